import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mmr
dev='cuda'
S=(256,256,256)
I=torch.rand((1,)+S+(1,),device=dev); J=torch.rand((1,)+S+(1,),device=dev); u=torch.randn((1,)+S+(3,),device=dev)
for name,fn in (("ncc_fwd",lambda: mmr.ops.ncc_loss(I,J,9)),("ncc_bwd",lambda: mmr.ops.ncc_loss_bwd(I,J)),("bend_fwd",lambda: mmr.ops.bending_energy(u)),("bend_bwd",lambda: mmr.ops.bending_energy_bwd(u))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter()-t)/10*1e3, "ms")
