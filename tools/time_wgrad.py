import os, sys
sys.path.insert(0, ".")
import torch, mmr
dev = torch.device("cuda", 0)
for (cin, cout, up) in ((64, 64, False), (128, 64, True)):
    if up:
        x0 = torch.randn((1, 80, 80, 80, 64), device=dev) * 0.5
        x1 = torch.randn((1, 160, 160, 160, 64), device=dev) * 0.5
    else:
        x0 = torch.randn((1, 160, 160, 160, cin), device=dev) * 0.5
        x1 = None
    dz = torch.randn((1, 160, 160, 160, cout), device=dev) * 0.1
    dw = torch.zeros((3, 3, 3, cin, cout), device=dev)
    for _ in range(2):
        mmr.ops.conv3d_k3_wgrad(x0, dz, dw, in1=x1, up0=up, x3=True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
    for _ in range(5):
        mmr.ops.conv3d_k3_wgrad(x0, dz, dw, in1=x1, up0=up, x3=True)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 5
    print(f"PF={os.environ.get('MMR_WGRAD_PF', '3')} cin {cin} cout {cout}: {ms:.3f} ms = {2*27*cin*cout*160**3/ms/1e9:.0f} TF-equiv")
