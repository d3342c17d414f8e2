#!/usr/bin/env python
"""The HBM-bound kernels of a C3 step's tail at their real sizes (160^3 field, 26 labels): label-map Dice forward / backward,
Grad-l2 forward / backward, the two resize adjoints.  HIP events, 20 launches each; MMR_LIB=<other .so> for an A/B on one box.
  python tools/time_tail_bwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmr
from mmr import ops
dev = torch.device("cuda", 0)
S, L = (160, 160, 160), 26
g = torch.Generator(device="cpu").manual_seed(0)
pos = (torch.nn.functional.interpolate(torch.randn((1, 3, 10, 10, 10), generator=g) * 2, size=S, mode="trilinear")
       .permute(0, 2, 3, 4, 1).contiguous().to(dev))
rng = np.random.default_rng(0)
mk = lambda: torch.from_numpy(np.repeat(np.repeat(np.repeat(rng.integers(0, L, (1, 20, 20, 20)), 8, 1), 8, 2), 8, 3).astype(np.uint8)[..., None]).to(dev)
lab1, lab2 = mk(), mk()
half = tuple(s // 2 for s in S)
dlo = torch.randn((1,) + half + (3,), generator=g).to(dev)
dhi = torch.randn((1,) + S + (3,), generator=g).to(dev)
def ev_time(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3
dice, tb = ops.dice_labels_fwd(lab1, lab2, pos, L)
dpos = ops.dice_labels_bwd(lab1, lab2, pos, tb, L, scale=1.0)
rows = [("dice_labels_fwd", lambda: ops.dice_labels_fwd(lab1, lab2, pos, L)),
        ("dice_labels_bwd", lambda: ops.dice_labels_bwd(lab1, lab2, pos, tb, L, scale=1.0)),
        ("grad_l2_loss", lambda: ops.grad_l2_loss(pos, 1.0)),
        ("grad_l2_bwd (accumulating)", lambda: ops.grad_l2_bwd(pos, 1.0, 1.0, out=dpos)),
        ("resize_trilinear_bwd 160^3 -> 80^3 (adjoint of the x2 resize)", lambda: ops.resize_trilinear_bwd(dhi, half, mul=2.0)),
        ("resize_trilinear_bwd 80^3 -> 160^3 (adjoint of the /2 resize)", lambda: ops.resize_trilinear_bwd(dlo, S, mul=0.5))]
chk = [float(dice), float(dpos.double().abs().sum()), float(ops.grad_l2_loss(pos, 1.0)[0]),
       float(ops.resize_trilinear_bwd(dhi, half, mul=2.0).double().abs().sum()), float(ops.resize_trilinear_bwd(dlo, S, mul=0.5).double().abs().sum())]
print("checksums:", " ".join(f"{c:.10e}" for c in chk))
for name, fn in rows:
    print(f"{name:66s} {ev_time(fn):7.1f} us", flush=True)
