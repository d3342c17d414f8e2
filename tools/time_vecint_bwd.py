#!/usr/bin/env python
"""VecInt backward (5 scaling-and-squaring steps) at the half resolution of a C3 step (80^3), on a field of realistic size:
  python tools/time_vecint_bwd.py      (MMR_LIB=<other libmmr_hip.so> for an A/B on one box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr
from mmr import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
vel = torch.nn.functional.interpolate(torch.randn((1, 3, 10, 10, 10), generator=g) * 1.5, size=(80, 80, 80), mode="trilinear")
vel = vel.permute(0, 2, 3, 4, 1).contiguous().to(dev)
out, steps = ops.vecint_save(vel, 5)
dout = torch.randn(out.shape, generator=g).to(dev)
def ev_time(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n
d = ops.vecint_bwd(vel, steps, dout, 5)
print(f"max |displacement| {float(out.abs().max()):.2f} voxels; checksum of d vel {float(d.double().abs().sum()):.9e}")
for r in range(3):
    print(f"vecint_bwd(5 steps, 80^3): {ev_time(lambda: ops.vecint_bwd(vel, steps, dout, 5)) * 1e3:.1f} us", flush=True)
