#!/usr/bin/env python
"""Time the two HBM-bound layers of the C2 forward (first layer with fused pooling, flow head), 160x160x192 x 256, bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mmr
dev = torch.device("cuda", 0)
shape, C = (160, 160, 192), 256
src = torch.rand((1,) + shape + (1,), device=dev); trg = torch.rand((1,) + shape + (1,), device=dev)
w0 = torch.randn((3, 3, 3, 2, C), device=dev) * 0.2; b0 = torch.zeros(C, device=dev)
x = (torch.randn((1,) + shape + (C,), device=dev) * 0.5).to(torch.bfloat16)
wf = torch.randn((3, 3, 3, C, 3), device=dev) * 0.02; bf = torch.zeros(3, device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize(); return t0.elapsed_time(t1) / n
nv = shape[0] * shape[1] * shape[2]
t = timeit(lambda: mmr.ops.conv3d_k3_cin2(src, trg, w0, b0, torch.bfloat16, pool=True))
print(f"first layer + fused pool: {t:.3f} ms  ({(nv * C * 2 * 1.125 + nv * 8) / t / 1e6:.0f} GB/s of output+input)")
t = timeit(lambda: mmr.ops.conv3d_k3_cout3(x, wf, bf))
print(f"flow head: {t:.3f} ms  ({(nv * C * 2 + nv * 12) / t / 1e6:.0f} GB/s algorithmic)")
