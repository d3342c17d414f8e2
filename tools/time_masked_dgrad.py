#!/usr/bin/env python
"""Masked data gradient (LeakyReLU backward + bias sums fused into the conv epilogue) against the same conv without the
mask, fp32x3, 64 -> 64 at 160^3 and 80^3: what the fused epilogue costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmr
dev = torch.device("cuda", 0)
def ev_time(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n
for shape in ((160, 160, 160), (80, 80, 80)):
    C = 64
    dz = torch.randn((1,) + shape + (C,), device=dev)
    y = torch.randn((1,) + shape + (C,), device=dev)
    w = torch.randn((3, 3, 3, C, C), device=dev) * 0.05
    wt = mmr.ops.pack_conv_weights(w, torch.float32, transpose_flip=True, x3=True)
    wp = mmr.ops.pack_conv_weights(w, torch.float32, x3=True)
    db = torch.zeros(C, device=dev)
    a, b = [], []
    for _ in range(4):
        a.append(ev_time(lambda: mmr.ops.conv3d_k3(dz, wp, None, C, leaky=False, x3=True)))
        b.append(ev_time(lambda: mmr.ops.conv3d_k3_dgrad_masked(dz, wt, C, y, db, x3=True)))
    print(f"{shape} 64->64 fp32x3: plain conv {np.median(a):.4f} ms, masked dgrad {np.median(b):.4f} ms", flush=True)
