#!/usr/bin/env python
"""Per-layer times of the C2 (160x160x192, 256 features, bf16) convs, one process, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmr
dev = torch.device("cuda", 0)
def ev_time(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n
dt = torch.bfloat16
for shape, cin in [((160, 160, 192), 256), ((160, 160, 192), 512), ((80, 80, 96), 256), ((80, 80, 96), 512)]:
    x = (torch.randn((1,) + shape + (cin,), device=dev) * 0.5).to(dt)
    w = torch.randn((3, 3, 3, cin, 256), device=dev) * 0.02
    b = torch.zeros(256, device=dev)
    wp = mmr.ops.pack_conv_weights(w, dt)
    ms = np.median([ev_time(lambda: mmr.ops.conv3d_k3(x, wp, b, 256)) for _ in range(3)])
    fl = 2 * 27 * cin * 256 * np.prod(shape)
    print(f"plain {cin}->256 at {shape}: {ms:.3f} ms = {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
    del x, w, wp
    torch.cuda.empty_cache()
