#!/usr/bin/env python
"""Layers whose tile count leaves a partial round of workgroups: one launch (mmr_conv3d_k3_fwd) vs the tail-split form
(mmr_conv3d_k3_fwd_ws with the work space the query asks for), interleaved in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmr
from mmr import _lib
dev = torch.device("cuda", 0)
lib = _lib.load()
def ev_time(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n
for dt, x3, shape, cin, cout in [(torch.bfloat16, False, (80, 80, 96), 256, 256), (torch.bfloat16, False, (40, 40, 48), 256, 256),
                                 (torch.float32, True, (160, 160, 160), 64, 64), (torch.float32, True, (80, 80, 80), 64, 64)]:
    x = (torch.randn((1,) + shape + (cin,), device=dev) * 0.5).to(dt)
    w = torch.randn((3, 3, 3, cin, cout), device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    wp = mmr.ops.pack_conv_weights(w, dt, x3=x3)
    m = mmr.ops.conv_mode(dt, x3)
    out = torch.empty((1,) + shape + (cout,), dtype=dt, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    one = lambda: lib.mmr_conv3d_k3_fwd(x.data_ptr(), cin, 0, None, 0, wp.data_ptr(), b.data_ptr(), out.data_ptr(), None, 1, *shape, cout, 1, 0.2, m, 0, st)
    a, s = [], []
    for _ in range(4):
        a.append(ev_time(one)); s.append(ev_time(lambda: mmr.ops.conv3d_k3(x, wp, b, cout, x3=x3)))
    print(f"{shape} {cin}->{cout} {'fp32x3' if x3 else 'bf16'}: one launch {np.median(a):.4f} ms, tail split {np.median(s):.4f} ms "
          f"(ws {lib.mmr_conv3d_k3_ksplit_ws_bytes(1, *shape, cin, cout, m) / 1e6:.1f} MB)", flush=True)
