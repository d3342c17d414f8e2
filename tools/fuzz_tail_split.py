#!/usr/bin/env python
"""Randomised check of the tail-split launch form against the one-launch form (mmr_conv3d_k3_fwd_ws with / without a work
space), bf16 and fp32x3, shapes with 257 .. 700 tiles: python tools/fuzz_tail_split.py [ncases] [seed]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmr
from mmr import _lib
ops = mmr.ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
lib = _lib.load()
worst = 0.0
hit = 0
for case in range(n):
    mode = ["bf16", "fp32x3"][case % 2]
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    x3 = mode == "fp32x3"
    cout = int(rng.choice([64, 128, 256]))
    cin = int(rng.choice([128, 192, 256])) if mode == "fp32x3" else int(rng.choice([256, 320]))
    tx = 4 if cout == 256 else 8
    while True:
        shape = (int(rng.integers(8, 60)), int(rng.integers(8, 70)), int(rng.integers(8, 70)))
        tiles = -(-shape[0] // tx) * -(-shape[1] // 8) * -(-shape[2] // 8)
        if 257 <= tiles <= 700:
            break
    B = 1
    m = ops.conv_mode(dt, x3)
    ws = lib.mmr_conv3d_k3_ksplit_ws_bytes(B, *shape, cin, cout, m)
    x = torch.from_numpy(rng.standard_normal((B,) + shape + (cin,)).astype(np.float32)).to(dev).to(dt)
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, cin, cout)) * 0.05).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(dev)
    wp = ops.pack_conv_weights(w, dt, x3=x3)
    y = ops.conv3d_k3(x, wp, b, cout, leaky=True, x3=x3)
    ref = torch.empty_like(y)
    rc = lib.mmr_conv3d_k3_fwd(x.data_ptr(), cin, 0, None, 0, wp.data_ptr(), b.data_ptr(), ref.data_ptr(), None, B, *shape, cout,
                               1, 0.2, m, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    err = float((y.float() - ref.float()).abs().max()) / float(ref.float().abs().max())
    tol = 1e-2 if mode == "bf16" else 1e-5
    hit += ws > 0
    worst = max(worst, err / tol)
    print(f"{case:3d} {mode:6s} shape={shape} tiles={tiles} cin={cin} cout={cout} ws={ws / 1e6:.1f}MB err={err:.2e}", flush=True)
    assert err <= tol, "MISMATCH"
print("cases with a tail split:", hit, "of", n, " worst err / tol:", worst)
