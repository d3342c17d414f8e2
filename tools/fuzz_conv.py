"""Randomised shape sweep of the MFMA conv (fwd with upsample/concat loader, every dtype mode) against torch-CPU fp64
conv3d.  Not part of the test suite (minutes of CPU convs); run on a GPU box: python tools/fuzz_conv.py [ncases] [seed]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import mmr
ops = mmr.ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
worst = {}
for case in range(n):
    mode = ["bf16", "fp32x3", "fp32"][case % 3]
    kc = 64 if mode == "bf16" else 32
    up0 = bool(rng.integers(2))
    has1 = bool(rng.integers(2))
    shape = tuple(int(2 * rng.integers(1, 9)) if up0 else int(rng.integers(1, 19)) for _ in range(3))
    B = int(rng.integers(1, 3))
    C0 = kc * int(rng.integers(1, 4))
    C1 = kc * int(rng.integers(1, 3)) if has1 else 0
    Cout = int(rng.choice([32, 64, 96, 128, 192, 256]))
    leaky = bool(rng.integers(2))
    s0 = tuple(s // 2 for s in shape) if up0 else shape
    x0 = rng.standard_normal((B,) + s0 + (C0,)).astype(np.float32)
    x1 = rng.standard_normal((B,) + shape + (C1,)).astype(np.float32) if has1 else None
    w = (rng.standard_normal((3, 3, 3, C0 + C1, Cout)) * 0.05).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    t0 = torch.from_numpy(x0).to(dev).to(dt)
    t1 = torch.from_numpy(x1).to(dev).to(dt) if has1 else None
    tw = torch.from_numpy(w).to(dev)
    wp = ops.pack_conv_weights(tw, dt, x3=(mode == "fp32x3"))
    y = ops.conv3d_k3(t0, wp, torch.from_numpy(b).to(dev), Cout, in1=t1, up0=up0, leaky=leaky, out_f32=True,
                      x3=(mode == "fp32x3")).cpu().double()
    # reference in fp64 on the values the kernel sees
    q = (lambda a: torch.from_numpy(a).to(torch.bfloat16).double()) if mode == "bf16" else (lambda a: torch.from_numpy(a).double())
    r0 = q(x0)
    if up0:
        r0 = r0.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
    xin = torch.cat([r0, q(x1)], -1) if has1 else r0
    ref = F.conv3d(xin.permute(0, 4, 1, 2, 3), q(w).permute(4, 3, 0, 1, 2), torch.from_numpy(b).double(), padding=1).permute(0, 2, 3, 4, 1)
    if leaky:
        ref = torch.where(ref < 0, 0.2 * ref, ref)
    err = float((y - ref).abs().max() / ref.abs().max())
    tol = {"bf16": 2e-5, "fp32x3": 1e-4, "fp32": 2e-5}[mode]
    worst[mode] = max(worst.get(mode, 0.0), err)
    flag = "" if err < tol else "   <-- FAIL"
    print(f"{case:3d} {mode:6s} B={B} shape={shape} C0={C0} C1={C1} up0={int(up0)} Cout={Cout} leaky={int(leaky)} err={err:.2e}{flag}", flush=True)
    if flag:
        sys.exit(1)
print("worst", worst)
