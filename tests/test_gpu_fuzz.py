"""Randomised shape sweeps of the MFMA conv as fixed-seed ``-m gpu`` tests (tools/fuzz_conv.py and tools/fuzz_tail_split.py
run the same generators open-ended): every case is drawn from its seed alone, so a failure names a reproducible shape."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(9))
def test_fuzz_conv_vs_oracle(dev, seed):
    """Forward conv with the upsample / concat loader in every arithmetic mode against the C oracle (double accumulation) on
    the values the kernel sees: random batch, ragged / odd sizes, 1-3 channel slices per source, every N tile, bias, leaky."""
    import mmr
    from oracle.cbind import conv3d_same
    from oracle.net_np import bf16_round
    from oracle import ops_np as O
    ops = mmr.ops
    rng = np.random.default_rng(1000 + seed)
    mode = ["bf16", "fp32x3", "fp32"][seed % 3]
    kc = 64 if mode == "bf16" else 32
    up0 = bool(rng.integers(2))
    has1 = bool(rng.integers(2)) or up0 and seed % 2 == 0
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
    if mode == "bf16":
        x0, w = bf16_round(x0), bf16_round(w)
        x1 = bf16_round(x1) if has1 else None
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    x3 = mode == "fp32x3"
    t0 = torch.from_numpy(x0).to(dev).to(dt)
    t1 = torch.from_numpy(x1).to(dev).to(dt) if has1 else None
    wp = ops.pack_conv_weights(torch.from_numpy(w).to(dev), dt, x3=x3)
    y = ops.conv3d_k3(t0, wp, torch.from_numpy(b).to(dev), Cout, in1=t1, up0=up0, leaky=leaky, out_f32=True, x3=x3).cpu().numpy()
    full = O.upsample2(x0) if up0 else x0
    if has1:
        full = np.concatenate([full, x1], -1)
    ref = conv3d_same(full, w, b, leaky=leaky, alpha=0.2)
    err = np.abs(y - ref).max() / np.abs(ref).max()
    print(f"fuzz conv seed {seed}: {mode} B={B} shape={shape} C0={C0} C1={C1} up0={int(up0)} Cout={Cout} leaky={int(leaky)} err={err:.2e}")
    assert y.shape == ref.shape and err < {"bf16": 2e-5, "fp32x3": 1e-4, "fp32": 2e-5}[mode]


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_tail_split_vs_one_launch_and_oracle_window(dev, seed):
    """Launches that leave a partial round of workgroups (257 .. 700 tiles on 256 CUs): the tail-split form
    (mmr_conv3d_k3_fwd_ws with its work space) against the one-launch form and, on a far-corner window whose tiles belong to
    the split tail, against the C oracle."""
    import mmr
    from mmr import _lib
    from oracle.cbind import conv3d_same
    ops = mmr.ops
    rng = np.random.default_rng(2000 + seed)
    lib = _lib.load()
    mode = ["bf16", "fp32x3"][seed % 2]
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    x3 = mode == "fp32x3"
    cout = int(rng.choice([64, 128, 256]))
    cin = int(rng.choice([128, 192, 256])) if x3 else int(rng.choice([256, 320]))
    tx = 4 if cout == 256 else 8
    while True:
        shape = (int(rng.integers(8, 60)), int(rng.integers(8, 70)), int(rng.integers(8, 70)))
        tiles = -(-shape[0] // tx) * -(-shape[1] // 8) * -(-shape[2] // 8)
        if 257 <= tiles <= 700:
            break
    m = ops.conv_mode(dt, x3)
    ws = lib.mmr_conv3d_k3_ksplit_ws_bytes(1, *shape, cin, cout, m)
    x = torch.from_numpy(rng.standard_normal((1,) + shape + (cin,)).astype(np.float32)).to(dev).to(dt)
    wk = (rng.standard_normal((3, 3, 3, cin, cout)) * 0.05).astype(np.float32)
    w = torch.from_numpy(wk).to(dev)
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(dev)
    wp = ops.pack_conv_weights(w, dt, x3=x3)
    y = ops.conv3d_k3(x, wp, b, cout, leaky=True, x3=x3, out_f32=True)
    ref = torch.empty_like(y)
    rc = lib.mmr_conv3d_k3_fwd(x.data_ptr(), cin, 0, None, 0, wp.data_ptr(), b.data_ptr(), ref.data_ptr(), None, 1, *shape, cout,
                               1, 0.2, m, 1, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    err = float((y - ref).abs().max()) / float(ref.abs().max())
    n = tuple(min(s, 10) for s in shape)
    crop = x[0, -n[0]:, -n[1]:, -n[2]:].float().cpu().numpy()[None]
    wq = w.to(dt).float().cpu().numpy() if not x3 else wk
    o = conv3d_same(crop, wq, b.cpu().numpy(), leaky=True, alpha=0.2)[0]
    keep = tuple(slice(0 if k == s else 1, None) for k, s in zip(n, shape))
    got = y[0, -n[0]:, -n[1]:, -n[2]:].cpu().numpy()[keep]
    oerr = np.abs(got - o[keep]).max() / np.abs(o[keep]).max()
    print(f"fuzz tail split seed {seed}: {mode} shape={shape} tiles={tiles} cin={cin} cout={cout} ws={ws / 1e6:.1f} MB "
          f"vs one launch {err:.2e}, window vs oracle {oerr:.2e}")
    assert err <= 1e-5 and oerr < (1e-4 if x3 else 2e-5)
