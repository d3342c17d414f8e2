#!/usr/bin/env python
"""Experimental instantiations of the fp32x3 64-column conv (diagnostic -DMMR_DIAG build; mmr_debug_set_variant):
  0 default | 1 BREG + prio | 2 BREG | 4 PRESPLIT (input stored [32 hi | 32 lo] per group, A tile by LDS-DMA) | 8 PRESPLIT + BREG + prio
  | 16 PRESPLIT + BREG
Each variant is checked bit for bit against the default kernel on the same layer, then timed (alternated).
  python tools/exp_x3_variants.py [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("mmr_build", os.path.join(_root, "multimodal-registration_amd", "build.py"))
_b = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_b)
os.environ["MMR_LIB"] = _b.DIAG_LIB if os.path.exists(_b.DIAG_LIB) else _b.build_diag()
import torch
import mmr
ops = mmr.ops
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = mmr._lib.load()
dl = ctypes.CDLL(mmr._lib.lib_path())


def presplit(x):
    """fp32 [..., C] -> the same bytes per 32-channel group as [32 hi bf16 | 32 lo bf16], viewed as fp32 [..., C]."""
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    G = x.shape[-1] // 32
    t = torch.stack([hi.view(*x.shape[:-1], G, 32), lo.view(*x.shape[:-1], G, 32)], dim=-2).contiguous()
    return t.view(torch.float32).view(x.shape)


def timed(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, shape, C0, C1, up0 in (("64->64 @160^3", (160, 160, 160), 64, 0, False), ("128->64 @160^3", (160, 160, 160), 128, 0, False),
                                 ("up(64)+64->64 @160^3", (160, 160, 160), 64, 64, True), ("64->64 @80^3", (80, 80, 80), 64, 0, False)):
    g = torch.Generator(device="cpu").manual_seed(1)
    s0 = tuple(s // 2 for s in shape) if up0 else shape
    x0 = torch.randn((1,) + s0 + (C0,), generator=g).to(dev)
    x1 = torch.randn((1,) + shape + (C1,), generator=g).to(dev) if C1 else None
    w = (torch.randn((3, 3, 3, C0 + C1, 64), generator=g) * 0.03).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    wp = ops.pack_conv_weights(w, torch.float32, x3=True)
    p0, p1 = presplit(x0), (presplit(x1) if C1 else None)
    run = lambda a0, a1: ops.conv3d_k3(a0, wp, b, 64, in1=a1, up0=up0, x3=True)
    dl.mmr_debug_set_variant(0)
    ref = run(x0, x1)
    res = {}
    for rnd in range(2):
        for v in (0, 1, 2, 4, 8, 16):
            dl.mmr_debug_set_variant(v)
            a0, a1 = (p0, p1) if v & (4 | 8 | 16) else (x0, x1)
            if rnd == 0:
                y = run(a0, a1)
                same = torch.equal(y, ref)
                err = float((y - ref).abs().max() / ref.abs().max())
                res[v] = [same, err]
            res[v].append(timed(lambda: run(a0, a1), reps))
    dl.mmr_debug_set_variant(0)
    print(name)
    for v, (same, err, t1, t2) in res.items():
        print(f"   variant {v:2d}: bitwise {same!s:5s} err {err:.1e}   {t1:.3f} / {t2:.3f} ms", flush=True)
