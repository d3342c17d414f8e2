#!/usr/bin/env python
"""Diagnostic (-DMMR_DIAG) build: experimental instantiations of the fp32x3 64-column conv selected with mmr_debug_set_variant,
checked bit for bit against the default kernel and timed alternately.   python tools/exp_conv_variant.py [reps] [variants...]"""
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
variants = [int(v) for v in sys.argv[2:]] or [1, 2]
lib = mmr._lib.load()
dl = ctypes.CDLL(mmr._lib.lib_path())


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


CASES = (("64->64 @160^3", (160, 160, 160), 64, 0, False, 64), ("128->64 @160^3", (160, 160, 160), 128, 0, False, 64),
         ("up(64)+64->64 @160^3", (160, 160, 160), 64, 64, True, 64), ("64->64 @80^3", (80, 80, 80), 64, 0, False, 64),
         ("64->64 @ 9x13x21 (ragged, split K)", (9, 13, 21), 64, 0, False, 64), ("96+32->64 @ 20^3 up", (20, 20, 20), 96, 32, True, 64),
         ("64->64 @40^3", (40, 40, 40), 64, 0, False, 64))
for name, shape, C0, C1, up0, Cout in CASES:
    g = torch.Generator(device="cpu").manual_seed(1)
    s0 = tuple(s // 2 for s in shape) if up0 else shape
    x0 = torch.randn((1,) + s0 + (C0,), generator=g).to(dev)
    x1 = torch.randn((1,) + shape + (C1,), generator=g).to(dev) if C1 else None
    w = (torch.randn((3, 3, 3, C0 + C1, Cout), generator=g) * 0.03).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    wp = ops.pack_conv_weights(w, torch.float32, x3=True)
    run = lambda: ops.conv3d_k3(x0, wp, b, Cout, in1=x1, up0=up0, x3=True)
    dl.mmr_debug_set_variant(0)
    ref = run()
    res = {}
    for rnd in range(2):
        for v in [0] + variants:
            dl.mmr_debug_set_variant(v)
            if rnd == 0:
                y = run()
                res[v] = [torch.equal(y, ref), float((y - ref).abs().max() / ref.abs().max())]
            res[v].append(timed(run, reps))
    dl.mmr_debug_set_variant(0)
    print(name)
    for v, (same, err, t1, t2) in res.items():
        print(f"   variant {v:2d}: bitwise {same!s:5s} err {err:.1e}   {t1:.3f} / {t2:.3f} ms", flush=True)
