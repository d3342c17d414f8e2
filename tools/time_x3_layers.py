#!/usr/bin/env python
"""fp32x3 64-column convs of the C3 step at their sizes (forward 64 -> 64, masked dgrad, 128 -> 64) -- ms per launch and the
fraction of the x3 ceiling (2.5 PFLOP/s / 3).   python tools/time_x3_layers.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
if "--ab" in sys.argv:      # diagnostic build: alternate the occupancy kernel (x3s) with the 8-wave kernel on the same box
    import importlib.util
    _root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _spec = importlib.util.spec_from_file_location("mmr_build", os.path.join(_root, "multimodal-registration_amd", "build.py"))
    _b = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_b)
    os.environ["MMR_LIB"] = _b.DIAG_LIB
import torch, mmr
ops = mmr.ops
dl = ctypes.CDLL(mmr._lib.lib_path()) if "--ab" in sys.argv else None
dev = torch.device("cuda", 0)


def timed(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, shape, Cin in (("64->64 @160^3", (160, 160, 160), 64), ("128->64 @160^3", (160, 160, 160), 128), ("64->64 @80^3", (80, 80, 80), 64),
                         ("64->64 @96x96x128", (96, 96, 128), 64)):
    x = torch.randn((1,) + shape + (Cin,), device=dev)
    w = torch.randn((3, 3, 3, Cin, 64), device=dev) * 0.03
    b = torch.randn(64, device=dev)
    wp = ops.pack_conv_weights(w, torch.float32, x3=True)
    fl = 2.0 * 27 * Cin * 64 * shape[0] * shape[1] * shape[2]
    for rnd in range(2 if dl else 1):
        for on in ((1, 2, 3, 4, 0) if dl else (1,)):
            if dl:
                dl.mmr_debug_x3s(on)
            t = timed(lambda: ops.conv3d_k3(x, wp, b, 64, x3=True))
            print(f"{name} [{('8-wave', 'x3s 4x8x8 3 WG/CU', 'x3s 4x8x8 2 WG/CU dbuf', 'x3s 8x8x8 2 WG/CU dbuf', 'x3s 8x8x8 single buf')[on]}]: {t:.3f} ms  = {fl / t / 1e9:.0f} TFLOP/s algorithmic = {fl / t / 1e9 / (2500 / 3):.3f} of the x3 ceiling", flush=True)
    if dl:
        dl.mmr_debug_x3s(1)
    if Cin == 64:
        y = torch.randn((1,) + shape + (64,), device=dev)
        db = torch.zeros(64, device=dev)
        wt = ops.pack_conv_weights(w, torch.float32, transpose_flip=True, x3=True)
        t = timed(lambda: ops.conv3d_k3_dgrad_masked(x, wt, 64, y, db, x3=True))
        print(f"   masked dgrad: {t:.3f} ms")
