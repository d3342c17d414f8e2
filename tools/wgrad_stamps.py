#!/usr/bin/env python
"""Where the cycles of the fp32x3 wgrad kernel go (diagnostic -DMMR_DIAG build): the C3 dec_final_0 layer
(concat 128 -> 64 at 160^3) and dec_final_1 (64 -> 64); prints per-wave shares of the in-kernel stamps per voxel tile."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
_spec = importlib.util.spec_from_file_location("mmr_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multimodal-registration_amd", "build.py"))
_b = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_b)
os.environ["MMR_LIB"] = _b.DIAG_LIB if os.path.exists(_b.DIAG_LIB) else _b.build_diag()   # -DMMR_DIAG twin of the library
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mmr

dev = torch.device("cuda", 0)
shape, Cin, Cout = (160, 160, 160), 64, 64
x = torch.randn((1,) + shape + (Cin,), device=dev) * 0.5
dz = torch.randn((1,) + shape + (Cout,), device=dev) * 0.1
dw = torch.zeros((3, 3, 3, Cin, Cout), device=dev)
_dl = ctypes.CDLL(mmr._lib.lib_path())
_dl.mmr_debug_set_stamps(1)
fn = _dl.mmr_debug_wgrad_stamps
buf = (ctypes.c_ulonglong * 64)()
for _ in range(2):
    mmr.ops.conv3d_k3_wgrad(x, dz, dw, x3=True)
torch.cuda.synchronize()
fn(buf)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(3):
    mmr.ops.conv3d_k3_wgrad(x, dz, dw, x3=True)
t1.record()
torch.cuda.synchronize()
fn(buf)
a = np.array(list(buf), dtype=np.float64).reshape(8, 8)
names = ["barrier0", "split+store+barrier", "issue next loads", "k-loop (16 k-blocks)"]
ms = t0.elapsed_time(t1) / 3
fl = 2.0 * 27 * Cin * Cout * np.prod(shape)
print(f"ms per launch: {ms:.3f}  ({fl / ms / 1e9:.0f} TFLOP/s-equivalent; x3 ceiling 833)")
for wv in range(8):
    n = a[wv, 4]
    print(f"wave {wv}: cycles/tile " + ", ".join(f"{nm} {a[wv, i] / n:8.1f}" for i, nm in enumerate(names)) +
          f" | total {a[wv, :4].sum() / n:8.1f}  (ideal MFMA time per tile 21504)")
