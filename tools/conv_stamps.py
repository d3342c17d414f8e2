#!/usr/bin/env python
"""Where the cycles of a tap of the BN=256 bf16 conv go (diagnostic -DMMR_DIAG build, python -m ... build.py --diag): runs the C2
dec_final_1 layer (256 -> 256, 160x160x192) and prints the per-wave shares of the in-kernel stamps."""
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
# default: C2 dec_final_1 (bf16, 256 -> 256 at 160x160x192); "train": C3 dec_final_1 (fp32x3, 64 -> 64 at 160^3)
train = len(sys.argv) > 1 and sys.argv[1] == "train"
shape, C = ((160, 160, 160), 64) if train else ((160, 160, 192), 256)
dt = torch.float32 if train else torch.bfloat16
x = (torch.randn((1,) + shape + (C,), device=dev) * 0.5).to(dt)
w = torch.randn((3, 3, 3, C, C), device=dev) * 0.02
b = torch.zeros(C, device=dev)
wp = mmr.ops.pack_conv_weights(w, dt, x3=train)
_conv = mmr.ops.conv3d_k3
mmr.ops.conv3d_k3 = lambda x, wp, b, C: _conv(x, wp, b, C, out_f32=train, x3=train)
lib = mmr._lib.load()
_dl = ctypes.CDLL(mmr._lib.lib_path())
_dl.mmr_debug_set_stamps(1)
fn = _dl.mmr_debug_conv_stamps
buf = (ctypes.c_ulonglong * 64)()
for _ in range(2):
    y = mmr.ops.conv3d_k3(x, wp, b, C)
torch.cuda.synchronize()
fn(buf)   # clear
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(3):
    y = mmr.ops.conv3d_k3(x, wp, b, C)
t1.record()
torch.cuda.synchronize()
fn(buf)
a = np.array(list(buf), dtype=np.float64).reshape(8, 8)
names = ["dma_issue", "reads+mfma", "vmcnt(0)", "barrier", "A restage", "taps"]
print("ms per launch (stamped build):", t0.elapsed_time(t1) / 3)
for wv in range(8):
    taps = a[wv, 5]
    tot = a[wv, :5].sum()
    print(f"wave {wv}: cycles/tap " + ", ".join(f"{n} {a[wv, i] / taps:7.1f}" for i, n in enumerate(names[:5])) +
          f" | total {tot / taps:7.1f}")
