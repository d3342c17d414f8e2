#!/usr/bin/env python
"""A/B in ONE process: a decoder layer conv(concat([up2(x), skip])) as one 27-tap launch vs the folded-upsampling pair, and the
whole C2 forward with fold_upsampling on / off (interleaved rounds, HIP events)."""
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

for (shape, C, dt, x3) in [((160, 160, 192), 256, torch.bfloat16, False), ((80, 80, 96), 256, torch.bfloat16, False),
                           ((160, 160, 160), 64, torch.float32, True)]:
    lo = tuple(s // 2 for s in shape)
    x = (torch.randn((1,) + lo + (C,), device=dev) * 0.5).to(dt)
    sk = (torch.randn((1,) + shape + (C,), device=dev) * 0.5).to(dt)
    w = torch.randn((3, 3, 3, 2 * C, C), device=dev) * 0.02
    b = torch.zeros(C, device=dev)
    wp = mmr.ops.pack_conv_weights(w, dt, x3=x3)
    wu, ws = mmr.ops.pack_upfold_weights(w, C, dt, x3=x3)
    plain = lambda: mmr.ops.conv3d_k3(x, wp, b, C, in1=sk, up0=True, x3=x3)
    fold = lambda: mmr.ops.conv3d_k3_upfold(x, sk, wu, ws, b, C, x3=x3)
    res = {"plain": [], "fold": []}
    for r in range(3):
        res["plain"].append(ev_time(plain)); res["fold"].append(ev_time(fold))
    mmr.ops.PROFILE = []
    fold(); torch.cuda.synchronize()
    parts = {f: e0.elapsed_time(e1) for f, tag, e0, e1, fl in mmr.ops.PROFILE}
    mmr.ops.PROFILE = None
    print(f"{shape} C={C} {dt}: plain {np.median(res['plain']):.3f} ms, folded pair {np.median(res['fold']):.3f} ms {parts}", flush=True)
    del x, sk, w, wp, wu, ws
    torch.cuda.empty_cache()

shape = (160, 160, 192)
for dtype in ("bf16", "fp32x3"):
    ms = {}
    models = {f: mmr.networks.VxmDense(shape, nb_unet_features=([256] * 4, [256] * 6), int_steps=5, int_resolution=2, svf_resolution=2,
                                       compute_dtype=dtype, device=dev, seed=0, fold_upsampling=f) for f in (False, True)}
    g = torch.Generator(device="cpu").manual_seed(0)
    mov = torch.rand((1,) + shape + (1,), generator=g).to(dev); fix = torch.rand((1,) + shape + (1,), generator=g).to(dev)
    for r in range(3):
        for f, m in models.items():
            ms.setdefault(f, []).append(ev_time(lambda: m.forward(mov, fix)["y_source"], 3))
    a = models[False].forward(mov, fix); bb = models[True].forward(mov, fix)
    d = float((a["pos_flow"] - bb["pos_flow"]).abs().max() / a["pos_flow"].abs().max())
    print(f"C2 forward {dtype}: one-launch {np.median(ms[False]):.2f} ms, folded {np.median(ms[True]):.2f} ms; pos_flow difference {d:.2e} of scale", flush=True)
    del models
    torch.cuda.empty_cache()
