"""Which way across PCIe is fastest on THIS box for predict()'s volumes (160x160x192: 39.3 MB of float64 in per volume,
19.7 + 7.4 MB of fp32 out)?  Times every strategy of mmr.hostio and the raw ingredients (pin / unpin, memcpy into the
different kinds of host memory, the zero-copy kernels), then the whole predict() in each mode.

    python tools/time_hostio.py [--no-predict]
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import mmr
from mmr import _lib, hostio

shape = (1, 160, 160, 192, 1)
rng = np.random.default_rng(0)
mov = rng.random(shape)
fix = rng.random(shape)
dev = torch.device("cuda", 0)
lib = _lib.load()


def t(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return f"{np.median(ts):7.2f} ms (min {min(ts):.2f}, max {max(ts):.2f})"


print(f"volume {shape}: {mov.nbytes / 1e6:.1f} MB float64; torch threads {torch.get_num_threads()}")
print("--- ingredients, one volume")


def reg_only():
    with hostio.Registered(mov) as r:
        assert r.ok
print("hipHostRegister + Unregister of the caller's 39 MB :", t(reg_only))
d32 = torch.empty(shape, dtype=torch.float32, device=dev)
with hostio.Registered(mov) as r:
    print("cast kernel reading the registered float64 pages   :", t(lambda: hostio._cast_launch(r.dev, d32, mov.size, hostio.F64)))
    ref = torch.from_numpy(mov).float()
    assert torch.equal(d32.cpu(), ref), "cast kernel result differs from the host conversion"
cached = hostio.Staging(mov.nbytes, cached=True)
coher = hostio.Staging(mov.nbytes, cached=False)
page = np.empty_like(mov)
print("memcpy float64 -> cached pinned (NonCoherent)        :", t(lambda: np.copyto(cached.view(np.float64, shape), mov)))
print("memcpy float64 -> coherent pinned                    :", t(lambda: np.copyto(coher.view(np.float64, shape), mov)))
print("memcpy float64 -> pageable                           :", t(lambda: np.copyto(page, mov)))
print("cast kernel reading cached pinned float64            :", t(lambda: hostio._cast_launch(cached.ptr, d32, mov.size, hostio.F64)))
print("cast kernel reading coherent pinned float64          :", t(lambda: hostio._cast_launch(coher.ptr, d32, mov.size, hostio.F64)))
c32 = hostio.Staging(mov.size * 4, cached=True)
print("host convert float64 -> fp32 into cached pinned      :", t(lambda: np.copyto(c32.view(np.float32, shape), mov, casting="unsafe")))
tp = torch.empty(shape, dtype=torch.float32).pin_memory()
tm = torch.from_numpy(mov)
print("host convert into torch pin_memory (round 4's path)  :", t(lambda: tp.copy_(tm)))
print("torch pinned fp32 -> device (copy engine)            :", t(lambda: tp.to(dev, non_blocking=True)))
print("pageable float64 -> device (.to) + device cast       :", t(lambda: tm.to(dev).float()))
print("--- whole strategies, one volume in")
for mode in ("register", "staging", "torch"):
    print(f"hostio.to_device_f32 mode={mode:9s}                :", t(lambda: hostio.to_device_f32(mov, dev, mode=mode)))
print("--- pair in (moving + fixed)")
for mode in ("register", "staging", "torch"):
    print(f"hostio.pair_to_device mode={mode:9s}               :", t(lambda: hostio.pair_to_device([mov, fix], dev, mode=mode)),
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in hostio.LAST["in"].items()})
print("--- out (moved 19.7 MB + half-res field 7.4 MB)")
ym = torch.rand(shape, device=dev)
yf = torch.rand((1, 80, 80, 96, 3), device=dev)
for mode in ("staging", "register", "torch"):
    print(f"hostio.many_to_host mode={mode:9s}                 :", t(lambda: hostio.many_to_host([ym, yf], mode=mode)),
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in hostio.LAST["out"].items()})
a, b = hostio.many_to_host([ym, yf], mode="staging")
assert np.array_equal(a, ym.cpu().numpy()) and np.array_equal(b, yf.cpu().numpy())
a, b = hostio.many_to_host([ym, yf], mode="register")
assert np.array_equal(a, ym.cpu().numpy()) and np.array_equal(b, yf.cpu().numpy())
del cached, coher, c32

if "--no-predict" not in sys.argv:
    print("--- predict() at C2 (bf16, 256 features)")
    m = mmr.networks.VxmDense(shape[1:4], nb_unet_features=([256] * 4, [256] * 6), int_steps=5, int_resolution=2, svf_resolution=2,
                              compute_dtype="bf16")
    a_, b_ = torch.from_numpy(mov).float().to(dev), torch.from_numpy(fix).float().to(dev)
    print("device-resident forward                              :", t(lambda: m.forward(a_, b_)))
    for mi, mo in (("register", "staging"), ("staging", "staging"), ("register", "register"), ("torch", "torch")):
        hostio.MODE_IN, hostio.MODE_OUT = mi, mo
        print(f"predict  in={mi:9s} out={mo:9s}               :", t(lambda: m.predict([mov, fix])))
    hostio.MODE_IN, hostio.MODE_OUT = "register", "staging"
    n = 6
    movb, fixb = np.repeat(mov, n, 0), np.repeat(fix, n, 0)
    m.predict([movb, fixb])
    t0 = time.perf_counter()
    m.predict([movb, fixb])
    print(f"predict on a batch of {n} pairs (copies overlapped)    : {(time.perf_counter() - t0) / n * 1e3:7.2f} ms/pair")
