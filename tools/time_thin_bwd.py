"""Times the thin-layer backward kernels of the C3 training step (160^3, 64 features): flow-head wgrad / dgrad and
first-layer wgrad, exact-fp32 and fp32x3 paths.  Usage: python tools/time_thin_bwd.py [S]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmr  # noqa: E402
from mmr import ops  # noqa: E402


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((1, S, S, S, 64), device=dev, generator=g)
    dflow = torch.randn((1, S, S, S, 3), device=dev, generator=g)
    src = torch.rand((1, S, S, S, 1), device=dev, generator=g)
    trg = torch.rand((1, S, S, S, 1), device=dev, generator=g)
    w = torch.randn((3, 3, 3, 64, 3), device=dev, generator=g) * 0.05
    dwf = torch.zeros((3, 3, 3, 64, 3), device=dev)
    dw0 = torch.zeros((3, 3, 3, 2, 64), device=dev)
    gb = x.numel() * 4 / 1e9
    for x3 in (False, True):
        t = timeit(lambda: ops.conv3d_k3_wgrad(x, dflow, dwf, x3=x3))
        print(f"flow-head wgrad  x3={x3}: {t:.3f} ms  ({gb / t * 1e3:.0f} GB/s of the dense operand)")
        t = timeit(lambda: ops.conv3d_k3_cin2_wgrad(src, trg, x, dw0, x3=x3))
        print(f"first-layer wgrad x3={x3}: {t:.3f} ms  ({gb / t * 1e3:.0f} GB/s)")
    db = torch.zeros(64, device=dev)
    for x3 in (False, True):
        t = timeit(lambda: ops.conv3d_k3_cout3_dgrad(dflow, w, x3=x3))
        print(f"flow-head dgrad x3={x3}: {t:.3f} ms  ({gb / t * 1e3:.0f} GB/s written)")
        t = timeit(lambda: ops.conv3d_k3_cout3_dgrad_masked(dflow, w, x, db, x3=x3))
        print(f"flow-head dgrad masked x3={x3}: {t:.3f} ms  ({2 * gb / t * 1e3:.0f} GB/s read + written)")
    b = torch.zeros(3, device=dev)
    t = timeit(lambda: ops.conv3d_k3_cout3(x, w, b, x3=True))
    print(f"flow head fwd x3: {t:.3f} ms  ({gb / t * 1e3:.0f} GB/s read)")
    xb = torch.randn((1, 160, 160, 192, 256), device=dev, generator=g).to(torch.bfloat16)
    wb = torch.randn((3, 3, 3, 256, 3), device=dev, generator=g) * 0.05
    t = timeit(lambda: ops.conv3d_k3_cout3(xb, wb, b))
    print(f"flow head fwd bf16 C2: {t:.3f} ms  ({xb.numel() * 2 / 1e9 / t * 1e3:.0f} GB/s read)")


def compose_bwd_times():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    for sc in (0.01, 0.5, 2.0):
        v = torch.randn((1, 80, 80, 80, 3), device=dev, generator=g) * sc
        go = torch.randn((1, 80, 80, 80, 3), device=dev, generator=g)
        out, steps = ops.vecint_save(v, 5)
        t = timeit(lambda: ops.vecint_bwd(v, steps, go, 5))
        print(f"vecint_bwd 80^3, 5 steps, velocity std {sc}: {t:.3f} ms")


if __name__ == "__main__":
    compose_bwd_times()
    main()
