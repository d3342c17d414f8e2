#!/usr/bin/env python
"""Benchmark of the hot path (see DESIGN.md "Measurement").

python bench.py --gpus N --steps K --warmup W [--workload infer|ncc]

Default workload = BASELINE.json configs[1]: 3d_reg.py-style inference, one
VxmDense forward (enc/dec = 256, int_steps 5, half-res SVF) on a 160x160x192
pair, bf16 MFMA with fp32 accumulate, inputs resident in HBM.  A "step" is one
pair.  N > 1 = independent replicas (one process per GPU, weak scaling, no
data-path collective: single-pair inference does not shard).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0
DOMINANT_KERNEL = "conv3d_k3_kernel<bf16, WM=2, WN=4, MT=4, NT=2> (all 256-wide U-Net convs)"


def synth_pair(shape, device, seed=0):
    """Smooth synthetic T1w/T2w-like pair in [0,1] (random low-res noise, trilinearly upsampled on device)."""
    import mmr
    g = torch.Generator(device="cpu").manual_seed(seed)
    outs = []
    for k in range(2):
        lo = torch.rand((1, shape[0] // 8, shape[1] // 8, shape[2] // 8, 1), generator=g).to(device)
        v = mmr.ops.resize_trilinear(lo.contiguous(), shape)
        v = (v - v.min()) / (v.max() - v.min())
        outs.append(v.contiguous())
    return outs


def cpu_baseline_infer(enc, dec, full_shape):
    """Oracle (CPU restatement, 'port') on a bounded sample of the same workload: the same network
    on a 32x32x48 crop-sized pair, all host cores (OpenMP C conv + NumPy tail)."""
    from oracle import net_np
    from oracle import cbind
    import oracle.ops_np as O
    sample = (32, 32, 48)
    rng = np.random.default_rng(0)
    mov = rng.random((1,) + sample + (1,)).astype(np.float32)
    fix = rng.random((1,) + sample + (1,)).astype(np.float32)
    w = net_np.init_weights(enc, dec, seed=0)
    real = cbind.conv3d_same
    cbind_fast = lambda x, w_, b=None, leaky=False, alpha=0.2: real(x, w_, b, leaky=leaky, alpha=alpha, f32acc=True)
    net_np.conv3d_same = cbind_fast
    try:
        t0 = time.perf_counter()
        net_np.vxm_dense_forward(mov, fix, w, enc, dec, 5, 2, 2)
        dt = time.perf_counter() - t0
    finally:
        net_np.conv3d_same = real
    frac = float(np.prod(sample)) / float(np.prod(full_shape))
    return {"value": frac / dt, "unit": "pairs/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"one VxmDense forward (enc/dec=256) on a {sample[0]}x{sample[1]}x{sample[2]} pair = "
                      f"{frac:.5f} of the 160x160x192 voxels, {dt:.1f} s wall; value = that fraction / wall "
                      f"(pairs/s-equivalent, work is linear in voxels); oracle/conv_c.c f32-accumulate + NumPy tail"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="infer", choices=["infer"])
    ap.add_argument("--features", type=int, default=256)
    ap.add_argument("--shape", type=int, nargs=3, default=[160, 160, 192])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import mmr
    shape = tuple(args.shape)
    enc, dec = [args.features] * 4, [args.features] * 6
    model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2,
                                  svf_resolution=2, compute_dtype=args.dtype, device=dev, seed=0)
    mov, fix = synth_pair(shape, dev, seed=rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.forward(mov, fix)
    barrier()
    model.layer_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model.forward(mov, fix)
    barrier()
    dt = time.perf_counter() - t0
    events = model.layer_events
    model.layer_events = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out["y_source"]).all()

    if rank == 0:
        # dominant kernel = the BN=256 MFMA conv instantiation: every layer with Cout = features
        dom = [(n, e0.elapsed_time(e1), fl) for (n, e0, e1, fl) in events if n != "flow"]
        dom_ms = sum(d[1] for d in dom)
        dom_flops = sum(d[2] for d in dom)
        n_launch = len(dom)
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        per_layer = {}
        for n, ms, fl in [(n, e0.elapsed_time(e1), fl) for (n, e0, e1, fl) in events]:
            a = per_layer.setdefault(n, [0.0, 0.0])
            a[0] += ms
            a[1] += fl
        res = {
            "metric": "volume-pairs/sec", "value": world * args.steps / dt, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"3d_reg.py inference (BASELINE configs[1]): VxmDense forward "
                                   f"{shape[0]}x{shape[1]}x{shape[2]}, enc/dec={args.features}, int_steps=5, "
                                   f"svf/int_res=2, inputs resident in HBM, 1 pair/step",
                       "parallelism": f"replicas x{world} (single-pair inference does not shard)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3,
                         "unit": "TFLOP/s", "frac": achieved / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3),
                         "traffic": traffic, "kernel": DOMINANT_KERNEL,
                         "launches_per_step": n_launch // max(args.steps, 1),
                         "avg_launch_ms": dom_ms / max(n_launch, 1),
                         "algorithmic_tflop_per_step": dom_flops / max(args.steps, 1) / 1e12},
            "layer_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in per_layer.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_infer(enc, dec, shape)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
