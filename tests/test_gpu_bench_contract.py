"""bench.py's output contract on a tiny configuration: ONE JSON line on stdout with the driver's keys, the roofline
and cpu_baseline objects, and the secondary (training) leg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config"}


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_infer_line_small():
    d = _run("--shape", "32", "32", "48", "--features", "64", "--steps", "2", "--warmup", "1", "--no-secondary")
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert 0 < d["ms_per_step_without_events"] < 2 * d["ms_per_step"]    # the same K steps without the per-launch events
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["dist"]["world_size"] == 1 and d["dist"]["backend"] is None


def test_default_line_carries_both_halves_of_the_metric():
    """The line the driver records: primary = inference, secondary = the training step with the SAME steps / warm-up,
    each with its own roofline and cpu_baseline."""
    d = _run("--shape", "32", "32", "32", "--features", "32", "--steps", "2", "--warmup", "1")
    sec = d["secondary"]
    assert "error" not in sec, sec
    assert sec["steps"] == d["steps"] == 2 and sec["warmup"] == d["warmup"] == 1 and sec["dtype"] == "fp32x3"
    for leg in (d, sec):
        assert leg["roofline"]["frac"] > 0 and leg["cpu_baseline"]["value"] > 0 and leg["cpu_baseline"]["kind"] == "port"
    assert "whole SynthMorph step" in sec["cpu_baseline"]["sample"]
    # the exact-fp32 step (the reference's arithmetic, train_synthmorph.py:308) is driver-timed beside the fp32x3 one, the
    # data-parallel leg is lifted to the top level, and the line says where every rank sits
    assert sec["same_workload_fp32"]["dtype"] == "fp32" and sec["same_workload_fp32"]["ms_per_step"] > 0
    dp = d["dp_training"]
    assert dp["value"] == sec["value"] and dp["n_gpus"] == 1 and dp["ms_per_step"] == sec["ms_per_step"]
    assert d["dist"]["visible_devices"] >= 1 and d["dist"]["rank_devices"]
    assert dp["early_reduce"] is True and dp["ms_per_step_single_bucket"] is None     # the A/B leg needs a process group
    # north_star's latency: predict() with float64 NumPy volumes in / NumPy out, with its PCIe / host split;
    # and BASELINE configs[3], the two-step cascade -- both driver-timed, neither part of `value`
    pr = d["predict"]
    assert "error" not in pr, pr
    assert pr["calls"] == 5 and pr["warmup"] == 2 and pr["ms_per_pair"] > 0 and len(pr["runs_ms"]) == 5
    for k in ("h2d_ms", "d2h_ms", "host_convert_ms", "host_pin_ms", "forward_ms_device_resident", "overhead_ms"):
        assert isinstance(pr[k], float), k
    assert pr["mode_in"] in ("register", "staging") and pr["mode_out"] in ("register", "staging")
    ca = d["cascade"]
    assert "error" not in ca, ca
    assert ca["steps"] == 3 and ca["ms_per_pair"] > d["ms_per_step_without_events"] and "cascade" in ca["workload"]


def test_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` without torchrun: the launcher starts two fresh rank processes (before touching the GPU)
    and the line reports the group they formed.  On this one-GPU box the ranks share the card and exchange gradients over
    gloo (MMR_BENCH_BACKEND); the driver's 8-GPU run takes the same path with nccl = RCCL."""
    d = _run("--gpus", "2", "--workload", "train", "--shape", "32", "32", "32", "--features", "32", "--steps", "2", "--warmup", "1",
             env={"MMR_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["dist"] == {**d["dist"], "world_size": 2, "backend": "gloo", "launched_by": "self"}
    assert d["dist"]["allreduce_exposed_ms_per_step"] > 0 and d["dist"]["allreduce_bytes_total"] > d["dist"]["early_bucket_bytes"] > 0
    assert d["dist"]["early_reduce"] is True and d["dist"]["encoder_bwd_ms"] > 0 and "dp2" in d["config"]["parallelism"]
    d1 = _run("--gpus", "2", "--workload", "train", "--shape", "32", "32", "32", "--features", "32", "--steps", "2", "--warmup", "1",
              "--no-early-reduce", env={"MMR_BENCH_BACKEND": "gloo"})
    assert d1["dist"]["early_reduce"] is False and d1["dist"]["early_bucket_bytes"] == 0
    assert d1["dist"]["allreduce_bytes_in_exposed_region"] == d1["dist"]["allreduce_bytes_total"] and "one bucket" in d1["config"]["parallelism"]
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    d = _run("--gpus", "2", "--shape", "32", "32", "32", "--features", "32", "--steps", "2", "--warmup", "1",
             env={"MMR_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["dist"]["world_size"] == 2 and d["secondary"]["n_gpus"] == 2
    assert d["secondary"]["dist"]["allreduce_exposed_ms_per_step"] > 0 and "cpu_baseline" not in d
    dp = d["dp_training"]
    assert dp["n_gpus"] == 2 and dp["allreduce_exposed_ms_per_step"] > 0 and dp["allreduce_bytes_total"] > 0 and dp["backend"] == "gloo"
    # the first multi-rank record answers "does the overlapped bucket help?" by itself: both forms timed, and the encoder's
    # backward segment with and without the collective in flight
    assert dp["early_reduce"] is True and dp["ms_per_step_single_bucket"] > 0
    assert dp["encoder_bwd_ms"] > 0 and dp["encoder_bwd_ms_single_bucket"] > 0
    assert "same_workload_fp32" not in d["secondary"]      # one-rank leg only
    assert [r["rank"] for r in d["dist"]["rank_devices"]] == [0, 1]


def test_train_and_ncc_lines_small():
    d = _run("--workload", "train", "--shape", "32", "32", "32", "--features", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert KEYS <= set(d) and d["dtype"] == "fp32x3" and d["value"] > 0 and "dp1" in d["config"]["parallelism"]
    d = _run("--workload", "ncc", "--shape", "64", "64", "64", "--steps", "3", "--warmup", "1")
    assert KEYS <= set(d) and d["roofline"]["bound"] == "hbm" and d["roofline"]["unit"] == "GB/s"
    bw = d["bwd"]      # configs[4] is a loss: its backward is timed too
    assert "error" not in bw, bw
    assert bw["ncc_bwd"]["ms"] > 0 and bw["bending_bwd"]["ms"] > 0 and abs(bw["ms_per_step"] - bw["ncc_bwd"]["ms"] - bw["bending_bwd"]["ms"]) < 1e-9
    assert 0 < bw["ncc_bwd"]["frac_of_hbm_peak"] < 1 and 0 < bw["bending_bwd"]["frac_of_hbm_peak"] < 1


def test_torchrun_form_two_ranks():
    """The driver's N > 1 form: `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` -- every process is
    one rank (WORLD_SIZE set, so nobody becomes the launcher), rank 0 prints the one line.  gloo here (two ranks on one card)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ, MMR_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "train", "--shape", "32", "32",
                        "32", "--features", "32", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist"]["world_size"] == 2 and d["dist"]["launched_by"] == "torchrun" and d["dist"]["allreduce_exposed_ms_per_step"] > 0


def test_default_line_over_rccl_world_1():
    """The whole default line with the process group forced onto `nccl` (= RCCL) at world size 1 (MMR_FORCE_DIST=1): the rank /
    device gather, the barriers and the training leg's gradient all-reduce run as real RCCL calls on the device, as in the
    driver's N > 1 launches; the line reports the backend and the all-reduce it timed."""
    d = _run("--shape", "32", "32", "32", "--features", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
             env={"MMR_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1"})
    assert d["dist"]["backend"] == "nccl" and d["dist"]["world_size"] == 1 and d["dist"]["rank_devices"][0]["rank"] == 0
    dp = d["dp_training"]
    assert dp["backend"] == "nccl" and dp["allreduce_exposed_ms_per_step"] > 0 and dp["allreduce_bytes_total"] > 0
    assert dp["ms_per_step_single_bucket"] > 0      # forced group: the single-bucket A/B leg runs over RCCL too
    assert "rccl" in d["config"]["collectives"]
