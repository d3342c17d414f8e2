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


def test_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` without torchrun: the launcher starts two fresh rank processes (before touching the GPU)
    and the line reports the group they formed.  On this one-GPU box the ranks share the card and exchange gradients over
    gloo (MMR_BENCH_BACKEND); the driver's 8-GPU run takes the same path with nccl = RCCL."""
    d = _run("--gpus", "2", "--workload", "train", "--shape", "32", "32", "32", "--features", "32", "--steps", "2", "--warmup", "1",
             env={"MMR_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["dist"] == {**d["dist"], "world_size": 2, "backend": "gloo", "launched_by": "self"}
    assert d["dist"]["allreduce_ms_per_step"] > 0 and d["dist"]["allreduce_bytes"] > 0 and "dp2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    d = _run("--gpus", "2", "--shape", "32", "32", "32", "--features", "32", "--steps", "2", "--warmup", "1",
             env={"MMR_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["dist"]["world_size"] == 2 and d["secondary"]["n_gpus"] == 2
    assert d["secondary"]["dist"]["allreduce_ms_per_step"] > 0 and "cpu_baseline" not in d
    dp = d["dp_training"]
    assert dp["n_gpus"] == 2 and dp["allreduce_ms_per_step"] > 0 and dp["allreduce_bytes"] > 0 and dp["backend"] == "gloo"
    assert [r["rank"] for r in d["dist"]["rank_devices"]] == [0, 1]


def test_train_and_ncc_lines_small():
    d = _run("--workload", "train", "--shape", "32", "32", "32", "--features", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert KEYS <= set(d) and d["dtype"] == "fp32x3" and d["value"] > 0 and "dp1" in d["config"]["parallelism"]
    d = _run("--workload", "ncc", "--shape", "64", "64", "64", "--steps", "3", "--warmup", "1")
    assert KEYS <= set(d) and d["roofline"]["bound"] == "hbm" and d["roofline"]["unit"] == "GB/s"


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
    assert d["n_gpus"] == 2 and d["dist"]["world_size"] == 2 and d["dist"]["launched_by"] == "torchrun" and d["dist"]["allreduce_ms_per_step"] > 0


def test_default_line_over_rccl_world_1():
    """The whole default line with the process group forced onto `nccl` (= RCCL) at world size 1 (MMR_FORCE_DIST=1): the rank /
    device gather, the barriers and the training leg's gradient all-reduce run as real RCCL calls on the device, as in the
    driver's N > 1 launches; the line reports the backend and the all-reduce it timed."""
    d = _run("--shape", "32", "32", "32", "--features", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
             env={"MMR_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1"})
    assert d["dist"]["backend"] == "nccl" and d["dist"]["world_size"] == 1 and d["dist"]["rank_devices"][0]["rank"] == 0
    dp = d["dp_training"]
    assert dp["backend"] == "nccl" and dp["allreduce_ms_per_step"] > 0 and dp["allreduce_bytes"] > 0
    assert "rccl" in d["config"]["collectives"]
