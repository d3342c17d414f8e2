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


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_infer_line_small():
    d = _run("--shape", "32", "32", "48", "--features", "64", "--steps", "2", "--warmup", "1", "--no-secondary")
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]


def test_train_and_ncc_lines_small():
    d = _run("--workload", "train", "--shape", "32", "32", "32", "--features", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert KEYS <= set(d) and d["dtype"] == "fp32x3" and d["value"] > 0 and "dp1" in d["config"]["parallelism"]
    d = _run("--workload", "ncc", "--shape", "64", "64", "64", "--steps", "3", "--warmup", "1")
    assert KEYS <= set(d) and d["roofline"]["bound"] == "hbm" and d["roofline"]["unit"] == "GB/s"
