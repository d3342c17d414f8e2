"""bench.py's own launcher (`--gpus N` without torchrun), CPU side: it must start N ranks with the torchrun environment and
give up at once when a rank dies instead of leaving the others waiting in a collective.  (The N-rank run itself is a GPU test:
tests/test_gpu_bench_contract.py::test_gpus_2_starts_two_ranks.)"""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_refuses_more_nccl_ranks_than_devices():
    """`--gpus N` with fewer than N visible devices: a clear message and exit code 2 before any rank starts (two nccl ranks on one
    device would sit in their first collective until the launch timeout)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a machine with fewer than two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MMR_BENCH_BACKEND")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "needs 2 visible GPUs" in r.stderr and "found " in r.stderr
    assert time.time() - t0 < 120


def test_launcher_gives_up_when_a_rank_dies():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU: there every rank dies in torch.cuda.set_device")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["MMR_BENCH_BACKEND"] = "gloo"       # past the device-count gate of the nccl form: the ranks start, and die
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""          # no JSON line, non-zero exit ...
    assert time.time() - t0 < 120                                   # ... and no 25-minute wait for the rendezvous
    assert "No HIP GPUs" in r.stderr or "cuda" in r.stderr.lower()


def test_rank_environment_of_the_children(monkeypatch, tmp_path):
    """The children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT like torchrun's; checked by pointing the
    launcher's interpreter at a stub that records its environment instead of running bench.py."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    stub = tmp_path / "python_stub.py"
    stub.write_text("import os, sys, json\n"
                    "open(os.path.join(os.environ['STUB_OUT'], 'rank' + os.environ['RANK'] + '.json'), 'w').write(json.dumps("
                    "{k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'MMR_BENCH_LAUNCHED')}))\n"
                    "print('{\"rank\": ' + os.environ['RANK'] + '}')\n")
    monkeypatch.setenv("STUB_OUT", str(tmp_path))
    real_popen = subprocess.Popen

    def fake_popen(cmd, **kw):           # run the stub in place of `python bench.py ...`, same env / pipes
        return real_popen([sys.executable, str(stub)], **kw)
    monkeypatch.setattr(bench.subprocess, "Popen", fake_popen)
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: (3, "test"))
    def no_hip(*a, **k):
        raise AssertionError("the launcher parent must not call into torch.cuda")
    monkeypatch.setattr(bench.torch.cuda, "device_count", no_hip)
    monkeypatch.setattr(bench.torch.cuda, "is_available", no_hip)
    monkeypatch.setattr(bench.sys, "stdout", open(tmp_path / "stdout.txt", "w"))
    rc = bench.launch_ranks(3, ["--gpus", "3"])
    bench.sys.stdout.close()
    assert rc == 0
    import json
    envs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" and e["MMR_BENCH_LAUNCHED"] == "self" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1
    assert open(tmp_path / "stdout.txt").read().strip() == '{"rank": 0}'     # only rank 0's line is relayed


def test_visible_gpu_count_needs_no_hip_call(monkeypatch):
    """The launcher counts devices from the environment / sysfs, never through torch.cuda in its own process."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")

    def no_hip(*a, **k):
        raise AssertionError("visible_gpu_count called torch.cuda in the parent")
    monkeypatch.setattr(bench.torch.cuda, "device_count", no_hip)
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2,5")
    assert bench.visible_gpu_count() == (3, "HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == (0, "HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_gpu_count() == (1, "ROCR_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    n, how = bench.visible_gpu_count()      # sysfs (no /dev/kfd here: zero GPU nodes -> falls through to the child) or the child
    assert n >= 0 and how in ("kfd topology", "child process", "unknown")
