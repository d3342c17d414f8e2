"""The RCCL path on ONE GPU (MMR_FORCE_DIST=1): the `nccl` backend (= RCCL on ROCm) is initialised with world_size 1
on the device, and the trainer's broadcast / SUM all-reduce / scalar reductions run as real collectives on device
tensors -- the code an 8-GPU node executes (train_synthmorph.py:284-285 MirroredStrategy equivalent), minus the
peers.  Runs in a child process so the process group never leaks into the other tests."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["MMR_ROOT"])
import torch.distributed as dist
import mmr
from mmr import parallel, synth, training
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
rank, world, local = parallel.init_from_env(device=dev)
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1 and parallel.active()
calls = {"all_reduce": 0, "broadcast": 0}
_ar, _bc = dist.all_reduce, dist.broadcast
def ar(t, *a, **k):
    calls["all_reduce"] += 1; assert t.is_cuda; return _ar(t, *a, **k)
def bc(t, *a, **k):
    calls["broadcast"] += 1; assert t.is_cuda; return _bc(t, *a, **k)
dist.all_reduce, dist.broadcast = ar, bc
shape, L = (16, 16, 32), 4
rng = np.random.default_rng(0)
lab = np.repeat(np.repeat(np.repeat(rng.integers(0, L, (1, 4, 4, 8)), 4, 1), 4, 2), 4, 3).astype(np.uint8)[..., None]
kw = dict(in_shape=shape, in_label_list=np.arange(L), out_label_list=np.arange(L), warp_std=1, warp_res=8, blur_std=1,
          bias_std=0.3, bias_res=8, gamma_std=0.25)
g1, g2 = synth.labels_to_image(**kw, id=0, seed=1), synth.labels_to_image(**kw, id=1, seed=2)
model = mmr.networks.VxmDense(shape, nb_unet_features=([32, 32], [32, 32, 32]), int_steps=3, int_resolution=2,
                              svf_resolution=2, compute_dtype="fp32x3", seed=5)
tr = training.SynthMorphTrainer(model, g1, g2, reg_param=0.5, optimizer=training.Adam(1e-3), world_size=world, rank=rank)
assert calls["broadcast"] == 1                     # parameters mirrored from rank 0 at construction
d1, d2 = g1.draw(1), g2.draw(1)
tr.forward_backward(lab, lab, d1, d2)
g_local = tr.gflat.clone()
parallel.allreduce_sum_(tr.gflat)                  # world 1: SUM over one rank must return the same bits
assert torch.equal(tr.gflat, g_local) and calls["all_reduce"] == 1
l0 = float(tr.train_step(lab, lab, d1, d2)["loss"])
assert calls["all_reduce"] == 3                    # a step = two buckets: decoder + flow under the encoder's backward, encoder after it
for _ in range(8):
    l1 = float(tr.train_step(lab, lab, d1, d2)["loss"])
assert l1 < l0
m = parallel.allreduce_mean_scalar(1.25, dev)
assert m == 1.25
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"ok": True, "calls": calls, "loss0": l0, "loss1": l1}))
"""


def _env():
    env = dict(os.environ, MMR_FORCE_DIST="1", MMR_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_rccl_world1_trainer_collectives(dev):
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["calls"]["all_reduce"] >= 10 and out["calls"]["broadcast"] == 1


def test_bench_train_leg_through_rccl(dev):
    """bench.py --workload train with the forced single-rank nccl group: the line is produced with the all-reduce in
    the timed step (what the driver's N > 1 launches run)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "train", "--shape", "32", "32", "32",
                        "--features", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"].get("collectives") == "rccl (forced single-rank group)"
