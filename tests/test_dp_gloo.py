"""world_size-2 gloo test of the data-parallel host logic (runs on CPU): flat-buffer SUM all-reduce +
1/world scaling equals the single-process gradient of the concatenated batch, rank sharding of the
data feed, parameter broadcast."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from mmr import data, parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # identical parameters after broadcast
    flat_w = torch.full((1000,), float(rank + 1))
    parallel.broadcast_(flat_w, 0)
    assert torch.all(flat_w == 1.0)
    # per-rank "gradient" of a quadratic loss on this rank's shard of a global batch
    rng = np.random.default_rng(0)
    maps = [rng.integers(0, 5, (4, 4, 4)).astype(np.uint8) for _ in range(6)]
    gen = data.gen_synthmorph_eb(maps, batch_size=4, same_subj=False, flip=False, random_zero_borders=False,
                                 rng=np.random.default_rng(123))  # same seed on every rank -> same global batch
    batch = next(gen)
    src, trg = parallel.shard_batch(batch, rank, world)
    assert src.shape[0] == 2
    theta = torch.linspace(-1, 1, 64, dtype=torch.float64)
    x = torch.from_numpy(src.reshape(src.shape[0], -1).astype(np.float64))
    # loss_b = sum_i (theta_i * x_bi)^2 ; Keras: sum over the local batch, scaled 1/world
    g_local = (2 * theta[None] * x * x).sum(0) / world
    flat = g_local.clone()
    parallel.allreduce_sum_(flat)
    torch.save({"g": flat, "src": torch.from_numpy(src.copy())}, os.path.join(out_dir, f"r{rank}.pt"))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_dp_allreduce_matches_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["g"], r1["g"])  # every rank holds the same reduced gradient
    sys.path.insert(0, ROOT)
    from mmr import data
    rng = np.random.default_rng(0)
    maps = [rng.integers(0, 5, (4, 4, 4)).astype(np.uint8) for _ in range(6)]
    (src, trg), _ = next(data.gen_synthmorph_eb(maps, batch_size=4, same_subj=False, flip=False,
                                                random_zero_borders=False, rng=np.random.default_rng(123)))
    assert np.array_equal(np.concatenate([r0["src"].numpy(), r1["src"].numpy()]), src)  # shards tile the global batch
    theta = torch.linspace(-1, 1, 64, dtype=torch.float64)
    x = torch.from_numpy(src.reshape(4, -1).astype(np.float64))
    g_full = (2 * theta[None] * x * x).sum(0) / world
    assert torch.allclose(r0["g"], g_full, rtol=1e-12)


def test_shard_helpers():
    sys.path.insert(0, ROOT)
    import pytest
    from mmr import parallel
    assert parallel.shard_rows(8, 3, 4) == slice(6, 8)
    with pytest.raises(ValueError):
        parallel.shard_rows(3, 0, 2)
    assert parallel.map_indices(10, 1, 4) == [1, 5, 9]


# ---- world 4, the two-bucket exchange of the real trainer's flat gradient buffer (host logic, CPU tensors) ----
ARCHS = {"64f": ([64] * 4, [64] * 6), "256f": ([256] * 4, [256] * 6)}


def _bucket_trainer(arch):
    """SynthMorphTrainer over a CPU-resident VxmDense: enough for the buffer layout and the bucket boundary (no kernel runs)."""
    sys.path.insert(0, ROOT)
    import mmr
    from mmr import training

    class _Gen:
        L = 4
    enc, dec = ARCHS[arch]
    model = mmr.networks.VxmDense((16, 16, 16), nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2,
                                  compute_dtype="fp32x3", device="cpu", seed=0)
    return training.SynthMorphTrainer(model, _Gen(), _Gen(), world_size=1, rank=0)


def _worker4(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mmr import parallel
    parallel.init_from_env(backend="gloo")
    res = {}
    for arch in ARCHS:
        tr = _bucket_trainer(arch)
        n = tr.gflat.numel()
        cut = tr.goff[2 * (tr._bucket_li + 1)]
        g = torch.Generator().manual_seed(1000 * rank + len(arch))
        local = torch.randn(n, generator=g)
        # as train_step does it: bucket 1 (everything behind the encoder) asynchronously, then the encoder's bucket, then wait
        tr.gflat.copy_(local)
        h = parallel.allreduce_sum_async(tr.gflat[cut:])
        assert h is not None
        parallel.allreduce_sum_(tr.gflat[:cut])
        h.wait()
        two = tr.gflat.clone()
        tr.gflat.copy_(local)
        parallel.allreduce_sum_(tr.gflat)
        res[arch] = {"two": two, "one": tr.gflat.clone(), "local": local, "cut": cut, "n": n}
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_boundary_of_both_architectures():
    """Bucket 1 = every gradient behind the last encoder conv (layers bucket_li + 1 .. flow head), bucket 2 = the encoder's:
    the cut is goff[2 * (bucket_li + 1)], the offset of the first decoder kernel in Keras order."""
    for arch, (enc, dec) in ARCHS.items():
        tr = _bucket_trainer(arch)
        f = enc[0]
        assert tr._bucket_li == len(enc) - 1 == 3
        enc_params = (27 * 2 * f + f) + 3 * (27 * f * f + f)          # conv0 (2 -> f) + three f -> f convs, kernels + biases
        assert tr.goff[2 * (tr._bucket_li + 1)] == enc_params
        assert tr.goff[2 * tr._bucket_li] < enc_params < tr.gflat.numel()
        total = enc_params + (27 * f * f + f) + 3 * (27 * 2 * f * f + f) + (27 * (2 * f) * f + f) + (27 * f * f + f) + (27 * f * 3 + 3)
        assert tr.gflat.numel() == total and len(tr.goff) == 22
        # sizes quoted in the docs: 5.8 MB at 64 f (5.4 behind the encoder), 92 MB at 256 f
        mb = tr.gflat.numel() * 4 / 1e6
        assert (5.7 < mb < 5.9) if arch == "64f" else (91 < mb < 93)


def test_world_4_two_bucket_allreduce_equals_one_bucket(tmp_path):
    world = 4
    mp.spawn(_worker4, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    for arch in ARCHS:
        want = sum(r[arch]["local"].double() for r in rs)
        for r in rs:
            assert torch.equal(r[arch]["two"], rs[0][arch]["two"])                # every rank ends with the same buffer
            assert torch.allclose(r[arch]["two"].double(), want, rtol=0, atol=1e-5)
            assert torch.allclose(r[arch]["two"], r[arch]["one"], rtol=0, atol=1e-5)   # same sums as the single all-reduce
