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
