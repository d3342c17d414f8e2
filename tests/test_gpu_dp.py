"""Data-parallel equivalence of the REAL trainer: two ranks (one GPU, gloo, one pair each) against one process with
the batch of two -- the SUM-all-reduced gradient buffer must equal the full-batch gradient, every rank must end a
step with identical weights, and the Adam step must use the 1/world scale (train_synthmorph.py:284-285,302-308)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE, ENC, DEC, L = (16, 16, 32), [32, 32], [32, 32, 32], 5


class _FixedGen:
    """Stands in for LabelsToImage: returns precomputed (image, labels) so that the comparison is deterministic."""

    def __init__(self, image, labels, L):
        self.image, self.labels, self.L = image, labels, L

    def generate(self, labels, draws=None, want_onehot=False):
        return dict(image=self.image, labels=self.labels, onehot=None)


def _inputs(seed=0):
    rng = np.random.default_rng(seed)
    mk = lambda: np.repeat(np.repeat(np.repeat(rng.integers(0, L, (2, 4, 4, 8)), 4, 1), 4, 2), 4, 3).astype(np.uint8)[..., None]
    lab1, lab2 = mk(), mk()
    img = lambda: rng.random((2,) + SHAPE + (1,)).astype(np.float32)
    return lab1, lab2, img(), img()


def _trainer(dev, sl, world, rank, pg=None, early_reduce=True, nrep=1):
    sys.path.insert(0, ROOT)
    import mmr
    from mmr import training
    from oracle import net_np
    lab1, lab2, im1, im2 = _inputs()
    if nrep > 1:   # four rows from the two drawn ones (rows 2, 3 = rows 1, 0 with the image pair swapped)
        lab1, lab2 = np.concatenate([lab1, lab2[::-1]]), np.concatenate([lab2, lab1[::-1]])
        im1, im2 = np.concatenate([im1, im2[::-1]]), np.concatenate([im2, im1[::-1]])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl])).to(dev)
    model = mmr.networks.VxmDense(SHAPE, nb_unet_features=(ENC, DEC), int_steps=3, int_resolution=2, svf_resolution=2,
                                  compute_dtype="fp32", device=dev, seed=rank + 7)  # different init per rank: broadcast fixes it
    if world == 1 or rank == 0:
        model.set_weights(net_np.init_weights(ENC, DEC, seed=3, flow_std=3e-2))
    tr = training.SynthMorphTrainer(model, _FixedGen(t(im1), t(lab1), L), _FixedGen(t(im2), t(lab2), L), reg_param=0.8,
                                    optimizer=training.Adam(1e-3), world_size=world, rank=rank, process_group=pg,
                                    early_reduce=early_reduce)
    return tr, t(lab1), t(lab2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    tr, l1, l2 = _trainer(dev, slice(rank, rank + 1), world, rank)
    w0 = tr.model._flat.clone()
    tr.forward_backward(l1, l2)
    sys.path.insert(0, ROOT)
    from mmr import parallel
    parallel.allreduce_sum_(tr.gflat)
    g = tr.gflat.clone()
    tr.train_step(l1, l2)
    torch.save({"w0": w0.cpu(), "g": g.cpu(), "w1": tr.model._flat.cpu()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_with_the_batch_of_two(dev, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["w0"], r1["w0"])                      # parameters were broadcast from rank 0
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["w1"], r1["w1"])
    tr, l1, l2 = _trainer(dev, slice(0, 2), 1, 0)
    assert torch.equal(tr.model._flat.cpu(), r0["w0"])
    tr.forward_backward(l1, l2)
    g_full = tr.gflat.cpu()
    rel = float((r0["g"] - g_full).abs().max() / g_full.abs().max())
    assert rel < 1e-5, rel                                       # sum of the per-rank gradients == full-batch gradient
    # Keras scales each replica's loss by 1/replicas: the DP Adam step sees g/2; replay it on the single process
    tr.opt.apply(tr.model._flat, tr.gflat, grad_scale=1.0 / world)
    rel_w = float((tr.model._flat.cpu() - r0["w1"]).abs().max() / (r0["w1"] - r0["w0"]).abs().max())
    assert rel_w < 1e-3, rel_w


def _worker4(rank, world, port, out_dir):
    """Four ranks on the one card over gloo, one pair each: two train_steps with the two-bucket exchange (bucket 1 asynchronous,
    started when the backward reaches the last encoder conv), then the same two steps with ONE all-reduce after the backward."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    out = {}
    for early in (True, False):
        tr, l1, l2 = _trainer(dev, slice(rank, rank + 1), world, rank, early_reduce=early, nrep=2)
        assert tr.early_reduce is early and tr._bucket_li == len(ENC) - 1
        if early:     # the reduced gradient of the FIRST step (before any Adam update), for the full-batch comparison
            import mmr.parallel as par0
            tr.forward_backward(l1, l2)
            par0.allreduce_sum_(tr.gflat)
            out["g_first"] = tr.gflat.cpu()
        seen = []
        import mmr.parallel as par
        real_async = par.allreduce_sum_async

        def spy(flat, group=None):
            seen.append(int(flat.numel()))
            return real_async(flat, group)
        par.allreduce_sum_async = spy
        try:
            for _ in range(2):
                tr.train_step(l1, l2)
        finally:
            par.allreduce_sum_async = real_async
        cut = tr.goff[2 * (tr._bucket_li + 1)]
        assert seen == ([tr.gflat.numel() - cut] * 2 if early else [])     # bucket 1 = everything behind the encoder, once per step
        assert tr._ar_early is None
        out["two" if early else "one"] = tr.model._flat.cpu()
        out["g_two" if early else "g_one"] = tr.gflat.cpu()
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_two_buckets_equal_one_bucket_and_the_batch_of_four(dev, tmp_path):
    world = 4
    mp.spawn(_worker4, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    for r in rs[1:]:
        assert torch.equal(r["two"], rs[0]["two"]) and torch.equal(r["one"], rs[0]["one"])     # ranks stay in lockstep
    # same reduced gradient either way (gloo sums in another chunk order: not bitwise), same weights after two Adam steps
    g2, g1 = rs[0]["g_two"], rs[0]["g_one"]
    assert float((g2 - g1).abs().max() / g1.abs().max()) < 1e-5
    step = float((rs[0]["one"] - _trainer(dev, slice(0, 4), 1, 0, nrep=2)[0].model._flat.cpu()).abs().max())
    assert float((rs[0]["two"] - rs[0]["one"]).abs().max()) < 2e-3 * step
    # and the world-4 gradient is the full-batch gradient of the four pairs
    tr, l1, l2 = _trainer(dev, slice(0, 4), 1, 0, nrep=2)
    tr.forward_backward(l1, l2)
    gf = rs[0]["g_first"]
    rel = float((gf - tr.gflat.cpu()).abs().max() / gf.abs().max())
    assert rel < 1e-5, rel
