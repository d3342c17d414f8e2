"""Data-parallel plumbing: one process per GPU, batch sharded by rank, ONE collective per step.

Replaces ``tf.distribute.MirroredStrategy`` (train_synthmorph.py:284-285, the reference's only
collective): Keras scales each replica's loss by 1/num_replicas and SUM-all-reduces the gradients.
Here every rank holds the 22 gradient tensors as views of one flat fp32 buffer, so the exchange is a
single ``all_reduce(SUM)`` of 5.8 MB (64 features) / 92 MB (256 features) over RCCL/xGMI followed by the
1/world scale folded into the Adam kernel.  Label-map synthesis and the data feed shard by index with
no communication.  Works on any torch.distributed backend (RCCL on GPUs, gloo in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """torchrun-style env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) -> (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world, local


def shard_rows(n_rows, rank, world):
    """Rows of the global batch owned by ``rank`` (batch_size // nb_devices each, train_synthmorph.py:193-194)."""
    if n_rows % world:
        raise ValueError(f"batch size {n_rows} not a multiple of the number of GPUs {world}")
    b = n_rows // world
    return slice(rank * b, (rank + 1) * b)


def shard_batch(batch, rank, world):
    (src, trg), _ = batch
    sl = shard_rows(src.shape[0], rank, world)
    return src[sl], trg[sl]


def allreduce_sum_(flat, group=None):
    """In-place SUM all-reduce of the flat gradient buffer (no-op for a single rank)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_(flat, src=0, group=None):
    """Make every rank start from rank ``src``'s parameters (MirroredStrategy mirrors variables)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def map_indices(num_maps, rank, world):
    """Label maps synthesised by ``rank`` (independent maps, no exchange)."""
    return list(range(rank, num_maps, world))
