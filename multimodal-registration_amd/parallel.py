"""Data-parallel plumbing: one process per GPU, batch sharded by rank, ONE collective per step.

Replaces ``tf.distribute.MirroredStrategy`` (train_synthmorph.py:284-285, the reference's only
collective): Keras scales each replica's loss by 1/num_replicas and SUM-all-reduces the gradients.
Here every rank holds the 22 gradient tensors as views of one flat fp32 buffer, so the exchange is a
``all_reduce(SUM)`` of 5.8 MB (64 features) / 92 MB (256 features) over RCCL/xGMI -- issued as two buckets: everything but the
encoder's gradients as soon as the decoder's backward is done (it runs under the encoder's backward, >= 3 ms at C3), the
encoder's few hundred KB at the end -- followed by the 1/world scale folded into the Adam kernel.  Label-map synthesis and the data feed shard by index with
no communication.  Works on any torch.distributed backend (RCCL on GPUs, gloo in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def forced():
    """MMR_FORCE_DIST=1: create the process group and run every collective even with ONE rank, so that a 1-GPU box
    exercises exactly the code an 8-GPU node runs (RCCL load, communicator creation on the device, all-reduce /
    broadcast of device tensors on the current stream)."""
    return os.environ.get("MMR_FORCE_DIST", "0") == "1"


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_from_env(backend=None, device=None):
    """torchrun-style env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) -> (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        backend = backend or os.environ.get("MMR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def active(group=None):
    """True when collectives must actually run: more than one rank, or a forced single-rank group."""
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or forced())


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
    if active(group):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def allreduce_sum_async(flat, group=None):
    """Start the in-place SUM all-reduce of a contiguous slice of the gradient buffer behind the work queued on the current
    stream so far; returns a handle whose ``wait()`` orders the current stream behind the collective (None for a single rank)."""
    if active(group):
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


def allreduce_mean_scalar(value, device, group=None):
    """Mean over ranks of a Python float (validation loss): one 4-byte all-reduce."""
    if not active(group):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t[0]) / dist.get_world_size(group)


def broadcast_(flat, src=0, group=None):
    """Make every rank start from rank ``src``'s parameters (MirroredStrategy mirrors variables)."""
    if active(group):
        dist.broadcast(flat, src=src, group=group)
    return flat


def gather_maps(mine, num_maps, rank, world, device, group=None):
    """Label maps synthesised shard-wise (map i on rank i % world) -> the full list on every rank: one broadcast
    of a uint8 volume per map from its owner (start-up only; a 160^3 map is 4 MB)."""
    if world == 1:
        return list(mine)
    import numpy as np
    shape = tuple(mine[0].shape) if mine else None
    shapes = [None] * world
    dist.all_gather_object(shapes, shape, group=group)
    shape = next(s for s in shapes if s is not None)
    out, k = [], 0
    for i in range(num_maps):
        owner = i % world
        if owner == rank:
            t = torch.as_tensor(np.ascontiguousarray(mine[k]), dtype=torch.uint8).to(device)
            k += 1
        else:
            t = torch.empty(shape, dtype=torch.uint8, device=device)
        dist.broadcast(t, src=owner, group=group)
        out.append(t.cpu().numpy())
    return out


def map_indices(num_maps, rank, world):
    """Label maps synthesised by ``rank`` (independent maps, no exchange)."""
    return list(range(rank, num_maps, world))
