"""Layer-level mirrors of ``vxm.layers`` (reference: train_synthmorph.py:298)."""
import numpy as np
import torch

from . import ops


_ASYNC_H2D_BYTES = 1 << 20  # small per-step parameters only: CPU writes into pinned (uncached) memory run at ~1 GB/s
                            # here, so volumes go through data.gen_synthmorph_eb(device=) / h2d_volume instead


_RING = {}      # (shape, dtype) -> [next index, [(pinned tensor, event | None)] * _RING_N]
_RING_N = 4
_RING_KEYS_MAX = 32


def _ring_slot(shape, dtype):
    """Next pinned staging buffer of a small per-shape ring; waits for the copy that last used it (never in practice:
    the host is at most one step ahead).  torch's own pin_memory() would call hipHostMalloc whenever its cached blocks
    are still owned by in-flight copies -- and that call synchronises with the device."""
    key = (tuple(shape), dtype)
    ring = _RING.get(key)
    if ring is None:
        if len(_RING) >= _RING_KEYS_MAX:
            _RING.pop(next(iter(_RING)))
        ring = _RING[key] = [0, [[torch.empty(tuple(shape), dtype=dtype).pin_memory(), None] for _ in range(_RING_N)]]
    slot = ring[1][ring[0]]
    ring[0] = (ring[0] + 1) % _RING_N
    if slot[1] is not None:
        slot[1].synchronize()
    return slot


def to_device(a, dtype=torch.float32, device="cuda"):
    """NumPy / tensor -> contiguous device tensor (Keras casts float64 inputs to fp32)."""
    if isinstance(a, torch.Tensor):
        if a.is_cuda or torch.device(device).type != "cuda" or a.numel() * a.element_size() <= _ASYNC_H2D_BYTES:
            return a.to(device=device, dtype=dtype).contiguous()
        a = a.detach().to(dtype).contiguous().numpy()      # large host tensor: the pinned-scratch path below
    a = np.asarray(a)
    if torch.device(device).type == "cuda" and a.size * np.dtype(a.dtype).itemsize <= _ASYNC_H2D_BYTES and a.size > 0:
        # per-step inputs (label maps, generator draws, blur kernels, ...): a pageable copy would block the host until the
        # stream reaches it, i.e. until the previous step has finished, and the GPU then idles between the generator's
        # small kernels; a pinned staging ring + non_blocking copy keeps the host running ahead
        slot = _ring_slot(a.shape, dtype)
        if any(st < 0 for st in a.strides):
            a = np.ascontiguousarray(a)  # torch cannot wrap negative strides (np.flip views of the batch generator)
        slot[0].copy_(torch.from_numpy(a))
        out = slot[0].to(device=device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(out.device))
        slot[1] = ev
        return out
    t = torch.from_numpy(np.ascontiguousarray(a))
    if torch.device(device).type == "cuda" and t.numel() > 0:
        # large host arrays (weights): cast on the host, then pinned scratch + copy engine (hostio.upload) -- a pageable
        # tensor.to(device) above 1 MB has the runtime pin the caller's own pages in place
        from . import hostio
        return hostio.upload(t.to(dtype).numpy(), device)
    return t.to(device=device, dtype=dtype).contiguous()


# Large host <-> device transfers of predict(): pageable memory moves at a few GB/s and a float64 input costs twice the
# bytes, which at C2 adds ~36 ms to a 45 ms forward.  A few cached pinned staging buffers (keyed by shape) take the
# dtype conversion on the host (torch's threaded copy) and move at PCIe speed.
_PINNED = {}
_PINNED_MAX = 16


def _pinned(shape, dtype, tag):
    key = (tuple(shape), dtype, tag)
    buf = _PINNED.get(key)
    if buf is None:
        if len(_PINNED) >= _PINNED_MAX:
            _PINNED.pop(next(iter(_PINNED)))
        buf = _PINNED[key] = torch.empty(tuple(shape), dtype=dtype).pin_memory()
    return buf


def h2d_volume(a, device, dtype=torch.float32, tag=0):
    """NumPy (any real dtype) / CPU tensor -> device tensor of ``dtype``.  fp32 targets go through hostio: the caller's
    pages are pinned in place (or memcpy'd into cached pinned staging memory) and a kernel converts to fp32 while it
    reads them over PCIe -- no host-side conversion, no write into uncached pinned memory."""
    if isinstance(a, torch.Tensor) and a.is_cuda:
        return a.to(dtype).contiguous()
    if torch.device(device).type != "cuda":
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(dtype).contiguous()
    if dtype == torch.float32:
        from . import hostio
        return hostio.to_device_f32(a.numpy() if isinstance(a, torch.Tensor) else a, device, tag=tag)
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    pin = _pinned(t.shape, dtype, ("in", tag))
    pin.copy_(t)  # threaded conversion + copy on the host
    out = pin.to(device, non_blocking=True)
    torch.cuda.current_stream().synchronize()  # the staging buffer is reused by the next call
    return out


def d2h_volume(t, tag=0):
    """Device tensor -> fresh NumPy array (through cached pinned staging memory, hostio.to_host)."""
    if not t.is_cuda:
        return t.detach().cpu().numpy()
    from . import hostio
    return hostio.to_host(t, tag=tag)


class SpatialTransformer:
    """``vxm.layers.SpatialTransformer(interp_method, name=, fill_value=None)([vol, flow])``.

    vol [B,X,Y,Z,C], flow [B,X,Y,Z,3] in voxels, 'ij' indexing, clamp-to-edge
    unless ``fill_value`` is given.
    """

    def __init__(self, interp_method="linear", indexing="ij", single_transform=False, fill_value=None,
                 shift_center=True, name=None, **kwargs):
        if indexing != "ij":
            raise ValueError("only indexing='ij' is supported (the reference never changes it)")
        ops.interp_code(interp_method)
        self.interp_method = interp_method
        self.fill_value = fill_value
        self.single_transform = single_transform
        self.name = name

    def __call__(self, inputs):
        vol, flow = inputs
        numpy_in = not isinstance(vol, torch.Tensor)
        vol = to_device(vol)
        flow = to_device(flow)
        if self.single_transform and flow.shape[0] == 1 and vol.shape[0] > 1:
            flow = flow.expand(vol.shape[0], *flow.shape[1:]).contiguous()
        out = ops.warp3d(vol, flow, self.interp_method, self.fill_value)
        return out.cpu().numpy() if numpy_in else out


class VecInt:
    """``vxm.layers.VecInt(method='ss', int_steps=)`` (config.json:41)."""

    def __init__(self, indexing="ij", method="ss", int_steps=7, name=None, **kwargs):
        if method not in ("ss", "scaling_and_squaring"):
            raise ValueError("only scaling-and-squaring integration is implemented (what the reference uses)")
        self.int_steps = int_steps

    def __call__(self, vel):
        numpy_in = not isinstance(vel, torch.Tensor)
        out = ops.vecint(to_device(vel), self.int_steps)
        return out.cpu().numpy() if numpy_in else out


class RescaleTransform:
    """``vxm.layers.RescaleTransform(zoom_factor)``: resize a dense field and scale its vectors."""

    def __init__(self, zoom_factor, interp_method="linear", name=None, **kwargs):
        if interp_method != "linear":
            raise ValueError("RescaleTransform supports interp_method='linear' only")
        self.zoom_factor = zoom_factor

    def __call__(self, trf):
        numpy_in = not isinstance(trf, torch.Tensor)
        out = ops.rescale_transform(to_device(trf), self.zoom_factor)
        return out.cpu().numpy() if numpy_in else out
