"""NumPy <-> HBM hand-over of whole volumes for ``predict`` (3d_reg.py:310-314: ``get_fdata()`` float64 arrays in, NumPy out).

The host does no arithmetic on the way in: the caller's float64 (or fp32 / uint8 / int16) buffer is either pinned in place
for the duration of the call (``mmr_host_register`` -- only buffers that certainly own their pages, see REGISTER_MIN_BYTES) or
memcpy'd into CPU-cached pinned staging memory (``mmr_host_alloc(cached=1)``), and ``mmr_cast_to_f32`` -- a kernel reading that
host memory over PCIe -- converts to fp32 straight into HBM.  Outputs are written by ``mmr_copy_to_host`` straight into result
arrays that live on pooled anonymous mappings of their own (``exclusive_empty``), pinned for the duration of the copy; the
staging mode copies through pinned staging memory into a fresh array instead.  (Round 4's path converted on the host into
torch's coherent pinned buffers: 1 GB/s on some boxes of the pool, predict() at 2x the forward.)

``MODE_IN`` / ``MODE_OUT`` select the strategy; tools/time_hostio.py measures all of them on the box at hand.
"""
import ctypes

import numpy as np
import torch

from . import _lib

F64, F32, U8, I16 = 0, 1, 2, 3
_CODES = {np.dtype(np.float64): F64, np.dtype(np.float32): F32, np.dtype(np.uint8): U8, np.dtype(np.int16): I16}

MODE_IN = "register"    # "register" | "staging" | "torch"
MODE_OUT = "register"   # "register" | "staging" | "torch"

_STAGING_MAX = 24
_staging = {}            # (nbytes, tag) -> Staging


class Staging:
    """``nbytes`` of pinned, device-mapped, CPU-cached host memory (freed with the object)."""

    def __init__(self, nbytes, cached=True):
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        _lib.check(_lib.load().mmr_host_alloc(ctypes.byref(p), self.nbytes, int(cached)), "mmr_host_alloc")
        self.ptr = p.value
        self._buf = (ctypes.c_uint8 * self.nbytes).from_address(self.ptr)
        self.event = None   # last GPU use of this buffer

    def view(self, dtype, shape):
        a = np.frombuffer(self._buf, dtype=dtype, count=int(np.prod(shape)))
        return a.reshape(shape)

    def wait(self):
        if self.event is not None:
            self.event.synchronize()
            self.event = None

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self.wait()          # no copy may still read the pages that are being released
                self._buf = None
                _lib.load().mmr_host_free(ctypes.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def staging(nbytes, tag):
    key = (int(nbytes), tag)
    s = _staging.get(key)
    if s is None:
        if len(_staging) >= _STAGING_MAX:
            _staging.pop(next(iter(_staging)))
        s = _staging[key] = Staging(nbytes)
    return s


def _stream():
    return torch.cuda.current_stream().cuda_stream


_SCRATCH = [None]


def upload(a, device):
    """C-contiguous NumPy array -> device tensor of the same dtype and shape, through ONE grow-only pinned scratch buffer and the
    copy engine (weights, whole volumes that are not pinned in place).  A plain ``tensor.to(device)`` of pageable memory above
    1 MB makes the runtime pin the CALLER's pages in place for the copy; this path never creates such a mapping."""
    a = np.ascontiguousarray(a)
    out = torch.empty(a.shape, dtype=torch.from_numpy(a.reshape(-1)[:0]).dtype, device=device)
    if a.nbytes == 0:
        return out
    st = _SCRATCH[0]
    if st is None or st.nbytes < a.nbytes:
        _SCRATCH[0] = None            # (the old buffer waits for its last copy in __del__)
        st = _SCRATCH[0] = Staging(max(a.nbytes, 1 << 24))
    st.wait()
    np.copyto(st.view(a.dtype, a.shape), a)
    rc = _lib.load().mmr_memcpy_async(out.data_ptr(), ctypes.c_void_p(st.ptr), a.nbytes, 0, _stream())
    _lib.check(rc, "mmr_memcpy_async")
    st.event = torch.cuda.Event()
    st.event.record()
    st.event.synchronize()
    return out


_LIVE = {}   # host address -> [references, nbytes, device address] of the regions this module has pinned

# Which caller buffers are pinned in place.  hipHostRegister makes a user-pointer mapping of whole PAGES.  A buffer that shares
# pages with other heap data (anything glibc serves from an arena: small arrays, and mid-sized ones once the dynamic mmap threshold
# has grown) drags that data's pages into the mapping, two buffers registered together can share a page, and the same pages come
# back later as the source of somebody's pageable copy, which the runtime pins in place again.  Round 5 saw three GPU memory
# faults inside plain torch host-to-device copies in test processes that had registered such small heap arrays before
# (profiles/HISTORY.md).  Only buffers that certainly own their pages are registered: above glibc's largest mmap threshold
# (32 MiB: always a mapping of their own; a 160 x 160 x 192 float64 volume is 39 MB), or allocated here page-aligned
# (``exclusive_empty``).  Everything else goes through the pinned staging buffers.
REGISTER_MIN_BYTES = (32 << 20) + 4096
_PAGE = 4096
_OWNED = set()   # addresses of exclusive_empty() arrays that are alive


_POOL, _POOL_MAX_PER_SIZE = {}, 3   # size -> idle anonymous mappings (pages already faulted in: pinning them again takes microseconds)


_POOL_MAX_BYTES = 1 << 30


def _recycle(size, buf, addr):
    lst = _POOL.setdefault(size, [])
    if len(lst) < _POOL_MAX_PER_SIZE and sum(k * len(v) for k, v in _POOL.items()) + size <= _POOL_MAX_BYTES:
        lst.append(buf)
    else:
        _OWNED.discard(addr)       # (not buf.close(): the dying array still holds its buffer export here; dropping the last
                                   # reference unmaps it a moment later)


def exclusive_empty(shape, dtype=np.float32):
    """``np.empty`` on an anonymous mapping of its own (page-aligned, whole pages): safe to pin.  The mapping goes back to a small
    pool when the last array that views it dies -- a FRESH mapping costs its page faults when it is pinned (3 ms for predict's
    27 MB of outputs), a recycled one does not; np.empty is fast for the same reason (glibc hands out used heap), which is
    exactly why its pages cannot be trusted to be the array's own."""
    import mmap
    import weakref
    count = int(np.prod(shape))
    size = max((count * np.dtype(dtype).itemsize + _PAGE - 1) // _PAGE * _PAGE, _PAGE)
    lst = _POOL.get(size)
    buf = lst.pop() if lst else mmap.mmap(-1, size)
    root = np.frombuffer(buf, dtype=dtype, count=count)      # every view handed out keeps `root` alive through .base
    _OWNED.add(int(root.ctypes.data))
    weakref.finalize(root, _recycle, size, buf, int(root.ctypes.data))
    return root.reshape(shape)


def registrable(a):
    """True if ``a`` certainly owns the pages it lies on (see REGISTER_MIN_BYTES)."""
    return a.nbytes >= REGISTER_MIN_BYTES or int(a.ctypes.data) in _OWNED



class Registered:
    """Context: the pages of a C-contiguous NumPy array pinned and mapped for the device; ``dev`` = device-side address.
    ``ok`` is False when the runtime refuses (read-only mapping, already registered, ...) -- callers fall back to staging.

    The same buffer may be asked for twice in one call (``predict([a, a])``): the runtime accepts a second registration of a
    pointer it already holds, replaces its table entry and leaks the first pin (the second unregister then fails with "pointer
    does not correspond to a registered memory region") -- so identical addresses are counted here and registered once."""

    def __init__(self, a):
        self.a, self.ok, self.dev, self._key = a, False, None, None

    def __enter__(self):
        key, nbytes = int(self.a.ctypes.data), int(self.a.nbytes)
        if not registrable(self.a):
            return self            # ok stays False: staging
        live = _LIVE.get(key)
        if live is not None:
            if live[1] >= nbytes:
                live[0] += 1
                self.ok, self.dev, self._key = True, live[2], key
            return self            # a longer region from the same address while a shorter one is pinned: staging
        d = ctypes.c_void_p()
        rc = _lib.load().mmr_host_register(ctypes.c_void_p(key), nbytes, ctypes.byref(d))
        self.ok = rc == 0 and bool(d.value)
        self.dev = d.value
        if self.ok:
            self._key = key
            _LIVE[key] = [1, nbytes, d.value]
        return self

    def __exit__(self, *exc):
        if self.ok:
            self.ok = False
            live = _LIVE[self._key]
            live[0] -= 1
            if live[0] == 0:
                del _LIVE[self._key]
                rc = _lib.load().mmr_host_unregister(ctypes.c_void_p(self._key))
                if rc != 0 and exc[0] is None:      # pages left pinned behind the caller's back: loud, not silent
                    _lib.check(rc, "mmr_host_unregister")
        return False


def _cast_launch(src_ptr, dst, n, code):
    rc = _lib.load().mmr_cast_to_f32(ctypes.c_void_p(src_ptr), dst.data_ptr(), int(n), code, _stream())
    _lib.check(rc, "mmr_cast_to_f32")


def to_device_f32(a, device, tag=0, mode=None):
    """NumPy array (float64 / float32 / uint8 / int16; other dtypes are cast to float64 on the host first) -> fp32 device
    tensor of the same shape.  Returns after the transfer has COMPLETED (the caller may reuse or free ``a``)."""
    mode = mode or MODE_IN
    a = np.asarray(a)
    if a.dtype not in _CODES:
        a = a.astype(np.float64)
    if not a.flags.c_contiguous:
        a = np.ascontiguousarray(a)
    out = torch.empty(a.shape, dtype=torch.float32, device=device)
    if a.size == 0:
        return out
    code = _CODES[a.dtype]
    if mode == "torch":
        pin = staging(a.size * 4, ("in32", tag))
        pin.wait()
        np.copyto(pin.view(np.float32, a.shape), a, casting="unsafe")   # host-side conversion (round 4's path, cached pages)
        _cast_launch(pin.ptr, out, a.size, F32)
        torch.cuda.current_stream().synchronize()
        return out
    if mode == "register":
        with Registered(a) as r:
            if r.ok:
                _cast_launch(r.dev, out, a.size, code)
                torch.cuda.current_stream().synchronize()   # before the pages are unpinned
                return out
    st = staging(a.nbytes, ("in", tag))
    st.wait()
    np.copyto(st.view(a.dtype, a.shape), a)         # plain memcpy into write-back pinned pages
    _cast_launch(st.ptr, out, a.size, code)
    ev = torch.cuda.Event()
    ev.record()
    st.event = ev
    ev.synchronize()
    return out


def to_host(t, tag=0, mode=None):
    """fp32 device tensor -> fresh NumPy array."""
    mode = mode or MODE_OUT
    t = t.detach()
    if not t.is_cuda:
        return t.cpu().numpy()
    if not t.is_contiguous():
        t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes == 0 or nbytes % 4 or t.data_ptr() % 16:
        return t.cpu().numpy()
    np_dtype = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32}.get(t.dtype)
    if np_dtype is None:
        return t.cpu().numpy()
    if mode == "register":
        out = exclusive_empty(tuple(t.shape), np_dtype)
        if out.ctypes.data % 16 == 0:
            with Registered(out) as r:
                if r.ok:
                    rc = _lib.load().mmr_copy_to_host(t.data_ptr(), ctypes.c_void_p(r.dev), nbytes, _stream())
                    _lib.check(rc, "mmr_copy_to_host")
                    torch.cuda.current_stream().synchronize()
                    return out
    if mode == "torch":
        return t.cpu().numpy()
    st = staging(nbytes, ("out", tag))
    st.wait()
    rc = _lib.load().mmr_copy_to_host(t.data_ptr(), ctypes.c_void_p(st.ptr), nbytes, _stream())
    _lib.check(rc, "mmr_copy_to_host")
    torch.cuda.current_stream().synchronize()
    return st.view(np_dtype, tuple(t.shape)).copy()


# ---- several volumes per call (predict's [moving, fixed] in, [moved, field] out): one synchronisation for all ----
LAST = {}   # wall-clock split of the most recent call of each kind (bench.py / tools/time_hostio.py read it)


def pair_to_device(arrays, device, mode=None):
    """``to_device_f32`` for a list of arrays with the transfers of all of them in flight together: every buffer is pinned
    (or staged) first, the cast kernels are queued back to back, ONE synchronisation, then the pages are released."""
    import time
    mode = mode or MODE_IN
    t0 = time.perf_counter()
    prep = []
    for k, a in enumerate(arrays):
        a = np.asarray(a)
        if a.dtype not in _CODES:
            a = a.astype(np.float64)
        if not a.flags.c_contiguous:
            a = np.ascontiguousarray(a)
        prep.append(a)
    if mode == "torch":
        outs = [to_device_f32(a, device, tag=("pair", k), mode="torch") for k, a in enumerate(prep)]
        LAST["in"] = {"mode": mode, "total_ms": (time.perf_counter() - t0) * 1e3}
        return outs
    regs, srcs, stage_ms = [], [], 0.0
    try:
        for k, a in enumerate(prep):
            r = Registered(a).__enter__() if (mode == "register" and a.size) else None
            if r is not None and r.ok:
                regs.append(r)
                srcs.append(r.dev)
            else:
                ts = time.perf_counter()
                st = staging(max(a.nbytes, 16), ("in", "pair", k))
                st.wait()
                np.copyto(st.view(a.dtype, a.shape), a)
                srcs.append(st.ptr)
                stage_ms += (time.perf_counter() - ts) * 1e3
        t1 = time.perf_counter()
        outs = []
        for a, src in zip(prep, srcs):
            o = torch.empty(a.shape, dtype=torch.float32, device=device)
            if a.size:
                _cast_launch(src, o, a.size, _CODES[a.dtype])
            outs.append(o)
        torch.cuda.current_stream().synchronize()
        t2 = time.perf_counter()
    finally:
        if regs:
            torch.cuda.current_stream().synchronize()   # also on an error path: no kernel may still read pages that are being unpinned
        for r in regs:
            r.__exit__(None, None, None)
    t3 = time.perf_counter()
    LAST["in"] = {"mode": "register" if regs else "staging", "pin_ms": (t1 - t0) * 1e3 - stage_ms + (t3 - t2) * 1e3,
                  "host_copy_ms": stage_ms, "transfer_ms": (t2 - t1) * 1e3, "total_ms": (t3 - t0) * 1e3,
                  "bytes": int(sum(a.nbytes for a in prep))}
    return outs


def many_to_host(tensors, mode=None):
    """``to_host`` for a list of fp32 device tensors: the copy kernels queued back to back behind whatever produces the
    tensors, ONE synchronisation, then the NumPy copies out of the staging memory."""
    import time
    mode = mode or MODE_OUT
    t0 = time.perf_counter()
    plain = all(t.is_cuda and t.dtype == torch.float32 and t.data_ptr() % 16 == 0 and t.numel() > 0 and t.is_contiguous()
                for t in tensors)
    if not plain or mode == "torch":
        outs = [to_host(t, tag=("many", k), mode=mode) for k, t in enumerate(tensors)]
        LAST["out"] = {"mode": mode, "total_ms": (time.perf_counter() - t0) * 1e3}
        return outs
    if mode == "register":
        # the result arrays themselves are pinned for the duration of the copy: the copy kernels write them over PCIe, no
        # staging buffer and no host-side copy
        outs = [exclusive_empty(tuple(t.shape), np.float32) for t in tensors]      # pages of their own: safe to pin
        regs = [Registered(o).__enter__() for o in outs]
        try:
            if all(r.ok and r.dev % 16 == 0 for r in regs):
                for t, r in zip(tensors, regs):
                    rc = _lib.load().mmr_copy_to_host(t.detach().data_ptr(), ctypes.c_void_p(r.dev), t.numel() * 4, _stream())
                    _lib.check(rc, "mmr_copy_to_host")
                t1 = time.perf_counter()
                torch.cuda.current_stream().synchronize()
                t2 = time.perf_counter()
                LAST["out"] = {"mode": "register", "enqueue_ms": (t1 - t0) * 1e3, "wait_ms": (t2 - t1) * 1e3, "host_copy_ms": 0.0,
                               "bytes": int(sum(t.numel() * 4 for t in tensors))}
                return outs
        finally:
            torch.cuda.current_stream().synchronize()
            for r in regs:
                r.__exit__(None, None, None)
            if "out" in LAST and LAST["out"].get("mode") == "register":
                LAST["out"]["total_ms"] = (time.perf_counter() - t0) * 1e3
        mode = "staging"
    sts = []
    for k, t in enumerate(tensors):
        st = staging(t.numel() * 4, ("out", "many", k))
        st.wait()
        rc = _lib.load().mmr_copy_to_host(t.detach().data_ptr(), ctypes.c_void_p(st.ptr), t.numel() * 4, _stream())
        _lib.check(rc, "mmr_copy_to_host")
        sts.append(st)
    t1 = time.perf_counter()
    torch.cuda.current_stream().synchronize()
    t2 = time.perf_counter()
    outs = [st.view(np.float32, tuple(t.shape)).copy() for st, t in zip(sts, tensors)]
    t3 = time.perf_counter()
    LAST["out"] = {"mode": "staging", "enqueue_ms": (t1 - t0) * 1e3, "wait_ms": (t2 - t1) * 1e3, "host_copy_ms": (t3 - t2) * 1e3,
                   "total_ms": (t3 - t0) * 1e3, "bytes": int(sum(t.numel() * 4 for t in tensors))}
    return outs
