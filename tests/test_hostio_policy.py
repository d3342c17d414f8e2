"""Which host buffers `mmr.hostio` pins in place (no GPU needed): only buffers that certainly own their pages -- arrays above glibc's
largest mmap threshold and arrays allocated on anonymous mappings of their own (`exclusive_empty`), which are pooled so that
their pages are faulted in once."""
import gc

import numpy as np


def test_registrable_only_for_exclusive_pages():
    from mmr import hostio
    small = np.zeros((16, 32, 16), np.float64)                 # glibc heap: shares pages with other objects
    mid = np.empty(20 << 20, np.uint8)                         # may be heap memory once the dynamic mmap threshold has grown
    big = np.empty(hostio.REGISTER_MIN_BYTES, np.uint8)        # above the largest mmap threshold: always a mapping of its own
    assert not hostio.registrable(small) and not hostio.registrable(mid) and hostio.registrable(big)
    assert hostio.REGISTER_MIN_BYTES > (32 << 20) and 160 * 160 * 192 * 8 >= hostio.REGISTER_MIN_BYTES      # the C2 float64 volume is
    r = hostio.Registered(small).__enter__()                   # refused before any runtime call: no library needed
    assert not r.ok and r.dev is None and not hostio._LIVE
    r.__exit__(None, None, None)


def test_exclusive_empty_is_page_aligned_pooled_and_never_shared():
    from mmr import hostio
    a = hostio.exclusive_empty((3, 5, 7), np.float32)
    a[...] = 1.0
    addr = a.ctypes.data
    assert addr % 4096 == 0 and hostio.registrable(a) and a.flags.writeable and a.shape == (3, 5, 7) and a.dtype == np.float32
    b = hostio.exclusive_empty((3, 5, 7), np.float32)          # `a` is alive: a different mapping
    assert b.ctypes.data != addr
    view = a[1:]
    del a
    gc.collect()
    c = hostio.exclusive_empty((3, 5, 7), np.float32)          # a view still holds the first mapping: not handed out again
    assert c.ctypes.data not in (addr, b.ctypes.data)
    del view
    gc.collect()
    d = hostio.exclusive_empty((3, 5, 7), np.float32)          # now it is idle: recycled, pages already faulted in
    assert d.ctypes.data == addr
    many = [hostio.exclusive_empty((11,), np.float64) for _ in range(8)]
    del many
    gc.collect()
    assert all(len(v) <= hostio._POOL_MAX_PER_SIZE for v in hostio._POOL.values())
    assert sum(k * len(v) for k, v in hostio._POOL.items()) <= hostio._POOL_MAX_BYTES
