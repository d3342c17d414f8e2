"""Build + ctypes binding of oracle/conv_c.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_c.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "conv_c.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", src, "-o", _SO, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def conv3d_same(x, w, b=None, leaky=False, alpha=0.2, f32acc=False):
    """x [B,X,Y,Z,Cin] f32, w [3,3,3,Cin,Cout] f32 -> [B,X,Y,Z,Cout] f32 (double accumulate)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    B, X, Y, Z, Cin = x.shape
    Cout = w.shape[-1]
    assert w.shape == (3, 3, 3, Cin, Cout)
    out = np.empty((B, X, Y, Z, Cout), dtype=np.float32)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    fn = lib().oracle_conv3d_k3_same_f32acc if f32acc else lib().oracle_conv3d_k3_same
    fn(_p(x), _p(w), _p(bb) if bb is not None else None, _p(out),
       B, X, Y, Z, Cin, Cout, int(bool(leaky)), ctypes.c_float(alpha))
    return out


def warp3d_linear(vol, flow):
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    flow = np.ascontiguousarray(flow, dtype=np.float32)
    B, X, Y, Z, C = vol.shape
    out = np.empty_like(vol)
    lib().oracle_warp3d_linear(_p(vol), _p(flow), _p(out), B, X, Y, Z, C)
    return out
