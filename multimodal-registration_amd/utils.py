"""Mirrors of ``vxm.utils`` helpers the reference calls on NumPy arrays
(train_synthmorph.py:67; bids_two_steps_registration.py:324; 3d_reg.py:394)."""
import numpy as np
import torch

from . import ops
from .layers import to_device


def _ret(out, numpy_in):
    return out.cpu().numpy() if numpy_in else out


def transform(vol, loc_shift, interp_method="linear", indexing="ij", fill_value=None):
    """Unbatched ``vxm.utils.transform``: vol [*S,C] (or [*S]), loc_shift [*S,3] or channel-wise [*S,C,3]."""
    if indexing != "ij":
        raise ValueError("only indexing='ij' is supported")
    numpy_in = not isinstance(vol, torch.Tensor)
    v = to_device(vol)
    f = to_device(loc_shift)
    squeeze = False
    if v.dim() == 3:
        v = v[..., None]
        squeeze = True
    out = ops.warp3d(v[None].contiguous(), f[None].contiguous(), interp_method, fill_value)[0]
    if squeeze:
        out = out[..., 0]
    return _ret(out, numpy_in)


def compose(transforms, interp_method="linear", shift_center=True, indexing="ij"):
    """``vxm.utils.compose([A, B, ...])`` for dense shifts: apply A first, then B (Appendix A5)."""
    if interp_method != "linear" or indexing != "ij":
        raise ValueError("compose supports linear interpolation and 'ij' indexing only")
    if len(transforms) < 2:
        raise ValueError("compose needs at least two transforms")
    numpy_in = not isinstance(transforms[0], torch.Tensor)
    ts = [to_device(t)[None].contiguous() for t in transforms]
    curr = ts[-1]
    for nxt in reversed(ts[:-1]):
        curr = ops.compose(nxt, curr)
    return _ret(curr[0], numpy_in)


def rescale_dense_transform(transform, factor, interp_method="linear", grid=None):
    """``vxm.utils.rescale_dense_transform``; batched iff rank > ndims + 1 (Appendix A4).  ``grid`` picks the
    upstream resize grid ('align_corners' | 'arange_over_f'); None = the setting of ``mmr.semantics``."""
    if interp_method != "linear":
        raise ValueError("rescale_dense_transform supports interp_method='linear' only")
    numpy_in = not isinstance(transform, torch.Tensor)
    t = to_device(transform)
    batched = t.dim() > t.shape[-1] + 1
    if not batched:
        t = t[None].contiguous()
    out = ops.rescale_transform(t, factor, grid=grid)
    if not batched:
        out = out[0]
    return _ret(out, numpy_in)
