"""NumPy restatement of the upstream operator semantics (TEST INFRASTRUCTURE).

PARITY UNPINNED (see oracle/__init__.py).  Every function cites the reference
call site it serves and the SURVEY.md Appendix A item it follows.  All tensors
are channels-last; spatial rank is 3 unless a function says otherwise.
Arithmetic runs in ``dtype`` (float32 mirrors the TF graph, float64 gives a
"true value" to bound fp32 rounding).
"""
import itertools

import numpy as np


# --------------------------------------------------------------------------- #
# interpolation (A3: neurite ``interpn``)                                      #
# --------------------------------------------------------------------------- #
def interpn(vol, loc, interp_method="linear", fill_value=None, dtype=np.float32):
    """vol [*S, C], loc [*O, D] (voxel coordinates, 'ij') -> [*O, C].

    Linear: clamp-to-edge, corner weights (loc1 - clipped) and 1 - that,
    corners accumulated in itertools.product order.  Nearest: round half to
    even then clamp.  Used under SpatialTransformer (train_synthmorph.py:298),
    Transform (3d_reg.py:331-334) and VecInt/compose/resize.
    """
    vol = np.asarray(vol, dtype=dtype)
    loc = np.asarray(loc, dtype=dtype)
    D = loc.shape[-1]
    S = vol.shape[:D]
    C = vol.shape[D] if vol.ndim > D else 1
    v = vol.reshape(S + (C,))
    maxl = [s - 1 for s in S]
    if interp_method == "nearest":
        idx = [np.clip(np.rint(loc[..., d]).astype(np.int64), 0, maxl[d]) for d in range(D)]
        out = v[tuple(idx)]
    elif interp_method == "linear":
        loc0 = np.floor(loc)
        clipped = [np.clip(loc[..., d], 0, maxl[d]).astype(dtype) for d in range(D)]
        l0 = [np.clip(loc0[..., d], 0, maxl[d]) for d in range(D)]
        l1 = [np.clip(l0[d] + 1, 0, maxl[d]) for d in range(D)]
        w_c0 = [(l1[d].astype(dtype) - clipped[d]).astype(dtype) for d in range(D)]
        w_c1 = [(dtype(1) - w_c0[d]).astype(dtype) for d in range(D)]
        locs = [[l0[d].astype(np.int64), l1[d].astype(np.int64)] for d in range(D)]
        wts = [[w_c0[d], w_c1[d]] for d in range(D)]
        out = np.zeros(loc.shape[:-1] + (C,), dtype=dtype)
        for corner in itertools.product([0, 1], repeat=D):
            idx = tuple(locs[d][corner[d]] for d in range(D))
            w = wts[0][corner[0]]
            for d in range(1, D):
                w = (w * wts[d][corner[d]]).astype(dtype)
            out = (out + w[..., None] * v[idx]).astype(dtype)
    else:
        raise ValueError(interp_method)
    if fill_value is not None:
        oob = np.zeros(loc.shape[:-1], dtype=bool)
        for d in range(D):
            oob |= (loc[..., d] < 0) | (loc[..., d] > maxl[d])
        out = np.where(oob[..., None], dtype(fill_value), out).astype(dtype)
    return out


def _grid(shape, dtype=np.float32):
    return np.stack(np.meshgrid(*[np.arange(s, dtype=dtype) for s in shape], indexing="ij"), -1)


def transform(vol, loc_shift, interp_method="linear", fill_value=None, dtype=np.float32):
    """A2 ``vxm.utils.transform`` (unbatched): out(x) = vol(x + u(x)).

    vol [*S, C], loc_shift [*S, D]; channel-wise form loc_shift [*S, C, D]
    (train_synthmorph.py:67) warps channel c with its own field.
    """
    vol = np.asarray(vol, dtype=dtype)
    loc_shift = np.asarray(loc_shift, dtype=dtype)
    D = loc_shift.shape[-1]
    S = loc_shift.shape[:D]
    if loc_shift.ndim == D + 2:  # channel-wise
        C = loc_shift.shape[D]
        out = np.empty(S + (C,), dtype=dtype)
        for c in range(C):
            out[..., c] = transform(vol[..., c:c + 1], loc_shift[..., c, :], interp_method, fill_value, dtype)[..., 0]
        return out
    loc = (_grid(S, dtype) + loc_shift).astype(dtype)
    if vol.ndim == D:
        vol = vol[..., None]
    return interpn(vol, loc, interp_method, fill_value, dtype)


def spatial_transformer(vol, flow, interp_method="linear", fill_value=None, dtype=np.float32):
    """``vxm.layers.SpatialTransformer`` (batched): vol [B,*S,C], flow [B,*S,3]."""
    return np.stack([transform(vol[b], flow[b], interp_method, fill_value, dtype) for b in range(vol.shape[0])])


def resize(vol, factor, dtype=np.float32, grid="align_corners"):
    """A4 ``ne.utils.resize`` on [*S, C]: new = int(old*f).  grid='align_corners' (default, neurite late 2021):
    sample grid linspace(0, old-1, new); grid='arange_over_f' (older neurite): arange(new) / f, clamp-to-edge
    through interpn."""
    vol = np.asarray(vol, dtype=dtype)
    S = vol.shape[:-1]
    new = [int(s * factor) for s in S]
    lin = []
    for s, n in zip(S, new):
        if grid == "align_corners":
            step = dtype(s - 1) / dtype(max(n - 1, 1))
            lin.append((np.arange(n, dtype=dtype) * step).astype(dtype))
        elif grid == "arange_over_f":
            lin.append((np.arange(n, dtype=dtype) / dtype(factor)).astype(dtype))
        else:
            raise ValueError(grid)
    loc = np.stack(np.meshgrid(*lin, indexing="ij"), -1)
    return interpn(vol, loc, "linear", None, dtype)


def rescale_dense_transform(trf, factor, dtype=np.float32, grid="align_corners"):
    """A4 ``vxm.utils.rescale_dense_transform`` (3d_reg.py:394), unbatched [*S, D]."""
    trf = np.asarray(trf, dtype=dtype)
    if factor < 1:
        return (resize(trf, factor, dtype, grid) * dtype(factor)).astype(dtype)
    return resize((trf * dtype(factor)).astype(dtype), factor, dtype, grid)


def vecint(vel, nsteps, dtype=np.float32):
    """``vxm.layers.VecInt('ss')`` (config.json:41), unbatched [*S, D]."""
    v = (np.asarray(vel, dtype=dtype) / dtype(2 ** nsteps)).astype(dtype)
    for _ in range(nsteps):
        v = (v + transform(v, v, "linear", None, dtype)).astype(dtype)
    return v


def compose(a, b, dtype=np.float32):
    """A5 ``vxm.utils.compose([A, B])`` (bids_two_steps_registration.py:324): B + A o (id + B)."""
    a = np.asarray(a, dtype=dtype)
    b = np.asarray(b, dtype=dtype)
    return (b + transform(a, b, "linear", None, dtype)).astype(dtype)


# --------------------------------------------------------------------------- #
# losses (A6-A8)                                                               #
# --------------------------------------------------------------------------- #
def _div_no_nan(a, b):
    out = np.zeros_like(a)
    np.divide(a, b, out=out, where=(b != 0))
    return out


def _dice_ratio(top, bot, eps_mode):
    if eps_mode == "divide_no_nan":
        return _div_no_nan(top, bot)
    if eps_mode == "max_eps":  # older voxelmorph: bottom = tf.maximum(sum(t + p), 1e-5)
        return top / np.maximum(bot, 1e-5)
    raise ValueError(eps_mode)


def dice_loss(y_true, y_pred, dtype=np.float64, eps_mode="divide_no_nan"):
    """A6 ``vxm.losses.Dice().loss`` (train_synthmorph.py:306): scalar -mean_{b,l} dice."""
    t = np.asarray(y_true, dtype=dtype)
    p = np.asarray(y_pred, dtype=dtype)
    ax = tuple(range(1, t.ndim - 1))
    top = 2 * (t * p).sum(ax)
    bot = (t + p).sum(ax)
    return -np.mean(_dice_ratio(top, bot, eps_mode))


def dice_loss_zeropad(y_true, y_pred, dtype=np.float64):
    """Intent of losses.py:13-21,34-69 (the reference function always raises, B1).

    Batch item 0 only (losses.py:38-39,53-54); voxels where channel 0 >= 1 in
    either map are masked out; mean Dice over labels 1..L-1; returns -dice.
    """
    t = np.asarray(y_true, dtype=dtype)[0]
    p = np.asarray(y_pred, dtype=dtype)[0]
    keep = ~((t[..., 0] >= 1) | (p[..., 0] >= 1))
    t = t * keep[..., None]
    p = p * keep[..., None]
    top = 2 * (t * p).sum((0, 1, 2))[1:]
    bot = (t + p).sum((0, 1, 2))[1:]
    return -np.mean(_div_no_nan(top, bot))


def grad_l2_loss(flow, loss_mult=1.0, dtype=np.float64):
    """A7 ``vxm.losses.Grad('l2', loss_mult).loss(None, flow)`` -> [B]."""
    y = np.asarray(flow, dtype=dtype)
    D = y.ndim - 2
    terms = []
    for d in range(D):
        sl_hi = [slice(None)] * y.ndim
        sl_lo = [slice(None)] * y.ndim
        sl_hi[1 + d] = slice(1, None)
        sl_lo[1 + d] = slice(0, -1)
        df = y[tuple(sl_hi)] - y[tuple(sl_lo)]
        terms.append((df * df).reshape(y.shape[0], -1).mean(1))
    return np.mean(terms, axis=0) * loss_mult


def _box_sum_same(x, win):
    """Sum over a win^3 window with 'SAME' zero padding, [B,*S] -> [B,*S]."""
    lo = (win - 1) // 2
    hi = win - 1 - lo
    out = x
    for ax in (1, 2, 3):
        pad = [(0, 0)] * out.ndim
        pad[ax] = (lo + 1, hi)
        c = np.cumsum(np.pad(out, pad), axis=ax)
        n = out.shape[ax]
        a = np.take(c, np.arange(win, win + n), axis=ax)
        b = np.take(c, np.arange(0, n), axis=ax)
        out = a - b
    return out


def ncc_loss(I, J, win=9, eps=1e-5, dtype=np.float64, form="classic"):
    """A8 ``vxm.losses.NCC(win).loss`` on [B,*S,1]; form='classic' (older upstream, default):
    cc = cross^2 / (I_var*J_var + eps); form='clamped' (newer upstream): cross, I_var, J_var = max(., eps),
    cc = (cross / I_var) * (cross / J_var); returns -mean(cc) per batch item [B]."""
    I = np.asarray(I, dtype=dtype)[..., 0]
    J = np.asarray(J, dtype=dtype)[..., 0]
    ws = float(win ** 3)
    Is, Js = _box_sum_same(I, win), _box_sum_same(J, win)
    I2, J2, IJ = _box_sum_same(I * I, win), _box_sum_same(J * J, win), _box_sum_same(I * J, win)
    uI, uJ = Is / ws, Js / ws
    cross = IJ - uJ * Is - uI * Js + uI * uJ * ws
    Iv = I2 - 2 * uI * Is + uI * uI * ws
    Jv = J2 - 2 * uJ * Js + uJ * uJ * ws
    if form == "classic":
        cc = cross * cross / (Iv * Jv + eps)
    elif form == "clamped":
        cross, Iv, Jv = np.maximum(cross, eps), np.maximum(Iv, eps), np.maximum(Jv, eps)
        cc = (cross / Iv) * (cross / Jv)
    else:
        raise ValueError(form)
    return -cc.reshape(cc.shape[0], -1).mean(1)


def bending_energy(flow, dtype=np.float64):
    """Bending energy defined by this build (no reference implementation, A8):
    mean over interior voxels and channels of dxx^2+dyy^2+dzz^2+2dxy^2+2dxz^2+2dyz^2
    with central second differences; [B,*S,3] -> [B]."""
    u = np.asarray(flow, dtype=dtype)
    c = u[:, 1:-1, 1:-1, 1:-1]
    dxx = u[:, 2:, 1:-1, 1:-1] - 2 * c + u[:, :-2, 1:-1, 1:-1]
    dyy = u[:, 1:-1, 2:, 1:-1] - 2 * c + u[:, 1:-1, :-2, 1:-1]
    dzz = u[:, 1:-1, 1:-1, 2:] - 2 * c + u[:, 1:-1, 1:-1, :-2]
    dxy = (u[:, 2:, 2:, 1:-1] - u[:, 2:, :-2, 1:-1] - u[:, :-2, 2:, 1:-1] + u[:, :-2, :-2, 1:-1]) / 4
    dxz = (u[:, 2:, 1:-1, 2:] - u[:, 2:, 1:-1, :-2] - u[:, :-2, 1:-1, 2:] + u[:, :-2, 1:-1, :-2]) / 4
    dyz = (u[:, 1:-1, 2:, 2:] - u[:, 1:-1, 2:, :-2] - u[:, 1:-1, :-2, 2:] + u[:, 1:-1, :-2, :-2]) / 4
    e = dxx ** 2 + dyy ** 2 + dzz ** 2 + 2 * dxy ** 2 + 2 * dxz ** 2 + 2 * dyz ** 2
    return e.reshape(e.shape[0], -1).mean(1)


# --------------------------------------------------------------------------- #
# U-Net pieces (A1)                                                            #
# --------------------------------------------------------------------------- #
def leaky_relu(x, alpha=0.2):
    return np.where(x >= 0, x, x * x.dtype.type(alpha))


def maxpool2(x):
    """MaxPooling3D(2) on [B,X,Y,Z,C] (floor on odd sizes, as Keras 'valid')."""
    B, X, Y, Z, C = x.shape
    x = x[:, :X // 2 * 2, :Y // 2 * 2, :Z // 2 * 2]
    return x.reshape(B, X // 2, 2, Y // 2, 2, Z // 2, 2, C).max((2, 4, 6))


def upsample2(x):
    """UpSampling3D(2): nearest repeat."""
    return x.repeat(2, 1).repeat(2, 2).repeat(2, 3)


def conv3d_same_np(x, w, b=None, dtype=np.float64):
    """Conv3D(k=3,'same',stride 1) on [B,X,Y,Z,Cin] with Keras kernel
    [3,3,3,Cin,Cout] (cross-correlation, zero padding).  Slow reference used
    for small cases; oracle/conv_c.c is the fast restatement."""
    x = np.asarray(x, dtype=dtype)
    w = np.asarray(w, dtype=dtype)
    B, X, Y, Z, _ = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)))
    out = np.zeros((B, X, Y, Z, w.shape[-1]), dtype=dtype)
    for i in range(3):
        for j in range(3):
            for k in range(3):
                out += xp[:, i:i + X, j:j + Y, k:k + Z] @ w[i, j, k]
    if b is not None:
        out += np.asarray(b, dtype=dtype)
    return out
