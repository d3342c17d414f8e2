"""Sub-volume tiling and overlap-weighted fusion (host side, NumPy).

Behavioural mirror of the reference's inference pre/post-processing:
tile grid = 3d_reg.py:159-207 (incl. the float-truncation quirks recorded in
SURVEY.md B4/B5), fusion = 3d_reg.py:214-259.  Pinned by golden vectors
generated from the reference (tests/golden/host_helpers.npz).
"""
import itertools

import numpy as np


def round_down_16(shape):
    """The reference 'pads' with int(ceil(s // 16)) * 16, which floors."""
    return tuple(int(s) // 16 * 16 for s in shape)


def normalise_overlap(p):
    """Out-of-range min_perc_overlap handling of 3d_reg.py:165-172."""
    if p >= 1:
        return p / 100 if p / 100 < 1 else 0.1
    if p <= 0:
        return 0.1
    return p


def axis_spans(size, tile, perc):
    """[(lo, hi)] along one axis: count int(S/(T-pT))+1, overlap (T-S/n)*n/(n-1), int()-truncated origins."""
    n = int(size / (tile - perc * tile)) + 1
    overlap = (tile - size / n) * (n / (n - 1)) if n > 1 else 0
    spans, hi = [], 0
    for i in range(n):
        lo = int(hi - overlap) if i else 0
        hi = int(lo + tile)
        spans.append((lo, hi))
    return spans


def subvolume_grid(vol_shape, subvol_size, min_perc_overlap):
    """-> (tile_shape, [(x0,x1,y0,y1,z0,z1)]) in x-major order like the reference's triple loop."""
    tile = round_down_16(subvol_size)
    p = normalise_overlap(min_perc_overlap)
    per_axis = [axis_spans(vol_shape[d], tile[d], p) for d in range(3)]
    coords = [(a[0], a[1], b[0], b[1], c[0], c[1]) for a, b, c in itertools.product(*per_axis)]
    return tile, coords


def extract_subvolumes(vol, coords):
    return [vol[c[0]:c[1], c[2]:c[3], c[4]:c[5]] for c in coords]


def pyramid_weights(tile_shape):
    """1 at the tile centre, falling linearly with Chebyshev distance: 1 - d_inf / (max + 1)."""
    half = [s // 2 for s in tile_shape]
    g = np.ogrid[-half[0]:half[0], -half[1]:half[1], -half[2]:half[2]]
    cheb = np.maximum(np.maximum(np.abs(g[0]), np.abs(g[1])), np.abs(g[2]))
    return 1 - cheb / (cheb.max() + 1)


def fuse_subvolume_fields(tile_shape, vol_shape, coords, fields):
    """Normalised pyramid-weighted blend of per-tile fields [T,T,T,3] into [X,Y,Z,3] (float64).

    Voxels no tile covers keep weight sum 0 -> forced to 1 -> zero displacement, as in the reference."""
    w = pyramid_weights(tile_shape)
    num = np.zeros(tuple(vol_shape[:3]) + (3,))
    den = np.zeros(tuple(vol_shape[:3]))
    for c, f in zip(coords, fields):
        sl = (slice(c[0], c[1]), slice(c[2], c[3]), slice(c[4], c[5]))
        den[sl] += w
        num[sl] += w[..., None] * np.asarray(f, dtype=np.float64)
    den[den == 0] = 1
    return num / den[..., None]
