"""Registration-quality metrics of the reference's eval_reg_*.py scripts, on device (fp64).

eval_reg_with_jacobian.py:62-91 (Jacobian determinant, folding %), eval_reg_with_mi.py:16-74,123-139
(zero-padding crop + 100-bin NMI), eval_reg_on_sc_seg.py:80-124 (overlap metrics on binary masks).
Pinned by golden vectors produced by running those scripts (tests/golden/make_golden_eval.py).
"""
import numpy as np
import torch

from . import _lib
from .layers import to_device
from .ops import _stream, _ws


def _dev64(a, device="cuda"):
    return to_device(np.asarray(a, dtype=np.float64) if not isinstance(a, torch.Tensor) else a, dtype=torch.float64,
                     device=device)


def jacobian_determinant(ddf, device="cuda"):
    """ddf [X,Y,Z,(1,)3] displacement field -> dict(det [X-4,Y-4,Z-4], percentage_negative, median, mean, std,
    n_total, n_negative) exactly as the reference summarises it."""
    d = _dev64(ddf, device)
    if d.dim() == 5:
        d = d[:, :, :, 0, :]
    if d.dim() != 4 or d.shape[-1] != 3:
        raise ValueError("ddf must be [X,Y,Z,3] or [X,Y,Z,1,3]")
    d = d.contiguous()
    X, Y, Z, _ = d.shape
    det = torch.empty((X - 4, Y - 4, Z - 4), dtype=torch.float64, device=d.device)
    _lib.check(_lib.load().mmr_jacobian_det_f64(d.data_ptr(), det.data_ptr(), X, Y, Z, _stream()), "mmr_jacobian_det_f64")
    flat = det.reshape(-1)
    n = flat.numel()
    srt = torch.sort(flat).values
    median = float(srt[n // 2]) if n % 2 else float(0.5 * (srt[n // 2 - 1] + srt[n // 2]))
    n_neg = int((flat < 0).sum())
    return dict(det=det.cpu().numpy()[..., None], percentage_negative=100.0 * n_neg / n, median=median,
                mean=float(flat.mean()), std=float(flat.std(unbiased=False)), n_total=n, n_negative=n_neg)


def _entropy(counts):
    p = np.asarray(counts, dtype=np.float64).reshape(-1)
    p = p / p.sum()
    p = p[p > 0]
    return float(-(p * np.log(p)).sum())


def normalized_mutual_information(image0, image1, bins=100, device="cuda"):
    """(H0 + H1) / H01 from a bins x bins joint histogram with numpy.histogramdd binning."""
    a, b = _dev64(image0, device).reshape(-1).contiguous(), _dev64(image1, device).reshape(-1).contiguous()
    if a.numel() != b.numel():
        raise ValueError("images must have the same number of voxels")
    edges = []
    for t in (a, b):
        lo, hi = float(t.min()), float(t.max())
        if lo == hi:
            lo, hi = lo - 0.5, hi + 0.5
        edges.append(torch.from_numpy(np.linspace(lo, hi, bins + 1)).to(a.device))
    hist = torch.empty((bins, bins), dtype=torch.int64, device=a.device)
    _lib.check(_lib.load().mmr_joint_hist_f64(a.data_ptr(), b.data_ptr(), edges[0].data_ptr(), edges[1].data_ptr(),
                                              hist.data_ptr(), a.numel(), int(bins), _stream()), "mmr_joint_hist_f64")
    h = hist.cpu().numpy()
    return (_entropy(h.sum(0)) + _entropy(h.sum(1))) / _entropy(h)


def detect_zero_padding(im):
    """Bounding box (x_min, y_min, z_min, x_max, y_max, z_max) of the planes whose sum is > 0."""
    im = np.asarray(im)
    box_lo, box_hi = [], []
    for ax in range(3):
        prof = im.sum(axis=tuple(a for a in range(3) if a != ax))
        nz = np.flatnonzero(prof > 0)
        box_lo.append(int(nz[0]))
        box_hi.append(int(nz[-1]))
    return tuple(box_lo) + tuple(box_hi)


def nmi_report(fixed, moving, moved, device="cuda"):
    """The four numbers eval_reg_with_mi.py writes: NMI before / after / moving-vs-moved, % improvement (rounded)."""
    x0, y0, z0, x1, y1, z1 = detect_zero_padding(moving)
    crop = lambda v: np.asarray(v)[x0:x1 + 1, y0:y1 + 1, z0:z1 + 1]
    f, m, w = crop(fixed), crop(moving), crop(moved)
    before = normalized_mutual_information(f, m, device=device)
    after = normalized_mutual_information(f, w, device=device)
    mm = normalized_mutual_information(m, w, device=device)
    return before, after, mm, float(np.round(100 * (after - before) / before, 2))


def overlap_metrics(fixed_seg, seg, device="cuda"):
    """Dice, Jaccard, sensitivity, precision, specificity, accuracy of ``seg`` against ``fixed_seg`` (binary masks)."""
    f, m = _dev64(fixed_seg, device).reshape(-1).contiguous(), _dev64(seg, device).reshape(-1).contiguous()
    lib = _lib.load()
    ws = _ws(lib.mmr_overlap_ws_bytes(), f.device)
    out = torch.empty(6, dtype=torch.float64, device=f.device)
    _lib.check(lib.mmr_overlap_sums_f64(f.data_ptr(), m.data_ptr(), out.data_ptr(), ws.data_ptr(), f.numel(), _stream()),
               "mmr_overlap_sums_f64")
    tp, fp, n1, n0, sm, n = (float(v) for v in out.cpu())
    tn, fn = n0 - fp, n1 - tp
    return dict(dice=2 * tp / (tp + tp + fp + fn), jaccard=tp / (tp + fp + fn), sensitivity=tp / (tp + fn),
                precision=tp / sm, specificity=tn / (tn + fp), accuracy=(tp + tn) / n)
