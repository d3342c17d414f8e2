"""Pair registration flow of the reference's inference scripts, on top of the device operators.

Mirrors 3d_reg.py:120-211,262-425, bids_registration.py:274-429 and
bids_two_steps_registration.py:274-546: min-max scaling, 1 mm resampling, crop to a multiple of 16,
whole-volume or sub-volume prediction with pyramid-weighted fusion, optional two-model cascade with
field composition, nearest / linear warping, resampling back to the moving image's grid and the
RAI-ordered 5-D warp (intent 1007) the Spinal Cord Toolbox expects.

Host-side resampling uses scipy.ndimage (nibabel / nilearn are not available here; their calls are
restated from their documented behaviour: ``resample_from_to`` = affine_transform with the
voxel-to-voxel matrix, ``resample_img(..., 'continuous')`` = cubic-spline affine_transform with zero
fill) -- UNPINNED against the originals; the tiling / fusion arithmetic it relies on is golden-pinned
(mmr.tiling).  Everything between pre- and post-processing runs through the HIP operators.
"""
import numpy as np
from scipy import ndimage

from . import networks, ops, tiling, utils

_ORDER = {"nn": 0, "nearest": 0, "linear": 1, "spline": 2}


class Volume:
    """Minimal stand-in for a nibabel image: data + 4x4 voxel-to-world affine."""

    def __init__(self, data, affine):
        self.data = np.asarray(data)
        self.affine = np.asarray(affine, dtype=np.float64)

    @property
    def shape(self):
        return self.data.shape

    def get_fdata(self):
        return self.data.astype(np.float64)

    @classmethod
    def load(cls, path):
        from .py_utils import read_nifti
        data, aff, _ = read_nifti(path)
        return cls(data, aff)

    def save(self, path, intent_code=0):
        from .py_utils import write_nifti
        write_nifti(self.data, path, self.affine, intent_code=intent_code)


def resample_from_to(vol, to_shape, to_affine, order=1, mode="constant", cval=0.0):
    """nibabel.processing.resample_from_to for 3-D data (extra trailing axes are resampled one by one)."""
    vox2vox = np.linalg.inv(vol.affine) @ np.asarray(to_affine, dtype=np.float64)
    mat, off = vox2vox[:3, :3], vox2vox[:3, 3]
    data = vol.get_fdata()
    if data.ndim == 3:
        out = ndimage.affine_transform(data, mat, off, output_shape=tuple(to_shape[:3]), order=order, mode=mode, cval=cval)
    else:
        flat = data.reshape(data.shape[:3] + (-1,))
        out = np.stack([ndimage.affine_transform(flat[..., k], mat, off, output_shape=tuple(to_shape[:3]), order=order,
                                                 mode=mode, cval=cval) for k in range(flat.shape[-1])], -1)
        out = out.reshape(tuple(to_shape[:3]) + data.shape[3:])
    return Volume(out, to_affine)


def resample_grid_mm(shape, affine, new_mm=(1, 1, 1)):
    """Target grid of ``resample_nib(..., new_size=new_mm, new_size_type='mm')`` (3d_reg.py:56-90): ``shape_r = round(shape *
    zoom / new_mm)`` with the zooms the header reports (the column norms of the affine), ``affine_r = affine . diag(shape /
    shape_r)``; one value = isotropic.  Pinned by tests/golden/resample_nib_grid.npz (the reference's own function, recorded)."""
    affine = np.asarray(affine, dtype=np.float64)
    zooms = np.sqrt((affine[:3, :3] ** 2).sum(0))
    new_mm = tuple(new_mm)
    if len(new_mm) == 1:
        new_mm = new_mm * 3
    shape = tuple(shape[:3])
    shape_r = tuple(int(np.round(shape[i] * float(zooms[i]) / float(new_mm[i]))) for i in range(3))
    if min(shape_r) < 1:
        raise ZeroDivisionError("Destination size is zero; check the NIfTI pixdim values")
    R = np.eye(4)
    for i in range(3):
        R[i, i] = shape[i] / float(shape_r[i])
    return shape_r, affine @ R


def resample_mm(vol, new_mm=(1, 1, 1), interpolation="linear", mode="constant"):
    """``resample_nib(..., new_size=[1,1,1], new_size_type='mm')`` (3d_reg.py:19-117)."""
    shape_r, affine_r = resample_grid_mm(vol.shape, vol.affine, new_mm)
    return resample_from_to(vol, shape_r, affine_r, order=_ORDER[interpolation], mode=mode)


def resample_img(vol, target_affine, target_shape, order=3):
    """nilearn.image.resample_img(..., interpolation='continuous'): spline resampling, zeros outside."""
    return resample_from_to(vol, target_shape, target_affine, order=order, mode="constant", cval=0.0)


def _minmax(a):
    a = np.asarray(a, dtype=np.float64)
    return (a - a.min()) / (a.max() - a.min())


def preprocess(specs, fixed, moving, resample_interp="linear"):
    """3d_reg.py:120-211 -> (fixed_proc, moving_proc, fixed tiles, moving tiles, tile coordinates)."""
    fx = Volume(_minmax(fixed.get_fdata()), fixed.affine)
    mv = Volume(_minmax(moving.get_fdata()), moving.affine)
    fx_r = resample_mm(fx, (1, 1, 1), resample_interp, mode="constant")
    mv_r = resample_from_to(mv, fx_r.shape, fx_r.affine, order=_ORDER[resample_interp], mode="constant")
    new_shape = tiling.round_down_16(max(fx_r.shape, mv_r.shape))  # lexicographic tuple max, floor
    fx_r = resample_img(fx_r, fx_r.affine, new_shape)
    mv_r = resample_img(mv_r, mv_r.affine, new_shape)
    tiles_fx, tiles_mv, coords = [], [], []
    if specs["use_subvol"]:
        _, coords = tiling.subvolume_grid(new_shape, specs["subvol_size"], specs["min_perc_overlap"])
        tiles_fx = tiling.extract_subvolumes(fx_r.data, coords)
        tiles_mv = tiling.extract_subvolumes(mv_r.data, coords)
    return fx_r, mv_r, tiles_fx, tiles_mv, coords


def axcodes(affine):
    """Closest world axis per voxel axis ('R','A','S' / 'L','P','I'), like nibabel.aff2axcodes for
    near-axis-aligned affines (column-wise arg-max of |RZS|)."""
    rzs = np.asarray(affine)[:3, :3]
    labels = (("L", "R"), ("P", "A"), ("I", "S"))
    codes, used = [], set()
    for j in range(3):
        col = np.abs(rzs[:, j]).copy()
        for u in used:
            col[u] = -1
        i = int(np.argmax(col))
        used.add(i)
        codes.append(labels[i][1 if rzs[i, j] > 0 else 0])
    return codes


def to_rai_warp(warp_full, fixed_affine):
    """Reorder / sign-flip the vector components for sct_apply_transfo (3d_reg.py:399-417) -> [X,Y,Z,1,3]."""
    orient = axcodes(-np.asarray(fixed_affine))
    opposite = {"L": "R", "R": "L", "A": "P", "P": "A", "I": "S", "S": "I"}
    perm, inv = [0, 1, 2], [1, 1, 1]
    for i, ch in enumerate("RAI"):
        if ch in orient:
            perm[i] = orient.index(ch)
        else:
            perm[i] = orient.index(opposite[ch])
            inv[i] = -1
    w = np.asarray(warp_full)
    return np.stack([inv[k] * w[..., perm[k]] for k in range(3)], -1)[:, :, :, None, :]


def _build(models, specs, shape, compute_dtype):
    out = []
    for m in models:
        net = networks.VxmDense(shape, int_steps=specs["int_steps"], int_resolution=specs["int_res"],
                                svf_resolution=specs["svf_res"], nb_unet_features=(specs["enc"], specs["dec"]),
                                compute_dtype=compute_dtype, device=m.device)
        net.set_weights(m.get_weights())  # 3d_reg.py:305-306: rebuild at the runtime shape, transplant weights
        out.append(net)
    return out


def _pair(a, b):
    return [a[None, ..., None], b[None, ..., None]]


def register(specs, models, fixed, moving, warp_interp="linear", resample_interp="linear", compute_dtype="fp32x3"):
    """Register ``moving`` to ``fixed`` (Volume objects) with one model (3d_reg.py / bids_registration.py) or a
    cascade of two (bids_two_steps_registration.py).  Returns dict(fixed_proc, moving_proc, moved, moved_original,
    warp (half-res field as the reference keeps it), warp_rai, warp_rai_original, scale)."""
    models = list(models) if isinstance(models, (list, tuple)) else [models]
    if warp_interp not in ("nearest", "linear"):
        warp_interp = "linear"
    if resample_interp not in ("nearest", "linear", "spline"):
        resample_interp = "linear"
    fx, mv, tiles_fx, tiles_mv, coords = preprocess(specs, fixed, moving, "nn" if resample_interp == "nearest" else resample_interp)
    in_shape = tiling.round_down_16(specs["subvol_size"]) if specs["use_subvol"] else fx.shape
    if specs["use_subvol"] and any(t > s for t, s in zip(in_shape, fx.shape)):
        raise ValueError(f"sub-volume size {in_shape} exceeds the preprocessed volume {fx.shape} (SURVEY B5)")
    nets = _build(models, specs, in_shape, compute_dtype)
    mv_data, fx_data = mv.get_fdata(), fx.get_fdata()

    def predict_field(net, moving_arr, fixed_arr, mt, ft):
        """-> (moved by the net (whole volume only), half/full-res field at volume scale, scale)."""
        if not specs["use_subvol"]:
            moved, warp = net.predict(_pair(moving_arr, fixed_arr))
            w = warp[0]
            return moved[0, ..., 0], w, (1 if w.shape[0] == in_shape[0] else 2)
        fields = [net.predict(_pair(m, f))[1][0] for f, m in zip(ft, mt)]
        half = fields[0].shape[0] != in_shape[0]
        sc = 2 if half else 1
        t_shape = tuple(s // sc for s in in_shape)
        v_shape = tuple(s // sc for s in moving_arr.shape)
        cds = [tuple(c // sc for c in cd) for cd in coords]
        return None, tiling.fuse_subvolume_fields(t_shape, v_shape, cds, fields), sc

    def apply(arr, field, sc):
        return networks.Transform(arr.shape, interp_method=warp_interp, rescale=sc, nb_feats=1,
                                  device=nets[0].device).predict([arr[None, ..., None], field[None]])[0, ..., 0]

    if len(nets) == 2 and specs["use_subvol"] and warp_interp == "linear":
        # bids_two_steps_registration.py:362-404: per tile, model 2 runs on model 1's OWN moved tile and the two
        # half-res fields are composed per tile; only then are the composed tile fields fused and applied once
        fields = []
        for f, m in zip(tiles_fx, tiles_mv):
            moved_t, w1 = nets[0].predict(_pair(m, f))
            _, w2 = nets[1].predict(_pair(moved_t[0, ..., 0], f))
            fields.append(np.asarray(utils.compose([np.asarray(w1[0], np.float32), np.asarray(w2[0], np.float32)])))
        scale = 2 if fields[0].shape[0] != in_shape[0] else 1
        warp = tiling.fuse_subvolume_fields(tuple(s // scale for s in in_shape), tuple(s // scale for s in mv_data.shape),
                                            [tuple(c // scale for c in cd) for cd in coords], fields)
        moved = apply(mv_data, warp, scale)
    else:
        moved1, warp, scale = predict_field(nets[0], mv_data, fx_data, tiles_mv, tiles_fx)
        if warp_interp != "linear" or moved1 is None:
            moved1 = apply(mv_data, warp, scale)
        moved = moved1
        if len(nets) == 2:
            # whole volume (:320-358), or sub-volumes with nearest interpolation (:406-470): stage 1 is fused and
            # applied to the whole volume, stage 2 runs on (tiles of) that moved volume, the fields compose globally
            tiles_m1 = tiling.extract_subvolumes(moved1, coords) if specs["use_subvol"] else []
            moved2, warp2, _ = predict_field(nets[1], moved1, fx_data, tiles_m1, tiles_fx)
            warp = utils.compose([np.asarray(warp, np.float32), np.asarray(warp2, np.float32)])  # first, then second
            if warp_interp == "linear" and moved2 is not None:
                moved = moved2
            else:
                moved = apply(mv_data, warp, scale)
    moved_vol = Volume(moved, fx.affine)
    moved_orig = resample_img(moved_vol, moving.affine, moving.shape[:3])
    full = utils.rescale_dense_transform(np.asarray(warp, np.float32)[None], scale)[0]
    warp_rai = Volume(to_rai_warp(full, fixed.affine), fx.affine)
    warp_orig = resample_img(warp_rai, moving.affine, moving.shape[:3])
    return dict(fixed_proc=fx, moving_proc=mv, moved=moved_vol, moved_original=moved_orig, warp=np.asarray(warp),
                warp_rai=warp_rai, warp_rai_original=warp_orig, scale=scale)


def run_3d_reg(specs, model_path, fx_im_path, mov_im_path, res_dir="res", warp_interp="linear", resample_interp="linear",
               out_im_path="warped_im", out_field_path="deform_field", compute_dtype="fp32x3", model_path_2=None):
    """File-level entry with 3d_reg.py's arguments; unlike the reference (NameError at 3d_reg.py:421)
    it also writes the deformation field."""
    import os
    models = [networks.VxmDense.load(model_path, input_model=None, compute_dtype=compute_dtype)]
    if model_path_2:
        models.append(networks.VxmDense.load(model_path_2, input_model=None, compute_dtype=compute_dtype))
    os.makedirs(res_dir, exist_ok=True)
    out = register(specs, models, Volume.load(fx_im_path), Volume.load(mov_im_path), warp_interp, resample_interp,
                   compute_dtype)
    out["moved_original"].save(os.path.join(res_dir, f"{out_im_path}.nii.gz"))
    out["warp_rai_original"].save(os.path.join(res_dir, f"{out_field_path}.nii.gz"), intent_code=1007)
    return out
