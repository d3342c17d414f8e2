"""Generates tests/golden/resample_nib_grid.npz by calling the reference's own ``resample_nib`` (3d_reg.py:19-117).

Run in the build container only (needs /root/reference):  python -B tests/golden/make_golden_resample.py

What is executed from the reference: the shape / affine arithmetic of ``resample_nib`` for ``new_size_type='mm'`` (the only
form the reference calls: 3d_reg.py:135-139 with new_size=[1, 1, 1]) and for a destination image (``image_dest``, the moving
volume's call) -- ``shape_r = round(shape * zoom / new_size)``, ``R = diag(shape / shape_r)``, ``affine_r = affine . R``, the
interpolation-order table and the ``mode`` it hands on.  DECLARED STUBS (none of them arithmetic under test):
  * tensorflow / voxelmorph / neurite / nibabel / nilearn modules: MagicMock;
  * ``nib.nifti1.Nifti1Image``: an in-memory image class whose header reports ``get_zooms()`` = the column norms of the
    affine (what a consistent NIfTI header holds) and ``get_data_shape()``;
  * ``resample_from_to``: records ``(to_vox_map, order, mode, cval)`` and returns the input -- the spline resampling itself is
    nibabel's / scipy's, not the reference's.
Only inputs and recorded outputs are stored (data, not source)."""
import importlib.util
import os
import sys
from unittest import mock

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
for name in ["tensorflow", "tensorflow.keras", "tensorflow.keras.backend", "voxelmorph", "neurite", "nibabel", "nibabel.processing",
             "nilearn", "nilearn.image", "matplotlib", "matplotlib.pyplot", "tqdm", "losses"]:
    sys.modules.setdefault(name, mock.MagicMock())


class FakeHeader:
    def __init__(self, shape, affine):
        self._shape, self._zooms = tuple(shape), tuple(float(z) for z in np.sqrt((np.asarray(affine)[:3, :3] ** 2).sum(0)))

    def get_zooms(self):
        return self._zooms

    def get_data_shape(self):
        return self._shape


class FakeNii:
    def __init__(self, shape, affine):
        self.shape, self.ndim = tuple(shape), len(shape)
        self.affine = np.array(affine, dtype=np.float64)
        self.header = FakeHeader(shape, affine)


def main():
    spec = importlib.util.spec_from_file_location("ref_3d_reg", os.path.join(REF, "3d_reg.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.nib.nifti1.Nifti1Image = FakeNii
    rec = {}

    def fake_resample_from_to(img, to_vox_map=None, order=None, mode=None, cval=None, out_class=None):
        rec["to"], rec["order"], rec["mode"], rec["cval"] = to_vox_map, order, mode, cval
        return img
    m.resample_from_to = fake_resample_from_to

    rng = np.random.default_rng(3)

    def affine_of(zooms, perm=(0, 1, 2), signs=(1, 1, 1), origin=(0, 0, 0), shear=0.0):
        a = np.zeros((4, 4))
        for j in range(3):
            a[perm[j], j] = signs[j] * zooms[j]
        a[:3, :3] += shear * rng.standard_normal((3, 3))     # oblique acquisitions
        a[:3, 3] = origin
        a[3, 3] = 1
        return a
    cases = [  # (shape, affine, new_size, interpolation, mode)
        ((40, 52, 17), affine_of((0.8, 0.8, 2.5)), [1, 1, 1], "linear", "constant"),
        ((33, 31, 64), affine_of((1.2, 0.9, 1.0), (2, 0, 1), (-1, 1, -1), (12.5, -40, 7)), [1, 1, 1], "linear", "constant"),
        ((25, 25, 25), affine_of((1.0, 1.0, 1.0)), [1, 1, 1], "nn", "nearest"),
        ((64, 48, 20), affine_of((0.5, 0.5, 3.0), (1, 0, 2), (1, -1, 1), (-3, 9, 100), shear=0.02), [1, 1, 1], "spline", "constant"),
        ((19, 23, 29), affine_of((0.7, 1.3, 0.45), shear=0.05), [2, 2, 2], "linear", "nearest"),
        ((30, 30, 12), affine_of((0.9375, 0.9375, 4.4)), [0.5, 1, 1.5], "linear", "constant"),
        ((21, 17, 9), affine_of((1.5, 1.5, 1.5), (0, 2, 1), (-1, -1, 1)), [1], "linear", "constant"),     # one value = isotropic
    ]
    out = {"n_cases": np.array(len(cases))}
    for i, (shape, aff, new_size, interp, mode) in enumerate(cases):
        m.resample_nib(FakeNii(shape, aff), new_size=list(new_size), new_size_type="mm", interpolation=interp, mode=mode)
        shape_r, affine_r = rec["to"]
        out[f"c{i}_shape"], out[f"c{i}_affine"] = np.array(shape), aff
        out[f"c{i}_new_size"] = np.array(new_size, dtype=np.float64)
        out[f"c{i}_interp"], out[f"c{i}_mode"] = np.array(interp), np.array(mode)
        out[f"c{i}_shape_r"], out[f"c{i}_affine_r"] = np.array(shape_r), np.array(affine_r)
        out[f"c{i}_order"], out[f"c{i}_mode_passed"], out[f"c{i}_cval"] = np.array(rec["order"]), np.array(rec["mode"]), np.array(rec["cval"])
    # the destination-image form (3d_reg.py:138-139): the reference hands the destination object itself on
    dest = FakeNii((10, 11, 12), affine_of((1, 1, 1)))
    m.resample_nib(FakeNii((5, 6, 7), affine_of((2, 2, 2))), image_dest=dest, interpolation="linear", mode="constant")
    out["dest_is_passed_through"] = np.array(rec["to"] is dest)
    out["dest_order"] = np.array(rec["order"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resample_nib_grid.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
