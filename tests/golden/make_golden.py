"""Generates tests/golden/host_helpers.npz from the reference's own NumPy helpers.

Run in the build container only (needs /root/reference):  python -B tests/golden/make_golden.py
The reference's third-party imports (tensorflow, voxelmorph, neurite, nibabel, nilearn, ...) are
absent here, so they are replaced by MagicMock stubs; only pure-NumPy code paths are executed:
  * 3d_reg.get_def_field_from_subvol            (3d_reg.py:214-259)
  * the tiling arithmetic inside 3d_reg.preprocess (3d_reg.py:157-207), driven through fake
    image objects with the resampling calls stubbed to identity / crop
  * train_synthmorph.set_random_zero_borders     (train_synthmorph.py:85-114) under np.random.seed
  * train_synthmorph.gen_synthmorph_eb           (train_synthmorph.py:117-165) with default_rng seeded
Only inputs and outputs are stored (data, not source).
"""
import importlib.util
import os
import sys
from unittest import mock

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
for name in ["tensorflow", "tensorflow.keras", "tensorflow.keras.backend", "voxelmorph", "neurite", "nibabel",
             "nibabel.processing", "nilearn", "nilearn.image", "matplotlib", "matplotlib.pyplot", "tqdm", "losses"]:
    sys.modules.setdefault(name, mock.MagicMock())


def load(fname, modname):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, fname))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class FakeNii:
    def __init__(self, data, affine=None):
        self._d = np.asarray(data, dtype=np.float64)
        self.affine = np.eye(4) if affine is None else affine

    def get_fdata(self):
        return self._d


def fake_resample_img(img, target_affine=None, target_shape=None, interpolation=None):
    out = np.zeros(target_shape)
    d = img.get_fdata()
    sl = tuple(slice(0, min(a, b)) for a, b in zip(d.shape, target_shape))
    out[sl] = d[sl]
    return FakeNii(out, img.affine)


def main():
    reg = load("3d_reg.py", "ref_3d_reg")
    tr = load("train_synthmorph.py", "ref_train")
    out = {}
    rng = np.random.default_rng(0)

    # ---- tiling coordinates through the real preprocess() -------------------------------------
    reg.nib.Nifti1Image = FakeNii
    reg.resample_nib = lambda img, **kw: img
    reg.resample_img = fake_resample_img
    cases = [((160, 160, 192), (80, 80, 96), 0.1), ((208, 96, 64), (80, 48, 32), 0.1), ((80, 80, 96), (80, 80, 96), 0.1),
             ((170, 130, 100), (90, 70, 50), 0.25), ((64, 64, 64), (32, 32, 32), 20), ((64, 64, 64), (32, 32, 32), 150),
             ((64, 48, 32), (32, 32, 32), -1), ((96, 96, 96), (47, 33, 64), 0.3)]
    for i, (shape, sub, perc) in enumerate(cases):
        fx = FakeNii(rng.random(shape))
        mv = FakeNii(rng.random(shape))
        specs = dict(use_subvol=True, subvol_size=list(sub), min_perc_overlap=perc)
        fxr, mvr, lfx, lmv, coords = reg.preprocess(specs, fx, mv, "linear")
        out[f"tile{i}_shape"] = np.array(shape)
        out[f"tile{i}_sub"] = np.array(sub)
        out[f"tile{i}_perc"] = np.array(float(perc))
        out[f"tile{i}_newshape"] = np.array(fxr.get_fdata().shape)
        out[f"tile{i}_coords"] = np.array(coords, dtype=np.int64)
        out[f"tile{i}_first_subvol_shape"] = np.array(lfx[0].shape)
    out["n_tile_cases"] = np.array(len(cases))

    # ---- fusion of per-tile fields ---------------------------------------------------------------
    fuse_cases = [((8, 8, 8), (12, 12, 12), [(0, 8, 0, 8, 0, 8), (4, 12, 0, 8, 0, 8), (0, 8, 4, 12, 4, 12), (4, 12, 4, 12, 4, 12)]),
                  ((4, 6, 8), (6, 6, 13), [(0, 4, 0, 6, 0, 8), (2, 6, 0, 6, 0, 8), (0, 4, 0, 6, 4, 12), (2, 6, 0, 6, 4, 12)]),
                  ((16, 8, 4), (16, 8, 4), [(0, 16, 0, 8, 0, 4), (0, 16, 0, 8, 0, 4)])]
    for i, (tshape, imshape, coords) in enumerate(fuse_cases):
        warps = [rng.standard_normal(tshape + (3,)) for _ in coords]
        res = reg.get_def_field_from_subvol(tshape, imshape, coords, warps)
        out[f"fuse{i}_tshape"] = np.array(tshape)
        out[f"fuse{i}_imshape"] = np.array(imshape)
        out[f"fuse{i}_coords"] = np.array(coords, dtype=np.int64)
        out[f"fuse{i}_warps"] = np.stack(warps)
        out[f"fuse{i}_out"] = res
    out["n_fuse_cases"] = np.array(len(fuse_cases))

    # ---- random zero borders under a seeded global RNG --------------------------------------------
    zb_cases = [((16, 12, 20, 1), 8, 0), ((16, 12, 20, 1), 4, 1), ((32, 32, 1), 8, 2), ((24, 24, 24, 1), 2, 3)]
    for i, (shape, scale, seed) in enumerate(zb_cases):
        im = rng.integers(1, 26, shape).astype(np.uint8)
        np.random.seed(seed)
        res = tr.set_random_zero_borders(im, scale)
        out[f"zb{i}_in"] = im
        out[f"zb{i}_scale"] = np.array(scale)
        out[f"zb{i}_seed"] = np.array(seed)
        out[f"zb{i}_out"] = res
    out["n_zb_cases"] = np.array(len(zb_cases))

    # ---- batch generator with default_rng seeded -----------------------------------------------------
    maps = [rng.integers(0, 26, (8, 6, 10)).astype(np.uint8) for _ in range(5)]
    out["gen_maps"] = np.stack(maps)
    gen_cases = [dict(batch_size=1, same_subj=True, flip=True, random_zero_borders=False),
                 dict(batch_size=2, same_subj=False, flip=True, random_zero_borders=False),
                 dict(batch_size=2, same_subj=True, flip=False, random_zero_borders=True, scale_zero_borders=4, frac_zero_bord=0.7)]
    for i, kw in enumerate(gen_cases):
        with mock.patch.object(np.random, "default_rng", lambda: np.random.Generator(np.random.PCG64(100 + i))):
            np.random.seed(200 + i)
            g = tr.gen_synthmorph_eb(maps, **kw)
            for step in range(3):
                (src, trg), voids = next(g)
                out[f"gen{i}_s{step}_src"] = np.array(src)
                out[f"gen{i}_s{step}_trg"] = np.array(trg)
            out[f"gen{i}_void_shape"] = np.array(voids[0].shape)
            out[f"gen{i}_void_dtype"] = np.array(str(voids[0].dtype))
    out["n_gen_cases"] = np.array(len(gen_cases))

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_helpers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
