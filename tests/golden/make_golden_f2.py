"""Generates tests/golden/host_preproc_rai.npz by DRIVING the reference's own ``bids_registration.register()``.

Run in the build container only (needs /root/reference):  python -B tests/golden/make_golden_f2.py

What is executed from the reference (bids_registration.py): ``preprocess`` lines 127-159 -- min-max scaling, the
lexicographic ``max(fx_shape, mov_shape)``, the ``int(np.ceil(s // 16)) * 16`` shape rule and the crop / pad call --
and ``register`` lines 274-429 on the whole-volume / linear branch: the ``_proc`` side files, the moved image, the
RAI permutation + sign table of the saved warp (lines 394-425) and the intent code.  Everything the image lacks is a
DECLARED STUB, none of it is arithmetic under test:
  * tensorflow / voxelmorph / neurite / nibabel / nilearn modules: MagicMock;
  * ``nib.load`` / ``nib.save`` / ``nib.Nifti1Image``: an in-memory image class (data + affine + header dict);
  * ``nib.aff2axcodes``: restated below for affines whose 3x3 part is a signed, scaled axis permutation (the only kind
    generated here): voxel axis j -> the world axis of its single non-zero entry, letter by its sign;
  * ``resample_nib``: identity (all inputs are already 1 mm isotropic); ``resample_img``: crop / zero-pad to the
    target shape (what 'continuous' resampling onto the same affine does at integer grid points);
  * ``VxmDense(...).predict``: returns the moving volume and a seeded random FULL-resolution field, so that
    ``scale == 1``; ``rescale_dense_transform`` / ``K.eval``: identity.
Only inputs and outputs are stored (data, not source)."""
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
        self.affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
        self.header = {}

    @property
    def shape(self):
        return self._d.shape

    def get_fdata(self):
        return self._d


def fake_resample_img(img, target_affine=None, target_shape=None, interpolation=None):
    d = img.get_fdata()
    out = np.zeros(tuple(target_shape) + d.shape[3:])
    sl = tuple(slice(0, min(a, b)) for a, b in zip(d.shape[:3], target_shape))
    out[sl] = d[sl]
    return FakeNii(out, target_affine)


def stub_aff2axcodes(aff):
    rzs = np.asarray(aff)[:3, :3]
    labels = (("L", "R"), ("P", "A"), ("I", "S"))
    out = []
    for j in range(3):
        nz = np.nonzero(rzs[:, j])[0]
        assert len(nz) == 1, "stub covers signed, scaled axis permutations only"
        out.append(labels[nz[0]][1 if rzs[nz[0], j] > 0 else 0])
    return tuple(out)


def signed_perm(perm, signs, origin):
    a = np.zeros((4, 4))
    for j, (i, s) in enumerate(zip(perm, signs)):
        a[i, j] = s
    a[:3, 3] = origin
    a[3, 3] = 1
    return a


def main():
    br = load("bids_registration.py", "ref_bids_registration")
    rng = np.random.default_rng(7)
    files, saved, volfiles = {}, {}, {}
    br.nib.load = lambda p: files[p]
    br.nib.save = lambda img, p: saved.__setitem__(p, img)
    br.nib.Nifti1Image = FakeNii
    br.nib.aff2axcodes = stub_aff2axcodes
    br.resample_nib = lambda image, **kw: image
    br.resample_img = fake_resample_img
    br.K.eval = lambda x: np.asarray(x)
    br.vxm.utils.rescale_dense_transform = lambda w, s, interp_method=None: w
    br.vxm.py.utils.save_volfile = lambda arr, p, aff: volfiles.__setitem__(p, (np.asarray(arr), aff))
    state = {}

    class StubModel:
        def __init__(self, **kw):
            state["inshape"] = tuple(kw["inshape"])

        def set_weights(self, w):
            pass

        def get_weights(self):
            return []

        def predict(self, inputs):
            mov = np.asarray(inputs[0])
            state["predict_moving"] = mov
            state["predict_fixed"] = np.asarray(inputs[1])
            return mov.copy(), state["field"][None].copy()   # the reference permutes the returned array IN PLACE
    br.vxm.networks.VxmDense = StubModel

    specs = dict(warp_interpolation="linear", resample_interpolation="linear", use_subvol=False, subvol_size=[32, 32, 32],
                 min_perc_overlap=0.1, int_steps=5, int_res=2, svf_res=2, enc=[16] * 4, dec=[16] * 6)
    cases = [  # (fixed shape, moving shape, fixed affine, moving affine)
        ((21, 17, 34), (21, 17, 34), np.eye(4), np.eye(4)),                                                  # RAS+
        ((33, 19, 16), (20, 35, 40), signed_perm((0, 1, 2), (-1, -1, 1), (90, 120, -70)), np.eye(4)),        # LPS; max() lexicographic
        ((17, 36, 31), (17, 40, 16), signed_perm((2, 0, 1), (1, 1, -1), (3, -4, 5)), signed_perm((0, 1, 2), (1, -1, 1), (0, 0, 0))),
        ((20, 33, 18), (19, 50, 50), signed_perm((1, 2, 0), (-1, 1, 1), (0, 0, 0)), np.eye(4)),
        ((32, 17, 23), (32, 17, 23), signed_perm((0, 2, 1), (1, -1, -1), (10, 20, 30)), signed_perm((0, 2, 1), (1, -1, -1), (10, 20, 30))),
        ((16, 32, 16), (16, 32, 16), -np.eye(4) + 2 * np.diag([0, 0, 0, 1.0]), np.eye(4)),                   # LPI
    ]
    out = {"n_cases": np.array(len(cases))}
    for i, (fs, ms, fa, ma) in enumerate(cases):
        fx = rng.integers(-200, 1200, fs).astype(np.float64)      # integer-valued: the fixture compresses
        mv = rng.integers(3, 90, ms).astype(np.float64) / 4
        files[f"fx{i}.nii.gz"] = FakeNii(fx, fa)
        files[f"mv{i}.nii.gz"] = FakeNii(mv, ma)
        new_shape = tuple(int(np.ceil(s // 16)) * 16 for s in max(fs, ms))   # only to size the stub's field
        state["field"] = rng.integers(-64, 64, new_shape + (3,)).astype(np.float64) / 16
        saved.clear()
        volfiles.clear()
        br.register(specs, StubModel(inshape=(16, 16, 16)), f"fx{i}.nii.gz", f"mv{i}.nii.gz", fx_contrast="T1w")
        out[f"c{i}_fixed"], out[f"c{i}_moving"] = fx.astype(np.float32), mv.astype(np.float32)   # exact: small integers / 4
        out[f"c{i}_fixed_affine"], out[f"c{i}_moving_affine"] = fa, ma
        out[f"c{i}_field"] = state["field"].astype(np.float32)
        out[f"c{i}_fixed_proc"] = saved[f"fx{i}_proc.nii.gz"].get_fdata()
        out[f"c{i}_moving_proc"] = saved[f"mv{i}_proc.nii.gz"].get_fdata()
        out[f"c{i}_model_inshape"] = np.array(state["inshape"])
        w = saved[f"mv{i}_proc_field_to_T1w.nii.gz"]
        out[f"c{i}_warp_rai"] = w.get_fdata().astype(np.float32)
        out[f"c{i}_warp_intent"] = np.array(int(w.header["intent_code"]))
        out[f"c{i}_warp_affine"] = w.affine
        wo = saved[f"mv{i}_warp_original_dim.nii.gz"]
        out[f"c{i}_warp_original_shape"] = np.array(wo.get_fdata().shape)
        out[f"c{i}_warp_original_intent"] = np.array(int(wo.header["intent_code"]))
        out[f"c{i}_moved_original_shape"] = np.array(saved[f"mv{i}_reg_original_dim.nii.gz"].get_fdata().shape)
        moved, aff = volfiles[f"mv{i}_proc_reg_to_T1w.nii.gz"]
        out[f"c{i}_moved_saved_equals_moving_proc"] = np.array(bool(np.array_equal(moved, saved[f"mv{i}_proc.nii.gz"].get_fdata())))
        out[f"c{i}_predict_moving_shape"] = np.array(state["predict_moving"].shape)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_preproc_rai.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
