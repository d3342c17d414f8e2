"""Generates tests/golden/eval_metrics.npz by RUNNING the reference's evaluation scripts on synthetic inputs
(build container only; needs /root/reference):  python -B tests/golden/make_golden_eval.py

nibabel is absent, so it is replaced by a stub whose ``load`` hands back in-memory arrays; the scripts'
own arithmetic (eval_reg_with_jacobian.py:62-91, eval_reg_with_mi.py:16-74,123-139,
eval_reg_on_sc_seg.py:80-124) runs unchanged and its CSV / image outputs are captured.  Only inputs and
outputs are stored."""
import csv
import importlib.util
import os
import runpy
import sys
import tempfile
import types
from unittest import mock

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True

STORE = {}
SAVED = {}


class FakeNii:
    def __init__(self, data, affine=None):
        self._d = np.asarray(data)
        self.affine = np.eye(4) if affine is None else affine

    def get_fdata(self):
        return self._d.astype(np.float64)


nib = types.ModuleType("nibabel")
nib.load = lambda p: FakeNii(STORE[os.path.basename(str(p))])
nib.Nifti1Image = lambda data, affine: FakeNii(data, affine)
nib.save = lambda img, p: SAVED.__setitem__(os.path.basename(str(p)), np.asarray(img._d))
sys.modules["nibabel"] = nib


def run_script(name, argv):
    old = sys.argv
    sys.argv = [name] + argv
    code = None
    try:
        runpy.run_path(os.path.join(REF, name), run_name="__main__")
    except SystemExit as e:
        code = e.code
    finally:
        sys.argv = old
    return code


def read_csv(path):
    with open(path) as f:
        rows = list(csv.reader(f))
    return rows[0], rows[-1]


def main():
    rng = np.random.default_rng(0)
    out = {}
    tmp = tempfile.mkdtemp()
    # ---- Jacobian determinant ------------------------------------------------------------------
    for i, (shape, amp) in enumerate([((12, 10, 14), 0.3), ((9, 9, 9), 2.5), ((16, 8, 11), 1.0)]):
        ddf = (rng.standard_normal(shape + (1, 3)) * amp)
        STORE["ddf.nii.gz"] = ddf
        csvp = os.path.join(tmp, f"jac{i}.csv")
        code = run_script("eval_reg_with_jacobian.py", ["--def-field-path", "ddf.nii.gz", "--sub-id", "s", "--out-file", csvp,
                                                        "--out-im-path", f"det{i}.nii.gz", "--append", "0"])
        hdr, row = read_csv(csvp)
        out[f"jac{i}_ddf"] = ddf
        out[f"jac{i}_det"] = SAVED[f"det{i}.nii.gz"]
        out[f"jac{i}_stats"] = np.array([float(v) for v in row[2:]])  # perc_neg, median, mean, std, n_total, n_neg
    out["n_jac"] = np.array(3)
    # ---- NMI ------------------------------------------------------------------------------------
    spec = importlib.util.spec_from_file_location("ref_mi", os.path.join(REF, "eval_reg_with_mi.py"))
    mi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mi)
    for i in range(3):
        a = rng.random((14, 12, 10))
        b = 0.6 * a + 0.4 * rng.random((14, 12, 10)) if i else a.copy()
        if i == 2:
            b = np.round(b * 7) / 7  # repeated values / edges
        out[f"nmi{i}_a"], out[f"nmi{i}_b"] = a, b
        out[f"nmi{i}_val"] = np.array(mi.normalized_mutual_information(a, b))
    pad = np.zeros((12, 10, 9))
    pad[2:9, 1:8, 3:7] = rng.random((7, 7, 4)) + 0.1
    out["zp_im"] = pad
    out["zp_box"] = np.array(mi.detect_zero_padding(pad))
    fx, mv = rng.random((12, 10, 9)), pad
    wr = np.clip(pad + 0.05 * rng.random(pad.shape) * (pad > 0), 0, None)
    STORE.update({"fx.nii.gz": fx, "mv.nii.gz": mv, "wr.nii.gz": wr})
    csvp = os.path.join(tmp, "nmi.csv")
    run_script("eval_reg_with_mi.py", ["--fx-im-path", "fx.nii.gz", "--moving-im-path", "mv.nii.gz", "--warped-im-path",
                                       "wr.nii.gz", "--sub-id", "s", "--out-file", csvp, "--append", "0"])
    hdr, row = read_csv(csvp)
    out["nmi_script_fx"], out["nmi_script_mv"], out["nmi_script_wr"] = fx, mv, wr
    out["nmi_script_vals"] = np.array([float(v) for v in row[2:]])
    # ---- overlap metrics on binary segmentations ---------------------------------------------------
    fxs = (rng.random((10, 11, 12)) > 0.6).astype(np.float64)
    mvs = np.roll(fxs, 2, axis=0)
    wrs = np.roll(fxs, 1, axis=2)
    STORE.update({"fxs.nii.gz": fxs, "mvs.nii.gz": mvs, "wrs.nii.gz": wrs})
    csvp = os.path.join(tmp, "seg.csv")
    code = run_script("eval_reg_on_sc_seg.py", ["--fx-seg-path", "fxs.nii.gz", "--moving-seg-path", "mvs.nii.gz",
                                                "--warped-seg-path", "wrs.nii.gz", "--sub-id", "s", "--out-file", csvp,
                                                "--append", "0", "--min-dice", "0", "--last-eval", "1"])
    hdr, row = read_csv(csvp)
    out["seg_fx"], out["seg_mv"], out["seg_wr"] = fxs, mvs, wrs
    out["seg_header"] = np.array(hdr[2:])
    out["seg_vals"] = np.array([float(v) for v in row[2:]])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eval_metrics.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", list(out["seg_header"]))


if __name__ == "__main__":
    main()
