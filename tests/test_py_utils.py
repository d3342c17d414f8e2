"""NIfTI / volume file helpers (host side, no GPU)."""
import gzip
import struct

import numpy as np
import pytest


def test_nifti_roundtrip(tmp_path):
    from mmr import py_utils as U
    rng = np.random.default_rng(0)
    aff = np.array([[0.0, 0, 2.0, -10], [-1.5, 0, 0, 20], [0, 1.0, 0, 5], [0, 0, 0, 1]])
    for dt in (np.uint8, np.int16, np.float32, np.float64):
        arr = (rng.random((5, 6, 7)) * 100).astype(dt)
        for ext in (".nii", ".nii.gz"):
            p = str(tmp_path / f"v_{np.dtype(dt).name}{ext}")
            U.save_volfile(arr, p, aff)
            got, gaff = U.load_volfile(p, ret_affine=True)
            assert got.dtype == dt and np.array_equal(got, arr)
            np.testing.assert_allclose(gaff, aff, atol=1e-6)
    field = rng.standard_normal((4, 5, 6, 1, 3)).astype(np.float32)  # the reference's 5-D warp (3d_reg.py:412-419)
    p = str(tmp_path / "warp.nii.gz")
    U.write_nifti(field, p, aff, intent_code=1007)
    data, _, hdr = U.read_nifti(p)
    assert hdr["intent_code"] == 1007 and data.shape == field.shape and np.array_equal(data, field)
    v = U.load_volfile(p, add_batch_axis=True)
    assert v.shape == (1, 4, 5, 6, 3)  # squeezed, then batch axis
    v2 = U.load_volfile(str(tmp_path / "v_float32.nii.gz"), add_batch_axis=True, add_feat_axis=True)
    assert v2.shape == (1, 5, 6, 7, 1)


def test_nifti_header_layout_and_scaling(tmp_path):
    """Hand-built big-endian file with scl_slope/inter and a qform-only orientation."""
    from mmr import py_utils as U
    h = bytearray(348)
    struct.pack_into(">i", h, 0, 348)
    struct.pack_into(">8h", h, 40, 3, 2, 3, 4, 1, 1, 1, 1)
    struct.pack_into(">hhh", h, 68, 0, 4, 16)  # int16
    struct.pack_into(">8f", h, 76, 1.0, 2.0, 3.0, 4.0, 1, 1, 1, 1)
    struct.pack_into(">fff", h, 108, 352.0, 0.5, 10.0)
    struct.pack_into(">hh", h, 252, 1, 0)  # qform only
    struct.pack_into(">6f", h, 256, 0.0, 0.0, 0.0, 7.0, 8.0, 9.0)  # identity rotation, offsets
    h[344:348] = b"n+1\x00"
    data = np.arange(24, dtype=">i2")
    p = tmp_path / "be.nii.gz"
    with gzip.open(p, "wb") as f:
        f.write(bytes(h) + b"\0\0\0\0" + data.tobytes())
    vol, aff, hdr = U.read_nifti(str(p))
    assert vol.shape == (2, 3, 4)
    np.testing.assert_allclose(vol, data.astype(np.float64).reshape((2, 3, 4), order="F") * 0.5 + 10)
    np.testing.assert_allclose(aff, [[2, 0, 0, 7], [0, 3, 0, 8], [0, 0, 4, 9], [0, 0, 0, 1]])


def test_load_labels_and_setup_device(tmp_path):
    from mmr import py_utils as U
    rng = np.random.default_rng(1)
    for i in range(3):
        U.save_volfile(rng.integers(0, 5, (4, 4, 4)).astype(np.uint8), str(tmp_path / f"label_map_{i}.nii.gz"), np.eye(4))
    np.save(tmp_path / "extra.npy", rng.integers(0, 7, (4, 4, 4)).astype(np.uint8))
    labels, maps = U.load_labels(str(tmp_path))
    assert len(maps) == 4 and labels.max() <= 6 and maps[0].shape == (4, 4, 4)
    np.save(tmp_path / "bad.npy", np.zeros((3, 3, 3), np.uint8))
    with pytest.raises(ValueError):
        U.load_labels(str(tmp_path))
    with pytest.raises(RuntimeError):
        U.setup_device("-1")
    dev, nb = U.setup_device("0")
    assert dev == "cuda:0" and nb == 1
