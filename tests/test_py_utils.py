"""NIfTI / volume file helpers (host side, no GPU)."""
import gzip
import os
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


def test_mgz_label_maps(tmp_path):
    """FreeSurfer .mgz (what ``vxm.py.utils.load_labels`` reads the public SynthMorph label maps from, train_synthmorph.py:207):
    the reader against a byte stream assembled here field by field from the published MGH layout (big-endian; 284-byte
    header: version, dims, type, dof, goodRASFlag, spacing, direction cosines column by column, centre; x-fastest data) -- not
    only against this package's own writer -- including the vox2ras affine nibabel derives (M = Mdc * spacing,
    P0 = c_ras - M . dims / 2).  No FreeSurfer-written file exists in this image; the layout is the format's definition."""
    import gzip
    import struct
    from mmr import py_utils as U
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 26, (5, 7, 3)).astype(np.uint8)
    spacing = (0.5, 2.0, 1.25)
    mdc_cols = np.array([[0.0, 0.0, -1.0], [-1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])     # x_ras, y_ras, z_ras
    c_ras = (10.0, -4.5, 3.0)
    hd = bytearray(284)
    struct.pack_into(">7i", hd, 0, 1, 5, 7, 3, 1, 0, 0)
    struct.pack_into(">h", hd, 28, 1)
    struct.pack_into(">3f", hd, 30, *spacing)
    struct.pack_into(">9f", hd, 42, *mdc_cols.reshape(-1))
    struct.pack_into(">3f", hd, 78, *c_ras)
    path = str(tmp_path / "seg_01.mgz")
    with gzip.open(path, "wb") as f:
        f.write(bytes(hd) + lab.tobytes(order="F"))
    vol, aff = U.load_volfile(path, ret_affine=True)
    assert vol.dtype == np.uint8 and np.array_equal(vol, lab)
    M = mdc_cols.T * np.array(spacing)
    assert np.allclose(aff[:3, :3], M) and np.allclose(aff[:3, 3], np.array(c_ras) - M @ (np.array([5, 7, 3]) / 2)) and aff[3, 3] == 1
    # every MGH voxel type, the writer / reader pair, frames, and the unflagged default orientation
    for dt in (np.uint8, np.int16, np.int32, np.float32):
        a = (rng.standard_normal((4, 3, 6)) * 50).astype(dt)
        p = str(tmp_path / f"v_{np.dtype(dt).name}.mgz")
        U.save_volfile(a, p, aff)
        b, aff2 = U.load_volfile(p, ret_affine=True)
        assert b.dtype == dt and np.array_equal(a, b) and np.allclose(aff2, aff, atol=1e-5)
    four = rng.integers(0, 9, (3, 4, 5, 2)).astype(np.int32)
    U.write_mgh(four, str(tmp_path / "f.mgh"))
    assert np.array_equal(U.read_mgh(str(tmp_path / "f.mgh"))[0], four)
    struct.pack_into(">h", hd, 28, 0)
    with open(str(tmp_path / "noras.mgh"), "wb") as f:
        f.write(bytes(hd) + lab.tobytes(order="F"))
    _, aff0, h0 = U.read_mgh(str(tmp_path / "noras.mgh"))
    assert np.allclose(aff0[:3, :3], [[-1, 0, 0], [0, 0, 1], [0, -1, 0]]) and h0["goodRASFlag"] == 0
    # load_labels over a folder of .mgz maps, like the reference's label_dir
    U.save_volfile(lab, str(tmp_path / "seg_02.mgz"))
    os.remove(str(tmp_path / "f.mgh")); os.remove(str(tmp_path / "noras.mgh"))
    for dt in ("uint8", "int16", "int32", "float32"):
        os.remove(str(tmp_path / f"v_{dt}.mgz"))
    labels, maps = U.load_labels(str(tmp_path))
    assert len(maps) == 2 and np.array_equal(maps[0], lab) and np.array_equal(labels, np.unique(lab))
    with pytest.raises(ValueError):
        U.read_nifti(path)   # an .mgz is not a NIfTI file
