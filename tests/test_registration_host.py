"""Host-side pieces of the pair-registration flow (no GPU): resampling, orientation, RAI reordering."""
import numpy as np


def test_resampling_helpers():
    from mmr.registration import Volume, resample_from_to, resample_img, resample_mm
    rng = np.random.default_rng(0)
    aff = np.diag([2.0, 2.0, 1.0, 1.0])
    aff[:3, 3] = [5, -3, 2]
    v = Volume(rng.random((10, 12, 16)), aff)
    same = resample_from_to(v, v.shape, v.affine, order=1)
    np.testing.assert_allclose(same.data, v.data, atol=1e-12)
    r = resample_mm(v, (1, 1, 1), "linear")
    assert r.shape == (20, 24, 16)
    np.testing.assert_allclose(np.sqrt((r.affine[:3, :3] ** 2).sum(0)), [1, 1, 1])
    np.testing.assert_allclose(r.data[::2, ::2, :][:-1, :-1], v.data[:-1, :-1], atol=1e-9)  # grid points coincide
    crop = resample_img(v, v.affine, (8, 8, 16))
    np.testing.assert_allclose(crop.data, v.data[:8, :8], atol=1e-9)
    pad = resample_img(v, v.affine, (12, 12, 16))
    assert np.allclose(pad.data[10:], 0) and np.allclose(pad.data[:10], v.data, atol=1e-9)


def test_orientation_and_rai():
    from mmr.registration import axcodes, to_rai_warp
    assert axcodes(np.eye(4)) == ["R", "A", "S"]
    assert axcodes(-np.eye(4)) == ["L", "P", "I"]
    lps = np.diag([-1.0, -1.0, 1.0, 1.0])
    assert axcodes(lps) == ["L", "P", "S"]
    perm = np.array([[0, 0, 1.0, 0], [1.0, 0, 0, 0], [0, -1.0, 0, 0], [0, 0, 0, 1]])
    assert axcodes(perm) == ["A", "I", "R"]
    w = np.random.default_rng(1).standard_normal((4, 5, 6, 3))
    out = to_rai_warp(w, np.eye(4))  # RAS+ image: -affine reads LPI -> (-x, -y, +z)
    assert out.shape == (4, 5, 6, 1, 3)
    np.testing.assert_array_equal(out[..., 0, :], w * np.array([-1, -1, 1]))
    out2 = to_rai_warp(w, -np.eye(4))  # -(-I) = RAS: R,A present, I missing -> S flipped
    np.testing.assert_array_equal(out2[..., 0, :], w * np.array([1, 1, -1]))
