"""Host helpers vs golden vectors generated from the reference (tests/golden/make_golden.py)."""
import os

import numpy as np

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "host_helpers.npz"))


def test_tile_grid_golden():
    from mmr import tiling
    for i in range(int(G["n_tile_cases"])):
        shape16 = tiling.round_down_16(G[f"tile{i}_shape"])
        assert shape16 == tuple(G[f"tile{i}_newshape"])
        tile, coords = tiling.subvolume_grid(shape16, G[f"tile{i}_sub"], float(G[f"tile{i}_perc"]))
        assert np.array_equal(np.array(coords), G[f"tile{i}_coords"]), f"case {i}"
        vol = np.zeros(shape16)
        assert tiling.extract_subvolumes(vol, coords)[0].shape == tuple(G[f"tile{i}_first_subvol_shape"])


def test_tile_grid_known_quirks():
    """regular cases exact; 208/80 leaves voxel 207 uncovered; S == T gives two identical tiles."""
    from mmr import tiling
    assert tiling.axis_spans(160, 80, 0.1) == [(0, 80), (40, 120), (80, 160)]
    assert tiling.axis_spans(192, 96, 0.1) == [(0, 96), (48, 144), (96, 192)]
    assert tiling.axis_spans(208, 80, 0.1) == [(0, 80), (63, 143), (127, 207)]
    assert tiling.axis_spans(80, 80, 0.1) == [(0, 80), (0, 80)]
    assert tiling.round_down_16((170, 31, 16)) == (160, 16, 16)


def test_fusion_golden():
    from mmr import tiling
    for i in range(int(G["n_fuse_cases"])):
        got = tiling.fuse_subvolume_fields(tuple(G[f"fuse{i}_tshape"]), tuple(G[f"fuse{i}_imshape"]),
                                           [tuple(c) for c in G[f"fuse{i}_coords"]], list(G[f"fuse{i}_warps"]))
        np.testing.assert_allclose(got, G[f"fuse{i}_out"], rtol=1e-12, atol=1e-14)


def test_zero_borders_golden():
    from mmr import data
    for i in range(int(G["n_zb_cases"])):
        np.random.seed(int(G[f"zb{i}_seed"]))
        got = data.set_random_zero_borders(G[f"zb{i}_in"], int(G[f"zb{i}_scale"]))
        assert got.dtype == G[f"zb{i}_out"].dtype
        assert np.array_equal(got, G[f"zb{i}_out"])


def test_batch_generator_golden():
    from mmr import data
    maps = list(G["gen_maps"])
    cases = [dict(batch_size=1, same_subj=True, flip=True, random_zero_borders=False),
             dict(batch_size=2, same_subj=False, flip=True, random_zero_borders=False),
             dict(batch_size=2, same_subj=True, flip=False, random_zero_borders=True, scale_zero_borders=4,
                  frac_zero_bord=0.7)]
    assert len(cases) == int(G["n_gen_cases"])
    for i, kw in enumerate(cases):
        np.random.seed(200 + i)
        g = data.gen_synthmorph_eb(maps, rng=np.random.Generator(np.random.PCG64(100 + i)), **kw)
        for step in range(3):
            (src, trg), voids = next(g)
            assert src.dtype == np.uint8 and src.shape == (kw["batch_size"], 8, 6, 10, 1)
            assert np.array_equal(src, G[f"gen{i}_s{step}_src"]), (i, step)
            assert np.array_equal(trg, G[f"gen{i}_s{step}_trg"]), (i, step)
            if kw["same_subj"] and not kw["random_zero_borders"]:
                assert np.array_equal(src, trg)
        assert voids[0].shape == tuple(G[f"gen{i}_void_shape"]) and str(voids[0].dtype) == str(G[f"gen{i}_void_dtype"])


def test_resident_batch_generator_matches_host_generator():
    """gen_synthmorph_eb(device=...) keeps the label maps on the device and must yield exactly the host batches
    (same RNG call order); run here on the CPU device."""
    import torch
    from mmr import data
    rng = np.random.default_rng(3)
    maps = [rng.integers(0, 7, (12, 10, 14)).astype(np.uint8) for _ in range(5)]
    for same_subj, bs in ((False, 2), (True, 1)):
        np.random.seed(11)
        host = data.gen_synthmorph_eb(maps, batch_size=bs, same_subj=same_subj, flip=True, rng=np.random.default_rng(5))
        hb = [next(host) for _ in range(6)]
        np.random.seed(11)
        devg = data.gen_synthmorph_eb(maps, batch_size=bs, same_subj=same_subj, flip=True, rng=np.random.default_rng(5),
                                      device=torch.device("cpu"))
        for (hs, ht), _ in hb:
            (ds, dt_), void = next(devg)
            assert isinstance(ds, torch.Tensor) and ds.dtype == torch.uint8 and tuple(ds.shape) == hs.shape
            assert np.array_equal(ds.numpy(), hs) and np.array_equal(dt_.numpy(), ht)
            assert void[0].shape == (bs, 12, 10, 14, 3)


def test_preprocess_and_rai_warp_match_reference_register():
    """goldens captured by driving the reference's own ``bids_registration.register()`` through declared stubs
    (tests/golden/make_golden_f2.py): min-max scaling, lexicographic ``max(shape)``, the floor-to-16 crop shape
    (bids_registration.py:127-159) and the RAI permutation / sign table + intent 1007 of the saved warp (:394-425)."""
    from mmr import registration as R
    from mmr import tiling
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "host_preproc_rai.npz"))
    specs = dict(use_subvol=False, subvol_size=[32, 32, 32], min_perc_overlap=0.1)
    seen_orient = set()
    for i in range(int(g["n_cases"])):
        fx, mv = g[f"c{i}_fixed"].astype(np.float64), g[f"c{i}_moving"].astype(np.float64)
        fa, ma = g[f"c{i}_fixed_affine"], g[f"c{i}_moving_affine"]
        want_fx, want_mv = g[f"c{i}_fixed_proc"], g[f"c{i}_moving_proc"]
        # shape rule on the reference's own (possibly different) shapes: tuple max, then floor to a multiple of 16
        assert tiling.round_down_16(max(fx.shape, mv.shape)) == want_fx.shape == tuple(g[f"c{i}_model_inshape"])
        # the product's preprocess on the fixed image alone (moving := fixed grid, so no stubbed resampling is involved):
        # min-max + crop must reproduce the reference's _proc volume
        fx_r, _, _, _, _ = R.preprocess(specs, R.Volume(fx, fa), R.Volume(fx, fa), "linear")
        same_shape = tiling.round_down_16(fx.shape)
        sl = tuple(slice(0, min(a, b)) for a, b in zip(same_shape, want_fx.shape))
        np.testing.assert_allclose(fx_r.data[sl], want_fx[sl], atol=1e-9)
        np.testing.assert_allclose(R._minmax(mv)[tuple(slice(0, min(a, b)) for a, b in zip(mv.shape, want_mv.shape))],
                                   want_mv[tuple(slice(0, min(a, b)) for a, b in zip(mv.shape, want_mv.shape))], atol=1e-12)
        assert 0.0 <= want_fx.min() and want_fx.max() <= 1.0
        # RAI reordering of the (full-resolution, scale 1) field
        got = R.to_rai_warp(g[f"c{i}_field"].astype(np.float64), fa)
        assert got.shape == g[f"c{i}_warp_rai"].shape and got.shape[3] == 1
        np.testing.assert_array_equal(got.astype(np.float32), g[f"c{i}_warp_rai"])
        assert int(g[f"c{i}_warp_intent"]) == 1007 and int(g[f"c{i}_warp_original_intent"]) == 1007
        np.testing.assert_array_equal(g[f"c{i}_warp_affine"], fa)          # saved with the FIXED image's affine
        assert tuple(g[f"c{i}_warp_original_shape"])[:3] == mv.shape and tuple(g[f"c{i}_moved_original_shape"]) == mv.shape
        assert bool(g[f"c{i}_moved_saved_equals_moving_proc"])
        seen_orient.add("".join(R.axcodes(-fa)))
    assert len(seen_orient) >= 5   # RAS, LPS, LPI and three axis permutations


def test_resample_grid_matches_reference_resample_nib():
    """registration.resample_grid_mm / the interpolation-order table against the reference's own ``resample_nib``
    (3d_reg.py:19-117; recorded by tests/golden/make_golden_resample.py with ``resample_from_to`` stubbed): anisotropic,
    permuted / flipped and oblique affines, 1 mm and other target resolutions, one-value (isotropic) form."""
    from mmr import registration as R
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "resample_nib_grid.npz"))
    for i in range(int(g["n_cases"])):
        shape, aff, new = tuple(g[f"c{i}_shape"]), g[f"c{i}_affine"], tuple(g[f"c{i}_new_size"])
        shape_r, affine_r = R.resample_grid_mm(shape, aff, new)
        assert tuple(shape_r) == tuple(int(v) for v in g[f"c{i}_shape_r"]), i
        np.testing.assert_allclose(affine_r, g[f"c{i}_affine_r"], rtol=1e-13, atol=1e-13)
        assert R._ORDER[str(g[f"c{i}_interp"])] == int(g[f"c{i}_order"])
        assert str(g[f"c{i}_mode_passed"]) == str(g[f"c{i}_mode"]) and float(g[f"c{i}_cval"]) == 0.0
    assert bool(g["dest_is_passed_through"]) and int(g["dest_order"]) == 1
    # and the resampling call built on that grid lands on it
    v = R.Volume(np.random.default_rng(0).random(tuple(g["c0_shape"])), g["c0_affine"])
    r = R.resample_mm(v, tuple(g["c0_new_size"]), "linear")
    assert r.shape == tuple(int(x) for x in g["c0_shape_r"]) and np.allclose(r.affine, g["c0_affine_r"])
