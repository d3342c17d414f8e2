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
