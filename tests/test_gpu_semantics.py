"""The three upstream behaviours that cannot be pinned are switchable per call
and process-wide (mmr.semantics); every variant is checked against its own oracle branch, forward and backward."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(got, ref):
    g = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    r = ref.detach().cpu().double().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref, np.float64)
    return np.abs(g - r).max() / max(np.abs(r).max(), 1e-30)


@pytest.mark.parametrize("shape,factor", [((16, 12, 20), 0.5), ((8, 6, 10), 2), ((7, 9, 5), 2), ((9, 6, 12), 1.5),
                                          ((10, 11, 13), 0.5)])
def test_resize_grid_variants(dev, shape, factor):
    import mmr
    from oracle import ops_np as O
    rng = np.random.default_rng(4)
    trf = rng.standard_normal(shape + (3,)).astype(np.float32)
    refs = {g: O.rescale_dense_transform(trf, factor, grid=g) for g in ("align_corners", "arange_over_f")}
    assert np.abs(refs["align_corners"] - refs["arange_over_f"]).max() > 1e-2  # the variants really differ
    for g, ref in refs.items():
        got = mmr.ops.rescale_transform(_t(trf[None], dev), factor, grid=g)[0].cpu().numpy()
        assert got.shape == ref.shape
        # 3e-6 where the sample coordinates are exact in fp32 (factor 2, 1/2); for factor 1.5 the kernel forms
        # i * (1/f), the oracle i / f: one ulp of a coordinate ~ 10 moves a value by a few 1e-6
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-6 if factor in (0.5, 2) else 2e-5, err_msg=g)
    # process-wide default
    assert mmr.semantics.get("resize_grid") == "align_corners"
    with mmr.semantics.using(resize_grid="arange_over_f"):
        np.testing.assert_allclose(mmr.utils.rescale_dense_transform(trf, factor), refs["arange_over_f"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(mmr.utils.rescale_dense_transform(trf, factor), refs["align_corners"], rtol=0, atol=3e-6)


def test_resize_arange_over_f_kat(dev):
    """arange(new)/f: factor 2 reads x/2 -> even outputs copy the input, odd ones are midpoints, the last one clamps
    to the edge; factor 1/2 reads every second sample exactly."""
    import mmr
    x = np.arange(5, dtype=np.float32) ** 2
    v = np.broadcast_to(x[:, None, None, None], (5, 3, 4, 1)).copy()
    up = mmr.ops.resize_trilinear(_t(v[None], dev), (10, 6, 8), grid="arange_over_f", zoom=2.0)[0].cpu().numpy()
    exp = np.array([0, .5, 1, 2.5, 4, 6.5, 9, 12.5, 16, 16], np.float32)
    np.testing.assert_allclose(up[:, 0, 0, 0], exp, atol=1e-6)
    np.testing.assert_allclose(up[:, 5, 7, 0], exp, atol=1e-6)
    dn = mmr.ops.resize_trilinear(_t(v[None], dev), (2, 1, 2), grid="arange_over_f", zoom=0.5)[0].cpu().numpy()
    np.testing.assert_array_equal(dn[:, 0, 0, 0], [0, 4])


@pytest.mark.parametrize("grid", ["align_corners", "arange_over_f"])
@pytest.mark.parametrize("shape,new,zoom", [((6, 8, 10), (12, 16, 20), 2.0), ((12, 16, 20), (6, 8, 10), 0.5),
                                            ((5, 7, 4), (7, 10, 6), 1.5),
                                            ((4, 5, 6), (14, 16, 19), 2.0)])   # Xo > X * zoom + 2: many outputs clamp onto the last voxel
def test_resize_bwd_is_adjoint_for_both_grids(dev, grid, shape, new, zoom):
    import mmr
    from oracle import grad_torch as G
    rng = np.random.default_rng(2)
    x = rng.standard_normal(shape + (3,)).astype(np.float32)
    g = rng.standard_normal(new + (3,)).astype(np.float32)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    (1.5 * G.resize(xt, new, grid=grid, zoom=zoom) * torch.from_numpy(g).double()).sum().backward()
    for separable in (True, False):   # three per-axis passes (what the trainer runs) / the one-launch 3-D gather
        got = mmr.ops.resize_trilinear_bwd(_t(g[None], dev), shape, mul=1.5, grid=grid, zoom=zoom, separable=separable)[0]
        assert _rel(got, xt.grad) < 1e-5, separable


@pytest.mark.parametrize("zlen", [37, 36])   # 37: one-z-per-lane forward kernel; 36 (Z % 4 == 0): four z per lane
@pytest.mark.parametrize("form", ["classic", "clamped"])
def test_ncc_form_variants(dev, form, zlen):
    import mmr
    from oracle import grad_torch as G, ops_np as O
    rng = np.random.default_rng(8)
    shape = (14, 19, zlen)
    I = rng.random((2,) + shape + (1,)).astype(np.float32)
    J = (0.5 * I + 0.5 * rng.random((2,) + shape + (1,))).astype(np.float32)
    # a block where BOTH images are constant: zero variances and zero cross -> classic 0 / (0 + eps) = 0, clamped
    # (eps / eps) * (eps / eps) = 1 for every window inside it
    I[:, 1:13, 3:16, 5:22] = 0.25
    J[:, 1:13, 3:16, 5:22] = 0.75
    ref = O.ncc_loss(I, J, 9, form=form)
    other = O.ncc_loss(I, J, 9, form="clamped" if form == "classic" else "classic")
    assert np.abs(ref - other).max() > 1e-4
    got = mmr.ops.ncc_loss(_t(I, dev), _t(J, dev), form=form).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4)
    It, Jt = torch.from_numpy(I).double().requires_grad_(True), torch.from_numpy(J).double().requires_grad_(True)
    gout = np.array([1.0, -0.5], np.float32)
    (G.ncc_loss(It, Jt, form=form) * torch.from_numpy(gout).double()).sum().backward()
    dI, dJ = mmr.ops.ncc_loss_bwd(_t(I, dev), _t(J, dev), _t(gout, dev), form=form)
    assert _rel(dI, It.grad) < 1e-4 and _rel(dJ, Jt.grad) < 1e-4
    with mmr.semantics.using(ncc_form=form):
        np.testing.assert_allclose(mmr.losses.NCC(9).loss(I, J).cpu().numpy(), ref, rtol=1e-4)


@pytest.mark.parametrize("mode", ["divide_no_nan", "max_eps"])
def test_dice_eps_variants(dev, mode):
    """bottom == 0 gives 0 under both; 0 < bottom < 1e-5 separates them (top / bottom vs top / 1e-5)."""
    import mmr
    from oracle import grad_torch as G, ops_np as O
    rng = np.random.default_rng(3)
    B, S, L = 2, (6, 7, 8), 4
    t = np.eye(L, dtype=np.float32)[rng.integers(0, 2, (B,) + S)]      # labels 2, 3 never true
    p = rng.random((B,) + S + (L,)).astype(np.float32)
    p[..., 2] = 0                                                       # label 2: bottom == 0
    p[..., 3] = 0
    p[:, 0, 0, 0, 3] = 2e-6                                             # label 3: bottom = 2e-6 < 1e-5, top = 0
    t[0, 0, 0, 0, :] = 0
    t[0, 0, 0, 0, 3] = 1                                                # item 0: top = 2 * 2e-6, bottom = 1 + 2e-6
    ref = O.dice_loss(t, p, eps_mode=mode)
    got = float(mmr.ops.dice_loss(_t(t, dev), _t(p, dev), eps_mode=mode))
    np.testing.assert_allclose(got, ref, rtol=1e-5)
    pt = torch.from_numpy(p).double().requires_grad_(True)
    G.dice_loss(torch.from_numpy(t).double(), pt, eps_mode=mode).backward()
    _, tb = mmr.ops.dice_loss(_t(t, dev), _t(p, dev), return_parts=True, eps_mode=mode)
    g = mmr.ops.dice_loss_bwd(_t(t, dev), tb, eps_mode=mode)
    assert _rel(g, pt.grad) < 1e-5
    if mode == "max_eps":  # item 1, label 3: bottom 2e-6 is clamped -> gradient 2 t / 1e-5 = 0 (t = 0), not -top/bot^2
        assert float(g[1, ..., 3].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["divide_no_nan", "max_eps"])
def test_dice_labels_eps_variants_match_dense(dev, mode):
    """The fused label-map Dice (what the trainer runs) honours the same switch as the dense one."""
    import mmr
    rng = np.random.default_rng(5)
    B, S, L = 1, (8, 9, 10), 5
    lab1 = rng.integers(0, 3, (B,) + S).astype(np.uint8)        # labels 3, 4 absent
    lab2 = rng.integers(0, 3, (B,) + S).astype(np.uint8)
    flow = (rng.standard_normal((B,) + S + (3,)) * 1.5).astype(np.float32)
    l1, l2, f = _t(lab1, dev), _t(lab2, dev), _t(flow, dev)
    loss, tb = mmr.ops.dice_labels_fwd(l1, l2, f, L, eps_mode=mode)
    pred = mmr.ops.warp3d(mmr.ops.onehot(l1[..., None].contiguous(), L), f, "linear", None)
    dense = mmr.ops.dice_loss(mmr.ops.onehot(l2[..., None].contiguous(), L), pred, eps_mode=mode)
    assert abs(float(loss) - float(dense)) < 1e-6
    g = mmr.ops.dice_labels_bwd(l1, l2, f, tb, L, eps_mode=mode)
    assert torch.isfinite(g).all()
