"""HIP tail kernels (warp / resize / compose / VecInt / losses) vs the CPU oracle.

Tolerances: the kernels run the same fp32 operation order as oracle/ops_np.py
with FP contraction off, so warp-family results are compared at 1e-6 abs
(bit-exact for nearest-neighbour); reductions at 1e-5 rel vs the fp64 oracle.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _rand_flow(rng, shape, amp):
    return (rng.standard_normal(shape + (3,)) * amp).astype(np.float32)


@pytest.mark.parametrize("shape,C", [((12, 10, 14), 1), ((9, 7, 5), 3), ((8, 8, 8), 26)])
@pytest.mark.parametrize("method", ["linear", "nearest"])
@pytest.mark.parametrize("fill", [None, 0.0, -3.5])
def test_warp_matches_oracle(dev, shape, C, method, fill):
    import mmr
    from oracle import ops_np as O
    rng = np.random.default_rng(1)
    vol = rng.standard_normal((2,) + shape + (C,)).astype(np.float32)
    flow = _rand_flow(rng, (2,) + shape, 3.0)
    got = _np(mmr.ops.warp3d(torch.from_numpy(vol).to(dev), torch.from_numpy(flow).to(dev), method, fill))
    ref = O.spatial_transformer(vol, flow, method, fill)
    if method == "nearest":
        assert np.array_equal(got, ref)
    else:
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)


def test_warp_kat(dev):
    import mmr
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((1, 6, 7, 8, 2)).astype(np.float32)
    v = torch.from_numpy(vol).to(dev)
    zero = torch.zeros((1, 6, 7, 8, 3), device=dev)
    for m in ("linear", "nearest"):
        assert np.array_equal(_np(mmr.ops.warp3d(v, zero, m)), vol)  # identity, bit-exact
    shift = zero.clone()
    shift[..., 2] = 1.0  # integer shift along z with edge replication
    got = _np(mmr.ops.warp3d(v, shift))
    exp = np.concatenate([vol[:, :, :, 1:], vol[:, :, :, -1:]], 3)
    assert np.array_equal(got, exp)
    # half-voxel shift on a linear ramp is exact
    ramp = np.broadcast_to(np.arange(8, dtype=np.float32)[None, None, None, :, None], (1, 6, 7, 8, 1)).copy()
    half = zero.clone()
    half[..., 2] = 0.5
    got = _np(mmr.ops.warp3d(torch.from_numpy(ramp).to(dev), half))
    exp = np.minimum(ramp + 0.5, 7.0)
    assert np.array_equal(got, exp)
    # out of bounds -> edge value, or fill value
    far = zero.clone()
    far[..., 0] = 100.0
    assert np.array_equal(_np(mmr.ops.warp3d(v, far)), np.broadcast_to(vol[:, -1:], vol.shape))
    assert np.all(_np(mmr.ops.warp3d(v, far, "linear", 7.0)) == 7.0)
    # nearest rounds half to even: x + 0.5 -> 0.5->0, 1.5->2, 2.5->2
    hx = zero.clone()
    hx[..., 0] = 0.5
    got = _np(mmr.ops.warp3d(v, hx, "nearest"))
    idx = np.clip(np.rint(np.arange(6) + 0.5).astype(int), 0, 5)
    assert np.array_equal(got, vol[:, idx])


def test_warp_channelwise(dev):
    import mmr
    from oracle import ops_np as O
    rng = np.random.default_rng(2)
    vol = rng.standard_normal((6, 5, 7, 4)).astype(np.float32)
    flow = (rng.standard_normal((6, 5, 7, 4, 3)) * 2).astype(np.float32)
    got = mmr.utils.transform(vol, flow)
    ref = O.transform(vol, flow)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)


def test_warp_nearest_u8_bitexact(dev):
    import mmr
    from oracle import ops_np as O
    rng = np.random.default_rng(3)
    lab = rng.integers(0, 26, (2, 10, 12, 9, 1)).astype(np.uint8)
    flow = _rand_flow(rng, (2, 10, 12, 9), 4.0)
    flow[0, :3] = np.round(flow[0, :3]) + 0.5  # exercise ties
    got = _np(mmr.ops.warp3d_nearest_u8(torch.from_numpy(lab).to(dev), torch.from_numpy(flow).to(dev), 0))
    ref = O.spatial_transformer(lab.astype(np.float32), flow, "nearest", 0.0).astype(np.uint8)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("shape,factor", [((16, 12, 20), 0.5), ((8, 6, 10), 2), ((10, 10, 10), 0.5), ((7, 9, 5), 2)])
def test_rescale_matches_oracle(dev, shape, factor):
    import mmr
    from oracle import ops_np as O
    rng = np.random.default_rng(4)
    trf = rng.standard_normal(shape + (3,)).astype(np.float32)
    got = mmr.utils.rescale_dense_transform(trf, factor)
    ref = O.rescale_dense_transform(trf, factor)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
    got_b = mmr.utils.rescale_dense_transform(trf[None], factor)
    np.testing.assert_array_equal(got_b[0], got)


def test_rescale_kat(dev):
    import mmr
    rng = np.random.default_rng(5)
    trf = rng.standard_normal((6, 6, 6, 3)).astype(np.float32)
    assert np.array_equal(mmr.utils.rescale_dense_transform(trf, 1), trf)
    # a field affine in x survives 1/2 then x2 resizing up to the vector scaling
    x = np.arange(9, dtype=np.float32)
    aff = np.broadcast_to((0.25 * x + 1.0)[:, None, None, None], (9, 9, 9, 3)).copy()
    down = mmr.utils.rescale_dense_transform(aff, 0.5)  # 4^3; align-corners keeps end points
    assert down.shape == (4, 4, 4, 3)
    np.testing.assert_allclose(down[0, 0, 0], 0.5 * aff[0, 0, 0], atol=1e-6)
    np.testing.assert_allclose(down[-1, 0, 0], 0.5 * aff[-1, 0, 0], atol=1e-6)


def test_compose_and_vecint(dev):
    import mmr
    from oracle import ops_np as O
    rng = np.random.default_rng(6)
    a = _rand_flow(rng, (10, 12, 8), 1.5)
    b = _rand_flow(rng, (10, 12, 8), 1.5)
    np.testing.assert_allclose(mmr.utils.compose([a, b]), O.compose(a, b), rtol=0, atol=1e-6)
    z = np.zeros_like(a)
    assert np.array_equal(mmr.utils.compose([a, z]), a)
    assert np.array_equal(mmr.utils.compose([z, b]), b)
    for n in (0, 1, 5, 7):
        got = mmr.layers.VecInt(int_steps=n)(a[None])[0]
        np.testing.assert_allclose(got, O.vecint(a, n), rtol=0, atol=2e-6)
    assert np.all(mmr.layers.VecInt(int_steps=5)(z[None]) == 0)
    # constant field integrates to itself (composition of constant shifts)
    c = np.broadcast_to(np.array([0.5, -0.25, 1.0], np.float32), (10, 12, 8, 3)).copy()
    np.testing.assert_allclose(mmr.layers.VecInt(int_steps=5)(c[None])[0], c, atol=1e-6)
    # three-way compose associates from the right like vxm.utils.compose
    c3 = mmr.utils.compose([a, b, z])
    np.testing.assert_allclose(c3, O.compose(a, O.compose(b, z)), atol=1e-6)


def test_dice_grad(dev):
    import mmr
    from oracle import ops_np as O
    rng = np.random.default_rng(7)
    lab = rng.integers(0, 26, (2, 12, 10, 14))
    t = np.eye(26, dtype=np.float32)[lab]
    p = rng.random((2, 12, 10, 14, 26)).astype(np.float32)
    p[..., 5] = 0
    t[..., 5] = 0  # empty label in both -> divide_no_nan gives 0
    got = float(mmr.losses.Dice().loss(t, p))
    np.testing.assert_allclose(got, O.dice_loss(t, p), rtol=1e-5)
    assert abs(float(mmr.losses.Dice().loss(t, t)) - (-25 / 26)) < 1e-6  # Dice(x,x) = 1 per non-empty label
    assert float(mmr.losses.Dice().loss(t, np.zeros_like(t))) == 0.0
    np.testing.assert_allclose(float(mmr.losses.dice_loss_zeropad(t, p)), O.dice_loss_zeropad(t, p), rtol=1e-5)
    flow = _rand_flow(rng, (2, 12, 10, 14), 2.0)
    got = _np(mmr.losses.Grad("l2", loss_mult=0.7).loss(None, flow))
    np.testing.assert_allclose(got, O.grad_l2_loss(flow, 0.7), rtol=1e-5)
    s = 0.3  # ramp of slope s along x in 1 of 3 channels: axis-x mean = s^2/3, mean over 3 axes = s^2/9
    ramp = np.zeros((1, 8, 8, 8, 3), np.float32)
    ramp[..., 0] = s * np.arange(8, dtype=np.float32)[None, :, None, None]
    got = _np(mmr.losses.Grad("l2", loss_mult=2.0).loss(None, ramp))
    np.testing.assert_allclose(got, [s * s / 9 * 2.0], rtol=1e-5)


@pytest.mark.parametrize("shape", [(20, 18, 40), (9, 33, 12), (5, 11, 256), (40, 9, 260)])   # Z = 256: every lane of the
def test_ncc_bending(dev, shape):                                                              # four-z kernel; 260: one-z kernel
    import mmr
    from oracle import ops_np as O
    rng = np.random.default_rng(8)
    I = rng.random((2,) + shape + (1,)).astype(np.float32)
    J = rng.random((2,) + shape + (1,)).astype(np.float32)
    got = _np(mmr.losses.NCC(win=9).loss(I, J))
    np.testing.assert_allclose(got, O.ncc_loss(I, J, 9), rtol=1e-5)   # measured <= 1e-7 (tools/ncc_error_probe.py)
    same = _np(mmr.losses.NCC(win=9).loss(I, I))
    np.testing.assert_allclose(same, O.ncc_loss(I, I, 9), rtol=1e-5)
    assert np.all(same < -0.99)
    # affine intensity change: invariant away from the zero-padded border only, so compare with the oracle
    J2 = (2.0 * I + 0.5).astype(np.float32)
    np.testing.assert_allclose(_np(mmr.losses.NCC().loss(I, J2)), O.ncc_loss(I, J2, 9), rtol=1e-5)
    flow = _rand_flow(rng, (2,) + shape, 2.0)
    np.testing.assert_allclose(_np(mmr.losses.BendingEnergy().loss(None, flow)), O.bending_energy(flow), rtol=1e-5)
    lin = np.zeros((1,) + shape + (3,), np.float32)
    lin[..., 1] = 0.5 * np.arange(shape[0], dtype=np.float32)[None, :, None, None]
    assert float(mmr.losses.BendingEnergy().loss(None, lin)[0]) < 1e-10  # affine field has no bending


@pytest.mark.parametrize("shape", [(64, 40, 64), (48, 26, 66), (70, 20, 256)])      # four-z kernel, one-z kernel, whole 256-z rows
@pytest.mark.parametrize("form", ["classic", "clamped"])
def test_ncc_unnormalised_intensities_with_zero_background(dev, shape, form):
    """Raw scanner intensities (0 .. 4095) inside a block, exact zeros around it -- the shape of a skull-stripped volume.  The
    x window sums slide (add the joining plane, subtract the leaving one): once a column has left the block the sums must be
    EXACTLY zero again (the reference's conv-based sums are, and cc = 0 / (0 + eps) = 0 there); a rounding residual of 1e-7 of
    4095^2 x 729 in the variances would score O(1) garbage over the whole background.  Against the float64 oracle."""
    import mmr
    from oracle import ops_np as O
    rng = np.random.default_rng(21)
    X, Y, Z = shape
    I = np.zeros((2,) + shape + (1,), np.float32)
    J = np.zeros_like(I)
    bx, by, bz = slice(X // 8, X // 8 + X // 4), slice(Y // 4, Y // 4 + Y // 3), slice(Z // 3 + 1, Z // 3 + 1 + Z // 4 + 1)   # ragged in z: lanes straddle
    blk = I[:, bx, by, bz].shape
    I[:, bx, by, bz] = rng.integers(0, 4096, blk).astype(np.float32)
    J[:, bx, by, bz] = np.clip(0.7 * I[:, bx, by, bz] + rng.integers(0, 1500, blk), 0, 4095).astype(np.float32)
    ref = O.ncc_loss(I, J, 9, form=form)
    got = _np(mmr.ops.ncc_loss(torch.from_numpy(I).to(dev), torch.from_numpy(J).to(dev), 9, form=form))
    np.testing.assert_allclose(got, ref, rtol=2e-4)
    # same volume, block pushed against the far x face: the window leaves it through the volume border
    I2, J2 = np.ascontiguousarray(I[:, ::-1]), np.ascontiguousarray(J[:, ::-1])
    got2 = _np(mmr.ops.ncc_loss(torch.from_numpy(I2).to(dev), torch.from_numpy(J2).to(dev), 9, form=form))
    np.testing.assert_allclose(got2, O.ncc_loss(I2, J2, 9, form=form), rtol=2e-4)
    if form == "classic":     # windows wholly in the background contribute exactly nothing: an all-zero pair scores exactly 0
        z = torch.zeros((1,) + shape + (1,), device=dev)
        assert float(mmr.ops.ncc_loss(z, z, 9)[0]) == 0.0
    if form == "classic" and shape[2] % 4 == 0:
        dI, dJ = mmr.ops.ncc_loss_bwd(torch.from_numpy(I).to(dev), torch.from_numpy(J).to(dev))
        from oracle import grad_torch as G
        It, Jt = torch.from_numpy(I).double().requires_grad_(True), torch.from_numpy(J).double().requires_grad_(True)
        G.ncc_loss(It, Jt).sum().backward()
        assert float((dI.cpu().double() - It.grad).abs().max() / It.grad.abs().max()) < 1e-3


def test_loss_reductions_finish_in_kernel_and_accumulate(dev):
    """mmr_ncc_fwd_ticket_f32 / mmr_bending_fwd_ticket_f32: the last workgroup adds the partials (no finalize launch), `out` may
    carry a running total (out += scale * loss), the ticket word returns to zero so that the next call finds it so, and the sums
    are reproducible bit for bit -- and equal to the two-launch entry points'."""
    import ctypes
    import mmr
    from mmr import _lib
    g = torch.Generator(device="cpu").manual_seed(4)
    S = (40, 24, 64)
    I, J = torch.rand((2,) + S + (1,), generator=g).to(dev), torch.rand((2,) + S + (1,), generator=g).to(dev)
    flow = torch.randn((2,) + S + (3,), generator=g).to(dev)
    a = mmr.ops.ncc_loss(I, J, 9)
    b = mmr.ops.bending_energy(flow)
    for _ in range(3):
        assert torch.equal(mmr.ops.ncc_loss(I, J, 9), a) and torch.equal(mmr.ops.bending_energy(flow), b)
    tot = mmr.ops.ncc_loss(I, J, 9)
    mmr.ops.bending_energy(flow, out=tot, scale=0.25)
    assert torch.allclose(tot, a + 0.25 * b, rtol=1e-6, atol=0)
    tot2 = mmr.ops.bending_energy(flow)
    mmr.ops.ncc_loss(I, J, 9, out=tot2, scale=2.0)
    assert torch.allclose(tot2, b + 2.0 * a, rtol=1e-6, atol=0)
    assert int(mmr.ops._ticket(I.device).abs().sum()) == 0
    # the two-launch entry points of the C-ABI give the same bits
    lib = _lib.load()
    B, X, Y, Z = 2, *S
    ws = torch.empty(int(lib.mmr_ncc_ws_bytes(B, X, Y, Z)) + 64, dtype=torch.uint8, device=dev)
    out = torch.empty(B, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.mmr_ncc_fwd_f32(I.data_ptr(), J.data_ptr(), out.data_ptr(), ws.data_ptr(), B, X, Y, Z, 9, 1e-5, 0, st) == 0
    assert torch.equal(out, a)
    ws = torch.empty(int(lib.mmr_bending_ws_bytes(B, X, Y, Z)) + 64, dtype=torch.uint8, device=dev)
    assert lib.mmr_bending_fwd_f32(flow.data_ptr(), out.data_ptr(), ws.data_ptr(), B, X, Y, Z, st) == 0
    assert torch.equal(out, b)
    assert lib.mmr_ncc_fwd_ticket_f32(I.data_ptr(), J.data_ptr(), out.data_ptr(), ws.data_ptr(), None, B, X, Y, Z, 9, 1e-5, 0, 1.0, 0, st) == -1
