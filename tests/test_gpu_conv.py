"""MFMA implicit-GEMM Conv3D (+ first layer, pooling) vs the C oracle.

fp32 path: tolerance 1e-4 relative to the output scale (north_star's fp32 bar;
the MFMA is an exact fp32 FMA chain, the oracle accumulates in double).
bf16 path: compared with the oracle fed the SAME bf16-rounded inputs/weights
(fp32 accumulate both sides), tolerance 2e-3 of the output scale for the
final bf16 store rounding.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scale_err(got, ref):
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


CASES = [
    # (shape, C0, C1, up0, Cout)
    ((8, 8, 8), 64, 0, False, 64),
    ((4, 8, 8), 64, 0, False, 256),
    ((6, 10, 12), 128, 0, False, 128),      # ragged tiles
    ((8, 16, 8), 64, 64, True, 64),         # upsample + skip concat
    ((4, 8, 16), 256, 256, True, 256),      # the dec_final_0 shape class
    ((5, 9, 7), 64, 0, False, 3),           # flow head (Cout padded to 32), odd sizes
    ((2, 2, 2), 64, 0, False, 64),          # deepest level of a 32^3 volume
    ((5, 9, 11), 128, 0, False, 256),       # 256-column tile, batch of 2, ragged on every axis: the restage's LDS offset table
]


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp32x3"])
@pytest.mark.parametrize("shape,C0,C1,up0,Cout", CASES)
def test_conv_matches_oracle(dev, dtype, shape, C0, C1, up0, Cout):
    import mmr
    from oracle.cbind import conv3d_same
    from oracle.net_np import bf16_round
    from oracle import ops_np as O
    rng = np.random.default_rng(hash((shape, C0, C1, Cout)) % 2 ** 31)
    B = 2 if np.prod(shape) < 600 else 1
    X, Y, Z = shape
    s0 = (B, X // 2, Y // 2, Z // 2, C0) if up0 else (B, X, Y, Z, C0)
    a0 = rng.standard_normal(s0).astype(np.float32)
    a1 = rng.standard_normal((B, X, Y, Z, C1)).astype(np.float32) if C1 else None
    w = (rng.standard_normal((3, 3, 3, C0 + C1, Cout)) * np.sqrt(2.0 / (27 * (C0 + C1)))).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32) * 0.1
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    x3 = dtype == "fp32x3"
    if dtype == "bf16":
        a0, w = bf16_round(a0), bf16_round(w)
        a1 = bf16_round(a1) if a1 is not None else None
    full = O.upsample2(a0) if up0 else a0
    if a1 is not None:
        full = np.concatenate([full, a1], -1)
    leaky = Cout != 3
    ref = conv3d_same(full, w, bias, leaky=leaky, alpha=0.2)
    wp = mmr.ops.pack_conv_weights(torch.from_numpy(w).to(dev), tdt, x3=x3)
    got = mmr.ops.conv3d_k3(torch.from_numpy(a0).to(dev).to(tdt), wp, torch.from_numpy(bias).to(dev), Cout,
                            in1=None if a1 is None else torch.from_numpy(a1).to(dev).to(tdt), up0=up0,
                            leaky=leaky, out_f32=(Cout == 3), x3=x3)
    got = got.float().cpu().numpy()
    assert got.shape == ref.shape
    # fp32: exact-fp32 MFMA; fp32x3: bf16 hi/lo split, three MFMAs per product (north_star's 1e-4 fp32 bar, measured ~1e-5)
    tol = 1e-4 if dtype in ("fp32", "fp32x3") else (1e-5 if Cout == 3 else 4e-3)
    if dtype == "fp32x3":
        print(f"fp32x3 rel-to-scale error {_scale_err(got, ref):.2e}")
    assert _scale_err(got, ref) < tol


UPFOLD_CASES = [
    # (full-res shape, C0 upsampled, C1 skip, Cout): every N tile (256 / 128 / 64), ragged low-res tiles, C0 != C1
    ((8, 16, 16), 64, 64, 64),
    ((8, 16, 32), 256, 256, 256),     # the dec_final_0 shape class of BASELINE configs[1]
    ((12, 20, 28), 128, 64, 128),     # low-res 6x10x14: partial tiles on every axis
    ((4, 4, 6), 64, 128, 64),         # volume smaller than one tile: every voxel on a zero-padded border
    ((10, 14, 18), 64, 64, 128),      # odd low-resolution sizes 5 x 7 x 9
]


@pytest.mark.parametrize("dtype", ["bf16", "fp32x3"])
@pytest.mark.parametrize("shape,C0,C1,Cout", UPFOLD_CASES)
def test_folded_upsampling_conv_matches_oracle(dev, dtype, shape, C0, C1, Cout):
    """mmr_conv3d_k3_upfold_fwd + mmr_conv3d_k3_fwd_init against the C oracle of conv(concat([UpSampling3D(2)(x), skip]))
    (upsampled channels first).  The folded half sums up to 8 weights before the bf16 / hi-lo
    encoding, so bf16 is compared with the oracle on bf16-rounded inputs and UNROUNDED weights at the bf16 bound; fp32x3
    at north_star's 1e-4.  Also: the partial tensor alone equals the oracle conv over the upsampled channels only."""
    import mmr
    from oracle.cbind import conv3d_same
    from oracle.net_np import bf16_round
    from oracle import ops_np as O
    rng = np.random.default_rng(hash((shape, C0, C1, Cout)) % 2 ** 31)
    X, Y, Z = shape
    B = 2 if np.prod(shape) < 600 else 1
    a0 = rng.standard_normal((B, X // 2, Y // 2, Z // 2, C0)).astype(np.float32)
    a1 = rng.standard_normal((B, X, Y, Z, C1)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, C0 + C1, Cout)) * np.sqrt(2.0 / (27 * (C0 + C1)))).astype(np.float32)
    bias = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    x3 = dtype == "fp32x3"
    if dtype == "bf16":
        a0, a1 = bf16_round(a0), bf16_round(a1)
    ref = conv3d_same(np.concatenate([O.upsample2(a0), a1], -1), w, bias, leaky=True, alpha=0.2)
    ref_up = conv3d_same(O.upsample2(a0), np.ascontiguousarray(w[:, :, :, :C0]), None, leaky=False)
    wd = torch.from_numpy(w).to(dev)
    w_up, w_skip = mmr.ops.pack_upfold_weights(wd, C0, tdt, x3=x3)
    t0, t1 = torch.from_numpy(a0).to(dev).to(tdt), torch.from_numpy(a1).to(dev).to(tdt)
    got = mmr.ops.conv3d_k3_upfold(t0, t1, w_up, w_skip, torch.from_numpy(bias).to(dev), Cout, x3=x3).float().cpu().numpy()
    tol = 1e-4 if x3 else 6e-3
    err = _scale_err(got, ref)
    print(f"folded upsampling [{dtype}] {shape} {C0}+{C1}->{Cout}: rel-to-scale error {err:.2e}")
    assert got.shape == ref.shape and err < tol
    # the partial tensor on its own (launch A): NaN-poisoned first, so that an unwritten element would show
    lib = mmr._lib.load()
    for half in ((False, True) if dtype == "bf16" else (False,)):
        part = torch.full((B, X, Y, Z, Cout), float("nan"), dtype=torch.float16 if half else torch.float32, device=dev)
        rc = lib.mmr_conv3d_k3_upfold_fwd(t0.data_ptr(), C0, w_up.data_ptr(), part.data_ptr(), int(half), B, X // 2, Y // 2,
                                          Z // 2, Cout, mmr.ops.conv_mode(tdt, x3), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        assert _scale_err(part.float().cpu().numpy(), ref_up) < (1e-4 if x3 else 4e-3)
    if dtype == "bf16":   # fp32 partial between the launches: same result up to the half rounding of the partial
        got32 = mmr.ops.conv3d_k3_upfold(t0, t1, w_up, w_skip, torch.from_numpy(bias).to(dev), Cout, x3=x3,
                                         half_partial=False).float().cpu().numpy()
        assert _scale_err(got32, ref) < tol and _scale_err(got, got32) < 8e-3
    # and the one-launch path of the same layer agrees with the folded pair to the arithmetic's accuracy
    plain = mmr.ops.conv3d_k3(t0, mmr.ops.pack_conv_weights(wd, tdt, x3=x3), torch.from_numpy(bias).to(dev), Cout, in1=t1,
                              up0=True, x3=x3).float().cpu().numpy()
    assert _scale_err(got, plain) < (2e-5 if x3 else 1.2e-2)


@pytest.mark.parametrize("shape,Cout,odt", [((8, 8, 16), 64, "fp32"), ((5, 6, 19), 256, "fp32"), ((8, 8, 16), 256, "bf16"),
                                            ((5, 9, 7), 64, "bf16"), ((8, 8, 16), 64, "fp32x3"), ((5, 6, 19), 256, "fp32x3"),
                                            ((4, 8, 8), 32, "fp32x3")])
def test_first_layer(dev, shape, Cout, odt):
    import mmr
    from oracle.cbind import conv3d_same
    rng = np.random.default_rng(11)
    src = rng.random((2,) + shape + (1,)).astype(np.float32)
    trg = rng.random((2,) + shape + (1,)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 2, Cout)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32) * 0.1
    ref = conv3d_same(np.concatenate([src, trg], -1), w, bias, leaky=True)
    tdt = torch.bfloat16 if odt == "bf16" else torch.float32
    got = mmr.ops.conv3d_k3_cin2(torch.from_numpy(src).to(dev), torch.from_numpy(trg).to(dev),
                                 torch.from_numpy(w).to(dev), torch.from_numpy(bias).to(dev), tdt, x3=(odt == "fp32x3"))
    assert got.dtype == tdt
    # bf16: inputs, weights and output rounded to bf16 (8-bit mantissa); fp32x3: hi/lo split; fp32: exact VALU kernel
    assert _scale_err(got.float().cpu().numpy(), ref) < {"fp32": 1e-5, "fp32x3": 1e-4, "bf16": 1.2e-2}[odt]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool(dev, dtype):
    import mmr
    from oracle import ops_np as O
    rng = np.random.default_rng(12)
    x = torch.from_numpy(rng.standard_normal((2, 6, 8, 10, 64)).astype(np.float32)).to(dev).to(dtype)
    got = mmr.ops.maxpool3d2(x).float().cpu().numpy()
    assert np.array_equal(got, O.maxpool2(x.float().cpu().numpy()))


@pytest.mark.parametrize("shape,Cin,mode", [((5, 9, 7), 64, "bf16"), ((8, 8, 16), 256, "bf16"), ((2, 4, 8), 32, "fp32x3"),
                                            ((6, 7, 19), 64, "fp32x3"), ((4, 4, 8), 128, "fp32x3"),
                                            ((20, 13, 30), 64, "bf16"), ((17, 6, 14), 32, "fp32x3"),  # several x segments / tiles
                                            ((9, 7, 15), 256, "fp32x3"),  # single P buffer
                                            # bf16, 256 channels, Y >= 28: the 16-row patches (14 x 14 outputs, eight waves, two outputs per
                                            # thread): exact, ragged in y and z, several x segments and patches
                                            ((13, 28, 14), 256, "bf16"), ((6, 30, 17), 256, "bf16"), ((21, 45, 33), 256, "bf16")])
def test_flow_head_folded_taps(dev, shape, Cin, mode):
    """Flow head with the taps folded into the GEMM N axis vs the C oracle."""
    import mmr
    from oracle.cbind import conv3d_same
    from oracle.net_np import bf16_round
    rng = np.random.default_rng(21)
    x = rng.standard_normal((2,) + shape + (Cin,)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, Cin, 3)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(3).astype(np.float32)
    if mode == "bf16":
        x, wq = bf16_round(x), bf16_round(w)
    else:
        wq = w
    ref = conv3d_same(x, wq, bias, leaky=False)
    xt = torch.from_numpy(x).to(dev)
    if mode == "bf16":
        xt = xt.to(torch.bfloat16)
    got = mmr.ops.conv3d_k3_cout3(xt, torch.from_numpy(w).to(dev), torch.from_numpy(bias).to(dev), x3=(mode == "fp32x3"))
    assert got.dtype == torch.float32 and tuple(got.shape) == (2,) + shape + (3,)
    assert _scale_err(got.cpu().numpy(), ref) < (1e-5 if mode == "bf16" else 1e-4)


@pytest.mark.parametrize("mode", ["bf16", "fp32x3", "fp32"])
def test_split_k_small_launch_matches_unsplit(dev, mode):
    """Few-workgroup launches split the K walk over several workgroups (ws given) -- same result as the unsplit
    launch up to the summation order, bitwise reproducible, and the work-space query says when it applies."""
    import mmr
    from mmr import _lib
    ops = mmr.ops
    rng = np.random.default_rng(31)
    shape, Cin, Cout = (8, 16, 8), 128, 64
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    x3 = mode == "fp32x3"
    x = torch.from_numpy(rng.standard_normal((1,) + shape + (Cin,)).astype(np.float32)).to(dev).to(dt)
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, Cin, Cout)) * 0.05).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32)).to(dev)
    wp = ops.pack_conv_weights(w, dt, x3=x3)
    lib = _lib.load()
    m = ops.conv_mode(dt, x3)
    assert lib.mmr_conv3d_k3_ksplit_ws_bytes(1, *shape, Cin, Cout, m) > 0
    assert lib.mmr_conv3d_k3_ksplit_ws_bytes(1, 128, 128, 128, Cin, Cout, m) == 0    # 4096 tiles: whole rounds of workgroups
    y1 = ops.conv3d_k3(x, wp, b, Cout, leaky=True, x3=x3)       # split path (ws allocated by the wrapper)
    y2 = ops.conv3d_k3(x, wp, b, Cout, leaky=True, x3=x3)
    assert torch.equal(y1, y2)
    ref = torch.empty_like(y1)
    rc = lib.mmr_conv3d_k3_fwd(x.data_ptr(), Cin, 0, None, 0, wp.data_ptr(), b.data_ptr(), ref.data_ptr(), None, 1, *shape, Cout,
                               1, 0.2, m, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    tol = 1e-2 if mode == "bf16" else 1e-5  # other summation order over K = 3456 (bf16: output rounding on top)
    assert float((y1.float() - ref.float()).abs().max()) <= tol * float(ref.float().abs().max())


@pytest.mark.parametrize("mode,shape,Cin,Cout", [("bf16", (44, 30, 60), 256, 256), ("bf16", (36, 32, 64), 320, 256),
                                                 ("fp32x3", (72, 60, 40), 128, 64), ("fp32x3", (48, 64, 60), 128, 128)])
def test_tail_split_matches_unsplit(dev, mode, shape, Cin, Cout):
    """A launch whose tile count leaves a partial round of workgroups (352 / 288 / 360 / 384 tiles on 256 CUs) runs its last tiles
    with the K walk split (csrc/conv3d.hip conv_tail_plan): same result as the one-launch form up to the summation order,
    bitwise reproducible, ragged tiles included."""
    import mmr
    from mmr import _lib
    ops = mmr.ops
    rng = np.random.default_rng(hash((mode, shape)) % 2 ** 31)
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    x3 = mode == "fp32x3"
    x = torch.from_numpy(rng.standard_normal((1,) + shape + (Cin,)).astype(np.float32)).to(dev).to(dt)
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, Cin, Cout)) * 0.05).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32)).to(dev)
    wp = ops.pack_conv_weights(w, dt, x3=x3)
    lib = _lib.load()
    m = ops.conv_mode(dt, x3)
    assert lib.mmr_conv3d_k3_ksplit_ws_bytes(1, *shape, Cin, Cout, m) > 0
    for out_f32 in (False, True):
        y1 = ops.conv3d_k3(x, wp, b, Cout, leaky=True, x3=x3, out_f32=out_f32)
        y2 = ops.conv3d_k3(x, wp, b, Cout, leaky=True, x3=x3, out_f32=out_f32)
        assert torch.equal(y1, y2)
        ref = torch.empty_like(y1)
        rc = lib.mmr_conv3d_k3_fwd(x.data_ptr(), Cin, 0, None, 0, wp.data_ptr(), b.data_ptr(), ref.data_ptr(), None, 1, *shape,
                                   Cout, 1, 0.2, m, int(out_f32), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        tol = 1e-2 if y1.dtype == torch.bfloat16 else 1e-5
        d = (y1.float() - ref.float()).abs()
        assert float(d.max()) <= tol * float(ref.float().abs().max())
        # the tiles of the whole rounds come from the same one-launch code path: bit-identical there
        assert float(d[:, :4].max()) == 0.0


@pytest.mark.parametrize("shape,Cout,odt", [((8, 8, 16), 64, "bf16"), ((6, 10, 12), 256, "bf16"), ((8, 8, 16), 64, "fp32x3"),
                                            ((5, 7, 9), 32, "fp32x3"), ((4, 8, 8), 256, "fp32x3")])
def test_first_layer_fused_maxpool(dev, shape, Cout, odt):
    """mmr_conv3d_k3_cin2_fwd with pool_out: the full-resolution output is unchanged bit for bit and the pooled output
    equals MaxPooling3D(2) of it bit for bit (floor on odd sizes, ragged tiles)."""
    import mmr
    rng = np.random.default_rng(3)
    B = 2
    src = torch.from_numpy(rng.standard_normal((B,) + shape + (1,)).astype(np.float32)).to(dev)
    trg = torch.from_numpy(rng.standard_normal((B,) + shape + (1,)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, 2, Cout)) * 0.3).astype(np.float32)).to(dev)
    b = torch.from_numpy((rng.standard_normal(Cout) * 0.1).astype(np.float32)).to(dev)
    tdt = torch.bfloat16 if odt == "bf16" else torch.float32
    x3 = odt == "fp32x3"
    assert mmr.ops.cin2_pool_supported(Cout, tdt, x3)
    plain = mmr.ops.conv3d_k3_cin2(src, trg, w, b, tdt, x3=x3)
    full, pooled = mmr.ops.conv3d_k3_cin2(src, trg, w, b, tdt, x3=x3, pool=True)
    assert torch.equal(full, plain)
    ref = mmr.ops.maxpool3d2(plain)
    assert pooled.shape == ref.shape == (B, shape[0] // 2, shape[1] // 2, shape[2] // 2, Cout)
    assert torch.equal(pooled, ref)
    from oracle import ops_np as O
    np.testing.assert_array_equal(pooled.float().cpu().numpy(), O.maxpool2(plain.float().cpu().numpy()))


def test_half_partial_of_folded_pair_saturates_at_the_half_range(dev):
    """include/mmr.h (mmr_conv3d_k3_upfold_fwd, partial_half): the tensor between the two launches of a bf16 layer is IEEE
    half SATURATED to +-65504.  Drive the upsampled half's sums past that (activations 512, output channel co sums co input
    channels with weight +-1/4: +-27 * 128 * co in the interior, fewer taps on the border): the half partial holds exactly
    the clamped sums (every unsaturated one is a multiple of 128 below 2^16: exact in half; never inf / NaN), the fp32 partial
    the true sums, and the second launch adds the skip half to the clamped value -- the documented behaviour for weights far
    outside the He-init range; ``half_partial=False`` / ``fold_upsampling=False`` are the exact ways to run such a layer."""
    import mmr
    from oracle.cbind import conv3d_same
    from oracle.net_np import bf16_round
    from oracle import ops_np as O
    ops = mmr.ops
    rng = np.random.default_rng(41)
    B, X, Y, Z, C0, C1, Cout = 1, 8, 16, 16, 64, 64, 64
    a0 = np.full((B, X // 2, Y // 2, Z // 2, C0), 512.0, np.float32)
    a1 = bf16_round(rng.standard_normal((B, X, Y, Z, C1)).astype(np.float32))
    w = rng.choice([-0.25, 0.25], size=(3, 3, 3, C0 + C1, Cout)).astype(np.float32)
    sgn = np.where(np.arange(Cout) % 2 == 0, 0.25, -0.25).astype(np.float32)
    w[:, :, :, :C0, :] = np.where(np.arange(C0)[:, None] < np.arange(Cout)[None, :], sgn[None, :], 0.0)
    wd = torch.from_numpy(w).to(dev)
    w_up, w_skip = ops.pack_upfold_weights(wd, C0, torch.bfloat16)
    t0, t1 = torch.from_numpy(a0).to(dev).bfloat16(), torch.from_numpy(a1).to(dev).bfloat16()
    lib = mmr._lib.load()
    ref_up = conv3d_same(O.upsample2(a0), np.ascontiguousarray(w[:, :, :, :C0]), None, leaky=False)
    ref_skip = conv3d_same(a1, np.ascontiguousarray(w[:, :, :, C0:]), None, leaky=False)
    sat = np.abs(ref_up) > 65504
    assert sat.mean() > 0.2 and (~sat).mean() > 0.2 and np.abs(ref_up).max() > 2e5
    parts = {}
    for half in (True, False):
        part = torch.full((B, X, Y, Z, Cout), float("nan"), dtype=torch.float16 if half else torch.float32, device=dev)
        rc = lib.mmr_conv3d_k3_upfold_fwd(t0.data_ptr(), C0, w_up.data_ptr(), part.data_ptr(), int(half), B, X // 2, Y // 2, Z // 2,
                                          Cout, ops.conv_mode(torch.bfloat16), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        parts[half] = part.float().cpu().numpy()
    assert np.array_equal(parts[False], ref_up)                      # integers x 128 below 2^24: exact in fp32
    assert np.isfinite(parts[True]).all()
    clamped = np.clip(ref_up, -65504.0, 65504.0)
    assert np.array_equal(parts[True], clamped)
    out = ops.conv3d_k3_upfold(t0, t1, w_up, w_skip, None, Cout, leaky=False, out_f32=True).cpu().numpy()
    assert np.isfinite(out).all()
    assert np.abs(out - (clamped + ref_skip)).max() < 1e-5 * 65504      # fp32 accumulation on a 65504 base (ulp 2^-8)
    exact = ops.conv3d_k3_upfold(t0, t1, w_up, w_skip, None, Cout, leaky=False, out_f32=True, half_partial=False).cpu().numpy()
    assert np.abs(exact - (ref_up + ref_skip)).max() < 1e-5 * np.abs(ref_up).max()


def test_pack_batch_equals_the_single_image_packs(dev):
    """mmr_conv3d_k3_pack_batch (ops.PackBook): every kind of weight image, whole kernels and channel slices read in place,
    bit for bit what the one-image entry points write from contiguous copies -- first use (one job per launch) and the
    refresh after the weights changed (all jobs in one launch)."""
    import mmr
    from mmr import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    w = (torch.randn((3, 3, 3, 128, 64), generator=g) * 0.1).to(dev)
    C0 = 64
    for dtype, x3 in ((torch.bfloat16, False), (torch.float32, True), (torch.float32, "hi"), (torch.float32, False)):
        mode = ops.conv_mode(dtype, x3)
        book = ops.PackBook()

        def images():
            got = {"fwd": book.get("fwd", ops.PACK_FWD, w, 0, 128, mode), "skip": book.get("skip", ops.PACK_FWD, w, C0, 64, mode),
                   "dg": book.get("dg", ops.PACK_DGRAD, w, 0, 128, mode), "dgs": book.get("dgs", ops.PACK_DGRAD, w, C0, 64, mode)}
            ws = w[:, :, :, C0:, :].contiguous()
            ref = {"fwd": ops.pack_conv_weights(w, dtype, x3=x3), "skip": ops.pack_conv_weights(ws, dtype, x3=x3),
                   "dg": ops.pack_conv_weights(w, dtype, transpose_flip=True, x3=x3),
                   "dgs": ops.pack_conv_weights(ws, dtype, transpose_flip=True, x3=x3)}
            if mode in (ops.BF16, ops.F32X3):
                got["up"] = book.get("up", ops.PACK_UPFOLD, w, 0, C0, mode)
                ref["up"] = ops.pack_upfold_weights(w, C0, dtype, x3=x3)[0]
            if mode in (ops.F32X3, ops.F32X1):
                got["dgf"] = book.get("dgf", ops.PACK_DGFOLD, w, 0, C0, mode)
                ref["dgf"] = ops.pack_dgrad_upfold_weights(w, C0, x3=x3)
            return got, ref
        got, ref = images()
        for k in ref:
            assert got[k].numel() == ref[k].numel() and torch.equal(got[k], ref[k]), (dtype, x3, k)
        ptrs = {k: v.data_ptr() for k, v in got.items()}
        w.mul_(-0.7).add_(0.01)
        book.invalidate()
        book.refresh()                                  # one launch for every image recorded
        assert all(e["valid"] for e in book._e.values())
        got, ref = images()
        for k in ref:
            assert got[k].data_ptr() == ptrs[k] and torch.equal(got[k], ref[k]), (dtype, x3, k, "refresh")
    with pytest.raises(mmr._lib.MmrError):
        ops.PackBook().get("bad", ops.PACK_UPFOLD, w, 0, C0, ops.F32)   # no folded image in exact fp32
