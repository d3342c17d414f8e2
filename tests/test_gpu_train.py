"""Backward kernels, Adam and the full SynthMorph training step vs torch-CPU float64 autograd of the
restated graph (oracle/grad_torch.py).  fp32 kernels vs fp64 oracle: tolerance 1e-4 of the gradient scale
(float atomics in the gather adjoints make the last bits order-dependent, far below this)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(got, ref):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    ref = ref.detach().cpu().double().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


def test_dice_from_labels_matches_onehot_formulation(dev):
    import mmr
    from oracle import grad_torch as G
    rng = np.random.default_rng(0)
    B, S, L = 2, (10, 12, 9), 7
    lab1 = rng.integers(0, L, (B,) + S).astype(np.uint8)
    lab2 = rng.integers(0, L - 1, (B,) + S).astype(np.uint8)  # label L-1 absent from the target
    flow = (rng.standard_normal((B,) + S + (3,)) * 2).astype(np.float32)
    loss, tb = mmr.ops.dice_labels_fwd(_t(lab1, dev), _t(lab2, dev), _t(flow, dev), L)
    # product's own one-hot path (materialised) must agree
    oh1, oh2 = mmr.ops.onehot(_t(lab1, dev), L), mmr.ops.onehot(_t(lab2, dev), L)
    pred = mmr.ops.warp3d(oh1, _t(flow, dev))
    loss_oh = mmr.ops.dice_loss(oh2, pred)
    assert abs(float(loss) - float(loss_oh)) < 1e-6
    # gradient wrt flow vs autograd
    f = torch.from_numpy(flow).double().requires_grad_(True)
    e = torch.eye(L, dtype=torch.float64)
    p = torch.stack([G.transform(e[torch.from_numpy(lab1[b]).long()], f[b]) for b in range(B)])
    ref = G.dice_loss(e[torch.from_numpy(lab2).long()], p)
    assert abs(float(loss) - float(ref.detach())) < 1e-6
    ref.backward()
    got = mmr.ops.dice_labels_bwd(_t(lab1, dev), _t(lab2, dev), _t(flow, dev), tb, L, scale=1.0)
    assert _rel(got, f.grad) < 1e-4
    acc = torch.ones_like(got)
    mmr.ops.dice_labels_bwd(_t(lab1, dev), _t(lab2, dev), _t(flow, dev), tb, L, scale=2.0, out=acc)
    assert _rel(acc - 1, 2 * f.grad) < 1e-4


def test_dice_zeropad_from_labels(dev):
    """Zero-pad-aware Dice (losses.py docstring intent) from label maps vs the one-hot oracle + autograd."""
    import mmr
    from oracle import grad_torch as G
    from oracle import ops_np as O
    rng = np.random.default_rng(11)
    B, S, L = 2, (10, 12, 9), 6
    lab1 = rng.integers(0, L, (B,) + S).astype(np.uint8)
    lab2 = rng.integers(0, L, (B,) + S).astype(np.uint8)
    lab1[:, :3] = 0  # zero-padded borders in both maps
    lab2[:, :, :2] = 0
    flow = (rng.standard_normal((B,) + S + (3,)) * 1.5).astype(np.float32)
    loss, tb = mmr.ops.dice_labels_fwd(_t(lab1, dev), _t(lab2, dev), _t(flow, dev), L, zeropad=True)
    e = np.eye(L, dtype=np.float32)
    pred = O.spatial_transformer(e[lab1], flow, "linear")
    ref = O.dice_loss_zeropad(e[lab2], pred)
    assert abs(float(loss) - ref) < 1e-5
    # the product's composed one-hot version agrees too
    comp = mmr.losses.dice_loss_zeropad(e[lab2], pred)
    assert abs(float(comp) - ref) < 1e-5
    # gradient: autograd through the same masked formula (mask treated as constant)
    f = torch.from_numpy(flow).double().requires_grad_(True)
    et = torch.eye(L, dtype=torch.float64)
    p = torch.stack([G.transform(et[torch.from_numpy(lab1[b]).long()], f[b]) for b in range(B)])
    t0, p0 = et[torch.from_numpy(lab2[0]).long()], p[0]
    # the mask is decided on the fp32 prediction, exactly as the forward does
    keep = ~((t0[..., 0] >= 1) | torch.from_numpy(pred[0][..., 0] >= 1))
    tm, pm = t0 * keep[..., None], p0 * keep[..., None]
    top = 2 * (tm * pm).sum((0, 1, 2))[1:]
    bot = (tm + pm).sum((0, 1, 2))[1:]
    (-(top / bot).mean()).backward()
    got = mmr.ops.dice_labels_bwd(_t(lab1, dev), _t(lab2, dev), _t(flow, dev), tb, L, scale=1.0, zeropad=True)
    assert _rel(got, f.grad) < 1e-4
    assert float(got[1].abs().max()) == 0.0  # only batch item 0 contributes (losses.py:38-39)


def test_grad_l2_bwd(dev):
    import mmr
    from oracle import grad_torch as G
    rng = np.random.default_rng(1)
    flow = rng.standard_normal((2, 7, 9, 8, 3)).astype(np.float32)
    f = torch.from_numpy(flow).double().requires_grad_(True)
    G.grad_l2(f, 0.7).sum().backward()
    assert _rel(mmr.ops.grad_l2_bwd(_t(flow, dev), 0.7), f.grad) < 1e-5


@pytest.mark.parametrize("shape,new,mul", [((6, 8, 10), (12, 16, 20), 2.0), ((12, 16, 20), (6, 8, 10), 0.5), ((5, 7, 4), (9, 8, 11), 1.5),
                                           ((1, 4, 3), (3, 8, 5), 1.0), ((4, 5, 6), (16, 5, 1), 1.0)])
def test_resize_bwd_is_adjoint(dev, shape, new, mul):
    import mmr
    from oracle import grad_torch as G
    rng = np.random.default_rng(2)
    x = rng.standard_normal(shape + (3,)).astype(np.float32)
    g = rng.standard_normal(new + (3,)).astype(np.float32)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    (mul * G.resize(xt, new) * torch.from_numpy(g).double()).sum().backward()
    for separable in (True, False):   # three per-axis passes (what the trainer runs) / the one-launch 3-D gather
        got = mmr.ops.resize_trilinear_bwd(_t(g[None], dev), shape, mul=mul, separable=separable)[0]
        assert _rel(got, xt.grad) < 1e-5, separable


@pytest.mark.parametrize("scale", [0.4, 2.5, "smooth", "smooth_large"])
def test_compose_bwd_tiled_multi_tile(dev, scale):
    """compose_bwd_tiled_kernel: several 8x8x16 tiles with overhang, batch of 2, splats inside the LDS image (small
    displacements) and beyond its 2-voxel margin (large ones, global-atomic path), aliased (VecInt) and separate outputs.
    'smooth': fields that vary slowly (what VecInt sees) -- there neighbouring voxels splat onto shared corners and the kernel merges
    them along z (DPP) and y (ds_bpermute) before the atomic, also across the clamped borders and the ragged tile edges."""
    import mmr
    from oracle import grad_torch as G
    from scipy.ndimage import zoom
    rng = np.random.default_rng(13)
    S = (19, 10, 37)
    if isinstance(scale, str):
        amp = 1.2 if scale == "smooth" else 4.0
        coarse = lambda: rng.standard_normal((2, 4, 3, 5, 3)) * amp
        up = lambda c: np.stack([np.stack([zoom(c[i, ..., k], [S[0] / 4, S[1] / 3, S[2] / 5], order=1) for k in range(3)], -1)
                                 for i in range(2)]).astype(np.float32)
        a, b = up(coarse()), up(coarse())
        assert a.shape == (2,) + S + (3,)
    else:
        a = (rng.standard_normal((2,) + S + (3,)) * scale).astype(np.float32)
        b = (rng.standard_normal((2,) + S + (3,)) * scale).astype(np.float32)
    g = rng.standard_normal((2,) + S + (3,)).astype(np.float32)
    da, db = mmr.ops.compose_bwd(_t(a, dev), _t(b, dev), _t(g, dev))
    for i in range(2):
        at, bt = torch.from_numpy(a[i]).double().requires_grad_(True), torch.from_numpy(b[i]).double().requires_grad_(True)
        ((bt + G.transform(at, bt)) * torch.from_numpy(g[i]).double()).sum().backward()
        assert _rel(da[i], at.grad) < 1e-5 and _rel(db[i], bt.grad) < 1e-5
    for i in range(2):
        vt = torch.from_numpy(a[i]).double().requires_grad_(True)
        (G.vecint(vt, 3) * torch.from_numpy(g[i]).double()).sum().backward()
        out, steps = mmr.ops.vecint_save(_t(a[i][None], dev), 3)
        dv = mmr.ops.vecint_bwd(_t(a[i][None], dev), steps, _t(g[i][None], dev), 3)
        assert _rel(dv[0], vt.grad) < 1e-5


def test_compose_vecint_warp_bwd(dev):
    import mmr
    from oracle import grad_torch as G
    rng = np.random.default_rng(3)
    S = (8, 9, 10)
    a = (rng.standard_normal(S + (3,)) * 1.5).astype(np.float32)
    b = (rng.standard_normal(S + (3,)) * 1.5).astype(np.float32)
    g = rng.standard_normal(S + (3,)).astype(np.float32)
    at, bt = torch.from_numpy(a).double().requires_grad_(True), torch.from_numpy(b).double().requires_grad_(True)
    ((bt + G.transform(at, bt)) * torch.from_numpy(g).double()).sum().backward()
    da, db = mmr.ops.compose_bwd(_t(a[None], dev), _t(b[None], dev), _t(g[None], dev))
    assert _rel(da[0], at.grad) < 1e-5 and _rel(db[0], bt.grad) < 1e-5
    for n in (0, 1, 5):
        v = (rng.standard_normal(S + (3,)) * 3).astype(np.float32)
        vt = torch.from_numpy(v).double().requires_grad_(True)
        (G.vecint(vt, n) * torch.from_numpy(g).double()).sum().backward()
        out, steps = mmr.ops.vecint_save(_t(v[None], dev), n)
        assert _rel(out, mmr.ops.vecint(_t(v[None], dev), n)) == 0
        dv = mmr.ops.vecint_bwd(_t(v[None], dev), steps, _t(g[None], dev), n)
        assert _rel(dv[0], vt.grad) < 1e-5, n
    vol = rng.standard_normal(S + (4,)).astype(np.float32)
    go = rng.standard_normal(S + (4,)).astype(np.float32)
    vt, ft = torch.from_numpy(vol).double().requires_grad_(True), torch.from_numpy(a).double().requires_grad_(True)
    (G.transform(vt, ft) * torch.from_numpy(go).double()).sum().backward()
    dvol, dflow = mmr.ops.warp3d_bwd(_t(vol[None], dev), _t(a[None], dev), _t(go[None], dev))
    assert _rel(dvol[0], vt.grad) < 1e-5 and _rel(dflow[0], ft.grad) < 1e-5


@pytest.mark.parametrize("shape,C0,C1,up0,Cout", [((8, 8, 8), 32, 0, False, 64), ((6, 10, 12), 64, 0, False, 32),
                                                    ((8, 16, 8), 32, 32, True, 64), ((4, 8, 8), 64, 64, True, 128),
                                                    ((5, 9, 7), 64, 0, False, 3)])
@pytest.mark.parametrize("x3", [False, True])
def test_conv_backward_pieces(dev, shape, C0, C1, up0, Cout, x3):
    import mmr
    from oracle import grad_torch as G
    rng = np.random.default_rng(4)
    B = 2
    X, Y, Z = shape
    s0 = (B, X // 2, Y // 2, Z // 2, C0) if up0 else (B, X, Y, Z, C0)
    a0 = rng.standard_normal(s0).astype(np.float32)
    a1 = rng.standard_normal((B, X, Y, Z, C1)).astype(np.float32) if C1 else None
    w = (rng.standard_normal((3, 3, 3, C0 + C1, Cout)) * 0.05).astype(np.float32)
    bias = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    gy = rng.standard_normal((B, X, Y, Z, Cout)).astype(np.float32)
    leaky = Cout != 3
    t0 = torch.from_numpy(a0).double().requires_grad_(True)
    t1 = torch.from_numpy(a1).double().requires_grad_(True) if C1 else None
    wt, bt = torch.from_numpy(w).double().requires_grad_(True), torch.from_numpy(bias).double().requires_grad_(True)
    xin = G.up(t0) if up0 else t0
    if C1:
        xin = torch.cat([xin, t1], -1)
    y = G.conv(xin, wt, bt, leaky)
    (y * torch.from_numpy(gy).double()).sum().backward()
    # product
    d0 = _t(a0, dev)
    d1 = _t(a1, dev) if C1 else None
    wp = mmr.ops.pack_conv_weights(_t(w, dev), torch.float32, x3=x3)
    yk = mmr.ops.conv3d_k3(d0, wp, _t(bias, dev), Cout, in1=d1, up0=up0, leaky=leaky, out_f32=True, x3=x3)
    assert _rel(yk, y) < (5e-5 if x3 else 1e-5)
    dy = _t(gy, dev).clone()
    db = torch.zeros(Cout, device=dev)
    dz = mmr.ops.leaky_bwd_bias_(yk if leaky else None, dy, db, leaky=leaky)
    assert _rel(db, bt.grad) < 1e-4
    dw = torch.zeros((3, 3, 3, C0 + C1, Cout), device=dev)
    mmr.ops.conv3d_k3_wgrad(d0, dz, dw, in1=d1, up0=up0, x3=x3)
    assert _rel(dw, wt.grad) < 1e-4
    if Cout == 3:
        dcat = mmr.ops.conv3d_k3_cout3_dgrad(dz, _t(w, dev))
    else:
        wtp = mmr.ops.pack_conv_weights(_t(w, dev), torch.float32, transpose_flip=True, x3=x3)
        dcat = mmr.ops.conv3d_k3(dz, wtp, None, C0 + C1, leaky=False, out_f32=True, x3=x3)
    if C1 or up0:
        g0, g1 = mmr.ops.upcat_bwd(dcat, C0, C1, up0)
        assert _rel(g0, t0.grad) < 1e-4
        if C1:
            assert _rel(g1, t1.grad) < 1e-4
    else:
        assert _rel(dcat, t0.grad) < 1e-4


@pytest.mark.parametrize("x3", [False, True])
def test_first_layer_wgrad_and_maxpool_bwd(dev, x3):
    import mmr
    from oracle import grad_torch as G
    rng = np.random.default_rng(5)
    S = (6, 8, 18)
    src = rng.random((2,) + S + (1,)).astype(np.float32)
    trg = rng.random((2,) + S + (1,)).astype(np.float32)
    gz = rng.standard_normal((2,) + S + (64,)).astype(np.float32)
    wt = torch.zeros((3, 3, 3, 2, 64), dtype=torch.float64, requires_grad=True)
    y = G.conv(torch.cat([torch.from_numpy(src), torch.from_numpy(trg)], -1).double(), wt, None, leaky=False)
    (y * torch.from_numpy(gz).double()).sum().backward()
    dw = torch.zeros((3, 3, 3, 2, 64), device=dev)
    mmr.ops.conv3d_k3_cin2_wgrad(_t(src, dev), _t(trg, dev), _t(gz, dev), dw, x3=x3)
    assert _rel(dw, wt.grad) < (2e-5 if x3 else 1e-5)
    mmr.ops.conv3d_k3_cin2_wgrad(_t(src, dev), _t(trg, dev), _t(gz, dev), dw, accumulate=True, x3=x3)
    assert _rel(dw, 2 * wt.grad) < (2e-5 if x3 else 1e-5)
    x = rng.standard_normal((2, 6, 8, 10, 32)).astype(np.float32)
    gp = rng.standard_normal((2, 3, 4, 5, 32)).astype(np.float32)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    (G.pool(xt) * torch.from_numpy(gp).double()).sum().backward()
    got = mmr.ops.maxpool3d2_bwd(_t(x, dev), _t(gp, dev))
    assert _rel(got, xt.grad) == 0
    acc = torch.ones_like(got)
    mmr.ops.maxpool3d2_bwd(_t(x, dev), _t(gp, dev), dx=acc)
    assert _rel(acc - 1, xt.grad) < 1e-6


@pytest.mark.parametrize("shape,Cin", [((4, 8, 8), 64), ((7, 13, 21), 128), ((2, 3, 5), 64)])
def test_flow_head_wgrad_x3_thin_kernel(dev, shape, Cin):
    """thin_wgrad_x3_kernel (taps folded into N, bf16 hi/lo products): several channel blocks, tiles that overhang
    the volume on every axis, volumes smaller than one tile, accumulate; vs float64 autograd and vs the exact path."""
    import mmr
    from oracle import grad_torch as G
    rng = np.random.default_rng(11)
    B = 2
    x = rng.standard_normal((B,) + shape + (Cin,)).astype(np.float32)
    gy = rng.standard_normal((B,) + shape + (3,)).astype(np.float32)
    wt = torch.zeros((3, 3, 3, Cin, 3), dtype=torch.float64, requires_grad=True)
    (G.conv(torch.from_numpy(x).double(), wt, None, leaky=False) * torch.from_numpy(gy).double()).sum().backward()
    dw = torch.full((3, 3, 3, Cin, 3), 7.0, device=dev)
    mmr.ops.conv3d_k3_wgrad(_t(x, dev), _t(gy, dev), dw, x3=True)
    assert _rel(dw, wt.grad) < 2e-5
    mmr.ops.conv3d_k3_wgrad(_t(x, dev), _t(gy, dev), dw, x3=True, accumulate=True)
    assert _rel(dw, 2 * wt.grad) < 2e-5
    ex = torch.zeros_like(dw)
    mmr.ops.conv3d_k3_wgrad(_t(x, dev), _t(gy, dev), ex, x3=False)
    assert _rel(ex, wt.grad) < 1e-5


def test_adam_matches_keras_formula(dev):
    import mmr
    rng = np.random.default_rng(6)
    w = rng.standard_normal(1000).astype(np.float32)
    m = np.zeros_like(w); v = np.zeros_like(w)
    wt, mt, vt = _t(w, dev), _t(m, dev), _t(v, dev)
    ref = w.astype(np.float64)
    for t in range(1, 4):
        g = rng.standard_normal(1000).astype(np.float32)
        mmr.ops.adam_step_(wt, _t(g, dev), mt, vt, t, lr=1e-2, grad_scale=0.5)
        gs = g.astype(np.float64) * 0.5
        m = 0.9 * m + 0.1 * gs
        v = 0.999 * v + 0.001 * gs * gs
        lr_t = 1e-2 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ref = ref - lr_t * m / (np.sqrt(v) + 1e-7)
    np.testing.assert_allclose(wt.cpu().numpy(), ref, rtol=2e-5, atol=1e-6)


_ARCHS = [([32, 32], [32, 32, 32]),               # 32-wide: 32x32 MFMA kernels (BN = 32)
          ([64, 64], [64, 64, 64]),               # 64-wide: 16x16x32 kernels with the transposed epilogue
          ([32, 64, 64], [64, 64, 32, 32, 32]),   # three levels, mixed widths
          ([64] * 4, [64] * 6)]                   # BASELINE configs[2] exactly: config/config.json:44-45 (4 levels, 64 wide)


@pytest.mark.parametrize("cdt,kinks,tol", [("fp32", False, 2e-4), ("fp32", True, 1e-4), ("fp32x3", True, 1e-4)])
@pytest.mark.parametrize("enc,dec", _ARCHS)
def test_full_training_step_gradients(dev, cdt, kinks, tol, enc, dec):
    """One SynthMorph step on a tiny U-Net: every gradient tensor vs autograd (several widths / depths, so that the
    pre-masked gradient bookkeeping sees plain, concat and pooled consumers on both conv kernel families).

    kinks=False: plain float64 autograd of the restated graph (exact-fp32 path, 2e-4).
    kinks=True: the oracle takes the LeakyReLU slopes and max-pool routing from the HIP forward's activations
    (oracle/grad_torch.py::unet), i.e. differentiates the same linear piece of the piecewise-linear net; both
    arithmetic modes -- fp32x3 is the API default and the only one with the split-store dgrad -- must then meet
    north_star's 1e-4 on EVERY gradient tensor."""
    _check_step_gradients(dev, cdt, kinks, tol, enc, dec)


def _check_step_gradients(dev, cdt, kinks, tol, enc, dec, shape=(16, 16, 32), L=5, B=2, block=4, int_steps=3, families=(), report=None,
                          tail_pins=False):
    """Every gradient tensor of one SynthMorph step (train_synthmorph.py:296-308) against oracle/grad_torch.synthmorph_loss;
    ``families``: kernel-family suffixes that must have run (ops.PROFILE) -- the folded launches at sizes where they engage."""
    import mmr
    from mmr import synth, training
    from oracle import grad_torch as G
    from oracle import net_np
    rng = np.random.default_rng(7)
    cs = tuple(s // block for s in shape)
    coarse = rng.integers(0, L, (B,) + cs)
    lab_s = np.repeat(np.repeat(np.repeat(coarse, block, 1), block, 2), block, 3).astype(np.uint8)[..., None]
    coarse = rng.integers(0, L, (B,) + cs)
    lab_t = np.repeat(np.repeat(np.repeat(coarse, block, 1), block, 2), block, 3).astype(np.uint8)[..., None]
    kw = dict(in_shape=shape, in_label_list=np.arange(L), out_label_list=np.arange(L), warp_std=2, warp_res=8, blur_std=1,
              bias_std=0.3, bias_res=8, gamma_std=0.25)
    g1, g2 = synth.labels_to_image(**kw, id=0, seed=1), synth.labels_to_image(**kw, id=1, seed=2)
    model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=int_steps, int_resolution=2, svf_resolution=2,
                                  compute_dtype=cdt)
    ws = net_np.init_weights(enc, dec, seed=3, flow_std=3e-2)
    for i in range(1, len(ws), 2):
        ws[i] = (rng.standard_normal(ws[i].shape) * 0.05).astype(np.float32)
    model.set_weights(ws)
    tr = training.SynthMorphTrainer(model, g1, g2, reg_param=0.8, optimizer=training.Adam(1e-3))
    out = tr.forward_backward(lab_s, lab_t)
    assert torch.isfinite(out["loss"]) and float(out["loss"]) > 0
    # deterministic comparison: run the trainer's kernels on fixed generator outputs
    gen1 = g1.generate(lab_s, want_onehot=True)
    gen2 = g2.generate(lab_t, want_onehot=True)
    ima1, ima2 = gen1["image"], gen2["image"]
    lab1, lab2 = gen1["labels"], gen2["labels"]
    tape = []
    tr.gflat.zero_()
    mmr.ops.PROFILE = [] if families else None
    try:
        flow = tr._forward(ima1, ima2, tape)
        svf, steps, pos_lo, pos = tr._tail_forward(flow)
        dice, tb = mmr.ops.dice_labels_fwd(lab1, lab2, pos, L)
        gl = mmr.ops.grad_l2_loss(pos, 0.8)
        dpos = mmr.ops.dice_labels_bwd(lab1, lab2, pos, tb, L, scale=float(B))
        mmr.ops.grad_l2_bwd(pos, 0.8, 1.0, out=dpos)
        tr._backward(tape, tr._tail_backward(dpos, svf, steps))
        ran = {f for f, *_ in (mmr.ops.PROFILE or [])}
    finally:
        mmr.ops.PROFILE = None
    for suffix in families:
        assert any(f.endswith(suffix) for f in ran), (suffix, sorted(ran))
    wt = [torch.from_numpy(w).double().requires_grad_(True) for w in ws]
    kk = None
    if kinks:  # activated outputs of the LeakyReLU layers, execution order (tape: conv0 / conv records)
        kk = [(r[4] if r[0] == "conv0" else r[5]).cpu().double() for r in tape
              if r[0] == "conv0" or (r[0] == "conv" and r[6])]
    tp = None
    if tail_pins:   # the tail's floor / clamp pieces of the HIP evaluation (oracle/grad_torch.interpn(pin=))
        tp = G.tail_pins_from(svf.cpu(), steps.cpu(), pos.cpu(), int_steps)
    total, rdice, rgl, rpos, rflow = G.synthmorph_loss(ima1.cpu().double(), ima2.cpu().double(), gen1["onehot"].cpu().double(),
                                                      gen2["onehot"].cpu().double(), wt, enc, dec, int_steps, 0.8, kinks=kk, tail_pins=tp)
    if report is not None:
        rflow.retain_grad()
    total.backward()
    assert np.abs(rpos.detach().numpy()).max() > 0.3, "flow too small to exercise the warp"
    assert _rel(flow, rflow) < 1e-4 and _rel(pos, rpos) < 1e-4
    assert abs(float(dice) - float(rdice.detach())) < 1e-5 and _rel(gl, rgl) < 1e-4
    names = [p[0] for p in model.plan]
    errs = [_rel(g, w.grad) for g, w in zip(tr.g, wt)]
    if report is not None:   # tools/grad_parity_fullsize.py: every error instead of the first failure, and how ill-conditioned the
        # flow head's bias gradient (a plain sum of d loss / d flow over all voxels) is: sum |terms| / |sum|
        dfl = tr._tail_backward(dpos, svf, steps).double()
        cond = (dfl.abs().sum((0, 1, 2, 3)) / dfl.sum((0, 1, 2, 3)).abs()).cpu().numpy()
        report.append(("flow bias cancellation sum|t| / |sum t| per channel", cond.tolist()))
        dd = (dfl.cpu() - rflow.grad).abs()
        big = dd > 1e-3 * rflow.grad.abs().max()   # voxels on another linear piece of the tail (floor / clamp of interpn)
        report.append(("d loss / d flow, elementwise (rel. to max)", float(dd.max() / rflow.grad.abs().max())))
        report.append(("   voxels off by more than 1e-3 of the max, and their share of the bias-gradient difference per channel",
                       [int(big.sum())] + ((dd * big).sum((0, 1, 2, 3)) / (dfl.cpu().sum((0, 1, 2, 3)) - rflow.grad.sum((0, 1, 2, 3))).abs()
                                            .clamp(min=1e-300)).tolist()))
        for i, err in enumerate(errs):
            report.append((f"{names[i // 2]} {'bias' if i % 2 else 'kernel'}", float(err)))
        return
    if families:
        print(f"step gradients at {shape} [{cdt}]: worst {max(errs):.2e} ({names[int(np.argmax(errs)) // 2]})")
    for i, err in enumerate(errs):
        assert err < tol, f"{names[i // 2]} {'bias' if i % 2 else 'kernel'}: {err:.2e}"


def test_folded_training_step_gradients_vs_oracle(dev):
    """BASELINE configs[2]'s architecture (config/config.json:44-45) at 96 x 96 x 128, where all three folded kernels of the
    training step engage for dec_final_0 (forward upfold + cinit, folded data gradient, folded weight gradient -- asserted):
    all 22 gradient tensors against the float64 gradient oracle on the HIP forward's linear piece at north_star's 1e-4.
    (The unfolded-vs-folded self-comparison below stays; this one is the oracle gate.)"""
    _check_step_gradients(dev, "fp32x3", True, 1e-4, [64] * 4, [64] * 6, shape=(96, 96, 128), L=6, B=1, block=8, int_steps=5,
                          families=("_upfold", "_cinit", "_dgfold", "wgrad_mfma_f32x3_upfold"))


def test_training_lowers_loss_and_is_reproducible(dev):
    import mmr
    from mmr import synth, training
    shape, enc, dec, L = (16, 16, 32), [32, 32], [32, 32, 32], 4
    rng = np.random.default_rng(8)
    mk = lambda: np.repeat(np.repeat(np.repeat(rng.integers(0, L, (1, 4, 4, 8)), 4, 1), 4, 2), 4, 3).astype(np.uint8)[..., None]
    lab_s, lab_t = mk(), mk()
    kw = dict(in_shape=shape, in_label_list=np.arange(L), out_label_list=np.arange(L), warp_std=1, warp_res=8, blur_std=1,
              bias_std=0.3, bias_res=8, gamma_std=0.25)

    def run():
        g1, g2 = synth.labels_to_image(**kw, id=0, seed=1), synth.labels_to_image(**kw, id=1, seed=2)
        d1, d2 = g1.draw(1), g2.draw(1)
        model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=3, int_resolution=2,
                                      svf_resolution=2, compute_dtype="fp32", seed=5)
        tr = training.SynthMorphTrainer(model, g1, g2, reg_param=0.1, optimizer=training.Adam(1e-3))
        return [float(tr.train_step(lab_s, lab_t, d1, d2)["loss"]) for _ in range(25)], model.get_weights()
    l1, w1 = run()
    assert l1[-1] < l1[0] - 0.02, l1
    l2, w2 = run()
    # float atomics in the gather adjoints make later steps drift in the last bits; the first steps must agree
    assert np.allclose(l1[:4], l2[:4], rtol=1e-4, atol=1e-5) and abs(l1[-1] - l2[-1]) < 0.05


def test_render_ahead_is_the_same_run(dev):
    """train_step(next_labels=): the next step's two renderings run on the generator stream behind the current step.  Same
    host draws in the same order -> the images are bit-identical to a twin generator's and the run is the run without it."""
    import mmr
    from mmr import synth, training
    shape, enc, dec, L = (16, 16, 32), [32, 32], [32, 32, 32], 4
    rng = np.random.default_rng(18)
    mk = lambda: torch.from_numpy(np.repeat(np.repeat(np.repeat(rng.integers(0, L, (1, 4, 4, 8)), 4, 1), 4, 2), 4, 3)
                                  .astype(np.uint8)[..., None]).to(dev)
    batches = [(mk(), mk()) for _ in range(5)]
    kw = dict(in_shape=shape, in_label_list=np.arange(L), out_label_list=np.arange(L), warp_std=1, warp_res=8, blur_std=1,
              bias_std=0.3, bias_res=8, gamma_std=0.25)

    def run(ahead):
        g1, g2 = synth.labels_to_image(**kw, id=0, seed=1), synth.labels_to_image(**kw, id=1, seed=2)
        model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=3, int_resolution=2, svf_resolution=2,
                                      compute_dtype="fp32x3", seed=5)
        tr = training.SynthMorphTrainer(model, g1, g2, reg_param=0.1, optimizer=training.Adam(1e-3))
        losses, imgs = [], []
        for i, b in enumerate(batches):
            nxt = batches[i + 1] if ahead and i + 1 < len(batches) else None
            losses.append(float(tr.train_step(*b, next_labels=nxt)["loss"]))
            if nxt is not None:
                torch.cuda.synchronize()
                imgs.append((tr._ahead[2][0].clone(), tr._ahead[2][3].clone()))
        return losses, imgs, tr
    l0, _, _ = run(False)
    l1, imgs, tr = run(True)
    assert tr._ahead is None and tr.gstream is not None
    t1, t2 = synth.labels_to_image(**kw, id=0, seed=1), synth.labels_to_image(**kw, id=1, seed=2)
    for i, b in enumerate(batches):
        a, c = t1.generate(b[0], want_onehot=False), t2.generate(b[1], want_onehot=False)
        if i:
            assert torch.equal(imgs[i - 1][0], a["image"]) and torch.equal(imgs[i - 1][1], c["labels"])
    assert np.allclose(l0[:3], l1[:3], rtol=1e-4, atol=1e-5) and abs(l0[-1] - l1[-1]) < 0.05, (l0, l1)
    tr.train_step(*batches[0], next_labels=batches[1])
    with pytest.raises(RuntimeError, match="announced other label maps"):
        tr.train_step(*batches[2])


def test_fit_with_and_without_render_ahead_is_the_same_history(dev):
    """fit(render_ahead=): two epochs with validation, device-resident and host label maps -- the next batch is pulled one step
    early inside an epoch only, so the data generator, both image generators and the validation generator are consumed in
    the same order either way."""
    import mmr
    from mmr import data, synth, training
    shape, enc, dec, L = (16, 16, 32), [32, 32], [32, 32, 32], 4
    rng = np.random.default_rng(28)
    maps = [np.repeat(np.repeat(np.repeat(rng.integers(0, L, (4, 4, 8)), 4, 0), 4, 1), 4, 2).astype(np.uint8) for _ in range(6)]
    kw = dict(in_shape=shape, in_label_list=np.arange(L), out_label_list=np.arange(L), warp_std=1, warp_res=8, blur_std=1,
              bias_std=0.3, bias_res=8, gamma_std=0.25)
    for device in (dev, None):
        hist = {}
        for ahead in (False, True):
            np.random.seed(7)   # set_random_zero_borders draws from the global NumPy state, like the reference's
            g1, g2 = synth.labels_to_image(**kw, id=0, seed=1), synth.labels_to_image(**kw, id=1, seed=2)
            model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=3, int_resolution=2, svf_resolution=2,
                                          compute_dtype="fp32x3", seed=5)
            # a small step: the float atomics of the gather adjoints make two runs of the SAME schedule drift apart (see
            # test_training_lowers_loss_and_is_reproducible); a batch consumed out of order would move the losses by ~1e-2
            tr = training.SynthMorphTrainer(model, g1, g2, reg_param=0.1, optimizer=training.Adam(1e-5))
            gen = data.gen_synthmorph_eb(maps[:4], batch_size=1, rng=np.random.default_rng(3), device=device)
            val = data.gen_synthmorph_eb(maps[4:], batch_size=1, rng=np.random.default_rng(4), device=device)
            hist[ahead] = tr.fit(gen, validation_data=val, validation_steps=2, epochs=2, steps_per_epoch=4, verbose=0,
                                 render_ahead=ahead)
            assert tr._ahead is None
        for a, b in zip(hist[False], hist[True]):
            assert abs(a["loss"] - b["loss"]) < 2e-5 and abs(a["val_loss"] - b["val_loss"]) < 2e-5, (hist[False], hist[True])


def test_optin_bf16_backward(dev):
    """Opt-in mixed-precision backward: forward unchanged (fp32x3), weight gradients within bf16-product accuracy
    of the fp32x3 gradients, and training still converges."""
    import mmr
    from mmr import synth, training
    shape, enc, dec, L = (16, 16, 32), [32, 32], [32, 32, 32], 4
    rng = np.random.default_rng(9)
    mk = lambda: np.repeat(np.repeat(np.repeat(rng.integers(0, L, (1, 4, 4, 8)), 4, 1), 4, 2), 4, 3).astype(np.uint8)[..., None]
    lab_s, lab_t = mk(), mk()
    kw = dict(in_shape=shape, in_label_list=np.arange(L), out_label_list=np.arange(L), warp_std=1, warp_res=8, blur_std=1,
              bias_std=0.3, bias_res=8, gamma_std=0.25)
    grads, losses = {}, {}
    for mode in (None, "bf16"):
        g1, g2 = synth.labels_to_image(**kw, id=0, seed=1), synth.labels_to_image(**kw, id=1, seed=2)
        d1, d2 = g1.draw(1), g2.draw(1)
        model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=3, int_resolution=2, svf_resolution=2,
                                      compute_dtype="fp32x3", seed=5)
        w = model.get_weights()
        w[-2] = (np.random.default_rng(3).standard_normal(w[-2].shape) * 3e-2).astype(np.float32)
        model.set_weights(w)
        tr = training.SynthMorphTrainer(model, g1, g2, reg_param=0.1, optimizer=training.Adam(1e-3), backward_precision=mode)
        out = tr.forward_backward(lab_s, lab_t, d1, d2)
        grads[mode] = tr.gflat.clone()
        losses[mode] = [float(out["loss"])] + [float(tr.train_step(lab_s, lab_t, d1, d2)["loss"]) for _ in range(20)]
    assert losses[None][0] == losses["bf16"][0]  # identical forward
    ref, got = grads[None], grads["bf16"]
    cos = float((ref * got).sum() / (ref.norm() * got.norm()))
    assert cos > 0.999, cos
    assert float((ref - got).norm() / ref.norm()) < 3e-2
    assert losses["bf16"][-1] < losses["bf16"][0] - 0.02


@pytest.mark.parametrize("shape,Cmid,Cin", [((8, 8, 8), 64, 64), ((6, 10, 12), 32, 128), ((9, 7, 5), 64, 32), ((8, 16, 8), 32, 256)])
@pytest.mark.parametrize("x3", [False, True, "hi"])
def test_dgrad_masked_equals_dgrad_then_leaky_bwd(dev, shape, Cmid, Cin, x3):
    """The fused epilogue (LeakyReLU backward + bias gradient of the producing layer) against the two-kernel path:
    same conv arithmetic, so the masked values must agree to fp32 rounding and the bias gradient to 1e-5."""
    import mmr
    ops = mmr.ops
    rng = np.random.default_rng(3)
    dz = _t(rng.standard_normal((1,) + shape + (Cmid,)).astype(np.float32), dev)
    w = _t((rng.standard_normal((3, 3, 3, Cin, Cmid)) * 0.05).astype(np.float32), dev)  # forward kernel Cin -> Cmid
    y = _t(rng.standard_normal((1,) + shape + (Cin,)).astype(np.float32), dev)           # activated output of the producer
    wt = ops.pack_conv_weights(w, torch.float32, transpose_flip=True, x3=x3)
    ref = ops.conv3d_k3(dz, wt, None, Cin, leaky=False, out_f32=True, x3=x3)
    db_ref = torch.zeros(Cin, device=dev)
    ref = ops.leaky_bwd_bias_(y, ref, db_ref, leaky=True)
    db = torch.full((Cin,), 7.0, device=dev)
    got = ops.conv3d_k3_dgrad_masked(dz, wt, Cin, y, db, x3=x3)
    # same MFMA arithmetic; the unfused conv of such a small volume takes the split-K path (other summation order)
    assert _rel(got, ref) < 2e-6
    assert _rel(db, db_ref) < 1e-5
    db2 = db.clone()
    ops.conv3d_k3_dgrad_masked(dz, wt, Cin, y, db2, accumulate=True, x3=x3)
    assert _rel(db2, 2 * db_ref) < 1e-5


@pytest.mark.parametrize("shape,Cmid,B", [((8, 8, 8), 64, 1), ((10, 12, 22), 32, 2), ((2, 2, 2), 64, 2), ((16, 8, 24), 128, 1)])
@pytest.mark.parametrize("x3", [True, "hi"])
def test_dgrad_masked_with_pooling_backward_in_the_epilogue(dev, shape, Cmid, B, x3):
    """mmr_conv3d_k3_dgrad_masked_pool: the gradient of a skip tensor y that also feeds MaxPooling3D(2) -- data gradient of the
    decoder conv + the pooling's gradient routed to each window's first maximum, times LeakyReLU'(y), bias gradient = column
    sums -- in ONE launch, against (a) float64 autograd of conv + max_pool3d on the same linear piece and (b) the two-launch path
    it replaces (dgrad_masked, then maxpool3d2_bwd(masked, accumulate)): same arithmetic, so equal to fp32 rounding.  Ragged
    tiles, a volume of one window, batch of 2, TIES inside windows (first maximum in x, y, z order wins)."""
    import mmr
    import torch.nn.functional as F
    ops = mmr.ops
    rng = np.random.default_rng(hash((shape, Cmid)) % 2 ** 31)
    Cy = 64
    X, Y, Z = shape
    dz = rng.standard_normal((B,) + shape + (Cmid,)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, Cy, Cmid)) * 0.05).astype(np.float32)              # forward kernel y (64 ch) -> Cmid
    y = rng.standard_normal((B,) + shape + (Cy,)).astype(np.float32)
    y[:, ::2, 1::2, :, :8] = 0.75          # ties: several equal maxima per window in the first channels
    y[:, :, :, :, 8:12] = np.round(y[:, :, :, :, 8:12])                                    # and many accidental ones
    dp = rng.standard_normal((B, X // 2, Y // 2, Z // 2, Cy)).astype(np.float32)
    assert ops.dgrad_masked_pool_supported(Cy, x3, X, Y, Z)
    wt = ops.pack_conv_weights(_t(w, dev), torch.float32, transpose_flip=True, x3=x3)
    db = torch.full((Cy,), 5.0, device=dev)
    got = ops.conv3d_k3_dgrad_masked(_t(dz, dev), wt, Cy, _t(y, dev), db, accumulate=True, x3=x3, pool_grad=_t(dp, dev))
    # (b) the two-launch path
    db2 = torch.zeros(Cy, device=dev)
    two = ops.conv3d_k3_dgrad_masked(_t(dz, dev), wt, Cy, _t(y, dev), db2, x3=x3)
    two = ops.maxpool3d2_bwd(_t(y, dev), _t(dp, dev), dx=two, masked=True, dbias=db2, acc_b=True)
    assert _rel(got, two) < 2e-6 and _rel(db - 5.0, db2) < 1e-5
    # (a) float64: d/dy_pre of sum(conv(y) * dz) + sum(pool(y) * dp) with y = LeakyReLU(y_pre) on y's linear piece
    yt = torch.from_numpy(y).double().requires_grad_(True)
    wk = torch.from_numpy(w).double()
    c = F.conv3d(yt.permute(0, 4, 1, 2, 3), wk.permute(4, 3, 0, 1, 2), padding=1).permute(0, 2, 3, 4, 1)
    pooled = F.max_pool3d(yt.permute(0, 4, 1, 2, 3), 2).permute(0, 2, 3, 4, 1)
    ((c * torch.from_numpy(dz).double()).sum() + (pooled * torch.from_numpy(dp).double()).sum()).backward()
    ref = yt.grad * torch.where(yt.detach() < 0, 0.2, 1.0)
    tol = 1e-4 if x3 is True else 1.5e-2
    # torch's max_pool3d backward also routes to the first maximum in (x, y, z) scan order
    assert _rel(got, ref) < tol, _rel(got, ref)
    assert _rel(db - 5.0, ref.sum((0, 1, 2, 3))) < (1e-4 if x3 is True else 2e-2)
    with pytest.raises(mmr._lib.MmrError):      # odd dims: no whole windows -> refused, the caller keeps the two-launch path
        ops.conv3d_k3_dgrad_masked(_t(dz[:, :-1], dev), wt, Cy, _t(y[:, :-1], dev), db, x3=x3, pool_grad=_t(dp, dev))


@pytest.mark.parametrize("shape,Cin", [((8, 8, 8), 64), ((5, 9, 11), 128), ((2, 17, 3), 64)])
def test_flow_dgrad_x3_kernel(dev, shape, Cin):
    """flow_dgrad_x3_kernel (channels as MFMA rows, K ordered (dx, dy | dz, co), bf16 hi/lo products): vs float64
    autograd, fused LeakyReLU mask + bias gradient vs the unfused pair (bit-equal), accumulate, batch of 2."""
    import mmr
    from oracle import grad_torch as G
    ops = mmr.ops
    rng = np.random.default_rng(12)
    B = 2
    gy = rng.standard_normal((B,) + shape + (3,)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, Cin, 3)) * 0.1).astype(np.float32)
    xt = torch.zeros((B,) + shape + (Cin,), dtype=torch.float64, requires_grad=True)
    (G.conv(xt, torch.from_numpy(w).double(), None, leaky=False) * torch.from_numpy(gy).double()).sum().backward()
    dy, wd = _t(gy, dev), _t(w, dev)
    got = ops.conv3d_k3_cout3_dgrad(dy, wd, x3=True)
    assert _rel(got, xt.grad) < 2e-5
    assert _rel(ops.conv3d_k3_cout3_dgrad(dy, wd), xt.grad) < 1e-5
    y = _t(rng.standard_normal((B,) + shape + (Cin,)).astype(np.float32), dev)
    db_ref = torch.zeros(Cin, device=dev)
    ref = ops.leaky_bwd_bias_(y, got.clone(), db_ref, leaky=True)
    db = torch.full((Cin,), -3.0, device=dev)
    fused = ops.conv3d_k3_cout3_dgrad_masked(dy, wd, y, db, x3=True)
    assert torch.equal(fused, ref) and _rel(db, db_ref) < 1e-5
    ops.conv3d_k3_cout3_dgrad_masked(dy, wd, y, db, accumulate=True, x3=True)
    assert _rel(db, 2 * db_ref) < 1e-5


@pytest.mark.parametrize("shape,Cin", [((8, 8, 8), 64), ((5, 9, 11), 128)])
def test_flow_dgrad_masked_equals_unfused(dev, shape, Cin):
    import mmr
    ops = mmr.ops
    rng = np.random.default_rng(4)
    dy = _t(rng.standard_normal((1,) + shape + (3,)).astype(np.float32), dev)
    w = _t((rng.standard_normal((3, 3, 3, Cin, 3)) * 0.1).astype(np.float32), dev)
    y = _t(rng.standard_normal((1,) + shape + (Cin,)).astype(np.float32), dev)
    db_ref = torch.zeros(Cin, device=dev)
    ref = ops.leaky_bwd_bias_(y, ops.conv3d_k3_cout3_dgrad(dy, w), db_ref, leaky=True)
    db = torch.full((Cin,), -3.0, device=dev)
    got = ops.conv3d_k3_cout3_dgrad_masked(dy, w, y, db)
    assert torch.equal(got, ref) and _rel(db, db_ref) < 1e-5
    assert ops.conv3d_k3_cout3_dgrad_masked(dy[..., :3], w[:, :, :, :32], y[..., :32].contiguous(), db[:32]) is None


@pytest.mark.parametrize("shape,C0,C1,up0", [((8, 8, 8), 32, 32, True), ((6, 10, 4), 64, 32, True), ((5, 7, 3), 32, 0, False),
                                              ((4, 6, 8), 36, 20, True)])
def test_upcat_bwd_masked_equals_unfused(dev, shape, C0, C1, up0):
    import mmr
    ops = mmr.ops
    rng = np.random.default_rng(5)
    dcat = _t(rng.standard_normal((1,) + shape + (C0 + C1,)).astype(np.float32), dev)
    s0 = tuple(s // 2 for s in shape) if up0 else shape
    y0 = _t(rng.standard_normal((1,) + s0 + (C0,)).astype(np.float32), dev)
    y1 = _t(rng.standard_normal((1,) + shape + (C1,)).astype(np.float32), dev) if C1 else None
    prev1 = _t(rng.standard_normal((1,) + shape + (C1,)).astype(np.float32), dev) if C1 else None
    # reference: scalar split, then the separate leaky-backward passes
    r0, r1 = ops.upcat_bwd(dcat, C0, C1, up0)
    b0_ref, b1_ref = torch.zeros(C0, device=dev), torch.zeros(max(C1, 1), device=dev)
    r0 = ops.leaky_bwd_bias_(y0, r0.clone(), b0_ref)
    if C1:
        r1 = ops.leaky_bwd_bias_(y1, r1.clone(), b1_ref[:C1]) + prev1
    b0 = torch.full((C0,), 5.0, device=dev)
    b1 = torch.full((max(C1, 1),), 1.0, device=dev)
    g0, g1 = ops.upcat_bwd(dcat, C0, C1, up0, d_in1=prev1.clone() if C1 else None, y0=y0, dbias0=b0, acc_b0=False,
                           y1=y1, dbias1=b1[:C1] if C1 else None, acc_b1=True)
    assert _rel(g0, r0) < 1e-6 and _rel(b0, b0_ref) < 1e-5
    if C1:
        assert _rel(g1, r1) < 1e-6 and _rel(b1[:C1] - 1.0, b1_ref[:C1]) < 1e-4
    # plain (unmasked) float4 path against the torch formulation
    p0, p1 = ops.upcat_bwd(dcat, C0, C1, up0)
    d = dcat[..., :C0]
    if up0:
        d = d.reshape(1, s0[0], 2, s0[1], 2, s0[2], 2, C0).sum(dim=(2, 4, 6))
    assert _rel(p0, d) < 1e-6 and (not C1 or torch.equal(p1, dcat[..., C0:].contiguous()))


@pytest.mark.parametrize("shape,C", [((8, 8, 8), 32), ((6, 4, 10), 64), ((4, 4, 4), 20)])
def test_maxpool_bwd_masked_equals_unfused(dev, shape, C):
    import mmr
    ops = mmr.ops
    rng = np.random.default_rng(6)
    x = _t(rng.standard_normal((1,) + shape + (C,)).astype(np.float32), dev)
    dp = _t(rng.standard_normal((1,) + tuple(s // 2 for s in shape) + (C,)).astype(np.float32), dev)
    prev = _t(rng.standard_normal((1,) + shape + (C,)).astype(np.float32), dev)
    xt = x.detach().clone().requires_grad_(True)
    torch.nn.functional.max_pool3d(xt.permute(0, 4, 1, 2, 3), 2).permute(0, 2, 3, 4, 1).backward(dp)
    plain = ops.maxpool3d2_bwd(x, dp)
    assert torch.equal(plain, xt.grad)
    db_ref = torch.zeros(C, device=dev)
    ref = ops.leaky_bwd_bias_(x, plain.clone(), db_ref) + prev
    db = torch.full((C,), 2.0, device=dev)
    got = ops.maxpool3d2_bwd(x, dp, dx=prev.clone(), masked=True, dbias=db, acc_b=True)
    assert _rel(got, ref) < 1e-6 and _rel(db - 2.0, db_ref) < 1e-4


@pytest.mark.parametrize("shape", [(12, 17, 60), (9, 9, 9), (20, 70, 13)])
def test_ncc_and_bending_backward(dev, shape):
    """Loss gradients vs torch-CPU float64 autograd of the restated formulas (which equal the numpy oracle's)."""
    import mmr
    from oracle import grad_torch as G, ops_np as O
    rng = np.random.default_rng(11)
    B = 2
    I = rng.random((B,) + shape + (1,)).astype(np.float32)
    J = (0.6 * I + 0.4 * rng.random((B,) + shape + (1,))).astype(np.float32)
    gout = np.array([1.0, -0.5], np.float32)
    It, Jt = torch.from_numpy(I).double().requires_grad_(True), torch.from_numpy(J).double().requires_grad_(True)
    loss = G.ncc_loss(It, Jt)
    assert np.allclose(loss.detach().numpy(), O.ncc_loss(I, J, 9), rtol=1e-9)  # the torch restatement IS the oracle formula
    (loss * torch.from_numpy(gout).double()).sum().backward()
    dI, dJ = mmr.ops.ncc_loss_bwd(_t(I, dev), _t(J, dev), _t(gout, dev))
    assert _rel(dI, It.grad) < 2e-5 and _rel(dJ, Jt.grad) < 2e-5   # measured <= 5e-6 (tools/ncc_error_probe.py)
    only_j = mmr.losses.NCC(9).grad(_t(I, dev), _t(J, dev))
    Jt.grad = None
    It.grad = None
    G.ncc_loss(It, Jt).sum().backward()
    assert _rel(only_j, Jt.grad) < 2e-5

    u = (rng.standard_normal((B,) + shape + (3,)) * 2).astype(np.float32)
    ut = torch.from_numpy(u).double().requires_grad_(True)
    e = G.bending_energy(ut)
    assert np.allclose(e.detach().numpy(), O.bending_energy(u), rtol=1e-9)
    (e * torch.from_numpy(gout).double()).sum().backward()
    du = mmr.ops.bending_energy_bwd(_t(u, dev), _t(gout, dev))
    assert _rel(du, ut.grad) < 1e-5
    acc = mmr.ops.bending_energy_bwd(_t(u, dev), _t(gout, dev), out=du.clone())
    assert _rel(acc, 2 * ut.grad) < 1e-5


@pytest.mark.parametrize("shape", [(19, 23, 64), (17, 40, 128), (33, 12, 256), (8, 8, 4), (70, 9, 12), (9, 16, 200)])
@pytest.mark.parametrize("form", ["classic", "clamped"])
def test_ncc_backward_two_pass_form(dev, shape, form):
    """Z % 4 == 0, Z <= 256: coefficient pass (the forward's march writing A, 2 Bc, 2 Cc and the two mean terms) + ONE box-filter
    pass that combines them with I_p, J_p.  Both gradients from one filter of five fields, or one gradient from three; several x
    segments (X = 70), partial y tiles, rows shorter than a wave, B = 2 with signed gout.  Against float64 autograd."""
    import mmr
    from oracle import grad_torch as G
    rng = np.random.default_rng(5)
    B = 2
    I = rng.random((B,) + shape + (1,)).astype(np.float32)
    J = (0.6 * I + 0.4 * rng.random((B,) + shape + (1,))).astype(np.float32)
    if form == "clamped":                      # a block where both are constant: all three maxima bind (see test_gpu_semantics)
        I[:, 1:7, 2:8, 0:4] = 0.25
        J[:, 1:7, 2:8, 0:4] = 0.75
    gout = np.array([1.0, -0.5], np.float32)
    It, Jt = torch.from_numpy(I).double().requires_grad_(True), torch.from_numpy(J).double().requires_grad_(True)
    (G.ncc_loss(It, Jt, form=form) * torch.from_numpy(gout).double()).sum().backward()
    Id, Jd, gd = _t(I, dev), _t(J, dev), _t(gout, dev)
    dI, dJ = mmr.ops.ncc_loss_bwd(Id, Jd, gd, form=form)
    assert _rel(dI, It.grad) < 2e-5 and _rel(dJ, Jt.grad) < 2e-5
    oI, none = mmr.ops.ncc_loss_bwd(Id, Jd, gd, want=("I",), form=form)
    assert none is None and _rel(oI, It.grad) < 2e-5
    none, oJ = mmr.ops.ncc_loss_bwd(Id, Jd, gd, want=("J",), form=form)
    assert none is None and _rel(oJ, Jt.grad) < 2e-5


def test_dense_dice_backward(dev):
    import mmr
    from oracle import grad_torch as G
    rng = np.random.default_rng(12)
    B, S, L = 2, (7, 9, 11), 5
    t = np.eye(L, dtype=np.float32)[rng.integers(0, L - 1, (B,) + S)]  # label L-1 never true
    p = rng.random((B,) + S + (L,)).astype(np.float32)
    p[..., L - 1] = 0  # ... nor predicted: bot == 0 -> divide_no_nan -> zero gradient
    pt = torch.from_numpy(p).double().requires_grad_(True)
    G.dice_loss(torch.from_numpy(t).double(), pt).backward()
    got = mmr.losses.Dice().grad(_t(t, dev), _t(p, dev))
    assert _rel(got, pt.grad) < 1e-5 and float(got[..., L - 1].abs().max()) == 0.0


@pytest.mark.parametrize("shape,C0,C1,Cz", [((8, 8, 8), 64, 64, 64), ((6, 10, 4), 32, 32, 32), ((4, 16, 8), 64, 64, 128)])
@pytest.mark.parametrize("x3", [True, "hi"])
def test_dgrad_split_equals_dgrad_then_upcat(dev, shape, C0, C1, Cz, x3):
    """Split-store dgrad of a concat layer (skip half masked in the epilogue, upsampled half compact) against
    conv3d_k3 -> upcat_bwd with the same masks."""
    import mmr
    ops = mmr.ops
    rng = np.random.default_rng(13)
    assert ops.dgrad_split_supported(C0, C1, x3) and not ops.dgrad_split_supported(C0, C1, False)
    dz = _t(rng.standard_normal((1,) + shape + (Cz,)).astype(np.float32), dev)
    w = _t((rng.standard_normal((3, 3, 3, C0 + C1, Cz)) * 0.05).astype(np.float32), dev)  # forward kernel (C0+C1) -> Cz
    half = tuple(s // 2 for s in shape)
    y0 = _t(rng.standard_normal((1,) + half + (C0,)).astype(np.float32), dev)
    y1 = _t(rng.standard_normal((1,) + shape + (C1,)).astype(np.float32), dev)
    wt = ops.pack_conv_weights(w, torch.float32, transpose_flip=True, x3=x3)
    dcat = ops.conv3d_k3(dz, wt, None, C0 + C1, leaky=False, out_f32=True, x3=x3)
    b0r, b1r = torch.zeros(C0, device=dev), torch.zeros(C1, device=dev)
    r0, r1 = ops.upcat_bwd(dcat, C0, C1, True, y0=y0, dbias0=b0r, y1=y1, dbias1=b1r)
    b0, b1 = torch.full((C0,), 3.0, device=dev), torch.full((C1,), -1.0, device=dev)
    d0c, d1 = ops.conv3d_k3_dgrad_split(dz, wt, C0, C1, y1=y1, dbias1=b1, x3=x3)
    g0, _ = ops.upcat_bwd(d0c, C0, 0, True, y0=y0, dbias0=b0)
    assert _rel(d1, r1) < 2e-6 and _rel(g0, r0) < 2e-6       # the unfused conv of these small volumes is split-K
    assert _rel(b1, b1r) < 1e-5 and _rel(b0, b0r) < 1e-5
    d0u, d1u = ops.conv3d_k3_dgrad_split(dz, wt, C0, C1, x3=x3)  # no mask
    assert _rel(d1u, dcat[..., C0:]) < 2e-6 and _rel(d0u, dcat[..., :C0]) < 2e-6


@pytest.mark.parametrize("shape,C0,C1,Cz", [((8, 16, 16), 64, 64, 64), ((12, 20, 28), 128, 64, 64), ((4, 4, 6), 64, 128, 32),
                                            ((16, 16, 32), 256, 256, 256), ((10, 14, 18), 64, 64, 96)])
@pytest.mark.parametrize("x3", [True, "hi"])
def test_folded_dgrad_of_upsampled_half(dev, shape, C0, C1, Cz, x3):
    """mmr_conv3d_k3_dgrad_upfold vs float64 autograd of conv(concat([UpSampling3D(2)(x_low), skip])) w.r.t. x_low: plain,
    and with the fused LeakyReLU backward + bias gradient of the layer that produced x_low (what the trainer uses instead of
    dgrad_split + the 2x2x2 pooling of upcat_bwd).  Ragged low-res tiles, a volume smaller than one tile, every N tile."""
    import mmr
    import torch.nn.functional as F
    ops = mmr.ops
    rng = np.random.default_rng(hash((shape, C0, C1, Cz)) % 2 ** 31)
    X, Y, Z = shape
    B = 2 if np.prod(shape) < 600 else 1
    xl = rng.standard_normal((B, X // 2, Y // 2, Z // 2, C0)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, C0 + C1, Cz)) * np.sqrt(2.0 / (27 * (C0 + C1)))).astype(np.float32)
    dz = rng.standard_normal((B, X, Y, Z, Cz)).astype(np.float32)
    xt = torch.from_numpy(xl).double().requires_grad_(True)
    up = xt.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
    wt = torch.from_numpy(w[:, :, :, :C0]).double()
    y = F.conv3d(up.permute(0, 4, 1, 2, 3), wt.permute(4, 3, 0, 1, 2), padding=1).permute(0, 2, 3, 4, 1)
    (y * torch.from_numpy(dz).double()).sum().backward()
    ref = xt.grad.numpy()
    wp = ops.pack_dgrad_upfold_weights(_t(w, dev), C0, x3=x3)
    got = ops.conv3d_k3_dgrad_upfold(_t(dz, dev), wp, C0, x3=x3)
    tol = 1e-4 if x3 is True else 1.5e-2     # 'hi': bf16-product backward (opt-in), bf16-grade
    assert tuple(got.shape) == ref.shape and _rel(got, ref) < tol, _rel(got, ref)
    # masked: times LeakyReLU'(x_low), bias gradient = column sums, accumulated onto an existing value
    mask = np.where(xl > 0, 1.0, 0.2)
    db = torch.full((C0,), 3.0, device=dev)
    gm = ops.conv3d_k3_dgrad_upfold(_t(dz, dev), wp, C0, ymask=_t(xl, dev), dbias=db, accumulate=True, x3=x3)
    assert _rel(gm, ref * mask) < tol
    assert _rel(db - 3.0, (ref * mask).sum((0, 1, 2, 3))) < (1e-4 if x3 is True else 2e-2)


def test_folded_training_step_matches_unfolded(dev):
    """The whole backward with the decoder layers folded (forward: upfold + cinit, backward: folded dgrad) against the same
    step with fold_upsampling=False, at a size where the fold engages (96^3, BASELINE configs[2]'s widths): every one of the
    22 gradient tensors to fp32x3 accuracy -- the two paths share no kernel for the upsampled halves."""
    import mmr
    from mmr import synth, training
    shape, L = (96, 96, 96), 6
    enc, dec = [64] * 4, [64] * 6
    rng = np.random.default_rng(3)
    lab = np.repeat(np.repeat(np.repeat(rng.integers(0, L, (1, 12, 12, 12)), 8, 1), 8, 2), 8, 3).astype(np.uint8)[..., None]
    kw = dict(in_shape=shape, in_label_list=np.arange(L), out_label_list=np.arange(L), warp_std=2, warp_res=16, blur_std=1,
              bias_std=0.3, bias_res=40, gamma_std=0.25)
    grads, used = {}, {}
    for fold in (False, True):
        g1, g2 = synth.labels_to_image(**kw, id=0, seed=1), synth.labels_to_image(**kw, id=1, seed=2)
        model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2,
                                      compute_dtype="fp32x3", seed=4, fold_upsampling=fold)
        w = model.get_weights()
        w[-2] = (np.random.default_rng(9).standard_normal(w[-2].shape) * 2e-2).astype(np.float32)
        model.set_weights(w)
        tr = training.SynthMorphTrainer(model, g1, g2, reg_param=1.0)
        mmr.ops.PROFILE = []
        out = tr.forward_backward(lab, lab, g1.draw(1), g2.draw(1))
        used[fold] = {f for f, *_ in mmr.ops.PROFILE}
        mmr.ops.PROFILE = None
        grads[fold] = [g.clone() for g in tr.g]
        assert torch.isfinite(out["loss"])
    assert any(f.endswith("_upfold") for f in used[True]) and any(f.endswith("_dgfold") for f in used[True])
    assert any(f.startswith("conv3d_k3_wgrad") and f.endswith("_upfold") for f in used[True])
    assert not any(f.endswith("_upfold") or f.endswith("_dgfold") for f in used[False])
    for i, (a, b) in enumerate(zip(grads[True], grads[False])):
        err = _rel(a, b)
        assert err < 2e-4, f"gradient tensor {i}: folded vs unfolded {err:.2e}"


@pytest.mark.parametrize("shape,C0,C1,Cout", [((8, 16, 16), 64, 64, 64), ((12, 20, 28), 32, 64, 128), ((4, 4, 6), 64, 32, 64),
                                              ((10, 14, 18), 64, 64, 64)])
@pytest.mark.parametrize("x3", [True, "hi"])
def test_folded_wgrad_of_concat_layer(dev, shape, C0, C1, Cout, x3):
    """mmr_conv3d_k3_wgrad_upfold vs float64 autograd of conv(concat([UpSampling3D(2)(x_low), skip])) w.r.t. the kernel: all
    27 x (C0 + C1) x Cout entries -- the upsampled rows through the per-class correlations on the low-resolution grid, the skip
    rows through the ordinary kernel -- plain and accumulated onto an existing gradient.  Ragged and odd low-res sizes."""
    import mmr
    import torch.nn.functional as F
    ops = mmr.ops
    rng = np.random.default_rng(hash((shape, C0, C1, Cout)) % 2 ** 31)
    X, Y, Z = shape
    B = 2 if np.prod(shape) < 600 else 1
    xl = rng.standard_normal((B, X // 2, Y // 2, Z // 2, C0)).astype(np.float32)
    sk = rng.standard_normal((B, X, Y, Z, C1)).astype(np.float32)
    dz = rng.standard_normal((B, X, Y, Z, Cout)).astype(np.float32)
    wt = torch.zeros((3, 3, 3, C0 + C1, Cout), dtype=torch.float64, requires_grad=True)
    up = torch.from_numpy(xl).double().repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
    cat = torch.cat([up, torch.from_numpy(sk).double()], -1)
    y = F.conv3d(cat.permute(0, 4, 1, 2, 3), wt.permute(4, 3, 0, 1, 2), padding=1).permute(0, 2, 3, 4, 1)
    (y * torch.from_numpy(dz).double()).sum().backward()
    ref = wt.grad.numpy()
    dw = torch.zeros((3, 3, 3, C0 + C1, Cout), device=dev)
    ops.conv3d_k3_wgrad_upfold(_t(xl, dev), _t(sk, dev), _t(dz, dev), dw, x3=x3)
    tol = 2e-5 if x3 is True else 1.5e-2
    assert _rel(dw[:, :, :, :C0], ref[:, :, :, :C0]) < tol, ("upsampled rows", _rel(dw[:, :, :, :C0], ref[:, :, :, :C0]))
    assert _rel(dw[:, :, :, C0:], ref[:, :, :, C0:]) < tol, ("skip rows", _rel(dw[:, :, :, C0:], ref[:, :, :, C0:]))
    ops.conv3d_k3_wgrad_upfold(_t(xl, dev), _t(sk, dev), _t(dz, dev), dw, accumulate=True, x3=x3)
    assert _rel(dw, 2 * ref) < tol
    # and against the one-launch kernel with the upsampling folded into its loader
    one = torch.zeros_like(dw)
    ops.conv3d_k3_wgrad(_t(xl, dev), _t(dz, dev), one, in1=_t(sk, dev), up0=True, x3=x3)
    assert _rel(dw, 2 * one) < (2e-5 if x3 is True else 2e-2)


@pytest.mark.parametrize("shape", [(3, 3, 3), (4, 8, 32), (5, 9, 33), (8, 16, 64), (7, 11, 40), (6, 3, 70), (16, 24, 100), (13, 27, 131)])
def test_bending_backward_tiled_shapes(dev, shape):
    """The tiled bending-energy gradient (4 x 8 x 32 tiles, field + halo 2 in LDS; tiles two voxels away from every face take the
    collapsed 25-point stencil, the others the 21 masked second differences) at the smallest volume, exactly one tile, one voxel
    past a tile in every axis, several tiles, thin volumes and volumes with both kinds of tile, vs float64 autograd."""
    import mmr
    from oracle import grad_torch as G
    rng = np.random.default_rng(17)
    u = (rng.standard_normal((2,) + shape + (3,)) * 2).astype(np.float32)
    gout = np.array([0.7, -1.3], np.float32)
    ut = torch.from_numpy(u).double().requires_grad_(True)
    (G.bending_energy(ut) * torch.from_numpy(gout).double()).sum().backward()
    du = mmr.ops.bending_energy_bwd(_t(u, dev), _t(gout, dev))
    assert _rel(du, ut.grad) < 1e-5
    assert _rel(mmr.ops.bending_energy_bwd(_t(u, dev)), torch.autograd.grad(G.bending_energy(ut).sum(), ut)[0]) < 1e-5


def test_bending_backward_full_size_properties(dev):
    """256^3 (BASELINE configs[4]): an affine field has no bending -> zero gradient; and the gradient is the derivative of the
    forward kernel's energy along a random direction (central difference of two forward evaluations, fp32: 2e-3)."""
    import mmr
    S = (256, 256, 256)
    x = torch.arange(S[0], dtype=torch.float32, device=dev).view(-1, 1, 1, 1)
    y = torch.arange(S[1], dtype=torch.float32, device=dev).view(1, -1, 1, 1)
    z = torch.arange(S[2], dtype=torch.float32, device=dev).view(1, 1, -1, 1)
    aff = (0.5 * x - 0.25 * y + 0.125 * z + torch.tensor([1.0, -2.0, 3.0], device=dev)).contiguous()[None]
    assert float(mmr.ops.bending_energy_bwd(aff).abs().max()) < 1e-9
    g = torch.Generator(device="cpu").manual_seed(2)
    lo = torch.randn((1, 32, 32, 32, 3), generator=g).to(dev)
    u = mmr.ops.resize_trilinear(lo.contiguous(), S, mul=3.0)          # smooth field: second differences well above fp32 noise
    d = mmr.ops.resize_trilinear(torch.randn((1, 32, 32, 32, 3), generator=g).to(dev).contiguous(), S)
    grad = mmr.ops.bending_energy_bwd(u)
    eps = 0.5
    fd = (float(mmr.ops.bending_energy(u + eps * d)) - float(mmr.ops.bending_energy(u - eps * d))) / (2 * eps)   # exact: E is quadratic
    an = float((grad.double() * d.double()).sum())
    assert abs(fd - an) < 2e-3 * abs(an), (fd, an)
