"""Pins the CPU oracle: analytic known-answer tests + cross-checks against independent
torch-CPU implementations of the same maths.  The reference ships
no golden vectors for these operators, so this is all the pinning available (PARITY UNPINNED)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import net_np
from oracle import ops_np as O
from oracle.cbind import conv3d_same, warp3d_linear

rng = np.random.default_rng(0)


def test_interp_kats():
    vol = rng.standard_normal((6, 7, 8, 2)).astype(np.float32)
    z = np.zeros((6, 7, 8, 3), np.float32)
    for m in ("linear", "nearest"):
        assert np.array_equal(O.transform(vol, z, m), vol)
    s = z.copy(); s[..., 1] = -2
    assert np.array_equal(O.transform(vol, s), np.concatenate([vol[:, :1], vol[:, :1], vol[:, :-2]], 1))
    ramp = np.broadcast_to(np.arange(8, dtype=np.float32), (6, 7, 8)).copy()
    h = z.copy(); h[..., 2] = 0.5
    assert np.array_equal(O.transform(ramp, h)[..., 0], np.minimum(ramp + 0.5, 7))
    far = z.copy(); far[..., 0] = -50
    assert np.array_equal(O.transform(vol, far), np.broadcast_to(vol[:1], vol.shape))
    assert np.all(O.transform(vol, far, fill_value=2.5) == 2.5)
    hx = z.copy(); hx[..., 0] = 0.5  # tf.round is half-to-even: 0.5->0, 1.5->2, 2.5->2, 3.5->4
    assert np.array_equal(O.transform(vol, hx, "nearest"), vol[[0, 2, 2, 4, 4, 5]])


def test_warp_vs_grid_sample():
    vol = rng.standard_normal((1, 6, 7, 8, 3)).astype(np.float32)
    flow = (rng.standard_normal((1, 6, 7, 8, 3)) * 2).astype(np.float32)
    ref = O.spatial_transformer(vol, flow)
    g = O._grid((6, 7, 8)) + flow[0]
    gn = np.stack([2 * g[..., 2] / 7 - 1, 2 * g[..., 1] / 6 - 1, 2 * g[..., 0] / 5 - 1], -1)[None]
    t = F.grid_sample(torch.from_numpy(vol).permute(0, 4, 1, 2, 3), torch.from_numpy(gn).float(), mode="bilinear",
                      padding_mode="border", align_corners=True).permute(0, 2, 3, 4, 1).numpy()
    np.testing.assert_allclose(ref, t, atol=5e-6)
    np.testing.assert_allclose(warp3d_linear(vol, flow), ref, atol=1e-6)


def test_resize_vs_interpolate():
    x = rng.standard_normal((8, 6, 10, 3)).astype(np.float32)
    xt = torch.from_numpy(x).permute(3, 0, 1, 2)[None]
    up = F.interpolate(xt, scale_factor=2, mode="trilinear", align_corners=True)[0].permute(1, 2, 3, 0).numpy()
    np.testing.assert_allclose(O.resize(x, 2), up, atol=5e-6)
    dn = F.interpolate(xt, size=(4, 3, 5), mode="trilinear", align_corners=True)[0].permute(1, 2, 3, 0).numpy()
    np.testing.assert_allclose(O.resize(x, 0.5), dn, atol=5e-6)
    assert np.array_equal(O.rescale_dense_transform(x, 1), x)
    np.testing.assert_allclose(O.rescale_dense_transform(x, 2), 2 * up, atol=1e-5)


def test_vecint_compose_kats():
    a = (rng.standard_normal((8, 8, 8, 3))).astype(np.float32)
    z = np.zeros_like(a)
    assert np.array_equal(O.compose(a, z), a) and np.array_equal(O.compose(z, a), a)
    assert np.all(O.vecint(z, 5) == 0)
    c = np.broadcast_to(np.array([0.5, -0.25, 1.0], np.float32), (8, 8, 8, 3)).copy()
    np.testing.assert_allclose(O.vecint(c, 5), c, atol=1e-6)
    # VecInt(v, n) = VecInt(v/2, n-1) composed with itself
    half = O.vecint(a / 2, 4)
    np.testing.assert_allclose(O.vecint(a, 5), O.compose(half, half), atol=1e-5)


def test_losses_kats():
    lab = rng.integers(0, 5, (2, 6, 6, 6))
    t = np.eye(5, dtype=np.float32)[lab]
    assert abs(O.dice_loss(t, t) + 1) < 1e-12
    assert O.dice_loss(t, np.zeros_like(t)) == 0
    ramp = np.zeros((1, 8, 8, 8, 1), np.float32)
    ramp[..., 0] = 0.3 * np.arange(8)[None, :, None, None]
    np.testing.assert_allclose(O.grad_l2_loss(ramp, 2.0), [0.3 ** 2 / 3 * 2.0], rtol=1e-6)  # single channel: s^2/3
    I = rng.random((1, 12, 10, 14, 1)).astype(np.float32)
    J = rng.random((1, 12, 10, 14, 1)).astype(np.float32)

    def box(a):
        return F.conv3d(torch.from_numpy(a).double().permute(0, 4, 1, 2, 3), torch.ones(1, 1, 9, 9, 9).double(),
                        padding=4).numpy()[0, 0]
    Is, Js, I2, J2, IJ = box(I), box(J), box(I * I), box(J * J), box(I * J)
    ws = 729.0
    uI, uJ = Is / ws, Js / ws
    cross = IJ - uJ * Is - uI * Js + uI * uJ * ws
    Iv = I2 - 2 * uI * Is + uI * uI * ws
    Jv = J2 - 2 * uJ * Js + uJ * uJ * ws
    np.testing.assert_allclose(O.ncc_loss(I, J)[0], -np.mean(cross * cross / (Iv * Jv + 1e-5)), rtol=1e-7)
    assert O.ncc_loss(I, I)[0] < -0.999
    aff = np.zeros((1, 6, 6, 6, 3)); aff[..., 0] = np.arange(6)[None, :, None, None] * 0.5
    assert O.bending_energy(aff)[0] < 1e-20
    q = np.zeros((1, 6, 6, 6, 3)); q[..., 0] = (np.arange(6) ** 2)[None, :, None, None]
    np.testing.assert_allclose(O.bending_energy(q), [4.0 / 3.0])  # dxx = 2 in one of three channels


def test_conv_pool_vs_torch():
    x = rng.standard_normal((1, 5, 6, 7, 8)).astype(np.float32)
    w = rng.standard_normal((3, 3, 3, 8, 4)).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    t = F.conv3d(torch.from_numpy(x).permute(0, 4, 1, 2, 3).double(), torch.from_numpy(w).permute(4, 3, 0, 1, 2).double(),
                 torch.from_numpy(b).double(), padding=1)
    np.testing.assert_allclose(conv3d_same(x, w, b), t.permute(0, 2, 3, 4, 1).numpy(), atol=1e-5)
    np.testing.assert_allclose(conv3d_same(x, w, b, f32acc=True), t.permute(0, 2, 3, 4, 1).numpy(), atol=1e-4)
    np.testing.assert_allclose(O.conv3d_same_np(x, w, b), t.permute(0, 2, 3, 4, 1).numpy(), atol=1e-12)
    lt = F.leaky_relu(t, 0.2).permute(0, 2, 3, 4, 1).numpy()
    np.testing.assert_allclose(conv3d_same(x, w, b, leaky=True), lt, atol=1e-5)
    xp = rng.standard_normal((1, 6, 8, 4, 3)).astype(np.float32)
    tp = F.max_pool3d(torch.from_numpy(xp).permute(0, 4, 1, 2, 3), 2).permute(0, 2, 3, 4, 1).numpy()
    assert np.array_equal(O.maxpool2(xp), tp)
    tu = F.interpolate(torch.from_numpy(xp).permute(0, 4, 1, 2, 3), scale_factor=2, mode="nearest").permute(0, 2, 3, 4, 1).numpy()
    assert np.array_equal(O.upsample2(xp), tu)


def test_full_forward_vs_torch_assembly():
    """16^3 forward with seeded weights vs an independent torch-CPU assembly of the same graph."""
    enc, dec = [8, 8], [8, 8, 8]
    shape = (16, 16, 16)
    ws = net_np.init_weights(enc, dec, seed=3, flow_std=5e-2)
    mov = rng.random((1,) + shape + (1,)).astype(np.float32)
    fix = rng.random((1,) + shape + (1,)).astype(np.float32)
    out = net_np.vxm_dense_forward(mov, fix, ws, enc, dec, int_steps=3, int_resolution=2, svf_resolution=2)

    def conv(x, i, act=True):
        y = F.conv3d(x, torch.from_numpy(ws[i]).permute(4, 3, 0, 1, 2).double(), torch.from_numpy(ws[i + 1]).double(), padding=1)
        return F.leaky_relu(y, 0.2) if act else y
    x = torch.from_numpy(np.concatenate([mov, fix], -1)).permute(0, 4, 1, 2, 3).double()
    e0 = conv(x, 0); e1 = conv(F.max_pool3d(e0, 2), 2)
    d0 = conv(F.max_pool3d(e1, 2), 4)
    d1 = conv(torch.cat([F.interpolate(d0, scale_factor=2), e1], 1), 6)
    f0 = conv(torch.cat([F.interpolate(d1, scale_factor=2), e0], 1), 8)
    flow = conv(f0, 10, act=False)
    svf = 0.5 * F.interpolate(flow, size=(8, 8, 8), mode="trilinear", align_corners=True)

    def warp(v, u):  # v,u [1,C,8,8,8]; border grid_sample in voxel units
        S = v.shape[2:]
        g = torch.stack(torch.meshgrid(*[torch.arange(s, dtype=torch.float64) for s in S], indexing="ij"), 0)[None] + u
        gn = torch.stack([2 * g[:, 2] / (S[2] - 1) - 1, 2 * g[:, 1] / (S[1] - 1) - 1, 2 * g[:, 0] / (S[0] - 1) - 1], -1)
        return F.grid_sample(v, gn, mode="bilinear", padding_mode="border", align_corners=True)
    v = svf / 8
    for _ in range(3):
        v = v + warp(v, v)
    pos = F.interpolate(2 * v, size=shape, mode="trilinear", align_corners=True)
    moved = warp(torch.from_numpy(mov).permute(0, 4, 1, 2, 3).double(), pos)
    np.testing.assert_allclose(out["preint_flow"], svf.permute(0, 2, 3, 4, 1).numpy(), atol=2e-5)
    np.testing.assert_allclose(out["pos_flow"], pos.permute(0, 2, 3, 4, 1).numpy(), atol=5e-5)
    np.testing.assert_allclose(out["moved"], moved.permute(0, 2, 3, 4, 1).numpy(), atol=5e-5)


def test_layer_plan_matches_keras_order():
    plan = net_np.layer_plan([64] * 4, [64] * 6)
    assert [p[0] for p in plan] == ["enc_conv_0", "enc_conv_1", "enc_conv_2", "enc_conv_3", "dec_conv_3", "dec_conv_2",
                                    "dec_conv_1", "dec_conv_0", "dec_final_0", "dec_final_1", "flow"]
    assert [p[1] for p in plan] == [2, 64, 64, 64, 64, 128, 128, 128, 128, 64, 64]
    n = sum(27 * ci * co + co for _, ci, co in plan)
    assert n == 1_454_211 or abs(n - 1.45e6) < 1e4  # SURVEY: 1.45 M params at 64 features
    p256 = net_np.layer_plan([256] * 4, [256] * 6)
    assert abs(sum(27 * ci * co + co for _, ci, co in p256) - 23.04e6) < 5e4


def test_gradient_oracle_kinks_self_consistent():
    """oracle/grad_torch.py::unet(kinks=...) with the kinks taken from its OWN activations is the same function and
    has the same gradients as the plain graph (LeakyReLU slopes and max-pool routing of the same linear piece);
    with kinks from a perturbed evaluation the forward moves by no more than the perturbation."""
    from oracle import grad_torch as G
    enc, dec = [4, 4], [4, 4, 4]
    r = np.random.default_rng(3)
    ws = [torch.from_numpy(w).double().requires_grad_(True) for w in net_np.init_weights(enc, dec, seed=1, flow_std=0.1)]
    src = torch.from_numpy(r.random((1, 8, 8, 8, 1)))
    trg = torch.from_numpy(r.random((1, 8, 8, 8, 1)))
    acts = []
    real = G.conv

    def rec(x, w, b, leaky=True):
        y = real(x, w, b, leaky)
        if leaky:
            acts.append(y.detach())
        return y
    G.conv = rec
    try:
        f0 = G.unet(src, trg, ws, enc, dec)
    finally:
        G.conv = real
    assert len(acts) == len(enc) + len(dec)
    g0 = torch.autograd.grad(f0.square().sum(), ws)
    f1 = G.unet(src, trg, ws, enc, dec, kinks=acts)
    g1 = torch.autograd.grad(f1.square().sum(), ws)
    assert torch.allclose(f0, f1, rtol=0, atol=1e-14)
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-14)
    f2 = G.unet(src, trg, ws, enc, dec, kinks=[a + 1e-6 * torch.randn_like(a) for a in acts])
    assert (f2 - f0).abs().max() < 1e-4 * f0.abs().max()


def test_generate_label_maps_oracle_kat():
    """oracle/synth_np.generate_label_maps (train_synthmorph.py:55-69): zero warp noise -> the label map is the
    argmax of the Perlin image itself; a constant image -> label 0 everywhere (tf.argmax returns the first maximum)."""
    from oracle import synth_np
    shape, L = (8, 8, 8), 4
    r = np.random.default_rng(0)
    d = {"im": {"stds": [1.0], "noise": [r.standard_normal((4, 4, 4, L)).astype(np.float32)]},
         "warp": {"stds": [0.0], "noise": [np.zeros((4, 4, 4, 2, 3), np.float32)]}}
    maps, ims = synth_np.generate_label_maps(shape, L, [d], [2], [2])
    ref = synth_np.perlin((*shape, L), [2], [1.0], d["im"]["noise"])
    assert maps[0].dtype == np.uint8 and np.array_equal(maps[0], np.argmax(ref, -1))
    d["im"]["stds"] = [0.0]
    maps, _ = synth_np.generate_label_maps(shape, L, [d], [2], [2])
    assert not maps[0].any()


def test_switchable_semantics_oracle_branches():
    """KATs for the oracle's own branches of the unpinnable upstream variants."""
    # A4: arange(new)/f -- factor 2: even outputs copy, odd are midpoints, the last clamps; factor .5: every 2nd sample
    v = (np.arange(5, dtype=np.float32) ** 2)[:, None, None, None] * np.ones((1, 2, 2, 1), np.float32)
    up = O.resize(v, 2, grid="arange_over_f")
    assert np.allclose(up[:, 0, 0, 0], [0, .5, 1, 2.5, 4, 6.5, 9, 12.5, 16, 16])
    assert np.array_equal(O.resize(v, 0.5, grid="arange_over_f")[:, 0, 0, 0], [0, 4])
    ac = O.resize(v, 2, grid="align_corners")
    assert np.isclose(ac[0, 0, 0, 0], 0) and np.isclose(ac[-1, 0, 0, 0], 16) and not np.allclose(ac, up)
    # torch interpolate(align_corners=False) is NOT the arange/f grid (it is half-pixel centred): document by test
    t = F.interpolate(torch.from_numpy(v).permute(3, 0, 1, 2)[None], scale_factor=2, mode="trilinear", align_corners=False)
    assert not np.allclose(t[0, 0, :, 0, 0].numpy(), up[:, 0, 0, 0])
    # A6: bottom == 0 -> 0 in both; tiny bottom separates them
    t1 = np.zeros((1, 2, 2, 2, 2), np.float32); p1 = np.zeros_like(t1)
    assert O.dice_loss(t1, p1) == 0 and O.dice_loss(t1, p1, eps_mode="max_eps") == 0
    p1[0, 0, 0, 0, 0] = 2e-6; t1[0, 0, 0, 0, 0] = 1e-6
    a, b = O.dice_loss(t1, p1), O.dice_loss(t1, p1, eps_mode="max_eps")
    assert np.isclose(a, -0.5 * (2 * 2e-12 / 3e-6)) and np.isclose(b, -0.5 * (2 * 2e-12 / 1e-5))
    # A8: identical images -> both forms give -1 away from eps; constant image: the 4^3 windows that lie wholly inside the
    # 12^3 volume have zero variance and zero cross (classic cc = 0/(0+eps) = 0, clamped cc = (eps/eps)^2 = 1); every
    # window that touches the zero padding has cc = 1 under both
    r = np.random.default_rng(0)
    I = r.random((1, 12, 12, 12, 1))
    assert np.allclose(O.ncc_loss(I, I, 9, form="clamped"), O.ncc_loss(I, I, 9, form="classic"), rtol=1e-3)
    c = np.full((1, 12, 12, 12, 1), 0.5)
    assert np.isclose(O.ncc_loss(c, c, 9, form="classic")[0], -(1.0 - 64 / 1728), atol=1e-6)
    assert np.isclose(O.ncc_loss(c, c, 9, form="clamped")[0], -1.0, atol=1e-6)
    from oracle import grad_torch as G
    for form in ("classic", "clamped"):
        assert np.allclose(G.ncc_loss(torch.from_numpy(I), torch.from_numpy(I * 0.5 + 0.1), form=form).numpy(),
                           O.ncc_loss(I, I * 0.5 + 0.1, 9, form=form), rtol=1e-9)


def test_torch_cpu_baseline_graph_equals_numpy_oracle():
    """oracle/net_torch.py (what bench.py times as cpu_baseline) computes the same VxmDense forward as oracle/net_np.py."""
    from oracle import net_torch
    enc, dec = [8, 8, 8], [8, 8, 8, 8]
    r = np.random.default_rng(1)
    w = net_np.init_weights(enc, dec, seed=2, flow_std=0.05)
    for i in range(1, len(w), 2):
        w[i] = (r.standard_normal(w[i].shape) * 0.05).astype(np.float32)
    mov = r.random((1, 16, 16, 24, 1)).astype(np.float32)
    fix = r.random((1, 16, 16, 24, 1)).astype(np.float32)
    ref = net_np.vxm_dense_forward(mov, fix, w, enc, dec, 5, 2, 2)
    got = net_torch.vxm_dense_forward(torch.from_numpy(mov), torch.from_numpy(fix), net_torch.prepare_weights(w), enc, dec, 5, 2, 2)
    assert np.abs(ref["pos_flow"]).max() > 0.3
    for k in ("moved", "preint_flow", "pos_flow"):
        assert np.abs(got[k].numpy() - ref[k]).max() < 2e-5 * max(np.abs(ref[k]).max(), 1), k


def test_torch_cpu_graph_with_bf16_rounding_points_equals_numpy_oracle():
    """net_torch's ``quant`` hooks (what the 80x80x96 bf16 whole-network GPU test compares with) round at the same points as
    net_np's: same bf16 values in, fp32-vs-double accumulation apart, so the two agree far inside one bf16 step except where a
    pre-rounding value sits within that accumulation difference of a rounding boundary (rare; bounded by one bf16 ulp)."""
    from oracle import net_torch
    enc, dec = [8, 8, 8], [8, 8, 8, 8]
    r = np.random.default_rng(4)
    w = net_np.init_weights(enc, dec, seed=3, flow_std=0.05)
    for i in range(1, len(w), 2):
        w[i] = (r.standard_normal(w[i].shape) * 0.05).astype(np.float32)
    x = r.standard_normal((3, 5, 7)).astype(np.float32)
    assert np.array_equal(net_torch.bf16_round(torch.from_numpy(x)).numpy(), net_np.bf16_round(x))
    mov = r.random((1, 16, 16, 24, 1)).astype(np.float32)
    fix = r.random((1, 16, 16, 24, 1)).astype(np.float32)
    ref = net_np.vxm_dense_forward(mov, fix, w, enc, dec, 5, 2, 2, quant=net_np.bf16_round)
    plain = net_np.vxm_dense_forward(mov, fix, w, enc, dec, 5, 2, 2)
    got = net_torch.vxm_dense_forward(torch.from_numpy(mov), torch.from_numpy(fix),
                                      net_torch.prepare_weights(w, quant=net_torch.bf16_round), enc, dec, 5, 2, 2,
                                      quant=net_torch.bf16_round)
    for k in ("moved", "preint_flow", "pos_flow"):
        sc = max(np.abs(ref[k]).max(), 1)
        assert np.abs(got[k].numpy() - ref[k]).max() < 2e-3 * sc, k
        assert np.abs(plain[k] - ref[k]).max() > 1e-4 * sc, k     # the hooks do something


def test_slab_conv_of_the_gradient_oracle_equals_the_plain_conv():
    """oracle/grad_torch.conv splits large float64 convs into x slabs (bounded im2col memory); values and gradients equal the
    one-call form."""
    from oracle import grad_torch as G
    r = np.random.default_rng(2)
    x = torch.from_numpy(r.standard_normal((2, 11, 6, 5, 3))).requires_grad_(True)
    w = torch.from_numpy(r.standard_normal((3, 3, 3, 3, 4))).requires_grad_(True)
    b = torch.from_numpy(r.standard_normal(4)).requires_grad_(True)
    g = torch.from_numpy(r.standard_normal((2, 11, 6, 5, 4)))
    outs = []
    for slab in (None, 4):
        keep = G.SLAB_VOXELS
        G.SLAB_VOXELS = None if slab is None else slab * 6 * 5
        try:
            y = G.conv(x, w, b, leaky=True)
        finally:
            G.SLAB_VOXELS = keep
        gr = torch.autograd.grad((y * g).sum(), (x, w, b))
        outs.append((y.detach(), *gr))
    for a, c in zip(*outs):
        assert torch.allclose(a, c, rtol=1e-12, atol=1e-12)


def test_cpu_training_step_restatement_runs_and_learns():
    """oracle/train_torch.CpuStep (bench.py's training cpu_baseline): generators + fwd + bwd + Adam in fp32 on torch-CPU;
    its loss equals the float64 gradient oracle's on the same generated pair and Adam moves the weights."""
    import torch
    from oracle import grad_torch as G, train_torch
    rng = np.random.default_rng(0)
    lab = np.repeat(np.repeat(np.repeat(rng.integers(0, 4, (4, 4, 4)), 4, 0), 4, 1), 4, 2).astype(np.uint8)
    st = train_torch.CpuStep(lab, 4, [8, 8], [8, 8, 8], lr=1e-2, seed=1, warp_res=(8,), bias_res=(8,))
    w0 = [w.detach().clone() for w in st.ws]
    losses = [st.step() for _ in range(4)]
    assert all(np.isfinite(l) for l in losses) and G.DT == torch.float64   # the dtype switch is restored
    assert any(not torch.equal(a, b.detach()) for a, b in zip(w0, st.ws))
    assert 0.0 < losses[-1] < 2.0


def test_gradient_oracle_interpn_pin_selects_the_cell_not_the_value():
    """oracle/grad_torch.interpn(pin=): with the cell taken from another evaluation of the location the VALUE is unchanged
    (the interpolant is continuous across a cell boundary) while the gradient w.r.t. the location is the pinned cell's slope."""
    import torch
    from oracle import grad_torch as G
    vol = torch.tensor([0.0, 1.0, 3.0, 6.0], dtype=torch.float64).view(4, 1, 1, 1).expand(4, 2, 2, 1).contiguous()
    for x, pin_x, slope in ((1.0 - 1e-9, 1.0 + 1e-9, 2.0), (1.0 + 1e-9, 1.0 - 1e-9, 1.0), (1.5, 1.5, 2.0)):
        loc = torch.tensor([[x, 0.0, 0.0]], dtype=torch.float64, requires_grad=True)
        pin = torch.tensor([[pin_x, 0.0, 0.0]], dtype=torch.float64)
        v = G.interpn(vol, loc, pin)
        v.sum().backward()
        free = G.interpn(vol, loc.detach())
        assert abs(float(v) - float(free)) < 1e-8              # same value up to the 1e-9 offset times a slope
        assert abs(float(loc.grad[0, 0]) - slope) < 1e-12       # ... but the slope of the PINNED cell
    # outside the volume on the pinned side: clamp-to-edge, no gradient
    loc = torch.tensor([[1e-9, 0.0, 0.0]], dtype=torch.float64, requires_grad=True)
    v = G.interpn(vol, loc, torch.tensor([[-1e-9, 0.0, 0.0]], dtype=torch.float64))
    v.sum().backward()
    assert float(loc.grad[0, 0]) == 0.0 and abs(float(v)) < 1e-8
    # pin == own location: identical to the unpinned call, values and gradients
    g = torch.Generator().manual_seed(0)
    vol = torch.rand((5, 6, 7, 2), generator=g, dtype=torch.float64)
    shift = (torch.rand((5, 6, 7, 3), generator=g, dtype=torch.float64) - 0.5) * 4
    a = shift.clone().requires_grad_(True)
    b = shift.clone().requires_grad_(True)
    G.transform(vol, a).sum().backward()
    G.transform(vol, b, G.grid((5, 6, 7)) + shift).sum().backward()
    assert torch.equal(a.grad, b.grad)
    pins = G.tail_pins_from(torch.zeros(1, 4, 4, 4, 3), torch.zeros(2, 1, 4, 4, 4, 3), torch.zeros(1, 8, 8, 8, 3), 3)
    assert len(pins) == 1 and len(pins[0]["vecint"]) == 3 and pins[0]["warp"].shape == (8, 8, 8, 3)
    assert torch.equal(pins[0]["vecint"][1], G.grid((4, 4, 4)))
