"""BASELINE.json's FULL sizes (C2 160x160x192 / 256 features, C3 160^3 / 64 features / 26 labels, C5 256^3), checked
through size-independent properties plus oracle spot checks on cropped windows: tile seams of the 19 200-workgroup
launches, exact linearity, the zero-flow identity of the whole network, Dice = -1 on identical maps, analytic
bending energy, NCC against the float64 oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

C2 = (160, 160, 192)


def _bf16_randn(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(dev)


def _window_vs_oracle(y, in0, in1, wq, origin, n, C2):
    """Window [origin, origin + n) of the full-resolution output against the C oracle run on the cropped input.  A side of
    the crop that coincides with a face of the volume keeps its outputs (the oracle's zero padding IS the layer's there);
    a side cut inside the volume loses its outermost output layer (the oracle padded where the layer had data)."""
    from oracle.cbind import conv3d_same
    lo = [o for o in origin]
    hi = [o + n for o in origin]
    assert all(o % 2 == 0 for o in lo) and all(h <= s for h, s in zip(hi, C2))
    up = in0[0, lo[0] // 2:hi[0] // 2, lo[1] // 2:hi[1] // 2, lo[2] // 2:hi[2] // 2].float()
    up = up.repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2)
    crop = torch.cat([up, in1[0, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].float()], -1).cpu().numpy()[None]
    ref = conv3d_same(crop, wq.cpu().numpy(), np.zeros(wq.shape[-1], np.float32), leaky=False)[0]
    keep = tuple(slice(0 if l == 0 else 1, n if h == s else n - 1) for l, h, s in zip(lo, hi, C2))
    vol = tuple(slice(l + k.start, l + k.stop) for l, k in zip(lo, keep))
    got = y[0][vol].float().cpu().numpy()
    return np.abs(got - ref[keep]).max() / np.abs(ref[keep]).max()


@pytest.mark.parametrize("path", ["upfold_pair", "one_launch"])
def test_c2_largest_conv_seams_linearity_and_window_vs_oracle(dev, path):
    """dec_final_0 of C2 (3d_reg.py:297-305 with config_inference.json:8-9): concat(up2(256 ch @ 80x80x96), 256 ch @ 160x160x192)
    -> 256, bf16.  ``upfold_pair`` is the path the benchmark runs since round 3 (networks.py::_conv: folded launch on the
    low-resolution grid into an IEEE-half partial, then the skip half started from it: 8 x 2 400 + 19 200 workgroups, LDS offset
    table, no tail split at this size); ``one_launch`` is the 27-tap kernel with the upsampling in its loader."""
    import mmr
    ops = mmr.ops
    C = 256
    half = tuple(s // 2 for s in C2)
    # weights = small integers x 2^-7 (std 0.016): exact in bf16, and so is every sum of up to 8 of them -- the folded weights
    # of launch A (pack_upfold_kernel sums in fp32, then rounds to bf16) carry NO rounding of their own, so the oracle
    # comparison below is as tight for the folded pair as for the one-launch kernel
    w = (torch.randint(-3, 4, (3, 3, 3, 2 * C, C), generator=torch.Generator().manual_seed(1)).float() * 2.0 ** -7).to(dev)
    wq = w.to(torch.bfloat16).float()
    assert torch.equal(w, wq)
    bias = torch.linspace(-1, 1, C, device=dev)
    fold = path == "upfold_pair"
    if fold:
        assert ops.upfold_supported(C, C, C, torch.bfloat16, False, 1, *C2)   # the gate networks.py::_conv asks
        w_up, w_skip = ops.pack_upfold_weights(w, C, torch.bfloat16)

        def conv(a0, a1, b, **kw):
            return ops.conv3d_k3_upfold(a0, a1, w_up, w_skip, b, C, **kw)
    else:
        wp = ops.pack_conv_weights(w, torch.bfloat16)

        def conv(a0, a1, b, **kw):
            kw.pop("half_partial", None)
            return ops.conv3d_k3(a0, wp, b, C, in1=a1, up0=True, **kw)

    # (1) per-channel-constant inputs: every voxel at least one step inside the volume sees the same products in the same
    #     order as every other voxel OF ITS PARITY CLASS (the fold pre-sums the weights per class; the one-launch kernel has
    #     one class) -> bitwise identical outputs across all tile / wave / lane seams of both launches
    c0 = torch.linspace(-1, 1, C, device=dev).to(torch.bfloat16)
    c1 = torch.linspace(0.5, -0.5, C, device=dev).to(torch.bfloat16)
    in0 = c0.expand((1,) + half + (C,)).contiguous()
    in1 = c1.expand((1,) + C2 + (C,)).contiguous()
    y = conv(in0, in1, bias, leaky=True)
    assert y.dtype == torch.bfloat16
    inner = y[0, 1:-1, 1:-1, 1:-1]
    ref = (torch.cat([c0.float(), c1.float()]).double() @ wq.double().sum(dim=(0, 1, 2)) + bias.double())
    ref = torch.where(ref < 0, 0.2 * ref, ref)
    for px in range(2):
        for py in range(2):
            for pz in range(2):
                cls = inner[px::2, py::2, pz::2]
                assert torch.equal(cls, cls[:1, :1, :1].expand_as(cls)), (path, px, py, pz)
                assert (cls[0, 0, 0].double() - ref).abs().max() < 2e-2 * ref.abs().max()
    if not fold:
        assert torch.equal(inner, inner[:1, :1, :1].expand_as(inner))
    del in0, in1, y, inner

    # (2) random inputs: linearity under a power-of-two scale -- exact with the fp32 partial (and for the one-launch kernel);
    #     with the half partial exact except where a partial sum falls into the half subnormals (|v| < 2^-14, spacing 2^-24:
    #     the skip half's fp32 accumulation then starts one subnormal step apart and may round differently a few times)
    in0 = _bf16_randn((1,) + half + (C,), dev, 2)
    in1 = _bf16_randn((1,) + C2 + (C,), dev, 3)
    y = conv(in0, in1, None, leaky=False, out_f32=True)
    y2 = conv(in0 * 2, in1 * 2, None, leaky=False, out_f32=True)
    if fold:
        assert float((y2 - 2 * y).abs().max()) <= 4 * 2.0 ** -23 * float(y2.abs().max())
        assert float(((y2 - 2 * y) != 0).float().mean()) < 1e-3
        del y2
        y32 = conv(in0, in1, None, leaky=False, out_f32=True, half_partial=False)
        y32b = conv(in0 * 2, in1 * 2, None, leaky=False, out_f32=True, half_partial=False)
        assert torch.equal(y32b, 2 * y32)
        assert float((y32 - y).abs().max()) < 2.0 ** -11 * float(y32.abs().max())   # half rounding of the partial only
        del y32b
    else:
        assert torch.equal(y2, 2 * y)
        del y2
    # (3) 20^3 windows against the C oracle on the cropped input: the far corner (three zero-padded faces, where the folded
    #     weights differ from the interior ones), the x = 0 face, and an interior block crossing voxel-tile seams of both
    #     launches on every axis (full-res tiles 4 x 8 x 8, low-res tiles 4 x 8 x 8 = 8 x 16 x 16 full-res voxels)
    n = 20
    for origin in ((C2[0] - n, C2[1] - n, C2[2] - n), (0, 70, 90), (70, 86, 118)):
        err = _window_vs_oracle(y, in0, in1, wq, origin, n, C2)
        print(f"C2 dec_final_0 [{path}] window at {origin}: rel-to-scale error {err:.2e}")
        # fp32 accumulation of 13 824 exact products on both sides; the folded pair adds the half rounding of its partial
        # (2^-12 of a partial sum that is ~0.7 of the output scale)
        assert err < (5e-4 if fold else 1e-4), (path, origin, err)
        if fold:
            err32 = _window_vs_oracle(y32, in0, in1, wq, origin, n, C2)
            print(f"C2 dec_final_0 [{path}, fp32 partial] window at {origin}: rel-to-scale error {err32:.2e}")
            assert err32 < 1e-4, (path, origin, err32)


def test_c2_network_zero_flow_head_is_identity(dev):
    """Whole C2 forward (bf16, 256 features): with a zero flow head the field is exactly 0 and moved == moving."""
    import mmr
    m = mmr.networks.VxmDense(C2, nb_unet_features=([256] * 4, [256] * 6), int_steps=5, int_resolution=2, svf_resolution=2,
                              compute_dtype="bf16", seed=3)
    w = m.get_weights()
    w[-2][:] = 0
    w[-1][:] = 0
    m.set_weights(w)
    g = torch.Generator(device="cpu").manual_seed(5)
    mov = torch.rand((1,) + C2 + (1,), generator=g).to(dev)
    fix = torch.rand((1,) + C2 + (1,), generator=g).to(dev)
    out = m.forward(mov, fix)
    assert torch.count_nonzero(out["preint_flow"]) == 0 and torch.count_nonzero(out["pos_flow"]) == 0
    assert torch.equal(out["y_source"], mov)
    # and with a non-zero head the result is finite, differs from the input and is reproducible bit for bit
    w[-2][:] = (np.random.default_rng(0).standard_normal(w[-2].shape) * 1e-3).astype(np.float32)
    m.set_weights(w)
    a = m.forward(mov, fix)
    b = m.forward(mov, fix)
    assert torch.isfinite(a["y_source"]).all() and not torch.equal(a["y_source"], mov)
    assert torch.equal(a["y_source"], b["y_source"]) and torch.equal(a["preint_flow"], b["preint_flow"])


def test_c2_whole_network_bf16_full_size_vs_rounding_oracle(dev):
    """BASELINE configs[1] exactly as bench.py times it -- 160x160x192, enc/dec = 256, bf16, three folded decoder layers, 19 200-
    workgroup launches -- as a WHOLE forward against oracle/net_torch with bf16 rounding at the same points (one full-size
    torch-CPU forward, ~1 min on the box's 16 cores; skipped when the host cannot hold its ~40 GB of fp32 activations).
    Gate 2e-2 of each output's scale, as at 80x80x96."""
    import os
    import mmr
    from oracle import net_np, net_torch
    avail = float("inf")
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) / 1e6
        v = open("/sys/fs/cgroup/memory.max").read().strip()
        if v.isdigit():
            avail = min(avail, int(v) / 1e9)
    except OSError:
        pass
    if avail < 70:
        pytest.skip(f"needs ~40 GB of host memory for the full-size CPU forward, {avail:.0f} GB available")
    enc, dec = [256] * 4, [256] * 6
    rng = np.random.default_rng(31)
    import scipy.ndimage as ndi
    n = lambda v: ((v - v.min()) / (v.max() - v.min())).astype(np.float32)
    mov = n(ndi.gaussian_filter(rng.random(C2, dtype=np.float32), 2.0))[None, ..., None]
    fix = n(ndi.gaussian_filter(rng.random(C2, dtype=np.float32), 2.0))[None, ..., None]
    weights = net_np.init_weights(enc, dec, seed=9, flow_std=1e-2)
    for i in range(1, len(weights), 2):
        weights[i] = (rng.standard_normal(weights[i].shape) * 0.05).astype(np.float32)
    model = mmr.networks.VxmDense(C2, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2, compute_dtype="bf16")
    model.set_weights(weights)
    mmr.ops.PROFILE = []
    try:
        moved, preint = model.predict([mov, fix])
        fams = [f for f, *_ in mmr.ops.PROFILE]
    finally:
        mmr.ops.PROFILE = None
    assert sum(f.endswith("bf16_bn256_upfold") for f in fams) == 3 and sum(f.endswith("bf16_bn256_cinit") for f in fams) == 3, fams
    pos = model.references.pos_flow.cpu().numpy()
    del model
    torch.cuda.empty_cache()
    q = net_torch.bf16_round
    torch.set_num_threads(max(1, min(torch.get_num_threads(), len(os.sched_getaffinity(0)))))
    ref = net_torch.vxm_dense_forward(torch.from_numpy(mov), torch.from_numpy(fix), net_torch.prepare_weights(weights, quant=q),
                                      enc, dec, 5, 2, 2, quant=q)
    ref = {k: v.numpy() for k, v in ref.items()}
    assert np.abs(ref["pos_flow"]).max() > 0.5, "test flow too small to be meaningful"
    errs = {}
    for name, got, exp in (("preint_flow", preint, ref["preint_flow"]), ("pos_flow", pos, ref["pos_flow"]), ("moved", moved, ref["moved"])):
        errs[name] = np.abs(got - exp).max() / np.abs(exp).max()
    print("160x160x192 / 256-feature whole-net parity [bf16, folded]: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    for name, err in errs.items():
        assert err < 2e-2, f"{name}: rel-to-scale err {err:.3e} >= 2e-2"


def test_c3_training_step_same_subject_zero_head(dev):
    """C3 sizes (160^3, 64 features, 26 labels): identical label maps + zero flow head -> every present label scores 2|t.p| / (|t|+|p|) = 1, i.e.
    Dice = -(labels present)/L and the Keras loss (1 + dice) + reg * grad follows; one full step (generators, fwd, bwd, Adam) stays finite."""
    import mmr
    from mmr import synth, training
    S, L = (160, 160, 160), 26
    maps = synth.generate_label_maps(S, L, 1, [16, 32, 64], [8, 16, 32], 1, 3, seed=7, device=dev)
    lab = torch.from_numpy(maps[0][None, ..., None]).to(dev)
    labels = np.arange(L)
    kw = dict(in_shape=S, in_label_list=labels, out_label_list=labels, warp_std=0, warp_res=16, blur_std=1, bias_std=0.3,
              bias_res=40, gamma_std=0.25, device=dev)  # warp_std 0: both generators keep the label map in place
    g1, g2 = synth.labels_to_image(**kw, id=0, seed=1), synth.labels_to_image(**kw, id=1, seed=2)
    model = mmr.networks.VxmDense(S, nb_unet_features=([64] * 4, [64] * 6), int_steps=5, int_resolution=2, svf_resolution=2,
                                  compute_dtype="fp32x3", seed=0)
    w = model.get_weights()
    w[-2][:] = 0
    w[-1][:] = 0
    model.set_weights(w)
    tr = training.SynthMorphTrainer(model, g1, g2, reg_param=1.0, optimizer=training.Adam(1e-4))
    out = tr.train_step(lab, lab)
    present = len(np.unique(maps[0]))  # labels absent from the map contribute 0 (divide_no_nan) to the mean over L
    assert abs(float(out["dice"]) + present / L) < 1e-6, (float(out["dice"]), present)
    assert abs(float(out["loss"]) - (1.0 - present / L)) < 1e-6 and float(out["grad"].abs().max()) == 0.0
    assert torch.isfinite(model._flat).all() and torch.isfinite(tr.gflat).all()
    assert float(tr.gflat.abs().max()) > 0  # the flow head does receive a gradient


def test_c5_ncc_and_bending_full_size(dev):
    import mmr
    from oracle import ops_np as O
    S = (256, 256, 256)
    g = torch.Generator(device="cpu").manual_seed(0)
    I = torch.rand((1,) + S + (1,), generator=g)
    J = torch.rand((1,) + S + (1,), generator=g)
    Id, Jd = I.to(dev), J.to(dev)
    got = float(mmr.ops.ncc_loss(Id, Jd, 9))
    ref = float(O.ncc_loss(I.numpy(), J.numpy(), 9)[0])  # float64 oracle on the full volume
    assert abs(got - ref) < 1e-5 * max(abs(ref), 1e-3), (got, ref)
    assert abs(float(mmr.ops.ncc_loss(Jd, Id, 9)) - got) < 1e-7
    assert abs(float(mmr.ops.ncc_loss(Id, Id, 9)) + 1.0) < 1e-3
    # bending energy of u_c = a_c x^2 + b_c x y (+ affine part) is 4 a_c^2 + 2 b_c^2 at every interior voxel
    a = np.array([0.01, -0.02, 0.005])
    b = np.array([0.03, 0.0, -0.01])
    x = torch.arange(S[0], dtype=torch.float64).view(-1, 1, 1, 1) - 128
    y = torch.arange(S[1], dtype=torch.float64).view(1, -1, 1, 1) - 128
    z = torch.arange(S[2], dtype=torch.float64).view(1, 1, -1, 1) - 128
    u = torch.from_numpy(a) * x * x + torch.from_numpy(b) * x * y + 0.5 * z + 0.25 * y - 3.0
    got = float(mmr.ops.bending_energy(u.float()[None].to(dev)))
    ref = float(np.mean(4 * a * a + 2 * b * b))
    assert abs(got - ref) < 2e-3 * ref, (got, ref)  # fp32 second differences of values up to ~330


def test_c5_ncc_backward_full_size_windows_vs_float64(dev):
    """BASELINE configs[4] as a LOSS at its own size: d(-mean cc) / dI, dJ at 256^3 from the two-pass backward (coefficient pass + one
    box filter), checked against float64 autograd on CROPS.  The gradient at a voxel depends on the images within 8 voxels (every
    9^3 window that contains it), so the oracle on a 72^3 crop is exact for the voxels 8 or more away from the crop's ARTIFICIAL
    faces; faces that are faces of the volume carry the same zero padding in both.  Three crops: a corner (three real faces), the
    centre, and the far corner; the mean's 1 / N is rescaled from the crop to the volume.  Plus exact linearity in gout and the
    one-gradient form (three fields) against the two-gradient form (five fields)."""
    import mmr
    from oracle import grad_torch as G
    S, C = (256, 256, 256), 72
    g = torch.Generator(device="cpu").manual_seed(7)
    I = torch.rand((1,) + S + (1,), generator=g)
    J = 0.6 * I + 0.4 * torch.rand((1,) + S + (1,), generator=g)
    Id, Jd = I.to(dev), J.to(dev)
    dI, dJ = mmr.ops.ncc_loss_bwd(Id, Jd)
    scale = float(dI.abs().max())
    worst = 0.0
    for o in ((0, 0, 0), (92, 92, 92), (S[0] - C, S[1] - C, S[2] - C)):
        sl = tuple(slice(a, a + C) for a in o)
        Ic = I[(slice(None),) + sl].double().requires_grad_(True)
        Jc = J[(slice(None),) + sl].double().requires_grad_(True)
        G.ncc_loss(Ic, Jc).sum().backward()
        k = (C ** 3) / float(S[0] * S[1] * S[2])            # -mean over the crop -> -mean over the volume
        keep = tuple(slice(0 if a == 0 else 8, C if a + C == n else C - 8) for a, n in zip(o, S))     # away from artificial faces
        for got, ref in ((dI, Ic.grad), (dJ, Jc.grad)):
            gw = got[(slice(None),) + sl][(slice(None),) + keep].cpu().double()
            rw = ref[(slice(None),) + keep] * k
            worst = max(worst, float((gw - rw).abs().max()) / scale)
    print(f"NCC backward at 256^3: worst |HIP - float64| / max|grad| over three 72^3 crops = {worst:.2e}")
    assert worst < 2e-5, worst
    gout = torch.tensor([-2.5], device=dev)
    sI, sJ = mmr.ops.ncc_loss_bwd(Id, Jd, gout)
    assert torch.equal(sI, dI * -2.5) or float((sI - dI * -2.5).abs().max()) <= 2e-7 * scale * 2.5
    oI, _ = mmr.ops.ncc_loss_bwd(Id, Jd, want=("I",))
    _, oJ = mmr.ops.ncc_loss_bwd(Id, Jd, want=("J",))
    assert float((oI - dI).abs().max()) <= 1e-6 * scale and float((oJ - dJ).abs().max()) <= 1e-6 * scale


def test_c3_thin_backward_kernels_full_size(dev):
    """The fp32x3 thin-layer backward kernels at C3's size (160^3 x 64, 32 000 tiles over persistent workgroups): the
    flow-head / first-layer weight gradients and the flow-head data gradient against the exact-fp32 kernels (independent
    code paths: different tilings, MFMA shapes and reductions), plus the fused LeakyReLU mask + bias gradient."""
    import mmr
    ops = mmr.ops
    S = (160, 160, 160)
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((1,) + S + (64,), generator=g).to(dev)
    dflow = torch.randn((1,) + S + (3,), generator=g).to(dev)
    src, trg = torch.rand((1,) + S + (1,), generator=g).to(dev), torch.rand((1,) + S + (1,), generator=g).to(dev)
    w = (torch.randn((3, 3, 3, 64, 3), generator=g) * 0.05).to(dev)

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())

    dw3, dwe = torch.zeros((3, 3, 3, 64, 3), device=dev), torch.zeros((3, 3, 3, 64, 3), device=dev)
    ops.conv3d_k3_wgrad(x, dflow, dw3, x3=True)
    ops.conv3d_k3_wgrad(x, dflow, dwe, x3=False)
    assert rel(dw3, dwe) < 2e-5
    d03, d0e = torch.zeros((3, 3, 3, 2, 64), device=dev), torch.zeros((3, 3, 3, 2, 64), device=dev)
    ops.conv3d_k3_cin2_wgrad(src, trg, x, d03, x3=True)
    ops.conv3d_k3_cin2_wgrad(src, trg, x, d0e, x3=False)
    assert rel(d03, d0e) < 2e-5
    dx3 = ops.conv3d_k3_cout3_dgrad(dflow, w, x3=True)
    dxe = ops.conv3d_k3_cout3_dgrad(dflow, w, x3=False)
    assert rel(dx3, dxe) < 2e-5
    del dxe
    db = torch.zeros(64, device=dev)
    fused = ops.conv3d_k3_cout3_dgrad_masked(dflow, w, x, db, x3=True)
    ref = torch.where(x < 0, 0.2 * dx3, dx3)
    assert torch.equal(fused, ref)
    assert rel(db, ref.double().sum(dim=(0, 1, 2, 3)).float()) < 1e-5


def _host_gb():
    avail = float("inf")
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) / 1e6
        v = open("/sys/fs/cgroup/memory.max").read().strip()
        if v.isdigit():
            avail = min(avail, int(v) / 1e9)
    except OSError:
        pass
    return avail


def test_c3_step_gradients_full_size_vs_float64_oracle(dev):
    """BASELINE configs[2]'s own size -- 160^3, enc/dec = 64 (config/config.json:44-45), 26 labels, fp32x3, every folded kernel and
    the pooling-backward epilogue engaged (asserted through the kernel families) -- all 22 gradient tensors of one step
    (train_synthmorph.py:296-308) against the float64 gradient oracle on the HIP evaluation's linear piece: LeakyReLU slopes
    and max-pool routing (``kinks``) AND the tail's floor / clamp cells (``tail_pins``; round 4's tool had the flow head's
    bias gradient at 1.25e-4 because 83 voxels of d loss / d flow sat in a neighbouring interpolation cell).  Gate: north_star's
    1e-4 of each tensor's max.  About five minutes of float64 CPU convolutions and ~45 GB of host memory: skipped below 60 GB
    or with MMR_SKIP_FULLSIZE_GRAD=1."""
    import os
    import sys
    import time
    if os.environ.get("MMR_SKIP_FULLSIZE_GRAD") == "1":
        pytest.skip("MMR_SKIP_FULLSIZE_GRAD=1")
    if _host_gb() < 60:
        pytest.skip(f"needs ~45 GB of host memory for the float64 oracle, {_host_gb():.0f} GB available")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_train as T
    torch.set_num_threads(max(1, min(torch.get_num_threads(), len(os.sched_getaffinity(0)))))
    t0 = time.time()
    rep = []
    T._check_step_gradients(dev, "fp32x3", True, 1e-4, [64] * 4, [64] * 6, shape=(160, 160, 160), L=26, B=1, block=8, int_steps=5,
                            families=("_upfold", "_cinit", "_dgfold", "wgrad_mfma_f32x3_upfold"), report=rep, tail_pins=True)
    errs = {n: v for n, v in rep if n.endswith((" kernel", " bias"))}
    assert len(errs) == 22, sorted(errs)
    worst = max(errs, key=errs.get)
    other = {n: v for n, v in rep if n not in errs}
    print(f"160^3 / 26-label step, 22 gradient tensors vs float64 oracle [fp32x3, folded, tail pinned]: worst {errs[worst]:.2e} ({worst}), "
          f"flow bias {errs['flow bias']:.2e}; d loss / d flow elementwise {other.get('d loss / d flow, elementwise (rel. to max)', float('nan')):.2e}; "
          f"{time.time() - t0:.0f} s")
    for n, v in errs.items():
        assert v < 1e-4, f"{n}: {v:.2e} (all: {errs})"


def test_kink_masks_of_the_hip_forward_match_the_float64_oracles_own(dev):
    """Independent check of what ``kinks`` takes on trust: the gradient oracle differentiates on the LeakyReLU slopes and
    max-pool routing of the HIP forward's activations, so a wrong sign mask in the HIP forward would be inherited.  Here the
    float64 oracle runs the SAME network on its own (no kinks) and its masks are compared with the HIP forward's (exact-fp32
    path, 96 x 96 x 128, config.json's architecture): they may differ only where an activation lies within fp32 rounding of a
    kink -- a fraction of order 1e-5.  Asserted: sign mismatches < 3e-5 of all activations and arg-max mismatches < 3e-5 of
    all pooling windows, in every layer."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mmr
    from mmr import training
    from oracle import grad_torch as G
    from oracle import net_np
    shape, enc, dec = (96, 96, 128), [64] * 4, [64] * 6
    rng = np.random.default_rng(11)
    import scipy.ndimage as ndi
    n = lambda v: ((v - v.min()) / (v.max() - v.min())).astype(np.float32)
    src = n(ndi.gaussian_filter(rng.random(shape, dtype=np.float32), 1.5))[None, ..., None]
    trg = n(ndi.gaussian_filter(rng.random(shape, dtype=np.float32), 1.5))[None, ..., None]
    ws = net_np.init_weights(enc, dec, seed=3, flow_std=3e-2)
    for i in range(1, len(ws), 2):
        ws[i] = (rng.standard_normal(ws[i].shape) * 0.05).astype(np.float32)
    model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2, compute_dtype="fp32")
    model.set_weights(ws)

    class _G:
        L = 2
    tr = training.SynthMorphTrainer(model, _G(), _G())
    tape = []
    tr._forward(torch.from_numpy(src).to(dev), torch.from_numpy(trg).to(dev), tape)
    hip = [(r[4] if r[0] == "conv0" else r[5]) for r in tape if r[0] == "conv0" or (r[0] == "conv" and r[6])]
    torch.set_num_threads(max(1, min(torch.get_num_threads(), len(os.sched_getaffinity(0)))))
    own = []
    with torch.no_grad():
        G.unet(torch.from_numpy(src).double(), torch.from_numpy(trg).double(), [torch.from_numpy(w).double() for w in ws], enc, dec, collect=own)
    assert len(own) == len(hip) == 10
    tot = bad = 0
    wtot = wbad = 0
    worst = 0.0
    for li, (h, o) in enumerate(zip(hip, own)):
        h = h.cpu().double()
        assert h.shape == o.shape
        m = int(((h > 0) != (o > 0)).sum())
        worst = max(worst, m / h.numel())
        bad, tot = bad + m, tot + h.numel()
        if li < len(enc):      # these four feed a MaxPooling3D(2)
            ah, ao = G._windows(h).argmax(-1), G._windows(o).argmax(-1)
            wm = int((ah != ao).sum())
            wbad, wtot = wbad + wm, wtot + ah.numel()
            assert wm < 3e-5 * ah.numel() + 2, (li, wm, ah.numel())
        assert m < 3e-5 * h.numel() + 2, (li, m, h.numel())
        # where the signs agree the values agree to fp32-grade rounding of this layer's scale
        assert float((h - o).abs().max() / o.abs().max()) < 2e-5, li
    print(f"kink masks, HIP exact-fp32 forward vs the float64 oracle's own at {shape}: LeakyReLU sign mismatches {bad} of {tot} "
          f"= {bad / tot:.2e} (worst layer {worst:.2e}), max-pool arg-max mismatches {wbad} of {wtot} = {wbad / max(wtot, 1):.2e}")


def test_c4_cascade_full_size_properties(dev):
    """BASELINE configs[3] at its own size (bids_two_steps_registration.py:311-325,484-499: model 1 on (moving, fixed), model 2 on
    (moved_1, fixed), compose the half-resolution fields, rescale x2, warp the ORIGINAL moving image) -- 160x160x192, enc/dec = 256,
    bf16 -- through size-independent properties: with a zero second flow head the composed field IS stage 1's (compose(a, 0) = a,
    bit for bit) and the final image is stage 1's moved image; with both heads live the result is finite, differs from both
    single stages and is reproducible bit for bit."""
    import mmr
    enc, dec = [256] * 4, [256] * 6
    g = torch.Generator(device="cpu").manual_seed(3)
    lo = torch.rand((2, 1, 20, 20, 24, 1), generator=g)
    mk = lambda t: mmr.ops.resize_trilinear(t.to(dev).contiguous(), C2)
    mov, fix = mk(lo[0]), mk(lo[1])
    m1 = mmr.networks.VxmDense(C2, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2, compute_dtype="bf16", seed=0)
    m2 = mmr.networks.VxmDense(C2, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2, compute_dtype="bf16", seed=1)
    for m, sd in ((m1, 5), (m2, 6)):     # flow heads large enough to move voxels (the init's N(0, 1e-5) would not)
        w = m.get_weights()
        w[-2][:] = (np.random.default_rng(sd).standard_normal(w[-2].shape) * 2e-2).astype(np.float32)
        m.set_weights(w)

    def cascade():
        o1 = m1.forward(mov, fix)
        o2 = m2.forward(o1["y_source"], fix)
        warp = mmr.ops.compose(o1["preint_flow"], o2["preint_flow"])
        full = mmr.ops.rescale_transform(warp, 2)
        return o1, o2, warp, mmr.ops.warp3d(mov, full, "linear", None)
    o1, o2, warp, out = cascade()
    assert torch.isfinite(out).all() and torch.isfinite(warp).all()
    assert float(o1["preint_flow"].abs().max()) > 0.05 and float(o2["preint_flow"].abs().max()) > 0.05
    assert not torch.equal(warp, o1["preint_flow"]) and not torch.equal(out, o1["y_source"])
    _, _, warp_b, out_b = cascade()
    assert torch.equal(warp, warp_b) and torch.equal(out, out_b)          # reproducible bit for bit
    s1_field, s1_moved = o1["preint_flow"].clone(), o1["y_source"].clone()
    w = m2.get_weights()
    w[-2][:] = 0
    w[-1][:] = 0
    m2.set_weights(w)
    o1z, o2z, warp_z, out_z = cascade()
    assert float(o2z["preint_flow"].abs().max()) == 0.0
    assert torch.equal(o1z["preint_flow"], s1_field)
    assert torch.equal(warp_z, s1_field)                                   # compose(a, 0) = 0 + a o id = a, exactly
    # the final image warps the ORIGINAL moving image by rescale(svf_1): NOT the integrated field of stage 1 (3d_reg.py:317-334
    # applies the pre-integration field), so it differs from stage 1's moved image but must equal Transform's result
    tr = mmr.networks.Transform(C2, interp_method="linear", rescale=2)
    ref = tr.predict([mov.cpu().numpy(), s1_field.cpu().numpy()])
    assert np.array_equal(out_z.cpu().numpy(), ref)
