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


def test_c2_largest_conv_seams_linearity_and_window_vs_oracle(dev):
    """dec_final_0 of C2: concat(up2(256 ch @ 80x80x96), 256 ch @ 160x160x192) -> 256, bf16, 19 200 workgroups."""
    import mmr
    from oracle.cbind import conv3d_same
    ops = mmr.ops
    C = 256
    half = tuple(s // 2 for s in C2)
    w = (torch.randn((3, 3, 3, 2 * C, C), generator=torch.Generator().manual_seed(1)) * 0.02).to(dev)
    wq = w.to(torch.bfloat16).float()
    bias = torch.linspace(-1, 1, C, device=dev)
    wp = ops.pack_conv_weights(w, torch.bfloat16)

    # (1) per-channel-constant inputs: every voxel at least one step inside the volume sees the same 27 x 512 products
    #     in the same order -> bitwise identical outputs across all tile / wave / lane seams
    c0 = torch.linspace(-1, 1, C, device=dev).to(torch.bfloat16)
    c1 = torch.linspace(0.5, -0.5, C, device=dev).to(torch.bfloat16)
    in0 = c0.expand((1,) + half + (C,)).contiguous()
    in1 = c1.expand((1,) + C2 + (C,)).contiguous()
    y = ops.conv3d_k3(in0, wp, bias, C, in1=in1, up0=True, leaky=True)
    inner = y[0, 1:-1, 1:-1, 1:-1]
    assert torch.equal(inner, inner[:1, :1, :1].expand_as(inner))
    ref = (torch.cat([c0.float(), c1.float()]).double() @ wq.double().sum(dim=(0, 1, 2)) + bias.double())
    ref = torch.where(ref < 0, 0.2 * ref, ref)
    assert (inner[0, 0, 0].double() - ref).abs().max() < 2e-2 * ref.abs().max()
    del in0, in1, y, inner

    # (2) random inputs: exact linearity under a power-of-two scale, and an 18^3 window in the far corner against the
    #     C oracle run on the cropped (haloed) input
    in0 = _bf16_randn((1,) + half + (C,), dev, 2)
    in1 = _bf16_randn((1,) + C2 + (C,), dev, 3)
    y = ops.conv3d_k3(in0, wp, None, C, in1=in1, up0=True, leaky=False, out_f32=True)
    y2 = ops.conv3d_k3(in0 * 2, wp, None, C, in1=in1 * 2, up0=True, leaky=False, out_f32=True)
    assert torch.equal(y2, 2 * y)
    x0, y0, z0, n = 136, 140, 170, 20  # window [x0, x0+n) incl. a 1-voxel halo; even origin for the x2 upsample
    up = in0[0, x0 // 2:(x0 + n) // 2, y0 // 2:(y0 + n) // 2, z0 // 2:(z0 + n) // 2].float()
    up = up.repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2)
    crop = torch.cat([up, in1[0, x0:x0 + n, y0:y0 + n, z0:z0 + n].float()], -1).cpu().numpy()[None]
    ref = conv3d_same(crop, wq.cpu().numpy(), np.zeros(C, np.float32), leaky=False)[0, 1:-1, 1:-1, 1:-1]
    got = y[0, x0 + 1:x0 + n - 1, y0 + 1:y0 + n - 1, z0 + 1:z0 + n - 1].cpu().numpy()
    assert np.abs(got - ref).max() < 1e-4 * np.abs(ref).max()


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
