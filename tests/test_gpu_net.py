"""End-to-end VxmDense forward (HIP) vs the oracle assembly of the same graph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(rng, shape):
    """Smooth-ish random image pair in [0,1]."""
    import scipy.ndimage as ndi
    a = ndi.gaussian_filter(rng.random(shape), 2.0)
    b = ndi.gaussian_filter(rng.random(shape), 2.0)
    n = lambda v: ((v - v.min()) / (v.max() - v.min())).astype(np.float32)
    return n(a)[None, ..., None], n(b)[None, ..., None]


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("fp32x3", 1e-4), ("bf16", 3e-2)])
def test_vxmdense_forward_matches_oracle(dev, dtype, tol):
    import mmr
    from oracle import net_np
    shape, enc, dec = (32, 32, 48), [64] * 4, [64] * 6
    rng = np.random.default_rng(0)
    mov, fix = _pair(rng, shape)
    # flow-head std large enough for voxel-scale displacements (the Keras init 1e-5 gives ~0 flow)
    weights = net_np.init_weights(enc, dec, seed=1, flow_std=2e-2)
    for i in range(1, len(weights), 2):
        weights[i] = (rng.standard_normal(weights[i].shape) * 0.05).astype(np.float32)
    model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2,
                                  svf_resolution=2, compute_dtype=dtype)
    model.set_weights(weights)
    moved, preint = model.predict([mov, fix])
    pos = model.references.pos_flow.cpu().numpy()
    quant = net_np.bf16_round if dtype == "bf16" else None
    ref = net_np.vxm_dense_forward(mov, fix, weights, enc, dec, 5, 2, 2, quant=quant)
    assert preint.shape == (1, 16, 16, 24, 3) and moved.shape == (1,) + shape + (1,)
    assert np.abs(ref["pos_flow"]).max() > 0.5, "test flow too small to be meaningful"
    for name, got, exp in (("preint_flow", preint, ref["preint_flow"]), ("pos_flow", pos, ref["pos_flow"]),
                           ("moved", moved, ref["moved"])):
        err = np.abs(got - exp).max() / np.abs(exp).max()
        assert err < tol, f"{name}: rel-to-scale err {err:.3e} >= {tol}"


def test_weight_transplant_is_shape_agnostic(dev):
    """3d_reg.py:305-306: rebuild at the runtime shape, then set_weights(get_weights())."""
    import mmr
    feats = ([64] * 4, [64] * 6)
    m1 = mmr.networks.VxmDense((16, 16, 16), nb_unet_features=feats, int_steps=5, int_resolution=2, svf_resolution=2,
                               compute_dtype="fp32", seed=3)
    m2 = mmr.networks.VxmDense((32, 16, 48), nb_unet_features=feats, int_steps=5, int_resolution=2, svf_resolution=2,
                               compute_dtype="fp32", seed=4)
    w = m1.get_weights()
    assert len(w) == 22 and w[0].shape == (3, 3, 3, 2, 64) and w[-2].shape == (3, 3, 3, 64, 3)
    m2.set_weights(w)
    for a, b in zip(m2.get_weights(), w):
        assert np.array_equal(a, b)
    rng = np.random.default_rng(0)
    moved, warp = m2.predict([rng.random((1, 32, 16, 48, 1)), rng.random((1, 32, 16, 48, 1))])
    assert moved.shape == (1, 32, 16, 48, 1) and warp.shape == (1, 16, 8, 24, 3)
    assert moved.dtype == np.float32


def test_predict_batch_overlaps_copies_and_equals_single_pairs(dev):
    """``predict`` on a batch of pairs (Keras semantics, 3d_reg.py:310-314 calls it with one): pairs are forwarded one at a time
    with the neighbouring pairs' host <-> device copies on a side stream; every pair's outputs equal the one-pair call's bit
    for bit, float64 inputs included."""
    import mmr
    shape, feats = (16, 32, 16), ([32, 32], [32, 32, 32])
    m = mmr.networks.VxmDense(shape, nb_unet_features=feats, int_steps=5, int_resolution=2, svf_resolution=2, compute_dtype="fp32x3", seed=2)
    w = m.get_weights()
    w[-2] = (np.random.default_rng(1).standard_normal(w[-2].shape) * 3e-2).astype(np.float32)
    m.set_weights(w)
    rng = np.random.default_rng(0)
    mov, fix = rng.random((5,) + shape + (1,)), rng.random((5,) + shape + (1,))       # float64, as nibabel hands them over
    moved, warp = m.predict([mov, fix])
    assert moved.shape == (5,) + shape + (1,) and warp.shape == (5, 8, 16, 8, 3) and moved.dtype == np.float32
    for b in range(5):
        m1, w1 = m.predict([mov[b:b + 1], fix[b:b + 1]])
        assert np.array_equal(moved[b:b + 1], m1) and np.array_equal(warp[b:b + 1], w1), b
    again = m.predict([mov, fix])
    assert np.array_equal(again[0], moved) and np.array_equal(again[1], warp)
    assert np.abs(warp).max() > 0.05 and not np.array_equal(moved[0], moved[1])


def test_transform_network(dev):
    import mmr
    from oracle import ops_np as O
    rng = np.random.default_rng(5)
    vol = rng.random((1, 8, 12, 10, 1))
    trf = (rng.standard_normal((1, 4, 6, 5, 3)) * 1.5).astype(np.float32)
    for method in ("linear", "nearest"):
        got = mmr.networks.Transform((8, 12, 10), interp_method=method, rescale=2, nb_feats=1).predict([vol, trf])
        full = O.rescale_dense_transform(trf[0], 2)
        ref = O.transform(vol[0].astype(np.float32), full, method)[None]
        np.testing.assert_allclose(got, ref, atol=2e-6)


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 3e-2)])
def test_default_voxelmorph_features_via_channel_padding(dev, dtype, tol):
    """voxelmorph's default widths (16/32) are not multiples of the MFMA slice: they run zero-padded and must
    give the same result as the oracle at the logical widths; get/set_weights exchange the logical arrays."""
    import mmr
    from oracle import net_np
    shape = (16, 16, 16)
    enc, dec = [16, 32, 32, 32], [32, 32, 32, 32, 32, 16, 16]
    rng = np.random.default_rng(3)
    mov, fix = _pair(rng, shape)
    weights = net_np.init_weights(enc, dec, seed=2, flow_std=3e-2)
    for i in range(1, len(weights), 2):
        weights[i] = (rng.standard_normal(weights[i].shape) * 0.05).astype(np.float32)
    model = mmr.networks.VxmDense(shape, int_steps=7, int_resolution=2, svf_resolution=1, compute_dtype=dtype)  # defaults
    assert [p[2] for p in model.plan] == enc + dec + [3]
    model.set_weights(weights)
    for a, b in zip(model.get_weights(), weights):
        assert a.shape == b.shape and np.array_equal(a, b)
    moved, preint = model.predict([mov, fix])
    quant = net_np.bf16_round if dtype == "bf16" else None
    ref = net_np.vxm_dense_forward(mov, fix, weights, enc, dec, 7, 2, 1, quant=quant)
    assert preint.shape == ref["preint_flow"].shape == (1, 8, 8, 8, 3)
    for got, exp in ((preint, ref["preint_flow"]), (moved, ref["moved"])):
        assert np.abs(got - exp).max() / np.abs(exp).max() < tol
    assert model.count_params() == sum(w.size for w in weights)


@pytest.mark.parametrize("dtype,tol", [("fp32x3", 1e-4), ("bf16", 2e-2)])
def test_vxmdense_256_features_matches_oracle(dev, dtype, tol):
    """The benchmarked width (BASELINE configs[1]: enc/dec = 256, the BN=256 MFMA tiles, split-K coarse levels)
    as a WHOLE forward against the oracle, not per layer: 3d_reg.py:297-314 builds exactly this network.
    fp32x3 (the API default) must hold north_star's 1e-4; bf16 is compared with the oracle rounding its conv
    inputs / weights to bf16 at the same points, bound 2e-2 of each output's scale over the 10-conv-deep net."""
    import mmr
    from oracle import net_np
    shape, enc, dec = (32, 32, 48), [256] * 4, [256] * 6
    rng = np.random.default_rng(11)
    mov, fix = _pair(rng, shape)
    weights = net_np.init_weights(enc, dec, seed=5, flow_std=1e-2)
    for i in range(1, len(weights), 2):
        weights[i] = (rng.standard_normal(weights[i].shape) * 0.05).astype(np.float32)
    model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2,
                                  svf_resolution=2, compute_dtype=dtype)
    model.set_weights(weights)
    moved, preint = model.predict([mov, fix])
    pos = model.references.pos_flow.cpu().numpy()
    quant = net_np.bf16_round if dtype == "bf16" else None
    ref = net_np.vxm_dense_forward(mov, fix, weights, enc, dec, 5, 2, 2, quant=quant)
    assert np.abs(ref["pos_flow"]).max() > 0.5, "test flow too small to be meaningful"
    errs = {}
    for name, got, exp in (("preint_flow", preint, ref["preint_flow"]), ("pos_flow", pos, ref["pos_flow"]),
                           ("moved", moved, ref["moved"])):
        errs[name] = np.abs(got - exp).max() / np.abs(exp).max()
    print(f"256-feature whole-net parity [{dtype}]: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    for name, err in errs.items():
        assert err < tol, f"{name}: rel-to-scale err {err:.3e} >= {tol}"


def test_vxmdense_256_features_80x80x96_matches_torch_oracle(dev):
    """Whole-network parity where the benchmark's launch geometry starts: 80x80x96 (an eighth of BASELINE configs[1]'s
    voxels), enc/dec = 256 -- 2 400 workgroups per full-res conv launch, no split-K at full resolution, split-K only at
    the two coarsest levels, an x-segmented flow head -- in the API's default arithmetic (fp32x3) against
    oracle/net_torch.vxm_dense_forward (torch-CPU fp32; itself checked against oracle/net_np.py on CPU).  north_star's
    1e-4 on every output."""
    import mmr
    from oracle import net_np, net_torch
    shape, enc, dec = (80, 80, 96), [256] * 4, [256] * 6
    rng = np.random.default_rng(21)
    mov, fix = _pair(rng, shape)
    weights = net_np.init_weights(enc, dec, seed=7, flow_std=1e-2)
    for i in range(1, len(weights), 2):
        weights[i] = (rng.standard_normal(weights[i].shape) * 0.05).astype(np.float32)
    model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2,
                                  svf_resolution=2, compute_dtype="fp32x3")
    model.set_weights(weights)
    moved, preint = model.predict([mov, fix])
    pos = model.references.pos_flow.cpu().numpy()
    ref = net_torch.vxm_dense_forward(torch.from_numpy(mov), torch.from_numpy(fix), net_torch.prepare_weights(weights),
                                      enc, dec, 5, 2, 2)
    ref = {k: v.numpy() for k, v in ref.items()}
    assert np.abs(ref["pos_flow"]).max() > 0.5, "test flow too small to be meaningful"
    errs = {}
    for name, got, exp in (("preint_flow", preint, ref["preint_flow"]), ("pos_flow", pos, ref["pos_flow"]),
                           ("moved", moved, ref["moved"])):
        errs[name] = np.abs(got - exp).max() / np.abs(exp).max()
    print("80x80x96 / 256-feature whole-net parity [fp32x3]: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    for name, err in errs.items():
        assert err < 1e-4, f"{name}: rel-to-scale err {err:.3e} >= 1e-4"


def test_vxmdense_256_features_80x80x96_bf16_folded_matches_rounding_oracle(dev):
    """The arithmetic the benchmark line is quoted in -- bf16, enc/dec = 256 (3d_reg.py:297-305 with
    config_inference.json:8-9) -- as a WHOLE forward at a size where networks.py::_conv folds decoder layers (asserted
    through ops.PROFILE: the `_upfold` and `_cinit` launches with the half partial ran), against oracle/net_torch with
    bf16 rounding at the same points (conv inputs, kernels, LeakyReLU outputs; checked against oracle/net_np's on CPU).
    Gate 2e-2 of each output's scale over the 10-conv-deep network, as for the unfolded 32x32x48 case."""
    import mmr
    from oracle import net_np, net_torch
    shape, enc, dec = (80, 80, 96), [256] * 4, [256] * 6
    rng = np.random.default_rng(21)
    mov, fix = _pair(rng, shape)
    weights = net_np.init_weights(enc, dec, seed=7, flow_std=1e-2)
    for i in range(1, len(weights), 2):
        weights[i] = (rng.standard_normal(weights[i].shape) * 0.05).astype(np.float32)
    model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2,
                                  svf_resolution=2, compute_dtype="bf16")
    model.set_weights(weights)
    mmr.ops.PROFILE = []
    try:
        moved, preint = model.predict([mov, fix])
        fams = [f for f, *_ in mmr.ops.PROFILE]
    finally:
        mmr.ops.PROFILE = None
    assert sum(f.endswith("bf16_bn256_upfold") for f in fams) >= 1 and sum(f.endswith("bf16_bn256_cinit") for f in fams) >= 1, fams
    pos = model.references.pos_flow.cpu().numpy()
    q = net_torch.bf16_round
    ref = net_torch.vxm_dense_forward(torch.from_numpy(mov), torch.from_numpy(fix), net_torch.prepare_weights(weights, quant=q),
                                      enc, dec, 5, 2, 2, quant=q)
    ref = {k: v.numpy() for k, v in ref.items()}
    assert np.abs(ref["pos_flow"]).max() > 0.5, "test flow too small to be meaningful"
    errs = {}
    for name, got, exp in (("preint_flow", preint, ref["preint_flow"]), ("pos_flow", pos, ref["pos_flow"]),
                           ("moved", moved, ref["moved"])):
        errs[name] = np.abs(got - exp).max() / np.abs(exp).max()
    print("80x80x96 / 256-feature whole-net parity [bf16, folded]: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    for name, err in errs.items():
        assert err < 2e-2, f"{name}: rel-to-scale err {err:.3e} >= 2e-2"


def test_input_model_wires_the_generator_pair(dev):
    """train_synthmorph.py:288-296: ``VxmDense(..., input_model=Model(labels -> (ima_1, ima_2)))`` -- the model's inputs are
    the two label maps, its source / target the generators' images.  Same seeds, same draws: predicting through the
    input_model equals rendering the pair by hand and predicting with a plain model; the trainer picks the generators up
    from the model."""
    import mmr
    from mmr import synth, training
    shape, L = (16, 16, 32), 4
    rng = np.random.default_rng(0)
    lab = np.repeat(np.repeat(np.repeat(rng.integers(0, L, (1, 4, 4, 8)), 4, 1), 4, 2), 4, 3).astype(np.uint8)[..., None]
    kw = dict(in_shape=shape, in_label_list=np.arange(L), out_label_list=np.arange(L), warp_std=1, warp_res=8, blur_std=1,
              bias_std=0.3, bias_res=8, gamma_std=0.25)
    feats = ([32, 32], [32, 32, 32])
    mk = lambda: (synth.labels_to_image(**kw, id=0, seed=1), synth.labels_to_image(**kw, id=1, seed=2))
    g1, g2 = mk()
    im = mmr.networks.InputModel(g1, g2)
    assert im.inputs == ["labels_input_0", "labels_input_1"]
    m = mmr.networks.VxmDense(shape, nb_unet_features=feats, int_steps=3, int_resolution=2, svf_resolution=2,
                              input_model=im, compute_dtype="fp32", seed=3)
    w = m.get_weights()
    w[-2] = (rng.standard_normal(w[-2].shape) * 3e-2).astype(np.float32)
    m.set_weights(w)
    moved, warp = m.predict([lab, lab])
    h1, h2 = mk()
    a, b = h1.generate(lab)["image"], h2.generate(lab)["image"]
    plain = mmr.networks.VxmDense(shape, nb_unet_features=feats, int_steps=3, int_resolution=2, svf_resolution=2,
                                  compute_dtype="fp32", seed=3)
    plain.set_weights(w)
    moved2, warp2 = plain.predict([a.cpu().numpy(), b.cpu().numpy()])
    assert np.array_equal(moved, moved2) and np.array_equal(warp, warp2) and np.abs(warp).max() > 0
    assert im.maps[0].dtype == torch.uint8 and tuple(im.maps[0].shape) == (1,) + shape + (1,)
    tr = training.SynthMorphTrainer(m, reg_param=0.5, optimizer=training.Adam(1e-3))
    assert tr.gen_1 is g1 and tr.gen_2 is g2 and np.isfinite(float(tr.train_step(lab, lab)["loss"]))
    with pytest.raises(TypeError):
        mmr.networks.VxmDense(shape, nb_unet_features=feats, input_model=object())


def test_predict_host_handover_every_input_kind_and_fallback(dev):
    """mmr.hostio (3d_reg.py:310-314's hand-over): the caller's pages pinned in place and cast on the GPU must give what Keras'
    host-side cast gives, for every dtype / layout a caller can hold -- float64, float32, uint8, int16, another dtype (int32 ->
    float64 on the host), Fortran-ordered and strided views, an unaligned buffer -- through each strategy ('register', the
    'staging' fallback the runtime's refusal leads to, 'torch'), and the outputs of each output strategy are the device values bit
    for bit.  predict() itself is then identical whichever strategy carried the volumes."""
    import mmr
    from mmr import hostio
    rng = np.random.default_rng(3)
    shape = (1, 16, 32, 16, 1)
    base = rng.random(shape) * 200.0
    raw = np.frombuffer(bytearray(base.nbytes + 8), dtype=np.uint8)[8:8 + base.nbytes].view(np.float64).reshape(shape)   # 8-B aligned only
    raw[...] = base
    cases = {"float64": base, "float32": base.astype(np.float32), "uint8": base.astype(np.uint8), "int16": (base * 50).astype(np.int16),
             "int32": (base * 1000).astype(np.int32), "fortran": np.asfortranarray(base), "strided": np.repeat(base, 2, axis=3)[:, :, :, ::2],
             "unaligned": raw, "readonly": base.copy()}
    cases["readonly"].setflags(write=False)
    for name, a in cases.items():
        want = torch.from_numpy(np.ascontiguousarray(a).astype(np.float64)).float()
        for mode in ("register", "staging", "torch"):
            got = hostio.to_device_f32(a, dev, mode=mode).cpu()
            assert got.shape == want.shape and torch.equal(got, want), (name, mode)
        pair = hostio.pair_to_device([a, a], dev)
        assert torch.equal(pair[0].cpu(), want) and torch.equal(pair[1].cpu(), want), name
        assert hostio.LAST["in"]["mode"] == "staging" and not hostio._LIVE, name       # small heap arrays are never pinned in place
        # the same values on pages of their own (what a > 32 MiB volume is by construction): pinned in place.  The SAME buffer twice
        # is pinned once, counted twice, released once -- nothing stays pinned behind the caller's back
        ac = np.ascontiguousarray(a)
        if ac.dtype in hostio._CODES:
            ex = hostio.exclusive_empty(ac.shape, ac.dtype)
            ex[...] = ac
            assert hostio.registrable(ex) and not hostio.registrable(ac)
            assert torch.equal(hostio.to_device_f32(ex, dev, mode="register").cpu(), want), name
            pair = hostio.pair_to_device([ex, ex], dev)
            assert torch.equal(pair[0].cpu(), want) and torch.equal(pair[1].cpu(), want), name
            assert hostio.LAST["in"]["mode"] == "register" and not hostio._LIVE, name
    t = torch.randn((1, 16, 32, 16, 3), device=dev)
    for mode in ("register", "staging", "torch"):
        assert np.array_equal(hostio.to_host(t, mode=mode), t.cpu().numpy()), mode
        outs = hostio.many_to_host([t, t[..., :1].contiguous()], mode=mode)
        assert np.array_equal(outs[0], t.cpu().numpy()) and np.array_equal(outs[1], t[..., :1].cpu().numpy()), mode
    m = mmr.networks.VxmDense((16, 32, 16), nb_unet_features=([32, 32], [32, 32, 32]), int_steps=5, int_resolution=2, svf_resolution=2,
                              compute_dtype="fp32x3", seed=2)
    w = m.get_weights()
    w[-2] = (np.random.default_rng(1).standard_normal(w[-2].shape) * 3e-2).astype(np.float32)
    m.set_weights(w)
    mov, fix = rng.random(shape), rng.random(shape)
    same = m.predict([mov, mov])                         # moving = fixed, one array
    assert not hostio._LIVE and np.isfinite(same[0]).all()
    big = hostio.exclusive_empty(shape, np.float64)      # stands in for a volume above the mmap threshold: pinned in place
    big[...] = mov
    same2 = m.predict([big, big])                        # one registration shared by both inputs, not doubled
    assert hostio.LAST["in"]["mode"] == "register" and not hostio._LIVE
    assert np.array_equal(same2[0], same[0]) and np.array_equal(same2[1], same[1])
    keep = (hostio.MODE_IN, hostio.MODE_OUT)
    res = {}
    try:
        for mi, mo in (("register", "register"), ("staging", "staging"), ("torch", "torch")):
            hostio.MODE_IN, hostio.MODE_OUT = mi, mo
            res[mi] = m.predict([mov, fix])
    finally:
        hostio.MODE_IN, hostio.MODE_OUT = keep
    for k in ("staging", "torch"):
        assert np.array_equal(res[k][0], res["register"][0]) and np.array_equal(res[k][1], res["register"][1]), k
    # a buffer that is already pinned by the caller is shared (counted), not pinned a second time
    with hostio.Registered(big) as r:
        assert r.ok
        again = m.predict([big, big])
        assert hostio._LIVE[int(big.ctypes.data)][0] == 1      # predict's references are gone, the caller's is still there
    assert np.array_equal(again[0], same[0]) and not hostio._LIVE
