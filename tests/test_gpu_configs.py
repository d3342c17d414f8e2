"""BASELINE.json configs as end-to-end cases (parity-test cases, not bench lines):
 C1: config/config.json-style SynthMorph training, 2 synthetic label maps, vol 64^3, 1 step;
 C4: bids_two_steps_registration.py cascade (two VxmDense passes + compose) on a synthetic pair."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# the 44 keys of the reference's config/config.json (values adapted to C1: 64^3, 2 maps, 1 epoch of 1 step)
CONFIG_C1 = {
    "model_dir": "models", "log_dir": "logs", "bool_sub_dir": False, "sub_dir": "train_ex", "gen_label_only": False,
    "gen_label": True, "save_label": False, "label_dir": "labels", "zero_borders_maps": False,
    "zero_borders_maps_val": False, "zero_bord_scale": 8, "zero_bord_frac": 0.5, "in_shape": [64, 64, 64],
    "num_labels": 26, "num_maps": 2, "im_scales": [16, 32, 64], "def_scales": [8, 16, 32], "im_max_std": 1,
    "def_max_std": 3, "add_str": "26lab_", "same_subj": True, "blur_std": 1, "gamma": 0.25, "vel_std": 3, "vel_res": 16,
    "bias_std": 0.3, "bias_res": 40, "gpu": "0", "epochs": 1, "batch_size": 1, "train_frac": 0.5, "batch_size_val": 1,
    "save_freq": 1, "bool_init_weights": False, "init_weights": "model.h5", "reg_param": 1, "lr": 1e-4, "init_epoch": 0,
    "verbose": 0, "int_steps": 5, "int_res": 2, "svf_res": 2, "enc": [64, 64, 64, 64], "dec": [64, 64, 64, 64, 64, 64],
}


def test_c1_training_from_reference_config(dev, tmp_path):
    from mmr import networks, training
    assert len(CONFIG_C1) == 44
    cfg = dict(CONFIG_C1, model_dir=str(tmp_path / "models"))
    trainer, hist = training.run_training(cfg, device=dev, seed=0)
    assert len(hist) == 1 and np.isfinite(hist[0]["loss"]) and 0 < hist[0]["loss"] < 2.5
    assert "val_loss" in hist[0] and np.isfinite(hist[0]["val_loss"])
    # ModelCheckpoint-style files: initial save (epoch 0) and the epoch-1 save; reloadable with shape change
    for ep in (0, 1):
        assert (tmp_path / "models" / f"{ep:04d}.h5").exists()
    m = networks.VxmDense.load(str(tmp_path / "models" / "0001.h5"), compute_dtype="fp32")
    assert m.inshape == (64, 64, 64) and len(m.get_weights()) == 22
    m2 = networks.VxmDense((32, 48, 32), nb_unet_features=(cfg["enc"], cfg["dec"]), int_steps=5, int_resolution=2,
                           svf_resolution=2, compute_dtype="fp32")
    m2.set_weights(m.get_weights())  # 3d_reg.py:305-306
    rng = np.random.default_rng(0)
    moved, warp = m2.predict([rng.random((1, 32, 48, 32, 1)), rng.random((1, 32, 48, 32, 1))])
    assert moved.shape == (1, 32, 48, 32, 1) and warp.shape == (1, 16, 24, 16, 3) and np.isfinite(moved).all()


def test_c4_two_step_cascade_matches_oracle(dev):
    """bids_two_steps_registration.py:318-325 (linear / whole volume)."""
    import mmr
    from oracle import net_np
    from oracle import ops_np as O
    shape, enc, dec = (32, 32, 32), [64] * 4, [64] * 6
    rng = np.random.default_rng(1)
    mov = rng.random((1,) + shape + (1,)).astype(np.float32)
    fix = rng.random((1,) + shape + (1,)).astype(np.float32)
    w1 = net_np.init_weights(enc, dec, seed=1, flow_std=2e-2)
    w2 = net_np.init_weights(enc, dec, seed=2, flow_std=2e-2)
    kw = dict(nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2, compute_dtype="fp32")
    m1, m2 = mmr.networks.VxmDense(shape, **kw), mmr.networks.VxmDense(shape, **kw)
    m1.set_weights(w1)
    m2.set_weights(w2)
    moved1, warp1 = m1.predict([mov, fix])
    moved, warp2 = m2.predict([moved1, fix])
    warp = mmr.utils.compose([warp1[0], warp2[0]])
    r1 = net_np.vxm_dense_forward(mov, fix, w1, enc, dec, 5, 2, 2)
    r2 = net_np.vxm_dense_forward(r1["moved"], fix, w2, enc, dec, 5, 2, 2)
    ref_warp = O.compose(r1["preint_flow"][0], r2["preint_flow"][0])
    scale = 1 if warp1[0].shape[0] == shape[0] else 2
    assert scale == 2 and warp.shape == (16, 16, 16, 3)
    assert np.abs(moved - r2["moved"]).max() / np.abs(r2["moved"]).max() < 1e-4
    assert np.abs(warp - ref_warp).max() / np.abs(ref_warp).max() < 1e-4
    # the field the reference would save: rescaled x2 to full resolution (bids_two_steps_registration.py:515)
    full = mmr.utils.rescale_dense_transform(warp[None], scale)
    np.testing.assert_allclose(full[0], O.rescale_dense_transform(ref_warp, 2), atol=2e-4 * np.abs(ref_warp).max() + 1e-6)
    # nearest-neighbour variant applies the composed field with Transform (…:354-355)
    tr = mmr.networks.Transform(shape, interp_method="nearest", rescale=scale, nb_feats=1).predict([mov, warp[None]])
    assert tr.shape == (1,) + shape + (1,)


def test_c4_cascade_bf16_256_features_matches_bf16_oracle(dev):
    """The BENCHMARKED cascade arithmetic (bench.py --workload cascade: bf16, enc/dec = 256) on 32x32x48 against the
    oracle that rounds conv inputs / weights to bf16 at the same points (bids_two_steps_registration.py:318-325):
    stage 2 sees stage 1's moved image, the two half-res fields are composed, rescaled x2 and applied."""
    import mmr
    from oracle import net_np
    from oracle import ops_np as O
    shape, enc, dec = (32, 32, 48), [256] * 4, [256] * 6
    rng = np.random.default_rng(4)
    import scipy.ndimage as ndi
    n = lambda v: ((v - v.min()) / (v.max() - v.min())).astype(np.float32)[None, ..., None]
    mov, fix = n(ndi.gaussian_filter(rng.random(shape), 2.0)), n(ndi.gaussian_filter(rng.random(shape), 2.0))
    w1 = net_np.init_weights(enc, dec, seed=1, flow_std=1e-2)
    w2 = net_np.init_weights(enc, dec, seed=2, flow_std=1e-2)
    kw = dict(nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2, compute_dtype="bf16")
    m1, m2 = mmr.networks.VxmDense(shape, **kw), mmr.networks.VxmDense(shape, **kw)
    m1.set_weights(w1)
    m2.set_weights(w2)
    moved1, warp1 = m1.predict([mov, fix])
    moved, warp2 = m2.predict([moved1, fix])
    warp = mmr.utils.compose([warp1[0], warp2[0]])
    final = mmr.networks.Transform(shape, interp_method="linear", rescale=2, nb_feats=1).predict([mov, warp[None]])
    r1 = net_np.vxm_dense_forward(mov, fix, w1, enc, dec, 5, 2, 2, quant=net_np.bf16_round)
    r2 = net_np.vxm_dense_forward(r1["moved"], fix, w2, enc, dec, 5, 2, 2, quant=net_np.bf16_round)
    ref_warp = O.compose(r1["preint_flow"][0], r2["preint_flow"][0])
    ref_final = O.transform(mov[0], O.rescale_dense_transform(ref_warp, 2), "linear")[None]
    assert np.abs(ref_warp).max() > 0.25, "composed field too small to be meaningful"
    errs = {"moved_1": np.abs(moved1 - r1["moved"]).max() / np.abs(r1["moved"]).max(),
            "moved_2": np.abs(moved - r2["moved"]).max() / np.abs(r2["moved"]).max(),
            "composed_warp": np.abs(warp - ref_warp).max() / np.abs(ref_warp).max(),
            "final": np.abs(final - ref_final).max() / np.abs(ref_final).max()}
    print("C4 cascade bf16 / 256 features: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    for k, v in errs.items():
        assert v < 3e-2, f"{k}: {v:.3e}"


def test_pair_registration_flow_files(dev, tmp_path):
    """3d_reg.py / bids_registration.py flow end to end on synthetic NIfTI files: whole volume and sub-volumes,
    one model and the two-step cascade; outputs written like the reference (moved image + RAI warp, intent 1007)."""
    import scipy.ndimage as ndi
    import mmr
    from mmr import py_utils, registration
    rng = np.random.default_rng(4)
    fx = ndi.gaussian_filter(rng.random((70, 40, 52)), 2.0)
    mv = ndi.shift(fx, (1.5, -1.0, 0.5), order=1) + 0.02 * rng.random(fx.shape)
    aff = np.diag([1.0, 1.0, 1.0, 1.0])
    aff[:3, 3] = [-30, -20, 10]
    py_utils.write_nifti(fx.astype(np.float32), str(tmp_path / "fx.nii.gz"), aff)
    py_utils.write_nifti(mv.astype(np.float32), str(tmp_path / "mv.nii.gz"), aff)
    specs = dict(use_subvol=False, subvol_size=[32, 32, 32], min_perc_overlap=0.1, int_steps=5, int_res=2, svf_res=2,
                 enc=[64] * 4, dec=[64] * 6, warp_interpolation="linear", resample_interpolation="linear")
    m = mmr.networks.VxmDense((16, 16, 16), nb_unet_features=(specs["enc"], specs["dec"]), int_steps=5, int_resolution=2,
                              svf_resolution=2, compute_dtype="fp32", seed=1)
    w = m.get_weights()
    w[-2] = (rng.standard_normal(w[-2].shape) * 2e-2).astype(np.float32)
    m.set_weights(w)
    m.save(str(tmp_path / "m1.h5"))
    out = registration.run_3d_reg(specs, str(tmp_path / "m1.h5"), str(tmp_path / "fx.nii.gz"),
                                  str(tmp_path / "mv.nii.gz"), res_dir=str(tmp_path / "res"), compute_dtype="fp32")
    assert out["fixed_proc"].shape == (64, 32, 48) and out["scale"] == 2 and out["warp"].shape == (32, 16, 24, 3)
    moved, aff2, _ = py_utils.read_nifti(str(tmp_path / "res" / "warped_im.nii.gz"))
    field, _, hdr = py_utils.read_nifti(str(tmp_path / "res" / "deform_field.nii.gz"))
    assert moved.shape == (70, 40, 52) and field.shape == (70, 40, 52, 1, 3) and hdr["intent_code"] == 1007
    np.testing.assert_allclose(aff2, aff)
    assert np.isfinite(moved).all() and np.abs(out["warp"]).max() > 0.05
    # nearest variant re-applies the (x2-rescaled) field with Transform: consistent with the device ops
    outn = registration.register(specs, m, registration.Volume(fx, aff), registration.Volume(mv, aff), warp_interp="nearest",
                                 compute_dtype="fp32")
    chk = mmr.networks.Transform((64, 32, 48), interp_method="nearest", rescale=2, nb_feats=1).predict(
        [outn["moving_proc"].get_fdata()[None, ..., None], outn["warp"][None]])[0, ..., 0]
    np.testing.assert_array_equal(outn["moved"].data, chk)
    # sub-volume path and two-step cascade run and give fields of the same geometry
    specs_sv = dict(specs, use_subvol=True)
    outs = registration.register(specs_sv, m, registration.Volume(fx, aff), registration.Volume(mv, aff), compute_dtype="fp32")
    assert outs["warp"].shape == (32, 16, 24, 3) and np.isfinite(outs["moved"].data).all()
    out2 = registration.register(specs, [m, m], registration.Volume(fx, aff), registration.Volume(mv, aff), compute_dtype="fp32")
    assert out2["warp"].shape == (32, 16, 24, 3) and out2["warp_rai"].shape == (64, 32, 48, 1, 3)


def test_training_cli_runs_the_reference_config(dev, tmp_path):
    """tools/train.py --config-path <44-key JSON> (the reference's command line, train_synthmorph.py:175-185)."""
    import json
    import subprocess
    import sys
    cfg = dict(CONFIG_C1, model_dir=str(tmp_path / "models"), in_shape=[32, 32, 32], im_scales=[8, 16], def_scales=[8, 16],
               enc=[32, 32], dec=[32, 32, 32], num_maps=4, epochs=2)
    path = tmp_path / "config.json"
    path.write_text(json.dumps(cfg))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "train.py"), "--config-path", str(path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["epochs"] == 2 and np.isfinite(out["last"]["loss"])
    assert (tmp_path / "models" / "0002.h5").exists()


def test_two_step_subvol_linear_composes_per_tile(dev):
    """bids_two_steps_registration.py:362-404 (use_subvol + linear): per tile, model 2 sees model 1's own moved tile and
    the two fields are composed per tile BEFORE fusion.  The expected field is assembled here tile by tile from the
    operators; it must differ from the nearest branch's order (fuse stage 1, re-tile, compose globally)."""
    import scipy.ndimage as ndi
    import mmr
    from mmr import registration, tiling, utils
    rng = np.random.default_rng(6)
    fx = ndi.gaussian_filter(rng.random((64, 32, 48)), 2.0)
    mv = ndi.shift(fx, (1.5, -1.0, 0.5), order=1) + 0.02 * rng.random(fx.shape)
    aff = np.eye(4)
    specs = dict(use_subvol=True, subvol_size=[32, 32, 32], min_perc_overlap=0.1, int_steps=5, int_res=2, svf_res=2,
                 enc=[32, 32], dec=[32, 32, 32])
    nets = []
    for seed in (1, 2):
        m = mmr.networks.VxmDense((32, 32, 32), nb_unet_features=(specs["enc"], specs["dec"]), int_steps=5, int_resolution=2,
                                  svf_resolution=2, compute_dtype="fp32", seed=seed)
        w = m.get_weights()
        w[-2] = (rng.standard_normal(w[-2].shape) * 3e-2).astype(np.float32)
        m.set_weights(w)
        nets.append(m)
    out = registration.register(specs, nets, registration.Volume(fx, aff), registration.Volume(mv, aff),
                                warp_interp="linear", compute_dtype="fp32")
    fxp, mvp, tiles_fx, tiles_mv, coords = registration.preprocess(specs, registration.Volume(fx, aff), registration.Volume(mv, aff))
    assert len(coords) >= 2
    fields = []
    for f, m in zip(tiles_fx, tiles_mv):
        moved_t, w1 = nets[0].predict([m[None, ..., None], f[None, ..., None]])
        _, w2 = nets[1].predict([moved_t, f[None, ..., None]])
        fields.append(np.asarray(utils.compose([w1[0], w2[0]])))
    exp = tiling.fuse_subvolume_fields((16, 16, 16), tuple(s // 2 for s in fxp.shape), [tuple(c // 2 for c in cd) for cd in coords], fields)
    np.testing.assert_array_equal(out["warp"], exp)
    assert np.abs(exp).max() > 0.05
    moved = mmr.networks.Transform(fxp.shape, interp_method="linear", rescale=2, nb_feats=1).predict(
        [mvp.get_fdata()[None, ..., None], exp[None]])[0, ..., 0]
    np.testing.assert_array_equal(out["moved"].data, moved)
    outn = registration.register(specs, nets, registration.Volume(fx, aff), registration.Volume(mv, aff),
                                 warp_interp="nearest", compute_dtype="fp32")
    assert np.abs(outn["warp"] - out["warp"]).max() > 1e-4   # the two branches are different computations
