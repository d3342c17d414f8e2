"""End-to-end walk through the reference's workflow on synthetic data: train (train_synthmorph.py) -> Keras .h5
checkpoint -> VxmDense.load + rebuild at a new shape + set_weights (3d_reg.py:277,297-306) -> predict -> warp the
moving label map with nearest interpolation (3d_reg.py:331-334) -> folding % (eval_reg_with_jacobian.py) and per-label
Dice before / after (eval_reg_on_sc_seg.py).  Usage: python tools/demo_train_then_register.py [steps]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mmr
from mmr import synth, training, data, networks, evaluation, utils
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
dev = torch.device("cuda", 0)
S, L = (96, 96, 96), 8
maps = synth.generate_label_maps(S, L, 8, [16, 32], [8, 16], 1, 3, seed=1, device=dev)
lab = np.arange(L)
kw = dict(in_shape=S, in_label_list=lab, out_label_list=lab, warp_std=5, warp_res=16, blur_std=1, bias_std=0.3, bias_res=40,
          gamma_std=0.25, device=dev)
g1, g2 = synth.labels_to_image(**kw, id=0, seed=11), synth.labels_to_image(**kw, id=1, seed=12)
enc, dec = [64] * 4, [64] * 6
model = networks.VxmDense(S, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2, device=dev, seed=0)
tr = training.SynthMorphTrainer(model, g1, g2, reg_param=1.0, optimizer=training.Adam(1e-4))
gen = data.gen_synthmorph_eb(list(maps[:6]), batch_size=1, same_subj=True, flip=True, random_zero_borders=False,
                             rng=np.random.default_rng(0), device=dev)
t = time.perf_counter()
hist = tr.fit(gen, epochs=steps // 100, steps_per_epoch=100, verbose=0)
print("trained %d steps in %.1f s; loss %.3f -> %.3f" % (len(hist) * 100, time.perf_counter() - t, hist[0]["loss"], hist[-1]["loss"]))
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "0001.h5")
    model.save(path)
    reg = networks.VxmDense.load(path, input_model=None)                       # 3d_reg.py:277
    net = networks.VxmDense(S, int_steps=5, int_resolution=2, svf_resolution=2, nb_unet_features=(enc, dec))
    net.set_weights(reg.get_weights())                                         # 3d_reg.py:305-306
# held-out subject: two differently deformed / contrasted renderings of label map 7
src = torch.from_numpy(maps[7][None, ..., None]).to(dev)
a, b = g1.generate(src, want_onehot=False), g2.generate(src, want_onehot=False)
mov, fix = a["image"].cpu().numpy().astype(np.float64), b["image"].cpu().numpy().astype(np.float64)
moved, preint = net.predict([mov, fix])
full = utils.rescale_dense_transform(net.references.pos_flow.cpu().numpy(), 1)  # integrated full-res warp
warped_lab = networks.Transform(S, interp_method="nearest", nb_feats=1).predict([a["labels"].cpu().numpy().astype(np.float32), full])
lm, lf, lw = a["labels"].cpu().numpy()[0, ..., 0], b["labels"].cpu().numpy()[0, ..., 0], warped_lab[0, ..., 0].astype(np.uint8)
before = [evaluation.overlap_metrics(lf == l, lm == l)["dice"] for l in range(1, L) if (lf == l).any() and (lm == l).any()]
after = [evaluation.overlap_metrics(lf == l, lw == l)["dice"] for l in range(1, L) if (lf == l).any() and (lw == l).any()]
jd = evaluation.jacobian_determinant(full[0])
print("mean label Dice before %.3f  after registration %.3f" % (np.mean(before), np.mean(after)))
print("folding: %.3f %% of voxels with det(J) < 0 (median det %.3f)" % (jd["percentage_negative"], jd["median"]))
assert np.mean(after) > np.mean(before)
