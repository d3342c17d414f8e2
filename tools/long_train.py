"""Longer SynthMorph run at C3 sizes with the reference's settings (config/config.json: same_subj true, lr 1e-4,
reg_param 1): prints the mean loss of every 500 steps.  python tools/long_train.py [steps] [num_maps] [same_subj 0|1]
(different-subject pairs of unrelated random label maps do not train -- 28 000 steps stayed at 0.905 -- which is why the
reference pairs two renderings of the SAME map.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mmr
from mmr import synth, training, data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 12
same = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
dev = torch.device("cuda", 0)
S, L = (160, 160, 160), 26
maps = synth.generate_label_maps(S, L, nm, [16, 32, 64], [8, 16, 32], 1, 3, seed=100, device=dev)
lab = np.arange(L)
kw = dict(in_shape=S, in_label_list=lab, out_label_list=lab, warp_std=3, warp_res=16, blur_std=1, bias_std=0.3, bias_res=40,
          gamma_std=0.25, device=dev)
g1, g2 = synth.labels_to_image(**kw, id=0, seed=11), synth.labels_to_image(**kw, id=1, seed=12)
model = mmr.networks.VxmDense(S, nb_unet_features=([64] * 4, [64] * 6), int_steps=5, int_resolution=2, svf_resolution=2,
                              compute_dtype="fp32x3", device=dev, seed=0)
tr = training.SynthMorphTrainer(model, g1, g2, reg_param=1.0, optimizer=training.Adam(1e-4))
gen = data.gen_synthmorph_eb(list(maps), batch_size=1, same_subj=same, flip=True, random_zero_borders=False,
                             rng=np.random.default_rng(0), device=dev)
t0 = time.perf_counter()
def log(rec):
    print("steps %6d  loss %.4f  %.1f ms/step  elapsed %.0f s" % (rec["epoch"] * 500, rec["loss"], rec["s_per_step"] * 1e3,
                                                                  time.perf_counter() - t0), flush=True)
tr.fit(gen, epochs=n // 500, steps_per_epoch=500, verbose=0, log=log)
