"""Wall time per step of SynthMorphTrainer.fit fed by gen_synthmorph_eb: host NumPy batches (two 4 MB H2D copies per
step) against the resident generator, next to the 33 ms resident-input step that bench.py times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mmr
from mmr import synth, training, data
dev = torch.device("cuda", 0)
S, L = (160, 160, 160), 26
maps = synth.generate_label_maps(S, L, 4, [16, 32, 64], [8, 16, 32], 1, 3, seed=100, device=dev)
lab = np.arange(L)
kw = dict(in_shape=S, in_label_list=lab, out_label_list=lab, warp_std=3, warp_res=16, blur_std=1, bias_std=0.3, bias_res=40,
          gamma_std=0.25, device=dev)
g1, g2 = synth.labels_to_image(**kw, id=0, seed=11), synth.labels_to_image(**kw, id=1, seed=12)
model = mmr.networks.VxmDense(S, nb_unet_features=([64] * 4, [64] * 6), int_steps=5, int_resolution=2, svf_resolution=2,
                              compute_dtype="fp32x3", device=dev, seed=0)
tr = training.SynthMorphTrainer(model, g1, g2, reg_param=1.0, optimizer=training.Adam(1e-4))
for name, kwg in (("host NumPy generator", {}), ("resident generator (device=)", {"device": dev})):
    gen = data.gen_synthmorph_eb(list(maps), batch_size=1, same_subj=False, flip=True, rng=np.random.default_rng(0), **kwg)
    tr.fit(gen, epochs=1, steps_per_epoch=3, verbose=0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    tr.fit(gen, epochs=2, steps_per_epoch=15, verbose=0)
    torch.cuda.synchronize()
    print("fit() with the %s: %.2f ms/step" % (name, (time.perf_counter() - t) / 30 * 1e3))
