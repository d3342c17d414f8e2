"""Soak: N training steps at C3 sizes through fit() with the resident generator; reports the loss trend, the step time
and the device-memory high-water marks at the start / end (leak check)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mmr
from mmr import synth, training, data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
S, L = (160, 160, 160), 26
maps = synth.generate_label_maps(S, L, 6, [16, 32, 64], [8, 16, 32], 1, 3, seed=100, device=dev)
lab = np.arange(L)
kw = dict(in_shape=S, in_label_list=lab, out_label_list=lab, warp_std=3, warp_res=16, blur_std=1, bias_std=0.3, bias_res=40,
          gamma_std=0.25, device=dev)
g1, g2 = synth.labels_to_image(**kw, id=0, seed=11), synth.labels_to_image(**kw, id=1, seed=12)
model = mmr.networks.VxmDense(S, nb_unet_features=([64] * 4, [64] * 6), int_steps=5, int_resolution=2, svf_resolution=2,
                              compute_dtype="fp32x3", device=dev, seed=0)
tr = training.SynthMorphTrainer(model, g1, g2, reg_param=1.0, optimizer=training.Adam(float(os.environ.get("MMR_SOAK_LR", "1e-4"))))
gen = data.gen_synthmorph_eb(list(maps), batch_size=1, same_subj=bool(os.environ.get("MMR_SOAK_SAME")), flip=True, rng=np.random.default_rng(0), device=dev)
h0 = tr.fit(gen, epochs=1, steps_per_epoch=10, verbose=0)
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated(), torch.cuda.max_memory_allocated()
t = time.perf_counter()
hist = tr.fit(gen, epochs=n // 50, steps_per_epoch=50, verbose=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t
m1 = torch.cuda.memory_allocated(), torch.cuda.max_memory_allocated()
print("steps %d  %.2f ms/step" % (len(hist) * 50, dt / (len(hist) * 50) * 1e3))
print("loss per 50 steps:", " ".join("%.4f" % h["loss"] for h in hist))
print("allocated MB start/end: %.0f / %.0f   peak MB start/end: %.0f / %.0f" % (m0[0] / 2**20, m1[0] / 2**20, m0[1] / 2**20, m1[1] / 2**20))
assert all(np.isfinite(h["loss"]) for h in hist) and m1[0] <= m0[0] * 1.05 + (64 << 20)
