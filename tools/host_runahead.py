"""How far does the host run ahead of the GPU in the training step?  Prints, per step, the host time until
train_step() returns (enqueue only) next to the wall time per step with a final synchronize."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mmr
from mmr import synth, training
dev = torch.device("cuda", 0)
S, L = (160, 160, 160), 26
maps = synth.generate_label_maps(S, L, 2, [16, 32, 64], [8, 16, 32], 1, 3, seed=100, device=dev)
lab = np.arange(L)
kw = dict(in_shape=S, in_label_list=lab, out_label_list=lab, warp_std=3, warp_res=16, blur_std=1, bias_std=0.3, bias_res=40,
          gamma_std=0.25, device=dev)
g1, g2 = synth.labels_to_image(**kw, id=0, seed=11), synth.labels_to_image(**kw, id=1, seed=12)
model = mmr.networks.VxmDense(S, nb_unet_features=([64] * 4, [64] * 6), int_steps=5, int_resolution=2, svf_resolution=2,
                              compute_dtype="fp32x3", device=dev, seed=0)
tr = training.SynthMorphTrainer(model, g1, g2, reg_param=1.0, optimizer=training.Adam(1e-4))
src = torch.from_numpy(maps[0][None, ..., None]).to(dev)
trg = torch.from_numpy(maps[1][None, ..., None]).to(dev)
for _ in range(3):
    tr.train_step(src, trg)
torch.cuda.synchronize()
t0 = time.perf_counter()
host = []
for _ in range(10):
    a = time.perf_counter()
    tr.train_step(src, trg)
    host.append((time.perf_counter() - a) * 1e3)
torch.cuda.synchronize()
print("host enqueue ms per step:", " ".join(f"{h:.1f}" for h in host))
print("wall ms per step:", (time.perf_counter() - t0) / 10 * 1e3)
