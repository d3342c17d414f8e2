#!/usr/bin/env python
"""Per-launch table of the matrix-core kernels of one C3 training step (HIP events around each launch, ops.PROFILE):
family, (channels, voxels) tag, ms, algorithmic TFLOP/s and its fraction of the fp32x3 ceiling (2.5 PF / 3).
  python tools/train_layer_table.py [steps]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmr
from mmr import synth, training
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
shape, L, feats = (160, 160, 160), 26, 64
enc, dec = [feats] * 4, [feats] * 6
maps = synth.generate_label_maps(shape, L, 1, [16, 32, 64], [8, 16, 32], 1, 3, seed=100, device=dev)
kw = dict(in_shape=shape, in_label_list=np.arange(L), out_label_list=np.arange(L), warp_std=3, warp_res=16, blur_std=1,
          bias_std=0.3, bias_res=40, gamma_std=0.25, device=dev)
src = torch.from_numpy(maps[0][None, ..., None]).to(dev)
g1, g2 = synth.labels_to_image(**kw, id=0, seed=11), synth.labels_to_image(**kw, id=1, seed=12)
model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2,
                              compute_dtype="fp32x3", device=dev, seed=0)
tr = training.SynthMorphTrainer(model, g1, g2, reg_param=1.0, optimizer=training.Adam(1e-4))
for _ in range(3):
    tr.train_step(src, src)
torch.cuda.synchronize()
mmr.ops.PROFILE = []
for _ in range(steps):
    tr.train_step(src, src)
torch.cuda.synchronize()
prof, mmr.ops.PROFILE = mmr.ops.PROFILE, None
per = len(prof) // steps
acc = collections.OrderedDict()
for i, (fam, tag, e0, e1, fl) in enumerate(prof):
    k = (i % per, fam, tag)
    a = acc.setdefault(k, [0.0, fl])
    a[0] += e0.elapsed_time(e1) / steps
tot = 0.0
print(f"{'#':>3s} {'family':44s} {'tag':28s} {'ms':>7s} {'TFLOP/s':>8s} {'of x3 peak':>10s}")
for (i, fam, tag), (ms, fl) in acc.items():
    tot += ms
    tf = fl / ms / 1e9 if fl else 0.0
    print(f"{i:3d} {fam:44s} {str(tag):28s} {ms:7.3f} {tf:8.1f} {tf / 833.3:10.3f}")
print(f"sum of timed launches {tot:.2f} ms/step")
