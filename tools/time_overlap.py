#!/usr/bin/env python
"""C3 training step A/B of one trainer option, alternated in one process, with a gradient comparison:
  python tools/time_overlap.py [steps] [option]     option: overlap_wgrad (default) | fuse_pool_bwd | batch_repack"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmr
from mmr import synth, training
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
opt = sys.argv[2] if len(sys.argv) > 2 else "overlap_wgrad"
shape, L, feats = (160, 160, 160), 26, 64
enc, dec = [feats] * 4, [feats] * 6
maps = synth.generate_label_maps(shape, L, 1, [16, 32, 64], [8, 16, 32], 1, 3, seed=100, device=dev)
kw = dict(in_shape=shape, in_label_list=np.arange(L), out_label_list=np.arange(L), warp_std=3, warp_res=16, blur_std=1,
          bias_std=0.3, bias_res=40, gamma_std=0.25, device=dev)
src = torch.from_numpy(maps[0][None, ..., None]).to(dev)
trs = {}
vals = (False, "small", True) if opt == "overlap_wgrad" else (False, True)
for ov in vals:
    g1, g2 = synth.labels_to_image(**kw, id=0, seed=11), synth.labels_to_image(**kw, id=1, seed=12)
    model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2,
                                  compute_dtype="fp32x3", device=dev, seed=0)
    trs[ov] = (training.SynthMorphTrainer(model, g1, g2, reg_param=1.0, optimizer=training.Adam(1e-4), **{opt: ov}), g1, g2)
# same draws -> same gradients
d = {}
for ov, (tr, g1, g2) in trs.items():
    g1b, g2b = synth.labels_to_image(**kw, id=0, seed=11), synth.labels_to_image(**kw, id=1, seed=12)
    tr.forward_backward(src, src, g1b.draw(1), g2b.draw(1))
    torch.cuda.synchronize()
    d[ov] = tr.gflat.clone()
print("gradients equal:", torch.equal(d[False], d[True]), float((d[False] - d[True]).abs().max() / d[False].abs().max()))
for rnd in range(3):
    for ov, (tr, _, _) in trs.items():
        for _ in range(2):
            tr.train_step(src, src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.train_step(src, src)
        torch.cuda.synchronize()
        print(f"round {rnd} {opt}={ov!s:5s}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step", flush=True)
