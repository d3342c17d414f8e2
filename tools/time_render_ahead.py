#!/usr/bin/env python
"""C3 training step with the image pair rendered inside the step vs one step ahead on the generator stream
(SynthMorphTrainer.train_step(next_labels=)), alternated in one process; also checks that both trainers see the same
images and end at the same weights (same random streams in the same order).
  python tools/time_render_ahead.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmr
from mmr import synth, training
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
shape, L, feats = (160, 160, 160), 26, 64
enc, dec = [feats] * 4, [feats] * 6
maps = synth.generate_label_maps(shape, L, 1, [16, 32, 64], [8, 16, 32], 1, 3, seed=100, device=dev)
kw = dict(in_shape=shape, in_label_list=np.arange(L), out_label_list=np.arange(L), warp_std=3, warp_res=16, blur_std=1,
          bias_std=0.3, bias_res=40, gamma_std=0.25, device=dev)
src = torch.from_numpy(maps[0][None, ..., None]).to(dev)
trs = {}
for ahead in (False, True):
    g1, g2 = synth.labels_to_image(**kw, id=0, seed=11), synth.labels_to_image(**kw, id=1, seed=12)
    model = mmr.networks.VxmDense(shape, nb_unet_features=(enc, dec), int_steps=5, int_resolution=2, svf_resolution=2,
                                  compute_dtype="fp32x3", device=dev, seed=0)
    trs[ahead] = training.SynthMorphTrainer(model, g1, g2, reg_param=1.0, optimizer=training.Adam(1e-4))
losses = {}
for ahead, tr in trs.items():
    losses[ahead] = [float(tr.train_step(src, src, next_labels=(src, src) if ahead and i < 3 else None)["loss"]) for i in range(4)]
torch.cuda.synchronize()
wa, wb = trs[False].model._flat, trs[True].model._flat
print("losses in step / ahead:", losses[False], losses[True])
print("weights after 4 steps: max |diff| / max |w| =", float((wa - wb).abs().max() / wa.abs().max()))
for rnd in range(3):
    for ahead, at in ((False, "tail"), (True, "end"), (True, "tail")):
        tr = trs[ahead]
        tr.render_at = at
        nl = (src, src) if ahead else None
        for _ in range(2):
            tr.train_step(src, src, next_labels=nl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.train_step(src, src, next_labels=nl)
        torch.cuda.synchronize()
        print(f"round {rnd} render_ahead={ahead!s:5s} at={at if ahead else '-':4s}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step", flush=True)
        if ahead:   # drop the pair rendered for a step that this loop does not run, so the next round starts clean
            tr._ahead = None
