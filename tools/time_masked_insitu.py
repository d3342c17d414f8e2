#!/usr/bin/env python
"""The masked data gradient at the arguments a real C3 training step hands it (real dz, real activations, the step's own
weights) against (a) the same kernel without mask / bias sums on the same dz, (b) the same on random dz, (c) the forward conv
of that layer on its real input: is the 2.6 ms of the masked dgrad (forward: 2.1 ms) the epilogue or the data?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmr
from mmr import synth, training, ops
dev = torch.device("cuda", 0)
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

def ev_time(fn, n=6):
    fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n

orig = ops.conv3d_k3_dgrad_masked
def probe(dz, wt, cin, ymask, dbias, alpha=0.2, accumulate=False, x3=False, pool_grad=None):
    if dz.shape[1] == 160 and pool_grad is None:
        db = torch.zeros_like(dbias)
        rnd = torch.randn_like(dz)
        small = rnd * float(dz.abs().mean())
        st = lambda t: f"mean|.| {float(t.abs().mean()):.3e} max {float(t.abs().max()):.3e} zeros {float((t == 0).float().mean()):.3f}"
        print("dz:", st(dz), "| ymask:", st(ymask), flush=True)
        rows = [("masked dgrad, real dz", lambda: orig(dz, wt, cin, ymask, db, alpha, False, x3)),
                ("plain conv, same weights, real dz", lambda: ops.conv3d_k3(dz, wt, None, cin, leaky=False, out_f32=True, x3=x3)),
                ("masked dgrad, N(0,1) dz", lambda: orig(rnd, wt, cin, ymask, db, alpha, False, x3)),
                ("plain conv, N(0,1) dz", lambda: ops.conv3d_k3(rnd, wt, None, cin, leaky=False, out_f32=True, x3=x3)),
                ("plain conv, N(0,1) scaled to dz's mean magnitude", lambda: ops.conv3d_k3(small, wt, None, cin, leaky=False, out_f32=True, x3=x3)),
                ("plain conv, real activations (ymask) as input", lambda: ops.conv3d_k3(ymask, wt, None, cin, leaky=False, out_f32=True, x3=x3))]
        for rnd_i in range(2):
            for name, fn in rows:
                print(f"  round {rnd_i} {name:52s} {ev_time(fn):.3f} ms", flush=True)
        # the same masked launch timed INSIDE a sequence, after another kernel (as in the step: the layer's wgrad precedes it)
        dw = torch.zeros((3, 3, 3, dz.shape[-1], cin), device=dz.device)   # wgrad of a cin -> Cz layer needs in0 = ymask-shaped tensor
        pre = {"masked dgrad itself": lambda: orig(dz, wt, cin, ymask, db, alpha, False, x3),
               "wgrad (ymask, dz)": lambda: ops.conv3d_k3_wgrad(ymask, dz, dw.view(3, 3, 3, cin, dz.shape[-1]) if cin == dz.shape[-1] else dw, x3=x3),
               "plain conv on other tensors": lambda: ops.conv3d_k3(rnd, wt, None, cin, leaky=False, out_f32=True, x3=x3),
               "torch fill of 1 GB": lambda: small.fill_(0.5)}
        for name, pf in pre.items():
            ts = []
            for _ in range(5):
                pf()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); orig(dz, wt, cin, ymask, db, alpha, False, x3); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print(f"  masked dgrad right after [{name}]: " + " ".join(f"{t:.3f}" for t in ts), flush=True)
    return orig(dz, wt, cin, ymask, dbias, alpha, accumulate, x3, pool_grad)
ops.conv3d_k3_dgrad_masked = probe
training.ops.conv3d_k3_dgrad_masked = probe
tr.train_step(src, src)
torch.cuda.synchronize()
