"""torch-CPU fp32 restatement of ONE WHOLE SynthMorph training step -- TEST INFRASTRUCTURE ONLY, PARITY UNPINNED like
the rest of oracle/: used by ``bench.py``'s ``cpu_baseline`` leg of the training workload (kind "port") and smoke-
tested on CPU in tests/test_oracle_kat.py.

What a step of ``train_synthmorph.py:284-344`` does on the reference's TF CPU path, restated with what this image has:
two ``labels_to_image`` generators (oracle/synth_np.py, NumPy, noise drawn with a NumPy generator -- drawing it is part
of the generator's work), U-Net forward + SVF tail + 26-channel one-hot warp + Dice + Grad-l2 and their backward through
torch autograd (oracle/grad_torch.py's graph in float32: oneDNN conv3d forward / backward, gather-form interpn), then
Adam(lr, eps 1e-7) over the 22 tensors.  BASELINE.md section 3 prescribes exactly this as the CPU figure."""
import math

import numpy as np
import torch

from . import grad_torch as G
from . import net_np, synth_np


class CpuStep:
    def __init__(self, label_map, L, enc, dec, reg_param=1.0, lr=1e-4, seed=0, warp_res=(16,), bias_res=(40,),
                 warp_std=3.0, blur_std=1.0, bias_std=0.3, gamma_std=0.25, int_steps=5):
        self.lab = np.ascontiguousarray(np.asarray(label_map, dtype=np.uint8))
        self.shape = tuple(self.lab.shape)
        self.L, self.enc, self.dec = int(L), list(enc), list(dec)
        self.reg_param, self.int_steps = float(reg_param), int_steps
        self.warp_res, self.bias_res = tuple(warp_res), tuple(bias_res)
        self.warp_std, self.blur_std, self.bias_std, self.gamma_std = warp_std, blur_std, bias_std, gamma_std
        self.ws = [torch.from_numpy(np.ascontiguousarray(w)).float().requires_grad_(True)
                   for w in net_np.init_weights(enc, dec, seed=seed)]
        self.opt = torch.optim.Adam(self.ws, lr=lr, betas=(0.9, 0.999), eps=1e-7)
        self.rng = np.random.default_rng(seed)
        self.last_loss = None

    def _draws(self):
        """One batch item's random draws in the layout oracle/synth_np.labels_to_image consumes (Appendix A9)."""
        r, L, S = self.rng, self.L, self.shape
        half = tuple(s // 2 for s in S)
        coarse = lambda shp, sc: tuple(int(math.ceil(s / sc)) for s in shp)
        d = {"vel_stds": [[float(r.uniform(0, self.warp_std)) for _ in self.warp_res]],
             "vel_noise": [[r.standard_normal(coarse(half, sc / 2) + (3,), dtype=np.float32) for sc in self.warp_res]],
             "means": r.uniform([0] + [25] * (L - 1), [225] * L, size=(1, L)).astype(np.float32),
             "stds": r.uniform([0] + [5] * (L - 1), [25] * L, size=(1, L)).astype(np.float32),
             "gmm_noise": [r.standard_normal(S, dtype=np.float32)],
             "sigma": r.uniform(0, self.blur_std, size=1).astype(np.float32),
             "bias_stds": [[float(r.uniform(0, self.bias_std)) for _ in self.bias_res]],
             "bias_noise": [[r.standard_normal(coarse(S, sc) + (1,), dtype=np.float32) for sc in self.bias_res]],
             "gamma": r.normal(0, self.gamma_std, size=1).astype(np.float32)}
        if r.uniform() < 0.2:   # zero_background
            d["means"][:, 0] = 0
            d["stds"][:, 0] = 0
        return d

    def step(self):
        lab = self.lab[None, ..., None]
        img1, _, oh1 = synth_np.labels_to_image(lab, self.L, self._draws(), self.warp_res, self.bias_res, self.blur_std)
        img2, _, oh2 = synth_np.labels_to_image(lab, self.L, self._draws(), self.warp_res, self.bias_res, self.blur_std)
        src, trg = torch.from_numpy(img1), torch.from_numpy(img2)
        o1, o2 = torch.from_numpy(oh1), torch.from_numpy(oh2)
        keep = G.DT
        G.DT = torch.float32
        try:
            total = G.synthmorph_loss(src, trg, o1, o2, self.ws, self.enc, self.dec, self.int_steps, self.reg_param)[0]
            self.opt.zero_grad(set_to_none=True)
            total.backward()
        finally:
            G.DT = keep
        self.opt.step()
        self.last_loss = float(total.detach())
        return self.last_loss
