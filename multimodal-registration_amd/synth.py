"""SynthMorph image synthesis on device: ``ne.models.labels_to_image``,
``ne.utils.augment.draw_perlin`` and the reference's ``generate_label_maps``.

Reference call sites: train_synthmorph.py:31-82 (label maps), :258-268,288-291
(generators).  Stage order and defaults follow SURVEY.md Appendix A9/A10.  TF's
Philox streams cannot be reproduced, so parity is defined as
"same output for the same injected draws": every random quantity is collected
in a ``draws`` dict that can be passed back in (and is what the CPU oracle in
oracle/synth_np.py consumes).  Small per-item parameters (stds, means, sigma,
gamma, flags) are drawn with a NumPy generator on the host; the big noise
fields are Philox4x32 draws on the device (mmr_philox_*, mmr_gmm_sample_f32).
"""
import math
import types

import numpy as np
import torch

from . import ops
from .layers import to_device


def _as_list(x, n=None):
    if np.isscalar(x):
        return [x] * (n or 1)
    return list(x)


def draw_perlin(out_shape, scales, min_std=0, max_std=1, modulate=True, rng=None, seed=None, draws=None,
                device="cuda"):
    """``ne.utils.augment.draw_perlin``: sum over scales of linearly upsampled N(0, std_s^2) noise drawn at
    ceil(shape/scale); std_s ~ U(min_std, max_std) (one scalar per scale).  out_shape = (*spatial, C) with 3
    spatial axes, or (*spatial, L, C) — the reference's 5-element call (train_synthmorph.py:61-64), where
    the label axis is a 4th interpolated axis: it is interpolated at the coarse resolution on
    the host (multilinear interpolation is separable), then the 3-D upsampling runs on device.

    Returns a device tensor of ``out_shape``.  ``draws`` (dict with 'stds', 'noise') injects / records the
    random draws."""
    out_shape = tuple(int(s) for s in out_shape)
    four_d = len(out_shape) == 5
    if len(out_shape) not in (4, 5):
        raise ValueError("draw_perlin needs out_shape = (X,Y,Z,C) or (X,Y,Z,L,C)")
    spatial = out_shape[:3]
    feat = out_shape[3:]
    scales = _as_list(scales)
    rng = np.random.default_rng(seed) if rng is None else rng
    rec = {"stds": [], "noise": []}
    C = int(np.prod(feat))
    out = None
    for i, scale in enumerate(scales):
        coarse_sp = tuple(int(math.ceil(s / scale)) for s in spatial)
        coarse = coarse_sp + ((int(math.ceil(feat[0] / scale)), feat[1]) if four_d else feat)
        std = float(draws["stds"][i]) if draws else float(rng.uniform(min_std, max_std) if modulate else max_std)
        if draws:
            g = draws["noise"][i]
            g = g.to(device) if isinstance(g, torch.Tensor) else to_device(np.asarray(g, dtype=np.float32), device=device)
            if tuple(g.shape) != tuple(coarse):
                raise ValueError(f"injected noise {tuple(g.shape)} != coarse shape {tuple(coarse)}")
        else:
            g = ops.philox_normal(coarse, seed=int(rng.integers(2 ** 62)), stream_id=i, device=device)
        rec["stds"].append(std)
        rec["noise"].append(g)
        if four_d:  # interpolate the label axis at coarse resolution (host, tiny), then treat (L,C) as channels
            gc = g.cpu().numpy()
            cl, L = gc.shape[3], feat[0]
            pos = np.arange(L, dtype=np.float32) * (np.float32(cl - 1) / np.float32(max(L - 1, 1)))
            l0 = np.clip(np.floor(pos), 0, cl - 1).astype(int)
            l1 = np.clip(l0 + 1, 0, cl - 1)
            w0 = (l1.astype(np.float32) - np.clip(pos, 0, cl - 1))[None, None, None, :, None]
            gc = w0 * gc[:, :, :, l0] + (1 - w0) * gc[:, :, :, l1]
            g = to_device(gc.astype(np.float32), device=device)
        g = g.reshape((1,) + coarse_sp + (C,)).contiguous()
        if out is None:  # first scale: the resize writes std * upsampled noise directly (no zero-fill, no axpy pass)
            out = ops.resize_trilinear(g, spatial, mul=std) if (scale != 1 or coarse_sp != spatial) else g * std
        elif scale == 1:
            ops.axpy_(out, g, std)
        else:
            ops.axpy_(out, ops.resize_trilinear(g, spatial), std)
    draw_perlin.last_draws = rec
    return out.reshape(out_shape)


def generate_label_maps(in_shape, num_labels, num_maps, im_scales, def_scales, im_max_std, def_max_std,
                        save_label=False, label_dir=None, add_str="", seed=None, device="cuda", shard=None,
                        draws=None):
    """Reference ``generate_label_maps`` (train_synthmorph.py:31-82): per map, Perlin image [*S,L] warped
    channel-wise by a Perlin field [*S,L,3], argmax over labels (first maximum, like ``tf.argmax``) -> uint8.
    ``shard=(rank, world)`` makes each rank synthesise maps rank, rank+world, ... (independent maps, no
    communication).  ``draws`` = one ``{"im": {...}, "warp": {...}}`` dict of
    ``draw_perlin`` draws per map injects the randomness (what oracle/synth_np.generate_label_maps consumes);
    the draws actually used are left in ``generate_label_maps.last_draws``.
    Saved as ``label_map_{add_str}{i}.nii.gz`` with an identity affine, like train_synthmorph.py:72-76."""
    rng = np.random.default_rng(seed)
    seeds = rng.integers(2 ** 62, size=num_maps)
    maps, rec, index = [], [], []
    rank, world = shard if shard else (0, 1)
    for i in range(num_maps):
        if i % world != rank:
            continue
        r = np.random.default_rng(int(seeds[i]))
        d = draws[i] if draws is not None else {}
        im = draw_perlin((*in_shape, num_labels), im_scales, max_std=im_max_std, rng=r, device=device,
                         draws=d.get("im"))
        d_im = draw_perlin.last_draws
        warp = draw_perlin((*in_shape, num_labels, len(in_shape)), def_scales, max_std=def_max_std, rng=r,
                           device=device, draws=d.get("warp"))
        rec.append({"im": d_im, "warp": draw_perlin.last_draws})
        moved = ops.warp3d(im[None].contiguous(), warp[None].contiguous(), "linear", None)
        maps.append(ops.argmax_u8(moved[0]).cpu().numpy())
        index.append(i)
        del im, warp, moved
    generate_label_maps.last_draws = rec
    if save_label and label_dir:
        import os
        from . import py_utils
        os.makedirs(label_dir, exist_ok=True)
        for i, m in zip(index, maps):
            py_utils.write_nifti(m, os.path.join(label_dir, f"label_map_{add_str}{i + 1}.nii.gz"), np.eye(4))
    return maps


def gaussian_kernels(sigmas, blur_std):
    """Per-item 1-D Gaussian kernels [B, W], W = 2*round(3*blur_std)+1, normalised to sum 1."""
    R = int(np.round(blur_std * 3))
    x = np.arange(-R, R + 1, dtype=np.float64)
    s = np.maximum(np.asarray(sigmas, dtype=np.float64), 1e-6)[:, None]
    k = np.exp(-0.5 * (x[None] / s) ** 2)
    return (k / k.sum(1, keepdims=True)).astype(np.float32)


class LabelsToImage:
    """``ne.models.labels_to_image``: label map [B,*S,1] uint8 -> (image [B,*S,1], one-hot [B,*S,L]) fp32."""

    def __init__(self, in_shape, in_label_list, out_label_list=None, out_shape=None, num_chan=1, mean_min=None,
                 mean_max=None, std_min=None, std_max=None, zero_background=0.2, warp_res=16, warp_std=0.5,
                 warp_modulate=True, bias_res=40, bias_std=0.3, bias_modulate=True, blur_std=1, blur_modulate=True,
                 normalize=True, gamma_std=0.25, dc_offset=0, one_hot=True, seeds=None, return_vel=False,
                 return_def=False, id=0, seed=None, device="cuda"):
        if num_chan != 1 or dc_offset != 0 or (out_shape is not None and tuple(out_shape) != tuple(in_shape)):
            raise NotImplementedError("num_chan=1, dc_offset=0, out_shape=in_shape only (what the reference uses)")
        self.in_shape = tuple(int(s) for s in in_shape)
        self.in_labels = np.unique(np.asarray(in_label_list)).astype(np.int64)
        out_list = self.in_labels if out_label_list is None else np.asarray(out_label_list).astype(np.int64)
        if not np.array_equal(np.unique(out_list), self.in_labels):
            raise NotImplementedError("out_label_list must equal in_label_list (train_synthmorph.py:233-234)")
        self.L = len(self.in_labels)
        if self.L > 256 or self.in_labels.max() > 255:
            raise ValueError("labels must fit uint8")
        lut = np.zeros(256, dtype=np.uint8)
        lut[self.in_labels] = np.arange(self.L, dtype=np.uint8)
        self.device = torch.device(device)
        self._lut = torch.from_numpy(lut).to(self.device)
        self._identity_lut = bool(np.array_equal(self.in_labels, np.arange(self.L)))
        L = self.L
        self.mean_min = np.asarray([0] + [25] * (L - 1), np.float32) if mean_min is None else np.asarray(mean_min, np.float32)
        self.mean_max = np.asarray([225] * L, np.float32) if mean_max is None else np.asarray(mean_max, np.float32)
        self.std_min = np.asarray([0] + [5] * (L - 1), np.float32) if std_min is None else np.asarray(std_min, np.float32)
        self.std_max = np.asarray([25] * L, np.float32) if std_max is None else np.asarray(std_max, np.float32)
        self.zero_background, self.warp_res, self.warp_std, self.warp_modulate = zero_background, _as_list(warp_res), warp_std, warp_modulate
        self.bias_res, self.bias_std, self.bias_modulate = _as_list(bias_res), bias_std, bias_modulate
        self.blur_std, self.blur_modulate, self.normalize, self.gamma_std = blur_std, blur_modulate, normalize, gamma_std
        self.one_hot, self.id = one_hot, id
        self.rng = np.random.default_rng(seed)
        self.inputs = [f"labels_input_{id}"]
        self.outputs = (f"image_{id}", f"labels_out_{id}")
        self.last_draws = None

    # -- host-side draws of the small per-item parameters ------------------------------------------
    def draw(self, B):
        r, L = self.rng, self.L
        d = {"seed": int(r.integers(2 ** 62))}
        if self.warp_std > 0:
            d["vel_stds"] = [[float(r.uniform(0, self.warp_std) if self.warp_modulate else self.warp_std)
                              for _ in self.warp_res] for _ in range(B)]
        d["means"] = r.uniform(self.mean_min, self.mean_max, size=(B, L)).astype(np.float32)
        d["stds"] = r.uniform(self.std_min, self.std_max, size=(B, L)).astype(np.float32)
        if self.zero_background > 0:
            keep = (r.uniform(size=B) >= self.zero_background).astype(np.float32)
            d["means"][:, 0] *= keep
            d["stds"][:, 0] *= keep
        if self.blur_std > 0:
            d["sigma"] = (r.uniform(0, self.blur_std, size=B) if self.blur_modulate else np.full(B, self.blur_std)).astype(np.float32)
        if self.bias_std > 0:
            d["bias_stds"] = [[float(r.uniform(0, self.bias_std) if self.bias_modulate else self.bias_std)
                               for _ in self.bias_res] for _ in range(B)]
        if self.gamma_std > 0:
            d["gamma"] = r.normal(0, self.gamma_std, size=B).astype(np.float32)
        return d

    def _perlin_batch(self, B, shape, scales, stds, noise, seed, tag):
        outs = []
        for b in range(B):
            dr = {"stds": stds[b], "noise": noise[b]} if noise is not None else None
            if dr is None:
                r = np.random.default_rng([int(seed), {"vel": 1, "bias": 2}[tag], b])
                dr = {"stds": stds[b], "noise": [
                    ops.philox_normal(tuple(int(math.ceil(s / sc)) for s in shape[:3]) + tuple(shape[3:]),
                                      seed=int(r.integers(2 ** 62)), stream_id=i, device=self.device)
                    for i, sc in enumerate(scales)]}
            outs.append(draw_perlin(shape, scales, draws=dr, device=self.device))
            self._rec.setdefault(f"{tag}_noise", []).append(dr["noise"])
        return torch.stack(outs)

    def generate(self, labels, draws=None, want_onehot=None):
        """labels: uint8 [B,*S,1] (NumPy or device). Returns dict(image, labels (uint8 indices), onehot|None)."""
        lab = to_device(labels, dtype=torch.uint8, device=self.device)
        if lab.dim() == 4:
            lab = lab[..., None].contiguous()
        B = lab.shape[0]
        if tuple(lab.shape[1:4]) != self.in_shape:
            raise ValueError(f"label map shape {tuple(lab.shape)} does not match in_shape {self.in_shape}")
        d = dict(draws) if draws is not None else self.draw(B)
        self._rec = dict(d)
        seed = d.get("seed", 0)
        if not self._identity_lut:
            lab = ops.lut_u8(lab, self._lut)
        if self.warp_std > 0:
            half = tuple(s // 2 for s in self.in_shape)
            vel = self._perlin_batch(B, half + (3,), [r / 2 for r in self.warp_res], d["vel_stds"], d.get("vel_noise"), seed, "vel")
            deff = ops.vecint(vel.contiguous(), 5)
            deff = ops.resize_trilinear(deff, self.in_shape, mul=2.0, pre_scale=True, zoom=2.0)
            lab = ops.warp3d_nearest_u8(lab, deff, fill_value=0)
        means = to_device(d["means"], device=self.device)
        stds = to_device(d["stds"], device=self.device)
        noise = d.get("gmm_noise")
        img = ops.gmm_sample(lab, means, stds, seed=seed, stream_id=7,
                             noise=to_device(noise, device=self.device) if noise is not None else None)
        if self.blur_std > 0:
            k = to_device(gaussian_kernels(d["sigma"], self.blur_std), device=self.device)
            img = ops.blur_separable(img, k)
        bias = None
        if self.bias_std > 0:
            bias = self._perlin_batch(B, self.in_shape + (1,), self.bias_res, d["bias_stds"], d.get("bias_noise"), seed, "bias").contiguous()
        gamma = to_device(d["gamma"], device=self.device) if (self.gamma_std > 0 and self.normalize) else None
        if self.normalize:
            ops.bias_clip_norm_gamma_(img, bias, gamma, 0.0, 255.0)
        elif bias is not None:
            raise NotImplementedError("normalize=False is not used by the reference")
        want = self.one_hot if want_onehot is None else want_onehot
        onehot = ops.onehot(lab, self.L) if want else None
        self.last_draws = self._rec
        return dict(image=img, labels=lab, onehot=onehot)

    def __call__(self, labels, draws=None):
        o = self.generate(labels, draws)
        return o["image"], (o["onehot"] if self.one_hot else o["labels"])

    def predict(self, labels, draws=None):
        img, m = self(labels, draws)
        return [img.cpu().numpy(), m.cpu().numpy()]


def labels_to_image(**kwargs):
    """Factory with neurite's name and keywords (train_synthmorph.py:288-289)."""
    return LabelsToImage(**kwargs)


models = types.SimpleNamespace(labels_to_image=labels_to_image)
