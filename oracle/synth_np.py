"""NumPy restatement of the SynthMorph generator stages (TEST INFRASTRUCTURE ONLY,
PARITY UNPINNED -- SURVEY.md Appendix A9/A10, call sites train_synthmorph.py:57-69,258-291).
All randomness is injected through the same ``draws`` dict the product records."""
import math

import numpy as np

from . import ops_np as O


def perlin(out_shape, scales, stds, noise):
    """draw_perlin with injected unit-normal coarse noise; out_shape (X,Y,Z,C) or (X,Y,Z,L,C)."""
    out_shape = tuple(out_shape)
    four_d = len(out_shape) == 5
    spatial, feat = out_shape[:3], out_shape[3:]
    C = int(np.prod(feat))
    out = np.zeros(spatial + (C,), np.float32)
    for scale, std, g in zip(scales, stds, noise):
        g = np.asarray(g, np.float32)
        if four_d:
            cl, L = g.shape[3], feat[0]
            pos = np.arange(L, dtype=np.float32) * (np.float32(cl - 1) / np.float32(max(L - 1, 1)))
            gl = O.interpn(np.moveaxis(g, 3, 0).reshape(cl, -1), pos[:, None], "linear")  # 1-D interp along labels
            g = np.moveaxis(gl.reshape((L,) + g.shape[:3] + g.shape[4:]), 0, 3)
        g = g.reshape(g.shape[:3] + (C,))
        if scale != 1:
            lin = [(np.arange(n, dtype=np.float32) * (np.float32(s - 1) / np.float32(max(n - 1, 1)))).astype(np.float32)
                   for s, n in zip(g.shape[:3], spatial)]
            g = O.interpn(g, np.stack(np.meshgrid(*lin, indexing="ij"), -1), "linear")
        out = (out + np.float32(std) * g).astype(np.float32)
    return out.reshape(out_shape)


def generate_label_maps(in_shape, num_labels, draws, im_scales, def_scales):
    """train_synthmorph.py:55-69 with injected draws: per map ``im = perlin((*S, L))``, ``warp = perlin((*S, L, 3))``,
    ``im = vxm.utils.transform(im, warp)`` (channel-wise linear warp, A2), ``lab = tf.argmax(im, -1)`` (index of the
    FIRST maximum), cast to uint8.  Also returns the warped images so a test can tell near-ties from errors."""
    maps, ims = [], []
    for d in draws:
        im = perlin((*in_shape, num_labels), im_scales, d["im"]["stds"], d["im"]["noise"])
        warp = perlin((*in_shape, num_labels, len(in_shape)), def_scales, d["warp"]["stds"], d["warp"]["noise"])
        im = O.transform(im, warp, "linear", None)
        ims.append(im)
        maps.append(np.argmax(im, axis=-1).astype(np.uint8))  # np.argmax: first occurrence, as tf.argmax
    return maps, ims


def gaussian_kernel(sigma, blur_std):
    R = int(np.round(blur_std * 3))
    x = np.arange(-R, R + 1, dtype=np.float64)
    k = np.exp(-0.5 * (x / max(float(sigma), 1e-6)) ** 2)
    return (k / k.sum()).astype(np.float32)


def blur(img, k):
    out = img.astype(np.float32)
    R = len(k) // 2
    for ax in range(3):
        pad = [(0, 0)] * 3
        pad[ax] = (R, R)
        p = np.pad(out, pad)
        acc = np.zeros_like(out)
        for j in range(len(k)):
            sl = [slice(None)] * 3
            sl[ax] = slice(j, j + out.shape[ax])
            acc = (acc + k[j] * p[tuple(sl)]).astype(np.float32)
        out = acc
    return out


def labels_to_image(labels, L, draws, warp_res=(16,), bias_res=(40,), blur_std=1.0, warp=True):
    """labels uint8 [B,X,Y,Z,1] (already 0..L-1) -> (image [B,X,Y,Z,1], labels_out uint8 [B,X,Y,Z,1], onehot)."""
    lab = np.asarray(labels)[..., 0]
    B, shape = lab.shape[0], lab.shape[1:]
    imgs, labs = [], []
    for b in range(B):
        l = lab[b]
        if warp:
            half = tuple(s // 2 for s in shape)
            vel = perlin(half + (3,), [r / 2 for r in warp_res], draws["vel_stds"][b], draws["vel_noise"][b])
            deff = O.vecint(vel, 5)
            deff = O.resize((deff * np.float32(2)).astype(np.float32), 2)
            l = O.transform(l.astype(np.float32)[..., None], deff, "nearest", 0.0)[..., 0].astype(np.uint8)
        img = (draws["means"][b][l] + draws["stds"][b][l] * np.asarray(draws["gmm_noise"][b]).reshape(shape)).astype(np.float32)
        if "sigma" in draws:
            img = blur(img, gaussian_kernel(draws["sigma"][b], blur_std))
        if "bias_stds" in draws:
            bias = perlin(shape + (1,), bias_res, draws["bias_stds"][b], draws["bias_noise"][b])[..., 0]
            img = (img * np.exp(bias)).astype(np.float32)
        img = np.clip(img, 0, 255)
        mn, mx = img.min(), img.max()
        img = (img - mn) / (mx - mn) if mx > mn else np.zeros_like(img)
        if "gamma" in draws:
            img = np.power(img.astype(np.float64), np.exp(np.float64(draws["gamma"][b]))).astype(np.float32)
        imgs.append(img)
        labs.append(l)
    labs = np.stack(labs)
    return np.stack(imgs)[..., None], labs[..., None], np.eye(L, dtype=np.float32)[labs]
