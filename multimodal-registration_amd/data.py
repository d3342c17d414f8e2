"""Host-side training data feed (NumPy): label-map pairs, flips, random zero borders.

Behavioural mirror of train_synthmorph.py:85-165; RNG call order is kept so that the
reference's outputs under a seeded generator are reproduced bit for bit (golden vectors in
tests/golden/host_helpers.npz).  Unlike the reference the ``void`` targets are built once and
never shipped to the GPU (the SynthMorph losses ignore them, train_synthmorph.py:136-137).
"""
import numpy as np


def set_random_zero_borders(im, scale=8):
    """Zero a random-width border (up to 1/scale per side) of channel 0 of im [*S, 1]."""
    ndim = im.ndim - 1
    box = []
    for d in range(ndim):
        n = im.shape[d]
        lo_cand = np.random.randint(0, n // scale)
        lo = np.random.choice([0, lo_cand])
        hi_cand = np.random.randint((scale - 1) * n // scale, n)
        hi = np.random.choice([hi_cand, n])
        box.append(slice(lo, hi))
    out = np.zeros_like(im)
    sel = tuple(box) + (0,)
    out[sel] = im[sel]
    return out


def gen_synthmorph_eb(label_maps, batch_size=1, same_subj=False, flip=True, random_zero_borders=True,
                      scale_zero_borders=8, frac_zero_bord=0.5, rng=None):
    """Endless generator of ([src, trg], [void, void]); src/trg uint8 [B,*S,1].

    ``rng``: optional np.random.Generator (the reference uses an unseeded default_rng)."""
    shape = label_maps[0].shape
    ndim = len(shape)
    void = np.zeros((batch_size, *shape, ndim), dtype="float32")
    rng = np.random.default_rng() if rng is None else rng
    while True:
        picks = rng.integers(len(label_maps), size=2 * batch_size)
        chosen = [label_maps[i] for i in picks]
        if same_subj:
            chosen = chosen[:batch_size] * 2
        x = np.stack(chosen)[..., None]
        if flip:
            axes = rng.choice(ndim, size=rng.integers(ndim + 1), replace=False, shuffle=False)
            x = np.flip(x, axis=axes + 1)
        src, trg = x[:batch_size], x[batch_size:]
        if random_zero_borders:
            for b in range(batch_size):
                if np.random.random() < frac_zero_bord:
                    trg[b] = set_random_zero_borders(trg[b], scale_zero_borders)
                if np.random.random() < frac_zero_bord:
                    src[b] = set_random_zero_borders(src[b], scale_zero_borders)
        yield [src, trg], [void] * 2
