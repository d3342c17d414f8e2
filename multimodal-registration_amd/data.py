"""Training data feed: label-map pairs, flips, random zero borders -- NumPy on the host like the reference, or with the
label maps kept resident in HBM (``device=``) so that no volume crosses PCIe per step.

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


def _zero_border_box(shape, scale):
    """The random box of set_random_zero_borders (same np.random call order), as slices."""
    box = []
    for n in shape:
        lo_cand = np.random.randint(0, n // scale)
        lo = np.random.choice([0, lo_cand])
        hi_cand = np.random.randint((scale - 1) * n // scale, n)
        hi = np.random.choice([hi_cand, n])
        box.append(slice(int(lo), int(hi)))
    return box


def _gen_device(label_maps, batch_size, same_subj, flip, random_zero_borders, scale_zero_borders, frac_zero_bord, rng, device):
    """gen_synthmorph_eb with the label maps resident on ``device``: per step only indices / flip axes / border boxes
    are drawn on the host (identical RNG call order -> identical batches), the volumes are gathered, flipped and
    masked by device kernels.  At 160^3 a host batch costs ~10 ms of NumPy plus two 4 MB H2D copies per step, which a
    33 ms GPU step does not hide once the copies synchronise with the stream; here the host work is microseconds."""
    import torch
    bank = torch.stack([torch.as_tensor(np.ascontiguousarray(m), dtype=torch.uint8) for m in label_maps]).to(device)
    shape = tuple(bank.shape[1:])
    ndim = len(shape)
    void = np.zeros((batch_size, *shape, ndim), dtype="float32")
    while True:
        picks = rng.integers(len(label_maps), size=2 * batch_size)
        if same_subj:
            picks = np.concatenate([picks[:batch_size]] * 2)
        x = bank[torch.as_tensor(picks, device=device)][..., None]          # [2B,*S,1], a fresh copy
        if flip:
            axes = rng.choice(ndim, size=rng.integers(ndim + 1), replace=False, shuffle=False)
            if len(axes):
                x = torch.flip(x, dims=[int(a) + 1 for a in axes])
        src, trg = x[:batch_size], x[batch_size:]
        if random_zero_borders:
            for b in range(batch_size):
                for t in (trg, src):  # same order of np.random draws as the host version
                    if np.random.random() < frac_zero_bord:
                        box = tuple(_zero_border_box(shape, scale_zero_borders))
                        keep = torch.zeros(shape, dtype=torch.bool, device=device)
                        keep[box] = True
                        t[b, ..., 0] *= keep
        yield [src.contiguous(), trg.contiguous()], [void] * 2


def gen_synthmorph_eb(label_maps, batch_size=1, same_subj=False, flip=True, random_zero_borders=True,
                      scale_zero_borders=8, frac_zero_bord=0.5, rng=None, device=None):
    """Endless generator of ([src, trg], [void, void]); src/trg uint8 [B,*S,1].

    ``rng``: optional np.random.Generator (the reference uses an unseeded default_rng).
    ``device``: keep the label maps on that device and yield device tensors (same batches, no per-step H2D)."""
    if device is not None:
        yield from _gen_device(label_maps, batch_size, same_subj, flip, random_zero_borders, scale_zero_borders,
                               frac_zero_bord, np.random.default_rng() if rng is None else rng, device)
        return
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
