"""Gradient oracle (TEST INFRASTRUCTURE ONLY, PARITY UNPINNED): the forward graph of the SynthMorph
training step restated with differentiable torch-CPU float64 ops, so that torch.autograd supplies the
reference gradients TF's autodiff would produce for the same formulas (SURVEY.md Appendix A1-A7, A11;
call sites train_synthmorph.py:296-308).  ``interpn`` is the gather formulation of Appendix A3 (not
grid_sample) so the clamp-to-edge gradient semantics (tf.clip_by_value passes gradient on [min, max],
floor passes none) are the ones differentiated."""
import itertools

import torch
import torch.nn.functional as F

DT = torch.float64


def interpn(vol, loc, pin=None):
    """vol [*S, C], loc [*O, 3] -> [*O, C]; linear, clamp-to-edge.

    ``pin``: ANOTHER evaluation of ``loc`` (the HIP path's fp32 locations, as float64 values).  interpn is piecewise
    multilinear in ``loc`` with a kink at every integer (``floor``) and at the two clamp bounds; where the two evaluations
    fall on different sides of a kink the VALUE still agrees to rounding (the interpolant is continuous) but the gradient
    w.r.t. ``loc`` is the slope of another cell.  With ``pin`` the cell index and the inside / outside decision are taken
    from that other evaluation, so autograd differentiates the SAME piece (the analogue of ``unet(kinks=)`` for the tail)."""
    S = vol.shape[:3]
    idx, wts = [], []
    for d in range(3):
        mx = float(S[d] - 1)
        l = loc[..., d]
        if pin is not None:
            pl = pin[..., d]
            inside = (pl >= 0.0) & (pl <= mx)
            clipped = torch.where(inside, l, torch.clamp(l.detach(), 0.0, mx))
            l0 = torch.clamp(torch.floor(pl), 0.0, mx)
        else:
            clipped = torch.clamp(l, 0.0, mx)
            l0 = torch.clamp(torch.floor(l.detach()), 0.0, mx)
        l1 = torch.clamp(l0 + 1, 0.0, mx)
        w0 = l1 - clipped
        idx.append((l0.long(), l1.long()))
        wts.append((w0, 1 - w0))
    out = 0
    for c in itertools.product([0, 1], repeat=3):
        w = wts[0][c[0]] * wts[1][c[1]] * wts[2][c[2]]
        out = out + w[..., None] * vol[idx[0][c[0]], idx[1][c[1]], idx[2][c[2]]]
    return out


def grid(shape):
    return torch.stack(torch.meshgrid(*[torch.arange(s, dtype=DT) for s in shape], indexing="ij"), -1)


def transform(vol, shift, pin=None):
    return interpn(vol, grid(shift.shape[:3]) + shift, pin)


def resize(vol, new_shape, grid="align_corners", zoom=None):
    S = vol.shape[:3]
    if grid == "arange_over_f":
        lin = [torch.arange(n, dtype=DT) / (zoom if zoom else n / s) for s, n in zip(S, new_shape)]
    else:
        lin = [torch.arange(n, dtype=DT) * ((s - 1) / max(n - 1, 1)) for s, n in zip(S, new_shape)]
    return interpn(vol, torch.stack(torch.meshgrid(*lin, indexing="ij"), -1))


def vecint(v, nsteps, pins=None):
    """``pins``: one location tensor per squaring step (see ``interpn``)."""
    v = v / (2 ** nsteps)
    for k in range(nsteps):
        v = v + transform(v, v, None if pins is None else pins[k])
    return v


def dice_loss(t, p, eps_mode="divide_no_nan"):
    """[B,*S,L] -> scalar (-mean divide_no_nan, or top / max(bot, 1e-5) for eps_mode='max_eps')."""
    top = 2 * (t * p).sum((1, 2, 3))
    bot = (t + p).sum((1, 2, 3))
    if eps_mode == "max_eps":
        return -(top / bot.clamp(min=1e-5)).mean()
    return -torch.where(bot != 0, top / torch.where(bot != 0, bot, torch.ones_like(bot)), torch.zeros_like(bot)).mean()


def grad_l2(y, loss_mult):
    """[B,*S,C] -> [B]."""
    terms = []
    for d in (1, 2, 3):
        df = y.narrow(d, 1, y.shape[d] - 1) - y.narrow(d, 0, y.shape[d] - 1)
        terms.append((df * df).reshape(y.shape[0], -1).mean(1))
    return torch.stack(terms).mean(0) * loss_mult


# torch-CPU float64 conv3d is an im2col GEMM: its column buffer is 27 * Cin * 8 bytes PER OUTPUT VOXEL (26 GB for a 128-channel
# layer at 96 x 96 x 128).  Above this many output voxels per call the conv runs slab by slab along x (same arithmetic, autograd
# differentiates each slab on its own), so that the gradient oracle can be asked for volumes where the folded kernels engage.
SLAB_VOXELS = 1 << 17


def conv(x, w, b, leaky=True):
    """x [B,X,Y,Z,Cin], keras w [3,3,3,Cin,Cout]."""
    xc, wc = x.permute(0, 4, 1, 2, 3), w.permute(4, 3, 0, 1, 2)
    B, X, Y, Z, _ = x.shape
    if SLAB_VOXELS is None or x.dtype != torch.float64 or B * X * Y * Z <= SLAB_VOXELS:
        y = F.conv3d(xc, wc, b, padding=1)
    else:
        step = max(SLAB_VOXELS // (B * Y * Z), 1)
        xp = F.pad(xc, (0, 0, 0, 0, 1, 1))
        y = torch.cat([F.conv3d(xp[:, :, x0:min(x0 + step, X) + 2], wc, b, padding=(0, 1, 1)) for x0 in range(0, X, step)], 2)
    y = y.permute(0, 2, 3, 4, 1)
    return F.leaky_relu(y, 0.2) if leaky else y


def pool(x):
    return F.max_pool3d(x.permute(0, 4, 1, 2, 3), 2).permute(0, 2, 3, 4, 1)


def _windows(x):
    """[B,X,Y,Z,C] -> [B,X/2,Y/2,Z/2,C,8] (the 2x2x2 MaxPooling3D windows, floor on odd sizes)."""
    B, X, Y, Z, C = x.shape
    x = x[:, :X // 2 * 2, :Y // 2 * 2, :Z // 2 * 2]
    x = x.reshape(B, X // 2, 2, Y // 2, 2, Z // 2, 2, C).permute(0, 1, 3, 5, 7, 2, 4, 6)
    return x.reshape(B, X // 2, Y // 2, Z // 2, C, 8)


def pool_routed(x, k):
    """MaxPooling3D(2) of x with the window element chosen where ``k`` (another evaluation of the same
    activation) has its maximum -- the linear piece of max() that evaluation is on."""
    idx = _windows(k).argmax(-1, keepdim=True)
    return _windows(x).gather(-1, idx)[..., 0]


def up(x):
    return x.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)


def unet(src, trg, ws, enc, dec, kinks=None, collect=None):
    """``kinks``: optional list with one tensor per LeakyReLU conv layer (execution order) holding ANOTHER
    evaluation of that layer's activated output (the HIP forward's).  The network is piecewise linear; two fp32-grade
    evaluations that differ by 5e-6 sit on different linear pieces at the ~1e-5 of the activations that lie that
    close to a LeakyReLU kink or a max-pool tie, and their exact gradients then differ by O(1) there.  With
    ``kinks`` the LeakyReLU slope (1 where kinks > 0, else 0.2) and the max-pool routing are taken from that other
    evaluation, so autograd returns the exact float64 gradient of the SAME linear piece -- which is what a
    gradient-parity test at 1e-4 has to compare with.  Forward values change by at most 0.8 * 5e-6.
    ``collect``: a list that receives every LeakyReLU layer's activated output in execution order (this evaluation's own
    kink positions, for tests that compare them with another evaluation's)."""
    nlev = len(enc)
    it = iter(range(0, len(ws), 2))
    kit = iter(kinks) if kinks is not None else None
    last_k = [None]

    def c(x, leaky=True):
        i = next(it)
        if not leaky or kit is None:
            y = conv(x, ws[i], ws[i + 1], leaky)
            if leaky and collect is not None:
                collect.append(y.detach())
            return y
        k = next(kit)
        last_k[0] = k
        z = conv(x, ws[i], ws[i + 1], False)
        return z * torch.where(k > 0, torch.ones_like(k), torch.full_like(k, 0.2))
    last = torch.cat([src, trg], -1)
    skips = []
    for _ in range(nlev):
        last = c(last)
        skips.append(last)
        last = pool(last) if kit is None else pool_routed(last, last_k[0])
    for _ in range(nlev):
        last = c(last)
        last = torch.cat([up(last), skips.pop()], -1)
    for _ in dec[nlev:]:
        last = c(last)
    return c(last, leaky=False)


def tail_pins_from(svf, steps, pos, int_steps):
    """The fp32 sample locations of the HIP tail, rebuilt from its own tensors exactly as its kernels form them
    (csrc/tail.hip: ``(float)x + f`` in fp32): per batch item {"vecint": [int_steps location tensors at half resolution],
    "warp": the full-resolution one}.  svf [B,*h,3], steps [int_steps - 1, B,*h,3] (inputs of squaring steps 1..), pos [B,*S,3]:
    fp32 CPU tensors."""
    out = []
    gh = grid(svf.shape[1:4]).float()
    gf = grid(pos.shape[1:4]).float()
    for b in range(svf.shape[0]):
        v = [gh + svf[b].float() * (1.0 / (1 << int_steps))] + [gh + steps[k][b].float() for k in range(int_steps - 1)]
        out.append({"vecint": [t.double() for t in v], "warp": (gf + pos[b].float()).double()})
    return out


def synthmorph_loss(src, trg, onehot1, onehot2, ws, enc, dec, int_steps, reg_param, kinks=None, tail_pins=None):
    """Sum over the batch of (Dice + 1) + Grad-l2 (what Keras differentiates, Appendix A11).
    Returns (total, dice, grad[B], pos_flow, flow).  ``kinks``: see ``unet``; ``tail_pins``: ``tail_pins_from`` (see ``interpn``)."""
    flow = unet(src, trg, ws, enc, dec, kinks)
    B = flow.shape[0]
    half = tuple(s // 2 for s in flow.shape[1:4])
    pos = []
    for b in range(B):
        svf = 0.5 * resize(flow[b], half)
        v = vecint(svf, int_steps, None if tail_pins is None else tail_pins[b]["vecint"])
        pos.append(resize(2 * v, flow.shape[1:4]))
    pos = torch.stack(pos)
    pred = torch.stack([transform(onehot1[b], pos[b], None if tail_pins is None else tail_pins[b]["warp"]) for b in range(B)])
    dice = dice_loss(onehot2, pred)
    gl = grad_l2(pos, reg_param)
    total = (dice + 1) * B + gl.sum()
    return total, dice, gl, pos, flow


def ncc_loss(I, J, win=9, eps=1e-5, form="classic"):
    """A8 ``vxm.losses.NCC(win).loss`` on [B,*S,1] (float64 tensors): -mean(cross^2 / (Iv*Jv + eps)) per batch item,
    window sums by a ones-kernel conv with zero ('SAME') padding -- the same formula as oracle/ops_np.py::ncc_loss."""
    import torch.nn.functional as F
    Ii, Ji = I[..., 0][:, None], J[..., 0][:, None]
    k = torch.ones((1, 1, win, win, win), dtype=I.dtype)
    box = lambda t: F.conv3d(t, k, padding=win // 2)
    ws = float(win ** 3)
    Is, Js, I2, J2, IJ = box(Ii), box(Ji), box(Ii * Ii), box(Ji * Ji), box(Ii * Ji)
    uI, uJ = Is / ws, Js / ws
    cross = IJ - uJ * Is - uI * Js + uI * uJ * ws
    Iv = I2 - 2 * uI * Is + uI * uI * ws
    Jv = J2 - 2 * uJ * Js + uJ * uJ * ws
    if form == "clamped":  # newer upstream: tf.maximum(., eps) on all three, (cross / Iv) * (cross / Jv)
        cross, Iv, Jv = cross.clamp(min=eps), Iv.clamp(min=eps), Jv.clamp(min=eps)
        cc = (cross / Iv) * (cross / Jv)
    else:
        cc = cross * cross / (Iv * Jv + eps)
    return -cc.flatten(1).mean(1)


def bending_energy(u):
    """Same definition as oracle/ops_np.py::bending_energy on a float64 tensor [B,*S,3] -> [B]."""
    c = u[:, 1:-1, 1:-1, 1:-1]
    dxx = u[:, 2:, 1:-1, 1:-1] - 2 * c + u[:, :-2, 1:-1, 1:-1]
    dyy = u[:, 1:-1, 2:, 1:-1] - 2 * c + u[:, 1:-1, :-2, 1:-1]
    dzz = u[:, 1:-1, 1:-1, 2:] - 2 * c + u[:, 1:-1, 1:-1, :-2]
    dxy = (u[:, 2:, 2:, 1:-1] - u[:, 2:, :-2, 1:-1] - u[:, :-2, 2:, 1:-1] + u[:, :-2, :-2, 1:-1]) / 4
    dxz = (u[:, 2:, 1:-1, 2:] - u[:, 2:, 1:-1, :-2] - u[:, :-2, 1:-1, 2:] + u[:, :-2, 1:-1, :-2]) / 4
    dyz = (u[:, 1:-1, 2:, 2:] - u[:, 1:-1, 2:, :-2] - u[:, 1:-1, :-2, 2:] + u[:, 1:-1, :-2, :-2]) / 4
    e = dxx ** 2 + dyy ** 2 + dzz ** 2 + 2 * dxy ** 2 + 2 * dxz ** 2 + 2 * dyz ** 2
    return e.flatten(1).mean(1)
