"""torch-CPU (oneDNN) restatement of the VxmDense forward -- TEST INFRASTRUCTURE ONLY, used by ``bench.py``'s
``cpu_baseline`` leg (kind "port") and checked against oracle/net_np.py in tests/test_oracle_kat.py.

BASELINE.md section 3 prescribes this as the CPU figure placed beside the GPU number: the reference's TF CPU path
(bids_registration.py:460-472 selects all cores or one thread) cannot be timed here (TensorFlow absent), so the same
graph is run with torch's multi-threaded CPU kernels: Conv3D / MaxPooling3D /
UpSampling3D through torch.nn.functional in channels-last-3d memory format, the SVF tail (RescaleTransform, VecInt,
SpatialTransformer) through the gather formulation of ``interpn`` (Appendix A3).  PARITY UNPINNED like the rest of
the oracle."""
import itertools

import torch
import torch.nn.functional as F


def interpn(vol, loc):
    """vol [*S, C], loc [*O, 3] -> [*O, C]; linear, clamp-to-edge (A3)."""
    S = vol.shape[:3]
    idx, wts = [], []
    for d in range(3):
        mx = float(S[d] - 1)
        l = loc[..., d]
        clipped = torch.clamp(l, 0.0, mx)
        l0 = torch.clamp(torch.floor(l), 0.0, mx)
        l1 = torch.clamp(l0 + 1, 0.0, mx)
        w0 = l1 - clipped
        idx.append((l0.long(), l1.long()))
        wts.append((w0, 1 - w0))
    out = 0
    for c in itertools.product([0, 1], repeat=3):
        w = wts[0][c[0]] * wts[1][c[1]] * wts[2][c[2]]
        out = out + w[..., None] * vol[idx[0][c[0]], idx[1][c[1]], idx[2][c[2]]]
    return out


def _grid(shape, dtype):
    return torch.stack(torch.meshgrid(*[torch.arange(s, dtype=dtype) for s in shape], indexing="ij"), -1)


def transform(vol, shift):
    return interpn(vol, _grid(shift.shape[:3], vol.dtype) + shift)


def resize(vol, new_shape):
    S = vol.shape[:3]
    lin = [torch.arange(n, dtype=vol.dtype) * ((s - 1) / max(n - 1, 1)) for s, n in zip(S, new_shape)]
    return interpn(vol, torch.stack(torch.meshgrid(*lin, indexing="ij"), -1))


def vecint(v, nsteps):
    v = v / (2 ** nsteps)
    for _ in range(nsteps):
        v = v + transform(v, v)
    return v


def bf16_round(t):
    """Round-to-nearest-even to bf16 and back (the same values as oracle/net_np.bf16_round)."""
    return t.to(torch.bfloat16).to(t.dtype)


def prepare_weights(weights, dtype=torch.float32, quant=None):
    """Keras [3,3,3,Cin,Cout] kernels -> torch [Cout,Cin,3,3,3] in channels-last-3d (done once, outside the timing).
    ``quant`` (e.g. bf16_round) is applied to the kernels, not to the biases, as oracle/net_np.vxm_dense_forward does."""
    out = []
    for i in range(0, len(weights), 2):
        w = torch.as_tensor(weights[i], dtype=dtype)
        if quant is not None:
            w = quant(w)
        w = w.permute(4, 3, 0, 1, 2).contiguous(memory_format=torch.channels_last_3d)
        out.append((w, torch.as_tensor(weights[i + 1], dtype=dtype)))
    return out


@torch.no_grad()
def vxm_dense_forward(moving, fixed, tw, enc, dec, int_steps=5, int_resolution=2, svf_resolution=2, quant=None):
    """moving / fixed: torch [B,X,Y,Z,1]; tw = prepare_weights(...) -> dict(moved, preint_flow, pos_flow) (NDHWC).
    ``quant`` (e.g. bf16_round; pass the same to prepare_weights) rounds every conv's input and every LeakyReLU conv's
    output -- the rounding points of a reduced-precision HIP path, exactly those of oracle/net_np.vxm_dense_forward(quant=)."""
    if int_resolution != svf_resolution:
        raise NotImplementedError("int_resolution == svf_resolution (every configuration the reference ships)")
    nlev = len(enc)
    x = torch.cat([moving, fixed], -1).permute(0, 4, 1, 2, 3).contiguous(memory_format=torch.channels_last_3d)
    it = iter(tw)

    q = (lambda a: a) if quant is None else quant

    def conv(a, leaky=True):
        w, b = next(it)
        y = F.conv3d(q(a), w, b, padding=1)
        return q(F.leaky_relu_(y, 0.2)) if leaky else y
    skips = []
    last = x
    for _ in range(nlev):
        last = conv(last)
        skips.append(last)
        last = F.max_pool3d(last, 2)
    for _ in range(nlev):
        last = conv(last)
        last = torch.cat([F.interpolate(last, scale_factor=2, mode="nearest"), skips.pop()], 1)
    for _ in dec[nlev:]:
        last = conv(last)
    flow = conv(last, leaky=False).permute(0, 2, 3, 4, 1).contiguous()
    B = flow.shape[0]
    full = tuple(flow.shape[1:4])
    half = tuple(s // svf_resolution for s in full)
    pre, pos = [], []
    for b in range(B):
        svf = resize(flow[b], half) * (1.0 / svf_resolution) if svf_resolution != 1 else flow[b]
        p = vecint(svf, int_steps) if int_steps > 0 else svf
        if int_steps > 0 and int_resolution != 1:
            p = resize(p * float(int_resolution), full)
        pre.append(svf)
        pos.append(p)
    pos = torch.stack(pos)
    moved = torch.stack([transform(moving[b], pos[b]) for b in range(B)])
    return dict(moved=moved, preint_flow=torch.stack(pre), pos_flow=pos)
