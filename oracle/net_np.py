"""VxmDense forward assembled from the oracle ops (TEST INFRASTRUCTURE ONLY).

Follows SURVEY.md section 3.4 / Appendix A1 (voxelmorph ``VxmDense`` as the
reference calls it at train_synthmorph.py:296 and 3d_reg.py:305-314).
PARITY UNPINNED (oracle/__init__.py).
"""
import numpy as np

from . import ops_np as O
from .cbind import conv3d_same


def bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (mirrors v_cvt_pk_bf16_f32)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.view(np.float32).reshape(x.shape)


def layer_plan(enc, dec):
    """[(name, cin, cout)] in Keras weight-creation order for ``get_weights()``."""
    plan = []
    cin = 2
    skips = []
    for i, nf in enumerate(enc):
        plan.append((f"enc_conv_{i}", cin, nf))
        skips.append(nf)
        cin = nf
    nlev = len(enc)
    for i in range(nlev):
        nf = dec[i]
        plan.append((f"dec_conv_{nlev - 1 - i}", cin, nf))
        cin = nf + skips.pop()
    for j, nf in enumerate(dec[nlev:]):
        plan.append((f"dec_final_{j}", cin, nf))
        cin = nf
    plan.append(("flow", cin, 3))
    return plan


def init_weights(enc, dec, seed=0, flow_std=1e-5):
    """He-normal kernels / zero bias; flow N(0, flow_std) (Appendix A1)."""
    rng = np.random.default_rng(seed)
    ws = []
    for name, cin, cout in layer_plan(enc, dec):
        std = flow_std if name == "flow" else np.sqrt(2.0 / (27 * cin))
        ws.append((rng.standard_normal((3, 3, 3, cin, cout)) * std).astype(np.float32))
        ws.append(np.zeros(cout, dtype=np.float32))
    return ws


def vxm_dense_forward(moving, fixed, weights, enc, dec, int_steps=5, int_resolution=2,
                      svf_resolution=2, quant=None):
    """moving/fixed [B,X,Y,Z,1] -> dict(moved, preint_flow, pos_flow, flow_full).

    ``quant`` (e.g. bf16_round) is applied to conv inputs/weights to mirror a
    reduced-precision HIP path's rounding points; None = pure fp32 in, double
    accumulate.
    """
    q = (lambda a: a) if quant is None else quant
    nlev = len(enc)
    x = np.concatenate([moving, fixed], -1).astype(np.float32)
    wi = iter(range(0, len(weights), 2))

    def conv(a, leaky=True):
        i = next(wi)
        return conv3d_same(q(a), q(weights[i]), weights[i + 1], leaky=leaky, alpha=0.2)

    skips = []
    last = x
    for _ in range(nlev):
        last = q(conv(last))
        skips.append(last)
        last = O.maxpool2(last)
    for _ in range(nlev):
        last = q(conv(last))
        last = np.concatenate([O.upsample2(last), skips.pop()], -1)
    for _ in dec[nlev:]:
        last = q(conv(last))
    flow = conv(last, leaky=False)

    B = flow.shape[0]
    svf = flow
    if svf_resolution != 1:
        svf = np.stack([O.rescale_dense_transform(flow[b], 1.0 / svf_resolution) for b in range(B)])
    preint = svf
    if int_resolution != svf_resolution:
        preint = np.stack([O.rescale_dense_transform(svf[b], svf_resolution / int_resolution) for b in range(B)])
    pos = preint
    if int_steps > 0:
        pos = np.stack([O.vecint(preint[b], int_steps) for b in range(B)])
        if int_resolution != 1:
            pos = np.stack([O.rescale_dense_transform(pos[b], int_resolution) for b in range(B)])
    moved = O.spatial_transformer(moving.astype(np.float32), pos, "linear", None)
    return dict(moved=moved, preint_flow=preint, pos_flow=pos, flow_full=flow)
