"""MI355X-native SynthMorph / VoxelMorph registration engine (hot path only).

Mirrors the operator surface the reference scripts call on voxelmorph/neurite
(SURVEY.md section 8b): ``networks.VxmDense``, ``networks.Transform``,
``layers.SpatialTransformer``, ``utils.transform/compose/rescale_dense_transform``,
``losses.Dice/Grad/NCC``.  All arithmetic runs in hand-written gfx950 HIP
kernels behind the C-ABI of include/mmr.h; there is no CPU fallback.
"""
from . import _lib, ops, semantics  # noqa: F401
from . import layers, losses, networks, utils  # noqa: F401
from . import data, evaluation, parallel, py_utils, registration, synth, tiling, training  # noqa: F401
from ._lib import MmrError  # noqa: F401

__all__ = ["ops", "semantics", "layers", "losses", "networks", "utils", "data", "evaluation", "parallel", "py_utils", "registration", "synth", "tiling",
           "training", "MmrError"]
