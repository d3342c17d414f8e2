"""ctypes binding of the C-ABI declared in include/mmr.h.

There is no CPU fallback: loading fails loudly when libmmr_hip.so is missing,
and every entry point raises ``MmrError`` on a non-zero return code.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_uint8, c_uint32, c_uint64, c_void_p

from . import build as _build

_LIB = None
ABI_VERSION = 100   # mmr_version() of csrc/api.hip

P = c_void_p
I = c_int
F = c_float

PACK_FWD, PACK_DGRAD, PACK_UPFOLD, PACK_DGFOLD = 0, 1, 2, 3


class PackJob(ctypes.Structure):
    """MmrPackJob of include/mmr.h."""
    _fields_ = [("w", c_void_p), ("out", c_void_p), ("kind", ctypes.c_int32), ("rows_total", ctypes.c_int32),
                ("row_off", ctypes.c_int32), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("reserved", ctypes.c_int32)]


# name -> (restype, argtypes); mirrors include/mmr.h one to one
SIGNATURES = {
    "mmr_version": (I, []),
    "mmr_error_string": (ctypes.c_char_p, [I]),
    "mmr_last_hip_error": (ctypes.c_char_p, []),
    "mmr_warp3d_f32": (I, [P, P, P, I, I, I, I, I, I, I, F, I, P]),
    "mmr_warp3d_nearest_u8": (I, [P, P, P, I, I, I, I, I, I, c_uint8, P]),
    "mmr_resize_trilinear_f32": (I, [P, P, I, I, I, I, I, I, I, I, F, I, I, F, P]),
    "mmr_compose_f32": (I, [P, P, P, I, I, I, I, P]),
    "mmr_vecint_f32": (I, [P, P, P, I, I, I, I, I, P]),
    "mmr_conv3d_k3_packed_bytes": (c_int64, [I, I, I]),
    "mmr_conv3d_k3_pack": (I, [P, P, I, I, I, I, P]),
    "mmr_conv3d_k3_fwd": (I, [P, I, I, P, I, P, P, P, P, I, I, I, I, I, I, F, I, I, P]),
    "mmr_conv3d_k3_ksplit_ws_bytes": (c_int64, [I, I, I, I, I, I, I]),
    "mmr_conv3d_k3_fwd_ws": (I, [P, I, I, P, I, P, P, P, P, I, I, I, I, I, I, F, I, I, P, P]),
    "mmr_conv3d_k3_upfold_packed_bytes": (c_int64, [I, I, I]),
    "mmr_conv3d_k3_upfold_pack": (I, [P, P, I, I, I, P]),
    "mmr_conv3d_k3_upfold_fwd": (I, [P, I, P, P, I, I, I, I, I, I, I, P]),
    "mmr_conv3d_k3_fwd_init": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, F, I, I, P, P]),
    "mmr_conv3d_k3_dgrad_upfold_packed_bytes": (c_int64, [I, I, I]),
    "mmr_conv3d_k3_dgrad_upfold_pack": (I, [P, P, I, I, I, P]),
    "mmr_conv3d_k3_pack_job_bytes": (c_int64, [I, I, I, I]),
    "mmr_conv3d_k3_pack_batch": (I, [P, I, I, P]),
    "mmr_conv3d_k3_dgrad_upfold_ws_bytes": (c_int64, [I, I, I, I, I]),
    "mmr_conv3d_k3_dgrad_upfold": (I, [P, I, P, P, I, I, I, I, I, P, F, P, P, I, I, P]),
    "mmr_conv3d_k3_cin2_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, F, I, P]),
    "mmr_conv3d_k3_cout3_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "mmr_maxpool3d2_fwd": (I, [P, P, I, I, I, I, I, I, P]),
    "mmr_dice_ws_bytes": (c_int64, [I, c_int64, I]),
    "mmr_dice_fwd_f32": (I, [P, P, P, P, P, I, c_int64, I, I, P]),
    "mmr_dice_zeropad_fwd_f32": (I, [P, P, P, P, P, I, c_int64, I, I, P]),
    "mmr_grad_l2_ws_bytes": (c_int64, [I, I, I, I, I]),
    "mmr_grad_l2_fwd_f32": (I, [P, P, P, I, I, I, I, I, F, P]),
    "mmr_ncc_ws_bytes": (c_int64, [I, I, I, I]),
    "mmr_ncc_fwd_f32": (I, [P, P, P, P, I, I, I, I, I, F, I, P]),
    "mmr_ncc_fwd_ticket_f32": (I, [P, P, P, P, P, I, I, I, I, I, F, I, F, I, P]),
    "mmr_bending_fwd_ticket_f32": (I, [P, P, P, P, I, I, I, I, F, I, P]),
    "mmr_bending_ws_bytes": (c_int64, [I, I, I, I]),
    "mmr_bending_fwd_f32": (I, [P, P, P, I, I, I, I, P]),
    "mmr_philox_normal_f32": (I, [P, c_int64, c_uint64, c_uint32, F, F, P]),
    "mmr_philox_uniform_f32": (I, [P, c_int64, c_uint64, c_uint32, F, F, P]),
    "mmr_lut_u8": (I, [P, P, P, c_int64, P]),
    "mmr_gmm_sample_f32": (I, [P, P, P, P, P, I, c_int64, I, c_uint64, c_uint32, P]),
    "mmr_blur_axis_f32": (I, [P, P, P, I, I, I, I, I, I, P]),
    "mmr_intensity_ws_bytes": (c_int64, [I]),
    "mmr_bias_clip_norm_gamma_f32": (I, [P, P, P, P, I, c_int64, F, F, P]),
    "mmr_onehot_f32": (I, [P, P, c_int64, I, P]),
    "mmr_argmax_u8": (I, [P, P, c_int64, I, P]),
    "mmr_axpy_f32": (I, [P, P, F, c_int64, P]),
    "mmr_dice_labels_ws_bytes": (c_int64, [I, c_int64, I]),
    "mmr_dice_labels_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "mmr_dice_labels_bwd": (I, [P, P, P, P, P, I, I, I, I, I, F, I, I, P]),
    "mmr_dice_labels_zeropad_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "mmr_dice_labels_zeropad_bwd": (I, [P, P, P, P, P, I, I, I, I, I, F, I, I, P]),
    "mmr_grad_l2_bwd_f32": (I, [P, P, I, I, I, I, I, F, F, I, P]),
    "mmr_resize_trilinear_bwd_f32": (I, [P, P, I, I, I, I, I, I, I, I, F, I, F, P]),
    "mmr_resize_trilinear_bwd_ws_bytes": (c_int64, [I, I, I, I, I, I, I, I]),
    "mmr_resize_trilinear_bwd_ws_f32": (I, [P, P, P, I, I, I, I, I, I, I, I, F, I, F, P]),
    "mmr_compose_bwd_f32": (I, [P, P, P, P, P, I, I, I, I, P]),
    "mmr_vecint_save_f32": (I, [P, P, P, I, I, I, I, I, P]),
    "mmr_vecint_bwd_f32": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "mmr_warp3d_bwd_flow_f32": (I, [P, P, P, P, I, I, I, I, I, P]),
    "mmr_warp3d_bwd_vol_f32": (I, [P, P, P, I, I, I, I, I, P]),
    "mmr_leaky_bwd_ws_bytes": (c_int64, [c_int64, I]),
    "mmr_leaky_bwd_bias_f32": (I, [P, P, P, P, P, c_int64, I, I, F, I, P]),
    "mmr_conv3d_k3_dgrad_masked_ws_bytes": (c_int64, [I, I, I, I, I]),
    "mmr_conv3d_k3_dgrad_masked": (I, [P, I, P, P, I, I, I, I, I, P, F, P, P, I, I, P]),
    "mmr_conv3d_k3_dgrad_masked_pool": (I, [P, I, P, P, I, I, I, I, I, P, F, P, P, I, I, P, P]),
    "mmr_conv3d_k3_cout3_dgrad_masked_ws_bytes": (c_int64, [I, I, I, I, I]),
    "mmr_conv3d_k3_cout3_dgrad_masked_f32": (I, [P, P, P, I, I, I, I, I, P, F, P, P, I, P]),
    "mmr_conv3d_k3_cout3_dgrad_masked_f32x3": (I, [P, P, P, I, I, I, I, I, P, F, P, P, I, P]),
    "mmr_dice_bwd_f32": (I, [P, P, P, I, c_int64, I, F, I, I, P]),
    "mmr_ncc_bwd_ws_bytes": (c_int64, [I, I, I, I]),
    "mmr_ncc_bwd_f32": (I, [P, P, P, P, P, P, I, I, I, I, I, F, I, P]),
    "mmr_bending_bwd_f32": (I, [P, P, P, I, I, I, I, I, P]),
    "mmr_conv3d_k3_dgrad_split_ws_bytes": (c_int64, [I, I, I, I, I]),
    "mmr_conv3d_k3_dgrad_split": (I, [P, I, P, P, P, I, I, I, I, I, I, P, F, P, P, I, I, P]),
    "mmr_upcat_bwd_masked_ws_bytes": (c_int64, [I, I]),
    "mmr_upcat_bwd_masked_f32": (I, [P, P, P, I, I, I, I, I, I, I, I, P, P, F, P, I, P, I, P, P]),
    "mmr_maxpool3d2_bwd_masked_ws_bytes": (c_int64, [I]),
    "mmr_maxpool3d2_bwd_masked_f32": (I, [P, P, P, I, I, I, I, I, I, I, F, P, I, P, P]),
    "mmr_upcat_bwd_f32": (I, [P, P, P, I, I, I, I, I, I, I, I, P]),
    "mmr_maxpool3d2_bwd_f32": (I, [P, P, P, I, I, I, I, I, I, P]),
    "mmr_conv3d_k3_wgrad_ws_bytes": (c_int64, [I, I, I, I, I, I]),
    "mmr_conv3d_k3_wgrad_f32": (I, [P, I, I, P, I, P, P, P, I, I, I, I, I, I, P]),
    "mmr_conv3d_k3_wgrad_f32x3": (I, [P, I, I, P, I, P, P, P, I, I, I, I, I, I, P]),
    "mmr_conv3d_k3_wgrad_f32x1": (I, [P, I, I, P, I, P, P, P, I, I, I, I, I, I, P]),
    "mmr_conv3d_k3_wgrad_upfold_ws_bytes": (c_int64, [I, I, I, I, I, I, I]),
    "mmr_conv3d_k3_wgrad_upfold": (I, [P, I, P, I, P, P, P, I, I, I, I, I, I, I, P]),
    "mmr_conv3d_k3_cin2_wgrad_ws_bytes": (c_int64, [I]),
    "mmr_conv3d_k3_cin2_wgrad_f32": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "mmr_conv3d_k3_cin2_wgrad_f32x3": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "mmr_conv3d_k3_cout3_dgrad_f32": (I, [P, P, P, I, I, I, I, I, P]),
    "mmr_conv3d_k3_cout3_dgrad_f32x3": (I, [P, P, P, I, I, I, I, I, P]),
    "mmr_adam_step_f32": (I, [P, P, P, P, c_int64, F, F, F, F, c_int64, F, P]),
    "mmr_jacobian_det_f64": (I, [P, P, I, I, I, P]),
    "mmr_joint_hist_f64": (I, [P, P, P, P, P, c_int64, I, P]),
    "mmr_overlap_ws_bytes": (c_int64, []),
    "mmr_overlap_sums_f64": (I, [P, P, P, P, c_int64, P]),
    "mmr_host_alloc": (I, [P, c_int64, I]),
    "mmr_host_free": (I, [P]),
    "mmr_host_register": (I, [P, c_int64, P]),
    "mmr_host_unregister": (I, [P]),
    "mmr_cast_to_f32": (I, [P, P, c_int64, I, P]),
    "mmr_copy_to_host": (I, [P, P, c_int64, P]),
    "mmr_memcpy_async": (I, [P, P, c_int64, I, P]),
}


class MmrError(RuntimeError):
    pass


def lib_path():
    """MMR_LIB=<path> loads a specific prebuilt library instead (development A/B of kernel builds in one gpurun call);
    the staleness check then does not apply."""
    return os.environ.get("MMR_LIB") or _build.LIB


def load():
    """Load libmmr_hip.so.  A missing library, or one whose recorded source hash no longer matches csrc/ while
    hipcc is available, is (re)built first -- serialised across the ranks of a node by build.build()'s file lock.
    A stale library without hipcc is refused rather than silently used."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    try:
        stale = not os.environ.get("MMR_LIB") and _build.needs_build()
    except OSError as e:
        raise MmrError(f"cannot check {path} against its sources: {e}") from e
    if stale:
        have_hipcc = os.path.exists(_build.hipcc_path())
        if not os.path.exists(path) or have_hipcc:
            try:
                _build.build()
            except Exception as e:  # no silent fallback
                raise MmrError(f"libmmr_hip.so is missing or stale and could not be built: {e}") from e
        else:
            raise MmrError(f"{path} does not match the sources in csrc/ and hipcc is not available to rebuild it")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MmrError(f"{path} does not export {name}; rebuild with __graft_entry__.build()") from e
        fn.restype = res
        fn.argtypes = args
    if os.environ.get("MMR_LIB") and lib.mmr_version() != ABI_VERSION:   # a hand-picked library skips the source-hash check
        raise MmrError(f"{path} reports ABI version {lib.mmr_version()}, this binding expects {ABI_VERSION}")
    _LIB = lib
    return lib


def check(rc, what):
    if rc != 0:
        lib = load()
        msg = lib.mmr_error_string(int(rc)).decode()
        hip = lib.mmr_last_hip_error().decode()
        raise MmrError(f"{what} failed: {msg}" + (f" [{hip}]" if rc == -2 and hip else ""))
