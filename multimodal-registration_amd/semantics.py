"""Switchable upstream semantics.

The arithmetic behind the reference lives in voxelmorph / neurite at pinned commits (README.md:35-37 of the
reference) that are not available here, so three points where upstream changed behaviour over time cannot be
verified (SURVEY.md Appendix A, items marked uncertain).  Each is an int flag on the C-ABI entry points
(include/mmr.h, MMR_RESIZE_* / MMR_DICE_* / MMR_NCC_*) and a keyword on the Python operators; this module holds
the process-wide default those keywords fall back to.  The defaults are what is recalled for late-2021 upstream:

  resize_grid  'align_corners'  ne.utils.resize samples linspace(0, old-1, new)              (A4, default)
               'arange_over_f'  older neurite: arange(new) / zoom_factor, clamped at the edge
  dice_eps     'divide_no_nan'  vxm.losses.Dice: tf.math.divide_no_nan(top, bottom)           (A6, default)
               'max_eps'        older voxelmorph: top / max(bottom, 1e-5)
  ncc_form     'classic'        vxm.losses.NCC: cc = cross^2 / (I_var * J_var + eps)          (A8, default)
               'clamped'        newer voxelmorph: cross, I_var, J_var >= eps; (cross/I_var)(cross/J_var)

A model trained upstream under the other variant is matched with ``mmr.semantics.set(...)`` (or the environment
variables MMR_RESIZE_GRID / MMR_DICE_EPS / MMR_NCC_FORM) without rebuilding anything.
"""
import contextlib
import os

CODES = {
    "resize_grid": {"align_corners": 0, "arange_over_f": 1},
    "dice_eps": {"divide_no_nan": 0, "max_eps": 1},
    "ncc_form": {"classic": 0, "clamped": 1},
}
_ENV = {"resize_grid": "MMR_RESIZE_GRID", "dice_eps": "MMR_DICE_EPS", "ncc_form": "MMR_NCC_FORM"}
_state = {}


def _check(key, value):
    if key not in CODES:
        raise KeyError(f"unknown semantics key {key!r}; known: {sorted(CODES)}")
    if value not in CODES[key]:
        raise ValueError(f"{key} must be one of {sorted(CODES[key])}, got {value!r}")
    return value


for _k, _e in _ENV.items():
    _state[_k] = _check(_k, os.environ.get(_e, next(iter(CODES[_k]))))


def get(key):
    if key not in CODES:
        raise KeyError(f"unknown semantics key {key!r}; known: {sorted(CODES)}")
    return _state[key]


def set(**kw):
    """mmr.semantics.set(resize_grid='arange_over_f', ncc_form='clamped', ...)"""
    for k, v in kw.items():
        _state[k] = _check(k, v)


def code(key, value=None):
    """C-ABI flag for ``value`` (None -> the current default of ``key``)."""
    return CODES[key][_check(key, _state[key] if value is None else value)]


@contextlib.contextmanager
def using(**kw):
    old = dict(_state)
    set(**kw)
    try:
        yield
    finally:
        _state.clear()
        _state.update(old)
