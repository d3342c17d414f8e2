"""Mirrors of ``vxm.losses`` used by the reference (train_synthmorph.py:303-307)
plus the NCC / bending-energy terms of BASELINE.json config 5."""
from . import ops
from .layers import to_device


class Dice:
    """``vxm.losses.Dice().loss(y_true, y_pred)`` -> scalar (Appendix A6).  ``eps_mode`` picks the upstream
    variant ('divide_no_nan' | 'max_eps'); None = the process-wide setting of ``mmr.semantics``."""

    def __init__(self, eps_mode=None):
        self.eps_mode = eps_mode

    def loss(self, y_true, y_pred):
        return ops.dice_loss(to_device(y_true), to_device(y_pred), eps_mode=self.eps_mode)

    def grad(self, y_true, y_pred, scale=1.0):
        """d loss / d y_pred, [B,*S,L]."""
        t = to_device(y_true)
        _, tb = ops.dice_loss(t, to_device(y_pred), return_parts=True, eps_mode=self.eps_mode)
        return ops.dice_loss_bwd(t, tb, scale, eps_mode=self.eps_mode)


class Grad:
    """``vxm.losses.Grad('l2', loss_mult=).loss(None, flow)`` -> [B] (Appendix A7)."""

    def __init__(self, penalty="l1", loss_mult=None, vox_weight=None):
        if penalty != "l2":
            raise ValueError("only penalty='l2' is implemented (what the reference uses)")
        self.loss_mult = 1.0 if loss_mult is None else float(loss_mult)

    def loss(self, _, y_pred):
        return ops.grad_l2_loss(to_device(y_pred), self.loss_mult)

    def grad(self, _, y_pred, scale=1.0):
        """d sum_b loss[b] / d y_pred."""
        return ops.grad_l2_bwd(to_device(y_pred), self.loss_mult, scale)


class NCC:
    """``vxm.losses.NCC(win).loss(I, J)`` -> [B]; cc = cross^2/(I_var*J_var+eps) (Appendix A8).  ``form``
    picks the upstream variant ('classic' | 'clamped'); None = the process-wide setting of ``mmr.semantics``."""

    def __init__(self, win=None, eps=1e-5, form=None):
        self.win = 9 if win is None else int(win if not isinstance(win, (list, tuple)) else win[0])
        self.eps = eps
        self.form = form

    def loss(self, y_true, y_pred):
        return ops.ncc_loss(to_device(y_true), to_device(y_pred), self.win, self.eps, form=self.form)

    def grad(self, y_true, y_pred, gout=None):
        """d loss / d y_pred (the moved image), [B,*S,1]."""
        return ops.ncc_loss_bwd(to_device(y_true), to_device(y_pred), gout, self.win, self.eps, want=("J",),
                                form=self.form)[1]


class BendingEnergy:
    """Mean squared second differences of a displacement field -> [B] (defined by this build)."""

    def loss(self, _, y_pred):
        return ops.bending_energy(to_device(y_pred))

    def grad(self, _, y_pred, gout=None):
        return ops.bending_energy_bwd(to_device(y_pred), gout)


def dice_loss_zeropad(y_true, y_pred):
    """Intent of the reference's ``losses.dice_loss_zeropad`` (losses.py:13-21; the reference
    function itself always raises): Dice over labels 1..L-1 of batch item 0 with
    voxels masked where channel 0 >= 1 in either map.  One HIP reduction (mmr_dice_zeropad_fwd_f32)."""
    return ops.dice_loss(to_device(y_true), to_device(y_pred), zeropad=True)
