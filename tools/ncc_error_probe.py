"""GPU-box probe: actual relative errors of the NCC forward / backward kernels vs the float64 oracle on the shapes the
parity tests use (to set the test gates from measurements, not guesses)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmr
from oracle import grad_torch as G, ops_np as O

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
rng = np.random.default_rng(8)
for shape in [(20, 18, 40), (9, 33, 12), (5, 11, 256), (40, 9, 260), (14, 19, 37), (14, 19, 36), (12, 17, 60), (9, 9, 9), (20, 70, 13)]:
    I = rng.random((2,) + shape + (1,)).astype(np.float32)
    for name, J in (("uncorr", rng.random((2,) + shape + (1,)).astype(np.float32)), ("same", I), ("affine", (2 * I + 0.5).astype(np.float32)),
                    ("mix", (0.6 * I + 0.4 * rng.random((2,) + shape + (1,))).astype(np.float32))):
        for form in ("classic", "clamped"):
            ref = O.ncc_loss(I, J, 9, form=form)
            got = mmr.ops.ncc_loss(t(I), t(J), form=form).cpu().numpy()
            It, Jt = torch.from_numpy(I).double().requires_grad_(True), torch.from_numpy(J.copy()).double().requires_grad_(True)
            gout = np.array([1.0, -0.5], np.float32)
            (G.ncc_loss(It, Jt, form=form) * torch.from_numpy(gout).double()).sum().backward()
            dI, dJ = mmr.ops.ncc_loss_bwd(t(I), t(J), t(gout), form=form)
            print(shape, name, form, "loss", ref.round(4), "fwd rel %.2e" % np.abs((got - ref) / ref).max(),
                  "bwd rel-to-scale dI %.2e dJ %.2e" % (rel(dI.cpu().numpy(), It.grad.numpy()), rel(dJ.cpu().numpy(), Jt.grad.numpy())), flush=True)
