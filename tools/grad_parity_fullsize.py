#!/usr/bin/env python
"""Every gradient tensor of ONE SynthMorph training step at BASELINE configs[2]'s own size (160^3, enc/dec = 64, 26 labels,
fp32x3, every folded kernel and the pooling-backward epilogue engaged) against the float64 gradient oracle on the HIP forward's
linear piece -- the same check as tests/test_gpu_train.py::test_folded_training_step_gradients_vs_oracle (96 x 96 x 128 there),
too slow for the suite (several minutes of float64 CPU convolutions).   python tools/grad_parity_fullsize.py [X Y Z]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import torch
import test_gpu_train as T
shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (160, 160, 160)
t0 = time.time()
rep = []
T._check_step_gradients(torch.device("cuda", 0), "fp32x3", True, 1e-4, [64] * 4, [64] * 6, shape=shape, L=26, B=1, block=8, int_steps=5,
                        families=("_upfold", "_cinit", "_dgfold", "wgrad_mfma_f32x3_upfold"), report=rep)
for name, v in rep:
    print(f"{name:60s} {v if isinstance(v, list) else format(v, '.2e')}")
errs = [v for n, v in rep if n.endswith((' kernel', ' bias'))]
print(f"worst of {len(errs)} gradient tensors at {shape}: {max(errs):.2e} (rel. to each tensor's max; north_star's bar 1e-4); {time.time() - t0:.0f} s")
