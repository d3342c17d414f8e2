"""End-to-end VxmDense.predict latency at C2 (NumPy float64 in -> NumPy fp32 out, as 3d_reg.py:310-314 calls it),
next to the device-resident forward that bench.py times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mmr
shape = (160, 160, 192)
m = mmr.networks.VxmDense(shape, nb_unet_features=([256] * 4, [256] * 6), int_steps=5, int_resolution=2, svf_resolution=2,
                          compute_dtype=sys.argv[1] if len(sys.argv) > 1 else "bf16")
rng = np.random.default_rng(0)
mov = rng.random((1,) + shape + (1,))  # float64, like nibabel's get_fdata()
fix = rng.random((1,) + shape + (1,))
for _ in range(2):
    m.predict([mov, fix])
t = time.perf_counter()
for _ in range(5):
    moved, flow = m.predict([mov, fix])
print("predict (float64 NumPy in, fp32 NumPy out): %.1f ms/pair" % ((time.perf_counter() - t) / 5 * 1e3))
a, b = torch.from_numpy(mov).float().cuda(), torch.from_numpy(fix).float().cuda()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    m.forward(a, b)
torch.cuda.synchronize()
print("device-resident forward: %.1f ms/pair" % ((time.perf_counter() - t) / 5 * 1e3))
# a batch of pairs: the copies of neighbouring pairs overlap the forward (VxmDense._predict_overlapped)
n = 6
movb, fixb = np.repeat(mov, n, 0), np.repeat(fix, n, 0)
m.predict([movb, fixb])
t = time.perf_counter()
m.predict([movb, fixb])
print("predict on a batch of %d pairs (copies overlapped): %.1f ms/pair" % (n, (time.perf_counter() - t) / n * 1e3))
