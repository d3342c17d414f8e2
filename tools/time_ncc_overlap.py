#!/usr/bin/env python
"""C5 step (NCC win 9 + bending energy, 256^3 fp32): one stream vs two streams vs the same captured in a HIP graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr
dev = torch.device("cuda", 0)
torch.manual_seed(0)
I = torch.rand((1, 256, 256, 256, 1), device=dev)
J = torch.rand((1, 256, 256, 256, 1), device=dev)
flow = torch.randn((1, 256, 256, 256, 3), device=dev)
side = torch.cuda.Stream(device=dev)

def seq():
    return mmr.ops.ncc_loss(I, J, 9) + mmr.ops.bending_energy(flow)

def two():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        b = mmr.ops.bending_energy(flow)
    a = mmr.ops.ncc_loss(I, J, 9)
    cur.wait_stream(side)
    return a + b

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): r = fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3, (float(r) if r is not None else float('nan'))

print("one stream        : %.1f us  loss %.6f" % timeit(seq))
print("two streams       : %.1f us  loss %.6f" % timeit(two))
for name, fn in (("graph, one stream ", seq), ("graph, two streams", two)):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    us, _ = timeit(g.replay)
    print("%s: %.1f us  loss %.6f" % (name, us, float(out)))
