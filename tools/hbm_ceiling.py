#!/usr/bin/env python
"""What a plain streaming kernel reaches on this box: write-only (fill), read-only (sum) and copy, on 2.5-GB tensors
(the size of the first layer's output / the flow head's input at BASELINE configs[1])."""
import torch
dev = torch.device("cuda", 0)
n = 160 * 160 * 192 * 256
x = torch.empty(n, dtype=torch.bfloat16, device=dev)
y = torch.empty(n, dtype=torch.bfloat16, device=dev)
def t(fn, k=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
gb = n * 2 / 1e9
ms = t(lambda: x.fill_(1.0)); print(f"fill  {gb:.2f} GB: {ms:.3f} ms = {gb / ms:.2f} TB/s")
ms = t(lambda: x.view(torch.int16).sum()); print(f"sum   {gb:.2f} GB: {ms:.3f} ms = {gb / ms:.2f} TB/s")
ms = t(lambda: y.copy_(x)); print(f"copy  {2 * gb:.2f} GB: {ms:.3f} ms = {2 * gb / ms:.2f} TB/s")
xf = x.view(torch.float32)
ms = t(lambda: xf.fill_(1.0)); print(f"fill f32 {gb:.2f} GB: {ms:.3f} ms = {gb / ms:.2f} TB/s")
ms = t(lambda: xf.sum()); print(f"sum f32  {gb:.2f} GB: {ms:.3f} ms = {gb / ms:.2f} TB/s")
