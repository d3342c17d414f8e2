import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
import mmr
dev = torch.device("cuda:0")
x = torch.randn((1, 160, 160, 192, 256), device=dev) * 0.5
w = torch.randn((3, 3, 3, 256, 256), device=dev) * 0.02
b = torch.zeros(256, device=dev)
wp = mmr.ops.pack_conv_weights(w, torch.float32, x3=True)
for _ in range(2): y = mmr.ops.conv3d_k3(x, wp, b, 256, out_f32=True, x3=True)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3): y = mmr.ops.conv3d_k3(x, wp, b, 256, out_f32=True, x3=True)
e.record(); torch.cuda.synchronize()
print("x3 256->256 C2 layer: %.3f ms" % (a.elapsed_time(e) / 3))
