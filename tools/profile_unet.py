#!/usr/bin/env python3
"""Run N eager U-Net forwards at the bench configuration (for rocprofv3 --kernel-trace): python tools/profile_unet.py [n] [dtype]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
model, pvae = bench.build(dtype, dev)
del pvae
T, h, w = 16, 40, 64
g = torch.Generator().manual_seed(0)
x = torch.randn((1, 16, T, h, w), generator=g).to(dev)
zc = torch.randn((1, 4, T, h, w), generator=g).to(dev)
ctx = torch.randn((1, 77 + 16 * T, 1024), generator=g).to(dev)
t = torch.tensor([500], device=dev)
fs = torch.tensor([24], device=dev)
cond = {"c_crossattn": [ctx], "c_concat": [zc]}
for i in range(n + 1):
    if i == 1:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    y = model.apply_model(x, t, cond, fs=fs)
e1.record()
torch.cuda.synchronize()
print(f"eager forward: {e0.elapsed_time(e1) / n:.2f} ms")
