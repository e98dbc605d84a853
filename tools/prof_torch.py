import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
model, pvae = bench.build("bf16", dev)
T, h, w = 16, 40, 64
x = torch.randn((1, 16, T, h, w), device=dev); zc = torch.randn((1, 4, T, h, w), device=dev)
ctx = torch.randn((1, 77 + 16 * T, 1024), device=dev); t = torch.tensor([500], device=dev); fs = torch.tensor([24], device=dev)
cond = {"c_crossattn": [ctx], "c_concat": [zc]}
model.apply_model(x, t, cond, fs=fs); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    model.apply_model(x, t, cond, fs=fs); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))
