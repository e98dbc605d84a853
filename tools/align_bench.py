#!/usr/bin/env python3
"""Per-iteration cost of the global-alignment loop (geo4d_amd/align.py + csrc/align.hip) at clip size: G windows of 16 frames at
HxW, stride 4 (BASELINE configs[2]/[3]: 14 windows for 64 frames, 30 for 128). Reports the fused residual kernel's achieved HBM
bandwidth against its algorithmic bytes (20 B read + 4 B written per window-frame pixel, + the depth map once per image).
usage: align_bench.py [n_frames] [H] [W]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geo4d_amd.align import GroupAligner
from geo4d_amd.pipeline import window_slices
n, H, W = (int(a) for a in (sys.argv[1:4] + ["128", "320", "512"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
groups = [list(range(s.start, s.stop)) for s in window_slices(n, 4, 16)]
G, S = len(groups), 16
g = torch.Generator().manual_seed(0)
pred = (torch.randn((G, S, H, W, 3), generator=g) * 0.3 + torch.tensor([0.0, 0.0, 2.0])).to(dev)
conf = (torch.rand((G, S, H, W), generator=g) * 4 + 0.5).to(dev)
for chunk in (1024, 2048, 4096):
    a = GroupAligner(groups, pred, conf, temporal_smoothing_weight=0.015, translation_weight=1.0, chunk_pixels=chunk)
    a.P["im_depthmaps"] += 0.7
    for _ in range(3):
        a.loss_and_grads()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        a.loss_and_grads()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    res = {}
    for graph in (False, True):
        a.compute_global_alignment(niter=5, lr=0.01, schedule="linear", use_graph=graph)      # warm-up (allocator, capture path)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.compute_global_alignment(niter=100, lr=0.01, schedule="linear", use_graph=graph)
        torch.cuda.synchronize()
        res[graph] = (time.perf_counter() - t0) * 1e3 / 100
    bytes_alg = G * S * H * W * 20 + a.n * H * W * 8            # residual kernel: pred 12 + conf 4 + depth gradient 4 B per slot-pixel
    bytes_it = bytes_alg + a.n * H * W * 24                      # + fused Adam on the depth maps: param, grad, m, v read, param, m, v written
    print(f"align {n} frames = {G} windows of 16 at {H}x{W}, chunk {chunk}: loss+grads eager {ms:.3f} ms; full Adam iteration eager "
          f"{res[False]:.3f} ms, hipGraph replay {res[True]:.3f} ms = {bytes_it / res[True] / 1e6:.0f} GB/s of {bytes_it / 1e9:.2f} GB algorithmic "
          f"({100 * bytes_it / res[True] / 1e6 / 8000:.1f} % of the 8 TB/s HBM peak)", flush=True)
