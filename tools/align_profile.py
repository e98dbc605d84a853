#!/usr/bin/env python3
"""50 replayed Adam iterations of the global alignment at clip size (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geo4d_amd.align import GroupAligner
from geo4d_amd.pipeline import window_slices
n, H, W = 128, 320, 512
dev = torch.device("cuda:0")
groups = [list(range(s.start, s.stop)) for s in window_slices(n, 4, 16)]
G, S = len(groups), 16
g = torch.Generator().manual_seed(0)
pred = (torch.randn((G, S, H, W, 3), generator=g) * 0.3 + torch.tensor([0.0, 0.0, 2.0])).to(dev)
conf = (torch.rand((G, S, H, W), generator=g) * 4 + 0.5).to(dev)
a = GroupAligner(groups, pred, conf, temporal_smoothing_weight=0.015, translation_weight=1.0)
a.P["im_depthmaps"] += 0.7
a.compute_global_alignment(niter=52, lr=0.01, schedule="linear", use_graph=True)
torch.cuda.synchronize()
