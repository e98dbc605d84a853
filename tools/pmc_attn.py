#!/usr/bin/env python3
"""Launch the level-0 self-attention shape N times (for rocprofv3 --pmc passes over flash_attn_kernel). usage: pmc_attn.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geo4d_amd import ops
dev, dt = torch.device("cuda:0"), torch.bfloat16
F_, H, N = 16, 5, 2560
C_ = H * 64
qk = torch.randn((F_ * N, 2 * C_), device=dev).to(dt); vt = torch.randn((F_ * C_, N), device=dev).to(dt)
for _ in range(3):
    ops.attention(qk[:, :C_], [(qk[:, C_:], vt, N, 1, C_ * N)], B=F_, H=H, Nq=N, scale=0.125)
torch.cuda.synchronize()
