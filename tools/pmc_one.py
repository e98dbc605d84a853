#!/usr/bin/env python3
"""Launch one conv_gemm shape N times (for rocprofv3 --pmc passes). usage: pmc_one.py <name-filter> [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geo4d_amd import ops
import tools.gemm_bench as gb  # noqa

flt, iters = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev, dt = torch.device("cuda:0"), torch.bfloat16
for name, kind, geo, N, Cin, cnt in gb.SHAPES:
    if flt not in name:
        continue
    if kind == "c3":
        F_, H, W = geo
        M, K = F_ * H * W, 9 * Cin
        x = torch.randn((M, Cin), device=dev).to(dt); w = (torch.randn((N, K), device=dev) / K ** 0.5).to(dt)
        b = torch.randn((N,), device=dev); r = torch.randn((M, N), device=dev).to(dt)
        for _ in range(iters):
            ops.conv2d(x, w, b, F=F_, Hin=H, Win=W, KH=3, KW=3, pad=1, residual=r)
    else:
        M, K = geo, Cin
        x = torch.randn((M, K), device=dev).to(dt); w = (torch.randn((N, K), device=dev) / K ** 0.5).to(dt)
        b = torch.randn((N,), device=dev)
        for _ in range(iters):
            ops.linear(x, w, b, act=2 if kind == "geglu" else 0)
    torch.cuda.synchronize()
    print("ran", name, M, N, K)
