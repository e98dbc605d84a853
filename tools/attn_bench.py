#!/usr/bin/env python3
"""Spatial self-attention throughput at the four U-Net levels (B*H, N, d=64) + MFMA fraction, for every A/B build of the
kernel (geo4d_attention_t.variant). usage: attn_bench.py [dtype ...]   (dtype: bf16 f16 f32 bf16x3)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geo4d_amd import ops
dev = torch.device("cuda:0")
LEVELS = ((16, 5, 2560, 5), (16, 10, 640, 5), (16, 20, 160, 5), (16, 20, 40, 1), (16, 5, 2304, 0), (4, 5, 9216, 0))
for name in (sys.argv[1:] or ["bf16"]):
    x3 = name == "bf16x3"
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "bf16x3": torch.float32}[name]
    passes = 3 if x3 else 1
    for variant in (1, 2, 3, 4, "1ps", "3ps", "4ps", "5ps"):
        ps = isinstance(variant, str)                  # bf16x3 only: q | k and V^T in the producers' pre-split format (qkv_split, round 4)
        if (variant == 2 and dt == torch.float32) or (ps and not x3) or (variant == 4 and dt == torch.float32):
            continue                                   # 4 / 5 = flash_attn2_kernel (round 4): 16-bit types and pre-split bf16x3 only
        vnum = int(variant[0]) if ps else variant
        tot_ms = 0
        for F_, H, N, cnt in LEVELS:
            C_ = H * 64
            qk = torch.randn((F_ * N, 2 * C_), device=dev).to(dt)
            vt = torch.randn((F_ * C_, N), device=dev).to(dt)
            if ps:
                from geo4d_amd import pack
                qs, vs = ops.SplitAct.wrap(pack.split_bf16(qk)), ops.SplitAct.wrap(pack.split_bf16(vt))
                fn = lambda: ops.attention(qs[:, :2 * C_], [(qs[:, 2 * C_:], vs, N, 1, C_ * 2 * N)], B=F_, H=H, Nq=N, scale=0.125, x3=True, variant=vnum, qkv_split=True)
            else:
                fn = lambda: ops.attention(qk[:, :C_], [(qk[:, C_:], vt, N, 1, C_ * N)], B=F_, H=H, Nq=N, scale=0.125, x3=x3, variant=variant)
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            fl = 4.0 * F_ * H * N * N * 64
            tot_ms += us * cnt / 1e3
            print(f"{name} v{variant} self-attn F*H={F_*H:4d} N={N:5d}: {us:8.1f} us  {fl/us/1e6:7.1f} algorithmic TF/s  ({passes*fl/us/1e6/2500*100:4.1f}% of bf16 MFMA peak issued)  x{cnt}")
        print(f"{name} v{variant}: spatial self-attention per U-Net forward: {tot_ms:.2f} ms", flush=True)
