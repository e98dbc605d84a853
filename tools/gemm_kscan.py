#!/usr/bin/env python3
"""Per-tile fixed cost vs per-slab slope of conv_gemm on plain linears: one (M, N) at several K, for given (tile, split) hints and epilogues
(bf16x3, pre-split operands, what the U-Net runs). t(K) = fixed + slope * K/32: `fixed` is prologue + epilogue + launch, `slope` the K loop.
usage (GPU box): python tools/gemm_kscan.py [--dtype bf16x3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geo4d_amd import ops, pack  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16x3")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    x3 = a.dtype == "bf16x3"
    dev = torch.device("cuda:0")
    dt = torch.float32 if x3 else torch.bfloat16
    cases = [  # (label, M, N, [(tile, split) ...], [act ...])
        ("L0 geglu-like", 40960, 2560, [(71, 1), (13, 1), (22, 1), (74, 1)], [0, 2]),
        ("L0 qkv-like", 40960, 960, [(2, 1), (71, 1), (72, 1), (25, 1)], [0]),
        ("L0 proj-like", 40960, 320, [(16, 1), (72, 1), (23, 1)], [0]),
        ("L1 proj-like", 10240, 640, [(4, 1), (1, 1), (17, 1), (25, 1), (74, 1)], [0]),
        ("L2 proj-like", 2560, 1280, [(4, 1), (1, 1), (3, 1), (25, 2), (28, 1)], [0]),
    ]
    for label, M, N, cfgs, acts in cases:
        for act in acts:
            for tile, split in cfgs:
                row = []
                for K in (320, 640, 1280, 2560):
                    x = torch.randn((M, K), device=dev).to(dt)
                    w = torch.randn((N, K), device=dev) / K ** 0.5
                    b = torch.randn((N,), device=dev)
                    if x3:
                        xa, wp = ops.SplitAct.wrap(pack.split_bf16(x)), pack.split_bf16(w)
                    else:
                        xa, wp = x, w.to(dt)
                    so = bool(x3 and act == 2)
                    fn = lambda: ops.linear(xa, wp, b, act=act, tile_hint=tile, split_k=split, split_out=so)
                    try:
                        for _ in range(3):
                            fn()
                    except RuntimeError:
                        row.append(None)
                        continue
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    row.append(e0.elapsed_time(e1) * 1e3 / a.iters)
                if any(v is None for v in row):
                    print(f"{label:14s} M={M:6d} N={N:5d} act={act} t{tile}/s{split}: n/a")
                    continue
                slope = (row[3] - row[1]) / ((2560 - 640) / 32)            # us per 32-k slab
                fixed = row[1] - slope * 640 / 32
                pf = 3 * 2.0 * M * N * 32 / slope / 1e9 if x3 else 2.0 * M * N * 32 / slope / 1e9      # MFMA-issued PF/s... (TF/s / 1000)
                print(f"{label:14s} M={M:6d} N={N:5d} act={act} t{tile}/s{split}: " + "  ".join(f"K{k}={v:7.1f}" for k, v in zip((320, 640, 1280, 2560), row)) +
                      f"  us | slope {slope:6.2f} us/slab = {pf:6.0f} TF/s issued in the loop | fixed {fixed:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
