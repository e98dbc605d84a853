#!/usr/bin/env python3
"""GroupNorm / LayerNorm op timing at the shapes of the Geo4D hot path (HIP events on the launch stream).
GB/s counts the algorithmic traffic: GroupNorm = 2 reads + 1 write of the tensor, LayerNorm = 1 read + 1 write.
Back-to-back eager calls: below ~30 us per op the number is the Python call overhead, not the GPU.
usage (GPU box): python tools/norm_bench.py [--dtype bf16]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geo4d_amd import ops

GN = [("unet L0 320ch", 16, 2560, 320), ("unet L0 640ch (skip concat)", 16, 2560, 640), ("unet L1 640ch", 16, 640, 640),
      ("unet L2 1280ch", 16, 160, 1280), ("unet L3 1280ch", 16, 40, 1280),
      ("vae 512ch @40x64", 16, 2560, 512), ("vae 512ch @80x128", 16, 10240, 512), ("vae 256ch @160x256", 16, 40960, 256),
      ("vae 128ch @320x512", 16, 163840, 128)]
LN = [("L0", 40960, 320), ("L1", 10240, 640), ("L2", 2560, 1280)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    dev = torch.device("cuda:0")
    print(f"{'groupnorm':34s} {'fps':>4s} {'MB':>8s} {'us':>9s} {'TB/s':>8s}")
    for name, F, HW, C in GN:
        x = torch.randn((F * HW, C), device=dev).to(dt)
        g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
        out = torch.empty_like(x)
        for fps in ((1, 16) if name.startswith("unet") else (1,)):
            us = timeit(lambda: ops.groupnorm(x, g, b, F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=True, out=out), a.iters)
            mb = x.numel() * x.element_size() / 1e6
            print(f"{name:34s} {fps:4d} {mb:8.1f} {us:9.1f} {3 * mb / us:8.2f}")
    print(f"{'layernorm':34s} {'':>4s} {'MB':>8s} {'us':>9s} {'TB/s':>8s}")
    for name, M, C in LN:
        x = torch.randn((M, C), device=dev).to(dt)
        g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
        out = torch.empty_like(x)
        us = timeit(lambda: ops.layernorm(x, g, b, out=out), a.iters)
        mb = x.numel() * x.element_size() / 1e6
        print(f"{name:34s} {'':4s} {mb:8.1f} {us:9.1f} {2 * mb / us:8.2f}")


if __name__ == "__main__":
    main()
