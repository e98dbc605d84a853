#!/usr/bin/env python3
"""Per-shape throughput of geo4d_conv_gemm on the GEMM / conv census of one U-Net forward at config A
 and of the VAE decoder. HIP-event timing on the launch stream, random bf16 data.

usage (GPU box): python tools/gemm_bench.py [--dtype bf16] [--iters 20] [--filter conv]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geo4d_amd import ops  # noqa: E402

# (name, kind, M-geometry, N, Cin, count per U-Net forward)
# kind: lin (K=Cin) | c3 (3x3 conv, K=9*Cin, geometry F,H,W) | t3 (temporal 3-tap, K=3*Cin) | geglu
SHAPES = [
    ("L0 conv3x3 320->320", "c3", (16, 40, 64), 320, 320, 7),
    ("L0 conv3x3 640->320", "c3", (16, 40, 64), 320, 640, 2),
    ("L0 conv3x3 960->320", "c3", (16, 40, 64), 320, 960, 1),
    ("L1 conv3x3 640->640", "c3", (16, 20, 32), 640, 640, 6),
    ("L1 conv3x3 1920->640", "c3", (16, 20, 32), 640, 1920, 1),
    ("L2 conv3x3 1280->1280", "c3", (16, 10, 16), 1280, 1280, 7),
    ("L2 conv3x3 2560->1280", "c3", (16, 10, 16), 1280, 2560, 2),
    ("L3 conv3x3 1280->1280", "c3", (16, 5, 8), 1280, 1280, 11),
    ("L3 conv3x3 2560->1280", "c3", (16, 5, 8), 1280, 2560, 3),
    ("L0 conv3d 320", "t3", (16, 2560), 320, 320, 20),
    ("L1 conv3d 640", "t3", (16, 640), 640, 640, 20),
    ("L2 conv3d 1280", "t3", (16, 160), 1280, 1280, 20),
    ("L3 conv3d 1280", "t3", (16, 40), 1280, 1280, 28),
    ("L0 qkv 320->960", "lin", 40960, 960, 320, 15),
    ("L0 proj 320->320", "lin", 40960, 320, 320, 45),
    ("L0 geglu 320->2560", "geglu", 40960, 2560, 320, 10),
    ("L0 ffout 1280->320", "lin", 40960, 320, 1280, 10),
    ("L1 qkv 640->1920", "lin", 10240, 1920, 640, 15),
    ("L1 proj 640->640", "lin", 10240, 640, 640, 45),
    ("L1 geglu 640->5120", "geglu", 10240, 5120, 640, 10),
    ("L1 ffout 2560->640", "lin", 10240, 640, 2560, 10),
    ("L2 qkv 1280->3840", "lin", 2560, 3840, 1280, 15),
    ("L2 proj 1280->1280", "lin", 2560, 1280, 1280, 45),
    ("L2 geglu 1280->10240", "geglu", 2560, 10240, 1280, 10),
    ("L2 ffout 5120->1280", "lin", 2560, 1280, 5120, 10),
    ("VAE conv3x3 512 @40x64 x48f", "c3", (48, 40, 64), 512, 512, 0),
    ("VAE conv3x3 512 @80x128 x48f", "c3", (48, 80, 128), 512, 512, 0),
    ("VAE conv3x3 256 @160x256 x48f", "c3", (48, 160, 256), 256, 256, 0),
    ("VAE conv3x3 128 @320x512 x16f", "c3", (16, 320, 512), 128, 128, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", help="bf16 | f16 | f32 | bf16x3 | f16x2 (two-pass f16 of the bf16x3m mode: 3x3 conv rows, pre-split operands)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--filter", default="")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--split", type=int, default=0)
    ap.add_argument("--vendor", action="store_true", help="also time the vendor library on the same shape (hipBLASLt via F.linear, MIOpen via F.conv2d channels_last): a calibration point, never used by the product path")
    ap.add_argument("--explore", action="store_true", help="time every (tile, split) pair per shape and report the best")
    ap.add_argument("--explore-all", action="store_true", help="per shape: every candidate of the host tuner (ops._CANDIDATES, all three kernel generations x split-K) "
                    "at `--iters` launches each, the 6 best re-measured interleaved; prints the table's choice next to the best (round 4: the tuner's own sample was too noisy)")
    ap.add_argument("--explore2", action="store_true", help="per shape: best 32x32-MFMA configuration (tile hints 1..17) vs best second-generation one (21..29), and every 21..29 tile at split 1")
    ap.add_argument("--tiles", default="", help="comma list of tile hints to time per shape (split 1), e.g. the ablation builds 40..64")
    ap.add_argument("--zeros", action="store_true", help="zero-filled operands: same instruction stream, no data toggling (how much of the time is the chip's power-limited clock?)")
    ap.add_argument("--presplit", action="store_true", help="bf16x3: hand the activation over in the producers' pre-split format (what the networks run)")
    ap.add_argument("--ablate", type=int, default=0, help="bf16x3 only: 1 = skip the in-register operand split (wrong numbers; measures its cost)")
    args = ap.parse_args()
    ops.DEBUG_ABLATE = args.ablate
    x2 = args.dtype == "f16x2"       # the two-pass f16 GEMM of the bf16x3m mode (dtype 4): conv3x3 rows only, pre-split operands
    x3 = args.dtype == "bf16x3"
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "bf16x3": torch.float32, "f16x2": torch.float32}[args.dtype]
    from geo4d_amd import pack
    wcast = (lambda w: pack.split_bf16(w)) if x3 else (lambda w: pack.split_f16(w)) if x2 else (lambda w: w.to(dt))
    acast = (lambda a: ops.SplitAct.wrap(pack.split_bf16(a))) if (x3 and args.presplit) else (lambda a: a)
    if x2:
        def acast(a):      # the two-pass GEMM's A operand: plain f16 rows (round 6)
            return a.to(torch.float16).contiguous()
    dev = torch.device("cuda:0")
    if args.zeros:
        torch.randn = lambda *a, **k: torch.zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "generator"})
    tot_ms, tot_tf = 0.0, 0.0
    print(f"{'shape':34s} {'M':>8s} {'N':>6s} {'K':>6s} {'us':>9s} {'TF/s':>8s}  x count -> ms/forward")
    for name, kind, geo, N, Cin, cnt in SHAPES:
        if args.filter and args.filter not in name:
            continue
        if kind == "c3":
            F_, H, W = geo
            M, K = F_ * H * W, 9 * Cin
            x = torch.randn((M, Cin), device=dev).to(dt)
            w = wcast(torch.randn((N, K), device=dev) / K ** 0.5)
            b = torch.randn((N,), device=dev)
            r = torch.randn((M, N), device=dev).to(dt)
            xa = acast(x)
            fn = lambda tile=args.tile, split=args.split: ops.conv2d(xa, w, b, F=F_, Hin=H, Win=W, KH=3, KW=3, pad=1, residual=r, tile_hint=tile, split_k=split)
        elif kind == "t3":
            T, HW = geo
            M, K = T * HW, 3 * Cin
            x = torch.randn((M, Cin), device=dev).to(dt)
            w = wcast(torch.randn((N, K), device=dev) / K ** 0.5)
            b = torch.randn((N,), device=dev)
            xa = acast(x)
            fn = lambda tile=args.tile, split=args.split: ops.conv_gemm(xa, w, torch.empty((M, N), device=dev, dtype=dt), M=M, N=N, K=K, Cin=Cin, lda=xa.stride(0), ldw=w.stride(0), ldo=N, T=T, Hin=HW, Win=1, Hout=HW, Wout=1, KT=3, pt=1, bias=b, residual=x, ldr=Cin, tile_hint=tile, split_k=split)
        else:
            M, K = geo, Cin
            x = torch.randn((M, K), device=dev).to(dt)
            w = wcast(torch.randn((N, K), device=dev) / K ** 0.5)
            b = torch.randn((N,), device=dev)
            act = 2 if kind == "geglu" else 0
            xa = acast(x)
            so = bool(x3 and args.presplit and kind == "geglu")      # the network's GEGLU writes the pre-split format for the ff-out GEMM
            fn = lambda tile=args.tile, split=args.split: ops.linear(xa, w, b, act=act, tile_hint=tile, split_k=split, split_out=so)
        def timeit(**kw):
            for _ in range(2):
                fn(**kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn(**kw)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / args.iters
        us = timeit()
        vend = None
        if args.vendor:
            import torch.nn.functional as Fn
            if kind == "c3" and x3:
                vfn = None
            elif kind == "c3":
                xi = x.view(F_, H, W, Cin).permute(0, 3, 1, 2)     # channels_last view
                wi = w.view(N, 3, 3, Cin).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
                vfn = lambda: Fn.conv2d(xi, wi, b.to(dt), padding=1)
            elif kind == "t3":
                vfn = None
            elif x3:     # bf16x3: the vendor library has no such mode; 3 x its bf16 time is the bound a 3-pass scheme can be held to
                xb, wb, bb = x.to(torch.bfloat16), torch.randn((N, K), device=dev).to(torch.bfloat16), b.to(torch.bfloat16)
                vfn = lambda: Fn.linear(xb, wb, bb)
            else:
                vfn = lambda: Fn.linear(x, w, b.to(dt))
            if vfn is not None:
                for _ in range(3):
                    vfn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    vfn()
                e1.record()
                torch.cuda.synchronize()
                vend = e0.elapsed_time(e1) * 1e3 / args.iters
        if args.explore:
            res = []
            for tile in (1, 2, 3, 4, 11, 13, 16, 17):
                for split in (1, 2, 4, 8):
                    try:
                        res.append((timeit(tile=tile, split=split), tile, split))
                    except RuntimeError:
                        pass
            res.sort()
            print("    auto %.1f us | best: %s" % (us, "  ".join("t%d/s%d %.1f" % (t, s2, u) for u, t, s2 in res[:6])))
            us = min(us, res[0][0])
        if args.explore_all:
            res = []
            for tile, split in ops._CANDIDATES:
                try:
                    res.append((timeit(tile=tile, split=split), tile, split))
                except RuntimeError:
                    pass
            res.sort()
            fin = {(t, s2): u for u, t, s2 in res[:6]}
            for _ in range(2):
                for (t, s2) in list(fin):
                    fin[(t, s2)] = min(fin[(t, s2)], timeit(tile=t, split=s2))
            order = sorted(fin.items(), key=lambda kv: kv[1])
            print("    table %.1f us | best: %s" % (us, "  ".join("t%d/s%d %.1f" % (t, s2, u) for (t, s2), u in order)))
            exp_tot = globals().setdefault("_TOTX", [0.0])
            exp_tot[0] += min(us, order[0][1]) * cnt / 1e3
            best_cfg = globals().setdefault("_BESTCFG", {})
            best_cfg[name] = (order[0][0], order[0][1], us)
            us_table = us
            us = min(us, order[0][1])
        if args.tiles:
            row = []
            for t in args.tiles.split(","):
                t, _, sp = t.partition("/")          # "71" or "71/2" (tile / split)
                try:
                    row.append("t%s%s %.1f" % (t, "/" + sp if sp else "", timeit(tile=int(t), split=int(sp or 1))))
                except RuntimeError as e:
                    row.append("t%s n/a" % t)
            print("    " + "  ".join(row))
        if args.explore2:
            def sweep(tiles):
                res = []
                for tile in tiles:
                    for split in (1, 2, 4, 8):
                        try:
                            res.append((timeit(tile=tile, split=split), tile, split))
                        except RuntimeError:
                            pass
                return sorted(res)
            v1, v2 = sweep((1, 2, 3, 4, 11, 13, 16, 17)), sweep(list(range(21, 30)) + [31, 33, 34, 35, 39])
            tot2 = globals().setdefault("_TOT2", [0.0, 0.0])
            tot2[0] += v1[0][0] * cnt / 1e3
            tot2[1] += v2[0][0] * cnt / 1e3
            print("    table %.1f us | v1 best t%d/s%d %.1f | v2 best t%d/s%d %.1f (%.2fx) | v2 split 1: %s" % (
                us, v1[0][1], v1[0][2], v1[0][0], v2[0][1], v2[0][2], v2[0][0], v1[0][0] / v2[0][0],
                "  ".join("t%d %.1f" % (t, u) for u, t, s2 in sorted(v2, key=lambda r: r[1]) if s2 == 1)))
            us = min(us, v1[0][0], v2[0][0])
        tf = 2.0 * M * N * K / us / 1e6
        tot_ms += us * cnt / 1e3
        tot_tf += 2.0 * M * N * K * cnt / 1e12
        print(f"{name:34s} {M:8d} {N:6d} {K:6d} {us:9.1f} {tf:8.1f}  x{cnt:3d} -> {us * cnt / 1e3:7.2f}" + (f"   vendor {vend:8.1f} us ({us / vend:4.2f}x ours/vendor)" if vend and not x3 else "") +
              (f"   3 x vendor bf16 {3 * vend:8.1f} us ({us / (3 * vend):4.2f}x ours/bound)" if vend and x3 else ""))
    if tot_ms:
        print(f"U-Net GEMM census: {tot_tf:.2f} TFLOP in {tot_ms:.1f} ms = {tot_tf / tot_ms * 1e3:.0f} TF/s")
    if args.explore_all:
        print(f"census with the best candidate per shape: {globals()['_TOTX'][0]:.1f} ms")
        for nm, (cfg, u, ut) in globals()["_BESTCFG"].items():
            print(f"  {nm:34s} best t{cfg[0]}/s{cfg[1]} {u:8.1f} us   table {ut:8.1f} us   ({ut / u:4.2f}x)")
    if args.explore2:
        t1, t2 = globals()["_TOT2"]
        print(f"census with the best 32x32-MFMA configuration per shape: {t1:.1f} ms; with the best 21..29 configuration per shape: {t2:.1f} ms")


if __name__ == "__main__":
    main()
