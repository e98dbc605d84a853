#!/usr/bin/env python3
"""EXPERIMENT (GEO4D_TSTAMP builds): where a bf16x3 256x128 workgroup spends its time on one long-K conv: prologue / steady K loop /
tail + epilogue, from per-workgroup wall-clock stamps (100 MHz)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geo4d_amd import ops, pack, _lib
dev = torch.device("cuda:0")
F_, H, W, Cin, N = 48, 40, 64, 512, 512
x = torch.randn((F_ * H * W, Cin), device=dev)
w = pack.split_bf16(torch.randn((N, 9 * Cin), device=dev) * 0.02)
b = torch.randn(N, device=dev)
r = torch.randn((F_ * H * W, N), device=dev)
fn = lambda: ops.conv2d(x, w, b, F=F_, Hin=H, Win=W, KH=3, KW=3, pad=1, residual=r, tile_hint=11, split_k=0)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
nblk = (F_ * H * W // 256) * (N // 128)
buf = (C.c_ulonglong * (8 * nblk))()
assert lib.geo4d_debug_tstamps(buf, 8 * nblk) == 0
raw = np.frombuffer(buf, dtype=np.uint64).reshape(nblk, 8).astype(np.int64)
t, cyc = raw[:, :4], raw[:, 4:]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
print(f"kernel {e0.elapsed_time(e1) * 1e3:.1f} us; {nblk} workgroups; span of stamps {us.max():.1f} us")
pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
nslab = 9 * Cin // 32
for name, v in (("prologue (entry -> steady loop)", pro), (f"steady loop ({nslab - 3}..{nslab - 2} slabs)", loop), ("tail slabs + epilogue", epi), ("whole workgroup", us[:, 3] - us[:, 0])):
    print(f"  {name:34s} mean {v.mean():8.2f} us  min {v.min():8.2f}  max {v.max():8.2f}")
print(f"  per slab in the steady loop: {loop.mean() / (nslab - 2) * 1e3:.0f} ns (24 MFMAs per wave, 48 per SIMD = 640 ns at the 2.4 GHz peak rate)")
print(f"  shader clock inside the steady loop: {((cyc[:, 2] - cyc[:, 1]) / (loop * 1e-6)).mean() / 1e9:.3f} GHz (s_memtime ticks / wall time)")
first = np.sort(us[:, 0])
print("  workgroup start times (us), every 256th:", np.round(first[::256], 1).tolist())
