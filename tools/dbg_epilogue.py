#!/usr/bin/env python3
"""debug: where does the fast register epilogue differ from the reference? (one-off)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geo4d_amd import ops, pack
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
M, K, N = 1000, 320, 456
x = torch.randn((M, K), generator=g).to(dev); w = (torch.randn((N, K), generator=g) * 0.05).to(dev)
b = torch.randn((N,), generator=g).to(dev); r = torch.randn((M, N), generator=g).to(dev)
wp = pack.pack_linear(w, "bf16x3")
xs = ops.SplitAct.wrap(pack.split_bf16(x))
for tile in (25, 22, 72, 28):
    for name, kw in (("plain", {}), ("bias", dict(bias=b)), ("res", dict(residual=r)), ("bias+res", dict(bias=b, residual=r))):
        ref = x.double() @ w.double().t() + (b.double() if "bias" in kw else 0) + (r.double() if "residual" in kw else 0)
        outs = []
        for rep in range(3):
            out = ops.linear(xs, wp, kw.get("bias"), residual=kw.get("residual"), tile_hint=tile)
            torch.cuda.synchronize()
            outs.append(out.clone())
        err = (outs[0].double() - ref).abs()
        bad = (err > 1e-3).nonzero()
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        msg = f"tile {tile} {name:9s}: max err {err.max().item():.3e}  bad {bad.shape[0]}  repeatable {same}"
        if bad.shape[0]:
            rows, cols = bad[:, 0], bad[:, 1]
            msg += f" | rows {rows.min().item()}..{rows.max().item()} (mod16 {sorted(set((rows % 16).tolist()))[:16]}) cols {cols.min().item()}..{cols.max().item()} (mod16 {sorted(set((cols % 16).tolist()))})"
            i = bad[0]
            msg += f" | first ({i[0].item()},{i[1].item()}): got {outs[0][i[0], i[1]].item():.4f} want {ref[i[0], i[1]].item():.4f} r {r[i[0], i[1]].item():.4f} b {b[i[1]].item():.4f}"
        print(msg, flush=True)
