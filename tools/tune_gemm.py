#!/usr/bin/env python3
"""Fill geo4d_amd/tuning/gfx950.json on a GPU box: run one eager U-Net forward + one 4-modality decode at the bench
configuration (and at the test configurations) with autotuning on, then save the table. Copy the printed JSON back
(gpurun merges gpurun_out/)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geo4d_amd import ops  # noqa: E402
import bench  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gfx950.json")
    dev = torch.device("cuda:0")
    old = dict(ops._tune_table())
    keep = "--keep" in sys.argv          # --keep: only shapes MISSING from the table are measured (a new mode's launches), the rest keep their entry
    drop = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--drop-prefix=")]      # with --keep: forget the entries whose key starts with this (e.g. 4/ = the two-pass f16 launches) so that they are re-measured
    for suf in [a.split("=", 1)[1] for a in sys.argv if a.startswith("--drop-suffix=")]:      # e.g. --drop-suffix=x11o: the f16-row-output launches
        for k in [k for k in ops._tune_table() if k.endswith(suf)]:
            del ops._tune_table()[k]
            old.pop(k, None)
    batches = [int(a.split("=", 1)[1]) for a in sys.argv if a.startswith("--batch=")] or [1]      # --batch=2: the shapes of two windows per step (run_clip window_batch)
    sys.argv = [a for a in sys.argv if not a.startswith("--drop-prefix=") and not a.startswith("--batch=") and not a.startswith("--drop-suffix=")]
    for pre in drop:
        for k in [k for k in ops._tune_table() if k.startswith(pre)]:
            del ops._tune_table()[k]
            old.pop(k, None)
    if "--exact" in sys.argv:            # pre-split launches get their own entries (ops.TUNE_EXACT) instead of borrowing the raw-activation one
        sys.argv.remove("--exact")
        ops.TUNE_EXACT = 1
    if keep:
        sys.argv.remove("--keep")
    else:
        ops._TUNE = {}                   # re-measure every shape this run touches; shapes it does not touch keep their entry
    for dtype in (sys.argv[2:] or ["bf16"]):
        model, pvae = bench.build(dtype, dev)
        T, h, w = 16, 40, 64
        g = torch.Generator().manual_seed(0)
        from geo4d_amd.pipeline import decode_modalities
        for nb in batches:
            x = torch.randn((nb, 16, T, h, w), generator=g).to(dev)
            zc = torch.randn((nb, 4, T, h, w), generator=g).to(dev)
            ctx = torch.randn((nb, 77 + 16 * T, 1024), generator=g).to(dev)
            t = torch.full((nb,), 500, device=dev)
            y = model.apply_model(x, t, {"c_crossattn": [ctx], "c_concat": [zc]}, fs=torch.full((nb,), 24, device=dev))
            decode_modalities(model, y, pvae)
            torch.cuda.synchronize()
        del model, pvae
    fresh = {k: v for k, v in ops._TUNE.items() if (not keep or k not in old)}
    changed = sum(1 for k, v in fresh.items() if old.get(k) != v)
    ops._TUNE = {**old, **fresh}
    print(f"re-measured {len(fresh)} shapes, {changed} changed")
    with open(out + ".log", "w") as f:
        for key, best, ms, fin in ops.TUNE_LOG:
            f.write(f"{key} -> t{best[0]}/s{best[1]} {ms * 1e3:.1f} us | was {old.get(key)} | " + "  ".join(f"t{t}/s{s2} {u}" for t, s2, u in fin) + "\n")
    ops.save_tuning(out)
    print("saved", len(ops._tune_table()), "entries to", out)


if __name__ == "__main__":
    main()
