#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes over `tools/profile_unet.py 1` (warm-up forward + ONE steady-state forward).

usage: pmc_summary.py <out.md> <out.json> <counter_collection.csv> [...]
Only the dispatches of the LAST U-Net forward are counted (it starts at the second-to-last timestep_embedding launch:
the forward embeds `t` and `fs` first). Corrections per MI355X_MICROARCH.md (HBM / rocprofv3 section):
FETCH_SIZE and WRITE_SIZE are in KiB -> x1024; gfx950 tallies wide coalesced reads at half size -> FETCH_SIZE x2
(all our global reads are 16 B per lane); GRBM_GUI_ACTIVE is summed over the 8 XCDs.
"""
import csv
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("geo4d_gemm::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^()]*?>)?)", name)
    s = m.group(1) if m else name
    return s[:70]


def klass(n):
    """Kernel class of a short kernel name (the classes of tools/prof_phases.py)."""
    for pre, c in (("conv_gemm_v3", "conv_gemm third generation"), ("conv_gemm_v2", "conv_gemm second generation"), ("conv_gemm_kernel", "conv_gemm first generation"),
                   ("splitk_reduce", "split-K reduce"), ("gn_", "GroupNorm"), ("ln_kernel", "LayerNorm"), ("ln16_kernel", "LayerNorm"), ("flash_attn", "attention"), ("temporal_attn", "attention")):
        if n.startswith(pre):
            return c
    return "other"


def main():
    out_md, out_json, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    per = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum
    calls = defaultdict(set)
    for fi, f in enumerate(files):
        rows = list(csv.DictReader(open(f, newline="")))
        marks = sorted({int(r["Dispatch_Id"]) for r in rows if "timestep_embedding" in r["Kernel_Name"]})
        assert len(marks) >= 2, f"{f}: no U-Net forward found"
        start = marks[-2]
        for r in rows:
            d = int(r["Dispatch_Id"])
            if d < start:
                continue
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if fi == 0:                  # dispatch ids differ between passes (first-use tuning): count launches in ONE pass
                calls[k].add(d)
    tot = defaultdict(float)
    for k in per:
        for c, v in per[k].items():
            tot[c] += v
    gb = lambda v: v * 1024 / 1e9
    has_mfma = "SQ_VALU_MFMA_BUSY_CYCLES" in tot and "GRBM_GUI_ACTIVE" in tot

    def busy(d):
        if not has_mfma or d.get("GRBM_GUI_ACTIVE", 0) == 0:
            return float("nan")
        return d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)

    lines = ["| kernel | launches | fetch GB (x2) | write GB | MFMA pipe busy |", "|---|---|---|---|---|"]
    for k in sorted(per, key=lambda k: -(2 * per[k].get("FETCH_SIZE", 0) + per[k].get("WRITE_SIZE", 0))):
        d = per[k]
        if gb(2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) < 0.05:
            continue
        lines.append(f"| `{k}` | {len(calls[k])} | {gb(2 * d.get('FETCH_SIZE', 0)):.2f} | {gb(d.get('WRITE_SIZE', 0)):.2f} | {100 * busy(d):.1f} % |")
    lines.append(f"| **whole forward** | {sum(len(v) for v in calls.values())} | **{gb(2 * tot.get('FETCH_SIZE', 0)):.1f}** | "
                 f"**{gb(tot.get('WRITE_SIZE', 0)):.1f}** | **{100 * busy(tot):.1f} %** |")
    by_class = defaultdict(float)
    for k, d in per.items():
        by_class[klass(k)] += gb(2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0))
    lines += ["", "| kernel class | GB per forward (2 x fetch + write) |", "|---|---|"] + [f"| {c} | {v:.1f} |" for c, v in sorted(by_class.items(), key=lambda kv: -kv[1])]
    open(out_md, "w").write("\n".join(lines) + "\n")
    json.dump({"per_unet_forward": {"fetch_bytes_x2": 2 * tot.get("FETCH_SIZE", 0) * 1024, "write_bytes": tot.get("WRITE_SIZE", 0) * 1024,
                                    "mfma_pipe_busy_frac": busy(tot), "dispatches": sum(len(v) for v in calls.values()),
                                    "by_class": {c: round(v, 2) for c, v in by_class.items()}},
               "source": out_md}, open(out_json, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
