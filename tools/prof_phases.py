#!/usr/bin/env python3
"""Split a rocprofv3 --kernel-trace CSV of a bench.py run into its phases BY THE KERNEL SEQUENCE and summarise each: replayed DDIM
steps (gather_timestep ... ddim_step / advance_index), eager U-Net forwards (the live GEMM / attention timelines of bench.py: the next
L kernels after a timestep_embedding outside a step, L = kernels per forward inside a step), and everything else of the library = the
4-modality VAE decode (+ the VAE encode of a clip run). Prints the counts the trace itself implies (windows, steps, forwards, decodes) -
the round-4 trace header got them wrong by hand - and one per-class / top-kernel table per phase.
usage: tools/prof_phases.py <..._kernel_trace.csv> [top_n]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"geo4d_gemm::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n).strip()


def klass(n):
    if n.startswith("conv_gemm_v3"):
        return "conv_gemm third generation"
    if n.startswith("conv_gemm_v2"):
        return "conv_gemm second generation"
    if n.startswith("conv_gemm_kernel"):
        return "conv_gemm first generation"
    if n.startswith("splitk_reduce"):
        return "split-K reduce"
    if n.startswith("gn_"):
        return "GroupNorm"
    if n.startswith("ln_kernel") or n.startswith("ln16_kernel"):
        return "LayerNorm"
    if n.startswith("flash_attn") or n.startswith("temporal_attn"):
        return "attention"
    if n.startswith("softmax_rows"):
        return "VAE attention softmax"
    return "other"


OURS = ("conv_gemm", "splitk_reduce", "gn_", "ln_kernel", "ln16_kernel", "cast_rows_f16", "split_rows_bf16", "flash_attn", "temporal_attn", "softmax_rows", "tokens_from_ncthw", "concat_channels",
        "timestep_embedding", "embed_tokens", "linear_small", "ddim_step", "cfg_combine", "gather_timestep", "advance_index", "ray_", "align_", "adam_",
        "lad_", "select_")


def main():
    path, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # kernels per forward inside a replayed step = the SHORTEST gather_timestep .. ddim_step span (the sampler's first, eager step of a
    # process also holds the tuner's trial launches of shapes missing from the table)
    L = None
    gat = [i for i, r in enumerate(rows) if r[2].startswith("gather_timestep")]
    dd = [i for i, r in enumerate(rows) if r[2].startswith("ddim_step")]
    import bisect
    for i in gat:
        k = bisect.bisect_right(dd, i)
        if k < len(dd):
            L = dd[k] - i - 1 if L is None else min(L, dd[k] - i - 1)
    phase = [None] * len(rows)
    i, steps, eager = 0, 0, 0
    while i < len(rows):
        n = rows[i][2]
        if n.startswith("gather_timestep") and L is not None:
            k = next((q for q in range(i, len(rows)) if rows[q][2].startswith("advance_index")), len(rows) - 1)
            replay = (k - i) <= L + 3
            for q in range(i, k + 1):
                phase[q] = "replayed DDIM step" if replay else "eager first step of a sampler (with tuner trial launches)"
            steps += 1 if replay else 0
            i = k + 1
            continue
        if n.startswith("timestep_embedding") and L is not None:
            for k in range(i, min(len(rows), i + L)):
                phase[k] = "eager U-Net forward"
            eager += 1
            i += L
            continue
        phase[i] = "decode / encode / other library kernels" if n.startswith(OURS) else "torch (ATen) kernels"
        i += 1
    tot = defaultdict(float)
    byk = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for (s, e, n), ph in zip(rows, phase):
        tot[ph] += (e - s) / 1e6
        c = byk[ph][n]
        c[0] += 1
        c[1] += (e - s) / 1e6
    heads = sum(c[0] for n, c in byk["decode / encode / other library kernels"].items() if n.startswith("conv_gemm_kernel") and ", 128, 32, " in n)
    print(f"# phases of `{path.split('/')[-1]}` by kernel sequence")
    print(f"{len(rows)} dispatches, {sum(tot.values()):.1f} ms of kernel time; kernels per U-Net forward inside a step: {L}; replayed DDIM steps: {steps} "
          f"(= {steps / 50:.2f} windows of 50); eager forwards: {eager}; NCTHW-head launches outside forwards: {heads} (5 per 4-modality decode -> {heads / 5:.1f} decodes)\n")
    for ph in ("replayed DDIM step", "eager first step of a sampler (with tuner trial launches)", "eager U-Net forward", "decode / encode / other library kernels", "torch (ATen) kernels"):
        if ph not in tot:
            continue
        units = {"replayed DDIM step": steps, "eager U-Net forward": eager, "decode / encode / other library kernels": max(1.0, heads / 5)}.get(ph, 1) or 1
        print(f"## {ph}: {tot[ph]:.1f} ms total = {tot[ph] / units:.2f} ms per {'unit' if ph.startswith('torch') else ph.split(' /')[0]}\n")
        cls = defaultdict(float)
        for n, (c, ms) in byk[ph].items():
            cls[klass(n)] += ms
        print("| class | ms | % of phase |\n|---|---|---|")
        for k, ms in sorted(cls.items(), key=lambda kv: -kv[1]):
            print(f"| {k} | {ms:.1f} | {100 * ms / tot[ph]:.1f} |")
        print("\n| kernel | calls | total ms | % of phase | avg us |\n|---|---|---|---|---|")
        for n, (c, ms) in sorted(byk[ph].items(), key=lambda kv: -kv[1][1])[:top]:
            print(f"| `{n[:100]}` | {c} | {ms:.1f} | {100 * ms / tot[ph]:.1f} | {1e3 * ms / c:.1f} |")
        print()


if __name__ == "__main__":
    main()
