#!/usr/bin/env python3
"""How much of a replayed DDIM step is idle between kernels? Reads the rocprofv3 --kernel-trace CSV of
`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode` and, for the timed window's 50 replayed steps (the last 50
runs of ddim_step_kernel before the final decode), compares the sum of kernel durations with the wall span, and lists the gap
distribution. usage: graph_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
csv.field_size_limit(1 << 30)
rows = []
for r in csv.DictReader(open(sys.argv[1], newline="")):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "ddim_step" in r[2]]
print(f"{len(rows)} dispatches, {len(marks)} ddim_step launches")
# steps = intervals between consecutive ddim_step kernels; take the last 49 full intervals
steps = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1)]
sel = [s for s in steps if 1000 < s[1] - s[0] < 1400][-49:]
busy = span = 0
gaps = []
by_prev = collections.Counter()
for a, b in sel:
    seg = rows[a + 1:b + 1]
    span += seg[-1][1] - rows[a][1]
    busy += sum(e - s for s, e, _ in seg)
    prev_end = rows[a][1]
    for (s, e, n), p in zip(seg, [rows[a]] + seg[:-1]):
        g = s - p[1]
        gaps.append(g)
        by_prev[p[2].split("<")[0].split("(")[0][-40:]] += max(g, 0)
n = len(sel)
print(f"{n} replayed steps of {len(rows[sel[0][0] + 1:sel[0][1] + 1])} kernels: span {span / n / 1e6:.3f} ms per step, kernel time {busy / n / 1e6:.3f} ms, "
      f"idle between kernels {(span - busy) / n / 1e6:.3f} ms ({100 * (span - busy) / span:.1f} %)")
gs = sorted(gaps)
q = lambda f: gs[int(f * (len(gs) - 1))] / 1e3
print(f"gap between consecutive kernels (end -> next start), us: min {q(0):.2f}  p10 {q(.1):.2f}  median {q(.5):.2f}  p90 {q(.9):.2f}  p99 {q(.99):.2f}  max {q(1):.2f}; "
      f"negative (overlapped) {sum(1 for g in gaps if g < 0)} of {len(gaps)}")
print("idle time by preceding kernel (us per step):")
for k, v in by_prev.most_common(10):
    print(f"  {k:42s} {v / n / 1e3:8.1f}")
