#!/usr/bin/env python3
"""Lists, for ONE replayed DDIM step of the rocprofv3 kernel trace of the bench command, every dispatch that is not one of this
library's kernels (torch elementwise / fill / copy kernels, blit copies) with its duration and the idle gap that follows it.
usage: graph_step_foreign.py <kernel_trace.csv>"""
import csv, sys
csv.field_size_limit(1 << 30)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"]) for r in csv.DictReader(open(sys.argv[1], newline="")))
marks = [i for i, r in enumerate(rows) if "ddim_step" in r[2]]
steps = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if 1000 < marks[i + 1] - marks[i] < 1400]
a, b = steps[-3]
seg = rows[a:b + 1]
ours = ("geo4d", "anonymous namespace")
for j, (s, e, n, gx) in enumerate(seg[:-1]):
    gap = seg[j + 1][0] - e
    if not any(o in n for o in ours) or gap > 3000:
        print(f"#{j:5d} {n[:110]:110s} grid {gx:>9s} dur {(e - s) / 1e3:8.1f} us  gap after {gap / 1e3:8.1f} us   next: {seg[j + 1][2][:60]}")
