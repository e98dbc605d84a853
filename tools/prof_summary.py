#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite) into the per-kernel table we commit under profiles/.
usage: tools/prof_summary.py <results.db> [top_n]"""
import re
import sqlite3
import sys

db, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
c = sqlite3.connect(db)
tot = c.execute("select sum(end-start)/1e6, count(*) from kernels").fetchone()
print(f"# rocprofv3 --kernel-trace summary: {tot[1]} dispatches, {tot[0]:.1f} ms total kernel time")
print("| kernel | calls | total ms | % | avg us | min us | max us | VGPR | LDS B |")
print("|---|---|---|---|---|---|---|---|---|")
q = ("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
     "max(vgpr_count), max(lds_size) from kernels group by name order by 3 desc limit ?")
for n, cnt, ms, avg, mn, mx, vg, lds in c.execute(q, (top,)):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n)[:80]
    print(f"| `{n}` | {cnt} | {ms:.2f} | {100 * ms / tot[0]:.1f} | {avg:.1f} | {mn:.1f} | {mx:.1f} | {vg} | {lds} |")
