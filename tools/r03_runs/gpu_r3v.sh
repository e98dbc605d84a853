#!/bin/bash
# GroupNorm with the merge folded into the apply launch (2 launches instead of 3): parity, then A/B against the three-launch path
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3v; mkdir -p $O; cd $R; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_presplit_gpu.py tests/test_parity_gpu.py -q -x ) > $O/tests.log 2>&1; tail -2 $O/tests.log
for m in 0 1 0 1; do echo "== GEO4D_GN_MERGE_IN_APPLY=$m" >> $O/norm.log; GEO4D_GN_MERGE_IN_APPLY=$m timeout 100 python tools/norm_bench.py --dtype f32 --iters 50 2>&1 | grep -v amdgpu | head -15 >> $O/norm.log; done
python - <<'PY'
import re
rows={}
cur=None
for l in open("gpurun_out/r3v/norm.log"):
    if l.startswith("=="): cur=l.strip()[-1]; continue
    m=re.match(r"(.{34})\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", l)
    if m: rows.setdefault((m.group(1).strip(), m.group(2)), {}).setdefault(cur, []).append(float(m.group(4)))
for k,v in rows.items(): print("%-34s fps %2s   3 launches %s us   2 launches %s us" % (k[0], k[1], v.get('0'), v.get('1')))
PY
