#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3m; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_align_gpu.py -q -x -s ) > $O/align_tests.log 2>&1; echo "rc=$?" >> $O/align_tests.log
grep -E "fused chain|passed|failed|rc=|Error" $O/align_tests.log | tail -8
ALIGN_BENCH_LATE=1 timeout 300 python tools/align_bench.py > $O/bench.log 2>&1; grep -v amdgpu.ids $O/bench.log | cut -c1-330
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/al -o kt -- python $R/tools/align_profile.py > $O/al.log 2>&1
f=$(find /tmp/prof/al -name "*kernel_stats.csv" | head -1); cp "$f" $O/align_kernel_stats.csv; python - <<PY
import csv
rows=list(csv.DictReader(open("$O/align_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per iteration %.3f ms, %.1f launches per iteration" % (tot/1e6/52, sum(int(r['Calls']) for r in rows)/52))
for r in rows[:8]: print(int(r['Calls']), "%.2f ms" % (float(r['TotalDurationNs'])/1e6), "%.1f us" % (float(r['AverageNs'])/1e3), r['Name'][:90])
PY
