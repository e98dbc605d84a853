#!/bin/bash
# round 2 run J: alignment late terms (inverse depth + trajectory): tests + bench
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2j; mkdir -p $O
( time timeout 300 python -m pytest tests/test_align_gpu.py -m gpu -q -s -x ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -E "^\[|passed|failed|rc=|^E  |^FAILED|Error|median|   " $O/tests.log | tail -60 | cut -c1-330
timeout 200 python tools/align_bench.py 128 320 512 > $O/align_bench.log 2>&1; echo "bench rc=$?"; tail -8 $O/align_bench.log | cut -c1-400
