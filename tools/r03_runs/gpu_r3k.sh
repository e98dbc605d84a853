#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3k; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_align_gpu.py -q -x ) > $O/align_tests.log 2>&1; echo "rc=$?" >> $O/align_tests.log; tail -4 $O/align_tests.log
ALIGN_BENCH_LATE=0 GEO4D_ALIGN_KERNEL=1 timeout 300 python tools/align_bench.py > $O/bench_v1.log 2>&1; grep "^align" $O/bench_v1.log | cut -c1-330
ALIGN_BENCH_LATE=1 timeout 300 python tools/align_bench.py > $O/bench_v2.log 2>&1; grep -v amdgpu.ids $O/bench_v2.log | cut -c1-330
