#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3d; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_presplit_gpu.py -q -x -s ) > $O/presplit.log 2>&1; echo "rc=$?" >> $O/presplit.log
grep -E "passed|failed|rc=|Error|error|presplit vs" $O/presplit.log | tail -8
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bf16x3_gpu.py tests/test_gemm_v2_gpu.py tests/test_parity_gpu.py -q -x ) > $O/kernels.log 2>&1; echo "rc=$?" >> $O/kernels.log
tail -4 $O/kernels.log
for ps in 1 0; do
  GEO4D_X3_PRESPLIT=$ps timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/bench_ps$ps.json 2> $O/bench_ps$ps.err; cut -c1-400 $O/bench_ps$ps.json; tail -2 $O/bench_ps$ps.err
done
GEO4D_GN_ONE_LAUNCH=0 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/bench_gn3.json 2> $O/bench_gn3.err; cut -c1-300 $O/bench_gn3.json
