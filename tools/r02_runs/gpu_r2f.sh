#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2f; mkdir -p $O
timeout 200 python -m pytest tests/test_align_gpu.py tests/test_bf16x3_gpu.py -m gpu -q -s > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "^\[align|passed|failed|rc=|^E  " $O/tests.log | tail | cut -c1-250
timeout 200 python tools/align_bench.py 128 320 512 > $O/align_bench.log 2>&1; grep "align " $O/align_bench.log
timeout 100 python tools/align_bench.py 64 256 576 >> $O/align_bench.log 2>&1; tail -3 $O/align_bench.log
timeout 120 python tools/gemm_bench.py --dtype bf16x3 --iters 10 > $O/gemm_x3.log 2>&1; tail -1 $O/gemm_x3.log
timeout 120 python tools/gemm_bench.py --dtype bf16 --iters 10 > $O/gemm_bf16.log 2>&1; tail -1 $O/gemm_bf16.log
( time timeout 400 python bench.py --steps 3 --warmup 1 ) > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json; tail -4 $O/bench.err
