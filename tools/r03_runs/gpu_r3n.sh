#!/bin/bash
# third-generation conv_gemm: parity tests, then per-shape timing against the committed table (pre-split bf16x3 operands, as the networks run)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3n; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gemm_v3_gpu.py -q -x ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|Error|error" $O/tests.log | tail -6
timeout 600 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 10 --tiles 71,72,73,74,71/2,72/2,73/2,74/2,73/4,74/4,74/8 > $O/bench_x3.log 2>&1; tail -64 $O/bench_x3.log | cut -c1-250
timeout 300 python tools/gemm_bench.py --dtype bf16 --iters 10 --tiles 71,72,73,74,73/2,74/2,74/4 > $O/bench_bf16.log 2>&1; tail -3 $O/bench_bf16.log | cut -c1-250
