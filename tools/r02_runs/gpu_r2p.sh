#!/bin/bash
# round 2 run P: epilogue with prefetched residual chunks: kernel tests, then the census in both modes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2p; mkdir -p $O
( timeout 300 python -m pytest tests/test_bf16x3_gpu.py tests/test_kernels_gpu.py -m gpu -q -x ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log | cut -c1-200
timeout 200 python tools/gemm_bench.py --dtype bf16x3 --iters 10 > $O/census_x3.log 2>&1; tail -1 $O/census_x3.log
timeout 200 python tools/gemm_bench.py --dtype bf16 --iters 10 > $O/census_bf16.log 2>&1; tail -1 $O/census_bf16.log
cut -c1-100 $O/census_x3.log | head -30
