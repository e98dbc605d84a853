#!/bin/bash
# round 2 run K: bf16x3 K loop with register-prefetch pipeline vs the batched-reads build (lib B): correctness, then the GEMM census
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2k; mkdir -p $O
( time timeout 300 python -m pytest tests/test_bf16x3_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "x3 or conv or linear or gemm" ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -6 $O/tests.log | cut -c1-200
timeout 200 python tools/gemm_bench.py --dtype bf16x3 --iters 10 > $O/census_pipe.log 2>&1; tail -1 $O/census_pipe.log
GEO4D_HIP_LIB=$R/geo4d_amd/csrc/libgeo4d_hip_b.so timeout 200 python tools/gemm_bench.py --dtype bf16x3 --iters 10 > $O/census_b.log 2>&1; tail -1 $O/census_b.log
paste <(cut -c1-78 $O/census_pipe.log) <(cut -c50-78 $O/census_b.log) | head -40
