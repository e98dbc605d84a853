#!/bin/bash
# re-tune the bf16x3 table with the third-generation tiles among the candidates, then check the gemm test files on the new library
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3t; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 500 python tools/tune_gemm.py $O/gfx950.json bf16x3 ) > $O/tune.log 2>&1; tail -4 $O/tune.log
( timeout 400 python -m pytest tests/test_gemm_v3_gpu.py tests/test_gemm_v2_gpu.py tests/test_presplit_gpu.py -q -x ) > $O/tests.log 2>&1; tail -2 $O/tests.log
