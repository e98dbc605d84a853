#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3c; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 700 python -m pytest tests/test_gemm_v2_gpu.py -q -x ) > $O/v2_tests.log 2>&1; echo "rc=$?" >> $O/v2_tests.log
tail -5 $O/v2_tests.log
timeout 400 python tools/gemm_bench.py --dtype bf16x3 --iters 5 --explore2 > $O/explore_x3.log 2>&1; grep -v "^L\|^VAE\|^shape" $O/explore_x3.log | cut -c1-330
