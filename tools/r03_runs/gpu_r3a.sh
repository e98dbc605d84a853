#!/bin/bash
# round 3, call A: parity of the second-generation GEMM tiles, the new full-size pins, per-shape A/B of v1 vs v2 tiles
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3a; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 700 python -m pytest tests/test_gemm_v2_gpu.py -q -x ) > $O/v2_tests.log 2>&1; echo "rc=$?" >> $O/v2_tests.log
tail -5 $O/v2_tests.log
timeout 300 python tools/gemm_bench.py --dtype bf16x3 --iters 5 --explore2 > $O/explore_x3.log 2>&1; tail -4 $O/explore_x3.log
timeout 300 python tools/gemm_bench.py --dtype bf16 --iters 5 --explore2 > $O/explore_bf16.log 2>&1; tail -2 $O/explore_bf16.log
( time timeout 500 python -m pytest tests/test_fullsize_gpu.py -q -x -k "multi_step or unet_full_size" -s ) > $O/fullsize.log 2>&1; echo "rc=$?" >> $O/fullsize.log
grep -E "3-step|passed|failed|rc=" $O/fullsize.log | tail -5
( time timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_bf16x3_gpu.py -q -k "attention_self" ) > $O/attn.log 2>&1; tail -3 $O/attn.log
