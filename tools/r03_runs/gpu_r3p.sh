#!/bin/bash
# phased K loop after the restructure: parity, per-shape timing, ablation (make ABLATION=1 build)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3p; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gemm_v3_gpu.py -q -x ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|Error|error" $O/tests.log | tail -6
timeout 300 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 20 --filter "L0 conv3x3 320->320" --tiles 23,72,81,82,83,84,85,86,87 > $O/abl.log 2>&1
timeout 300 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 20 --filter "L0 proj" --tiles 23,72,81,82,83,84,85,86,87 >> $O/abl.log 2>&1
timeout 300 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 10 --filter "VAE conv3x3 512 @80x128" --tiles 22,71,91,92,93,94,95,96,97 >> $O/abl.log 2>&1
timeout 300 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 20 --filter "L2 geglu" --tiles 22,71,91,92,93,94,95,96,97 >> $O/abl.log 2>&1
grep -v "amdgpu.ids\|^shape\|census" $O/abl.log | cut -c1-220
timeout 600 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 10 --tiles 71,72,73,74,71/2,72/2,73/2,74/2,73/4,74/4,74/8 > $O/bench_x3.log 2>&1; grep -v "amdgpu.ids" $O/bench_x3.log | cut -c1-250
