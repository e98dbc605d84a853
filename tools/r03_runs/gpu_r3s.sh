#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3s; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 60 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 20 --filter "L0 conv3x3 320->320" --tiles 23,72,83,81,82,85,87 > $O/abl.log 2>&1
timeout 60 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 20 --filter "L0 ffout" --tiles 23,72,83 >> $O/abl.log 2>&1
timeout 60 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 10 --filter "VAE conv3x3 512 @80x128" --tiles 22,71,93,91,92 >> $O/abl.log 2>&1
timeout 60 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 20 --filter "L1 geglu" --tiles 22,71,93 >> $O/abl.log 2>&1
grep -v "amdgpu.ids\|^shape\|census" $O/abl.log | cut -c1-220
