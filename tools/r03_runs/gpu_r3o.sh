#!/bin/bash
# ablation of the phased K loop (make ABLATION=1 build): where a phase's time goes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3o; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 300 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 20 --filter "L0 conv3x3 320->320" --tiles 23,72,81,82,83,84,85,86,87 > $O/abl.log 2>&1
timeout 300 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 20 --filter "L0 proj" --tiles 23,72,81,82,83,84,85,86,87 >> $O/abl.log 2>&1
timeout 300 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 10 --filter "VAE conv3x3 512 @80x128" --tiles 22,71,91,92,93,94,95,96,97 >> $O/abl.log 2>&1
timeout 300 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 20 --filter "L2 geglu" --tiles 22,71,91,92,93,94,95,96,97 >> $O/abl.log 2>&1
grep -v "amdgpu.ids\|^shape\|census" $O/abl.log | cut -c1-220
