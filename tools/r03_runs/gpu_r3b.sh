#!/bin/bash
# round 3, call B: where does a v2 workgroup's time go? ablation builds (make ABLATION=1) of tiles 23 / 22 / 25 on census shapes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3b; mkdir -p $O; cd $R; export TMPDIR=/tmp
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.2; done ) > $O/smi.log 2>&1 &
SMI=$!
for f in "L0 conv3x3 320->320" "L0 proj" "L0 ffout" "L0 conv3d"; do
  timeout 120 python tools/gemm_bench.py --dtype bf16x3 --iters 20 --filter "$f" --tiles 23,41,42,43,44 2>&1 | grep -v amdgpu.ids
done > $O/abl_t23.log
for f in "VAE conv3x3 512 @80x128" "L2 geglu" "L0 geglu"; do
  timeout 120 python tools/gemm_bench.py --dtype bf16x3 --iters 10 --filter "$f" --tiles 22,51,52,53,54 2>&1 | grep -v amdgpu.ids
done > $O/abl_t22.log
for f in "L1 conv3d" "L1 proj" "L1 ffout" "L1 conv3x3 640"; do
  timeout 120 python tools/gemm_bench.py --dtype bf16x3 --iters 20 --filter "$f" --tiles 25,61,62,63,64 2>&1 | grep -v amdgpu.ids
done > $O/abl_t25.log
kill $SMI
cat $O/abl_t23.log $O/abl_t22.log $O/abl_t25.log
sort $O/smi.log | uniq -c | sort -rn | head -8
