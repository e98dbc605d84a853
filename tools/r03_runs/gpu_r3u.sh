#!/bin/bash
# random vs zero-filled operands, same launches: the data-dependent (power / clock) share of the K loop's time
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3u; mkdir -p $O; cd $R; export TMPDIR=/tmp
for z in "" "--zeros"; do
  echo "== operands: ${z:-random}" >> $O/zeros.log
  timeout 60 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 30 $z --filter "L0 conv3x3 320->320" --tiles 23,72,16 >> $O/zeros.log 2>&1
  timeout 60 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 30 $z --filter "L1 geglu" --tiles 22,71 >> $O/zeros.log 2>&1
  timeout 60 python tools/gemm_bench.py --dtype bf16 --iters 30 $z --filter "L0 conv3x3 320->320" --tiles 23,72,16 >> $O/zeros.log 2>&1
done
grep -v "amdgpu.ids\|^shape\|census" $O/zeros.log | cut -c1-200
