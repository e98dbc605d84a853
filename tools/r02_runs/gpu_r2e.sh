#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2e; mkdir -p $O
timeout 200 python -m pytest tests/test_align_gpu.py -m gpu -q -s > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "^\[|passed|failed|rc=|^E  " $O/tests.log | tail | cut -c1-250
timeout 200 python tools/align_bench.py 128 320 512 > $O/align_bench.log 2>&1; grep align $O/align_bench.log
timeout 100 python tools/align_bench.py 64 256 576 >> $O/align_bench.log 2>&1; tail -3 $O/align_bench.log
for dt in bf16 bf16x3; do for ab in 0 2 3; do
  timeout 120 python tools/gemm_bench.py --dtype $dt --iters 10 --ablate $ab > $O/gemm_${dt}_ab$ab.log 2>&1; echo "$dt ablate=$ab: $(tail -1 $O/gemm_${dt}_ab$ab.log)"
done; done
