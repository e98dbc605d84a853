#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2h; mkdir -p $O
( time timeout 400 python -m pytest "tests/test_kernels_gpu.py::test_groupnorm_statistics_from_the_gemm_epilogue" "tests/test_kernels_gpu.py::test_groupnorm" \
   tests/test_parity_gpu.py tests/test_fullsize_gpu.py -m gpu -q -s ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -E "fused|window e2e|full-size|drift|passed|failed|rc=|^E  |^FAILED" $O/tests.log | tail -30 | cut -c1-260
GEO4D_GN_FUSED=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err; python -c "
import json;d=json.loads(open('$O/bench_fused.json').read().strip().splitlines()[-1]);print('fused', d['value'], d['ms_per_step'], d['fast_mode']['value'])"
GEO4D_GN_FUSED=0 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err; python -c "
import json;d=json.loads(open('$O/bench_plain.json').read().strip().splitlines()[-1]);print('plain', d['value'], d['ms_per_step'], d['fast_mode']['value'])"
