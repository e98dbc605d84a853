#!/bin/bash
# round 2 run I: fused-GN finalize v2 A/B, BASELINE configs[4] and configs[2] bench lines
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2i; mkdir -p $O
( time timeout 200 python -m pytest "tests/test_kernels_gpu.py::test_groupnorm_statistics_from_the_gemm_epilogue" "tests/test_kernels_gpu.py::test_groupnorm" -m gpu -q ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -5 $O/tests.log | cut -c1-200
line() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], (d.get('fast_mode') or {}).get('value'), d['roofline']['frac'])"; }
timeout 420 python bench.py --height 576 --width 1024 --batch 4 --dtype f16 --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/bench_cfg4_f16.json 2> $O/bench_cfg4.err; echo rc=$?; tail -3 $O/bench_cfg4.err | cut -c1-300; line $O/bench_cfg4_f16.json cfg4
timeout 200 python bench.py --height 256 --width 576 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_cfg2_x3.json 2> $O/bench_cfg2.err; echo rc=$?; line $O/bench_cfg2_x3.json cfg2
GEO4D_GN_FUSED=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err; line $O/bench_fused.json fused
GEO4D_GN_FUSED=0 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err; line $O/bench_plain.json plain
