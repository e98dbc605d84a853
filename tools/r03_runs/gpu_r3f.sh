#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_presplit_gpu.py -q -x -s ) > $O/presplit.log 2>&1; echo "rc=$?" >> $O/presplit.log
grep -E "passed|failed|rc=|presplit vs" $O/presplit.log | tail -5
( time GEO4D_AUTOTUNE=1 timeout 900 python tools/tune_gemm.py $O/gfx950.json bf16x3 ) > $O/tune.log 2>&1; tail -3 $O/tune.log
cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
for ps in 1 0; do
  GEO4D_X3_PRESPLIT=$ps timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/bench_ps$ps.json 2> $O/bench_ps$ps.err; cut -c1-200 $O/bench_ps$ps.json; python -c "
import json;d=json.load(open('$O/bench_ps$ps.json'));print(d['split_ms_per_step'], d['roofline']['ms_per_unet_forward'], d['roofline']['frac'])"
done
