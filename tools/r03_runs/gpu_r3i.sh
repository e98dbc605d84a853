#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3i; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_presplit_gpu.py tests/test_kernels_gpu.py -q -x -k "groupnorm" ) > $O/gn.log 2>&1; echo "rc=$?" >> $O/gn.log; tail -3 $O/gn.log
for tw in 1 0; do
  GEO4D_GN_TWO_LAUNCH=$tw timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/bench_tw$tw.json 2> $O/bench_tw$tw.err; python -c "
import json;d=json.load(open('$O/bench_tw$tw.json'));print('two_launch=$tw', d['value'], d['split_ms_per_step'])"
done
