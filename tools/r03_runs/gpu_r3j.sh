#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3j; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" $O/pytest.log | tail -6
timeout 600 python bench.py --height 576 --width 1024 --batch 4 --dtype f16 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-250 $O/bench_cfg4.json
timeout 400 python bench.py --height 256 --width 576 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; cut -c1-250 $O/bench_cfg2.json
