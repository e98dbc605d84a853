#!/bin/bash
# round 2 run Q: final state: the whole GPU suite, smoke(), the default bench line (CPU baseline left to the driver's run)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2q; mkdir -p $O
( time timeout 420 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^FAILED|^ERROR|real" $O/pytest.log | tail -8 | cut -c1-200
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log | cut -c1-300
timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
