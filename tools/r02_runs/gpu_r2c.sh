#!/bin/bash
# Round-2 GPU call C: the FULL gpu suite (what the driver runs at round end), smoke(), kernel trace of the bench command and
# three PMC passes over one bf16x3 U-Net forward. Summaries land in gpurun_out/r2c/.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
export TMPDIR=/tmp
O=$R/gpurun_out/r2c; mkdir -p $O
( time timeout 840 python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^FAILED|^ERROR|real|s call|s setup" $O/pytest.log | tail -40 | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -4 $O/smoke.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/kt.log 2>&1
cp /tmp/prof/kt/kt_kernel_stats.csv $O/ 2>/dev/null || find /tmp/prof/kt -name "*kernel_stats.csv" -exec cp {} $O/kt_kernel_stats.csv \;
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof/pmc$i -o p -- python $R/tools/profile_unet.py 1 bf16x3 > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc.md $O/pmc.json $(find /tmp/prof/pmc1 /tmp/prof/pmc2 /tmp/prof/pmc3 -name "*counter_collection.csv") > $O/pmc_summary.log 2>&1
tail -4 $O/pmc_summary.log; tail -2 $O/kt.log | cut -c1-300
