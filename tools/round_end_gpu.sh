#!/bin/bash
# Everything the round-end evidence needs, in ONE gpurun call: gpu tests, smoke, bench line, kernel trace of the bench
# command, three PMC passes over one U-Net forward. Raw profiler output stays in /tmp; summaries land in gpurun_out/final/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 500 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
export TMPDIR=/tmp
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
DB=$(find /tmp/prof/kt -name "*.db" | head -1)
python $R/tools/prof_summary.py "$DB" 40 > $O/kernel_trace.md 2>> $O/kt.log
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof/pmc$i -o p -- python $R/tools/profile_unet.py 1 > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc.md $O/pmc.json $(find /tmp/prof/pmc1 /tmp/prof/pmc2 /tmp/prof/pmc3 -name "*counter_collection.csv") > $O/pmc_summary.log 2>&1
tail -3 $O/pytest.log $O/smoke.log; cut -c1-300 $O/bench.json; tail -3 $O/pmc_summary.log
