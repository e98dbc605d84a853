R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-shipped-setting --no-clip-leg --no-batched-windows > $O/kt.log 2>&1
find /tmp/prof/kt -name "*kernel_stats.csv" -exec cp {} $O/kt_kernel_stats.csv \;
KT=$(find /tmp/prof/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/prof_phases.py $KT 40 > $O/phases.md 2>&1
head -16 $O/phases.md | cut -c1-200
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof/pmc$i -o p -- python $R/tools/profile_unet.py 1 bf16x3m > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc.md $O/pmc.json $(find /tmp/prof/pmc1 /tmp/prof/pmc2 /tmp/prof/pmc3 -name "*counter_collection.csv") > $O/pmc_summary.log 2>&1
tail -14 $O/pmc_summary.log
