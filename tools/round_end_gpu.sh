#!/bin/bash
# Round-end evidence in ONE gpurun call (round 6 form, ~16 min): the whole GPU suite (-s: the parity numbers land in the log), smoke(), the
# DEFAULT bench line (what the driver runs, with fewer steps), a same-box A/B against the round-5 end state (gpurun_ab_r5/ = `git archive
# 1f11639`, built in the container), a kernel trace of the bench split into phases, three PMC passes over one U-Net forward of the headline
# mode, the GEMM census (three-pass, two-pass, vendor single-pass bf16) and the attention bench. Raw profiler output stays in /tmp;
# summaries land in gpurun_out/final/ (copy what should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
MODE=${1:-bf16x3m}
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -s --durations=10 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" $O/pytest.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -5 $O/smoke.log
( time timeout 900 python bench.py --steps 5 --warmup 2 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-400 $O/bench.json
# same-box A/B against the round-5 end state (gpurun_ab_r5/ = `git archive 1f11639`, library built in the container: it travels with the snapshot)
if [ -f $R/gpurun_ab_r5/geo4d_amd/csrc/libgeo4d_hip.so ]; then
  for i in 1 2; do
    cd $R/gpurun_ab_r5 && timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/ab_r5_$i.json 2> $O/ab_r5_$i.err
    cd $R && timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/ab_r6_$i.json 2> $O/ab_r6_$i.err
  done
  cd $R && python - <<PY
import json
for n in ("ab_r5_1", "ab_r6_1", "ab_r5_2", "ab_r6_2"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % n) if l.startswith("{")][-1]); print(n, d["dtype"], round(d["value"], 3), "frames/s", {k: round(v) for k, v in d["split_ms_per_step"].items()}, "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
    except Exception as e:
        print(n, "failed", e)
PY
fi
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --dtype $MODE --no-cpu-baseline --no-fast-mode --no-strict-mode --no-shipped-setting --no-clip-leg --no-batched-windows > $O/kt.log 2>&1
find /tmp/prof/kt -name "*kernel_stats.csv" -exec cp {} $O/kt_kernel_stats.csv \;
KT=$(find /tmp/prof/kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python $R/tools/prof_phases.py $KT 30 > $O/phases.md 2>&1
head -4 $O/phases.md | cut -c1-400
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof/pmc$i -o p -- python $R/tools/profile_unet.py 1 $MODE > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc.md $O/pmc.json $(find /tmp/prof/pmc1 /tmp/prof/pmc2 /tmp/prof/pmc3 -name "*counter_collection.csv") > $O/pmc_summary.log 2>&1
tail -12 $O/pmc_summary.log
cd $R
timeout 300 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 10 --vendor > $O/gemm_x3.log 2>&1; tail -2 $O/gemm_x3.log
timeout 300 python tools/gemm_bench.py --dtype f16x2 --iters 10 > $O/gemm_x2.log 2>&1; tail -2 $O/gemm_x2.log
timeout 300 python tools/gemm_bench.py --dtype bf16 --iters 10 --vendor > $O/gemm_bf16_vendor.log 2>&1; tail -2 $O/gemm_bf16_vendor.log
timeout 200 python tools/attn_bench.py bf16 f16 bf16x3 > $O/attn.log 2>&1; grep "2560\|forward" $O/attn.log | grep "v4\|v3 \|v1 " | tail -8
