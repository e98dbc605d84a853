#!/bin/bash
# One entry point for the GPU-box jobs of a round: `gpurun --timeout N -- 'bash tools/gpu_jobs.sh <job> [args]'`.
# Every job writes under gpurun_out/<tag>/ (merged back by gpurun); summaries worth judging are copied into profiles/ by hand.
# (Rounds 2-3 kept one script per call under tools/r0*_runs/: those are in the history; this file replaces the pattern.)
# Round 5's calls, in order: r5a (two-pass f16 GEMM: first measurement) r5b (census of every row in two passes) r5c (classes widened)
# r5d (stream-rounding classes: measured, later removed) r5e / r5f (suite dry runs) r5g (phases of a step) r5h (one-launch GroupNorm: slower)
# r5i (A64) r5j (320x160 tile: no gain) r5k (same-box A/B of A64) r5l / r5m (grouped tile order) r5n (the other BASELINE configs)
# r5x (attention staging) r5y (same-box A/B vs round 4) r5z / r5final (HEAD as the driver runs it + its trace / PMC); tools/round_end_gpu.sh
# is the all-in-one evidence job. A job that names build_b/ or gpurun_ab_r4/ needs that directory prepared in the container first.
R=${GRAFT_REPO_ROOT:-/root/repo}
JOB=${1:?job name}; shift
TAG=${TAG:-r6_$JOB}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
show() { grep -v "amdgpu.ids" "$1" | cut -c1-${2:-260}; }

case $JOB in
  census)      # per-shape GEMM census: every tuner candidate per shape vs the committed table, vendor calibration next to it
    MODE=${1:-bf16x3}
    if [ "$MODE" = bf16x3 ]; then EXTRA="--presplit"; else EXTRA=""; fi
    timeout 900 python tools/gemm_bench.py --dtype $MODE $EXTRA --iters 10 --explore-all --vendor > $O/census_$MODE.log 2>&1
    show $O/census_$MODE.log | grep -v "^    table" | tail -70
    ;;
  retune)      # re-measure the tuning table with the two-stage tuner (modes as arguments), keep the result for the next steps
    timeout 1500 python tools/tune_gemm.py $O/gfx950.json "$@" > $O/tune.log 2>&1; show $O/tune.log | tail -5
    ;;
  bench)       # bench line (arguments passed through)
    timeout 900 python bench.py "$@" > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-1500 $O/bench.json
    ;;
  attn)
    timeout 600 python tools/attn_bench.py "$@" > $O/attn.log 2>&1; show $O/attn.log
    ;;
  r6a)         # first call of round 6: plain f16 activation rows for the two-pass GEMMs (ABI 8) - correct? same smoke bits as round 5 (5.298e-4)? faster?
    ( time timeout 900 python -m pytest tests/test_f16x2_gpu.py tests/test_presplit_gpu.py tests/test_gemm_v3_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest_x2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_x2.log
    grep -E "passed|failed|rc=|Error|error|assert" $O/pytest_x2.log | tail -12
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -5 $O/smoke.log
    ( time timeout 1200 python -m pytest tests/test_fullsize_gpu.py tests/test_parity_gpu.py -m gpu -q -x -s ) > $O/pytest_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_full.log
    grep -E "^\[|passed|failed|rc=|Error|assert" $O/pytest_full.log | cut -c1-300 | tail -14
    for i in 1 2; do
      timeout 400 python bench.py --steps 3 --warmup 1 --dtype bf16x3m --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_$i.json 2> $O/bench_$i.err
      python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/bench_$i.json") if l.startswith("{")][-1]); r = d["roofline"]
    print("run $i:", round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "issued", round(r["frac_issued"], 3), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print("run $i failed", e)
PY
    done
    ;;
  r6b)         # class "attn" (f16 self-attention chain) + bf16x3m in every MODES list: parity everywhere? same-box three-way: round-5 tree / plain f16 rows / + attn
    ( time timeout 900 python -m pytest tests/test_f16x2_gpu.py tests/test_host_logic.py -m gpu -q -x --durations=5 ) > $O/pytest_x2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_x2.log
    grep -E "passed|failed|rc=|Error|error|assert" $O/pytest_x2.log | tail -8
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -5 $O/smoke.log
    ( time timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_frontend_gpu.py tests/test_sizes_gpu.py tests/test_fullsize_gpu.py -m gpu -q -s -k "not 50_step_window" ) > $O/pytest_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_full.log
    grep -E "bf16x3m|passed|failed|rc=|Error|^FAILED" $O/pytest_full.log | cut -c1-420 | tail -40
    bl() { python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "issued", round(r["frac_issued"], 3), "attn", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.get("attention", {}).items()}, "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
    }
    for i in 1 2; do
      ( cd gpurun_ab_r5 && timeout 400 python bench.py --steps 3 --warmup 1 --dtype bf16x3m --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_r5_$i.json 2> $O/bench_r5_$i.err ); bl $O/bench_r5_$i.json "r5 tree run $i:"
      GEO4D_TWO_PASS=conv3x3,vae3x3,tconv,ln,ff timeout 400 python bench.py --steps 3 --warmup 1 --dtype bf16x3m --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_noattn_$i.json 2> $O/bench_noattn_$i.err; bl $O/bench_noattn_$i.json "r6 f16 rows, no attn class run $i:"
      timeout 400 python bench.py --steps 3 --warmup 1 --dtype bf16x3m --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_attn_$i.json 2> $O/bench_attn_$i.err; bl $O/bench_attn_$i.json "r6 + attn class run $i:"
    done
    ;;
  r6c)         # split-K without a reduce launch (wave tickets): bit-equal to the reduce kernel? faster? (same-box A/B by environment switch)
    ( time timeout 1200 python -m pytest tests/test_splitk_fused_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py tests/test_f16x2_gpu.py tests/test_bf16x3_gpu.py tests/test_presplit_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gemm.log
    grep -E "passed|failed|rc=|Error|error|assert" $O/pytest_gemm.log | tail -8
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -5 $O/smoke.log
    bl() { python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "frac", round(r["frac"], 4), "issued", round(r["frac_issued"], 3), "attn ms", round(r["attention"]["ms_per_unet_forward"], 3), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
    }
    for i in 1 2; do
      GEO4D_SPLITK_FUSED=0 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_reduce_$i.json 2> $O/bench_reduce_$i.err; bl $O/bench_reduce_$i.json "reduce kernel run $i:"
      timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_fused_$i.json 2> $O/bench_fused_$i.err; bl $O/bench_fused_$i.json "fused split-K run $i:"
    done
    cd /tmp
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-shipped-setting --no-clip-leg > $O/kt.log 2>&1
    find /tmp/prof/kt -name "*kernel_stats.csv" -exec cp {} $O/kt_kernel_stats.csv \;
    KT=$(find /tmp/prof/kt -name "*kernel_trace.csv" | head -1)
    [ -n "$KT" ] && python $R/tools/prof_phases.py $KT 40 > $O/phases.md 2>&1
    head -60 $O/phases.md | cut -c1-200
    ;;
  r6d)         # split-K last arrival with sc1 write-through slabs instead of fences: correct (incl. 10x repeat at full chip)? A/B
    ( time timeout 600 python -m pytest tests/test_splitk_fused_gpu.py -m gpu -q -x ) > $O/pytest_splitk.log 2>&1; echo "pytest rc=$?" >> $O/pytest_splitk.log
    grep -E "passed|failed|rc=|Error|error|assert" $O/pytest_splitk.log | tail -8
    bl() { python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "frac", round(r["frac"], 4), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
    }
    for i in 1 2; do
      GEO4D_SPLITK_FUSED=0 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_reduce_$i.json 2> $O/bench_reduce_$i.err; bl $O/bench_reduce_$i.json "reduce kernel run $i:"
      timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_fused_$i.json 2> $O/bench_fused_$i.err; bl $O/bench_fused_$i.json "fused split-K run $i:"
    done
    ;;
  r6e)         # classes cattn / tattn (cross- and temporal-attention chains on f16 rows): parity + frames/s per class, same box; the 50-step reference fixture
    BASE=conv3x3,vae3x3,tconv,ln,ff,attn
    bl() { python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "frac", round(r["frac"], 4), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
    }
    for cls in $BASE $BASE,cattn $BASE,tattn $BASE,cattn,tattn; do
      GEO4D_TWO_PASS=$cls timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$cls.log 2>&1; echo "$cls: $(grep 'bf16x3m' $O/smoke_$cls.log)"
    done
    for i in 1 2; do
      for cls in $BASE $BASE,cattn $BASE,tattn $BASE,cattn,tattn; do
        GEO4D_TWO_PASS=$cls timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_${cls}_$i.json 2> $O/bench_${cls}_$i.err; bl $O/bench_${cls}_$i.json "$cls run $i:"
      done
    done
    ( time GEO4D_TWO_PASS=$BASE,cattn,tattn timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_parity_gpu.py -m gpu -q -s -k "full_size or window_end_to_end or ddim_sampler" ) > $O/pytest_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_full.log
    grep -E "bf16x3m|passed|failed|rc=|Error|^FAILED" $O/pytest_full.log | cut -c1-520 | tail -20
    ;;
  r6f)         # class vaeup (two-pass Upsample convs on an f16 copy of the decoder stream): decoder-alone parity vs the reference, decode ms, frames/s
    BASE=conv3x3,vae3x3,tconv,ln,ff,attn,cattn,tattn
    bl() { python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
    }
    for cls in $BASE $BASE,vaeup; do
      ( GEO4D_TWO_PASS=$cls timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_parity_gpu.py tests/test_sizes_gpu.py -m gpu -q -s -k "vae_decode or window_end_to_end or 50_step_window" ) > $O/pytest_$cls.log 2>&1; echo "pytest rc=$?" >> $O/pytest_$cls.log
      echo "== $cls"; grep -E "bf16x3m|passed|failed|rc=|Error|^FAILED" $O/pytest_$cls.log | cut -c1-420 | tail -8
    done
    for i in 1 2; do
      for cls in $BASE $BASE,vaeup; do
        GEO4D_TWO_PASS=$cls timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_${cls}_$i.json 2> $O/bench_${cls}_$i.err; bl $O/bench_${cls}_$i.json "$cls run $i:"
      done
    done
    ;;
  r6g)         # re-measure the tuning table for the two-pass launches (their A rows are 2-byte now) + the new f16-row shapes; A/B old vs new table
    bl() { python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
    }
    cp geo4d_amd/tuning/gfx950.json $O/gfx950_before.json
    timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_old_1.json 2> $O/bench_old_1.err; bl $O/bench_old_1.json "old table run 1:"
    timeout 1500 python tools/tune_gemm.py $O/gfx950.json --keep --drop-prefix=4/ bf16x3m > $O/tune.log 2>&1; show $O/tune.log | tail -3
    if [ -s $O/gfx950.json ]; then
      cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
      timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_new_1.json 2> $O/bench_new_1.err; bl $O/bench_new_1.json "new table run 1:"
      cp $O/gfx950_before.json geo4d_amd/tuning/gfx950.json
      timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_old_2.json 2> $O/bench_old_2.err; bl $O/bench_old_2.json "old table run 2:"
      cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
      timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_new_2.json 2> $O/bench_new_2.err; bl $O/bench_new_2.json "new table run 2:"
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -4 $O/smoke.log
    fi
    ;;
  r6h)         # split-K reduce emitting the GroupNorm column sums: tests, A/B by environment switch
    ( time timeout 900 python -m pytest tests/test_f16x2_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py tests/test_kernels_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error|assert" $O/pytest.log | tail -8
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -4 $O/smoke.log
    bl() { python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "frac", round(r["frac"], 4), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
    }
    for i in 1 2; do
      GEO4D_SPLITK_COLSUM=0 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg --no-batched-windows > $O/bench_off_$i.json 2> $O/bench_off_$i.err; bl $O/bench_off_$i.json "split-K colsum off run $i:"
      timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg --no-batched-windows > $O/bench_on_$i.json 2> $O/bench_on_$i.err; bl $O/bench_on_$i.json "split-K colsum on run $i:"
    done
    ;;
  r6i)         # the f16-row epilogue on every two-pass tile (160x320 for N = 640 / 960 / 1920 ...): tests, re-tune the "o" launches, A/B old vs new table
    ( time timeout 600 python -m pytest tests/test_f16x2_gpu.py tests/test_gemm_v3_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error|assert" $O/pytest.log | tail -6
    bl() { python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "frac", round(r["frac"], 4))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
    }
    cp geo4d_amd/tuning/gfx950.json $O/gfx950_before.json
    timeout 1200 python tools/tune_gemm.py $O/gfx950.json --keep --drop-suffix=x11o --batch=1 --batch=2 bf16x3m > $O/tune.log 2>&1; show $O/tune.log | tail -3
    grep -c "" $O/gfx950.json.log; grep "x11o" $O/gfx950.json.log | cut -c1-230 | head -40
    if [ -s $O/gfx950.json ]; then
      for i in 1 2; do
        cp $O/gfx950_before.json geo4d_amd/tuning/gfx950.json
        timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg --no-batched-windows --no-shipped-setting > $O/bench_old_$i.json 2> $O/bench_old_$i.err; bl $O/bench_old_$i.json "old table run $i:"
        cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
        timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg --no-batched-windows --no-shipped-setting > $O/bench_new_$i.json 2> $O/bench_new_$i.err; bl $O/bench_new_$i.json "new table run $i:"
      done
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
    fi
    ;;
  r6j)         # pre-split pass in front of the Upsample convolutions (U-Net + VAE): bit-identical? faster? (A/B by environment switch)
    ( time timeout 600 python -m pytest tests/test_presplit_gpu.py tests/test_f16x2_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error|assert" $O/pytest.log | tail -6
    bl() { python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "frac", round(r["frac"], 4), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
    }
    for i in 1 2; do
      GEO4D_X3_PRESPLIT_UP=0 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg --no-batched-windows > $O/bench_off_$i.json 2> $O/bench_off_$i.err; bl $O/bench_off_$i.json "presplit-up off run $i:"
      timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg --no-batched-windows > $O/bench_on_$i.json 2> $O/bench_on_$i.err; bl $O/bench_on_$i.json "presplit-up on run $i:"
    done
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
    ;;
  r6k)         # own table entries for the pre-split bf16x3 launches that borrowed the raw-activation entry (incl. the pre-split Upsample convs); A/B
    bl() { python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "frac", round(r["frac"], 4), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
    }
    cp geo4d_amd/tuning/gfx950.json $O/gfx950_before.json
    timeout 1500 python tools/tune_gemm.py $O/gfx950.json --keep --exact --batch=1 --batch=2 bf16x3m > $O/tune.log 2>&1; show $O/tune.log | tail -3
    grep -c "" $O/gfx950.json.log; cut -c1-200 $O/gfx950.json.log | head -50
    if [ -s $O/gfx950.json ]; then
      for i in 1 2; do
        cp $O/gfx950_before.json geo4d_amd/tuning/gfx950.json
        timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg --no-batched-windows > $O/bench_old_$i.json 2> $O/bench_old_$i.err; bl $O/bench_old_$i.json "old table run $i:"
        cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
        timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg --no-batched-windows > $O/bench_new_$i.json 2> $O/bench_new_$i.err; bl $O/bench_new_$i.json "new table run $i:"
      done
    fi
    ;;
  r6l)         # skip concatenations written by their producers (no concat_channels): bit-identical? A/B by environment switch
    ( time timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "concatenation or unet or ddim_sampler or window_end or full_size_vs_reference" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error|assert" $O/pytest.log | tail -6
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
    bl() { python - "$1" "$2" <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
    }
    for i in 1 2; do
      GEO4D_FUSED_CONCAT=0 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg --no-batched-windows --no-shipped-setting > $O/bench_off_$i.json 2> $O/bench_off_$i.err; bl $O/bench_off_$i.json "concat copies run $i:"
      timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg --no-batched-windows --no-shipped-setting > $O/bench_on_$i.json 2> $O/bench_on_$i.err; bl $O/bench_on_$i.json "producers write the halves run $i:"
    done
    ;;
  r5a)         # first call of round 5: the two-pass f16 GEMM / bf16x3m mode - correct? how much faster per conv? accurate at size over 50 steps? end to end?
    ( time timeout 600 python -m pytest tests/test_f16x2_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest_x2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_x2.log
    grep -E "passed|failed|rc=|Error|error|assert" $O/pytest_x2.log | tail -12
    timeout 300 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 10 --filter conv3x3 > $O/census_x3_conv.log 2>&1; show $O/census_x3_conv.log | tail -16
    timeout 600 python tools/gemm_bench.py --dtype f16x2 --iters 10 --explore-all --filter conv3x3 > $O/census_x2_conv.log 2>&1; show $O/census_x2_conv.log | grep -v "^    table" | tail -32
    cp geo4d_amd/tuning/gfx950.json $O/gfx950_before.json
    timeout 900 python tools/tune_gemm.py $O/gfx950.json --keep bf16x3m > $O/tune.log 2>&1; show $O/tune.log | tail -4
    [ -s $O/gfx950.json ] && cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
    ( time timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s -k "vs_reference or 50_step or multi_step" ) > $O/pytest_fullsize.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fullsize.log
    grep -E "^\[|passed|failed|rc=|Error|assert" $O/pytest_fullsize.log | cut -c1-400 | tail -14
    for i in 1 2; do
      for m in bf16x3 bf16x3m; do
        timeout 400 python bench.py --steps 3 --warmup 1 --dtype $m --no-cpu-baseline --no-fast-mode > $O/bench_${m}_$i.json 2> $O/bench_${m}_$i.err
        python - <<PY
import json
try:
    d = json.load(open("$O/bench_${m}_$i.json")); r = d["roofline"]
    print("$m run $i:", round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "issued", round(r["frac_issued"], 3), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print("$m run $i failed", e)
PY
      done
    done
    cd /tmp
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --dtype bf16x3m --no-cpu-baseline --no-fast-mode --no-shipped-setting > $O/kt.log 2>&1
    find /tmp/prof/kt -name "*kernel_stats.csv" -exec cp {} $O/kt_kernel_stats.csv \;
    KT=$(find /tmp/prof/kt -name "*kernel_trace.csv" | head -1)
    [ -n "$KT" ] && python $R/tools/prof_phases.py $KT 16 > $O/phases_bf16x3m.md 2>&1
    head -5 $O/phases_bf16x3m.md | cut -c1-600
    ;;
  r5final)     # the round's last call: HEAD as the driver will run it (suite, smoke, default line) + the trace / PMC evidence of that very tree
    ( time timeout 1500 python -m pytest tests -m gpu -q -x -s ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=|^FAILED|^ERROR" $O/pytest.log | tail -5
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -5 $O/smoke.log
    ( time timeout 900 python bench.py --steps 5 --warmup 2 ) > $O/bench.json 2> $O/bench.err; tail -4 $O/bench.err
    python -c "import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print('default line:', d['dtype'], round(d['value'],3), d['split_ms_per_step'], 'strict', round(d['strict_mode']['value'],3), 'fast', round(d['fast_mode']['value'],3), 'shipped', round(d['shipped_setting']['value'],2), 'clip s', round(d['clip_mode']['ms_per_step']/1e3,2), 'frac', round(d['roofline']['frac'],4), 'issued', round(d['roofline']['frac_issued'],3), 'attn issued', round(d['roofline']['attention']['frac_issued'],3))"
    cd /tmp
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-shipped-setting --no-clip-leg > $O/kt.log 2>&1
    find /tmp/prof/kt -name "*kernel_stats.csv" -exec cp {} $O/kt_kernel_stats.csv \;
    KT=$(find /tmp/prof/kt -name "*kernel_trace.csv" | head -1)
    [ -n "$KT" ] && python $R/tools/prof_phases.py $KT 30 > $O/phases.md 2>&1
    head -16 $O/phases.md | cut -c1-300
    i=0
    for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof/pmc$i -o p -- python $R/tools/profile_unet.py 1 bf16x3m > $O/pmc$i.log 2>&1
    done
    python $R/tools/pmc_summary.py $O/pmc.md $O/pmc.json $(find /tmp/prof/pmc1 /tmp/prof/pmc2 /tmp/prof/pmc3 -name "*counter_collection.csv") > $O/pmc_summary.log 2>&1
    grep "whole forward" $O/pmc.md; tail -9 $O/pmc.md
    ;;
  r5x)         # flash_attn2 staged through buffer resources (fewer live registers): correct? scratch traffic down -> faster?
    ( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bf16x3_gpu.py tests/test_presplit_gpu.py tests/test_sizes_gpu.py -m gpu -q -k "attention or attn or presplit or flash or sizes" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "^FAILED|^ERROR" $O/pytest.log | head
    timeout 300 python tools/attn_bench.py bf16 bf16x3 > $O/attn.log 2>&1; grep "2560\|forward\| 640\|9216" $O/attn.log | grep "v4\|v1 " | tail -14
    ;;
  r5y)         # a second same-box A/B against the round-4 end state (gpurun_ab_r4/ built on the box), on whatever box this call gets
    cd $R/gpurun_ab_r4 && ( time make -j64 > $O/build_r4.log 2>&1 ) 2>&1 | grep real
    for i in 1 2; do
      cd $R/gpurun_ab_r4 && timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/ab_r4_$i.json 2> $O/ab_r4_$i.err
      cd $R && timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/ab_r5_$i.json 2> $O/ab_r5_$i.err
    done
    cd $R && python - <<PY
import json
for n in ("ab_r4_1", "ab_r5_1", "ab_r4_2", "ab_r5_2"):
    try:
        d = json.load(open("$O/%s.json" % n)); print(n, d["dtype"], round(d["value"], 3), "frames/s", {k: round(v) for k, v in d["split_ms_per_step"].items()}, "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
    except Exception as e:
        print(n, "failed", e)
PY
    ;;
  r5z)         # last call of the round: HEAD as the driver will run it - the whole suite, smoke, a default bench line
    ( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=|^FAILED|^ERROR" $O/pytest.log | tail -5
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -5 $O/smoke.log
    ( time timeout 900 python bench.py --steps 5 --warmup 2 ) > $O/bench.json 2> $O/bench.err; tail -4 $O/bench.err
    python -c "import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print('default line:', d['dtype'], round(d['value'],3), d['split_ms_per_step'], 'strict', round(d['strict_mode']['value'],3), 'fast', round(d['fast_mode']['value'],3), 'shipped', round(d['shipped_setting']['value'],2), 'clip s', round(d['clip_mode']['ms_per_step']/1e3,2), 'frac', round(d['roofline']['frac'],4), 'issued', round(d['roofline']['frac_issued'],3), 'attn issued', round(d['roofline']['attention']['frac_issued'],3))"
    ;;
  r5n)         # the other BASELINE configs on one GPU at HEAD (bf16x3m): configs[2] at the Sintel size, the 128-frame clip of configs[3], configs[4]'s 576x1024 batch; + one more default line (another box's headline)
    timeout 600 python bench.py --clip-frames 64 --height 256 --width 576 > $O/bench_clip64_sintel.json 2> $O/bench_clip64_sintel.err; cut -c1-900 $O/bench_clip64_sintel.json | python -c "import sys,json; d=json.loads(sys.stdin.read()+'' if False else open('$O/bench_clip64_sintel.json').read()); print('configs[2] 64x256x576:', round(d['value'],3), d['unit'], d['phase_seconds'], d['alignment_vs_scene_truth'])"
    timeout 600 python bench.py --clip-frames 128 > $O/bench_clip128.json 2> $O/bench_clip128.err; python -c "import json; d=json.load(open('$O/bench_clip128.json')); print('configs[3]-size 128x320x512:', round(d['value'],3), d['unit'], d['phase_seconds'])"
    timeout 600 python bench.py --height 576 --width 1024 --batch 4 --dtype f16 --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-shipped-setting --no-clip-leg > $O/bench_configs4.json 2> $O/bench_configs4.err; python -c "import json; d=json.load(open('$O/bench_configs4.json')); print('configs[4] 4x16x576x1024 f16:', round(d['value'],3), d['unit'], round(d['ms_per_step']), 'ms per 4-clip step')"
    ( time timeout 900 python bench.py --steps 5 --warmup 2 ) > $O/bench_default.json 2> $O/bench_default.err; python -c "import json; d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1]); print('default line:', round(d['value'],3), d['split_ms_per_step'], 'strict', round(d['strict_mode']['value'],3), 'fast', round(d['fast_mode']['value'],3), 'shipped', round(d['shipped_setting']['value'],2), 'clip s', round(d['clip_mode']['ms_per_step']/1e3,2), 'traffic GB', round((d['roofline']['traffic'] or 0)/1e9,1), d['roofline']['traffic_by_kernel_class_gb'])"
    ;;
  r5m)         # r5l's A/B again with the switch actually reaching the library (bench only + the column-fastest PMC passes)
    run_bench() {   # name, extra env
      env $2 timeout 400 python bench.py --steps 3 --warmup 1 --dtype bf16x3m --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg --no-shipped-setting > $O/bench_$1.json 2> $O/bench_$1.err
      python - <<PY
import json
try:
    d = json.load(open("$O/bench_$1.json")); r = d["roofline"]
    print("$1:", round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2))
except Exception as e:
    print("$1 failed", e)
PY
    }
    for i in 1 2; do
      run_bench ncol_$i "GEO4D_DEBUG_ABLATE=17"
      run_bench group4_$i "A=1"
    done
    cd /tmp
    i=0
    for c in "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      GEO4D_DEBUG_ABLATE=17 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof/pmc_17_$i -o p -- python $R/tools/profile_unet.py 1 bf16x3m > $O/pmc_17_$i.log 2>&1
    done
    python $R/tools/pmc_summary.py $O/pmc_17.md $O/pmc_17.json $(find /tmp/prof/pmc_17_1 /tmp/prof/pmc_17_2 -name "*counter_collection.csv") > $O/pmc_summary_17.log 2>&1
    grep "whole forward" $O/pmc_17.md; tail -9 $O/pmc_17.md
    ;;
  r5l)         # grouped tile order (GROUP_M = 4) against the column-fastest order of rounds 1-4 (GEO4D_DEBUG_ABLATE=17), same library, same box: bits? frames/s? L2-miss traffic?
    ( timeout 600 python -m pytest tests/test_f16x2_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py tests/test_presplit_gpu.py -m gpu -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "^FAILED|^ERROR" $O/pytest.log | head
    run_bench() {   # name, extra env
      env $2 timeout 400 python bench.py --steps 3 --warmup 1 --dtype bf16x3m --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_$1.json 2> $O/bench_$1.err
      python - <<PY
import json
try:
    d = json.load(open("$O/bench_$1.json")); r = d["roofline"]
    print("$1:", round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print("$1 failed", e)
PY
    }
    for i in 1 2; do
      run_bench ncol_$i "GEO4D_DEBUG_ABLATE=17"
      run_bench group4_$i "A=1"
    done
    run_bench group8 "GEO4D_DEBUG_ABLATE=24"
    run_bench group2 "GEO4D_DEBUG_ABLATE=18"
    cd /tmp
    for v in 17 0; do
      i=0
      for c in "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i+1))
        GEO4D_DEBUG_ABLATE=$v timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof/pmc_${v}_$i -o p -- python $R/tools/profile_unet.py 1 bf16x3m > $O/pmc_${v}_$i.log 2>&1
      done
      python $R/tools/pmc_summary.py $O/pmc_$v.md $O/pmc_$v.json $(find /tmp/prof/pmc_${v}_1 /tmp/prof/pmc_${v}_2 -name "*counter_collection.csv") > $O/pmc_summary_$v.log 2>&1
      echo "debug_ablate=$v:"; grep "whole forward" $O/pmc_$v.md; tail -9 $O/pmc_$v.md
    done
    ;;
  r5k)         # same-box A/B of A64: the pre-A64 library + its table (build_b/, built in the container from commit 0f96093) against HEAD, interleaved
    run_bench() {   # name, extra env
      env $2 timeout 400 python bench.py --steps 3 --warmup 1 --dtype bf16x3m --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_$1.json 2> $O/bench_$1.err
      python - <<PY
import json
try:
    d = json.load(open("$O/bench_$1.json")); r = d["roofline"]
    print("$1:", round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print("$1 failed", e)
PY
    }
    cp geo4d_amd/tuning/gfx950.json $O/gfx950_head.json
    for i in 1 2; do
      cp build_b/gfx950_pre_a64.json geo4d_amd/tuning/gfx950.json
      run_bench pre_a64_$i "GEO4D_HIP_LIB=$R/build_b/libgeo4d_hip_pre_a64.so"
      cp $O/gfx950_head.json geo4d_amd/tuning/gfx950.json
      run_bench head_$i "A=1"
    done
    ;;
  r5j)         # the 320x160 two-pass tile (hint 75): same bits? does the tuner take it? end to end?
    ( timeout 600 python -m pytest tests/test_f16x2_gpu.py -m gpu -q ) > $O/pytest_x2.log 2>&1; tail -3 $O/pytest_x2.log; grep -E "^FAILED|^ERROR" $O/pytest_x2.log | head
    timeout 600 python tools/gemm_bench.py --dtype f16x2 --iters 10 --filter L0 --tiles 72,75,73,74,23 > $O/census_l0.log 2>&1; show $O/census_l0.log | tail -20
    timeout 600 python tools/gemm_bench.py --dtype f16x2 --iters 10 --filter VAE --tiles 71,73,75 > $O/census_vae.log 2>&1; show $O/census_vae.log | tail -10
    cp geo4d_amd/tuning/gfx950.json $O/gfx950_before.json
    timeout 900 python tools/tune_gemm.py $O/gfx950.json --keep --drop-prefix=4/ bf16x3m > $O/tune.log 2>&1; show $O/tune.log | tail -3
    run_bench() {   # name, dtype, extra env
      env $3 timeout 400 python bench.py --steps 3 --warmup 1 --dtype $2 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_$1.json 2> $O/bench_$1.err
      python - <<PY
import json
try:
    d = json.load(open("$O/bench_$1.json")); r = d["roofline"]
    print("$1:", round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "issued", round(r["frac_issued"], 3), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print("$1 failed", e)
PY
    }
    run_bench old_table_1 bf16x3m "A=1"
    cp geo4d_amd/tuning/gfx950.json $O/gfx950_old.json
    [ -s $O/gfx950.json ] && cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
    run_bench new_table_1 bf16x3m "A=1"
    cp $O/gfx950_old.json geo4d_amd/tuning/gfx950.json
    run_bench old_table_2 bf16x3m "A=1"
    cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
    run_bench new_table_2 bf16x3m "A=1"
    python - <<PY
import json
a, b = json.load(open("$O/gfx950_old.json")), json.load(open("$O/gfx950.json"))
print("entries taking tile 75:", sum(1 for k, v in b.items() if v[0] == 75), [k for k, v in b.items() if v[0] == 75][:12])
PY
    ;;
  r5i)         # A64: the two-pass kernels stage only the activation's hi chunks (64-byte A rows, half the A-side requests): same bits? faster?
    ( timeout 600 python -m pytest tests/test_f16x2_gpu.py -m gpu -q ) > $O/pytest_x2.log 2>&1; tail -3 $O/pytest_x2.log; grep -E "^FAILED|^ERROR" $O/pytest_x2.log | head
    timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; grep bf16x3m $O/smoke.log
    timeout 900 python tools/gemm_bench.py --dtype f16x2 --iters 10 > $O/census_x2_table.log 2>&1; show $O/census_x2_table.log | tail -31
    cp geo4d_amd/tuning/gfx950.json $O/gfx950_before.json
    timeout 900 python tools/tune_gemm.py $O/gfx950.json --keep --drop-prefix=4/ bf16x3m > $O/tune.log 2>&1; show $O/tune.log | tail -3
    run_bench() {   # name, dtype, extra env
      env $3 timeout 400 python bench.py --steps 3 --warmup 1 --dtype $2 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_$1.json 2> $O/bench_$1.err
      python - <<PY
import json
try:
    d = json.load(open("$O/bench_$1.json")); r = d["roofline"]
    print("$1:", round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "issued", round(r["frac_issued"], 3), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print("$1 failed", e)
PY
    }
    run_bench old_table bf16x3m "A=1"
    [ -s $O/gfx950.json ] && cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
    run_bench new_table_1 bf16x3m "A=1"
    run_bench new_table_2 bf16x3m "A=1"
    ( timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "50_step or multi_step" ) > $O/pytest_fullsize.log 2>&1; grep -E "\[3-step|\[50-step|passed|failed" $O/pytest_fullsize.log | cut -c1-600
    ;;
  r5h)         # one-launch GroupNorm for small statistics: correct? faster? (A/B in one box: GEO4D_GN_SMALL=2 = the three-launch form everywhere)
    ( timeout 600 python -m pytest tests/test_gn_small_gpu.py tests/test_kernels_gpu.py tests/test_bf16x3_gpu.py tests/test_parity_gpu.py -m gpu -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log; grep -E "^FAILED|^ERROR" $O/pytest.log | head
    timeout 200 python tools/norm_bench.py --dtype f32 > $O/norm.log 2>&1; show $O/norm.log | tail -24
    run_bench() {   # name, dtype, extra env
      env $3 timeout 400 python bench.py --steps 3 --warmup 1 --dtype $2 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-clip-leg > $O/bench_$1.json 2> $O/bench_$1.err
      python - <<PY
import json
try:
    d = json.load(open("$O/bench_$1.json")); r = d["roofline"]
    print("$1:", round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print("$1 failed", e)
PY
    }
    run_bench three_1 bf16x3m "GEO4D_GN_SMALL=2"
    run_bench small_1 bf16x3m "A=1"
    run_bench three_2 bf16x3m "GEO4D_GN_SMALL=2"
    run_bench small_2 bf16x3m "A=1"
    ;;
  r5g)         # where does a bf16x3m step go now? (kernel trace of the bench split into phases) + the two-pass test file
    ( timeout 300 python -m pytest tests/test_f16x2_gpu.py -m gpu -q ) > $O/pytest_x2.log 2>&1; tail -3 $O/pytest_x2.log
    cd /tmp
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode --no-strict-mode --no-shipped-setting --no-clip-leg > $O/kt.log 2>&1
    find /tmp/prof/kt -name "*kernel_stats.csv" -exec cp {} $O/kt_kernel_stats.csv \;
    KT=$(find /tmp/prof/kt -name "*kernel_trace.csv" | head -1)
    [ -n "$KT" ] && python $R/tools/prof_phases.py $KT 40 > $O/phases.md 2>&1
    head -60 $O/phases.md | cut -c1-200
    ;;
  r5f)         # the whole GPU suite (no -x: every failure shows)
    ( time timeout 1800 python -m pytest tests -m gpu -q --durations=12 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|^FAILED|^ERROR" $O/pytest.log | tail -20; grep -E "^[0-9.]+s " $O/pytest.log | head -12
    grep -E "\[clip|\[full|\[3-step|\[50-step|\[window e2e\]" $O/pytest.log | cut -c1-500 | tail -12
    ;;
  r5e)         # mid-round dry run of what the driver runs: the whole GPU suite, smoke, the default bench line (clip leg, strict / fast legs, CPU baseline)
    ( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|^FAILED|assert" $O/pytest.log | tail -12; grep -E "^[0-9.]+s " $O/pytest.log | head -12
    timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -5 $O/smoke.log
    ( time timeout 900 python bench.py --steps 3 --warmup 1 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("headline", d["dtype"], round(d["value"], 3), d["split_ms_per_step"], "strict", d.get("strict_mode", {}).get("value"), "fast", d.get("fast_mode", {}).get("value"), "shipped", d.get("shipped_setting", {}).get("value"))
print("clip", json.dumps(d.get("clip_mode"))[:1500])
print("cpu", d.get("cpu_baseline", {}).get("value"))
PY
    ;;
  r5d)         # + ff-out -> proj_out pre-split chain (both modes), raw-activation two-pass (down / up / skip / VAE upsamplers): correct? accurate (smoke!)? faster?
    ( time timeout 900 python -m pytest tests/test_f16x2_gpu.py tests/test_presplit_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gemm.log
    grep -E "passed|failed|rc=|Error|error|assert" $O/pytest_gemm.log | tail -12
    cp geo4d_amd/tuning/gfx950.json $O/gfx950_before.json
    timeout 900 python tools/tune_gemm.py $O/gfx950.json --keep bf16x3m bf16x3 > $O/tune.log 2>&1; show $O/tune.log | tail -4
    [ -s $O/gfx950.json ] && cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
    timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -5 $O/smoke.log
    ( time timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_parity_gpu.py -m gpu -q -x -s -k "vs_reference or 50_step or multi_step or window_end_to_end" ) > $O/pytest_fullsize.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fullsize.log
    grep -E "\[full|\[3-step|\[50-step|rel|passed|failed|rc=|Error|assert" $O/pytest_fullsize.log | cut -c1-700 | tail -16
    run_bench() {   # name, dtype, extra env
      env $3 timeout 400 python bench.py --steps 3 --warmup 1 --dtype $2 --no-cpu-baseline --no-fast-mode --no-clip-leg > $O/bench_$1.json 2> $O/bench_$1.err
      python - <<PY
import json
try:
    d = json.load(open("$O/bench_$1.json")); r = d["roofline"]
    print("$1:", round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "issued", round(r["frac_issued"], 3), "passes", round(r["mfma_passes_per_product"], 2), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print("$1 failed", e)
PY
    }
    run_bench x3_1 bf16x3 "A=1"
    run_bench x3m_all_1 bf16x3m "A=1"
    run_bench x3m_r5c bf16x3m "GEO4D_TWO_PASS=conv3x3,vae3x3,tconv,proj_in,ln,ff"
    run_bench x3m_all_2 bf16x3m "A=1"
    ;;
  r5c)         # two-pass f16 widened to tconv / proj_in / LayerNorm-fed plain-row projections / the GEGLU feed-forward (f16 o_split): correct? accurate? faster?
    ( time timeout 900 python -m pytest tests/test_f16x2_gpu.py tests/test_presplit_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gemm.log
    grep -E "passed|failed|rc=|Error|error|assert" $O/pytest_gemm.log | tail -12
    cp geo4d_amd/tuning/gfx950.json $O/gfx950_before.json
    timeout 900 python tools/tune_gemm.py $O/gfx950.json --keep bf16x3m > $O/tune.log 2>&1; show $O/tune.log | tail -4
    [ -s $O/gfx950.json ] && cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
    ( time timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s -k "vs_reference or 50_step or multi_step" ) > $O/pytest_fullsize.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fullsize.log
    grep -E "\[full|\[3-step|\[50-step|passed|failed|rc=|Error|assert" $O/pytest_fullsize.log | cut -c1-700 | tail -14
    run_bench() {   # name, dtype, extra env
      env $3 timeout 400 python bench.py --steps 3 --warmup 1 --dtype $2 --no-cpu-baseline --no-fast-mode --no-clip-leg > $O/bench_$1.json 2> $O/bench_$1.err
      python - <<PY
import json
try:
    d = json.load(open("$O/bench_$1.json")); r = d["roofline"]
    print("$1:", round(d["value"], 3), "frames/s", {k: round(v, 1) for k, v in d["split_ms_per_step"].items()}, "gemm ms/fwd", round(r["ms_per_unet_forward"], 2), "issued", round(r["frac_issued"], 3), "passes", round(r["mfma_passes_per_product"], 2), "shipped", round(d.get("shipped_setting", {}).get("value", 0), 2))
except Exception as e:
    print("$1 failed", e)
PY
    }
    run_bench x3_1 bf16x3 "A=1"
    run_bench x3m_all_1 bf16x3m "A=1"
    run_bench x3m_conv_only bf16x3m "GEO4D_TWO_PASS=conv3x3,vae3x3"
    run_bench x3m_all_2 bf16x3m "A=1"
    run_bench x3m_no_ff bf16x3m "GEO4D_TWO_PASS=conv3x3,vae3x3,tconv,proj_in,ln"
    ;;
  r5b)         # does the two-pass f16 form buy the short-K linears / temporal convs anything at the same bytes? (census A/B on one box)
    timeout 400 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 10 > $O/census_x3_all.log 2>&1; show $O/census_x3_all.log | tail -32
    timeout 1200 python tools/gemm_bench.py --dtype f16x2 --iters 10 --explore-all > $O/census_x2_all.log 2>&1; show $O/census_x2_all.log | grep -v "^    table" | tail -64
    ;;
  r4a)         # first call of round 4: where do the linears stand (all generations, vendor), is the table mis-tuned, what does a re-tune buy
    timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/bench_old_table.json 2> $O/bench_old.err; cut -c1-400 $O/bench_old_table.json
    TAG=$TAG bash $0 census bf16x3
    TAG=$TAG bash $0 census bf16
    TAG=$TAG bash $0 retune bf16x3 bf16
    cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
    timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_new_table.json 2> $O/bench_new.err; cut -c1-400 $O/bench_new_table.json
    timeout 300 python tools/attn_bench.py bf16 bf16x3 > $O/attn.log 2>&1; show $O/attn.log | grep "v1\|v3" | grep "2560\|forward"
    timeout 200 python tools/norm_bench.py --dtype f32 > $O/norm.log 2>&1; show $O/norm.log | tail -20
    ;;
  r4b)         # second call: the pruned library + o_split everywhere + pre-split attention inputs + GroupNorm chunking: correct? faster?
    ( time timeout 900 python -m pytest tests/test_presplit_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py tests/test_kernels_gpu.py tests/test_bf16x3_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error" $O/pytest.log | tail -8
    timeout 300 python tools/attn_bench.py bf16x3 > $O/attn.log 2>&1; show $O/attn.log | grep "2560\|forward"
    timeout 200 python tools/norm_bench.py --dtype f32 > $O/norm.log 2>&1; show $O/norm.log | grep unet
    TAG=$TAG bash $0 retune bf16x3
    cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
    timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-600 $O/bench.json
    ;;
  r4c)         # re-entry call: the groundwork commit on the GPU (tests of the touched kernels), census with every candidate + vendor bound, attention / norm benches, bench line
    ( time timeout 700 python -m pytest tests/test_presplit_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py tests/test_kernels_gpu.py tests/test_bf16x3_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error" $O/pytest.log | tail -8
    timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-900 $O/bench.json
    TAG=$TAG bash $0 census bf16x3
    timeout 300 python tools/attn_bench.py bf16 bf16x3 > $O/attn.log 2>&1; show $O/attn.log | grep "2560\|forward"
    timeout 200 python tools/norm_bench.py --dtype f32 > $O/norm.log 2>&1; show $O/norm.log | grep "unet\|L[012]"
    ;;
  r4d)         # the skewed-block attention kernel (variants 4 / 5): correct? faster?  + the o_split table fallback fix under the bench
    ( time timeout 600 python -m pytest tests/test_presplit_gpu.py tests/test_kernels_gpu.py tests/test_bf16x3_gpu.py -m gpu -q -x -k "attention or presplit or split" --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error|variant" $O/pytest.log | tail -12
    timeout 300 python tools/attn_bench.py bf16 bf16x3 > $O/attn.log 2>&1; show $O/attn.log | grep "2560\|forward\| 640"
    timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-1200 $O/bench.json
    ;;
  r4e)         # after the o_split fixes + variant 4 as the attention default: the touched tests, then the bench line
    ( time timeout 900 python -m pytest tests/test_presplit_gpu.py tests/test_kernels_gpu.py tests/test_bf16x3_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error|variant" $O/pytest.log | tail -12
    timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-1500 $O/bench.json
    ;;
  r4f)         # paired 128-byte epilogue stores (second / third generation): bit-identical? fixed cost per launch? census + bench
    ( time timeout 900 python -m pytest tests/test_presplit_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py tests/test_bf16x3_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error" $O/pytest.log | tail -6
    timeout 300 python tools/gemm_kscan.py > $O/kscan.log 2>&1; show $O/kscan.log 230
    timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-420 $O/bench.json
    ;;
  r4g)         # the table re-measured with the round-4 epilogue (fixed cost per tile 30 -> ~5 us on the phased tiles), then the bench line on it
    TAG=$TAG bash $0 retune bf16x3 bf16
    cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
    grep -c "" $O/gfx950.json.log; awk '{print $3}' $O/gfx950.json.log | sort | uniq -c | sort -rn | head -12
    timeout 500 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-420 $O/bench.json
    ;;
  r4h)         # 16-bit-row fast paths: the GEMM tests of every generation / dtype, bf16 table re-measured, bench line
    ( time timeout 900 python -m pytest tests/test_presplit_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py tests/test_bf16x3_gpu.py tests/test_kernels_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error" $O/pytest.log | tail -6
    TAG=$TAG bash $0 retune bf16
    cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
    timeout 500 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-420 $O/bench.json
    ;;
  r4i)         # per-row bias in the fast epilogue (V^T projections): tests, bench; the configs[2] clip line (64 frames -> 14 windows + alignment) on one GPU
    ( time timeout 900 python -m pytest tests/test_presplit_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py tests/test_bf16x3_gpu.py tests/test_kernels_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error" $O/pytest.log | tail -6
    timeout 500 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-420 $O/bench.json
    timeout 600 python bench.py --clip-frames 64 --no-cpu-baseline > $O/clip64.json 2> $O/clip64.err; tail -2 $O/clip64.err; cut -c1-2000 $O/clip64.json
    ;;
  r4j)         # row-bias table + per-row bias in the fast epilogue: tests, table re-measured (both modes), bench, the 64-frame clip line
    ( time timeout 900 python -m pytest tests/test_presplit_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py tests/test_bf16x3_gpu.py tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error" $O/pytest.log | tail -6
    TAG=$TAG bash $0 retune bf16x3 bf16
    cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
    timeout 500 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-420 $O/bench.json
    timeout 600 python bench.py --clip-frames 64 --no-cpu-baseline > $O/clip64.json 2> $O/clip64.err; tail -2 $O/clip64.err; cut -c1-2400 $O/clip64.json
    ;;
  r4k)         # GroupNorm statistics from the fast epilogue (GEO4D_GN_FUSED=1): correct? faster? (A/B on one box, default stays off)
    ( time timeout 600 python -m pytest tests/test_bf16x3_gpu.py tests/test_kernels_gpu.py tests/test_gemm_v2_gpu.py tests/test_gemm_v3_gpu.py tests/test_presplit_gpu.py -m gpu -q -x -k "statistics or groupnorm or gemm or conv or linear or epilogue" --durations=3 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error|error" $O/pytest.log | tail -6
    for f in 0 1 0 1; do
      GEO4D_GN_FUSED=$f timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/bench_fused$f.json 2> $O/bench_fused$f.err
      python -c "import json; d=json.load(open('$O/bench_fused$f.json')); print('GN_FUSED=$f', round(d['value'],3), {k:round(v) for k,v in d['split_ms_per_step'].items()})" || tail -3 $O/bench_fused$f.err
    done
    ;;
  tests)       # gpu test files given as arguments (default: all)
    ( time timeout 1200 python -m pytest ${@:-tests} -m gpu -q -x --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
    grep -E "passed|failed|rc=|Error" $O/pytest.log | tail -8
    ;;
  *) echo "unknown job $JOB"; exit 2;;
esac
