#!/bin/bash
# round 2 run N: SQ counters of the bf16x3 256x128 kernel on one long-K conv (full loop, accumulator-major = lib a8; MFMA-only loop = lib a15)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r2n; mkdir -p $O
for v in 8 15; do
  i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    GEO4D_HIP_LIB=$R/geo4d_amd/csrc/libgeo4d_hip_a$v.so timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof/n${v}_$i -o p -- python $R/tools/gemm_bench.py --dtype bf16x3 --iters 3 --tile 11 --filter "512 @40x64" > $O/pmc_${v}_$i.log 2>&1
    f=$(find /tmp/prof/n${v}_$i -name "*counter_collection.csv" | head -1)
    python - "$f" "$v" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "conv_gemm_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("lib a" + sys.argv[2], {k: round(sum(v) / len(v)) for k, v in acc.items()}, "launches", {k: len(v) for k, v in acc.items()})
PY
  done
done 2>&1 | tee $O/summary.log
