R=$PWD; mkdir -p $R/gpurun_out/r4_pmc_attn; export TMPDIR=/tmp; cd /tmp
i=0
for c in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof/pa$i -o p -- python $R/tools/pmc_attn.py > $R/gpurun_out/r4_pmc_attn/log$i.txt 2>&1
done
python - <<'PY'
import csv, glob, collections
csv.field_size_limit(1 << 30)
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('/tmp/prof/pa*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f, newline='')):
        if 'flash_attn' in r['Kernel_Name']:
            tot[(r['Kernel_Name'][:60], r['Counter_Name'])] += float(r['Counter_Value']); n[(r['Kernel_Name'][:60], r['Counter_Name'])] += 1
for k in sorted(tot): print(k[0], k[1], tot[k] / n[k], n[k])
PY
