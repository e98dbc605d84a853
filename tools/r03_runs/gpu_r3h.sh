#!/bin/bash
# new reference pins + bf16 re-tune (v1 + v2 tiles) + fast-mode check
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3h; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_sizes_gpu.py tests/test_parity_gpu.py -q -x -s -k "reference or mask" ) > $O/pins.log 2>&1; echo "rc=$?" >> $O/pins.log
grep -E "vs reference|ddim mask|passed|failed|rc=" $O/pins.log | tail -8
( time GEO4D_AUTOTUNE=1 timeout 900 python tools/tune_gemm.py $O/gfx950.json bf16 ) > $O/tune.log 2>&1; tail -2 $O/tune.log
cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'], d['split_ms_per_step'], d['roofline']['ms_per_unet_forward'], d['roofline']['frac'], d['fast_mode']['value'])"
