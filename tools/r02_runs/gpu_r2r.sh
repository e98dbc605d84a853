#!/bin/bash
# round 2 run R: same box A/B of the bf16x3 bench line: epilogue with prefetched residual chunks (current) vs the previous epilogue
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2r; mkdir -p $O
line() { python -c "
import json;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['split_ms_per_step'], d['roofline']['ms_per_unet_forward'])"; }
timeout 80 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/new.json 2> $O/new.err; line $O/new.json new
GEO4D_HIP_LIB=$R/geo4d_amd/csrc/libgeo4d_hip_old.so timeout 80 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/old.json 2> $O/old.err; line $O/old.json old
