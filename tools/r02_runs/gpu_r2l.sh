#!/bin/bash
# round 2 run L: where does a bf16x3 slab go? 256x128 tile forced, steady-loop ablation builds (GEO4D_ABL bits: 1 no DMA, 2 no fragment reads, 4 no split)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2l; mkdir -p $O
for v in 0 8 7 15; do
  GEO4D_HIP_LIB=$R/geo4d_amd/csrc/libgeo4d_hip_a$v.so timeout 120 python tools/gemm_bench.py --dtype bf16x3 --iters 10 --tile 11 > $O/abl_$v.log 2>&1
  echo "ABL=$v: $(tail -1 $O/abl_$v.log)"
done
paste <(cut -c1-78 $O/abl_0.log) <(cut -c60-69 $O/abl_8.log) <(cut -c60-69 $O/abl_7.log) <(cut -c60-69 $O/abl_15.log) | head -36
