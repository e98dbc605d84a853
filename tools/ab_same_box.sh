#!/bin/bash
# Same-box A/B of the bench line: the round-3 tree (exported to gpurun_ab_r3/ by `git archive d7ec541`, built ON the GPU box) against this
# tree, interleaved A B A B so that clock drift between the runs shows. Prepare (build container): mkdir gpurun_ab_r3 && git archive d7ec541 | tar -x -C gpurun_ab_r3
# (git-ignored; drop its tests/golden and profiles to keep the push small). usage (via gpurun): bash tools/ab_same_box.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_ab; mkdir -p $O
cd $R/gpurun_ab_r3 && ( time make -j64 > $O/build_r3.log 2>&1 ) 2>&1 | grep real
ls -la geo4d_amd/csrc/libgeo4d_hip.so | cut -c1-120
for i in 1 2; do
  cd $R/gpurun_ab_r3 && timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/r3_$i.json 2> $O/r3_$i.err
  cd $R && timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/r4_$i.json 2> $O/r4_$i.err
done
python - <<PY
import json
for n in ("r3_1","r4_1","r3_2","r4_2"):
    try:
        d=json.load(open("$O/%s.json"%n)); print(n, round(d["value"],3), "frames/s", {k:round(v) for k,v in d["split_ms_per_step"].items()})
    except Exception as e: print(n, "failed", e)
PY
