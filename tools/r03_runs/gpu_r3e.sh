#!/bin/bash
# kernel-trace A/B of one eager U-Net forward: baseline (raw activations, 3-launch GN) vs one-launch GN vs pre-split producers
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3e; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/$name -o kt -- python $R/tools/profile_unet.py 3 bf16x3 > $O/$name.log 2>&1
  f=$(find /tmp/prof/$name -name "*kernel_stats.csv" | head -1)
  echo "== $name: $(grep 'eager forward' $O/$name.log)"; head -14 "$f" | cut -d, -f1-6 | cut -c1-200
  cp "$f" $O/${name}_kernel_stats.csv
}
run base GEO4D_X3_PRESPLIT=0 GEO4D_GN_ONE_LAUNCH=0
run gn1 GEO4D_X3_PRESPLIT=0 GEO4D_GN_ONE_LAUNCH=1
run ps GEO4D_X3_PRESPLIT=1 GEO4D_GN_ONE_LAUNCH=0
