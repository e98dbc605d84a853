#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2o; mkdir -p $O
for v in 8 15; do echo "== lib t$v"; GEO4D_HIP_LIB=$R/geo4d_amd/csrc/libgeo4d_hip_t$v.so timeout 120 python tools/probe/tstamp_run.py 2>&1 | grep -v amdgpu.ids | tee $O/tstamp_$v.log; done
