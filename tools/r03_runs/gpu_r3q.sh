#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3q; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 120 python tools/v3_timeline.py conv > $O/timeline_conv.log 2>&1; grep -v amdgpu.ids $O/timeline_conv.log
timeout 120 python tools/v3_timeline.py lin > $O/timeline_lin.log 2>&1; grep -v amdgpu.ids $O/timeline_lin.log
