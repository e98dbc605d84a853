#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3l; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/al -o kt -- python $R/tools/align_profile.py > $O/al.log 2>&1
f=$(find /tmp/prof/al -name "*kernel_stats.csv" | head -1); cp "$f" $O/align_kernel_stats.csv; head -25 "$f" | cut -d, -f1-6 | cut -c1-170
