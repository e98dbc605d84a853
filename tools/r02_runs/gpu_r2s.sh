#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
O=$R/gpurun_out/r2s; mkdir -p $O
timeout 60 python -m pytest tests/test_align_gpu.py -m gpu -q -s > $O/tests.log 2>&1; echo "rc=$?"; grep -E "^\[align focal|^\[post_opt|passed|failed|^E " $O/tests.log | cut -c1-200 | tail -8
