#!/bin/bash
# Round-2 GPU call B (budgeted ~14 min): new / changed tests, attention A/B builds, the bench line.
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
( time timeout 420 python -m pytest tests/test_frontend_gpu.py tests/test_fullsize_gpu.py tests/test_sizes_gpu.py \
    "tests/test_kernels_gpu.py::test_attention_variants" "tests/test_bf16x3_gpu.py::test_attention_self" \
    "tests/test_parity_gpu.py::test_unet_full_config_vs_reference_golden" "tests/test_parity_gpu.py::test_rccl_path_executes_on_one_gpu" \
    -m gpu -q -x -s --durations=12 ) > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
grep -E "^\[|passed|failed|rc=|Error|error|real|s call" $O/tests.log | tail -50 | cut -c1-260
timeout 150 python tools/attn_bench.py bf16 bf16x3 > $O/attn.log 2>&1; grep "per U-Net" $O/attn.log
( time timeout 480 python bench.py --steps 2 --warmup 1 ) > $O/bench.json 2> $O/bench.err
tail -c 2500 $O/bench.json; tail -5 $O/bench.err
