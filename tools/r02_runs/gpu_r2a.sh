#!/bin/bash
# Round-2 GPU call A: tune the bf16x3 GEMM shapes, run the new / affected tests, bench (bf16x3 + fast mode + CPU baseline),
# rocprofv3 kernel trace of the bench command. Everything lands in gpurun_out/.
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
timeout 600 python tools/tune_gemm.py $O/gfx950.json bf16x3 > $O/tune.log 2>&1 && cp $O/gfx950.json geo4d_amd/tuning/gfx950.json
tail -3 $O/tune.log
timeout 1500 python -m pytest tests/test_bf16x3_gpu.py tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_sizes_gpu.py \
    "tests/test_kernels_gpu.py::test_linear_bias_residual" "tests/test_kernels_gpu.py::test_attention_self" \
    "tests/test_kernels_gpu.py::test_batched_gemm_rowbias_alpha" "tests/test_kernels_gpu.py::test_lds_dma_pipelines_are_race_free" \
    -m gpu -q -x -s > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
grep -E "^\[|passed|failed|rc=|Error|error" $O/tests.log | tail -60
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la $O/prof | head
find $O/prof -name "*stats*" | head
find $O/prof -type f -size +8M -delete
du -sh $O
