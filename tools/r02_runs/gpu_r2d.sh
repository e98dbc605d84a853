#!/bin/bash
# Round-2 GPU call D: the tests that failed / are new, the N > 1 bench path on a 2-rank gloo rig sharing the GPU (logic check of
# window-DP + frame-sharded decode + async gathers; RCCL itself runs in the world-1 test and in the driver's 8-GPU run),
# per-shape bf16x3 GEMM throughput, a clean kernel trace + PMC passes of the bf16x3 path.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R || exit 1
export TMPDIR=/tmp
O=$R/gpurun_out/r2d; mkdir -p $O
( time timeout 300 python -m pytest tests/test_align_gpu.py tests/test_frontend_gpu.py "tests/test_fullsize_gpu.py::test_folded_depth_mean_head" \
    "tests/test_parity_gpu.py::test_captured_step_is_reused_across_windows_and_survives_model_changes" \
    "tests/test_parity_gpu.py::test_guided_synthesis_encodes_the_latent_for_every_branch" -m gpu -q -s ) > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
grep -E "^\[|passed|failed|rc=|^FAILED|^E  " $O/tests.log | tail -30 | cut -c1-250
for mode in sharded local; do
  GEO4D_DIST_BACKEND=gloo GEO4D_SINGLE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
     bench.py --gpus 2 --steps 1 --warmup 1 --decode $mode --no-fast-mode > $O/bench_2rank_$mode.json 2> $O/bench_2rank_$mode.err
  echo "2-rank $mode rc=$?"; tail -c 600 $O/bench_2rank_$mode.json; tail -3 $O/bench_2rank_$mode.err | cut -c1-300
done
timeout 200 python tools/gemm_bench.py --dtype bf16x3 --iters 10 > $O/gemm_x3.log 2>&1; tail -4 $O/gemm_x3.log
timeout 120 python tools/gemm_bench.py --dtype bf16x3 --iters 10 --ablate 1 > $O/gemm_x3_nosplit.log 2>&1; tail -2 $O/gemm_x3_nosplit.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode > $O/kt.log 2>&1
find /tmp/prof/kt -name "*kernel_stats.csv" -exec cp {} $O/kt_kernel_stats.csv \;
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof/pmc$i -o p -- python $R/tools/profile_unet.py 1 bf16x3 > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc.md $O/pmc.json $(find /tmp/prof/pmc1 /tmp/prof/pmc2 /tmp/prof/pmc3 -name "*counter_collection.csv") > $O/pmc_summary.log 2>&1
tail -3 $O/pmc_summary.log
