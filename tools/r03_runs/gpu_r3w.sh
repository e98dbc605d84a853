#!/bin/bash
# last A/B of the round on the phased K loop (make ABLATION=1 build): no s_setprio (x4), waves 4-7 issuing their DMA pieces mid-quadrant (x9)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3w; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 50 python - > $O/equal.log 2>&1 <<'PY'
import torch
from geo4d_amd import ops, pack
dev = torch.device("cuda:0"); torch.manual_seed(0)
F_, H, W, Cin, N = 4, 40, 64, 320, 320
x = torch.randn((F_ * H * W, Cin), device=dev); w = pack.split_bf16(torch.randn((N, 9 * Cin), device=dev) / 50); b = torch.randn((N,), device=dev)
xa = ops.SplitAct.wrap(pack.split_bf16(x))
run = lambda t: ops.conv2d(xa, w, b, F=F_, Hin=H, Win=W, KH=3, KW=3, pad=1, tile_hint=t, split_k=1)[0]
ref = run(72)
print({t: bool(torch.equal(run(t), ref)) for t in (84, 89, 94, 99, 25)})
PY
cat $O/equal.log | grep -v amdgpu
for f in "L0 conv3x3 320->320" "L0 conv3x3 640->320" "L0 ffout" "L0 conv3d"; do timeout 40 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 20 --filter "$f" --tiles 23,72,84,89 >> $O/ab.log 2>&1; done
for f in "VAE conv3x3 512 @40x64" "L1 geglu"; do timeout 40 python tools/gemm_bench.py --dtype bf16x3 --presplit --iters 10 --filter "$f" --tiles 22,71,94,99 >> $O/ab.log 2>&1; done
grep -v "amdgpu.ids\|^shape\|census" $O/ab.log | cut -c1-200
