#!/usr/bin/env python3
"""Guided sampling (classifier-free guidance, scripts/infer_geo4d.sh:29-32 "better visual results": 2 or 3 U-Net evaluations per step) at BASELINE
size: the evaluations of a step as ONE batched forward (round 6, DDIMSampler(batch_cfg=True), the default) against one forward per conditioning.
usage (GPU box): python tools/cfg_bench.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    model, pvae = bench.build("bf16x3m", dev)
    from geo4d_amd.ddim import DDIMSampler
    from geo4d_amd.ddim_multiplecond import DDIMSampler as Multi
    T, h, w = 16, 40, 64
    g = torch.Generator().manual_seed(5)
    mk = lambda: {"c_crossattn": [torch.randn((1, 77 + 16 * T, 1024), generator=g).to(dev)], "c_concat": [zc]}
    zc = torch.randn((1, 4, T, h, w), generator=g).to(dev)
    cond, uc, uci = mk(), mk(), mk()
    x_T = torch.randn((1, 16, T, h, w), generator=g).to(dev)
    fs = torch.tensor([24], device=dev)
    for name, cls, extra in (("2-way CFG 7.5", DDIMSampler, {}), ("3-way CFG 7.5 / img 2.0", Multi, dict(cfg_img=2.0, unconditional_conditioning_img_nonetext=uci))):
        outs = {}
        for batched in (False, True):
            s = cls(model, batch_cfg=batched)
            run = lambda: s.sample(S=S, conditioning=cond, batch_size=1, shape=[16, T, h, w], verbose=False, eta=0.0, unconditional_guidance_scale=7.5,
                                   unconditional_conditioning=uc, fs=fs, x_T=x_T, timestep_spacing="uniform_trailing", guidance_rescale=0.7, **extra)[0]
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = run()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            outs[batched] = out
            print(f"{name}: {'one batched forward per step' if batched else 'one forward per conditioning'}: {1e3 * dt / S:7.1f} ms per DDIM step")
        a, b = outs[True].double(), outs[False].double()
        print(f"{name}: batched vs separate, latent rel L2 {float((a - b).norm() / b.norm()):.2e}", flush=True)


if __name__ == "__main__":
    main()
