#!/usr/bin/env python3
"""bench.py — Geo4D hot path on MI355X: denoised latent frames/sec.

One "step" = one pass of the hot path over one 16-frame window at 16x3x320x512 (latent 16x16x40x64):
50-step DDIM (eta 0, CFG 1.0, uniform_trailing, dynamic rescale) over the 1.44 B-parameter 3D U-Net + the 4-modality
VAE decode (point map + confidence, ray, ray-moment, inverse depth). BASELINE.json configs[1].
Inputs are synthetic (seeded), weights random-init, everything resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0). `roofline` prices the whole step against the dense bf16 MFMA peak with the algorithmic
FLOP count of SURVEY.md §8(d) (733.6 TFLOP per window); `cpu_baseline` times the oracle (CPU port of the reference path)
on a bounded sample on the host cores of the same box.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from geo4d_amd import dist as gdist  # noqa: E402
from geo4d_amd.ddim import DDIMSampler  # noqa: E402
from geo4d_amd.pipeline import decode_modalities  # noqa: E402
from geo4d_amd.registry import instantiate_from_config, load_config  # noqa: E402

TFLOP_UNET_STEP = 12.61      # SURVEY.md §6 [probe]: one U-Net forward at 1x20x16x40x64
TFLOP_DECODE_FRAME = 6.4475  # 1.757 (conf decode) + 3 x 1.563 per frame
MFMA_PEAK_TF = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}


def build(dtype, dev):
    cfg = load_config(os.path.join(ROOT, "configs", "inference_geo4d.yaml"))
    mcfg = cfg.pop("model")
    mcfg["params"]["unet_config"]["params"]["use_checkpoint"] = False
    mcfg["params"]["unet_config"]["params"]["compute_dtype"] = dtype
    mcfg["params"]["first_stage_config"]["params"]["compute_dtype"] = dtype
    pcfg = cfg.pop("pointmap_vae_config")
    pcfg["params"]["compute_dtype"] = dtype
    torch.manual_seed(0)
    model = instantiate_from_config(mcfg).to(dev)
    pvae = instantiate_from_config(pcfg).to(dev)
    return model, pvae


def cpu_baseline(model, pvae, ddim_steps, T, h, w):
    """Oracle (CPU restatement of the reference path, fp32) on a bounded sample, extrapolated by token count."""
    from oracle import unet as ounet
    from oracle import vae as ovae
    cores = min(os.cpu_count() or 1, 32)   # more OpenMP threads than that make these small CPU convs slower, not faster
    torch.set_num_threads(cores)
    usd = {k: v.detach().float().cpu() for k, v in model.model.diffusion_model.state_dict().items()}
    vsd = {k: v.detach().float().cpu() for k, v in pvae.state_dict().items()}
    ucfg = dict(model.model.diffusion_model.cfg)
    hs, ws = 8, 8                                    # sample: latent 8x8 instead of 40x64 (1/40 of the tokens)
    g = torch.Generator().manual_seed(1)
    x = torch.randn((1, 20, T, hs, ws), generator=g)
    ctx = torch.randn((1, 77 + 16 * T, ucfg["context_dim"]), generator=g)
    t0 = time.time()
    ounet.unet_forward(usd, ucfg, x, torch.tensor([499]), ctx, torch.tensor([24]))
    t_unet = time.time() - t0
    z = torch.randn((1, 4, hs, ws), generator=g)
    t0 = time.time()
    ovae.decode_with_conf_adaptor(vsd, pvae.ddconfig, pvae.adaptorconfig, z)
    t_dec = time.time() - t0
    scale = (h * w) / (hs * ws)
    t_window = ddim_steps * t_unet * scale + T * 4 * t_dec * scale
    return {"value": T / t_window, "unit": "denoised latent frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32: 1 U-Net forward at 1x20x{T}x{hs}x{ws} ({t_unet:.2f} s) + 1 conf-decode frame at {hs}x{ws} latents "
                      f"({t_dec:.2f} s), scaled linearly by token count x{scale:.0f} to {h}x{w} and to {ddim_steps} steps + {4 * T} frame "
                      "decodes (attention's quadratic term ignored: favours the CPU)"}


def gemm_timeline(model, x_T, cond, fs, dev):
    """Dominant kernel, measured live: one EAGER U-Net forward (outside the timed region) with every geo4d_conv_gemm launch
    bracketed by HIP events on the launch stream. Returns (launches, algorithmic TFLOP, total ms)."""
    from geo4d_amd import ops
    t = torch.tensor([499], device=dev)
    model.apply_model(x_T, t, cond, fs=fs)
    torch.cuda.synchronize()
    ops.GEMM_TIMELINE = []
    model.apply_model(x_T, t, cond, fs=fs)
    torch.cuda.synchronize()
    tl, ops.GEMM_TIMELINE = ops.GEMM_TIMELINE, None
    return len(tl), sum(f for f, _, _ in tl) / 1e12, sum(a.elapsed_time(b) for _, a, b in tl)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, world, local = gdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local)
    T, h, w = args.frames, args.height // 8, args.width // 8
    model, pvae = build(args.dtype, dev)
    g = torch.Generator().manual_seed(123 + rank)
    ctx = torch.randn((1, 77 + 16 * T, 1024), generator=g).to(dev)
    zc = torch.randn((1, 4, T, h, w), generator=g).to(dev)
    cond = {"c_crossattn": [ctx], "c_concat": [zc]}
    fs = torch.tensor([24], dtype=torch.long, device=dev)
    sampler = DDIMSampler(model, use_graph=not args.no_graph)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    split = [0.0, 0.0]

    def one_window(seed, timed):
        x_T = torch.randn((1, 16, T, h, w), generator=torch.Generator().manual_seed(seed)).to(dev)
        if timed:
            ev[0].record()
        lat, _ = sampler.sample(S=args.ddim_steps, conditioning=cond, batch_size=1, shape=[16, T, h, w], verbose=False,
                                unconditional_guidance_scale=1.0, unconditional_conditioning=None, eta=0.0, cfg_img=None,
                                fs=fs, x_T=x_T, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                unconditional_conditioning_img_nonetext=None)
        if timed:
            ev[1].record()
        out = decode_modalities(model, lat, pvae)
        if world > 1:
            out = gdist.all_gather_windows(out, world)       # one window per rank per step -> whole clip on every rank
        if timed:
            ev[2].record()
            torch.cuda.synchronize()
            split[0] += ev[0].elapsed_time(ev[1])
            split[1] += ev[1].elapsed_time(ev[2])
        return out

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_window(1000 + i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_window(2000 + i, True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    assert torch.isfinite(out).all()

    if rank == 0:
        frames = T * args.steps * world
        tflop_window = TFLOP_UNET_STEP * args.ddim_steps + TFLOP_DECODE_FRAME * T
        if (args.height, args.width, T) != (320, 512, 16):
            tflop_window *= (h * w * T) / (40 * 64 * 16)
        achieved = tflop_window * args.steps * world / dt / world      # per-GPU TFLOP/s, whole step
        peak = MFMA_PEAK_TF[args.dtype]
        x_T = torch.randn((1, 16, T, h, w), generator=torch.Generator().manual_seed(7)).to(dev)
        n_gemm, tf_gemm, ms_gemm = gemm_timeline(model, x_T, cond, fs, dev)
        traffic, traffic_note = None, "no PMC summary committed for this dtype"
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc.json")
        if args.dtype == "bf16" and (args.height, args.width, T) == (320, 512, 16) and os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)["per_unet_forward"]
            traffic = pmc["fetch_bytes_x2"] + pmc["write_bytes"]
            traffic_note = ("L2-miss (fabric-side) bytes of ONE U-Net forward, ALL its kernels (conv_gemm is ~80 % of them): "
                            "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 from separate rocprofv3 --pmc passes (profiles/r01_pmc_unet_forward.md; "
                            "Infinity-Cache hits included, so an upper bound on HBM bytes)")
        res = {
            "metric": "denoised latent frames/sec (16x320x512, 50-step DDIM)",
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (seeded N(0,1) latents/context, random-init weights)",
            "config": {"workload": f"1 window = {T}x3x{args.height}x{args.width} clip (latent 16x{T}x{h}x{w}): {args.ddim_steps}-step DDIM "
                                   f"(eta 0, CFG 1, uniform_trailing, dynamic rescale) over the 1.44B-param 3D U-Net + 4-modality VAE decode; "
                                   f"BASELINE.json configs[1]", "windows_per_gpu_per_step": 1,
                       "parallelism": f"window-dp{world}" + (" + all-gather of decoded maps" if world > 1 else ""),
                       "hipgraph": not args.no_graph},
            "split_ms_per_step": {"ddim_denoise": split[0] / args.steps, "vae_decode_4_modalities": split[1] / args.steps},
            "roofline": {"bound": "mfma", "kernel": "conv_gemm_kernel (MFMA implicit GEMM: every conv / linear / batched GEMM of the path)",
                         "achieved": tf_gemm / ms_gemm * 1e3, "peak": peak, "unit": "TFLOP/s", "frac": tf_gemm / ms_gemm * 1e3 / peak,
                         "launches_per_unet_forward": n_gemm, "tflop_per_unet_forward": tf_gemm, "ms_per_unet_forward": ms_gemm,
                         "avg_launch_us": 1e3 * ms_gemm / n_gemm,
                         "traffic": traffic, "traffic_note": traffic_note,
                         "note": "sum of 2*M*N*K over the conv_gemm launches of ONE eager U-Net forward / sum of their HIP-event "
                                 "durations on the launch stream (brackets include a split-K launch's reduce kernel and ~2 us of "
                                 "dispatch gap each; the rocprofv3 kernel trace in profiles/ gives the pure kernel time)",
                         "whole_step": {"achieved": achieved, "frac": achieved / peak,
                                        "note": f"{tflop_window:.1f} algorithmic TFLOP per window (SURVEY §8d: {TFLOP_UNET_STEP} x S + "
                                                f"{TFLOP_DECODE_FRAME} x T) / measured step time, per GPU"}},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(model, pvae, args.ddim_steps, T, h, w)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()        # rank 0 is still timing its conv_gemm timeline: nobody tears the group down under it
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
