#!/usr/bin/env python3
"""Calibrates bench.py's `cpu_baseline.kind = "port"`: wall time of the ORACLE (oracle/unet.py, oracle/vae.py — what the GPU box can
run) against the IMPORTED REFERENCE (/root/reference lvdm classes — only present in the build container) on the same host, same
threads, same seeded weights and inputs, at the BASELINE size (U-Net forward at 1x20x16x40x64, conf-adaptor decode of one 40x64
latent frame). Also re-checks that both produce the same numbers. Build container only:
    python tools/cpu_crosscheck.py > profiles/r04_cpu_baseline_crosscheck.md
The reference is imported the way tests/golden/generate.py imports it (absent third-party modules stubbed); nothing is copied."""
import importlib.util
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("golden_generate", os.path.join(ROOT, "tests", "golden", "generate.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)          # sets sys.path for /root/reference and installs the stubs generate.py needs


def timed(fn, reps=2):
    best, out = float("inf"), None
    for _ in range(reps):
        t0 = time.time()
        with torch.no_grad():
            out = fn()
        best = min(best, time.time() - t0)
    return best, out


def main():
    import yaml
    from oracle import unet as ounet, vae as ovae
    from oracle.params import fill_module_
    cores = os.cpu_count() or 8
    torch.set_num_threads(cores)
    with open(os.path.join(gen.REF, "configs/inference_geo4d.yaml")) as f:
        ycfg = yaml.safe_load(f)
    ucfg = dict(ycfg["model"]["params"]["unet_config"]["params"])
    ucfg["use_checkpoint"] = False
    from lvdm.modules.networks.openaimodel3d import UNetModel
    from lvdm.models.autoencoder import AutoencoderKL
    inp = gen.fullsize_inputs()
    print("# Round 4 — CPU wall time: oracle (bench.py's `cpu_baseline`, kind = port) vs the imported reference, same host\n")
    print(f"`python tools/cpu_crosscheck.py` in the build container: {cores} threads, fp32, seeded weights (oracle/params.py), best of 2.\n")
    print("| piece | reference (imported `/root/reference` classes) | oracle (`oracle/`) | oracle / reference | outputs rel L2 |")
    print("|---|---|---|---|---|")
    ref = UNetModel(**ucfg).eval()
    fill_module_(ref)
    t_ref, y_ref = timed(lambda: ref(inp["x"], inp["t"], context=inp["context"], fs=inp["fs"]))
    usd = {k: v.detach() for k, v in ref.state_dict().items()}
    del ref
    t_or, y_or = timed(lambda: ounet.unet_forward(usd, ucfg, inp["x"], inp["t"], inp["context"], inp["fs"]))
    e = ((y_or - y_ref).norm() / y_ref.norm()).item()
    print(f"| U-Net forward 1x20x16x40x64 (1.44 B parameters) | {t_ref:.1f} s | {t_or:.1f} s | {t_or / t_ref:.2f} | {e:.1e} |", flush=True)
    del usd
    vparams = ycfg["pointmap_vae_config"]["params"]
    vae = AutoencoderKL(**{k: (gen.ad(v) if isinstance(v, dict) else v) for k, v in vparams.items()}).eval()
    fill_module_(vae)
    t_refv, d_ref = timed(lambda: vae.decode_with_conf_adaptor(inp["z"]))
    vsd = {k: v.detach() for k, v in vae.state_dict().items()}
    del vae
    t_orv, d_or = timed(lambda: ovae.decode_with_conf_adaptor(vsd, vparams["ddconfig"], vparams["adaptorconfig"], inp["z"]))
    ev = ((d_or - d_ref).norm() / d_ref.norm()).item()
    print(f"| conf-adaptor decode of one 40x64 latent frame -> 320x512 | {t_refv:.1f} s | {t_orv:.1f} s | {t_orv / t_refv:.2f} | {ev:.1e} |")
    S, T = 50, 16
    w_ref, w_or = S * t_ref + 4 * T * t_refv, S * t_or + 4 * T * t_orv
    print(f"\nOne BASELINE window (50 forwards + 16 frames x 4 decodes, the composition bench.py uses): reference {w_ref:.0f} s = "
          f"{T / w_ref:.4f} frames/s, oracle {w_or:.0f} s = {T / w_or:.4f} frames/s on these {cores} threads: the port's time is "
          f"{w_or / w_ref:.2f}x the reference's, so `cpu_baseline.value` (the oracle on the GPU box's host) "
          f"{'understates' if w_or > w_ref else 'overstates'} the reference's CPU rate by that factor.")


if __name__ == "__main__":
    main()
