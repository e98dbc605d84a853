"""The other resolutions of BASELINE.json's configs on ONE GPU: 576x256 Sintel windows (configs[2]: latent 32x72, N = 2304
tokens per frame) and 576x1024 (configs[4]: latent 72x128, N = 9216, batch 4, fp16). Parity where the CPU oracle finishes in
seconds (tiny config at 32x72); at the shipped 1.44 B config the reduced-precision modes are checked against the engine's
exact-f32 mode (pinned to the reference at 40x64 by tests/test_fullsize_gpu.py) plus finiteness / determinism properties.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def test_unet_tiny_config_at_sintel_window_size_vs_oracle(dev):
    """576x256 eval resolution (eval_dataset_geo4d.py:15) -> latent 32 x 72: ragged against every tile size (N = 2304 = 18 x 128)."""
    from geo4d_amd.unet import UNetModel
    from oracle import unet as ounet
    from oracle.params import seeded_state_dict
    g = torch.load(os.path.join(G, "unet_tiny.pt"), weights_only=False)
    sd = seeded_state_dict(g["shapes"])
    gen = torch.Generator().manual_seed(72)
    B, T, h, w = 1, 16, 32, 72
    x = torch.randn((B, 20, T, h, w), generator=gen)
    ctx = torch.randn((B, 77 + 16 * T, g["unet_config"]["context_dim"]), generator=gen)
    t, fs = torch.tensor([601]), torch.tensor([24])
    from conftest import cpu_threads
    cpu_threads()
    ref = ounet.unet_forward(sd, g["unet_config"], x, t, ctx, fs)
    m = UNetModel(**g["unet_config"], compute_dtype="f32")
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    errs = {}
    for mode in ("f32", "bf16x3", "bf16x3m", "f16", "bf16"):
        m.set_compute_dtype(mode)
        errs[mode] = rel(m(x.to(dev), t.to(dev), context=ctx.to(dev), fs=fs.to(dev)).cpu(), ref)
    print("[tiny U-Net at 32x72 latents vs oracle] " + "  ".join(f"{k}: {v:.3e}" for k, v in errs.items()))
    assert errs["f32"] < 2e-4 and errs["bf16x3"] < 2e-4 and errs["f16"] < 1e-2 and errs["bf16"] < 5e-2, errs
    assert errs["bf16x3m"] < 1e-3, errs          # the headline mode at the Sintel latent size (BASELINE configs[2])


@pytest.fixture
def engine(dev, full_engine):
    import bench
    bench.set_mode(*full_engine, "f16")
    return full_engine


def test_full_config_at_576x1024_latents(engine, dev):
    """BASELINE configs[4] shapes through the shipped 1.44 B config: latent 72 x 128 (N = 9216), batch 2 (the bench runs 4),
    fp16 — finite, deterministic, and within the fp16 distance of the exact-f32 engine; bf16x3 within 2e-4 of it."""
    model, _ = engine
    net = model.model.diffusion_model
    gen = torch.Generator().manual_seed(9216)
    B, T, h, w = 2, 16, 72, 128
    x = torch.randn((B, 16, T, h, w), generator=gen).to(dev)
    zc = torch.randn((B, 4, T, h, w), generator=gen).to(dev)
    ctx = torch.randn((B, 77 + 16 * T, 1024), generator=gen).to(dev)
    t, fs = torch.tensor([500, 120], device=dev), torch.tensor([24, 24], device=dev)
    y16 = net(x, t, context=ctx, fs=fs, c_concat=zc).clone()
    y16b = net(x, t, context=ctx, fs=fs, c_concat=zc)
    assert torch.isfinite(y16).all() and torch.equal(y16, y16b)
    net.set_compute_dtype("f32")
    y32 = net(x[:1], t[:1], context=ctx[:1], fs=fs[:1], c_concat=zc[:1]).clone()
    net.set_compute_dtype("bf16x3")
    y3 = net(x[:1], t[:1], context=ctx[:1], fs=fs[:1], c_concat=zc[:1]).clone()
    net.set_compute_dtype("bf16x3m")
    y3m = net(x[:1], t[:1], context=ctx[:1], fs=fs[:1], c_concat=zc[:1]).clone()
    net.set_compute_dtype("f16")
    e16, e3, e3m = rel(y16[:1], y32), rel(y3, y32), rel(y3m, y32)
    print(f"[1.44 B U-Net at 72x128 latents] f16 vs exact-f32 engine {e16:.3e}; bf16x3 {e3:.3e}; bf16x3m {e3m:.3e}")
    assert e16 < 1e-2 and e3 < 2e-4 and e3m < 1e-3


def test_vae_decode_at_576x1024_and_chunked_decode(engine, dev):
    """VAE AttnBlock at N = 9216 (scores materialised per bounded chunk of frames), decode at 576x1024, and the clip-chunked
    4-modality decode (pixel budget) against the un-chunked one."""
    import geo4d_amd.pipeline as pipe
    model, pvae = engine
    gen = torch.Generator().manual_seed(1024)
    z = torch.randn((2, 4, 72, 128), generator=gen).to(dev) / 0.18215
    d16 = pvae.decode_with_conf_adaptor(z)
    assert d16.shape == (2, 4, 576, 1024) and torch.isfinite(d16).all()
    pvae.set_compute_dtype("f32")
    pvae.ATTN_SCRATCH_BYTES = 600 << 20            # forces the frame loop of the attention scratch (340 MB of scores per frame)
    d32 = pvae.decode_with_conf_adaptor(z)
    pvae.set_compute_dtype("bf16x3")
    d3 = pvae.decode_with_conf_adaptor(z)
    pvae.set_compute_dtype("bf16x3m")
    d3m = pvae.decode_with_conf_adaptor(z)
    pvae.set_compute_dtype("f16")
    del pvae.ATTN_SCRATCH_BYTES
    print(f"[VAE decode 576x1024] f16 vs exact-f32 engine {rel(d16, d32):.3e}; bf16x3 {rel(d3, d32):.3e}; bf16x3m {rel(d3m, d32):.3e}")
    assert rel(d16, d32) < 1e-2 and rel(d3, d32) < 2e-4 and rel(d3m, d32) < 1e-3
    lat = torch.randn((3, 16, 2, 16, 24), generator=gen).to(dev)
    whole = pipe.decode_modalities(model, lat, pvae)
    old = pipe.DECODE_PIXEL_BUDGET
    try:
        pipe.DECODE_PIXEL_BUDGET = 3 * 2 * 128 * 192       # one clip per pass
        chunked = pipe.decode_modalities(model, lat, pvae)
    finally:
        pipe.DECODE_PIXEL_BUDGET = old
    assert whole.shape == (3, 11, 2, 128, 192) and rel(chunked, whole) < 2e-2     # fp16: tile / split-K choice differs with the batch


def test_full_config_at_576x1024_vs_the_reference(full_engine, dev):
    """BASELINE configs[4]'s spatial size against the REFERENCE itself: tests/golden/hires.pt = the reference UNetModel (shipped 1.44 B
    yaml config, name-keyed seeded weights) at x = [1,20,2,72,128] (N = 9216 tokens per frame, 2 frames so that the reference's einsum
    attention fits a CPU run; `generate.py hires`). f32 and bf16x3 < 2e-4, f16 (the mode configs[4] names) and bf16 reported and bounded."""
    from oracle.params import seeded_state_dict
    ref = torch.load(os.path.join(G, "hires.pt"), weights_only=False)
    g = torch.load(os.path.join(G, "unet_full.pt"), weights_only=False)
    net = full_engine[0].model.diffusion_model
    net.load_state_dict(seeded_state_dict(g["shapes"]), strict=True)
    rn = lambda shape, seed: torch.randn(shape, generator=torch.Generator().manual_seed(seed))
    T = 2
    x, ctx = rn((1, 20, T, 72, 128), 920).to(dev), rn((1, 77 + 16 * T, 1024), 921).to(dev)
    t, fs = torch.tensor([499], device=dev), torch.tensor([24], device=dev)
    errs = {}
    for mode in ("f32", "bf16x3", "bf16x3m", "f16", "bf16"):
        net.set_compute_dtype(mode)
        errs[mode] = rel(net(x, t, context=ctx, fs=fs).cpu(), ref["unet_out"])
    net.set_compute_dtype("bf16")
    print("[1.44 B U-Net at 72x128 latents (N = 9216) vs reference] " + "  ".join(f"{k}: {v:.3e}" for k, v in errs.items()))
    assert errs["f32"] < 2e-4 and errs["bf16x3"] < 2e-4 and errs["f16"] < 1e-2 and errs["bf16"] < 5e-2, errs
    assert errs["bf16x3m"] < 1e-3, errs          # the headline mode at BASELINE configs[4]'s latent size, against the reference's own output
