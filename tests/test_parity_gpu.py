"""Parity of the HIP path (U-Net, DDIM sampler, VAE decode, 4-modality decode) against
 (a) golden outputs produced by the reference itself (tests/golden/*.pt, see generate.py) and
 (b) the oracle on further seeded inputs.
Tolerances (relative L2 on the whole tensor), written per mode. The north-star bar is 1e-3 on the point map:
   f32    (exact-f32 MFMA)                                    : we assert 2e-4
   bf16x3 (f32 storage, 3-term bf16 split MFMA; BENCH mode)   : we assert 2e-4 — the mode bench.py quotes meets the bar
   f16    (single f16 pass, fp32 accumulate)                  : 1e-2   (fast mode, reported)
   bf16   (single bf16 pass, fp32 accumulate)                 : 5e-2   (fast mode, reported: 8-bit mantissas through ~300 GEMMs)
   bf16x3m (bf16x3 + two-pass f16 on branch activations; the HEADLINE mode since round 5): 1e-3, the north-star bar itself, on every
           fixture below (round 6: it sits in MODES; the tiny random-weight configurations are its worst case)
tests/precision_sim.py + tests/test_precision_floor.py show why no single 16-bit pass can meet 1e-3 on this network.
"""
import os

import pytest
import torch

from oracle import ddim as oddim
from oracle import pipeline as opipe
from oracle import unet as ounet
from oracle import vae as ovae
from oracle.params import seeded_state_dict

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = [("f32", 2e-4), ("bf16x3", 2e-4), ("bf16x3m", 1e-3), ("f16", 1e-2), ("bf16", 5e-2)]


def load(name):
    return torch.load(os.path.join(G, name), weights_only=False)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm()).item()


def build_unet(cfg, shapes, dev, mode):
    from geo4d_amd.unet import UNetModel
    m = UNetModel(**cfg, compute_dtype=mode)
    m.load_state_dict(seeded_state_dict(shapes), strict=True)
    return m.to(dev)


def build_vae(g, dev, mode):
    from geo4d_amd.vae import AutoencoderKL
    m = AutoencoderKL(ddconfig=g["ddconfig"], lossconfig={"target": "torch.nn.Identity"}, embed_dim=4,
                      adaptorconfig=g["adaptorconfig"], compute_dtype=mode)
    m.load_state_dict(seeded_state_dict(g["shapes"]), strict=True)
    return m.to(dev)


@pytest.mark.parametrize("mode,tol", MODES)
@pytest.mark.parametrize("case", ["t16_8x8", "b2_t5_8x16"])
def test_unet_vs_reference_golden(dev, mode, tol, case):
    g = load("unet_tiny.pt")
    m = build_unet(g["unet_config"], g["shapes"], dev, mode)
    c = g["cases"][case]
    xc = torch.cat([c["x"], c["c_concat"]], 1).to(dev)
    y = m(xc, c["t"].to(dev), context=c["context"].to(dev), fs=c["fs"].to(dev), cfg_img=None,
          unconditional_conditioning_img_nonetext=None)
    y2 = m(c["x"].to(dev), c["t"].to(dev), context=c["context"].to(dev), fs=c["fs"].to(dev), c_concat=c["c_concat"].to(dev))
    e = rel(y, c["out"])
    print(f"[unet {case}] mode={mode} rel_l2 vs reference = {e:.3e} (tol {tol:.0e})")
    assert y.shape == c["out"].shape and y.dtype == torch.float32
    assert torch.equal(y, y2), "pre-concatenated and two-source inputs must give identical results"
    assert e < tol


@pytest.mark.parametrize("mode", ["f32", "bf16x3m", "bf16"])
def test_skip_concatenation_written_by_its_producers_is_bit_identical(dev, mode):
    """Round 6 (unet.FUSED_CONCAT): the last launch of an input block / of the previous output block writes straight into the buffer the
    output block reads as torch.cat([h, hs.pop()], 1) - no concat_channels copy. Same kernels on the same operands (only row pitches
    differ): the U-Net output equals the copying path bit for bit, in every storage type, for both golden cases."""
    import geo4d_amd.unet as gu
    g = load("unet_tiny.pt")
    m = build_unet(g["unet_config"], g["shapes"], dev, mode)
    for case in ("t16_8x8", "b2_t5_8x16"):
        c = g["cases"][case]
        args = (torch.cat([c["x"], c["c_concat"]], 1).to(dev), c["t"].to(dev))
        kw = dict(context=c["context"].to(dev), fs=c["fs"].to(dev))
        old = gu.FUSED_CONCAT
        try:
            gu.FUSED_CONCAT = False
            ref = m(*args, **kw).clone()
            gu.FUSED_CONCAT = True
            out = m(*args, **kw)
        finally:
            gu.FUSED_CONCAT = old
        assert torch.equal(out, ref), f"{mode} {case}: {rel(out, ref):.3e}"


def test_unet_full_config_vs_reference_golden(dev, full_engine):
    """The shipped yaml config (1.44 B parameters) against the reference UNetModel's own output (8x8 latents; the 40x64 pin is
    tests/test_fullsize_gpu.py). One shared model instance, every compute mode."""
    g = load("unet_full.pt")
    m = full_engine[0].model.diffusion_model
    m.load_state_dict(seeded_state_dict(g["shapes"]), strict=True)
    for mode, tol in (("f32", 2e-4), ("bf16x3", 2e-4), ("bf16x3m", 1e-3), ("bf16", 5e-2)):
        m.set_compute_dtype(mode)
        y = m(g["x"].to(dev), g["t"].to(dev), context=g["context"].to(dev), fs=g["fs"].to(dev))
        e = rel(y, g["out"])
        print(f"[unet full config] mode={mode} rel_l2 vs reference = {e:.3e} (tol {tol:.0e})")
        assert e < tol


@pytest.mark.parametrize("mode,tol", MODES)
def test_vae_decode_vs_reference_golden(dev, mode, tol):
    g = load("vae_tiny.pt")
    m = build_vae(g, dev, mode)
    d = m.decode(g["z"].to(dev))
    dc = m.decode_with_conf_adaptor(g["z"].to(dev))
    e1, e2 = rel(d, g["decode"]), rel(dc, g["decode_with_conf_adaptor"])
    print(f"[vae] mode={mode} decode {e1:.3e} decode_with_conf_adaptor {e2:.3e} (tol {tol:.0e})")
    assert d.shape == g["decode"].shape and dc.shape == g["decode_with_conf_adaptor"].shape
    assert e1 < tol and e2 < tol


@pytest.mark.parametrize("mode,tol", MODES)
def test_vae_encode_vs_reference_golden(dev, mode, tol):
    """SURVEY §8(f) N3, first piece: AutoencoderKL.encode / encode_with_adaptor (Encoder + stride-2 Downsample with (0,1,0,1)
    padding + mid attention + quant_conv folded into conv_out) vs moments produced by the reference itself."""
    g = load("vae_encode_tiny.pt")
    m = build_vae(g, dev, mode)
    post = m.encode(g["x"].to(dev))
    post_a = m.encode_with_adaptor(g["x"].to(dev))
    e1, e2 = rel(post.parameters, g["moments"]), rel(post_a.parameters, g["moments_adaptor"])
    print(f"[vae encode] mode={mode} encode {e1:.3e} encode_with_adaptor {e2:.3e} (tol {tol:.0e})")
    assert post.parameters.shape == g["moments"].shape and post.mean.shape[1] == 4
    assert e1 < tol and e2 < tol
    with pytest.raises(ValueError):
        m.encode(torch.zeros((1, 3, 20, 64), device=dev))     # 20 is not a multiple of 8


def test_encode_first_stage_consumes_rng_like_reference(dev):
    """LatentDiffusion.encode_first_stage (ddpm3d.py:683-707, test_geo4d.py:110-113): same seed -> same sampled z_video as the
    reference (posterior noise is drawn on the CPU generator frame by frame because perframe_ae=True)."""
    g = load("vae_encode_tiny.pt")
    m, _, _ = _diffusion(dev, "f32")
    from geo4d_amd.pipeline import get_latent_z
    torch.manual_seed(g["seed"])
    z = get_latent_z(m, g["video"].to(dev))
    e = rel(z, g["z_first_stage"])
    print(f"[encode_first_stage] rel_l2 vs reference = {e:.3e}")
    assert z.shape == g["z_first_stage"].shape and e < 2e-4


def _diffusion(dev, mode):
    from geo4d_amd.diffusion import LatentVisualDiffusion
    u, v = load("unet_tiny.pt"), load("vae_tiny.pt")
    vae_cfg = {"target": "geo4d_amd.vae.AutoencoderKL",
               "params": dict(ddconfig=v["ddconfig"], lossconfig={"target": "torch.nn.Identity"}, embed_dim=4,
                              adaptorconfig=v["adaptorconfig"], compute_dtype=mode)}
    m = LatentVisualDiffusion(unet_config={"target": "geo4d_amd.unet.UNetModel", "params": dict(u["unet_config"], compute_dtype=mode)},
                              first_stage_config=vae_cfg, parameterization="v", conditioning_key="hybrid",
                              rescale_betas_zero_snr=True, linear_start=0.00085, linear_end=0.012, use_dynamic_rescale=True,
                              base_scale=0.7, scale_factor=0.18215, perframe_ae=True, modality="pc_ray_cross_depth", channels=16)
    m.model.diffusion_model.load_state_dict(seeded_state_dict(u["shapes"]), strict=True)
    m.first_stage_model.load_state_dict(seeded_state_dict(v["shapes"]), strict=True)
    return m.to(dev), u, v


@pytest.mark.parametrize("mode,tol,graph", [("f32", 2e-4, False), ("f32", 2e-4, True), ("bf16x3", 2e-4, True), ("bf16x3m", 1e-3, True), ("bf16", 1e-1, True)])
def test_ddim_sampler_vs_reference_golden(dev, mode, tol, graph):
    """Same call as test_geo4d.py:212-227; golden = reference DDIMSampler + LatentDiffusion.apply_model, S=4, eta 0."""
    from geo4d_amd.ddim import DDIMSampler
    g = load("ddim_tiny.pt")
    m, _, _ = _diffusion(dev, mode)
    cond = {"c_crossattn": [g["context"].to(dev)], "c_concat": [g["c_concat"].to(dev)]}
    s = DDIMSampler(m, use_graph=graph)
    out, inter = s.sample(S=g["S"], conditioning=cond, batch_size=1, shape=list(g["x_T"].shape[1:]), verbose=False,
                          unconditional_guidance_scale=1.0, unconditional_conditioning=None, eta=0.0, cfg_img=None, mask=None,
                          x0=None, fs=g["fs"].to(dev), x_T=g["x_T"].to(dev), timestep_spacing="uniform_trailing",
                          guidance_rescale=0.7, unconditional_conditioning_img_nonetext=None)
    e = rel(out, g["samples"])
    print(f"[ddim] mode={mode} graph={graph} rel_l2 vs reference = {e:.3e} (tol {tol:.0e})")
    assert s.ts_table.tolist() == [249, 499, 749, 999]
    assert e < tol
    ray = m.decode_first_stage(g["samples"][:, 4:8].to(dev))
    e2 = rel(ray, g["decode_first_stage_4_8"])
    print(f"[decode_first_stage] mode={mode} rel_l2 vs reference = {e2:.3e}")
    assert ray.shape == g["decode_first_stage_4_8"].shape and e2 < max(tol, 2e-4) * 2


@pytest.mark.parametrize("mode,tol", [("f32", 2e-4), ("bf16x3", 2e-4), ("bf16x3m", 1e-3)])
def test_ddim_mask_blending_vs_reference_golden(dev, mode, tol):
    """mask / x0 (ddim.py:173-180, clean_cond): golden = the reference DDIMSampler on the tiny LatentDiffusion (generate.py ddim_mask).
    The masked region ends one step away from x0, the rest is sampled; an all-zero mask reproduces the unmasked sampler bit for bit."""
    from geo4d_amd.ddim import DDIMSampler
    g = load("ddim_mask_tiny.pt")
    m, _, _ = _diffusion(dev, mode)
    cond = {"c_crossattn": [g["context"].to(dev)], "c_concat": [g["c_concat"].to(dev)]}
    kw = dict(S=g["S"], conditioning=cond, batch_size=1, shape=list(g["x_T"].shape[1:]), verbose=False, unconditional_guidance_scale=1.0,
              unconditional_conditioning=None, eta=0.0, cfg_img=None, fs=g["fs"].to(dev), x_T=g["x_T"].to(dev),
              timestep_spacing="uniform_trailing", guidance_rescale=0.7, unconditional_conditioning_img_nonetext=None)
    out, _ = DDIMSampler(m).sample(mask=g["mask"].to(dev), x0=g["x0"].to(dev), clean_cond=True, **kw)
    e = rel(out, g["samples"])
    print(f"[ddim mask] mode={mode} rel_l2 vs reference = {e:.3e}")
    assert e < tol
    plain, _ = DDIMSampler(m, use_graph=False).sample(**kw)
    zero, _ = DDIMSampler(m).sample(mask=torch.zeros_like(g["mask"]).to(dev), x0=g["x0"].to(dev), clean_cond=True, **kw)
    assert torch.equal(plain, zero)
    with pytest.raises(ValueError):
        DDIMSampler(m).sample(mask=g["mask"].to(dev), x0=None, **kw)


@pytest.mark.parametrize("mode,tol", [("f32", 1e-3), ("bf16x3", 1e-3), ("bf16x3m", 1e-3), ("f16", 2e-2), ("bf16", 1e-1)])
def test_window_end_to_end_vs_oracle(dev, mode, tol):
    """One window: DDIM (S=3, eta 0, uniform_trailing) + 4-modality decode, HIP vs oracle on identical inputs.
    North-star bar: point-map relative L2 <= 1e-3 — asserted, un-loosened, for the exact f32 mode, for bf16x3 AND for bf16x3m, the
    mode bench.py quotes its number in (this tiny random-weight config is its worst case: ~5e-4 here, 1.1e-4 over 50 steps at
    BASELINE size, tests/test_fullsize_gpu.py); f16 / bf16 are the reported fast modes."""
    from geo4d_amd.pipeline import image_guided_synthesis, postprocess_window
    from geo4d_amd.vae import AutoencoderKL
    m, u, v = _diffusion(dev, mode)
    pv_sd = seeded_state_dict({k: s for k, s in v["shapes"].items()}, gain=0.9)   # a second, different "fine-tuned" VAE
    pvae = AutoencoderKL(ddconfig=v["ddconfig"], lossconfig=None, embed_dim=4, adaptorconfig=v["adaptorconfig"], compute_dtype=mode)
    pvae.load_state_dict(pv_sd, strict=True)
    pvae = pvae.to(dev)
    gen = torch.Generator().manual_seed(777)
    B, T, h, w = 1, 16, 8, 8
    x_T = torch.randn((B, 16, T, h, w), generator=gen)
    ctx = torch.randn((B, 77 + 16 * T, u["unet_config"]["context_dim"]), generator=gen)
    zc = torch.randn((B, 4, T, h, w), generator=gen)
    fs = 24
    cond = {"c_crossattn": [ctx.to(dev)], "c_concat": [zc.to(dev)]}
    out = image_guided_synthesis(m, [""], None, [B, 16, T, h, w], 1, 3, 0.0, 1.0, None, fs, True, False, False, False,
                                 "uniform_trailing", 0.7, pointmap_vae=pvae, cond=cond, x_T=x_T.to(dev))
    assert out.shape == (B, 1, 11, T, 8 * h, 8 * w)
    # oracle
    usd, vsd = seeded_state_dict(u["shapes"]), seeded_state_dict(v["shapes"])

    def apply_model(x, t):
        return ounet.unet_forward(usd, u["unet_config"], torch.cat([x, zc], 1), t, ctx, torch.tensor([fs]))
    ref_lat = oddim.ddim_sample(apply_model, oddim.make_schedule(), oddim.make_scale_arr(), 3, x_T, eta=0.0)
    ref = opipe.decode_modalities(vsd, pv_sd, v["ddconfig"], v["adaptorconfig"], ref_lat)
    got = out[:, 0].cpu()
    e_pts, e_all = rel(got[:, 0:3], ref[:, 0:3]), rel(got, ref)
    print(f"[window e2e] mode={mode} point-map rel_l2 = {e_pts:.3e}  all 11 channels = {e_all:.3e} (tol {tol:.0e})")
    assert e_pts < tol and e_all < tol * 2
    if mode in ("f32", "bf16x3"):
        po, pr = postprocess_window(out[:, 0]), opipe.postprocess_window(ref)
        flips = (po["valid"].cpu() != ~pr["invalid"]).float().mean().item()
        assert flips < 1e-3 and rel(po["inverse_depthmap"], pr["inverse_depthmap"]) < 1e-3


def test_cfg_path_matches_reference_formula(dev):
    """unconditional_guidance_scale != 1 (test_geo4d.py:171-188, ddim.py:218-231): two U-Net evaluations per step combined
    as e_u + s (e_c - e_u) and rescaled by utils_diffusion.rescale_noise_cfg. Checked against the oracle restatement."""
    from geo4d_amd.ddim import DDIMSampler
    m, u, _ = _diffusion(dev, "f32")
    gen = torch.Generator().manual_seed(99)
    B, T, h, w = 1, 4, 8, 8
    x_T = torch.randn((B, 16, T, h, w), generator=gen)
    zc = torch.randn((B, 4, T, h, w), generator=gen)
    ctx_c = torch.randn((B, 77 + 16 * T, u["unet_config"]["context_dim"]), generator=gen)
    ctx_u = torch.randn((B, 77 + 16 * T, u["unet_config"]["context_dim"]), generator=gen)
    cond = {"c_crossattn": [ctx_c.to(dev)], "c_concat": [zc.to(dev)]}
    uc = {"c_crossattn": [ctx_u.to(dev)], "c_concat": [zc.to(dev)]}
    out, _ = DDIMSampler(m).sample(S=3, conditioning=cond, batch_size=B, shape=[16, T, h, w], verbose=False, eta=0.0,
                                   unconditional_guidance_scale=7.5, unconditional_conditioning=uc, fs=torch.tensor([24], device=dev),
                                   x_T=x_T.to(dev), timestep_spacing="uniform_trailing", guidance_rescale=0.7)
    usd = seeded_state_dict(u["shapes"])

    def apply_model(x, t):
        fs = torch.tensor([24])
        e_c = ounet.unet_forward(usd, u["unet_config"], torch.cat([x, zc], 1), t, ctx_c, fs)
        e_u = ounet.unet_forward(usd, u["unet_config"], torch.cat([x, zc], 1), t, ctx_u, fs)
        o = e_u + 7.5 * (e_c - e_u)
        dims = list(range(1, o.ndim))
        resc = o * (e_c.std(dim=dims, keepdim=True) / o.std(dim=dims, keepdim=True))
        return 0.7 * resc + 0.3 * o
    ref = oddim.ddim_sample(apply_model, oddim.make_schedule(), oddim.make_scale_arr(), 3, x_T, eta=0.0)
    e = rel(out, ref)
    print(f"[cfg 7.5 + rescale 0.7] rel_l2 vs oracle = {e:.3e}")
    assert e < 2e-4


def test_multicond_guidance_and_captured_cfg_step(dev):
    """ddim_multiplecond.py:229-236: out = e_u + cfg_img (e_i - e_u) + s (e_c - e_i), then rescale_noise_cfg — against the
    oracle DDIM loop driven with that formula. The guided step (3 U-Net evaluations + combination) is hipGraph-captured;
    the captured and the eager sampler must agree bit for bit."""
    from geo4d_amd.ddim_multiplecond import DDIMSampler as Multi
    m, u, _ = _diffusion(dev, "f32")
    gen = torch.Generator().manual_seed(314)
    B, T, h, w = 1, 4, 8, 8
    cd = u["unet_config"]["context_dim"]
    x_T = torch.randn((B, 16, T, h, w), generator=gen)
    zc = torch.randn((B, 4, T, h, w), generator=gen)
    ctx = [torch.randn((B, 77 + 16 * T, cd), generator=gen) for _ in range(3)]      # cond, uncond, image-yes/text-""
    mk = lambda c: {"c_crossattn": [c.to(dev)], "c_concat": [zc.to(dev)]}
    kw = dict(S=4, conditioning=mk(ctx[0]), batch_size=B, shape=[16, T, h, w], verbose=False, eta=0.0,
              unconditional_guidance_scale=7.5, unconditional_conditioning=mk(ctx[1]), cfg_img=2.0,
              unconditional_conditioning_img_nonetext=mk(ctx[2]), fs=torch.tensor([24], device=dev), x_T=x_T.to(dev),
              timestep_spacing="uniform_trailing", guidance_rescale=0.7)
    out_g, _ = Multi(m, use_graph=True).sample(**kw)
    out_e, _ = Multi(m, use_graph=False).sample(**kw)
    assert torch.equal(out_g, out_e), "captured guided step differs from the eager one"
    usd = seeded_state_dict(u["shapes"])

    def apply_model(x, t):
        fs = torch.tensor([24])
        e_c, e_u, e_i = (ounet.unet_forward(usd, u["unet_config"], torch.cat([x, zc], 1), t, c, fs) for c in ctx)
        o = e_u + 2.0 * (e_i - e_u) + 7.5 * (e_c - e_i)
        dims = list(range(1, o.ndim))
        resc = o * (e_c.std(dim=dims, keepdim=True) / o.std(dim=dims, keepdim=True))
        return 0.7 * resc + 0.3 * o
    ref = oddim.ddim_sample(apply_model, oddim.make_schedule(), oddim.make_scale_arr(), 4, x_T, eta=0.0)
    e = rel(out_g, ref)
    print(f"[3-way cfg 7.5 / img 2.0 + rescale 0.7] rel_l2 vs oracle = {e:.3e}")
    assert e < 2e-4
    with pytest.raises(ValueError):
        Multi(m).sample(**dict(kw, unconditional_conditioning_img_nonetext=None))


def test_synthesis_encodes_video_when_c_concat_is_missing(dev):
    """image_guided_synthesis computes z_video = get_latent_z(model, videos) itself (test_geo4d.py:159-170) when the caller
    passes only the cross-attention context; same seed => same result as passing that latent explicitly."""
    from geo4d_amd.pipeline import get_latent_z, image_guided_synthesis
    m, u, _ = _diffusion(dev, "f32")
    gen = torch.Generator().manual_seed(27)
    B, T = 1, 4
    videos = (torch.rand((B, 3, T, 64, 64), generator=gen) * 2 - 1).to(dev)
    ctx = torch.randn((B, 77 + 16 * T, u["unet_config"]["context_dim"]), generator=gen).to(dev)
    x_T = torch.randn((B, 16, T, 8, 8), generator=gen).to(dev)
    kw = dict(n_samples=1, ddim_steps=3, ddim_eta=0.0, fs=24, timestep_spacing="uniform_trailing", guidance_rescale=0.7, x_T=x_T)
    torch.manual_seed(5)
    a = image_guided_synthesis(m, [""], videos, [B, 16, T, 8, 8], cond={"c_crossattn": [ctx]}, **kw)
    torch.manual_seed(5)
    z = get_latent_z(m, videos)
    b = image_guided_synthesis(m, [""], videos, [B, 16, T, 8, 8], cond={"c_crossattn": [ctx], "c_concat": [z]}, **kw)
    assert a.shape == (B, 1, 11, T, 64, 64) and torch.isfinite(a).all()
    assert torch.equal(a, b)


def test_run_clip_windows(dev):
    """run_clip = the window loop of run_inference (test_geo4d.py:396-443) on the HIP path: 20 frames -> windows (0,16),
    (4,20) and the always-appended tail (4,20); deterministic; window 0 equals a direct image_guided_synthesis call with that
    window's seeds."""
    from geo4d_amd.pipeline import image_guided_synthesis, run_clip
    m, u, _ = _diffusion(dev, "bf16")
    gen = torch.Generator().manual_seed(8)
    video = (torch.rand((1, 3, 20, 64, 64), generator=gen) * 2 - 1).to(dev)
    ctx = torch.randn((1, 77 + 16 * 16, u["unet_config"]["context_dim"]), generator=gen).to(dev)
    kw = dict(ddim_steps=3, seed=123)
    slices, maps = run_clip(m, video, ctx, **kw)
    _, maps2, traj = run_clip(m, video, lambda frames: ctx, with_cameras=True, **kw)
    assert traj.shape == (3, 16, 4, 4) and torch.isfinite(traj).all()
    Rm = traj[:, :, :3, :3].double()
    assert torch.allclose(Rm @ Rm.transpose(-1, -2), torch.eye(3, dtype=torch.float64, device=dev).expand_as(Rm), atol=1e-5)
    assert [(s.start, s.stop) for s in slices] == [(0, 16), (4, 20), (4, 20)]
    assert maps.shape == (3, 11, 16, 64, 64) and torch.isfinite(maps).all() and torch.equal(maps, maps2)
    assert not torch.equal(maps[1], maps[2])            # same frames, different window index => different noise
    wseed = 123 * 1000003 + 0
    x_T = torch.randn([1, 16, 16, 8, 8], generator=torch.Generator().manual_seed(wseed)).to(dev)
    torch.manual_seed(wseed)
    direct = image_guided_synthesis(m, [""], video[:, :, 0:16], [1, 16, 16, 8, 8], n_samples=1, ddim_steps=3, ddim_eta=0.0, fs=24,
                                    timestep_spacing="uniform_trailing", guidance_rescale=0.7, cond={"c_crossattn": [ctx]}, x_T=x_T)
    assert torch.equal(direct[:, 0], maps[0:1])


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16x3", 2e-4), ("bf16x3m", 1e-3)])
def test_run_clip_window_batch_matches_one_window_at_a_time(dev, mode, tol):
    """Round 6: run_clip(window_batch = 2) denoises and decodes two of a rank's windows as one batch (levels 1-3 of the U-Net cannot fill
    the chip at one window). Per-window noise / VAE-encode sampling / conditioning are unchanged, so every window's maps equal the
    one-at-a-time run up to what GEMM tile choice (it follows M) does to fp32 summation order - the mode's own rounding level; the
    cameras follow; a ragged last group (3 windows: 2 + 1) is covered."""
    from geo4d_amd.pipeline import run_clip
    m, u, _ = _diffusion(dev, mode)
    gen = torch.Generator().manual_seed(18)
    video = (torch.rand((1, 3, 22, 64, 64), generator=gen) * 2 - 1).to(dev)
    ctx = torch.randn((1, 77 + 16 * 16, u["unet_config"]["context_dim"]), generator=gen).to(dev)
    kw = dict(ddim_steps=3, seed=77, with_cameras=True)
    slices, one, traj1 = run_clip(m, video, ctx, window_batch=1, **kw)
    _, two, traj2 = run_clip(m, video, lambda frames: ctx, window_batch=2, **kw)
    assert [(s.start, s.stop) for s in slices] == [(0, 16), (4, 20), (6, 22)] and one.shape == two.shape == (3, 11, 16, 64, 64)
    errs = [rel(two[i], one[i]) for i in range(3)]
    print(f"[run_clip window_batch 2 vs 1] mode={mode} per-window rel_l2 {['%.2e' % e for e in errs]} (tol {tol:.0e})")
    assert max(errs) < tol and torch.isfinite(two).all()
    assert traj2.shape == (3, 16, 4, 4) and torch.allclose(traj2, traj1, atol=50 * tol)


def test_plucker_cameras_vs_reference_golden(dev):
    """raymap_to_camera_matrix on the device (csrc/rays.hip) vs matrices produced by the reference's own
    functions (fixtures), vs the fp64 oracle on a larger window taken as channel views of a decoded [1,11,T,H,W] tensor, and on
    a square frame (where the reference itself raises)."""
    from geo4d_amd.rays import cameras_from_plucker, raymap_to_camera_matrix
    from oracle import rays as orays
    for name, c in load("rays.pt").items():
        P = raymap_to_camera_matrix(c["raymap"].to(dev), c["crossmap"].to(dev))
        err = (P.cpu() - c["P_c2w"]).abs().max().item()
        print(f"[plucker {name}] max abs err vs reference = {err:.2e}")
        assert P.shape == c["P_c2w"].shape and err < 5e-5
    for (T, H, W) in ((16, 64, 96), (16, 40, 40)):
        gen = torch.Generator().manual_seed(T + H)
        # synthetic smooth path: directions through a pinhole + noise (same recipe as the fixture generator, inlined)
        ys, xs = torch.meshgrid(torch.linspace(-0.5, 0.5, H), torch.linspace(-0.8, 0.8, W), indexing="ij")
        cam = torch.nn.functional.normalize(torch.stack([xs, ys, torch.ones_like(xs)], -1), dim=-1)
        maps = torch.zeros((1, 11, T, H, W))
        for t in range(T):
            a = torch.tensor(0.04 * t)
            R = torch.tensor([[torch.cos(a), 0, torch.sin(a)], [0, 1, 0], [-torch.sin(a), 0, torch.cos(a)]])
            cc = torch.tensor([0.1 * t, -0.03 * t, 0.02 * t * t])
            d = cam @ R.T
            maps[0, 4:7, t] = (d * 1.3 + 0.01 * torch.randn(d.shape, generator=gen)).permute(2, 0, 1)
            maps[0, 7:10, t] = (torch.cross(cc.expand_as(d), d, dim=-1) + 0.01 * torch.randn(d.shape, generator=gen)).permute(2, 0, 1)
        md = maps.to(dev)
        P = raymap_to_camera_matrix(md[:, 4:7], md[:, 7:10])
        ref = orays.raymap_to_camera_matrix(maps[:, 4:7], maps[:, 7:10])
        err = (P.cpu().double() - ref).abs().max().item()
        print(f"[plucker {T}x{H}x{W}] max abs err vs oracle = {err:.2e}")
        assert err < 5e-5
        R_, T_, c_ = cameras_from_plucker(md[:, 4:7], md[:, 7:10])
        assert torch.allclose(c_.cpu().double(), ref[:, :3, 3], atol=5e-5) and R_.shape == (T, 3, 3) and T_.shape == (T, 3)
    with pytest.raises(RuntimeError):
        raymap_to_camera_matrix(torch.zeros((1, 3, 2, 8, 11), device=dev), torch.zeros((1, 3, 2, 8, 11), device=dev))   # odd |H - W|


def test_guided_samplers_vs_reference_golden(dev):
    """2-way (geo4d_amd.ddim) and 3-way (geo4d_amd.ddim_multiplecond) guidance + guidance_rescale vs outputs of the
    reference's own lvdm.models.samplers.ddim / ddim_multiplecond classes on the same inputs (tests/golden/ddim_cfg_tiny.pt)."""
    from geo4d_amd.ddim import DDIMSampler
    from geo4d_amd.ddim_multiplecond import DDIMSampler as Multi
    g = load("ddim_cfg_tiny.pt")
    # round 6: the reduced-precision modes on the guidance fixtures too. Guidance MULTIPLIES rounding error: out = e_u + s (e_c - e_u) with s = 7.5
    # weighs the two forwards' errors by |1 - s| + s = 14, so on this (LATENT, tiny worst-case config) fixture bf16x3 lands at ~1e-4 and the
    # headline mode at ~2e-3 - reported and bounded at 5e-3, NOT claimed to meet the CFG = 1 benchmark's 1e-3 point-map bar: the strict
    # bf16x3 mode is the one to run guided sampling in when 1e-3 is required
    for mode, tol in (("f32", 2e-4), ("bf16x3", 1e-3), ("bf16x3m", 5e-3)):
        m, u, _ = _diffusion(dev, mode)
        mk = lambda c: {"c_crossattn": [c.to(dev)], "c_concat": [g["c_concat"].to(dev)]}
        c_c, c_u, c_i = g["contexts"]
        kw = dict(S=g["S"], conditioning=mk(c_c), batch_size=1, shape=list(g["x_T"].shape[1:]), verbose=False, eta=0.0,
                  unconditional_guidance_scale=g["scale"], unconditional_conditioning=mk(c_u), fs=g["fs"].to(dev), x_T=g["x_T"].to(dev),
                  timestep_spacing="uniform_trailing", guidance_rescale=g["guidance_rescale"])
        two, _ = DDIMSampler(m).sample(cfg_img=None, unconditional_conditioning_img_nonetext=None, **kw)
        three, _ = Multi(m).sample(cfg_img=g["cfg_img"], unconditional_conditioning_img_nonetext=mk(c_i), **kw)
        e2, e3 = rel(two, g["samples_2way"]), rel(three, g["samples_3way"])
        print(f"[guided samplers] mode={mode} 2-way {e2:.3e} 3-way {e3:.3e} vs reference (tol {tol:.0e})")
        assert e2 < tol and e3 < tol


def test_stochastic_ddim_matches_oracle_on_the_same_noise(dev):
    """eta = 1: the sampler draws one torch.randn(size, device) per step exactly where the reference does (ddim.py:271), so the
    device RNG stream can be recorded up front and replayed into the oracle loop (itself pinned to the reference at eta = 1 by
    tests/golden/ddim_eta_tiny.pt): sigma_t, the sqrt(1 - a_prev - sigma_t^2) direction term and the noise term of the fused
    update kernel are checked end to end."""
    from geo4d_amd.ddim import DDIMSampler
    g = load("ddim_eta_tiny.pt")
    m, u, _ = _diffusion(dev, "f32")
    size = tuple(g["x_T"].shape)
    torch.manual_seed(4321)
    recorded = [torch.randn(size, device=dev).cpu() for _ in range(g["S"])]
    torch.manual_seed(4321)
    cond = {"c_crossattn": [g["context"].to(dev)], "c_concat": [g["c_concat"].to(dev)]}
    out, _ = DDIMSampler(m).sample(S=g["S"], conditioning=cond, batch_size=1, shape=list(size[1:]), verbose=False, eta=g["eta"],
                                   unconditional_guidance_scale=1.0, unconditional_conditioning=None, fs=g["fs"].to(dev),
                                   x_T=g["x_T"].to(dev), timestep_spacing="uniform_trailing", guidance_rescale=0.7)
    usd = seeded_state_dict(u["shapes"])
    it = iter(recorded)

    def apply_model(x, t):
        return ounet.unet_forward(usd, g["unet_config"], torch.cat([x, g["c_concat"]], 1), t, g["context"], g["fs"])
    ref = oddim.ddim_sample(apply_model, oddim.make_schedule(), oddim.make_scale_arr(), g["S"], g["x_T"], eta=g["eta"],
                            noise_fn=lambda shape: next(it))
    e = rel(out, ref)
    print(f"[ddim eta=1, replayed device noise] rel_l2 vs oracle = {e:.3e}")
    assert e < 2e-4


def test_stochastic_ddim_runs(dev):
    """eta > 0 draws torch noise per step (no hipGraph); RNG streams differ from the CPU reference, so only sanity here."""
    from geo4d_amd.ddim import DDIMSampler
    m, u, _ = _diffusion(dev, "bf16")
    B, T, h, w = 1, 4, 8, 8
    cond = {"c_crossattn": [torch.randn((B, 77 + 16 * T, u["unet_config"]["context_dim"]), device=dev)],
            "c_concat": [torch.randn((B, 4, T, h, w), device=dev)]}
    out, _ = DDIMSampler(m).sample(S=3, conditioning=cond, batch_size=B, shape=[16, T, h, w], verbose=False, eta=1.0,
                                   fs=torch.tensor([24], device=dev), timestep_spacing="uniform_trailing")
    assert out.shape == (B, 16, T, h, w) and torch.isfinite(out).all()


def test_rccl_path_executes_on_one_gpu(dev):
    """The `nccl` (= RCCL) code path of geo4d_amd.dist on a world-size-1 group: init, the window all-gather (sync + async) and the
    frame all-gather really call ncclAllGather on device tensors (the 8-GPU run is the driver's; multi-rank logic is covered
    by the gloo world-2 tests in tests/test_dist_cpu.py)."""
    import socket
    import torch.distributed as dist
    from geo4d_amd import dist as gd
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        maps = torch.randn((2, 11, 4, 16, 24), device=dev)
        full = gd.all_gather_windows(maps, 2, force=True)
        pend = gd.all_gather_windows(maps, 2, async_op=True, force=True)
        frames = gd.all_gather_frames(maps, 4, dim=2, force=True)
        pf = gd.all_gather_frames(maps, 4, dim=2, async_op=True, force=True)
        torch.cuda.synchronize()
        assert torch.equal(full, maps) and torch.equal(pend.wait(), maps) and torch.equal(frames, maps) and torch.equal(pf.wait(), maps)
        t = torch.ones(3, device=dev)
        dist.all_reduce(t)
        assert t.tolist() == [1.0, 1.0, 1.0]
    finally:
        dist.destroy_process_group()


def test_captured_step_is_reused_across_windows_and_survives_model_changes(dev):
    """The sampler owns static copies of the conditioning: a second sample() with NEW conditioning values of the same shapes
    replays the captured step (context K/V refreshed in place) and must equal a fresh sampler bit for bit; after the weights
    change (load_state_dict -> re-pack) the capture is dropped instead of replaying over freed tensors."""
    from geo4d_amd.ddim import DDIMSampler
    m, u, _ = _diffusion(dev, "bf16")
    gen = torch.Generator().manual_seed(41)
    B, T, h, w = 1, 4, 8, 8
    cd = u["unet_config"]["context_dim"]
    mk = lambda: {"c_crossattn": [torch.randn((B, 77 + 16 * T, cd), generator=gen).to(dev)], "c_concat": [torch.randn((B, 4, T, h, w), generator=gen).to(dev)]}
    cond_a, cond_b = mk(), mk()
    x_T = torch.randn((B, 16, T, h, w), generator=gen).to(dev)
    kw = dict(S=4, batch_size=B, shape=[16, T, h, w], verbose=False, eta=0.0, fs=torch.tensor([24], device=dev), x_T=x_T,
              timestep_spacing="uniform_trailing", unconditional_conditioning_img_nonetext=None)
    s = DDIMSampler(m)
    a1, _ = s.sample(conditioning=cond_a, **kw)
    g_first = s._static["g"]
    b1, _ = s.sample(conditioning=cond_b, **kw)                       # same shapes, new values: replay
    assert s._static["g"] is g_first, "the captured step should have been reused"
    b_ref, _ = DDIMSampler(m).sample(conditioning=cond_b, **kw)
    a_ref, _ = DDIMSampler(m, use_graph=False).sample(conditioning=cond_a, **kw)
    assert torch.equal(b1, b_ref) and torch.equal(a1, a_ref) and not torch.equal(a1, b1)
    sd = {k: v * 1.01 for k, v in m.model.diffusion_model.state_dict().items()}
    m.model.diffusion_model.load_state_dict(sd)                       # re-pack: the old capture's weight pointers are gone
    c1, _ = s.sample(conditioning=cond_b, **kw)
    assert s._static["g"] is not g_first, "a capture must not outlive the packed weights it baked in"
    c_ref, _ = DDIMSampler(m, use_graph=False).sample(conditioning=cond_b, **kw)
    assert torch.equal(c1, c_ref) and not torch.equal(c1, b1)


def test_guided_synthesis_encodes_the_latent_for_every_branch(dev):
    """CFG through image_guided_synthesis: the caller supplies only the cross-attention contexts; the video latent
    computed from `videos` must reach the unconditional (and image-only) branches too (test_geo4d.py:184-195) — equal to a
    direct sampler call with hand-built dicts."""
    from geo4d_amd.ddim_multiplecond import DDIMSampler as Multi
    from geo4d_amd.pipeline import decode_modalities, get_latent_z, image_guided_synthesis
    m, u, _ = _diffusion(dev, "f32")
    gen = torch.Generator().manual_seed(43)
    B, T = 1, 4
    cd = u["unet_config"]["context_dim"]
    videos = (torch.rand((B, 3, T, 64, 64), generator=gen) * 2 - 1).to(dev)
    ctx = [torch.randn((B, 77 + 16 * T, cd), generator=gen).to(dev) for _ in range(3)]
    x_T = torch.randn((B, 16, T, 8, 8), generator=gen).to(dev)
    torch.manual_seed(9)
    out = image_guided_synthesis(m, [""], videos, [B, 16, T, 8, 8], n_samples=1, ddim_steps=3, ddim_eta=0.0, unconditional_guidance_scale=7.5,
                                 cfg_img=2.0, fs=24, multiple_cond_cfg=True, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                 cond={"c_crossattn": [ctx[0]]}, x_T=x_T, unconditional_conditioning={"c_crossattn": [ctx[1]]},
                                 unconditional_conditioning_img_nonetext={"c_crossattn": [ctx[2]]})
    torch.manual_seed(9)
    z = get_latent_z(m, videos)
    mk = lambda c: {"c_crossattn": [c], "c_concat": [z]}
    lat, _ = Multi(m).sample(S=3, conditioning=mk(ctx[0]), batch_size=B, shape=[16, T, 8, 8], verbose=False, eta=0.0,
                             unconditional_guidance_scale=7.5, unconditional_conditioning=mk(ctx[1]), cfg_img=2.0,
                             unconditional_conditioning_img_nonetext=mk(ctx[2]), fs=torch.tensor([24], device=dev), x_T=x_T,
                             timestep_spacing="uniform_trailing", guidance_rescale=0.7)
    assert torch.isfinite(out).all() and torch.equal(out[:, 0], decode_modalities(m, lat, None))
