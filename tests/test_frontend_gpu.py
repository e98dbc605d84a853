"""Conditioning front-end on the HIP kernels against tests/golden/clip_tiny.pt:
outputs of the reference Resampler class and of HuggingFace transformers' CLIP text / vision models on the same open_clip-format
weights (tests/golden/generate.py frontend). Tolerances as in tests/test_parity_gpu.py."""
import os

import pytest
import torch

from oracle.params import seeded_state_dict

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# bf16x3m (round 6): the front-end has no two-pass class (precision.TWO_PASS_CLASSES live in the U-Net and the VAE decoder), so the headline mode
# runs bf16x3 arithmetic here and is held to bf16x3's tolerance
MODES = [("f32", 2e-4), ("bf16x3", 2e-4), ("bf16x3m", 2e-4), ("f16", 1e-2), ("bf16", 5e-2)]


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def fix():
    return torch.load(os.path.join(G, "clip_tiny.pt"), weights_only=False)


@pytest.mark.parametrize("mode,tol", MODES)
def test_resampler_vs_reference(fix, dev, mode, tol):
    from geo4d_amd.encoders import Resampler
    r = fix["resampler"]
    m = Resampler(**r["cfg"], compute_dtype=mode)
    m.load_state_dict(seeded_state_dict(r["shapes"]), strict=True)
    m = m.to(dev)
    e3, e4 = rel(m(r["x3"].to(dev)), r["y3"]), rel(m(r["x4"].to(dev)), r["y4"])
    print(f"[resampler] mode={mode} 3-D input {e3:.3e}  per-frame 4-D input {e4:.3e} (tol {tol:.0e})")
    assert e3 < tol and e4 < tol


@pytest.mark.parametrize("mode,tol", MODES)
def test_openclip_towers_vs_hf_transformers(fix, dev, mode, tol):
    from geo4d_amd.encoders import FrozenOpenCLIPEmbedder, FrozenOpenCLIPImageEmbedderV2
    c = fix["clip"]
    sd = seeded_state_dict({**c["text_shapes"], **c["vision_shapes"]})
    W = c["text_shapes"]["positional_embedding"][1]
    VW = c["vision_shapes"]["visual.class_embedding"][0]
    nl = 1 + max(int(k.split(".")[2]) for k in c["text_shapes"] if k.startswith("transformer.resblocks."))
    nv = 1 + max(int(k.split(".")[3]) for k in c["vision_shapes"] if k.startswith("visual.transformer.resblocks."))
    grid = int(round((c["vision_shapes"]["visual.positional_embedding"][0] - 1) ** 0.5))
    for layer, key in (("penultimate", "text_penultimate"), ("last", "text_last")):
        t = FrozenOpenCLIPEmbedder(layer=layer, width=W, layers=nl, heads=c["text_heads"], vocab_size=c["text_shapes"]["token_embedding.weight"][0],
                                   compute_dtype=mode)
        own = t.state_dict()
        t.load_state_dict({**own, **{"model." + k: v for k, v in sd.items() if not k.startswith("visual.")}}, strict=True)
        t = t.to(dev)
        e = rel(t(c["tokens"].to(dev)), c[key])
        print(f"[clip text {layer}] mode={mode} rel_l2 vs HF = {e:.3e} (tol {tol:.0e})")
        assert e < tol
    v = FrozenOpenCLIPImageEmbedderV2(width=VW, layers=nv, heads=c["vision_heads"], image_size=14 * grid, patch_size=14, compute_dtype=mode)
    own = v.state_dict()
    v.load_state_dict({**own, **{"model." + k: val for k, val in sd.items() if k.startswith("visual.")}}, strict=True)
    v = v.to(dev)
    e = rel(v.encode_with_vision_transformer(c["pixels"].to(dev), preprocess=False), c["vision_tokens"])
    print(f"[clip vision, head width 80] mode={mode} rel_l2 vs HF = {e:.3e} (tol {tol:.0e})")
    assert e < tol
    z = v(torch.zeros((1, 3, 40, 64), device=dev))          # the shipped path: a zero image of any size -> constant pixels
    assert z.shape == (1, 1 + grid * grid, VW) and torch.isfinite(z).all()


def test_synthesis_builds_its_own_context(dev):
    """image_guided_synthesis without `cond`: text tower on "" + Resampler over the image tower's tokens of a zero image
    (test_geo4d.py:124-158), built lazily from the yaml sections; equals passing that context explicitly, and is cached."""
    from geo4d_amd.diffusion import LatentVisualDiffusion
    from geo4d_amd.pipeline import image_guided_synthesis
    u = torch.load(os.path.join(G, "unet_tiny.pt"), weights_only=False)
    v = torch.load(os.path.join(G, "vae_tiny.pt"), weights_only=False)
    cd = u["unet_config"]["context_dim"]
    vae_cfg = {"target": "geo4d_amd.vae.AutoencoderKL", "params": dict(ddconfig=v["ddconfig"], lossconfig=None, embed_dim=4,
                                                                       adaptorconfig=v["adaptorconfig"], compute_dtype="f32")}
    m = LatentVisualDiffusion(
        unet_config={"target": "geo4d_amd.unet.UNetModel", "params": dict(u["unet_config"], compute_dtype="f32")}, first_stage_config=vae_cfg,
        cond_stage_config={"target": "geo4d_amd.encoders.FrozenOpenCLIPEmbedder", "params": dict(layer="penultimate", width=cd, layers=2, heads=2, vocab_size=49408)},
        img_cond_stage_config={"target": "geo4d_amd.encoders.FrozenOpenCLIPImageEmbedderV2", "params": dict(width=160, layers=2, heads=2, image_size=56)},
        image_proj_stage_config={"target": "geo4d_amd.encoders.Resampler", "params": dict(dim=128, depth=1, dim_head=64, heads=2, num_queries=16,
                                                                                          embedding_dim=160, output_dim=cd, video_length=4)},
        parameterization="v", conditioning_key="hybrid", rescale_betas_zero_snr=True, linear_start=0.00085, linear_end=0.012,
        use_dynamic_rescale=True, base_scale=0.7, scale_factor=0.18215, perframe_ae=True, modality="pc_ray_cross_depth", channels=16).to(dev)
    assert "cond_stage_model" not in m._modules                      # nothing built until it is needed
    gen = torch.Generator().manual_seed(3)
    B, T = 1, 4
    videos = (torch.rand((B, 3, T, 64, 64), generator=gen) * 2 - 1).to(dev)
    x_T = torch.randn((B, 16, T, 8, 8), generator=gen).to(dev)
    kw = dict(n_samples=1, ddim_steps=3, ddim_eta=0.0, fs=24, timestep_spacing="uniform_trailing", guidance_rescale=0.7, x_T=x_T)
    torch.manual_seed(5)
    a = image_guided_synthesis(m, ["ignored: text_input is False"], videos, [B, 16, T, 8, 8], **kw)
    assert {"cond_stage_model", "embedder", "image_proj_model"} <= set(m._modules)
    ctx = m.context_for([""], image=videos[:, :, 0])
    assert ctx.shape == (B, 77 + 16 * T, cd)
    torch.manual_seed(5)
    b = image_guided_synthesis(m, [""], videos, [B, 16, T, 8, 8], cond={"c_crossattn": [ctx]}, **kw)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    keys = set(m.state_dict())
    assert "cond_stage_model.model.transformer.resblocks.0.attn.in_proj_weight" in keys and "embedder.model.visual.conv1.weight" in keys \
        and "image_proj_model.layers.0.0.to_kv.weight" in keys          # the reference checkpoint's names
