"""TEST INFRASTRUCTURE ONLY — functional fp32 restatement of the conditioning front-end.

* ``resampler_forward`` follows lvdm/modules/encoders/resampler.py:96-163 (Resampler.forward), :46-93 (PerceiverAttention),
  :27-34 (FeedForward) on a reference-format state_dict. PINNED: tests/golden/clip_tiny.pt holds outputs of the reference
  Resampler class itself (tests/golden/generate.py frontend).
* ``text_transformer_forward`` / ``vision_transformer_forward`` restate what lvdm/modules/encoders/condition.py:174-234
  (FrozenOpenCLIPEmbedder.encode_with_transformer, layer="penultimate") and :295-372 (FrozenOpenCLIPImageEmbedderV2.
  encode_with_vision_transformer) execute inside ``open_clip_torch==2.22.0`` (requirements.txt:22; NOT under /root/reference and
  not installed here): ViT-H-14 towers built from ``ResidualAttentionBlock``s — x = x + attn(ln_1(x)); x = x + mlp(ln_2(x)) with
  nn.MultiheadAttention (in_proj_weight / in_proj_bias / out_proj), mlp = c_fc -> GELU(erf) -> c_proj, a causal additive mask
  in the text tower, conv1 patch embedding (no bias) + class token + positional embedding + ln_pre in the vision tower.
  State_dict names are open_clip's (``token_embedding.weight``, ``positional_embedding``, ``transformer.resblocks.N.*``,
  ``ln_final.*``; ``visual.conv1.weight``, ``visual.class_embedding``, ``visual.positional_embedding``, ``visual.ln_pre.*``,
  ``visual.transformer.resblocks.N.*``). PINNED against an independent implementation of the same published architecture:
  HuggingFace ``transformers`` CLIPTextModel / CLIPVisionModel with the weights mapped name by name (generate.py frontend);
  the open_clip package itself is absent, so the pin is on the architecture, not on that package's code.
"""
import math

import torch
import torch.nn.functional as F


# ---- Resampler ----------------------------------------------------------------------------------------------------------------
def _perceiver_attention(sd, p, x, latents, heads):
    x = F.layer_norm(x, x.shape[-1:], sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
    latents = F.layer_norm(latents, latents.shape[-1:], sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
    b, l, _ = latents.shape
    q = F.linear(latents, sd[p + ".to_q.weight"])
    k, v = F.linear(torch.cat((x, latents), dim=-2), sd[p + ".to_kv.weight"]).chunk(2, dim=-1)
    split = lambda t: t.reshape(b, t.shape[1], heads, -1).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    scale = 1 / math.sqrt(math.sqrt(q.shape[-1]))
    w = torch.softmax(((q * scale) @ (k * scale).transpose(-2, -1)).float(), dim=-1)
    out = (w @ v).permute(0, 2, 1, 3).reshape(b, l, -1)
    return F.linear(out, sd[p + ".to_out.weight"])


@torch.no_grad()
def resampler_forward(sd, x, heads, num_queries, video_length=None, prefix=""):
    """x [B, L, C] (or [B, T, L, C]) image tokens -> [B, Q, D] (or [B, T*q, D]) context tokens."""
    g = lambda n: sd[prefix + n]
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    depth = 1 + max(int(k.split(".")[1]) for k in sub if k.startswith("layers."))
    dim = g("latents").shape[-1]
    four_d = x.dim() == 4
    if four_d:
        B, T, L, C = x.shape
        latents = g("latents").repeat(B, 1, 1).reshape(B * T, num_queries, dim)
        x = x.reshape(B * T, L, C)
    else:
        latents = g("latents").repeat(x.shape[0], 1, 1)
    x = F.linear(x, g("proj_in.weight"), g("proj_in.bias"))
    for i in range(depth):
        latents = _perceiver_attention(sub, f"layers.{i}.0", x, latents, heads) + latents
        h = F.layer_norm(latents, (dim,), sub[f"layers.{i}.1.0.weight"], sub[f"layers.{i}.1.0.bias"])
        latents = F.linear(F.gelu(F.linear(h, sub[f"layers.{i}.1.1.weight"])), sub[f"layers.{i}.1.3.weight"]) + latents
    latents = F.linear(latents, g("proj_out.weight"), g("proj_out.bias"))
    latents = F.layer_norm(latents, latents.shape[-1:], g("norm_out.weight"), g("norm_out.bias"))
    if four_d:
        latents = latents.reshape(B, T * num_queries, -1)
    return latents


# ---- open_clip towers -----------------------------------------------------------------------------------------------------------
def _resblock(sd, p, x, heads, mask=None):
    """open_clip ResidualAttentionBlock (batch-first here): x [B, N, W]."""
    B, N, W = x.shape
    d = W // heads
    h = F.layer_norm(x, (W,), sd[p + ".ln_1.weight"], sd[p + ".ln_1.bias"])
    qkv = F.linear(h, sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"])
    q, k, v = (t.reshape(B, N, heads, d).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    s = q @ k.transpose(-1, -2) * d ** -0.5
    if mask is not None:
        s = s + mask
    a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, N, W)
    x = x + F.linear(a, sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"])
    h = F.layer_norm(x, (W,), sd[p + ".ln_2.weight"], sd[p + ".ln_2.bias"])
    h = F.linear(F.gelu(F.linear(h, sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"])), sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])
    return x + h


def _nblocks(sd, prefix):
    return 1 + max(int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix))


@torch.no_grad()
def text_transformer_forward(sd, tokens, heads, layer="penultimate", prefix=""):
    """tokens int64 [B, 77] -> [B, 77, W] (condition.py:217-234: all blocks but the last `layer_idx`, then ln_final)."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"]
    n = tokens.shape[1]
    mask = torch.full((n, n), float("-inf")).triu_(1)
    nb = _nblocks(sd, "transformer.resblocks.")
    for i in range(nb - (1 if layer == "penultimate" else 0)):
        x = _resblock(sd, f"transformer.resblocks.{i}", x, heads, mask)
    return F.layer_norm(x, x.shape[-1:], sd["ln_final.weight"], sd["ln_final.bias"])


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(x, size=224):
    """condition.py:332-340 without kornia: bicubic resize (align_corners=True) to 224, [-1,1] -> [0,1], CLIP normalisation.
    kornia's `antialias` pre-blur when DOWN-scaling is not reproduced (unpinned dependency); for the shipped config the image
    is all zeros (test_geo4d.py:151-157) and the result is the constant (0.5 - mean) / std whatever the resampling."""
    if x.shape[-2:] != (size, size):
        x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=True)
    x = (x + 1.) / 2.
    mean = torch.tensor(CLIP_MEAN, dtype=x.dtype, device=x.device).reshape(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=x.dtype, device=x.device).reshape(1, 3, 1, 1)
    return (x - mean) / std


@torch.no_grad()
def vision_transformer_forward(sd, image, heads, prefix="visual.", preprocess=True):
    """image [B, 3, H, W] in [-1, 1] -> all tokens [B, 1 + grid^2, W] after the last block (no ln_post / proj): what
    FrozenOpenCLIPImageEmbedderV2.encode_with_vision_transformer returns (condition.py:346-372)."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    patch = sd["conv1.weight"].shape[-1]
    x = clip_preprocess(image, size=patch * int(round(math.sqrt(sd["positional_embedding"].shape[0] - 1)))) if preprocess else image
    x = F.conv2d(x, sd["conv1.weight"], stride=patch)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    x = torch.cat([sd["class_embedding"].reshape(1, 1, -1).expand(x.shape[0], 1, -1), x], dim=1) + sd["positional_embedding"]
    x = F.layer_norm(x, x.shape[-1:], sd["ln_pre.weight"], sd["ln_pre.bias"])
    for i in range(_nblocks(sd, "transformer.resblocks.")):
        x = _resblock(sd, f"transformer.resblocks.{i}", x, heads)
    return x
