"""TEST INFRASTRUCTURE ONLY — functional fp32 restatement of the reference 3D U-Net forward.

Follows lvdm/modules/networks/openaimodel3d.py (UNetModel.forward :558-634, ResBlock._forward :210-236,
TemporalConvBlock.forward :272-279, Downsample/Upsample :51-106, TimestepEmbedSequential :36-48) and
lvdm/modules/attention.py (CrossAttention.forward :81-144, BasicTransformerBlock._forward :242-246,
SpatialTransformer.forward :294-310, TemporalTransformer.forward :365-412, GEGLU :415-422); timestep embedding
from lvdm/models/utils_diffusion.py:8-28. Operates directly on a reference-format ``state_dict``.
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def unet_layout(cfg):
    """Module census as (kind, state_dict prefix, info) lists, derived from the yaml unet_config
    (configs/inference_geo4d.yaml:62-93) the same way UNetModel.__init__ (openaimodel3d.py:386-556) does."""
    mc, mult, nres = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    attn_res, dh = set(cfg["attention_resolutions"]), cfg["num_head_channels"]
    temporal = cfg.get("temporal_attention", True)

    def attn_layers(prefix, start, ch):
        out = [("spatial", f"{prefix}.{start}", dict(ch=ch, heads=ch // dh))]
        if temporal:
            out.append(("temporal", f"{prefix}.{start + 1}", dict(ch=ch, inner=ch, heads=ch // dh)))
        return out

    inputs = [[("conv_in", "input_blocks.0.0", dict(cin=cfg["in_channels"], cout=mc))]]
    chans, ch, ds, idx = [mc], mc, 1, 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            layers = [("res", f"input_blocks.{idx}.0", dict(cin=ch, cout=m * mc))]
            ch = m * mc
            if ds in attn_res:
                layers += attn_layers(f"input_blocks.{idx}", 1, ch)
            inputs.append(layers); chans.append(ch); idx += 1
        if level != len(mult) - 1:
            inputs.append([("down", f"input_blocks.{idx}.0", dict(ch=ch))])
            chans.append(ch); idx += 1; ds *= 2
    middle = [("res", "middle_block.0", dict(cin=ch, cout=ch)), ("spatial", "middle_block.1", dict(ch=ch, heads=ch // dh))]
    if temporal:
        middle.append(("temporal", "middle_block.2", dict(ch=ch, inner=ch, heads=ch // dh)))
    middle.append(("res", f"middle_block.{len(middle)}", dict(cin=ch, cout=ch)))
    outputs, idx = [], 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            ich = chans.pop()
            layers = [("res", f"output_blocks.{idx}.0", dict(cin=ch + ich, cout=m * mc))]
            ch = m * mc
            if ds in attn_res:
                layers += attn_layers(f"output_blocks.{idx}", 1, ch)
            if level and i == nres:
                layers.append(("up", f"output_blocks.{idx}.{len(layers)}", dict(ch=ch)))
                ds //= 2
            outputs.append(layers); idx += 1
    return dict(inputs=inputs, middle=middle, outputs=outputs, final_ch=ch)


class _SD:
    def __init__(self, sd):
        self.sd = sd

    def lin(self, x, p, bias=True):
        w = self.sd[p + ".weight"]
        return F.linear(x, w.reshape(w.shape[0], -1), self.sd.get(p + ".bias") if bias else None)

    def gn(self, x, p, eps):
        return F.group_norm(x, 32, self.sd[p + ".weight"], self.sd[p + ".bias"], eps)

    def ln(self, x, p):
        return F.layer_norm(x, x.shape[-1:], self.sd[p + ".weight"], self.sd[p + ".bias"], 1e-5)

    def conv2(self, x, p, stride=1, pad=1):
        return F.conv2d(x, self.sd[p + ".weight"], self.sd[p + ".bias"], stride=stride, padding=pad)

    def conv3(self, x, p):
        return F.conv3d(x, self.sd[p + ".weight"], self.sd[p + ".bias"], padding=(1, 0, 0))


def _mha(q, k, v, heads):
    b, n, _ = q.shape
    d = q.shape[-1] // heads
    def split(t):
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
    w = torch.softmax(split(q) @ split(k).transpose(-1, -2) * d ** -0.5, dim=-1)
    return (w @ split(v)).permute(0, 2, 1, 3).reshape(b, n, heads * d)


def _attention(S, x, p, heads, context=None, image_cross=False):
    q = S.lin(x, p + ".to_q", bias=False)
    if context is None:
        out = _mha(q, S.lin(x, p + ".to_k", False), S.lin(x, p + ".to_v", False), heads)
    else:
        text, img = context[:, :77], context[:, 77:]
        out = _mha(q, S.lin(text, p + ".to_k", False), S.lin(text, p + ".to_v", False), heads)
        if image_cross:
            out = out + 1.0 * _mha(q, S.lin(img, p + ".to_k_ip", False), S.lin(img, p + ".to_v_ip", False), heads)
    return S.lin(out, p + ".to_out.0")


def _block(S, x, p, heads, context, image_cross):
    x = _attention(S, S.ln(x, p + ".norm1"), p + ".attn1", heads) + x
    x = _attention(S, S.ln(x, p + ".norm2"), p + ".attn2", heads, context, image_cross) + x
    h = S.lin(S.ln(x, p + ".norm3"), p + ".ff.net.0.proj")
    a, gate = h.chunk(2, dim=-1)
    return S.lin(a * F.gelu(gate), p + ".ff.net.2") + x


def _res(S, x, emb, p, b, temporal_conv=True):
    h = S.conv2(F.silu(S.gn(x, p + ".in_layers.0", 1e-5)), p + ".in_layers.2")
    h = h + S.lin(F.silu(emb), p + ".emb_layers.1")[:, :, None, None]
    h = S.conv2(F.silu(S.gn(h, p + ".out_layers.0", 1e-5)), p + ".out_layers.3")
    skip = x if (p + ".skip_connection.weight") not in S.sd else S.conv2(x, p + ".skip_connection", pad=0)
    h = skip + h
    if temporal_conv and (p + ".temopral_conv.conv1.0.weight") in S.sd:
        bt, c, hh, ww = h.shape
        z = h.reshape(b, bt // b, c, hh, ww).permute(0, 2, 1, 3, 4)
        y = z
        for name, conv_idx in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
            q = f"{p}.temopral_conv.{name}"
            y = S.conv3(F.silu(S.gn(y, q + ".0", 1e-5)), f"{q}.{conv_idx}")
        h = (z + y).permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
    return h


def _spatial(S, x, p, heads, context):
    bt, c, hh, ww = x.shape
    y = S.gn(x, p + ".norm", 1e-6).permute(0, 2, 3, 1).reshape(bt, hh * ww, c)
    y = S.lin(y, p + ".proj_in")
    y = _block(S, y, p + ".transformer_blocks.0", heads, context, image_cross=True)
    y = S.lin(y, p + ".proj_out")
    return y.reshape(bt, hh, ww, c).permute(0, 3, 1, 2) + x


def _temporal(S, x, p, heads, b):
    bt, c, hh, ww = x.shape
    t = bt // b
    z = x.reshape(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)               # b c t h w
    y = S.gn(z, p + ".norm", 1e-6).permute(0, 3, 4, 2, 1).reshape(b * hh * ww, t, c)  # (b h w) t c
    y = S.lin(y, p + ".proj_in")                                        # Linear or Conv1d(k=1): same math
    y = _block(S, y, p + ".transformer_blocks.0", heads, None, False)
    y = S.lin(y, p + ".proj_out")
    y = y.reshape(b, hh, ww, t, c).permute(0, 4, 3, 1, 2)               # b c t h w
    return (y + z).permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)


def _run(S, layers, h, emb, context, b):
    for kind, p, info in layers:
        if kind == "conv_in":
            h = S.conv2(h, p)
        elif kind == "res":
            h = _res(S, h, emb, p, b)
        elif kind == "spatial":
            h = _spatial(S, h, p, info["heads"], context)
        elif kind == "temporal":
            h = _temporal(S, h, p, info["heads"], b)
        elif kind == "down":
            h = S.conv2(h, p + ".op", stride=2)
        elif kind == "up":
            h = S.conv2(F.interpolate(h, scale_factor=2, mode="nearest"), p + ".conv")
    return h


@torch.no_grad()
def unet_forward(sd, cfg, x, timesteps, context, fs=None):
    S = _SD(sd)
    lay = unet_layout(cfg)
    b, _, t, hh, ww = x.shape
    mc = cfg["model_channels"]
    emb = S.lin(F.silu(S.lin(timestep_embedding(timesteps, mc), "time_embed.0")), "time_embed.2")
    if context.shape[1] == 77 + t * 16:
        text = context[:, :77].repeat_interleave(t, dim=0)
        img = context[:, 77:].reshape(b * t, 16, context.shape[-1])
        context = torch.cat([text, img], dim=1)
    else:
        context = context.repeat_interleave(t, dim=0)
    emb = emb.repeat_interleave(t, dim=0)
    if cfg.get("fs_condition", False):
        if fs is None:
            fs = torch.full((b,), cfg.get("default_fs", 4), dtype=torch.long)
        fe = S.lin(F.silu(S.lin(timestep_embedding(fs, mc), "fps_embedding.0")), "fps_embedding.2")
        emb = emb + fe.repeat_interleave(t, dim=0)
    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], hh, ww)
    hs = []
    for i, layers in enumerate(lay["inputs"]):
        h = _run(S, layers, h, emb, context, b)
        if i == 0 and cfg.get("addition_attention", False):
            h = _temporal(S, h, "init_attn.0", 8, b)
        hs.append(h)
    h = _run(S, lay["middle"], h, emb, context, b)
    for layers in lay["outputs"]:
        h = _run(S, layers, torch.cat([h, hs.pop()], dim=1), emb, context, b)
    y = S.conv2(F.silu(S.gn(h, "out.0", 1e-5)), "out.2")
    return y.reshape(b, t, -1, hh, ww).permute(0, 2, 1, 3, 4)
