"""DEV TOOL (test infrastructure, CPU only) — where does the reduced-precision error of the HIP engine come from?

Re-runs the oracle U-Net / VAE decode with the engine's rounding points emulated in PyTorch (fp32 math, explicit
`.to(dtype).float()` at every place the engine stores a tensor), so that a mixed-precision scheme can be chosen on
the CPU before it is built in HIP. A scheme names the storage dtype of
    w   : GEMM weights                          a : branch activations (GN / LN outputs, q/k/v, attention out, GEGLU out, conv1 out)
    s   : residual streams (h between layers, x inside a transformer block)
    p   : softmax probabilities fed to the PV MFMA
`python tests/precision_sim.py` prints relative L2 vs the fp32 oracle for a list of schemes on the tiny golden config
and on one window (3-step DDIM + 4-modality decode), the exact setting of tests/test_parity_gpu.py::test_window_end_to_end_vs_oracle.
"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ddim as oddim  # noqa: E402
from oracle import unet as ounet  # noqa: E402
from oracle.params import seeded_state_dict  # noqa: E402

DT = {"f32": None, "f16": torch.float16, "bf16": torch.bfloat16, "bf16x2": "bf16x2", "f16x2": "f16x2", None: None}


class Scheme:
    def __init__(self, w="f32", a="f32", s="f32", p=None, h1=None, name=None, c3a=None, c3w=None, vae3=False, two_pass=(), attn1=None):
        """`c3a` / `c3w` (round 5, the mixed-pass question of VERDICT r4 #8): storage of the A operand / the weights of the long-K 3x3
        convolutions ONLY (ResBlock in / out convs, down / up samplers; with `vae3` also the VAE's 3x3 convs); None = as `a` / `w`."""
        self.w, self.a, self.s, self.p, self.h1 = DT[w], DT[a], DT[s], DT[p if p else a], DT[h1 if h1 else a]
        self.c3a, self.c3w, self.vae3 = (DT[c3a] if c3a else self.a), (DT[c3w] if c3w else self.w), vae3
        # classes of GEMMs (beyond the 3x3 convs) whose A operand is additionally rounded to f16 (weights stay ~exact: f16 hi + lo):
        # "tconv" temporal 3-tap convs, "ff" GEGLU feed-forward (both linears), "proj" q / k / v / out / proj_in / proj_out linears
        self.two_pass = frozenset(two_pass)
        # round 6: operands of the SPATIAL SELF-attention's two GEMMs (attention.hip): None = as `a` / `p`; "x2" = q and P one f16, K and V f16 hi +
        # lo (two f16 MFMAs per product); "f16" = q, K, V, P one f16 each (one MFMA per product)
        self.attn1 = attn1
        self.attnT = None       # "f16": the TEMPORAL attention's q / k / v arrive as f16 rows (its arithmetic stays fp32 VALU)
        self.name = name or f"w={w} a={a} s={s} p={p or a} h1={h1 or a}" + (f" c3a={c3a} c3w={c3w}" if (c3a or c3w) else "")

    @staticmethod
    def _q(x, dt):
        if dt is None:
            return x
        if dt == "bf16x2":                      # the operand the bf16x3 MFMA scheme sees: bf16 hi + bf16 lo (~16 mantissa bits)
            hi = x.to(torch.bfloat16).float()
            return hi + (x - hi).to(torch.bfloat16).float()
        if dt == "f16x2":                       # f16 hi + f16 lo of a value pre-scaled by 2^12 (keeps lo out of the f16 subnormals): ~22 bits
            y = x * 4096.0
            hi = y.to(torch.float16).float()
            return (hi + (y - hi).to(torch.float16).float()) / 4096.0
        return x.to(dt).float()

    def qw(self, x): return self._q(x, self.w)
    def qa(self, x): return self._q(x, self.a)
    def qs(self, x): return self._q(x, self.s)
    def qp(self, x): return self._q(x, self.p)
    def qh1(self, x): return self._q(x, self.h1)
    def q3a(self, x): return self._q(x, self.c3a)
    def q3w(self, x): return self._q(x, self.c3w)


class S_:
    """state_dict access with weight rounding"""
    def __init__(self, sd, sc):
        self.sd, self.sc = sd, sc
        self._wq = {}

    def w(self, name):
        t = self._wq.get(name)
        if t is None:
            t = self._wq[name] = self.sc.qw(self.sd[name])
        return t

    def lin(self, x, p, bias=True, cls=None):
        if cls is not None and any(c in self.sc.two_pass for c in ((cls,) if isinstance(cls, str) else cls)):
            x = x.to(torch.float16).float()
        w = self.w(p + ".weight")
        if cls is not None and "v1" in cls and "wv16" in self.sc.two_pass:      # round 6: the V^T projection as ONE f16 pass (weight rounded to f16 too)
            w = self.sd[p + ".weight"].to(torch.float16).float()
        w16 = getattr(self.sc, "w16", ())            # classes whose WEIGHTS are ONE f16 as well (a single-pass f16 GEMM)
        if cls is not None and any(c in w16 for c in ((cls,) if isinstance(cls, str) else cls)):
            w = self.sd[p + ".weight"].to(torch.float16).float()
        return F.linear(x, w.reshape(w.shape[0], -1), self.sd.get(p + ".bias") if bias else None)

    def gn(self, x, p, eps):
        return F.group_norm(x, 32, self.sd[p + ".weight"], self.sd[p + ".bias"], eps)

    def ln(self, x, p):
        return F.layer_norm(x, x.shape[-1:], self.sd[p + ".weight"], self.sd[p + ".bias"], 1e-5)

    def conv2(self, x, p, stride=1, pad=1):
        return F.conv2d(x, self.w(p + ".weight"), self.sd[p + ".bias"], stride=stride, padding=pad)

    def conv3x3(self, x, p, stride=1):
        """A long-K 3x3 convolution on the UNROUNDED input `x`: operand storage per Scheme.c3a / c3w."""
        t = self._wq.get("3|" + p)
        if t is None:
            t = self._wq["3|" + p] = self.sc.q3w(self.sd[p + ".weight"])
        return F.conv2d(self.sc.q3a(x), t, self.sd[p + ".bias"], stride=stride, padding=1)

    def conv3(self, x, p):
        if "tconv" in self.sc.two_pass:
            x = x.to(torch.float16).float()
        w = self.sd[p + ".weight"].to(torch.float16).float() if "tconv" in getattr(self.sc, "w16", ()) else self.w(p + ".weight")
        return F.conv3d(x, w, self.sd[p + ".bias"], padding=(1, 0, 0))


def _mha(sc, q, k, v, heads, self1=False, tself_f16=False, cross=False):
    b, n, _ = q.shape
    d = q.shape[-1] // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
    mode = sc.attn1 if self1 else (getattr(sc, "attnC", None) if cross else None)
    h16 = lambda t: t.to(torch.float16).float()
    if tself_f16:
        q, k, v = h16(q), h16(k), h16(v)
    if mode == "x2":
        q, k, v = h16(q), Scheme._q(k, "f16x2"), Scheme._q(v, "f16x2")
    elif mode == "f16":
        q, k, v = h16(q), h16(k), h16(v)
    s = split(q) @ split(k).transpose(-1, -2) * d ** -0.5
    m = s.amax(-1, keepdim=True)
    pr = torch.exp(s - m)
    l = pr.sum(-1, keepdim=True)                        # the engine sums the UNROUNDED fp32 p
    o = ((h16(pr) if mode else sc.qp(pr)) @ split(v)) / l
    return o.permute(0, 2, 1, 3).reshape(b, n, heads * d)


def _attention(S, x, p, heads, context=None, image_cross=False):
    sc = S.sc
    # classes as the engine groups them (geo4d_amd/precision.py TWO_PASS_CLASSES): "ln" = LayerNorm-fed projections writing plain rows
    # (cross-attention q; temporal q | k | v - `tself`), "qk1" / "v1" = the spatial self-attention's q | k and V^T, "attn_out" = to_out
    tself = getattr(S, "_temporal_block", False)
    q = sc.qa(S.lin(x, p + ".to_q", False, cls=("proj", "ln" if (context is not None or tself) else "qk1")))
    if context is None:
        out = _mha(sc, q, sc.qa(S.lin(x, p + ".to_k", False, cls=("proj", "ln" if tself else "qk1"))),
                   sc.qa(S.lin(x, p + ".to_v", False, cls=("proj", "ln" if tself else "v1"))), heads, self1=not tself, tself_f16=bool(tself and sc.attnT == "f16"))
    else:
        text, img = sc.qa(context[:, :77]), sc.qa(context[:, 77:])
        out = _mha(sc, q, sc.qa(S.lin(text, p + ".to_k", False)), sc.qa(S.lin(text, p + ".to_v", False)), heads, cross=True)
        if image_cross:
            out = out + _mha(sc, q, sc.qa(S.lin(img, p + ".to_k_ip", False)), sc.qa(S.lin(img, p + ".to_v_ip", False)), heads, cross=True)
    return S.lin(sc.qa(out), p + ".to_out.0", cls=("proj", "attn_out", "attn_out_t" if tself else "attn_out_c" if context is not None else "attn_out_s"))


def _block(S, x, p, heads, context, image_cross):
    sc = S.sc
    x = sc.qs(_attention(S, sc.qa(S.ln(x, p + ".norm1")), p + ".attn1", heads) + x)
    x = sc.qs(_attention(S, sc.qa(S.ln(x, p + ".norm2")), p + ".attn2", heads, context, image_cross) + x)
    h = S.lin(sc.qa(S.ln(x, p + ".norm3")), p + ".ff.net.0.proj", cls="ff")
    a, gate = h.chunk(2, dim=-1)
    return sc.qs(S.lin(sc.qa(a * F.gelu(gate)), p + ".ff.net.2", cls="ff") + x)


def _res(S, x, emb, p, b):
    sc = S.sc
    h = S.conv3x3(F.silu(S.gn(x, p + ".in_layers.0", 1e-5)), p + ".in_layers.2")
    h = sc.qh1(h + F.linear(F.silu(emb), S.sd[p + ".emb_layers.1.weight"], S.sd[p + ".emb_layers.1.bias"])[:, :, None, None])
    h = S.conv3x3(F.silu(S.gn(h, p + ".out_layers.0", 1e-5)), p + ".out_layers.3")
    skip = x if (p + ".skip_connection.weight") not in S.sd else S.conv2(sc.qa(x) if "raw" not in sc.two_pass else sc.qa(x).to(torch.float16).float(), p + ".skip_connection", pad=0)
    h = sc.qs(skip + h)
    if (p + ".temopral_conv.conv1.0.weight") in S.sd:
        bt, c, hh, ww = h.shape
        z = h.reshape(b, bt // b, c, hh, ww).permute(0, 2, 1, 3, 4)
        y = z
        for i, (name, ci) in enumerate((("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3))):
            q = f"{p}.temopral_conv.{name}"
            y = S.conv3(sc.qa(F.silu(S.gn(y, q + ".0", 1e-5))), f"{q}.{ci}")
            if i < 3:
                y = sc.qh1(y)
        h = sc.qs(z + y).permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
    return h


def _spatial(S, x, p, heads, context):
    sc = S.sc
    bt, c, hh, ww = x.shape
    y = sc.qa(S.gn(x, p + ".norm", 1e-6)).permute(0, 2, 3, 1).reshape(bt, hh * ww, c)
    y = sc.qs(S.lin(y, p + ".proj_in", cls=("proj", "proj_in")))
    S._temporal_block = False
    y = _block(S, y, p + ".transformer_blocks.0", heads, context, True)
    y = S.lin(sc.qa(y), p + ".proj_out", cls=("proj", "proj_out"))
    return sc.qs(y.reshape(bt, hh, ww, c).permute(0, 3, 1, 2) + x)


def _temporal(S, x, p, heads, b):
    sc = S.sc
    bt, c, hh, ww = x.shape
    t = bt // b
    z = x.reshape(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)
    y = sc.qa(S.gn(z, p + ".norm", 1e-6)).permute(0, 3, 4, 2, 1).reshape(b * hh * ww, t, c)
    y = sc.qs(S.lin(y, p + ".proj_in", cls=("proj", "proj_in")))
    S._temporal_block = True
    y = _block(S, y, p + ".transformer_blocks.0", heads, None, False)
    S._temporal_block = False
    y = S.lin(sc.qa(y), p + ".proj_out", cls=("proj", "proj_out"))
    y = y.reshape(b, hh, ww, t, c).permute(0, 4, 3, 1, 2)
    return sc.qs(y + z).permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)


def _run(S, layers, h, emb, context, b):
    sc = S.sc
    for kind, p, info in layers:
        if kind == "conv_in":
            h = sc.qs(S.conv2(sc.qa(h), p))
        elif kind == "res":
            h = _res(S, h, emb, p, b)
        elif kind == "spatial":
            h = _spatial(S, h, p, info["heads"], context)
        elif kind == "temporal":
            h = _temporal(S, h, p, info["heads"], b)
        elif kind == "down":
            h = sc.qs(S.conv3x3(h, p + ".op", stride=2))
        elif kind == "up":
            h = sc.qs(S.conv3x3(F.interpolate(h, scale_factor=2, mode="nearest"), p + ".conv"))
    return h


@torch.no_grad()
def unet_forward(sd, cfg, x, timesteps, context, fs, sc):
    S = sd if isinstance(sd, S_) else S_(sd, sc)
    lay = ounet.unet_layout(cfg)
    b, _, t, hh, ww = x.shape
    mc = cfg["model_channels"]
    lin32 = lambda v, p: F.linear(v, S.sd[p + ".weight"], S.sd[p + ".bias"])
    emb = lin32(F.silu(lin32(ounet.timestep_embedding(timesteps, mc), "time_embed.0")), "time_embed.2")
    text = context[:, :77].repeat_interleave(t, dim=0)
    img = context[:, 77:].reshape(b * t, 16, context.shape[-1])
    context = torch.cat([text, img], dim=1)
    emb = emb.repeat_interleave(t, dim=0)
    if cfg.get("fs_condition", False):
        fe = lin32(F.silu(lin32(ounet.timestep_embedding(fs, mc), "fps_embedding.0")), "fps_embedding.2")
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
    y = S.conv2(sc.qa(F.silu(S.gn(h, "out.0", 1e-5))), "out.2")
    return y.reshape(b, t, -1, hh, ww).permute(0, 2, 1, 3, 4)


# ---- VAE decode with the same rounding points ---------------------------------------------------------------------------
def _vgn(S, x, p):
    return F.group_norm(x, 32, S.sd[p + ".weight"], S.sd[p + ".bias"], 1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _vresnet(S, x, p):
    sc = S.sc
    c3 = (lambda t, q: S.conv3x3(t, q)) if sc.vae3 else (lambda t, q: S.conv2(sc.qa(t), q))
    h = sc.qh1(c3(_swish(_vgn(S, x, p + ".norm1")), p + ".conv1"))
    h = c3(_swish(_vgn(S, h, p + ".norm2")), p + ".conv2")
    if (p + ".nin_shortcut.weight") in S.sd:
        x = S.conv2(sc.qa(x), p + ".nin_shortcut", pad=0)
    return sc.qs(x + h)


def _vattn(S, x, p):
    sc = S.sc
    b, c, hh, ww = x.shape
    h = sc.qa(_vgn(S, x, p + ".norm"))
    q, k, v = (sc.qa(S.conv2(h, f"{p}.{n}", pad=0)).reshape(b, c, hh * ww) for n in ("q", "k", "v"))
    w = sc.qp(torch.softmax(torch.bmm(q.permute(0, 2, 1), k) * (int(c) ** -0.5), dim=2))
    h = sc.qa(torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww))
    return sc.qs(x + S.conv2(h, p + ".proj_out", pad=0))


def vae_features(S, ddconfig, z, prefix="decoder"):
    sc = S.sc
    nres, nlev = ddconfig["num_res_blocks"], len(ddconfig["ch_mult"])
    h = sc.qs(S.conv2(sc.qa(z), prefix + ".conv_in"))
    h = _vresnet(S, h, prefix + ".mid.block_1")
    h = _vattn(S, h, prefix + ".mid.attn_1")
    h = _vresnet(S, h, prefix + ".mid.block_2")
    for lvl in reversed(range(nlev)):
        for blk in range(nres + 1):
            h = _vresnet(S, h, f"{prefix}.up.{lvl}.block.{blk}")
        if lvl != 0:
            hu = sc.qa(h).to(torch.float16).float() if (sc.vae3 and "raw" in sc.two_pass) else sc.qa(h)
            h = sc.qs(S.conv2(F.interpolate(hu, scale_factor=2.0, mode="nearest"), f"{prefix}.up.{lvl}.upsample.conv"))
    return h


def _vhead(S, h, prefix):
    return S.conv2(S.sc.qa(_swish(_vgn(S, h, prefix + ".norm_out"))), prefix + ".conv_out")


def vae_decode(sd, ddconfig, adaptorconfig, z, sc, conf):
    S = S_(sd, sc)
    z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])      # folded into conv_in / tiny: fp32
    feat = vae_features(S, ddconfig, z)
    rgb = _vhead(S, feat, "decoder")
    if not conf:
        return rgb
    h = feat
    for blk in range(adaptorconfig["num_res_blocks"] + 1):
        h = _vresnet(S, h, f"decoder_adaptor.up.0.block.{blk}")
    return torch.cat([rgb, _vhead(S, h, "decoder_adaptor")], dim=1)


def decode_modalities(fsd, psd, ddconfig, adaptorconfig, samples, sc, scale_factor=0.18215):
    b, _, t, h, w = samples.shape
    frames = lambda z: z.permute(0, 2, 1, 3, 4).reshape(b * t, 4, h, w) / scale_factor
    back = lambda y: y.reshape(b, t, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)
    pc = back(vae_decode(psd, ddconfig, adaptorconfig, frames(samples[:, 0:4]), sc, True))
    ray = back(vae_decode(fsd, ddconfig, adaptorconfig, frames(samples[:, 4:8]), sc, False))
    cross = back(vae_decode(fsd, ddconfig, adaptorconfig, frames(samples[:, 8:12]), sc, False))
    depth = back(vae_decode(fsd, ddconfig, adaptorconfig, frames(samples[:, 12:16]), sc, False)).mean(dim=1, keepdim=True)
    return torch.cat([pc, ray, cross, depth], dim=1)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


SCHEMES = [
    Scheme("f16", "f16", "f16", name="f16 everywhere (round-1 f16 mode)"),
    Scheme("f16", "f16", "f32", name="f16 operands, fp32 residual streams"),
    Scheme("f16", "f16", "f32", h1="f32", name="f16 operands, fp32 streams + fp32 conv1/tconv intermediates"),
    Scheme("f16", "f32", "f32", name="only weights f16"),
    Scheme("f32", "f16", "f32", name="only branch activations f16 (fp32 streams)"),
    Scheme("bf16", "bf16", "bf16", name="bf16 everywhere (round-1 bench mode)"),
    Scheme("bf16", "bf16", "f32", name="bf16 operands, fp32 residual streams"),
    Scheme("bf16x2", "bf16x2", "f32", name="bf16x3 mode: f32 storage, operands split hi+lo, 3 bf16 MFMAs per product"),
    # round 5: two-pass f16 products (A one f16, W f16 hi + lo) on classes of GEMMs, everything else as bf16x3 (profiles/r05_mixed_pass_sim.md)
    Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", name="two-pass f16 on the U-Net's 3x3 convs"),
    Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=("tconv", "ln", "ff"), name="bf16x3m mode: two-pass f16 on conv3x3 + vae3x3 + tconv + ln + ff"),
    Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=("tconv", "ln", "ff", "proj_in", "proj_out", "raw"), name="... + the stream-rounding classes (proj_in / proj_out / raw): rejected"),
    # round 6: the spatial self-attention's own GEMMs (S = q.K^T, O = P.V), alone and on top of the bf16x3m classes
    Scheme("bf16x2", "bf16x2", "f32", attn1="x2", name="bf16x3 + self-attention two-pass (q, P one f16; K, V f16 hi + lo)"),
    Scheme("bf16x2", "bf16x2", "f32", attn1="f16", name="bf16x3 + self-attention ONE pass (q, K, V, P one f16)"),
    Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=("tconv", "ln", "ff"), attn1="x2", name="bf16x3m + self-attention two-pass"),
    Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=("tconv", "ln", "ff"), attn1="f16", name="bf16x3m + self-attention ONE pass"),
    Scheme("bf16x2", "bf16x2", "f32", two_pass=("qk1", "v1", "attn_out"), attn1="f16", name="bf16x3 + self-attention CHAIN (n1 f16 -> two-pass q|k, V; one-pass attention; two-pass to_out)"),
    Scheme("bf16x2", "bf16x2", "f32", two_pass=("qk1", "v1", "attn_out", "wv16"), attn1="f16", name="bf16x3 + self-attention CHAIN, V^T projection in ONE f16 pass (W_v rounded to f16)"),
    Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=("tconv", "ln", "ff", "qk1", "v1", "attn_out"), attn1="f16", name="bf16x3m + self-attention CHAIN"),
    Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=("tconv", "ln", "ff", "qk1", "v1", "attn_out", "wv16"), attn1="f16", name="bf16x3m + self-attention CHAIN, V^T in ONE f16 pass"),
]
def _with(sc, **kw):
    for k, v in kw.items():
        setattr(sc, k, v)
    return sc


M6 = ("tconv", "ln", "ff", "qk1", "v1", "attn_out_s", "wv16")      # the round-6 default: bf16x3m classes + the spatial self-attention chain ("attn")
SCHEMES += [
    Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=M6, attn1="f16", name="R6 default: bf16x3m + attn (to_out of the SPATIAL self-attention only)"),
    _with(Scheme("bf16x2", "bf16x2", "f32", two_pass=("ln", "attn_out_t"), name="R6 bf16x3 + temporal attention chain alone (q | k | v f16 rows, two-pass to_out)"), attnT="f16"),
    _with(Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=M6 + ("attn_out_t",), attn1="f16", name="R6 default + temporal attention chain (tattn)"), attnT="f16"),
    _with(Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=M6 + ("attn_out_c",), attn1="f16", name="R6 default + cross-attention chain (q, K, V, P one f16; two-pass to_out) (cattn)"), attnC="f16"),
    _with(Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=M6 + ("attn_out_c", "attn_out_t"), attn1="f16", name="R6 default + tattn + cattn"), attnC="f16", attnT="f16"),
]
M6D = M6 + ("attn_out_t", "attn_out_c")
SCHEMES += [
    _with(Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=M6D, attn1="f16", name="R6 HEAD default (attn + cattn + tattn)"), attnC="f16", attnT="f16"),
    _with(Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=M6D, attn1="f16", name="R6 HEAD + ff weights ONE f16 (single-pass GEGLU / ff-out)"), attnC="f16", attnT="f16", w16=("ff",)),
    _with(Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=M6D, attn1="f16", name="R6 HEAD + tconv weights ONE f16"), attnC="f16", attnT="f16", w16=("tconv",)),
    _with(Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16x2", vae3=True, two_pass=M6D, attn1="f16", name="R6 HEAD + ln / qk1 / attn_out weights ONE f16 (all attention projections single-pass)"), attnC="f16", attnT="f16", w16=("ln", "qk1", "attn_out")),
    _with(Scheme("bf16x2", "bf16x2", "f32", c3a="f16", c3w="f16", vae3=False, two_pass=M6D, attn1="f16", name="R6 HEAD + U-Net conv3x3 weights ONE f16 (VAE stays two-pass)"), attnC="f16", attnT="f16"),
]
if os.environ.get("SIM_ONLY"):
    SCHEMES = [sc for sc in SCHEMES if any(k in sc.name for k in os.environ["SIM_ONLY"].split(","))]


def main():
    torch.set_num_threads(int(os.environ.get("SIM_THREADS", os.cpu_count() or 1)))
    G = os.path.join(ROOT, "tests", "golden")
    u = torch.load(os.path.join(G, "unet_tiny.pt"), weights_only=False)
    v = torch.load(os.path.join(G, "vae_tiny.pt"), weights_only=False)
    usd, vsd = seeded_state_dict(u["shapes"]), seeded_state_dict(v["shapes"])
    psd = seeded_state_dict(dict(v["shapes"]), gain=0.9)
    cfg = u["unet_config"]
    gen = torch.Generator().manual_seed(777)
    B, T, h, w = 1, 16, 8, 8
    x_T = torch.randn((B, 16, T, h, w), generator=gen)
    ctx = torch.randn((B, 77 + 16 * T, cfg["context_dim"]), generator=gen)
    zc = torch.randn((B, 4, T, h, w), generator=gen)
    fs = torch.tensor([24])
    S = int(os.environ.get("SIM_STEPS", "3"))
    exact = Scheme()

    def run(sc):
        S_u = S_(usd, sc)
        am = lambda x, t: unet_forward(S_u, cfg, torch.cat([x, zc], 1), t, ctx, fs, sc)
        v1 = am(x_T, torch.tensor([999]))
        lat = oddim.ddim_sample(am, oddim.make_schedule(), oddim.make_scale_arr(), S, x_T, eta=0.0)
        return v1, lat, decode_modalities(vsd, psd, v["ddconfig"], v["adaptorconfig"], lat, sc)

    r_v, r_lat, r_out = run(exact)
    # sanity: the exact scheme IS the oracle
    chk = ounet.unet_forward(usd, cfg, torch.cat([x_T, zc], 1), torch.tensor([999]), ctx, fs)
    print(f"sim(fp32) vs oracle: {rel(r_v, chk):.2e}")
    print(f"{'scheme':70s} {'1 fwd':>9s} {'latent':>9s} {'pts':>9s} {'all11':>9s}   (S={S})")
    for sc in SCHEMES:
        a_v, a_lat, a_out = run(sc)
        # decode-only error: exact latent through the rounded decoder
        print(f"{sc.name:70s} {rel(a_v, r_v):9.2e} {rel(a_lat, r_lat):9.2e} {rel(a_out[:, :3], r_out[:, :3]):9.2e} {rel(a_out, r_out):9.2e}")


if __name__ == "__main__":
    main()
