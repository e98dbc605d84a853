"""Conditioning front-end on the HIP kernels: drop-ins for the three yaml targets
``lvdm.modules.encoders.condition.FrozenOpenCLIPEmbedder`` (cond_stage_config, text, layer "penultimate"),
``lvdm.modules.encoders.condition.FrozenOpenCLIPImageEmbedderV2`` (img_cond_stage_config) and
``lvdm.modules.encoders.resampler.Resampler`` (image_proj_stage_config) — same constructor keywords, same ``state_dict`` names
(``model.transformer.resblocks.N...`` / ``model.visual...`` as open_clip_torch 2.22 names them, so the reference checkpoint's
``cond_stage_model.*`` / ``embedder.*`` / ``image_proj_model.*`` tensors load), same call surface and outputs.

These run ONCE per clip (the shipped config feeds a fixed prompt and a zero image: constant across windows), so they reuse the
U-Net's kernels rather than getting their own: LayerNorm, conv_gemm linears with bias / residual / GELU epilogues, the d = 64
flash kernel for the Resampler's PerceiverAttention, and — for the ViT towers, whose heads are 64 (text, causal) and 80 (vision)
wide — per-head batched conv_gemm QK^T / PV around a (causal) row-softmax kernel with the head dimension zero-padded to the
GEMM's K granularity at pack time. The architecture constants default to ViT-H-14 (open_clip model config: text width 1024 /
24 layers / 16 heads / vocab 49408 / context 77; vision width 1280 / 32 layers / 16 heads / patch 14 / image 224).

Tokenisation (geo4d_amd/tokenizer.py): the CLIP byte-pair encoder of ``open_clip.tokenize`` (condition.py:207-210) restated; its merge
table ``bpe_simple_vocab_16e6.txt.gz`` ships in the open_clip wheel, not in the reference repository, so it is loaded from a
user-supplied path (``bpe_path=`` / ``GEO4D_CLIP_BPE``). ``FrozenOpenCLIPEmbedder`` takes prompts (tokenised with that table),
int64 token ids, or a ``tokenizer`` callable. The shipped scripts run with ``--text_input`` (scripts/infer_geo4d.sh:24), i.e. they
encode the fixed prompt of test_geo4d.py:410; without ``--text_input`` the prompt is blanked to ``""`` (test_geo4d.py:124-126), which
is <start> <end> + padding and needs no table.
"""
import math

import torch

from . import ops, pack
from .unet import ParamTree, init_params_, resolve_dtype

SOT, EOT = 49406, 49407
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class _Packed(ParamTree):
    """Shared plumbing: compute mode, lazy weight packing, no CPU fallback."""

    def _setup(self, compute_dtype):
        self.compute_dtype = resolve_dtype(compute_dtype)
        self._packed = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    def invalidate(self):
        self._packed = None

    def set_compute_dtype(self, d):
        self.compute_dtype = resolve_dtype(d)
        self.invalidate()
        return self

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    @property
    def storage_dtype(self):
        return self.compute_dtype.storage

    def _device_check(self, t, who):
        if t.device.type != "cuda":
            raise ops._lib.Geo4DNativeError(f"geo4d_amd.{who} runs only on a HIP device: call .cuda() first (there is no CPU fallback)")
        ops._lib.load()


# ---- ViT transformer blocks (open_clip ResidualAttentionBlock) --------------------------------------------------------------
def _block_shapes(add, p, w):
    add(p + ".ln_1.weight", (w,)); add(p + ".ln_1.bias", (w,))
    add(p + ".attn.in_proj_weight", (3 * w, w)); add(p + ".attn.in_proj_bias", (3 * w,))
    add(p + ".attn.out_proj.weight", (w, w)); add(p + ".attn.out_proj.bias", (w,))
    add(p + ".ln_2.weight", (w,)); add(p + ".ln_2.bias", (w,))
    add(p + ".mlp.c_fc.weight", (4 * w, w)); add(p + ".mlp.c_fc.bias", (4 * w,))
    add(p + ".mlp.c_proj.weight", (w, 4 * w)); add(p + ".mlp.c_proj.bias", (w,))


def _pack_block(sd, p, w, heads, dt):
    d = w // heads
    dp = pack.pad_to(d, ops.k_align(dt))                 # head width as a K dimension: zero padded to the GEMM slab
    f32 = lambda n: sd[n].float().contiguous()
    wq, wk, wv = sd[p + ".attn.in_proj_weight"].float().chunk(3, 0)
    bq, bk, bv = sd[p + ".attn.in_proj_bias"].float().chunk(3, 0)

    def head_pad(m):                                      # [heads*d, ...] -> [heads*dp, ...], zero rows after each head's d
        out = m.new_zeros((heads, dp) + tuple(m.shape[1:]))
        out[:, :d] = m.reshape((heads, d) + tuple(m.shape[1:]))
        return out.reshape((heads * dp,) + tuple(m.shape[1:]))
    return dict(ln1=(f32(p + ".ln_1.weight"), f32(p + ".ln_1.bias")), ln2=(f32(p + ".ln_2.weight"), f32(p + ".ln_2.bias")),
                qk=(pack.pack_linear(torch.cat([head_pad(wq), head_pad(wk)], 0), dt), torch.cat([head_pad(bq), head_pad(bk)]).contiguous()),
                v=(pack.pack_linear(wv, dt), bv.contiguous()),
                o=(pack.pack_linear(sd[p + ".attn.out_proj.weight"], dt), f32(p + ".attn.out_proj.bias")),
                fc=(pack.pack_linear(sd[p + ".mlp.c_fc.weight"], dt), f32(p + ".mlp.c_fc.bias")),
                proj=(pack.pack_linear(sd[p + ".mlp.c_proj.weight"], dt), f32(p + ".mlp.c_proj.bias")), d=d, dp=dp)


def _vit_block(e, x, B, N, heads, prec, causal):
    """x [B*N, W] tokens -> x + attn(ln_1 x) -> + mlp(ln_2 .): MultiheadAttention as per-head batched GEMMs + row softmax."""
    d, dp, x3 = e["d"], e["dp"], prec.x3
    dt = x.dtype
    ka = ops.k_align(prec)
    Np = pack.pad_to(N, ka)
    h = ops.layernorm(x, *e["ln1"])
    qk = ops.linear(h, *e["qk"])                                                   # [B*N, 2*heads*dp], heads zero-padded to dp
    att = torch.empty((B * N, heads * d), device=x.device, dtype=dt)
    scores = torch.empty((heads * N, N), device=x.device, dtype=torch.float32)
    probs = torch.zeros((heads * N, Np), device=x.device, dtype=dt)                # K padding of the PV GEMM stays zero
    vt = torch.zeros((heads * d, Np), device=x.device, dtype=dt)
    for b in range(B):
        rows = slice(b * N, (b + 1) * N)
        ops.linear_t(e["v"][0], h[rows], e["v"][1], out=vt[:, :N])                  # V^T [heads*d, N]: row = channel, column = key
        ops.batched_gemm(qk[rows, :heads * dp], qk[rows, heads * dp:], scores, batch=heads, M=N, N=N, K=dp, a_bs=dp, b_bs=dp,
                         o_bs=N * N, alpha=float(d) ** -0.5, x3=x3)
        ops.softmax_rows(scores, 1.0, dt, out=probs, causal_period=N if causal else 0)
        ops.batched_gemm(probs, vt, att[rows], batch=heads, M=N, N=d, K=Np, a_bs=N * Np, b_bs=d * Np, o_bs=d, x3=x3)
    x = ops.linear(att, *e["o"], residual=x)
    g = ops.linear(ops.layernorm(x, *e["ln2"]), *e["fc"], act=3)                    # GELU (erf) in the GEMM epilogue
    return ops.linear(g, *e["proj"], residual=x)


class FrozenOpenCLIPEmbedder(_Packed):
    """Text tower (condition.py:174-234). ``forward(text)``: list of prompts (tokenised by geo4d_amd.tokenizer with the merge
    table at ``bpe_path`` / $GEO4D_CLIP_BPE; "" needs none) or int64 token ids [B, 77] -> [B, 77, width]."""
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, layer="last",
                 width=1024, layers=24, heads=16, vocab_size=49408, tokenizer=None, bpe_path=None, compute_dtype=None):
        super().__init__()
        assert layer in self.LAYERS
        if arch != "ViT-H-14":
            raise NotImplementedError("only the ViT-H-14 text tower of configs/inference_geo4d.yaml is described here")
        self.max_length, self.layer, self.layer_idx = max_length, layer, (1 if layer == "penultimate" else 0)
        self.width, self.layers, self.heads, self.vocab_size, self.tokenizer = width, layers, heads, vocab_size, tokenizer
        self.bpe_path = bpe_path
        add = self.insert
        add("model.token_embedding.weight", (vocab_size, width)); add("model.positional_embedding", (max_length, width))
        for i in range(layers):
            _block_shapes(add, f"model.transformer.resblocks.{i}", width)
        add("model.ln_final.weight", (width,)); add("model.ln_final.bias", (width,))
        add("model.text_projection", (width, width)); add("model.logit_scale", ())       # carried by the checkpoint, unused on this path
        init_params_(self)
        self._setup(compute_dtype)

    def tokenize(self, text):
        if self.tokenizer is not None:
            return torch.as_tensor(self.tokenizer(text), dtype=torch.long)
        if isinstance(text, str):
            text = [text]
        if all(t == "" for t in text):                       # <start> <end> padding: no merge table needed
            toks = torch.zeros((len(text), self.max_length), dtype=torch.long)
            toks[:, 0], toks[:, 1] = SOT, EOT
            return toks
        from .tokenizer import SimpleTokenizer
        self.tokenizer = SimpleTokenizer(self.bpe_path, context_length=self.max_length)   # FileNotFoundError names what to supply
        toks = self.tokenizer(text)
        if int(toks.max()) >= self.vocab_size:
            raise ValueError(f"token id {int(toks.max())} outside the embedding table ({self.vocab_size} rows): wrong merge table?")
        return toks

    @torch.no_grad()
    def _pack(self):
        sd = dict(self.named_parameters())
        self._device_check(sd["model.ln_final.weight"], "FrozenOpenCLIPEmbedder")
        dt = self.compute_dtype
        f32 = lambda n: sd[n].float().contiguous()
        self._packed = dict(table=f32("model.token_embedding.weight"), pos=f32("model.positional_embedding"),
                            blocks=[_pack_block(sd, f"model.transformer.resblocks.{i}", self.width, self.heads, dt) for i in range(self.layers)],
                            ln=(f32("model.ln_final.weight"), f32("model.ln_final.bias")))
        return self._packed

    @torch.no_grad()
    def encode_with_transformer(self, tokens):
        P = self._packed or self._pack()
        dev = P["table"].device
        tokens = tokens.to(dev).contiguous()
        B, N = tokens.shape
        x = ops.embed_tokens(tokens, P["table"], P["pos"], self.storage_dtype)
        for e in P["blocks"][: self.layers - self.layer_idx]:          # condition.py:223-231: stop `layer_idx` blocks early
            x = _vit_block(e, x, B, N, self.heads, self.compute_dtype, causal=True)
        return ops.layernorm(x, *P["ln"]).reshape(B, N, self.width).float()

    def forward(self, text):
        tokens = text if isinstance(text, torch.Tensor) else self.tokenize(text)
        return self.encode_with_transformer(tokens)

    encode = forward


class FrozenOpenCLIPImageEmbedderV2(_Packed):
    """Vision tower returning ALL tokens after the last block (condition.py:295-372). ``forward(image [b,3,h,w] in [-1,1])`` ->
    [b, 1 + (image_size / patch)^2, width]."""

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", freeze=True, layer="pooled", antialias=True,
                 width=1280, layers=32, heads=16, image_size=224, patch_size=14, embed_dim=1024, compute_dtype=None):
        super().__init__()
        if layer == "penultimate":
            raise NotImplementedError()      # as the reference does
        self.width, self.layers, self.heads, self.image_size, self.patch_size = width, layers, heads, image_size, patch_size
        self.grid = image_size // patch_size
        add = self.insert
        add("model.visual.class_embedding", (width,)); add("model.visual.positional_embedding", (1 + self.grid ** 2, width))
        add("model.visual.conv1.weight", (width, 3, patch_size, patch_size))
        add("model.visual.ln_pre.weight", (width,)); add("model.visual.ln_pre.bias", (width,))
        for i in range(layers):
            _block_shapes(add, f"model.visual.transformer.resblocks.{i}", width)
        add("model.visual.ln_post.weight", (width,)); add("model.visual.ln_post.bias", (width,)); add("model.visual.proj", (width, embed_dim))
        add("model.logit_scale", ())                                                    # carried by the checkpoint, unused here
        init_params_(self)
        self._setup(compute_dtype)

    def preprocess(self, x):
        """condition.py:332-340: resize to 224 (bicubic, align_corners), [-1,1] -> [0,1], CLIP mean / std. kornia's antialias
        pre-blur on down-scaling is not reproduced (the shipped config feeds a ZERO image, whose result does not depend on it)."""
        s = self.image_size
        if x.shape[-2:] != (s, s):
            x = torch.nn.functional.interpolate(x.float(), size=(s, s), mode="bicubic", align_corners=True)
        mean = torch.tensor(CLIP_MEAN, device=x.device).reshape(1, 3, 1, 1)
        std = torch.tensor(CLIP_STD, device=x.device).reshape(1, 3, 1, 1)
        return ((x.float() + 1.) / 2. - mean) / std

    @torch.no_grad()
    def _pack(self):
        sd = dict(self.named_parameters())
        self._device_check(sd["model.visual.ln_pre.weight"], "FrozenOpenCLIPImageEmbedderV2")
        dt = self.compute_dtype
        f32 = lambda n: sd[n].float().contiguous()
        pos = f32("model.visual.positional_embedding")
        self._packed = dict(conv=pack.pack_linear(sd["model.visual.conv1.weight"].reshape(self.width, -1), dt),
                            kpad=pack.pad_to(3 * self.patch_size ** 2, ops.k_align(dt)),
                            cls=(f32("model.visual.class_embedding") + pos[0]).to(self.storage_dtype), pos=pos[1:].contiguous(),
                            ln_pre=(f32("model.visual.ln_pre.weight"), f32("model.visual.ln_pre.bias")),
                            blocks=[_pack_block(sd, f"model.visual.transformer.resblocks.{i}", self.width, self.heads, dt) for i in range(self.layers)])
        return self._packed

    @torch.no_grad()
    def encode_with_vision_transformer(self, image, preprocess=True):
        P = self._packed or self._pack()
        dev = P["pos"].device
        x = self.preprocess(image.to(dev)) if preprocess else image.to(dev).float()
        B, p, g = x.shape[0], self.patch_size, self.grid
        N = 1 + g * g
        dt = self.storage_dtype
        # conv1 (stride = kernel = patch, no bias) as a linear over unfolded patches (host-side re-layout of a 224x224 image)
        patches = x.reshape(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, 3 * p * p)
        a = torch.zeros((B * g * g, P["kpad"]), device=dev, dtype=dt)
        a[:, : 3 * p * p] = patches.to(dt)
        tok = torch.empty((B, N, self.width), device=dev, dtype=dt)
        tok[:, 0] = P["cls"]                                                            # class token + its positional embedding
        pos = P["pos"].to(dt)
        for b in range(B):                                                             # + positional embedding in the GEMM epilogue
            ops.linear(a[b * g * g:(b + 1) * g * g], P["conv"], residual=pos, out=tok[b, 1:])
        x = ops.layernorm(tok.reshape(B * N, self.width), *P["ln_pre"])
        for e in P["blocks"]:
            x = _vit_block(e, x, B, N, self.heads, self.compute_dtype, causal=False)
        return x.reshape(B, N, self.width).float()

    def forward(self, image, no_dropout=False):
        return self.encode_with_vision_transformer(image)

    encode = forward


# ---- Resampler (lvdm/modules/encoders/resampler.py:96-163) ------------------------------------------------------------------
class Resampler(_Packed):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024, ff_mult=4,
                 video_length=None, compute_dtype=None):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("PerceiverAttention runs on the d_head = 64 flash kernel (yaml dim_head: 64)")
        self.dim, self.depth, self.heads, self.num_queries, self.video_length = dim, depth, heads, num_queries, video_length
        self.output_dim = output_dim
        nq = num_queries * video_length if video_length is not None else num_queries
        inner, add = dim_head * heads, self.insert
        add("latents", (1, nq, dim))
        add("proj_in.weight", (dim, embedding_dim)); add("proj_in.bias", (dim,))
        add("proj_out.weight", (output_dim, dim)); add("proj_out.bias", (output_dim,))
        add("norm_out.weight", (output_dim,)); add("norm_out.bias", (output_dim,))
        for i in range(depth):
            p = f"layers.{i}"
            for n in ("norm1", "norm2"):
                add(f"{p}.0.{n}.weight", (dim,)); add(f"{p}.0.{n}.bias", (dim,))
            add(f"{p}.0.to_q.weight", (inner, dim)); add(f"{p}.0.to_kv.weight", (2 * inner, dim)); add(f"{p}.0.to_out.weight", (dim, inner))
            add(f"{p}.1.0.weight", (dim,)); add(f"{p}.1.0.bias", (dim,))
            add(f"{p}.1.1.weight", (int(dim * ff_mult), dim)); add(f"{p}.1.3.weight", (dim, int(dim * ff_mult)))
        init_params_(self)
        with torch.no_grad():
            self.latents.copy_(torch.randn(self.latents.shape, generator=torch.Generator().manual_seed(0)) / dim ** 0.5)
        self._setup(compute_dtype)

    @torch.no_grad()
    def _pack(self):
        sd = dict(self.named_parameters())
        self._device_check(sd["norm_out.weight"], "Resampler")
        dt = self.compute_dtype
        f32 = lambda n: sd[n].float().contiguous()
        inner = 64 * self.heads
        L = []
        for i in range(self.depth):
            p = f"layers.{i}"
            wk, wv = sd[f"{p}.0.to_kv.weight"].float().chunk(2, 0)
            assert wk.shape[0] == inner
            L.append(dict(n1=(f32(f"{p}.0.norm1.weight"), f32(f"{p}.0.norm1.bias")), n2=(f32(f"{p}.0.norm2.weight"), f32(f"{p}.0.norm2.bias")),
                          q=pack.pack_linear(sd[f"{p}.0.to_q.weight"], dt), k=pack.pack_linear(wk, dt), v=pack.pack_linear(wv, dt),
                          o=pack.pack_linear(sd[f"{p}.0.to_out.weight"], dt), ffn=(f32(f"{p}.1.0.weight"), f32(f"{p}.1.0.bias")),
                          ff1=pack.pack_linear(sd[f"{p}.1.1.weight"], dt), ff2=pack.pack_linear(sd[f"{p}.1.3.weight"], dt)))
        self._packed = dict(layers=L, latents=sd["latents"].float()[0].to(self.storage_dtype).contiguous(),
                            pin=(pack.pack_linear(sd["proj_in.weight"], dt), f32("proj_in.bias")),
                            pout=(pack.pack_linear(sd["proj_out.weight"], dt), f32("proj_out.bias")),
                            nout=(f32("norm_out.weight"), f32("norm_out.bias")))
        return self._packed

    @torch.no_grad()
    def forward(self, x):
        P = self._packed or self._pack()
        dev, dt, x3 = P["latents"].device, self.storage_dtype, self.compute_dtype.x3
        four_d = x.dim() == 4
        if four_d:                                              # per-frame queries (resampler.py:133-147)
            B0, T, Lx, C = x.shape
            S, Q = B0 * T, self.num_queries
            lat = P["latents"].repeat(B0, 1)                    # [(b t q), dim]: frame t of a sample uses queries t*Q .. t*Q+Q
            x = x.reshape(S, Lx, C)
        else:
            S, Lx, C = x.shape
            Q = P["latents"].shape[0]
            lat = P["latents"].repeat(S, 1)
        ka = ops.k_align(self.compute_dtype)
        xin = torch.zeros((S * Lx, pack.pad_to(C, ka)), device=dev, dtype=dt)
        xin[:, :C] = x.reshape(S * Lx, C).to(dev).to(dt)
        xf = ops.linear(xin, *P["pin"])                          # [S*Lx, dim]
        H, inner = self.heads, 64 * self.heads
        Nk = Lx + Q
        epc = 4 if dt == torch.float32 else 8
        Nkp = pack.pad_to(Nk, epc)
        kbuf = torch.empty((S, Nk, inner), device=dev, dtype=dt)
        vtbuf = torch.zeros((S, inner, Nkp), device=dev, dtype=dt)
        for e in P["layers"]:
            xn, ln = ops.layernorm(xf, *e["n1"]), ops.layernorm(lat, *e["n2"])
            q = ops.linear(ln, e["q"])
            for s in range(S):                                   # keys / values of cat(x, latents) written side by side (resampler.py:76-78)
                xs, ls = xn[s * Lx:(s + 1) * Lx], ln[s * Q:(s + 1) * Q]
                ops.linear(xs, e["k"], out=kbuf[s, :Lx]); ops.linear(ls, e["k"], out=kbuf[s, Lx:])
                ops.linear_t(e["v"], xs, out=vtbuf[s][:, :Lx]); ops.linear_t(e["v"], ls, out=vtbuf[s][:, Lx:Nk])
            att = ops.attention(q, [(kbuf.reshape(S * Nk, inner), vtbuf.reshape(S * inner, Nkp), Nk, 1, inner * Nkp)], B=S, H=H, Nq=Q,
                                scale=0.125, x3=x3)              # (q * 64^-1/4) . (k * 64^-1/4) = q.k / 8
            lat = ops.linear(att, e["o"], residual=lat)
            g = ops.linear(ops.layernorm(lat, *e["ffn"]), e["ff1"], act=3)
            lat = ops.linear(g, e["ff2"], residual=lat)
        out = ops.layernorm(ops.linear(lat, *P["pout"]), *P["nout"]).float()
        return out.reshape(B0, T * Q, self.output_dim) if four_d else out.reshape(S, Q, self.output_dim)
