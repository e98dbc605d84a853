"""MI355X-native 3D U-Net — drop-in for ``lvdm.modules.networks.openaimodel3d.UNetModel`` (yaml ``unet_config.target``).

Same constructor kwargs (configs/inference_geo4d.yaml:62-93), same ``state_dict`` keys and shapes (1516 tensors at the
shipped config, including the reference's ``temopral_conv`` spelling, openaimodel3d.py:190), same ``forward`` signature
(openaimodel3d.py:558: ``x [B,20,T,h,w], timesteps int64[B], context [B,77+16T,ctx], fs int64[B], **ignored``) and the
same output ``[B,16,T,h,w]`` — but nothing is computed by PyTorch: ``forward`` enqueues hand-written HIP kernels
(include/geo4d_hip.h) on the current stream, is free of host synchronisation and therefore hipGraph-capturable.

Engine layout: activations are channels-last tokens ``[(b t) h w, C]`` in the storage dtype of the compute mode (bf16 by
default, f16; f32 for ``bf16x3`` = the mode that meets the 1e-3 parity bar, and for ``f32`` = exact parity, see precision.py); the frame-major token order is kept through the temporal layers (the temporal kernels
gather across T themselves), so no rearrange is ever materialised. Fusions vs. the reference op list: bias / timestep
embedding / residual adds / GEGLU / SiLU live in GEMM epilogues, q-k-v projections are one GEMM, nearest-2x upsample
is folded into the conv gather, cross-attention K/V of the (step-constant) context are projected once and cached, all
22 ResBlock embedding projections are one small GEMV.
"""
import math
import os

import torch
import torch.nn as nn

from . import ops, pack
from .precision import resolve as resolve_dtype   # compute mode: "bf16" | "f16" | "bf16x3" | "f32" (geo4d_amd/precision.py)


class ParamTree(nn.Module):
    """Parameter container whose ``state_dict`` names are given dotted names (no compute lives in modules)."""

    def __init__(self):
        super().__init__()

    def insert(self, dotted, shape):
        node = self
        parts = dotted.split(".")
        for part in parts[:-1]:
            child = node._modules.get(part)
            if child is None:
                child = ParamTree()
                node.add_module(part, child)
            node = child
        node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape), requires_grad=False))


def init_params_(module, seed=0):
    """Deterministic default init (checkpoints overwrite it): weights N(0, 1/fan_in), norm gains 1, biases 0. The normals come
    from ONE 4 Mi-sample draw that the tensors walk through cyclically (a 1.44 B-parameter model would otherwise spend minutes in
    the scalar CPU generator before every test and bench run)."""
    g = torch.Generator().manual_seed(seed)
    pool = torch.randn((1 << 22) + 8191, generator=g)
    pos = 0
    for name, p in module.named_parameters():
        if p.is_meta:
            continue
        if p.dim() <= 1:
            p.data.fill_(1.0 if name.endswith("weight") else 0.0)
        else:
            n, scale = p.numel(), 1.0 / math.sqrt(p[0].numel())
            flat = p.data.view(-1)
            done = 0
            while done < n:
                m = min(n - done, pool.numel() - pos)
                torch.mul(pool[pos:pos + m], scale, out=flat[done:done + m])
                done += m
                pos = (pos + m) % pool.numel()


# ------------------------------------------------------------------------------------------------------
# layout (which layers exist, in which order, under which state_dict prefix)
# ------------------------------------------------------------------------------------------------------
class Layer:
    __slots__ = ("kind", "prefix", "cin", "cout", "heads", "inner", "conv1d")

    def __init__(self, kind, prefix, cin=0, cout=0, heads=0, inner=0, conv1d=False):
        self.kind, self.prefix, self.cin, self.cout, self.heads, self.inner, self.conv1d = kind, prefix, cin, cout, heads, inner, conv1d


def build_layout(c):
    mc, mult, nres, dh = c["model_channels"], list(c["channel_mult"]), c["num_res_blocks"], c["num_head_channels"]
    attn_ds = set(c["attention_resolutions"])

    def transformers(base, first, ch):
        ls = [Layer("spatial", f"{base}.{first}", cin=ch, heads=ch // dh, inner=ch)]
        if c["temporal_attention"]:
            ls.append(Layer("temporal", f"{base}.{first + 1}", cin=ch, heads=ch // dh, inner=ch))
        return ls

    inputs = [[Layer("conv_in", "input_blocks.0.0", cin=c["in_channels"], cout=mc)]]
    skip_ch, ch, ds, n = [mc], mc, 1, 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            blk = [Layer("res", f"input_blocks.{n}.0", cin=ch, cout=m * mc)]
            ch = m * mc
            if ds in attn_ds:
                blk += transformers(f"input_blocks.{n}", 1, ch)
            inputs.append(blk); skip_ch.append(ch); n += 1
        if level != len(mult) - 1:
            inputs.append([Layer("down", f"input_blocks.{n}.0", cin=ch, cout=ch)])
            skip_ch.append(ch); n += 1; ds *= 2
    middle = [Layer("res", "middle_block.0", cin=ch, cout=ch)] + transformers("middle_block", 1, ch)
    middle.append(Layer("res", f"middle_block.{len(middle)}", cin=ch, cout=ch))
    outputs, n = [], 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            blk = [Layer("res", f"output_blocks.{n}.0", cin=ch + skip_ch.pop(), cout=m * mc)]
            ch = m * mc
            if ds in attn_ds:
                blk += transformers(f"output_blocks.{n}", 1, ch)
            if level and i == nres:
                blk.append(Layer("up", f"output_blocks.{n}.{len(blk)}", cin=ch, cout=ch))
                ds //= 2
            outputs.append(blk); n += 1
    init_attn = None
    if c["addition_attention"]:
        init_attn = Layer("temporal", "init_attn.0", cin=mc, heads=8, inner=8 * dh, conv1d=True)
    return inputs, middle, outputs, init_attn, ch


def _block_shapes(add, p, dim, ctx, image_cross):
    for a, kdim in (("attn1", dim), ("attn2", ctx if ctx else dim)):
        add(f"{p}.{a}.to_q.weight", (dim, dim))
        add(f"{p}.{a}.to_k.weight", (dim, kdim))
        add(f"{p}.{a}.to_v.weight", (dim, kdim))
        add(f"{p}.{a}.to_out.0.weight", (dim, dim))
        add(f"{p}.{a}.to_out.0.bias", (dim,))
    if image_cross and ctx:
        add(f"{p}.attn2.to_k_ip.weight", (dim, ctx))
        add(f"{p}.attn2.to_v_ip.weight", (dim, ctx))
    add(f"{p}.ff.net.0.proj.weight", (8 * dim, dim))
    add(f"{p}.ff.net.0.proj.bias", (8 * dim,))
    add(f"{p}.ff.net.2.weight", (dim, 4 * dim))
    add(f"{p}.ff.net.2.bias", (dim,))
    for n in ("norm1", "norm2", "norm3"):
        add(f"{p}.{n}.weight", (dim,))
        add(f"{p}.{n}.bias", (dim,))


def layer_shapes(add, L, c):
    p, emb_ch = L.prefix, 4 * c["model_channels"]
    if L.kind == "conv_in":
        add(p + ".weight", (L.cout, L.cin, 3, 3)); add(p + ".bias", (L.cout,))
    elif L.kind == "res":
        add(p + ".in_layers.0.weight", (L.cin,)); add(p + ".in_layers.0.bias", (L.cin,))
        add(p + ".in_layers.2.weight", (L.cout, L.cin, 3, 3)); add(p + ".in_layers.2.bias", (L.cout,))
        add(p + ".emb_layers.1.weight", (L.cout, emb_ch)); add(p + ".emb_layers.1.bias", (L.cout,))
        add(p + ".out_layers.0.weight", (L.cout,)); add(p + ".out_layers.0.bias", (L.cout,))
        add(p + ".out_layers.3.weight", (L.cout, L.cout, 3, 3)); add(p + ".out_layers.3.bias", (L.cout,))
        if L.cin != L.cout:
            add(p + ".skip_connection.weight", (L.cout, L.cin, 1, 1)); add(p + ".skip_connection.bias", (L.cout,))
        if c["temporal_conv"]:
            for name, ci in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
                q = f"{p}.temopral_conv.{name}"
                add(q + ".0.weight", (L.cout,)); add(q + ".0.bias", (L.cout,))
                add(f"{q}.{ci}.weight", (L.cout, L.cout, 3, 1, 1)); add(f"{q}.{ci}.bias", (L.cout,))
    elif L.kind in ("spatial", "temporal"):
        add(p + ".norm.weight", (L.cin,)); add(p + ".norm.bias", (L.cin,))
        tail = (1,) if L.conv1d else ()
        add(p + ".proj_in.weight", (L.inner, L.cin) + tail); add(p + ".proj_in.bias", (L.inner,))
        _block_shapes(add, p + ".transformer_blocks.0", L.inner, c["context_dim"] if L.kind == "spatial" else None,
                      c["image_cross_attention"] and L.kind == "spatial")
        add(p + ".proj_out.weight", (L.cin, L.inner) + tail); add(p + ".proj_out.bias", (L.cin,))
    elif L.kind == "down":
        add(p + ".op.weight", (L.cout, L.cin, 3, 3)); add(p + ".op.bias", (L.cout,))
    elif L.kind == "up":
        add(p + ".conv.weight", (L.cout, L.cin, 3, 3)); add(p + ".conv.bias", (L.cout,))


PRESPLIT = os.environ.get("GEO4D_X3_PRESPLIT", "1") != "0"
PRESPLIT_UP = os.environ.get("GEO4D_X3_PRESPLIT_UP", "1") != "0"
FUSED_CONCAT = os.environ.get("GEO4D_FUSED_CONCAT", "1") != "0"     # the skip concatenations' producers write into the consumer's buffer (no concat_channels launches); 0: A/B      # A/B switch of round 6: pre-split pass in front of the Upsample convolutions (U-Net and VAE)


# ------------------------------------------------------------------------------------------------------
class UNetModel(ParamTree):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0.0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None, use_scale_shift_norm=False,
                 resblock_updown=False, num_heads=-1, num_head_channels=-1, transformer_depth=1, use_linear=False,
                 use_checkpoint=False, temporal_conv=False, tempspatial_aware=False, temporal_attention=True,
                 use_relative_position=True, use_causal_attention=False, temporal_length=None, use_fp16=False,
                 addition_attention=False, temporal_selfatt_only=True, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False, default_fs=4, fs_condition=False, task_condition=False,
                 compute_dtype=None):
        super().__init__()
        unsupported = dict(dims=dims != 2, use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                           num_head_channels=num_head_channels != 64, transformer_depth=transformer_depth != 1,
                           use_linear=not use_linear, tempspatial_aware=tempspatial_aware,
                           use_relative_position=use_relative_position, use_causal_attention=use_causal_attention,
                           conv_resample=not conv_resample, task_condition=task_condition,
                           image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                           context_dim=context_dim is None)
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"geo4d_amd.UNetModel: option(s) {bad} are outside the Geo4D inference config "
                                      "(configs/inference_geo4d.yaml) and have no HIP path")
        self.cfg = dict(in_channels=in_channels, model_channels=model_channels, out_channels=out_channels,
                        num_res_blocks=num_res_blocks, attention_resolutions=list(attention_resolutions),
                        channel_mult=list(channel_mult), num_head_channels=num_head_channels, context_dim=context_dim,
                        temporal_conv=temporal_conv, temporal_attention=temporal_attention,
                        addition_attention=addition_attention, image_cross_attention=image_cross_attention,
                        default_fs=default_fs, fs_condition=fs_condition)
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.temporal_length, self.default_fs, self.fs_condition = temporal_length, default_fs, fs_condition
        self.use_checkpoint = use_checkpoint  # accepted and ignored: inference only (test_geo4d.py:321-322)
        self.dtype = torch.float32            # reference attribute: dtype of the module interface
        self.compute_dtype = resolve_dtype(compute_dtype)
        self.layout = build_layout(self.cfg)
        emb = 4 * model_channels
        self.insert("time_embed.0.weight", (emb, model_channels)); self.insert("time_embed.0.bias", (emb,))
        self.insert("time_embed.2.weight", (emb, emb)); self.insert("time_embed.2.bias", (emb,))
        if fs_condition:
            self.insert("fps_embedding.0.weight", (emb, model_channels)); self.insert("fps_embedding.0.bias", (emb,))
            self.insert("fps_embedding.2.weight", (emb, emb)); self.insert("fps_embedding.2.bias", (emb,))
        inputs, middle, outputs, init_attn, final_ch = self.layout
        for L in self.all_layers():
            layer_shapes(self.insert, L, self.cfg)
        self.insert("out.0.weight", (final_ch,)); self.insert("out.0.bias", (final_ch,))
        self.insert("out.2.weight", (out_channels, model_channels, 3, 3)); self.insert("out.2.bias", (out_channels,))
        init_params_(self)
        self._packed = None
        self._ctx_cache = {}
        self.generation = 0
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    def all_layers(self):
        inputs, middle, outputs, init_attn, _ = self.layout
        for blk in inputs:
            for L in blk:
                yield L
            if blk is inputs[0] and init_attn is not None:
                yield init_attn
        for L in middle:
            yield L
        for blk in outputs:
            for L in blk:
                yield L

    # ---- weight packing (once per load / dtype) ---------------------------------------------------------
    def invalidate(self):
        self._packed = None
        self._ctx_cache = {}
        self.generation = getattr(self, "generation", 0) + 1   # packed weights / K/V tensors of older captures are gone

    def set_compute_dtype(self, d):
        self.compute_dtype = resolve_dtype(d)
        self.invalidate()
        return self

    @property
    def presplit(self):
        """bf16x3: producers whose only consumers are GEMMs (GroupNorm / LayerNorm / attention / GEGLU epilogue) write the pre-split
        operand format, so conv_gemm's K loop does not split activation fragments (measured: 12-29 % of a GEMM's time)."""
        return bool(self.compute_dtype.x3) and PRESPLIT

    @property
    def storage_dtype(self):
        """torch dtype of the activations (and of the cached context K/V) in the current compute mode."""
        return self.compute_dtype.storage

    def _apply(self, fn, *a, **k):  # .cuda() / .to(): repack on the new device
        self.invalidate()
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def _pack(self):
        sd = dict(self.named_parameters())
        dev = sd["out.2.weight"].device
        if dev.type != "cuda":
            raise ops._lib.Geo4DNativeError("geo4d_amd.UNetModel runs only on a HIP device: call .cuda() first "
                                            "(there is no CPU fallback)")
        ops._lib.load()
        dt = self.compute_dtype
        P = {}
        f32 = lambda n: sd[n].float().contiguous()

        def norm(p):
            return f32(p + ".weight"), f32(p + ".bias")

        # bf16x3m (precision.py): which GEMM classes take the two-pass f16 form - their weights are packed as f16 hi | lo and the
        # producers of their A operands (GroupNorm / LayerNorm / the GEGLU epilogue) write f16 halves
        x2 = lambda cls: bool(dt.two_pass(cls) and self.presplit)
        lin_ln = pack.pack_linear_x2 if x2("ln") else pack.pack_linear

        def block(p, cross):
            # class "attn" (round 6): the spatial SELF-attention chain in f16 - LayerNorm writes f16 rows, q | k leave a two-pass GEMM as f16
            # rows, V^T a one-pass f16 GEMM (the weight rounded to f16 too: it is the MFMA's A side there), the attention kernel is the
            # single-pass f16 one, its f16 output feeds a two-pass to_out (tests/precision_sim.py "self-attention CHAIN")
            # class "tattn" (round 6): the TEMPORAL attention branch on f16 rows - its two-pass q | k | v projection (class "ln") writes f16 rows,
            # temporal_attn_kernel's f16 instantiation reads half the bytes (the kernel is HBM-bound; its arithmetic is fp32 VALU either way) and
            # writes f16 rows, which feed a two-pass to_out
            # class "cattn" (round 6): the spatial CROSS-attention on f16 rows - q leaves its two-pass projection (class "ln") as f16 rows, the cached
            # context K / V^T are kept as f16 copies, the dual-KV attention kernel runs its single-pass f16 instantiation, to_out is two-pass
            b = {"x2ln": x2("ln"), "x2ff": x2("ff"), "x2attn": bool(cross and x2("attn")), "x2tattn": bool(not cross and x2("tattn") and x2("ln")),
                 "x2cattn": bool(cross and x2("cattn") and x2("ln"))}
            for a in ("attn1", "attn2"):
                x2o = False
                if a == "attn2" and cross:
                    b[a + ".q"] = lin_ln(sd[f"{p}.{a}.to_q.weight"], dt)            # LayerNorm -> q (plain rows): class "ln"
                    x2o = b["x2cattn"]
                elif cross:      # spatial self-attention: q|k fused, V projected transposed (flash kernel wants V^T)
                    wqk = torch.cat([sd[f"{p}.{a}.to_q.weight"], sd[f"{p}.{a}.to_k.weight"]], 0)
                    if b["x2attn"]:
                        b[a + ".qk"] = pack.pack_linear_x2(wqk, dt)
                        b[a + ".v"] = pack.pack_linear(sd[f"{p}.{a}.to_v.weight"], "f16")
                        x2o = True
                    else:
                        b[a + ".qk"] = pack.pack_linear(wqk, dt)
                        b[a + ".v"] = pack.pack_linear(sd[f"{p}.{a}.to_v.weight"], dt)
                else:            # temporal attention: LayerNorm -> q | k | v (plain rows): class "ln"
                    b[a + ".qkv"] = lin_ln(torch.cat([sd[f"{p}.{a}.to_q.weight"], sd[f"{p}.{a}.to_k.weight"],
                                                      sd[f"{p}.{a}.to_v.weight"]], 0), dt)
                    x2o = b["x2tattn"]
                b[a + ".o"] = ((pack.pack_linear_x2 if x2o else pack.pack_linear)(sd[f"{p}.{a}.to_out.0.weight"], dt), f32(f"{p}.{a}.to_out.0.bias"))
            if x2("ff"):
                b["ff1"] = pack.pack_geglu_x2(sd[p + ".ff.net.0.proj.weight"], sd[p + ".ff.net.0.proj.bias"], dt)
                b["ff2"] = (pack.pack_linear_x2(sd[p + ".ff.net.2.weight"], dt), f32(p + ".ff.net.2.bias"))
            else:
                b["ff1"] = pack.pack_geglu(sd[p + ".ff.net.0.proj.weight"], sd[p + ".ff.net.0.proj.bias"], dt)
                b["ff2"] = (pack.pack_linear(sd[p + ".ff.net.2.weight"], dt), f32(p + ".ff.net.2.bias"))
            for n in ("norm1", "norm2", "norm3"):
                b[n] = norm(f"{p}.{n}")
            return b

        emb_w, emb_b, off = [], [], 0
        k_text, k_img, kv_off = [], [], 0
        for L in self.all_layers():
            p, e = L.prefix, {}
            if L.kind == "conv_in":
                e["w"] = pack.pack_conv2d(sd[p + ".weight"], dt); e["b"] = f32(p + ".bias")
                e["cpad"] = pack.pad_to(L.cin, ops.k_align(dt))
            elif L.kind == "res":
                e["gn1"], e["gn2"] = norm(p + ".in_layers.0"), norm(p + ".out_layers.0")
                # bf16x3m: the ResBlocks' two 3x3 convolutions take the two-pass f16 form (precision.py; their GroupNorms then write f16 hi | lo)
                pk = pack.pack_conv2d_x2 if x2("conv3x3") else pack.pack_conv2d
                e["x2"] = pk is pack.pack_conv2d_x2
                e["w1"], e["b1"] = pk(sd[p + ".in_layers.2.weight"], dt), f32(p + ".in_layers.2.bias")
                e["w2"], e["b2"] = pk(sd[p + ".out_layers.3.weight"], dt), f32(p + ".out_layers.3.bias")
                emb_w.append(sd[p + ".emb_layers.1.weight"].float()); emb_b.append(sd[p + ".emb_layers.1.bias"].float())
                e["emb"] = (off, off + L.cout); off += L.cout
                if L.cin != L.cout:
                    e["skip"] = (pack.pack_linear(sd[p + ".skip_connection.weight"], dt), f32(p + ".skip_connection.bias"))
                if self.cfg["temporal_conv"]:
                    e["tc"] = []
                    e["x2t"] = x2("tconv")
                    pkt = pack.pack_conv3d_t_x2 if e["x2t"] else pack.pack_conv3d_t
                    for name, ci in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
                        q = f"{p}.temopral_conv.{name}"
                        e["tc"].append((norm(q + ".0"), pkt(sd[f"{q}.{ci}.weight"], dt), f32(f"{q}.{ci}.bias")))
            elif L.kind in ("spatial", "temporal"):
                e["norm"] = norm(p + ".norm")
                e["x2in"] = x2("proj_in")
                e["in"] = ((pack.pack_linear_x2 if e["x2in"] else pack.pack_linear)(sd[p + ".proj_in.weight"], dt), f32(p + ".proj_in.bias"))
                e["out"] = (pack.pack_linear(sd[p + ".proj_out.weight"], dt), f32(p + ".proj_out.bias"))
                e["blk"] = block(p + ".transformer_blocks.0", cross=L.kind == "spatial")
                if L.kind == "spatial":
                    a = p + ".transformer_blocks.0.attn2"
                    k_text.append(sd[a + ".to_k.weight"])
                    e["wv_text"] = pack.pack_linear(sd[a + ".to_v.weight"], dt)
                    if self.cfg["image_cross_attention"]:
                        k_img.append(sd[a + ".to_k_ip.weight"])
                        e["wv_img"] = pack.pack_linear(sd[a + ".to_v_ip.weight"], dt)
                    e["kv"] = (kv_off, L.inner); kv_off += L.inner
            elif L.kind == "down":
                e["w"], e["b"] = pack.pack_conv2d(sd[p + ".op.weight"], dt), f32(p + ".op.bias")
            elif L.kind == "up":
                e["w"], e["b"] = pack.pack_conv2d(sd[p + ".conv.weight"], dt), f32(p + ".conv.bias")
            P[p] = e
        P["emb_w"], P["emb_b"] = torch.cat(emb_w, 0).contiguous(), torch.cat(emb_b, 0).contiguous()
        P["k_text"] = pack.pack_linear(torch.cat(k_text, 0), dt)
        P["k_img"] = pack.pack_linear(torch.cat(k_img, 0), dt) if k_img else None
        P["time"] = [f32("time_embed.0.weight"), f32("time_embed.0.bias"), f32("time_embed.2.weight"), f32("time_embed.2.bias")]
        if self.fs_condition:
            P["fps"] = [f32("fps_embedding.0.weight"), f32("fps_embedding.0.bias"), f32("fps_embedding.2.weight"), f32("fps_embedding.2.bias")]
        P["out_gn"] = norm("out.0")
        P["out_w"], P["out_b"] = pack.pack_conv2d(sd["out.2.weight"], dt), f32("out.2.bias")
        half = self.model_channels // 2
        P["freqs"] = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(dev)
        self._packed = P
        return P

    # ---- cross-attention K/V of the context: constant over DDIM steps -> projected once --------------------
    def _context_kv(self, P, context, B, T):
        """K / V^T of the cross-attention context for every spatial transformer, cached per context BUFFER.

        The cache key is the buffer identity (pointer, shape, compute dtype); the tensor's ``_version`` is stored in the
        entry. A hit whose version moved (the caller refilled the same buffer, e.g. the sampler's static conditioning
        between windows) re-projects IN PLACE into the same K/V tensors, so device pointers baked into a captured
        hipGraph stay valid and see the new values. Evicting an entry or re-packing the weights bumps ``generation``;
        samplers put it in their graph key and re-capture when it moves (a captured step must never outlive its K/V)."""
        key = (context.data_ptr(), tuple(context.shape), context.dtype, self.compute_dtype)
        hit = self._ctx_cache.get(key)
        if hit is not None and hit["version"] == context._version:
            return hit["kv"]
        L = context.shape[1]
        if L != 77 + 16 * T:
            raise NotImplementedError(f"context length {L} != 77 + 16*T ({77 + 16 * T}): only the per-frame image-token "
                                      "layout of Geo4D inference (openaimodel3d.py:576-580) has a HIP path")
        dt = self.storage_dtype
        ctx = context.to(dt)
        text = ctx[:, :77].reshape(B * 77, -1).contiguous()
        img = ctx[:, 77:].reshape(B * T * 16, -1).contiguous()
        if hit is None:
            k_t = k_i = None
            vt_t, vt_i = {}, {}
        else:
            k_t, k_i, vt_t, vt_i = hit["kv"][:4]
        k_t = ops.linear(text, P["k_text"], out=k_t)                          # every layer's text keys: [B*77, sum C]
        k_i = ops.linear(img, P["k_img"], out=k_i) if P["k_img"] is not None else None
        for L in self.all_layers():                                            # values, transposed per layer (V^T rows = channels)
            if L.kind != "spatial":
                continue
            e = P[L.prefix]
            v = vt_t.get(L.prefix)
            if v is None:                                                      # 77 keys padded to a 16-byte multiple (pad stays 0)
                v = vt_t[L.prefix] = torch.zeros((B, L.inner, 80), device=ctx.device, dtype=dt)
            for b in range(B):
                ops.linear_t(e["wv_text"], text[b * 77:(b + 1) * 77], out=v[b])
            if k_i is not None:
                vt_i[L.prefix] = ops.linear_t(e["wv_img"], img, out=vt_i.get(L.prefix))   # [C, B*T*16], frame f at columns 16f..
        # f16 copies for the "cattn" class (refreshed in place like the f32 ones: a captured graph keeps valid pointers)
        h16 = hit["kv"][5] if hit is not None else {}
        if self.compute_dtype.two_pass("cattn") and self.compute_dtype.two_pass("ln") and self.presplit:
            def half(name, t):
                if t is None:
                    return None
                c = h16.get(name)
                if c is None:
                    c = h16[name] = torch.empty(t.shape, device=t.device, dtype=torch.float16)
                c.copy_(t)
                return c
            half("k_t", k_t); half("k_i", k_i)
            for pfx, v in vt_t.items():
                half("vt_t." + pfx, v)
            for pfx, v in vt_i.items():
                half("vt_i." + pfx, v)
        kv = (k_t, k_i, vt_t, vt_i, context, h16)   # keeps `context` alive so its data_ptr stays unique
        if hit is None:
            if len(self._ctx_cache) >= 8:
                self._ctx_cache.pop(next(iter(self._ctx_cache)))
                self.generation += 1             # a captured graph may have been reading the evicted tensors
            self._ctx_cache[key] = {"version": context._version, "kv": kv}
        else:
            hit["version"] = context._version
        return kv

    def prepare_context(self, context, T):
        """Project (or refresh in place) the context K/V outside of a captured step; returns the token a sampler compares
        before replaying a graph: it changes whenever pointers the graph baked in may have been invalidated."""
        P = self._packed or self._pack()
        kv = self._context_kv(P, context, context.shape[0], T)
        return (self.generation, kv[0].data_ptr())

    # ---- layer executors (all enqueue HIP kernels; tensors are token matrices [(b t) hw, C]) -------------
    def _res(self, e, L, h, emb_all, B, T, H, W, out=None):
        """`out` (every layer executor): where the layer's LAST launch writes its result - a column view of a wider buffer (FUSED_CONCAT)."""
        F_, HW = B * T, H * W
        sp = self.presplit
        sp3 = "f16" if e.get("x2") else sp           # operand format of the two 3x3 convolutions
        a = ops.groupnorm(h, *e["gn1"], F=F_, HW=HW, eps=1e-5, silu=True, split_out=sp3)
        lo, hi = e["emb"]
        h1, _, _ = ops.conv2d(a, e["w1"], e["b1"], F=F_, Hin=H, Win=W, KH=3, KW=3, pad=1, rowbias=emb_all[:, lo:hi],
                              rowbias_div=T * HW, gn_stats=True)      # gn_stats: the epilogue sums the next GroupNorm's statistics
        a = ops.groupnorm(h1, *e["gn2"], F=F_, HW=HW, eps=1e-5, silu=True, split_out=sp3)
        skip = ops.linear(h, *e["skip"]) if "skip" in e else h
        h2, _, _ = ops.conv2d(a, e["w2"], e["b2"], F=F_, Hin=H, Win=W, KH=3, KW=3, pad=1, residual=skip, gn_stats=True, out=None if "tc" in e else out)
        if "tc" in e:
            y = h2
            spt = "f16" if e.get("x2t") else sp
            for i, (gn, w, b) in enumerate(e["tc"]):
                a = ops.groupnorm(y, *gn, F=F_, HW=HW, eps=1e-5, frames_per_stat=T, silu=True, split_out=spt)
                y = ops.conv_temporal(a, w, b, B=B, T=T, HW=HW, residual=h2 if i == 3 else None, gn_stats=True, out=out if i == 3 else None)
            h2 = y
        return h2

    def _ff(self, blk, x):
        # (round 5 also tried the ff-out epilogue writing proj_out's pre-split operand: no gain in either mode, profiles/r05_two_pass_f16.md)
        sp = "f16" if blk.get("x2ff") else self.presplit      # two-pass f16 feed-forward: LayerNorm and the GEGLU epilogue write f16 halves
        g = ops.linear(ops.layernorm(x, *blk["norm3"], split_out=sp), *blk["ff1"], act=2, split_out=sp)
        return ops.linear(g, *blk["ff2"], residual=x)

    def _spatial(self, e, L, h, kv, B, T, H, W, out=None):
        F_, N, C_, heads = B * T, H * W, L.inner, L.heads
        blk = e["blk"]
        sp = self.presplit
        x = ops.linear(ops.groupnorm(h, *e["norm"], F=F_, HW=N, eps=1e-6, split_out="f16" if e.get("x2in") else sp), *e["in"])
        x3 = self.compute_dtype.x3
        if blk.get("x2attn"):
            # bf16x3m class "attn": the whole self-attention branch in f16 (one MFMA per product inside the attention kernel, two in the
            # projections around it); everything it touches is a normalised branch activation, the residual stream `x` stays f32
            n1 = ops.layernorm(x, *blk["norm1"], split_out="f16")
            qk = ops.linear(n1, blk["attn1.qk"], split_out="f16")                         # f16 rows [M, 2C]
            vt, npad = ops.linear_t_batched(blk["attn1.v"], n1, F_, N)                    # f16 V^T per frame: [F, C, Npad]
            att = ops.attention(qk[:, :C_], [(qk[:, C_:], vt.reshape(-1, npad), N, 1, C_ * npad)], B=F_, H=heads, Nq=N, scale=0.125)
        elif sp and N % 8 == 0:
            n1 = ops.layernorm(x, *blk["norm1"], split_out=sp)
            # bf16x3: q | k and V^T leave their projections in the pre-split operand format (o_split epilogue), so the attention kernel
            # does not split K / V^T fragments per tile and wave (round 4: -30 % of its VALU instructions). 2 bf16 per element:
            qk = ops.linear(n1, blk["attn1.qk"], split_out=True)                          # SplitAct [M, 2 * 2C]
            vt, npad = ops.linear_t_batched(blk["attn1.v"], n1, F_, N, split_out=True)    # SplitAct [F, C, 2 * Npad]
            att = ops.attention(qk[:, :2 * C_], [(qk[:, 2 * C_:], vt.reshape(-1, 2 * npad), N, 1, C_ * 2 * npad)], B=F_, H=heads, Nq=N,
                                scale=0.125, x3=True, split_out=sp, qkv_split=True)
        else:
            n1 = ops.layernorm(x, *blk["norm1"], split_out=sp)
            qk = ops.linear(n1, blk["attn1.qk"])
            vt, npad = ops.linear_t_batched(blk["attn1.v"], n1, F_, N)             # V^T per frame: [F, C, Npad]
            att = ops.attention(qk[:, :C_], [(qk[:, C_:], vt.reshape(-1, npad), N, 1, C_ * npad)], B=F_, H=heads, Nq=N, scale=0.125, x3=x3, split_out=sp)
        x = ops.linear(att, *blk["attn1.o"], residual=x)
        c16 = blk.get("x2cattn")
        q = ops.linear(ops.layernorm(x, *blk["norm2"], split_out="f16" if blk.get("x2ln") else sp), blk["attn2.q"], split_out="f16" if c16 else False)
        k_t, k_i, vt_t, vt_i, _, h16 = kv
        off, _ = e["kv"]
        if c16:
            sets = [(h16["k_t"][:, off:off + C_], h16["vt_t." + L.prefix].reshape(-1, 80), 77, T, C_ * 80)]
            if k_i is not None:
                sets.append((h16["k_i"][:, off:off + C_], h16["vt_i." + L.prefix], 16, 1, 16))
            att = ops.attention(q, sets, B=F_, H=heads, Nq=N, scale=0.125)
        else:
            sets = [(k_t[:, off:off + C_], vt_t[L.prefix].reshape(-1, 80), 77, T, C_ * 80)]
            if k_i is not None:
                sets.append((k_i[:, off:off + C_], vt_i[L.prefix], 16, 1, 16))
            att = ops.attention(q, sets, B=F_, H=heads, Nq=N, scale=0.125, x3=x3, split_out=sp)
        x = ops.linear(att, *blk["attn2.o"], residual=x)
        x = self._ff(blk, x)
        return ops.linear(x, *e["out"], residual=h, gn_stats=True, out=out)

    def _temporal(self, e, L, h, B, T, H, W, out=None):
        F_, HW, C_, heads = B * T, H * W, L.inner, L.heads
        blk = e["blk"]
        sp = self.presplit
        x = ops.linear(ops.groupnorm(h, *e["norm"], F=F_, HW=HW, eps=1e-6, frames_per_stat=T, split_out="f16" if e.get("x2in") else sp), *e["in"])
        t16 = blk.get("x2tattn")
        for a, n in (("attn1", "norm1"), ("attn2", "norm2")):
            qkv = ops.linear(ops.layernorm(x, *blk[n], split_out="f16" if blk.get("x2ln") else sp), blk[a + ".qkv"], split_out="f16" if t16 else False)
            att = ops.temporal_attention(qkv[:, :C_], qkv[:, C_:2 * C_], qkv[:, 2 * C_:], B=B, T=T, HW=HW, H=heads, scale=0.125, split_out=False if t16 else sp)
            x = ops.linear(att, *blk[a + ".o"], residual=x)
        x = self._ff(blk, x)
        return ops.linear(x, *e["out"], residual=h, gn_stats=True, out=out)

    def _run(self, P, layers, h, emb_all, kv, B, T, H, W, out=None):
        """`out`: a function (rows) -> the [rows, C] view the block's LAST layer writes its result into (or None): the next skip
        concatenation's buffer, so that torch.cat([h, hs.pop()], 1) (openaimodel3d.py:624-626) never runs as a copy (FUSED_CONCAT)."""
        for L in layers:
            e = P[L.prefix]
            o = None
            if out is not None and L is layers[-1]:
                Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if L.kind == "down" else (2 * H, 2 * W) if L.kind == "up" else (H, W)
                o = out(B * T * Ho * Wo)
            if L.kind == "res":
                h = self._res(e, L, h, emb_all, B, T, H, W, out=o)
            elif L.kind == "spatial":
                h = self._spatial(e, L, h, kv, B, T, H, W, out=o)
            elif L.kind == "temporal":
                h = self._temporal(e, L, h, B, T, H, W, out=o)
            elif L.kind == "down":
                h, H, W = ops.conv2d(h, e["w"], e["b"], F=B * T, Hin=H, Win=W, KH=3, KW=3, stride=2, pad=1, gn_stats=True, out=o)
            elif L.kind == "up":
                # (bf16x3 modes: the stream is split ONCE by its own pass instead of per K slab and wave inside the conv - same bits, PRESPLIT_UP)
                a = ops.presplit(h) if (self.presplit and PRESPLIT_UP) else h
                h, H, W = ops.conv2d(a, e["w"], e["b"], F=B * T, Hin=H, Win=W, KH=3, KW=3, pad=1, ups=2, out=o)
        return h, H, W

    # ---- public forward -------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, timesteps, context=None, features_adapter=None, fs=None, task=None, c_concat=None, **kwargs):
        """x [B, C, T, h, w] (already channel-concatenated, or pass the second part as ``c_concat``);
        extra kwargs (cfg_img, unconditional_conditioning_img_nonetext, ...) are accepted and ignored exactly like
        the reference forward (openaimodel3d.py:558, ddim.py:217)."""
        if features_adapter is not None or task is not None:
            raise NotImplementedError("features_adapter / task conditioning are not part of Geo4D inference")
        if timesteps.dim() != 1:
            raise NotImplementedError("per-frame timesteps (2-D) have no HIP path")
        P = self._packed or self._pack()
        B, C0, T, H, W = x.shape
        C1 = 0 if c_concat is None else c_concat.shape[1]
        assert C0 + C1 == self.in_channels, f"expected {self.in_channels} input channels, got {C0}+{C1}"
        if T > 16:
            raise NotImplementedError("temporal_length > 16 has no HIP path (configs/inference_geo4d.yaml:89)")
        dt = self.storage_dtype
        inputs, middle, outputs, init_attn, _ = self.layout
        # embeddings (fp32, M = B rows)
        t_emb = ops.timestep_embedding(timesteps.to(torch.int64), P["freqs"])
        w0, b0, w2, b2 = P["time"]
        emb = ops.linear_small(ops.linear_small(t_emb, w0, b0, act_out=True), w2, b2)
        if self.fs_condition:
            if fs is None:
                fs = torch.full((B,), self.default_fs, dtype=torch.int64, device=x.device)
            w0, b0, w2, b2 = P["fps"]
            f_emb = ops.timestep_embedding(fs.to(torch.int64), P["freqs"])
            emb = ops.linear_small(ops.linear_small(f_emb, w0, b0, act_out=True), w2, b2, add=emb)
        emb_all = ops.linear_small(emb, P["emb_w"], P["emb_b"], act_in=True)      # every ResBlock's emb_layers at once
        kv = self._context_kv(P, context, B, T)
        # tokens
        e0 = P[inputs[0][0].prefix]
        h = ops.tokens_from_ncthw(x.float().contiguous(), None if c_concat is None else c_concat.float().contiguous(), e0["cpad"], dt)
        # FUSED_CONCAT (round 6): output block j reads torch.cat([h, hs.pop()], 1). Instead of copying both halves into a new tensor
        # (concat_channels: 12 launches per forward), both PRODUCERS write straight into that block's buffer: the input block's last launch
        # into columns [Ch, Ch + Cs) when the skip is made (its result then flows on as a column view: every kernel takes a row pitch), the
        # previous output block's (or the middle block's) last launch into columns [0, Ch). Same kernels, same operands: same bits.
        n_in = len(inputs)
        cats = {}                                             # index of the consuming output block -> its concatenated input buffer

        def half(j, left):
            """-> function(rows) returning the [rows, C] view of output block j's input that a producer fills."""
            if not FUSED_CONCAT or j >= len(outputs):
                return None
            ccat = outputs[j][0].cin                          # = Ch (incoming h) + Cs (the skip popped for block j)

            def view(rows, j=j, left=left, ccat=ccat):
                buf = cats.get(j)
                if buf is None:
                    buf = cats[j] = torch.empty((rows, ccat), device=x.device, dtype=dt)
                assert buf.shape[0] == rows, "latent height/width must be multiples of 8"
                cs = skip_ch[n_in - 1 - j]
                return buf[:, :ccat - cs] if left else buf[:, ccat - cs:]
            return view
        skip_ch = [blk[-1].cout if blk[-1].kind != "temporal" and blk[-1].kind != "spatial" else blk[-1].cin for blk in inputs]
        o0 = half(n_in - 1, False)
        if init_attn is not None:
            h, _, _ = ops.conv2d(h, e0["w"], e0["b"], F=B * T, Hin=H, Win=W, KH=3, KW=3, pad=1, gn_stats=True)
            h = self._temporal(P[init_attn.prefix], init_attn, h, B, T, H, W, out=None if o0 is None else o0(B * T * H * W))
        else:
            h, _, _ = ops.conv2d(h, e0["w"], e0["b"], F=B * T, Hin=H, Win=W, KH=3, KW=3, pad=1, gn_stats=True, out=None if o0 is None else o0(B * T * H * W))
        hs = [(h, H, W)]
        for i, blk in enumerate(inputs[1:], start=1):
            h, H, W = self._run(P, blk, h, emb_all, kv, B, T, H, W, out=half(n_in - 1 - i, False))
            hs.append((h, H, W))
        h, H, W = self._run(P, middle, h, emb_all, kv, B, T, H, W, out=half(0, True))
        for j, blk in enumerate(outputs):
            s, sh, sw = hs.pop()
            assert (sh, sw) == (H, W), "latent height/width must be multiples of 8"
            hc = cats.pop(j) if j in cats else ops.concat_channels(h, s)
            h, H, W = self._run(P, blk, hc, emb_all, kv, B, T, H, W, out=half(j + 1, True))
        a = ops.groupnorm(h, *P["out_gn"], F=B * T, HW=H * W, eps=1e-5, silu=True, split_out=self.presplit)
        y, _, _ = ops.conv2d(a, P["out_w"], P["out_b"], F=B * T, Hin=H, Win=W, KH=3, KW=3, pad=1, T=T, out_nchw=True,
                             out_dtype=torch.float32)
        return y if x.dtype == torch.float32 else y.to(x.dtype)
