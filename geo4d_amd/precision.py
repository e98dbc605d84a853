"""Compute-precision modes of the HIP engine.

=========  ===================  ============================  ==========================================================
name       activation storage   GEMM / attention arithmetic   point-map parity vs the fp32 reference (1e-3 is the bar)
=========  ===================  ============================  ==========================================================
``bf16``   bfloat16             bf16 MFMA, fp32 accumulate    ~2e-2  (8-bit mantissas through ~300 GEMMs)
``f16``    float16              f16 MFMA, fp32 accumulate     ~2.5e-3
``bf16x3`` float32              3 bf16 MFMAs per product on   < 1e-4: meets the bar at ~1/3 of the bf16 MFMA rate
                                a hi/lo split of each operand
``bf16x3m`` float32 streams;    bf16x3, except the GEMMs fed  1.8e-4 against the REFERENCE's own 50-step window at
           f16 rows for         by a normalised branch        BASELINE size (tests/golden/fullsize_ddim50.pt,
           branch activations   activation (TWO_PASS_CLASSES  tests/test_fullsize_gpu.py); 6.5e-4 on the tiny smoke
                                below): two f16 MFMAs per     window (its worst case)
                                product on an f16 activation
                                x an f16 hi + lo weight; the
                                three attention kernels: ONE
                                f16 MFMA per product
``f32``    float32              v_mfma_f32_32x32x2_f32        ~2e-6 (exact f32; 1/16 of the bf16 MFMA rate)
=========  ===================  ============================  ==========================================================

``bf16x3m`` (round 5, "mixed passes"): the error budget of the 1e-3 bar is spent where it buys MFMA time. Rounding ONLY the
A operand of the 3x3 convolutions to f16 (weights kept to ~22 bits as f16 hi + lo) moves the point map by 1.3e-4 .. 2.9e-4
on the simulated window (tests/precision_sim.py: 3 / 10 DDIM steps), an order of magnitude less than rounding every GEMM
input (1.2e-3), because these 44 + 3 x 24 launches are 33 % of the FLOPs but a small share of the network's rounding points;
the same rounding on the projections alone costs 5.8e-4, on everything 8.3e-4 - so only the convolutions take it.

Why ``bf16x3`` exists: rounding ONLY the weights of the U-Net to f16 (activations exact) already costs 1.4e-3 on the point
map, rounding only the GEMM inputs another 1.2e-3 (tests/precision_sim.py, tests/test_precision_floor.py) — no single pass
over 11-bit (f16) or 8-bit (bf16) mantissas can meet 1e-3 on this network, whatever is done to the residual streams. The
split ``x = hi + lo`` (two bf16) carries ~16 mantissa bits; ``x.w ~ hi.hi + hi.lo + lo.hi`` needs three MFMAs and an fp32
accumulator, which is what the matrix cores provide. Activations stay f32 in memory (every HBM-bound kernel simply runs its
f32 instantiation), weights are split once at pack time.
"""
import os

import torch


# Which classes of GEMMs the bf16x3m mode runs in the two-pass f16 form (comma list in $GEO4D_TWO_PASS; A/B runs and the error-budget
# table of profiles/r05_two_pass_f16.md): conv3x3 = U-Net ResBlock convolutions, vae3x3 = VAE decoder ResnetBlock convolutions,
# tconv = temporal 3-tap convolutions, proj_in = the transformers' GroupNorm -> proj_in linears, ln = the LayerNorm-fed projections that
# write plain rows (cross-attention q, temporal q | k | v), ff = the GEGLU feed-forward (both linears), attn (round 6) = the spatial
# self-attention branch: LayerNorm -> f16 rows -> two-pass q | k + one-pass f16 V^T -> the single-pass f16 attention kernel -> two-pass to_out;
# cattn = the spatial cross-attention on f16 rows (f16 q, f16 copies of the cached context K / V^T, single-pass f16 dual-KV kernel, two-pass
# to_out); tattn = the temporal attention on f16 rows (its q | k | v projection writes f16 rows, the HBM-bound kernel reads half the bytes,
# two-pass to_out). Error / frames/s per class: profiles/r06_attention_chains_ab.md.
# Default = classes whose A operand is a NORMALISED branch activation (GroupNorm / LayerNorm / GEGLU outputs) AND that pay for their
# error (profiles/r05_two_pass_f16.md: frames/s and point-map error per class). proj_in is off by default: 3.7e-4 on the simulated smoke
# window (tests/precision_sim.py) for 0.2 % of the step. Built, measured and REMOVED in round 5: the classes that round a RESIDUAL STREAM
# to f16 (proj_out; the raw-activation launches: down / up samplers, skip connections, VAE upsamplers) - at BASELINE size they took the
# 50-step point-map drift from 1.1e-4 to 5.3e-4 (the tiny smoke window to 9.98e-4) for +1 % frames/s.
TWO_PASS_CLASSES = frozenset(c.strip() for c in os.environ.get("GEO4D_TWO_PASS", "conv3x3,vae3x3,tconv,ln,ff,attn,cattn,tattn").split(",") if c.strip())


class Precision:
    __slots__ = ("name", "storage", "x3", "two_pass_conv")

    def __init__(self, name, storage, x3=False, two_pass_conv=False):
        self.name, self.storage, self.x3, self.two_pass_conv = name, storage, x3, two_pass_conv

    def __repr__(self):
        return f"Precision({self.name})"

    def __eq__(self, other):
        other = _coerce(other)
        return isinstance(other, Precision) and other.name == self.name

    def __hash__(self):
        return hash(self.name)

    def two_pass(self, cls):
        """Does this mode run GEMM class `cls` (see TWO_PASS_CLASSES) in the two-pass f16 form?"""
        return bool(self.two_pass_conv) and cls in TWO_PASS_CLASSES


BF16 = Precision("bf16", torch.bfloat16)
F16 = Precision("f16", torch.float16)
F32 = Precision("f32", torch.float32)
BF16X3 = Precision("bf16x3", torch.float32, x3=True)
BF16X3M = Precision("bf16x3m", torch.float32, x3=True, two_pass_conv=True)

_BY_NAME = {"bf16": BF16, "bfloat16": BF16, "f16": F16, "fp16": F16, "float16": F16, "f32": F32, "fp32": F32, "float32": F32,
            "bf16x3": BF16X3, "bf16_3x": BF16X3, "3xbf16": BF16X3, "bf16x3m": BF16X3M, "mixed": BF16X3M}
_BY_TORCH = {torch.bfloat16: BF16, torch.float16: F16, torch.float32: F32}


def _coerce(d):
    if isinstance(d, Precision):
        return d
    if isinstance(d, torch.dtype):
        return _BY_TORCH.get(d)
    if isinstance(d, str):
        return _BY_NAME.get(d.lower())
    return None


def resolve(d=None):
    """None -> $GEO4D_DTYPE or bf16; str / torch.dtype / Precision -> Precision."""
    if d is None:
        d = os.environ.get("GEO4D_DTYPE", "bf16")
    p = _coerce(d)
    if p is None:
        raise ValueError(f"geo4d_amd: unknown compute dtype {d!r} (bf16, f16, bf16x3, bf16x3m, f32)")
    return p
