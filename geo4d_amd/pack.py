"""One-time weight re-layout from the reference ``state_dict`` format into the kernel format.

Runs once per model load (not on the hot path): conv weights become ``[Cout][taps][Cin_pad]`` K-major rows,
GEGLU projections get their value/gate rows interleaved in blocks of 32 so the GEMM epilogue can multiply them.
"""
import torch

from .ops import X2Weight, k_align
from .precision import resolve


def split_bf16(w):
    """[N, K] fp32 (K % 8 == 0) -> the pre-split operand format of the bf16x3 GEMM: a bf16 tensor [N, 2K] holding, per 8
    K-elements, [8 x hi | 8 x lo] with hi = bf16(w), lo = bf16(w - hi) — 32 bytes, the footprint of the 8 f32 it replaces."""
    n, k = w.shape
    assert k % 8 == 0
    w = w.float()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.stack([hi.reshape(n, k // 8, 8), lo.reshape(n, k // 8, 8)], dim=2).reshape(n, 2 * k).contiguous()


def split_f16(w):
    """[N, K] fp32 (K % 8 == 0) -> the pre-split weight of the two-pass f16 GEMM (include/geo4d_hip.h dtype 4): a float16 tensor
    [N, 2K] holding, per 8 K-elements, [8 x hi | 8 x lo] with hi = f16(s w), lo = f16(s w - hi) and s a power of two that puts the
    largest |w| at 2^13..2^14 (hi well inside the f16 range, lo of every weight that matters a NORMAL f16 number: hi + lo then carries
    ~22 bits). The launch multiplies its accumulators by 1 / s (`X2Weight._x2_alpha`, exact), before bias / residual.
    One scale per TENSOR is enough (ADVICE r5 asked for one per output row): a weight far below the tensor's largest one has a SUBNORMAL
    lo half, whose absolute spacing is 2^-24 of the scaled range - hi + lo then carries an absolute error <= 2^-38 |w|_max, i.e. full
    22-bit relative precision for every |w| >= 2^-16 |w|_max and an error of 2^-38 |w|_max below that: against a row whose rms is even
    1000x below its outlier that is 2^-28 of the row's contribution (tests/test_f16x2_gpu.py::test_outlier_heavy_weights)."""
    n, k = w.shape
    assert k % 8 == 0
    w = w.float()
    amax = float(w.abs().max())
    e = 0 if amax == 0.0 or not torch.isfinite(torch.tensor(amax)) else 13 - int(torch.floor(torch.log2(torch.tensor(amax))))
    e = max(-24, min(24, e))
    ws = w * (2.0 ** e)
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    out = torch.stack([hi.reshape(n, k // 8, 8), lo.reshape(n, k // 8, 8)], dim=2).reshape(n, 2 * k).contiguous()
    return X2Weight.wrap(out, 2.0 ** -e)


def pack_conv2d_x2(w, dtype, cin_pad=None):
    """nn.Conv2d weight -> the two-pass f16 operand ([Cout, 2 * KH*KW*Cin_pad] float16 + `_x2_alpha`), same K order as pack_conv2d."""
    co, ci, kh, kw = w.shape
    cp = cin_pad or pad_to(ci, k_align(dtype))
    w = w.permute(0, 2, 3, 1)
    if cp != ci:
        w = torch.nn.functional.pad(w, (0, cp - ci))
    return split_f16(w.reshape(co, kh * kw * cp))


def pack_conv3d_t_x2(w, dtype):
    """nn.Conv3d weight [Cout, Cin, 3, 1, 1] -> the two-pass f16 operand, same K order as pack_conv3d_t."""
    co, ci, kt, kh, kw = w.shape
    assert kh == 1 and kw == 1
    return split_f16(w.reshape(co, ci, kt).permute(0, 2, 1).reshape(co, kt * ci))


def pack_linear_x2(w, dtype):
    """nn.Linear / 1x1 conv / Conv1d(k = 1) weight -> the two-pass f16 operand, same layout as pack_linear."""
    w = w.reshape(w.shape[0], -1)
    kp = pad_to(w.shape[1], k_align(dtype))
    if kp != w.shape[1]:
        w = torch.nn.functional.pad(w, (0, kp - w.shape[1]))
    return split_f16(w)


def pack_geglu_x2(w, b, dtype):
    inner = w.shape[0] // 2
    perm = geglu_perm(inner, w.device)
    return pack_linear_x2(w[perm], dtype), b[perm].float().contiguous()


def cast(w, dtype):
    """2-D K-major weight -> operand format of the compute mode (``dtype``: Precision, name or torch dtype)."""
    prec = resolve(dtype)
    return split_bf16(w) if prec.x3 else w.to(prec.storage).contiguous()


def pad_to(n, m):
    return (n + m - 1) // m * m


def pack_linear(w, dtype):
    """nn.Linear / 1x1 conv weight [N, K(,1,1)] -> [N, Kpad]."""
    w = w.reshape(w.shape[0], -1)
    kp = pad_to(w.shape[1], k_align(dtype))
    if kp != w.shape[1]:
        w = torch.nn.functional.pad(w, (0, kp - w.shape[1]))
    return cast(w, dtype)


def pack_conv2d(w, dtype, cin_pad=None):
    """nn.Conv2d weight [Cout, Cin, KH, KW] -> [Cout, KH*KW*Cin_pad] (tap-major, channel fastest)."""
    co, ci, kh, kw = w.shape
    cp = cin_pad or pad_to(ci, k_align(dtype))
    w = w.permute(0, 2, 3, 1)
    if cp != ci:
        w = torch.nn.functional.pad(w, (0, cp - ci))
    return cast(w.reshape(co, kh * kw * cp), dtype)


def pack_conv3d_t(w, dtype):
    """nn.Conv3d weight [Cout, Cin, 3, 1, 1] -> [Cout, 3*Cin]."""
    co, ci, kt, kh, kw = w.shape
    assert kh == 1 and kw == 1
    return cast(w.reshape(co, ci, kt).permute(0, 2, 1).reshape(co, kt * ci), dtype)


def geglu_perm(inner, device=None):
    """Row order that interleaves 32 value rows with their 32 gate rows (GEGLU.proj: [value | gate])."""
    assert inner % 32 == 0
    j = torch.arange(inner // 32, device=device)
    blk = torch.arange(32, device=device)
    val = (j[:, None] * 32 + blk[None, :])
    gate = val + inner
    return torch.cat([val, gate], dim=1).reshape(-1)


def pack_geglu(w, b, dtype):
    inner = w.shape[0] // 2
    perm = geglu_perm(inner, w.device)
    return pack_linear(w[perm], dtype), b[perm].float().contiguous()
