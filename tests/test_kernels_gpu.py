"""Kernel-level parity of the HIP path (through the C ABI) against plain PyTorch fp32 math on the same inputs.

Tolerances (relative L2, printed for every case): f32 mode 2e-5 (exact-f32 MFMA, only summation order differs);
bf16 / f16 modes compare against the fp32 result computed from the SAME rounded inputs, so what remains is the
output rounding + fp32 accumulation order: 6e-3 (bf16), 1e-3 (f16).
"""
import math

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16, torch.float16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 6e-3, torch.float16: 1e-3}


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def check(name, got, ref, dtype, scale=1.0):
    e = rel(got.float(), ref.float())
    m = (got.float() - ref.float()).abs().max().item()
    print(f"[{name}] dtype={dtype} rel_l2={e:.3e} max_abs={m:.3e} tol={TOL[dtype] * scale:.1e}")
    assert math.isfinite(e) and e <= TOL[dtype] * scale, f"{name}: rel_l2 {e:.3e} > {TOL[dtype] * scale:.1e}"


def rnd(shape, dev, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 11, 13, 16, 17])
def test_linear_bias_residual(dev, dtype, tile):
    from geo4d_amd import ops
    M, K, N = 300, 320, 200  # ragged M and N
    x, w = rnd((M, K), dev, dtype, 1), rnd((N, K), dev, dtype, 2, 0.05)
    b = rnd((N,), dev, torch.float32, 3)
    r = rnd((M, N), dev, dtype, 4)
    out = ops.linear(x, w, b, residual=r, tile_hint=tile)
    ref = x.float() @ w.float().t() + b + r.float()
    check(f"linear tile{tile}", out, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_is_not_transposed(dev, dtype):
    """A = I with an asymmetric W catches a swapped C/D mapping (cdna_hip_programming.md rule 16)."""
    from geo4d_amd import ops
    K = 128
    x = torch.eye(K, device=dev, dtype=dtype)
    w = (torch.arange(96 * K, device=dev, dtype=torch.float32).reshape(96, K) % 17 - 8).to(dtype)
    out = ops.linear(x, w)
    check("identity", out, w.float().t(), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_silu_and_strided_views(dev, dtype):
    from geo4d_amd import ops
    M, K, N = 257, 128, 64
    big = rnd((M, 3 * K), dev, dtype, 5)
    x = big[:, K:2 * K]  # column slice: row pitch 3K
    w = rnd((N, K), dev, dtype, 6, 0.1)
    outbuf = torch.zeros((M, 2 * N), device=dev, dtype=dtype)
    ops.linear(x, w, None, act=1, out=outbuf[:, N:])
    ref = TF.silu(x.float() @ w.float().t())
    check("linear silu strided", outbuf[:, N:], ref, dtype)
    assert outbuf[:, :N].abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_geglu(dev, dtype):
    from geo4d_amd import ops, pack
    M, K, inner = 200, 64 if dtype != torch.float32 else 32, 128
    x = rnd((M, K), dev, dtype, 7)
    w, b = rnd((2 * inner, K), dev, torch.float32, 8, 0.2), rnd((2 * inner,), dev, torch.float32, 9)
    wp, bp = pack.pack_geglu(w, b, dtype)
    out = ops.linear(x, wp, bp, act=2)
    h = x.float() @ w.to(dtype).float().t() + b
    ref = h[:, :inner] * TF.gelu(h[:, inner:])
    assert out.shape == (M, inner)
    check("geglu", out, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [11, 13, 16, 17])
def test_big_tiles_geglu_and_deep_conv(dev, dtype, tile):
    """The 8- / 10- / 5-wave configurations (tile hints 11, 13, 16, 17): GEGLU epilogue with 128-wide wave tiles, and a 3x3 conv
    whose K loop (36 slabs in 16-bit) is longer than any ring, ragged M and N, with split-K 2 as well."""
    from geo4d_amd import ops, pack
    M, K, inner = 700, 256, 320
    x = rnd((M, K), dev, dtype, 7)
    w, b = rnd((2 * inner, K), dev, torch.float32, 8, 0.1), rnd((2 * inner,), dev, torch.float32, 9)
    wp, bp = pack.pack_geglu(w, b, dtype)
    if tile in (16, 17):      # 160-column wave tiles cannot pair value / gate blocks: the C ABI must refuse, not mis-compute
        with pytest.raises(RuntimeError):
            ops.linear(x, wp, bp, act=2, tile_hint=tile)
    else:
        out = ops.linear(x, wp, bp, act=2, tile_hint=tile)
        h = x.float() @ w.to(dtype).float().t() + b
        check(f"geglu tile{tile}", out, h[:, :inner] * TF.gelu(h[:, inner:]), dtype)
    F, H, W, Ci, Co = 5, 12, 9, 256, 200
    x_nchw = rnd((F, Ci, H, W), dev, dtype, 10)
    wc = rnd((Co, Ci, 3, 3), dev, torch.float32, 11, 0.03)
    bc = rnd((Co,), dev, torch.float32, 12)
    xt = x_nchw.permute(0, 2, 3, 1).reshape(F * H * W, Ci).contiguous()
    r = rnd((F * H * W, Co), dev, dtype, 14)
    ref = TF.conv2d(x_nchw.float(), wc.to(dtype).float(), bc, padding=1).permute(0, 2, 3, 1).reshape(F * H * W, Co) + r.float()
    for split in (1, 2):
        o, _, _ = ops.conv2d(xt, pack.pack_conv2d(wc, dtype), bc, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, residual=r, tile_hint=tile, split_k=split)
        check(f"conv tile{tile} split{split}", o, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [dict(stride=1, ups=1), dict(stride=2, ups=1), dict(stride=1, ups=2)])
def test_conv3x3(dev, dtype, cfg):
    from geo4d_amd import ops, pack
    F, H, W, Ci, Co = 3, 9, 7, 64, 96
    x_nchw = rnd((F, Ci, H, W), dev, dtype, 10)
    w = rnd((Co, Ci, 3, 3), dev, torch.float32, 11, 0.05)
    b = rnd((Co,), dev, torch.float32, 12)
    emb = rnd((F, Co), dev, torch.float32, 13)
    x = x_nchw.permute(0, 2, 3, 1).reshape(F * H * W, Ci).contiguous()
    wp = pack.pack_conv2d(w, dtype)
    xin = x_nchw.float()
    if cfg["ups"] == 2:
        xin = TF.interpolate(xin, scale_factor=2, mode="nearest")
    ref = TF.conv2d(xin, w.to(dtype).float(), b, stride=cfg["stride"], padding=1) + emb[:, :, None, None]
    Ho, Wo = ref.shape[-2:]
    out, ho, wo = ops.conv2d(x, wp, b, F=F, Hin=H, Win=W, KH=3, KW=3, stride=cfg["stride"], pad=1, ups=cfg["ups"],
                             rowbias=emb, rowbias_div=Ho * Wo)
    assert (ho, wo) == (Ho, Wo)
    check(f"conv3x3 {cfg}", out.reshape(F, Ho, Wo, Co).permute(0, 3, 1, 2), ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_out_ncthw_small_n(dev, dtype):
    from geo4d_amd import ops, pack
    B, T, H, W, Ci, Co = 2, 3, 6, 5, 64, 16
    x_nchw = rnd((B * T, Ci, H, W), dev, dtype, 14)
    w = rnd((Co, Ci, 3, 3), dev, torch.float32, 15, 0.05)
    b = rnd((Co,), dev, torch.float32, 16)
    x = x_nchw.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous()
    out, _, _ = ops.conv2d(x, pack.pack_conv2d(w, dtype), b, F=B * T, Hin=H, Win=W, KH=3, KW=3, pad=1, T=T, out_nchw=True)
    ref = TF.conv2d(x_nchw.float(), w.to(dtype).float(), b, padding=1).reshape(B, T, Co, H, W).permute(0, 2, 1, 3, 4)
    assert out.shape == (B, Co, T, H, W) and out.dtype == torch.float32
    check("conv ncthw", out, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_temporal(dev, dtype):
    from geo4d_amd import ops, pack
    B, T, HW, Cc = 2, 5, 12, 64
    x5 = rnd((B, Cc, T, HW, 1), dev, dtype, 17)
    w = rnd((Cc, Cc, 3, 1, 1), dev, torch.float32, 18, 0.08)
    b = rnd((Cc,), dev, torch.float32, 19)
    x = x5.permute(0, 2, 3, 4, 1).reshape(B * T * HW, Cc).contiguous()
    out = ops.conv_temporal(x, pack.pack_conv3d_t(w, dtype), b, B=B, T=T, HW=HW, residual=x)
    ref = TF.conv3d(x5.float(), w.to(dtype).float(), b, padding=(1, 0, 0)) + x5.float()
    check("conv3d(3,1,1)", out.reshape(B, T, HW, 1, Cc).permute(0, 4, 1, 2, 3), ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_batched_gemm_rowbias_alpha(dev, dtype):
    from geo4d_amd import ops
    Z, M, N, K = 3, 70, 100, 64
    a, b = rnd((Z * M, K), dev, dtype, 20), rnd((Z * N, K), dev, dtype, 21)
    bias = rnd((M,), dev, torch.float32, 22)
    out = torch.empty((Z * M, N), device=dev, dtype=torch.float32)
    ops.batched_gemm(a, b, out, batch=Z, M=M, N=N, K=K, a_bs=M * K, b_bs=N * K, o_bs=M * N, bias=bias, bias_per_row=True,
                     alpha=0.25)
    ref = 0.25 * torch.einsum("zmk,znk->zmn", a.float().reshape(Z, M, K), b.float().reshape(Z, N, K)) + bias[None, :, None]
    check("batched gemm", out.reshape(Z, M, N), ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(4, 6 * 7, 320, 1), (6, 5 * 4, 64, 3), (2, 70 * 33, 128, 1), (4, 1, 960, 2)])
def test_groupnorm(dev, dtype, case):
    from geo4d_amd import ops
    F, HW, Cc, fps = case
    x = rnd((F * HW, Cc), dev, dtype, 23) * 2 + 0.7
    g, b = rnd((Cc,), dev, torch.float32, 24), rnd((Cc,), dev, torch.float32, 25)
    out = ops.groupnorm(x, g, b, F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=True)
    x5 = x.float().reshape(F // fps, fps, HW, Cc).permute(0, 3, 1, 2)  # [stat, C, fps, HW]
    ref = TF.silu(TF.group_norm(x5, 32, g, b, 1e-5)).permute(0, 2, 3, 1).reshape(F * HW, Cc)
    check(f"groupnorm {case}", out, ref, dtype, scale=2.0)
    out2 = ops.groupnorm(x, g, b, F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=True)
    assert torch.equal(out, out2), "groupnorm must be run-to-run deterministic"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Cc", [320, 640, 1280, 512])
def test_layernorm(dev, dtype, Cc):
    from geo4d_amd import ops
    x = rnd((77, Cc), dev, dtype, 26) * 3 + 1
    g, b = rnd((Cc,), dev, torch.float32, 27), rnd((Cc,), dev, torch.float32, 28)
    out = ops.layernorm(x, g, b, 1e-5)
    check(f"layernorm {Cc}", out, TF.layer_norm(x.float(), (Cc,), g, b, 1e-5), dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_rows(dev, dtype):
    from geo4d_amd import ops
    x = rnd((50, 333), dev, torch.float32, 29) * 4
    out = ops.softmax_rows(x, 0.3, dtype)
    check("softmax", out, torch.softmax(x * 0.3, -1), dtype)


def _sdpa(q, k, v, scale):
    return torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v


def _vt(v, b, n, pad=None):
    """[b*n, C] row-major values -> V^T [C, b*n] (+ zero padded columns), the layout geo4d_attention consumes."""
    t = v.t().contiguous()
    if pad:
        t = torch.nn.functional.pad(t, (0, pad - t.shape[1]))
    return t.contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N", [40, 160, 200, 640, 2560, 2304, 9216])   # 2304 / 9216 = tokens per frame at 576x256 / 576x1024 (BASELINE configs[2], [4])
def test_attention_self(dev, dtype, N):
    from geo4d_amd import ops
    B, H = (3, 5) if N < 600 else ((16, 10) if N == 640 else (1, 2))
    C_ = H * 64
    qkv = rnd((B * N, 3 * C_), dev, dtype, 30)
    vt = _vt(qkv[:, 2 * C_:], B, N)
    out = ops.attention(qkv[:, :C_], [(qkv[:, C_:2 * C_], vt, N, 1, N)], B=B, H=H, Nq=N, scale=0.125)
    f = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = _sdpa(f[0], f[1], f[2], 0.125).permute(0, 2, 1, 3).reshape(B * N, C_)
    check(f"attn self N={N}", out, ref, dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("variant", [1, 2, 3, 4])
@pytest.mark.parametrize("N", [40, 200, 640, 2560])
def test_attention_variants(dev, dtype, variant, N):
    """The A/B builds of the flash kernel (geo4d_attention_t.variant: 4-waves-per-SIMD build, two query blocks per wave with
    256-row workgroups — ragged N = 200 leaves one block half empty) compute the same attention."""
    from geo4d_amd import ops
    if variant == 4 and dtype == torch.float32:
        pytest.skip("variant 4 (flash_attn2_kernel, skewed query blocks) is built for the 16-bit types and pre-split bf16x3")
    B, H = 2, 3
    C_ = H * 64
    qkv = rnd((B * N, 3 * C_), dev, dtype, 130 + N)
    vt = torch.stack([_vt(qkv[b * N:(b + 1) * N, 2 * C_:], 1, N) for b in range(B)])      # [B, C, N]
    out = ops.attention(qkv[:, :C_], [(qkv[:, C_:2 * C_], vt.reshape(-1, N), N, 1, C_ * N)], B=B, H=H, Nq=N, scale=0.125, variant=variant)
    f = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = _sdpa(f[0], f[1], f[2], 0.125).permute(0, 2, 1, 3).reshape(B * N, C_)
    check(f"attn variant {variant} N={N}", out, ref, dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_v_projected_transposed(dev, dtype):
    """linear_t writes V^T directly (operand-swapped GEMM): same numbers as projecting then transposing."""
    from geo4d_amd import ops
    M, K, N = 200, 128, 192
    x, w = rnd((M, K), dev, dtype, 60), rnd((N, K), dev, dtype, 61, 0.1)
    vt = ops.linear_t(w, x)
    check("linear_t", vt, (x.float() @ w.float().t()).t(), dtype)
    vt2 = ops.linear_t(w, x[:77], pad_cols=80)
    assert vt2.shape == (N, 80) and vt2[:, 77:].abs().max().item() == 0
    check("linear_t padded", vt2[:, :77], (x[:77].float() @ w.float().t()).t(), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_cross_two_sets(dev, dtype):
    """77 text keys shared by all T frames of a sample + 16 per-frame image keys, outputs summed (attention.py:128-142)."""
    from geo4d_amd import ops
    Bs, T, N, H = 2, 3, 50, 2
    C_ = H * 64
    q = rnd((Bs * T * N, C_), dev, dtype, 31)
    kt, vt = rnd((Bs * 77, C_), dev, dtype, 32), rnd((Bs * 77, C_), dev, dtype, 33)
    ki, vi = rnd((Bs * T * 16, C_), dev, dtype, 34), rnd((Bs * T * 16, C_), dev, dtype, 35)
    vt_text = torch.stack([_vt(vt[b * 77:(b + 1) * 77], 1, 77, pad=80) for b in range(Bs)])      # [Bs, C, 80]
    vt_img = _vt(vi, Bs * T, 16)                                                                  # [C, Bs*T*16]
    out = ops.attention(q, [(kt, vt_text.reshape(-1, 80), 77, T, C_ * 80), (ki, vt_img, 16, 1, 16)], B=Bs * T, H=H, Nq=N, scale=0.125)

    def heads(x, b, n):
        return x.float().reshape(b, n, H, 64).permute(0, 2, 1, 3)
    qh = heads(q, Bs * T, N)
    kth = heads(kt, Bs, 77).repeat_interleave(T, 0)
    vth = heads(vt, Bs, 77).repeat_interleave(T, 0)
    ref = _sdpa(qh, kth, vth, 0.125) + _sdpa(qh, heads(ki, Bs * T, 16), heads(vi, Bs * T, 16), 0.125)
    check("attn cross", out, ref.permute(0, 2, 1, 3).reshape(Bs * T * N, C_), dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_online_softmax_rescale(dev, dtype):
    """A key far above the rest in a LATE tile forces the running-max rescale branch (rule 26)."""
    from geo4d_amd import ops
    N, H = 256, 1
    q, k, v = rnd((N, 64), dev, dtype, 36), rnd((N, 64), dev, dtype, 37), rnd((N, 64), dev, dtype, 38)
    k[200] = (q[7].float() * 4).to(dtype)
    out = ops.attention(q, [(k, _vt(v, 1, N), N, 1, N)], B=1, H=H, Nq=N, scale=0.125)
    check("attn spike", out, _sdpa(q.float(), k.float(), v.float(), 0.125), dtype, scale=2.0)
    if dtype != torch.float32:         # the skewed-block kernel: the spiked row sits in the wave's FIRST query block, a second spike in its second block
        k[130] = (q[40].float() * 4).to(dtype)
        out = ops.attention(q, [(k, _vt(v, 1, N), N, 1, N)], B=1, H=H, Nq=N, scale=0.125, variant=4)
        check("attn spike (variant 4)", out, _sdpa(q.float(), k.float(), v.float(), 0.125), dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T", [16, 5])
def test_temporal_attention(dev, dtype, T):
    from geo4d_amd import ops
    B, HW, H = 2, 7, 3
    C_ = H * 64
    qkv = rnd((B * T * HW, 3 * C_), dev, dtype, 39)
    out = ops.temporal_attention(qkv[:, :C_], qkv[:, C_:2 * C_], qkv[:, 2 * C_:], B=B, T=T, HW=HW, H=H, scale=0.125)
    f = qkv.float().reshape(B, T, HW, 3, H, 64).permute(3, 0, 2, 4, 1, 5)  # [3, B, HW, H, T, 64]
    ref = _sdpa(f[0], f[1], f[2], 0.125).permute(0, 3, 1, 2, 4).reshape(B * T * HW, C_)
    check(f"temporal attn T={T}", out, ref, dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_and_concat(dev, dtype):
    from geo4d_amd import ops
    B, T, H, W = 2, 3, 4, 5
    a, b = rnd((B, 16, T, H, W), dev, torch.float32, 40), rnd((B, 4, T, H, W), dev, torch.float32, 41)
    tok = ops.tokens_from_ncthw(a, b, 32, dtype)
    ref = torch.cat([a, b], 1).permute(0, 2, 3, 4, 1).reshape(-1, 20).to(dtype)
    assert torch.equal(tok[:, :20], ref) and tok[:, 20:].abs().max().item() == 0
    x, y = rnd((33, 64), dev, dtype, 42), rnd((33, 128), dev, dtype, 43)
    assert torch.equal(ops.concat_channels(x, y), torch.cat([x, y], 1))


def test_embedding_and_small_linear(dev):
    from geo4d_amd import ops
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(dev)
    t = torch.tensor([999, 19], device=dev, dtype=torch.int64)
    e = ops.timestep_embedding(t, freqs)
    args = t[:, None].float() * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], -1)
    assert (e - ref).abs().max().item() < 2e-4  # t*freq up to 999 rad: device sin/cos vs torch differ by ulps of the argument
    w, b = rnd((300, 320), dev, torch.float32, 44, 0.05), rnd((300,), dev, torch.float32, 45)
    add = rnd((2, 300), dev, torch.float32, 46)
    out = ops.linear_small(e, w, b, add=add, act_in=True, act_out=True)
    check("linear_small", out, TF.silu(TF.silu(e) @ w.t() + b) + add, torch.float32)


@pytest.mark.parametrize("three_way", [False, True])
@pytest.mark.parametrize("rescale", [0.0, 0.7])
def test_cfg_combine(dev, three_way, rescale):
    """geo4d_cfg_combine vs the reference formulas (ddim.py:216-229, ddim_multiplecond.py:229-236, utils_diffusion.py:147-158)."""
    from geo4d_amd import ops
    g = torch.Generator().manual_seed(11)
    shape = (2, 16, 5, 7, 9)                     # n = 5040 per sample: not a multiple of the chunking
    e_c, e_u, e_i = (torch.randn(shape, generator=g).to(dev) * (1.0 + k) + 0.1 * k for k in range(3))
    got = ops.cfg_combine(e_c, e_u, e_i if three_way else None, scale=7.5, cfg_img=2.0 if three_way else None, guidance_rescale=rescale)
    ref = e_u + 2.0 * (e_i - e_u) + 7.5 * (e_c - e_i) if three_way else e_u + 7.5 * (e_c - e_u)
    if rescale > 0:
        dims = list(range(1, ref.ndim))
        r = ref * (e_c.std(dim=dims, keepdim=True) / ref.std(dim=dims, keepdim=True))
        ref = rescale * r + (1 - rescale) * ref
    err = ((got - ref).norm() / ref.norm()).item()
    print(f"[cfg_combine three_way={three_way} rescale={rescale}] rel_l2={err:.2e}")
    assert got.shape == ref.shape and err < 2e-6


def test_ddim_step(dev):
    from geo4d_amd import ops
    x, v, nz = rnd((1000,), dev, torch.float32, 47), rnd((1000,), dev, torch.float32, 48), rnd((1000,), dev, torch.float32, 49)
    coef = torch.tensor([[0.0] * 6, [0.6, 0.8, 0.9, 0.7, 0.5, 0.1]], device=dev)
    idx = torch.tensor([1], device=dev, dtype=torch.int32)
    x0 = torch.empty_like(x)
    xr = x.clone()
    ops.ddim_step(x, v, coef, idx, noise=nz, pred_x0=x0)
    e_t = 0.6 * v + 0.8 * xr
    p0 = (0.6 * xr - 0.8 * v) * 0.9
    check("ddim x0", x0, p0, torch.float32)
    check("ddim x_prev", x, 0.7 * p0 + 0.5 * e_t + 0.1 * nz, torch.float32)
    ops.advance_index(idx, -1)
    assert idx.item() == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("split", [2, 4])
def test_split_k_conv_matches(dev, dtype, split):
    """Deep-K, small-M problems (5x8 / 10x16 U-Net levels) run split-K: fp32 slabs + fixed-order reduce with the epilogue."""
    from geo4d_amd import ops, pack
    F, H, W, Ci, Co = 2, 5, 8, 256, 96
    x_nchw = rnd((F, Ci, H, W), dev, dtype, 50)
    w = rnd((Co, Ci, 3, 3), dev, torch.float32, 51, 0.03)
    b = rnd((Co,), dev, torch.float32, 52)
    emb = rnd((F, Co), dev, torch.float32, 53)
    x = x_nchw.permute(0, 2, 3, 1).reshape(F * H * W, Ci).contiguous()
    res = rnd((F * H * W, Co), dev, dtype, 54)
    wp = pack.pack_conv2d(w, dtype)
    out, _, _ = ops.conv2d(x, wp, b, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, rowbias=emb, rowbias_div=H * W, residual=res, split_k=split)
    one, _, _ = ops.conv2d(x, wp, b, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, rowbias=emb, rowbias_div=H * W, residual=res, split_k=1)
    ref = TF.conv2d(x_nchw.float(), w.to(dtype).float(), b, padding=1) + emb[:, :, None, None]
    ref = ref.permute(0, 2, 3, 1).reshape(F * H * W, Co) + res.float()
    check(f"split-k {split}", out, ref, dtype)
    check(f"split-k {split} vs unsplit", out, one.float(), dtype)
    again, _, _ = ops.conv2d(x, wp, b, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, rowbias=emb, rowbias_div=H * W, residual=res, split_k=split)
    assert torch.equal(out, again), "split-K reduce must be deterministic"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_lds_dma_pipelines_are_race_free(dev, dtype):
    """The GEMM and attention kernels stage tiles by LDS-DMA, which hipcc does not track: a missing `s_waitcnt vmcnt`
    before a barrier shows up as run-to-run differences / NaNs on a busy chip. Full-chip shapes, repeated, bit-compared."""
    from geo4d_amd import ops
    B, H, N = 16, 10, 640
    C_ = H * 64
    qk = rnd((B * N, 2 * C_), dev, dtype, 70)
    vt = rnd((B * C_, N), dev, dtype, 71)
    f = qk.float().reshape(B, N, 2, H, 64).permute(2, 0, 3, 1, 4)
    v = vt.float().reshape(B, H, 64, N).permute(0, 1, 3, 2)
    ref = _sdpa(f[0], f[1], v, 0.125).permute(0, 2, 1, 3).reshape(B * N, C_)
    first = None
    for _ in range(6):
        out = ops.attention(qk[:, :C_], [(qk[:, C_:], vt, N, 1, C_ * N)], B=B, H=H, Nq=N, scale=0.125)
        assert torch.isfinite(out.float()).all()
        first = out if first is None else first
        assert torch.equal(out, first), "attention output changed between identical launches"
    check("attn full-chip", first, ref, dtype, scale=2.0)
    M, K, Nn = 10240, 5760, 640
    x, w = rnd((M, 640), dev, dtype, 72), rnd((Nn, K), dev, dtype, 73, 0.02)
    outs = [ops.conv2d(x, w, None, F=16, Hin=20, Win=32, KH=3, KW=3, pad=1, tile_hint=t, split_k=1)[0] for t in (1, 1, 1, 2, 3, 4, 11, 11, 13, 13, 16, 16, 17, 17)]   # same split => same fp32 association
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "conv_gemm output depends on launch / tile shape"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(4, 64, 320, 1, 0), (4, 160, 640, 4, 11), (2, 2560, 320, 2, 16), (3, 96, 128, 1, 2)])
def test_groupnorm_statistics_from_the_gemm_epilogue(dev, dtype, case):
    """conv_gemm(gn_stats=True) leaves per-32-row-block column sums on its output; the GroupNorm that consumes that tensor skips
    its statistics pass. Same result as the three-pass GroupNorm (to fp32 re-association) and as F.group_norm; every tile shape
    writes every (block, column) exactly once; a residual and a bias are part of what is summed."""
    from geo4d_amd import ops
    monkey = ops.GN_FUSED_STATS
    ops.GN_FUSED_STATS = 2                 # mode 2: also on first-generation tiles (default 1 = second / third generation producers only, ops.py)
    try:
        _fused_stats_case(dev, dtype, case)
    finally:
        ops.GN_FUSED_STATS = monkey


def _fused_stats_case(dev, dtype, case):
    from geo4d_amd import ops
    F, HW, Cc, fps, tile = case
    M, K = F * HW, 128
    x, w = rnd((M, K), dev, dtype, 300), rnd((Cc, K), dev, dtype, 301, 0.1)
    b, r = rnd((Cc,), dev, torch.float32, 302), rnd((M, Cc), dev, dtype, 303)
    g, be = rnd((Cc,), dev, torch.float32, 304), rnd((Cc,), dev, torch.float32, 305)
    h = ops.linear(x, w, b, residual=r, tile_hint=tile, gn_stats=True)
    rows = h._gn_colsum_rows
    assert rows == 32 and hasattr(h, "_gn_colsum") and h._gn_colsum.shape == (M // rows, Cc, 2)      # (first-generation tiles: 32-row blocks)
    cs_ref = torch.stack([h.float().reshape(M // 32, 32, Cc).sum(1), (h.float() ** 2).reshape(M // 32, 32, Cc).sum(1)], -1)
    assert rel(h._gn_colsum, cs_ref) < (2e-3 if dtype == torch.bfloat16 else 3e-4)      # sums of the un-rounded fp32 values
    fused = ops.groupnorm(h, g, be, F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=True)
    plain = ops.groupnorm(h.clone(), g, be, F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=True)     # the clone carries no sums
    x5 = h.float().reshape(F // fps, fps, HW, Cc).permute(0, 3, 1, 2)
    ref = TF.silu(TF.group_norm(x5, 32, g, be, 1e-5)).permute(0, 2, 3, 1).reshape(M, Cc)
    check(f"groupnorm fused stats {case}", fused, ref, dtype, scale=2.0)
    check(f"groupnorm fused vs three-pass {case}", fused, plain.float(), dtype, scale=1.0)
    fused2 = ops.groupnorm(ops.linear(x, w, b, residual=r, tile_hint=tile, gn_stats=True), g, be, F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=True)
    assert torch.equal(fused, fused2), "fused-statistics GroupNorm must be run-to-run deterministic"
