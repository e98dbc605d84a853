"""bf16x3 compute mode (geo4d_amd/precision.py): f32 storage, every product as three bf16 MFMAs on a hi/lo split of both
operands. Kernel-level parity against plain PyTorch fp32 math computed from the UNROUNDED inputs: what remains is the
2^-17-class truncation of the split (+ the dropped lo.lo term + fp32 accumulation order), so the tolerance is 5e-5 relative
L2 — two orders of magnitude tighter than f16, one looser than the exact-f32 mode.
"""
import math

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu
TOL = 5e-5
X3 = "bf16x3"


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def check(name, got, ref, scale=1.0):
    e = rel(got.float(), ref.float())
    print(f"[x3 {name}] rel_l2={e:.3e} tol={TOL * scale:.1e}")
    assert math.isfinite(e) and e <= TOL * scale, f"{name}: rel_l2 {e:.3e} > {TOL * scale:.1e}"


def rnd(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def test_split_layout_roundtrip(dev):
    """pack.split_bf16: [N, K] -> bf16 [N, 2K] = per 8 K-elements [8 hi | 8 lo]; hi + lo reproduces w to ~2^-17."""
    from geo4d_amd import pack
    w = rnd((48, 64), dev, 1)
    s = pack.split_bf16(w)
    assert s.dtype == torch.bfloat16 and s.shape == (48, 128)
    parts = s.float().reshape(48, 8, 2, 8)
    back = (parts[:, :, 0] + parts[:, :, 1]).reshape(48, 64)
    assert torch.equal(parts[:, :, 0].reshape(48, 64), w.to(torch.bfloat16).float())
    assert rel(back, w) < 1e-5


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 11, 13, 16, 17])
def test_linear_presplit_weights(dev, tile):
    from geo4d_amd import ops, pack
    M, K, N = 300, 320, 200  # ragged M and N
    x, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.05)
    b, r = rnd((N,), dev, 3), rnd((M, N), dev, 4)
    out = ops.linear(x, pack.pack_linear(w, X3), b, residual=r, tile_hint=tile)
    assert out.dtype == torch.float32
    check(f"linear tile{tile}", out, x @ w.t() + b + r)


def test_linear_is_not_transposed_and_operands_not_swapped(dev):
    from geo4d_amd import ops, pack
    K = 128
    x = torch.eye(K, device=dev) * 1.0009765625          # needs the lo part of x: 1 + 2^-10 is not a bf16
    w = (torch.arange(96 * K, device=dev, dtype=torch.float32).reshape(96, K) % 17 - 8) * 1.001
    check("identity", ops.linear(x, pack.pack_linear(w, X3)), (x @ w.t()))


def test_linear_t_presplit_a_operand(dev):
    """operand-swapped projection (V^T): the pre-split weight is the A operand, the f32 activation the W operand."""
    from geo4d_amd import ops, pack
    M, K, N = 200, 128, 192
    x, w = rnd((M, K), dev, 60), rnd((N, K), dev, 61, 0.1)
    wp = pack.pack_linear(w, X3)
    check("linear_t", ops.linear_t(wp, x), (x @ w.t()).t())
    vt, rp = ops.linear_t_batched(wp, x[:198], 2, 99)
    assert vt.shape == (2, N, 100) and rp == 100 and vt[:, :, 99:].abs().max().item() == 0
    ref = torch.stack([(x[b * 99:(b + 1) * 99] @ w.t()).t() for b in range(2)])
    check("linear_t_batched", vt[:, :, :99], ref)


def test_batched_gemm_two_activations(dev):
    from geo4d_amd import ops
    Z, M, N, K = 3, 70, 100, 64
    a, b = rnd((Z * M, K), dev, 20), rnd((Z * N, K), dev, 21)
    bias = rnd((M,), dev, 22)
    out = torch.empty((Z * M, N), device=dev, dtype=torch.float32)
    ops.batched_gemm(a, b, out, batch=Z, M=M, N=N, K=K, a_bs=M * K, b_bs=N * K, o_bs=M * N, bias=bias, bias_per_row=True,
                     alpha=0.25, x3=True)
    ref = 0.25 * torch.einsum("zmk,znk->zmn", a.reshape(Z, M, K), b.reshape(Z, N, K)) + bias[None, :, None]
    check("batched gemm", out.reshape(Z, M, N), ref)
    exact = torch.empty_like(out)
    ops.batched_gemm(a, b, exact, batch=Z, M=M, N=N, K=K, a_bs=M * K, b_bs=N * K, o_bs=M * N, bias=bias, bias_per_row=True, alpha=0.25)
    assert not torch.equal(out, exact), "x3=True must select the split-bf16 kernel, not the exact-f32 one"


def test_geglu(dev):
    from geo4d_amd import ops, pack
    M, K, inner = 200, 64, 128
    x = rnd((M, K), dev, 7)
    w, b = rnd((2 * inner, K), dev, 8, 0.2), rnd((2 * inner,), dev, 9)
    wp, bp = pack.pack_geglu(w, b, X3)
    h = x @ w.t() + b
    check("geglu", ops.linear(x, wp, bp, act=2), h[:, :inner] * TF.gelu(h[:, inner:]), scale=2.0)


@pytest.mark.parametrize("cfg", [dict(stride=1, ups=1), dict(stride=2, ups=1), dict(stride=1, ups=2)])
def test_conv3x3(dev, cfg):
    from geo4d_amd import ops, pack
    F, H, W, Ci, Co = 3, 9, 7, 64, 96
    x_nchw, w = rnd((F, Ci, H, W), dev, 10), rnd((Co, Ci, 3, 3), dev, 11, 0.05)
    b, emb = rnd((Co,), dev, 12), rnd((F, Co), dev, 13)
    x = x_nchw.permute(0, 2, 3, 1).reshape(F * H * W, Ci).contiguous()
    xin = TF.interpolate(x_nchw, scale_factor=2, mode="nearest") if cfg["ups"] == 2 else x_nchw
    ref = TF.conv2d(xin, w, b, stride=cfg["stride"], padding=1) + emb[:, :, None, None]
    Ho, Wo = ref.shape[-2:]
    out, ho, wo = ops.conv2d(x, pack.pack_conv2d(w, X3), b, F=F, Hin=H, Win=W, KH=3, KW=3, stride=cfg["stride"], pad=1,
                             ups=cfg["ups"], rowbias=emb, rowbias_div=Ho * Wo)
    assert (ho, wo) == (Ho, Wo)
    check(f"conv3x3 {cfg}", out.reshape(F, Ho, Wo, Co).permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("split", [1, 2, 4])
def test_deep_conv_split_k_and_heads(dev, split):
    from geo4d_amd import ops, pack
    F, H, W, Ci, Co = 2, 5, 8, 256, 96
    x_nchw, w, b = rnd((F, Ci, H, W), dev, 50), rnd((Co, Ci, 3, 3), dev, 51, 0.03), rnd((Co,), dev, 52)
    x = x_nchw.permute(0, 2, 3, 1).reshape(F * H * W, Ci).contiguous()
    res = rnd((F * H * W, Co), dev, 54)
    out, _, _ = ops.conv2d(x, pack.pack_conv2d(w, X3), b, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, residual=res, split_k=split)
    ref = TF.conv2d(x_nchw, w, b, padding=1).permute(0, 2, 3, 1).reshape(F * H * W, Co) + res
    check(f"conv split{split}", out, ref)
    if split == 1:      # NCTHW head with N = 3 (direct epilogue path)
        w3, b3 = rnd((3, Ci, 3, 3), dev, 55, 0.03), rnd((3,), dev, 56)
        o, _, _ = ops.conv2d(x, pack.pack_conv2d(w3, X3), b3, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, T=1, out_nchw=True)
        check("conv ncthw N=3", o[:, :, 0], TF.conv2d(x_nchw, w3, b3, padding=1))


def test_conv_temporal(dev):
    from geo4d_amd import ops, pack
    B, T, HW, Cc = 2, 5, 12, 64
    x5, w, b = rnd((B, Cc, T, HW, 1), dev, 17), rnd((Cc, Cc, 3, 1, 1), dev, 18, 0.08), rnd((Cc,), dev, 19)
    x = x5.permute(0, 2, 3, 4, 1).reshape(B * T * HW, Cc).contiguous()
    out = ops.conv_temporal(x, pack.pack_conv3d_t(w, X3), b, B=B, T=T, HW=HW, residual=x)
    ref = TF.conv3d(x5, w, b, padding=(1, 0, 0)) + x5
    check("conv3d(3,1,1)", out.reshape(B, T, HW, 1, Cc).permute(0, 4, 1, 2, 3), ref)


def _sdpa(q, k, v, scale):
    return torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v


@pytest.mark.parametrize("variant", [1, 3])
@pytest.mark.parametrize("N", [40, 200, 640, 2304, 9216])   # 2304 / 9216: BASELINE configs[2] / [4] tokens per frame
def test_attention_self(dev, N, variant):
    from geo4d_amd import ops
    B, H = (3, 5) if N < 600 else (2, 2)
    C_ = H * 64
    qkv = rnd((B * N, 3 * C_), dev, 30)
    Np = (N + 3) // 4 * 4
    vt = torch.zeros((B, C_, Np), device=dev)
    vt[:, :, :N] = qkv[:, 2 * C_:].reshape(B, N, C_).permute(0, 2, 1)
    out = ops.attention(qkv[:, :C_], [(qkv[:, C_:2 * C_], vt.reshape(-1, Np), N, 1, C_ * Np)], B=B, H=H, Nq=N, scale=0.125, x3=True, variant=variant)
    f = qkv.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = _sdpa(f[0], f[1], f[2], 0.125).permute(0, 2, 1, 3).reshape(B * N, C_)
    check(f"attn self N={N}", out, ref, scale=2.0)


def test_attention_cross_two_sets_and_rescale(dev):
    from geo4d_amd import ops
    Bs, T, N, H = 2, 3, 50, 2
    C_ = H * 64
    q = rnd((Bs * T * N, C_), dev, 31)
    kt, vt = rnd((Bs * 77, C_), dev, 32), rnd((Bs * 77, C_), dev, 33)
    ki, vi = rnd((Bs * T * 16, C_), dev, 34), rnd((Bs * T * 16, C_), dev, 35)
    vt_text = torch.zeros((Bs, C_, 80), device=dev)
    vt_text[:, :, :77] = vt.reshape(Bs, 77, C_).permute(0, 2, 1)
    vt_img = vi.t().contiguous()
    out = ops.attention(q, [(kt, vt_text.reshape(-1, 80), 77, T, C_ * 80), (ki, vt_img, 16, 1, 16)], B=Bs * T, H=H, Nq=N,
                        scale=0.125, x3=True)
    heads = lambda x, b, n: x.reshape(b, n, H, 64).permute(0, 2, 1, 3)
    qh = heads(q, Bs * T, N)
    ref = _sdpa(qh, heads(kt, Bs, 77).repeat_interleave(T, 0), heads(vt, Bs, 77).repeat_interleave(T, 0), 0.125) + \
        _sdpa(qh, heads(ki, Bs * T, 16), heads(vi, Bs * T, 16), 0.125)
    check("attn cross", out, ref.permute(0, 2, 1, 3).reshape(Bs * T * N, C_), scale=2.0)
    # a key far above the rest in a LATE tile forces the running-max rescale branch
    Nn = 256
    q1, k1, v1 = rnd((Nn, 64), dev, 36), rnd((Nn, 64), dev, 37), rnd((Nn, 64), dev, 38)
    k1[200] = q1[7] * 4
    o1 = ops.attention(q1, [(k1, v1.t().contiguous(), Nn, 1, Nn)], B=1, H=1, Nq=Nn, scale=0.125, x3=True)
    check("attn spike", o1, _sdpa(q1, k1, v1, 0.125), scale=2.0)


def test_x3_kernels_are_deterministic_on_a_full_chip(dev):
    """Same race screen as tests/test_kernels_gpu.py::test_lds_dma_pipelines_are_race_free, for the x3 instantiations."""
    from geo4d_amd import ops, pack
    B, H, N = 8, 10, 640
    C_ = H * 64
    qk, vt = rnd((B * N, 2 * C_), dev, 70), rnd((B * C_, N), dev, 71)
    first = None
    for _ in range(4):
        out = ops.attention(qk[:, :C_], [(qk[:, C_:], vt, N, 1, C_ * N)], B=B, H=H, Nq=N, scale=0.125, x3=True)
        first = out if first is None else first
        assert torch.isfinite(out).all() and torch.equal(out, first)
    x, w = rnd((10240, 640), dev, 72), pack.pack_conv2d(rnd((640, 640, 3, 3), dev, 73, 0.02), X3)
    outs = [ops.conv2d(x, w, None, F=16, Hin=20, Win=32, KH=3, KW=3, pad=1, tile_hint=t, split_k=1)[0] for t in (1, 1, 2, 3, 4, 11, 13, 13, 16, 17)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "conv_gemm (bf16x3) output depends on launch / tile shape"


@pytest.mark.parametrize("tile", [22, 23, 25, 27, 28, 71, 72, 73, 74])
@pytest.mark.parametrize("F,fps", [(2, 1), (2, 2), (1, 1)])       # F = 1: M = 1920 leaves the last row tile of the 256- / 192- / 160-row tiles ragged (waves beyond M write nothing)
def test_groupnorm_statistics_from_the_fast_epilogue(dev, tile, F, fps):
    """Round 4: the second / third generation GEMMs' fast epilogue emits the consumer GroupNorm's column sums per WAVE-TILE row range
    (geo4d_conv_gemm_colsum_rows: 32..128 rows per entry), bias / row-bias table / residual included; the GroupNorm fed by them equals
    the three-pass GroupNorm and F.group_norm. (Mode 2 = also on first-generation tiles; the default since round 4 is 1: second / third generation producers only.)"""
    import torch.nn.functional as TF
    from geo4d_amd import ops, pack
    HW, Cc, K = 1920, 320, 256
    M = F * HW
    g = torch.Generator().manual_seed(500 + tile)
    x, w = torch.randn((M, K), generator=g).to(dev), (torch.randn((Cc, K), generator=g) * 0.1).to(dev)
    b, r = torch.randn((Cc,), generator=g).to(dev), torch.randn((M, Cc), generator=g).to(dev)
    emb = torch.randn((F, Cc), generator=g).to(dev)
    gam, bet = torch.randn((Cc,), generator=g).to(dev), torch.randn((Cc,), generator=g).to(dev)
    wp = pack.pack_linear(w, "bf16x3")
    old = ops.GN_FUSED_STATS
    ops.GN_FUSED_STATS = 2
    try:
        for kw in (dict(residual=r), dict(rowbias=emb, rowbias_div=HW), dict()):
            h = torch.empty((M, Cc), device=dev)
            ops.conv_gemm(x, wp, h, M=M, N=Cc, K=K, Cin=K, lda=K, ldw=wp.stride(0), ldo=Cc, bias=b, ldr=Cc if "residual" in kw else 0,
                          tile_hint=tile, split_k=1, gn_stats=True, **kw)
            rows = getattr(h, "_gn_colsum_rows", 0)
            assert rows in (32, 64, 80, 96, 128) and h._gn_colsum.shape == (M // rows, Cc, 2), (tile, rows)
            hf = h.double()
            cs_ref = torch.stack([hf.reshape(M // rows, rows, Cc).sum(1), (hf ** 2).reshape(M // rows, rows, Cc).sum(1)], -1).float()
            assert ((h._gn_colsum - cs_ref).norm() / cs_ref.norm()).item() < 2e-6, (tile, list(kw))
            fused = ops.groupnorm(h, gam, bet, F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=True)
            plain = ops.groupnorm(h.clone(), gam, bet, F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=True)      # the clone carries no sums
            x5 = h.reshape(F // fps, fps, HW, Cc).permute(0, 3, 1, 2)
            ref = TF.silu(TF.group_norm(x5, 32, gam, bet, 1e-5)).permute(0, 2, 3, 1).reshape(M, Cc)
            assert ((fused - ref).norm() / ref.norm()).item() < 2e-5 and ((fused - plain).norm() / plain.norm()).item() < 1e-5, (tile, list(kw))
    finally:
        ops.GN_FUSED_STATS = old
