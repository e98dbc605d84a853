"""Second-generation conv_gemm kernel (tile hints 22 / 23 / 25 / 27 / 28 after the round-4 pruning: 16x16x32 MFMA, register epilogue,
persistent workgroups; geo4d_amd/csrc/gemm_kernel_v2.h) against plain PyTorch fp32 math, every element type it serves (bf16, bf16x3;
f16 stays on the first generation), through the C ABI.
Each case runs twice: with the production grid and with `debug_ablate = 2` (3 workgroups: the persistent tile loop, the next-tile
prefetch under the epilogue and the gather-table reuse are exercised on small shapes), and the two must agree bit for bit."""
import math

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

V2_TILES = [22, 23, 25, 27, 28]
GEGLU_TILES = {22, 25, 27}
MODES = ["bf16", "bf16x3"]
TOL = {"bf16": 6e-3, "f16": 1e-3, "bf16x3": 3e-5}


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def rnd(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def act_dtype(mode):
    return {"bf16": torch.bfloat16, "f16": torch.float16, "bf16x3": torch.float32}[mode]


def pack_mode(mode):
    return "bf16x3" if mode == "bf16x3" else act_dtype(mode)


def rounded(t, mode):
    """The value the kernel multiplies: operands are rounded to the mode's storage type (bf16x3 keeps ~16 mantissa bits: exact here)."""
    return t.to(act_dtype(mode)).float()


def both_grids(fn):
    """fn() with the production grid and with 3 persistent workgroups; returns the production result after checking equality."""
    from geo4d_amd import ops
    a = fn()
    ops.DEBUG_ABLATE = 2
    try:
        b = fn()
    finally:
        ops.DEBUG_ABLATE = 0
    assert torch.equal(a, b), f"persistent-loop result differs from one-tile-per-workgroup result: {rel(b, a):.3e}"
    return a


def check(name, got, ref, mode, scale=1.0):
    e = rel(got.float(), ref.float())
    print(f"[{name}] {mode} rel_l2={e:.3e} tol={TOL[mode] * scale:.1e}")
    assert math.isfinite(e) and e <= TOL[mode] * scale, f"{name}: {e:.3e}"


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", V2_TILES)
def test_linear_bias_residual_ragged(dev, mode, tile):
    from geo4d_amd import ops, pack
    M, K, N = 1000, 320, 456          # ragged in M and N for every tile; several tiles per workgroup under debug_ablate = 2
    x, w = rnd((M, K), dev, 1).to(act_dtype(mode)), rnd((N, K), dev, 2, 0.05)
    b, r = rnd((N,), dev, 3), rnd((M, N), dev, 4).to(act_dtype(mode))
    wp = pack.pack_linear(w, pack_mode(mode))
    out = both_grids(lambda: ops.linear(x, wp, b, residual=r, tile_hint=tile))
    check(f"linear tile{tile}", out, x.float() @ rounded(w, mode).t() + b + r.float(), mode)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", V2_TILES)
def test_geglu_and_silu(dev, mode, tile):
    from geo4d_amd import ops, pack
    M, K, inner = 700, 256, 320
    x = rnd((M, K), dev, 7).to(act_dtype(mode))
    w, b = rnd((2 * inner, K), dev, 8, 0.1), rnd((2 * inner,), dev, 9)
    wp, bp = pack.pack_geglu(w, b, pack_mode(mode))
    if tile not in GEGLU_TILES:       # wave tiles that cannot pair value / gate blocks: the C ABI must refuse, not mis-compute
        with pytest.raises(RuntimeError):
            ops.linear(x, wp, bp, act=2, tile_hint=tile)
    else:
        out = both_grids(lambda: ops.linear(x, wp, bp, act=2, tile_hint=tile))
        h = x.float() @ rounded(w, mode).t() + b
        check(f"geglu tile{tile}", out, h[:, :inner] * TF.gelu(h[:, inner:]), mode)
    w2 = rnd((200, K), dev, 10, 0.1)
    out = both_grids(lambda: ops.linear(x, pack.pack_linear(w2, pack_mode(mode)), None, act=1, tile_hint=tile))
    check(f"silu tile{tile}", out, TF.silu(x.float() @ rounded(w2, mode).t()), mode)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", V2_TILES)
@pytest.mark.parametrize("cfg", [dict(stride=1, ups=1), dict(stride=2, ups=1), dict(stride=1, ups=2)])
def test_conv3x3_rowbias_residual_split_k(dev, mode, tile, cfg):
    from geo4d_amd import ops, pack
    F, H, W, Ci, Co = 5, 12, 9, 256, 200
    x_nchw = rnd((F, Ci, H, W), dev, 10).to(act_dtype(mode))
    wc, bc = rnd((Co, Ci, 3, 3), dev, 11, 0.03), rnd((Co,), dev, 12)
    emb = rnd((F, Co), dev, 13)
    xt = x_nchw.permute(0, 2, 3, 1).reshape(F * H * W, Ci).contiguous()
    xin = x_nchw.float()
    if cfg["ups"] == 2:
        xin = TF.interpolate(xin, scale_factor=2, mode="nearest")
    ref = TF.conv2d(xin, rounded(wc, mode), bc, stride=cfg["stride"], padding=1) + emb[:, :, None, None]
    Ho, Wo = ref.shape[-2:]
    r = rnd((F * Ho * Wo, Co), dev, 14).to(act_dtype(mode))
    ref = ref.permute(0, 2, 3, 1).reshape(F * Ho * Wo, Co) + r.float()
    wp = pack.pack_conv2d(wc, pack_mode(mode))
    for split in (1, 2):
        o = both_grids(lambda: ops.conv2d(xt, wp, bc, F=F, Hin=H, Win=W, KH=3, KW=3, stride=cfg["stride"], pad=1, ups=cfg["ups"], rowbias=emb,
                                          rowbias_div=Ho * Wo, residual=r, tile_hint=tile, split_k=split)[0])
        check(f"conv tile{tile} {cfg} split{split}", o, ref, mode)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", [22, 23, 25, 28])
def test_temporal_conv(dev, mode, tile):
    from geo4d_amd import ops, pack
    B, T, HW, C = 2, 7, 45, 128
    x = rnd((B * T, HW, C), dev, 20).to(act_dtype(mode))
    w, b = rnd((C, C, 3, 1, 1), dev, 21, 0.05), rnd((C,), dev, 22)
    r = rnd((B * T * HW, C), dev, 23).to(act_dtype(mode))
    out = both_grids(lambda: ops.conv_temporal(x.reshape(B * T * HW, C), pack.pack_conv3d_t(w, pack_mode(mode)), b, B=B, T=T, HW=HW, residual=r, tile_hint=tile))
    x5 = x.float().reshape(B, T, HW, 1, C).permute(0, 4, 1, 2, 3)
    ref = TF.conv3d(x5, rounded(w, mode), b, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(B * T * HW, C) + r.float()
    check(f"temporal tile{tile}", out, ref, mode)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", [22, 25, 28])
def test_unaligned_output_takes_the_scalar_path(dev, mode, tile):
    """N = 77 columns with a row pitch of 77 elements: 4-element vectors are not aligned, the epilogue must fall back to scalar stores."""
    from geo4d_amd import ops, pack
    M, K, N = 333, 128, 77
    x, w, b = rnd((M, K), dev, 30).to(act_dtype(mode)), rnd((N, K), dev, 31, 0.1), rnd((N,), dev, 32)
    r = rnd((M, N), dev, 33).to(act_dtype(mode))
    out = both_grids(lambda: ops.linear(x, pack.pack_linear(w, pack_mode(mode)), b, residual=r, tile_hint=tile))
    check(f"unaligned tile{tile}", out, x.float() @ rounded(w, mode).t() + b + r.float(), mode)


@pytest.mark.parametrize("mode", MODES)
def test_ncthw_and_f32_are_refused(dev, mode):
    from geo4d_amd import ops, pack
    F, H, W, Ci = 2, 5, 8, 64
    x = rnd((F * H * W, Ci), dev, 40).to(act_dtype(mode))
    w3, b3 = rnd((3, Ci, 3, 3), dev, 41, 0.03), rnd((3,), dev, 42)
    with pytest.raises(RuntimeError):
        ops.conv2d(x, pack.pack_conv2d(w3, pack_mode(mode)), b3, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, T=1, out_nchw=True, tile_hint=22)
    xf, wf = rnd((64, 64), dev, 43), rnd((64, 64), dev, 44)
    with pytest.raises(RuntimeError):
        ops.linear(xf, wf, None, tile_hint=22)          # exact-f32 mode


@pytest.mark.parametrize("mode", MODES)
def test_full_chip_persistent_rounds_are_deterministic(dev, mode):
    """More tiles than resident workgroups (64x64 tiles: 2560 of them), long K: every workgroup walks several tiles with the DMA of the
    next tile's first slab in flight under its epilogue. Three runs bit-identical and equal to the 32x32-MFMA kernel up to fp32 order."""
    from geo4d_amd import ops, pack
    M, K, N = 4096, 1024, 2560
    x, w, b = rnd((M, K), dev, 50).to(act_dtype(mode)), rnd((N, K), dev, 51, 0.03), rnd((N,), dev, 52)
    r = rnd((M, N), dev, 53).to(act_dtype(mode))
    wp = pack.pack_linear(w, pack_mode(mode))
    for tile in (28, 25, 22, 23, 27):
        outs = [ops.linear(x, wp, b, residual=r, tile_hint=tile) for _ in range(3)]
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), f"tile {tile}: runs differ"
        check(f"full chip tile{tile}", outs[0], x.float() @ rounded(w, mode).t() + b + r.float(), mode)
        assert rel(outs[0], ops.linear(x, wp, b, residual=r, tile_hint=11)) < 2 * TOL[mode]


@pytest.mark.parametrize("mode", MODES)
def test_batched_gemm_alpha(dev, mode):
    """Batched x3 / 16-bit GEMM (the VAE AttnBlock's Q.K^T form): batch strides, alpha, two activations."""
    from geo4d_amd import ops
    Bz, M, N, K = 3, 200, 136, 128
    a, bt = rnd((Bz, M, K), dev, 60).to(act_dtype(mode)), rnd((Bz, N, K), dev, 61).to(act_dtype(mode))
    out = torch.empty((Bz, M, N), device=dev, dtype=act_dtype(mode))
    for tile in (25, 28):
        both_grids(lambda: ops.conv_gemm(a, bt, out, M=M, N=N, K=K, Cin=K, lda=K, ldw=K, ldo=N, batch=Bz, a_bs=M * K, w_bs=N * K, o_bs=M * N,
                                         alpha=0.25, tile_hint=tile, x3=(mode == "bf16x3")).clone())
        check(f"batched tile{tile}", out, 0.25 * a.float() @ bt.float().transpose(1, 2), mode)
