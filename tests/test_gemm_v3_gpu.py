"""Third-generation conv_gemm kernel (tile hints 71..74: the second generation's MFMA form, LDS image and register epilogue under a
phased, software-pipelined K loop with counted DMA waits and buffer-resource staging; geo4d_amd/csrc/gemm_kernel_v3.h) through the C ABI.

The K order and the per-accumulator summation order are those of the second generation, so every case is checked TWICE: against plain
PyTorch fp32 math, and bit for bit against a second-generation tile (hint 25) on the same operands. Each case also runs with
`debug_ablate = 2` (3 persistent workgroups: the staging cursors cross tile boundaries - the next tile's buffer windows and first slabs
are issued inside the current tile's last slabs - on small shapes). The races this schedule could have (a fragment
read before its half panel landed, a half panel re-staged under a late reader) do not show as a fixed wrong answer, so the full-chip
case repeats launches and compares them bit for bit."""
import math

import pytest
import torch
import torch.nn.functional as TF

from test_gemm_v2_gpu import MODES, TOL, act_dtype, both_grids, check, pack_mode, rel, rnd, rounded

pytestmark = pytest.mark.gpu

V3_TILES = [71, 72, 73, 74]      # 192x256, 160x320, 256x128, 128x256 on 2 x 4 waves
GEGLU_TILES = {71, 74}           # wave tiles a multiple of 64 columns wide


def same_as_v2(name, got, fn_v2):
    ref = fn_v2()
    assert torch.equal(got, ref), f"{name}: differs from the second-generation kernel ({rel(got.float(), ref.float()):.3e})"


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", V3_TILES)
def test_linear_bias_residual_ragged(dev, mode, tile):
    from geo4d_amd import ops, pack
    M, K, N = 1000, 512, 456          # ragged in M and N for every tile; several tiles per workgroup under debug_ablate = 2; 8 / 16 K slabs
    x, w = rnd((M, K), dev, 1).to(act_dtype(mode)), rnd((N, K), dev, 2, 0.05)
    b, r = rnd((N,), dev, 3), rnd((M, N), dev, 4).to(act_dtype(mode))
    wp = pack.pack_linear(w, pack_mode(mode))
    out = both_grids(lambda: ops.linear(x, wp, b, residual=r, tile_hint=tile))
    check(f"linear tile{tile}", out, x.float() @ rounded(w, mode).t() + b + r.float(), mode)
    same_as_v2(f"linear tile{tile}", out, lambda: ops.linear(x, wp, b, residual=r, tile_hint=25))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", V3_TILES)
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
        same_as_v2(f"geglu tile{tile}", out, lambda: ops.linear(x, wp, bp, act=2, tile_hint=25))
    w2 = rnd((200, K), dev, 10, 0.1)
    out = both_grids(lambda: ops.linear(x, pack.pack_linear(w2, pack_mode(mode)), None, act=1, tile_hint=tile))
    check(f"silu tile{tile}", out, TF.silu(x.float() @ rounded(w2, mode).t()), mode)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", V3_TILES)
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
    for split in (1, 2, 3):          # 72 (bf16x3) / 36 (16-bit) slabs: every split divides them evenly
        def run(t):
            return ops.conv2d(xt, wp, bc, F=F, Hin=H, Win=W, KH=3, KW=3, stride=cfg["stride"], pad=1, ups=cfg["ups"], rowbias=emb,
                              rowbias_div=Ho * Wo, residual=r, tile_hint=t, split_k=split)[0]
        o = both_grids(lambda: run(tile))
        check(f"conv tile{tile} {cfg} split{split}", o, ref, mode)
        same_as_v2(f"conv tile{tile} {cfg} split{split}", o, lambda: run(25))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", V3_TILES)
def test_temporal_conv(dev, mode, tile):
    from geo4d_amd import ops, pack
    B, T, HW, C = 2, 7, 45, 128
    x = rnd((B * T, HW, C), dev, 20).to(act_dtype(mode))
    w, b = rnd((C, C, 3, 1, 1), dev, 21, 0.05), rnd((C,), dev, 22)
    r = rnd((B * T * HW, C), dev, 23).to(act_dtype(mode))
    wp = pack.pack_conv3d_t(w, pack_mode(mode))
    out = both_grids(lambda: ops.conv_temporal(x.reshape(B * T * HW, C), wp, b, B=B, T=T, HW=HW, residual=r, tile_hint=tile))
    x5 = x.float().reshape(B, T, HW, 1, C).permute(0, 4, 1, 2, 3)
    ref = TF.conv3d(x5, rounded(w, mode), b, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(B * T * HW, C) + r.float()
    check(f"temporal tile{tile}", out, ref, mode)
    same_as_v2(f"temporal tile{tile}", out, lambda: ops.conv_temporal(x.reshape(B * T * HW, C), wp, b, B=B, T=T, HW=HW, residual=r, tile_hint=25))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", [71, 72])
def test_launches_the_phased_stream_cannot_take_fall_back(dev, mode, tile):
    """One K slab per tile (K = 32 f32 / 64 16-bit elements), an odd slab count, an uneven split-K, an unaligned output pitch: the library
    routes these to the second-generation tile of the same shape instead of refusing."""
    from geo4d_amd import ops, pack
    M = 333
    for K, N, split in ((64 if mode != "bf16x3" else 32, 128, 1), (320 if mode != "bf16x3" else 160, 128, 1), (448, 128, 4), (128, 77, 1)):
        x, w, b = rnd((M, K), dev, 30).to(act_dtype(mode)), rnd((N, K), dev, 31, 0.1), rnd((N,), dev, 32)
        r = rnd((M, N), dev, 33).to(act_dtype(mode))
        out = both_grids(lambda: ops.linear(x, pack.pack_linear(w, pack_mode(mode)), b, residual=r, tile_hint=tile, split_k=split))
        check(f"fallback tile{tile} K{K} N{N} split{split}", out, x.float() @ rounded(w, mode).t() + b + r.float(), mode)


@pytest.mark.parametrize("mode", MODES)
def test_ncthw_and_f32_are_refused(dev, mode):
    from geo4d_amd import ops, pack
    F, H, W, Ci = 2, 5, 8, 64
    x = rnd((F * H * W, Ci), dev, 40).to(act_dtype(mode))
    w3, b3 = rnd((3, Ci, 3, 3), dev, 41, 0.03), rnd((3,), dev, 42)
    with pytest.raises(RuntimeError):
        ops.conv2d(x, pack.pack_conv2d(w3, pack_mode(mode)), b3, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, T=1, out_nchw=True, tile_hint=71)
    xf, wf = rnd((64, 64), dev, 43), rnd((64, 64), dev, 44)
    with pytest.raises(RuntimeError):
        ops.linear(xf, wf, None, tile_hint=71)          # exact-f32 mode


@pytest.mark.parametrize("mode", MODES)
def test_full_chip_stream_is_deterministic(dev, mode):
    """More tiles than workgroups (M = 8192, N = 2560 on 128x256 tiles: 640 of them on 256 CUs), long K, every workgroup streams several
    tiles with two slabs of DMA in flight across the tile boundaries. Five launches per tile shape bit-identical, equal to the
    second-generation kernel bit for bit and to fp32 math within the mode's tolerance; also with split-K."""
    from geo4d_amd import ops, pack
    M, K, N = 8192, 1024, 2560
    x, w, b = rnd((M, K), dev, 50).to(act_dtype(mode)), rnd((N, K), dev, 51, 0.03), rnd((N,), dev, 52)
    r = rnd((M, N), dev, 53).to(act_dtype(mode))
    wp = pack.pack_linear(w, pack_mode(mode))
    ref = x.float() @ rounded(w, mode).t() + b + r.float()
    v2 = ops.linear(x, wp, b, residual=r, tile_hint=25)
    for tile in V3_TILES:
        for split in (1, 2):
            outs = [ops.linear(x, wp, b, residual=r, tile_hint=tile, split_k=split) for _ in range(5)]
            torch.cuda.synchronize()
            assert all(torch.equal(outs[0], o) for o in outs[1:]), f"tile {tile} split {split}: runs differ"
            check(f"full chip tile{tile} split{split}", outs[0], ref, mode)
            if split == 1:
                assert torch.equal(outs[0], v2), f"tile {tile}: differs from the second-generation kernel"


def test_presplit_operands_and_split_output(dev):
    """bf16x3 with the activation in the producers' pre-split format (a_split) and the GEGLU epilogue writing it (o_split)."""
    from geo4d_amd import ops, pack
    from test_presplit_gpu import decode_split, make_split
    M, K, N = 2000, 640, 512
    x, w, b = rnd((M, K), dev, 60), rnd((N, K), dev, 61, 0.05), rnd((N,), dev, 62)
    r = rnd((M, N), dev, 63)
    wp = pack.pack_linear(w, "bf16x3")
    for tile in V3_TILES:
        raw = both_grids(lambda: ops.linear(x, wp, b, residual=r, tile_hint=tile))
        pre = both_grids(lambda: ops.linear(make_split(x), wp, b, residual=r, tile_hint=tile))
        assert torch.equal(raw, pre), f"tile {tile}"
    F, H, W, Ci, Co = 3, 10, 12, 128, 96
    xc = rnd((F * H * W, Ci), dev, 64)
    wc, bc = pack.pack_conv2d(rnd((Co, Ci, 3, 3), dev, 65, 0.03), "bf16x3"), rnd((Co,), dev, 66)
    for tile in V3_TILES:
        a = ops.conv2d(xc, wc, bc, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, tile_hint=tile)[0]
        c = both_grids(lambda: ops.conv2d(make_split(xc), wc, bc, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, tile_hint=tile)[0])
        assert torch.equal(a, c) and torch.equal(a, ops.conv2d(xc, wc, bc, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, tile_hint=25)[0]), f"conv tile {tile}"
    inner = 640
    wg, bg = rnd((2 * inner, K), dev, 67, 0.05), rnd((2 * inner,), dev, 68)
    wgp, bgp = pack.pack_geglu(wg, bg, "bf16x3")
    for tile in sorted(GEGLU_TILES):
        plain = ops.linear(make_split(x), wgp, bgp, act=2, tile_hint=tile)
        split = both_grids(lambda: ops.linear(make_split(x), wgp, bgp, act=2, tile_hint=tile, split_out=True).as_subclass(torch.Tensor))
        hi, lo = decode_split(split)
        assert torch.equal(hi, plain.to(torch.bfloat16).float()) and ((hi + lo - plain).norm() / plain.norm()).item() < 2e-5, tile
        assert torch.equal(split, ops.linear(make_split(x), wgp, bgp, act=2, tile_hint=22, split_out=True).as_subclass(torch.Tensor)), tile
        assert math.isfinite(plain.sum().item())


@pytest.mark.parametrize("mode", MODES)
def test_batched_gemm_alpha(dev, mode):
    """Batched GEMM with both operands raw (the VAE AttnBlock's Q.K^T form; bf16x3 splits both fragments in registers): batch strides,
    alpha, the workgroups' tile stream crossing batch boundaries."""
    from geo4d_amd import ops
    Bz, M, N, K = 3, 200, 136, 256
    a, bt = rnd((Bz, M, K), dev, 60).to(act_dtype(mode)), rnd((Bz, N, K), dev, 61).to(act_dtype(mode))

    def run(tile):
        out = torch.empty((Bz, M, N), device=dev, dtype=act_dtype(mode))
        ops.conv_gemm(a, bt, out, M=M, N=N, K=K, Cin=K, lda=K, ldw=K, ldo=N, batch=Bz, a_bs=M * K, w_bs=N * K, o_bs=M * N, alpha=0.25,
                      tile_hint=tile, x3=(mode == "bf16x3"))
        return out
    ref = run(25)
    check("batched v2", ref, 0.25 * a.float() @ bt.float().transpose(1, 2), mode)
    for tile in V3_TILES:
        assert torch.equal(both_grids(lambda: run(tile)), ref), f"tile {tile}"
