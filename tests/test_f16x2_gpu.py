"""The two-pass f16 GEMM of the bf16x3m mode (include/geo4d_hip.h dtype 4, round 5; operand layout of round 6): a pre-split
[8 x f16 hi | 8 x f16 lo] weight x PLAIN f16 activation rows, product = a.w_hi + a.w_lo on the second- / third-generation tiles, and its
producers (GroupNorm / LayerNorm / the GEGLU epilogue with split_out = "f16": plain f16 rows, clamped, NaN kept).

What is checked is the ARITHMETIC THE MODE CLAIMS, not a loose tolerance: the reference is PyTorch fp64 math on exactly the operands the
kernel multiplies - the activation rounded to f16, the weight as (f16 hi + f16 lo) / scale - so the only difference left is the fp32
accumulation order (1e-5 relative on the output norm); a kernel that dropped the lo pass, read the wrong half, or mis-scaled would be
off by 1e-4 .. 1 there. Every case also runs on three persistent workgroups (`debug_ablate = 2`: the tile stream crosses tile
boundaries on small shapes), and the third-generation tiles are compared bit for bit with a second-generation tile (same summation
order by construction)."""
import pytest
import torch
import torch.nn.functional as TF

from test_gemm_v2_gpu import both_grids, rel, rnd

pytestmark = pytest.mark.gpu

V2_TILES = [22, 23, 25, 27, 28]
V3_TILES = [71, 72, 73, 74]


def split_f16_act(x):
    """Host twin of store4_f16 (common.h): f32 [M, K] -> the two-pass GEMM's A operand = plain float16 rows [M, K], clamped to the finite
    f16 range (round 5 stored [8 x hi | 8 x lo] groups whose lo half no kernel read; hi is exactly this value)."""
    return x.float().clamp(-65504.0, 65504.0).to(torch.float16).contiguous()


def weight_seen(wp):
    """The weight values a pack.split_f16 operand represents: (hi + lo) * alpha, fp64, [N, K]."""
    n, k2 = wp.shape
    g = wp.reshape(n, k2 // 16, 2, 8).double()
    return ((g[:, :, 0] + g[:, :, 1]).reshape(n, k2 // 2)) * wp._x2_alpha


def a_seen(x):
    return x.float().clamp(-65504.0, 65504.0).to(torch.float16).double()


def close(name, got, ref, tol=1e-5):
    e = rel(got.double(), ref.double())
    assert torch.isfinite(got).all() and e < tol, f"{name}: rel_l2 {e:.3e} (tol {tol:.0e})"


def test_pack_split_f16_represents_the_weight_to_22_bits(dev):
    from geo4d_amd import pack
    for scale in (1e-4, 0.03, 1.0, 300.0):
        w = rnd((96, 256), dev, 1, scale)
        wp = pack.split_f16(w)
        assert wp.dtype == torch.float16 and wp.shape == (96, 512) and torch.isfinite(wp.float()).all()
        e = rel(weight_seen(wp), w.double())
        assert e < 2e-6, f"scale {scale}: hi + lo represents w to {e:.2e}"
        hi_only = wp.reshape(96, 32, 2, 8)[:, :, 0].reshape(96, 256).double() * wp._x2_alpha
        assert rel(hi_only, w.double()) > 1e-4          # (the lo half is doing the work)


@pytest.mark.parametrize("tile", V2_TILES + V3_TILES)
def test_linear_bias_residual_ragged(dev, tile):
    from geo4d_amd import ops, pack
    M, K, N = 1000, 512, 456
    x, w = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.05)
    b, r = rnd((N,), dev, 3), rnd((M, N), dev, 4)
    wp, xs = pack.split_f16(w), split_f16_act(x)
    out = both_grids(lambda: ops.linear(xs, wp, b, residual=r, tile_hint=tile))
    assert out.dtype == torch.float32 and out.shape == (M, N)
    close(f"linear tile{tile}", out, a_seen(x) @ weight_seen(wp).t() + b.double() + r.double())
    if tile >= 71:
        assert torch.equal(out, ops.linear(xs, wp, b, residual=r, tile_hint=25)), f"tile {tile} differs from the second generation"
    # the error the mode accepts: the activation is an f16, nothing else
    full = x.double() @ w.double().t() + b.double() + r.double()
    e = rel(out.double(), full)
    assert 1e-5 < e < 6e-4, f"distance to exact math {e:.2e}: expected the f16 rounding of the activation (~2e-4), not more, not less"


@pytest.mark.parametrize("tile", V3_TILES + [23, 25])
@pytest.mark.parametrize("stride", [1, 2])
def test_conv3x3_rowbias_residual_split_k(dev, tile, stride):
    from geo4d_amd import ops, pack
    F, H, W, Ci, Co = 5, 12, 9, 256, 200
    x_nchw = rnd((F, Ci, H, W), dev, 10)
    wc, bc = rnd((Co, Ci, 3, 3), dev, 11, 0.03), rnd((Co,), dev, 12)
    emb = rnd((F, Co), dev, 13)
    xt = x_nchw.permute(0, 2, 3, 1).reshape(F * H * W, Ci).contiguous()
    wp = pack.pack_conv2d_x2(wc, "bf16x3m")
    ws = weight_seen(wp).reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2)                    # back to [Co, Ci, 3, 3]
    xa = a_seen(xt).reshape(F, H, W, Ci).permute(0, 3, 1, 2)
    ref = TF.conv2d(xa, ws, bc.double(), stride=stride, padding=1) + emb.double()[:, :, None, None]
    Ho, Wo = ref.shape[-2:]
    r = rnd((F * Ho * Wo, Co), dev, 14)
    ref = ref.permute(0, 2, 3, 1).reshape(F * Ho * Wo, Co) + r.double()
    xs = split_f16_act(xt)
    for split in (1, 2, 3):
        def run(t):
            return ops.conv2d(xs, wp, bc, F=F, Hin=H, Win=W, KH=3, KW=3, stride=stride, pad=1, rowbias=emb, rowbias_div=Ho * Wo, residual=r,
                              tile_hint=t, split_k=split)[0]
        o = both_grids(lambda: run(tile))
        close(f"conv tile{tile} stride{stride} split{split}", o, ref)
        if tile >= 71:
            assert torch.equal(o, run(25)), f"conv tile{tile} split{split}: differs from the second generation"


def test_default_tile_and_fallbacks(dev):
    """tile_hint 0 (the library picks), an odd K-slab count on a third-generation hint (runs on its second-generation twin), and what
    the C ABI must refuse: first-generation hints, raw (un-split) operands, a pre-split output."""
    from geo4d_amd import ops, pack
    M, N = 640, 128
    for K in (160, 320, 1024):
        x, w, b = rnd((M, K), dev, 30), rnd((N, K), dev, 31, 0.1), rnd((N,), dev, 32)
        wp, xs = pack.split_f16(w), split_f16_act(x)
        ref = a_seen(x) @ weight_seen(wp).t() + b.double()
        for tile in (0, 72, 74):
            close(f"K{K} tile{tile}", ops.linear(xs, wp, b, tile_hint=tile, split_k=1 if tile else 0), ref)
    with pytest.raises(RuntimeError):
        ops.linear(xs, wp, b, tile_hint=1, split_k=1)
    with pytest.raises(AssertionError):
        ops.linear(xs, wp, b, split_out=True)                       # the two-pass GEMM writes plain f32 or plain f16 rows
    with pytest.raises(AssertionError):
        ops.linear(xs, pack.split_bf16(w), b)                       # an f16 activation needs an f16-split weight
    with pytest.raises(AssertionError):
        ops.linear(x, wp, b)                                        # ... and an f16-split weight an f16 activation (no raw-f32 launch)
    # a non-GEGLU launch writing f16 rows (o_split = 2) with the library's own tile choice, M >= 4096 (the default used to
    # pick the 160x320 tile, which has no f16-row epilogue -> EINVAL)
    big = rnd((4608, 1024), dev, 33)
    old = ops.AUTOTUNE
    try:
        ops.AUTOTUNE = False
        ops._tune_table().pop("4/0|4608x128x1024|c1024|t111s1u1|a0r0n0|b1|x11o", None)
        g = ops.linear(split_f16_act(big), wp, b, split_out="f16")
    finally:
        ops.AUTOTUNE = old
    assert g.dtype == torch.float16 and g.shape == (4608, N)
    want = (a_seen(big) @ weight_seen(wp).t() + b.double())
    assert bool(((g.double() - want).abs() <= want.abs() * 2.0 ** -11 + 1e-5 * float(want.abs().max())).all())


def test_saturation_counter_and_nan_propagation(dev):
    """The f16-clamping stores (GroupNorm / LayerNorm / GEGLU epilogue writing the two-pass GEMM's operand): values beyond +-65504 are clamped AND
    counted when ops.SAT_COUNTER is set (the full-size tests assert the count stays 0); a NaN stays a NaN (fminf(fmaxf()) laundered it
    into -65504, hiding a diverged activation from every isfinite check downstream)."""
    from geo4d_amd import ops, pack
    M, C = 256, 64
    x = rnd((M, C), dev, 90)
    x[3, 5], x[7, 9], x[100, 1] = 3.0e6, -4.0e6, float("nan")          # LayerNorm of a row with one 3e6 entry: ~8 x gamma, so scale gamma up
    gamma, beta = torch.full((C,), 3.0e4, device=dev), torch.zeros((C,), device=dev)
    cnt = torch.zeros(1, device=dev, dtype=torch.int64)
    old = ops.SAT_COUNTER
    try:
        ops.SAT_COUNTER = cnt
        y = ops.layernorm(x, gamma, beta, split_out="f16")
        n_ln = int(cnt.item())
        plain = ops.layernorm(x, gamma, beta)
        assert torch.isnan(plain[100]).all() and torch.isnan(y[100].float()).all(), "NaN must survive the clamp"
        big = plain.abs() > 65504.0
        assert big.any() and n_ln >= 1 and bool((y.float().abs()[big] == 65504.0).all()) and torch.isfinite(y[:100].float()).all()
        # GroupNorm: one frame, gamma large -> saturates; counter moves on
        g = ops.groupnorm(x[:100].contiguous(), gamma[:C], beta[:C], F=1, HW=100, eps=1e-5, groups=32, split_out="f16")
        assert int(cnt.item()) > n_ln and float(g.float().abs().max()) == 65504.0
        # GEGLU epilogue writing f16 rows: a huge bias on the value half saturates the product
        n0 = int(cnt.item())
        K, inner = 64, 64
        w, b = rnd((2 * inner, K), dev, 91, 0.1), torch.zeros((2 * inner,), device=dev)
        b[:inner] = 1.0e5; b[inner:] = 10.0
        wp, bp = pack.pack_geglu_x2(w, b, "bf16x3m")
        o = ops.linear(split_f16_act(rnd((128, K), dev, 92)), wp, bp, act=2, split_out="f16", tile_hint=25, split_k=1)
        assert int(cnt.item()) > n0 and float(o.float().abs().max()) == 65504.0
        ops.SAT_COUNTER = None
        n1 = int(cnt.item())
        ops.layernorm(x, gamma, beta, split_out="f16")
        assert int(cnt.item()) == n1, "no counter, no counting"
    finally:
        ops.SAT_COUNTER = old


def test_outlier_heavy_weights(dev):
    """real checkpoints have weight outliers, and pack.split_f16 scales per TENSOR - the lo halves of small weights in
    a tensor with a 1000x outlier are f16 subnormals. What that costs (pack.split_f16 docstring): hi + lo keeps an absolute error of
    2^-38 |w|_max. Heavy-tailed weights (Student-t, 2 dof, + planted 1000x outliers) and activations with 50x outlier channels: the
    operand still represents the weight to 2^-20 of each ROW's norm, and the GEMM matches fp64 math on the operands it multiplies."""
    from geo4d_amd import ops, pack
    M, K, N = 512, 1024, 256
    g = torch.Generator(device="cpu").manual_seed(7)
    t = torch.distributions.StudentT(2.0).sample((N, K)).clamp(-200, 200) * 0.02
    t[::7, ::131] *= 1000.0
    w = t.to(dev)
    x = rnd((M, K), dev, 95)
    x[:, ::97] *= 50.0
    wp = pack.split_f16(w)
    rep = weight_seen(wp)
    row_err = (rep - w.double()).norm(dim=1) / w.double().norm(dim=1)
    assert float(row_err.max()) < 2.0 ** -20, f"weight rows represented to {float(row_err.max()):.2e}"
    # per weight: 2^-22 relative while lo is a normal f16, 2^-38 |w|_max absolute once it is subnormal (one factor 2 of slack on each)
    amax = float(w.abs().max())
    assert bool(((rep - w.double()).abs() <= w.double().abs() * 2.0 ** -21 + amax * 2.0 ** -37).all())
    assert bool((w.abs() < amax * 2.0 ** -14).any()), "the test must contain weights whose lo half is subnormal"
    out = ops.linear(split_f16_act(x), wp, None)
    close("outlier GEMM vs fp64 on its operands", out, a_seen(x) @ rep.t())
    full = x.double() @ w.double().t()
    assert rel(out.double(), full) < 6e-4


@pytest.mark.parametrize("fps", [1, 4])
def test_groupnorm_f16_split_output_and_fused_statistics(dev, fps):
    """GroupNorm(+SiLU) writing the f16 pre-split format == the host split of the plain-f32 GroupNorm of the same input; then the whole
    chain the networks run: GroupNorm -> two-pass conv (emits the next GroupNorm's column sums) -> GroupNorm from those sums."""
    from geo4d_amd import ops, pack
    F, H, W, C = 4, 10, 16, 320
    x = rnd((F * H * W, C), dev, 40) * 3.0 + 0.5
    gamma, beta = rnd((C,), dev, 41) + 1.0, rnd((C,), dev, 42)
    plain = ops.groupnorm(x, gamma, beta, F=F, HW=H * W, eps=1e-5, frames_per_stat=fps, silu=True)
    sp = ops.groupnorm(x, gamma, beta, F=F, HW=H * W, eps=1e-5, frames_per_stat=fps, silu=True, split_out="f16")
    assert not isinstance(sp, ops.SplitAct) and sp.dtype == torch.float16 and sp.shape == (F * H * W, C)
    assert torch.equal(sp.view(torch.int16), split_f16_act(plain).view(torch.int16)), "f16 output differs from f16(plain GroupNorm)"
    wc, bc = rnd((C, C, 3, 3), dev, 43, 0.02), rnd((C,), dev, 44)
    wp = pack.pack_conv2d_x2(wc, "bf16x3m")
    old = ops.GN_FUSED_STATS
    try:
        outs = {}
        for fused in (0, 1):
            ops.GN_FUSED_STATS = fused
            h, _, _ = ops.conv2d(sp, wp, bc, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, gn_stats=True, tile_hint=72, split_k=1)
            assert (getattr(h, "_gn_colsum", None) is not None) == bool(fused)
            outs[fused] = (h, ops.groupnorm(h, gamma, beta, F=F, HW=H * W, eps=1e-5, frames_per_stat=fps, silu=True))
    finally:
        ops.GN_FUSED_STATS = old
    assert torch.equal(outs[0][0], outs[1][0])
    close("GroupNorm from the two-pass conv's column sums", outs[1][1], outs[0][1], tol=2e-6)
    ws = weight_seen(wp).reshape(C, 3, 3, C).permute(0, 3, 1, 2)
    ref = TF.conv2d(a_seen(plain).reshape(F, H, W, C).permute(0, 3, 1, 2), ws, bc.double(), padding=1).permute(0, 2, 3, 1).reshape(F * H * W, C)
    close("GroupNorm -> two-pass conv", outs[0][0], ref)


def test_full_chip_stream_is_deterministic(dev):
    """The U-Net's largest 3x3 convolution (M = 40960, K = 2880, N = 320) on every tile that can serve it: five launches bit-identical,
    third == second generation bit for bit."""
    from geo4d_amd import ops, pack
    F, H, W, C = 16, 40, 64, 320
    xt = rnd((F * H * W, C), dev, 50)
    wc, bc = rnd((C, C, 3, 3), dev, 51, 0.02), rnd((C,), dev, 52)
    wp, xs = pack.pack_conv2d_x2(wc, "bf16x3m"), split_f16_act(xt)
    base = None
    for tile in (23, 71, 72, 74):
        outs = [ops.conv2d(xs, wp, bc, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, tile_hint=tile, split_k=1)[0].clone() for _ in range(5)]
        assert all(torch.equal(outs[0], o) for o in outs[1:]), f"tile {tile}: launches differ"
        base = outs[0] if base is None else base
        assert torch.equal(outs[0], base), f"tile {tile} differs from tile 23"
    ws = weight_seen(wp).reshape(C, 3, 3, C).permute(0, 3, 1, 2)
    sub = slice(0, 2)                                                     # two frames of the fp64 reference are enough
    ref = TF.conv2d(a_seen(xt).reshape(F, H, W, C)[sub].permute(0, 3, 1, 2), ws, bc.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, C)
    close("L0 conv3x3", base[: 2 * H * W], ref)


def split_halves(t):
    """(hi, lo) fp64 [M, K] of a pre-split [M, 2K] tensor (either format)."""
    m, k2 = t.shape
    g = t.as_subclass(torch.Tensor).reshape(m, k2 // 16, 2, 8).double()
    return g[:, :, 0].reshape(m, k2 // 2), g[:, :, 1].reshape(m, k2 // 2)


@pytest.mark.parametrize("C", [320, 640, 1280])
def test_layernorm_f16_split_output(dev, C):
    from geo4d_amd import ops
    M = 777
    x = rnd((M, C), dev, 60) * 2.0 + 0.3
    gamma, beta = rnd((C,), dev, 61) + 1.0, rnd((C,), dev, 62)
    plain = ops.layernorm(x, gamma, beta)
    sp = ops.layernorm(x, gamma, beta, split_out="f16")
    assert not isinstance(sp, ops.SplitAct) and sp.dtype == torch.float16 and sp.shape == (M, C)
    assert torch.equal(sp.view(torch.int16), split_f16_act(plain).view(torch.int16)), "f16 output differs from f16(plain LayerNorm)"
    b16 = ops.layernorm(x, gamma, beta, split_out=True)                       # the bf16 format is unchanged
    hi, lo = split_halves(b16)
    assert b16.dtype == torch.bfloat16 and rel(hi + lo, plain.double()) < 1e-5


@pytest.mark.parametrize("tile", [23, 25, 72, 74])
def test_temporal_conv_two_pass(dev, tile):
    from geo4d_amd import ops, pack
    B, T, HW, C = 2, 7, 45, 128
    x = rnd((B * T * HW, C), dev, 70)
    w, b = rnd((C, C, 3, 1, 1), dev, 71, 0.05), rnd((C,), dev, 72)
    r = rnd((B * T * HW, C), dev, 73)
    wp = pack.pack_conv3d_t_x2(w, "bf16x3m")
    out = both_grids(lambda: ops.conv_temporal(split_f16_act(x), wp, b, B=B, T=T, HW=HW, residual=r, tile_hint=tile, split_k=1))
    ws = weight_seen(wp).reshape(C, 3, C).permute(0, 2, 1).reshape(C, C, 3, 1, 1)           # [Cout, 3, Cin] -> [Cout, Cin, 3, 1, 1]
    x5 = a_seen(x).reshape(B, T, HW, 1, C).permute(0, 4, 1, 2, 3)
    ref = TF.conv3d(x5, ws, b.double(), padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(B * T * HW, C) + r.double()
    close(f"temporal conv tile{tile}", out, ref)


@pytest.mark.parametrize("tile", [0, 25, 27, 71, 74])
def test_geglu_feed_forward_chain_two_pass(dev, tile):
    """LayerNorm (f16 rows) -> two-pass GEGLU linear writing f16 rows (o_split = 2) -> two-pass ff-out with the residual: the chain
    unet._ff runs in the bf16x3m mode, each stage against fp64 math on the operands it reads (the ff-out sees the f16 the GEGLU
    epilogue stored). Tiles whose wave tiles cannot pair value / gate blocks must be refused."""
    from geo4d_amd import ops, pack
    M, K, inner = 700, 256, 320
    x = rnd((M, K), dev, 80)
    w, b = rnd((2 * inner, K), dev, 81, 0.1), rnd((2 * inner,), dev, 82)
    w2, b2 = rnd((K, inner), dev, 83, 0.1), rnd((K,), dev, 84)
    wp, bp = pack.pack_geglu_x2(w, b, "bf16x3m")
    wp2 = pack.pack_linear_x2(w2, "bf16x3m")
    xs = split_f16_act(x)
    g = both_grids(lambda: ops.linear(xs, wp, bp, act=2, split_out="f16", tile_hint=tile, split_k=1 if tile else 0))
    assert not isinstance(g, ops.SplitAct) and g.dtype == torch.float16 and g.shape == (M, inner)
    perm = pack.geglu_perm(inner, dev)
    wu = torch.empty((2 * inner, K), device=dev, dtype=torch.float64)
    wu[perm] = weight_seen(wp)
    h = a_seen(x) @ wu.t() + b.double()
    ref = h[:, :inner] * TF.gelu(h[:, inner:])
    hi = g.double()
    # the stored f16 is the fp32 epilogue value rounded once: within half an f16 ulp of the fp64 reference (+ the fp32 accumulation error)
    assert bool(((hi - ref).abs() <= ref.abs() * 2.0 ** -11 + 1e-5 * float(ref.abs().max()) + 2.0 ** -24).all()), "GEGLU f16 rows are further than half an ulp from fp64 math"
    y = both_grids(lambda: ops.linear(g, wp2, b2, residual=x, tile_hint=tile if tile != 27 else 25, split_k=1 if tile else 0))
    close(f"ff-out tile{tile}", y, hi @ weight_seen(wp2).t() + b2.double() + x.double())
    for bad in (23, 72, 73):
        with pytest.raises(RuntimeError):
            ops.linear(xs, wp, bp, act=2, split_out="f16", tile_hint=bad, split_k=1)


@pytest.mark.parametrize("H,W,rows", [(10, 16, 32), (5, 8, 8)])
def test_split_k_launch_emits_groupnorm_sums_from_its_reduce(dev, H, W, rows):
    """Round 6: a split-K launch of the second / third generation emits the consumer GroupNorm's column sums from its reduce launch
    (splitk_reduce_colsum_kernel: per 32 rows, or per 8 where a frame has 40 rows - the 5 x 8 level), so the GroupNorms behind the level-2 / 3
    convolutions skip their statistics pass like the others. The conv output is bit-identical to the plain reduce kernel's; the GroupNorm from
    the sums equals the one that makes its own pass to fp32 round-off (per-frame and 5-D statistics)."""
    from geo4d_amd import ops, pack
    F, C = 16, 1280
    xt = rnd((F * H * W, C), dev, 100).to(torch.float16).contiguous()
    wp, bc = pack.pack_conv2d_x2(rnd((C, C, 3, 3), dev, 101, 0.01), "bf16x3m"), rnd((C,), dev, 102)
    emb, r = rnd((1, C), dev, 103), rnd((F * H * W, C), dev, 104)
    gamma, beta = rnd((C,), dev, 105) + 1.0, rnd((C,), dev, 106)
    for tile in (23, 72, 73):
        run = lambda gs: ops.conv2d(xt, wp, bc, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, rowbias=emb, rowbias_div=F * H * W, residual=r, tile_hint=tile,
                                    split_k=4, gn_stats=gs)[0]
        plain, h = run(False), run(True)
        assert torch.equal(h, plain), f"tile {tile}"
        assert getattr(h, "_gn_colsum", None) is not None and h._gn_colsum_rows == rows and tuple(h._gn_colsum.shape) == (F * H * W // rows, C, 2)
        want = plain.double().reshape(-1, rows, C)
        assert rel(h._gn_colsum[:, :, 0], want.sum(1)) < 1e-5 and rel(h._gn_colsum[:, :, 1], (want * want).sum(1)) < 1e-5
        for fps in (1, 16):
            a = ops.groupnorm(h, gamma, beta, F=F, HW=H * W, eps=1e-5, frames_per_stat=fps, silu=True, split_out="f16")
            b = ops.groupnorm(plain.clone(), gamma, beta, F=F, HW=H * W, eps=1e-5, frames_per_stat=fps, silu=True, split_out="f16")
            assert rel(a.float(), b.float()) < 1e-4 and float((a.float() - b.float()).abs().max()) < 0.02       # (f16 outputs: an ulp here and there)


def test_cast_rows_f16(dev):
    """geo4d_cast_rows_f16 (class `vaeup`: an f16 copy of an f32 stream as a two-pass GEMM's A operand): f16(clamp(x)) element for element,
    pitched input rows, NaN kept, clamped lanes counted."""
    from geo4d_amd import ops
    wide = rnd((300, 512), dev, 110) * 100.0
    x = wide[:, 128:384]                                     # a column view: row pitch 512
    x[5, 7], x[9, 0], x[11, 3] = 1.0e6, -2.0e6, float("nan")
    cnt = torch.zeros(1, device=dev, dtype=torch.int64)
    old = ops.SAT_COUNTER
    try:
        ops.SAT_COUNTER = cnt
        y = ops.cast_f16(x)
    finally:
        ops.SAT_COUNTER = old
    want = x.clamp(-65504.0, 65504.0).to(torch.float16)
    ok = ~torch.isnan(x)
    assert y.dtype == torch.float16 and y.shape == x.shape and torch.equal(y[ok], want[ok]) and bool(torch.isnan(y[11, 3])) and int(cnt.item()) == 2
