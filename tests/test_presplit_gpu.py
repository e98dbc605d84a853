"""bf16x3 producer / consumer chain in the PRE-SPLIT operand format (rounds 3-4).

Producers (GroupNorm, LayerNorm, both attention kernels, the GEGLU epilogue) can write their f32 result as [8 x bf16 hi | 8 x bf16 lo]
per 8 channels (ops.SplitAct); conv_gemm then multiplies it without splitting fragments in its K loop. Checked here:
  * every producer's split output decodes (hi + lo) to its own plain f32 output within bf16x3 resolution, and hi is exactly bf16(f32);
  * conv_gemm on a SplitAct A operand (linear / 3x3 conv with gather + zero padding / temporal conv / operand-swapped V^T projection)
    equals conv_gemm on the same values given as raw f32 BIT FOR BIT (the split happens before the MFMA either way), on first- and
    second-generation tiles;
  * (round 4) every GEMM epilogue can write the pre-split format (q | k and V^T projections), and the spatial attention kernel fed with
    pre-split q / K / V^T equals the kernel fed with raw f32 bit for bit."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def rnd(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def decode_split(t):
    """SplitAct [M, 2K] bf16 -> (hi, lo) as f32 [M, K]."""
    m, k2 = t.shape
    v = t.as_subclass(torch.Tensor).reshape(m, k2 // 16, 2, 8).float()
    return v[:, :, 0].reshape(m, k2 // 2), v[:, :, 1].reshape(m, k2 // 2)


def check_split(name, split, plain):
    hi, lo = decode_split(split)
    assert torch.equal(hi, plain.to(torch.bfloat16).float()), f"{name}: hi is not bf16(value)"
    err = ((hi + lo - plain).norm() / plain.norm()).item()
    print(f"[{name}] split decode rel_l2 = {err:.3e}")
    assert err < 2e-5


def make_split(x):
    """Host-side reference packer (pack.split_bf16 is the weight-side twin)."""
    from geo4d_amd import ops, pack
    return ops.SplitAct.wrap(pack.split_bf16(x))


@pytest.mark.parametrize("case", [(4, 6 * 7, 320, 1, True), (6, 5 * 4, 64, 3, False), (2, 70 * 33, 128, 1, True), (16, 160, 1280, 16, True)])
def test_groupnorm_split_output(dev, case):
    from geo4d_amd import ops
    F, HW, C, fps, silu = case
    x = rnd((F * HW, C), dev, 1) * 2 + 0.5
    g, b = rnd((C,), dev, 2) + 1, rnd((C,), dev, 3)
    y3 = ops.groupnorm(x, g, b, F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=silu)
    y3b = ops.groupnorm(x, g, b, F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=silu)
    s3 = ops.groupnorm(x, g, b, F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=silu, split_out=True)
    assert torch.equal(y3, y3b), "GroupNorm is order-deterministic"
    ref = TF.group_norm(x.reshape(F // fps, fps * HW, C).permute(0, 2, 1), 32, g, b, 1e-5).permute(0, 2, 1).reshape(F * HW, C)
    if silu:
        ref = TF.silu(ref)
    assert ((y3 - ref).norm() / ref.norm()).item() < 2e-5
    check_split(f"groupnorm {case}", s3, y3)


def test_groupnorm_vae_sized_back_to_back(dev):
    """VAE-sized tensor (48 frames x 40x64 x 512 channels), several GroupNorms in a row on one stream: bit-identical every time."""
    from geo4d_amd import ops
    F, HW, C = 48, 2560, 512
    x = rnd((F * HW, C), dev, 5)
    g, b = rnd((C,), dev, 6) + 1, rnd((C,), dev, 7)
    outs = [ops.groupnorm(x, g, b, F=F, HW=HW, eps=1e-6, silu=True) for _ in range(4)]
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    ref = TF.silu(TF.group_norm(x.reshape(F, HW, C).permute(0, 2, 1), 32, g, b, 1e-6).permute(0, 2, 1).reshape(F * HW, C))
    assert ((outs[0] - ref).norm() / ref.norm()).item() < 2e-5


@pytest.mark.parametrize("C", [320, 640, 1280, 512])
def test_layernorm_split(dev, C):
    from geo4d_amd import ops
    x = rnd((777, C), dev, 10) * 3
    g, b = rnd((C,), dev, 11) + 1, rnd((C,), dev, 12)
    check_split(f"layernorm {C}", ops.layernorm(x, g, b, split_out=True), ops.layernorm(x, g, b))


def test_attention_kernels_split_output(dev):
    from geo4d_amd import ops
    B, H, N = 2, 5, 200
    C_ = H * 64
    qkv = rnd((B * N, 3 * C_), dev, 20)
    Np = (N + 3) // 4 * 4
    vt = torch.zeros((B, C_, Np), device=dev)
    vt[:, :, :N] = qkv[:, 2 * C_:].reshape(B, N, C_).permute(0, 2, 1)
    kv = [(qkv[:, C_:2 * C_], vt.reshape(-1, Np), N, 1, C_ * Np)]
    for x3 in (True, False):
        plain = ops.attention(qkv[:, :C_], kv, B=B, H=H, Nq=N, scale=0.125, x3=x3)
        check_split(f"flash x3={x3}", ops.attention(qkv[:, :C_], kv, B=B, H=H, Nq=N, scale=0.125, x3=x3, split_out=True), plain)
    Bt, T, HW = 1, 16, 37
    q3 = rnd((Bt * T * HW, 3 * C_), dev, 21)
    plain = ops.temporal_attention(q3[:, :C_], q3[:, C_:2 * C_], q3[:, 2 * C_:], B=Bt, T=T, HW=HW, H=H, scale=0.125)
    check_split("temporal", ops.temporal_attention(q3[:, :C_], q3[:, C_:2 * C_], q3[:, 2 * C_:], B=Bt, T=T, HW=HW, H=H, scale=0.125, split_out=True), plain)


@pytest.mark.parametrize("tile", [0, 1, 3, 11, 13, 16, 17, 22, 23, 25, 27, 28, 71, 72, 73, 74])
def test_gemm_on_presplit_activations_equals_raw(dev, tile):
    from geo4d_amd import ops, pack
    M, K, N = 1000, 320, 456
    x, w, b = rnd((M, K), dev, 30), rnd((N, K), dev, 31, 0.05), rnd((N,), dev, 32)
    r = rnd((M, N), dev, 33)
    wp = pack.pack_linear(w, "bf16x3")
    raw = ops.linear(x, wp, b, residual=r, tile_hint=tile)
    pre = ops.linear(make_split(x), wp, b, residual=r, tile_hint=tile)
    assert pre.dtype == torch.float32 and torch.equal(raw, pre), f"tile {tile}: {((raw - pre).norm() / raw.norm()).item():.3e}"
    assert ((raw - (x @ w.t() + b + r)).norm() / raw.norm()).item() < 3e-5
    # 3x3 conv: gather + zero padding + stride on split rows
    F, H, W, Ci, Co = 3, 12, 9, 64, 96
    xc = rnd((F * H * W, Ci), dev, 34)
    wc, bc = pack.pack_conv2d(rnd((Co, Ci, 3, 3), dev, 35, 0.05), "bf16x3"), rnd((Co,), dev, 36)
    for stride, ups in ((1, 1), (2, 1), (1, 2)):
        a = ops.conv2d(xc, wc, bc, F=F, Hin=H, Win=W, KH=3, KW=3, stride=stride, pad=1, ups=ups, tile_hint=tile)[0]
        c = ops.conv2d(make_split(xc), wc, bc, F=F, Hin=H, Win=W, KH=3, KW=3, stride=stride, pad=1, ups=ups, tile_hint=tile)[0]
        assert torch.equal(a, c), f"conv tile {tile} stride {stride} ups {ups}"
    # temporal conv
    Bt, T, HW, Cc = 2, 7, 45, 128
    xt = rnd((Bt * T * HW, Cc), dev, 37)
    wt, bt = pack.pack_conv3d_t(rnd((Cc, Cc, 3, 1, 1), dev, 38, 0.05), "bf16x3"), rnd((Cc,), dev, 39)
    assert torch.equal(ops.conv_temporal(xt, wt, bt, B=Bt, T=T, HW=HW, tile_hint=tile),
                       ops.conv_temporal(make_split(xt), wt, bt, B=Bt, T=T, HW=HW, tile_hint=tile))


def test_operand_swapped_projection_and_geglu_split_output(dev):
    from geo4d_amd import ops, pack
    F, N, C = 3, 200, 320
    x = rnd((F * N, C), dev, 40)
    wv = pack.pack_linear(rnd((C, C), dev, 41, 0.05), "bf16x3")
    a, npa = ops.linear_t_batched(wv, x, F, N)
    b, npb = ops.linear_t_batched(wv, make_split(x), F, N)
    assert npa == npb and torch.equal(a, b)
    # GEGLU epilogue writing the next GEMM's operand (every GEGLU-capable tile, first-generation hints are re-routed)
    M, K, inner = 700, 256, 320
    xg = rnd((M, K), dev, 42)
    w, bb = rnd((2 * inner, K), dev, 43, 0.1), rnd((2 * inner,), dev, 44)
    wp, bp = pack.pack_geglu(w, bb, "bf16x3")
    for tile in (0, 1, 11, 13, 22, 25, 27, 71, 74):
        plain = ops.linear(xg, wp, bp, act=2, tile_hint=tile)
        split = ops.linear(make_split(xg), wp, bp, act=2, tile_hint=tile, split_out=True)
        hi, lo = decode_split(split)
        h = xg @ w.t() + bb
        ref = h[:, :inner] * TF.gelu(h[:, inner:])
        assert ((hi + lo - ref).norm() / ref.norm()).item() < 3e-5, tile
        assert ((hi + lo - plain).norm() / plain.norm()).item() < 2e-5, tile
    w2 = pack.pack_linear(rnd((200, inner), dev, 45, 0.1), "bf16x3")
    out = ops.linear(split, w2, None, residual=None)
    assert ((out - ref @ rnd((200, inner), dev, 45, 0.1).t()).norm() / out.norm()).item() < 3e-5


def test_presplit_network_equals_raw_network(dev):
    """The tiny U-Net + VAE decode in bf16x3 with and without pre-split producers: same values up to the split's own rounding
    (the GEMMs see identical hi / lo pairs either way, so the only differences are fp32 re-association from tile choices)."""
    import os
    from geo4d_amd import unet as U
    from geo4d_amd.unet import UNetModel
    from oracle.params import seeded_state_dict
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = torch.load(os.path.join(ROOT, "tests", "golden", "unet_tiny.pt"), weights_only=False)
    net = UNetModel(**dict(g["unet_config"], compute_dtype="bf16x3"))
    net.load_state_dict(seeded_state_dict(g["shapes"]))
    net = net.to(dev)
    c = g["cases"]["t16_8x8"]
    outs = {}
    for flag in (True, False):
        U.PRESPLIT = flag
        try:
            outs[flag] = net(torch.cat([c["x"], c["c_concat"]], 1).to(dev), c["t"].to(dev), context=c["context"].to(dev), fs=c["fs"].to(dev)).clone()
        finally:
            U.PRESPLIT = True
    e = ((outs[True] - outs[False]).norm() / outs[False].norm()).item()
    ref = ((outs[True].cpu() - c["out"]).norm() / c["out"].norm()).item()
    print(f"[presplit vs raw tiny U-Net] rel_l2 = {e:.3e}; vs reference {ref:.3e}")
    assert e < 5e-5 and ref < 2e-4     # two bf16x3 evaluations with different tiles: each is ~2.5e-5 from the reference


@pytest.mark.parametrize("tile", [0, 1, 3, 4, 11, 13, 16, 22, 23, 25, 27, 28, 71, 72, 73, 74])
def test_plain_epilogue_split_output(dev, tile):
    """Round 4: o_split on the plain epilogue (bias / residual / SiLU), every second- and third-generation tile (first-generation hints
    are re-routed): the split output is exactly the split of the plain f32 output of the same launch configuration."""
    from geo4d_amd import ops, pack
    M, K, N = 1000, 512, 456          # ragged in M and N for every tile; 16 K slabs (even, >= 4: the phased tiles take it)
    x, w, b = rnd((M, K), dev, 50), rnd((N, K), dev, 51, 0.05), rnd((N,), dev, 52)
    r = rnd((M, N), dev, 53)
    wp = pack.pack_linear(w, "bf16x3")
    xs = make_split(x)
    for kw in (dict(), dict(residual=r), dict(act=1)):
        plain = ops.linear(xs, wp, b, tile_hint=tile, **kw)
        split = ops.linear(xs, wp, b, tile_hint=tile, split_out=True, **kw)
        assert isinstance(split, ops.SplitAct) and split.shape == (M, 2 * N)
        hi, lo = decode_split(split)
        if tile < 20:      # tile 0: table / tuner pick per launch signature; first-generation hints: the split output is re-routed to the
            # second-generation twin (16x16x32 MFMA instead of 32x32x16): the two launches associate the fp32 sums differently
            assert ((hi + lo) - plain).abs().max().item() < 2e-5 * plain.abs().max().item(), f"tile {tile} {list(kw)}"
            continue
        want_hi = plain.to(torch.bfloat16).float()
        assert torch.equal(hi, want_hi), f"tile {tile} {list(kw)}: hi"
        assert torch.equal(lo, (plain - want_hi).to(torch.bfloat16).float()), f"tile {tile} {list(kw)}: lo"
    with pytest.raises(RuntimeError):                      # the reduce kernel writes plain f32: no split-K with o_split
        ops.linear(xs, wp, b, tile_hint=25, split_k=2, split_out=True)


def test_spatial_attention_on_presplit_qkv_equals_raw(dev):
    """q | k (one GEMM, N = 2C) and V^T (operand-swapped, batched per frame) written pre-split by their projections; the flash kernel on
    those (qkv_split) == the flash kernel on the raw f32 projections, bit for bit; ragged last key tile (N = 200 = 3 x 64 + 8)."""
    from geo4d_amd import ops, pack
    for F, N, H in ((3, 200, 5), (2, 2560, 5), (2, 40, 20)):
        C = 64 * H
        n1 = rnd((F * N, C), dev, 60)
        wqk = pack.pack_linear(rnd((2 * C, C), dev, 61, 0.05), "bf16x3")
        wv = pack.pack_linear(rnd((C, C), dev, 62, 0.05), "bf16x3")
        xs = make_split(n1)
        qk = ops.linear(xs, wqk)
        vt, npad = ops.linear_t_batched(wv, xs, F, N)
        raw = ops.attention(qk[:, :C], [(qk[:, C:], vt.reshape(-1, npad), N, 1, C * npad)], B=F, H=H, Nq=N, scale=0.125, x3=True)
        qks = ops.linear(xs, wqk, split_out=True)
        vts, npad2 = ops.linear_t_batched(wv, xs, F, N, split_out=True)
        assert npad2 % 8 == 0 and vts.shape == (F, C, 2 * npad2)
        h, l = decode_split(vts.reshape(F * C, 2 * npad2))
        vref = vt.reshape(F * C, npad)[:, :N]         # (the two launches are tuned separately: same values up to the fp32 summation order)
        assert ((h + l)[:, :N] - vref).abs().max().item() < 2e-5 * vref.abs().max().item()
        # (q | k / V^T of the raw and the pre-split path come from separately tuned launches: same values up to the fp32 summation order,
        # so the kernels are compared bit for bit on the SAME pre-split inputs and to 1e-5 across the two paths)
        akw = dict(B=F, H=H, Nq=N, scale=0.125, x3=True, qkv_split=True)
        akv = [(qks[:, 2 * C:], vts.reshape(-1, 2 * npad2), N, 1, C * 2 * npad2)]
        pre1 = ops.attention(qks[:, :2 * C], akv, variant=1, **akw)
        assert ((pre1 - raw).norm() / raw.norm()).item() < 1e-5, f"F{F} N{N}: {((pre1 - raw).norm() / raw.norm()).item():.3e}"
        check_split(f"attention qkv_split F{F} N{N}", ops.attention(qks[:, :2 * C], akv, variant=1, split_out=True, **akw), pre1)
        for variant in (0, 4, 5):   # flash_attn2_kernel (skewed query blocks; one / two waves per SIMD; 0 = the library default): same sums in the same order
            pre = ops.attention(qks[:, :2 * C], akv, variant=variant, **akw)
            e = ((pre - pre1).norm() / pre1.norm()).item()
            print(f"[attention qkv_split variant {variant} F{F} N{N}] vs the first-generation kernel: rel_l2 = {e:.2e}, equal = {torch.equal(pre, pre1)}")
            assert e < 2e-6, (variant, F, N, e)
            check_split(f"attention qkv_split variant {variant} F{F} N{N}", ops.attention(qks[:, :2 * C], akv, variant=variant, split_out=True, **akw), pre)
        q, k = qk[:, :C].reshape(F, N, H, 64).permute(0, 2, 1, 3), qk[:, C:].reshape(F, N, H, 64).permute(0, 2, 1, 3)
        v = vt.reshape(F, H, 64, npad)[..., :N].permute(0, 1, 3, 2)
        ref = TF.scaled_dot_product_attention(q.double(), k.double(), v.double()).permute(0, 2, 1, 3).reshape(F * N, C).float()
        assert ((raw - ref).norm() / ref.norm()).item() < 3e-5


def test_presplit_pass_in_front_of_an_upsample_conv_is_bit_identical(dev):
    """Round 6: ops.presplit (geo4d_split_rows_bf16) turns an f32 stream into the pre-split operand ONCE; the nearest-2x Upsample convolution
    that consumes it (second generation: the gather handles the upsampling) then skips its in-register split - the same arithmetic, so the
    same bits as the raw-activation launch of the same tile; pitched input rows."""
    from geo4d_amd import ops, pack
    from test_gemm_v2_gpu import rnd
    F, H, W, C = 3, 10, 16, 256
    wide = rnd((F * H * W, 2 * C), dev, 7)
    x = wide[:, C:]                                          # row pitch 2C
    w, b = pack.pack_conv2d(rnd((192, C, 3, 3), dev, 8, 0.03), "bf16x3"), rnd((192,), dev, 9)
    xs = ops.presplit(x)
    assert isinstance(xs, ops.SplitAct) and xs.dtype == torch.bfloat16 and tuple(xs.shape) == (F * H * W, 2 * C)
    assert torch.equal(xs.as_subclass(torch.Tensor), pack.split_bf16(x.contiguous()))
    for tile in (22, 23, 25):
        raw = ops.conv2d(x, w, b, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, ups=2, tile_hint=tile, split_k=1)[0]
        pre = ops.conv2d(xs, w, b, F=F, Hin=H, Win=W, KH=3, KW=3, pad=1, ups=2, tile_hint=tile, split_k=1)[0]
        assert raw.shape == (F * 4 * H * W, 192) and torch.equal(pre, raw), f"tile {tile}"
