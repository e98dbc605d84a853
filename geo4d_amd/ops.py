"""Torch-tensor front-ends of the C ABI (include/geo4d_hip.h).

PyTorch is plumbing only: it owns device memory (``torch.empty``) and the HIP stream. Every function takes 2-D
"token" views ``[rows, channels]`` (unit inner stride, arbitrary row pitch), hands raw device pointers to
libgeo4d_hip.so and returns immediately (stream ordered, hipGraph-capturable). There is no CPU/eager fallback:
tensors that are not on a HIP device raise.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import BF16, BF16X3, F16, F16X2, F32, Attention, ConvGemm, GroupNorm
from .precision import resolve as _resolve_precision

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
_TD = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}


def dt_code(dtype):
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"geo4d_amd: unsupported dtype {dtype} (float32, bfloat16, float16 only)")


def k_align(dtype):
    """K / Cin granularity of conv_gemm in elements: one 128-byte LDS slab (4-byte elements for f32 and bf16x3)."""
    return 32 if _resolve_precision(dtype).storage == torch.float32 else 64


class X2Weight(torch.Tensor):
    """A pack.split_f16 weight: float16 [N, 2K] = f16 hi | lo per 8 K-elements of s * w, carrying alpha = 1 / s of its launch. A Tensor
    subclass so that the scale travels with the data: .to() / .clone() / .detach() / .contiguous() and same-shape views keep it (a
    plain attribute on a plain tensor was lost by all of them); slicing or reshaping gives a plain float16 tensor, which
    conv_gemm refuses as a two-pass weight."""
    _KEEP = {"to", "clone", "detach", "contiguous", "cuda", "cpu", "pin_memory"}

    @staticmethod
    def wrap(t, alpha):
        r = t.as_subclass(X2Weight)
        r._x2_alpha = float(alpha)
        return r

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        out = super().__torch_function__(func, types, args, kwargs or {})
        if isinstance(out, X2Weight):
            src = next((a for a in args if isinstance(a, X2Weight) and hasattr(a, "_x2_alpha")), None)
            if src is not None and getattr(func, "__name__", "") in cls._KEEP and out.shape == src.shape and out.dtype == torch.float16:
                out._x2_alpha = src._x2_alpha
            elif not hasattr(out, "_x2_alpha"):
                out = out.as_subclass(torch.Tensor)        # anything else is just a float16 matrix
        return out


def is_x2_weight(w):
    """A pack.split_f16 weight (X2Weight: float16 [N, 2K] = f16 hi | lo per 8 K-elements, carrying the 1 / scale of its launch)."""
    return isinstance(w, X2Weight) and w.dtype == torch.float16 and hasattr(w, "_x2_alpha")


def is_split(w, other):
    """True when `w` is a PRE-SPLIT bf16x3 operand (pack.split_bf16: bf16 [rows, 2K]) multiplied with a raw f32 operand."""
    return w.dtype == torch.bfloat16 and other.dtype == torch.float32


class SplitAct(torch.Tensor):
    """(The two-pass f16 consumers take PLAIN float16 rows since round 6 - an ordinary tensor, no wrapper: new_split(fmt="f16").)
    A bf16 [rows, 2K] activation in the pre-split bf16x3 operand format (per 8 K-elements: 8 x bf16 hi | 8 x bf16 lo), written by
    the producers that feed GEMMs (GroupNorm / LayerNorm / attention / GEGLU epilogue with split_out) so that conv_gemm does not
    split the fragments again in its K loop. A Tensor subclass only to make the format visible to conv_gemm; every other entry point
    of this module rejects it (`_dev`), and producers require a whole contiguous matrix as their split output (`_split_out_ok`).
    Plain torch ops on it (.float(), slicing at non-8-element offsets) still see a bf16 matrix: do not use them."""
    @staticmethod
    def wrap(t):
        return t.as_subclass(SplitAct)


def new_split(rows, k, device, fmt="bf16"):
    """The A operand a producer writes for the GEMM that follows. `fmt`: "bf16" = the pre-split bf16x3 operand format (SplitAct, bf16
    [rows, 2k]); "f16" = the two-pass f16 consumers' operand (dtype 4): PLAIN float16 rows [rows, k] - the activation is multiplied as
    one f16, so since round 6 nothing else is stored (round 5 wrote an f16 lo half beside it that no kernel read)."""
    if fmt == "f16":
        return torch.empty((rows, k), device=device, dtype=torch.float16)
    return SplitAct.wrap(torch.empty((rows, 2 * k), device=device, dtype=torch.bfloat16))


def split_fmt(split_out):
    """groupnorm(split_out=...) argument -> (ABI code, format name): False / None -> 0, True / "bf16" -> 1, "f16" -> 2."""
    if not split_out:
        return 0, None
    if split_out is True or split_out == "bf16":
        return 1, "bf16"
    if split_out == "f16":
        return 2, "f16"
    raise ValueError(f"split_out={split_out!r}: False, True / 'bf16' or 'f16'")


def act_k(x):
    """Channels (K elements) of an activation matrix: a SplitAct stores 2 bf16 per element."""
    return x.shape[1] // 2 if isinstance(x, SplitAct) else x.shape[1]


def kdim(w, other):
    """Logical K extent of a 2-D operand (a pre-split operand stores 2 bf16 / f16 per K element)."""
    return w.shape[1] // 2 if (is_split(w, other) or isinstance(other, SplitAct) or is_x2_weight(w)) else w.shape[1]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, what, split_ok=False):
    """Every operand goes through here: it must live on a HIP device, and a SplitAct (bf16 [rows, 2K] hi | lo image) may only reach
    the operands that understand the format - conv_gemm's A / W / out and the producers' `out`; anywhere else it would be silently
    reinterpreted as a bf16 matrix."""
    if not t.is_cuda:
        raise _lib.Geo4DNativeError(f"geo4d_amd.ops: `{what}` lives on {t.device}; the HIP path needs a GPU tensor "
                                    "(there is no CPU fallback)")
    if isinstance(t, SplitAct) and not split_ok:
        raise TypeError(f"geo4d_amd.ops: `{what}` is a pre-split bf16x3 activation (SplitAct); only GEMM operands take that format")
    return t


def _split_out_ok(out, rows):
    """Producers write the pre-split format with row-start-relative addressing (store_split4): the buffer must be the whole
    [rows, 2K] matrix - contiguous, starting at the row, 16-byte aligned - not a column-offset view of a wider one."""
    assert out.dim() == 2 and out.shape[0] == rows and out.is_contiguous() and out.data_ptr() % 16 == 0, \
        "a SplitAct output must be a contiguous [rows, 2K] matrix (no column-offset views)"


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _ld(t):
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), f"need a [rows, cols] view with unit inner stride, got {t.shape} {t.stride()}"
    return t.stride(0)


_WS = {}
WORKSPACE_BYTES = 256 << 20
# (Round 6 built split-K WITHOUT the reduce launch - wave tickets, the last arrival of a (tile, wave) position sums the slabs and runs the
# epilogue, bit-equal to the reduce kernel - and measured it slower twice: with agent-scope release / acquire fences 6.37 -> 5.57 frames/s,
# with sc1 write-through slabs and no fence 6.31 -> 6.08. Removed; profiles/r06_splitk_fused.md.)

# ---- per-shape launch tuning ------------------------------------------------------------------------------------
# geo4d_conv_gemm has three kernel generations x tile shapes x split-K factors (include/geo4d_hip.h tile_hint); the C-side heuristic is a fallback. The host keeps a table
# problem-signature -> (tile_hint, split_k): loaded from geo4d_amd/tuning/gfx950.json (measured on MI355X by
# tools/tune_gemm.py) and, for shapes not in it, filled by timing the candidates on first eager use (never while a
# hipGraph is being captured). Every candidate computes the same sums in the same k order per output element
# (split-K only regroups them), so tuning never changes a GEMM's result beyond fp32 re-association. Since round 4 (GN_FUSED_STATS = 1)
# the tile choice ALSO sets the row granularity of the GroupNorm statistics its epilogue emits (32 ... 128 rows per entry), so the
# consumer GroupNorm - and with it the U-Net output - differs at fp32 round-off between tile choices: bit-reproducibility holds PER
# TUNING TABLE (the committed gfx950.json + whatever this process measured for shapes missing from it), not per candidate. Processes
# that must agree bit for bit (ranks comparing outputs, a captured graph against a later eager run of a shape first met during capture)
# need the same table: ship one (save_tuning) or set GEO4D_AUTOTUNE=0 / GEO4D_GN_FUSED=0.
import json as _json
import os as _os

_TUNE_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tuning", "gfx950.json")
_TUNE = None
_CANDIDATES = [(1, 1), (2, 1), (3, 1), (4, 1), (11, 1), (13, 1), (16, 1), (17, 1), (17, 2), (16, 2), (17, 8), (1, 2), (1, 4), (1, 8), (11, 2), (11, 8), (13, 2), (13, 4), (3, 2), (4, 2), (4, 4), (3, 4), (2, 4), (5, 1),
               # second generation (gemm_kernel_v2.h: 16x16x32 MFMA, register epilogue, persistent workgroups; bf16 / bf16x3, no NCTHW outputs);
               # round 4 kept the five tiles the measured table selects
               (22, 1), (23, 1), (25, 1), (27, 1), (28, 1), (22, 2), (22, 4), (23, 2), (25, 2), (25, 4), (25, 8), (27, 2), (27, 4), (28, 2), (28, 4), (28, 8),
               # third generation (gemm_kernel_v3.h: phased K loop, counted DMA waits; 8-wave tiles, one workgroup per CU).
               # Splits that do not divide the K slabs evenly run on the second-generation twin inside the library.
               (71, 1), (72, 1), (73, 1), (74, 1), (71, 2), (72, 2), (73, 2), (74, 2), (71, 3), (72, 3), (73, 3), (74, 3), (71, 4), (73, 4), (74, 4),
               (71, 5), (73, 5), (74, 5), (73, 6), (74, 6), (73, 8), (74, 8), (73, 10), (74, 10)]
_VALID_HINTS = {t for t, _ in _CANDIDATES} | {0}
AUTOTUNE = _os.environ.get("GEO4D_AUTOTUNE", "1") != "0"
# GEMM epilogues can emit the next GroupNorm's column sums (gn_stats=True call sites), which removes that GroupNorm's statistics pass
# over the tensor. 1 (default since round 4): where the second / third generation's fast epilogue does it per wave-tile row range at
# ~0.3 us per tile (f32 rows, i.e. the bf16x3 mode: same-box A/B +1.2 % frames/s, decode -2.5 %, profiles/r04_gn_fused_stats.md);
# 2: also on the first-generation tiles (round 2: measured slower there: -3.5 % bf16x3, -7 % bf16); 0: off.
def _env_level(name, default):
    """0 / 1 / 2 ... or the usual words: off / false / no -> 0, on / true / yes (and anything else that is not a number) -> 1."""
    v = _os.environ.get(name)
    if v is None or not v.strip():
        return default
    v = v.strip().lower()
    try:
        return int(v)
    except ValueError:
        return 0 if v in ("off", "false", "no", "none") else 1


GN_FUSED_STATS = _env_level("GEO4D_GN_FUSED", 1)
SPLITK_COLSUM = _env_level("GEO4D_SPLITK_COLSUM", 1)
TUNE_EXACT = _env_level("GEO4D_TUNE_EXACT", 0)     # tools/tune_gemm.py: measure a pre-split launch under its OWN key instead of borrowing the raw-activation entry of the same shape       # 0: split-K launches leave the GroupNorm statistics to the GroupNorm (the state before round 6)
TUNE_LOG = []          # (key, chosen (tile, split), ms per launch, finalists) of every shape autotuned in this process (tools/tune_gemm.py prints it)
DEBUG_ABLATE = _env_level("GEO4D_DEBUG_ABLATE", 0)       # tests / A-B runs: 2 = three persistent workgroups; 16 + g = tile order with GROUP_M = g (17 = column-fastest)
GEMM_TIMELINE = None   # set to a list to have conv_gemm bracket every launch with HIP events: (flops, start, end, MFMA passes per product)
ATTN_TIMELINE = None   # the same for the spatial self-attention launches (one key/value set) of ops.attention
SAT_COUNTER = None     # debug: an int64 [1] device tensor -> every f16-clamping store (GroupNorm / LayerNorm / GEGLU epilogue writing the two-pass
                       # GEMM's f16 operand) adds the lanes it clamped (|x| > 65504); tests assert 0 at full size. None in production.


def _tune_table():
    global _TUNE
    if _TUNE is None:
        _TUNE = {}
        if _os.path.exists(_TUNE_PATH):
            with open(_TUNE_PATH) as f:
                _TUNE = {k: tuple(v) for k, v in _json.load(f).items() if v[0] in _VALID_HINTS}     # (entries naming a retired tile hint are re-tuned)
    return _TUNE


def save_tuning(path=None):
    with open(path or _TUNE_PATH, "w") as f:
        _json.dump({k: list(v) for k, v in sorted(_tune_table().items())}, f, indent=0)


def _time_launches(launch, tile, split, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        launch(tile, split)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def _autotune(launch, key):
    """Two stages (round 4: a single 3-launch sample per candidate mis-ranked candidates that differ by 30-40 % when re-measured in
    isolation, profiles/r04_gemm_retune.md): (1) every candidate, 4 launches after a warm-up; (2) the 6 fastest re-measured in three
    interleaved rounds of 12 launches, ranked by their best round (the chip's clock moves under load: the minimum is the least noisy
    estimate of what a launch costs inside a replayed graph)."""
    stage1 = []
    for tile, split in _CANDIDATES:
        try:
            launch(tile, split)
        except RuntimeError:
            continue
        stage1.append((_time_launches(launch, tile, split, 4), tile, split))
    if not stage1:
        _tune_table()[key] = (0, 0)
        return (0, 0)
    stage1.sort()
    finalists = stage1[:6]
    best_of = {(t, s): float("inf") for _, t, s in finalists}
    for _ in range(3):
        for _, tile, split in finalists:
            best_of[(tile, split)] = min(best_of[(tile, split)], _time_launches(launch, tile, split, 12))
    best = min(best_of, key=best_of.get)
    _tune_table()[key] = best
    TUNE_LOG.append((key, best, best_of[best], [(t, s, round(v * 1e3, 1)) for (t, s), v in sorted(best_of.items(), key=lambda kv: kv[1])]))
    return best


def workspace(device):
    """One static fp32 scratch per device for split-K slabs (stream-ordered reuse; allocated outside graph capture)."""
    ws = _WS.get(device)
    if ws is None:
        ws = _WS[device] = (torch.empty(WORKSPACE_BYTES, device=device, dtype=torch.uint8),
                            torch.zeros(256, device=device, dtype=torch.uint8))
    return ws


def conv_gemm(a, w, out, *, M, N, K, Cin, lda, ldw, ldo, T=1, Hin=1, Win=1, Hout=1, Wout=1, KT=1, KH=1, KW=1,
              pt=0, ph=0, pw=0, stride=1, ups=1, bias=None, bias_per_row=False, rowbias=None, rowbias_div=0,
              residual=None, ldr=0, act=0, out_nchw=False, alpha=1.0, batch=1, a_bs=0, w_bs=0, o_bs=0, r_bs=0,
              tile_hint=0, split_k=0, x3=False, gn_stats=False):
    """`lda` / `ldw` / `a_bs` / `w_bs` are strides of the tensors as passed (torch elements). bf16x3 mode is selected by the
    operands: f32 activations against a pre-split bf16 weight (either side), or two f32 operands with `x3=True`."""
    lib = _lib.load()
    _dev(a, "A", True); _dev(w, "W", True); _dev(out, "out", True)
    if getattr(out, "_gn_colsum", None) is not None:       # sums of an EARLIER launch into this tensor object: this launch either sets fresh ones or leaves none
        out._gn_colsum = None
        out._gn_colsum_rows = 32
        out._gn_colsum_tag = None
    for opt, nm in ((bias, "bias"), (rowbias, "rowbias"), (residual, "residual")):
        if opt is not None:
            _dev(opt, nm)
    if isinstance(out, SplitAct) and batch == 1:
        _split_out_ok(out, M)
    a_split, w_split = is_split(a, w), is_split(w, a)
    assert not is_x2_weight(a), "a pack.split_f16 weight is the W operand"
    if isinstance(w, X2Weight):
        # two-pass f16 (dtype 4): plain f16 activation rows (GroupNorm / LayerNorm / GEGLU-epilogue output) x a pack.split_f16 weight (which carries its 1 / scale)
        assert is_x2_weight(w), "a two-pass weight without its scale"
        assert a.dtype == torch.float16 and not isinstance(a, SplitAct), "a pack.split_f16 weight multiplies plain float16 activation rows (the two-pass form has no raw-f32 launch)"
        assert out.dtype in (torch.float16, torch.float32) and not isinstance(out, SplitAct) and not out_nchw, "the two-pass f16 GEMM writes plain f32 rows or plain f16 rows"
        assert ldw % 2 == 0 and w_bs % 2 == 0
        a_split, w_split = 2, True
        code = F16X2
        alpha = alpha * w._x2_alpha
        ldw, w_bs = ldw // 2, w_bs // 2
    elif isinstance(a, SplitAct) or isinstance(w, SplitAct):     # pre-split activations x pre-split weights (bf16 storage, 4 bytes per K element)
        assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and lda % 2 == 0 and ldw % 2 == 0 and a_bs % 2 == 0 and w_bs % 2 == 0
        a_split = w_split = True
        code = BF16X3
        lda, a_bs, ldw, w_bs = lda // 2, a_bs // 2, ldw // 2, w_bs // 2
    elif a_split or w_split:
        code = BF16X3
        if a_split:
            assert lda % 2 == 0 and a_bs % 2 == 0
            lda, a_bs = lda // 2, a_bs // 2          # -> K elements (4 bytes each)
        else:
            assert ldw % 2 == 0 and w_bs % 2 == 0
            ldw, w_bs = ldw // 2, w_bs // 2
    else:
        assert a.dtype == w.dtype, (a.dtype, w.dtype)
        code = dt_code(a.dtype)
        if x3:
            assert a.dtype == torch.float32, "x3=True multiplies two f32 operands with the bf16x3 scheme"
            code = BF16X3
    p = ConvGemm()
    p.A, p.W, p.O = a.data_ptr(), w.data_ptr(), out.data_ptr()
    p.bias, p.rowbias, p.R = _ptr(bias), _ptr(rowbias), _ptr(residual)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous()
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rowbias.dim() == 2 and rowbias.stride(1) == 1
        p.ldrb = rowbias.stride(0)
    if residual is not None:
        assert residual.dtype == (torch.float32 if isinstance(out, SplitAct) else out.dtype)      # (a SplitAct is f32 values in bf16 storage)
    p.lda, p.ldw, p.ldo, p.ldr = lda, ldw, ldo, ldr
    p.a_bs, p.w_bs, p.o_bs, p.r_bs = a_bs, w_bs, o_bs, r_bs
    p.M, p.N, p.K, p.batch, p.Cin = M, N, K, batch, Cin
    p.T, p.Hin, p.Win, p.Hout, p.Wout = T, Hin, Win, Hout, Wout
    p.KT, p.KH, p.KW, p.pt, p.ph, p.pw, p.stride, p.ups = KT, KH, KW, pt, ph, pw, stride, ups
    p.rowbias_div, p.bias_per_row, p.act = rowbias_div, int(bias_per_row), act
    p.dtype, p.out_dtype, p.out_nchw, p.tile_hint = code, (F32 if code == F16X2 else dt_code(out.dtype)), int(out_nchw), tile_hint
    p.alpha, p.split_k = alpha, split_k
    p.debug_ablate = DEBUG_ABLATE
    p.a_split, p.w_split = int(a_split), int(w_split)
    p.o_split = 1 if isinstance(out, SplitAct) else 2 if (code == F16X2 and out.dtype == torch.float16) else 0
    if p.o_split == 1:
        assert code == BF16X3, "pre-split (bf16 hi | lo) outputs come from bf16x3 launches"
        assert ldo % 2 == 0 and o_bs % 2 == 0
        p.ldo, p.o_bs, p.out_dtype = ldo // 2, o_bs // 2, F32
    elif p.o_split == 2:            # plain f16 rows (clamped): ldo / o_bs stay in f16 elements
        assert residual is None and rowbias is None and not bias_per_row and act in (0, 2), "the f16-row epilogue of the two-pass GEMM: column bias (+ GEGLU) only"
    p.gn_colsum = 0
    p.sat_count = _ptr(SAT_COUNTER) if p.o_split == 2 else 0
    ws, zeros = workspace(a.device)
    p.workspace, p.workspace_bytes, p.zeros = ws.data_ptr(), ws.numel(), zeros.data_ptr()

    def launch(tile, split):
        p.tile_hint, p.split_k = tile, split
        _lib.check(lib.geo4d_conv_gemm(C.byref(p), _stream()), "geo4d_conv_gemm")

    if tile_hint == 0 and split_k == 0:
        key = f"{p.dtype}/{p.out_dtype}|{M}x{N}x{K}|c{Cin}|t{KT}{KH}{KW}s{stride}u{ups}|a{act}r{int(residual is not None)}n{int(out_nchw)}|b{batch}"
        if code in (BF16X3, F16X2):
            key += f"|x{int(bool(a_split))}{int(w_split)}" + ("o" if p.o_split else "")       # (dtype 4's "o" = its f16 rows)
        cfg = _tune_table().get(key)
        if cfg is None and code == BF16X3 and a_split and w_split and not (TUNE_EXACT and AUTOTUNE and not torch.cuda.is_current_stream_capturing()):
            # pre-split activations: same tile geometry as the raw-activation launch of the same shape (table measured on those)
            base = key.split("|x")[0]
            cfg = (_tune_table().get(base + "|x11") if p.o_split else None) or _tune_table().get(base + "|x01") or _tune_table().get(base + "|x10")
            if cfg is not None and p.o_split and cfg[1] > 1:
                # the pre-split output has no split-K form (the reduce kernel writes plain f32): measure this launch on its own when
                # that is allowed, else keep the tile and drop the split
                cfg = None if (AUTOTUNE and not torch.cuda.is_current_stream_capturing()) else (cfg[0], 1)
        if cfg is None and code == F16X2 and not (AUTOTUNE and not torch.cuda.is_current_stream_capturing()):
            # no entry and no measuring now: the bf16x3 launch of the same shape is the same tile geometry and the same bytes
            cfg = _tune_table().get("3" + key[1:])
            if cfg is not None and cfg[0] < 22:
                cfg = None                   # (the two-pass kernels exist on the second / third generation only: library default)
            elif cfg is not None and p.o_split and cfg[1] > 1:
                cfg = (cfg[0], 1)            # (a pre-split output has no split-K form)
        if cfg is None and AUTOTUNE and not torch.cuda.is_current_stream_capturing():
            cfg = _autotune(launch, key)
        tile_hint, split_k = cfg if cfg is not None else (0, 0)
        if p.o_split == 2:
            # the f16-row epilogue has no split-K; its GEGLU form lives on the tiles whose wave tiles are a multiple of 64 columns wide
            # (25, 27, 71, 74): a tile borrowed from the bf16x3 entry of the same shape goes to its nearest such neighbour
            split_k = min(split_k, 1)
            if act == 2:
                tile_hint = {22: 25, 23: 25, 28: 27, 72: 71, 73: 74}.get(tile_hint, tile_hint)
    if gn_stats and GN_FUSED_STATS and not p.o_split and batch == 1 and out.dim() == 2 and out.shape[0] == M:
        # the consumer GroupNorm's statistics pass, for free: per row block and column (sum, sum of squares) from the epilogue. The library
        # says how many rows one entry of THIS launch covers (32 for the first generation, the wave tile's rows for the second / third; 0 =
        # this configuration cannot emit them - split-K, activations, unaligned rows: the GroupNorm then runs its own pass)
        p.tile_hint, p.split_k = tile_hint, split_k
        rows = lib.geo4d_conv_gemm_colsum_rows(C.byref(p)) if (tile_hint >= 21 or int(GN_FUSED_STATS) >= 2) else 0
        if split_k > 1 and not SPLITK_COLSUM:      # (A/B switch: round 6 lets a split-K launch's reduce kernel emit the sums)
            rows = 0
        if rows == 0 and int(GN_FUSED_STATS) >= 2:
            # mode 2 (A/B, tests): a tuned configuration that cannot emit the sums (split-K; a second / third generation tile on 16-bit rows)
            # gives way to an un-split first-generation launch, as this path did before round 4
            p.tile_hint, p.split_k = (tile_hint if tile_hint < 21 else 0), 1
            rows = lib.geo4d_conv_gemm_colsum_rows(C.byref(p))
            if rows > 0:
                tile_hint, split_k = p.tile_hint, 1
        if rows > 0:
            split_k = split_k or 1           # (the library's own tile choice may not split a launch that emits the sums)
            cs = torch.empty((M // rows, N, 2), device=out.device, dtype=torch.float32)
            p.gn_colsum = cs.data_ptr()
            out._gn_colsum = cs
            out._gn_colsum_rows = rows
            out._gn_colsum_tag = (out.data_ptr(), out._version)      # groupnorm() ignores the sums if the buffer was rewritten by torch since
    if GEMM_TIMELINE is not None:     # bench.py's per-launch HIP-event timeline of the dominant kernel (never on while capturing)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch(tile_hint, split_k)
        e1.record()
        GEMM_TIMELINE.append((2.0 * M * N * K * batch, e0, e1, {BF16X3: 3, F16X2: 2}.get(code, 1)))    # (flops, events, MFMAs issued per product)
        return out
    launch(tile_hint, split_k)
    return out


def _out_dtype(x, out_dtype, w=None):
    """Default output dtype of a GEMM: f32 rows for the 4-byte modes' operand formats (a SplitAct, or f16 rows against a two-pass weight)."""
    return out_dtype or (torch.float32 if (isinstance(x, SplitAct) or (w is not None and is_x2_weight(w))) else x.dtype)


def linear(x, w, bias=None, *, residual=None, act=0, out=None, out_dtype=None, alpha=1.0, tile_hint=0, split_k=0, gn_stats=False,
           split_out=False):
    """x [M, K] (row pitch free), w packed [N, K]; GEGLU (act=2) returns [M, N/2]. `split_out` (bf16x3): the result is written as a
    SplitAct (the next GEMM's pre-split A operand)."""
    M, K = x.shape[0], act_k(x)
    N = w.shape[0]
    assert kdim(w, x) == K, (w.shape, x.shape)
    if out is None:
        nout = N // 2 if act == 2 else N
        out = new_split(M, nout, x.device, split_fmt(split_out)[1]) if split_out else torch.empty((M, nout), device=x.device, dtype=_out_dtype(x, out_dtype, w))
    return conv_gemm(x, w, out, M=M, N=N, K=K, Cin=K, lda=_ld(x), ldw=_ld(w), ldo=_ld(out), bias=bias,
                     residual=residual, ldr=_ld(residual) if residual is not None else 0, act=act, alpha=alpha,
                     tile_hint=tile_hint, split_k=split_k, gn_stats=gn_stats)


def conv2d(x, w, bias, *, F, Hin, Win, KH, KW, stride=1, pad=0, pad_end=0, ups=1, T=1, rowbias=None, rowbias_div=0,
           residual=None, act=0, out=None, out_dtype=None, out_nchw=False, nchw_channels=None, tile_hint=0, split_k=0, gn_stats=False):
    """x tokens [F*Hin*Win, Cin]; w packed [N, KH*KW*Cin]. Output tokens [F*Hout*Wout, N], or with out_nchw a
    [B, C, T, Hout, Wout] tensor (`out` may be a channel-offset view of a wider tensor with `nchw_channels` channels).
    `pad_end` = extra zero rows / columns at the bottom / right only (ae_modules.py:102-106 pads (0,1,0,1) before its
    stride-2 conv): the gather treats every out-of-image tap as zero, so only the output size changes."""
    Cin = act_k(x)
    N = w.shape[0]
    Hs, Ws = Hin * ups, Win * ups
    Hout = (Hs + 2 * pad + pad_end - KH) // stride + 1
    Wout = (Ws + 2 * pad + pad_end - KW) // stride + 1
    M = F * Hout * Wout
    assert x.shape[0] == F * Hin * Win, (x.shape, F, Hin, Win)
    assert kdim(w, x) == KH * KW * Cin, (w.shape, KH, KW, Cin)
    if out is None:
        if out_nchw:
            out = torch.empty((F // T, N, T, Hout, Wout), device=x.device, dtype=out_dtype or torch.float32)
        else:
            out = torch.empty((M, N), device=x.device, dtype=_out_dtype(x, out_dtype, w))
    conv_gemm(x, w, out, M=M, N=N, K=KH * KW * Cin, Cin=Cin, lda=_ld(x), ldw=_ld(w), ldo=(nchw_channels or N) if out_nchw else _ld(out), T=T,
              Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, KH=KH, KW=KW, ph=pad, pw=pad, stride=stride, ups=ups, bias=bias,
              rowbias=rowbias, rowbias_div=rowbias_div, residual=residual,
              ldr=_ld(residual) if residual is not None else 0, act=act, out_nchw=out_nchw, tile_hint=tile_hint,
              split_k=split_k, gn_stats=gn_stats)
    return out, Hout, Wout


def conv_temporal(x, w, bias, *, B, T, HW, residual=None, out=None, gn_stats=False, tile_hint=0, split_k=0):
    """nn.Conv3d kernel (3,1,1), padding (1,0,0) on tokens [(b t) hw, C]; w packed [N, 3*C]."""
    Cin = act_k(x)
    N = w.shape[0]
    M = B * T * HW
    assert x.shape[0] == M and kdim(w, x) == 3 * Cin
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=_out_dtype(x, None, w))
    return conv_gemm(x, w, out, M=M, N=N, K=3 * Cin, Cin=Cin, lda=_ld(x), ldw=_ld(w), ldo=_ld(out), T=T, Hin=HW, Win=1,
                     Hout=HW, Wout=1, KT=3, pt=1, bias=bias, residual=residual,
                     ldr=_ld(residual) if residual is not None else 0, gn_stats=gn_stats, tile_hint=tile_hint, split_k=split_k)


def batched_gemm(a, b, out, *, batch, M, N, K, a_bs, b_bs, o_bs, bias=None, bias_per_row=False, alpha=1.0, x3=False):
    """out[z] = alpha * a[z] @ b[z]^T (+bias); a [.., K] rows, b [.., K] rows (both K-major). `x3`: both operands are f32
    activations to be multiplied with the bf16x3 scheme (the VAE AttnBlock GEMMs of the bf16x3 mode)."""
    return conv_gemm(a, b, out, M=M, N=N, K=K, Cin=K, lda=_ld(a), ldw=_ld(b), ldo=_ld(out), batch=batch, a_bs=a_bs,
                     w_bs=b_bs, o_bs=o_bs, bias=bias, bias_per_row=bias_per_row, alpha=alpha, x3=x3)


_gn_ws = {}


def groupnorm(x, gamma, beta, *, F, HW, eps, groups=32, frames_per_stat=1, silu=False, out=None, split_out=False):
    """`split_out` (f32 input of the bf16x3 mode): y is returned as a SplitAct, the pre-split A operand of the conv that follows -
    True / "bf16": bf16 hi | lo (three-pass bf16x3 consumer); "f16": f16 hi | lo (two-pass f16 consumer, pack.split_f16 weights)."""
    lib = _lib.load()
    _dev(x, "x")
    assert not isinstance(x, SplitAct), "GroupNorm reads plain activations"
    Cc = x.shape[1]
    assert x.shape[0] == F * HW, (x.shape, F, HW)
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
    if out is None:
        out = new_split(F * HW, Cc, x.device, split_fmt(split_out)[1]) if split_out else torch.empty((F * HW, Cc), device=x.device, dtype=x.dtype)
    split_out = 1 if isinstance(out, SplitAct) else 2 if (out.dtype == torch.float16 and x.dtype == torch.float32) else 0
    if split_out:
        assert x.dtype == torch.float32, "the GEMM-operand producer formats are written from f32 activations"
    if split_out == 1:
        _split_out_ok(out, F * HW)
    need = lib.geo4d_groupnorm_workspace(F, HW, groups, frames_per_stat)
    ws = torch.empty(need, device=x.device, dtype=torch.uint8)
    p = GroupNorm()
    p.x, p.y, p.gamma, p.beta = x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    p.workspace, p.workspace_bytes = ws.data_ptr(), need
    p.ldx, p.ldy = _ld(x), (_ld(out) // 2 if split_out == 1 else _ld(out))
    p.split_out = int(split_out)
    p.sat_count = _ptr(SAT_COUNTER) if split_out == 2 else 0
    p.F, p.HW, p.C, p.groups, p.frames_per_stat = F, HW, Cc, groups, frames_per_stat
    p.act, p.dtype, p.eps = int(silu), dt_code(x.dtype), eps
    cs = getattr(x, "_gn_colsum", None)     # column sums left on this very tensor object by the GEMM that produced it
    if cs is not None and getattr(x, "_gn_colsum_tag", None) != (x.data_ptr(), x._version):
        cs = None                            # the tensor was modified in place (or re-viewed) after the GEMM wrote it: stale sums
    rows = getattr(x, "_gn_colsum_rows", 32)
    ok = cs is not None and (frames_per_stat * HW) % rows == 0 and tuple(cs.shape) == (F * HW // rows, Cc, 2)
    p.colsum, p.colsum_rows = (cs.data_ptr(), rows) if ok else (0, 0)
    _lib.check(lib.geo4d_groupnorm(C.byref(p), _stream()), "geo4d_groupnorm")
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None, split_out=False):
    lib = _lib.load()
    _dev(x, "x")
    assert not isinstance(x, SplitAct), "LayerNorm reads plain activations"
    M, Cc = x.shape
    if split_out or isinstance(out, SplitAct) or (out is not None and out.dtype == torch.float16 and x.dtype == torch.float32):
        assert x.dtype == torch.float32
        if out is None:
            out = new_split(M, Cc, x.device, split_fmt(split_out)[1])
        if isinstance(out, SplitAct):
            _split_out_ok(out, M)
            _lib.check(lib.geo4d_layernorm_split(x.data_ptr(), _ld(x), out.data_ptr(), _ld(out) // 2, M, Cc, eps, gamma.data_ptr(),
                                                 beta.data_ptr(), 1, 0, _stream()), "geo4d_layernorm_split")
        else:           # plain f16 rows: the two-pass GEMM's A operand
            assert out.dtype == torch.float16
            _lib.check(lib.geo4d_layernorm_split(x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), M, Cc, eps, gamma.data_ptr(),
                                                 beta.data_ptr(), 2, _ptr(SAT_COUNTER), _stream()), "geo4d_layernorm_split")
        return out
    if out is None:
        out = torch.empty((M, Cc), device=x.device, dtype=x.dtype)
    _lib.check(lib.geo4d_layernorm(x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), M, Cc, eps, gamma.data_ptr(),
                                   beta.data_ptr(), dt_code(x.dtype), _stream()), "geo4d_layernorm")
    return out


def softmax_rows(x, scale, out_dtype, out=None, causal_period=0):
    """softmax(scale * x) per row of fp32 scores; `out` may be a wider zero-initialised buffer (K padding of the next GEMM);
    `causal_period` > 0: row r only sees columns <= r % causal_period (text-transformer mask)."""
    lib = _lib.load()
    _dev(x, "x")
    assert x.dtype == torch.float32
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), device=x.device, dtype=out_dtype)
    if causal_period:
        _lib.check(lib.geo4d_softmax_rows_causal(x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), rows, cols, scale,
                                                 dt_code(out.dtype), causal_period, _stream()), "geo4d_softmax_rows_causal")
    else:
        _lib.check(lib.geo4d_softmax_rows(x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), rows, cols, scale,
                                          dt_code(out.dtype), _stream()), "geo4d_softmax_rows")
    return out


ATTN_VARIANT = int(_os.environ.get("GEO4D_ATTN_VARIANT", "0"))   # 0 = library default; 1..3 = A/B builds (include/geo4d_hip.h)


def attention(q, kv, *, B, H, Nq, scale, out=None, x3=False, variant=None, split_out=False, qkv_split=False):
    """q [B*Nq, >=H*64] view; kv = list of (k, vt, Nk, kv_div, vt_bs): k [(B/kv_div)*Nk, >=H*64] view, vt = V TRANSPOSED
    as a [>=H*64, ld] view (row = channel, column = key) whose batch b' starts vt_bs elements after batch b'-1.
    `qkv_split` (bf16x3, one key/value set): q, k and vt are SplitAct views (bf16, 2 per element: what the projections wrote with
    split_out=True), strides / vt_bs in bf16 elements as torch reports them."""
    lib = _lib.load()
    if qkv_split:
        assert len(kv) == 1 and all(isinstance(t, SplitAct) for t in (q, kv[0][0], kv[0][1])), "qkv_split takes SplitAct q / k / vt of one key/value set"
        return _attention_presplit(lib, q, kv[0], B=B, H=H, Nq=Nq, scale=scale, out=out, variant=variant, split_out=split_out)
    _dev(q, "q")
    if out is None:
        out = new_split(B * Nq, H * 64, q.device) if split_out else torch.empty((B * Nq, H * 64), device=q.device, dtype=q.dtype)
    split_out = isinstance(out, SplitAct)
    if split_out:
        _split_out_ok(out, B * Nq)
    p = Attention()
    p.q, p.o, p.ldq, p.ldo = q.data_ptr(), out.data_ptr(), _ld(q), (_ld(out) // 2 if split_out else _ld(out))
    p.split_out = int(split_out)
    assert 1 <= len(kv) <= 2
    for i, (k, vt, nk, div, vt_bs) in enumerate(kv):
        assert k.dtype == q.dtype and vt.dtype == q.dtype
        _dev(k, "k"); _dev(vt, "vt")
        p.k[i], p.vt[i], p.ldk[i], p.ldvt[i], p.vt_bs[i], p.Nk[i], p.kv_div[i] = k.data_ptr(), vt.data_ptr(), _ld(k), _ld(vt), vt_bs, nk, div
    p.zeros = workspace(q.device)[1].data_ptr()
    code = dt_code(q.dtype)
    if x3:
        assert q.dtype == torch.float32, "x3 attention runs on f32 q / k / v^T"
        code = BF16X3
    p.B, p.H, p.Nq, p.nseg, p.head_dim, p.dtype, p.scale = B, H, Nq, len(kv), 64, code, scale
    p.variant = ATTN_VARIANT if variant is None else variant
    _launch_attention(lib, p)
    return out


def _attention_presplit(lib, q, kv0, *, B, H, Nq, scale, out, variant, split_out):
    k, vt, nk, div, vt_bs = kv0
    for t, nm in ((q, "q"), (k, "k"), (vt, "vt")):
        _dev(t, nm, True)
        assert t.dtype == torch.bfloat16 and _ld(t) % 2 == 0
    if out is None:
        out = new_split(B * Nq, H * 64, q.device) if split_out else torch.empty((B * Nq, H * 64), device=q.device, dtype=torch.float32)
    split_out = isinstance(out, SplitAct)
    if split_out:
        _split_out_ok(out, B * Nq)
    assert vt_bs % 2 == 0
    p = Attention()
    p.q, p.o, p.ldq, p.ldo = q.data_ptr(), out.data_ptr(), _ld(q) // 2, (_ld(out) // 2 if split_out else _ld(out))
    p.split_out, p.qkv_split = int(split_out), 1
    p.k[0], p.vt[0], p.ldk[0], p.ldvt[0], p.vt_bs[0], p.Nk[0], p.kv_div[0] = k.data_ptr(), vt.data_ptr(), _ld(k) // 2, _ld(vt) // 2, vt_bs // 2, nk, div
    p.zeros = workspace(q.device)[1].data_ptr()
    p.B, p.H, p.Nq, p.nseg, p.head_dim, p.dtype, p.scale = B, H, Nq, 1, 64, BF16X3, scale
    p.variant = ATTN_VARIANT if variant is None else variant
    _launch_attention(lib, p)
    return out


def _launch_attention(lib, p):
    if ATTN_TIMELINE is not None and p.nseg == 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.geo4d_attention(C.byref(p), _stream()), "geo4d_attention")
        e1.record()
        ATTN_TIMELINE.append((4.0 * p.B * p.H * p.Nq * p.Nk[0] * 64, e0, e1, 3 if p.dtype == BF16X3 else 1))     # (flops, events, MFMAs issued per product)
        return
    _lib.check(lib.geo4d_attention(C.byref(p), _stream()), "geo4d_attention")


def linear_t_batched(w, x, batch, rows, dtype_align=None, split_out=False):
    """Per-batch operand-swapped projection: x [batch*rows, K] -> out [batch, N, rows_pad] with out[b, n, m] =
    sum_k w[n, k] x[b*rows + m, k]; rows_pad = rows rounded up to a 16-byte multiple (zero filled). This is V^T per frame.
    `split_out` (bf16x3, pre-split x): out is a SplitAct [batch, N, 2*rows_pad] (rows_pad a multiple of 8) - V^T in the format the
    attention kernel takes with qkv_split."""
    N, K = w.shape[0], kdim(w, x)
    if split_out:
        assert isinstance(x, SplitAct)
        rp = (rows + 7) // 8 * 8
        out = SplitAct.wrap((torch.zeros if rp != rows else torch.empty)((batch, N, 2 * rp), device=x.device, dtype=torch.bfloat16))
        conv_gemm(w, x, out, M=N, N=rows, K=K, Cin=K, lda=_ld(w), ldw=_ld(x), ldo=2 * rp, batch=batch, a_bs=0, w_bs=rows * _ld(x), o_bs=N * 2 * rp)
        return out, rp
    odt = _out_dtype(x, None)
    epc = 4 if odt == torch.float32 else 8
    rp = (rows + epc - 1) // epc * epc
    out = (torch.zeros if rp != rows else torch.empty)((batch, N, rp), device=x.device, dtype=odt)
    conv_gemm(w, x, out, M=N, N=rows, K=K, Cin=K, lda=_ld(w), ldw=_ld(x), ldo=rp, batch=batch, a_bs=0, w_bs=rows * _ld(x), o_bs=N * rp)
    return out, rp


def linear_t(w, x, bias=None, *, out=None, pad_cols=None):
    """Operand-swapped projection: out[n, m] = sum_k w[n, k] * x[m, k] (+ bias[n]) -> [N, M]: the transposed layout the
    attention kernel wants for V. `pad_cols`: allocate zero-filled columns up to this count (16-byte row alignment)."""
    N, K = w.shape[0], kdim(w, x)
    M = x.shape[0]
    if out is None:
        cols = pad_cols or M
        odt = _out_dtype(x, None)
        out = torch.zeros((N, cols), device=x.device, dtype=odt) if cols != M else torch.empty((N, M), device=x.device, dtype=odt)
    return conv_gemm(w, x, out, M=N, N=M, K=K, Cin=K, lda=_ld(w), ldw=_ld(x), ldo=_ld(out), bias=bias, bias_per_row=True)


def temporal_attention(q, k, v, *, B, T, HW, H, scale, out=None, split_out=False):
    lib = _lib.load()
    _dev(q, "q")
    if out is None:
        out = new_split(B * T * HW, H * 64, q.device) if split_out else torch.empty((B * T * HW, H * 64), device=q.device, dtype=q.dtype)
    split_out = isinstance(out, SplitAct)
    _dev(k, "k"); _dev(v, "v")
    if split_out:
        _split_out_ok(out, B * T * HW)
    _lib.check(lib.geo4d_temporal_attention2(q.data_ptr(), _ld(q), k.data_ptr(), _ld(k), v.data_ptr(), _ld(v), out.data_ptr(),
                                             _ld(out) // 2 if split_out else _ld(out), B, T, HW, H, 64, scale, dt_code(q.dtype),
                                             int(split_out), _stream()), "geo4d_temporal_attention")
    return out


def tokens_from_ncthw(src0, src1, cpad, dtype):
    """[B,C0,T,H,W] (+[B,C1,T,H,W]) fp32 -> tokens [(b t) h w, cpad] of `dtype`, zero padded channels."""
    lib = _lib.load()
    _dev(src0, "src0")
    assert src0.dtype == torch.float32 and src0.is_contiguous()
    B, C0, T, H, W = src0.shape
    C1 = 0
    if src1 is not None:
        assert src1.dtype == torch.float32 and src1.is_contiguous() and src1.shape[0] == B and src1.shape[2:] == src0.shape[2:]
        C1 = src1.shape[1]
    out = torch.empty((B * T * H * W, cpad), device=src0.device, dtype=dtype)
    _lib.check(lib.geo4d_tokens_from_ncthw(src0.data_ptr(), C0, _ptr(src1), C1, out.data_ptr(), cpad, B, T, H * W,
                                           dt_code(dtype), _stream()), "geo4d_tokens_from_ncthw")
    return out


def cast_f16(x, out=None):
    """f32 rows [M, C] -> plain f16 rows (clamped, NaN kept): the two-pass GEMM's A operand made from a tensor without a normalising producer
    (bf16x3m class "vaeup": the VAE decoder's stream in front of its upsampling convolutions)."""
    lib = _lib.load()
    _dev(x, "x")
    assert x.dtype == torch.float32 and x.dim() == 2
    M, Cc = x.shape
    if out is None:
        out = torch.empty((M, Cc), device=x.device, dtype=torch.float16)
    _lib.check(lib.geo4d_cast_rows_f16(x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), M, Cc, _ptr(SAT_COUNTER), _stream()), "geo4d_cast_rows_f16")
    return out


def presplit(x):
    """f32 rows [M, C] -> SplitAct (the bf16x3 pre-split operand format): conv_gemm then skips its in-register split of the A fragments (12-29 %
    of a launch, paid per K slab and wave). Worth its own pass only where every input element is used many times: the 3x3 Upsample convolutions."""
    lib = _lib.load()
    _dev(x, "x")
    assert x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] % 8 == 0
    M, Cc = x.shape
    out = new_split(M, Cc, x.device)
    _lib.check(lib.geo4d_split_rows_bf16(x.data_ptr(), _ld(x), out.data_ptr(), _ld(out) // 2, M, Cc, _stream()), "geo4d_split_rows_bf16")
    return out


def concat_channels(a, b):
    lib = _lib.load()
    _dev(a, "a"); _dev(b, "b")
    M, Ca = a.shape
    Cb = b.shape[1]
    assert b.shape[0] == M and a.dtype == b.dtype
    out = torch.empty((M, Ca + Cb), device=a.device, dtype=a.dtype)
    _lib.check(lib.geo4d_concat_channels(a.data_ptr(), _ld(a), Ca, b.data_ptr(), _ld(b), Cb, out.data_ptr(), _ld(out), M,
                                         dt_code(a.dtype), _stream()), "geo4d_concat_channels")
    return out


def timestep_embedding(t, freqs):
    """t int64 [B] on device, freqs fp32 [dim/2] -> fp32 [B, dim] = [cos | sin]."""
    lib = _lib.load()
    _dev(t, "t")
    assert t.dtype == torch.int64 and freqs.dtype == torch.float32
    B, half = t.shape[0], freqs.shape[0]
    out = torch.empty((B, 2 * half), device=t.device, dtype=torch.float32)
    _lib.check(lib.geo4d_timestep_embedding(t.data_ptr(), freqs.data_ptr(), out.data_ptr(), B, 2 * half, _stream()),
               "geo4d_timestep_embedding")
    return out


def embed_tokens(tokens, table, pos, dtype):
    """tokens int64 [B, n_ctx] (device), table fp32 [vocab, W], pos fp32 [n_ctx, W] -> [B * n_ctx, W] of `dtype`."""
    lib = _lib.load()
    _dev(tokens, "tokens")
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and table.dtype == torch.float32 and pos.dtype == torch.float32
    B, n = tokens.shape
    out = torch.empty((B * n, table.shape[1]), device=tokens.device, dtype=dtype)
    _lib.check(lib.geo4d_embed_tokens(tokens.data_ptr(), table.data_ptr(), pos.data_ptr(), out.data_ptr(), _ld(out), B * n, n,
                                      table.shape[1], table.shape[0], dt_code(dtype), _stream()), "geo4d_embed_tokens")
    return out


def linear_small(x, w, bias=None, *, add=None, act_in=False, act_out=False, out=None):
    """fp32 [M,K] x fp32 [N,K]^T for M = batch-sized rows (time / fps / ResBlock embedding MLPs)."""
    lib = _lib.load()
    _dev(x, "x")
    assert x.dtype == torch.float32 and w.dtype == torch.float32
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    _lib.check(lib.geo4d_linear_small(x.data_ptr(), _ld(x), w.data_ptr(), _ld(w), _ptr(bias), _ptr(add),
                                      _ld(add) if add is not None else 0, out.data_ptr(), _ld(out), M, N, K, int(act_in),
                                      int(act_out), _stream()), "geo4d_linear_small")
    return out


def ddim_step(x, v, coef, step_index, noise=None, pred_x0=None):
    lib = _lib.load()
    _dev(x, "x")
    assert x.dtype == torch.float32 and v.dtype == torch.float32 and x.is_contiguous() and v.is_contiguous()
    assert coef.dtype == torch.float32 and step_index.dtype == torch.int32
    _lib.check(lib.geo4d_ddim_step(x.data_ptr(), v.data_ptr(), _ptr(noise), _ptr(pred_x0), coef.data_ptr(),
                                   step_index.data_ptr(), x.numel(), _stream()), "geo4d_ddim_step")
    return x


def cfg_combine(e_c, e_u, e_i=None, *, scale, cfg_img=None, guidance_rescale=0.0):
    """Classifier-free guidance on fp32 U-Net outputs [B, ...]: 2-way, or 3-way with e_i (image yes / text ""), + rescale_noise_cfg."""
    lib = _lib.load()
    _dev(e_c, "e_c")
    ts = [t for t in (e_c, e_u, e_i) if t is not None]
    assert all(t.dtype == torch.float32 and t.is_contiguous() and t.shape == e_c.shape for t in ts)
    B = e_c.shape[0]
    n = e_c.numel() // B
    out = torch.empty_like(e_c)
    need = lib.geo4d_cfg_combine_workspace(B)
    ws = torch.empty(need // 8, device=e_c.device, dtype=torch.float64)
    _lib.check(lib.geo4d_cfg_combine(e_c.data_ptr(), e_u.data_ptr(), _ptr(e_i), out.data_ptr(), B, n, float(scale),
                                     float(scale if cfg_img is None else cfg_img), float(guidance_rescale), ws.data_ptr(), need,
                                     _stream()), "geo4d_cfg_combine")
    return out


def advance_index(idx, delta):
    lib = _lib.load()
    _lib.check(lib.geo4d_advance_index(idx.data_ptr(), delta, _stream()), "geo4d_advance_index")


def gather_timestep(idx, table, ts):
    """ts[:] = table[idx] on the device (idx int32[1], table int64[S], ts int64[B])."""
    lib = _lib.load()
    assert idx.dtype == torch.int32 and table.dtype == torch.int64 and ts.dtype == torch.int64
    _lib.check(lib.geo4d_gather_timestep(idx.data_ptr(), table.data_ptr(), ts.data_ptr(), ts.numel(), _stream()),
               "geo4d_gather_timestep")
    return ts


def plucker_cameras(raymap, crossmap):
    """raymap, crossmap: fp32 [1, 3, T, H, W] (may be channel views of the decoded [B, 11, T, H, W] tensor) -> P_c2w [T, 4, 4]."""
    lib = _lib.load()
    _dev(raymap, "raymap"); _dev(crossmap, "crossmap")
    assert raymap.dtype == torch.float32 and crossmap.dtype == torch.float32 and raymap.shape == crossmap.shape
    B, Cc, T, H, W = raymap.shape
    assert B == 1 and Cc == 3, raymap.shape
    for m in (raymap, crossmap):
        assert m.stride(4) == 1 and m.stride(3) == W, "pixel planes must be contiguous"
    assert raymap.stride(1) == crossmap.stride(1) and raymap.stride(2) == crossmap.stride(2)
    need = lib.geo4d_plucker_cameras_workspace(T, H, W)
    ws = torch.empty(need // 8, device=raymap.device, dtype=torch.float64)
    out = torch.empty((T, 4, 4), device=raymap.device, dtype=torch.float32)
    _lib.check(lib.geo4d_plucker_cameras(raymap.data_ptr(), crossmap.data_ptr(), raymap.stride(1), raymap.stride(2), T, H, W,
                                         ws.data_ptr(), need, out.data_ptr(), _stream()), "geo4d_plucker_cameras")
    return out
