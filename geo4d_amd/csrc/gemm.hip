// geo4d_amd/csrc/gemm.hip — the workhorse: implicit-GEMM convolution / linear / batched GEMM on MFMA.
//
// One kernel covers every dense contraction of the Geo4D hot path:
//   nn.Linear                         (1 tap, Hin=Win=Hout=Wout=1, F = M rows)
//   nn.Conv2d 3x3 / 1x1, stride 1|2   (openaimodel3d.py:154,179 ResBlock; :51-78 Downsample)
//   nearest-2x upsample + Conv2d 3x3  (openaimodel3d.py:80-106, ae_modules.py:111-127) — the upsample is
//                                      folded into the gather (src = dst >> 1), nothing is materialised
//   nn.Conv3d (3,1,1), pad (1,0,0)    (openaimodel3d.py:257-266 TemporalConvBlock) — 3 temporal taps
//   batched Q.K^T / P.V GEMMs         (ae_modules.py:53-78 VAE AttnBlock, d = 512)
// Element types: bf16 / f16 (one MFMA 32x32x16 per 16 k), f32 (exact, v_mfma_f32_32x32x2_f32) and bf16x3 — f32 in memory,
// split into bf16 hi + lo in registers (weights pre-split at pack time) and multiplied with three bf16 MFMAs per 16 k: the
// precision of ~16-bit mantissas (point-map parity 1e-3 needs more than any single 16-bit pass gives, tests/precision_sim.py)
// at 3/16 of the f32-MFMA cost.
// Activations are channels-last tokens [F, H*W, C]; weights are packed [N][taps*Cin] (K-major, tap-major).
// out[m][n] = epilogue(alpha * sum_k A_gather[m][k] * W[n][k])
//
// Structure (kernel template in gemm_kernel.h; 4, 5, 8 or 10 waves, tile BM x BN x 128 B of K per stage; tiles 128x128, 128x64,
// 64x128, 64x64, 128x32, 256x128, 256x256, 160x320, 160x160 picked per problem by the host's measured table):
//   * gather table: the source pixel of every (tile row, tap) is resolved ONCE per workgroup into LDS
//     (-1 = zero padding), so the K loop does one ds_read + one 64-bit mad per 16-byte chunk instead of
//     re-deriving (frame, y, x), bounds and upsample/stride arithmetic every stage;
//   * global->LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR round trip and no
//     ds_write: the ds_write_b128 path tops out at ~79 B/clk/CU and was the limiter of the register-staged version),
//     2-deep LDS ring, one EXPLICIT `s_waitcnt vmcnt(0)` + barrier per stage (hipcc does not track LDS-DMA and drops the
//     wait inside loops). K is walked channel-slab major / tap minor so a conv's re-reads of its input hit the XCD's L2.
//     The DMA image is lane-linear (8 lanes = one 128-byte row),
//     so bank conflicts are removed by an XOR swizzle applied on the SOURCE address (slot = chunk ^ ((row>>1)&7))
//     and mirrored on the fragment ds_read_b128 (conflict-free for all four 16-lane service groups);
//     zero padding = lanes whose tap falls outside the image fetch from a 16-byte zero block;
//   * swapped MFMA operands: the weight fragment is the MFMA "A" side, the activation fragment the "B" side,
//     so a lane owns ONE output row m and 4 consecutive output columns n per register quad;
//   * epilogue through LDS (fp32): bias / row-bias / SiLU / GEGLU applied in registers, each wave transposes its tile through
//     a private 32 x 64 (or 32 x 32) block of the idle stage buffers, then residual add + rounding + coalesced 16-byte stores;
//   * optional split-K (gridDim.z) into fp32 partial slabs + a deterministic fixed-order reduce kernel
//     carrying the epilogue — for the 5x8 / 10x16 levels whose M x N tile count cannot fill 256 CUs;
//   * workgroup ids remapped so each XCD (private 4 MiB L2) walks a contiguous range of tiles with the
//     N-tile index fastest: neighbours share the gathered A panel.
#include "gemm_kernel.h"

namespace geo4d_gemm {
extern template int launch_typed<float>(const geo4d_conv_gemm_t&, hipStream_t);
extern template int launch_typed<bf16_t>(const geo4d_conv_gemm_t&, hipStream_t);
extern template int launch_typed<f16_t>(const geo4d_conv_gemm_t&, hipStream_t);
extern template int launch_typed<bf16x3_t>(const geo4d_conv_gemm_t&, hipStream_t);
}  // namespace geo4d_gemm
namespace geo4d_gemm {
template <typename T> int colsum_rows_v23(const geo4d_conv_gemm_t& p);      // gemm_kernel_v3.h, instantiated in gemm_v3_bf16x3.hip / gemm_v3_bf16.hip
extern template int colsum_rows_v23<bf16x3_t>(const geo4d_conv_gemm_t&);
extern template int colsum_rows_v23<bf16_t>(const geo4d_conv_gemm_t&);
extern template int colsum_rows_v23<f16x2p_t>(const geo4d_conv_gemm_t&);
// f16x2 (dtype 4) exists on the second / third generation only
extern template int launch_v2_typed<f16x2p_t>(const geo4d_conv_gemm_t&, hipStream_t);
extern template int launch_v3_typed<f16x2p_t>(const geo4d_conv_gemm_t&, hipStream_t);
}  // namespace geo4d_gemm
using geo4d_gemm::BKC;
using geo4d_gemm::MAXTAP;
using geo4d_gemm::launch_typed;

extern "C" int geo4d_conv_gemm(const geo4d_conv_gemm_t* pp, void* stream) {
    if (!pp) return GEO4D_EINVAL;
    geo4d_conv_gemm_t p = *pp;
    const int esz = (p.dtype == GEO4D_F32 || p.dtype == GEO4D_BF16X3 || p.dtype == GEO4D_F16X2) ? 4 : 2;
    const int epc = 16 / esz;
    const int bk = BKC * epc;
    if (p.dtype < 0 || p.dtype > 4 || p.out_dtype < 0 || p.out_dtype > 2) { geo4d_set_error("conv_gemm: bad dtype"); return GEO4D_EINVAL; }
    if (p.dtype != GEO4D_BF16X3 && p.dtype != GEO4D_F16X2 && (p.a_split || p.w_split)) { geo4d_set_error("conv_gemm: a_split / w_split are bf16x3 / f16x2 (dtype 3 / 4) options"); return GEO4D_EINVAL; }
    if (p.dtype == GEO4D_F16X2) {
        // two-pass f16: plain f16 activation rows x a pre-split f16 weight, f32 rows (or, o_split = 2, plain f16 rows) out, second / third
        // generation tiles (0 = a default per shape)
        if (p.a_split != 2 || !p.w_split || (p.o_split != 0 && p.o_split != 2) || p.out_nchw || p.out_dtype != GEO4D_F32 || (p.tile_hint != 0 && p.tile_hint < 22)) {
            geo4d_set_error("conv_gemm: f16x2 (dtype 4) needs a_split = 2 (plain f16 activation rows) and w_split, a row-major output (f32 rows, or o_split = 2: plain f16 rows), and tile hint 0 or >= 22");
            return GEO4D_EINVAL;
        }
        // (GEGLU lives on the tiles whose wave tiles are a multiple of 64 columns wide)
        if (p.tile_hint == 0) p.tile_hint = p.M >= 4096 ? (p.act == 2 ? 71 : 72) : 25;
        if (p.split_k == 0) p.split_k = 1;
    }
    if (p.o_split) {
        const int nst = p.act == 2 ? p.N / 2 : p.N;
        if ((p.dtype != GEO4D_BF16X3 && p.dtype != GEO4D_F16X2) || (p.dtype == GEO4D_BF16X3 && p.o_split != 1) || p.out_dtype != GEO4D_F32 || !p.a_split || !p.w_split || p.split_k > 1 || p.out_nchw || p.gn_colsum ||
            (nst % 8) || (p.ldo % 8) || (p.o_bs % 8) || ((uintptr_t)p.O % (p.o_split == 2 ? 16 : 32))) {
            geo4d_set_error("conv_gemm: o_split needs a pre-split x pre-split bf16x3 launch, f32 row-major output, stored columns % 8 == 0, 32-byte aligned rows, no split-K");
            return GEO4D_EINVAL;
        }
    }
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.batch <= 0) { geo4d_set_error("conv_gemm: empty problem"); return GEO4D_EINVAL; }
    if (p.KT <= 0 || p.KH <= 0 || p.KW <= 0 || p.KT * p.KH * p.KW > MAXTAP) { geo4d_set_error("conv_gemm: at most 9 taps"); return GEO4D_EINVAL; }
    if (p.Cin % bk || p.K != p.Cin * p.KT * p.KH * p.KW) { geo4d_set_error("conv_gemm: Cin must be a multiple of the 128-byte K slab and K = taps*Cin"); return GEO4D_EINVAL; }
    const int esz_a = p.dtype == GEO4D_F16X2 ? 2 : esz;      // (dtype 4: the activation is plain f16 rows)
    if ((p.lda * esz_a) % 16 || (p.ldw * esz) % 16 || ((uintptr_t)p.A % 16) || ((uintptr_t)p.W % 16) || (p.a_bs * esz_a) % 16 || (p.w_bs * esz) % 16) {
        geo4d_set_error("conv_gemm: operands must be 16-byte aligned");
        return GEO4D_EINVAL;
    }
    if (p.T <= 0 || p.Hout <= 0 || p.Wout <= 0 || p.Hin <= 0 || p.Win <= 0 || p.M % (p.Hout * p.Wout)) { geo4d_set_error("conv_gemm: bad geometry"); return GEO4D_EINVAL; }
    if ((p.M / (p.Hout * p.Wout)) % p.T) { geo4d_set_error("conv_gemm: frames not a multiple of T"); return GEO4D_EINVAL; }
    if ((long)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Win > 2147483647L) { geo4d_set_error("conv_gemm: more than 2^31 input pixels"); return GEO4D_EINVAL; }
    if (p.ups != 1 && p.ups != 2) { geo4d_set_error("conv_gemm: ups must be 1 or 2"); return GEO4D_EINVAL; }
    if (p.act == 2 && (p.N % 64 || p.out_nchw || p.R)) { geo4d_set_error("conv_gemm: GEGLU needs N % 64 == 0, row-major output, no residual"); return GEO4D_EINVAL; }
    if (p.act < 0 || p.act > 3) { geo4d_set_error("conv_gemm: act must be 0 (none), 1 (SiLU), 2 (GEGLU) or 3 (GELU)"); return GEO4D_EINVAL; }
    if (p.act == 3) {   // GELU lives in the vectorised epilogue only (the scalar NCTHW / odd-shape path stays small enough to unroll)
        const int oesz = p.out_dtype == GEO4D_F32 ? 4 : 2;
        if (p.out_nchw || p.R || (p.N % 8) || ((p.ldo * oesz) % 16) || ((uintptr_t)p.O % 16) || ((p.o_bs * oesz) % 16)) {
            geo4d_set_error("conv_gemm: GELU needs a row-major output with N % 8 == 0, 16-byte aligned rows and no residual");
            return GEO4D_EINVAL;
        }
    }
    if (p.gn_colsum && p.tile_hint < 21) {      // (first generation: one entry per 32 rows; the second / third generation launchers check their own form)
        const int oesz = p.out_dtype == GEO4D_F32 ? 4 : 2;
        if ((p.M % 32) || (p.N % 8) || p.out_nchw || p.act == 2 || p.batch != 1 || p.split_k != 1 || ((p.ldo * oesz) % 16) || ((uintptr_t)p.O % 16) ||
            ((uintptr_t)p.gn_colsum % 16) || (p.R && (((p.ldr * oesz) % 16) || ((uintptr_t)p.R % 16)))) {
            geo4d_set_error("conv_gemm: gn_colsum needs M % 32 == 0, N % 8 == 0, a row-major 16-byte aligned output, batch 1, split_k = 1, no GEGLU");
            return GEO4D_EINVAL;
        }
    }
    if (p.rowbias && p.rowbias_div <= 0) { geo4d_set_error("conv_gemm: rowbias_div"); return GEO4D_EINVAL; }
    if (p.batch > 65535) { geo4d_set_error("conv_gemm: batch too large"); return GEO4D_EINVAL; }
    if (!p.zeros || ((uintptr_t)p.zeros % 16)) { geo4d_set_error("conv_gemm: `zeros` must point at 16 zero bytes (16-byte aligned) in device memory"); return GEO4D_EINVAL; }
    if (p.workspace && ((uintptr_t)p.workspace % 16)) { geo4d_set_error("conv_gemm: workspace alignment"); return GEO4D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    switch (p.dtype) {
        case GEO4D_F32: return launch_typed<float>(p, s);
        case GEO4D_BF16: return launch_typed<bf16_t>(p, s);
        case GEO4D_BF16X3: return launch_typed<bf16x3_t>(p, s);
        case GEO4D_F16X2: return p.tile_hint >= 71 ? geo4d_gemm::launch_v3_typed<f16x2p_t>(p, s) : geo4d_gemm::launch_v2_typed<f16x2p_t>(p, s);
        default: return launch_typed<f16_t>(p, s);
    }
}

// Rows of the output that ONE gn_colsum entry of the launch `*pp` describes covers (tile_hint / split_k as they will be launched;
// the gn_colsum field itself is ignored): 32 for the first-generation tiles, the wave tile's rows for the second / third generation,
// 0 = this launch cannot emit the sums (the caller then leaves gn_colsum null and the GroupNorm runs its own statistics pass).
extern "C" int geo4d_conv_gemm_colsum_rows(const geo4d_conv_gemm_t* pp) {
    if (!pp) return 0;
    const geo4d_conv_gemm_t& p = *pp;
    if (p.tile_hint < 21) {
        const int oesz = p.out_dtype == GEO4D_F32 ? 4 : 2;
        const bool ok = !(p.M % 32) && !(p.N % 8) && !p.out_nchw && p.act != 2 && p.batch == 1 && p.split_k <= 1 && !p.o_split && !((p.ldo * oesz) % 16) &&
                        !((uintptr_t)p.O % 16) && (!p.R || (!((p.ldr * oesz) % 16) && !((uintptr_t)p.R % 16)));
        return ok ? 32 : 0;
    }
    if (p.dtype == GEO4D_BF16X3) return geo4d_gemm::colsum_rows_v23<bf16x3_t>(p);
    if (p.dtype == GEO4D_BF16) return geo4d_gemm::colsum_rows_v23<bf16_t>(p);
    if (p.dtype == GEO4D_F16X2) return geo4d_gemm::colsum_rows_v23<f16x2p_t>(p);
    return 0;
}
