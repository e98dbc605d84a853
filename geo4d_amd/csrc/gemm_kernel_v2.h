// geo4d_amd/csrc/gemm_kernel_v2.h — second-generation implicit-GEMM kernel (round 3). Included by gemm_kernel.h; same ABI struct,
// same gather / LDS-DMA ring / K order as conv_gemm_kernel, with the three changes the round-2 probes justified
// (profiles/r02_mfma_probe_and_kloop.md, tools/probe/NEXT_KERNEL_NOTES.md):
//   1. v_mfma_f32_16x16x32_{bf16,f16}: 86-87 % of the dense peak on random data where the 32x32x16 form keeps 76 %. Wave tiles are
//      multiples of 16 rows / columns (80x80 becomes possible: 160x320 and 160x160 tiles on 8 / 4 waves, balanced over the 4 SIMDs,
//      instead of the 10- / 5-wave layouts of tile hints 16 / 17).
//      bf16x3 gets its own source-side LDS swizzle (xor key p ^ ((p>>2 ^ p>>1) & 1) << 1 on the row pair p): with the 16-row
//      fragment a lane quarter reads chunk pair (2c, 2c+1), for which the 32x32 key is 2-way conflicted.
//   2. register epilogue: with swapped operands a lane of the 16x16 accumulator owns 4 CONSECUTIVE output columns of one row, so
//      bias / row-bias / activation / residual / rounding happen in registers and the stores are 16-byte (f32) or 8-byte (16-bit)
//      vectors straight to global memory - no LDS transpose, hence the ring is free while the epilogue runs, hence
//   3. persistent workgroups: grid = resident workgroups, a static round-robin tile loop; the NEXT tile's gather table, weight-row
//      pointers and first K slab (LDS-DMA) are issued BEFORE the current tile's epilogue, so the DMA latency and the table's
//      integer divides run under the epilogue's global traffic instead of in front of the next tile's first MFMA.
// Not covered here (the host keeps such launches on conv_gemm_kernel): f32 element type, NCTHW outputs.
#pragma once
#include "gemm_kernel.h"

namespace geo4d_gemm {

template <typename T> __device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma16<bf16_t>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<f16_t>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
// acc += a.b with a = ah + al, b = bh + bl (al.bl dropped): three dense bf16 MFMAs of the 16x16x32 form
__device__ __forceinline__ void mma16_x3(f32x4& acc, const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, al), __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
}

// f16x2 (two-pass): acc += ah.(bh + bl) - the activation's hi half against both halves of the weight, two f16 MFMAs (lo term first,
// like mma16_x3; `bh` / `bl` are the MFMA's A side = the weight, `ah` its B side = the activation)
__device__ __forceinline__ void mma16_x2(f32x4& acc, const u32x4& bh, const u32x4& bl, const u32x4& ah) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, bl), __builtin_bit_cast(f16x8_t, ah), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, bh), __builtin_bit_cast(f16x8_t, ah), acc, 0, 0, 0);
}

// LDS slot of logical 16-byte chunk c in panel row r = c ^ swz_key<T>(r). ds_read_b128 is served in lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with lane = 16 * quarter + row, a group holds rows {0-3,12-15} of one quarter and
// rows {4-11} of the next. 16-bit types read chunk 4h + quarter -> the plain row-pair key keeps the 16 slots distinct; bf16x3 reads
// chunk 2 * quarter (+1), and needs rows {4-11} keyed with bit 1 flipped.
template <typename T> __device__ __forceinline__ int swz_key(int row) {
    const int p = (row >> 1) & 7;
    if constexpr (IsX3<T>::value) return p ^ ((((p >> 2) ^ (p >> 1)) & 1) << 1);
    else return p;
}

// Tile order inside one (batch, split) slab of tiles_m x tiles_n tiles. The persistent workgroups of an XCD work on ~32 CONSECUTIVE
// tile ids at a time (xcd_remap): with the N index fastest (the default) those are 1 row-tile x 32 column-tiles - one A panel shared, 32
// weight panels streamed, and every row-tile streams the whole weight matrix again (PMC: the level-2 GEGLU launch fetches 14 x its 52 MB
// weight, profiles/r05_head_pmc_bf16x3m.md). Round 5 built the grouped order (GROUP_M row-tiles fastest inside a band of GROUP_M rows: the
// 32 tiles become a GROUP_M x 32 / GROUP_M block) and measured it (profiles/r05_tile_order.md): L2-miss fetch of a U-Net forward 78.1 ->
// 70.8 GB (-9 %), frames/s 5.925 -> 5.905 (-0.3 %, two interleaved pairs on one box): the misses are served from the Infinity Cache at
// a rate the K loop does not wait for, while the grouped order shares each A panel among fewer neighbours. The column-fastest order
// stays the default; debug_ablate = 16 + g selects GROUP_M = g. Same tiles, same arithmetic either way: bits unchanged.
__device__ __forceinline__ void tile_of(long t, int tiles_n, int tiles_mn, int group_m, int& tm, int& tn) {
    const int tiles_m = tiles_mn / tiles_n;
    if (group_m <= 1 || tiles_n < 2 || tiles_m < 2) {
        tm = (int)(t / tiles_n);
        tn = (int)(t - (long)tm * tiles_n);
        return;
    }
    const int per_band = group_m * tiles_n;
    const int band = (int)(t / per_band);
    const int first_m = band * group_m;
    const int rows = min(tiles_m - first_m, group_m);
    const int r = (int)(t - (long)band * per_band);
    tn = r / rows;
    tm = first_m + (r - tn * rows);
}
// debug_ablate 16 + g (A/B runs, tests): GROUP_M = g; otherwise 1 = column-fastest
__device__ __forceinline__ int tile_group_m(const geo4d_conv_gemm_t& p) { return (p.debug_ablate >= 16 && p.debug_ablate < 48) ? p.debug_ablate - 16 : 1; }

// The two-pass f16 type's activation panel has 64-byte rows (round 5, "A64"): it multiplies ONE f16 per activation element, and the
// cost of staging is per 1 KB LDS-DMA request (profiles/r03_gemm_v2_explore_and_ablation.md), so a request covers 16 rows of the 32-k slab
// instead of 8: half the A-side requests. Round 5 fetched the four hi chunks out of [8 hi | 8 lo] groups (16 of every 32 bytes of a row
// whose lo halves nobody read); round 6: the producers write PLAIN f16 rows and a panel row is 64 CONTIGUOUS bytes of the source row -
// the same LDS image, the same arithmetic, the same bits, half the activation bytes written and fetched.
// LDS slot of K-group g in panel row r = g ^ swz_key_a64(r); ds_read_b128 service groups (see swz_key): with 4 slots per row the 16
// lanes of a group hit 16 distinct 16-byte bank groups when rows {0-3, 4-7, 8-11, 12-15} of a 16-row block are keyed {0, 3, 2, 1}.
__device__ __forceinline__ int swz_key_a64(int row) { return (4 - ((row >> 2) & 3)) & 3; }
constexpr int PITCH_A64 = 64;

// ---- register epilogue shared by the second- and third-generation kernels: acc[a][b][j] is out[m_w0 + 16a + lr][n_w0 + 16b + 4lq + j]
// (bias / row-bias / activation / GEGLU / residual / rounding in registers, 16-byte (f32) or 8-byte (16-bit) stores). `partial`:
// split-K launch, the raw fp32 slab of split e_kz goes to the workspace and the epilogue runs in splitk_reduce_kernel.
// Round 4 - fast paths for 4-byte rows (every bf16x3 projection / conv except the ResBlocks' emb-add convs and the per-row-bias V^T
// projections). tools/gemm_kscan.py measured a K-INDEPENDENT ~30 us per tile in these kernels (profiles/r04_gemm_kscan.md: 97 us of a
// 151 us level-0 qkv launch): the generic code below loads bias / residual element by element behind per-element branches, each load
// followed by s_waitcnt vmcnt(0) - and vmcnt is in order over loads AND stores, so every block's loads waited until the previous
// block's stores were acknowledged (~1.3 us x 24-25 blocks per tile). The fast paths have no per-element branches: the column bias
// is ONE 16-byte load per block outside the row loop, the residual one 16-byte load issued a row block AHEAD of the stores (counted
// wait), and the pre-split output is written as whole 16-byte hi / lo chunks (the two lanes of an 8-column group trade their halves
// with v_permlane16_swap) instead of two 8-byte pieces per lane. Same values, same addresses.
// gn_colsum from the fast path (round 4): per WAVE-TILE row range (MB x 16 rows) and output column, (sum, sum of squares) of the values
// this launch stores - the consumer GroupNorm's statistics pass, skipped (norm.hip gn_finalize_cols, rows per entry = the wave tile's
// rows). Available where the plain f32-row fast path runs (no activation, no split-K, no pre-split output) and M is a multiple of the
// wave tile's rows; geo4d_conv_gemm_colsum_rows() tells the host which rows a launch will use (0 = not available).
inline bool colsum_fast_ok(const geo4d_conv_gemm_t& p, int sp) {
    return sp == 1 && p.act == 0 && !p.o_split && !p.out_nchw && p.batch == 1 && p.out_dtype == GEO4D_F32 && (p.N & 3) == 0 && (p.ldo & 3) == 0 &&
           ((uintptr_t)p.O % 16) == 0 && (!p.R || ((p.ldr & 3) == 0 && ((uintptr_t)p.R % 16) == 0));
}
// The two-pass f16 type has no 256x256 instantiation (that tile spills a few registers around its K loop, and the long-K convolutions
// the type exists for run on the phased tiles): hint 22 means the 160x320 tile there - same bits, every tile sums in the same order.
// (GEGLU needs wave tiles a multiple of 64 columns wide, which 160x320 is not: 128x128 there.)
template <typename T> inline int v2_effective_hint(int hint, int act = 0, int o_split = 0) { return (IsTwoPass<T>::value && hint == 22) ? (act == 2 ? 25 : 23) : hint; }
inline int v2_wave_rows(int hint) { return hint == 22 ? 64 : hint == 23 ? 80 : hint == 25 ? 64 : (hint == 27 || hint == 28) ? 32 : 0; }
__device__ __forceinline__ float row16_sum(float v) {      // sum over the 16 lanes lr of a 16-lane row, fixed order (DPP: xor 1, xor 2, half mirror, mirror)
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));     // row_mirror
    return v;
}
template <int MB, int NB, bool OSPLIT, bool VECONLY = false, bool ROWS4 = false, bool OH = false>   // OH: the two-pass f16 type - its OSPLIT output (o_split = 2) is PLAIN f16 rows instead of bf16 hi | lo halves; VECONLY: the host checked the vector-store conditions (no scalar fallback code); ROWS4: 4-byte element kernels (bf16x3) - the 16-bit-row fast paths are not instantiated
__device__ __forceinline__ void reg_epilogue(const geo4d_conv_gemm_t& p, const f32x4 (&acc)[MB][NB], const int m_w0, const int n_w0,
                                             const long e_bz, const int e_kz, const bool partial, const int lr, const int lq) {
    const int odt = partial ? GEO4D_F32 : p.out_dtype;
    const int oesz = odt == GEO4D_F32 ? 4 : 2;
    const bool geglu = !partial && p.act == 2;
    const int nout = geglu ? (p.N >> 1) : p.N;
    void* O = partial ? (void*)((float*)p.workspace + ((long)e_kz * p.batch + e_bz) * (long)p.M * p.N) : p.O;
    const long ldo = partial ? (long)p.N : p.ldo;
    const long obase = partial ? 0 : e_bz * p.o_bs;
    const bool has_res = !partial && p.R != nullptr;
    const long rbase = e_bz * p.r_bs;
    // 4-element vectors need 4-element aligned rows and bases (16 B for f32, 8 B for 16-bit outputs)
    const bool vec_ok = VECONLY || ((nout & 3) == 0 && (ldo & 3) == 0 && (((uintptr_t)O + obase * oesz) % (4 * oesz)) == 0 &&
                                    (!has_res || ((p.ldr & 3) == 0 && (((uintptr_t)p.R + rbase * oesz) % (4 * oesz)) == 0)));
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    // this lane's 16-byte chunk `lq` of a block's 64 output bytes (4-byte rows)
    auto chunk_of = [&](const float (&e)[4]) __attribute__((always_inline)) -> u32x4 {
        if constexpr (OSPLIT) {
            unsigned int h0, h1, l0, l1;
            h0 = f32x2_to_bf16x2(e[0], e[1]); h1 = f32x2_to_bf16x2(e[2], e[3]);
            l0 = f32x2_to_bf16x2(e[0] - __uint_as_float(h0 << 16), e[1] - __uint_as_float(h0 & 0xffff0000u));
            l1 = f32x2_to_bf16x2(e[2] - __uint_as_float(h1 << 16), e[3] - __uint_as_float(h1 & 0xffff0000u));
            // rows of 16 lanes = lq: odd rows of (h) <-> even rows of (l): even lq ends with [h own | h of lq + 1] = the group's hi chunk,
            // odd lq with [l of lq - 1 | l own] = its lo chunk (every lane of the wave takes part: no divergence before this point)
            const u32x2 s0 = __builtin_amdgcn_permlane16_swap(h0, l0, false, false);
            const u32x2 s1 = __builtin_amdgcn_permlane16_swap(h1, l1, false, false);
            return u32x4{s0[0], s1[0], s0[1], s1[1]};
        } else {
            return u32x4{__float_as_uint(e[0]), __float_as_uint(e[1]), __float_as_uint(e[2]), __float_as_uint(e[3])};
        }
    };
    if constexpr (OSPLIT && OH) {
        // o_split = 2 (the two-pass f16 type; round 6): O = PLAIN f16 rows - the A operand of the next dtype-4 launch (GEGLU -> ff-out). The
        // launcher checked: no split-K, no residual / row biases, act 0 or GEGLU, stored columns % 8 == 0, 16-byte aligned rows. 8-byte
        // vector stores through a buffer resource (a lane outside M x N offers an out-of-window offset), values clamped to the finite f16
        // range with NaN kept (common.h pack4_f16_sat); p.sat_count (debug) counts the clamped lanes.
        constexpr unsigned OOB2 = 0x80000000u;
        auto uptr = [](const void* q) __attribute__((always_inline)) -> void* {
            const unsigned long long v = (unsigned long long)q;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return (void*)(((unsigned long long)hi << 32) | lo);
        };
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
            uptr((unsigned short*)p.O + e_bz * p.o_bs + (long)m_w0 * p.ldo + (geglu ? (n_w0 >> 1) : n_w0)), 0, OOB2, 0x00020000);
        const int nleft = p.N - n_w0;
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
            uptr(p.bias ? (const void*)(p.bias + n_w0) : p.zeros), 0, (p.bias && nleft > 0) ? (unsigned)nleft * 4u : 0u, 0x00020000);
        const unsigned offO = (unsigned)(lr * (int)p.ldo + 4 * lq) * 2u, rowO = (unsigned)p.ldo * 32u;       // bytes per 16-row block
        if (geglu) {
            if constexpr (NB % 4 == 0) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if ((b & 3) >= 2) continue;
                    const bool grp = n_w0 + 16 * (b & ~3) + 64 <= p.N;       // whole 64-column value | gate groups only (N % 64 == 0)
                    const u32x4 bv = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(4 * lq) * 4u + 64u * b, 0, 0);
                    const u32x4 bg = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(4 * lq) * 4u + 64u * b + 128u, 0, 0);
#pragma unroll
                    for (int a = 0; a < MB; ++a) {
                        const bool ok = grp && m_w0 + a * 16 + lr < p.M;
                        float e[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            e[j] = (acc[a][b][j] * p.alpha + __uint_as_float(bv[j])) * gelu_erf_f(acc[a][b + 2 < NB ? b + 2 : b][j] * p.alpha + __uint_as_float(bg[j]));
                        count_f16_saturation(p.sat_count, e);
                        __builtin_amdgcn_raw_buffer_store_b64(pack4_f16_sat(e), rsO, (ok ? offO + a * rowO : OOB2) + (unsigned)(32 * (b >> 2) + 16 * (b & 1)) * 2u, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const bool colok = n_w0 + 16 * b + 4 * lq < p.N;
                const u32x4 bcu = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(4 * lq) * 4u + 64u * b, 0, 0);
#pragma unroll
                for (int a = 0; a < MB; ++a) {
                    const bool ok = colok && m_w0 + a * 16 + lr < p.M;
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = acc[a][b][j] * p.alpha + __uint_as_float(bcu[j]);
                    count_f16_saturation(p.sat_count, e);
                    __builtin_amdgcn_raw_buffer_store_b64(pack4_f16_sat(e), rsO, (ok ? offO + a * rowO : OOB2) + 32u * b, 0, 0);
                }
            }
        }
        return;
    }
    const bool wide = vec_ok && odt == GEO4D_F32;       // (o_split: N % 8 == 0, so the two lanes of an 8-column group are in range together)
    // Fast paths: every access goes through a raw buffer resource whose base is this wave's tile corner (wave-uniform, SGPRs): a lane
    // outside M x N offers an offset beyond the 2 GB window (stores dropped, loads return 0 - no exec-masked branches, so hipcc's
    // waits stay COUNTED), an absent bias / residual is a resource with zero records (its loads return 0 without touching memory).
    constexpr unsigned OOB = 0x80000000u;
    auto uniform_ptr = [](const void* q) __attribute__((always_inline)) -> void* {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (void*)(((unsigned long long)hi << 32) | lo);
    };
    if (wide && !geglu && (partial || p.act == 0)) {
        const float alpha = partial ? 1.0f : p.alpha;
        const bool hb = !partial && p.bias != nullptr && !p.bias_per_row;      // bias per output column
        const bool hr = !partial && p.bias != nullptr && p.bias_per_row;       // bias per output row (the operand-swapped V^T projection)
        const bool ht = !partial && p.rowbias != nullptr;                      // row-bias table (the ResBlocks' per-frame emb add): row m / rowbias_div
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((float*)O + obase + (long)m_w0 * ldo + n_w0), 0, OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(has_res ? (const void*)((const float*)p.R + rbase + (long)m_w0 * p.ldr + n_w0) : p.zeros), 0, has_res ? OOB : 0u, 0x00020000);
        const int nleft = p.N - n_w0, mleft = p.M - m_w0;
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(hb ? (const void*)(p.bias + n_w0) : p.zeros), 0, (hb && nleft > 0) ? (unsigned)nleft * 4u : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsBr = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(hr ? (const void*)(p.bias + m_w0) : p.zeros), 0, (hr && mleft > 0) ? (unsigned)mleft * 4u : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ht ? (const void*)(p.rowbias + n_w0) : p.zeros), 0, ht ? OOB : 0u, 0x00020000);
        const unsigned offO = (unsigned)(lr * (int)ldo + 4 * lq) * 4u, offR = (unsigned)(lr * (int)p.ldr + 4 * lq) * 4u;
        const unsigned rowO = (unsigned)ldo * 64u, rowR = (unsigned)p.ldr * 64u;       // bytes per 16-row block
        // (the launcher checked M % (16 MB) == 0 and the fast-path conditions; a wave tile that starts beyond M - the last row tile of a
        // ragged M - has no entry: its chunk index would lie beyond the [M / rows] buffer)
        const bool gn = !partial && p.gn_colsum != nullptr && m_w0 < p.M;
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(gn ? (const void*)(p.gn_colsum + ((long)(m_w0 / (16 * MB)) * p.N + n_w0) * 2) : p.zeros), 0, gn ? OOB : 0u, 0x00020000);
        unsigned offT[MB];                                                              // byte offset of this lane's row-bias row per row block
#pragma unroll
        for (int a = 0; a < MB; ++a) offT[a] = 0u;
        if (ht) {
            const unsigned ldt = (unsigned)(p.ldrb ? p.ldrb : (long)p.N);
#pragma unroll
            for (int a = 0; a < MB; ++a) {
                const int m = m_w0 + a * 16 + lr;
                offT[a] = ((unsigned)((m < p.M ? m : p.M - 1) / p.rowbias_div) * ldt + 4u * lq) * 4u;
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const bool colok = n_w0 + 16 * b + 4 * lq < p.N;
            const u32x4 bcu = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(4 * lq) * 4u + 64u * b, 0, 0);
            u32x4 ru = __builtin_amdgcn_raw_buffer_load_b128(rsR, ((colok && m_w0 + lr < p.M) ? offR : OOB) + 64u * b, 0, 0);
            u32x4 tu = __builtin_amdgcn_raw_buffer_load_b128(rsT, (colok ? offT[0] : OOB) + 64u * b, 0, 0);
            unsigned bru = __builtin_amdgcn_raw_buffer_load_b32(rsBr, (unsigned)lr * 4u, 0, 0);
            float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < MB; ++a) {
                const bool ok = colok && m_w0 + a * 16 + lr < p.M;
                const float brow = __uint_as_float(bru);
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    e[j] = (((acc[a][b][j] * alpha + brow) + __uint_as_float(bcu[j])) + __uint_as_float(tu[j])) + __uint_as_float(ru[j]);
                if (gn) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { cs[j] += e[j]; cq[j] += e[j] * e[j]; }
                }
                const u32x4 c = chunk_of(e);
                if (a + 1 < MB) {            // the next row block's residual / row biases go out BEFORE this block's store: their wait stays counted
                    ru = __builtin_amdgcn_raw_buffer_load_b128(rsR, ((colok && m_w0 + (a + 1) * 16 + lr < p.M) ? offR : OOB) + 64u * b, (a + 1) * rowR, 0);
                    tu = __builtin_amdgcn_raw_buffer_load_b128(rsT, (colok ? offT[a + 1 < MB ? a + 1 : a] : OOB) + 64u * b, 0, 0);
                    bru = __builtin_amdgcn_raw_buffer_load_b32(rsBr, (unsigned)(lr + 16 * (a + 1)) * 4u, 0, 0);
                }
                // (row-block offset in the VGPR offset, soffset = 0: with an SGPR soffset hipcc's hazard recognizer assumes a 16-byte buffer
                // store's data registers may be overwritten right away - on gfx950 the last lanes of every 16 then stored the NEXT block's
                // values, tools/dbg_epilogue.py; measured round 4)
                __builtin_amdgcn_raw_buffer_store_b128(c, rsO, (ok ? offO + a * rowO : OOB) + 64u * b, 0, 0);
            }
            if (gn) {                        // [chunk][n][2]: lane lr == 0 of every 16-lane row writes its 4 columns' (sum, sum of squares)
#pragma unroll
                for (int j = 0; j < 4; ++j) { cs[j] = row16_sum(cs[j]); cq[j] = row16_sum(cq[j]); }
                const unsigned offC = (lr == 0 && colok) ? (unsigned)(16 * b + 4 * lq) * 8u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(cs[0]), __float_as_uint(cq[0]), __float_as_uint(cs[1]), __float_as_uint(cq[1])}, rsC, offC, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(cs[2]), __float_as_uint(cq[2]), __float_as_uint(cs[3]), __float_as_uint(cq[3])}, rsC, offC + 16u, 0, 0);
            }
        }
        return;
    }
    if (wide && geglu) {             // bias (value | gate columns) once per block, outside the row loop; only stores inside
        if constexpr (NB % 4 == 0) {
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((float*)O + obase + (long)m_w0 * ldo + (n_w0 >> 1)), 0, OOB, 0x00020000);
            const int nleft = p.N - n_w0;
            const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
                uniform_ptr(p.bias ? (const void*)(p.bias + n_w0) : p.zeros), 0, (p.bias && nleft > 0) ? (unsigned)nleft * 4u : 0u, 0x00020000);
            const unsigned offO = (unsigned)(lr * (int)ldo + 4 * lq) * 4u, rowO = (unsigned)ldo * 64u;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if ((b & 3) >= 2) continue;
                // whole 64-column value | gate groups only (N % 64 == 0): wave-uniform, folded into the store offset
                const bool grp = n_w0 + 16 * (b & ~3) + 64 <= p.N;
                const u32x4 bv = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(4 * lq) * 4u + 64u * b, 0, 0);
                const u32x4 bg = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(4 * lq) * 4u + 64u * b + 128u, 0, 0);
#pragma unroll
                for (int a = 0; a < MB; ++a) {
                    const bool ok = grp && m_w0 + a * 16 + lr < p.M;
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        e[j] = (acc[a][b][j] * p.alpha + __uint_as_float(bv[j])) * gelu_erf_f(acc[a][b + 2 < NB ? b + 2 : b][j] * p.alpha + __uint_as_float(bg[j]));
                    const u32x4 c = chunk_of(e);
                    __builtin_amdgcn_raw_buffer_store_b128(c, rsO, (ok ? offO + a * rowO : OOB) + (unsigned)(32 * (b >> 2) + 16 * (b & 1)) * 4u, 0, 0);
                }
            }
        }
        return;
    }
    if constexpr (!ROWS4)
    if (vec_ok && odt != GEO4D_F32 && !geglu && p.act == 0 && !p.rowbias && !(p.bias && p.bias_per_row)) {
        // 16-bit rows (the bf16 / f16 modes): the same structure with 8-byte vectors
        const bool isbf = odt == GEO4D_BF16;
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((unsigned short*)O + obase + (long)m_w0 * ldo + n_w0), 0, OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(has_res ? (const void*)((const unsigned short*)p.R + rbase + (long)m_w0 * p.ldr + n_w0) : p.zeros), 0, has_res ? OOB : 0u, 0x00020000);
        const int nleft = p.N - n_w0;
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(p.bias ? (const void*)(p.bias + n_w0) : p.zeros), 0, (p.bias && nleft > 0) ? (unsigned)nleft * 4u : 0u, 0x00020000);
        const unsigned offO = (unsigned)(lr * (int)ldo + 4 * lq) * 2u, offR = (unsigned)(lr * (int)p.ldr + 4 * lq) * 2u;
        const unsigned rowO = (unsigned)ldo * 32u, rowR = (unsigned)p.ldr * 32u;       // bytes per 16-row block
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const bool colok = n_w0 + 16 * b + 4 * lq < p.N;
            const u32x4 bcu = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(4 * lq) * 4u + 64u * b, 0, 0);
            u32x2 ru = __builtin_amdgcn_raw_buffer_load_b64(rsR, ((colok && m_w0 + lr < p.M) ? offR : OOB) + 32u * b, 0, 0);
#pragma unroll
            for (int a = 0; a < MB; ++a) {
                const bool ok = colok && m_w0 + a * 16 + lr < p.M;
                float rf[4];
                if (isbf) {
                    rf[0] = __uint_as_float(ru[0] << 16); rf[1] = __uint_as_float(ru[0] & 0xffff0000u);
                    rf[2] = __uint_as_float(ru[1] << 16); rf[3] = __uint_as_float(ru[1] & 0xffff0000u);
                } else {
                    rf[0] = f16_bits_to_f32((unsigned short)(ru[0] & 0xffffu)); rf[1] = f16_bits_to_f32((unsigned short)(ru[0] >> 16));
                    rf[2] = f16_bits_to_f32((unsigned short)(ru[1] & 0xffffu)); rf[3] = f16_bits_to_f32((unsigned short)(ru[1] >> 16));
                }
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = (acc[a][b][j] * p.alpha + __uint_as_float(bcu[j])) + rf[j];
                const u32x2 c = isbf ? u32x2{f32x2_to_bf16x2(e[0], e[1]), f32x2_to_bf16x2(e[2], e[3])} : u32x2{f32x2_to_f16x2(e[0], e[1]), f32x2_to_f16x2(e[2], e[3])};
                if (a + 1 < MB)
                    ru = __builtin_amdgcn_raw_buffer_load_b64(rsR, ((colok && m_w0 + (a + 1) * 16 + lr < p.M) ? offR : OOB) + 32u * b, (a + 1) * rowR, 0);
                __builtin_amdgcn_raw_buffer_store_b64(c, rsO, (ok ? offO + a * rowO : OOB) + 32u * b, 0, 0);
            }
        }
        return;
    }
    if constexpr (!ROWS4)
    if (vec_ok && odt != GEO4D_F32 && geglu) {
        if constexpr (NB % 4 == 0) {
            const bool isbf = odt == GEO4D_BF16;
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((unsigned short*)O + obase + (long)m_w0 * ldo + (n_w0 >> 1)), 0, OOB, 0x00020000);
            const int nleft = p.N - n_w0;
            const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
                uniform_ptr(p.bias ? (const void*)(p.bias + n_w0) : p.zeros), 0, (p.bias && nleft > 0) ? (unsigned)nleft * 4u : 0u, 0x00020000);
            const unsigned offO = (unsigned)(lr * (int)ldo + 4 * lq) * 2u, rowO = (unsigned)ldo * 32u;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if ((b & 3) >= 2) continue;
                const bool grp = n_w0 + 16 * (b & ~3) + 64 <= p.N;
                const u32x4 bv = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(4 * lq) * 4u + 64u * b, 0, 0);
                const u32x4 bg = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(4 * lq) * 4u + 64u * b + 128u, 0, 0);
#pragma unroll
                for (int a = 0; a < MB; ++a) {
                    const bool ok = grp && m_w0 + a * 16 + lr < p.M;
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        e[j] = (acc[a][b][j] * p.alpha + __uint_as_float(bv[j])) * gelu_erf_f(acc[a][b + 2 < NB ? b + 2 : b][j] * p.alpha + __uint_as_float(bg[j]));
                    const u32x2 c = isbf ? u32x2{f32x2_to_bf16x2(e[0], e[1]), f32x2_to_bf16x2(e[2], e[3])} : u32x2{f32x2_to_f16x2(e[0], e[1]), f32x2_to_f16x2(e[2], e[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(c, rsO, (ok ? offO + a * rowO : OOB) + (unsigned)(32 * (b >> 2) + 16 * (b & 1)) * 2u, 0, 0);
                }
            }
        }
        return;
    }
    // ---- generic path (activations, row-bias tables, per-row bias, 16-bit rows, unaligned rows): element by element ----------------
#pragma unroll
    for (int a = 0; a < MB; ++a) {
        const int m = m_w0 + a * 16 + lr;
        if (m >= p.M) continue;
        const float brow = (!partial && p.bias && p.bias_per_row) ? p.bias[m] : 0.f;
        const long rboff = (!partial && p.rowbias) ? (long)(m / p.rowbias_div) * (p.ldrb ? p.ldrb : (long)p.N) : 0;
        if (geglu) {
            if constexpr (NB % 4 == 0) {
                // packed GEGLU weights interleave value / gate in 32-column blocks: 16-blocks 4j, 4j+1 = value, 4j+2, 4j+3 = gate
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if ((b & 3) >= 2) continue;
                    const int n = n_w0 + 16 * b + 4 * lq;                       // value column; its gate sits 32 columns further
                    if (n + 32 >= p.N) continue;
                    const int oc = (n_w0 >> 1) + 32 * (b >> 2) + 16 * (b & 1) + 4 * lq;
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xv = acc[a][b][j] * p.alpha + (p.bias ? p.bias[n + j] : 0.f);
                        const float gv = acc[a][b + 2 < NB ? b + 2 : b][j] * p.alpha + (p.bias ? p.bias[n + 32 + j] : 0.f);
                        e[j] = xv * gelu_erf_f(gv);
                    }
                    const long oidx = obase + (long)m * ldo + oc;
                    if (vec_ok) {
                        if constexpr (OSPLIT) store_split4((float*)O + obase + (long)m * ldo, oc >> 2, e);
                        else if (odt == GEO4D_F32) *(f32x4*)((float*)O + oidx) = f32x4{e[0], e[1], e[2], e[3]};
                        else if (odt == GEO4D_BF16) *(u32x2*)((unsigned short*)O + oidx) = u32x2{f32x2_to_bf16x2(e[0], e[1]), f32x2_to_bf16x2(e[2], e[3])};
                        else *(u32x2*)((unsigned short*)O + oidx) = u32x2{f32x2_to_f16x2(e[0], e[1]), f32x2_to_f16x2(e[2], e[3])};
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) store_out(O, oidx + j, e[j], odt);
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int n = n_w0 + 16 * b + 4 * lq;
            if (n >= p.N) continue;
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = acc[a][b][j];
            const long oidx = obase + (long)m * ldo + n;
            if (!partial) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = e[j] * p.alpha + brow;
                    if (n + j < p.N) {
                        if (p.bias && !p.bias_per_row) v += p.bias[n + j];
                        if (p.rowbias) v += p.rowbias[rboff + n + j];
                    }
                    if (p.act == 1) v = silu_f(v);
                    else if (p.act == 3) v = gelu_erf_f(v);
                    e[j] = v;
                }
            }
            if (vec_ok) {
                if (odt == GEO4D_F32) {
                    if (has_res) {
                        const f32x4 r = *(const f32x4*)((const float*)p.R + rbase + (long)m * p.ldr + n);
#pragma unroll
                        for (int j = 0; j < 4; ++j) e[j] += r[j];
                    }
                    if constexpr (OSPLIT) store_split4((float*)O + obase + (long)m * ldo, n >> 2, e);     // (OSPLIT launches are never partial)
                    else *(f32x4*)((float*)O + oidx) = f32x4{e[0], e[1], e[2], e[3]};
                } else {
                    if (has_res) {
                        const u32x2 r = *(const u32x2*)((const unsigned short*)p.R + rbase + (long)m * p.ldr + n);
                        if (odt == GEO4D_BF16) {
                            e[0] += __uint_as_float(r[0] << 16); e[1] += __uint_as_float(r[0] & 0xffff0000u);
                            e[2] += __uint_as_float(r[1] << 16); e[3] += __uint_as_float(r[1] & 0xffff0000u);
                        } else {
                            e[0] += f16_bits_to_f32((unsigned short)(r[0] & 0xffffu)); e[1] += f16_bits_to_f32((unsigned short)(r[0] >> 16));
                            e[2] += f16_bits_to_f32((unsigned short)(r[1] & 0xffffu)); e[3] += f16_bits_to_f32((unsigned short)(r[1] >> 16));
                        }
                    }
                    if (odt == GEO4D_BF16) *(u32x2*)((unsigned short*)O + oidx) = u32x2{f32x2_to_bf16x2(e[0], e[1]), f32x2_to_bf16x2(e[2], e[3])};
                    else *(u32x2*)((unsigned short*)O + oidx) = u32x2{f32x2_to_f16x2(e[0], e[1]), f32x2_to_f16x2(e[2], e[3])};
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (n + j >= p.N) continue;
                    float v = e[j];
                    if (has_res) v += load_res(p.R, rbase + (long)m * p.ldr + n + j, odt);
                    store_out(O, oidx + j, v, odt);
                }
            }
        }
    }
}

template <int BM, int BN>
constexpr int v2_smem_bytes() { return (2 * BM + 2 * BN) * PITCH + BM * MAXTAP * 4; }

// OSPLIT: the epilogue writes the pre-split operand format (o_split: the GEGLU output feeding ff.net.2, q | k and V^T feeding the
// attention kernel); a separate instantiation (the extra epilogue code costs registers).
// (Round 3 also built a third A-panel buffer - the gathered panel two slabs ahead - and ablation builds of this loop; measured: the
// third buffer never beat two, profiles/r03_gemm_v2_explore_and_ablation.md. Removed from the shipped sources in round 4.)
template <typename T, int BM, int BN, int WM, int WN, int HOT, bool OSPLIT = false>   // HOT: 0 generic, 1 raw A x split W, 2 split A x split W
__global__ __launch_bounds__(WM * WN * 64) void conv_gemm_v2_kernel(const geo4d_conv_gemm_t p, const int splits, const int tiles_mn) {
    constexpr int NT = WM * WN * 64;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = BKC * EPC;
    constexpr int WTM = BM / WM, WTN = BN / WN;       // wave tile
    constexpr int MB = WTM / 16, NB = WTN / 16;       // 16x16 accumulator blocks per wave
    constexpr int RSTEP = NT / 8;                     // panel rows covered by one staging pass of the workgroup
    constexpr bool A64 = IsTwoPass<T>::value;         // activation panel rows = the 4 hi chunks of the slab (64 bytes): 16 rows per 1 KB request
    constexpr int RSTEP_A = A64 ? NT / 4 : RSTEP;
    constexpr int ACH = (BM + RSTEP_A - 1) / RSTEP_A, BCH = (BN + RSTEP - 1) / RSTEP;
    constexpr int RING = (2 * BM + 2 * BN) * PITCH;
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0 && BM % 8 == 0 && BN % 8 == 0, "wave tiles are multiples of 16");
    static_assert(!std::is_same<T, float>::value, "v2 serves the 16-bit MFMA forms (bf16, bf16x3)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* rowpix = (int*)(smem + RING);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int wr = wave / WN, wc = wave % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int ntap = p.KT * p.KH * p.KW;
    const int hw = p.Hout * p.Wout;
    const long total = (long)tiles_mn * p.batch * splits;
    const long G = gridDim.x;
    const bool direct_rows = ntap == 1 && p.stride == 1 && p.ups == 1 && p.ph == 0 && p.pw == 0 && p.pt == 0 && p.Hin * p.Win == hw;
    const int nslab_all = p.K / BK;
    const int per = (nslab_all + splits - 1) / splits;
    const T* __restrict__ Z = (const T*)p.zeros;
    const int ccol = tid & 7;
    const int r0 = tid >> 3;
    const int r0a = A64 ? (tid >> 2) : r0;            // this lane's A-panel row inside a staging pass

    // ---- state of the tile being staged (the NEXT tile while the current one is in its epilogue) ---------------------------------
    int tm = 0, tn = 0, kz = 0, nslab = 0, tap = 0, c0 = 0, tapB = 0, c0B = 0;   // (tap, c0): A-panel cursor; (tapB, c0B): B-panel cursor
    long bz = 0;
    const T* __restrict__ A = nullptr;
    int pix[ACH];
    const T* wptr[BCH];
    int cA[ACH], cB[BCH];                              // source-side swizzle: LDS slot `ccol` of panel row r holds chunk ccol ^ key(r)
#pragma unroll
    for (int i = 0; i < ACH; ++i)
        cA[i] = A64 ? ((tid & 3) ^ swz_key_a64(r0a + i * RSTEP_A)) * 16           // BYTES: K-group g of the slab = 8 consecutive f16 of the plain f16 row
                    : (ccol ^ swz_key<T>(r0 + i * RSTEP)) * EPC;
#pragma unroll
    for (int i = 0; i < BCH; ++i) cB[i] = (ccol ^ swz_key<T>(r0 + i * RSTEP)) * EPC;

    auto fetch_pix = [&]() {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int row = r0a + i * RSTEP_A;
            if (direct_rows) {
                const int m = tm * BM + row;
                pix[i] = (row < BM && m < p.M) ? m : -1;
            } else {
                pix[i] = row < BM ? rowpix[row * ntap + tap] : -1;
            }
        }
    };
    auto setup_tile = [&](long w) {
        const long t = w % tiles_mn, rest = w / tiles_mn;
        bz = rest % p.batch;
        kz = (int)(rest / p.batch);
        tile_of(t, tiles_n, tiles_mn, tile_group_m(p), tm, tn);
        A = A64 ? (const T*)((const char*)p.A + bz * p.a_bs * 2) : (const T*)p.A + bz * p.a_bs;      // (A64: 2-byte elements)
        const T* __restrict__ W = (const T*)p.W + bz * p.w_bs;
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int row = r0 + i * RSTEP, n = tn * BN + row;
            wptr[i] = (row < BN && n < p.N) ? W + (long)n * p.ldw + cB[i] : nullptr;
        }
        const int s_begin = kz * per;
        nslab = min(nslab_all, s_begin + per) - s_begin;
        if (!direct_rows) {
            // gather table: source pixel of (tile row, tap), -1 = zero padding. The previous tile's table is dead here: its last
            // fetch_pix ran before the K loop's final barrier.
            const int hlim = p.ups == 2 ? 2 * p.Hin : p.Hin, wlim = p.ups == 2 ? 2 * p.Win : p.Win;
            const int ush = p.ups == 2 ? 1 : 0;
            for (int e = tid; e < BM * ntap; e += NT) {
                const int row = e / ntap, tp = e - row * ntap;
                const int m = tm * BM + row;
                int px = -1;
                if (m < p.M) {
                    const int f = m / hw, rem = m - f * hw;
                    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                    const int kt = tp / (p.KH * p.KW), r2 = tp - kt * (p.KH * p.KW);
                    const int ky = r2 / p.KW, kx = r2 - ky * p.KW;
                    const int iy = oy * p.stride - p.ph + ky, ix = ox * p.stride - p.pw + kx;
                    const int tt = (f % p.T) + kt - p.pt;
                    if ((unsigned)iy < (unsigned)hlim && (unsigned)ix < (unsigned)wlim && (unsigned)tt < (unsigned)p.T)
                        px = ((f + kt - p.pt) * p.Hin + (iy >> ush)) * p.Win + (ix >> ush);
                }
                rowpix[e] = px;
            }
            __syncthreads();
        }
        tap = tapB = s_begin % ntap;
        c0 = c0B = (s_begin / ntap) * BK;
        fetch_pix();
    };
    // one LDS-DMA per 8 panel rows and wave: wave-uniform destination + lane * 16 B; ragged last passes are skipped per wave.
    // K order: channel-slab major, tap minor (a conv's re-reads of its input hit the XCD's L2).
    auto issue_A = [&](int ab) {
        char* base = smem + ab * BM * PITCH + wave * 1024;
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            // (A64: a wave's request covers 16 rows of 64 bytes - the same 1 KB at the same byte offsets, RSTEP * PITCH == RSTEP_A * 64)
            if ((ACH * RSTEP_A == BM) || (wave * (A64 ? 16 : 8) + j * RSTEP_A < BM)) {
                const T* src = pix[j] < 0 ? Z : A64 ? (const T*)((const char*)A + ((long)pix[j] * p.lda + c0) * 2 + cA[j]) : A + (long)pix[j] * p.lda + c0 + cA[j];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + j * RSTEP * PITCH), 16, 0, 0);
            }
        }
        if (++tap == ntap) { tap = 0; c0 += BK; }
        if (ntap > 1 && c0 < p.Cin) fetch_pix();
    };
    auto issue_B = [&](int bb) {
        char* base = smem + (2 * BM + bb * BN) * PITCH + wave * 1024;
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            if ((BCH * RSTEP == BN) || (wave * 8 + i * RSTEP < BN)) {
                const T* src = wptr[i] ? wptr[i] + (long)tapB * p.Cin + c0B : Z;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + i * RSTEP * PITCH), 16, 0, 0);
            }
        }
        if (++tapB == ntap) { tapB = 0; c0B += BK; }
    };
    f32x4 acc[MB][NB];
    // fragment offsets inside a 16-row block: lane (lr, lq) reads row lr; 16-bit types: chunk 4h + lq of half h; bf16x3: chunks 2lq, 2lq + 1
    const int fkey = swz_key<T>(lr);
    int foff[2];
    if constexpr (IsX3<T>::value) {
        foff[0] = lr * PITCH + (((2 * lq) ^ fkey) << 4);
        foff[1] = lr * PITCH + (((2 * lq + 1) ^ fkey) << 4);
    } else {
        foff[0] = lr * PITCH + ((lq ^ fkey) << 4);
        foff[1] = lr * PITCH + (((4 + lq) ^ fkey) << 4);
    }
    const bool a_split = HOT ? (HOT == 2) : (p.a_split != 0), w_split = HOT ? true : (p.w_split != 0);
    auto compute_slab = [&](int buf) {
        const char* abase = smem + (buf * BM + wr * WTM) * PITCH;
        const char* bbase = smem + (2 * BM + buf * BN + wc * WTN) * PITCH;
        if constexpr (IsTwoPass<T>::value) {
            // f16x2: pre-split weights x plain f16 activation rows (A64 panel: 4 K-groups of 8 f16 per row). (Round 5 also built the form that converts a RAW f32
            // activation in registers - residual streams: down / up samplers, skip connections, proj_out, the VAE's upsamplers - and removed
            // it again: rounding a STREAM to f16 took the 50-step point-map drift from 1.1e-4 to 5.3e-4 for +1 % frames/s)
            u32x4 ah[MB];
            const char* abase64 = smem + buf * BM * PITCH + (wr * WTM) * PITCH_A64 + lr * PITCH_A64 + ((lq ^ swz_key_a64(lr)) << 4);
#pragma unroll
            for (int a = 0; a < MB; ++a) ah[a] = *(const u32x4*)(abase64 + a * 16 * PITCH_A64);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const u32x4 bh = *(const u32x4*)(bbase + b * 16 * PITCH + foff[0]);
                const u32x4 bl = *(const u32x4*)(bbase + b * 16 * PITCH + foff[1]);
#pragma unroll
                for (int a = 0; a < MB; ++a) mma16_x2(acc[a][b], bh, bl, ah[a]);      // C rows = n, C cols = m
            }
        } else if constexpr (IsX3<T>::value) {
            // one 128-byte slab = 32 k = ONE 16x16x32 step; the activation fragments are split once and reused by every column block
            u32x4 ah[MB], al[MB];
#pragma unroll
            for (int a = 0; a < MB; ++a) {
                const u32x4 x0 = *(const u32x4*)(abase + a * 16 * PITCH + foff[0]);
                const u32x4 x1 = *(const u32x4*)(abase + a * 16 * PITCH + foff[1]);
                if (a_split) { ah[a] = x0; al[a] = x1; }
                else split8_bf16(x0, x1, ah[a], al[a]);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const u32x4 y0 = *(const u32x4*)(bbase + b * 16 * PITCH + foff[0]);
                const u32x4 y1 = *(const u32x4*)(bbase + b * 16 * PITCH + foff[1]);
                u32x4 bh, bl;
                if (w_split) { bh = y0; bl = y1; }
                else split8_bf16(y0, y1, bh, bl);
#pragma unroll
                for (int a = 0; a < MB; ++a) mma16_x3(acc[a][b], bh, bl, ah[a], al[a]);      // C rows = n, C cols = m
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 fa[MB];
#pragma unroll
                for (int a = 0; a < MB; ++a) fa[a] = *(const u32x4*)(abase + a * 16 * PITCH + foff[h]);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const u32x4 fb = *(const u32x4*)(bbase + b * 16 * PITCH + foff[h]);
#pragma unroll
                    for (int a = 0; a < MB; ++a) mma16<T>(acc[a][b], fb, fa[a]);
                }
            }
        }
    };

    const bool partial = splits > 1;                  // split-K: raw fp32 slab, the epilogue runs in the reduce kernel
    auto epilogue = [&](int e_tm, int e_tn, long e_bz, int e_kz) {
        reg_epilogue<MB, NB, OSPLIT, false, IsX3<T>::value, IsTwoPass<T>::value>(p, acc, e_tm * BM + wr * WTM, e_tn * BN + wc * WTN, e_bz, e_kz, partial, lr, lq);
    };

    // ---- persistent tile loop --------------------------------------------------------------------------------------------------------
    long w = xcd_remap((long)blockIdx.x, G);           // each XCD walks a contiguous range of every round of G tiles
    if (w >= total) return;
    setup_tile(w);
    auto prologue = [&]() {                                 // the tile's first slab
        if (nslab > 0) { issue_A(0); issue_B(0); }
    };
    prologue();
    while (true) {
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ns = nslab;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA is drained explicitly before every barrier
        __syncthreads();
        for (int s = 0; s < ns; ++s) {
            const int buf = s & 1;
            if (s + 1 < ns) { issue_A(buf ^ 1); issue_B(buf ^ 1); }
            compute_slab(buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                               // next slab landed for every wave, everyone is done reading this one
        }
        const int e_tm = tm, e_tn = tn, e_kz = kz;
        const long e_bz = bz;
        const long wn = w + G;
        const bool more = wn < total;
        if (more) {                                        // the next tile's table + first slab go out BEFORE this tile's epilogue
            setup_tile(wn);
            prologue();
        }
        epilogue(e_tm, e_tn, e_bz, e_kz);
        if (!more) break;
        w = wn;
    }
}

// resident workgroups per launch = CUs x workgroups that fit a CU (queried once per kernel instantiation)
template <typename T, int BM, int BN, int WM, int WN, int HOT, bool OSPLIT = false>
int launch_v2_kernel(const geo4d_conv_gemm_t& p, int splits, hipStream_t stream) {
    constexpr int smem = v2_smem_bytes<BM, BN>();
    static_assert(smem <= 160 * 1024, "LDS");
    static int resident = 0;
    auto kern = conv_gemm_v2_kernel<T, BM, BN, WM, WN, HOT, OSPLIT>;
    if (!resident) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
            geo4d_set_error("hipFuncSetAttribute(max dynamic LDS) failed");
            return GEO4D_EIO;
        }
        int dev = 0, cus = 0, occ = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, WM * WN * 64, smem) != hipSuccess || cus <= 0 || occ <= 0) {
            geo4d_set_error("conv_gemm v2: occupancy query failed");
            return GEO4D_EIO;
        }
        resident = cus * occ;
    }
    const int tiles_mn = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const long total = (long)tiles_mn * p.batch * splits;
    // debug_ablate = 2 (tests only): 3 workgroups whatever the problem, so that small shapes walk the persistent tile loop
    const long cap = p.debug_ablate == 2 ? 3 : resident;
    const unsigned grid = (unsigned)(total < cap ? total : cap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), smem, stream, p, splits, tiles_mn);
    GEO4D_CHECK_LAUNCH();
    if (splits > 1) return launch_splitk_reduce<T>(p, splits, stream);
    return GEO4D_OK;
}

// o_split (bf16x3, pre-split x pre-split operands only): the output in the producers' pre-split format - needs whole 8-column groups
// and 16-byte aligned rows, has no split-K (the reduce kernel writes plain f32) and no residual-free restrictions otherwise
inline bool o_split_ok(const geo4d_conv_gemm_t& p, int splits) {
    const long nout = p.act == 2 ? (p.N >> 1) : p.N;
    return splits == 1 && p.w_split && p.a_split && (nout & 7) == 0 && (p.ldo & 7) == 0 && ((uintptr_t)p.O % 32) == 0 &&
           (p.batch == 1 || (p.o_bs & 7) == 0) && (!p.R || ((p.ldr & 3) == 0 && ((uintptr_t)p.R % 16) == 0 && (p.batch == 1 || (p.r_bs & 3) == 0)));
}

// o_split = 2 (the two-pass f16 type): plain f16 rows out - column bias + alpha (+ GEGLU) only, whole 8-column groups, 16-byte aligned rows
inline bool o_f16_ok(const geo4d_conv_gemm_t& p, int splits) {
    const long nout = p.act == 2 ? (p.N >> 1) : p.N;
    return splits == 1 && p.w_split && p.a_split == 2 && (p.act == 0 || p.act == 2) && !p.R && !p.rowbias && !p.bias_per_row && (nout & 7) == 0 && (p.ldo & 7) == 0 &&
           ((uintptr_t)p.O % 16) == 0 && (p.batch == 1 || (p.o_bs & 7) == 0);
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_v2_cfg(const geo4d_conv_gemm_t& p, int splits, hipStream_t stream) {
    if (p.act == 2 && ((BN / WN / 16) % 4)) {
        geo4d_set_error("conv_gemm v2: GEGLU needs wave tiles that are a multiple of 64 columns wide");
        return GEO4D_EINVAL;
    }
    if (p.o_split && !IsX3<T>::value) { geo4d_set_error("conv_gemm: o_split is a bf16x3 option"); return GEO4D_EINVAL; }
    if constexpr (IsTwoPass<T>::value) {       // f16x2: pre-split x pre-split; plain f32 rows out, or (o_split = 2) the f16 pre-split format
        if (p.a_split != 2 || !p.w_split || (p.o_split && p.o_split != 2)) { geo4d_set_error("conv_gemm: f16x2 (dtype 4) takes plain f16 activation rows (a_split = 2) and a pre-split f16 weight; o_split 0 or 2 (plain f16 rows out)"); return GEO4D_EINVAL; }
        if (p.o_split) {
            // (the f16-row epilogue: every tile for plain rows - q | k, q | k | v, cross-attention q -, the GEGLU form on the tiles whose wave
            // tiles are a multiple of 64 columns wide: checked at the top of this function)
            if (o_f16_ok(p, splits)) return launch_v2_kernel<T, BM, BN, WM, WN, 2, true>(p, splits, stream);
            geo4d_set_error("conv_gemm: o_split = 2 (plain f16 rows out) needs no split-K / residual / row biases / SiLU / GELU, stored columns % 8 == 0 and 16-byte aligned output rows");
            return GEO4D_EINVAL;
        }
        return launch_v2_kernel<T, BM, BN, WM, WN, 2>(p, splits, stream);
    } else {
    if constexpr (IsX3<T>::value) {
        if (p.o_split) {
            if (o_split_ok(p, splits)) return launch_v2_kernel<T, BM, BN, WM, WN, 2, true>(p, splits, stream);
            geo4d_set_error("conv_gemm: o_split needs pre-split x pre-split operands, no split-K, N % 8 == 0 and 32-byte aligned output rows");
            return GEO4D_EINVAL;
        }
        if (p.w_split && !p.a_split) return launch_v2_kernel<T, BM, BN, WM, WN, 1>(p, splits, stream);
        if (p.w_split && p.a_split) return launch_v2_kernel<T, BM, BN, WM, WN, 2>(p, splits, stream);
    }
    return launch_v2_kernel<T, BM, BN, WM, WN, 0>(p, splits, stream);
    }
}

// tile hints of the second generation (16x16x32 MFMA, register epilogue, persistent workgroups). Round 4 kept the five the measured
// table selects (profiles/r04_gemm_census_bf16x3.log); 21 / 24 / 26 / 29 and the three-A-buffer twins 31..39 are gone:
//   22: 256x256, 8 waves (64x128 wave tiles)    23: 160x320, 8 waves (80x80)    25: 128x128, 4 waves (64x64)
//   27: 64x128, 4 waves (32x64)                 28: 64x64, 4 waves (32x32)
template <typename T>
int launch_v2_typed(const geo4d_conv_gemm_t& p_in, hipStream_t stream) {
    geo4d_conv_gemm_t p = p_in;
    p.tile_hint = v2_effective_hint<T>(p_in.tile_hint, p_in.act, p_in.o_split);
    if constexpr (std::is_same<T, float>::value || std::is_same<T, f16_t>::value) {
        geo4d_set_error("conv_gemm: tile hints 22..28 serve bf16 / bf16x3 (the exact-f32 and the f16 modes stay on hints 0..17)");
        return GEO4D_EINVAL;
    } else {
        if (p.out_nchw) {
            geo4d_set_error("conv_gemm: tile hints 22..28 have no NCTHW epilogue");
            return GEO4D_EINVAL;
        }
        int sp = 1;
        if (p.split_k > 1) {
            if (!p.workspace || p.act == 2 || (p.N % 8) || (size_t)p.split_k * p.batch * p.M * p.N * 4 > p.workspace_bytes ||
                p.K / (BKC * Elem<T>::EPC) / p.split_k < 1) {
                geo4d_set_error("conv_gemm: split_k not applicable (workspace too small / epilogue not splittable)");
                return GEO4D_EINVAL;
            }
            sp = p.split_k;
        }
        if (p.gn_colsum && sp > 1) {      // split-K: the sums come from the reduce launch (gemm_kernel.h splitk_reduce_colsum_kernel)
            if (!splitk_colsum_rows(p, sp)) { geo4d_set_error("conv_gemm: this split-K launch cannot emit gn_colsum (geo4d_conv_gemm_colsum_rows)"); return GEO4D_EINVAL; }
        } else
        if (p.gn_colsum && (!colsum_fast_ok(p, sp) || v2_wave_rows(p.tile_hint) == 0 || p.M % v2_wave_rows(p.tile_hint) || ((uintptr_t)p.gn_colsum % 16))) {
            geo4d_set_error("conv_gemm: gn_colsum on tile hints 22..28 needs the plain f32-row epilogue (no activation / split-K / o_split / batch) and M % wave-tile rows == 0 (geo4d_conv_gemm_colsum_rows)");
            return GEO4D_EINVAL;
        }
        switch (p.tile_hint) {
            case 22:
                if constexpr (IsTwoPass<T>::value) return GEO4D_EINVAL;      // (unreachable: v2_effective_hint)
                else return launch_v2_cfg<T, 256, 256, 4, 2>(p, sp, stream);
            case 23: return launch_v2_cfg<T, 160, 320, 2, 4>(p, sp, stream);
            case 25: return launch_v2_cfg<T, 128, 128, 2, 2>(p, sp, stream);
            case 27: return launch_v2_cfg<T, 64, 128, 2, 2>(p, sp, stream);
            case 28: return launch_v2_cfg<T, 64, 64, 2, 2>(p, sp, stream);
        }
        geo4d_set_error("conv_gemm: unknown tile_hint");
        return GEO4D_EINVAL;
    }
}

}  // namespace geo4d_gemm
