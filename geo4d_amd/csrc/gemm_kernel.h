// geo4d_amd/csrc/gemm_kernel.h — the implicit-GEMM kernel template + tile dispatch (see gemm.hip for the design notes).
// Included by gemm.hip (C ABI + validation) and by gemm_{bf16,f16,f32}.hip, each of which instantiates launch_typed<T>
// for ONE element type so the three sets of kernels compile in parallel.
#pragma once
#include <type_traits>
#include "common.h"
#include "geo4d_hip.h"

namespace geo4d_gemm {


constexpr int PITCH = 128;  // LDS row pitch in bytes: 8 x 16-byte slots, XOR-swizzled
constexpr int BKC = 8;      // 16-byte chunks per row per stage
constexpr int MAXTAP = 9;

__device__ __forceinline__ void store_out(void* O, long idx, float v, int dt) {
    if (dt == GEO4D_F32) ((float*)O)[idx] = v;
    else if (dt == GEO4D_BF16) ((unsigned short*)O)[idx] = f32_to_bf16_bits(v);
    else ((unsigned short*)O)[idx] = f32_to_f16_bits(v);
}
__device__ __forceinline__ float load_res(const void* R, long idx, int dt) {
    if (dt == GEO4D_F32) return ((const float*)R)[idx];
    if (dt == GEO4D_BF16) return bf16_bits_to_f32(((const unsigned short*)R)[idx]);
    return f16_bits_to_f32(((const unsigned short*)R)[idx]);
}

template <int BM, int BN, int WM, int WN, int ST>
constexpr int stage_bytes() {
    constexpr int ring = ST * (BM + BN) * PITCH;
    constexpr int epi = WM * WN * 32 * ((((BN / WN / 32) % 2 == 0) ? 64 : 32) + 4) * 4;   // one fp32 32 x CG transpose block per wave
    return ring > epi ? ring : epi;
}
template <int BM, int BN, int WM, int WN, int ST>
constexpr int smem_bytes() {
    return stage_bytes<BM, BN, WM, WN, ST>() + BM * MAXTAP * 4;
}

// HOT (bf16x3 only): the layout of the two operands is fixed at compile time to the one every conv / linear of the networks uses
// (raw f32 activations, pre-split weights), so the fragment loop carries no per-fragment branch on the layout flags — measured:
// uniform branches inside this K loop cost ~10 % (profiles/r02_gemm_x3.md). The generic build serves linear_t / batched GEMMs.
template <typename T, int BM, int BN, int WM, int WN, int ST, int HOT = 0>   // HOT: 0 generic, 1 = raw A x split W, 2 = split A x split W
__global__ __launch_bounds__(WM * WN * 64) void conv_gemm_kernel(const geo4d_conv_gemm_t p) {
    constexpr int NT = WM * WN * 64;              // 256 threads (4 waves) or 512 (8 waves: one 256x128 tile per CU)
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = BKC * EPC;
    constexpr int MB = BM / WM / 32, NB = BN / WN / 32;
    constexpr int ACH = BM * BKC / NT, BCH = BN * BKC / NT;
    constexpr int RSTEP = NT / 8;                 // rows covered by one staging pass of the whole workgroup
    constexpr int WTM = MB * 32, WTN = NB * 32;   // wave tile
    constexpr int CG = (NB % 2 == 0) ? 64 : 32;   // columns staged per epilogue pass
    constexpr int NG = WTN / CG;
    constexpr int SP = CG + 4;                    // fp32 staging pitch (floats)
    static_assert(WM * WN == 4 || WM * WN == 5 || WM * WN == 8 || WM * WN == 10, "4, 5, 8 or 10 waves");
    static_assert((BM * BKC) % NT == 0 && (BN * BKC) % NT == 0 && (NT / 8) % 2 == 0, "staging passes must tile the panel");
    static_assert(MB >= 1 && NB >= 1 && ACH >= 1 && BCH >= 1, "tile too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* rowpix = (int*)(smem + stage_bytes<BM, BN, WM, WN, ST>());

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, g = lane >> 5;
    const int wr = wave / WN, wc = wave % WN;
    const long lid = xcd_remap((long)blockIdx.x, (long)gridDim.x);
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tm = (int)(lid / tiles_n), tn = (int)(lid - (long)tm * tiles_n);
    const long bz = blockIdx.y;
    const int kz = blockIdx.z;
    const T* __restrict__ A = (const T*)p.A + bz * p.a_bs;
    const T* __restrict__ W = (const T*)p.W + bz * p.w_bs;
    const int ntap = p.KT * p.KH * p.KW;
    const int hw = p.Hout * p.Wout;

    // ---- gather table: source pixel index of (tile row, tap), -1 where the tap falls into padding ------
    // Plain linears / batched GEMMs (1 tap, no stride / upsample, same pixel count in and out: source row == output row) skip
    // the table, its integer divides and its barrier: their rows are resolved straight into registers below.
    const bool direct_rows = ntap == 1 && p.stride == 1 && p.ups == 1 && p.ph == 0 && p.pw == 0 && p.pt == 0 &&
                             p.Hin * p.Win == hw;
    if (!direct_rows) {
        const int hlim = p.ups == 2 ? 2 * p.Hin : p.Hin, wlim = p.ups == 2 ? 2 * p.Win : p.Win;
        const int ush = p.ups == 2 ? 1 : 0;
        for (int e = tid; e < BM * ntap; e += NT) {
            const int row = e / ntap, tap = e - row * ntap;
            const int m = tm * BM + row;
            int pix = -1;
            if (m < p.M) {
                const int f = m / hw, rem = m - f * hw;
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                const int kt = tap / (p.KH * p.KW), r2 = tap - kt * (p.KH * p.KW);
                const int ky = r2 / p.KW, kx = r2 - ky * p.KW;
                const int iy = oy * p.stride - p.ph + ky, ix = ox * p.stride - p.pw + kx;
                const int tt = (f % p.T) + kt - p.pt;
                if ((unsigned)iy < (unsigned)hlim && (unsigned)ix < (unsigned)wlim && (unsigned)tt < (unsigned)p.T)
                    pix = ((f + kt - p.pt) * p.Hin + (iy >> ush)) * p.Win + (ix >> ush);
            }
            rowpix[e] = pix;
        }
        __syncthreads();
    }

    const int ccol = tid & 7;
    const int r0 = tid >> 3;
    // source-side swizzle: LDS slot `ccol` of row r holds global chunk ccol ^ ((r >> 1) & 7). Rows advance by RSTEP per staging
    // pass; when RSTEP / 2 is a multiple of 8 (4, 8, 10 waves) the XOR term is a per-thread constant, otherwise (5 waves) it
    // alternates with the pass index, which is a compile-time constant of the unrolled issue loop.
    const int csrc = (ccol ^ ((r0 >> 1) & 7)) * EPC;
    constexpr bool SWZ_CONST = ((RSTEP / 2) % 8) == 0;
    auto csrc_of = [&](int i) { return SWZ_CONST ? csrc : (ccol ^ (((r0 + i * RSTEP) >> 1) & 7)) * EPC; };
    const T* __restrict__ Z = (const T*)p.zeros;
    const T* wptr[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int n = tn * BN + r0 + i * RSTEP;
        wptr[i] = n < p.N ? W + (long)n * p.ldw + csrc_of(i) : nullptr;
    }

    // K range of this workgroup (split-K over gridDim.z)
    const int nslab_all = p.K / BK;
    const int per = (nslab_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int s_begin = kz * per;
    const int s_end = min(nslab_all, s_begin + per);
    const int nslab = s_end - s_begin;

    // K order is channel-slab major, TAP MINOR: consecutive stages touch the same input rows shifted by one pixel / row /
    // frame, so the re-reads of a 3x3 (or 3-tap temporal) gather hit the XCD's L2 instead of going back to Infinity Cache / HBM
    // (tap-major order re-touched a line only after a full sweep over Cin: 42 GB of L2-miss traffic per U-Net forward).
    int pix[ACH];
    int tap = s_begin % ntap;
    int c0 = (s_begin / ntap) * BK;
    auto fetch_pix = [&]() {
        if (direct_rows) {
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int m = tm * BM + r0 + i * RSTEP;
                pix[i] = m < p.M ? m : -1;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < ACH; ++i) pix[i] = rowpix[(r0 + i * RSTEP) * ntap + tap];
    };
    // one LDS-DMA per 8 rows: wave-uniform destination (M0) + lane * 16 B. A stage = ACH + BCH pieces per wave, issued
    // back-to-back at the top of the stage (A/B-tested against spreading them between the MFMA groups: 4-10 % slower).
    constexpr int NP = ACH + BCH;
    auto issue_piece = [&](int buf, int j) {
        char* base = smem + buf * (BM + BN) * PITCH + wave * 1024;
        if (j < ACH) {
            const T* src = pix[j] >= 0 ? A + (long)pix[j] * p.lda + c0 + csrc_of(j) : Z;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + j * RSTEP * PITCH), 16, 0, 0);
        } else {
            const int i = j - ACH;
            const T* src = wptr[i] ? wptr[i] + (long)tap * p.Cin + c0 : Z;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + (BM + i * RSTEP) * PITCH), 16, 0, 0);
        }
    };
    auto advance_cursor = [&]() {
        if (++tap == ntap) { tap = 0; c0 += BK; }
        if (ntap > 1 && c0 < p.Cin) fetch_pix();   // linear layers (1 tap) keep their row table entries in registers
    };
    auto issue_slab = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NP; ++j) issue_piece(buf, j);
        advance_cursor();
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment read offsets: logical chunk (2kk + g) of row li lives in slot (2kk + g) ^ ((li >> 1) & 7)
    int foff[BKC / 2];
#pragma unroll
    for (int kk = 0; kk < BKC / 2; ++kk) foff[kk] = li * PITCH + (((2 * kk + g) ^ ((li >> 1) & 7)) << 4);

    // one K slab of MFMAs out of ring buffer `buf`. Small wave tiles double-buffer their fragments in registers (the ds_reads
    // of step kk+1 are in flight under the MFMAs of step kk); the 64x128 wave tile of the 256x256 configuration has no VGPRs
    // to spare for that and relies on its second wave per SIMD instead.
    // bf16x3: a 128-byte slab holds 32 K-elements = two MFMA k-steps of 16; lane (li, g) owns elements 16s + 8g .. +8 of its
    // row = chunks 4s + 2g and 4s + 2g + 1 (raw f32 -> split in registers; pre-split weights: chunk 0 = hi, chunk 1 = lo)
    int foffx[2][2];
    if constexpr (IsX3<T>::value) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 2; ++j) foffx[s2][j] = li * PITCH + (((4 * s2 + 2 * g + j) ^ ((li >> 1) & 7)) << 4);
    }
    const bool a_split = HOT ? (HOT == 2) : (p.a_split != 0), w_split = HOT ? true : (p.w_split != 0);
    auto compute_slab = [&](int buf) {
        const char* abase = smem + buf * (BM + BN) * PITCH + (wr * WTM) * PITCH;
        const char* bbase = smem + buf * (BM + BN) * PITCH + (BM + wc * WTN) * PITCH;
        if constexpr (IsX3<T>::value) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 ah[MB], al[MB], bh[NB], bl[NB];
#pragma unroll
                for (int a = 0; a < MB; ++a) {
                    const u32x4 c0 = *(const u32x4*)(abase + a * 32 * PITCH + foffx[s2][0]);
                    const u32x4 c1 = *(const u32x4*)(abase + a * 32 * PITCH + foffx[s2][1]);
                    if (a_split) { ah[a] = c0; al[a] = c1; }
                    else split8_bf16(c0, c1, ah[a], al[a]);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const u32x4 c0 = *(const u32x4*)(bbase + b * 32 * PITCH + foffx[s2][0]);
                    const u32x4 c1 = *(const u32x4*)(bbase + b * 32 * PITCH + foffx[s2][1]);
                    if (w_split) { bh[b] = c0; bl[b] = c1; }
                    else split8_bf16(c0, c1, bh[b], bl[b]);
                }
#pragma unroll
                for (int a = 0; a < MB; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b) mma_x3(acc[a][b], bh[b], bl[b], ah[a], al[a]);   // C rows = n, C cols = m
            }
        } else if constexpr (MB * NB <= 4) {
            u32x4 fa[2][MB], fb[2][NB];
#pragma unroll
            for (int a = 0; a < MB; ++a) fa[0][a] = *(const u32x4*)(abase + a * 32 * PITCH + foff[0]);
#pragma unroll
            for (int b = 0; b < NB; ++b) fb[0][b] = *(const u32x4*)(bbase + b * 32 * PITCH + foff[0]);
#pragma unroll
            for (int kk = 0; kk < BKC / 2; ++kk) {
                if (kk + 1 < BKC / 2) {
#pragma unroll
                    for (int a = 0; a < MB; ++a) fa[(kk + 1) & 1][a] = *(const u32x4*)(abase + a * 32 * PITCH + foff[kk + 1]);
#pragma unroll
                    for (int b = 0; b < NB; ++b) fb[(kk + 1) & 1][b] = *(const u32x4*)(bbase + b * 32 * PITCH + foff[kk + 1]);
                }
#pragma unroll
                for (int a = 0; a < MB; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b) cmma<T>(acc[a][b], fb[kk & 1][b], fa[kk & 1][a]);   // C rows = n, C cols = m
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BKC / 2; ++kk) {
                u32x4 fa[MB], fb[NB];
#pragma unroll
                for (int a = 0; a < MB; ++a) fa[a] = *(const u32x4*)(abase + a * 32 * PITCH + foff[kk]);
#pragma unroll
                for (int b = 0; b < NB; ++b) fb[b] = *(const u32x4*)(bbase + b * 32 * PITCH + foff[kk]);
#pragma unroll
                for (int a = 0; a < MB; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b) cmma<T>(acc[a][b], fb[b], fa[a]);
            }
        }
    };

    static_assert(ST == 2, "two ring stages (deeper rings never beat their 2-stage twins: profiles/r01_gemm_tiles.md; removed in round 4)");
    if (nslab > 0) {
        fetch_pix();
        issue_slab(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA must be drained explicitly before the barrier
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) issue_slab(buf ^ 1);   // all pieces up front: measured faster than spreading them over the MFMA groups
        compute_slab(buf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next stage landed (explicit: never rely on hipcc for DMA)
        __syncthreads();                                    // ... for every wave, and everyone is done reading this one
    }

    // ---- epilogue ------------------------------------------------------------------------------------------
    // acc[a][b][r]: m = m_w0 + a*32 + li ; n = n_w0 + b*32 + 8*(r>>2) + 4*g + (r&3)
    const int m_w0 = tm * BM + wr * WTM;
    const int n_w0 = tn * BN + wc * WTN;
    const bool partial = gridDim.z > 1;          // split-K: raw fp32 slab, epilogue runs in the reduce kernel
    const int odt = partial ? GEO4D_F32 : p.out_dtype;
    void* O = partial ? (void*)((float*)p.workspace + ((long)kz * p.batch + bz) * (long)p.M * p.N) : p.O;
    const long ldo = partial ? (long)p.N : p.ldo;
    const long obase = partial ? 0 : bz * p.o_bs;
    const bool geglu = !partial && p.act == 2;
    const bool gn_on = !partial && p.gn_colsum != nullptr;
    const int oesz = odt == GEO4D_F32 ? 4 : 2;
    const int nout = geglu ? (p.N >> 1) : p.N;
    const bool vec_ok = !p.out_nchw && ((ldo * oesz) & 15) == 0 && (nout & 7) == 0 && (((uintptr_t)O + obase * oesz) & 15) == 0 &&
                        (partial || !p.R || (((p.ldr * oesz) & 15) == 0 && (((uintptr_t)p.R + bz * p.r_bs * oesz) & 15) == 0));

    if (!vec_ok) {
        // direct path (NCTHW heads with N = 16 / 3 / 1, odd shapes): lanes run along m -> coalesced along pixels
#pragma unroll
        for (int a = 0; a < MB; ++a) {
            const int m = m_w0 + a * 32 + li;
            if (m >= p.M) continue;
            long orow, ocol;
            if (p.out_nchw && !partial) {
                const int f = m / hw;
                const int bb = f / p.T, tt = f - bb * p.T;
                orow = ((long)bb * p.ldo * p.T + tt) * hw + (m - f * hw);   // ldo = channels of the NCTHW tensor
                ocol = (long)p.T * hw;
            } else {
                orow = (long)m * ldo;
                ocol = 1;
            }
            const float brow = (!partial && p.bias && p.bias_per_row) ? p.bias[m] : 0.f;
            const long rboff = (!partial && p.rowbias) ? (long)(m / p.rowbias_div) * (p.ldrb ? p.ldrb : (long)p.N) : 0;
            if (geglu) {
                if constexpr (NB % 2 == 0) {
#pragma unroll
                    for (int b2 = 0; b2 < NB / 2; ++b2)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int nn = n_w0 + 64 * b2 + 8 * (r >> 2) + 4 * g + (r & 3);
                            if (nn + 32 >= p.N) continue;
                            const float xv = acc[a][2 * b2][r] * p.alpha + (p.bias ? p.bias[nn] : 0.f);
                            const float gv = acc[a][2 * b2 + 1][r] * p.alpha + (p.bias ? p.bias[nn + 32] : 0.f);
                            store_out(O, obase + orow + (long)((n_w0 >> 1) + 32 * b2 + 8 * (r >> 2) + 4 * g + (r & 3)) * ocol, xv * gelu_erf_f(gv), odt);
                        }
                }
                continue;
            }
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n_w0 + b * 32 + 8 * (r >> 2) + 4 * g + (r & 3);
                    if (n >= p.N) continue;
                    float v = acc[a][b][r];
                    if (!partial) {
                        v = v * p.alpha + brow;
                        if (p.bias && !p.bias_per_row) v += p.bias[n];
                        if (p.rowbias) v += p.rowbias[rboff + n];
                        if (p.act == 1) v = silu_f(v);
                        if (p.R) v += load_res(p.R, bz * p.r_bs + (long)m * p.ldr + n, odt);
                    }
                    store_out(O, obase + orow + (long)n * ocol, v, odt);
                }
        }
        return;
    }

    // staged path: registers -> fp32 LDS block [32 m][CG n] private to the wave -> coalesced 16-byte rows. One 32-row block of
    // the wave tile and one group of CG columns (64, or 32 when the wave tile is an odd number of 32-column blocks wide) at a
    // time, so the staging area stays 32 x (CG + 4) floats per wave whatever the tile (256x256 and 160x320 tiles fit).
    float* stg = (float*)smem + wave * (32 * SP);
    const int ocol_w0 = geglu ? (n_w0 >> 1) : n_w0;
    constexpr int BPG = CG / 32;                              // 32-column accumulator blocks per group
    const int gcols = geglu ? CG / 2 : CG;                   // staged (= stored) columns per group
    const int cpr = gcols / 8;                                // 8-element chunks per staged row (8, 4 or 2)
    const int rows_per_pass = 64 / cpr;
    const int lc = lane % cpr, lr = lane / cpr;
#pragma unroll
    for (int a = 0; a < MB; ++a) {
        const int m = m_w0 + a * 32 + li;
        const float brow = (!partial && p.bias && p.bias_per_row && m < p.M) ? p.bias[m] : 0.f;
        const long rboff = (!partial && p.rowbias && m < p.M) ? (long)(m / p.rowbias_div) * (p.ldrb ? p.ldrb : (long)p.N) : 0;
#pragma unroll
        for (int cg = 0; cg < NG; ++cg) {
            // the residual chunks this lane will add in the read-back below are requested BEFORE the block is staged, so the global
            // round trip (a workgroup's epilogue overlaps with nothing: one workgroup per CU) runs under the 16 LDS writes and the
            // fences instead of once per read-back row - measured: tail + epilogue were 12 % of a long-K workgroup's life
            u32x4 rpre[4][2];
            const bool has_res = !partial && p.R != nullptr;
            if (has_res) {
                const int n = ocol_w0 + cg * gcols + lc * 8;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int rr = lr + k * rows_per_pass;
                    const int mo = m_w0 + a * 32 + rr;
                    if (rr < 32 && mo < p.M && n < nout) {
                        if (odt == GEO4D_F32) {
                            const float* rp = (const float*)p.R + bz * p.r_bs + (long)mo * p.ldr + n;
                            rpre[k][0] = *(const u32x4*)rp;
                            rpre[k][1] = *(const u32x4*)(rp + 4);
                        } else {
                            rpre[k][0] = *(const u32x4*)((const unsigned short*)p.R + bz * p.r_bs + (long)mo * p.ldr + n);
                        }
                    }
                }
            }
            if (geglu) {
                if constexpr (BPG == 2) {
                    // packed GEGLU weights interleave value / gate in 32-column blocks: block 2j = value, 2j + 1 = gate
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nn = n_w0 + 64 * cg + 8 * q + 4 * g;
                        f32x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool ok = nn + 32 + j < p.N;
                            const float xv = acc[a][2 * cg][4 * q + j] * p.alpha + ((p.bias && ok) ? p.bias[nn + j] : 0.f);
                            const float gv = acc[a][2 * cg + 1][4 * q + j] * p.alpha + ((p.bias && ok) ? p.bias[nn + 32 + j] : 0.f);
                            o[j] = xv * gelu_erf_f(gv);
                        }
                        *(f32x4*)(stg + li * SP + 8 * q + 4 * g) = o;
                    }
                }
            } else {
#pragma unroll
                for (int bb = 0; bb < BPG; ++bb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int b = cg * BPG + bb;
                        const int n = n_w0 + b * 32 + 8 * q + 4 * g;
                        f32x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = acc[a][b][4 * q + j];
                            if (!partial) {
                                v = v * p.alpha + brow;
                                if (n + j < p.N) {
                                    if (p.bias && !p.bias_per_row) v += p.bias[n + j];
                                    if (p.rowbias) v += p.rowbias[rboff + n + j];
                                }
                                if (p.act == 1) v = silu_f(v);
                        else if (p.act == 3) v = gelu_erf_f(v);
                            }
                            o[j] = v;
                        }
                        *(f32x4*)(stg + li * SP + bb * 32 + 8 * q + 4 * g) = o;
                    }
            }
            // the block is private to this wave and LDS operations of one wave execute in order: no workgroup barrier needed, only a
            // fence that keeps the compiler from moving the reads above the writes (and the next group's writes above these reads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // read back: 8 output elements per lane (two 16-byte LDS reads)
            float cs[8], cq[8];                      // GroupNorm column sums of this 32-row block (gn_colsum)
#pragma unroll
            for (int j = 0; j < 8; ++j) { cs[j] = 0.f; cq[j] = 0.f; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rr = lr + k * rows_per_pass;
                if (rr >= 32) break;
                const int mo = m_w0 + a * 32 + rr;
                const int n = ocol_w0 + cg * gcols + lc * 8;
                if (mo >= p.M || n >= nout) continue;
                const f32x4 v0 = *(const f32x4*)(stg + rr * SP + lc * 8);
                const f32x4 v1 = *(const f32x4*)(stg + rr * SP + lc * 8 + 4);
                float e[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const long oidx = obase + (long)mo * ldo + n;
                if (odt == GEO4D_F32) {
                    if (has_res) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { e[j] += __uint_as_float(rpre[k][0][j]); e[4 + j] += __uint_as_float(rpre[k][1][j]); }
                    }
                    f32x4 o0 = {e[0], e[1], e[2], e[3]}, o1 = {e[4], e[5], e[6], e[7]};
                    *(f32x4*)((float*)O + oidx) = o0;
                    *(f32x4*)((float*)O + oidx + 4) = o1;
                } else if (odt == GEO4D_BF16) {
                    if (has_res) {
                        float r[8];
                        chunk_to_f32<bf16_t>(rpre[k][0], r);
#pragma unroll
                        for (int j = 0; j < 8; ++j) e[j] += r[j];
                    }
                    *(u32x4*)((unsigned short*)O + oidx) = f32_to_chunk<bf16_t>(e);
                } else {
                    if (has_res) {
                        float r[8];
                        chunk_to_f32<f16_t>(rpre[k][0], r);
#pragma unroll
                        for (int j = 0; j < 8; ++j) e[j] += r[j];
                    }
                    *(u32x4*)((unsigned short*)O + oidx) = f32_to_chunk<f16_t>(e);
                }
                if (gn_on) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { cs[j] += e[j]; cq[j] += e[j] * e[j]; }
                }
            }
            if (gn_on) {
                // per-column sum / sum of squares over the 32 rows of this block: lanes that hold the same 8 columns sit cpr apart
                // (fixed-order xor butterfly: deterministic); the consumer GroupNorm merges the blocks (norm.hip gn_finalize_cols)
                for (int o = cpr; o < 64; o <<= 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { cs[j] += __shfl_xor(cs[j], o); cq[j] += __shfl_xor(cq[j], o); }
                }
                const int mb = m_w0 + a * 32;
                const int n = ocol_w0 + cg * gcols + lc * 8;
                if (lane < cpr && mb < p.M && n < nout) {
                    float* dst = p.gn_colsum + ((long)(mb >> 5) * nout + n) * 2;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        f32x4 v = {cs[j], cq[j], cs[j + 1], cq[j + 1]};
                        *(f32x4*)(dst + 2 * j) = v;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// out = epilogue(sum_z partial[z]) in fixed z order (deterministic); 8 consecutive n per thread
template <typename T>   // T only gives each translation unit its own copy (the kernel reads dtypes from p)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const geo4d_conv_gemm_t p, int splits) {
    const long total = (long)p.batch * p.M * (p.N / 8);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i % (p.N / 8)) * 8;
    const long bm = i / (p.N / 8);
    const int m = (int)(bm % p.M);
    const long bz = bm / p.M;
    const long slab = (long)p.batch * p.M * p.N;
    const float* src = (const float*)p.workspace + (bz * p.M + m) * (long)p.N + n;
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = 0.f;
    for (int z = 0; z < splits; ++z) {
        const f32x4 a = *(const f32x4*)(src + z * slab), b = *(const f32x4*)(src + z * slab + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { e[j] += a[j]; e[4 + j] += b[j]; }
    }
    const float brow = (p.bias && p.bias_per_row) ? p.bias[m] : 0.f;
    const long rboff = p.rowbias ? (long)(m / p.rowbias_div) * (p.ldrb ? p.ldrb : (long)p.N) : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = e[j] * p.alpha + brow;
        if (p.bias && !p.bias_per_row) v += p.bias[n + j];
        if (p.rowbias) v += p.rowbias[rboff + n + j];
        if (p.act == 1) v = silu_f(v);
        else if (p.act == 3) v = gelu_erf_f(v);
        if (p.R) v += load_res(p.R, bz * p.r_bs + (long)m * p.ldr + n + j, p.out_dtype);
        e[j] = v;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) store_out(p.O, bz * p.o_bs + (long)m * p.ldo + n + j, e[j], p.out_dtype);
}

// The same reduction for launches that ALSO emit the consumer GroupNorm's column sums (round 6; geo4d_conv_gemm_t.gn_colsum on a split-K
// launch of the second / third generation): a workgroup owns RB rows x (256 / RB) groups of 8 columns, thread (r, cg) finishes 8 outputs of row r
// exactly like splitk_reduce_kernel (same summation order, same epilogue order: same bits), then the RB rows of every column are added in a
// fixed tree (xor shuffles inside the wave, the 4 waves through LDS in wave order) into gn_colsum[M / RB][N][2] = (sum, sum of squares).
// batch 1, M % RB == 0, N % (8 * 256 / RB) == 0, f32 rows (splitk_colsum_rows below).
template <typename T, int RB>
__global__ __launch_bounds__(256) void splitk_reduce_colsum_kernel(const geo4d_conv_gemm_t p, int splits) {
    constexpr int CG = 256 / RB;                 // 8-column groups per workgroup
    __shared__ float part[4][CG * 8][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = tid / CG, cg = tid % CG;
    const int tiles_n = p.N / (8 * CG);
    const int bm = blockIdx.x / tiles_n, bn = blockIdx.x - bm * tiles_n;
    const int m = bm * RB + r, n = (bn * CG + cg) * 8;
    const long slab = (long)p.M * p.N;
    const float* src = (const float*)p.workspace + (long)m * p.N + n;
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = 0.f;
    for (int z = 0; z < splits; ++z) {
        const f32x4 a = *(const f32x4*)(src + z * slab), b = *(const f32x4*)(src + z * slab + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { e[j] += a[j]; e[4 + j] += b[j]; }
    }
    const long rboff = p.rowbias ? (long)(m / p.rowbias_div) * (p.ldrb ? p.ldrb : (long)p.N) : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = e[j] * p.alpha;
        if (p.bias) v += p.bias[n + j];
        if (p.rowbias) v += p.rowbias[rboff + n + j];
        if (p.R) v += ((const float*)p.R)[(long)m * p.ldr + n + j];
        e[j] = v;
    }
    *(f32x4*)((float*)p.O + (long)m * p.ldo + n) = f32x4{e[0], e[1], e[2], e[3]};
    *(f32x4*)((float*)p.O + (long)m * p.ldo + n + 4) = f32x4{e[4], e[5], e[6], e[7]};
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = e[j] * e[j];
    // rows of one wave: lanes that differ by a multiple of CG hold the same columns
#pragma unroll
    for (int o = CG; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { e[j] += __shfl_xor(e[j], o); q[j] += __shfl_xor(q[j], o); }
    }
    if (lane < CG) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { part[wave][lane * 8 + j][0] = e[j]; part[wave][lane * 8 + j][1] = q[j]; }
    }
    __syncthreads();
    if (tid < CG * 8) {
        float s0 = part[0][tid][0], s1 = part[0][tid][1];
        for (int w2 = 1; w2 < 4; ++w2) { s0 += part[w2][tid][0]; s1 += part[w2][tid][1]; }      // fixed wave order
        float* dst = p.gn_colsum + ((long)bm * p.N + bn * CG * 8 + tid) * 2;
        dst[0] = s0; dst[1] = s1;
    }
}
// rows per gn_colsum entry a split-K launch of the second / third generation emits through splitk_reduce_colsum_kernel (0 = it cannot): 32, or 8
// where a frame's rows are a multiple of 8 but not of 32 (the 5 x 8 level: per-frame GroupNorms need blocks that do not straddle frames)
inline int splitk_colsum_rows(const geo4d_conv_gemm_t& p, int sp) {
    if (sp <= 1 || p.batch != 1 || p.act != 0 || p.o_split || p.out_nchw || p.out_dtype != GEO4D_F32 || p.bias_per_row || (p.ldo & 3) || ((uintptr_t)p.O % 16) ||
        (p.R && ((uintptr_t)p.R % 4))) return 0;
    const int hw = p.Hout * p.Wout;
    if (hw % 32 == 0 && p.M % 32 == 0 && p.N % 64 == 0) return 32;
    if (hw % 8 == 0 && p.M % 8 == 0 && p.N % 256 == 0) return 8;
    return 0;
}
// the reduce launch of a split-K GEMM (every generation's launcher ends here): with gn_colsum, the column-sum form
template <typename T>
int launch_splitk_reduce(const geo4d_conv_gemm_t& p, int splits, hipStream_t stream) {
    if (p.gn_colsum) {
        const int rows = splitk_colsum_rows(p, splits);
        if (rows == 32) hipLaunchKernelGGL((splitk_reduce_colsum_kernel<T, 32>), dim3((unsigned)((p.M / 32) * (p.N / 64))), dim3(256), 0, stream, p, splits);
        else if (rows == 8) hipLaunchKernelGGL((splitk_reduce_colsum_kernel<T, 8>), dim3((unsigned)((p.M / 8) * (p.N / 256))), dim3(256), 0, stream, p, splits);
        else { geo4d_set_error("conv_gemm: gn_colsum on a split-K launch needs batch 1, f32 rows, no activation, frame rows % 8 == 0 (geo4d_conv_gemm_colsum_rows)"); return GEO4D_EINVAL; }
    } else {
        const long tot = (long)p.batch * p.M * (p.N / 8);
        hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, p, splits);
    }
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

template <typename T, int BM, int BN, int WM, int WN, int ST, int HOT>
int launch_kernel(const geo4d_conv_gemm_t& p, int splits, hipStream_t stream) {
    constexpr int smem = smem_bytes<BM, BN, WM, WN, ST>();
    static bool attr_set = false;
    auto kern = conv_gemm_kernel<T, BM, BN, WM, WN, ST, HOT>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
            geo4d_set_error("hipFuncSetAttribute(max dynamic LDS) failed");
            return GEO4D_EIO;
        }
        attr_set = true;
    }
    const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    dim3 grid((unsigned)tiles, (unsigned)p.batch, (unsigned)splits);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, p);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

template <typename T, int BM, int BN, int WM, int WN, int ST>
int launch_cfg(const geo4d_conv_gemm_t& p, int splits, hipStream_t stream) {
    if (p.act == 2 && ((BN / WN / 32) & 1)) {
        geo4d_set_error("conv_gemm: GEGLU needs wave tiles that are a multiple of 64 columns wide (this tile has an odd number of 32-column blocks)");
        return GEO4D_EINVAL;
    }
    int rc;
    if constexpr (IsX3<T>::value) {
        if (p.w_split && !p.a_split) rc = launch_kernel<T, BM, BN, WM, WN, ST, 1>(p, splits, stream);
        else if (p.w_split && p.a_split) rc = launch_kernel<T, BM, BN, WM, WN, ST, 2>(p, splits, stream);
        else rc = launch_kernel<T, BM, BN, WM, WN, ST, 0>(p, splits, stream);
    } else {
        rc = launch_kernel<T, BM, BN, WM, WN, ST, 0>(p, splits, stream);
    }
    if (rc != GEO4D_OK) return rc;
    if (splits > 1) {
        const long total = (long)p.batch * p.M * (p.N / 8);
        hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p, splits);
        GEO4D_CHECK_LAUNCH();
    }
    return GEO4D_OK;
}

// tile hints 22..28 (gemm_kernel_v2.h: 16x16x32 MFMA, register epilogue, persistent workgroups) are instantiated in their own
// translation units (gemm_v2_*.hip) so that the two kernel generations compile in parallel
template <typename T> int launch_v2_typed(const geo4d_conv_gemm_t& p, hipStream_t stream);
// tile hints 71..74 (gemm_kernel_v3.h: the same with a phased, counted-wait K loop): gemm_v3_*.hip
template <typename T> int launch_v3_typed(const geo4d_conv_gemm_t& p, hipStream_t stream);

// Tile choice: score = MFMA efficiency of the tile shape x useful fraction x how full the last wave of
// workgroups is (2 workgroups fit per CU by LDS => 512 slots on 256 CUs). Split-K multiplies the workgroup count
// when M x N alone cannot fill the chip and K is deep enough to amortise the extra fp32 slab traffic.
struct TileCfg { int bm, bn; float eff; };

template <typename T>
int launch_typed(const geo4d_conv_gemm_t& p, hipStream_t stream) {
    static constexpr TileCfg cfgs[] = {{128, 128, 1.00f}, {128, 64, 0.85f}, {64, 128, 0.80f}, {64, 64, 0.62f}, {128, 32, 0.50f}};
    if (p.tile_hint >= 71) return launch_v3_typed<T>(p, stream);
    if (p.tile_hint >= 21) return launch_v2_typed<T>(p, stream);
    if (p.o_split) {     // the pre-split output format lives in the register epilogue of the second-generation kernel only
        geo4d_conv_gemm_t q = p;
        // (GEGLU needs wave tiles that are a multiple of 64 columns wide: 22, 25, 27 of the second generation)
        const bool gg = p.act == 2;
        q.tile_hint = (p.tile_hint == 13 || p.tile_hint == 11) ? 22 : p.tile_hint == 16 ? (gg ? 22 : 23) : p.tile_hint == 3 ? 27 : p.tile_hint == 4 ? (gg ? 27 : 28) : 25;
        return launch_v2_typed<T>(q, stream);
    }
    if (p.tile_hint >= 11) {
        // explicit big-tile / deep-ring configurations, chosen by the host tuning table only. What they trade:
        // a CU can hold at most ~128 KB of LDS-DMA destinations, and a stage lands ~1 us after it is issued, so the
        // flops a CU can retire per microsecond are (bytes in flight) x (flops per byte of the tile shape).
        //   11: 256x128, 8 waves     13: 256x256, 8 waves (128 flop/B)
        //   16: 160x320, 10 waves (M = 40960, N = 320: 256 tiles = one per CU, each input row and each weight read once per tile)
        // Measured (profiles/r01_gemm_tiles.md): deeper rings (former hints 12, 14) never beat their 2-stage twins - the fill rate
        // per CU does not grow with more DMAs in flight - while the fatter tiles (11, 13) do: fewer L2->LDS bytes per flop.
        int sp = 1;
        if (p.split_k > 1) {
            if (!p.workspace || p.act == 2 || p.out_nchw || (p.N % 8) || (size_t)p.split_k * p.batch * p.M * p.N * 4 > p.workspace_bytes ||
                p.K / (BKC * Elem<T>::EPC) / p.split_k < 1) {
                geo4d_set_error("conv_gemm: split_k not applicable (workspace too small / epilogue not splittable)");
                return GEO4D_EINVAL;
            }
            sp = p.split_k;
        }
        switch (p.tile_hint) {
            case 11: return launch_cfg<T, 256, 128, 4, 2, 2>(p, sp, stream);
            case 13: return launch_cfg<T, 256, 256, 4, 2, 2>(p, sp, stream);
            case 16: return launch_cfg<T, 160, 320, 5, 2, 2>(p, sp, stream);   // 10 waves: all 320 columns of the level-0 layers in ONE tile
            case 17: return launch_cfg<T, 160, 160, 5, 1, 2>(p, sp, stream);   // 5 waves: M = 10240, N = 640 -> 256 tiles
        }
        geo4d_set_error("conv_gemm: unknown tile_hint");
        return GEO4D_EINVAL;
    }
    const int bk = BKC * Elem<T>::EPC;
    const int nslab = p.K / bk;
    const bool can_split = p.workspace && p.act != 2 && !p.out_nchw && (p.N % 8) == 0 && p.split_k != 1;
    int best = -1, best_split = 1;
    float best_score = -1.f;
    const int hint_tile = p.tile_hint;
    for (int i = 0; i < 5; ++i) {
        const TileCfg& c = cfgs[i];
        if (p.act == 2 && i >= 3) continue;  // GEGLU needs 64-wide wave tiles (NB == 2)
        if (hint_tile && i != hint_tile - 1) continue;
        const double tm = (p.M + c.bm - 1) / c.bm, tn = (p.N + c.bn - 1) / c.bn;
        const double tiles = tm * tn * p.batch;
        const double useful = ((double)p.M * p.N * p.batch) / (tiles * c.bm * c.bn);
        for (int s = 1; s <= 16; s *= 2) {
            if (s > 1 && (!can_split || nslab / s < 8)) break;
            if (p.split_k > 1 && s != p.split_k) continue;
            if (s > 1 && (size_t)s * p.batch * p.M * p.N * 4 > p.workspace_bytes) break;
            const double wgs = tiles * s;
            const double waves = (double)(long)((wgs + 511) / 512);
            const double fill = wgs / (waves * 512);
            const double split_cost = s > 1 ? 0.92 : 1.0;      // slab write + reduce kernel
            const float score = (float)(c.eff * useful * (0.30 + 0.70 * fill) * split_cost);
            if (score > best_score) { best_score = score; best = i; best_split = s; }
        }
    }
    switch (best) {
        case 0: return launch_cfg<T, 128, 128, 2, 2, 2>(p, best_split, stream);
        case 1: return launch_cfg<T, 128, 64, 4, 1, 2>(p, best_split, stream);
        case 2: return launch_cfg<T, 64, 128, 2, 2, 2>(p, best_split, stream);
        case 3: return launch_cfg<T, 64, 64, 2, 2, 2>(p, best_split, stream);
        case 4: return launch_cfg<T, 128, 32, 4, 1, 2>(p, best_split, stream);
    }
    geo4d_set_error("conv_gemm: no tile configuration (split_k / tile_hint not applicable to this problem?)");
    return GEO4D_EINVAL;
}


}  // namespace geo4d_gemm
