// geo4d_amd/csrc/gemm_kernel_v3.h — third-generation implicit-GEMM kernel (round 3, tile hints 71..74): the second generation's
// gather, LDS image, 16x16x32 MFMA form and register epilogue under a PHASED K loop.
//
// Why (profiles/r03_gemm_v2_explore_and_ablation.md): in the first two generations every wave of the single resident workgroup issues
// its 7-8 LDS-DMA pieces of a K slab back to back (60-185 cycles each while the request queue of the CU is full), then reads its
// fragments, then runs its MFMAs; all eight waves do this in lockstep between two barriers, so the matrix pipe idles while the
// requests are issued and the request path idles while the MFMAs run (full ~ MFMA-only + 0.75 x DMA-only). Here
//   * a K slab is cut into FOUR phases, one per quadrant of the wave tile (rows R0 | R1 x columns C0 | C1); each phase reads only the
//     fragments it newly needs (A0 + B0, B1, A1, none), issues ONE quarter of a future slab's LDS-DMA pieces (a "half panel": the R0 /
//     R1 rows of every wave's A rows, the C0 / C1 rows of the weight panel) and then runs the quadrant's MFMAs;
//   * the DMA waits are COUNTED (s_waitcnt vmcnt(pieces of one slab), never 0): a half panel is issued 4-5 phases before the wait
//     that retires it and read one phase after that wait, so a full slab (60-64 KB) is always in flight across the barriers;
//   * the two wave groups (waves 0-3 / 4-7: one wave of each on every SIMD) run STAGGERED by one barrier: while one group is in its
//     MFMA segment the other one reads fragments and issues DMA, and s_setprio keeps the MFMA segment ahead on the shared SIMD;
//   * workgroups are persistent and the staging cursor runs two slabs ahead ACROSS tiles: the next tile's gather table is built and
//     its first two slabs are issued during the current tile's last two slabs, the epilogue overlaps the other group's segment.
// Hazards (MI355X_MICROARCH.md "nothing orders a ds_read behind a pending LDS-DMA except the issuing wave's vmcnt plus a barrier"):
//   RAW  a half panel is read in phase k + 1 at the earliest when every wave's covering vmcnt sits before the first barrier of
//        phase k (group 1 executes that wait one barrier interval later than group 0, still before group 0's phase k + 1);
//   WAR  a half panel is re-staged two phases or more after the phase that read it (group 1's reads retire one interval late).
// Same ABI struct, same K order (channel-slab major, tap minor), same per-accumulator summation order as conv_gemm_v2_kernel:
// results are bit-identical to tile hints 21..29.
#pragma once
#include "gemm_kernel_v2.h"

namespace geo4d_gemm {

// the fall-back tiles live in the gemm_v2_*.hip translation units
extern template int launch_v2_typed<bf16x3_t>(const geo4d_conv_gemm_t&, hipStream_t);
extern template int launch_v2_typed<bf16_t>(const geo4d_conv_gemm_t&, hipStream_t);
extern template int launch_v2_typed<f16_t>(const geo4d_conv_gemm_t&, hipStream_t);

template <int BM, int BN>
constexpr int v3_smem_bytes() { return 2 * (BM + BN) * PITCH + BM * MAXTAP * 4; }

#define GEO4D_V3_BAR()                          \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

template <typename T, int BM, int BN, int WM, int WN, int HOT, bool OSPLIT = false>   // HOT: 0 generic, 1 raw A x split W, 2 split A x split W
__global__ __launch_bounds__(512) void conv_gemm_v3_kernel(const geo4d_conv_gemm_t p, const int splits, const int tiles_mn) {
    static_assert(WM * WN == 8, "two groups of four waves");
    static_assert(!std::is_same<T, float>::value, "v3 serves the 16-bit MFMA forms (bf16, f16, bf16x3)");
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = BKC * EPC;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MB = WTM / 16, NB = WTN / 16;
    constexpr int MB0 = (MB + 1) / 2, MB1 = MB - MB0, NB0 = (NB + 1) / 2, NB1 = NB - NB0;     // 16-blocks of R0 | R1, C0 | C1
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0 && MB1 >= 1 && NB1 >= 1, "wave tiles of at least 32 x 32");
    constexpr int RA0 = WM * MB0 * 16, RA1 = WM * MB1 * 16, RB0 = WN * NB0 * 16, RB1 = WN * NB1 * 16;   // rows of the half panels
    constexpr int OA0 = 0, OA1 = RA0 * PITCH, OB0 = BM * PITCH, OB1 = (BM + RB0) * PITCH;               // their offsets in a stage
    constexpr int STAGE = (BM + BN) * PITCH;
    constexpr int PA0 = (RA0 + 63) / 64, PA1 = (RA1 + 63) / 64, PB0 = (RB0 + 63) / 64, PB1 = (RB1 + 63) / 64;   // 64-row staging passes
    // pieces EVERY wave issues per slab (ragged last passes are issued by the first waves only): the counted wait. A wave that
    // issued more has more than NWAIT younger pieces outstanding, for which vmcnt(NWAIT) is the stricter wait.
    constexpr int NWAIT = RA0 / 64 + RA1 / 64 + RB0 / 64 + RB1 / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* rowpix = (int*)(smem + 2 * STAGE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int wr = wave / WN, wc = wave % WN;
    const int grp = wave >> 2;                         // the second group runs one barrier behind the first
    const int tiles_n = (p.N + BN - 1) / BN;
    const int ntap = p.KT * p.KH * p.KW;
    const int hw = p.Hout * p.Wout;
    const long total = (long)tiles_mn * p.batch * splits;
    const long G = gridDim.x;
    const bool direct_rows = ntap == 1 && p.stride == 1 && p.ups == 1 && p.ph == 0 && p.pw == 0 && p.pt == 0 && p.Hin * p.Win == hw;
    const int ns = p.K / BK / splits;                  // slabs per tile: the host guarantees an even split and ns >= 2
    const T* __restrict__ Z = (const T*)p.zeros;
    const int ccol = tid & 7;
    const int r0 = tid >> 3;

    // ---- staging side: LDS row r0 + 64 j of a half panel <-> tile row, source-side swizzle ----------------------------------------
    auto tile_row = [&](int r, int rows, int per_wave, int wt, int off) { return r < rows ? (r / per_wave) * wt + off + r % per_wave : -1; };
    auto rowA = [&](int h, int j) { return h ? tile_row(r0 + 64 * j, RA1, MB1 * 16, WTM, MB0 * 16) : tile_row(r0 + 64 * j, RA0, MB0 * 16, WTM, 0); };
    auto rowB = [&](int h, int j) { return h ? tile_row(r0 + 64 * j, RB1, NB1 * 16, WTN, NB0 * 16) : tile_row(r0 + 64 * j, RB0, NB0 * 16, WTN, 0); };
    auto chunk = [&](int j) { return (ccol ^ swz_key<T>(r0 + 64 * j)) * EPC; };   // LDS slot `ccol` of panel row r holds chunk ccol ^ key(r)

    // staging cursor (two slabs ahead of the MFMAs, across tiles) and the tile it is in
    long wS = 0;
    bool validS = false;
    int tmS = 0, tnS = 0, kzS = 0;
    long bzS = 0;
    const T* __restrict__ A_S = nullptr;               // activation base of the A cursor's tile
    const T* __restrict__ A_N = nullptr;               // ... of the tile the B cursor already moved to
    bool validA = false;                               // the A cursor's tile (it follows the B cursor one phase later)
    int tmA = 0;
    int tapA = 0, c0A = 0, tapB = 0, c0B = 0, tap_beg = 0, c0_beg = 0;
    const T* arow[2][PA0];                             // source of this thread's chunk of the A cursor's slab (without c0A); null = zeros
    const T* wrow[2][PB0];                             // weight rows of the B cursor's tile (without the K offset)

    auto fetch_A = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < (h ? PA1 : PA0); ++j) {
                const int row = rowA(h, j);
                const T* ptr = nullptr;
                if (row >= 0 && validA) {
                    int px;
                    if (direct_rows) {
                        const int m = tmA * BM + row;
                        px = m < p.M ? m : -1;
                    } else {
                        px = rowpix[row * ntap + tapA];
                    }
                    if (px >= 0) ptr = A_S + (long)px * p.lda + chunk(j);
                }
                arow[h][j] = ptr;
            }
        }
    };
    // moves the staging tile to w_ (B side at once, A side at the next advance_A): ids, weight rows, gather table
    auto stage_tile = [&](long w_) {
        wS = w_;
        validS = w_ < total;
        if (!validS) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < PB0; ++j) wrow[h][j] = nullptr;
            return;
        }
        const long t = w_ % tiles_mn, rest = w_ / tiles_mn;
        bzS = rest % p.batch;
        kzS = (int)(rest / p.batch);
        tmS = (int)(t / tiles_n);
        tnS = (int)(t - (long)tmS * tiles_n);
        A_N = (const T*)p.A + bzS * p.a_bs;
        const T* __restrict__ W = (const T*)p.W + bzS * p.w_bs;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < (h ? PB1 : PB0); ++j) {
                const int row = rowB(h, j);
                const int n = tnS * BN + row;
                wrow[h][j] = (row >= 0 && n < p.N) ? W + (long)n * p.ldw + chunk(j) : nullptr;
            }
        }
        const int s_begin = kzS * ns;
        tap_beg = s_begin % ntap;
        c0_beg = (s_begin / ntap) * BK;
        if (!direct_rows) {
            // gather table: source pixel of (tile row, tap), -1 = zero padding. The previous tile's table is dead: its last fetch
            // ran three phases ago. Visible to fetch_A one phase later (lgkmcnt(0) here, then the phase's barriers).
            const int hlim = p.ups == 2 ? 2 * p.Hin : p.Hin, wlim = p.ups == 2 ? 2 * p.Win : p.Win;
            const int ush = p.ups == 2 ? 1 : 0;
            for (int e = tid; e < BM * ntap; e += 512) {
                const int row = e / ntap, tp = e - row * ntap;
                const int m = tmS * BM + row;
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
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    // one LDS-DMA piece per 8 panel rows and wave: wave-uniform destination + lane * 16 B
    auto issue_A = [&](auto hc, int st) {
        constexpr int H = decltype(hc)::value;
        constexpr int R = H ? RA1 : RA0;
        char* base = smem + st * STAGE + (H ? OA1 : OA0) + wave * 1024;
#pragma unroll
        for (int j = 0; j < (H ? PA1 : PA0); ++j) {
            if ((j + 1) * 64 <= R || wave * 8 + j * 64 < R) {
                const T* src = arow[H][j] ? arow[H][j] + c0A : Z;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + j * 64 * PITCH), 16, 0, 0);
            }
        }
    };
    auto issue_B = [&](auto hc, int st) {
        constexpr int H = decltype(hc)::value;
        constexpr int R = H ? RB1 : RB0;
        char* base = smem + st * STAGE + (H ? OB1 : OB0) + wave * 1024;
        const long koff = (long)tapB * p.Cin + c0B;
#pragma unroll
        for (int j = 0; j < (H ? PB1 : PB0); ++j) {
            if ((j + 1) * 64 <= R || wave * 8 + j * 64 < R) {
                const T* src = wrow[H][j] ? wrow[H][j] + koff : Z;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + j * 64 * PITCH), 16, 0, 0);
            }
        }
    };
    // after B1 of slab t + 1: the B cursor moves to slab t + 2, which opens the next tile when t + 2 == ns
    auto advance_B = [&](int t) {
        if (t + 2 == ns) {
            stage_tile(wS + G);
            tapB = tap_beg;
            c0B = c0_beg;
        } else if (++tapB == ntap) {
            tapB = 0;
            c0B += BK;
        }
    };
    // after A1 of slab t + 1: the same for the A cursor (a new tile's table was written one phase ago)
    auto advance_A = [&](int t) {
        if (t + 2 == ns) {
            A_S = A_N;
            validA = validS;
            tmA = tmS;
            tapA = tap_beg;
            c0A = c0_beg;
            fetch_A();
        } else {
            if (++tapA == ntap) {
                tapA = 0;
                c0A += BK;
            }
            if (!direct_rows) fetch_A();
        }
    };

    // ---- MFMA side -------------------------------------------------------------------------------------------------------------------
    f32x4 acc[MB][NB];
    // fragment offsets inside a 16-row block: lane (lr, lq) reads row lr; 16-bit types: chunk 4h + lq of K half h; bf16x3: chunks 2lq (hi), 2lq + 1 (lo)
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
    u32x4 fa[2][MB0];                                  // the A half in use: [bf16x3: hi | lo; 16-bit: K half][block]
    u32x4 fb[2][2][NB0];                               // both B halves (C0 is used again by the slab's last phase)
    auto read_A = [&](auto hc, const char* sb) {
        constexpr int H = decltype(hc)::value;
        const char* base = sb + (H ? OA1 : OA0) + wr * ((H ? MB1 : MB0) * 16) * PITCH;
#pragma unroll
        for (int a = 0; a < (H ? MB1 : MB0); ++a) {
            fa[0][a] = *(const u32x4*)(base + a * 16 * PITCH + foff[0]);
            fa[1][a] = *(const u32x4*)(base + a * 16 * PITCH + foff[1]);
        }
    };
    auto read_B = [&](auto hc, const char* sb) {
        constexpr int H = decltype(hc)::value;
        const char* base = sb + (H ? OB1 : OB0) + wc * ((H ? NB1 : NB0) * 16) * PITCH;
#pragma unroll
        for (int b = 0; b < (H ? NB1 : NB0); ++b) {
            fb[H][0][b] = *(const u32x4*)(base + b * 16 * PITCH + foff[0]);
            fb[H][1][b] = *(const u32x4*)(base + b * 16 * PITCH + foff[1]);
        }
    };
    // raw f32 operands of the bf16x3 type are split once, in the MFMA segment of the phase that read them
    auto split_A = [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        if constexpr (IsX3<T>::value && HOT != 2) {
            if (!a_split) {
#pragma unroll
                for (int a = 0; a < (H ? MB1 : MB0); ++a) {
                    const u32x4 x0 = fa[0][a], x1 = fa[1][a];
                    split8_bf16(x0, x1, fa[0][a], fa[1][a]);
                }
            }
        }
    };
    auto split_B = [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        if constexpr (IsX3<T>::value && HOT == 0) {
            if (!w_split) {
#pragma unroll
                for (int b = 0; b < (H ? NB1 : NB0); ++b) {
                    const u32x4 y0 = fb[H][0][b], y1 = fb[H][1][b];
                    split8_bf16(y0, y1, fb[H][0][b], fb[H][1][b]);
                }
            }
        }
    };
    // quadrant (HA, HB): term-major so that consecutive MFMAs write different accumulators; per accumulator the terms keep the order
    // of mma16_x3 (w.lo x a.hi, w.hi x a.lo, w.hi x a.hi; C rows = n, C cols = m) resp. K half 0, 1
    auto mma_quadrant = [&](auto hac, auto hbc) {
        constexpr int HA = decltype(hac)::value, HB = decltype(hbc)::value;
        constexpr int MBH = HA ? MB1 : MB0, NBH = HB ? NB1 : NB0, AO = HA ? MB0 : 0, BO = HB ? NB0 : 0;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (IsX3<T>::value) {
#pragma unroll
            for (int term = 0; term < 3; ++term) {
#pragma unroll
                for (int b = 0; b < NBH; ++b) {
#pragma unroll
                    for (int a = 0; a < MBH; ++a) {
                        const u32x4& wv = fb[HB][term == 0 ? 1 : 0][b];
                        const u32x4& av = fa[term == 1 ? 1 : 0][a];
                        acc[AO + a][BO + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv), __builtin_bit_cast(bf16x8_t, av),
                                                                                      acc[AO + a][BO + b], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int b = 0; b < NBH; ++b) {
#pragma unroll
                    for (int a = 0; a < MBH; ++a) mma16<T>(acc[AO + a][BO + b], fb[HB][h][b], fa[h][a]);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    const bool partial = splits > 1;                   // split-K: raw fp32 slab, the epilogue runs in the reduce kernel

    // ---- first tile: table, then the steady state's in-flight set A0(0) B0(0) B1(0) A1(0) A0(1) B0(1) ----------------------------------
    const long w0 = xcd_remap((long)blockIdx.x, G);    // each XCD walks a contiguous range of every round of G tiles
    if (w0 >= total) return;
    stage_tile(w0);
    __syncthreads();
    A_S = A_N;
    validA = true;
    tmA = tmS;
    tapA = tapB = tap_beg;
    c0A = c0B = c0_beg;
    fetch_A();
    auto step_cursor = [&](int& tap, int& c0) {
        if (++tap == ntap) {
            tap = 0;
            c0 += BK;
        }
    };
    issue_A(H0{}, 0);
    issue_B(H0{}, 0);
    issue_B(H1{}, 0);
    step_cursor(tapB, c0B);
    issue_A(H1{}, 0);
    step_cursor(tapA, c0A);
    if (!direct_rows) fetch_A();
    issue_A(H0{}, 1);
    issue_B(H0{}, 1);
    int tmC = tmS, tnC = tnS, kzC = kzS;               // the tile the MFMAs are in
    long bzC = bzS;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
    __syncthreads();
    if (grp == 1) GEO4D_V3_BAR();

    int g = 0;                                         // ring stage of the current slab (slabs are counted across tiles)
    while (true) {
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < ns; ++t, g ^= 1) {
            const char* sb = smem + g * STAGE;
            // phase 0: fragments A0, B0 | B1 of slab t + 1 | quadrant (R0, C0)
            read_A(H0{}, sb);
            read_B(H0{}, sb);
            issue_B(H1{}, g ^ 1);
            advance_B(t);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");    // B1 of this slab
            GEO4D_V3_BAR();
            split_A(H0{});
            split_B(H0{});
            mma_quadrant(H0{}, H0{});
            GEO4D_V3_BAR();
            // phase 1: fragments B1 | A1 of slab t + 1 | quadrant (R0, C1)
            read_B(H1{}, sb);
            issue_A(H1{}, g ^ 1);
            advance_A(t);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");    // A1 of this slab
            GEO4D_V3_BAR();
            split_B(H1{});
            mma_quadrant(H0{}, H1{});
            GEO4D_V3_BAR();
            // phase 2: fragments A1 | A0 of slab t + 2 | quadrant (R1, C1)
            read_A(H1{}, sb);
            issue_A(H0{}, g);
            GEO4D_V3_BAR();
            split_A(H1{});
            mma_quadrant(H1{}, H1{});
            GEO4D_V3_BAR();
            // phase 3: no new fragments | B0 of slab t + 2 | quadrant (R1, C0)
            issue_B(H0{}, g);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");    // A0, B0 of slab t + 1
            GEO4D_V3_BAR();
            mma_quadrant(H1{}, H0{});
            if (t + 1 < ns) GEO4D_V3_BAR();
        }
        // the tile's closing barrier, then the epilogue (group 1's runs beside group 0's next fragment reads / DMA issue).
        // (Putting group 1's epilogue BEFORE the barrier so that both share one interval makes the register allocator spill ~900
        // VGPRs, including reloads inside the K loop whose vmcnt(0) drain the DMA queue - measured on the ISA, not worth it.)
        GEO4D_V3_BAR();
        reg_epilogue<MB, NB, OSPLIT, true>(p, acc, tmC * BM + wr * WTM, tnC * BN + wc * WTN, bzC, kzC, partial, lr, lq);
        // a REAL s_waitcnt vmcnt(0) (the builtin, which the compiler's wait-count pass tracks; an inline-asm one it does not see):
        // without it the pass has to assume pending loads into VGPRs at the K loop's header and drains the DMA queue every slab
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (!validS) break;                            // the staging tile is the next tile of the MFMAs
        tmC = tmS; tnC = tnS; kzC = kzS; bzC = bzS;
    }
    if (grp == 0) GEO4D_V3_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy pieces of the stream's tail must land before the LDS is released
}

// one workgroup per CU (96-155 KB of LDS); persistent over the tile list
template <typename T, int BM, int BN, int WM, int WN, int HOT, bool OSPLIT = false>
int launch_v3_kernel(const geo4d_conv_gemm_t& p, int splits, hipStream_t stream) {
    constexpr int smem = v3_smem_bytes<BM, BN>();
    static_assert(smem <= 160 * 1024, "LDS");
    static int resident = 0;
    auto kern = conv_gemm_v3_kernel<T, BM, BN, WM, WN, HOT, OSPLIT>;
    if (!resident) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
            geo4d_set_error("hipFuncSetAttribute(max dynamic LDS) failed");
            return GEO4D_EIO;
        }
        int dev = 0, cus = 0, occ = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 512, smem) != hipSuccess || cus <= 0 || occ <= 0) {
            geo4d_set_error("conv_gemm v3: occupancy query failed");
            return GEO4D_EIO;
        }
        resident = cus * occ;
    }
    const int tiles_mn = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const long total = (long)tiles_mn * p.batch * splits;
    const long cap = p.debug_ablate == 2 ? 3 : resident;          // tests: 3 workgroups, so that small shapes walk the tile stream
    const unsigned grid = (unsigned)(total < cap ? total : cap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, stream, p, splits, tiles_mn);
    GEO4D_CHECK_LAUNCH();
    if (splits > 1) {
        const long tot = (long)p.batch * p.M * (p.N / 8);
        hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, p, splits);
        GEO4D_CHECK_LAUNCH();
    }
    return GEO4D_OK;
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_v3_cfg(const geo4d_conv_gemm_t& p, int splits, hipStream_t stream) {
    if (p.act == 2 && ((BN / WN / 16) % 4)) {
        geo4d_set_error("conv_gemm v3: GEGLU needs wave tiles that are a multiple of 64 columns wide");
        return GEO4D_EINVAL;
    }
    if (p.o_split && !IsX3<T>::value) { geo4d_set_error("conv_gemm: o_split is a bf16x3 option"); return GEO4D_EINVAL; }
    if constexpr (IsX3<T>::value) {
        if (p.o_split) {
            if constexpr ((BN / WN / 16) % 4 == 0) {
                if (p.act == 2 && p.w_split && p.a_split) return launch_v3_kernel<T, BM, BN, WM, WN, 2, true>(p, splits, stream);
            }
            geo4d_set_error("conv_gemm: o_split is built for the GEGLU epilogue (act 2) of pre-split x pre-split launches on GEGLU-capable tiles");
            return GEO4D_EINVAL;
        }
        if (p.w_split && !p.a_split) return launch_v3_kernel<T, BM, BN, WM, WN, 1>(p, splits, stream);
        if (p.w_split && p.a_split) return launch_v3_kernel<T, BM, BN, WM, WN, 2>(p, splits, stream);
    }
    return launch_v3_kernel<T, BM, BN, WM, WN, 0>(p, splits, stream);
}

// tile hints 71..74: phased K loop on 8 waves (2 x 4), one workgroup per CU
//   71: 256x256 (wave tiles 128x64)   72: 160x320 (80x80)   73: 256x128 (128x32)   74: 128x256 (64x64)
// Launches the phased stream cannot take (fewer than 2 slabs per tile, an uneven split-K, outputs that are not 4-element aligned)
// fall back to the second-generation tile of the same shape.
template <typename T>
int launch_v3_typed(const geo4d_conv_gemm_t& p, hipStream_t stream) {
    if constexpr (std::is_same<T, float>::value) {
        geo4d_set_error("conv_gemm: tile hints 71..74 serve bf16 / f16 / bf16x3 (the exact-f32 mode stays on hints 0..17)");
        return GEO4D_EINVAL;
    } else {
        if (p.out_nchw || p.gn_colsum || p.debug_ablate == 1) {
            geo4d_set_error("conv_gemm: tile hints 71..74 have no NCTHW / gn_colsum epilogue");
            return GEO4D_EINVAL;
        }
        int sp = 1;
        const int nslab = p.K / (BKC * Elem<T>::EPC);
        if (p.split_k > 1) {
            if (!p.workspace || p.act == 2 || (p.N % 8) || (size_t)p.split_k * p.batch * p.M * p.N * 4 > p.workspace_bytes || nslab / p.split_k < 1) {
                geo4d_set_error("conv_gemm: split_k not applicable (workspace too small / epilogue not splittable)");
                return GEO4D_EINVAL;
            }
            sp = p.split_k;
        }
        if (p.tile_hint < 71 || p.tile_hint > 74) {
            geo4d_set_error("conv_gemm: unknown tile_hint");
            return GEO4D_EINVAL;
        }
        // the phased kernel carries the vector-store epilogue only (its scalar fallback costs ~900 spilled registers there)
        const bool geglu = sp == 1 && p.act == 2;
        const long nout = geglu ? (p.N >> 1) : p.N;
        const long oesz = (sp > 1 || p.out_dtype == GEO4D_F32) ? 4 : 2;
        bool vec_ok = (nout & 3) == 0;
        if (sp > 1) {
            vec_ok = vec_ok && ((uintptr_t)p.workspace % 16) == 0;
        } else {
            vec_ok = vec_ok && (p.ldo & 3) == 0 && ((uintptr_t)p.O % (4 * oesz)) == 0 && (p.batch == 1 || (p.o_bs & 3) == 0);
            if (p.R) vec_ok = vec_ok && (p.ldr & 3) == 0 && ((uintptr_t)p.R % (4 * oesz)) == 0 && (p.batch == 1 || (p.r_bs & 3) == 0);
        }
        if (nslab % sp || nslab / sp < 2 || !vec_ok) {
            geo4d_conv_gemm_t q = p;
            q.tile_hint = p.tile_hint == 71 ? 22 : p.tile_hint == 72 ? 23 : p.tile_hint == 73 ? 21 : 29;
            return launch_v2_typed<T>(q, stream);
        }
        switch (p.tile_hint) {
            case 71: return launch_v3_cfg<T, 256, 256, 2, 4>(p, sp, stream);
            case 72: return launch_v3_cfg<T, 160, 320, 2, 4>(p, sp, stream);
            case 73: return launch_v3_cfg<T, 256, 128, 2, 4>(p, sp, stream);
            case 74: return launch_v3_cfg<T, 128, 256, 2, 4>(p, sp, stream);
        }
        return GEO4D_EINVAL;
    }
}

#undef GEO4D_V3_BAR

}  // namespace geo4d_gemm
