// geo4d_amd/csrc/gemm_kernel_v3.h — third-generation implicit-GEMM kernel (round 3, tile hints 71..74): the second generation's LDS
// image, 16x16x32 MFMA form and register epilogue under a PHASED, software-pipelined K loop with a different staging side.
//
// Why (profiles/r03_gemm_v2_explore_and_ablation.md, profiles/r03_gemm_v3_phased.md): in the first two generations every wave of the
// single resident workgroup computes 7-8 source pointers per K slab (gather-table read + ~25 VALU instructions of 64-bit arithmetic
// each), issues its LDS-DMA pieces back to back (60-185 cycles each while the CU's request queue is full), reads its fragments, waits
// for them, then runs its MFMAs - one after the other, all eight waves in lockstep between two barriers. Here
//   * staging goes through RAW BUFFER RESOURCES (`buffer_load_dwordx4 ... lds`): per tile a 2 GB window per operand in SGPRs and a
//     32-bit byte offset per lane and piece in VGPRs that is constant for the tile; per slab only a wave-uniform SGPR offset (tap,
//     channel slab) changes. Zero padding = a lane offset beyond the window (the hardware writes zeros, tools/probe/bufload_probe.hip)
//     selected by one bit of a per-row tap mask: 3 VALU instructions per activation piece, none per weight piece, no gather table;
//   * a K slab is cut into FOUR phases, one per quadrant of the wave tile (rows R0 | R1 x columns C0 | C1), walked so that consecutive
//     phases share one operand half; every phase issues ONE half panel of a future slab (the R0 / R1 rows of every wave's A rows, the
//     C0 / C1 weight rows), reads the fragments of the half the NEXT phase newly needs (their registers are free by construction of
//     the walk) and runs its quadrant's MFMAs with fragments that arrived a phase ago;
//   * the DMA waits are COUNTED (`s_waitcnt vmcnt(pieces of four half panels)`, never 0): half panels are issued in the order they
//     are read, five phases ahead, and re-staged three phases after their read, so more than a full slab (60-64 KB) stays in flight
//     across the barriers; one barrier per phase;
//   * workgroups are persistent and the staging cursors run two (activations) and three (weights) slabs ahead ACROSS tiles: the next
//     tile's windows and first slabs are issued during the current tile's last slabs.
// Hazards (MI355X_MICROARCH.md: "nothing orders a ds_read behind a pending LDS-DMA except the issuing wave's vmcnt plus a barrier"):
//   RAW  a half panel is read in phase k + 1 when every wave's covering vmcnt sits before the barrier that closes phase k;
//   WAR  a half panel is re-staged three phases after the phase that read it (two would do: the reads retire before the MFMAs of the
//        following phase, i.e. before that phase's barrier).
// What it bought (MI355X, bf16x3, pre-split operands): the staging side of a slab went from ~2.5 us to ~1.6 us, but the whole kernel
// only from 222 us to 210 us on the largest conv of the U-Net (+4.5 %), +/- 4 % elsewhere: with the DMA and the fragment reads off
// the critical path the loop sits at ~75 % of what its MFMAs alone take on this chip (163 us, not the 103 us of the nominal clock).
// Same ABI struct, same K order (channel-slab major, tap minor), same per-accumulator summation order as conv_gemm_v2_kernel:
// results are bit-identical to tile hints 21..29.
#pragma once
#include "gemm_kernel_v2.h"

namespace geo4d_gemm {

// the fall-back tiles live in the gemm_v2_*.hip translation units
extern template int launch_v2_typed<bf16x3_t>(const geo4d_conv_gemm_t&, hipStream_t);
extern template int launch_v2_typed<bf16_t>(const geo4d_conv_gemm_t&, hipStream_t);
extern template int launch_v2_typed<f16x2p_t>(const geo4d_conv_gemm_t&, hipStream_t);

template <int BM, int BN>
constexpr int v3_smem_bytes() { return 2 * (BM + BN) * PITCH; }

#define GEO4D_V3_BAR()                          \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

// (Round 3 also carried ablation builds of this loop - no DMA / no MFMA / no reads / no waits / no barriers / s_memtime stamps - a
// variant with the two wave groups staggered by a second barrier per phase (4-10 % slower) and one with s_setprio around the MFMAs
// (1-4 % slower in this lockstep schedule): profiles/r03_gemm_v3_phased.md has their numbers; round 4 removed them from the sources.)
template <typename T, int BM, int BN, int WM, int WN, int HOT, bool OSPLIT = false>   // HOT: 0 generic, 1 raw A x split W, 2 split A x split W
__global__ __launch_bounds__(512) void conv_gemm_v3_kernel(const geo4d_conv_gemm_t p, const int splits, const int tiles_mn) {
    static_assert(WM * WN == 8, "two groups of four waves");
    static_assert(!std::is_same<T, float>::value, "v3 serves the 16-bit MFMA forms (bf16, bf16x3)");
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = BKC * EPC;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MB = WTM / 16, NB = WTN / 16;
    constexpr int MB0 = (MB + 1) / 2, MB1 = MB - MB0, NB0 = (NB + 1) / 2, NB1 = NB - NB0;     // 16-blocks of R0 | R1, C0 | C1
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0 && MB1 >= 1 && NB1 >= 1, "wave tiles of at least 32 x 32");
    constexpr int RA0 = WM * MB0 * 16, RA1 = WM * MB1 * 16, RB0 = WN * NB0 * 16, RB1 = WN * NB1 * 16;   // rows of the half panels
    // A64 (the two-pass f16 type, gemm_kernel_v2.h swz_key_a64): the activation is ONE f16 per element, stored as plain f16 rows (round 6) -
    // panel rows of 64 bytes = 64 contiguous bytes of the source row, 16 rows per 1 KB request, 128 rows per staging pass of the 8 waves:
    // half the A-side LDS-DMA requests of a slab
    constexpr bool A64 = IsTwoPass<T>::value;
    constexpr int PITCH_A = A64 ? PITCH_A64 : PITCH, RPA = A64 ? 128 : 64, RWA = A64 ? 16 : 8;          // A row pitch, rows per pass, rows per wave request
    constexpr int OA0 = 0, OA1 = RA0 * PITCH_A, OB0 = BM * PITCH, OB1 = (BM + RB0) * PITCH;             // their offsets in a stage
    constexpr int STAGE = (BM + BN) * PITCH;
    constexpr int PA0 = (RA0 + RPA - 1) / RPA, PA1 = (RA1 + RPA - 1) / RPA, PB0 = (RB0 + 63) / 64, PB1 = (RB1 + 63) / 64;   // staging passes
    // pieces EVERY wave issues per slab (ragged last passes are issued by the first waves only): the counted wait. A wave that
    // issued more has more than NWAIT younger pieces outstanding, for which vmcnt(NWAIT) is the stricter wait.
    constexpr int CA0 = RA0 / RPA, CA1 = RA1 / RPA, CB0 = RB0 / 64, CB1 = RB1 / 64;
    constexpr int NWAIT = CA0 + CA1 + CB0 + CB1;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int wr = wave / WN, wc = wave % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int ntap = p.KT * p.KH * p.KW;
    const int hw = p.Hout * p.Wout;
    const long total = (long)tiles_mn * p.batch * splits;
    const long G = gridDim.x;
    const bool direct_rows = ntap == 1 && p.stride == 1 && p.ups == 1 && p.ph == 0 && p.pw == 0 && p.pt == 0 && p.Hin * p.Win == hw;
    const int ns = p.K / BK / splits;                  // slabs per tile: the host guarantees an even split, ns even and >= 4
    const int ccol = tid & 7;
    const int r0 = tid >> 3;
    const int r0a = A64 ? (tid >> 2) : r0;             // this lane's row inside an A staging pass

    // ---- staging side -----------------------------------------------------------------------------------------------------------------
    // Every LDS-DMA piece is a `buffer_load_dwordx4 ... lds` on a raw buffer resource (base + 2 GB window in SGPRs): the per-lane part
    // of the address is a 32-bit byte offset that is CONSTANT for a tile (row x pitch + swizzled chunk), the per-slab part (tap, channel
    // slab) is a wave-uniform SGPR offset. A lane whose tap falls into zero padding (or whose row is beyond M / N) offers an offset
    // beyond the window: the hardware writes zeros for it (tools/probe/bufload_probe.hip). What this replaces: a gather table in LDS
    // plus ~25 VALU instructions per piece and slab of 64-bit pointer arithmetic - measured 1100-1700 cycles per slab in the MFMA
    // segment that followed it (profiles/r03_gemm_v3_timeline.md), i.e. the longest segment of the whole loop.
    constexpr unsigned OOB = 0x80000000u;              // = num_records of both resources
    constexpr int ESZ = 16 / EPC;                      // bytes per element (bf16x3 stores f32)
    constexpr int ESZ_A = A64 ? 2 : ESZ;               // ... of the activation operand (A64: plain f16 rows)
    auto tile_row = [&](int r, int rows, int per_wave, int wt, int off) { return r < rows ? (r / per_wave) * wt + off + r % per_wave : -1; };
    auto rowA = [&](int h, int j) { return h ? tile_row(r0a + RPA * j, RA1, MB1 * 16, WTM, MB0 * 16) : tile_row(r0a + RPA * j, RA0, MB0 * 16, WTM, 0); };
    auto rowB = [&](int h, int j) { return h ? tile_row(r0 + 64 * j, RB1, NB1 * 16, WTN, NB0 * 16) : tile_row(r0 + 64 * j, RB0, NB0 * 16, WTN, 0); };
    auto chunk = [&](int j) { return (ccol ^ swz_key<T>(r0 + 64 * j)) * EPC; };   // LDS slot `ccol` of panel row r holds chunk ccol ^ key(r)
    // A64: slot (tid & 3) of panel row r holds K-group slot ^ key_a64(r) of the slab = 8 consecutive f16 (16 bytes) of the source row
    auto chunkA = [&](int j) { return A64 ? ((tid & 3) ^ swz_key_a64(r0a + RPA * j)) * 8 : chunk(j); };
    const int khw = p.KH * p.KW;

    // staging cursors (two slabs ahead of the MFMAs, across tiles) and the tile they are in
    long wS = 0;
    bool validS = false;
    int tmS = 0, tnS = 0, kzS = 0;
    long bzS = 0;
    unsigned voffA[2][PA0], maskA[2][PA0];             // per lane: byte offset of (row's tap-0 pixel, chunk) in the A window; valid-tap bits
    unsigned voffB[2][PB0];                            // per lane: byte offset of (weight row, chunk) in the B window, OOB beyond N
    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.zeros, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rB = rA;
    // A cursor: tap (kt, ky, kx) and channel slab -> SGPR byte offset + bit index of the tap; B cursor: the same for the weight's K axis
    int ktA = 0, kyA = 0, kxA = 0, c0A = 0, tapB = 0, c0B = 0, tap_beg = 0, c0_beg = 0;
    unsigned soffA = 0, bitA = 0, soffB = 0;
    // (readfirstlane: the values are wave-uniform, but after a tile switch they come out of VALU code (integer divisions), and a buffer
    // load whose SGPR offset operand sits in a VGPR is legalised with a waterfall loop around every piece)
    auto refresh_A = [&]() {
        soffA = __builtin_amdgcn_readfirstlane((unsigned)((((long)ktA * p.Hin + kyA) * p.Win + kxA) * p.lda + c0A) * ESZ_A);
        bitA = __builtin_amdgcn_readfirstlane((unsigned)((ktA * p.KH + kyA) * p.KW + kxA));
    };
    auto refresh_B = [&]() { soffB = __builtin_amdgcn_readfirstlane((unsigned)(tapB * p.Cin + c0B) * ESZ); };
    auto begin_A = [&]() {                             // cursor at the tile's first slab
        ktA = tap_beg / khw;
        const int r2 = tap_beg - ktA * khw;
        kyA = r2 / p.KW;
        kxA = r2 - kyA * p.KW;
        c0A = c0_beg;
        refresh_A();
    };
    auto step_A = [&]() {                              // K order: channel-slab major, tap minor
        if (++kxA == p.KW) {
            kxA = 0;
            if (++kyA == p.KH) {
                kyA = 0;
                if (++ktA == p.KT) {
                    ktA = 0;
                    c0A += BK;
                }
            }
        }
        refresh_A();
    };
    auto step_B = [&]() {
        if (++tapB == ntap) {
            tapB = 0;
            c0B += BK;
        }
        refresh_B();
    };
    // the A window of tile (tm_, bz_): base = the tap-0 source pixel of the tile's first row (the smallest address any of its rows
    // reads; it may lie before the tensor when that pixel is padding - only in-image taps are ever dereferenced)
    auto tap0_pixel = [&](int m) -> long {
        const int f = m / hw, rem = m - f * hw;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        return ((long)(f - p.pt) * p.Hin + (oy * p.stride - p.ph)) * p.Win + (ox * p.stride - p.pw);
    };
    auto window_A = [&](bool valid, int tm_, long bz_) {
        const long P0 = valid ? tap0_pixel(tm_ * BM) : 0;
        rA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + (bz_ * p.a_bs + P0 * p.lda) * ESZ_A), 0, OOB, 0x00020000);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < (h ? PA1 : PA0); ++j) {
                const int row = rowA(h, j);
                const int m = tm_ * BM + row;
                unsigned off = 0, mask = 0;
                if (valid && row >= 0 && m < p.M) {
                    const int f = m / hw, rem = m - f * hw;
                    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                    const int iy0 = oy * p.stride - p.ph, ix0 = ox * p.stride - p.pw, ft = (f % p.T) - p.pt;
                    const long px0 = ((long)(f - p.pt) * p.Hin + iy0) * p.Win + ix0;
                    off = (unsigned)(((px0 - P0) * p.lda + chunkA(j)) * ESZ_A);
                    unsigned bit = 1;
                    for (int kt = 0; kt < p.KT; ++kt)
                        for (int ky = 0; ky < p.KH; ++ky)
                            for (int kx = 0; kx < p.KW; ++kx, bit <<= 1)
                                if ((unsigned)(ft + kt) < (unsigned)p.T && (unsigned)(iy0 + ky) < (unsigned)p.Hin && (unsigned)(ix0 + kx) < (unsigned)p.Win)
                                    mask |= bit;
                }
                voffA[h][j] = off;
                maskA[h][j] = mask;
            }
        }
    };
    // moves the staging tile to w_: ids and the B window at once, the A window at the next advance_A (one phase later)
    auto stage_tile = [&](long w_) {
        wS = w_;
        validS = w_ < total;
        if (!validS) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < PB0; ++j) voffB[h][j] = OOB;     // the stream's tail stages zeros
            return;
        }
        const long t = w_ % tiles_mn, rest = w_ / tiles_mn;
        bzS = rest % p.batch;
        kzS = (int)(rest / p.batch);
        tile_of(t, tiles_n, tiles_mn, tile_group_m(p), tmS, tnS);
        rB = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.W + bzS * p.w_bs + (long)tnS * BN * p.ldw), 0, OOB, 0x00020000);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < (h ? PB1 : PB0); ++j) {
                const int row = rowB(h, j);
                voffB[h][j] = (row >= 0 && tnS * BN + row < p.N) ? (unsigned)(((long)row * p.ldw + chunk(j)) * ESZ) : OOB;
            }
        }
        const int s_begin = kzS * ns;
        tap_beg = s_begin % ntap;
        c0_beg = (s_begin / ntap) * BK;
    };
    // one LDS-DMA piece per 8 panel rows and wave: wave-uniform destination + lane * 16 B
    auto issue_A = [&](auto hc, int st) {
        constexpr int H = decltype(hc)::value;
        constexpr int R = H ? RA1 : RA0;
        char* base = smem + st * STAGE + (H ? OA1 : OA0) + wave * 1024;
#pragma unroll
        for (int j = 0; j < (H ? PA1 : PA0); ++j) {
            if ((j + 1) * RPA <= R || wave * RWA + j * RPA < R) {
                const unsigned v = ((maskA[H][j] >> bitA) & 1u) ? voffA[H][j] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(base + j * RPA * PITCH_A), 16, (int)v, (int)soffA, 0, 0);
            }
        }
    };
    auto issue_B = [&](auto hc, int st) {
        constexpr int H = decltype(hc)::value;
        constexpr int R = H ? RB1 : RB0;
        char* base = smem + st * STAGE + (H ? OB1 : OB0) + wave * 1024;
#pragma unroll
        for (int j = 0; j < (H ? PB1 : PB0); ++j) {
            if ((j + 1) * 64 <= R || wave * 8 + j * 64 < R)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(base + j * 64 * PITCH), 16, (int)voffB[H][j], (int)soffB, 0, 0);
        }
    };
    // after both B half panels of a slab went out: the B cursor moves on; past the tile's last slab it opens the next tile
    int sA = 0, sB = 0;                                // slab of the A / B cursor inside its tile
    auto advance_B = [&]() {
        if (++sB == ns) {
            sB = 0;
            stage_tile(wS + G);
            tapB = tap_beg;
            c0B = c0_beg;
            refresh_B();
        } else {
            step_B();
        }
    };
    // the same for the A cursor, which follows the B cursor by a phase or more (B runs three slabs ahead of the MFMAs, A two)
    auto advance_A = [&]() {
        if (++sA == ns) {
            sA = 0;
            window_A(validS, tmS, bzS);
            begin_A();
        } else {
            step_A();
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
    const int foffA = lr * PITCH_A64 + ((lq ^ swz_key_a64(lr)) << 4);
    const bool a_split = HOT ? (HOT == 2) : (p.a_split != 0), w_split = HOT ? true : (p.w_split != 0);
    u32x4 fa[2][2][MB0];                               // A fragments [half][bf16x3: hi | lo; 16-bit: K half][block]: every half panel has its own
    u32x4 fb[2][2][NB0];                               // registers, so that a phase can read the set the NEXT phase multiplies
    auto read_A = [&](auto hc, const char* sb) {
        constexpr int H = decltype(hc)::value;
        const char* base = sb + (H ? OA1 : OA0) + wr * ((H ? MB1 : MB0) * 16) * PITCH_A;
#pragma unroll
        for (int a = 0; a < (H ? MB1 : MB0); ++a) {
            if constexpr (A64) {                     // hi chunks only: slot of K-group lq in row lr
                fa[H][0][a] = *(const u32x4*)(base + a * 16 * PITCH_A + foffA);
            } else {
                fa[H][0][a] = *(const u32x4*)(base + a * 16 * PITCH + foff[0]);
                fa[H][1][a] = *(const u32x4*)(base + a * 16 * PITCH + foff[1]);
            }
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
                    const u32x4 x0 = fa[H][0][a], x1 = fa[H][1][a];
                    split8_bf16(x0, x1, fa[H][0][a], fa[H][1][a]);
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
        if constexpr (IsTwoPass<T>::value) {
            // f16x2: w.lo x a.hi, then w.hi x a.hi (the order of mma16_x2)
#pragma unroll
            for (int term = 0; term < 2; ++term) {
#pragma unroll
                for (int b = 0; b < NBH; ++b) {
#pragma unroll
                    for (int a = 0; a < MBH; ++a)
                        acc[AO + a][BO + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fb[HB][term == 0 ? 1 : 0][b]),
                                                                                     __builtin_bit_cast(f16x8_t, fa[HA][0][a]), acc[AO + a][BO + b], 0, 0, 0);
                }
            }
        } else if constexpr (IsX3<T>::value) {
#pragma unroll
            for (int term = 0; term < 3; ++term) {
#pragma unroll
                for (int b = 0; b < NBH; ++b) {
#pragma unroll
                    for (int a = 0; a < MBH; ++a) {
                        const u32x4& wv = fb[HB][term == 0 ? 1 : 0][b];
                        const u32x4& av = fa[HA][term == 1 ? 1 : 0][a];
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
                    for (int a = 0; a < MBH; ++a) mma16<T>(acc[AO + a][BO + b], fb[HB][h][b], fa[HA][h][a]);
                }
            }
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    const bool partial = splits > 1;                   // split-K: raw fp32 slab, the epilogue runs in the reduce kernel

    // ---- first tile: windows, then the steady state's in-flight set A0(0) B0(0) B1(0) A1(0) A0(1) B1(1) B0(1) and the fragments of
    // the first quadrant ----------------------------------------------------------------------------------------------------------------
    const long w0 = xcd_remap((long)blockIdx.x, G);    // each XCD walks a contiguous range of every round of G tiles
    if (w0 >= total) return;
    stage_tile(w0);
    window_A(true, tmS, bzS);
    begin_A();
    tapB = tap_beg;
    c0B = c0_beg;
    refresh_B();
    issue_A(H0{}, 0);
    issue_B(H0{}, 0);
    issue_B(H1{}, 0);
    advance_B();
    issue_A(H1{}, 0);
    advance_A();
    issue_A(H0{}, 1);
    issue_B(H1{}, 1);
    issue_B(H0{}, 1);
    advance_B();
    int tmC = tmS, tnC = tnS, kzC = kzS;               // the tile the MFMAs are in
    long bzC = bzS;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");   // A0(0) B0(0) B1(0) landed; A1(0) A0(1) B1(1) B0(1) may be in flight
    __syncthreads();
    read_A(H0{}, smem);
    read_B(H0{}, smem);

    while (true) {
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        // One slab = four phases, one per quadrant of the wave tile; the quadrants are walked so that consecutive phases share one
        // operand half, and the half the NEXT phase needs is read while this phase's MFMAs run (its registers are free by construction):
        //   slab parity 0: (A0,B0) (A0,B1) (A1,B1) (A1,B0)      parity 1: (A0,B1) (A0,B0) (A1,B0) (A1,B1)      [X = parity, Y = 1 - X]
        //   phase 0: issue A1 of slab c + 1 (other ring stage) | read B_Y of slab c      | MFMA (A0, B_X)
        //   phase 1: issue A0 of slab c + 2                    | read A1 of slab c       | MFMA (A0, B_Y)
        //   phase 2: issue B_X of slab c + 2                   | read A0 of slab c + 1   | MFMA (A1, B_Y)
        //   phase 3: issue B_Y of slab c + 2                   | read B_Y of slab c + 1  | MFMA (A1, B_X)
        // Half panels are issued in the order they are read, FIVE phases ahead, and re-staged three phases after their read. The wait of
        // a phase retires the half panel the next phase reads: everything but the four half panels issued after it.
        auto slab = [&](auto xc) __attribute__((always_inline)) {
            constexpr int X = decltype(xc)::value, Y = 1 - X;
            using HX = std::integral_constant<int, X>;
            using HY = std::integral_constant<int, Y>;
            const char* sb = smem + X * STAGE;           // slabs of parity X live in ring stage X: tiles are whole pairs of slabs
            const char* so = smem + Y * STAGE;
            auto phase = [&](auto jc, auto issue, auto read, auto split, auto mma, auto after) __attribute__((always_inline)) {
                constexpr int J = decltype(jc)::value;
                constexpr int NW = J == 2 ? 2 * (X ? CB1 : CB0) + CA1 + CA0 : NWAIT;
                issue();
                read();
                split();
                mma();
                after();
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
            };
            phase(std::integral_constant<int, 0>{}, [&] { issue_A(H1{}, Y); }, [&] { read_B(HY{}, sb); },
                  [&] { split_A(H0{}); split_B(HX{}); }, [&] { mma_quadrant(H0{}, HX{}); }, [&] { advance_A(); });
            GEO4D_V3_BAR();
            phase(std::integral_constant<int, 1>{}, [&] { issue_A(H0{}, X); }, [&] { read_A(H1{}, sb); },
                  [&] { split_B(HY{}); }, [&] { mma_quadrant(H0{}, HY{}); }, [&] {});
            GEO4D_V3_BAR();
            phase(std::integral_constant<int, 2>{}, [&] { issue_B(HX{}, X); }, [&] { read_A(H0{}, so); },
                  [&] { split_A(H1{}); }, [&] { mma_quadrant(H1{}, HY{}); }, [&] {});
            GEO4D_V3_BAR();
            phase(std::integral_constant<int, 3>{}, [&] { issue_B(HY{}, X); }, [&] { read_B(HY{}, so); },
                  [&] {}, [&] { mma_quadrant(H1{}, HX{}); }, [&] { advance_B(); });
        };
        for (int t = 0; t < ns; t += 2) {
            slab(H0{});
            GEO4D_V3_BAR();
            slab(H1{});
            if (t + 2 < ns) GEO4D_V3_BAR();
        }
        // the tile's closing barrier, then the epilogue; the next tile's first slabs are already in flight. (An epilogue placed on either
        // side of the barrier depending on the wave group made the register allocator spill ~900 VGPRs, with reloads - and their
        // vmcnt(0) - inside the K loop.)
        GEO4D_V3_BAR();
        {   // the lane's row / column indices go through an opaque copy: otherwise the tile-invariant parts of the epilogue's addresses
            // (dozens of 64-bit row offsets) are hoisted out of the tile loop and live - spilled - across the K loop
            int lr_ = lr, lq_ = lq;
            asm volatile("" : "+v"(lr_), "+v"(lq_));
            reg_epilogue<MB, NB, OSPLIT, true, IsX3<T>::value, IsTwoPass<T>::value>(p, acc, tmC * BM + wr * WTM, tnC * BN + wc * WTN, bzC, kzC, partial, lr_, lq_);
        }
        // a REAL s_waitcnt vmcnt(0) (the builtin, which the compiler's wait-count pass tracks; an inline-asm one it does not see):
        // without it the pass has to assume pending loads into VGPRs at the K loop's header and drains the DMA queue every slab
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (!validS) break;                            // the staging tile is the next tile of the MFMAs
        tmC = tmS; tnC = tnS; kzC = kzS; bzC = bzS;
    }
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
    if (splits > 1) return launch_splitk_reduce<T>(p, splits, stream);
    return GEO4D_OK;
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_v3_cfg(const geo4d_conv_gemm_t& p, int splits, hipStream_t stream) {
    if (p.act == 2 && ((BN / WN / 16) % 4)) {
        geo4d_set_error("conv_gemm v3: GEGLU needs wave tiles that are a multiple of 64 columns wide");
        return GEO4D_EINVAL;
    }
    if (p.o_split && !IsX3<T>::value) { geo4d_set_error("conv_gemm: o_split is a bf16x3 option"); return GEO4D_EINVAL; }
    if constexpr (IsTwoPass<T>::value) {
        if (p.a_split != 2 || !p.w_split || (p.o_split && p.o_split != 2)) { geo4d_set_error("conv_gemm: f16x2 (dtype 4) takes plain f16 activation rows (a_split = 2) and a pre-split f16 weight; o_split 0 or 2 (plain f16 rows out)"); return GEO4D_EINVAL; }
        if (p.o_split) {
            // (the f16-row epilogue: every tile for plain rows - q | k, q | k | v, cross-attention q -, the GEGLU form on the tiles whose wave
            // tiles are a multiple of 64 columns wide: checked at the top of this function)
            if (o_f16_ok(p, splits)) return launch_v3_kernel<T, BM, BN, WM, WN, 2, true>(p, splits, stream);
            geo4d_set_error("conv_gemm: o_split = 2 (plain f16 rows out) needs no split-K / residual / row biases / SiLU / GELU, stored columns % 8 == 0 and 16-byte aligned output rows");
            return GEO4D_EINVAL;
        }
        return launch_v3_kernel<T, BM, BN, WM, WN, 2>(p, splits, stream);
    } else {
    if constexpr (IsX3<T>::value) {
        if (p.o_split) {
            if (o_split_ok(p, splits)) return launch_v3_kernel<T, BM, BN, WM, WN, 2, true>(p, splits, stream);
            geo4d_set_error("conv_gemm: o_split needs pre-split x pre-split operands, no split-K, N % 8 == 0 and 32-byte aligned output rows");
            return GEO4D_EINVAL;
        }
        if (p.w_split && !p.a_split) return launch_v3_kernel<T, BM, BN, WM, WN, 1>(p, splits, stream);
        if (p.w_split && p.a_split) return launch_v3_kernel<T, BM, BN, WM, WN, 2>(p, splits, stream);
    }
    return launch_v3_kernel<T, BM, BN, WM, WN, 0>(p, splits, stream);
    }
}

// tile hints 71..74: phased K loop on 8 waves (2 x 4), one workgroup per CU
//   71: 192x256 (wave tiles 96x64)   72: 160x320 (80x80)   73: 256x128 (128x32)   74: 128x256 (64x64)
// (256x256 does not fit: 128 accumulators + the four fragment sets of the prefetching walk spill)
// Launches the phased stream cannot take (an odd number or fewer than 4 K slabs per tile, an uneven split-K, outputs that are not 4-element aligned,
// nearest-upsampling gathers, operands beyond the 2 GB buffer window) fall back to the second-generation tile of the same shape.
inline int v3_wave_rows(int hint) { return hint == 71 ? 96 : hint == 72 ? 80 : hint == 73 ? 128 : 64; }
inline int v3_fallback_hint(int hint) { return hint == 71 ? 22 : hint == 72 ? 23 : 25; }
// does the phased stream take this launch (else its second-generation twin does)?
template <typename T>
bool v3_native(const geo4d_conv_gemm_t& p, int sp) {
    const int nslab = p.K / (BKC * Elem<T>::EPC);
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
    // the staging side addresses each operand through a 2 GB buffer window per tile (see the kernel): nearest-upsampling gathers have
    // no uniform tap offsets, and a tile's rows plus its taps must stay inside the window
    const long esz = 16 / Elem<T>::EPC, esz_a = IsTwoPass<T>::value ? 2 : esz;
    const long frames = 256 / ((long)p.Hout * p.Wout) + 2 + p.KT;
    const bool window_ok = p.ups == 1 && frames * p.Hin * p.Win * p.lda * esz_a < (1L << 31) && (320L * p.ldw + p.K) * esz < (1L << 31);
    return !(nslab % sp || ((nslab / sp) & 1) || nslab / sp < 4 || !vec_ok || !window_ok);
}
// rows per gn_colsum entry of the launch `p` describes (tile_hint >= 21), 0 = this launch cannot emit the sums
template <typename T>
int colsum_rows_v23(const geo4d_conv_gemm_t& p) {
    const int sp = p.split_k > 1 ? p.split_k : 1;
    if (sp > 1) return (p.gn_colsum == nullptr || ((uintptr_t)p.gn_colsum % 8) == 0) ? splitk_colsum_rows(p, sp) : 0;     // from the reduce launch
    int rows = 0;
    if (p.tile_hint >= 71 && p.tile_hint <= 74) rows = v3_native<T>(p, sp) ? v3_wave_rows(p.tile_hint) : v2_wave_rows(v2_effective_hint<T>(v3_fallback_hint(p.tile_hint)));
    else rows = v2_wave_rows(v2_effective_hint<T>(p.tile_hint));
    if (rows == 0 || !colsum_fast_ok(p, sp) || p.M % rows) return 0;
    return rows;
}

template <typename T>
int launch_v3_typed(const geo4d_conv_gemm_t& p, hipStream_t stream) {
    if constexpr (std::is_same<T, float>::value || std::is_same<T, f16_t>::value) {
        geo4d_set_error("conv_gemm: tile hints 71..74 serve bf16 / bf16x3 (the exact-f32 and the f16 modes stay on hints 0..17)");
        return GEO4D_EINVAL;
    } else {
        if (p.out_nchw) {
            geo4d_set_error("conv_gemm: tile hints 71..74 have no NCTHW epilogue");
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
        if (!v3_native<T>(p, sp)) {
            geo4d_conv_gemm_t q = p;
            q.tile_hint = v3_fallback_hint(p.tile_hint);      // (every tile sums in the same order: same bits)
            return launch_v2_typed<T>(q, stream);
        }
        if (p.gn_colsum && sp > 1) {      // split-K: the sums come from the reduce launch (gemm_kernel.h splitk_reduce_colsum_kernel)
            if (!splitk_colsum_rows(p, sp)) { geo4d_set_error("conv_gemm: this split-K launch cannot emit gn_colsum (geo4d_conv_gemm_colsum_rows)"); return GEO4D_EINVAL; }
        } else
        if (p.gn_colsum && (!colsum_fast_ok(p, sp) || p.M % v3_wave_rows(p.tile_hint) || ((uintptr_t)p.gn_colsum % 16))) {
            geo4d_set_error("conv_gemm: gn_colsum on tile hints 71..74 needs the plain f32-row epilogue (no activation / split-K / o_split / batch) and M % wave-tile rows == 0 (geo4d_conv_gemm_colsum_rows)");
            return GEO4D_EINVAL;
        }
        switch (p.tile_hint) {
            case 71: return launch_v3_cfg<T, 192, 256, 2, 4>(p, sp, stream);
            case 72: return launch_v3_cfg<T, 160, 320, 2, 4>(p, sp, stream);
            case 73: return launch_v3_cfg<T, 256, 128, 2, 4>(p, sp, stream);
            case 74: return launch_v3_cfg<T, 128, 256, 2, 4>(p, sp, stream);
        }
        return GEO4D_EINVAL;
    }
}

#undef GEO4D_V3_BAR

}  // namespace geo4d_gemm
