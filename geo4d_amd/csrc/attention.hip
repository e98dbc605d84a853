// geo4d_amd/csrc/attention.hip — spatiotemporal attention for the Geo4D U-Net.
//
// (1) flash_attn_kernel: softmax(Q K^T * scale) V with d_head = 64, no mask, for
//       * spatial self-attention  (attention.py:101-125 / xformers path :146-209), N in {2560,640,160,40}
//       * spatial cross-attention with TWO independent key/value sets whose normalised outputs are summed
//         (77 text tokens shared by every frame + 16 per-frame image tokens; attention.py:128-142)
//     Structure (one workgroup = 128 query rows = 4 waves x 32 rows, K/V tiles of 64 keys):
//       - swapped QK^T: S^T = K.Q^T on MFMA 32x32, so one lane owns one query column and the softmax row
//         reduction is in-register (one cross-half __shfl_xor(32) per tile) — wave-64 idiom, no LDS round trip;
//       - P never leaves registers: its C-layout registers ARE the B operand of the PV MFMA (O^T = V^T.P^T);
//       - V arrives already transposed ([head*64 + d][key], written that way by an operand-swapped projection GEMM), so
//         K and V^T tiles are both plain row tiles moved by LDS-DMA (global_load_lds_dwordx4) into a 2-deep ring with a
//         source-side XOR swizzle: no ds_write at all, next tile in flight while the current one is consumed,
//         ONE barrier per tile;
//       - softmax on raw scores with the scale folded into one fma + v_exp_f32 (exp2) per element.
// (2) temporal_attn_kernel: self-attention over T <= 16 frames for every (pixel, head)
//     (attention.py:365-412 TemporalTransformer, both attn1 and attn2). 0.02 TFLOP per forward but ~100 MB of
//     q/k/v/o traffic per layer: HBM-bound, so plain VALU with one wave per (pixel, head) and fp32 K/V in LDS.
//     Tokens stay in frame-major order [T, HW, C]; the kernel gathers the T rows of a pixel itself.
#include "common.h"
#include "geo4d_hip.h"

namespace {

// K tiles are [64 keys][64 d] and V^T tiles [64 d][64 keys]: both are 64 rows of 16-byte slots filled by LDS-DMA
// (lane-linear image), XOR-swizzled on the source side (128-byte rows: slot = chunk ^ ((row >> 1) & 7) as in gemm.hip;
// 256-byte f32 rows: slot = chunk ^ (row & 15)).
// QB = 32-row query blocks per wave (1: 128 query rows per workgroup; 2: 256 — every K / V^T fragment read from LDS feeds
// two MFMAs, and a launch needs half as many wave-slots). OCC = waves per SIMD the register allocator must leave room for.
// PS (bf16x3 only): q, K and V^T arrive in the producers' PRE-SPLIT format (geo4d_attention_t.qkv_split: per 8 elements of a row
// [8 x bf16 hi | 8 x bf16 lo], the same 4 bytes per element and the same 16-byte chunk addresses), so the kernel does not split K / V^T
// fragments per tile and wave (16 of its ~20 split8_bf16 per 64-key tile: ~30 % of the loop's VALU instructions). Same arithmetic as
// the in-kernel split -> bit-identical results.
template <typename T, int NSEG, int QB, int OCC, bool PS = false>
__global__ __launch_bounds__(256, OCC) void flash_attn_kernel(const geo4d_attention_t p) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int ES = (int)sizeof(T);
    constexpr int SLOTS = 64 / EPC;               // 16-byte slots per 64-element row: 8 (16-bit) or 16 (f32)
    constexpr int ROWB = 64 * ES;                 // row bytes: 128 / 256
    constexpr int NKK = SLOTS / 2;                // cmma steps over d
    constexpr int TILE = 64 * ROWB;               // one operand tile
    constexpr int NDMA = TILE / 1024 / 4;         // DMA instructions per wave per operand: 2 / 4
    constexpr int RPI = 1024 / ROWB;              // rows per DMA instruction: 8 / 4
    constexpr bool X3 = IsX3<T>::value;           // f32 storage, bf16 hi/lo split in registers, 3 bf16 MFMAs per product
    constexpr int PCH = X3 ? 2 : 16 / EPC;        // P chunks per 32-key block (x3: 16 keys per bf16 MFMA step, like the 16-bit types)
    constexpr int NQ = X3 ? 4 : NKK;              // MFMA k-steps over d = 64 (16 per step for the bf16 MFMA of the x3 path)
    static_assert(QB == 1 || NSEG == 1, "two query blocks per wave are built for single-segment (self) attention only");
    static_assert(!PS || X3, "pre-split inputs are a bf16x3 option");
    __shared__ __attribute__((aligned(16))) char lds[2 * 2 * TILE];   // [buf][K | Vt]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, g = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const T* __restrict__ Z = (const T*)p.zeros;

    int qrow[QB];
    bool qok[QB];
    u32x4 qf[QB][NQ], ql[QB][X3 ? 4 : 1];         // x3: qf = hi parts, ql = lo parts
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        qrow[qb] = blockIdx.x * (128 * QB) + wave * (32 * QB) + qb * 32 + li;
        qok[qb] = qrow[qb] < p.Nq;
        const T* qp = (const T*)p.q + ((long)b * p.Nq + (qok[qb] ? qrow[qb] : 0)) * p.ldq + h * 64;
        if constexpr (X3) {
            // k-step s2 covers d = 16 s2 .. +16; lane (li, g) owns d = 16 s2 + 8 g .. +8 = f32 chunks 4 s2 + 2 g, + 1
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                u32x4 c0 = {0u, 0u, 0u, 0u}, c1 = {0u, 0u, 0u, 0u};
                if (qok[qb]) {
                    c0 = *(const u32x4*)(qp + (4 * s2 + 2 * g) * EPC);
                    c1 = *(const u32x4*)(qp + (4 * s2 + 2 * g + 1) * EPC);
                }
                if constexpr (PS) { qf[qb][s2] = c0; ql[qb][s2] = c1; }      // chunk pair (2j, 2j+1) = (hi, lo) of d = 8j .. 8j+7
                else split8_bf16(c0, c1, qf[qb][s2], ql[qb][s2]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (qok[qb]) v = *(const u32x4*)(qp + (2 * kk + g) * EPC);
                qf[qb][kk] = v;
            }
        }
    }
    f32x16 of[NSEG == 1 ? 1 : 2], oa[QB][2];    // `of` (sum over segments) only exists for the dual-KV cross attention (QB = 1)
    float m_run[QB], l_run[QB];                 // running max of RAW scores; sums of exp2((s - m) * c)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -INFINITY;
        l_run[qb] = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                oa[qb][d][r] = 0.f;
                if constexpr (NSEG > 1) of[d][r] = 0.f;
            }
    }
    const float c2 = p.scale * 1.4426950408889634f;
    // lazy rescale (defer-max): the running max is only raised when some row's tile max exceeds it by more than THR raw
    // units, i.e. P = exp2((s - m) * c2) is allowed to reach 2^6 — exact in fp32, same relative precision in bf16/f16.
    // Order: decide + rescale O and l BEFORE this tile's P is formed and after the previous tile's P.V completed.
    const float thr = 6.0f / c2;

    // staging geometry of this lane: DMA instruction i of this wave covers rows (wave*NDMA + i)*RPI .. +RPI.
    // Per-lane source pointers are resolved once per segment; a full tile then costs one 64-bit add per DMA.
    const int srow = lane / SLOTS, sslot = lane % SLOTS;
    const T* ksrc[NDMA];
    const T* vsrc[NDMA];
    int krow[NDMA], vkey[NDMA];
    long kstep = 0;
    auto setup_seg = [&](int seg) {
        const int nk = seg ? p.Nk[1] : p.Nk[0];
        const long ldk = seg ? p.ldk[1] : p.ldk[0], ldvt = seg ? p.ldvt[1] : p.ldvt[0];
        const long kvb = b / (seg ? p.kv_div[1] : p.kv_div[0]);
        const T* kp = (const T*)(seg ? p.k[1] : p.k[0]) + kvb * nk * ldk + h * 64;
        const T* vp = (const T*)(seg ? p.vt[1] : p.vt[0]) + kvb * (seg ? p.vt_bs[1] : p.vt_bs[0]) + (long)h * 64 * ldvt;
        kstep = 64 * ldk;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int row = (wave * NDMA + i) * RPI + srow;            // key index in tile / d index
            const int chunk = sslot ^ (ES == 2 ? ((row >> 1) & 7) : (row & 15));   // logical 16-byte chunk fetched into this slot
            krow[i] = row;
            vkey[i] = PS ? (chunk >> 1) * 8 : chunk * EPC;              // first key of this lane's V^T chunk (pre-split: hi | lo chunks of 8 keys)
            ksrc[i] = kp + (long)row * ldk + chunk * EPC;
            vsrc[i] = vp + (long)row * ldvt + chunk * EPC;
        }
    };
    auto issue_tile = [&](int seg, int tile, int buf) {
        const int nk = seg ? p.Nk[1] : p.Nk[0];
        char* kb_ = lds + buf * 2 * TILE;
        const bool full = tile * 64 + 64 <= nk;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const T* ks = ksrc[i] + (long)tile * kstep;
            const T* vs = vsrc[i] + tile * 64;
            if (!full) {
                if (tile * 64 + krow[i] >= nk) ks = Z;
                if (tile * 64 + vkey[i] >= nk) vs = Z;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ks,
                                             (__attribute__((address_space(3))) void*)(kb_ + (wave * NDMA + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vs,
                                             (__attribute__((address_space(3))) void*)(kb_ + TILE + (wave * NDMA + i) * 1024), 16, 0, 0);
        }
    };

    // swizzled fragment offsets (row li of a 32-row block; (li >> 1) & (SLOTS-1) is block independent)
    const int swz = ES == 2 ? ((li >> 1) & 7) : (li & 15);   // 128-byte rows: 2 rows per bank row; 256-byte rows: 1
    int seg = 0, tile = 0, buf = 0;
    setup_seg(0);
    issue_tile(0, 0, 0);
    while (true) {
        // LDS-DMA is invisible to hipcc's waitcnt insertion inside the loop: drain it explicitly, THEN barrier
        // (without the explicit wait the barrier can release while this tile is still in flight -> stale LDS reads)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // this tile landed for every wave; everyone finished the previous tile
        const int nk = seg ? p.Nk[1] : p.Nk[0];
        const int ntile = (nk + 63) >> 6;
        int nseg2 = seg, ntile2 = tile + 1;
        if (ntile2 >= ntile) { nseg2 = seg + 1; ntile2 = 0; }
        const bool has_next = nseg2 < p.nseg;
        if (has_next) {
            if (nseg2 != seg) setup_seg(nseg2);
            issue_tile(nseg2, ntile2, buf ^ 1);                // flies while this tile is consumed
        }
        const char* ktile = lds + buf * 2 * TILE;
        const char* vtile = ktile + TILE;

        // ---- S^T = K.Q^T: every K fragment is read once and multiplied with the Q fragments of all QB query blocks ----------
        f32x16 st[QB][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[qb][kb][r] = 0.f;
            const char* krow_ = ktile + (kb * 32 + li) * ROWB;
            if constexpr (X3) {
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const u32x4 c0 = *(const u32x4*)(krow_ + (((4 * s2 + 2 * g) ^ swz) << 4));
                    const u32x4 c1 = *(const u32x4*)(krow_ + (((4 * s2 + 2 * g + 1) ^ swz) << 4));
                    u32x4 kh, kl;
                    if constexpr (PS) { kh = c0; kl = c1; }
                    else split8_bf16(c0, c1, kh, kl);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) mma_x3(st[qb][kb], kh, kl, qf[qb][s2], ql[qb][s2]);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const u32x4 a = *(const u32x4*)(krow_ + (((2 * kk + g) ^ swz) << 4));
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) cmma<T>(st[qb][kb], a, qf[qb][kk]);
                }
            }
        }
        // ---- online softmax on raw scores; exp2 with the scale folded in ----------------------
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (tile * 64 + 64 > nk) {               // only a ragged last tile pays for masking (wave-uniform branch)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (tile * 64 + kb * 32 + acc_row(r, g) >= nk) st[qb][kb][r] = -INFINITY;
            }
            float mt = st[qb][0][0];
#pragma unroll
            for (int r = 1; r < 16; r += 2)          // v_max3_f32: 2 new values per instruction, no canonicalising v_max pairs
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mt) : "v"(mt), "v"(st[qb][0][r]), "v"(st[qb][0][r + 1 < 16 ? r + 1 : r]));
#pragma unroll
            for (int r = 0; r < 16; r += 2)
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mt) : "v"(mt), "v"(st[qb][1][r]), "v"(st[qb][1][r + 1]));
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            if (__any(mt > m_run[qb] + thr)) {
                const float m_new = fmaxf(m_run[qb], mt);
                const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c2);   // m_run = -inf on the first tile -> 0
                l_run[qb] *= alpha;
                m_run[qb] = m_new;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oa[qb][d][r] *= alpha;
            }
            const f32x2 c2v = {c2, c2};
            const f32x2 mcv = {-m_run[qb] * c2, -m_run[qb] * c2};
            f32x2 ls2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {     // v_pk_fma_f32 + 2 x raw v_exp_f32 + v_pk_add_f32 per pair
                    f32x2 x = {st[qb][kb][r], st[qb][kb][r + 1]};
                    x = __builtin_elementwise_fma(x, c2v, mcv);
                    x[0] = __builtin_amdgcn_exp2f(x[0]);
                    x[1] = __builtin_amdgcn_exp2f(x[1]);
                    st[qb][kb][r] = x[0];
                    st[qb][kb][r + 1] = x[1];
                    ls2 += x;
                }
            l_run[qb] += ls2[0] + ls2[1];
        }
        // ---- O^T += V^T.P^T: every V^T fragment is read once for all QB query blocks --------------------------------------
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int c = 0; c < PCH; ++c) {
                if constexpr (X3) {
                    // P registers 8c .. 8c+7 of this lane = keys 32 kb + 16 c + 4 g + {0..3} and + 8 (see acc_row): split them once,
                    // fetch the same keys of V^T (f32 chunks 8 kb + 4 c + g and + 2) per 32-channel block and split those
                    u32x4 ph[QB], pl[QB];
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        float pv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) pv[j] = st[qb][kb][c * 8 + j];
                        split8_bf16(pv, ph[qb], pl[qb]);
                    }
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const char* vrow = vtile + (d * 32 + li) * ROWB;
                        u32x4 vh, vl;
                        if constexpr (PS) {
                            // keys 32 kb + 16 c + 4 g + {0..3} and + 8: positions 4g .. 4g+3 of the 8-key groups 4 kb + 2 c and + 1, whose hi / lo
                            // halves are the 16-byte chunks 2 G and 2 G + 1 of the row: four 8-byte reads (2-way bank conflict: 32 lanes x 8 B
                            // of 256-byte rows with one half-select per lane group; the b128 alternative reads twice the bytes for the same cycles)
                            const u32x2 h0 = *(const u32x2*)(vrow + (((8 * kb + 4 * c) ^ swz) << 4) + 8 * g);
                            const u32x2 l0 = *(const u32x2*)(vrow + (((8 * kb + 4 * c + 1) ^ swz) << 4) + 8 * g);
                            const u32x2 h1 = *(const u32x2*)(vrow + (((8 * kb + 4 * c + 2) ^ swz) << 4) + 8 * g);
                            const u32x2 l1 = *(const u32x2*)(vrow + (((8 * kb + 4 * c + 3) ^ swz) << 4) + 8 * g);
                            vh = u32x4{h0[0], h0[1], h1[0], h1[1]};
                            vl = u32x4{l0[0], l0[1], l1[0], l1[1]};
                        } else {
                            const u32x4 c0 = *(const u32x4*)(vrow + (((8 * kb + 4 * c + g) ^ swz) << 4));
                            const u32x4 c1 = *(const u32x4*)(vrow + (((8 * kb + 4 * c + 2 + g) ^ swz) << 4));
                            split8_bf16(c0, c1, vh, vl);
                        }
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) mma_x3(oa[qb][d], vh, vl, ph[qb], pl[qb]);
                    }
                } else {
                    u32x4 bch[QB];
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        float pv[EPC];
#pragma unroll
                        for (int j = 0; j < EPC; ++j) pv[j] = st[qb][kb][c * EPC + j];
                        bch[qb] = f32_to_chunk<T>(pv);
                    }
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        u32x4 a;
                        const char* vrow = vtile + (d * 32 + li) * ROWB;
                        if constexpr (ES == 2) {
                            // keys kb*32 + 16c + 4g + {0..3} and +8: byte offset 64kb + 32c + 8g (+16): slots 4kb + 2c (+1), half g
                            const u32x2 lo = *(const u32x2*)(vrow + (((4 * kb + 2 * c) ^ swz) << 4) + 8 * g);
                            const u32x2 hi = *(const u32x2*)(vrow + (((4 * kb + 2 * c + 1) ^ swz) << 4) + 8 * g);
                            a[0] = lo[0]; a[1] = lo[1]; a[2] = hi[0]; a[3] = hi[1];
                        } else {
                            // keys kb*32 + 8c + 4g + {0..3}: slot 8kb + 2c + g
                            a = *(const u32x4*)(vrow + (((8 * kb + 2 * c + g) ^ swz) << 4));
                        }
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) cmma<T>(oa[qb][d], a, bch[qb]);
                    }
                }
            }
        // ---- segment end: normalise and fold into the summed output ---------------------------
        if (ntile2 == 0) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float lt = l_run[qb] + __shfl_xor(l_run[qb], 32);
                const float inv = 1.0f / lt;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if constexpr (NSEG > 1) { of[d][r] += oa[qb][d][r] * inv; oa[qb][d][r] = 0.f; }
                        else oa[qb][d][r] *= inv;
                    }
                m_run[qb] = -INFINITY;
                l_run[qb] = 0.f;
            }
        }
        if (!has_next) break;
        seg = nseg2;
        tile = ntile2;
        buf ^= 1;
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        if (!qok[qb]) continue;
        T* op = (T*)p.o + ((long)b * p.Nq + qrow[qb]) * p.ldo + h * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const f32x16& res = NSEG > 1 ? of[d % (NSEG == 1 ? 1 : 2)] : oa[qb][d];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int dcol = d * 32 + 8 * i + 4 * g;
                if constexpr (ES == 2) {
                    float e[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { e[j] = res[4 * i + j]; e[4 + j] = 0.f; }
                    const u32x4 c = f32_to_chunk<T>(e);
                    u32x2 o2; o2[0] = c[0]; o2[1] = c[1];
                    *(u32x2*)(op + dcol) = o2;
                } else {
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = res[4 * i + j];
                    if (p.split_out) store_split4(op - h * 64, (h * 64 + dcol) >> 2, e);       // pre-split operand of the to_out GEMM
                    else *(u32x4*)(op + dcol) = f32_to_chunk<T>(e);
                }
            }
        }
    }
}

// ---- flash_attn2_kernel (round 4): the self-attention loop with the two query blocks of a wave SKEWED by one phase -----------------
// flash_attn_kernel<.., QB = 2> runs QK^T of both blocks, then both softmaxes, then both P.V: matrix and vector work alternate, and a
// wave's own MFMAs never have VALU work beside them (measured 27 % bf16 / 31 % bf16x3 of the MFMA peak, unchanged when 30 % of the VALU
// instructions were removed by the pre-split inputs: the loop is serialised, not issue-bound). Here the per-tile order of ONE wave is
//     A: S0 = K.Q0^T   |  head0 (row max, lazy rescale)  |  B: S1 = K.Q1^T  beside  P0 = exp2(S0)  |  head1  |
//     C: O0 += V^T.P0  beside  P1 = exp2(S1)             |  D: O1 += V^T.P1
// with B and C written as 8 slices of {fragment read for the next MFMA, one MFMA (three for bf16x3), the exponentials / row sums /
// bf16 conversions (hi / lo split) of 4 scores of the other block}, fenced by sched_barrier(0) so the interleave the source states
// is the interleave that is issued: the matrix pipe works on a slice's MFMA while the wave issues that slice's ~10-20 VALU
// instructions. K / V^T fragments are read per block (the shared reads of the QB = 2 build tied both blocks to one phase); masking
// code for a ragged last key tile is kept out of the full-tile path (hipcc had hoisted its 128 compares into every iteration).
// One key/value set (nseg == 1), d_head = 64, 16-bit types and bf16x3 (PS: pre-split q / K / V^T as in flash_attn_kernel).
template <typename T, bool PS, int OCC>
__global__ __launch_bounds__(256, OCC) void flash_attn2_kernel(const geo4d_attention_t p) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int ES = (int)sizeof(T);
    constexpr int SLOTS = 64 / EPC;
    constexpr int ROWB = 64 * ES;
    constexpr int TILE = 64 * ROWB;
    constexpr int NDMA = TILE / 1024 / 4;
    constexpr int RPI = 1024 / ROWB;
    constexpr bool X3 = IsX3<T>::value;
    static_assert(ES == 2 || X3, "16-bit types and bf16x3 only");
    static_assert(!PS || X3, "pre-split inputs are a bf16x3 option");
    __shared__ __attribute__((aligned(16))) char lds[2 * 2 * TILE];   // [buf][K | Vt]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, g = lane >> 5;
    // XCD-aware mapping: consecutive hardware workgroup ids go to different XCDs (8 private L2s); remapped so that the query blocks
    // of one (frame, head) - which all stream the same K / V^T - are neighbours on ONE XCD
    const long nwg = (long)gridDim.x * gridDim.y * gridDim.z;
    const long lid = xcd_remap(blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z), nwg);
    const int bx = (int)(lid % gridDim.x), h = (int)((lid / gridDim.x) % gridDim.y), b = (int)(lid / ((long)gridDim.x * gridDim.y));
    const T* __restrict__ Z = (const T*)p.zeros;
    const int nk = p.Nk[0];
    const int ntile = (nk + 63) >> 6;

    int qrow[2];
    bool qok[2];
    u32x4 qf[2][4], ql[2][X3 ? 4 : 1];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qrow[qb] = bx * 256 + wave * 64 + qb * 32 + li;
        qok[qb] = qrow[qb] < p.Nq;
        const T* qp = (const T*)p.q + ((long)b * p.Nq + (qok[qb] ? qrow[qb] : 0)) * p.ldq + h * 64;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            if constexpr (X3) {
                u32x4 c0 = {0u, 0u, 0u, 0u}, c1 = {0u, 0u, 0u, 0u};
                if (qok[qb]) {
                    c0 = *(const u32x4*)(qp + (4 * s2 + 2 * g) * EPC);
                    c1 = *(const u32x4*)(qp + (4 * s2 + 2 * g + 1) * EPC);
                }
                if constexpr (PS) { qf[qb][s2] = c0; ql[qb][s2] = c1; }
                else split8_bf16(c0, c1, qf[qb][s2], ql[qb][s2]);
            } else {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (qok[qb]) v = *(const u32x4*)(qp + (2 * s2 + g) * EPC);
                qf[qb][s2] = v;
            }
        }
    }
    f32x16 oa[2][2];
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oa[qb][d][r] = 0.f;
    const float c2 = p.scale * 1.4426950408889634f;
    const float thr = 6.0f / c2;                                        // lazy rescale threshold, as in flash_attn_kernel

    // staging geometry (identical to flash_attn_kernel: same LDS image, same source-side swizzle). Round 5: the pieces go through RAW
    // BUFFER RESOURCES - per lane ONE 32-bit byte offset per piece, the tile's position in a wave-uniform SGPR offset - instead of a
    // 64-bit pointer + a row / key index per piece: the 24 registers of that state were what the two-waves-per-SIMD bf16x3 build
    // spilled, and reloaded from scratch inside every key tile (120 bytes of scratch, ~20 scratch loads per tile).
    // In the ragged last tile a K row / a V^T key column at or beyond nk gets an offset beyond the 2 GB window (the hardware writes
    // zeros; the bounds check covers the VGPR offset only, not the SGPR tile offset, so the selection is explicit) - the row / key
    // indices are re-derived from the lane id there: nothing is kept live for it.
    const int srow = lane / SLOTS, sslot = lane % SLOTS;
    constexpr unsigned OOB = 0x80000000u;
    // piece i of a wave stages rows (wave NDMA + i) RPI + srow: its swizzled chunk is chunk(0) ^ 4 i (the row term of the key moves by
    // 4 per piece), so ONE offset per operand + the two byte deltas of flipping chunk bits 2 / 3 describe all NDMA pieces of a lane; the
    // row step of a piece is wave-uniform and rides in the SGPR offset
    unsigned voffK0, voffV0;
    int dflip2, dflip3;
    __amdgpu_buffer_rsrc_t rK, rV;
    auto piece_chunk = [&](int i) {
        const int row = (wave * NDMA + i) * RPI + srow;
        return sslot ^ (ES == 2 ? ((row >> 1) & 7) : (row & 15));
    };
    auto piece_delta = [&](int i) { return ((i & 1) ? dflip2 : 0) + ((i & 2) ? dflip3 : 0); };
    {
        const long kvb = b / p.kv_div[0];
        const T* kp = (const T*)p.k[0] + kvb * nk * p.ldk[0] + h * 64;
        const T* vp = (const T*)p.vt[0] + kvb * p.vt_bs[0] + (long)h * 64 * p.ldvt[0];
        // 2 GB windows from this (batch, head)'s first key row / first V^T channel row (the host checks that a window covers them)
        rK = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, OOB, 0x00020000);
        rV = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, OOB, 0x00020000);
        const int row0 = wave * NDMA * RPI + srow, c0 = piece_chunk(0);
        voffK0 = (unsigned)(((long)row0 * p.ldk[0] + c0 * EPC) * ES);
        voffV0 = (unsigned)(((long)row0 * p.ldvt[0] + c0 * EPC) * ES);
        dflip2 = (c0 & 4) ? -64 : 64;                   // chunk ^ 4: +- 4 chunks of 16 bytes
        dflip3 = (c0 & 8) ? -128 : 128;                 // chunk ^ 8 (4-byte storage: 16 chunks per row, 4 pieces per wave)
    }
    const unsigned kstepb = (unsigned)(64 * p.ldk[0] * ES);
    const unsigned prowK = (unsigned)(RPI * p.ldk[0] * ES), prowV = (unsigned)(RPI * p.ldvt[0] * ES);      // bytes from a piece's rows to the next piece's
    auto issue_tile = [&](int tile, int buf) {
        char* kb_ = lds + buf * 2 * TILE;
        const bool full = tile * 64 + 64 <= nk;
        const unsigned soffK = __builtin_amdgcn_readfirstlane((unsigned)tile * kstepb), soffV = __builtin_amdgcn_readfirstlane((unsigned)(tile * 64 * ES));
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            unsigned vk = voffK0 + (unsigned)piece_delta(i), vv = voffV0 + (unsigned)piece_delta(i);
            if (!full) {
                const int chunk = piece_chunk(i);
                if (tile * 64 + (wave * NDMA + i) * RPI + srow >= nk) vk = OOB;
                if (tile * 64 + (PS ? (chunk >> 1) * 8 : chunk * EPC) >= nk) vv = OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (__attribute__((address_space(3))) void*)(kb_ + (wave * NDMA + i) * 1024), 16, (int)vk, (int)(soffK + i * prowK), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (__attribute__((address_space(3))) void*)(kb_ + TILE + (wave * NDMA + i) * 1024), 16, (int)vv, (int)(soffV + i * prowV), 0, 0);
        }
    };
    const int swz = ES == 2 ? ((li >> 1) & 7) : (li & 15);

// fragment reads: K step s (16 d) of key block kb; V^T for keys of P chunk (kb, c) and channel block d (layouts: flash_attn_kernel)
#define A2_READ_K(KT, KB, S, FH, FL)                                                                   \
    do {                                                                                               \
        const char* krow_ = (KT) + ((KB) * 32 + li) * ROWB;                                            \
        if constexpr (X3) {                                                                            \
            const u32x4 c0_ = *(const u32x4*)(krow_ + (((4 * (S) + 2 * g) ^ swz) << 4));               \
            const u32x4 c1_ = *(const u32x4*)(krow_ + (((4 * (S) + 2 * g + 1) ^ swz) << 4));           \
            if constexpr (PS) { FH = c0_; FL = c1_; }                                                  \
            else split8_bf16(c0_, c1_, FH, FL);                                                        \
        } else {                                                                                       \
            FH = *(const u32x4*)(krow_ + (((2 * (S) + g) ^ swz) << 4));                                \
        }                                                                                              \
    } while (0)
#define A2_READ_V(VT, KB, CC, D, FH, FL)                                                               \
    do {                                                                                               \
        const char* vrow_ = (VT) + ((D) * 32 + li) * ROWB;                                             \
        if constexpr (X3 && PS) {                                                                      \
            const u32x2 h0_ = *(const u32x2*)(vrow_ + (((8 * (KB) + 4 * (CC)) ^ swz) << 4) + 8 * g);     \
            const u32x2 l0_ = *(const u32x2*)(vrow_ + (((8 * (KB) + 4 * (CC) + 1) ^ swz) << 4) + 8 * g); \
            const u32x2 h1_ = *(const u32x2*)(vrow_ + (((8 * (KB) + 4 * (CC) + 2) ^ swz) << 4) + 8 * g); \
            const u32x2 l1_ = *(const u32x2*)(vrow_ + (((8 * (KB) + 4 * (CC) + 3) ^ swz) << 4) + 8 * g); \
            FH = u32x4{h0_[0], h0_[1], h1_[0], h1_[1]};                                                \
            FL = u32x4{l0_[0], l0_[1], l1_[0], l1_[1]};                                                \
        } else if constexpr (X3) {                                                                     \
            const u32x4 c0_ = *(const u32x4*)(vrow_ + (((8 * (KB) + 4 * (CC) + g) ^ swz) << 4));       \
            const u32x4 c1_ = *(const u32x4*)(vrow_ + (((8 * (KB) + 4 * (CC) + 2 + g) ^ swz) << 4));   \
            split8_bf16(c0_, c1_, FH, FL);                                                             \
        } else {                                                                                       \
            const u32x2 lo_ = *(const u32x2*)(vrow_ + (((4 * (KB) + 2 * (CC)) ^ swz) << 4) + 8 * g);   \
            const u32x2 hi_ = *(const u32x2*)(vrow_ + (((4 * (KB) + 2 * (CC) + 1) ^ swz) << 4) + 8 * g); \
            FH = u32x4{lo_[0], lo_[1], hi_[0], hi_[1]};                                                \
        }                                                                                              \
    } while (0)
#define A2_MMA(ACC, AH, AL, BH, BL)                                  \
    do {                                                             \
        if constexpr (X3) mma_x3(ACC, AH, AL, BH, BL);               \
        else cmma<T>(ACC, AH, BH);                                   \
    } while (0)
// exponentials of scores 4 j .. 4 j + 3 of key block kb (slice I = 4 kb + j) -> words 2 (j & 1), + 1 of P chunk (kb, j >> 1); row sums
#define A2_EXP_SLICE(ST, PH, PL, LS, I)                                                                \
    do {                                                                                               \
        constexpr int kb_ = (I) >> 2, j_ = (I) & 3, c_ = j_ >> 1, w_ = 2 * (j_ & 1);                   \
        f32x2 x0_ = {ST[kb_][4 * j_], ST[kb_][4 * j_ + 1]}, x1_ = {ST[kb_][4 * j_ + 2], ST[kb_][4 * j_ + 3]}; \
        x0_ = __builtin_elementwise_fma(x0_, c2v, mcv);                                                \
        x1_ = __builtin_elementwise_fma(x1_, c2v, mcv);                                                \
        x0_[0] = __builtin_amdgcn_exp2f(x0_[0]);                                                       \
        x0_[1] = __builtin_amdgcn_exp2f(x0_[1]);                                                       \
        x1_[0] = __builtin_amdgcn_exp2f(x1_[0]);                                                       \
        x1_[1] = __builtin_amdgcn_exp2f(x1_[1]);                                                       \
        LS[0] += x0_[0];                                                                               \
        LS[1] += x0_[1];                                                                               \
        LS[0] += x1_[0];                                                                               \
        LS[1] += x1_[1];                                                                               \
        /* the empty asm pins the results HERE: without it LLVM sinks the whole slice next to its first use (the P.V phase) */ \
        if constexpr (X3) {                                                                            \
            unsigned int h0_ = f32x2_to_bf16x2(x0_[0], x0_[1]), h1_ = f32x2_to_bf16x2(x1_[0], x1_[1]); \
            unsigned int l0_ = f32x2_to_bf16x2(x0_[0] - __uint_as_float(h0_ << 16), x0_[1] - __uint_as_float(h0_ & 0xffff0000u)); \
            unsigned int l1_ = f32x2_to_bf16x2(x1_[0] - __uint_as_float(h1_ << 16), x1_[1] - __uint_as_float(h1_ & 0xffff0000u)); \
            asm volatile("" : "+v"(h0_), "+v"(h1_), "+v"(l0_), "+v"(l1_), "+v"(LS[0]), "+v"(LS[1]));   \
            PH[kb_][c_][w_] = h0_;                                                                     \
            PH[kb_][c_][w_ + 1] = h1_;                                                                 \
            PL[kb_][c_][w_] = l0_;                                                                     \
            PL[kb_][c_][w_ + 1] = l1_;                                                                 \
        } else {                                                                                       \
            unsigned int h0_, h1_;                                                                     \
            if constexpr (Elem<T>::DT == GEO4D_BF16) { h0_ = f32x2_to_bf16x2(x0_[0], x0_[1]); h1_ = f32x2_to_bf16x2(x1_[0], x1_[1]); } \
            else { h0_ = f32x2_to_f16x2(x0_[0], x0_[1]); h1_ = f32x2_to_f16x2(x1_[0], x1_[1]); }      \
            asm volatile("" : "+v"(h0_), "+v"(h1_), "+v"(LS[0]), "+v"(LS[1]));                         \
            PH[kb_][c_][w_] = h0_;                                                                     \
            PH[kb_][c_][w_ + 1] = h1_;                                                                 \
        }                                                                                              \
    } while (0)

    // row max of one block's 64 scores, the lazy-rescale decision (T13 order: the block's previous P.V is complete - it is the
    // producer of oa - and this tile's P is formed only afterwards), the exponent's fma constants
    auto head = [&](f32x16 (&st)[2], f32x16 (&o)[2], float& m, float& l, int tile, f32x2& mcv) {
        if (tile * 64 + 64 > nk) {
            int t0 = tile * 64 + 4 * g, nko = nk;
            asm volatile("" : "+v"(t0), "+s"(nko));      // opaque: keeps the 64 compares (and their lane constants) of the ragged tile out of the full-tile path
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t0 + kb * 32 + acc_row(r, 0) >= nko) st[kb][r] = -INFINITY;
        }
        float mt = st[0][0];
#pragma unroll
        for (int r = 1; r < 16; r += 2)
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mt) : "v"(mt), "v"(st[0][r]), "v"(st[0][r + 1 < 16 ? r + 1 : r]));
#pragma unroll
        for (int r = 0; r < 16; r += 2)
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mt) : "v"(mt), "v"(st[1][r]), "v"(st[1][r + 1]));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        if (__any(mt > m + thr)) {
            const float m_new = fmaxf(m, mt);
            const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c2);
            l *= alpha;
            m = m_new;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
        mcv = f32x2{-m * c2, -m * c2};
    };
    const f32x2 c2v = {c2, c2};

    int buf = 0;
    issue_tile(0, 0);
    for (int tile = 0; tile < ntile; ++tile, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA is invisible to hipcc's waitcnt insertion: drain, THEN barrier
        __syncthreads();
        if (tile + 1 < ntile) issue_tile(tile + 1, buf ^ 1);
        const char* ktile = lds + buf * 2 * TILE;
        const char* vtile = ktile + TILE;
        f32x16 s0[2], s1[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[kb][r] = 0.f; s1[kb][r] = 0.f; }
        u32x4 ph0[2][2], pl0[2][X3 ? 2 : 1], ph1[2][2], pl1[2][X3 ? 2 : 1];
        u32x4 fh[3], fl[3];
        f32x2 mcv;
        // ---- A: S0 = K.Q0^T (slice i: key block i & 1, k-step i >> 1: the two accumulators alternate) -----------------------------
        // (fragments are requested two slices ahead through a 3-slot register ring: fragment i sits in slot i % 3)
        A2_READ_K(ktile, 0, 0, fh[0], fl[0]);
        A2_READ_K(ktile, 1, 0, fh[1], fl[1]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < 6) A2_READ_K(ktile, (i + 2) & 1, (i + 2) >> 1, fh[(i + 2) % 3], fl[(i + 2) % 3]);
            A2_MMA(s0[i & 1], fh[i % 3], fl[i % 3], qf[0][i >> 1], ql[0][X3 ? (i >> 1) : 0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        head(s0, oa[0], m_run[0], l_run[0], tile, mcv);
        __builtin_amdgcn_sched_barrier(0);
        // ---- B: S1 = K.Q1^T beside P0 = exp2(S0 c - m c) ---------------------------------------------------------------------------
        {
            float ls[2] = {0.f, 0.f};
            A2_READ_K(ktile, 0, 0, fh[0], fl[0]);
            A2_READ_K(ktile, 1, 0, fh[1], fl[1]);
#define A2_B_SLICE(I)                                                                                                  \
            if ((I) < 6) A2_READ_K(ktile, ((I) + 2) & 1, ((I) + 2) >> 1, fh[((I) + 2) % 3], fl[((I) + 2) % 3]);        \
            A2_MMA(s1[(I) & 1], fh[(I) % 3], fl[(I) % 3], qf[1][(I) >> 1], ql[1][X3 ? ((I) >> 1) : 0]);                \
            A2_EXP_SLICE(s0, ph0, pl0, ls, I);                                                                         \
            __builtin_amdgcn_sched_barrier(0);
            A2_B_SLICE(0) A2_B_SLICE(1) A2_B_SLICE(2) A2_B_SLICE(3) A2_B_SLICE(4) A2_B_SLICE(5) A2_B_SLICE(6) A2_B_SLICE(7)
#undef A2_B_SLICE
            l_run[0] += ls[0] + ls[1];
        }
        head(s1, oa[1], m_run[1], l_run[1], tile, mcv);
        __builtin_amdgcn_sched_barrier(0);
        // ---- C: O0 += V^T.P0 beside P1 = exp2(S1 c - m c) (slice i: P chunk (i >> 2, (i >> 1) & 1), channel block i & 1) ------------
        {
            float ls[2] = {0.f, 0.f};
            A2_READ_V(vtile, 0, 0, 0, fh[0], fl[0]);
            A2_READ_V(vtile, 0, 0, 1, fh[1], fl[1]);
#define A2_C_SLICE(I)                                                                                                  \
            if ((I) < 6) A2_READ_V(vtile, ((I) + 2) >> 2, (((I) + 2) >> 1) & 1, ((I) + 2) & 1, fh[((I) + 2) % 3], fl[((I) + 2) % 3]); \
            A2_MMA(oa[0][(I) & 1], fh[(I) % 3], fl[(I) % 3], ph0[(I) >> 2][((I) >> 1) & 1], pl0[(I) >> 2][X3 ? (((I) >> 1) & 1) : 0]); \
            A2_EXP_SLICE(s1, ph1, pl1, ls, I);                                                                         \
            __builtin_amdgcn_sched_barrier(0);
            A2_C_SLICE(0) A2_C_SLICE(1) A2_C_SLICE(2) A2_C_SLICE(3) A2_C_SLICE(4) A2_C_SLICE(5) A2_C_SLICE(6) A2_C_SLICE(7)
#undef A2_C_SLICE
            l_run[1] += ls[0] + ls[1];
        }
        // ---- D: O1 += V^T.P1 ----------------------------------------------------------------------------------------------------------
        A2_READ_V(vtile, 0, 0, 0, fh[0], fl[0]);
        A2_READ_V(vtile, 0, 0, 1, fh[1], fl[1]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < 6) A2_READ_V(vtile, (i + 2) >> 2, ((i + 2) >> 1) & 1, (i + 2) & 1, fh[(i + 2) % 3], fl[(i + 2) % 3]);
            A2_MMA(oa[1][i & 1], fh[i % 3], fl[i % 3], ph1[i >> 2][(i >> 1) & 1], pl1[i >> 2][X3 ? ((i >> 1) & 1) : 0]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef A2_READ_K
#undef A2_READ_V
#undef A2_MMA
#undef A2_EXP_SLICE
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float lt = l_run[qb] + __shfl_xor(l_run[qb], 32);
        const float inv = 1.0f / lt;
        if (!qok[qb]) continue;
        T* op = (T*)p.o + ((long)b * p.Nq + qrow[qb]) * p.ldo + h * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int dcol = d * 32 + 8 * i + 4 * g;
                if constexpr (ES == 2) {
                    float e[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { e[j] = oa[qb][d][4 * i + j] * inv; e[4 + j] = 0.f; }
                    const u32x4 c = f32_to_chunk<T>(e);
                    u32x2 o2; o2[0] = c[0]; o2[1] = c[1];
                    *(u32x2*)(op + dcol) = o2;
                } else {
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = oa[qb][d][4 * i + j] * inv;
                    if (p.split_out) store_split4(op - h * 64, (h * 64 + dcol) >> 2, e);
                    else *(u32x4*)(op + dcol) = f32_to_chunk<T>(e);
                }
            }
    }
}

// one wave per (batch, pixel, head); lane = (query frame tq = lane >> 2, d-slice dp = lane & 3 of 16 channels).
// Q, K, V of the unit are staged as fp32 in LDS; the d-slice is walked in 4-wide steps by a ROLLED loop so the
// kernel stays at ~64 VGPRs (a fully unrolled body made hipcc hoist all 128 ds_read_b128 and spill).
template <typename T>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ k, long ldk,
                                                            const T* __restrict__ v, long ldv, T* __restrict__ o, long ldo, int B,
                                                            int Tn, int HW, int H, float scale, int split_out) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int DCH = 64 / EPC;
    __shared__ __attribute__((aligned(16))) float qf[4][16][64];
    __shared__ __attribute__((aligned(16))) float kf[4][16][64];
    __shared__ __attribute__((aligned(16))) float vf[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long total = (long)B * HW * H;
    long unit = (long)blockIdx.x * 4 + wave;
    const bool uok = unit < total;
    if (!uok) unit = total - 1;
    const int h = (int)(unit % H);
    const long bp = unit / H;
    const int pix = (int)(bp % HW);
    const int b = (int)(bp / HW);
    const long row0 = (long)b * Tn * HW + pix;  // row of frame t = row0 + t*HW

#pragma unroll
    for (int i = 0; i < 16 * DCH / 64; ++i) {
        const int idx = lane + i * 64;
        const int t = idx / DCH, col = idx % DCH;
        float qe[EPC], ke[EPC], ve[EPC];
        if (t < Tn) {
            const long r = row0 + (long)t * HW;
            const u32x4 qc = *(const u32x4*)(q + r * ldq + h * 64 + col * EPC);
            const u32x4 kc = *(const u32x4*)(k + r * ldk + h * 64 + col * EPC);
            const u32x4 vc = *(const u32x4*)(v + r * ldv + h * 64 + col * EPC);
            chunk_to_f32<T>(qc, qe);
            chunk_to_f32<T>(kc, ke);
            chunk_to_f32<T>(vc, ve);
        } else {
#pragma unroll
            for (int j = 0; j < EPC; ++j) { qe[j] = 0.f; ke[j] = 0.f; ve[j] = 0.f; }
        }
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            qf[wave][t][col * EPC + j] = qe[j];
            kf[wave][t][col * EPC + j] = ke[j];
            vf[wave][t][col * EPC + j] = ve[j];
        }
    }
    __syncthreads();
    const int tq = lane >> 2, dp = lane & 3;
    float s[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) s[kk] = 0.f;
#pragma unroll 1
    for (int j4 = 0; j4 < 4; ++j4) {
        const int d0 = dp * 16 + j4 * 4;
        const f32x4 q4 = *(const f32x4*)&qf[wave][tq][d0];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const f32x4 k4 = *(const f32x4*)&kf[wave][kk][d0];
            s[kk] += q4[0] * k4[0] + q4[1] * k4[1] + q4[2] * k4[2] + q4[3] * k4[3];
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        float a = s[kk];
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        a = kk < Tn ? a * scale : -INFINITY;
        s[kk] = a;
        m = fmaxf(m, a);
    }
    float l = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) { s[kk] = __expf(s[kk] - m); l += s[kk]; }
    const float inv = 1.0f / l;
    const bool wok = uok && tq < Tn;
    T* op = o + (row0 + (long)(tq < Tn ? tq : 0) * HW) * ldo + h * 64;
#pragma unroll 1
    for (int j4 = 0; j4 < 4; ++j4) {
        const int d0 = dp * 16 + j4 * 4;
        float o4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const f32x4 v4 = *(const f32x4*)&vf[wave][kk][d0];
            o4[0] += s[kk] * v4[0]; o4[1] += s[kk] * v4[1]; o4[2] += s[kk] * v4[2]; o4[3] += s[kk] * v4[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] *= inv;
        if (wok) {
            if constexpr (sizeof(T) == 2) {
                float e[8] = {o4[0], o4[1], o4[2], o4[3], 0.f, 0.f, 0.f, 0.f};
                const u32x4 c = f32_to_chunk<T>(e);
                u32x2 o2; o2[0] = c[0]; o2[1] = c[1];
                *(u32x2*)(op + d0) = o2;
            } else {
                if (split_out) store_split4(op - h * 64, (h * 64 + d0) >> 2, o4);
                else *(u32x4*)(op + d0) = f32_to_chunk<T>(o4);
            }
        }
    }
}

}  // namespace

extern "C" int geo4d_attention(const geo4d_attention_t* pp, void* stream) {
    if (!pp) return GEO4D_EINVAL;
    const geo4d_attention_t& p = *pp;
    const int esz = (p.dtype == GEO4D_F32 || p.dtype == GEO4D_BF16X3) ? 4 : 2;
    if (p.dtype < 0 || p.dtype > 3 || p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.nseg < 1 || p.nseg > 2) { geo4d_set_error("attention: bad arguments"); return GEO4D_EINVAL; }
    if (p.head_dim != 64) { geo4d_set_error("attention: only d_head = 64 is built (yaml num_head_channels: 64)"); return GEO4D_ENOTSUP; }
    if ((p.ldq * esz) % 16 || (p.ldo * esz) % 16 || ((uintptr_t)p.q % 16) || ((uintptr_t)p.o % 16)) { geo4d_set_error("attention: q/o alignment"); return GEO4D_EINVAL; }
    for (int s = 0; s < p.nseg; ++s) {
        if (p.Nk[s] <= 0 || p.kv_div[s] <= 0 || !p.k[s] || !p.vt[s] || (p.ldk[s] * esz) % 16 || (p.ldvt[s] * esz) % 16 || (p.vt_bs[s] * esz) % 16 ||
            ((uintptr_t)p.k[s] % 16) || ((uintptr_t)p.vt[s] % 16) || p.ldvt[s] * (16 / esz) < 0 || ((p.Nk[s] + 16 / esz - 1) / (16 / esz)) * (16 / esz) > p.ldvt[s]) {
            geo4d_set_error("attention: bad key/value segment");
            return GEO4D_EINVAL;
        }
    }
    if (!p.zeros || ((uintptr_t)p.zeros % 16)) { geo4d_set_error("attention: `zeros` must point at 16 zero bytes"); return GEO4D_EINVAL; }
    if (p.H > 65535 || p.B > 65535) { geo4d_set_error("attention: grid too large"); return GEO4D_EINVAL; }
    if (p.split_out && esz != 4) { geo4d_set_error("attention: split_out is the producer format of the 4-byte storage modes (f32 / bf16x3)"); return GEO4D_EINVAL; }
    if (p.qkv_split) {
        if (p.dtype != GEO4D_BF16X3 || p.nseg != 1 || (p.Nk[0] % 8) || (p.ldq % 8) || (p.ldk[0] % 8) || (p.ldvt[0] % 8) || (p.vt_bs[0] % 8) ||
            ((uintptr_t)p.q % 32) || ((uintptr_t)p.k[0] % 32) || ((uintptr_t)p.vt[0] % 32)) {
            geo4d_set_error("attention: qkv_split needs dtype bf16x3, one key/value set, Nk % 8 == 0, leading dimensions % 8 == 0 and 32-byte aligned bases");
            return GEO4D_EINVAL;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    // variant: 0 = host default; explicit: 1 = 128 rows / workgroup at 3 waves per SIMD (round-1 kernel), 2 = the same at 4 waves
    // per SIMD (128-VGPR budget), 3 = 256 rows / workgroup, two query blocks per wave (self-attention only)
    int variant = p.variant;
    if (variant < 0 || variant > 5) { geo4d_set_error("attention: unknown variant"); return GEO4D_EINVAL; }
    // 4 / 5 = flash_attn2_kernel (round 4: skewed query blocks, matrix work beside every softmax): one key/value set, 16-bit types
    // (4) and bf16x3 on pre-split inputs (4 = one wave per SIMD with the whole register file, 5 = two waves per SIMD).
    // Default since round 4 wherever it applies and a 256-row workgroup is not mostly empty (measured, profiles/r04_attention.md:
    // bf16 N = 2560 199.9 -> 189.9 us, N = 640 33.6 -> 31.4; pre-split bf16x3 N = 2560 524 -> 443 us, N = 640 76.3 -> 67.8)
    // (bf16x3: the two-waves-per-SIMD build wins on the long level-0 sequences once the workgroups of a (frame, head) share an XCD:
    // N = 2560 467 us (variant 4) vs 443 us (variant 5); N = 640 67.8 vs 70.3 us)
    // (round 5: that kernel stages K / V^T through 2 GB buffer windows with 32-bit offsets: one (batch, head)'s keys and V^T rows must
    // fit one - they do by orders of magnitude at every size of BASELINE.json; otherwise the pointer-staged kernels below serve the call)
    const long esz_w = (p.dtype == GEO4D_BF16 || p.dtype == GEO4D_F16) ? 2 : 4;
    const bool window_ok = p.nseg == 1 && ((long)(p.Nk[0] + 64) * p.ldk[0] + 64) * esz_w < (1L << 31) && (64L * p.ldvt[0] + p.Nk[0] + 64) * esz_w < (1L << 31);
    if (variant == 0 && p.nseg == 1 && p.Nq >= 256 && window_ok &&
        (p.dtype == GEO4D_BF16 || p.dtype == GEO4D_F16 || (p.dtype == GEO4D_BF16X3 && p.qkv_split)))
        variant = (p.dtype == GEO4D_BF16X3 && p.Nq >= 1024) ? 5 : 4;
    if (variant >= 4) {
        const bool ok16 = (p.dtype == GEO4D_BF16 || p.dtype == GEO4D_F16) && variant == 4;
        const bool okx3 = p.dtype == GEO4D_BF16X3 && p.qkv_split;
        if (p.nseg != 1 || !(ok16 || okx3)) { geo4d_set_error("attention: variants 4 / 5 take one key/value set in bf16 / f16 (4) or pre-split bf16x3"); return GEO4D_EINVAL; }
        if (!window_ok) { geo4d_set_error("attention: variants 4 / 5 need one (batch, head)'s keys and V^T rows inside a 2 GB window"); return GEO4D_EINVAL; }
        const dim3 grid2((p.Nq + 255) / 256, p.H, p.B);
        if (p.dtype == GEO4D_BF16) hipLaunchKernelGGL((flash_attn2_kernel<bf16_t, false, 2>), grid2, dim3(256), 0, st, p);
        else if (p.dtype == GEO4D_F16) hipLaunchKernelGGL((flash_attn2_kernel<f16_t, false, 2>), grid2, dim3(256), 0, st, p);
        else if (variant == 4) hipLaunchKernelGGL((flash_attn2_kernel<bf16x3_t, true, 1>), grid2, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((flash_attn2_kernel<bf16x3_t, true, 2>), grid2, dim3(256), 0, st, p);
        GEO4D_CHECK_LAUNCH();
        return GEO4D_OK;
    }
    // default (measured, profiles/r02_attention_variants.md): two query blocks per wave for 16-bit self-attention (bf16 N = 2560:
    // 245 -> 199 us), one for the 4-byte storage modes (their two-block build spills: bf16x3 523 -> 563 us)
    if (variant == 0) variant = (p.nseg == 1 && (p.dtype == GEO4D_BF16 || p.dtype == GEO4D_F16)) ? 3 : 1;
    if (p.nseg == 2 || p.dtype == GEO4D_F32 || p.dtype == GEO4D_BF16X3) {
        if (variant == 3 && p.nseg == 2) variant = 1;     // dual-KV cross attention: one query block per wave
        if (variant == 2 && (p.dtype == GEO4D_F32 || p.dtype == GEO4D_BF16X3)) variant = 1;   // 4-byte storage needs > 128 VGPRs
    }
    const int rows = variant == 3 ? 256 : 128;
    const dim3 grid((p.Nq + rows - 1) / rows, p.H, p.B);
#define ATT_LAUNCH(TT, NS, QB_, OCC_) hipLaunchKernelGGL((flash_attn_kernel<TT, NS, QB_, OCC_>), grid, dim3(256), 0, st, p)
#define ATT_TYPED(TT)                                                        \
    do {                                                                     \
        if (p.nseg == 2) { ATT_LAUNCH(TT, 2, 1, 1); }                        \
        else if (variant == 3) { ATT_LAUNCH(TT, 1, 2, 2); }                  \
        else { ATT_LAUNCH(TT, 1, 1, 1); }                                    \
    } while (0)
    switch (p.dtype) {
        case GEO4D_F32: ATT_TYPED(float); break;
        case GEO4D_BF16X3:
            if (p.qkv_split && variant == 3) hipLaunchKernelGGL((flash_attn_kernel<bf16x3_t, 1, 2, 2, true>), grid, dim3(256), 0, st, p);
            else if (p.qkv_split) hipLaunchKernelGGL((flash_attn_kernel<bf16x3_t, 1, 1, 1, true>), grid, dim3(256), 0, st, p);
            else ATT_TYPED(bf16x3_t);
            break;
        case GEO4D_BF16:
            if (p.nseg == 1 && variant == 2) ATT_LAUNCH(bf16_t, 1, 1, 4);
            else ATT_TYPED(bf16_t);
            break;
        default:
            if (p.nseg == 1 && variant == 2) ATT_LAUNCH(f16_t, 1, 1, 4);
            else ATT_TYPED(f16_t);
            break;
    }
#undef ATT_TYPED
#undef ATT_LAUNCH
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_temporal_attention(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                                        int B, int T, int HW, int H, int head_dim, float scale, int dtype, void* stream) {
    return geo4d_temporal_attention2(q, ldq, k, ldk, v, ldv, o, ldo, B, T, HW, H, head_dim, scale, dtype, 0, stream);
}
extern "C" int geo4d_temporal_attention2(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                                         int B, int T, int HW, int H, int head_dim, float scale, int dtype, int split_out, void* stream) {
    const int esz = dtype == GEO4D_F32 ? 4 : 2;
    if (split_out && dtype != GEO4D_F32) { geo4d_set_error("temporal_attention: split_out needs f32 storage"); return GEO4D_EINVAL; }
    if (dtype < 0 || dtype > 2 || B <= 0 || T <= 0 || HW <= 0 || H <= 0) { geo4d_set_error("temporal_attention: bad arguments"); return GEO4D_EINVAL; }
    if (T > 16) { geo4d_set_error("temporal_attention: T > 16 not built (yaml temporal_length: 16)"); return GEO4D_ENOTSUP; }
    if (head_dim != 64) { geo4d_set_error("temporal_attention: only d_head = 64 is built"); return GEO4D_ENOTSUP; }
    if ((ldq * esz) % 16 || (ldk * esz) % 16 || (ldv * esz) % 16 || (ldo * esz) % 16 || ((uintptr_t)q % 16) || ((uintptr_t)k % 16) || ((uintptr_t)v % 16) || ((uintptr_t)o % 16)) {
        geo4d_set_error("temporal_attention: alignment");
        return GEO4D_EINVAL;
    }
    const long units = (long)B * HW * H;
    const dim3 grid((unsigned)((units + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case GEO4D_F32: hipLaunchKernelGGL(temporal_attn_kernel<float>, grid, dim3(256), 0, st, (const float*)q, ldq, (const float*)k, ldk, (const float*)v, ldv, (float*)o, ldo, B, T, HW, H, scale, split_out); break;
        case GEO4D_BF16: hipLaunchKernelGGL(temporal_attn_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, B, T, HW, H, scale, 0); break;
        default: hipLaunchKernelGGL(temporal_attn_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)q, ldq, (const f16_t*)k, ldk, (const f16_t*)v, ldv, (f16_t*)o, ldo, B, T, HW, H, scale, 0); break;
    }
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}
