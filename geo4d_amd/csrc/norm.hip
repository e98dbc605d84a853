// geo4d_amd/csrc/norm.hip — HBM-bound normalisation kernels:
//   GroupNorm(32) [+SiLU] over channels-last tokens, both flavours used by the reference:
//     * per-frame 4-D   (basics.py:76-87 GroupNormSpecific in ResBlock / `out`; attention.py:265 SpatialTransformer;
//                         ae_modules.py:15-16 VAE Normalize)            -> frames_per_stat = 1
//     * across-time 5-D  (openaimodel3d.py:256-266 TemporalConvBlock; attention.py:331,368 TemporalTransformer)
//                                                                        -> frames_per_stat = T
//   LayerNorm over C     (attention.py:225-227 BasicTransformerBlock norm1/2/3)
//   row softmax          (ae_modules.py:66-68 VAE AttnBlock, scores kept in fp32)
// Statistics are always fp32 (partials merged with Chan's formula in fp64), activations are read and written
// as 16-byte chunks (8 x bf16/f16 or 4 x f32 per lane) — cdna_hip_programming.md G13.
// All reductions are order-deterministic (no atomics): the same input gives bit-identical output.
#include <type_traits>
#include "common.h"
#include "geo4d_hip.h"

namespace {

// rows of one frame per workgroup of the statistics / apply launches. Round 4: sized so that a launch has ~1024 workgroups whatever the
// level - the former HW / 64 with a floor of 32 rows left the 10x16 and 5x8 levels of the U-Net with 80 and 32 workgroups on 256 CUs
// (their three launches cost ~10 us each: 86 of the 166 GroupNorms of a forward, profiles/r04_groupnorm_chunks.md); a multiple of 4
// rows (the 4 loads in flight per lane), at most 1024.
__host__ __device__ inline int gn_rows_per_chunk(int HW, int F) {
    long r = ((long)HW * F + 1023) / 1024;
    r = (r + 3) / 4 * 4;
    if (r < 4) r = 4;
    if (r > 1024) r = 1024;
    return (int)r;
}

// ---- GroupNorm pass 1: per (frame, row-chunk, group) partial (n, mean, M2) ----------------
template <typename T>
__device__ __forceinline__ void gn_partial_body(char* smem, const T* __restrict__ x, long ldx, int HW, int C, int G,
                                                int R, int nchunk, float* __restrict__ part, int chunk, int f) {
    constexpr int EPC = Elem<T>::EPC;
    float* stage = (float*)smem;               // [256][2*EPC]
    float* chs = stage + 256 * 2 * EPC;        // [C] channel sums
    float* chq = chs + C;                      // [C] channel sums of squares
    const int tid = threadIdx.x;
    const int CPR = C / EPC;
    const int TC = CPR < 256 ? CPR : 256;
    const int TR = 256 / TC;
    const int tc = tid % TC, tr = tid / TC;
    const bool active = tr < TR;
    const int row0 = chunk * R;
    const int rows = min(R, HW - row0);
    const T* base = x + ((long)f * HW + row0) * ldx;
    for (int cc0 = 0; cc0 < CPR; cc0 += TC) {
        const int cc = cc0 + tc;
        float s[EPC], q[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) { s[j] = 0.f; q[j] = 0.f; }
        if (active && cc < CPR) {
            for (int r = tr; r < rows; r += 4 * TR) {
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {   // 4 independent 16-byte loads in flight per lane
                    const int ru = r + u * TR;
                    v[u] = ru < rows ? *(const u32x4*)(base + (long)ru * ldx + cc * EPC) : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float e[EPC];
                    chunk_to_f32<T>(v[u], e);
#pragma unroll
                    for (int j = 0; j < EPC; ++j) { s[j] += e[j]; q[j] += e[j] * e[j]; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < EPC; ++j) { stage[tid * 2 * EPC + j] = s[j]; stage[tid * 2 * EPC + EPC + j] = q[j]; }
        __syncthreads();
        if (tr == 0 && cc < CPR) {
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                float ss = 0.f, qq = 0.f;
                for (int t2 = 0; t2 < TR; ++t2) {
                    ss += stage[(t2 * TC + tc) * 2 * EPC + j];
                    qq += stage[(t2 * TC + tc) * 2 * EPC + EPC + j];
                }
                chs[cc * EPC + j] = ss;
                chq[cc * EPC + j] = qq;
            }
        }
        __syncthreads();
    }
    if (tid < G) {
        const int cpg = C / G;
        float ss = 0.f, qq = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { ss += chs[c]; qq += chq[c]; }
        const float n = (float)rows * (float)cpg;
        const float mean = ss / n;
        float m2 = qq - ss * mean;
        if (m2 < 0.f) m2 = 0.f;
        float* o = part + (((long)f * nchunk + chunk) * G + tid) * 3;
        o[0] = n; o[1] = mean; o[2] = m2;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, long ldx, int HW, int C, int G,
                                                         int R, int nchunk, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gn_partial_body<T>(smem, x, ldx, HW, C, G, R, nchunk, part, blockIdx.x, blockIdx.y);
}

// ---- GroupNorm pass 2: merge partials -> (mean, rstd) per (stat, group) --------------------
// one wave per (stat, group): lanes merge strided subsets with Chan's formula, then a fixed-order xor-butterfly merges
// the 64 lane results (deterministic: the pairing does not depend on timing)
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float qb) {
    if (nb > 0.f) {
        const float nn = n + nb;
        const float d = mb - mean;
        const float w = nb / nn;
        mean += d * w;
        m2 += qb + d * d * n * w;
        n = nn;
    }
}
// one wave merges the fps * nchunk partials of (stat, grp); lane 0 returns (mean, rstd)
__device__ __forceinline__ f32x2 gn_merge_unit(const float* __restrict__ part, int total, int stat, int grp, int G, float eps, int lane) {
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int i = lane; i < total; i += 64) {
        const float* pp = part + (((long)stat * total + i) * G + grp) * 3;
        chan_merge(n, mean, m2, pp[0], pp[1], pp[2]);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float nb = __shfl_xor(n, o), mb = __shfl_xor(mean, o), qb = __shfl_xor(m2, o);
        // both partners must compute the same merged value: order the pair by lane id
        float n0 = n, a0 = mean, q0 = m2, n1 = nb, a1 = mb, q1 = qb;
        if (lane & o) { n0 = nb; a0 = mb; q0 = qb; n1 = n; a1 = mean; q1 = m2; }
        chan_merge(n0, a0, q0, n1, a1, q1);
        n = n0; mean = a0; m2 = q0;
    }
    return f32x2{mean, 1.0f / sqrtf(m2 / n + eps)};
}
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, int nchunk, int fps, int G,
                                                          float eps, float* __restrict__ stats, int nstat) {
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= nstat * G) return;
    const int stat = unit / G, grp = unit - stat * G;
    const f32x2 r = gn_merge_unit(part, fps * nchunk, stat, grp, G, eps, lane);
    if (lane == 0) {
        stats[((long)stat * G + grp) * 2 + 0] = r[0];
        stats[((long)stat * G + grp) * 2 + 1] = r[1];
    }
}

// ---- GroupNorm pass 2': the same merge fed by the column sums the producing conv_gemm's epilogue wrote (gn_colsum): one item
// per (32-row block, channel of the group) = (n = 32, mean = s / 32, M2 = q - s * mean); replaces pass 1 + pass 2 ----------------
__global__ __launch_bounds__(256) void gn_finalize_cols_kernel(const float* __restrict__ colsum, int blocks_per_stat, int C, int G, float eps,
                                                               float* __restrict__ stats, int nstat, int rows) {
    // one WORKGROUP per (stat, group): a 5-D GroupNorm at level 0 has 32 units of 12800 items each, too few and too long for one
    // wave per unit (first version: slower than the statistics pass it replaced).
    // Round 6: the items are raw (sum, sum of squares) pairs, so they are simply ADDED - in fp64, fixed order (strided per thread, xor tree in the
    // wave, the 4 waves in order) - and mean / variance come from the two totals at the end. (Rounds 4-5 turned every item into (n, mean, M2)
    // and Chan-merged them: a division per item in a dependent chain for no accuracy - each item's M2 was already q - s * mean in fp32 - and the
    // kernel ran 7.8 us for a few hundred items.)
    __shared__ double red[4][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int unit = blockIdx.x;
    const int stat = unit / G, grp = unit - stat * G;
    const int cpg = C / G;
    double s = 0.0, q = 0.0;
    const int total = blocks_per_stat * cpg;
    for (int i = tid; i < total; i += 256) {
        const int rb = i / cpg, c = grp * cpg + (i - rb * cpg);
        const f32x2 sq = *(const f32x2*)(colsum + (((long)stat * blocks_per_stat + rb) * C + c) * 2);
        s += (double)sq[0];
        q += (double)sq[1];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    if (lane == 0) { red[wave][0] = s; red[wave][1] = q; }
    __syncthreads();
    if (tid == 0) {
        s = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];      // fixed order
        q = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
        const double n = (double)blocks_per_stat * rows * cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[((long)stat * G + grp) * 2 + 0] = (float)mean;
        stats[((long)stat * G + grp) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// ---- GroupNorm pass 3: y = (x - mean) * rstd * gamma + beta, optional SiLU -----------------
// st: (mean, rstd) per group of this frame's statistic (global memory or LDS)
template <typename T, int SPLIT>      // SPLIT: 0 plain rows, 1 pre-split bf16 hi | lo (bf16x3 consumers), 2 plain f16 rows from f32 input (f16x2 consumers; ldy in f16 elements)
__device__ __forceinline__ void gn_apply_body(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                              int HW, int C, int G, int R, const float* st, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, int act, int chunk, int f, unsigned long long* sat) {
    constexpr int EPC = Elem<T>::EPC;
    const int tid = threadIdx.x;
    const int CPR = C / EPC;
    const int TC = CPR < 256 ? CPR : 256;
    const int TR = 256 / TC;
    const int tc = tid % TC, tr = tid / TC;
    if (tr >= TR) return;
    const int row0 = chunk * R;
    const int rows = min(R, HW - row0);
    const int cpg = C / G;
    const T* xb = x + ((long)f * HW + row0) * ldx;
    T* yb = y + ((long)f * HW + row0) * ldy;
    for (int cc = tc; cc < CPR; cc += TC) {
        float sc[EPC], sh[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            const int c = cc * EPC + j;
            const int gi = c / cpg;
            const float mean = st[gi * 2], rstd = st[gi * 2 + 1];
            sc[j] = rstd * gamma[c];
            sh[j] = beta[c] - mean * sc[j];
        }
        for (int r = tr; r < rows; r += 4 * TR) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ru = r + u * TR;
                v[u] = ru < rows ? *(const u32x4*)(xb + (long)ru * ldx + cc * EPC) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ru = r + u * TR;
                if (ru >= rows) break;
                float e[EPC];
                chunk_to_f32<T>(v[u], e);
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    float o = e[j] * sc[j] + sh[j];
                    if (act == 1) o = silu_f(o);
                    e[j] = o;
                }
                if constexpr (SPLIT == 1) store_split4(yb + (long)ru * ldy, cc, e);          // (ldy counts K elements of 4 bytes, like ldx)
                else if constexpr (SPLIT == 2) {
                    count_f16_saturation(sat, e);
                    store4_f16((char*)y + ((long)f * HW + row0 + ru) * ldy * 2, cc, e);
                } else *(u32x4*)(yb + (long)ru * ldy + cc * EPC) = f32_to_chunk<T>(e);
            }
        }
    }
}
template <typename T, int SPLIT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                       int HW, int C, int G, int fps, int R,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int act, unsigned long long* sat) {
    const int f = blockIdx.y;
    gn_apply_body<T, SPLIT>(x, ldx, y, ldy, HW, C, G, R, stats + (long)(f / fps) * G * 2, gamma, beta, act, blockIdx.x, f, sat);
}

// ---- LayerNorm: one wave per row, row held in registers ------------------------------------
template <typename T, int MAXC, int SPLIT = 0>  // MAXC = chunks per lane; SPLIT (f32 only): 1 = write the pre-split bf16 hi | lo operand format, 2 = plain f16 rows (ldy in f16 elements)
__global__ __launch_bounds__(256) void ln_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int M, int C,
                                                 float eps, const float* __restrict__ gamma, const float* __restrict__ beta, unsigned long long* sat) {
    constexpr int EPC = Elem<T>::EPC;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int CPR = C / EPC;
    float e[MAXC][EPC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int cc = lane + i * 64;
        if (cc < CPR) {
            const u32x4 v = *(const u32x4*)(x + row * ldx + cc * EPC);
            chunk_to_f32<T>(v, e[i]);
#pragma unroll
            for (int j = 0; j < EPC; ++j) s += e[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < EPC; ++j) e[i][j] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int cc = lane + i * 64;
        if (cc < CPR) {
#pragma unroll
            for (int j = 0; j < EPC; ++j) { const float d = e[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int cc = lane + i * 64;
        if (cc < CPR) {
            float o[EPC];
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                const int c = cc * EPC + j;
                o[j] = (e[i][j] - mean) * rstd * gamma[c] + beta[c];
            }
            if constexpr (SPLIT == 1) store_split4(y + row * ldy, cc, o);
            else if constexpr (SPLIT == 2) {
                count_f16_saturation(sat, o);
                store4_f16((char*)y + row * ldy * 2, cc, o);
            }
            else *(u32x4*)(y + row * ldy + cc * EPC) = f32_to_chunk<T>(o);
        }
    }
}

// ---- LayerNorm, 16 lanes per row (round 6): the widths of U-Net levels 0 / 1 (320 / 640 f32 channels = 80 / 160 16-byte chunks) are
// multiples of 16 chunks but not of 64: the one-wave-per-row kernel above leaves 3/8 of its lanes idle at C = 320 (80 chunks on 64 lanes: two passes,
// the second with 16 lanes). Here a wave owns FOUR rows, a quarter-wave each: lane (row = lane >> 4, l = lane & 15) holds chunks l, l + 16, ... of its
// row (every load instruction of a quarter-wave = 256 contiguous bytes), statistics by a fixed DPP tree inside the 16 lanes. f32 input only (the
// storage type of the modes that use it); same outputs as ln_kernel up to the summation order of the statistics.
__device__ __forceinline__ float quarter_sum(float v) {      // sum over the 16 lanes of a DPP row, every lane gets the total
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));     // row_mirror
    return v;
}
template <int NCH, int SPLIT>        // NCH = chunks per lane (C = 64 NCH); SPLIT as ln_kernel
__global__ __launch_bounds__(256) void ln16_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy, int M,
                                                   float eps, const float* __restrict__ gamma, const float* __restrict__ beta, unsigned long long* sat) {
    constexpr int C = 64 * NCH;
    const int lane = threadIdx.x & 63, l = lane & 15;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool ok = row < M;
    const float* xr = x + (ok ? row : 0) * ldx;
    float e[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const u32x4 v = *(const u32x4*)(xr + (l + 16 * i) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { e[i][j] = __uint_as_float(v[j]); s += e[i][j]; }
    }
    const float mean = quarter_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = e[i][j] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(quarter_sum(q) / (float)C + eps);
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cc = l + 16 * i;
        const f32x4 g = *(const f32x4*)(gamma + cc * 4), b = *(const f32x4*)(beta + cc * 4);
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (e[i][j] - mean) * rstd * g[j] + b[j];
        if constexpr (SPLIT == 1) store_split4(y + row * ldy, cc, o);
        else if constexpr (SPLIT == 2) {
            count_f16_saturation(sat, o);
            store4_f16((char*)y + row * ldy * 2, cc, o);
        } else *(f32x4*)(y + row * ldy + cc * 4) = f32x4{o[0], o[1], o[2], o[3]};
    }
}

// ---- row softmax: y = softmax(scale * x), x fp32, one workgroup per row --------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                           int cols_all, float scale, int causal_period) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const float* xr = x + (long)blockIdx.x * ldx;
    T* yr = y + (long)blockIdx.x * ldy;
    // causal: row r of each `causal_period`-row score matrix sees columns 0 .. r; the masked tail is written as exact zeros
    const int cols = causal_period > 0 ? min(cols_all, (int)(blockIdx.x % causal_period) + 1) : cols_all;
    for (int c = cols + tid; c < cols_all; c += 256) Elem<T>::st(yr + c, 0.f);
    float m = -INFINITY;
    for (int c = tid; c < cols; c += 256) m = fmaxf(m, xr[c] * scale);
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = tid; c < cols; c += 256) s += __expf(xr[c] * scale - m);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / s;
    for (int c = tid; c < cols; c += 256) Elem<T>::st(yr + c, __expf(xr[c] * scale - m) * inv);
}

template <typename T, int SPLIT>
int groupnorm_typed(const geo4d_groupnorm_t& p, hipStream_t s) {
    constexpr int EPC = Elem<T>::EPC;
    int R = gn_rows_per_chunk(p.HW, p.F);
    int nchunk = (p.HW + R - 1) / R;
    float* part = (float*)p.workspace;
    const size_t smem = (size_t)256 * 2 * EPC * 4 + (size_t)2 * p.C * 4;
    const int nstat = p.F / p.frames_per_stat;
    float* stats = part + (size_t)p.F * nchunk * p.groups * 3;
    if (p.colsum) {     // statistics already summed per block of `colsum_rows` rows by the producing GEMM's epilogue: no pass over x
        const int crows = p.colsum_rows > 0 ? p.colsum_rows : 32;
        hipLaunchKernelGGL(gn_finalize_cols_kernel, dim3(nstat * p.groups), dim3(256), 0, s, p.colsum, (p.frames_per_stat * p.HW) / crows,
                           p.C, p.groups, p.eps, stats, nstat, crows);
        GEO4D_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(gn_partial_kernel<T>, dim3(nchunk, p.F), dim3(256), smem, s, (const T*)p.x, (long)p.ldx, p.HW, p.C, p.groups,
                           R, nchunk, part);
        GEO4D_CHECK_LAUNCH();
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((nstat * p.groups + 3) / 4), dim3(256), 0, s, part, nchunk, p.frames_per_stat,
                           p.groups, p.eps, stats, nstat);
        GEO4D_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL((gn_apply_kernel<T, SPLIT>), dim3(nchunk, p.F), dim3(256), 0, s, (const T*)p.x, (long)p.ldx, (T*)p.y, (long)p.ldy, p.HW,
                       p.C, p.groups, p.frames_per_stat, R, stats, p.gamma, p.beta, p.act, SPLIT == 2 ? p.sat_count : nullptr);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

template <typename T, int SPLIT = 0>
int layernorm_typed(const void* x, long ldx, void* y, long ldy, int M, int C, float eps, const float* g, const float* b,
                    hipStream_t s, unsigned long long* sat = nullptr) {
    constexpr int EPC = Elem<T>::EPC;
    if constexpr (std::is_same<T, float>::value) {
        // f32 rows whose width is a multiple of 64 channels (every width of the U-Net): four rows per wave, no idle lanes
        const bool al = ((uintptr_t)g % 16) == 0 && ((uintptr_t)b % 16) == 0;
        const dim3 grid16((M + 15) / 16);
#define LN16_LAUNCH(NC) hipLaunchKernelGGL((ln16_kernel<NC, SPLIT>), grid16, dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, M, eps, g, b, sat)
        // measured in the HEAD trace (us per launch, one-wave-per-row -> this kernel): C = 320 at M = 40960 20.4 -> 15.1, C = 640 at M = 10240 11.8 -> 10.6;
        // C = 1280 at M = 2560 7.8 -> 12.7 (160 workgroups of long per-lane chains: it stays on ln_kernel, as does C = 512 = exactly 2 chunks per lane there)
        if (al && C == 320) { LN16_LAUNCH(5); GEO4D_CHECK_LAUNCH(); return GEO4D_OK; }
        if (al && C == 640) { LN16_LAUNCH(10); GEO4D_CHECK_LAUNCH(); return GEO4D_OK; }
#undef LN16_LAUNCH
    }
    const int per_lane = (C / EPC + 63) / 64;
    const dim3 grid((M + 3) / 4);
#define LN_LAUNCH(MC) hipLaunchKernelGGL((ln_kernel<T, MC, SPLIT>), grid, dim3(256), 0, s, (const T*)x, ldx, (T*)y, ldy, M, C, eps, g, b, sat)
    if (per_lane <= 1) LN_LAUNCH(1);
    else if (per_lane <= 2) LN_LAUNCH(2);
    else if (per_lane <= 3) LN_LAUNCH(3);
    else if (per_lane <= 5) LN_LAUNCH(5);
    else if (per_lane <= 10) LN_LAUNCH(10);
    else { geo4d_set_error("layernorm: C too large"); return GEO4D_ENOTSUP; }
#undef LN_LAUNCH
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

}  // namespace

extern "C" size_t geo4d_groupnorm_workspace(int F, int HW, int groups, int frames_per_stat) {
    const int R = gn_rows_per_chunk(HW, F);
    const int nchunk = (HW + R - 1) / R;
    return ((size_t)F * nchunk * groups * 3 + (size_t)(F / frames_per_stat) * groups * 2) * sizeof(float);
}

extern "C" int geo4d_groupnorm(const geo4d_groupnorm_t* pp, void* stream) {
    if (!pp) return GEO4D_EINVAL;
    const geo4d_groupnorm_t& p = *pp;
    const int esz = p.dtype == GEO4D_F32 ? 4 : 2, epc = 16 / esz;
    if (p.dtype < 0 || p.dtype > 2) { geo4d_set_error("groupnorm: bad dtype"); return GEO4D_EINVAL; }
    if (p.F <= 0 || p.HW <= 0 || p.C <= 0 || p.groups <= 0 || p.groups > 32 || p.C % p.groups || p.C % epc) {
        geo4d_set_error("groupnorm: need C % groups == 0, C % (16 B) == 0, groups <= 32");
        return GEO4D_EINVAL;
    }
    if (p.frames_per_stat <= 0 || p.F % p.frames_per_stat) { geo4d_set_error("groupnorm: F % frames_per_stat"); return GEO4D_EINVAL; }
    const int yesz = p.split_out == 2 ? 2 : esz;       // (split_out 2: f16 rows from f32 input)
    if ((p.ldx * esz) % 16 || (p.ldy * yesz) % 16 || ((uintptr_t)p.x % 16) || ((uintptr_t)p.y % 16)) { geo4d_set_error("groupnorm: alignment"); return GEO4D_EINVAL; }
    if (p.workspace_bytes < geo4d_groupnorm_workspace(p.F, p.HW, p.groups, p.frames_per_stat)) { geo4d_set_error("groupnorm: workspace too small"); return GEO4D_EINVAL; }
    if (p.F > 65535) { geo4d_set_error("groupnorm: too many frames"); return GEO4D_EINVAL; }
    {
        const int crows = p.colsum_rows > 0 ? p.colsum_rows : 32;      // blocks must not straddle two statistics
        if (p.colsum && ((((long)p.frames_per_stat * p.HW) % crows) || ((uintptr_t)p.colsum % 8))) {
            geo4d_set_error("groupnorm: colsum needs (frames_per_stat x HW) % colsum_rows == 0");
            return GEO4D_EINVAL;
        }
    }
    if (p.split_out && (p.dtype != GEO4D_F32 || (p.C % 8) || p.split_out > 2 || p.split_out < 0)) { geo4d_set_error("groupnorm: split_out (1 = bf16 hi | lo, 2 = plain f16 rows) is a producer format for f32 input, C % 8 == 0"); return GEO4D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    switch (p.dtype) {
        case GEO4D_F32: return p.split_out == 2 ? groupnorm_typed<float, 2>(p, s) : p.split_out ? groupnorm_typed<float, 1>(p, s) : groupnorm_typed<float, 0>(p, s);
        case GEO4D_BF16: return groupnorm_typed<bf16_t, 0>(p, s);
        default: return groupnorm_typed<f16_t, 0>(p, s);
    }
}

extern "C" int geo4d_layernorm(const void* x, long ldx, void* y, long ldy, int M, int C, float eps, const float* gamma,
                               const float* beta, int dtype, void* stream) {
    const int esz = dtype == GEO4D_F32 ? 4 : 2, epc = 16 / esz;
    if (dtype < 0 || dtype > 2 || M <= 0 || C <= 0 || C % epc || (ldx * esz) % 16 || (ldy * esz) % 16 || ((uintptr_t)x % 16) || ((uintptr_t)y % 16)) {
        geo4d_set_error("layernorm: bad arguments");
        return GEO4D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case GEO4D_F32: return layernorm_typed<float>(x, ldx, y, ldy, M, C, eps, gamma, beta, s);
        case GEO4D_BF16: return layernorm_typed<bf16_t>(x, ldx, y, ldy, M, C, eps, gamma, beta, s);
        default: return layernorm_typed<f16_t>(x, ldx, y, ldy, M, C, eps, gamma, beta, s);
    }
}

extern "C" int geo4d_layernorm_split(const void* x, long ldx, void* y, long ldy, int M, int C, float eps, const float* gamma,
                                     const float* beta, int fmt, unsigned long long* sat_count, void* stream) {
    if (M <= 0 || C <= 0 || C % 8 || (ldx * 4) % 16 || (ldy * (fmt == 2 ? 2 : 4)) % 16 || ((uintptr_t)x % 16) || ((uintptr_t)y % 16) || (fmt != 1 && fmt != 2)) {
        geo4d_set_error("layernorm_split: bad arguments (f32 input, C % 8 == 0, 16-byte aligned rows, fmt 1 = bf16 halves or 2 = plain f16 rows)");
        return GEO4D_EINVAL;
    }
    if (fmt == 2) return layernorm_typed<float, 2>(x, ldx, y, ldy, M, C, eps, gamma, beta, (hipStream_t)stream, sat_count);
    return layernorm_typed<float, 1>(x, ldx, y, ldy, M, C, eps, gamma, beta, (hipStream_t)stream);
}

static int softmax_rows_launch(const float* x, long ldx, void* y, long ldy, long rows, int cols, float scale, int out_dtype,
                               int causal_period, void* stream);
extern "C" int geo4d_softmax_rows(const float* x, long ldx, void* y, long ldy, long rows, int cols, float scale, int out_dtype,
                                  void* stream) {
    return softmax_rows_launch(x, ldx, y, ldy, rows, cols, scale, out_dtype, 0, stream);
}
extern "C" int geo4d_softmax_rows_causal(const float* x, long ldx, void* y, long ldy, long rows, int cols, float scale, int out_dtype,
                                         int causal_period, void* stream) {
    if (causal_period <= 0) { geo4d_set_error("softmax_rows_causal: causal_period must be positive"); return GEO4D_EINVAL; }
    return softmax_rows_launch(x, ldx, y, ldy, rows, cols, scale, out_dtype, causal_period, stream);
}
static int softmax_rows_launch(const float* x, long ldx, void* y, long ldy, long rows, int cols, float scale, int out_dtype,
                               int causal_period, void* stream) {
    if (rows <= 0 || cols <= 0 || out_dtype < 0 || out_dtype > 2 || rows > 2147483647L) { geo4d_set_error("softmax_rows: bad arguments"); return GEO4D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)rows);
    switch (out_dtype) {
        case GEO4D_F32: hipLaunchKernelGGL(softmax_rows_kernel<float>, grid, dim3(256), 0, s, x, ldx, (float*)y, ldy, cols, scale, causal_period); break;
        case GEO4D_BF16: hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, grid, dim3(256), 0, s, x, ldx, (bf16_t*)y, ldy, cols, scale, causal_period); break;
        default: hipLaunchKernelGGL(softmax_rows_kernel<f16_t>, grid, dim3(256), 0, s, x, ldx, (f16_t*)y, ldy, cols, scale, causal_period); break;
    }
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}
