// geo4d_amd/csrc/attention.hip — spatiotemporal attention for the Geo4D U-Net.
//
// (1) flash_attn_kernel: softmax(Q K^T * scale) V with d_head = 64, no mask, for
//       * spatial self-attention  (attention.py:101-125 / xformers path :146-209), N in {2560,640,160,40}
//       * spatial cross-attention with TWO independent key/value sets whose normalised outputs are summed
//         (77 text tokens shared by every frame + 16 per-frame image tokens; attention.py:128-142)
//     Structure (one workgroup = 128 query rows = 4 waves x 32 rows, K/V tiles of 64 keys):
//       - swapped QK^T: S^T = K.Q^T on MFMA 32x32, so one lane owns one query column and the softmax row
//         reduction is in-register (one cross-half __shfl_xor(32) per tile) — wave-64 idiom, no LDS round trip;
//       - P never leaves registers: its C-layout registers ARE the B operand of the PV MFMA
//         (O^T = V^T.P^T); V is staged transposed in LDS so the matching A fragments are 8/16-byte reads;
//       - K rows padded to 144 B / V^T rows to 136 B: conflict-free ds_read_b128 / ds_read_b64;
//       - next K/V tile prefetched into registers while the current one is consumed.
// (2) temporal_attn_kernel: self-attention over T <= 16 frames for every (pixel, head)
//     (attention.py:365-412 TemporalTransformer, both attn1 and attn2). 0.02 TFLOP per forward but ~100 MB of
//     q/k/v/o traffic per layer: HBM-bound, so plain VALU with one wave per (pixel, head) and fp32 K/V in LDS.
//     Tokens stay in frame-major order [T, HW, C]; the kernel gathers the T rows of a pixel itself.
#include "common.h"
#include "geo4d_hip.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void flash_attn_kernel(const geo4d_attention_t p) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int ES = (int)sizeof(T);
    constexpr int DCH = 64 / EPC;                 // 16-byte chunks per 64-wide head row
    constexpr int NKK = DCH / 2;                  // cmma steps over d
    constexpr int KPITCH = 64 * ES + 16;
    constexpr int VPITCH = 64 * ES + (ES == 2 ? 8 : 16);
    constexpr int NST = 64 * DCH / 256;           // staged chunks per thread per operand
    constexpr int PCH = 16 / EPC;                 // P chunks per 32-key block
    __shared__ __attribute__((aligned(16))) char ktile[64 * KPITCH];
    __shared__ __attribute__((aligned(16))) char vtile[64 * VPITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, g = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int qrow = blockIdx.x * 128 + wave * 32 + li;
    const bool qok = qrow < p.Nq;

    u32x4 qf[NKK];
    {
        const T* qp = (const T*)p.q + ((long)b * p.Nq + (qok ? qrow : 0)) * p.ldq + h * 64;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (qok) v = *(const u32x4*)(qp + (2 * kk + g) * EPC);
            qf[kk] = v;
        }
    }
    f32x16 of[2], oa[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { of[d][r] = 0.f; oa[d][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    u32x4 kreg[NST], vreg[NST];
    auto load_tile = [&](int seg, int tile) {
        // explicit selects: dynamic indexing of kernel-argument arrays would spill them to scratch
        const int nk = seg ? p.Nk[1] : p.Nk[0];
        const long ldk = seg ? p.ldk[1] : p.ldk[0], ldv = seg ? p.ldv[1] : p.ldv[0];
        const long kvb = b / (seg ? p.kv_div[1] : p.kv_div[0]);
        const T* kp = (const T*)(seg ? p.k[1] : p.k[0]) + kvb * nk * ldk + h * 64;
        const T* vp = (const T*)(seg ? p.v[1] : p.v[0]) + kvb * nk * ldv + h * 64;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / DCH, col = idx % DCH;
            const int key = tile * 64 + row;
            u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (key < nk) {
                kv = *(const u32x4*)(kp + (long)key * ldk + col * EPC);
                vv = *(const u32x4*)(vp + (long)key * ldv + col * EPC);
            }
            kreg[i] = kv;
            vreg[i] = vv;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / DCH, col = idx % DCH;
            *(u32x4*)(ktile + row * KPITCH + col * 16) = kreg[i];
            if constexpr (ES == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    *(unsigned short*)(vtile + (col * 8 + 2 * j) * VPITCH + row * 2) = (unsigned short)(vreg[i][j] & 0xffffu);
                    *(unsigned short*)(vtile + (col * 8 + 2 * j + 1) * VPITCH + row * 2) = (unsigned short)(vreg[i][j] >> 16);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) *(unsigned int*)(vtile + (col * 4 + j) * VPITCH + row * 4) = vreg[i][j];
            }
        }
    };

    int seg = 0, tile = 0;
    load_tile(0, 0);
    while (true) {
        __syncthreads();
        store_tile();
        __syncthreads();
        const int nk = seg ? p.Nk[1] : p.Nk[0];
        const int ntile = (nk + 63) >> 6;
        int nseg2 = seg, ntile2 = tile + 1;
        if (ntile2 >= ntile) { nseg2 = seg + 1; ntile2 = 0; }
        const bool has_next = nseg2 < p.nseg;
        if (has_next) load_tile(nseg2, ntile2);

        // ---- S^T = K.Q^T --------------------------------------------------------------------
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const u32x4 a = *(const u32x4*)(ktile + (kb * 32 + li) * KPITCH + (2 * kk + g) * 16);
                cmma<T>(st[kb], a, qf[kk]);
            }
        }
        // ---- online softmax (row = this lane's query; keys live in registers) ----------------
        float mt = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = tile * 64 + kb * 32 + acc_row(r, g);
                const float s = key < nk ? st[kb][r] * p.scale : -INFINITY;
                st[kb][r] = s;
                mt = fmaxf(mt, s);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        float ls = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(st[kb][r] - m_new);
                st[kb][r] = e;
                ls += e;
            }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oa[d][r] *= alpha;
        // ---- O^T += V^T.P^T -------------------------------------------------------------------
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int c = 0; c < PCH; ++c) {
                float pv[EPC];
#pragma unroll
                for (int j = 0; j < EPC; ++j) pv[j] = st[kb][c * EPC + j];
                const u32x4 bch = f32_to_chunk<T>(pv);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    u32x4 a;
                    const char* vrow = vtile + (d * 32 + li) * VPITCH;
                    if constexpr (ES == 2) {
                        const u32x2 lo = *(const u32x2*)(vrow + (kb * 32 + 16 * c + 4 * g) * 2);
                        const u32x2 hi = *(const u32x2*)(vrow + (kb * 32 + 16 * c + 8 + 4 * g) * 2);
                        a[0] = lo[0]; a[1] = lo[1]; a[2] = hi[0]; a[3] = hi[1];
                    } else {
                        a = *(const u32x4*)(vrow + (kb * 32 + 8 * c + 4 * g) * 4);
                    }
                    cmma<T>(oa[d], a, bch);
                }
            }
        // ---- segment end: normalise and fold into the summed output ---------------------------
        if (ntile2 == 0) {
            const float lt = l_run + __shfl_xor(l_run, 32);
            const float inv = 1.0f / lt;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) { of[d][r] += oa[d][r] * inv; oa[d][r] = 0.f; }
            m_run = -INFINITY;
            l_run = 0.f;
        }
        if (!has_next) break;
        seg = nseg2;
        tile = ntile2;
    }
    if (qok) {
        T* op = (T*)p.o + ((long)b * p.Nq + qrow) * p.ldo + h * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int dcol = d * 32 + 8 * i + 4 * g;
                if constexpr (ES == 2) {
                    float e[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { e[j] = of[d][4 * i + j]; e[4 + j] = 0.f; }
                    const u32x4 c = f32_to_chunk<T>(e);
                    u32x2 o2; o2[0] = c[0]; o2[1] = c[1];
                    *(u32x2*)(op + dcol) = o2;
                } else {
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = of[d][4 * i + j];
                    *(u32x4*)(op + dcol) = f32_to_chunk<T>(e);
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
                                                            int Tn, int HW, int H, float scale) {
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
                *(u32x4*)(op + d0) = f32_to_chunk<T>(o4);
            }
        }
    }
}

}  // namespace

extern "C" int geo4d_attention(const geo4d_attention_t* pp, void* stream) {
    if (!pp) return GEO4D_EINVAL;
    const geo4d_attention_t& p = *pp;
    const int esz = p.dtype == GEO4D_F32 ? 4 : 2;
    if (p.dtype < 0 || p.dtype > 2 || p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.nseg < 1 || p.nseg > 2) { geo4d_set_error("attention: bad arguments"); return GEO4D_EINVAL; }
    if (p.head_dim != 64) { geo4d_set_error("attention: only d_head = 64 is built (yaml num_head_channels: 64)"); return GEO4D_ENOTSUP; }
    if ((p.ldq * esz) % 16 || (p.ldo * esz) % 16 || ((uintptr_t)p.q % 16) || ((uintptr_t)p.o % 16)) { geo4d_set_error("attention: q/o alignment"); return GEO4D_EINVAL; }
    for (int s = 0; s < p.nseg; ++s) {
        if (p.Nk[s] <= 0 || p.kv_div[s] <= 0 || !p.k[s] || !p.v[s] || (p.ldk[s] * esz) % 16 || (p.ldv[s] * esz) % 16 || ((uintptr_t)p.k[s] % 16) || ((uintptr_t)p.v[s] % 16)) {
            geo4d_set_error("attention: bad key/value segment");
            return GEO4D_EINVAL;
        }
    }
    if (p.H > 65535 || p.B > 65535) { geo4d_set_error("attention: grid too large"); return GEO4D_EINVAL; }
    const dim3 grid((p.Nq + 127) / 128, p.H, p.B);
    hipStream_t st = (hipStream_t)stream;
    switch (p.dtype) {
        case GEO4D_F32: hipLaunchKernelGGL(flash_attn_kernel<float>, grid, dim3(256), 0, st, p); break;
        case GEO4D_BF16: hipLaunchKernelGGL(flash_attn_kernel<bf16_t>, grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(flash_attn_kernel<f16_t>, grid, dim3(256), 0, st, p); break;
    }
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_temporal_attention(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                                        int B, int T, int HW, int H, int head_dim, float scale, int dtype, void* stream) {
    const int esz = dtype == GEO4D_F32 ? 4 : 2;
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
        case GEO4D_F32: hipLaunchKernelGGL(temporal_attn_kernel<float>, grid, dim3(256), 0, st, (const float*)q, ldq, (const float*)k, ldk, (const float*)v, ldv, (float*)o, ldo, B, T, HW, H, scale); break;
        case GEO4D_BF16: hipLaunchKernelGGL(temporal_attn_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, B, T, HW, H, scale); break;
        default: hipLaunchKernelGGL(temporal_attn_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)q, ldq, (const f16_t*)k, ldk, (const f16_t*)v, ldv, (f16_t*)o, ldo, B, T, HW, H, scale); break;
    }
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}
