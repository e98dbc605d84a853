// geo4d_amd/csrc/elementwise.hip — layout, glue and sampler kernels (all HBM- or latency-bound).
//   tokens_from_ncthw  DiffusionWrapper 'hybrid' concat + layout change (ddpm3d.py:2540-2544; openaimodel3d.py:588)
//   concat_channels    U-Net skip connection torch.cat([h, hs.pop()], dim=1) (openaimodel3d.py:624-626)
//   timestep_embedding sinusoidal embedding [cos | sin] (utils_diffusion.py:8-28)
//   linear_small       time_embed / fps_embedding / ResBlock emb_layers (openaimodel3d.py:367-384, 166-172): M = batch rows
//   ddim_step          v-prediction DDIM update with dynamic rescale (ddim.py:206-279; ddpm3d.py:278-290)
#include "common.h"
#include "geo4d_hip.h"

namespace {

// out[(b*T + t)*HW + p][c] = src[b][c][t][p]  for c < C0 (src0) / C0 <= c < C0+C1 (src1), zero for the pad
template <typename T>
__global__ __launch_bounds__(256) void tokens_from_ncthw_kernel(const float* __restrict__ s0, int C0, const float* __restrict__ s1,
                                                                int C1, T* __restrict__ out, int Cpad, int B, int Tn, int HW) {
    const long total = (long)B * Tn * HW;
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= total) return;
    const int pix = (int)(row % HW);
    const long bt = row / HW;
    const int t = (int)(bt % Tn);
    const int b = (int)(bt / Tn);
    T* o = out + row * Cpad;
    for (int c = 0; c < Cpad; ++c) {
        float v = 0.f;
        if (c < C0) v = s0[(((long)b * C0 + c) * Tn + t) * HW + pix];
        else if (c < C0 + C1) v = s1[(((long)b * C1 + (c - C0)) * Tn + t) * HW + pix];
        Elem<T>::st(o + c, v);
    }
}

// 16-byte chunk copy: out[m] = [a[m] | b[m]]
__global__ __launch_bounds__(256) void concat_channels_kernel(const u32x4* __restrict__ a, long lda, int ca, const u32x4* __restrict__ b,
                                                              long ldb, int cb, u32x4* __restrict__ o, long ldo, long M) {
    const int cpr = ca + cb;  // chunks per output row
    const long total = M * cpr;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / cpr;
        const int c = (int)(idx - m * cpr);
        o[m * ldo + c] = c < ca ? a[m * lda + c] : b[m * ldb + (c - ca)];
    }
}

__global__ void timestep_embedding_kernel(const long* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out,
                                          int B, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i - b * half;
    const float a = (float)t[b] * freqs[j];
    out[(long)b * 2 * half + j] = cosf(a);
    out[(long)b * 2 * half + half + j] = sinf(a);
}

// one wave per output feature n, all M rows at once (M <= 8 per pass)
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w, long ldw,
                                                           const float* __restrict__ bias, const float* __restrict__ add, long ldadd,
                                                           float* __restrict__ out, long ldo, int M, int N, int K, int act_in,
                                                           int act_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    for (int m0 = 0; m0 < M; m0 += 8) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float wv = w[(long)n * ldw + k];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (m0 + i < M) {
                    float xv = x[(long)(m0 + i) * ldx + k];
                    if (act_in) xv = silu_f(xv);
                    acc[i] += xv * wv;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float s = wave_sum(acc[i]);
            if (lane == 0 && m0 + i < M) {
                float v = s + (bias ? bias[n] : 0.f);
                if (act_out) v = silu_f(v);
                if (add) v += add[(long)(m0 + i) * ldadd + n];
                out[(long)(m0 + i) * ldo + n] = v;
            }
        }
    }
}

// coef row (6 floats): sqrt(ac_t), sqrt(1-ac_t), scale_prev/scale_t, sqrt(a_prev), sqrt(1-a_prev-sigma^2), sigma
__global__ __launch_bounds__(256) void ddim_step_kernel(float* __restrict__ x, const float* __restrict__ v, const float* __restrict__ noise,
                                                        float* __restrict__ pred_x0, const float* __restrict__ coef,
                                                        const int* __restrict__ step_index, long n) {
    const float* c = coef + (long)(*step_index) * 6;
    const float sa = c[0], s1 = c[1], rs = c[2], sp = c[3], dc = c[4], sg = c[5];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float xi = x[i], vi = v[i];
        const float e_t = sa * vi + s1 * xi;     // predict_eps_from_z_and_v
        float x0 = sa * xi - s1 * vi;            // predict_start_from_z_and_v
        x0 *= rs;                                // dynamic rescale
        float xp = sp * x0 + dc * e_t;
        if (noise) xp += sg * noise[i];
        x[i] = xp;
        if (pred_x0) pred_x0[i] = x0;
    }
}

__global__ void advance_index_kernel(int* idx, int delta) { *idx += delta; }
__global__ void gather_timestep_kernel(const int* idx, const long* table, long* ts, int B) {
    const int b = threadIdx.x;
    if (b < B) ts[b] = table[*idx];
}

}  // namespace

extern "C" int geo4d_tokens_from_ncthw(const float* src0, int C0, const float* src1, int C1, void* out, int Cpad, int B, int T, int HW,
                                       int dtype, void* stream) {
    if (!src0 || C0 <= 0 || (C1 > 0 && !src1) || C1 < 0 || Cpad < C0 + C1 || B <= 0 || T <= 0 || HW <= 0 || dtype < 0 || dtype > 2) {
        geo4d_set_error("tokens_from_ncthw: bad arguments");
        return GEO4D_EINVAL;
    }
    const long total = (long)B * T * HW;
    const dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case GEO4D_F32: hipLaunchKernelGGL(tokens_from_ncthw_kernel<float>, grid, dim3(256), 0, s, src0, C0, src1, C1, (float*)out, Cpad, B, T, HW); break;
        case GEO4D_BF16: hipLaunchKernelGGL(tokens_from_ncthw_kernel<bf16_t>, grid, dim3(256), 0, s, src0, C0, src1, C1, (bf16_t*)out, Cpad, B, T, HW); break;
        default: hipLaunchKernelGGL(tokens_from_ncthw_kernel<f16_t>, grid, dim3(256), 0, s, src0, C0, src1, C1, (f16_t*)out, Cpad, B, T, HW); break;
    }
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_concat_channels(const void* a, long lda, int Ca, const void* b, long ldb, int Cb, void* out, long ldo, long M,
                                     int dtype, void* stream) {
    const int esz = dtype == GEO4D_F32 ? 4 : 2, epc = 16 / esz;
    if (dtype < 0 || dtype > 2 || M <= 0 || Ca <= 0 || Cb <= 0 || Ca % epc || Cb % epc || lda % epc || ldb % epc || ldo % epc ||
        ((uintptr_t)a % 16) || ((uintptr_t)b % 16) || ((uintptr_t)out % 16)) {
        geo4d_set_error("concat_channels: channel counts and pitches must be multiples of 16 bytes");
        return GEO4D_EINVAL;
    }
    const long total = M * ((Ca + Cb) / epc);
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(concat_channels_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)a, lda / epc,
                       Ca / epc, (const u32x4*)b, ldb / epc, Cb / epc, (u32x4*)out, ldo / epc, M);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_timestep_embedding(const long* t, const float* freqs, float* out, int B, int dim, void* stream) {
    if (!t || !freqs || !out || B <= 0 || dim <= 0 || dim % 2) { geo4d_set_error("timestep_embedding: bad arguments (dim must be even)"); return GEO4D_EINVAL; }
    const int n = B * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, freqs, out, B, dim / 2);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_linear_small(const float* x, long ldx, const float* w, long ldw, const float* bias, const float* add, long ldadd,
                                  float* out, long ldo, int M, int N, int K, int act_in, int act_out, void* stream) {
    if (!x || !w || !out || M <= 0 || N <= 0 || K <= 0) { geo4d_set_error("linear_small: bad arguments"); return GEO4D_EINVAL; }
    hipLaunchKernelGGL(linear_small_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, w, ldw, bias, add, ldadd, out,
                       ldo, M, N, K, act_in, act_out);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_ddim_step(float* x, const float* v, const float* noise, float* pred_x0, const float* coef, const int* step_index,
                               long n, void* stream) {
    if (!x || !v || !coef || !step_index || n <= 0) { geo4d_set_error("ddim_step: bad arguments"); return GEO4D_EINVAL; }
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ddim_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, v, noise, pred_x0, coef,
                       step_index, n);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_advance_index(int* idx, int delta, void* stream) {
    if (!idx) return GEO4D_EINVAL;
    hipLaunchKernelGGL(advance_index_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, idx, delta);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_gather_timestep(const int* idx, const long* table, long* ts, int B, void* stream) {
    if (!idx || !table || !ts || B <= 0 || B > 1024) { geo4d_set_error("gather_timestep: bad arguments"); return GEO4D_EINVAL; }
    hipLaunchKernelGGL(gather_timestep_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, idx, table, ts, B);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

// ---- error string / version ------------------------------------------------------------------
static thread_local char g_err[512] = "";
void geo4d_set_error(const char* msg) {
    size_t i = 0;
    for (; msg && msg[i] && i + 1 < sizeof(g_err); ++i) g_err[i] = msg[i];
    g_err[i] = 0;
}
extern "C" const char* geo4d_last_error(void) { return g_err; }
extern "C" int geo4d_abi_version(void) { return GEO4D_ABI_VERSION; }
