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

// f32 rows -> f16 rows (8 elements per lane: two 16-byte loads, one 16-byte store), clamped to the finite f16 range, NaN kept
__global__ __launch_bounds__(256) void cast_rows_f16_kernel(const float* __restrict__ x, long ldx, unsigned short* __restrict__ y, long ldy, long M,
                                                            int c8, unsigned long long* sat) {
    const long total = M * c8;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / c8;
        const int c = (int)(idx - m * c8) * 8;
        const u32x4 a = *(const u32x4*)(x + m * ldx + c), b = *(const u32x4*)(x + m * ldx + c + 4);
        const float e0[4] = {__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3])};
        const float e1[4] = {__uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3])};
        count_f16_saturation(sat, e0);
        count_f16_saturation(sat, e1);
        const u32x2 h0 = pack4_f16_sat(e0), h1 = pack4_f16_sat(e1);
        *(u32x4*)(y + m * ldy + c) = u32x4{h0[0], h0[1], h1[0], h1[1]};
    }
}

// f32 rows -> the bf16x3 PRE-SPLIT operand format ([8 x bf16 hi | 8 x bf16 lo] per 8 elements: what conv_gemm's in-register split computes per
// fragment, done once): 8 elements per lane, two 16-byte loads, two 16-byte stores
__global__ __launch_bounds__(256) void split_rows_bf16_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy, long M, int c8) {
    const long total = M * c8;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / c8;
        const int g = (int)(idx - m * c8);
        const u32x4 a = *(const u32x4*)(x + m * ldx + g * 8), b = *(const u32x4*)(x + m * ldx + g * 8 + 4);
        const float e[8] = {__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3]),
                            __uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3])};
        store_split8(y + m * ldy, g, e);
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

extern "C" int geo4d_cast_rows_f16(const float* x, long ldx, void* y, long ldy, long M, int C, unsigned long long* sat_count, void* stream) {
    if (M <= 0 || C <= 0 || (C % 8) || (ldx % 4) || (ldy % 8) || ((uintptr_t)x % 16) || ((uintptr_t)y % 16)) {
        geo4d_set_error("cast_rows_f16: C % 8 == 0, 16-byte aligned rows");
        return GEO4D_EINVAL;
    }
    const long total = M * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(cast_rows_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, (unsigned short*)y, ldy, M, C / 8, sat_count);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_split_rows_bf16(const float* x, long ldx, void* y, long ldy, long M, int C, void* stream) {
    if (M <= 0 || C <= 0 || (C % 8) || (ldx % 4) || (ldy % 8) || ((uintptr_t)x % 16) || ((uintptr_t)y % 32)) {
        geo4d_set_error("split_rows_bf16: C % 8 == 0, 16-byte aligned input rows, 32-byte aligned output rows");
        return GEO4D_EINVAL;
    }
    const long total = M * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(split_rows_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, (float*)y, ldy, M, C / 8);
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

// ---- classifier-free guidance: combine 2 (or 3) U-Net outputs + rescale_noise_cfg ------------------------------------------
// out = e_u + cfg_img (e_i - e_u) + scale (e_c - e_i)     (e_i == nullptr: out = e_u + scale (e_c - e_u))
// pass 1 writes the combination and the per-sample sums of e_c, e_c^2, out, out^2 (fp64 partials per chunk, no atomics);
// pass 2 (guidance_rescale > 0) merges the partials in a fixed order, forms the unbiased std ratio and blends in place.
constexpr int CFG_CHUNKS = 64;

__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* __restrict__ ec, const float* __restrict__ eu, const float* __restrict__ ei,
                                                          float* __restrict__ out, long n, float scale, float cfg_img, double* __restrict__ part) {
    __shared__ double red[4][4];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const long per = (n + CFG_CHUNKS - 1) / CFG_CHUNKS;
    const long i0 = (long)chunk * per, i1 = min(n, i0 + per);
    const long base = (long)b * n;
    double sc = 0.0, qc = 0.0, so = 0.0, qo = 0.0;
    for (long i = i0 + tid; i < i1; i += 256) {
        const float c = ec[base + i], u = eu[base + i];
        float o;
        if (ei) {
            const float im = ei[base + i];
            o = u + cfg_img * (im - u) + scale * (c - im);
        } else {
            o = u + scale * (c - u);
        }
        out[base + i] = o;
        sc += c; qc += (double)c * c; so += o; qo += (double)o * o;
    }
    double v[4] = {sc, qc, so, qo};
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double x = v[k];
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) x += __shfl_down(x, o2);
        if (lane == 0) red[wave][k] = x;
    }
    __syncthreads();
    if (tid < 4) part[((long)b * CFG_CHUNKS + chunk) * 4 + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

__global__ __launch_bounds__(256) void cfg_rescale_kernel(float* __restrict__ out, long n, float rescale, const double* __restrict__ part) {
    __shared__ float ratio_s;
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid == 0) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int c = 0; c < CFG_CHUNKS; ++c)
            for (int k = 0; k < 4; ++k) s[k] += part[((long)b * CFG_CHUNKS + c) * 4 + k];
        const double dn = (double)n;
        const double var_c = (s[1] - s[0] * s[0] / dn) / (dn - 1.0), var_o = (s[3] - s[2] * s[2] / dn) / (dn - 1.0);   // torch.std: unbiased
        ratio_s = (float)sqrt(var_c > 0.0 ? var_c : 0.0) / (float)sqrt(var_o > 0.0 ? var_o : 0.0);
    }
    __syncthreads();
    const float ratio = ratio_s;
    const long base = (long)b * n;
    for (long i = (long)blockIdx.x * 256 + tid; i < n; i += (long)gridDim.x * 256) {
        const float o = out[base + i];
        out[base + i] = rescale * (o * ratio) + (1.0f - rescale) * o;
    }
}

extern "C" size_t geo4d_cfg_combine_workspace(int B) { return (size_t)B * CFG_CHUNKS * 4 * sizeof(double); }

extern "C" int geo4d_cfg_combine(const float* e_c, const float* e_u, const float* e_i, float* out, int B, long n, float scale, float cfg_img,
                                 float guidance_rescale, void* workspace, size_t workspace_bytes, void* stream) {
    if (!e_c || !e_u || !out || B <= 0 || B > 65535 || n <= 1) { geo4d_set_error("cfg_combine: bad arguments"); return GEO4D_EINVAL; }
    if (!workspace || workspace_bytes < geo4d_cfg_combine_workspace(B) || ((uintptr_t)workspace & 7)) { geo4d_set_error("cfg_combine: workspace too small / unaligned"); return GEO4D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(CFG_CHUNKS, B), dim3(256), 0, s, e_c, e_u, e_i, out, n, scale, cfg_img, (double*)workspace);
    GEO4D_CHECK_LAUNCH();
    if (guidance_rescale > 0.f) {
        long blocks = (n + 255) / 256;
        if (blocks > 256) blocks = 256;
        hipLaunchKernelGGL(cfg_rescale_kernel, dim3((unsigned)blocks, B), dim3(256), 0, s, out, n, guidance_rescale, (const double*)workspace);
        GEO4D_CHECK_LAUNCH();
    }
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

// ---- token embedding lookup + positional embedding (OpenCLIP text tower, condition.py:212-214) ----------------------------
// out[(b n), :] = table[tokens[b][n], :] + pos[n, :]   (fp32 tables, output in the engine's storage dtype)
template <typename T>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const long* __restrict__ tokens, const float* __restrict__ table,
                                                           const float* __restrict__ pos, T* __restrict__ out, long ldo, int n_ctx,
                                                           int width, int vocab) {
    const long row = blockIdx.x;
    long tok = tokens[row];
    if (tok < 0 || tok >= vocab) tok = 0;
    const float* tr = table + tok * width;
    const float* pr = pos + (row % n_ctx) * (long)width;
    for (int c = threadIdx.x; c < width; c += 256) Elem<T>::st(out + row * ldo + c, tr[c] + pr[c]);
}

extern "C" int geo4d_embed_tokens(const long* tokens, const float* table, const float* pos, void* out, long ldo, long rows, int n_ctx,
                                  int width, int vocab, int dtype, void* stream) {
    if (!tokens || !table || !pos || !out || rows <= 0 || rows > 2147483647L || n_ctx <= 0 || width <= 0 || vocab <= 0 || dtype < 0 || dtype > 2) {
        geo4d_set_error("embed_tokens: bad arguments");
        return GEO4D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)rows);
    switch (dtype) {
        case GEO4D_F32: hipLaunchKernelGGL(embed_tokens_kernel<float>, grid, dim3(256), 0, s, tokens, table, pos, (float*)out, ldo, n_ctx, width, vocab); break;
        case GEO4D_BF16: hipLaunchKernelGGL(embed_tokens_kernel<bf16_t>, grid, dim3(256), 0, s, tokens, table, pos, (bf16_t*)out, ldo, n_ctx, width, vocab); break;
        default: hipLaunchKernelGGL(embed_tokens_kernel<f16_t>, grid, dim3(256), 0, s, tokens, table, pos, (f16_t*)out, ldo, n_ctx, width, vocab); break;
    }
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
extern "C" size_t geo4d_abi_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(geo4d_conv_gemm_t);
        case 1: return sizeof(geo4d_groupnorm_t);
        case 2: return sizeof(geo4d_attention_t);
        case 3: return sizeof(geo4d_align_t);
        case 4: return sizeof(geo4d_align_small_t);
        default: return 0;
    }
}
