// geo4d_amd/csrc/align.hip — multi-window global alignment, the per-iteration hot loop.
//
// The reference optimises, with 500 Adam iterations, log-depth maps, camera poses and a focal per image plus one sim(3) per
// 16-frame window so that every window's predicted point map, moved by its sim(3), agrees with the points re-projected from the
// per-image depth / pose / focal (dust3r/cloud_opt/optimizer_group.py:440-455 `forward`, term `li`, + depth_to_pts3d :407-417):
//     loss = 1/A * sum_{slot s = (window g, frame k), pixel p}  min(conf_s[p], 10) * | X_i(s)[p] - (sR_g P_s[p] + st_g) |
//     X_i[p] = R_i (d (u - ppx)/f_i, d (v - ppy)/f_i, d) + t_i,   d = exp(logdepth_i[p]).
// PyTorch autograd runs ~40 elementwise kernels over [G*16, H*W, 3] tensors per iteration for this (2.2 GB of traffic at 30
// windows); here ONE kernel per iteration reads every prediction once (16 B / slot-pixel) and the log-depth once and
// writes (a) the gradient with respect to the log-depth map, (b) per-(image, pixel-chunk) partial sums of dL/dR_i (3x3), dL/dt_i,
// dL/df_i and the loss, (c) per-(slot, chunk) partial sums of dL/d(sR_g) and dL/d(st_g). A second small kernel adds the chunks
// in a fixed order (deterministic: no atomics). The chain rule from those 3x3 / 3-vector sums to quaternion, log-translation,
// log-focal and log-scale parameters is a few hundred flops per image and stays on the host side (geo4d_amd/align.py), as does
// the camera temporal-smoothing term. Everything here is HBM-bound: 20 B read + 4 B written per slot-pixel.
#include "common.h"
#include "geo4d_hip.h"

namespace {

constexpr int MAXS = 6;        // windows an image can belong to (stride 4, length 16 -> 4, + the duplicated tail window = 5)
constexpr int IMG_SUMS = 14;   // dL/dR (9) | dL/dt (3) | dL/df (1) | loss (1)
constexpr int SLOT_SUMS = 12;  // dL/d(sR) (9) | dL/d(st) (3)

// fixed-order block reduction of NV values per thread: xor-butterfly inside the wave, then the 4 waves through LDS
template <int NV>
__device__ __forceinline__ void block_sum(float* v, float* red, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (tid < NV) v[0] = (red[tid] + red[NV + tid]) + (red[2 * NV + tid] + red[3 * NV + tid]);
    __syncthreads();
}

__global__ __launch_bounds__(256) void align_residual_kernel(const geo4d_align_t p) {
    __shared__ float red[4 * SLOT_SUMS > 4 * IMG_SUMS ? 4 * SLOT_SUMS : 4 * IMG_SUMS];
    const int tid = threadIdx.x, chunk = blockIdx.x, img = blockIdx.y;
    const int HW = p.H * p.W;
    const float* cam = p.cams + img * 16;       // R (9, row-major) | t (3) | f | ppx | ppy | unused
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = cam[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = cam[9 + i];
    const float f = cam[12], inv_f = 1.0f / f, ppx = cam[13], ppy = cam[14];
    const int s0 = p.slot_ptr[img], ns = p.slot_ptr[img + 1] - s0;
    float simg[IMG_SUMS];
    float sslot[MAXS][SLOT_SUMS];
#pragma unroll
    for (int i = 0; i < IMG_SUMS; ++i) simg[i] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXS; ++j)
#pragma unroll
        for (int i = 0; i < SLOT_SUMS; ++i) sslot[j][i] = 0.f;

    const int p_end = min(HW, (chunk + 1) * p.chunk_pixels);
    for (int px = chunk * p.chunk_pixels + tid; px < p_end; px += 256) {
        const float d = __expf(p.logdepth[(long)img * HW + px]);
        const int v = px / p.W, u = px - v * p.W;
        const float xc = d * ((float)u - ppx) * inv_f, yc = d * ((float)v - ppy) * inv_f, zc = d;
        const float X0 = R[0] * xc + R[1] * yc + R[2] * zc + t[0];
        const float X1 = R[3] * xc + R[4] * yc + R[5] * zc + t[1];
        const float X2 = R[6] * xc + R[7] * yc + R[8] * zc + t[2];
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXS; ++j) {
            if (j >= ns) break;
            const int slot = p.slot_idx[s0 + j];
            const float* T = p.slot_trf + slot * 12;            // sR (9) | st (3) of the slot's window
            const float3 Pv = *(const float3*)(p.pred + ((long)slot * HW + px) * 3);   // one 12-byte load per lane: 768 contiguous bytes per wave
            const float P0 = Pv.x, P1 = Pv.y, P2 = Pv.z;
            const float r0 = X0 - (T[0] * P0 + T[1] * P1 + T[2] * P2 + T[9]);
            const float r1 = X1 - (T[3] * P0 + T[4] * P1 + T[5] * P2 + T[10]);
            const float r2 = X2 - (T[6] * P0 + T[7] * P1 + T[8] * P2 + T[11]);
            const float nr = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
            const float w = fminf(p.conf[(long)slot * HW + px], p.conf_clamp) * p.inv_area;
            simg[13] += w * nr;
            const float k = nr > 0.f ? w / nr : 0.f;            // d|r|/dr = r/|r| (0 at r = 0, as torch's norm backward)
            const float a0 = k * r0, a1 = k * r1, a2 = k * r2;  // dL/dX of this slot
            g0 += a0; g1 += a1; g2 += a2;
            sslot[j][0] -= a0 * P0; sslot[j][1] -= a0 * P1; sslot[j][2] -= a0 * P2;
            sslot[j][3] -= a1 * P0; sslot[j][4] -= a1 * P1; sslot[j][5] -= a1 * P2;
            sslot[j][6] -= a2 * P0; sslot[j][7] -= a2 * P1; sslot[j][8] -= a2 * P2;
            sslot[j][9] -= a0; sslot[j][10] -= a1; sslot[j][11] -= a2;
        }
        // camera-frame gradient gc = R^T g; dX_c/dlogdepth = X_c; dX_c/df = (-xc/f, -yc/f, 0)
        const float c0 = R[0] * g0 + R[3] * g1 + R[6] * g2;
        const float c1 = R[1] * g0 + R[4] * g1 + R[7] * g2;
        const float c2 = R[2] * g0 + R[5] * g1 + R[8] * g2;
        p.grad_logdepth[(long)img * HW + px] = c0 * xc + c1 * yc + c2 * zc;
        simg[0] += g0 * xc; simg[1] += g0 * yc; simg[2] += g0 * zc;
        simg[3] += g1 * xc; simg[4] += g1 * yc; simg[5] += g1 * zc;
        simg[6] += g2 * xc; simg[7] += g2 * yc; simg[8] += g2 * zc;
        simg[9] += g0; simg[10] += g1; simg[11] += g2;
        simg[12] -= (c0 * xc + c1 * yc) * inv_f;
    }
    block_sum<IMG_SUMS>(simg, red, tid);
    if (tid < IMG_SUMS) p.img_part[((long)img * gridDim.x + chunk) * IMG_SUMS + tid] = simg[0];
#pragma unroll
    for (int j = 0; j < MAXS; ++j) {
        if (j >= ns) break;                                      // uniform over the block
        block_sum<SLOT_SUMS>(sslot[j], red, tid);
        if (tid < SLOT_SUMS) p.slot_part[((long)(s0 + j) * gridDim.x + chunk) * SLOT_SUMS + tid] = sslot[j][0];
    }
}

// out[row][c] = sum over chunks (fixed order) of part[row][chunk][c]; one thread per (row, c)
__global__ __launch_bounds__(256) void align_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int rows, int nchunk,
                                                           int width) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * width) return;
    const int row = i / width, c = i - row * width;
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += part[((long)row * nchunk + k) * width + c];
    out[i] = s;
}

// Adam (torch.optim.Adam semantics: no weight decay, no amsgrad, bias correction, eps added after the sqrt of the corrected v)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ prm, const float* __restrict__ grad, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float bc1,
                                                   float bc2_sqrt) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = grad[i];
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    prm[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}

// the same with lr and the two bias corrections read from device memory (hyper = {lr, 1 - b1^t, sqrt(1 - b2^t)}): the whole
// iteration can then live in one captured hipGraph whose per-iteration scalars are produced on the device
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ prm, const float* __restrict__ grad, float* __restrict__ m,
                                                       float* __restrict__ v, long n, const float* __restrict__ hyper, float b1, float b2,
                                                       float eps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float lr = hyper[0], bc1 = hyper[1], bc2_sqrt = hyper[2];
    const float g = grad[i];
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    prm[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}

}  // namespace

extern "C" size_t geo4d_align_workspace(int n_imgs, int n_slots, int H, int W, int chunk_pixels) {
    if (n_imgs <= 0 || n_slots <= 0 || H <= 0 || W <= 0 || chunk_pixels <= 0) return 0;
    const size_t nchunk = ((size_t)H * W + chunk_pixels - 1) / chunk_pixels;
    return ((size_t)n_imgs * nchunk * IMG_SUMS + (size_t)n_slots * nchunk * SLOT_SUMS) * sizeof(float);
}

extern "C" int geo4d_align_residual(const geo4d_align_t* pp, void* stream) {
    if (!pp) return GEO4D_EINVAL;
    geo4d_align_t p = *pp;
    if (p.n_imgs <= 0 || p.n_slots <= 0 || p.H <= 0 || p.W <= 0 || p.chunk_pixels < 256 || (p.chunk_pixels % 256) || !p.pred || !p.conf ||
        !p.logdepth || !p.cams || !p.slot_trf || !p.slot_ptr || !p.slot_idx || !p.grad_logdepth || !p.img_sums || !p.slot_sums || !p.workspace) {
        geo4d_set_error("align_residual: bad arguments (chunk_pixels must be a positive multiple of 256)");
        return GEO4D_EINVAL;
    }
    if (p.max_slots_per_image > MAXS) { geo4d_set_error("align_residual: an image belongs to more than 6 windows"); return GEO4D_ENOTSUP; }
    const int HW = p.H * p.W;
    const int nchunk = (HW + p.chunk_pixels - 1) / p.chunk_pixels;
    if (p.workspace_bytes < geo4d_align_workspace(p.n_imgs, p.n_slots, p.H, p.W, p.chunk_pixels)) { geo4d_set_error("align_residual: workspace too small"); return GEO4D_EINVAL; }
    if (p.n_imgs > 65535) { geo4d_set_error("align_residual: too many images"); return GEO4D_EINVAL; }
    p.img_part = (float*)p.workspace;
    p.slot_part = p.img_part + (size_t)p.n_imgs * nchunk * IMG_SUMS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(align_residual_kernel, dim3(nchunk, p.n_imgs), dim3(256), 0, s, p);
    GEO4D_CHECK_LAUNCH();
    hipLaunchKernelGGL(align_reduce_kernel, dim3((p.n_imgs * IMG_SUMS + 255) / 256), dim3(256), 0, s, p.img_part, p.img_sums, p.n_imgs, nchunk, IMG_SUMS);
    GEO4D_CHECK_LAUNCH();
    hipLaunchKernelGGL(align_reduce_kernel, dim3((p.n_slots * SLOT_SUMS + 255) / 256), dim3(256), 0, s, p.slot_part, p.slot_sums, p.n_slots, nchunk, SLOT_SUMS);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                               float eps, int step, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) { geo4d_set_error("adam_step: bad arguments"); return GEO4D_EINVAL; }
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, bc1, sqrtf(bc2));
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, const float* hyper,
                                   float beta1, float beta2, float eps, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper || n <= 0) { geo4d_set_error("adam_step_dev: bad arguments"); return GEO4D_EINVAL; }
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       hyper, beta1, beta2, eps);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}
