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
#include <cstdlib>
#include "common.h"
#include "geo4d_hip.h"

namespace {

constexpr int MAXS = 6;        // windows an image can belong to (stride 4, length 16 -> 4, + the duplicated tail window = 5)
constexpr int IMG_SUMS = 14;   // dL/dR (9) | dL/dt (3) | dL/df (1) | loss (1)
constexpr int SLOT_SUMS = 14;  // dL/d(sR) (9) | dL/d(st) (3) | inverse-depth term: dL/ds_g, dL/dt_g

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

// DEPTH adds the inverse-depth term of optimizer_group.py:470-494 to the same pass:
//     + depth_weight * sum_{slot, pixel: q > 0.05, window accepted}  | 1 / (d + 1e-6) - (s_g q_s[p] + t_g) |
// (4 more bytes read per slot-pixel); slot_st[slot] = (s_g, t_g, accepted ? 1 : 0).
template <bool DEPTH>
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
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, gd = 0.f;
        const float invd = 1.0f / (d + 1e-6f);
#pragma unroll
        for (int j = 0; j < MAXS; ++j) {
            if (j >= ns) break;
            const int slot = p.slot_idx[s0 + j];
            if constexpr (DEPTH) {
                const float* st = p.slot_st + slot * 3;
                const float q = p.invdepth[(long)slot * HW + px];
                const float wd = (q > 0.05f ? p.depth_weight : 0.f) * st[2];
                const float rd = invd - (st[0] * q + st[1]);
                const float sg = rd > 0.f ? wd : (rd < 0.f ? -wd : 0.f);      // d|r|/dr = sign(r), 0 at 0 (torch.abs backward)
                simg[13] += wd * fabsf(rd);
                gd += sg;
                sslot[j][12] -= sg * q;
                sslot[j][13] -= sg;
            }
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
        // d(1 / (d + eps)) / dlogdepth = -d / (d + eps)^2
        p.grad_logdepth[(long)img * HW + px] = c0 * xc + c1 * yc + c2 * zc - gd * d * invd * invd;
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

// ---- round 3: the same arithmetic with the SLOT loop outside and the pixel loop inside ----------------------------------------------
// align_residual_kernel keeps 14 image sums + 6 x 14 slot sums live across its pixel loop (188 VGPRs, 2 waves per SIMD: 1.34 TB/s on
// the 128-frame clip, profiles/r02_align_bench.md). Every sum is LINEAR in the per-slot residual gradient a_j, so the slots can be
// walked one after the other: a thread owns NPX = chunk_pixels / 256 fixed pixels, keeps only their scalar dL/dlogdepth in registers
// (NPX values) + the 14 image sums + the CURRENT slot's 14 sums, re-derives the camera-frame point of a pixel per slot pass (one
// v_exp + ~12 flops; the log-depth re-read hits L1 / L2) and reduces the slot sums at the end of each pass. ~90 VGPRs instead of 188.
// Same partial-sum layout, same fixed-order reductions (deterministic); sums are accumulated slot-major instead of pixel-major, so the
// results differ from the first kernel by fp32 re-association only.
template <bool DEPTH, int NPX>
__global__ __launch_bounds__(256) void align_residual_kernel2(const geo4d_align_t p) {
    __shared__ float red[4 * SLOT_SUMS > 4 * IMG_SUMS ? 4 * SLOT_SUMS : 4 * IMG_SUMS];
    const int tid = threadIdx.x, chunk = blockIdx.x, img = blockIdx.y;
    const int HW = p.H * p.W;
    const float* cam = p.cams + img * 16;
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = cam[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = cam[9 + i];
    const float f = cam[12], inv_f = 1.0f / f, ppx = cam[13], ppy = cam[14];
    const int s0 = p.slot_ptr[img], ns = p.slot_ptr[img + 1] - s0;
    const int px0 = chunk * p.chunk_pixels + tid;
    float simg[IMG_SUMS], gl[NPX];
#pragma unroll
    for (int i = 0; i < IMG_SUMS; ++i) simg[i] = 0.f;
#pragma unroll
    for (int k = 0; k < NPX; ++k) gl[k] = 0.f;
    const float* ld = p.logdepth + (long)img * HW;
    for (int j = 0; j < ns; ++j) {                                   // uniform over the block
        const int slot = p.slot_idx[s0 + j];
        const float* T = p.slot_trf + slot * 12;
        float Tm[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) Tm[i] = T[i];
        float st0 = 0.f, st1 = 0.f, st2 = 0.f;
        if constexpr (DEPTH) { st0 = p.slot_st[slot * 3]; st1 = p.slot_st[slot * 3 + 1]; st2 = p.slot_st[slot * 3 + 2]; }
        const float* pr = p.pred + (long)slot * HW * 3;
        const float* cf = p.conf + (long)slot * HW;
        // launder the log-depth pointer once per slot pass: otherwise the loop-invariant code motion hoists d, the camera-frame point
        // and X of all NPX pixels out of the slot loop and keeps them live (8 registers per pixel: 225 VGPRs at NPX = 16)
        const float* ldj = ld;
        int px0j = px0;                              // ... and the pixel index, or every pixel's (u, v) and load addresses stay live too
        asm volatile("" : "+s"(ldj), "+v"(px0j));
        float ss[SLOT_SUMS];
#pragma unroll
        for (int i = 0; i < SLOT_SUMS; ++i) ss[i] = 0.f;
#pragma unroll
        for (int k = 0; k < NPX; ++k) {
            // at most 4 pixels' loads in flight per thread: without the fence hipcc hoists all NPX iterations' loads (240 VGPRs,
            // 2 waves per SIMD again); with it ~130 VGPRs and 4 waves per SIMD hide the latency instead
            if (k % 4 == 0 && k > 0) asm volatile("" ::: "memory");
            const int px = px0j + k * 256;
            if (px >= HW) continue;
            const float d = __expf(ldj[px]);
            const int v = px / p.W, u = px - v * p.W;
            const float xc = d * ((float)u - ppx) * inv_f, yc = d * ((float)v - ppy) * inv_f, zc = d;
            const float X0 = R[0] * xc + R[1] * yc + R[2] * zc + t[0];
            const float X1 = R[3] * xc + R[4] * yc + R[5] * zc + t[1];
            const float X2 = R[6] * xc + R[7] * yc + R[8] * zc + t[2];
            float gd = 0.f;
            if constexpr (DEPTH) {
                const float invd = 1.0f / (d + 1e-6f);
                const float q = p.invdepth[(long)slot * HW + px];
                const float wd = (q > 0.05f ? p.depth_weight : 0.f) * st2;
                const float rd = invd - (st0 * q + st1);
                const float sg = rd > 0.f ? wd : (rd < 0.f ? -wd : 0.f);
                simg[13] += wd * fabsf(rd);
                gd = sg * d * invd * invd;
                ss[12] -= sg * q;
                ss[13] -= sg;
            }
            const float3 Pv = *(const float3*)(pr + (long)px * 3);
            const float P0 = Pv.x, P1 = Pv.y, P2 = Pv.z;
            const float r0 = X0 - (Tm[0] * P0 + Tm[1] * P1 + Tm[2] * P2 + Tm[9]);
            const float r1 = X1 - (Tm[3] * P0 + Tm[4] * P1 + Tm[5] * P2 + Tm[10]);
            const float r2 = X2 - (Tm[6] * P0 + Tm[7] * P1 + Tm[8] * P2 + Tm[11]);
            const float nr = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
            const float w = fminf(cf[px], p.conf_clamp) * p.inv_area;
            simg[13] += w * nr;
            const float kk = nr > 0.f ? w / nr : 0.f;
            const float a0 = kk * r0, a1 = kk * r1, a2 = kk * r2;
            ss[0] -= a0 * P0; ss[1] -= a0 * P1; ss[2] -= a0 * P2;
            ss[3] -= a1 * P0; ss[4] -= a1 * P1; ss[5] -= a1 * P2;
            ss[6] -= a2 * P0; ss[7] -= a2 * P1; ss[8] -= a2 * P2;
            ss[9] -= a0; ss[10] -= a1; ss[11] -= a2;
            const float c0 = R[0] * a0 + R[3] * a1 + R[6] * a2;
            const float c1 = R[1] * a0 + R[4] * a1 + R[7] * a2;
            const float c2 = R[2] * a0 + R[5] * a1 + R[8] * a2;
            gl[k] += c0 * xc + c1 * yc + c2 * zc - gd;
            simg[0] += a0 * xc; simg[1] += a0 * yc; simg[2] += a0 * zc;
            simg[3] += a1 * xc; simg[4] += a1 * yc; simg[5] += a1 * zc;
            simg[6] += a2 * xc; simg[7] += a2 * yc; simg[8] += a2 * zc;
            simg[9] += a0; simg[10] += a1; simg[11] += a2;
            simg[12] -= (c0 * xc + c1 * yc) * inv_f;
        }
        block_sum<SLOT_SUMS>(ss, red, tid);
        if (tid < SLOT_SUMS) p.slot_part[((long)(s0 + j) * gridDim.x + chunk) * SLOT_SUMS + tid] = ss[0];
    }
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        const int px = px0 + k * 256;
        if (px < HW) p.grad_logdepth[(long)img * HW + px] = gl[k];
    }
    block_sum<IMG_SUMS>(simg, red, tid);
    if (tid < IMG_SUMS) p.img_part[((long)img * gridDim.x + chunk) * IMG_SUMS + tid] = simg[0];
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


// ---- start-up of the inverse-depth term: per-window least-absolute-deviation fit (optimizer_group.py:333-372) ------------------------
// The reference fits, per window, (s, t) minimising sum | s q + t - g | with 5000 Adam iterations of torch ops over the window's
// S*H*W pixels (dust3r/depth_eval.py:112-145), starting from s = median(g) / median(q) (:218-221), then scores the fit by the share
// of pixels with max(a/g, g/a) < 1.25 (:297-301). Here: target g materialised once, both medians by a 4-pass radix select
// (bit-exact lower median = torch.median), ONE launch per Adam iteration for ALL windows (every workgroup re-derives the window's
// Adam step from the previous launch's partial sums, so the update needs no launch of its own), one pass for the score.
constexpr int LAD_STATE = 12;   // s, t, m_s, m_t, v_s, v_t, prev_loss, has_prev, done, steps, unused x2

__global__ __launch_bounds__(256) void lad_target_kernel(const float* __restrict__ logdepth, const int* __restrict__ slot_img,
                                                         float* __restrict__ target, int HW) {
    const int slot = blockIdx.y;
    const long img = slot_img[slot];
    for (int px = blockIdx.x * 256 + threadIdx.x; px < HW; px += gridDim.x * 256)
        target[(long)slot * HW + px] = 1.0f / (expf(logdepth[img * HW + px]) + 1e-6f);
}

__device__ __forceinline__ unsigned ordered_key(float x) {          // monotone float -> unsigned
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// sel[row] = {prefix, mask, rank, unused}; hist[row][256]
__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ x, long n, const unsigned* __restrict__ sel, int shift,
                                                          unsigned* __restrict__ hist) {
    __shared__ unsigned h[256];
    const int row = blockIdx.y;
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned prefix = sel[row * 4 + 0], mask = sel[row * 4 + 1];
    const float* xr = x + (long)row * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned k = ordered_key(xr[i]);
        if ((k & mask) == prefix) atomicAdd(&h[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[row * 256 + threadIdx.x], h[threadIdx.x]);      // integer adds: order-independent
}

__global__ void select_pick_kernel(unsigned* __restrict__ sel, unsigned* __restrict__ hist, int shift, int rows, float* __restrict__ out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    unsigned rank = sel[row * 4 + 2], cum = 0;
    int bin = 255;
    for (int b = 0; b < 256; ++b) {
        const unsigned c = hist[row * 256 + b];
        if (rank < cum + c) { bin = b; break; }
        cum += c;
    }
    for (int b = 0; b < 256; ++b) hist[row * 256 + b] = 0;
    sel[row * 4 + 0] |= (unsigned)bin << shift;
    sel[row * 4 + 1] |= 255u << shift;
    sel[row * 4 + 2] = rank - cum;
    if (shift == 0) out[row] = key_to_float(sel[row * 4 + 0]);
}

__global__ void select_init_kernel(unsigned* __restrict__ sel, unsigned* __restrict__ hist, int rows, unsigned rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) { sel[i * 4 + 0] = 0; sel[i * 4 + 1] = 0; sel[i * 4 + 2] = rank; sel[i * 4 + 3] = 0; }
    if (i < rows * 256) hist[i] = 0;
}

// state layout per window: LAD_STATE floats. init: s = median(g) / median(q), everything else 0; inactive windows: done = 1
__global__ void lad_init_kernel(float* __restrict__ state, const float* __restrict__ med_g, const float* __restrict__ med_q,
                                const unsigned char* __restrict__ active, int G) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    float* st = state + g * LAD_STATE;
    for (int i = 0; i < LAD_STATE; ++i) st[i] = 0.f;
    st[0] = med_g[g] / med_q[g];
    st[8] = (active && !active[g]) ? 1.f : 0.f;
}

// One Adam iteration for every window. Launch `it` (0-based): (1) it > 0: add the partial sums launch it-1 left (loss and gradient at
// the parameters of step it-1) in a fixed order and apply Adam step `it` exactly as torch.optim.Adam does (+ the reference's stop
// test: |previous loss - loss| < tol ends the fit AFTER this step); (2) unless finished, accumulate this chunk's partial sums at
// the new parameters. Every workgroup of a window derives the same new state; chunk 0 stores it.
__global__ __launch_bounds__(256) void lad_iter_kernel(const float* __restrict__ q, const float* __restrict__ target, long n, int nchunk,
                                                       long chunk_elems, const float* __restrict__ state_in, float* __restrict__ state_out,
                                                       const float* __restrict__ part_in, float* __restrict__ part_out, int it, int last,
                                                       float step_size, float bc2_sqrt, float b1, float b2, float eps, float tol) {
    __shared__ float red[4 * 3];
    const int tid = threadIdx.x, chunk = blockIdx.x, g = blockIdx.y;
    float st[LAD_STATE];
#pragma unroll
    for (int i = 0; i < LAD_STATE; ++i) st[i] = state_in[g * LAD_STATE + i];
    if (it > 0 && st[8] == 0.f) {
        float v[3] = {0.f, 0.f, 0.f};
        for (int c = tid; c < nchunk; c += 256) {
            const float* pp = part_in + ((long)g * nchunk + c) * 3;
            v[0] += pp[0]; v[1] += pp[1]; v[2] += pp[2];
        }
        block_sum<3>(v, red, tid);
        __shared__ float tot[3];
        if (tid < 3) tot[tid] = v[0];
        __syncthreads();
        const float loss = tot[0], gs = tot[1], gt = tot[2];
        // exp_avg.lerp_(grad, 1 - b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad, value = 1 - b2); param.addcdiv_(exp_avg, denom, -step_size)
        st[2] = st[2] + (1.f - b1) * (gs - st[2]);
        st[3] = st[3] + (1.f - b1) * (gt - st[3]);
        st[4] = b2 * st[4] + (1.f - b2) * gs * gs;
        st[5] = b2 * st[5] + (1.f - b2) * gt * gt;
        st[0] -= step_size * (st[2] / (sqrtf(st[4]) / bc2_sqrt + eps));
        st[1] -= step_size * (st[3] / (sqrtf(st[5]) / bc2_sqrt + eps));
        if (st[7] != 0.f && fabsf(st[6] - loss) < tol) st[8] = 1.f;
        st[6] = loss;
        st[7] = 1.f;
        st[9] += 1.f;
    }
    if (chunk == 0 && tid < LAD_STATE) state_out[g * LAD_STATE + tid] = st[tid];
    if (last || st[8] != 0.f) return;
    const float s = st[0], t = st[1];
    const long e0 = (long)chunk * chunk_elems, e1 = min(n, e0 + chunk_elems);
    const float* qr = q + (long)g * n;
    const float* tr = target + (long)g * n;
    float a[3] = {0.f, 0.f, 0.f};
    auto term = [&](float p, float tg) {
        const float r = s * p + t - tg;
        const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);     // d|r|/dr as torch.abs' backward: 0 at 0
        a[0] += fabsf(r);
        a[1] += sg * p;
        a[2] += sg;
    };
    if ((n & 3) == 0) {                                               // 16-byte loads: rows start 16-byte aligned when n % 4 == 0
        for (long e = e0 + 4 * tid; e < e1; e += 1024) {
            const f32x4 pv = *(const f32x4*)(qr + e), tv = *(const f32x4*)(tr + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) term(pv[j], tv[j]);
        }
    } else {
        for (long e = e0 + tid; e < e1; e += 256) term(qr[e], tr[e]);
    }
    block_sum<3>(a, red, tid);
    if (tid < 3) part_out[((long)g * nchunk + chunk) * 3 + tid] = a[0];
}

// counts[g] = {pixels of the metric mask (conf > conf_thr and q > q_thr) with max(a/g, g/a) < 1.25, pixels of the mask}, a = max(s q + t, 1e-5)
__global__ __launch_bounds__(256) void lad_delta_kernel(const float* __restrict__ q, const float* __restrict__ target, const float* __restrict__ conf,
                                                        const float* __restrict__ st2, long n, float conf_thr, float conf_clamp, float q_thr,
                                                        unsigned* __restrict__ counts) {
    const int g = blockIdx.y;
    const float s = st2[g * 2], t = st2[g * 2 + 1];
    unsigned ok = 0, all = 0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float p = q[(long)g * n + e];
        if (fminf(conf[(long)g * n + e], conf_clamp) > conf_thr && p > q_thr) {
            const float a = fmaxf(s * p + t, 1e-5f), gt = target[(long)g * n + e];
            ++all;
            if (fmaxf(a / gt, gt / a) < 1.25f) ++ok;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { ok += __shfl_xor(ok, o); all += __shfl_xor(all, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[g * 2], ok); atomicAdd(&counts[g * 2 + 1], all); }
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
    if (p.invdepth && !p.slot_st) { geo4d_set_error("align_residual: invdepth needs slot_st"); return GEO4D_EINVAL; }
    if (p.max_slots_per_image > MAXS) { geo4d_set_error("align_residual: an image belongs to more than 6 windows"); return GEO4D_ENOTSUP; }
    const int HW = p.H * p.W;
    const int nchunk = (HW + p.chunk_pixels - 1) / p.chunk_pixels;
    if (p.workspace_bytes < geo4d_align_workspace(p.n_imgs, p.n_slots, p.H, p.W, p.chunk_pixels)) { geo4d_set_error("align_residual: workspace too small"); return GEO4D_EINVAL; }
    if (p.n_imgs > 65535) { geo4d_set_error("align_residual: too many images"); return GEO4D_EINVAL; }
    p.img_part = (float*)p.workspace;
    p.slot_part = p.img_part + (size_t)p.n_imgs * nchunk * IMG_SUMS;
    hipStream_t s = (hipStream_t)stream;
    // slot-major kernel (round 3) for the chunk sizes it is instantiated for; the pixel-major kernel serves any other multiple of 256
#define GEO4D_ALIGN2(NPX_)                                                                                                        \
    do {                                                                                                                           \
        if (p.invdepth) hipLaunchKernelGGL((align_residual_kernel2<true, NPX_>), dim3(nchunk, p.n_imgs), dim3(256), 0, s, p);       \
        else hipLaunchKernelGGL((align_residual_kernel2<false, NPX_>), dim3(nchunk, p.n_imgs), dim3(256), 0, s, p);                  \
    } while (0)
    const bool v1 = getenv("GEO4D_ALIGN_KERNEL") && atoi(getenv("GEO4D_ALIGN_KERNEL")) == 1;
    if (!v1 && p.chunk_pixels == 4096) GEO4D_ALIGN2(16);
    else if (!v1 && p.chunk_pixels == 2048) GEO4D_ALIGN2(8);
    else if (!v1 && p.chunk_pixels == 1024) GEO4D_ALIGN2(4);
    else if (!v1 && p.chunk_pixels == 256) GEO4D_ALIGN2(1);
    else if (p.invdepth) hipLaunchKernelGGL(align_residual_kernel<true>, dim3(nchunk, p.n_imgs), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(align_residual_kernel<false>, dim3(nchunk, p.n_imgs), dim3(256), 0, s, p);
#undef GEO4D_ALIGN2
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

// ---- LAD start-up: C entry points ----------------------------------------------------------------------------------------------------
static constexpr long LAD_CHUNK = 8192;
extern "C" size_t geo4d_lad_workspace(int G, long n) {
    if (G <= 0 || n <= 0) return 0;
    const size_t nchunk = (size_t)((n + LAD_CHUNK - 1) / LAD_CHUNK);
    // 2 x state | 2 x partial sums | select state + histograms for 2 x G rows | medians 2 x G | counts
    return (2 * (size_t)G * LAD_STATE + 2 * (size_t)G * nchunk * 3 + 2 * (size_t)G * 4 + 2 * (size_t)G * 256 + 2 * (size_t)G + 2 * (size_t)G) * 4 + 64;
}

extern "C" int geo4d_lad_target(const float* logdepth, const int* slot_img, float* target, int n_slots, int HW, void* stream) {
    if (!logdepth || !slot_img || !target || n_slots <= 0 || HW <= 0 || n_slots > 65535) { geo4d_set_error("lad_target: bad arguments"); return GEO4D_EINVAL; }
    hipLaunchKernelGGL(lad_target_kernel, dim3(min((HW + 255) / 256, 1024), n_slots), dim3(256), 0, (hipStream_t)stream, logdepth, slot_img, target, HW);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

// lower median of every row of x [rows][n] (== torch.median), bit-exact
static int lower_median_rows(const float* x, int rows, long n, unsigned* sel, unsigned* hist, float* out, hipStream_t s) {
    hipLaunchKernelGGL(select_init_kernel, dim3((rows * 256 + 255) / 256), dim3(256), 0, s, sel, hist, rows, (unsigned)((n - 1) / 2));
    GEO4D_CHECK_LAUNCH();
    const int nb = (int)min((n + 255) / 256, (long)2048);
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(select_hist_kernel, dim3(nb, rows), dim3(256), 0, s, x, n, sel, shift, hist);
        GEO4D_CHECK_LAUNCH();
        hipLaunchKernelGGL(select_pick_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, sel, hist, shift, rows, out);
        GEO4D_CHECK_LAUNCH();
    }
    return GEO4D_OK;
}

extern "C" int geo4d_lower_median(const float* x, int rows, long n, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || rows <= 0 || n <= 0 || rows > 65535 || n >= (1L << 32) || !workspace || workspace_bytes < (size_t)rows * (4 + 256) * 4) {
        geo4d_set_error("lower_median: bad arguments (workspace: rows * 260 * 4 bytes)");
        return GEO4D_EINVAL;
    }
    unsigned* sel = (unsigned*)workspace;
    return lower_median_rows(x, rows, n, sel, sel + (size_t)rows * 4, out, (hipStream_t)stream);
}

extern "C" int geo4d_lad_fit(const float* q, const float* target, int G, long n, const unsigned char* active, float lr, int max_iters, float tol,
                             float* st_out, float* info_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!q || !target || !st_out || G <= 0 || n <= 0 || G > 65535 || n >= (1L << 32) || max_iters <= 0 || !workspace ||
        workspace_bytes < geo4d_lad_workspace(G, n)) {
        geo4d_set_error("lad_fit: bad arguments / workspace too small");
        return GEO4D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = (int)((n + LAD_CHUNK - 1) / LAD_CHUNK);
    float* state[2] = {(float*)workspace, (float*)workspace + (size_t)G * LAD_STATE};
    float* part[2] = {state[1] + (size_t)G * LAD_STATE, state[1] + (size_t)G * LAD_STATE + (size_t)G * nchunk * 3};
    unsigned* sel = (unsigned*)(part[1] + (size_t)G * nchunk * 3);
    unsigned* hist = sel + 2 * (size_t)G * 4;
    float* med = (float*)(hist + 2 * (size_t)G * 256);           // [2][G]: target, then q
    int rc = lower_median_rows(target, G, n, sel, hist, med, s);
    if (rc) return rc;
    rc = lower_median_rows(q, G, n, sel, hist, med + G, s);
    if (rc) return rc;
    hipLaunchKernelGGL(lad_init_kernel, dim3((G + 63) / 64), dim3(64), 0, s, state[0], med, med + G, active, G);
    GEO4D_CHECK_LAUNCH();
    const double b1 = 0.9, b2 = 0.999;                           // torch.optim.Adam defaults (absolute_value_scaling2 passes only lr)
    for (int it = 0; it <= max_iters; ++it) {
        const double bc1 = 1.0 - pow(b1, (double)it), bc2 = 1.0 - pow(b2, (double)it);
        const float step_size = it ? (float)((double)lr / bc1) : 0.f, bc2_sqrt = it ? (float)sqrt(bc2) : 1.f;
        hipLaunchKernelGGL(lad_iter_kernel, dim3(nchunk, G), dim3(256), 0, s, q, target, n, nchunk, LAD_CHUNK, state[it & 1], state[(it + 1) & 1],
                           part[(it + 1) & 1], part[it & 1], it, it == max_iters ? 1 : 0, step_size, bc2_sqrt, (float)b1, (float)b2, 1e-8f, tol);
        GEO4D_CHECK_LAUNCH();
    }
    const float* fin = state[(max_iters + 1) & 1];
    // st_out [G][2] = (s, t); info_out [G][2] = (Adam steps taken, last loss) when asked for
    if (hipMemcpy2DAsync(st_out, 2 * sizeof(float), fin, LAD_STATE * sizeof(float), 2 * sizeof(float), G, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        geo4d_set_error("lad_fit: result copy failed");
        return GEO4D_EIO;
    }
    if (info_out) {
        if (hipMemcpy2DAsync(info_out, 2 * sizeof(float), fin + 9, LAD_STATE * sizeof(float), sizeof(float), G, hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpy2DAsync(info_out + 1, 2 * sizeof(float), fin + 6, LAD_STATE * sizeof(float), sizeof(float), G, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            geo4d_set_error("lad_fit: info copy failed");
            return GEO4D_EIO;
        }
    }
    return GEO4D_OK;
}

extern "C" int geo4d_lad_delta(const float* q, const float* target, const float* conf, const float* st, int G, long n, float conf_thr,
                               float conf_clamp, float q_thr, unsigned* counts, void* stream) {
    if (!q || !target || !conf || !st || !counts || G <= 0 || n <= 0 || G > 65535) { geo4d_set_error("lad_delta: bad arguments"); return GEO4D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, (size_t)G * 2 * sizeof(unsigned), s) != hipSuccess) { geo4d_set_error("lad_delta: memset failed"); return GEO4D_EIO; }
    hipLaunchKernelGGL(lad_delta_kernel, dim3((int)min((n + 255) / 256, (long)1024), G), dim3(256), 0, s, q, target, conf, st, n, conf_thr, conf_clamp, q_thr, counts);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}
