// geo4d_amd/csrc/align_small.hip — the "small parameter" half of one global-alignment iteration, round 3.
//
// The fused residual kernel (align.hip) leaves, per image and per (window, frame) slot, the gradient SUMS of the loss with respect to
// the matrices it consumed: dL/dR_i (3x3), dL/dt_i, dL/df_i, and dL/d(sR_g), dL/d(st_g) (+ dL/ds_g, dL/dt_g of the inverse-depth fit).
// What remains of `loss.backward()` of LightPointCloudGroupOptimizer.forward (optimizer_group.py:440-525) is the chain rule through
// the parameterisation (base_opt_group.py:262-327: XYZW quaternion -> rotation via roma.RigidUnitQuat(...).normalize(), signed-expm1
// translations, focal = exp(l / 20), window scale = exp(l_g) exp(log base_scale - mean l)) and the camera temporal-smoothing term
// (relative_pose_loss, :529-541) — a few hundred flops per image. Round 2 left it to autograd over [n, 7] tensors: ~450 kernel
// launches per iteration, 0.9 ms of a 1.46 ms replayed iteration (profiles/r03_align_iteration.md). Here it is TWO launches:
//   align_refresh_kernel  parameters -> the residual kernel's inputs: cams [n][16] = (R | t | f | ppx | ppy | 0), slot_trf [slots][12]
//   align_small_kernel    gradient sums + parameters -> loss and the gradients of im_poses [n][7], im_focals, pw_poses [G][8]
//                         (+ s_depth, t_depth), including the smoothing term and its gradient
// One workgroup each (n, G are a few hundred at most); every sum is taken in a fixed order (deterministic).
#include "common.h"
#include "geo4d_hip.h"

namespace {

// fp64 inside: a few hundred flops per image, and several of the sums cancel (the scale gradients against their mean, the quaternion
// gradient against its radial part) - the inputs (fp32 gradient sums) are the only fp32 noise left
typedef double real;

__device__ __forceinline__ void quat_rot(const float* q, real* R, real* u, real& nrm) {
    const real q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    nrm = sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    const real x = q0 / nrm, y = q1 / nrm, z = q2 / nrm, w = q3 / nrm;
    u[0] = x; u[1] = y; u[2] = z; u[3] = w;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
// dL/dq from G = dL/dR (row-major 3x3), through R(u) and u = q / |q|
__device__ __forceinline__ void quat_rot_backward(const real* G, const real* u, real nrm, real* gq) {
    const real x = u[0], y = u[1], z = u[2], w = u[3];
    real g[4];
    g[0] = 2 * (G[1] * y + G[2] * z + G[3] * y - 2 * G[4] * x - G[5] * w + G[6] * z + G[7] * w - 2 * G[8] * x);
    g[1] = 2 * (-2 * G[0] * y + G[1] * x + G[2] * w + G[3] * x + G[5] * z - G[6] * w + G[7] * z - 2 * G[8] * y);
    g[2] = 2 * (-2 * G[0] * z - G[1] * w + G[2] * x + G[3] * w - 2 * G[4] * z + G[5] * y + G[6] * x + G[7] * y);
    g[3] = 2 * (-G[1] * z + G[2] * y + G[3] * z - G[5] * x - G[6] * y + G[7] * x);
    const real dot = g[0] * x + g[1] * y + g[2] * z + g[3] * w;
#pragma unroll
    for (int k = 0; k < 4; ++k) gq[k] = (g[k] - u[k] * dot) / nrm;
}
__device__ __forceinline__ real signed_expm1_f(real v) { return v > 0 ? expm1(v) : (v < 0 ? -expm1(-v) : 0.0); }
// d/dv sign(v) expm1(|v|) as torch differentiates it: exp(|v|), and 0 at exactly v = 0 (sign' = 0, |.|' = 0 there)
__device__ __forceinline__ real signed_expm1_grad(real v) { return v != 0 ? exp(fabs(v)) : 0.0; }

__device__ __forceinline__ real pw_log_shift(const float* pw, int G, float base_scale, int norm) {
    if (!norm) return 0.0;
    real m = 0;
    for (int g = 0; g < G; ++g) m += pw[g * 8 + 7];      // fixed order
    return log((real)base_scale) - m / (real)G;
}

__global__ __launch_bounds__(256) void align_refresh_kernel(const geo4d_align_small_t p) {
    const int tid = threadIdx.x;
    for (int i = tid; i < p.n_imgs; i += 256) {
        real R[9], u[4], nrm;
        quat_rot(p.im_poses + i * 7, R, u, nrm);
        float* c = p.cams + i * 16;
#pragma unroll
        for (int k = 0; k < 9; ++k) c[k] = (float)R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) c[9 + k] = (float)signed_expm1_f(p.im_poses[i * 7 + 4 + k]);
        c[12] = (float)exp((real)p.im_focals[p.n_focals == 1 ? 0 : i] / p.focal_break);
        c[13] = p.ppx; c[14] = p.ppy; c[15] = 0.f;
    }
    const real shift = pw_log_shift(p.pw_poses, p.n_groups, p.base_scale, p.norm_pw_scale);
    for (int s = tid; s < p.n_slots; s += 256) {
        const int g = s / p.slots_per_group;
        const float* pw = p.pw_poses + g * 8;
        real R[9], u[4], nrm;
        quat_rot(pw, R, u, nrm);
        const real sc = exp((real)pw[7] + shift);
        float* o = p.slot_trf + s * 12;
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = (float)(R[k] * sc);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[9 + k] = (float)(signed_expm1_f(pw[4 + k]) * sc);
        if (p.slot_st) {                                    // inverse-depth term: (s_g, t_g, accepted_g) per slot
            p.slot_st[s * 3 + 0] = p.s_depth[g];
            p.slot_st[s * 3 + 1] = p.t_depth[g];
            p.slot_st[s * 3 + 2] = p.depth_ok[g];
        }
    }
}

// trajectory term (optimizer_group.py:496-512): for a valid window g and its frame k (image i): M = T_g [R_k | e^l_g t_k] with
// T_g = (Ra | ta) from traj_align_poses[g]; term = 0.005 (|Rm^T R_i - I|_F + tw |Rm^T (t_i - tm)|). Fills Rm, tm, the two normalised
// directions and returns the term; the caller turns them into the gradient it owns.
struct TrajPair { real Rm[9], tm[3], A[9], bv[3], d[3], ir, ib, term; };
__device__ __forceinline__ void traj_pair(const real* Ra, const real* ta, real sc, const float* Mk /* 4x4 row-major */, const real* Ri, const real* ti,
                                          real tw, TrajPair& o) {
    real Rk[9], tk[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Rk[r * 3 + c] = Mk[r * 4 + c];
        tk[r] = Mk[r * 4 + 3];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o.Rm[r * 3 + c] = Ra[r * 3 + 0] * Rk[0 * 3 + c] + Ra[r * 3 + 1] * Rk[1 * 3 + c] + Ra[r * 3 + 2] * Rk[2 * 3 + c];
        o.tm[r] = (Ra[r * 3 + 0] * tk[0] + Ra[r * 3 + 1] * tk[1] + Ra[r * 3 + 2] * tk[2]) * sc + ta[r];
    }
    real rho = 0, beta = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const real v = o.Rm[0 * 3 + r] * Ri[0 * 3 + c] + o.Rm[1 * 3 + r] * Ri[1 * 3 + c] + o.Rm[2 * 3 + r] * Ri[2 * 3 + c] - (r == c ? 1.0 : 0.0);
            o.A[r * 3 + c] = v;
            rho += v * v;
        }
    rho = sqrt(rho);
#pragma unroll
    for (int r = 0; r < 3; ++r) o.d[r] = ti[r] - o.tm[r];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o.bv[r] = o.Rm[0 * 3 + r] * o.d[0] + o.Rm[1 * 3 + r] * o.d[1] + o.Rm[2 * 3 + r] * o.d[2];
        beta += o.bv[r] * o.bv[r];
    }
    beta = sqrt(beta);
    o.ir = rho > 0 ? 1.0 / rho : 0.0;
    o.ib = beta > 0 ? tw / beta : 0.0;
    o.term = rho + tw * beta;
}

__global__ __launch_bounds__(256) void align_small_kernel(const geo4d_align_small_t p) {
    __shared__ real red[256];
    __shared__ real sh_shift, sh_mean_ds;
    const int tid = threadIdx.x;
    const int n = p.n_imgs, G = p.n_groups;
    // ---- per-window sums of the slot gradients (CSR order of the residual kernel -> fixed order per window) ------------------------
    for (int item = tid; item < G * 14; item += 256) {          // one (window, component) per thread, entries in CSR order
        const int g = item / 14, k = item - g * 14;
        real acc = 0;
        for (int j = p.group_ptr[g]; j < p.group_ptr[g + 1]; ++j) acc += p.slot_sums[(long)p.group_entries[j] * 14 + k];   // ascending CSR entry
        p.group_sums[item] = acc;
    }
    __syncthreads();
    if (tid == 0) sh_shift = pw_log_shift(p.pw_poses, G, p.base_scale, p.norm_pw_scale);
    __syncthreads();
    // ---- window parameters: dL/dq_g, dL/dlt_g, D_g s_g (the scale gradient before the mean coupling) ------------------------------
    for (int g = tid; g < G; g += 256) {
        const float* pw = p.pw_poses + g * 8;
        const real* S = p.group_sums + g * 14;
        real R[9], u[4], nrm;
        quat_rot(pw, R, u, nrm);
        const real sc = exp((real)pw[7] + sh_shift);
        real GR[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) GR[k] = S[k] * sc;
        real gq[4];
        quat_rot_backward(GR, u, nrm, gq);
        float* o = p.grad_pw_poses + g * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (float)gq[k];
        real D = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) D += S[k] * R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const real lt = pw[4 + k];
            o[4 + k] = (float)(S[9 + k] * sc * signed_expm1_grad(lt));
            D += S[9 + k] * signed_expm1_f(lt);
        }
        p.scale_terms[g] = D * sc;                                         // completed below: - mean_g (D_g s_g) when the scales are normalised
        if (p.grad_s_depth) { p.grad_s_depth[g] = (float)S[12]; p.grad_t_depth[g] = (float)S[13]; }
    }
    __syncthreads();
    if (tid == 0) {
        real m = 0;
        if (p.norm_pw_scale) {
            for (int g = 0; g < G; ++g) m += p.scale_terms[g];
            m /= (real)G;
        }
        sh_mean_ds = m;
    }
    __syncthreads();
    for (int g = tid; g < G; g += 256) p.grad_pw_poses[g * 8 + 7] = (float)(p.scale_terms[g] - sh_mean_ds);
    // ---- trajectory term, window side: gradient of traj_align_poses[g] (quaternion, signed-log translation, log scale) ---------------
    real loss_part = 0, focal_part = 0;
    const bool traj_on = p.traj != nullptr;
    if (traj_on) {
        for (int g = tid; g < G; g += 256) {
            float* o = p.grad_traj + g * 8;
            if (!p.traj_valid[g]) {
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = 0.f;
                continue;
            }
            const float* tp = p.traj_align + g * 8;
            real Ra[9], ua[4], na, ta[3];
            quat_rot(tp, Ra, ua, na);
#pragma unroll
            for (int k = 0; k < 3; ++k) ta[k] = signed_expm1_f(tp[4 + k]);
            const real sc = exp((real)tp[7]);
            real GRa[9], Gta[3], Gsc = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) GRa[k] = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) Gta[k] = 0;
            for (int k = 0; k < p.slots_per_group; ++k) {
                const int i = p.slot_img[g * p.slots_per_group + k];
                real Ri[9], ui[4], ni, ti[3];
                quat_rot(p.im_poses + i * 7, Ri, ui, ni);
#pragma unroll
                for (int c = 0; c < 3; ++c) ti[c] = signed_expm1_f(p.im_poses[i * 7 + 4 + c]);
                const float* Mk = p.traj + ((long)g * p.slots_per_group + k) * 16;
                TrajPair tr;
                traj_pair(Ra, ta, sc, Mk, Ri, ti, p.translation_weight, tr);
                loss_part += p.traj_weight * tr.term;
                // dterm/dRm = Ri (A/|A|)^T + tw d (b/|b|)^T ; dterm/dtm = -tw Rm (b/|b|)
                real GRm[9], Gtm[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        GRm[r * 3 + c] = (Ri[r * 3 + 0] * tr.A[c * 3 + 0] + Ri[r * 3 + 1] * tr.A[c * 3 + 1] + Ri[r * 3 + 2] * tr.A[c * 3 + 2]) * tr.ir + tr.d[r] * tr.bv[c] * tr.ib;
                    Gtm[r] = -(tr.Rm[r * 3 + 0] * tr.bv[0] + tr.Rm[r * 3 + 1] * tr.bv[1] + tr.Rm[r * 3 + 2] * tr.bv[2]) * tr.ib;
                }
                // Rm = Ra Rk, tm = sc Ra tk + ta
                real tk[3] = {Mk[3], Mk[7], Mk[11]};
                real Ratk[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) Ratk[r] = Ra[r * 3 + 0] * tk[0] + Ra[r * 3 + 1] * tk[1] + Ra[r * 3 + 2] * tk[2];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        GRa[r * 3 + c] += GRm[r * 3 + 0] * Mk[c * 4 + 0] + GRm[r * 3 + 1] * Mk[c * 4 + 1] + GRm[r * 3 + 2] * Mk[c * 4 + 2] + Gtm[r] * sc * tk[c];
                    Gta[r] += Gtm[r];
                    Gsc += Gtm[r] * Ratk[r];
                }
            }
            real gq[4];
            quat_rot_backward(GRa, ua, na, gq);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (float)(p.traj_weight * gq[k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) o[4 + k] = (float)(p.traj_weight * Gta[k] * signed_expm1_grad(tp[4 + k]));
            o[7] = (float)(p.traj_weight * Gsc * sc);
        }
    }
    // ---- image parameters: data term + temporal smoothing (relative_pose_loss between consecutive cameras) -------------------------
    const bool smooth = p.smooth_weight > 0.f && n > 1;
    for (int i = tid; i < n; i += 256) {
        const float* I = p.img_sums + i * 14;
        const float* q = p.im_poses + i * 7;
        real R[9], u[4], nrm;
        quat_rot(q, R, u, nrm);
        real t[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = signed_expm1_f(q[4 + k]);
        real GR[9], Gt[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) GR[k] = I[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) Gt[k] = I[9 + k];
        loss_part += I[13];
        if (traj_on) {
            // camera side of the trajectory term: every (valid window, frame) this image belongs to (full slot lists, CSR by image)
            for (int j = p.img_slot_ptr[i]; j < p.img_slot_ptr[i + 1]; ++j) {
                const int slot = p.img_slot_idx[j], g = slot / p.slots_per_group;
                if (!p.traj_valid[g]) continue;
                const float* tp = p.traj_align + g * 8;
                real Ra[9], ua[4], na, ta[3];
                quat_rot(tp, Ra, ua, na);
#pragma unroll
                for (int k = 0; k < 3; ++k) ta[k] = signed_expm1_f(tp[4 + k]);
                TrajPair tr;
                traj_pair(Ra, ta, exp((real)tp[7]), p.traj + (long)slot * 16, R, t, p.translation_weight, tr);
                // dterm/dRi = Rm (A/|A|) ; dterm/dti = tw Rm (b/|b|)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        GR[r * 3 + c] += p.traj_weight * (tr.Rm[r * 3 + 0] * tr.A[0 * 3 + c] + tr.Rm[r * 3 + 1] * tr.A[1 * 3 + c] + tr.Rm[r * 3 + 2] * tr.A[2 * 3 + c]) * tr.ir;
                    Gt[r] += p.traj_weight * (tr.Rm[r * 3 + 0] * tr.bv[0] + tr.Rm[r * 3 + 1] * tr.bv[1] + tr.Rm[r * 3 + 2] * tr.bv[2]) * tr.ib;
                }
            }
        }
        if (smooth) {
            // pair (a, b) = (i, i + 1): A = Ra^T Rb - I, b_ = Ra^T (tb - ta); term = |A|_F + tw |b_|
            //   dterm/dRa = Rb (A/|A|)^T + tw (tb - ta) (b_/|b_|)^T ; dterm/dRb = Ra (A/|A|) ; dterm/dtb = tw Ra (b_/|b_|) = -dterm/dta
            for (int side = 0; side < 2; ++side) {
                const int a = side == 0 ? i : i - 1, b = a + 1;
                if (a < 0 || b >= n) continue;
                real Ra[9], Rb[9], ta[3], tb[3], uu[4], nn;
                if (side == 0) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) Ra[k] = R[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) ta[k] = t[k];
                    quat_rot(p.im_poses + b * 7, Rb, uu, nn);
#pragma unroll
                    for (int k = 0; k < 3; ++k) tb[k] = signed_expm1_f(p.im_poses[b * 7 + 4 + k]);
                } else {
#pragma unroll
                    for (int k = 0; k < 9; ++k) Rb[k] = R[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) tb[k] = t[k];
                    quat_rot(p.im_poses + a * 7, Ra, uu, nn);
#pragma unroll
                    for (int k = 0; k < 3; ++k) ta[k] = signed_expm1_f(p.im_poses[a * 7 + 4 + k]);
                }
                real A[9], rho = 0;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        real v = Ra[0 * 3 + r] * Rb[0 * 3 + c] + Ra[1 * 3 + r] * Rb[1 * 3 + c] + Ra[2 * 3 + r] * Rb[2 * 3 + c] - (r == c ? 1.0 : 0.0);
                        A[r * 3 + c] = v;
                        rho += v * v;
                    }
                rho = sqrt(rho);
                const real d[3] = {tb[0] - ta[0], tb[1] - ta[1], tb[2] - ta[2]};
                real bv[3], beta = 0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    bv[r] = Ra[0 * 3 + r] * d[0] + Ra[1 * 3 + r] * d[1] + Ra[2 * 3 + r] * d[2];
                    beta += bv[r] * bv[r];
                }
                beta = sqrt(beta);
                const real ir = rho > 0 ? 1.0 / rho : 0.0, ib = beta > 0 ? (real)p.translation_weight / beta : 0.0;
                if (side == 0) loss_part += p.smooth_weight * (rho + p.translation_weight * beta);    // each pair counted once (by its first image)
                const real wgt = p.smooth_weight;
                if (side == 0) {              // this image is `a`
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            GR[r * 3 + c] += wgt * ((Rb[r * 3 + 0] * A[c * 3 + 0] + Rb[r * 3 + 1] * A[c * 3 + 1] + Rb[r * 3 + 2] * A[c * 3 + 2]) * ir + d[r] * bv[c] * ib);
#pragma unroll
                    for (int r = 0; r < 3; ++r) Gt[r] -= wgt * (Ra[r * 3 + 0] * bv[0] + Ra[r * 3 + 1] * bv[1] + Ra[r * 3 + 2] * bv[2]) * ib;
                } else {                      // this image is `b`
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            GR[r * 3 + c] += wgt * (Ra[r * 3 + 0] * A[0 * 3 + c] + Ra[r * 3 + 1] * A[1 * 3 + c] + Ra[r * 3 + 2] * A[2 * 3 + c]) * ir;
#pragma unroll
                    for (int r = 0; r < 3; ++r) Gt[r] += wgt * (Ra[r * 3 + 0] * bv[0] + Ra[r * 3 + 1] * bv[1] + Ra[r * 3 + 2] * bv[2]) * ib;
                }
            }
        }
        real gq[4];
        quat_rot_backward(GR, u, nrm, gq);
        float* o = p.grad_im_poses + i * 7;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (float)gq[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) o[4 + k] = (float)(Gt[k] * signed_expm1_grad(q[4 + k]));
        const real f = exp((real)p.im_focals[p.n_focals == 1 ? 0 : i] / p.focal_break);
        const real gf = I[12] * f / p.focal_break;
        if (p.n_focals == 1) focal_part += gf;
        else p.grad_im_focals[i] = (float)gf;
    }
    // ---- loss and the shared focal's gradient: per-thread partials added in thread order ----------------------------------------------
    red[tid] = loss_part;
    __syncthreads();
    if (tid == 0) {
        real s = 0;
        for (int k = 0; k < 256; ++k) s += red[k];
        p.loss[0] = (float)s;
    }
    __syncthreads();
    if (p.n_focals == 1) {
        red[tid] = focal_part;
        __syncthreads();
        if (tid == 0) {
            real s = 0;
            for (int k = 0; k < 256; ++k) s += red[k];
            p.grad_im_focals[0] = (float)s;
        }
    }
}

int check(const geo4d_align_small_t& p, bool grads) {
    if (p.n_imgs <= 0 || p.n_groups <= 0 || p.n_slots <= 0 || p.slots_per_group <= 0 || p.n_slots != p.n_groups * p.slots_per_group ||
        (p.n_focals != 1 && p.n_focals != p.n_imgs) || !p.im_poses || !p.im_focals || !p.pw_poses || p.focal_break <= 0.f || p.base_scale <= 0.f) {
        geo4d_set_error("align_small: bad arguments");
        return GEO4D_EINVAL;
    }
    if (!grads && (!p.cams || !p.slot_trf)) { geo4d_set_error("align_refresh: cams / slot_trf"); return GEO4D_EINVAL; }
    if (p.slot_st && (!p.s_depth || !p.t_depth || !p.depth_ok)) { geo4d_set_error("align_refresh: slot_st needs s_depth / t_depth / depth_ok"); return GEO4D_EINVAL; }
    if (grads && p.traj && (!p.traj_align || !p.traj_valid || !p.slot_img || !p.img_slot_ptr || !p.img_slot_idx || !p.grad_traj)) {
        geo4d_set_error("align_small_grads: the trajectory term needs traj_align / traj_valid / slot_img / img_slot_ptr / img_slot_idx / grad_traj");
        return GEO4D_EINVAL;
    }
    if (grads && (!p.img_sums || !p.slot_sums || !p.group_ptr || !p.group_entries || !p.group_sums || !p.scale_terms || !p.grad_im_poses || !p.grad_im_focals || !p.grad_pw_poses || !p.loss ||
                  p.n_listed_slots < 0 || p.n_listed_slots > p.n_slots || (!p.grad_s_depth) != (!p.grad_t_depth))) {
        geo4d_set_error("align_small_grads: bad arguments");
        return GEO4D_EINVAL;
    }
    return GEO4D_OK;
}

}  // namespace

extern "C" int geo4d_align_refresh(const geo4d_align_small_t* pp, void* stream) {
    if (!pp) return GEO4D_EINVAL;
    const int rc = check(*pp, false);
    if (rc != GEO4D_OK) return rc;
    hipLaunchKernelGGL(align_refresh_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *pp);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

extern "C" int geo4d_align_small_grads(const geo4d_align_small_t* pp, void* stream) {
    if (!pp) return GEO4D_EINVAL;
    const int rc = check(*pp, true);
    if (rc != GEO4D_OK) return rc;
    hipLaunchKernelGGL(align_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *pp);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}
