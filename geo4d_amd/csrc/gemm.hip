// geo4d_amd/csrc/gemm.hip — the workhorse: implicit-GEMM convolution / linear / batched GEMM on MFMA.
//
// One kernel covers every dense contraction of the Geo4D hot path:
//   nn.Linear                         (1 tap, Hin=Win=Hout=Wout=1, F = M rows)
//   nn.Conv2d 3x3 / 1x1, stride 1|2   (openaimodel3d.py:154,179 ResBlock; :51-78 Downsample)
//   nearest-2x upsample + Conv2d 3x3  (openaimodel3d.py:80-106, ae_modules.py:111-127) — the upsample is
//                                      folded into the gather (src = dst >> 1), nothing is materialised
//   nn.Conv3d (3,1,1), pad (1,0,0)    (openaimodel3d.py:257-266 TemporalConvBlock) — 3 temporal taps
//   batched Q.K^T / P.V GEMMs         (ae_modules.py:53-78 VAE AttnBlock, d = 512)
// Activations are channels-last tokens [F, H*W, C]; weights are packed [N][taps*Cin] (K-major, tap-major).
// out[m][n] = epilogue(alpha * sum_k A_gather[m][k] * W[n][k])
//
// Tile: BM x BN x 128 bytes of K per stage, 256 threads = 4 waves, each wave owns (BM/WM) x (BN/WN) as
// 32x32 MFMA blocks. Register-staged global->LDS with a 2-deep LDS ring (one barrier per stage), LDS rows
// padded to 144 B so that both the ds_write_b128 (8 lanes = one row) and the fragment ds_read_b128
// (16-lane groups, 16 distinct rows) are bank-conflict free (MI355X_MICROARCH.md §LDS).
// Workgroup ids are remapped so each XCD (private 4 MiB L2) walks a contiguous range of tiles with the
// N-tile index fastest: neighbours share the gathered A panel.
#include "common.h"
#include "geo4d_hip.h"

namespace {

constexpr int PITCH = 144;  // LDS row pitch in bytes (128 B payload + 16 B pad)
constexpr int BKC = 8;      // 16-byte chunks per row per stage

__device__ __forceinline__ void store_out(void* O, long idx, float v, int dt) {
    if (dt == GEO4D_F32) ((float*)O)[idx] = v;
    else if (dt == GEO4D_BF16) ((unsigned short*)O)[idx] = f32_to_bf16_bits(v);
    else ((unsigned short*)O)[idx] = f32_to_f16_bits(v);
}
__device__ __forceinline__ float load_res(const void* R, long idx, int dt) {
    if (dt == GEO4D_F32) return ((const float*)R)[idx];
    if (dt == GEO4D_BF16) return bf16_bits_to_f32(((const unsigned short*)R)[idx]);
    return f16_bits_to_f32(((const unsigned short*)R)[idx]);
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const geo4d_conv_gemm_t p) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = BKC * EPC;
    constexpr int MB = BM / WM / 32, NB = BN / WN / 32;
    constexpr int ACH = BM * BKC / 256, BCH = BN * BKC / 256;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(MB >= 1 && NB >= 1 && ACH >= 1 && BCH >= 1, "tile too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, g = lane >> 5;
    const int wr = wave / WN, wc = wave % WN;
    const long lid = xcd_remap((long)blockIdx.x, (long)gridDim.x);
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tm = (int)(lid / tiles_n), tn = (int)(lid - (long)tm * tiles_n);
    const long bz = blockIdx.y;
    const T* __restrict__ A = (const T*)p.A + bz * p.a_bs;
    const T* __restrict__ W = (const T*)p.W + bz * p.w_bs;

    const int ccol = tid & 7;
    const int r0 = tid >> 3;

    // ---- decode the output pixels of the A rows this thread stages ------------------------
    int rf[ACH], rt[ACH], ry[ACH], rx[ACH];
    bool rv[ACH];
    const int hw = p.Hout * p.Wout;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int m = tm * BM + r0 + i * 32;
        rv[i] = m < p.M;
        const int mm = rv[i] ? m : 0;
        const int f = mm / hw;
        const int rem = mm - f * hw;
        const int oy = rem / p.Wout;
        const int ox = rem - oy * p.Wout;
        rf[i] = f;
        rt[i] = f % p.T;
        ry[i] = oy * p.stride - p.ph;
        rx[i] = ox * p.stride - p.pw;
    }
    long wrow[BCH];
    bool wv[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int n = tn * BN + r0 + i * 32;
        wv[i] = n < p.N;
        wrow[i] = (long)(wv[i] ? n : 0) * p.ldw + ccol * EPC;
    }
    const int hlim = p.ups == 2 ? 2 * p.Hin : p.Hin;
    const int wlim = p.ups == 2 ? 2 * p.Win : p.Win;
    const int ush = p.ups == 2 ? 1 : 0;

    u32x4 ra[ACH], rb[BCH];
    // slab cursor (uniform across the block)
    int c0 = 0, kx = 0, ky = 0, kt = 0;
    long k0 = 0;
    auto load_slab = [&]() {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int iy = ry[i] + ky, ix = rx[i] + kx, tt = rt[i] + kt - p.pt;
            const bool ok = rv[i] && (unsigned)iy < (unsigned)hlim && (unsigned)ix < (unsigned)wlim &&
                            (unsigned)tt < (unsigned)p.T;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) {
                const long pix = ((long)(rf[i] + kt - p.pt) * p.Hin + (iy >> ush)) * p.Win + (ix >> ush);
                v = *(const u32x4*)(A + pix * p.lda + c0 + ccol * EPC);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (wv[i]) v = *(const u32x4*)(W + wrow[i] + k0);
            rb[i] = v;
        }
        // advance cursor
        k0 += BK;
        c0 += BK;
        if (c0 >= p.Cin) {
            c0 = 0;
            if (++kx == p.KW) { kx = 0; if (++ky == p.KH) { ky = 0; ++kt; } }
        }
    };
    auto store_slab = [&](int buf) {
        char* base = smem + buf * (BM + BN) * PITCH;
#pragma unroll
        for (int i = 0; i < ACH; ++i) *(u32x4*)(base + (r0 + i * 32) * PITCH + ccol * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < BCH; ++i) *(u32x4*)(base + (BM + r0 + i * 32) * PITCH + ccol * 16) = rb[i];
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nslab = p.K / BK;
    load_slab();
    store_slab(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) load_slab();
        const char* abase = smem + buf * (BM + BN) * PITCH + (wr * MB * 32 + li) * PITCH + g * 16;
        const char* bbase = smem + buf * (BM + BN) * PITCH + (BM + wc * NB * 32 + li) * PITCH + g * 16;
#pragma unroll
        for (int kk = 0; kk < BKC / 2; ++kk) {
            u32x4 fa[MB], fb[NB];
#pragma unroll
            for (int a = 0; a < MB; ++a) fa[a] = *(const u32x4*)(abase + a * 32 * PITCH + kk * 32);
#pragma unroll
            for (int b = 0; b < NB; ++b) fb[b] = *(const u32x4*)(bbase + b * 32 * PITCH + kk * 32);
#pragma unroll
            for (int a = 0; a < MB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) cmma<T>(acc[a][b], fa[a], fb[b]);
        }
        if (s + 1 < nslab) store_slab(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue -------------------------------------------------------------------------
    void* O = (char*)p.O;
    const long obase = bz * p.o_bs;
    const long rbase = bz * p.r_bs;
    const int row_w0 = tm * BM + wr * MB * 32;
    const int col_w0 = tn * BN + wc * NB * 32;
    if (p.act == 2) {
        if constexpr (NB == 2) {
            // GEGLU: packed weights interleave 32 value columns with their 32 gate columns.
            const int ocol = (col_w0 >> 1) + li;
            const int ncol = p.N >> 1;
            const float bx = p.bias ? p.bias[col_w0 + li] : 0.f;
            const float bg = p.bias ? p.bias[col_w0 + 32 + li] : 0.f;
            if (col_w0 + 32 + li < p.N) {
#pragma unroll
                for (int a = 0; a < MB; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row_w0 + a * 32 + acc_row(r, g);
                        if (row < p.M) {
                            const float xv = acc[a][0][r] * p.alpha + bx;
                            const float gv = acc[a][1][r] * p.alpha + bg;
                            store_out(O, obase + (long)row * p.ldo + ocol, xv * gelu_erf_f(gv), p.out_dtype);
                        }
                    }
            }
            (void)ncol;
        }
        return;
    }
    float bcol[NB];
    bool cok[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = col_w0 + b * 32 + li;
        cok[b] = col < p.N;
        bcol[b] = (p.bias && !p.bias_per_row && cok[b]) ? p.bias[col] : 0.f;
    }
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_w0 + a * 32 + acc_row(r, g);
            if (row >= p.M) continue;
            const float brow = (p.bias && p.bias_per_row) ? p.bias[row] : 0.f;
            const long rboff = p.rowbias ? (long)(row / p.rowbias_div) * (p.ldrb ? p.ldrb : (long)p.N) : 0;
            long obase_row, ostride_col;
            if (p.out_nchw) {
                // [B][N][T][hw]  (T = 1 gives plain NCHW per frame)
                const int f = row / hw;
                const int bb = f / p.T, tt = f - bb * p.T;
                obase_row = ((long)bb * p.ldo * p.T + tt) * hw + (row - f * hw);  // ldo = channels of the NCTHW tensor
                ostride_col = (long)p.T * hw;
            } else {
                obase_row = (long)row * p.ldo;
                ostride_col = 1;
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (!cok[b]) continue;
                const int col = col_w0 + b * 32 + li;
                float v = acc[a][b][r] * p.alpha + bcol[b] + brow;
                if (p.rowbias) v += p.rowbias[rboff + col];
                if (p.act == 1) v = silu_f(v);
                if (p.R) v += load_res(p.R, rbase + (long)row * p.ldr + col, p.out_dtype);
                store_out(O, obase + obase_row + (long)col * ostride_col, v, p.out_dtype);
            }
        }
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg(const geo4d_conv_gemm_t& p, hipStream_t stream) {
    constexpr int smem = 2 * (BM + BN) * PITCH;
    static bool attr_set = false;
    auto kern = conv_gemm_kernel<T, BM, BN, WM, WN>;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
            geo4d_set_error("hipFuncSetAttribute(max dynamic LDS) failed");
            return GEO4D_EIO;
        }
        attr_set = true;
    }
    const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    dim3 grid((unsigned)tiles, (unsigned)p.batch, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
    GEO4D_CHECK_LAUNCH();
    return GEO4D_OK;
}

// Tile choice: score = MFMA efficiency of the tile shape x useful fraction x how full the last wave of
// workgroups is (2 workgroups fit per CU by LDS => 512 slots on 256 CUs).
struct TileCfg { int bm, bn; float eff; };

template <typename T>
int launch_typed(const geo4d_conv_gemm_t& p, hipStream_t stream) {
    static const TileCfg cfgs[] = {{128, 128, 1.00f}, {128, 64, 0.85f}, {64, 128, 0.80f}, {64, 64, 0.62f}, {128, 32, 0.50f}};
    int best = -1;
    float best_score = -1.f;
    for (int i = 0; i < 5; ++i) {
        const TileCfg& c = cfgs[i];
        if (p.act == 2 && i >= 3) continue;  // GEGLU needs 64-wide wave tiles (NB == 2)
        if (p.tile_hint && i != p.tile_hint - 1) continue;
        const double tm = (p.M + c.bm - 1) / c.bm, tn = (p.N + c.bn - 1) / c.bn;
        const double tiles = tm * tn * p.batch;
        const double useful = ((double)p.M * p.N * p.batch) / (tiles * c.bm * c.bn);
        const double waves = (tiles + 511) / 512;
        const double fill = tiles / (double)((long)waves * 512);
        const float score = (float)(c.eff * useful * (0.35 + 0.65 * fill));
        if (score > best_score) { best_score = score; best = i; }
    }
    switch (best) {
        case 0: return launch_cfg<T, 128, 128, 2, 2>(p, stream);
        case 1: return launch_cfg<T, 128, 64, 4, 1>(p, stream);
        case 2: return launch_cfg<T, 64, 128, 2, 2>(p, stream);
        case 3: return launch_cfg<T, 64, 64, 2, 2>(p, stream);
        case 4: return launch_cfg<T, 128, 32, 4, 1>(p, stream);
    }
    geo4d_set_error("conv_gemm: no tile configuration");
    return GEO4D_EINVAL;
}

}  // namespace

extern "C" int geo4d_conv_gemm(const geo4d_conv_gemm_t* pp, void* stream) {
    if (!pp) return GEO4D_EINVAL;
    geo4d_conv_gemm_t p = *pp;
    const int esz = p.dtype == GEO4D_F32 ? 4 : 2;
    const int epc = 16 / esz;
    const int bk = BKC * epc;
    if (p.dtype < 0 || p.dtype > 2 || p.out_dtype < 0 || p.out_dtype > 2) { geo4d_set_error("conv_gemm: bad dtype"); return GEO4D_EINVAL; }
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.batch <= 0) { geo4d_set_error("conv_gemm: empty problem"); return GEO4D_EINVAL; }
    if (p.Cin % bk || p.K != p.Cin * p.KT * p.KH * p.KW) { geo4d_set_error("conv_gemm: Cin must be a multiple of the 128-byte K slab and K = taps*Cin"); return GEO4D_EINVAL; }
    if ((p.lda * esz) % 16 || (p.ldw * esz) % 16 || ((uintptr_t)p.A % 16) || ((uintptr_t)p.W % 16) || (p.a_bs * esz) % 16 || (p.w_bs * esz) % 16) {
        geo4d_set_error("conv_gemm: operands must be 16-byte aligned");
        return GEO4D_EINVAL;
    }
    if (p.T <= 0 || p.Hout <= 0 || p.Wout <= 0 || p.Hin <= 0 || p.Win <= 0 || p.M % (p.Hout * p.Wout)) { geo4d_set_error("conv_gemm: bad geometry"); return GEO4D_EINVAL; }
    if ((p.M / (p.Hout * p.Wout)) % p.T) { geo4d_set_error("conv_gemm: frames not a multiple of T"); return GEO4D_EINVAL; }
    if (p.ups != 1 && p.ups != 2) { geo4d_set_error("conv_gemm: ups must be 1 or 2"); return GEO4D_EINVAL; }
    if (p.act == 2 && (p.N % 64 || p.out_nchw || p.R)) { geo4d_set_error("conv_gemm: GEGLU needs N % 64 == 0, row-major output, no residual"); return GEO4D_EINVAL; }
    if (p.rowbias && p.rowbias_div <= 0) { geo4d_set_error("conv_gemm: rowbias_div"); return GEO4D_EINVAL; }
    if (p.batch > 65535) { geo4d_set_error("conv_gemm: batch too large"); return GEO4D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    switch (p.dtype) {
        case GEO4D_F32: return launch_typed<float>(p, s);
        case GEO4D_BF16: return launch_typed<bf16_t>(p, s);
        default: return launch_typed<f16_t>(p, s);
    }
}
