/* geo4d_hip.h — C ABI of libgeo4d_hip.so, the MI355X (gfx950 / CDNA4) kernel library behind the Geo4D
 * denoise + decode hot path.
 *
 * The reference (jzr99/Geo4D) has NO native boundary: its plug-in surface is the Python config registry
 * (utils/utils.py:27-42 instantiate_from_config) and below it only PyTorch ATen ops + xformers. This header is
 * therefore the boundary a reference maintainer would bind with ctypes: every entry point
 * replaces the ATen/xformers call(s) cited next to it. Conventions:
 *   - plain pointers and sizes only; all pointers are DEVICE pointers owned by the caller (PyTorch allocator);
 *     the library never allocates, frees, synchronises or copies — every call only enqueues kernels on `stream`
 *     (a hipStream_t passed as void*), so every call is hipGraph-capturable;
 *   - activations are channels-last "tokens": row (b*T + t)*H*W + y*W + x, `ld*` = row pitch in ELEMENTS;
 *     dtype codes: 0 = f32 (exact parity mode, v_mfma_f32_32x32x2_f32), 1 = bf16, 2 = f16 (MFMA 32x32x16, fp32 acc);
 *     3 = bf16x3 (conv_gemm / attention only): operands are f32 in memory (4 bytes per element, `ld*` in elements) and
 *     are multiplied as bf16 hi + bf16 lo with three bf16 MFMAs per product (~16 mantissa bits at 1/3 of the bf16
 *     rate instead of 1/16 for f32 MFMA) — the mode that meets the 1e-3 parity bar; every other entry point takes 0 for it;
 *     4 = f16x2 (conv_gemm only, round 5; operand layout of round 6 = ABI 8): W PRE-SPLIT as [8 x f16 hi | 8 x f16 lo] per 8
 *     K-elements (4 bytes per element, `ldw` / `w_bs` in elements), A = PLAIN f16 rows (2 bytes per element, `lda` / `a_bs` in f16
 *     elements; a_split must be 2); the product is a.w_hi + a.w_lo — TWO f16 MFMAs: the activation carries 11 mantissa bits, the
 *     weight ~22. Used for the GEMMs of the "bf16x3m" mode whose A operand is a normalised branch activation (DESIGN.md section 3;
 *     tests/precision_sim.py); tile hints 0 or >= 22; f32 rows out, or (o_split = 2) plain f16 rows = the next dtype-4 launch's A;
 *   - every row pitch / base pointer must be 16-byte aligned (kernels move 16-byte chunks);
 *   - return 0 on success, negative errno-style code otherwise (-22 EINVAL, -95 ENOTSUP, -5 EIO = HIP launch
 *     error); geo4d_last_error() returns a thread-local message. Kernels never abort().
 *   - thread model: one host thread per device/stream; no global mutable state except per-kernel attribute caches.
 */
#ifndef GEO4D_HIP_H
#define GEO4D_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GEO4D_ABI_VERSION 8

/* Implicit-GEMM convolution / linear / batched GEMM:  out = epilogue(alpha * gather(A) . W^T)
 * replaces F.linear (attention.py:52-56,420,437), F.conv2d 3x3/1x1 stride 1|2 (openaimodel3d.py:154,179,65-67;
 * ae_modules.py:199-226), F.interpolate(nearest,2x)+conv (openaimodel3d.py:99-105; ae_modules.py:123-127),
 * F.conv3d (3,1,1) (openaimodel3d.py:257-266), torch.bmm (ae_modules.py:64,72), `+ emb_out[..., None, None]`
 * (openaimodel3d.py:222-230), residual adds, GEGLU (attention.py:415-422), SiLU. */
typedef struct geo4d_conv_gemm_t {
    const void* A;       /* input tokens [F*Hin*Win][lda]                              */
    const void* W;       /* packed weights [N][ldw], K = KT*KH*KW*Cin, tap-major        */
    void* O;             /* output [M][ldo] (or NCTHW when out_nchw)                    */
    const float* bias;   /* [N] (or [M] when bias_per_row), may be NULL                 */
    const float* rowbias;/* [M/rowbias_div][N] fp32 added per row group, may be NULL    */
    const void* R;       /* residual [M][ldr], dtype = out_dtype, may be NULL           */
    const void* zeros;   /* >= 16 zero bytes in device memory (source of padding taps)  */
    void* workspace;     /* fp32 scratch for split-K slabs (caller-owned, may be NULL)  */
    size_t workspace_bytes;
    long lda, ldw, ldo, ldr;
    long ldrb;           /* row pitch of rowbias in floats (0 = N)                      */
    long a_bs, w_bs, o_bs, r_bs; /* batch strides in elements (batched GEMM)            */
    int M, N, K, batch;
    int Cin;             /* channels per tap; multiple of 128 B worth of elements       */
    int T, Hin, Win, Hout, Wout;
    int KT, KH, KW, pt, ph, pw, stride, ups;
    int rowbias_div;
    int bias_per_row;
    int act;             /* 0 none, 1 SiLU, 2 GEGLU (packed pairs of 32 columns), 3 GELU (erf)   */
    int dtype;           /* A/W element type                                            */
    int out_dtype;       /* O/R element type                                            */
    int out_nchw;        /* 1: store O as [B][ldo][T][Hout*Wout] (ldo = channel count of the
                            destination tensor; O may point at a channel offset inside it) */
    int tile_hint;       /* 0 auto; 1..5 = 128x128, 128x64, 64x128, 64x64, 128x32 (4 waves, 2-stage ring);
                            8 waves, one tile per CU: 11 = 256x128, 13 = 256x256; 16 = 160x320 with 10 waves
                            (N = 320 layers: one tile per CU at M = 40960), 17 = 160x160 with 5 waves; no
                            GEGLU on 16 / 17. Second generation, bf16 / bf16x3 only (16x16x32 MFMA, register epilogue, persistent
                            workgroups; no NCTHW output, no gn_colsum): 22 = 256x256, 23 = 160x320 (8 waves, 80x80 wave
                            tiles), 25 = 128x128, 27 = 64x128, 28 = 64x64 (4 waves); GEGLU on 22, 25, 27. Third generation: the
                            same MFMA form and epilogue under a PHASED K loop (4 phases per slab, counted LDS-DMA waits, the
                            staging cursor two slabs ahead across tiles; 8 waves, one workgroup per CU): 71 = 192x256,
                            72 = 160x320, 73 = 256x128, 74 = 128x256; GEGLU on 71, 74; launches with an odd number or fewer
                            than 4 K slabs per tile, an uneven split-K or outputs that are not 4-element aligned run on
                            22 / 23 / 25 / 25 instead (same bits: every tile sums in the same order).
                            Others (incl. the hints retired in round 4: 12, 14, 21, 24, 26, 29, 31..39): -EINVAL */
    int split_k;         /* 0 auto (powers of two), 1 never, n >= 2: n-way split (needs workspace; tile hints >= 21 take any n) */
    int debug_ablate;    /* 0 in production. 2 (tests only, tile hints >= 22): launch 3 persistent workgroups whatever the problem size,
                            so that small test shapes walk the persistent tile loop; 16 + g (A/B runs): tile order with GROUP_M = g row-tiles
                            fastest instead of the column-fastest default (-9 % L2-miss fetch, -0.3 % frames/s: profiles/r05_tile_order.md) */
    float alpha;
    int a_split, w_split;/* dtype 3 (bf16x3): the operand is stored PRE-SPLIT, per 8 K-elements
                            [8 x bf16 hi | 8 x bf16 lo] (32 bytes, pack.py split_bf16) instead of 8 raw f32; dtype 4 (f16x2): w_split = 1
                            (f16 halves: pack.py split_f16) and a_split = 2 = A is plain f16 rows (geo4d_groupnorm_t.split_out = 2,
                            geo4d_layernorm_split fmt 2, o_split = 2 of a previous dtype-4 launch) */
    int o_split;         /* 2 (dtype 4 only; out_dtype F32, no residual, no split-K): O is written as PLAIN f16 rows (ldo / o_bs in f16 elements,
                            stored columns % 8 == 0, 16-byte aligned rows), values clamped to the finite f16 range, NaN kept - the A operand of a following
                            dtype-4 launch (GEGLU -> FF-out). 1: dtype 3 (bf16x3), a_split and w_split set, out_dtype F32: O is written in the pre-split operand
                            format ([8 x bf16 hi | 8 x bf16 lo] per 8 output columns; ldo / o_bs still count columns) - the producer
                            side of a_split (GEGLU -> FF-out chain; q | k and V^T of the spatial attention, geo4d_attention_t.qkv_split).
                            Stored columns % 8 == 0, ldo % 8 == 0, 32-byte aligned rows, no split-K, any epilogue incl. residual;
                            served by the second- / third-generation tiles (first-generation hints are re-routed) */
    float* gn_colsum;    /* optional [M/rows][N][2] fp32 (rows = geo4d_conv_gemm_colsum_rows(p): 32 for the first generation): per row block and output column, (sum, sum of squares) of the values this
                            launch stores - the statistics pass of the GroupNorm that consumes O, produced for free by the epilogue
                            (geo4d_groupnorm_t.colsum). Needs M % 32 == 0, N % 8 == 0, row-major 16-byte aligned output, batch 1,
                            no GEGLU; no split-K on the first generation. Round 6: a split-K launch of tile hints >= 22 emits them from its
                            reduce launch, per 32 rows (N % 64 == 0) or, where a frame's Hout x Wout rows are a multiple of 8 but not of 32, per
                            8 rows (N % 256 == 0): geo4d_conv_gemm_colsum_rows says which. */
    unsigned long long* sat_count; /* optional DEBUG counter in device memory (NULL in production): o_split = 2 launches add the number of
                            (wave, store) lanes whose value lay beyond the finite f16 range and was clamped. Tests assert it stays 0. */
} geo4d_conv_gemm_t;
int geo4d_conv_gemm(const geo4d_conv_gemm_t* p, void* stream);
/* Rows of the output that ONE gn_colsum entry of the launch `*p` describes covers (tile_hint / split_k as they will be launched, the
   gn_colsum field itself ignored): 32 for the first-generation tiles, the wave tile's rows (32..128) for tile hints >= 21 without split-K,
   32 or 8 for their split-K launches (the reduce launch sums them); 0 = this launch cannot emit the sums. gn_colsum then has
   [M / rows][N][2] floats. */
int geo4d_conv_gemm_colsum_rows(const geo4d_conv_gemm_t* p);

/* GroupNorm(groups) [+SiLU] over tokens [F][HW][C]; statistics per (F / frames_per_stat, group) in fp32.
 * replaces nn.GroupNorm / GroupNormSpecific + nn.SiLU (basics.py:76-87; openaimodel3d.py:151-153,174-176,256-266;
 * attention.py:265,331; ae_modules.py:10-16). */
typedef struct geo4d_groupnorm_t {
    const void* x; void* y;
    const float* gamma; const float* beta;
    void* workspace; size_t workspace_bytes;
    long ldx, ldy;
    int F, HW, C, groups, frames_per_stat;
    int act;             /* 0 none, 1 SiLU / swish                                       */
    int dtype;
    float eps;
    const float* colsum; /* optional: [F*HW/colsum_rows][C][2] column sums written by the producing geo4d_conv_gemm (gn_colsum); when
                            given the pass over x that computes the statistics is skipped */
    int split_out;       /* producers of GEMM operands (dtype F32 only, C % 8 == 0): 1: y is written in the PRE-SPLIT operand format of
                            geo4d_conv_gemm_t.a_split, per 8 channels [8 x bf16 hi | 8 x bf16 lo] (bf16x3 consumers; ldy counts channels = 4-byte
                            units); 2: y = PLAIN f16 rows (ldy in f16 elements), values clamped to the finite f16 range, NaN kept (dtype-4 consumers) */
    int colsum_rows;     /* rows per `colsum` entry (what geo4d_conv_gemm_colsum_rows returned for the producing launch); 0 = 32;
                            (frames_per_stat x HW) % colsum_rows == 0 */
    unsigned long long* sat_count; /* optional DEBUG counter (see geo4d_conv_gemm_t.sat_count) of the split_out = 2 clamp; NULL in production */
} geo4d_groupnorm_t;
size_t geo4d_groupnorm_workspace(int F, int HW, int groups, int frames_per_stat);
int geo4d_groupnorm(const geo4d_groupnorm_t* p, void* stream);

/* LayerNorm over the last dim; replaces nn.LayerNorm (attention.py:225-227). */
int geo4d_layernorm(const void* x, long ldx, void* y, long ldy, int M, int C, float eps, const float* gamma,
                    const float* beta, int dtype, void* stream);

/* the same for an f32 x, writing y in a GEMM-operand format (geo4d_groupnorm_t.split_out: fmt 1 = pre-split bf16 hi | lo, ldy in
 * 4-byte units; fmt 2 = plain f16 rows, ldy in f16 elements, clamped, `sat_count` = optional debug counter of the clamp or NULL); C % 8 == 0. */
int geo4d_layernorm_split(const void* x, long ldx, void* y, long ldy, int M, int C, float eps, const float* gamma,
                          const float* beta, int fmt, unsigned long long* sat_count, void* stream);

/* y = softmax(scale * x) per row, x fp32; replaces F.softmax in the VAE AttnBlock (ae_modules.py:66-68). */
int geo4d_softmax_rows(const float* x, long ldx, void* y, long ldy, long rows, int cols, float scale, int out_dtype,
                       void* stream);
/* the same with a causal mask: row r attends to columns <= r % causal_period (the attn_mask of the OpenCLIP text transformer,
 * condition.py:217-221; rows of several (batch, head) score matrices of `causal_period` rows each stacked). */
int geo4d_softmax_rows_causal(const float* x, long ldx, void* y, long ldy, long rows, int cols, float scale, int out_dtype,
                              int causal_period, void* stream);

/* Fused multi-head attention, d_head = 64, no mask, up to two key/value sets with independent softmaxes whose
 * outputs are summed. q rows: b*Nq + i, head h at columns [64h, 64h+64). K rows of set s: (b / kv_div[s])*Nk[s] + j.
 * V is passed TRANSPOSED: vt[s] + (b / kv_div[s]) * vt_bs[s] + (64h + d) * ldvt[s] + j  (row = channel, column = key;
 * the row must be readable — zero or finite — up to the next multiple of 16 bytes past Nk). `zeros`: >= 16 zero bytes.
 * replaces CrossAttention.forward / efficient_forward = xformers.ops.memory_efficient_attention
 * (attention.py:81-144, 146-209). */
typedef struct geo4d_attention_t {
    const void* q; void* o;
    const void* k[2]; const void* vt[2];
    const void* zeros;
    long ldq, ldo, ldk[2], ldvt[2], vt_bs[2];
    int Nk[2], kv_div[2];
    int B, H, Nq, nseg, head_dim, dtype;
    float scale;
    int split_out;       /* 4-byte storage only (dtype F32 or BF16X3 - the bf16x3 mode stores its activations as f32): o is written in the pre-split operand format of geo4d_conv_gemm_t.a_split (ldo still
                            counts channels) - the to_out projection consumes it without splitting again */
    int variant;         /* 0 = default; A/B builds of the same math: 1 = 128 query rows per workgroup, 2 = the same compiled for
                            4 waves per SIMD (16-bit types), 3 = 256 rows per workgroup, two query blocks per wave (nseg == 1),
                            4 = 256 rows per workgroup with the two blocks' phases skewed so every softmax has the other block's MFMAs
                            beside it (nseg == 1; bf16 / f16, and bf16x3 with qkv_split at one wave per SIMD), 5 = the bf16x3 form of 4
                            at two waves per SIMD */
    int qkv_split;       /* dtype 3 (bf16x3), nseg == 1 only: q, k[0] and vt[0] are stored in the PRE-SPLIT operand format (written by
                            geo4d_conv_gemm_t.o_split projections): per 8 elements of a row [8 x bf16 hi | 8 x bf16 lo]; ld* / vt_bs still
                            count 4-byte elements and must be multiples of 8, bases 32-byte aligned, Nk % 8 == 0. Same results, bit
                            for bit, as the raw-f32 inputs (the split is the same arithmetic, done once by the producer). */
} geo4d_attention_t;
int geo4d_attention(const geo4d_attention_t* p, void* stream);

/* Self-attention over T <= 16 frames for every (batch, pixel, head); tokens stay frame-major [B*T][HW][C].
 * replaces the einsum/softmax path of CrossAttention inside TemporalTransformer (attention.py:101-125, 365-412). */
int geo4d_temporal_attention(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                             int B, int T, int HW, int H, int head_dim, float scale, int dtype, void* stream);
/* the same with `split_out` (dtype GEO4D_F32 only = the storage type of the bf16x3 mode; dtype 3 is NOT accepted here): o in the
 * pre-split operand format, see geo4d_attention_t.split_out */
int geo4d_temporal_attention2(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                              int B, int T, int HW, int H, int head_dim, float scale, int dtype, int split_out, void* stream);

/* [B,C0,T,H,W] (+ [B,C1,T,H,W]) fp32 -> tokens [(b t) hw][Cpad] (zero padded);
 * replaces torch.cat([x] + c_concat, 1) + rearrange (ddpm3d.py:2540-2544; openaimodel3d.py:588). */
int geo4d_tokens_from_ncthw(const float* src0, int C0, const float* src1, int C1, void* out, int Cpad, int B, int T, int HW,
                            int dtype, void* stream);

/* out[m] = [a[m] | b[m]]; replaces torch.cat([h, hs.pop()], dim=1) (openaimodel3d.py:624-626). */
int geo4d_concat_channels(const void* a, long lda, int Ca, const void* b, long ldb, int Cb, void* out, long ldo, long M,
                          int dtype, void* stream);

/* y = f16(x) row by row: the A operand of a dtype-4 (two-pass f16) geo4d_conv_gemm launch made from an f32 tensor that has no normalising
 * producer in front of it - the VAE decoder's residual stream in front of its Upsample convolutions (ae_modules.py:111-127), bf16x3m class
 * "vaeup". Values clamped to the finite f16 range, NaN kept; `sat_count` = optional debug counter of the clamp (NULL in production). C % 8 == 0. */
int geo4d_cast_rows_f16(const float* x, long ldx, void* y, long ldy, long M, int C, unsigned long long* sat_count, void* stream);

/* y = x in the PRE-SPLIT bf16x3 operand format (geo4d_conv_gemm_t.a_split = 1: per 8 elements [8 x bf16 hi | 8 x bf16 lo]; ldy counts 4-byte units like
 * ldx): what the GEMM's in-register split computes per fragment and K slab, done ONCE - for the long-K launches whose A operand is a residual stream
 * (the Upsample convolutions of the U-Net and of the VAE decoder: every input element is gathered 9 x 4 times). Same arithmetic, same bits. C % 8 == 0. */
int geo4d_split_rows_bf16(const float* x, long ldx, void* y, long ldy, long M, int C, void* stream);

/* out[b] = [cos(t_b * f) | sin(t_b * f)]; replaces timestep_embedding (utils_diffusion.py:8-28). */
int geo4d_timestep_embedding(const long* t, const float* freqs, float* out, int B, int dim, void* stream);

/* out = act_out(act_in(x) . W^T + bias) + add, all fp32, M small; replaces time_embed / fps_embedding /
 * ResBlock.emb_layers (openaimodel3d.py:367-384, 166-172, 222). */
int geo4d_linear_small(const float* x, long ldx, const float* w, long ldw, const float* bias, const float* add, long ldadd,
                       float* out, long ldo, int M, int N, int K, int act_in, int act_out, void* stream);

/* In-place DDIM update of x with v-prediction and dynamic rescale; coefficients are read from
 * coef[*step_index] (6 floats per step) so one captured hipGraph replays for every step.
 * replaces DDIMSampler.p_sample_ddim arithmetic (ddim.py:232-277) + DDPM.predict_* (ddpm3d.py:278-290). */
int geo4d_ddim_step(float* x, const float* v, const float* noise, float* pred_x0, const float* coef, const int* step_index,
                    long n, void* stream);
/* Classifier-free guidance on fp32 [B][n] U-Net outputs:
 *   out = e_u + cfg_img (e_i - e_u) + scale (e_c - e_i)          (e_i == NULL: out = e_u + scale (e_c - e_u))
 *   guidance_rescale > 0: out = r * out * std(e_c)/std(out) + (1 - r) * out, unbiased std per batch sample over the n other elements
 * replaces the combination in DDIMSampler.p_sample_ddim (ddim.py:216-229; 3-way: ddim_multiplecond.py:229-236) and
 * rescale_noise_cfg (utils_diffusion.py:147-158). workspace: geo4d_cfg_combine_workspace(B) bytes, 8-byte aligned. */
size_t geo4d_cfg_combine_workspace(int B);
int geo4d_cfg_combine(const float* e_c, const float* e_u, const float* e_i, float* out, int B, long n, float scale, float cfg_img,
                      float guidance_rescale, void* workspace, size_t workspace_bytes, void* stream);
/* out[(b n)][:] = table[tokens[b][n]][:] + pos[n][:]; replaces token_embedding(text) + positional_embedding of the OpenCLIP text
 * tower (condition.py:212-214). table [vocab][width], pos [n_ctx][width] fp32; rows = B * n_ctx; out-of-range ids read row 0. */
int geo4d_embed_tokens(const long* tokens, const float* table, const float* pos, void* out, long ldo, long rows, int n_ctx,
                       int width, int vocab, int dtype, void* stream);
int geo4d_advance_index(int* idx, int delta, void* stream);
/* ts[b] = table[*idx] for b < B: the per-step `ts = torch.full((b,), step)` of ddim.py:170, read from a device table so
 * the captured step graph is step-independent. */
int geo4d_gather_timestep(const int* idx, const long* table, long* ts, int B, void* stream);

/* Plücker ray map -> camera-to-world matrices of one window. replaces raymap_to_camera_matrix
 * (scripts/evaluation/test_geo4d.py:539-557) -> cameras_from_plucker / rays_to_cameras (utils/rays.py:387-433, 301-368) ->
 * intersect_skew_lines_high_dim (utils/normalize.py:25-51) + compute_optimal_rotation_alignment (utils/rays.py:579-595), which the
 * reference runs on the host. ray / moment: fp32 planes [3][T][H][W] addressed as base + c*channel_stride + t*frame_stride + y*W + x
 * (so they can be channel views of the decoded [B,11,T,H,W] tensor); frame 0 is the reference frame; the maps are centre-cropped to
 * min(H,W)^2 like the reference. P_c2w: [T][4][4] fp32 row-major. workspace: geo4d_plucker_cameras_workspace bytes, 8-byte aligned. */
size_t geo4d_plucker_cameras_workspace(int T, int H, int W);
int geo4d_plucker_cameras(const float* ray, const float* moment, long channel_stride, long frame_stride, int T, int H, int W,
                          void* workspace, size_t workspace_bytes, float* P_c2w, void* stream);

/* Multi-window global alignment, one iteration's residual + gradients. replaces the autograd forward /
 * backward of LightPointCloudGroupOptimizer.forward's point-map term (dust3r/cloud_opt/optimizer_group.py:440-455 with
 * depth_to_pts3d :407-417 and l1_dist commons.py:86-87). A "slot" is one (window, frame) prediction; image i owns the slots
 * slot_idx[slot_ptr[i] .. slot_ptr[i+1]).  loss = inv_area * sum min(conf, conf_clamp) * | X_i - (sR_g P + st_g) |.
 *   pred [n_slots][H*W][3], conf [n_slots][H*W], logdepth / grad_logdepth [n_imgs][H*W]          (fp32, device)
 *   cams [n_imgs][16] = R (9, row-major) | t (3) | focal | ppx | ppy | 0      slot_trf [n_slots][12] = sR (9) | st (3)
 *   img_sums [n_imgs][14] = dL/dR (9) | dL/dt (3) | dL/dfocal | loss share
 *   slot_sums [n_slots][14] = dL/d(sR) (9) | dL/d(st) (3) | dL/ds_depth, dL/dt_depth (inverse-depth term; 0 when off), in CSR order
 * Inverse-depth term (optimizer_group.py:470-494), fused into the same pass when invdepth != NULL:
 *   loss += depth_weight * sum_{slot, pixel: q > 0.05} accepted_g * | 1 / (exp(logdepth) + 1e-6) - (s_g q + t_g) |
 *   invdepth [n_slots][H*W] = q, slot_st [n_slots][3] = (s_g, t_g, accepted_g in {0, 1}); the reference's factor 2 / total area is depth_weight.
 * workspace: geo4d_align_workspace bytes. Deterministic (fixed-order reductions). */
typedef struct geo4d_align_t {
    const float* pred; const float* conf; const float* logdepth; const float* cams; const float* slot_trf;
    const int* slot_ptr; const int* slot_idx;
    float* grad_logdepth; float* img_sums; float* slot_sums;
    void* workspace; size_t workspace_bytes;
    float* img_part; float* slot_part;   /* set by the library (views of workspace) */
    const float* invdepth; const float* slot_st;   /* inverse-depth term (NULL: off), see below */
    int n_imgs, n_slots, H, W, chunk_pixels, max_slots_per_image;
    float conf_clamp, inv_area, depth_weight;
} geo4d_align_t;
size_t geo4d_align_workspace(int n_imgs, int n_slots, int H, int W, int chunk_pixels);
int geo4d_align_residual(const geo4d_align_t* p, void* stream);
/* One torch.optim.Adam step on a flat fp32 tensor (no weight decay / amsgrad); `step` counts from 1.
 * replaces optimizer.step() of global_alignment_iter (dust3r/cloud_opt/base_opt_group.py:596-626) for the depth maps. */
int geo4d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                    float eps, int step, void* stream);
/* The same with {lr, 1 - beta1^t, sqrt(1 - beta2^t)} read from device memory (3 floats), so a captured hipGraph of one whole
 * alignment iteration can be replayed while the schedule advances on the device. */
int geo4d_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, const float* hyper, float beta1,
                        float beta2, float eps, void* stream);

/* The parameter side of one alignment iteration (round 3; base_opt_group.py:262-327 parameterisation, optimizer_group.py:529-541
 * relative_pose_loss): geo4d_align_refresh writes the residual kernel's inputs from the parameters - cams [n_imgs][16] =
 * (R row-major | t | f | ppx | ppy | 0) with R = rotation of the NORMALISED XYZW quaternion im_poses[i][0:4], t = signed_expm1 of
 * im_poses[i][4:7], f = exp(im_focals / focal_break); slot_trf [n_slots][12] = (s_g R_g | s_g t_g) of the slot's window g = slot /
 * slots_per_group with s_g = exp(pw_poses[g][7] + (norm_pw_scale ? log base_scale - mean_g pw_poses[g][7] : 0)).
 * geo4d_align_small_grads turns the residual kernel's gradient sums (img_sums [n_imgs][14]; slot_sums [n_listed_slots][14] in the CSR
 * order of slot_idx, listed per window by group_ptr / group_entries) into loss[0] = sum_i img_sums[i][13] + smooth_weight * sum_i
 * relative_pose_loss(cam_i, cam_i+1) and the gradients of im_poses [n_imgs][7], im_focals [n_focals] (1 = shared), pw_poses
 * [n_groups][8] and, when given, s_depth / t_depth [n_groups]. One workgroup, fixed-order sums, fp64 inside (several of the sums cancel).
 * replaces loss.backward() through the tiny parameter tensors (round 2: ~450 autograd launches per iteration). */
typedef struct geo4d_align_small_t {
    const float* im_poses; const float* im_focals; const float* pw_poses;
    float* cams; float* slot_trf;
    const float* img_sums; const float* slot_sums;
    const int* group_ptr; const int* group_entries;   /* window g owns CSR entries group_entries[group_ptr[g] .. group_ptr[g + 1]) (ascending) */
    double* group_sums; double* scale_terms;   /* scratch: [n_groups][14] and [n_groups] fp64 */
    float* grad_im_poses; float* grad_im_focals; float* grad_pw_poses; float* grad_s_depth; float* grad_t_depth; float* loss;
    /* inverse-depth term (all NULL: off): geo4d_align_refresh also writes slot_st [n_slots][3] = (s_depth[g], t_depth[g], depth_ok[g]) */
    float* slot_st; const float* s_depth; const float* t_depth; const float* depth_ok;
    /* trajectory term (traj NULL: off; optimizer_group.py:496-512): traj [n_slots][4][4] = every window's predicted camera-to-world
     * matrices, traj_align [n_groups][8] = traj_align_poses, traj_valid [n_groups] (0 / 1), slot_img [n_slots] = image of a slot,
     * img_slot_ptr / img_slot_idx = CSR of ALL slots per image; loss += traj_weight * sum relative_pose_loss(T_g [R_k | e^l t_k], cam_i);
     * grad_traj [n_groups][8] receives d/d traj_align_poses (zero rows for invalid windows); the camera side is added to grad_im_poses */
    const float* traj; const float* traj_align; const int* traj_valid; const int* slot_img; const int* img_slot_ptr; const int* img_slot_idx;
    float* grad_traj;
    int n_imgs, n_groups, n_slots, slots_per_group, n_listed_slots, n_focals, norm_pw_scale;
    float focal_break, base_scale, ppx, ppy, smooth_weight, translation_weight, traj_weight;
} geo4d_align_small_t;
int geo4d_align_refresh(const geo4d_align_small_t* p, void* stream);
int geo4d_align_small_grads(const geo4d_align_small_t* p, void* stream);

/* Start-up of the inverse-depth term (LightPointCloudGroupOptimizer._set_st_depth, optimizer_group.py:333-372, which calls
 * dust3r/depth_eval.py depth_evaluation(align_with_lad2=True) :147-330 per window). All arrays fp32 on the device, windows
 * contiguous: q / target / conf are [G][n], n = S * H * W.
 *   geo4d_lad_target: target[slot][px] = 1 / (exp(logdepth[slot_img[slot]][px]) + 1e-6)                     (:336-337)
 *   geo4d_lower_median: out[r] = torch.median(x[r]) (the LOWER median), bit-exact, by radix select; workspace rows * 260 * 4 bytes
 *   geo4d_lad_fit: (s, t)[g] minimising sum | s q + t - target | by torch.optim.Adam's arithmetic (default betas), started at
 *     s = median(target) / median(q), t = 0; a window stops when its loss repeats within tol (depth_eval.py:112-145, 218-221).
 *     active (NULL = all): windows with 0 are left at their start values. st_out [G][2]; info_out (may be NULL) [G][2] = (steps, last loss).
 *   geo4d_lad_delta: counts[g] = {#(max(a/target, target/a) < 1.25), #mask}, a = max(s q + t, 1e-5), mask = min(conf, conf_clamp) >
 *     conf_thr and q > q_thr (:297-301 under _set_st_depth's custom mask :340-342). */
size_t geo4d_lad_workspace(int G, long n);
int geo4d_lad_target(const float* logdepth, const int* slot_img, float* target, int n_slots, int HW, void* stream);
int geo4d_lower_median(const float* x, int rows, long n, float* out, void* workspace, size_t workspace_bytes, void* stream);
int geo4d_lad_fit(const float* q, const float* target, int G, long n, const unsigned char* active, float lr, int max_iters, float tol,
                  float* st_out, float* info_out, void* workspace, size_t workspace_bytes, void* stream);
int geo4d_lad_delta(const float* q, const float* target, const float* conf, const float* st, int G, long n, float conf_thr,
                    float conf_clamp, float q_thr, unsigned* counts, void* stream);

const char* geo4d_last_error(void);
int geo4d_abi_version(void);
/* sizeof of the parameter structs as the LIBRARY was compiled (which: 0 conv_gemm, 1 groupnorm, 2 attention, 3 align, 4 align_small; else 0):
 * bindings compare it with their own layout at load time. */
size_t geo4d_abi_struct_size(int which);

#ifdef __cplusplus
}
#endif
#endif
