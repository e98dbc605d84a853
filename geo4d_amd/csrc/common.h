// geo4d_amd/csrc/common.h — shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Everything here is written for ONE target: wave64, MFMA 32x32 tiles, 16-byte LDS/global
// accesses. There is no other backend and no portability layer.
//
// Central abstraction: the "16-byte k-chunk". Every MFMA operand (A or B) is fed per lane as one
// 16-byte chunk holding EPC = 16/sizeof(T) consecutive K elements of one row:
//   * 16-bit types (bf16/f16): EPC = 8  -> one v_mfma_f32_32x32x16_{bf16,f16}; lane (i = lane&31,
//     g = lane>>5) supplies K elements [8g, 8g+8) of row i.
//   * f32: EPC = 4 -> four v_mfma_f32_32x32x2_f32; MFMA j pairs element j of the A chunk with element
//     j of the B chunk (the two half-waves g = 0/1 cover two different k). Any bijection of k is legal
//     for a dot product as long as A and B use the same one, so a chunk-pair always contributes the
//     products of its EPC * 2 k-values.
// One cmma() therefore covers K = 2*EPC elements (16 for bf16/f16, 8 for f32) and the callers
// are dtype-agnostic: bf16 (bench), f16 and exact-f32 (parity mode) share one kernel source.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GEO4D_F32 0
#define GEO4D_BF16 1
#define GEO4D_F16 2
#define GEO4D_BF16X3 3   // f32 STORAGE, bf16 MFMA on a 3-term hi/lo split (x_hi.w_hi + x_hi.w_lo + x_lo.w_hi): ~16-bit mantissas
#define GEO4D_F16X2 4    // conv_gemm only: the weight PRE-SPLIT as [8 x f16 hi | 8 x f16 lo] per 8 K-elements, the activation PLAIN f16 rows (round 6;
                         // round 5 stored an unread lo half beside it); product = a.w_hi + a.w_lo (two f16 MFMAs): the activation carries 11
                         // mantissa bits, the weight ~22

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

struct bf16_t { unsigned short v; };
struct f16_t { unsigned short v; };
// "bf16x3": an operand element occupies 4 bytes like an f32. Activations ARE plain f32 in memory (every non-GEMM kernel
// runs its f32 instantiation on them) and are split into bf16 hi + bf16 lo in registers right before the MFMA; weights are
// split once at pack time and stored per 8 K-elements as [8 x bf16 hi | 8 x bf16 lo] (two 16-byte chunks = the same 32 bytes
// 8 f32 would take), so a fragment read is the same two ds_read_b128 for both layouts.
struct bf16x3_t { float v; };
// "f16x2" (round 5): the WEIGHT keeps 4 bytes per K element in the same [8 hi | 8 lo] grouping with f16 halves; the ACTIVATION is one f16
// per element (round 6: stored as plain f16 rows, 2 bytes per element): a.w ~ a.w_hi + a.w_lo - two f16 MFMAs per product instead of
// three bf16 ones. The activation is an f16 (11 bits), the weight hi + lo ~22 bits (pre-scaled by a power of two at pack time so that lo stays out of the f16 subnormals; the
// launch's alpha undoes the scale). Why that is enough for the long-K 3x3 convolutions and only for them: tests/precision_sim.py,
// tests/test_precision_floor.py (point-map drift 1.3e-4 .. 2.9e-4 where a full f16 pass costs 2e-3).
struct f16x2p_t { float v; };

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}
// round-to-nearest-even in hardware: v_cvt_pk_bf16_f32 (gfx950) — one instruction per PAIR instead of ~5 VALU per element
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int f32x2_to_bf16x2(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ unsigned int f32x2_to_f16x2(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    return (unsigned short)(f32x2_to_bf16x2(f, 0.f) & 0xffffu);
}
__device__ __forceinline__ float f16_bits_to_f32(unsigned short b) {
    return (float)__builtin_bit_cast(_Float16, b);
}
__device__ __forceinline__ unsigned short f32_to_f16_bits(float f) {
    return __builtin_bit_cast(unsigned short, (_Float16)f);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int EPC = 4;      // elements per 16-byte chunk
    static constexpr int DT = GEO4D_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int EPC = 8;
    static constexpr int DT = GEO4D_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_bits_to_f32(p->v); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { p->v = f32_to_bf16_bits(v); }
};
template <> struct Elem<f16_t> {
    static constexpr int EPC = 8;
    static constexpr int DT = GEO4D_F16;
    __device__ static __forceinline__ float ld(const f16_t* p) { return f16_bits_to_f32(p->v); }
    __device__ static __forceinline__ void st(f16_t* p, float v) { p->v = f32_to_f16_bits(v); }
};
template <> struct Elem<bf16x3_t> {
    static constexpr int EPC = 4;      // storage granularity is f32: 4 elements per 16-byte chunk
    static constexpr int DT = GEO4D_BF16X3;
    __device__ static __forceinline__ float ld(const bf16x3_t* p) { return p->v; }
    __device__ static __forceinline__ void st(bf16x3_t* p, float v) { p->v = v; }
};
template <> struct Elem<f16x2p_t> {
    static constexpr int EPC = 4;      // 4 bytes per K element, like bf16x3
    static constexpr int DT = GEO4D_F16X2;
    __device__ static __forceinline__ float ld(const f16x2p_t* p) { return p->v; }
    __device__ static __forceinline__ void st(f16x2p_t* p, float v) { p->v = v; }
};
// IsX3: the 4-byte split-operand LAYOUT family (LDS swizzle, fragment chunks 2q / 2q + 1, f32 rows in the epilogue); IsTwoPass: of
// that family, the f16 form that multiplies only the activation's hi half
template <typename T> struct IsX3 { static constexpr bool value = false; };
template <> struct IsX3<bf16x3_t> { static constexpr bool value = true; };
template <> struct IsX3<f16x2p_t> { static constexpr bool value = true; };
template <typename T> struct IsTwoPass { static constexpr bool value = false; };
template <> struct IsTwoPass<f16x2p_t> { static constexpr bool value = true; };

// ---- chunk <-> float conversion -----------------------------------------------------------
template <typename T> __device__ __forceinline__ void chunk_to_f32(const u32x4& c, float* out);
template <> __device__ __forceinline__ void chunk_to_f32<float>(const u32x4& c, float* out) {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = __uint_as_float(c[j]);
}
template <> __device__ __forceinline__ void chunk_to_f32<bf16_t>(const u32x4& c, float* out) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        out[2 * j] = __uint_as_float(c[j] << 16);
        out[2 * j + 1] = __uint_as_float(c[j] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void chunk_to_f32<f16_t>(const u32x4& c, float* out) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        out[2 * j] = f16_bits_to_f32((unsigned short)(c[j] & 0xffffu));
        out[2 * j + 1] = f16_bits_to_f32((unsigned short)(c[j] >> 16));
    }
}
template <typename T> __device__ __forceinline__ u32x4 f32_to_chunk(const float* in);
template <> __device__ __forceinline__ void chunk_to_f32<bf16x3_t>(const u32x4& c, float* out) {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = __uint_as_float(c[j]);
}
template <> __device__ __forceinline__ u32x4 f32_to_chunk<float>(const float* in) {
    u32x4 c;
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = __float_as_uint(in[j]);
    return c;
}
template <> __device__ __forceinline__ u32x4 f32_to_chunk<bf16x3_t>(const float* in) {
    u32x4 c;
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = __float_as_uint(in[j]);
    return c;
}
template <> __device__ __forceinline__ u32x4 f32_to_chunk<bf16_t>(const float* in) {
    u32x4 c;
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = f32x2_to_bf16x2(in[2 * j], in[2 * j + 1]);
    return c;
}
template <> __device__ __forceinline__ u32x4 f32_to_chunk<f16_t>(const float* in) {
    u32x4 c;
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = f32x2_to_f16x2(in[2 * j], in[2 * j + 1]);
    return c;
}

// ---- chunk-MMA: acc(32x32) += A-chunk x B-chunk -------------------------------------------
// C/D layout of every 32x32 MFMA on gfx950: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
template <typename T> __device__ __forceinline__ void cmma(f32x16& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void cmma<bf16_t>(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void cmma<f16_t>(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void cmma<float>(f32x16& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), acc, 0, 0, 0);
}
// ---- bf16x3: 8 f32 -> 8 bf16 hi + 8 bf16 lo (x ~ hi + lo to ~16 mantissa bits), and the 3-term product ----------------
// element 2j sits in the low half of word j (the order v_cvt_pk_bf16_f32 packs and the order pack.py writes split weights)
__device__ __forceinline__ void split8_bf16(const float* x, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned int h = f32x2_to_bf16x2(x[2 * j], x[2 * j + 1]);
        hi[j] = h;
        lo[j] = f32x2_to_bf16x2(x[2 * j] - __uint_as_float(h << 16), x[2 * j + 1] - __uint_as_float(h & 0xffff0000u));
    }
}
__device__ __forceinline__ void split8_bf16(const u32x4& c0, const u32x4& c1, u32x4& hi, u32x4& lo) {
    const float x[8] = {__uint_as_float(c0[0]), __uint_as_float(c0[1]), __uint_as_float(c0[2]), __uint_as_float(c0[3]),
                        __uint_as_float(c1[0]), __uint_as_float(c1[1]), __uint_as_float(c1[2]), __uint_as_float(c1[3])};
    split8_bf16(x, hi, lo);
}
// bf16x3 producers: store 4 (8) consecutive f32 K-elements of a row in the PRE-SPLIT operand format (per 8 K-elements: 8 x bf16 hi |
// 8 x bf16 lo, 32 bytes - what pack.py split_bf16 writes for weights and what conv_gemm's a_split consumes). `row_base` is the row's
// first byte, `cc` the index of the 4-element (16-byte) f32 chunk inside the row: half `cc & 1` of group `cc >> 1`.
__device__ __forceinline__ void store_split4(void* row_base, int cc, const float* e) {
    const unsigned int h0 = f32x2_to_bf16x2(e[0], e[1]), h1 = f32x2_to_bf16x2(e[2], e[3]);
    const unsigned int l0 = f32x2_to_bf16x2(e[0] - __uint_as_float(h0 << 16), e[1] - __uint_as_float(h0 & 0xffff0000u));
    const unsigned int l1 = f32x2_to_bf16x2(e[2] - __uint_as_float(h1 << 16), e[3] - __uint_as_float(h1 & 0xffff0000u));
    char* g = (char*)row_base + (cc >> 1) * 32 + (cc & 1) * 8;
    *(u32x2*)g = u32x2{h0, h1};
    *(u32x2*)(g + 16) = u32x2{l0, l1};
}
// the f16x2 producers' form (round 6: PLAIN f16 rows - the two-pass consumer multiplies only the activation's f16 value, so there is
// no lo half to keep: 2 bytes per element written, 64 contiguous bytes per K slab read). Values are clamped to the finite f16 range first
// (an overflow would reach the MFMA as inf); a NaN stays a NaN, so a diverged activation is still visible downstream.
__device__ __forceinline__ float clamp_f16_range(float e) {
    const float c = fminf(fmaxf(e, -65504.0f), 65504.0f);
    return e != e ? e : c;
}
__device__ __forceinline__ u32x2 pack4_f16_sat(const float* e) {
    return u32x2{f32x2_to_f16x2(clamp_f16_range(e[0]), clamp_f16_range(e[1])), f32x2_to_f16x2(clamp_f16_range(e[2]), clamp_f16_range(e[3]))};
}
// debug counter of the clamping stores (geo4d_conv_gemm_t.sat_count & co; NULL in production): += the lanes of this wave that hold a value
// beyond the finite f16 range (|e| > 65504, inf included; NaN is not a saturation). One atomic per wave that saw any. Wave-uniform call sites only.
__device__ __forceinline__ void count_f16_saturation(unsigned long long* counter, const float* e) {
    if (counter) {
        const bool any = fabsf(e[0]) > 65504.0f || fabsf(e[1]) > 65504.0f || fabsf(e[2]) > 65504.0f || fabsf(e[3]) > 65504.0f;
        const unsigned long long b = __ballot(any);
        if (b && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)b) - 1)) atomicAdd(counter, (unsigned long long)__popcll(b));
    }
}
// `row_base` = first byte of the f16 row, `cc` = index of the 4-element group inside the row
__device__ __forceinline__ void store4_f16(void* row_base, int cc, const float* e) {
    *(u32x2*)((char*)row_base + cc * 8) = pack4_f16_sat(e);
}
__device__ __forceinline__ void store_split8(void* row_base, int group8, const float* e) {
    u32x4 hi, lo;
    split8_bf16(e, hi, lo);
    char* g = (char*)row_base + group8 * 32;
    *(u32x4*)g = hi;
    *(u32x4*)(g + 16) = lo;
}

// acc += a.b with a = ah + al, b = bh + bl, dropping al.bl (2^-16 x 2^-16): three dense bf16 MFMAs, fp32 accumulate
__device__ __forceinline__ void mma_x3(f32x16& acc, const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, al), __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
}
// row index inside a 32x32 accumulator block for register r on half-wave hi
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---- misc ---------------------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 round-off class): 5 fma + v_rcp + v_exp instead of
// libm's branchy ~25-instruction erff — the GEGLU epilogue runs it on every element of the widest tensors of the U-Net.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return copysignf(fmaf(-p * t, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware bijective remap of a linear workgroup id (8 XCDs, block b lands on XCD b % 8):
// gives every XCD a contiguous range of logical tiles so neighbouring tiles share its private L2.
__device__ __forceinline__ long xcd_remap(long bid, long nwg) {
    const long q = nwg >> 3, r = nwg & 7;
    const long xcd = bid & 7, idx = bid >> 3;
    const long base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// host-side error codes (errno-style, negative)
#define GEO4D_OK 0
#define GEO4D_EINVAL (-22)
#define GEO4D_ENOTSUP (-95)
#define GEO4D_EIO (-5)

#define GEO4D_CHECK_LAUNCH()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) { geo4d_set_error(hipGetErrorString(e__)); return GEO4D_EIO; } \
    } while (0)

void geo4d_set_error(const char* msg);
