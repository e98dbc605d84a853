// Probe 2: start from the MFMA-only K loop of conv_gemm (bf16x3, 256x128 tile: 8 waves, 24 MFMAs per slab in chains of 3 over 4
// accumulators, one s_barrier per slab) and add, one at a time, what the real loop has around it. One workgroup per CU (100 KB LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// FLAGS: 1 distinct operand registers (8 A + 8 B fragments)   2 four ds_read_b32 per slab + lgkmcnt(0) before the barrier
//        4 ~230 live VGPRs (a second fragment set kept alive) 8 s_waitcnt vmcnt(0) before the barrier
//        16 fragments re-read from LDS every slab (16 ds_read_b128, register prefetch one slab ahead)
//        128 pin the chains of 3 (sched_barrier after each triple: hipcc otherwise interleaves the four accumulators)
//        32 allocate all 256 VGPRs (clobber v255)    64 s_setprio 1 on the younger half of the workgroup (waves 4-7)
template <int FLAGS>
__global__ __launch_bounds__(512) void probe(float* out, const unsigned* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 24 * 1024; i += 512) ((unsigned*)lds)[i] = src[i & 4095];
    __syncthreads();
    u32x4 A[2][2][2], B[2][2][2], A2[2][2][2], B2[2][2][2];
    for (int s = 0; s < 2; ++s) for (int a = 0; a < 2; ++a) for (int j = 0; j < 2; ++j) {
        A[s][a][j] = *(const u32x4*)(lds + ((s * 4 + a * 2 + j) * 512 + tid) * 16 % 65536);
        B[s][a][j] = *(const u32x4*)(lds + ((8 + s * 4 + a * 2 + j) * 512 + tid) * 16 % 65536);
        A2[s][a][j] = A[s][a][j]; B2[s][a][j] = B[s][a][j];
    }
    if constexpr (FLAGS & 32) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if constexpr (FLAGS & 64) { if (tid >= 256) __builtin_amdgcn_s_setprio(1); }
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    int pix[4] = {0, 0, 0, 0};
    const int* rowpix = (const int*)(lds + 80 * 1024);
    auto mfma = [&](f32x16& c, const u32x4& x, const u32x4& y) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
    };
    auto slab = [&](u32x4 (&FA)[2][2][2], u32x4 (&FB)[2][2][2], u32x4 (&NA)[2][2][2], u32x4 (&NB)[2][2][2], int it) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FLAGS & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (FLAGS & (2 | 16)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (FLAGS & 16) {
            const int buf = (it & 1) * 40 * 1024;
            for (int s = 0; s < 2; ++s) for (int a = 0; a < 2; ++a) for (int j = 0; j < 2; ++j) {
                NA[s][a][j] = *(const u32x4*)(lds + buf + ((s * 4 + a * 2 + j) * 64 + (tid & 63)) * 16);
                NB[s][a][j] = *(const u32x4*)(lds + buf + 8192 + ((s * 4 + a * 2 + j) * 64 + (tid & 63)) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if constexpr (FLAGS & 1) {
                        mfma(acc[a][b], FB[s][b][0], FA[s][a][1]);
                        mfma(acc[a][b], FB[s][b][1], FA[s][a][0]);
                        mfma(acc[a][b], FB[s][b][0], FA[s][a][0]);
                        if constexpr (FLAGS & 128) __builtin_amdgcn_sched_barrier(0);    // keep the three dependent MFMAs together
                    } else {
                        mfma(acc[a][b], FB[0][0][0], FA[0][0][0]);
                        mfma(acc[a][b], FB[0][0][0], FA[0][0][0]);
                        mfma(acc[a][b], FB[0][0][0], FA[0][0][0]);
                    }
                }
        if constexpr (FLAGS & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pix[i] = rowpix[(tid >> 3) + i * 64 + (it & 7)];
        }
    };
    for (int it = 0; it < iters; it += 2) {
        if constexpr (FLAGS & (4 | 16)) { slab(A, B, A2, B2, it); slab(A2, B2, A, B, it + 1); }
        else { slab(A, B, A, B, it); slab(A, B, A, B, it + 1); }
    }
    float r = (float)(pix[0] + pix[1] + pix[2] + pix[3]);
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) r += acc[a][b][i];
    for (int s = 0; s < 2; ++s) for (int a = 0; a < 2; ++a) for (int j = 0; j < 2; ++j) r += (float)(A2[s][a][j][0] + B2[s][a][j][1]);
    out[blockIdx.x * 512 + tid] = r;
}

template <int FLAGS>
void run(const char* name, float* out, const unsigned* src) {
    const int iters = 2000, blocks = 256 * 4;
    hipFuncSetAttribute((const void*)probe<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<FLAGS>), dim3(blocks), dim3(512), 100 * 1024, 0, out, src, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = 24.0 * 32768 * iters * 8 * blocks / (ms * 1e-3) / 1e12;
    printf("%-84s %8.3f ms  %6.0f TF/s (%4.1f %%)\n", name, ms, tf, tf / 25.0);
}

int main() {
    float* out; unsigned* src;
    hipMalloc(&out, 256 * 4 * 512 * sizeof(float));
    hipMalloc(&src, 4096 * 4);
    hipMemset(src, 0x3c, 4096 * 4);
    auto fill_random = [&]() {                       // uniform random bf16 pairs in [-1, 1): operand toggling as in a real GEMM
        static unsigned h[4096];
        unsigned st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (st >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f; };
        for (int i = 0; i < 4096; ++i) {
            float a = rnd(), b = rnd();
            unsigned ua, ub;
            __builtin_memcpy(&ua, &a, 4); __builtin_memcpy(&ub, &b, 4);
            h[i] = (ua >> 16) | (ub & 0xffff0000u);
        }
        hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    };
    run<0>("MFMA chains of 3 + barrier per 24 (same operand registers)", out, src);
    run<1>("+ distinct operand registers (8 A + 8 B fragments)", out, src);
    run<1 | 2>("+ 4 ds_read_b32 per slab, lgkmcnt(0) before the barrier", out, src);
    run<1 | 2 | 8>("+ s_waitcnt vmcnt(0) before the barrier", out, src);
    run<1 | 2 | 4 | 8>("+ second fragment set alive (~2x fragment registers), sets alternate", out, src);
    run<1 | 2 | 8 | 16>("+ 16 ds_read_b128 per slab, prefetched one slab ahead (the real LDS read traffic)", out, src);
    run<1 | 2 | 8 | 128>("MFMA + waits + barrier, chains of 3 PINNED (dependent neighbours)", out, src);
    run<1 | 2 | 8 | 16 | 128>("... with the 16 ds_read_b128 per slab, chains of 3 pinned", out, src);
    fill_random();
    printf("-- operands: uniform random bf16 in [-1, 1) from here on\n");
    run<1 | 2 | 8 | 128>("MFMA + waits + barrier, chains of 3 pinned, RANDOM operands", out, src);
    run<1 | 2 | 8>("MFMA + waits + barrier, compiler order (round-robin), RANDOM operands", out, src);
    run<1 | 2 | 8 | 16 | 128>("... with the 16 ds_read_b128 per slab, RANDOM operands", out, src);
    run<1 | 2 | 8 | 32>("MFMA + waits + barrier, 256 VGPRs allocated per wave", out, src);
    run<1 | 2 | 8 | 16 | 32>("... with the 16 ds_read_b128 per slab, 256 VGPRs allocated", out, src);
    run<1 | 2 | 8 | 16 | 32 | 64>("... and s_setprio 1 on waves 4-7", out, src);
    return 0;
}
