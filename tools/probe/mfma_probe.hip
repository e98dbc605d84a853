// Standalone probe of the gfx950 matrix pipe under the issue patterns a GEMM K loop can produce (not part of the library).
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/probe/mfma_probe.hip -o gpurun_bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: one accumulator, every MFMA depends on the previous        MODE 1: 4 accumulators round-robin (independent neighbours)
// MODE 2: 4 accumulators, chains of 3 on each (the bf16x3 pattern)    MODE 3: 4 accumulators, chains of 6
// MODE 4: 16x16x32, 16 accumulators round-robin                       MODE 5: 16x16x32, chains of 4 on each of 16 accumulators
template <int MODE, bool BAR>
__global__ __launch_bounds__(512) void probe(float* out, int iters, int nwaves_active, const unsigned* src) {
    extern __shared__ float lds[];                      // sized by the launch so that ONE workgroup fits a CU (occupancy = nwaves_active / 4 per SIMD)
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) lds[0] = 0.f;
    if (wave >= nwaves_active) return;
    bf16x8 a, b;
    if (src) {                                          // operands from memory: uniform random bf16 in [-1, 1) (data-dependent MFMA rate)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        a = __builtin_bit_cast(bf16x8, *(const u32x4*)(src + (threadIdx.x & 511) * 4));
        b = __builtin_bit_cast(bf16x8, *(const u32x4*)(src + 2048 + (threadIdx.x & 511) * 4));
    } else {
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    }
    float r = 0.f;
    if constexpr (MODE <= 3) {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 24; ++k) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
            } else if constexpr (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 24; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k & 3], 0, 0, 0);
            } else if constexpr (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 24; ++k) acc[(k / 3) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(k / 3) & 3], 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 24; ++k) acc[(k / 6) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(k / 6) & 3], 0, 0, 0);
            }
            if constexpr (BAR) __builtin_amdgcn_s_barrier();
        }
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) r += acc[j][i];
    } else {
        f32x4 acc[16];
        for (int j = 0; j < 16; ++j) for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 4) {
#pragma unroll
                for (int k = 0; k < 48; ++k) acc[k & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 15], 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 48; ++k) acc[(k / 4) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[(k / 4) & 15], 0, 0, 0);
            }
            if constexpr (BAR) __builtin_amdgcn_s_barrier();
        }
        for (int j = 0; j < 16; ++j) for (int i = 0; i < 4; ++i) r += acc[j][i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE, bool BAR>
void run(const char* name, float* out, int waves, const unsigned* src = nullptr) {
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // BAR kernels run all 8 waves (a barrier with exited waves is fine on AMD, but keep it simple)
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, BAR>), dim3(blocks), dim3(waves * 64), 100 * 1024, 0, out, iters, waves, src);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_wave_flops = (MODE <= 3 ? 24.0 * 32768 : 48.0 * 16384) * iters;
    const double tf = per_wave_flops * waves * blocks / (ms * 1e-3) / 1e12;
    printf("%-58s waves/CU %d  %8.3f ms  %7.0f TF/s  (%4.1f %% of 2500)\n", name, waves, ms, tf, tf / 25.0);
}

int main() {
    hipFuncSetAttribute((const void*)probe<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    float* out;
    hipMalloc(&out, 256 * 4 * 512 * sizeof(float));
    for (int waves : {4, 8}) {                           // one workgroup per CU (100 KB of LDS each): 1 or 2 waves per SIMD
        run<0, false>("32x32x16 one accumulator (all dependent)", out, waves);
        run<1, false>("32x32x16 4 accumulators round-robin", out, waves);
        run<2, false>("32x32x16 4 accumulators, chains of 3", out, waves);
        run<3, false>("32x32x16 4 accumulators, chains of 6", out, waves);
        run<4, false>("16x16x32 16 accumulators round-robin", out, waves);
        run<5, false>("16x16x32 16 accumulators, chains of 4", out, waves);
    }
    unsigned* src;
    hipMalloc(&src, 4096 * 4);
    {
        static unsigned h[4096];
        unsigned st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (st >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f; };
        for (int i = 0; i < 4096; ++i) {
            float x = rnd(), y = rnd();
            unsigned ux, uy;
            __builtin_memcpy(&ux, &x, 4); __builtin_memcpy(&uy, &y, 4);
            h[i] = (ux >> 16) | (uy & 0xffff0000u);
        }
        hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    }
    printf("-- uniform random bf16 operands in [-1, 1), 8 waves per CU\n");
    run<1, false>("32x32x16 4 accumulators round-robin, RANDOM operands", out, 8, src);
    run<3, false>("32x32x16 4 accumulators, chains of 6, RANDOM operands", out, 8, src);
    run<4, false>("16x16x32 16 accumulators round-robin, RANDOM operands", out, 8, src);
    run<5, false>("16x16x32 16 accumulators, chains of 4, RANDOM operands", out, 8, src);
    run<4, false>("16x16x32 16 accumulators round-robin, RANDOM operands, 4 waves per CU", out, 4, src);
    printf("-- constant operands again\n");
    run<2, true>("32x32x16 chains of 3 + s_barrier every 24 MFMAs", out, 8);
    run<1, true>("32x32x16 round-robin + s_barrier every 24 MFMAs", out, 8);
    run<4, true>("16x16x32 round-robin + s_barrier every 48 MFMAs", out, 8);
    return 0;
}
