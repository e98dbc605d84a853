// Standalone probe (not part of the library): how fast can ONE workgroup per CU fill its LDS ring with the operand slabs of the bf16x3
// conv GEMM (160x320 tile: 480 rows x 128 B = 60 KB per K slab), and what does the rate depend on?
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/fill_probe.hip -o gpurun_bin/fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 160, BN = 320, PITCH = 128, NT = 512, RSTEP = NT / 8;
constexpr int ACH = (BM + RSTEP - 1) / RSTEP, BCH = BN / RSTEP;   // 3 (ragged), 5

__device__ __forceinline__ int key16(int row) { const int p = (row >> 1) & 7; return p ^ ((((p >> 2) ^ (p >> 1)) & 1) << 1); }

// MODE 0: LDS-DMA, 16-byte XOR swizzle on the source (the product's layout)      1: LDS-DMA, linear source
//      2: LDS-DMA, 32-byte (chunk pair) XOR swizzle                               3: as 0 with aux = nt
//      4: global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging)         5: global_load_dwordx4 -> VGPR only (no LDS)
//      6: as 0, two slabs in flight (counted vmcnt, rate without the latency of a lone slab)
//      7: as 0, A panel only (160 rows)                                           8: as 0, B panel only (320 rows)
//      9: as 0 but every lane of an 8-lane row group reads the SAME 128-byte row in natural order and the swizzle is applied by
//         permuting which LDS row-slot... (not possible with DMA: kept as linear + row rotation) -> alias of 1 with rows rotated
template <int MODE>
__global__ __launch_bounds__(NT) void fill(const float* __restrict__ A, const float* __restrict__ W, int M, int Cin, int nslab, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ccol = tid & 7, r0 = tid >> 3;
    const int tm = blockIdx.x;
    const int K = 9 * Cin;
    u32x4 acc = {0, 0, 0, 0};
    int c0 = 0, tap = 0;
    auto chunk = [&](int row) {
        if (MODE == 1 || MODE == 4 || MODE == 5) return ccol;
        if (MODE == 2) return (((ccol >> 1) ^ ((row >> 1) & 3)) << 1) | (ccol & 1);
        return ccol ^ key16(row);
    };
    auto issue = [&](int buf) {
        char* base = smem + buf * (BM + BN) * PITCH + wave * 1024;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            if (MODE == 8) break;
            if (wave * 8 + j * RSTEP < BM) {
                const int row = r0 + j * RSTEP;
                long m = (long)tm * BM + row + dy * 64 + dx;
                m = m < 0 ? 0 : (m >= M ? M - 1 : m);
                const float* src = A + m * Cin + c0 + chunk(row) * 4;
                if (MODE == 4 || MODE == 5) {
                    const u32x4 v = *(const u32x4*)src;
                    if (MODE == 4) *(u32x4*)(base + j * RSTEP * PITCH + (lane >> 3) * PITCH + ((ccol ^ key16(row)) << 4)) = v;
                    else acc ^= v;
                } else if (MODE == 3) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(base + j * RSTEP * PITCH), 16, 0, 2);
                } else {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(base + j * RSTEP * PITCH), 16, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            if (MODE == 7) break;
            const int row = r0 + i * RSTEP;
            const float* src = W + (long)row * K + tap * Cin + c0 + chunk(row) * 4;
            if (MODE == 4 || MODE == 5) {
                const u32x4 v = *(const u32x4*)src;
                if (MODE == 4) *(u32x4*)(base + (BM + i * RSTEP) * PITCH + (lane >> 3) * PITCH + ((ccol ^ key16(row)) << 4)) = v;
                else acc ^= v;
            } else if (MODE == 3) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + (BM + i * RSTEP) * PITCH), 16, 0, 2);
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + (BM + i * RSTEP) * PITCH), 16, 0, 0);
            }
        }
        if (++tap == 9) { tap = 0; c0 += 32; }
    };
    if (MODE == 6) {
        issue(0);
        for (int s = 1; s < nslab; ++s) {
            issue(s & 1);
            // this wave's pieces of the previous slab have landed when at most the pieces of the newest slab are outstanding
            if (wave < 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int s = 0; s < nslab; ++s) {
            issue(s & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (MODE == 5) { if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[tid] = 1; }
    else if (tid == 0 && nslab < 0) sink[0] = *(unsigned*)smem;
}

template <int MODE>
void run(const char* name, const float* A, const float* W, int M, int Cin, int nslab, unsigned* sink, int bytes_per_slab) {
    const int smem = 2 * (BM + BN) * PITCH;
    hipFuncSetAttribute((const void*)fill<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = M / BM;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fill<MODE>, dim3(grid), dim3(NT), smem, 0, A, W, M, Cin, nslab, sink);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fill<MODE>, dim3(grid), dim3(NT), smem, 0, A, W, M, Cin, nslab, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-74s %8.1f us  %6.0f ns/slab  %6.1f GB/s per CU  %6.2f TB/s chip\n", name, us, us * 1e3 / nslab, bytes_per_slab / (us * 1e3 / nslab),
           (double)bytes_per_slab * nslab * grid / us / 1e6);
}

int main() {
    const int M = 40960, Cin = 320, nslab = 90;
    float *A, *W;
    unsigned* sink;
    hipMalloc(&A, (size_t)M * Cin * 4);
    hipMalloc(&W, (size_t)BN * 9 * Cin * 4);
    hipMalloc(&sink, 4096);
    std::vector<float> h((size_t)M * Cin);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)BN * 9 * Cin * 4, hipMemcpyHostToDevice);
    printf("fill of the 160x320 bf16x3 tile's K slabs (L0 conv3x3 320->320: 256 workgroups, one per CU, 90 slabs of 60 KB, 8 waves)\n");
    const int full = (BM + BN) * PITCH;
    run<0>("0 LDS-DMA, 16-byte XOR swizzled source (product)", A, W, M, Cin, nslab, sink, full);
    run<1>("1 LDS-DMA, linear source (bank-conflicted fragment reads)", A, W, M, Cin, nslab, sink, full);
    run<2>("2 LDS-DMA, 32-byte pair swizzle", A, W, M, Cin, nslab, sink, full);
    run<3>("3 LDS-DMA, 16-byte swizzle, aux = nt", A, W, M, Cin, nslab, sink, full);
    run<4>("4 global_load_dwordx4 -> ds_write_b128 (swizzled destination)", A, W, M, Cin, nslab, sink, full);
    run<5>("5 global_load_dwordx4 -> VGPR only", A, W, M, Cin, nslab, sink, full);
    run<6>("6 LDS-DMA 16-byte swizzle, two slabs in flight (counted vmcnt)", A, W, M, Cin, nslab, sink, full);
    run<7>("7 LDS-DMA, A panel only (160 gathered rows)", A, W, M, Cin, nslab, sink, BM * PITCH);
    run<8>("8 LDS-DMA, B panel only (320 weight rows, same for every CU)", A, W, M, Cin, nslab, sink, BN * PITCH);
    return 0;
}
