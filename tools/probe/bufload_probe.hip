// tools/probe/bufload_probe.hip — what `buffer_load_dwordx4 ... lds` (raw buffer resource, LDS destination) does with out-of-range
// lanes on gfx950: the conv_gemm gather wants zeros written for padding taps. Build: hipcc --offload-arch=gfx950 -O2 -o bufload_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* a, float* o, unsigned num_records, int soff, unsigned oob_voff) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s = (float*)smem;
    for (int i = threadIdx.x; i < 64 * 4; i += 64) s[i] = -7.f;          // stale LDS content
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, num_records, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if ((threadIdx.x & 7) == 5) voff = oob_voff;                           // "padding" lanes
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, (int)voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4; i += 64) o[i] = s[i];
}

int main() {
    const int n = 1 << 20;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.f + i;
    float *a, *o;
    hipMalloc(&a, n * 4);
    hipMalloc(&o, 256 * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    struct { unsigned nr; int soff; unsigned oob; const char* what; } cases[] = {
        {0x80000000u, 0, 0x80000000u, "num_records 2^31, voffset 2^31, soffset 0"},
        {0x80000000u, 4096, 0x80000000u, "num_records 2^31, voffset 2^31, soffset 4096"},
        {0x80000000u, 4096, 0xFFFFF000u, "num_records 2^31, voffset 2^32-4096, soffset 4096 (sum wraps to 0)"},
        {(unsigned)n * 4, 4096, 0x80000000u, "num_records = size, voffset 2^31, soffset 4096"},
    };
    for (auto& c : cases) {
        hipMemset(o, 0xff, 256 * 4);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 1024, 0, a, o, c.nr, c.soff, c.oob);
        std::vector<float> r(256);
        hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        // lane l wrote LDS floats 4l..4l+3: valid lanes expect a[(16 l + soff) / 4 ...]
        int ok_valid = 0, zero_pad = 0, stale_pad = 0, other_pad = 0;
        for (int l = 0; l < 64; ++l) {
            const float exp = 1.f + (16 * l + c.soff) / 4;
            if ((l & 7) == 5) {
                if (r[4 * l] == 0.f && r[4 * l + 3] == 0.f) ++zero_pad;
                else if (r[4 * l] == -7.f) ++stale_pad;
                else ++other_pad;
            } else if (r[4 * l] == exp && r[4 * l + 3] == exp + 3) ++ok_valid;
        }
        printf("%-70s valid lanes ok %d/56 | padding lanes: zeros %d, stale %d, other %d (first %g)\n", c.what, ok_valid, zero_pad, stale_pad, other_pad, r[20]);
    }
    return 0;
}
