// geo4d_amd/csrc/gemm_v2_f32.hip — second-generation conv_gemm kernels (tile hints 21..39) for element type float
// (one translation unit per type: parallel build) - the exact-f32 mode has none: this instantiates the refusal.
#include "gemm_kernel_v2.h"

namespace geo4d_gemm {
template int launch_v2_typed<float>(const geo4d_conv_gemm_t&, hipStream_t);
}  // namespace geo4d_gemm
