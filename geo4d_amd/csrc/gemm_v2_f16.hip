// geo4d_amd/csrc/gemm_v2_f16.hip — second-generation conv_gemm kernels (tile hints 21..39) for element type f16_t
// (one translation unit per type: parallel build).
#include "gemm_kernel_v2.h"

namespace geo4d_gemm {
template int launch_v2_typed<f16_t>(const geo4d_conv_gemm_t&, hipStream_t);
}  // namespace geo4d_gemm
