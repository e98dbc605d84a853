// geo4d_amd/csrc/gemm_v2_f16x2.hip — second-generation conv_gemm kernels (tile hints 22..28) for element type f16x2p_t: pre-split
// f16 hi | lo operands, two f16 MFMAs per product (round 5; one translation unit per type: parallel build).
#include "gemm_kernel_v2.h"

namespace geo4d_gemm {
template int launch_v2_typed<f16x2p_t>(const geo4d_conv_gemm_t&, hipStream_t);
}  // namespace geo4d_gemm
