// geo4d_amd/csrc/gemm_v3_f16x2.hip — third-generation conv_gemm kernels (tile hints 71..74, phased K loop) for element type f16x2p_t:
// pre-split f16 hi | lo operands, two f16 MFMAs per product (round 5; one translation unit per type: parallel build).
#include "gemm_kernel_v3.h"

namespace geo4d_gemm {
template int launch_v3_typed<f16x2p_t>(const geo4d_conv_gemm_t&, hipStream_t);
template int colsum_rows_v23<f16x2p_t>(const geo4d_conv_gemm_t&);
}  // namespace geo4d_gemm
