// geo4d_amd/csrc/gemm_v3_bf16x3.hip — third-generation conv_gemm kernels (tile hints 71..74, phased K loop) for element type bf16x3_t
// (one translation unit per type: parallel build).
#include "gemm_kernel_v3.h"

namespace geo4d_gemm {
template int launch_v3_typed<bf16x3_t>(const geo4d_conv_gemm_t&, hipStream_t);
template int colsum_rows_v23<bf16x3_t>(const geo4d_conv_gemm_t&);
}  // namespace geo4d_gemm
