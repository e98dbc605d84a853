// geo4d_amd/csrc/gemm_f16.hip — conv_gemm kernels for element type f16_t (one translation unit per type: parallel build).
#include "gemm_kernel.h"

namespace geo4d_gemm {
template int launch_typed<f16_t>(const geo4d_conv_gemm_t&, hipStream_t);
}  // namespace geo4d_gemm
