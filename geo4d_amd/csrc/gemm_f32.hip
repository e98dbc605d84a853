// geo4d_amd/csrc/gemm_f32.hip — conv_gemm kernels for element type float (one translation unit per type: parallel build).
#include "gemm_kernel.h"

namespace geo4d_gemm {
template int launch_typed<float>(const geo4d_conv_gemm_t&, hipStream_t);
}  // namespace geo4d_gemm
