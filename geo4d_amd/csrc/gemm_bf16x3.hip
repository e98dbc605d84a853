// geo4d_amd/csrc/gemm_bf16x3.hip — conv_gemm kernels for element type bf16x3_t: f32 storage, 3-term bf16 split MFMA
// (one translation unit per type: parallel build).
#include "gemm_kernel.h"

namespace geo4d_gemm {
template int launch_typed<bf16x3_t>(const geo4d_conv_gemm_t&, hipStream_t);
}  // namespace geo4d_gemm
