// fp16 instantiations of the decode GEMM: the quantised formats (int4, int8) with plain stores, and the 32x32x16 kernel.
#include "gemm_impl.h"
#include "gemm_wide.h"

void nv_gemm_launch_q_f16(const GemmArgs& a) { gemm_dispatch<f16_t, false, false, true>(a); }
void nv_gemm_wide_launch_f16(const GemmArgs& a) { gemm_wide_dispatch<f16_t>(a); }
