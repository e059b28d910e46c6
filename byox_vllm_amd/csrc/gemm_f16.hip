// fp16 instantiations of the decode GEMM.
#include "gemm_impl.h"
#include "gemm_wide.h"

void nv_gemm_launch_f16(const GemmArgs& a) { gemm_dispatch<f16_t>(a); }
void nv_gemm_reduce_f16(const GemmArgs& a) { reduce_dispatch<f16_t>(a); }
void nv_gemm_wide_launch_f16(const GemmArgs& a) { gemm_wide_dispatch<f16_t>(a); }
