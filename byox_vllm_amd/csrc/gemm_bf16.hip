// bf16 instantiations of the decode GEMM.
#include "gemm_impl.h"
#include "gemm_wide.h"

void nv_gemm_launch_bf16(const GemmArgs& a) { gemm_dispatch<bf16_t>(a); }
void nv_gemm_reduce_bf16(const GemmArgs& a) { reduce_dispatch<bf16_t>(a); }
void nv_gemm_wide_launch_bf16(const GemmArgs& a) { gemm_wide_dispatch<bf16_t>(a); }
