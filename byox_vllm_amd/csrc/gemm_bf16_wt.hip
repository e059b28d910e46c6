// bf16 instantiations of the decode GEMM with write-through epilogues (the result-store policy, common.h out_store).
#include "gemm_impl.h"

void nv_gemm_launch_wt_bf16(const GemmArgs& a) { gemm_dispatch<bf16_t, true>(a); }
