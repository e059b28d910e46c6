// fp16 instantiations of the decode GEMM with write-through epilogues (the result-store policy, common.h out_store).
#include "gemm_impl.h"

void nv_gemm_launch_wt_f16(const GemmArgs& a) { gemm_dispatch<f16_t, true>(a); }
