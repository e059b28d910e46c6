// fp16 instantiations of the decode GEMM: the dense formats with plain stores.  The quantised formats are in
// gemm_f16_q.hip, the write-through epilogues of both in gemm_f16_wt.hip: three units that compile side by side.
#include "gemm_impl.h"

void nv_gemm_launch_wt_f16(const GemmArgs& a);
void nv_gemm_launch_q_f16(const GemmArgs& a);
void nv_gemm_launch_f16(const GemmArgs& a) {
  if (gemm_wants_wt(a)) nv_gemm_launch_wt_f16(a);
  else if (gemm_is_quant(a)) nv_gemm_launch_q_f16(a);
  else gemm_dispatch<f16_t, false, true, false>(a);
}
void nv_gemm_reduce_f16(const GemmArgs& a) { reduce_dispatch<f16_t>(a); }
