// Many-row GEMMs (prefill) through the vendor library.
//
// nv_gemm's kernels are built for the decode step: up to a few hundred rows of x against weights streamed once.  A
// prefill step multiplies thousands of rows by the same weights, which is a plain dense GEMM once the weights are in
// the model dtype: hipBLASLt runs those shapes at 1.1-1.45 PFLOP/s on this part where nv_gemm's tiles reach 0.65-0.95.
// So for many rows the weight is first written out row-major in the model dtype (nv_weight_expand: the tiled dense
// layout un-tiled, int4 / int8 dequantised exactly as the oracle does, (u - 8) * s in fp32 rounded once) into a scratch
// buffer the caller owns, and the product is one hipblasLtMatmul (nv_gemm_lt).  The expansion moves N*K*2 bytes and
// is a few per cent of the product from about two thousand rows on.
//
// hipBLASLt is loaded with dlopen on first use, not linked: the library loads, and everything else in it works, on a
// machine without it; nv_gemm_lt then returns NV_ERR_INVALID with the loader's message and nv_gemm_lt_available()
// says 0.  No allocation happens here: scratch and the library's workspace are arguments.
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

#include "common.h"

namespace {

struct LtApi {
  bool ok = false;
  char why[256] = {0};
  decltype(&hipblasLtCreate) Create = nullptr;
  decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
  decltype(&hipblasLtMatmulDescDestroy) DescDestroy = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
  decltype(&hipblasLtMatrixLayoutDestroy) LayoutDestroy = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
  decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
  decltype(&hipblasLtMatmul) Matmul = nullptr;
};

template <typename F>
bool bind(void* so, const char* name, F& fn, LtApi& api) {
  fn = reinterpret_cast<F>(dlsym(so, name));
  if (!fn) snprintf(api.why, sizeof api.why, "hipBLASLt lacks %s", name);
  return fn != nullptr;
}

LtApi& lt_api() {
  static LtApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* so = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) so = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
    if (!so) {
      snprintf(api.why, sizeof api.why, "hipBLASLt not loadable: %s", dlerror());
      return;
    }
    api.ok = bind(so, "hipblasLtCreate", api.Create, api) && bind(so, "hipblasLtMatmulDescCreate", api.DescCreate, api) &&
             bind(so, "hipblasLtMatmulDescSetAttribute", api.DescSet, api) &&
             bind(so, "hipblasLtMatmulDescDestroy", api.DescDestroy, api) &&
             bind(so, "hipblasLtMatrixLayoutCreate", api.LayoutCreate, api) &&
             bind(so, "hipblasLtMatrixLayoutDestroy", api.LayoutDestroy, api) &&
             bind(so, "hipblasLtMatmulPreferenceCreate", api.PrefCreate, api) &&
             bind(so, "hipblasLtMatmulPreferenceSetAttribute", api.PrefSet, api) &&
             bind(so, "hipblasLtMatmulPreferenceDestroy", api.PrefDestroy, api) &&
             bind(so, "hipblasLtMatmulAlgoGetHeuristic", api.Heuristic, api) && bind(so, "hipblasLtMatmul", api.Matmul, api);
  });
  return api;
}

// One product shape on one device: descriptors and the algorithm the library's heuristic chose for it.  A model has four
// shapes per row count and a serving process sees many row counts: the table is emptied when it reaches NV_LT_MAX_PLANS
// (descriptors are host objects, read when a product is launched, so none is in use between calls).
#define NV_LT_MAX_PLANS 512
struct LtPlan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t workspace = 0;
};
typedef std::tuple<int, int, int, int, int, int, int64_t, int64_t, size_t> LtKey;   // device, M, N, K, dtype, bias, strides, workspace

std::mutex g_lt_mutex;
std::map<int, hipblasLtHandle_t> g_lt_handles;
std::map<LtKey, LtPlan> g_lt_plans;

void destroy_plan(LtApi& api, LtPlan& p) {
  if (p.desc) api.DescDestroy(p.desc);
  if (p.a) api.LayoutDestroy(p.a);
  if (p.b) api.LayoutDestroy(p.b);
  if (p.c) api.LayoutDestroy(p.c);
  p = LtPlan();
}

}  // namespace

extern "C" int nv_gemm_lt_available(void) { return lt_api().ok ? 1 : 0; }

extern "C" int nv_gemm_lt(const void* x, int64_t x_row_stride, const void* w_rowmajor, const void* bias, void* out,
                          int64_t out_row_stride, int M, int N, int K, int dtype, void* workspace,
                          size_t workspace_bytes, nv_stream_t stream) {
  NV_CHECK_ARG(x && w_rowmajor && out, "nv_gemm_lt: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_gemm_lt: bad dtype %d", dtype);
  NV_CHECK_ARG(M > 0 && N > 0 && K > 0 && x_row_stride >= K && out_row_stride >= N, "nv_gemm_lt: M=%d N=%d K=%d strides %lld %lld",
               M, N, K, (long long)x_row_stride, (long long)out_row_stride);
  NV_CHECK_ARG(((uintptr_t)x | (uintptr_t)w_rowmajor | (uintptr_t)out | (uintptr_t)workspace) % 16 == 0 && K % 8 == 0 &&
                   N % 8 == 0 && x_row_stride % 8 == 0 && out_row_stride % 8 == 0,
               "nv_gemm_lt: operands must be 16-byte aligned (K, N and the row strides multiples of 8)");
  LtApi& api = lt_api();
  NV_CHECK_ARG(api.ok, "nv_gemm_lt: %s", api.why);
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) {
    nv_set_error("nv_gemm_lt: no current device");
    return NV_ERR_LAUNCH;
  }
  std::lock_guard<std::mutex> lock(g_lt_mutex);
  hipblasLtHandle_t& handle = g_lt_handles[device];
  if (!handle && api.Create(&handle) != HIPBLAS_STATUS_SUCCESS) {
    handle = nullptr;
    nv_set_error("nv_gemm_lt: hipblasLtCreate failed");
    return NV_ERR_LAUNCH;
  }
  const LtKey key(device, M, N, K, dtype, bias ? 1 : 0, x_row_stride, out_row_stride, workspace_bytes);
  auto it = g_lt_plans.find(key);
  if (it == g_lt_plans.end()) {
    // Row-major out[M, N] = x[M, K] * W[N, K]^T is, read column-major, D (N x M) = W^T-as-stored (K x N, transposed)
    // times x-as-stored (K x M): the library's A is the weight, its B the activations.
    LtPlan p;
    const hipDataType t = dtype == NV_BF16 ? HIP_R_16BF : HIP_R_16F;
    const hipblasOperation_t op_t = HIPBLAS_OP_T, op_n = HIPBLAS_OP_N;
    bool ok = api.DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op_t, sizeof op_t) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op_n, sizeof op_n) == HIPBLAS_STATUS_SUCCESS;
    if (ok && bias) {
      const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
      const int32_t bias_type = (int32_t)t;
      ok = api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof epi) == HIPBLAS_STATUS_SUCCESS &&
           api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bias_type, sizeof bias_type) == HIPBLAS_STATUS_SUCCESS;
    }
    ok = ok && api.LayoutCreate(&p.a, t, (uint64_t)K, (uint64_t)N, (int64_t)K) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && api.LayoutCreate(&p.b, t, (uint64_t)K, (uint64_t)M, x_row_stride) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && api.LayoutCreate(&p.c, t, (uint64_t)N, (uint64_t)M, out_row_stride) == HIPBLAS_STATUS_SUCCESS;
    if (!ok) {
      destroy_plan(api, p);
      nv_set_error("nv_gemm_lt: descriptor set-up failed (M=%d N=%d K=%d)", M, N, K);
      return NV_ERR_LAUNCH;
    }
    if (bias) {   // the heuristic of some versions inspects the pointer: give it this call's
      api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias);
    }
    hipblasLtMatmulPreference_t pref = nullptr;
    constexpr int ASK = 8;
    hipblasLtMatmulHeuristicResult_t found[ASK];
    int n_found = 0;
    const uint64_t limit = workspace ? workspace_bytes : 0;
    ok = api.PrefCreate(&pref) == HIPBLAS_STATUS_SUCCESS &&
         api.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &limit, sizeof limit) == HIPBLAS_STATUS_SUCCESS &&
         api.Heuristic(handle, p.desc, p.a, p.b, p.c, p.c, pref, ASK, found, &n_found) == HIPBLAS_STATUS_SUCCESS && n_found > 0;
    if (pref) api.PrefDestroy(pref);
    // the library's ranking, best first; some versions list algorithms that want more workspace than the limit says
    int pick = -1;
    for (int i = 0; ok && i < n_found && pick < 0; ++i)
      if (found[i].state == HIPBLAS_STATUS_SUCCESS && found[i].workspaceSize <= limit) pick = i;
    if (!ok || pick < 0) {
      destroy_plan(api, p);
      nv_set_error("nv_gemm_lt: the library offers no algorithm for M=%d N=%d K=%d within %llu bytes of workspace", M, N, K,
                   (unsigned long long)limit);
      return NV_ERR_INVALID;
    }
    p.algo = found[pick].algo;
    p.workspace = found[pick].workspaceSize;
    if (g_lt_plans.size() >= NV_LT_MAX_PLANS) {
      for (auto& kv : g_lt_plans) destroy_plan(api, kv.second);
      g_lt_plans.clear();
    }
    it = g_lt_plans.emplace(key, p).first;
  }
  LtPlan& p = it->second;
  NV_CHECK_ARG(p.workspace <= workspace_bytes && (p.workspace == 0 || workspace), "nv_gemm_lt: workspace of %zu bytes needed",
               p.workspace);
  if (bias && api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias) != HIPBLAS_STATUS_SUCCESS) {
    nv_set_error("nv_gemm_lt: cannot set the bias pointer");
    return NV_ERR_LAUNCH;
  }
  const float one = 1.f, zero = 0.f;
  const hipblasStatus_t st = api.Matmul(handle, p.desc, &one, w_rowmajor, p.a, x, p.b, &zero, out, p.c, out, p.c, &p.algo,
                                        workspace, p.workspace, (hipStream_t)stream);
  if (st != HIPBLAS_STATUS_SUCCESS) {
    nv_set_error("nv_gemm_lt: hipblasLtMatmul failed with status %d (M=%d N=%d K=%d)", (int)st, M, N, K);
    return NV_ERR_LAUNCH;
  }
  return NV_OK;
}

// ---------------------------------------------------------------- weights of any nv_gemm format -> row-major
// One thread per 8 consecutive input columns of one output row, in the order the tiled formats store them (tile of
// 16 rows, block of 128 columns, lane (c, g), k-step ks: row 16 nt + c, columns 128 kb + 32 ks + 8 g ..+7), so the
// reads are contiguous and every thread writes one 16-byte piece.  GATE_UP: the rows are stored as nv_gemm's
// silu(gate)*up epilogue wants them, (gate 2i, gate 2i+1, up 2i, up 2i+1), and are written back as [gate rows; up rows].
template <typename T, int FMT, bool GATE_UP>
__global__ __launch_bounds__(256) void weight_expand_kernel(const void* __restrict__ data, const uint16_t* __restrict__ scales,
                                                            uint16_t* __restrict__ out, int N, int K) {
  typedef typename VecOf<T>::x8 vec_t;
  const int nkb = K >> 7;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)(N >> 4) * nkb * 256) return;
  int ks, lane;
  if constexpr (FMT == NV_W_DENSE_TILED) {   // [nt][kb][ks][lane][8]
    lane = (int)(idx & 63);
    ks = (int)((idx >> 6) & 3);
  } else {                                   // [nt][kb][lane][ks ...]
    ks = (int)(idx & 3);
    lane = (int)((idx >> 2) & 63);
  }
  const int64_t tile_block = idx >> 8;
  const int kb = (int)(tile_block % nkb), nt = (int)(tile_block / nkb);
  const int c = lane & 15, g = lane >> 4;
  const int row = 16 * nt + c, col = 128 * kb + 32 * ks + 8 * g;
  vec_t v;
  if constexpr (FMT == NV_W_DENSE_TILED) {
    v = reinterpret_cast<const vec_t*>(data)[idx];
  } else if constexpr (FMT == NV_W_DENSE) {
    v = *reinterpret_cast<const vec_t*>(reinterpret_cast<const uint16_t*>(data) + (int64_t)row * K + col);
  } else {
    const float s = (float)__builtin_bit_cast(T, scales[(int64_t)kb * N + row]);
    if constexpr (FMT == NV_W_INT4) {
      const uint32_t word = reinterpret_cast<const uint32_t*>(data)[idx];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (T)((float)((int)((word >> (4 * e)) & 15u) - 8) * s);
    } else {
      const u32x2 two = reinterpret_cast<const u32x2*>(data)[idx];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t word = e < 4 ? two[0] : two[1];
        v[e] = (T)((float)(int8_t)((word >> (8 * (e & 3))) & 255u) * s);
      }
    }
  }
  int out_row = row;
  if constexpr (GATE_UP) {
    const int quad = row >> 2, j = row & 3;
    out_row = j < 2 ? 2 * quad + j : (N >> 1) + 2 * quad + (j - 2);
  }
  *reinterpret_cast<vec_t*>(out + (int64_t)out_row * K + col) = v;
}

template <typename T, int FMT>
static void weight_expand_go(const void* data, const void* scales, void* out, int N, int K, bool gate_up, hipStream_t s) {
  const int64_t threads = (int64_t)(N >> 4) * (K >> 7) * 256;
  const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  if (gate_up)
    weight_expand_kernel<T, FMT, true><<<grid, block, 0, s>>>(data, (const uint16_t*)scales, (uint16_t*)out, N, K);
  else
    weight_expand_kernel<T, FMT, false><<<grid, block, 0, s>>>(data, (const uint16_t*)scales, (uint16_t*)out, N, K);
}

extern "C" int nv_weight_expand(const void* data, const void* scales, int fmt, int N, int K, int gate_up_rows, int dtype,
                                void* out, nv_stream_t stream) {
  NV_CHECK_ARG(data && out, "nv_weight_expand: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_weight_expand: bad dtype %d", dtype);
  NV_CHECK_ARG(fmt == NV_W_DENSE || fmt == NV_W_DENSE_TILED || fmt == NV_W_INT4 || fmt == NV_W_INT8, "nv_weight_expand: bad format %d", fmt);
  NV_CHECK_ARG((fmt != NV_W_INT4 && fmt != NV_W_INT8) || scales, "nv_weight_expand: quantised weights need their scales");
  NV_CHECK_ARG(N > 0 && K > 0 && N % 16 == 0 && K % 128 == 0, "nv_weight_expand: N=%d must be a multiple of 16, K=%d of 128", N, K);
  NV_CHECK_ARG(!gate_up_rows || N % 4 == 0, "nv_weight_expand: gate/up rows come in fours");
  NV_CHECK_ARG(((uintptr_t)data | (uintptr_t)out) % 16 == 0, "nv_weight_expand: pointers must be 16-byte aligned");
  NV_CHECK_ARG((int64_t)(N >> 4) * (K >> 7) < (1ll << 31), "nv_weight_expand: weight too large");
  hipStream_t s = (hipStream_t)stream;
  const bool gu = gate_up_rows != 0;
#define EXPAND(T)                                                                                   \
  switch (fmt) {                                                                                    \
    case NV_W_DENSE: weight_expand_go<T, NV_W_DENSE>(data, scales, out, N, K, gu, s); break;        \
    case NV_W_DENSE_TILED: weight_expand_go<T, NV_W_DENSE_TILED>(data, scales, out, N, K, gu, s); break; \
    case NV_W_INT4: weight_expand_go<T, NV_W_INT4>(data, scales, out, N, K, gu, s); break;          \
    default: weight_expand_go<T, NV_W_INT8>(data, scales, out, N, K, gu, s); break;                 \
  }
  if (dtype == NV_BF16) {
    EXPAND(bf16_t)
  } else {
    EXPAND(f16_t)
  }
#undef EXPAND
  NV_CHECK_LAUNCH();
  return NV_OK;
}
