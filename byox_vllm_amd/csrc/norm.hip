// RMSNorm and fused residual-add RMSNorm for gfx950.
// One 64-lane wavefront owns one row: 16-byte loads (8 elements per lane per
// step), an xor-shuffle reduction of the sum of squares, 16-byte stores.  The
// rows are a few KB, so the second pass over the row is served by the CU's L1.
#include "common.h"
#include "splitk.h"

// x as the GEMM would have written it: the split-K slabs summed in slab order, rounded to the model dtype.
template <typename T>
__device__ __forceinline__ u32x4 sum_partials(const float* __restrict__ part, int64_t slab, int nparts, int64_t off) {
  f32x4 acc[2];
  slab_sum<2>(part + off, slab, nparts, acc);
  return u32x4{pack2<T>(acc[0][0], acc[0][1]), pack2<T>(acc[0][2], acc[0][3]), pack2<T>(acc[1][0], acc[1][1]),
               pack2<T>(acc[1][2], acc[1][3])};
}

template <typename T, bool ADD, bool PARTIAL>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const uint16_t* x, const uint16_t* res,
                                                      const uint16_t* __restrict__ w, uint16_t* out,
                                                      uint16_t* res_out, int64_t rows, int hidden,
                                                      int64_t x_stride, int64_t out_stride, float eps,
                                                      const float* __restrict__ part, int nparts) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = hidden >> 3;
  const u32x4* xv = reinterpret_cast<const u32x4*>(x + row * x_stride);
  const u32x4* rv = ADD ? reinterpret_cast<const u32x4*>(res + row * x_stride) : nullptr;
  u32x4* ro = ADD ? reinterpret_cast<u32x4*>(res_out + row * x_stride) : nullptr;
  const u32x4* wv = reinterpret_cast<const u32x4*>(w);
  u32x4* ov = reinterpret_cast<u32x4*>(out + row * out_stride);

  const int64_t slab = rows * hidden;
  float ss = 0.f;
  for (int i = lane; i < nvec; i += 64) {
    u32x4 a = PARTIAL ? sum_partials<T>(part, slab, nparts, row * hidden + 8 * i) : xv[i];
    u32x4 b = {0, 0, 0, 0};
    if (ADD) b = rv[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = lo_f32<T>(a[j]), hi = hi_f32<T>(a[j]);
      if (ADD) {
        lo = lo + lo_f32<T>(b[j]);
        hi = hi + hi_f32<T>(b[j]);
      }
      ss = fmaf(lo, lo, ss);
      ss = fmaf(hi, hi, ss);
    }
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);

  for (int i = lane; i < nvec; i += 64) {
    u32x4 a = PARTIAL ? sum_partials<T>(part, slab, nparts, row * hidden + 8 * i) : xv[i];
    u32x4 b = {0, 0, 0, 0};
    if (ADD) b = rv[i];
    u32x4 ww = wv[i];
    u32x4 o, r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = lo_f32<T>(a[j]), hi = hi_f32<T>(a[j]);
      if (ADD) {
        lo = lo + lo_f32<T>(b[j]);
        hi = hi + hi_f32<T>(b[j]);
        r[j] = pack2<T>(lo, hi);
      }
      // normalise in fp32, round to the model dtype, THEN scale by the weight
      const float nlo = bits_to_f32<T>(f32_to_bits<T>(lo * inv));
      const float nhi = bits_to_f32<T>(f32_to_bits<T>(hi * inv));
      o[j] = pack2<T>(nlo * lo_f32<T>(ww[j]), nhi * hi_f32<T>(ww[j]));
    }
    if (ADD) ro[i] = r;
    ov[i] = o;
  }
}

// One workgroup per row, the row held in registers between the reduction and the scaling: used for the
// model's hidden-size rows (few rows, each a few KB, possibly still spread over split-K slabs).
template <typename T, bool ADD, bool PARTIAL, int NT, int NV, bool RNT>
__global__ __launch_bounds__(NT) void rmsnorm_row_kernel(const uint16_t* x, const uint16_t* res,
                                                         const uint16_t* __restrict__ w, uint16_t* out,
                                                         uint16_t* res_out, int64_t rows, int hidden, int64_t x_stride,
                                                         int64_t out_stride, float eps, const float* __restrict__ part,
                                                         int nparts) {
  __shared__ float s_part[NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t row = blockIdx.x;
  const int nvec = hidden >> 3;
  const int64_t slab = rows * hidden;
  float v[NV][8];
  float ss = 0.f;
  // the norm weights are fetched together with the row, not after the reduction: one memory round trip less
  u32x4 wreg[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = tid + k * NT;
    wreg[k] = (i < nvec) ? reinterpret_cast<const u32x4*>(w)[i] : u32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = tid + k * NT;
    if (i < nvec) {
      u32x4 a;
      if (PARTIAL) a = sum_partials<T>(part, slab, nparts, row * hidden + 8 * i);
      else a = reinterpret_cast<const u32x4*>(x + row * x_stride)[i];
      u32x4 b = {0, 0, 0, 0};
      if (ADD) b = reinterpret_cast<const u32x4*>(res + row * x_stride)[i];
      u32x4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float lo = lo_f32<T>(a[j]), hi = hi_f32<T>(a[j]);
        if (ADD) {
          lo = lo + lo_f32<T>(b[j]);
          hi = hi + hi_f32<T>(b[j]);
          r[j] = pack2<T>(lo, hi);
        }
        v[k][2 * j] = lo;
        v[k][2 * j + 1] = hi;
        ss = fmaf(lo, lo, ss);
        ss = fmaf(hi, hi, ss);
      }
      if (ADD) out_store<RNT>(&reinterpret_cast<u32x4*>(res_out + row * x_stride)[i], r);
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) s_part[wid] = ss;
  __syncthreads();
  ss = 0.f;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) ss += s_part[k];
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = tid + k * NT;
    if (i < nvec) {
      const u32x4 ww = wreg[k];
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float nlo = bits_to_f32<T>(f32_to_bits<T>(v[k][2 * j] * inv));
        const float nhi = bits_to_f32<T>(f32_to_bits<T>(v[k][2 * j + 1] * inv));
        o[j] = pack2<T>(nlo * lo_f32<T>(ww[j]), nhi * hi_f32<T>(ww[j]));
      }
      out_store<RNT>(&reinterpret_cast<u32x4*>(out + row * out_stride)[i], o);
    }
  }
}

// The same with EPT = 4 (or 2) elements per thread (8- or 4-byte loads of the 16-bit rows, one float4 / float2 per slab):
// more wavefronts per row, for rows short enough that one workgroup covers them.  A norm launch is a chain of memory
// latencies, and more wavefronts per row keep more of the row's loads in flight at once.
template <typename T, bool ADD, bool PARTIAL, bool RNT, int EPT, int NT>
__global__ __launch_bounds__(NT) void rmsnorm_rowe_kernel(const uint16_t* x, const uint16_t* res,
                                                          const uint16_t* __restrict__ w, uint16_t* out,
                                                          uint16_t* res_out, int64_t rows, int hidden, int64_t x_stride,
                                                          int64_t out_stride, float eps, const float* __restrict__ part,
                                                          int nparts) {
  typedef uint32_t pk_t __attribute__((ext_vector_type(EPT / 2)));   // EPT 16-bit values
  typedef float fv_t __attribute__((ext_vector_type(EPT)));
  __shared__ float s_part[NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t row = blockIdx.x;
  const bool live = tid < hidden / EPT;
  const int i = live ? tid : 0;
  const pk_t wreg = reinterpret_cast<const pk_t*>(w)[i];
  float a[EPT];
  if (PARTIAL) {
    const float* p = part + row * hidden + EPT * i;
    const int64_t slab = rows * hidden;
    fv_t acc = *reinterpret_cast<const fv_t*>(p);
    int z = 1;
    for (; z + 8 <= nparts; z += 8) {      // eight slabs' loads go out before the first add; slab order is kept
      fv_t v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const fv_t*>(p + (z + k) * slab);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k];
    }
    {
      fv_t v[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) v[k] = *reinterpret_cast<const fv_t*>(p + (z + k < nparts ? z + k : 0) * slab);
#pragma unroll
      for (int k = 0; k < 7; ++k) if (z + k < nparts) acc += v[k];
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) a[j] = bits_to_f32<T>(f32_to_bits<T>(acc[j]));   // what the GEMM's reduction would store
  } else {
    const pk_t xv = reinterpret_cast<const pk_t*>(x + row * x_stride)[i];
#pragma unroll
    for (int j = 0; j < EPT / 2; ++j) { a[2 * j] = lo_f32<T>(xv[j]); a[2 * j + 1] = hi_f32<T>(xv[j]); }
  }
  if (ADD) {
    const pk_t rv = reinterpret_cast<const pk_t*>(res + row * x_stride)[i];
    pk_t r;
#pragma unroll
    for (int j = 0; j < EPT / 2; ++j) {
      a[2 * j] += lo_f32<T>(rv[j]);
      a[2 * j + 1] += hi_f32<T>(rv[j]);
      r[j] = pack2<T>(a[2 * j], a[2 * j + 1]);
    }
    if (live) out_store<RNT>(&reinterpret_cast<pk_t*>(res_out + row * x_stride)[i], r);
  }
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < EPT; ++j) ss = fmaf(a[j], a[j], ss);
  ss = wave_sum(live ? ss : 0.f);
  if (lane == 0) s_part[wid] = ss;
  __syncthreads();
  ss = 0.f;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) ss += s_part[k];
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
  pk_t o;
#pragma unroll
  for (int j = 0; j < EPT / 2; ++j)
    o[j] = pack2<T>(bits_to_f32<T>(f32_to_bits<T>(a[2 * j] * inv)) * lo_f32<T>(wreg[j]),
                    bits_to_f32<T>(f32_to_bits<T>(a[2 * j + 1] * inv)) * hi_f32<T>(wreg[j]));
  if (live) out_store<RNT>(&reinterpret_cast<pk_t*>(out + row * out_stride)[i], o);
}

template <typename T, bool ADD, bool PARTIAL, bool RNT>
static bool launch_norm_row(const void* x, const void* res, const void* w, void* out, void* res_out, int64_t rows,
                            int hidden, int64_t xs, int64_t os, float eps, hipStream_t s, const float* part,
                            int nparts) {
  const int nvec = hidden >> 3;
#define ROW_GO(NTV, NVV)                                                                                             \
  rmsnorm_row_kernel<T, ADD, PARTIAL, NTV, NVV, RNT><<<dim3((unsigned)rows), dim3(NTV), 0, s>>>(                     \
      (const uint16_t*)x, (const uint16_t*)res, (const uint16_t*)w, (uint16_t*)out, (uint16_t*)res_out, rows, hidden, \
      xs, os, eps, part, nparts)
#define ROWE_GO(EPTV, NTV)                                                                                           \
  rmsnorm_rowe_kernel<T, ADD, PARTIAL, RNT, EPTV, NTV><<<dim3((unsigned)rows), dim3(NTV), 0, s>>>(                    \
      (const uint16_t*)x, (const uint16_t*)res, (const uint16_t*)w, (uint16_t*)out, (uint16_t*)res_out, rows, hidden, \
      xs, os, eps, part, nparts)
  // hidden <= 4096: four elements per thread, one workgroup of up to 1024 threads per row.  Measured on the 896-wide rows of Qwen2-0.5B (48
  // launches per decode step): 8 elements per thread in 2 wavefronts 4.6 us per launch, 4 per thread in 4 wavefronts
  // 0.15-0.25 us less, 2 per thread in 8 wavefronts no better, all 16 per lane in one wavefront 0.65 us more.
  // The 3584-wide rows of Qwen2-7B (56 launches per step): 6.1 us with 8 elements per thread twice in 256 threads,
  // 0.56 us less with 4 per thread in a workgroup of 1024.
  if (nvec <= 128) ROWE_GO(4, 256);
  else if (nvec <= 256) ROWE_GO(4, 512);
  else if (nvec <= 512) ROWE_GO(4, 1024);
  else if (nvec <= 1024) ROW_GO(256, 4);
  else return false;
#undef ROW_GO
#undef ROWE_GO
  return true;
}

template <bool ADD, bool PARTIAL = false>
static int launch_norm(const void* x, const void* res, const void* w, void* out, void* res_out, int64_t rows,
                       int hidden, int64_t xs, int64_t os, float eps, int dtype, hipStream_t s,
                       const float* part = nullptr, int nparts = 0) {
  if (rows == 0) return NV_OK;
  // few long rows: one workgroup per row; many short rows (per-head norms): one wavefront per row
  if ((PARTIAL || hidden >= 512) && rows <= 0x7fffffff) {
    // the model's hidden-size rows are what the next GEMM reads: they follow the result-store policy
    const bool nt = nv_result_policy() != 0;
#define ROW(TT, RNT) launch_norm_row<TT, ADD, PARTIAL, RNT>(x, res, w, out, res_out, rows, hidden, xs, os, eps, s, part, nparts)
    const bool ok = (dtype == NV_BF16) ? (nt ? ROW(bf16_t, true) : ROW(bf16_t, false))
                                       : (nt ? ROW(f16_t, true) : ROW(f16_t, false));
#undef ROW
    if (ok) {
      NV_CHECK_LAUNCH();
      return NV_OK;
    }
    if (PARTIAL) {
      nv_set_error("nv_add_rmsnorm_splitk: hidden=%d above the 8192 this kernel covers", hidden);
      return NV_ERR_UNSUPPORTED;
    }
  }
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (dtype == NV_BF16)
    rmsnorm_kernel<bf16_t, ADD, PARTIAL><<<grid, block, 0, s>>>((const uint16_t*)x, (const uint16_t*)res,
                                                                (const uint16_t*)w, (uint16_t*)out, (uint16_t*)res_out,
                                                                rows, hidden, xs, os, eps, part, nparts);
  else
    rmsnorm_kernel<f16_t, ADD, PARTIAL><<<grid, block, 0, s>>>((const uint16_t*)x, (const uint16_t*)res,
                                                               (const uint16_t*)w, (uint16_t*)out, (uint16_t*)res_out,
                                                               rows, hidden, xs, os, eps, part, nparts);
  NV_CHECK_LAUNCH();
  return NV_OK;
}

extern "C" int nv_rmsnorm(const void* x, const void* weight, void* out, int64_t rows, int hidden, int64_t x_row_stride,
                          int64_t out_row_stride, float eps, int dtype, nv_stream_t stream) {
  NV_CHECK_ARG(x && weight && out, "nv_rmsnorm: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_rmsnorm: bad dtype %d", dtype);
  NV_CHECK_ARG(hidden > 0 && hidden % 8 == 0, "nv_rmsnorm: hidden=%d must be a positive multiple of 8", hidden);
  NV_CHECK_ARG(x_row_stride % 8 == 0 && out_row_stride % 8 == 0, "nv_rmsnorm: row strides must be multiples of 8");
  NV_CHECK_ARG(((uintptr_t)x | (uintptr_t)out | (uintptr_t)weight) % 16 == 0, "nv_rmsnorm: pointers must be 16-byte aligned");
  return launch_norm<false>(x, nullptr, weight, out, nullptr, rows, hidden, x_row_stride, out_row_stride, eps, dtype,
                            (hipStream_t)stream);
}

extern "C" int nv_add_rmsnorm(const void* x, const void* residual, const void* weight, void* out, void* residual_out,
                              int64_t rows, int hidden, float eps, int dtype, nv_stream_t stream) {
  NV_CHECK_ARG(x && residual && weight && out && residual_out, "nv_add_rmsnorm: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_add_rmsnorm: bad dtype %d", dtype);
  NV_CHECK_ARG(hidden > 0 && hidden % 8 == 0, "nv_add_rmsnorm: hidden=%d must be a positive multiple of 8", hidden);
  NV_CHECK_ARG(((uintptr_t)x | (uintptr_t)out | (uintptr_t)weight | (uintptr_t)residual | (uintptr_t)residual_out) % 16 == 0,
               "nv_add_rmsnorm: pointers must be 16-byte aligned");
  return launch_norm<true>(x, residual, weight, out, residual_out, rows, hidden, hidden, hidden, eps, dtype,
                           (hipStream_t)stream);
}

extern "C" int nv_add_rmsnorm_splitk(const float* partials, int num_partials, const void* residual, const void* weight,
                                     void* out, void* residual_out, int64_t rows, int hidden, float eps, int dtype,
                                     nv_stream_t stream) {
  NV_CHECK_ARG(partials && residual && weight && out && residual_out, "nv_add_rmsnorm_splitk: null pointer");
  NV_CHECK_ARG(num_partials >= 1, "nv_add_rmsnorm_splitk: num_partials=%d", num_partials);
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_add_rmsnorm_splitk: bad dtype %d", dtype);
  NV_CHECK_ARG(hidden > 0 && hidden % 8 == 0, "nv_add_rmsnorm_splitk: hidden=%d must be a positive multiple of 8", hidden);
  NV_CHECK_ARG(((uintptr_t)partials | (uintptr_t)out | (uintptr_t)weight | (uintptr_t)residual | (uintptr_t)residual_out) % 16 == 0,
               "nv_add_rmsnorm_splitk: pointers must be 16-byte aligned");
  return launch_norm<true, true>(nullptr, residual, weight, out, residual_out, rows, hidden, hidden, hidden, eps, dtype,
                                 (hipStream_t)stream, partials, num_partials);
}
