// RoPE (+ optional per-head RMSNorm) on q/k in place, fused with the KV-cache
// store, and the stand-alone KV-cache store.
//
// One wavefront per (token, head-slot).  Head-slots 0..H-1 are q heads,
// H..H+Hkv-1 are k heads (rotated in place and copied to the cache row of the
// token's slot), H+Hkv..H+2Hkv-1 are v heads (copied only).  A lane owns the
// pair (x[i], x[i + D/2]) for i = lane, lane+64... so the rotation needs no
// cross-lane traffic; the per-head norm is one wave reduction.
#include "common.h"

template <typename T, bool PARTIAL>
__global__ __launch_bounds__(256) void rope_kv_kernel(uint16_t* qkv, int64_t row_stride,
                                                      const int64_t* __restrict__ positions,
                                                      const float* __restrict__ cos_sin,
                                                      const int32_t* __restrict__ slots, uint16_t* k_cache,
                                                      uint16_t* v_cache, const uint16_t* __restrict__ qn_w,
                                                      const uint16_t* __restrict__ kn_w, float eps, int64_t ntok, int H,
                                                      int Hkv, int D, const float* __restrict__ part, int nparts,
                                                      const uint16_t* __restrict__ bias) {
  const int lane = threadIdx.x & 63;
  const int heads = H + 2 * Hkv;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= ntok * heads) return;
  const int64_t t = item / heads;
  const int hs = (int)(item - t * heads);
  uint16_t* row = qkv + t * row_stride + (int64_t)hs * D;
  const int half = D >> 1;
  const bool is_q = hs < H, is_k = !is_q && hs < H + Hkv;
  const int slot = slots ? slots[t] : -1;

  // PARTIAL: the qkv row does not exist yet; build this head from the GEMM's split-K slabs (+ bias),
  // rounded to the model dtype exactly as the GEMM epilogue would have stored it.
  const int ncols = heads * D;
  auto from_partials = [&](int col) -> uint16_t {
    float a = 0.f;
    for (int z = 0; z < nparts; ++z) a += part[((int64_t)z * ntok + t) * ncols + col];
    if (bias) a += bits_to_f32<T>(bias[col]);
    return f32_to_bits<T>(a);
  };

  if (!is_q && !is_k) {  // v head: copy to the cache (and materialise it in the qkv buffer)
    uint16_t* dst = (v_cache && slot >= 0) ? v_cache + ((int64_t)slot * Hkv + (hs - H - Hkv)) * D : nullptr;
    for (int i = lane; i < D; i += 64) {
      uint16_t val;
      if (PARTIAL) {
        val = from_partials(hs * D + i);
        row[i] = val;
      } else {
        val = row[i];
      }
      if (dst) dst[i] = val;
    }
    return;
  }

  const uint16_t* nw = is_q ? qn_w : kn_w;
  const float* cs = cos_sin + positions[t] * D;
  // D <= 128: one (x1, x2) pair per lane (lanes >= D/2 idle for D = 64)
  const bool act = lane < half;
  float x1 = 0.f, x2 = 0.f;
  if (act) {
    x1 = bits_to_f32<T>(PARTIAL ? from_partials(hs * D + lane) : row[lane]);
    x2 = bits_to_f32<T>(PARTIAL ? from_partials(hs * D + lane + half) : row[lane + half]);
  }
  if (nw) {  // per-head RMSNorm (Qwen3): fp32 normalise, round, scale by weight in the model dtype
    float ss = wave_sum(x1 * x1 + x2 * x2);
    const float inv = 1.0f / sqrtf(ss / (float)D + eps);
    if (act) {
      x1 = bits_to_f32<T>(f32_to_bits<T>(bits_to_f32<T>(f32_to_bits<T>(x1 * inv)) * bits_to_f32<T>(nw[lane])));
      x2 = bits_to_f32<T>(f32_to_bits<T>(bits_to_f32<T>(f32_to_bits<T>(x2 * inv)) * bits_to_f32<T>(nw[lane + half])));
    }
  }
  if (act) {
    const float c = cs[lane], s = cs[half + lane];
    // separate roundings for each product and the sum, as eager fp32 PyTorch does
    const float y1 = __fsub_rn(__fmul_rn(x1, c), __fmul_rn(x2, s));
    const float y2 = __fadd_rn(__fmul_rn(x2, c), __fmul_rn(x1, s));
    const uint16_t b1 = f32_to_bits<T>(y1), b2 = f32_to_bits<T>(y2);
    row[lane] = b1;
    row[lane + half] = b2;
    if (is_k && k_cache && slot >= 0) {
      uint16_t* dst = k_cache + ((int64_t)slot * Hkv + (hs - H)) * D;
      dst[lane] = b1;
      dst[lane + half] = b2;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void kv_store_kernel(const uint16_t* __restrict__ key, int64_t ks,
                                                       const uint16_t* __restrict__ value, int64_t vs,
                                                       uint16_t* k_cache, uint16_t* v_cache,
                                                       const int32_t* __restrict__ slots, int64_t ntok, int row_elems) {
  const int64_t t = blockIdx.x;
  const int slot = slots[t];
  if (slot < 0) return;
  const int nvec = row_elems >> 3;
  const u32x4* ksrc = reinterpret_cast<const u32x4*>(key + t * ks);
  const u32x4* vsrc = reinterpret_cast<const u32x4*>(value + t * vs);
  u32x4* kd = reinterpret_cast<u32x4*>(k_cache + (int64_t)slot * row_elems);
  u32x4* vd = reinterpret_cast<u32x4*>(v_cache + (int64_t)slot * row_elems);
  for (int i = threadIdx.x; i < nvec; i += blockDim.x) {
    kd[i] = ksrc[i];
    vd[i] = vsrc[i];
  }
}

static int rope_launch(void* qkv, int64_t qkv_row_stride, const int64_t* positions, const float* cos_sin,
                       const int32_t* slot_mapping, void* k_cache, void* v_cache, const void* q_norm_w,
                       const void* k_norm_w, float eps, int64_t T, int H, int Hkv, int D, int dtype,
                       const float* partials, int num_partials, const void* bias, nv_stream_t stream) {
  NV_CHECK_ARG(qkv && positions && cos_sin, "nv_rope_kv_store: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_rope_kv_store: bad dtype %d", dtype);
  NV_CHECK_ARG(D == 64 || D == 128 || D == 32 || D == 16, "nv_rope_kv_store: head_dim %d not in {16,32,64,128}", D);
  NV_CHECK_ARG(H >= 0 && Hkv >= 0 && H + Hkv > 0, "nv_rope_kv_store: bad head counts");
  NV_CHECK_ARG((k_cache == nullptr) == (v_cache == nullptr), "nv_rope_kv_store: k_cache and v_cache must both be set or both NULL");
  NV_CHECK_ARG(!k_cache || slot_mapping, "nv_rope_kv_store: cache given without slot_mapping");
  NV_CHECK_ARG((q_norm_w == nullptr) == (k_norm_w == nullptr), "nv_rope_kv_store: q_norm_w and k_norm_w go together");
  NV_CHECK_ARG(qkv_row_stride >= (int64_t)(H + 2 * Hkv) * D, "nv_rope_kv_store: row stride smaller than a qkv row");
  if (T == 0) return NV_OK;
  const int64_t items = T * (H + 2 * Hkv);
  dim3 grid((unsigned)((items + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define ROPE_GO(TT, PP)                                                                                               \
  rope_kv_kernel<TT, PP><<<grid, block, 0, s>>>((uint16_t*)qkv, qkv_row_stride, positions, cos_sin, slot_mapping,    \
                                                (uint16_t*)k_cache, (uint16_t*)v_cache, (const uint16_t*)q_norm_w,   \
                                                (const uint16_t*)k_norm_w, eps, T, H, Hkv, D, partials, num_partials, \
                                                (const uint16_t*)bias)
  if (dtype == NV_BF16) { if (partials) ROPE_GO(bf16_t, true); else ROPE_GO(bf16_t, false); }
  else                  { if (partials) ROPE_GO(f16_t, true); else ROPE_GO(f16_t, false); }
#undef ROPE_GO
  NV_CHECK_LAUNCH();
  return NV_OK;
}

extern "C" int nv_rope_kv_store(void* qkv, int64_t qkv_row_stride, const int64_t* positions, const float* cos_sin,
                                const int32_t* slot_mapping, void* k_cache, void* v_cache, const void* q_norm_w,
                                const void* k_norm_w, float eps, int64_t T, int H, int Hkv, int D, int dtype,
                                nv_stream_t stream) {
  return rope_launch(qkv, qkv_row_stride, positions, cos_sin, slot_mapping, k_cache, v_cache, q_norm_w, k_norm_w, eps,
                     T, H, Hkv, D, dtype, nullptr, 0, nullptr, stream);
}

extern "C" int nv_rope_kv_store_splitk(const float* partials, int num_partials, const void* bias, void* qkv,
                                       int64_t qkv_row_stride, const int64_t* positions, const float* cos_sin,
                                       const int32_t* slot_mapping, void* k_cache, void* v_cache, const void* q_norm_w,
                                       const void* k_norm_w, float eps, int64_t T, int H, int Hkv, int D, int dtype,
                                       nv_stream_t stream) {
  NV_CHECK_ARG(partials && num_partials >= 1, "nv_rope_kv_store_splitk: no partials");
  return rope_launch(qkv, qkv_row_stride, positions, cos_sin, slot_mapping, k_cache, v_cache, q_norm_w, k_norm_w, eps,
                     T, H, Hkv, D, dtype, partials, num_partials, bias, stream);
}

extern "C" int nv_kv_store(const void* key, int64_t key_row_stride, const void* value, int64_t value_row_stride,
                           void* k_cache, void* v_cache, const int32_t* slot_mapping, int64_t T, int row_elems,
                           int dtype, nv_stream_t stream) {
  NV_CHECK_ARG(key && value && k_cache && v_cache && slot_mapping, "nv_kv_store: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_kv_store: bad dtype %d", dtype);
  NV_CHECK_ARG(row_elems > 0 && row_elems % 8 == 0, "nv_kv_store: row_elems=%d must be a multiple of 8", row_elems);
  NV_CHECK_ARG(key_row_stride % 8 == 0 && value_row_stride % 8 == 0, "nv_kv_store: row strides must be multiples of 8");
  NV_CHECK_ARG(((uintptr_t)key | (uintptr_t)value | (uintptr_t)k_cache | (uintptr_t)v_cache) % 16 == 0,
               "nv_kv_store: pointers must be 16-byte aligned");
  if (T == 0) return NV_OK;
  hipStream_t s = (hipStream_t)stream;
  int threads = row_elems / 8 >= 256 ? 256 : (row_elems / 8 > 64 ? 128 : 64);
  kv_store_kernel<bf16_t><<<dim3((unsigned)T), dim3(threads), 0, s>>>((const uint16_t*)key, key_row_stride,
                                                                      (const uint16_t*)value, value_row_stride,
                                                                      (uint16_t*)k_cache, (uint16_t*)v_cache,
                                                                      slot_mapping, T, row_elems);
  NV_CHECK_LAUNCH();
  return NV_OK;
}
