// RoPE (+ optional per-head RMSNorm) on q/k in place, fused with the KV-cache
// store, and the stand-alone KV-cache store.
//
// Work items are (token, head-slot) pairs.  Head-slots 0..H-1 are q heads,
// H..H+Hkv-1 are k heads (rotated in place and copied to the cache row of the
// token's slot), H+Hkv..H+2Hkv-1 are v heads (copied only).  D/8 lanes share a
// head: a lane owns elements [4i, 4i+4) of the first half and the matching four
// of the second half, so the rotation needs no cross-lane traffic, every access
// is 8 or 16 bytes wide, and the per-head norm is a reduction over D/8 lanes.
#include "common.h"
#include "rope_common.h"
#include "splitk.h"

template <typename T, bool PARTIAL, bool NT>  // PARTIAL: the qkv rows exist only as split-K slabs; NT: results written through, common.h out_store
__global__ __launch_bounds__(256) void rope_kv_kernel(uint16_t* qkv, int64_t row_stride, RopeArgs ra, int64_t ntok,
                                                      const float* __restrict__ part, int nparts,
                                                      const uint16_t* __restrict__ bias) {
  const int D = ra.D, H = ra.H, Hkv = ra.Hkv;
  const int lph = D >> 3;  // lanes per head: 2, 4, 8 or 16
  const int heads = H + 2 * Hkv;
  const int64_t item = ((int64_t)blockIdx.x * 256 + threadIdx.x) / lph;
  const int sub = threadIdx.x & (lph - 1);
  const bool live = item < ntok * heads;
  const int64_t t = live ? item / heads : 0;
  const int hs = live ? (int)(item - t * heads) : 0;
  const int c1 = 4 * sub, c2 = (D >> 1) + 4 * sub;  // this lane's columns inside the head
  uint16_t* row = qkv + t * row_stride + (int64_t)hs * D;

  f32x4 x1, x2;
  if (PARTIAL) {
    // the qkv row does not exist yet: build this head from the GEMM's split-K slabs (+ bias), rounded to the
    // model dtype exactly as the GEMM epilogue would have stored it
    const int ncols = heads * D;
    const int64_t slab = ntok * ncols;
    const float* p = part + t * ncols + hs * D;
    f32x4 a1[1], a2[1];
    slab_sum<1>(p + c1, slab, nparts, a1);
    slab_sum<1>(p + c2, slab, nparts, a2);
    x1 = a1[0];
    x2 = a2[0];
    if (bias) {
      x1 += unpack4<T>(*reinterpret_cast<const u32x2*>(bias + hs * D + c1));
      x2 += unpack4<T>(*reinterpret_cast<const u32x2*>(bias + hs * D + c2));
    }
    x1 = round4<T>(x1);
    x2 = round4<T>(x2);
  } else {
    x1 = unpack4<T>(*reinterpret_cast<const u32x2*>(row + c1));
    x2 = unpack4<T>(*reinterpret_cast<const u32x2*>(row + c2));
  }
  rope_finish<T, NT>(x1, x2, ra, hs, t, sub, row, PARTIAL, live);
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
  NV_CHECK_ARG(qkv_row_stride % 4 == 0 && (uintptr_t)qkv % 8 == 0 && (uintptr_t)cos_sin % 16 == 0,
               "nv_rope_kv_store: qkv must be 8-byte aligned with a row stride that is a multiple of 4, cos_sin 16-byte aligned");
  NV_CHECK_ARG(((uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)q_norm_w | (uintptr_t)k_norm_w | (uintptr_t)bias) % 8 == 0 &&
               (uintptr_t)partials % 16 == 0, "nv_rope_kv_store: misaligned pointer");
  const int64_t items = T * (H + 2 * Hkv);
  const int64_t lanes = items * (D / 8);
  NV_CHECK_ARG((lanes + 255) / 256 <= 0x7fffffff, "nv_rope_kv_store: T=%lld too large for one launch", (long long)T);
  dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  RopeArgs ra;
  ra.positions = positions; ra.cos_sin = cos_sin; ra.slots = slot_mapping; ra.k_cache = (uint16_t*)k_cache;
  ra.v_cache = (uint16_t*)v_cache; ra.qn_w = (const uint16_t*)q_norm_w; ra.kn_w = (const uint16_t*)k_norm_w;
  ra.eps = eps; ra.H = H; ra.Hkv = Hkv; ra.D = D;
#define ROPE_GO(TT, PP, NT) \
  rope_kv_kernel<TT, PP, NT><<<grid, block, 0, s>>>((uint16_t*)qkv, qkv_row_stride, ra, T, partials, num_partials, (const uint16_t*)bias)
#define ROPE_PICK(TT, PP) do { if (nv_result_policy()) ROPE_GO(TT, PP, true); else ROPE_GO(TT, PP, false); } while (0)
  if (dtype == NV_BF16) { if (partials) ROPE_PICK(bf16_t, true); else ROPE_PICK(bf16_t, false); }
  else                  { if (partials) ROPE_PICK(f16_t, true); else ROPE_PICK(f16_t, false); }
#undef ROPE_PICK
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
