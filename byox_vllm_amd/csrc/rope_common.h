// The tail of the q/k/v projection shared by the stand-alone RoPE kernel and the GEMM's RoPE epilogue: per-head
// RMSNorm (Qwen3), rotation, store to the qkv buffer and to the token's KV-cache row.
#pragma once
#include "common.h"

struct RopeArgs {
  const int64_t* positions;   // [T]
  const float* cos_sin;       // [max_position, D]: cos(D/2) | sin(D/2), fp32
  const int32_t* slots;       // [T] cache row of each token (negative: no store), or NULL
  uint16_t* k_cache;          // [rows, Hkv, D] or NULL
  uint16_t* v_cache;
  const uint16_t* qn_w;       // [D] per-head norm weights or NULL
  const uint16_t* kn_w;
  float eps;
  int H, Hkv, D;
};

template <typename T>
__device__ __forceinline__ f32x4 unpack4(u32x2 w) {
  return f32x4{lo_f32<T>(w[0]), hi_f32<T>(w[0]), lo_f32<T>(w[1]), hi_f32<T>(w[1])};
}
template <typename T>
__device__ __forceinline__ u32x2 pack4(f32x4 v) {
  return u32x2{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
}
// round to the model dtype and back
template <typename T>
__device__ __forceinline__ f32x4 round4(f32x4 v) {
  return unpack4<T>(pack4<T>(v));
}

// Head-slot hs of token t: 0..H-1 query heads, H..H+Hkv-1 key heads (rotated, copied to the cache), then value
// heads (copied only).  D/8 consecutive lanes share the head; lane `sub` owns elements [4 sub, 4 sub + 4) of the
// first half (x1) and the matching four of the second half (x2), already rounded to the model dtype.
// `row` = the head's place in the qkv buffer; write_v_row: value heads are written there too (they are not there yet).
// Every lane of a head group must make the call (shuffles); lanes with live == false only take part in those.
template <typename T, bool NT = false>   // NT: q/k/v rows and cache rows written through (common.h out_store)
__device__ __forceinline__ void rope_finish(f32x4 x1, f32x4 x2, const RopeArgs& ra, int hs, int64_t t, int sub,
                                            uint16_t* row, bool write_v_row, bool live) {
  const int D = ra.D, H = ra.H, Hkv = ra.Hkv, lph = D >> 3;
  const int c1 = 4 * sub, c2 = (D >> 1) + 4 * sub;
  const bool is_q = hs < H, is_k = !is_q && hs < H + Hkv;
  const int slot = (ra.slots && live) ? ra.slots[t] : -1;

  if (!is_q && !is_k) {  // v head: copy to the cache (and materialise it in the qkv buffer)
    if (!live) return;
    const u32x2 w1 = pack4<T>(x1), w2 = pack4<T>(x2);
    if (write_v_row) {
      out_store<NT>(reinterpret_cast<u32x2*>(row + c1), w1);
      out_store<NT>(reinterpret_cast<u32x2*>(row + c2), w2);
    }
    if (ra.v_cache && slot >= 0) {
      uint16_t* dst = ra.v_cache + ((int64_t)slot * Hkv + (hs - H - Hkv)) * D;
      out_store<NT>(reinterpret_cast<u32x2*>(dst + c1), w1);
      out_store<NT>(reinterpret_cast<u32x2*>(dst + c2), w2);
    }
    return;
  }

  const uint16_t* nw = is_q ? ra.qn_w : ra.kn_w;
  if (nw) {  // per-head RMSNorm (Qwen3): fp32 normalise, round, scale by weight in the model dtype
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) ss += x1[j] * x1[j] + x2[j] * x2[j];
    for (int o = lph >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = 1.0f / sqrtf(ss / (float)D + ra.eps);
    const f32x4 w1 = unpack4<T>(*reinterpret_cast<const u32x2*>(nw + c1));
    const f32x4 w2 = unpack4<T>(*reinterpret_cast<const u32x2*>(nw + c2));
    x1 = round4<T>(round4<T>(x1 * inv) * w1);
    x2 = round4<T>(round4<T>(x2 * inv) * w2);
  }
  if (!live) return;
  const float* cs = ra.cos_sin + ra.positions[t] * D;
  const f32x4 c = *reinterpret_cast<const f32x4*>(cs + c1), sn = *reinterpret_cast<const f32x4*>(cs + c2);
  f32x4 y1, y2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // separate roundings for each product and the sum, as eager fp32 PyTorch does
    y1[j] = __fsub_rn(__fmul_rn(x1[j], c[j]), __fmul_rn(x2[j], sn[j]));
    y2[j] = __fadd_rn(__fmul_rn(x2[j], c[j]), __fmul_rn(x1[j], sn[j]));
  }
  const u32x2 w1 = pack4<T>(y1), w2 = pack4<T>(y2);
  out_store<NT>(reinterpret_cast<u32x2*>(row + c1), w1);
  out_store<NT>(reinterpret_cast<u32x2*>(row + c2), w2);
  if (is_k && ra.k_cache && slot >= 0) {
    uint16_t* dst = ra.k_cache + ((int64_t)slot * Hkv + (hs - H)) * D;
    out_store<NT>(reinterpret_cast<u32x2*>(dst + c1), w1);
    out_store<NT>(reinterpret_cast<u32x2*>(dst + c2), w2);
  }
}
