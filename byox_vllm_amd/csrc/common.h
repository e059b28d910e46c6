// Shared device helpers for the gfx950 kernels behind the nv_* C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nv_hip.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef f16_t f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Store of a kernel's result that the next launch of a chain reads.  WT = written through at device scope (sc1): the
// line goes to the memory side while the kernel still runs instead of sitting dirty in the writing XCD's L2 until the
// kernel ends, which is where a consumer on another XCD has to fetch it from anyway; and it leaves the L2 and the
// memory-side cache to the weights.  Measured per decode step against plain stores: Qwen2-0.5B int4 -2.4 %, Qwen2-7B
// int4 -2.4 %, 0.5B int8 -0.7 %; a non-temporal store (nt) in its place gains 1.6 % on the first and LOSES 1.3-1.8 % on
// the others (the consumer then reads HBM), nt together with sc1 loses 6 %.  The choice is a template parameter of
// the kernels, made by their launchers from the process-wide result-store policy (nv_set_result_policy): written as
// a run-time branch around the store it costs every kernel about 0.1 us with the policy off.
template <bool WT, typename V>
__device__ __forceinline__ void out_store(V* p, V v) {
  static_assert(sizeof(V) == 16 || sizeof(V) == 8 || sizeof(V) == 4, "out_store: 4, 8 or 16 bytes");
  if constexpr (!WT) *p = v;
  else if constexpr (sizeof(V) == 16) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (sizeof(V) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
int nv_result_policy();   // 1: results are written through (misc.hip)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define NV_WAVE 64

// Last error text, readable through nv_last_error().
void nv_set_error(const char* fmt, ...);

#define NV_CHECK_ARG(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      nv_set_error(__VA_ARGS__);         \
      return NV_ERR_INVALID;             \
    }                                    \
  } while (0)

#define NV_CHECK_LAUNCH()                                              \
  do {                                                                 \
    hipError_t e_ = hipGetLastError();                                 \
    if (e_ != hipSuccess) {                                            \
      nv_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
      return NV_ERR_LAUNCH;                                            \
    }                                                                  \
  } while (0)

template <typename T> struct VecOf;
template <> struct VecOf<bf16_t> { typedef bf16x8 x8; };
template <> struct VecOf<f16_t> { typedef f16x8 x8; };

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// 16-bit storage <-> float.  The 16-bit payload travels as raw bits in uint16/uint32.
template <typename T> __device__ __forceinline__ float bits_to_f32(uint16_t b);
template <> __device__ __forceinline__ float bits_to_f32<bf16_t>(uint16_t b) {
  return __uint_as_float(((uint32_t)b) << 16);
}
template <> __device__ __forceinline__ float bits_to_f32<f16_t>(uint16_t b) {
  return (float)__builtin_bit_cast(f16_t, b);
}
template <typename T> __device__ __forceinline__ uint16_t f32_to_bits(float f) {
  return __builtin_bit_cast(uint16_t, (T)f);   // round to nearest even
}
// low / high half of a packed pair
template <typename T> __device__ __forceinline__ float lo_f32(uint32_t w) { return bits_to_f32<T>((uint16_t)(w & 0xffffu)); }
template <typename T> __device__ __forceinline__ float hi_f32(uint32_t w) { return bits_to_f32<T>((uint16_t)(w >> 16)); }
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  return (uint32_t)f32_to_bits<T>(lo) | ((uint32_t)f32_to_bits<T>(hi) << 16);
}

// The four 16-lane rows of a wavefront (lanes l, l^16, l^32, l^48) exchanged on the VALU: v_permlane16_swap and
// v_permlane32_swap trade rows / halves between two registers without the LDS crossbar a __shfl_xor goes through.
// With both operands holding x, rows_pair16 returns (value of the even row, value of the odd row) of the lane's row
// pair, rows_pair32 (value of the lower half, value of the upper half) at the lane's position: the same two numbers in
// the same order in both lanes of a pair, so a reduction over them gives every lane identical bits.
typedef uint32_t u32x2_swap __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rows_pair16(uint32_t x, uint32_t& even, uint32_t& odd) {
  const u32x2_swap r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  even = r[0];
  odd = r[1];
}
__device__ __forceinline__ void rows_pair32(uint32_t x, uint32_t& lower, uint32_t& upper) {
  const u32x2_swap r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  lower = r[0];
  upper = r[1];
}
__device__ __forceinline__ void rows_pair16(float x, float& even, float& odd) {
  uint32_t a, b;
  rows_pair16(__float_as_uint(x), a, b);
  even = __uint_as_float(a);
  odd = __uint_as_float(b);
}
__device__ __forceinline__ void rows_pair32(float x, float& lower, float& upper) {
  uint32_t a, b;
  rows_pair32(__float_as_uint(x), a, b);
  lower = __uint_as_float(a);
  upper = __uint_as_float(b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
