// Per-row statistics of the logits that the inverse-CDF sampler draws from (sampler.py:11-15 of the reference samples
// from softmax(logits / T); greedy rows take the arg max).  A row of V logits is cut into parts of 64 columns; a part
// is described by its maximum m (over the logits as stored, i.e. rounded to the model dtype) and its mass
// s = sum exp((l - m) / T) (0 for greedy rows).  Two producers fill the same arrays with the same arithmetic, so a
// token does not depend on which one ran: the stand-alone scan over stored logits (sample_parts_scan_kernel, misc.hip)
// and the epilogue of the LM-head GEMM, which has the logits in registers anyway (gemm_impl.h, EPI == 4).  Both hold
// a 16-column tile as 4 consecutive columns in each of the lanes l, l^16, l^32, l^48 of a wavefront and reduce it
// with the functions below; a part is four such tiles joined in column order.  Which column holds a maximum is left
// to the consumer, which reads the 64 logits of the one part it ends up in anyway.
#pragma once
#include "common.h"

#define SAMPLE_PART_COLS 64
#define NV_SAMPLE_MAX_VOCAB (64 * 8128)   /* the draw keeps a row's part maxima and masses (8 bytes per 64-column part) in dynamic LDS next to 512 bytes reserved for its static variables: 64 KB in all */
#define SAMPLE_LOG2E 1.4426950408889634f

struct SampleParts {     // [rows][P] each, P = ceil(V / 64)
  float* m;
  float* s;
  int P;
  const float* temps;    // [rows]
};

static inline size_t sample_parts_bytes(int64_t rows, int V) {
  return (size_t)rows * (size_t)((V + SAMPLE_PART_COLS - 1) / SAMPLE_PART_COLS) * 8;
}

// log2(e) / T, or 0 for a greedy row
__device__ __forceinline__ float sample_exp_scale(float temp) { return temp != 0.0f ? SAMPLE_LOG2E / temp : 0.0f; }

// One 16-column tile: v = this lane's 4 consecutive logits.  All four lanes of the tile return the same (m, s).
// Every lane of the wavefront must call it (the lanes trade registers).  The roundings are spelled out (fma, add)
// so that both producers form bit-identical sums.
__device__ __forceinline__ void sample_tile16(const float (&v)[4], float c2, float& m, float& s) {
  m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
  float a, b;
  rows_pair16(m, a, b);          // over the tile's four lanes: rows l^16, then halves l^32 (VALU lane swaps, no LDS)
  m = fmaxf(a, b);
  rows_pair32(m, a, b);
  m = fmaxf(a, b);
  const float off = (m == -INFINITY) ? 0.f : -m * c2;   // a tile of -inf only: exp2(-inf) = 0 instead of exp2(nan)
  float t = __fadd_rn(__fadd_rn(__builtin_amdgcn_exp2f(__fmaf_rn(v[0], c2, off)), __builtin_amdgcn_exp2f(__fmaf_rn(v[1], c2, off))),
                      __fadd_rn(__builtin_amdgcn_exp2f(__fmaf_rn(v[2], c2, off)), __builtin_amdgcn_exp2f(__fmaf_rn(v[3], c2, off))));
  rows_pair16(t, a, b);
  t = __fadd_rn(a, b);
  rows_pair32(t, a, b);
  t = __fadd_rn(a, b);
  s = (c2 != 0.0f) ? t : 0.f;
}

// Join the four tiles of a part, in column order: m = max m_t, s = ((w_0 + w_1) + w_2) + w_3 with
// w_t = s_t exp((m_t - m) / T).  The two halves below let a producer that holds the tiles in four different lanes
// compute one weight per lane.
__device__ __forceinline__ float sample_join_max(const float (&tm)[4]) {
  return fmaxf(fmaxf(tm[0], tm[1]), fmaxf(tm[2], tm[3]));
}
__device__ __forceinline__ float sample_join_weight(float tm, float ts, float m, float c2) {
  const float off = (m == -INFINITY) ? 0.f : -m * c2;
  return (ts > 0.f) ? __fmul_rn(ts, __builtin_amdgcn_exp2f(__fmaf_rn(tm, c2, off))) : 0.f;
}
__device__ __forceinline__ float sample_join_sum(const float (&w)[4]) {
  return __fadd_rn(__fadd_rn(__fadd_rn(w[0], w[1]), w[2]), w[3]);
}
__device__ __forceinline__ void sample_join4(const float (&tm)[4], const float (&ts)[4], float c2, float& m, float& s) {
  m = sample_join_max(tm);
  float w[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) w[t] = sample_join_weight(tm[t], ts[t], m, c2);
  s = sample_join_sum(w);
}
