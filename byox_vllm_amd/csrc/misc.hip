// Small memory-bound kernels of the decode step: SiLU-and-multiply, embedding
// row gather, Gumbel-max / argmax sampler, graph-replay input staging; plus the
// library's version and error-string entry points.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";
void nv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* nv_last_error(void) { return g_err; }
extern "C" int nv_version(void) { return 100; }

// ---------------------------------------------------------------- silu * up
template <typename T>
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out,
                                                       int64_t T_, int I) {
  const int nvec = I >> 3;
  const int64_t total = T_ * nvec;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = v / nvec;
    const int c = (int)(v - t * nvec);
    const u32x4 g = reinterpret_cast<const u32x4*>(x + t * 2 * I)[c];
    const u32x4 u = reinterpret_cast<const u32x4*>(x + t * 2 * I + I)[c];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g0 = lo_f32<T>(g[j]), g1 = hi_f32<T>(g[j]);
      // silu rounds to the model dtype, then the product rounds again (eager semantics)
      const float s0 = bits_to_f32<T>(f32_to_bits<T>(g0 / (1.0f + expf(-g0))));
      const float s1 = bits_to_f32<T>(f32_to_bits<T>(g1 / (1.0f + expf(-g1))));
      o[j] = pack2<T>(s0 * lo_f32<T>(u[j]), s1 * hi_f32<T>(u[j]));
    }
    reinterpret_cast<u32x4*>(out + t * I)[c] = o;
  }
}

extern "C" int nv_silu_mul(const void* x, void* out, int64_t T, int I, int dtype, nv_stream_t stream) {
  NV_CHECK_ARG(x && out, "nv_silu_mul: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_silu_mul: bad dtype %d", dtype);
  NV_CHECK_ARG(I > 0 && I % 8 == 0, "nv_silu_mul: I=%d must be a multiple of 8", I);
  NV_CHECK_ARG(((uintptr_t)x | (uintptr_t)out) % 16 == 0, "nv_silu_mul: pointers must be 16-byte aligned");
  if (T == 0) return NV_OK;
  const int64_t total = T * (I / 8);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == NV_BF16)
    silu_mul_kernel<bf16_t><<<dim3((unsigned)blocks), dim3(256), 0, s>>>((const uint16_t*)x, (uint16_t*)out, T, I);
  else
    silu_mul_kernel<f16_t><<<dim3((unsigned)blocks), dim3(256), 0, s>>>((const uint16_t*)x, (uint16_t*)out, T, I);
  NV_CHECK_LAUNCH();
  return NV_OK;
}

// ---------------------------------------------------------------- embedding
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, const uint16_t* __restrict__ w,
                                                    uint16_t* __restrict__ out, int64_t T_, int hidden,
                                                    int64_t vocab_start, int64_t vocab_local) {
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T_) return;
  const int lane = threadIdx.x & 63;
  const int64_t id = ids[t] - vocab_start;
  const bool ok = id >= 0 && id < vocab_local;
  const u32x4* src = reinterpret_cast<const u32x4*>(w + (ok ? id : 0) * hidden);
  u32x4* dst = reinterpret_cast<u32x4*>(out + t * hidden);
  const u32x4 zero = {0, 0, 0, 0};
  for (int i = lane; i < (hidden >> 3); i += 64) dst[i] = ok ? src[i] : zero;
}

extern "C" int nv_embed(const int64_t* ids, const void* weight, void* out, int64_t T, int hidden, int64_t vocab_start,
                        int64_t vocab_local, int dtype, nv_stream_t stream) {
  NV_CHECK_ARG(ids && weight && out, "nv_embed: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_embed: bad dtype %d", dtype);
  NV_CHECK_ARG(hidden > 0 && hidden % 8 == 0, "nv_embed: hidden=%d must be a multiple of 8", hidden);
  NV_CHECK_ARG(((uintptr_t)weight | (uintptr_t)out) % 16 == 0, "nv_embed: pointers must be 16-byte aligned");
  if (T == 0) return NV_OK;
  embed_kernel<<<dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
      ids, (const uint16_t*)weight, (uint16_t*)out, T, hidden, vocab_start, vocab_local);
  NV_CHECK_LAUNCH();
  return NV_OK;
}

// ---------------------------------------------------------------- sampler
// argmax_v softmax(l/T)_v / e_v  ==  argmax_v (l_v / T - log e_v): the softmax normaliser and the
// row maximum are common to the row, and log is monotonic, so one pass over the logits suffices.
// Stage 1: grid (splits, B), each workgroup scans a slice of the row with 16-byte loads and writes
// its best (score, index); stage 2: one wavefront per row picks the winner (lowest index on ties).
__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser: one per (launch, row)
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
// Per-element random bits: a 32-bit avalanche hash (two multiplies; 64-bit multiplies are several quarter-rate
// instructions each on this chip and made the sampler VALU-bound) of the index keyed by the row's 64-bit key.
__device__ __forceinline__ uint32_t mix32(uint32_t x, uint32_t k0, uint32_t k1) {
  x ^= k0;
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x ^ k1;
}

struct Best {
  float v;
  int i;
};
__device__ __forceinline__ Best better(Best a, Best b) {
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ Best wave_best(Best best) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best other = {__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)};
    best = better(best, other);
  }
  return best;
}

#define SAMPLE_SPLITS 8
#define LOG2E_F 1.4426950408889634f

template <typename T>
__device__ __forceinline__ float sample_score(float l, int b, int v, int V, float inv_temp, bool greedy,
                                              const float* __restrict__ noise, uint64_t seed) {
  if (greedy) return l;
  float e;
  if (noise) {
    e = noise[(int64_t)b * V + v];
  } else {  // Exp(1) = -log(U), U uniform in (0, 1]
    const uint32_t r = mix32((uint32_t)v, (uint32_t)seed, (uint32_t)(seed >> 32));
    e = -__logf(((float)(r >> 8) + 1.0f) * (1.0f / 16777216.0f));
  }
  return l * inv_temp - __logf(fmaxf(e, 1e-10f));
}

template <typename T>
__global__ __launch_bounds__(256) void sample_partial_kernel(const uint16_t* __restrict__ logits, int64_t row_stride,
                                                             const float* __restrict__ temps,
                                                             const float* __restrict__ noise, float* __restrict__ pv,
                                                             int* __restrict__ pi, int V, uint64_t seed,
                                                             const uint64_t* __restrict__ seed_offset) {
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (seed_offset) seed += seed_offset[0] * 0x9e3779b97f4a7c15ull;   // per-replay stream inside a captured graph
  seed = mix64(seed ^ ((uint64_t)b << 32));                          // this row's key
  const uint16_t* row = logits + (int64_t)b * row_stride;
  const float temp = temps[b];
  const bool greedy = temp == 0.0f;
  const float inv_temp = greedy ? 1.0f : 1.0f / temp;
  // slice of the row in 8-element vectors (rows are 16-byte aligned when V % 8 == 0 and the stride allows)
  const bool vec_ok = (V % 8 == 0) && (row_stride % 8 == 0) && (((uintptr_t)logits) % 16 == 0);
  Best best = {-INFINITY, 0x7fffffff};
  if (vec_ok) {
    const int nvec = V >> 3;
    const int per = (nvec + SAMPLE_SPLITS - 1) / SAMPLE_SPLITS;
    const int v0 = blockIdx.x * per;
    int v1 = v0 + per;
    if (v1 > nvec) v1 = nvec;
    for (int i = v0 + tid; i < v1; i += 256) {
      const u32x4 t = reinterpret_cast<const u32x4*>(row)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int v = i * 8 + 2 * j;
        best = better(best, Best{sample_score<T>(lo_f32<T>(t[j]), b, v, V, inv_temp, greedy, noise, seed), v});
        best = better(best, Best{sample_score<T>(hi_f32<T>(t[j]), b, v + 1, V, inv_temp, greedy, noise, seed), v + 1});
      }
    }
  } else {
    const int per = (V + SAMPLE_SPLITS - 1) / SAMPLE_SPLITS;
    const int v0 = blockIdx.x * per;
    int v1 = v0 + per;
    if (v1 > V) v1 = V;
    for (int v = v0 + tid; v < v1; v += 256)
      best = better(best, Best{sample_score<T>(bits_to_f32<T>(row[v]), b, v, V, inv_temp, greedy, noise, seed), v});
  }
  best = wave_best(best);
  if (lane == 0) {
    s_val[wid] = best.v;
    s_idx[wid] = best.i;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) best = better(best, Best{s_val[w], s_idx[w]});
    pv[b * SAMPLE_SPLITS + blockIdx.x] = best.v;
    pi[b * SAMPLE_SPLITS + blockIdx.x] = best.i;
  }
}

__global__ __launch_bounds__(256) void sample_final_kernel(const float* __restrict__ pv, const int* __restrict__ pi,
                                                           int64_t* __restrict__ tokens, int B) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  Best best = {-INFINITY, 0x7fffffff};
  if (lane < SAMPLE_SPLITS) best = Best{pv[b * SAMPLE_SPLITS + lane], pi[b * SAMPLE_SPLITS + lane]};
  best = wave_best(best);
  if (lane == 0) tokens[b] = best.i;
}

// ---- sampling without injected noise: inverse CDF --------------------------------------------------------------
// Drawing a token from softmax(l / T) by the Gumbel / exponential-noise argmax costs a hash and two logarithms per
// logit (VALU-bound: 30 us for 128 x 152k logits).  With the noise drawn in-kernel anyway, the same distribution comes
// cheaper from the inverse CDF with ONE uniform per row: stage 1 (grid (splits, B)) gives every slice its maximum m_k
// (also the greedy candidate) and s_k = sum exp((l - m_k) / T); stage 2 (one workgroup per row) picks the slice by
// the weights s_k exp((m_k - M) / T), then walks that slice's prefix sums to the token.  One exp per logit.
template <typename T>
__device__ __forceinline__ void load8(const uint16_t* __restrict__ row, int i, int V, bool vec_ok, float (&x)[8]) {
  if (vec_ok) {
    const u32x4 t = reinterpret_cast<const u32x4*>(row)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[2 * j] = lo_f32<T>(t[j]);
      x[2 * j + 1] = hi_f32<T>(t[j]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (8 * i + j < V) ? bits_to_f32<T>(row[8 * i + j]) : -INFINITY;
  }
}

// 2^x by the hardware instruction alone (arguments here are <= 0 and results only feed sums: no denormal care)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {   // 256 threads, result in every thread
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

#define SAMPLE_QUARTERS 4   // wavefronts per slice: each owns a contiguous quarter and reports its own mass

template <typename T>
__global__ __launch_bounds__(256) void sample_cdf_scan_kernel(const uint16_t* __restrict__ logits, int64_t row_stride,
                                                              const float* __restrict__ temps, float* __restrict__ pv,
                                                              int* __restrict__ pi, float* __restrict__ ps, int V) {
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint16_t* row = logits + (int64_t)b * row_stride;
  const float temp = temps[b];
  const bool vec_ok = (V % 8 == 0) && (row_stride % 8 == 0) && (((uintptr_t)logits) % 16 == 0);
  const int nvec = (V + 7) >> 3;
  const int per = (nvec + SAMPLE_SPLITS - 1) / SAMPLE_SPLITS;
  const int v0 = blockIdx.x * per;
  const int v1 = v0 + per < nvec ? v0 + per : nvec;
  const int quarter = (per + SAMPLE_QUARTERS - 1) / SAMPLE_QUARTERS;
  const int q0 = v0 + wid * quarter, q1 = (q0 + quarter < v1) ? q0 + quarter : v1;
  // one pass: every thread keeps its best (value, index) and, for sampled rows, the sum of exp((x - best) / T)
  // relative to its own running maximum (rescaled when the maximum moves)
  const bool sampled = temp != 0.0f;
  const float c = sampled ? LOG2E_F / temp : 0.f;
  Best best = {-INFINITY, 0x7fffffff};
  float sum = 0.f;
  for (int i = q0 + lane; i < q1; i += 64) {
    float x[8];
    load8<T>(row, i, V, vec_ok, x);
    Best vb = {x[0], 8 * i};
#pragma unroll
    for (int j = 1; j < 8; ++j) vb = better(vb, Best{x[j], 8 * i + j});
    if (sampled && vb.v > -INFINITY) {
      const float m_new = fmaxf(best.v, vb.v);
      float add = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) add += fast_exp2((x[j] - m_new) * c);
      sum = sum * fast_exp2((best.v - m_new) * c) + add;   // best.v == -inf: sum is 0 and the factor exp2(-inf) = 0
    }
    best = better(best, vb);
  }
  const float my_max = best.v;
  best = wave_best(best);
  if (lane == 0) {
    s_val[wid] = best.v;
    s_idx[wid] = best.i;
  }
  __syncthreads();
  for (int w = 0; w < 4; ++w) best = better(best, Best{s_val[w], s_idx[w]});   // the slice's best, in every thread
  if (sampled && my_max > -INFINITY) sum *= fast_exp2((my_max - best.v) * c);
  else sum = 0.f;
  sum = wave_sum(sum);                     // this quarter's mass relative to the slice maximum
  const int slice = b * SAMPLE_SPLITS + blockIdx.x;
  if (lane == 0) ps[slice * SAMPLE_QUARTERS + wid] = sum;
  if (tid == 0) {
    pv[slice] = best.v;
    pi[slice] = best.i;
  }
}

// One wavefront per row: choose (slice, quarter) by the reported masses, then walk that quarter 64 vectors at a time
// (lane l takes vector base + l: coalesced; four batches of loads in flight) with a prefix scan over the lanes.
template <typename T>
__global__ __launch_bounds__(64) void sample_cdf_pick_kernel(const uint16_t* __restrict__ logits, int64_t row_stride,
                                                             const float* __restrict__ temps,
                                                             const float* __restrict__ pv, const int* __restrict__ pi,
                                                             const float* __restrict__ ps, int64_t* __restrict__ tokens,
                                                             int V, uint64_t seed,
                                                             const uint64_t* __restrict__ seed_offset) {
  constexpr int NP = SAMPLE_SPLITS * SAMPLE_QUARTERS;   // 32 parts per row: one lane each
  const int b = blockIdx.x, lane = threadIdx.x;
  const float temp = temps[b];
  const int part_k = (lane < NP ? lane : NP - 1) / SAMPLE_QUARTERS;
  const float m_part = pv[b * SAMPLE_SPLITS + part_k];
  const int i_part = pi[b * SAMPLE_SPLITS + part_k];
  const float s_part = lane < NP ? ps[b * NP + lane] : 0.f;
  const Best top = wave_best(Best{m_part, i_part});
  if (temp == 0.0f || !(top.v > -INFINITY)) {   // greedy (or a row without a finite logit)
    if (lane == 0) tokens[b] = top.i;
    return;
  }
  const float c = LOG2E_F / temp;
  if (seed_offset) seed += seed_offset[0] * 0x9e3779b97f4a7c15ull;   // per-replay stream inside a captured graph
  const uint64_t key = mix64(seed ^ ((uint64_t)b << 32));
  const float u = ((float)(uint32_t)(key >> 40) + 0.5f) * (1.0f / 16777216.0f);   // uniform in (0, 1)
  const float weight = s_part > 0.f ? s_part * fast_exp2((m_part - top.v) * c) : 0.f;
  float cum = weight;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float up = __shfl_up(cum, o, 64);
    if (lane >= o) cum += up;
  }
  const float target = u * __shfl(cum, 63, 64);
  // the first part with mass whose running total reaches the target; rounding may leave none: take the last with mass
  const unsigned long long reach = __ballot(weight > 0.f && cum >= target);
  const unsigned long long any = __ballot(weight > 0.f);
  const int sel = reach ? __ffsll((long long)reach) - 1 : 63 - __clzll((long long)any);
  const float m_sel = __shfl(m_part, sel, 64);
  const float resid = (target - (__shfl(cum, sel, 64) - __shfl(weight, sel, 64))) / fast_exp2((m_sel - top.v) * c);
  int token = __shfl(i_part, sel, 64);          // fall-back: the selected slice's best token

  const uint16_t* row = logits + (int64_t)b * row_stride;
  const bool vec_ok = (V % 8 == 0) && (row_stride % 8 == 0) && (((uintptr_t)logits) % 16 == 0);
  const int nvec = (V + 7) >> 3;
  const int per = (nvec + SAMPLE_SPLITS - 1) / SAMPLE_SPLITS;
  const int quarter = (per + SAMPLE_QUARTERS - 1) / SAMPLE_QUARTERS;
  const int v0 = (sel / SAMPLE_QUARTERS) * per;
  const int v1 = v0 + per < nvec ? v0 + per : nvec;
  const int q0 = v0 + (sel % SAMPLE_QUARTERS) * quarter, q1 = (q0 + quarter < v1) ? q0 + quarter : v1;
  float running = 0.f;
  bool done = false;
  for (int base = q0; base < q1 && !done; base += 256) {
    float p[4][8];
#pragma unroll
    for (int u4 = 0; u4 < 4; ++u4) {               // four batches of loads in flight
      const int i = base + 64 * u4 + lane;
      float x[8];
      if (i < q1) load8<T>(row, i, V, vec_ok, x);
#pragma unroll
      for (int j = 0; j < 8; ++j) p[u4][j] = (i < q1) ? fast_exp2((x[j] - m_sel) * c) : 0.f;
    }
#pragma unroll
    for (int u4 = 0; u4 < 4; ++u4) {
      if (done) break;
      const int i = base + 64 * u4 + lane;
      float part = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) part += p[u4][j];
      float incl = part;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      const float excl = running + incl - part;
      const bool mine = part > 0.f && excl < resid && resid <= running + incl;
      if (mine) {
        float acc = excl;
        int pick = -1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc += p[u4][j];
          if (pick < 0 && p[u4][j] > 0.f && acc >= resid) pick = 8 * i + j;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)                 // rounding: the last token of the vector that carries mass
          if (pick < 0 && p[u4][7 - j] > 0.f) pick = 8 * i + 7 - j;
        token = pick;
      }
      const unsigned long long found = __ballot(mine);
      if (found) {
        token = __shfl(token, __ffsll((long long)found) - 1, 64);
        done = true;
      }
      running += __shfl(incl, 63, 64);
    }
  }
  if (lane == 0) tokens[b] = token;
}

extern "C" size_t nv_sample_workspace_bytes(int B) { return (size_t)B * SAMPLE_SPLITS * (8 + 4 * SAMPLE_QUARTERS); }

extern "C" int nv_sample(const void* logits, int64_t logits_row_stride, const float* temperatures, const float* noise,
                         int64_t* tokens, void* workspace, size_t workspace_bytes, int B, int V, uint64_t seed,
                         const uint64_t* seed_offset, int dtype, nv_stream_t stream) {
  NV_CHECK_ARG(logits && temperatures && tokens, "nv_sample: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_sample: bad dtype %d", dtype);
  NV_CHECK_ARG(V > 0 && B >= 0 && B <= 65535, "nv_sample: bad shape B=%d V=%d", B, V);
  if (B == 0) return NV_OK;
  NV_CHECK_ARG(workspace && workspace_bytes >= nv_sample_workspace_bytes(B) && (uintptr_t)workspace % 4 == 0,
               "nv_sample: workspace %zu bytes < %zu needed", workspace_bytes, nv_sample_workspace_bytes(B));
  hipStream_t s = (hipStream_t)stream;
  float* pv = (float*)workspace;
  int* pi = (int*)(pv + (size_t)B * SAMPLE_SPLITS);
  dim3 grid(SAMPLE_SPLITS, B);
  if (!noise) {   // greedy rows and in-kernel randomness: maxima + partition sums, then one inverse-CDF draw per row
    float* ps = (float*)(pi + (size_t)B * SAMPLE_SPLITS);
    if (dtype == NV_BF16) {
      sample_cdf_scan_kernel<bf16_t><<<grid, dim3(256), 0, s>>>((const uint16_t*)logits, logits_row_stride, temperatures, pv, pi, ps, V);
      NV_CHECK_LAUNCH();
      sample_cdf_pick_kernel<bf16_t><<<dim3(B), dim3(64), 0, s>>>((const uint16_t*)logits, logits_row_stride, temperatures,
                                                                   pv, pi, ps, tokens, V, seed, seed_offset);
    } else {
      sample_cdf_scan_kernel<f16_t><<<grid, dim3(256), 0, s>>>((const uint16_t*)logits, logits_row_stride, temperatures, pv, pi, ps, V);
      NV_CHECK_LAUNCH();
      sample_cdf_pick_kernel<f16_t><<<dim3(B), dim3(64), 0, s>>>((const uint16_t*)logits, logits_row_stride, temperatures,
                                                                  pv, pi, ps, tokens, V, seed, seed_offset);
    }
    NV_CHECK_LAUNCH();
    return NV_OK;
  }
  if (dtype == NV_BF16)
    sample_partial_kernel<bf16_t><<<grid, dim3(256), 0, s>>>((const uint16_t*)logits, logits_row_stride, temperatures,
                                                             noise, pv, pi, V, seed, seed_offset);
  else
    sample_partial_kernel<f16_t><<<grid, dim3(256), 0, s>>>((const uint16_t*)logits, logits_row_stride, temperatures,
                                                            noise, pv, pi, V, seed, seed_offset);
  NV_CHECK_LAUNCH();
  sample_final_kernel<<<dim3((B + 3) / 4), dim3(256), 0, s>>>(pv, pi, tokens, B);
  NV_CHECK_LAUNCH();
  return NV_OK;
}

// ---------------------------------------------------------------- graph input staging
__global__ __launch_bounds__(256) void stage_kernel(const int64_t* ids, const int64_t* pos, const int32_t* slots,
                                                    const int32_t* ctx, const int32_t* bt, int bt_cols, int64_t* g_ids,
                                                    int64_t* g_pos, int32_t* g_slots, int32_t* g_ctx, int32_t* g_bt,
                                                    int g_bt_cols, int bs, int padded) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
  for (int i = tid; i < padded; i += nthr) {
    const bool live = i < bs;
    g_ids[i] = live ? ids[i] : 0;
    g_pos[i] = live ? pos[i] : 0;
    g_slots[i] = live ? slots[i] : -1;
    g_ctx[i] = live ? ctx[i] : 0;
  }
  for (int i = tid; i < bs * bt_cols; i += nthr) {
    const int r = i / bt_cols, c = i - r * bt_cols;
    g_bt[r * g_bt_cols + c] = bt[i];
  }
}

extern "C" int nv_stage_decode_inputs(const int64_t* ids, const int64_t* positions, const int32_t* slots,
                                      const int32_t* context_lens, const int32_t* block_tables, int bt_cols,
                                      int64_t* g_ids, int64_t* g_positions, int32_t* g_slots, int32_t* g_context_lens,
                                      int32_t* g_block_tables, int g_bt_cols, int bs, int padded_bs,
                                      nv_stream_t stream) {
  NV_CHECK_ARG(ids && positions && slots && context_lens && block_tables, "nv_stage_decode_inputs: null input");
  NV_CHECK_ARG(g_ids && g_positions && g_slots && g_context_lens && g_block_tables, "nv_stage_decode_inputs: null buffer");
  NV_CHECK_ARG(bs >= 0 && padded_bs >= bs && bt_cols >= 0 && bt_cols <= g_bt_cols,
               "nv_stage_decode_inputs: bs=%d padded=%d bt_cols=%d g_bt_cols=%d", bs, padded_bs, bt_cols, g_bt_cols);
  if (padded_bs == 0) return NV_OK;
  int work = bs * bt_cols > padded_bs ? bs * bt_cols : padded_bs;
  int blocks = (work + 255) / 256;
  if (blocks > 64) blocks = 64;
  stage_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(ids, positions, slots, context_lens, block_tables,
                                                                    bt_cols, g_ids, g_positions, g_slots,
                                                                    g_context_lens, g_block_tables, g_bt_cols, bs,
                                                                    padded_bs);
  NV_CHECK_LAUNCH();
  return NV_OK;
}

// Packed form of the staging above: ONE host->device copy carries the whole step
// [ids | positions | slots | context_lens | temperature bits | block_tables], all as 32-bit words.
// Advance form (header word 1 carries NV_STAGE_ADVANCE, prev_tokens given): the step continues the previous one for the
// same rows, so nothing but the header is read from the host's buffer: every live row takes the token sampled for it
// in the previous step, moves one position on, and derives its cache slot from the block table already on the device
// (the bookkeeping of nanovllm/engine/model_runner.py:206-226, one step of it, without the host's lists).
__global__ __launch_bounds__(256) void stage_packed_kernel(const int32_t* __restrict__ p, int bs, int padded, int bt_cols,
                                                           int64_t* g_ids, int64_t* g_pos, int32_t* g_slots,
                                                           int32_t* g_ctx, int32_t* g_bt, int g_bt_cols, float* g_temps,
                                                           uint64_t* step_counter, const int64_t* prev_tokens,
                                                           int block_size) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
  if (bs < 0) {  // header form: the batch size and table width of this step travel in the buffer itself
    bs = p[0];
    bt_cols = p[1];
    p += 2;
    const bool advance = prev_tokens != nullptr && bt_cols >= 0 && (bt_cols & NV_STAGE_ADVANCE);
    bs = bs < 0 ? 0 : (bs > padded ? padded : bs);
    if (advance) {
      int cols = bt_cols & ~NV_STAGE_ADVANCE;   // the table width of the last full staging: columns past it are stale
      cols = cols > g_bt_cols ? g_bt_cols : cols;
      for (int i = tid; i < bs; i += nthr) {
        const int64_t pos = g_pos[i] + 1;
        const int col = (int)(pos / block_size);
        const int32_t blk = col < cols ? g_bt[(int64_t)i * g_bt_cols + col] : -1;
        g_ids[i] = prev_tokens[i];
        g_pos[i] = pos;
        g_ctx[i] += 1;
        g_slots[i] = blk >= 0 ? blk * block_size + (int32_t)(pos - (int64_t)col * block_size) : -1;
      }
      if (tid == 0 && step_counter) step_counter[0] += 1;
      return;
    }
    bt_cols = bt_cols < 0 ? 0 : (bt_cols > g_bt_cols ? g_bt_cols : bt_cols);
  }
  for (int i = tid; i < padded; i += nthr) {
    const bool live = i < bs;
    g_ids[i] = live ? p[i] : 0;
    g_pos[i] = live ? p[bs + i] : 0;
    g_slots[i] = live ? p[2 * bs + i] : -1;
    g_ctx[i] = live ? p[3 * bs + i] : 0;
    if (g_temps) g_temps[i] = live ? __int_as_float(p[4 * bs + i]) : 1.0f;
  }
  const int32_t* bt = p + 5 * bs;
  for (int i = tid; i < bs * bt_cols; i += nthr) {
    const int r = i / bt_cols, c = i - r * bt_cols;
    g_bt[r * g_bt_cols + c] = bt[i];
  }
  if (tid == 0 && step_counter) step_counter[0] += 1;
}

static int stage_packed_go(const int32_t* packed, int bs, int padded_bs, int bt_cols, int64_t* g_ids,
                           int64_t* g_positions, int32_t* g_slots, int32_t* g_context_lens, int32_t* g_block_tables,
                           int g_bt_cols, float* g_temperatures, uint64_t* step_counter, const int64_t* prev_tokens,
                           int block_size, nv_stream_t stream);

extern "C" int nv_stage_decode_packed(const int32_t* packed, int bs, int padded_bs, int bt_cols, int64_t* g_ids,
                                      int64_t* g_positions, int32_t* g_slots, int32_t* g_context_lens,
                                      int32_t* g_block_tables, int g_bt_cols, float* g_temperatures,
                                      uint64_t* step_counter, nv_stream_t stream) {
  return stage_packed_go(packed, bs, padded_bs, bt_cols, g_ids, g_positions, g_slots, g_context_lens, g_block_tables,
                         g_bt_cols, g_temperatures, step_counter, nullptr, 0, stream);
}

extern "C" int nv_stage_decode_packed_advance(const int32_t* packed, int padded_bs, int64_t* g_ids, int64_t* g_positions,
                                              int32_t* g_slots, int32_t* g_context_lens, int32_t* g_block_tables,
                                              int g_bt_cols, float* g_temperatures, uint64_t* step_counter,
                                              const int64_t* prev_tokens, int block_size, nv_stream_t stream) {
  NV_CHECK_ARG(prev_tokens && block_size > 0, "nv_stage_decode_packed_advance: prev_tokens and block_size are required");
  return stage_packed_go(packed, -1, padded_bs, 0, g_ids, g_positions, g_slots, g_context_lens, g_block_tables,
                         g_bt_cols, g_temperatures, step_counter, prev_tokens, block_size, stream);
}

static int stage_packed_go(const int32_t* packed, int bs, int padded_bs, int bt_cols, int64_t* g_ids,
                           int64_t* g_positions, int32_t* g_slots, int32_t* g_context_lens, int32_t* g_block_tables,
                           int g_bt_cols, float* g_temperatures, uint64_t* step_counter, const int64_t* prev_tokens,
                           int block_size, nv_stream_t stream) {
  NV_CHECK_ARG(packed && g_ids && g_positions && g_slots && g_context_lens && g_block_tables,
               "nv_stage_decode_packed: null pointer");
  NV_CHECK_ARG(bs < 0 || (padded_bs >= bs && bt_cols >= 0 && bt_cols <= g_bt_cols),
               "nv_stage_decode_packed: bs=%d padded=%d bt_cols=%d g_bt_cols=%d", bs, padded_bs, bt_cols, g_bt_cols);
  NV_CHECK_ARG(padded_bs >= 0 && g_bt_cols >= 0, "nv_stage_decode_packed: padded=%d g_bt_cols=%d", padded_bs, g_bt_cols);
  if (padded_bs == 0) return NV_OK;
  if (bs < 0) bs = -1;
  // header form: the grid is sized for the largest step the static buffers can take
  const int rows = bs < 0 ? padded_bs : bs, cols = bs < 0 ? g_bt_cols : bt_cols;
  int work = rows * cols > padded_bs ? rows * cols : padded_bs;
  int blocks = (work + 255) / 256;
  if (blocks > 64) blocks = 64;
  stage_packed_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(
      packed, bs, padded_bs, bt_cols, g_ids, g_positions, g_slots, g_context_lens, g_block_tables, g_bt_cols,
      g_temperatures, step_counter, prev_tokens, block_size);
  NV_CHECK_LAUNCH();
  return NV_OK;
}
