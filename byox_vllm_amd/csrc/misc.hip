// Small memory-bound kernels of the decode step: SiLU-and-multiply, embedding
// row gather, Gumbel-max / argmax sampler, graph-replay input staging; plus the
// library's version and error-string entry points.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"
#include "sample_parts.h"

static thread_local char g_err[512] = "";
void nv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* nv_last_error(void) { return g_err; }
extern "C" int nv_version(void) { return 100; }

// Result-store policy (common.h out_store): read by the launch wrappers, so a captured launch keeps the policy that was
// in force at capture time.  One word, written by set-up code only.
static int g_result_policy = 0;
int nv_result_policy() { return g_result_policy; }
extern "C" int nv_set_result_policy(int write_through) {
  NV_CHECK_ARG(write_through == 0 || write_through == 1, "nv_set_result_policy: %d is neither 0 nor 1", write_through);
  g_result_policy = write_through;
  return NV_OK;
}
extern "C" int nv_get_result_policy(void) { return g_result_policy; }

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
// cheaper from the inverse CDF with ONE uniform per row: stage 1 describes every 64-column part of the row by its
// maximum m_k (also the greedy candidate) and its mass s_k = sum exp((l - m_k) / T) (sample_parts.h; the LM-head GEMM
// can deliver the same arrays from its epilogue, then stage 1 is not launched at all); stage 2 (one wavefront per
// row) picks the part by the weights s_k exp((m_k - M) / T), then the column inside it.  One exp per logit.

// 2^x by the hardware instruction alone (arguments here are <= 0 and results only feed sums: no denormal care)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Stage 1 over stored logits.  A wavefront takes 256 columns (four parts) of one row at a time: lane (c, g) holds
// columns 16 c + 4 g .. + 3 of them, which is the register layout the GEMM epilogue has.  The four tiles of a part sit
// in four neighbouring lanes (a quad), so the join reads them with quad broadcasts.  Eight such groups are loaded
// before the first is reduced; groups past the row's end are loaded from the last group and not stored.
template <int Q>
__device__ __forceinline__ float quad_bcast(float x) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), Q * 0x55, 0xf, 0xf, false));
}

#define SAMPLE_SCAN_DEPTH 8
template <typename T>
__global__ __launch_bounds__(256) void sample_parts_scan_kernel(const uint16_t* __restrict__ logits, int64_t row_stride,
                                                                SampleParts sp, int V) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wid = threadIdx.x >> 6;
  const uint16_t* row = logits + (int64_t)b * row_stride;
  const float c2 = sample_exp_scale(sp.temps[b]);
  const bool vec_ok = (row_stride % 4 == 0) && (((uintptr_t)logits) % 8 == 0);
  const int groups = (sp.P + 3) >> 2;                                // 256-column groups in the row
  // wavefront w of workgroup x takes groups (4 x + w) * DEPTH .. + DEPTH - 1, then the same 4 * gridDim.x * DEPTH further
  for (int g0 = (blockIdx.x * 4 + wid) * SAMPLE_SCAN_DEPTH; g0 < groups; g0 += 4 * gridDim.x * SAMPLE_SCAN_DEPTH) {
    float v[SAMPLE_SCAN_DEPTH][4];
#pragma unroll
    for (int u = 0; u < SAMPLE_SCAN_DEPTH; ++u) {
      const int gi = min(g0 + u, groups - 1);
      const int col = 256 * gi + 16 * c + 4 * g;
      if (vec_ok && col + 4 <= V) {
        const u32x2 t = *reinterpret_cast<const u32x2*>(row + col);
        v[u][0] = lo_f32<T>(t[0]); v[u][1] = hi_f32<T>(t[0]); v[u][2] = lo_f32<T>(t[1]); v[u][3] = hi_f32<T>(t[1]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = (col + e < V) ? bits_to_f32<T>(row[col + e]) : -INFINITY;
      }
    }
#pragma unroll
    for (int u = 0; u < SAMPLE_SCAN_DEPTH; ++u) {
      float tm_own, ts_own;
      sample_tile16(v[u], c2, tm_own, ts_own);
      const float tm[4] = {quad_bcast<0>(tm_own), quad_bcast<1>(tm_own), quad_bcast<2>(tm_own), quad_bcast<3>(tm_own)};
      const float m = sample_join_max(tm);
      const float w_own = sample_join_weight(tm_own, ts_own, m, c2);      // this lane's tile; the quad holds all four
      const float w[4] = {quad_bcast<0>(w_own), quad_bcast<1>(w_own), quad_bcast<2>(w_own), quad_bcast<3>(w_own)};
      const float s = sample_join_sum(w);
      const int part = 4 * (g0 + u) + (c >> 2);
      if (g == 0 && (c & 3) == 0 && part < sp.P) {
        const int64_t o = (int64_t)b * sp.P + part;
        sp.m[o] = m;
        sp.s[o] = s;
      }
    }
  }
}

// Stage 2: one workgroup of four wavefronts per row.  The row's part maxima and masses go to LDS with coalesced
// loads (all in flight at once); thread t then owns the contiguous run of parts [t per, (t + 1) per): the runs'
// masses, a scan over the threads for the run that holds the target, and - on the first wavefront alone - a scan over
// that run's parts and one over the 64 columns of the chosen part (one column per lane).  A greedy row goes straight
// to the columns of the first part that holds the row's maximum.
template <typename T>
__global__ __launch_bounds__(256) void sample_parts_pick_kernel(const uint16_t* __restrict__ logits, int64_t row_stride,
                                                                SampleParts sp, int64_t* __restrict__ tokens, int V,
                                                                uint64_t seed, const uint64_t* __restrict__ seed_offset) {
  extern __shared__ float pick_lds[];            // [P] maxima, [P] weights
  __shared__ float s_top_v[4], s_total[4], s_below;
  __shared__ int s_top_i[4];
  __shared__ unsigned long long s_reach[4], s_any[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, P = sp.P;
  const float temp = sp.temps[b];
  const float* pm = sp.m + (int64_t)b * P;
  const float* ps = sp.s + (int64_t)b * P;
  const uint16_t* row = logits + (int64_t)b * row_stride;
  float* lm = pick_lds;
  float* lw = pick_lds + P;
  Best top = {-INFINITY, 0x7fffffff};            // (maximum, part): parts are in column order, so the lowest part wins ties
  // The first 12 x 256 parts (a 151936-entry vocabulary has 2374) are fetched with all loads in flight at once, on
  // clamped indices so that no branch splits them: the kernel is one workgroup per row and this read is a third of it.
  {
    constexpr int DEPTH = 12;
    float m[DEPTH], w[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const int k = tid + 256 * j, kk = k < P ? k : P - 1;
      m[j] = pm[kk];
      w[j] = ps[kk];
    }
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const int k = tid + 256 * j;
      if (k < P) {
        lm[k] = m[j];
        lw[k] = w[j];
        top = better(top, Best{m[j], k});
      }
    }
    for (int k = tid + 256 * DEPTH; k < P; k += 256) {
      const float mk = pm[k];
      lm[k] = mk;
      lw[k] = ps[k];
      top = better(top, Best{mk, k});
    }
  }
  top = wave_best(top);
  if (lane == 0) {
    s_top_v[wid] = top.v;
    s_top_i[wid] = top.i;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; ++w) top = better(top, Best{s_top_v[w], s_top_i[w]});
  // the column of a part's maximum: the first of its 64 logits that equals it
  auto column_of_max = [&](int part, float m) {
    const int col = part * SAMPLE_PART_COLS + lane;
    const float x = col < V ? bits_to_f32<T>(row[col]) : -INFINITY;
    const unsigned long long hit = __ballot(x == m);
    return hit ? part * SAMPLE_PART_COLS + __ffsll((long long)hit) - 1 : 0;
  };
  if (temp == 0.0f || !(top.v > -INFINITY)) {    // greedy (or a row without a finite logit)
    if (wid == 0) {
      const int tok = top.i < P ? column_of_max(top.i, top.v) : 0;
      if (lane == 0) tokens[b] = tok;
    }
    return;
  }
  const float c2 = SAMPLE_LOG2E / temp;
  if (seed_offset) seed += seed_offset[0] * 0x9e3779b97f4a7c15ull;   // per-replay stream inside a captured graph
  const uint64_t key = mix64(seed ^ ((uint64_t)b << 32));
  const float u = ((float)(uint32_t)(key >> 40) + 0.5f) * (1.0f / 16777216.0f);   // uniform in (0, 1)
  for (int k = tid; k < P; k += 256) {           // masses relative to the row maximum (a thread's own writes: no barrier)
    const float s = lw[k];
    lw[k] = s > 0.f ? s * fast_exp2((lm[k] - top.v) * c2) : 0.f;
  }
  __syncthreads();
  auto scan = [&](float x) {                     // inclusive prefix sum over the lanes of a wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float up = __shfl_up(x, o, 64);
      if (lane >= o) x += up;
    }
    return x;
  };
  // the first lane with mass whose running total reaches the target; rounding may leave none: the last with mass
  auto choose = [&](float w, float incl, float target) {
    const unsigned long long reach = __ballot(w > 0.f && incl >= target);
    const unsigned long long any = __ballot(w > 0.f);
    return reach ? __ffsll((long long)reach) - 1 : (any ? 63 - __clzll((long long)any) : 0);
  };
  const int per = (P + 255) >> 8;
  const int k0 = min(P, tid * per), k1 = min(P, k0 + per);
  float run = 0.f;
  for (int k = k0; k < k1; ++k) run += lw[k];
  float run_incl = scan(run);
  if (lane == 63) s_total[wid] = run_incl;
  __syncthreads();
  float before = 0.f, grand = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wid) before += s_total[w];
    grand += s_total[w];
  }
  run_incl += before;
  const float target = u * grand;
  {
    const unsigned long long reach = __ballot(run > 0.f && run_incl >= target);
    const unsigned long long any = __ballot(run > 0.f);
    if (lane == 0) {
      s_reach[wid] = reach;
      s_any[wid] = any;
    }
  }
  __syncthreads();
  int sel_thread = -1, last_mass = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (sel_thread < 0 && s_reach[w]) sel_thread = 64 * w + __ffsll((long long)s_reach[w]) - 1;
    if (s_any[w]) last_mass = 64 * w + 63 - __clzll((long long)s_any[w]);
  }
  if (sel_thread < 0) sel_thread = last_mass;
  if (tid == sel_thread) s_below = run_incl - run;   // mass before the chosen run
  __syncthreads();
  if (wid != 0) return;
  float below = s_below;
  // the part inside the run, 64 parts at a time
  const int r0 = min(P, sel_thread * per), r1 = min(P, r0 + per);
  int part = r0 < P ? r0 : P - 1;
  float part_below = below;
  for (int base = r0; base < r1; base += 64) {
    const int k = base + lane;
    const float w = k < r1 ? lw[k] : 0.f;
    const float incl = scan(w);
    const float total = __shfl(incl, 63, 64);
    const bool last = base + 64 >= r1;
    if (below + total >= target || last) {
      const int sel = choose(w, below + incl, target);
      part = base + sel;
      part_below = below + __shfl(incl, sel, 64) - __shfl(w, sel, 64);
      break;
    }
    below += total;
  }
  // the column inside the part: masses relative to the part's own maximum
  const float m_part = lm[part];
  const float resid = (target - part_below) / fast_exp2((m_part - top.v) * c2);
  const int col = part * SAMPLE_PART_COLS + lane;
  const float x = col < V ? bits_to_f32<T>(row[col]) : -INFINITY;
  const float p = x > -INFINITY ? fast_exp2((x - m_part) * c2) : 0.f;
  const float p_incl = scan(p);
  const int sel_col = choose(p, p_incl, resid);
  int token = part * SAMPLE_PART_COLS + sel_col;
  if (!__ballot(p > 0.f) || token >= V) token = column_of_max(top.i, top.v);   // cannot happen for a part with mass
  if (lane == 0) tokens[b] = token;
}

extern "C" size_t nv_sample_workspace_bytes(int B, int V) {
  const size_t noise_path = (size_t)B * SAMPLE_SPLITS * 8, parts = sample_parts_bytes(B, V);
  return noise_path > parts ? noise_path : parts;
}

// Views of a workspace of sample_parts_bytes(B, V) bytes.
static SampleParts sample_parts_of(void* workspace, int64_t B, int V, const float* temperatures) {
  SampleParts sp;
  sp.P = (V + SAMPLE_PART_COLS - 1) / SAMPLE_PART_COLS;
  sp.m = (float*)workspace;
  sp.s = sp.m + B * sp.P;
  sp.temps = temperatures;
  return sp;
}

// Stage 2 alone: the statistics are already in the workspace (written by the LM-head GEMM's epilogue).
int nv_sample_pick_launch(const void* logits, int64_t logits_row_stride, const float* temperatures, int64_t* tokens,
                          void* workspace, int B, int V, uint64_t seed, const uint64_t* seed_offset, int dtype,
                          hipStream_t s) {
  const SampleParts sp = sample_parts_of(workspace, B, V, temperatures);
  const size_t lds = (size_t)sp.P * 8;         // nv_sample / nv_lm_head_sample bound V so that this fits
  if (dtype == NV_BF16)
    sample_parts_pick_kernel<bf16_t><<<dim3(B), dim3(256), lds, s>>>((const uint16_t*)logits, logits_row_stride, sp, tokens, V, seed, seed_offset);
  else
    sample_parts_pick_kernel<f16_t><<<dim3(B), dim3(256), lds, s>>>((const uint16_t*)logits, logits_row_stride, sp, tokens, V, seed, seed_offset);
  NV_CHECK_LAUNCH();
  return NV_OK;
}

extern "C" int nv_sample(const void* logits, int64_t logits_row_stride, const float* temperatures, const float* noise,
                         int64_t* tokens, void* workspace, size_t workspace_bytes, int B, int V, uint64_t seed,
                         const uint64_t* seed_offset, int dtype, nv_stream_t stream) {
  NV_CHECK_ARG(logits && temperatures && tokens, "nv_sample: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_sample: bad dtype %d", dtype);
  NV_CHECK_ARG(V > 0 && V <= NV_SAMPLE_MAX_VOCAB && B >= 0 && B <= 65535, "nv_sample: bad shape B=%d V=%d", B, V);
  if (B == 0) return NV_OK;
  NV_CHECK_ARG(workspace && workspace_bytes >= nv_sample_workspace_bytes(B, V) && (uintptr_t)workspace % 4 == 0,
               "nv_sample: workspace %zu bytes < %zu needed", workspace_bytes, nv_sample_workspace_bytes(B, V));
  hipStream_t s = (hipStream_t)stream;
  float* pv = (float*)workspace;
  int* pi = (int*)(pv + (size_t)B * SAMPLE_SPLITS);
  dim3 grid(SAMPLE_SPLITS, B);
  if (!noise) {   // greedy rows and in-kernel randomness: part maxima + masses, then one inverse-CDF draw per row
    const SampleParts sp = sample_parts_of(workspace, B, V, temperatures);
    const int groups = (sp.P + 3) / 4;
    int slices = (groups + 4 * SAMPLE_SCAN_DEPTH - 1) / (4 * SAMPLE_SCAN_DEPTH);   // one round of loads per wavefront
    while (slices > 1 && (int64_t)slices * B > 4096) slices = (slices + 1) / 2;     // unless that floods the chip
    if (slices < 1) slices = 1;
    if (dtype == NV_BF16)
      sample_parts_scan_kernel<bf16_t><<<dim3(slices, B), dim3(256), 0, s>>>((const uint16_t*)logits, logits_row_stride, sp, V);
    else
      sample_parts_scan_kernel<f16_t><<<dim3(slices, B), dim3(256), 0, s>>>((const uint16_t*)logits, logits_row_stride, sp, V);
    NV_CHECK_LAUNCH();
    return nv_sample_pick_launch(logits, logits_row_stride, temperatures, tokens, workspace, B, V, seed, seed_offset,
                                 dtype, s);
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
    // NV_STAGE_STEP_BACK: the previous launch's step was thrown away by the host, its count is taken back first
    if (bt_cols >= 0 && (bt_cols & NV_STAGE_STEP_BACK)) {
      bt_cols &= ~NV_STAGE_STEP_BACK;
      if (tid == 0 && step_counter) step_counter[0] -= 1;
    }
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
