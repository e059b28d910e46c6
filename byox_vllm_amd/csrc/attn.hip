// Paged decode attention (flash-decoding) and causal variable-length prefill
// attention for gfx950, one code path for both.
//
// Work decomposition.  A wavefront owns 16 "query columns" that share one KV
// head: in decode the (up to 16) query heads of a GQA group for one sequence,
// in prefill 16 consecutive query tokens of one head.  It walks its keys in
// chunks of 32 tokens:
//
//   S^T[32 x 16] = K_chunk[32 x D] . Q^T[D x 16]      (2 x D/32 MFMA 16x16x32)
//   online softmax down each column (column = lane & 15, so the running max,
//   the row sum and the rescale factor are lane-local; one 2-step xor shuffle
//   joins the four 16-lane groups for the max)
//   O^T[D x 16] += V_chunk^T[D x 32] . P^T[32 x 16]   (D/16 MFMA 16x16x32)
//
// The S^T accumulator layout (column on the lane, four consecutive tokens in
// the four registers) is already the B-operand layout of the second product if
// k-slot (g, j) of that MFMA is taken to mean token 4g+j (j<4) / 16+4g+j-4
// (j>=4), so P never leaves registers.  The V^T operand is built from 16-byte
// row loads (lane (c, g) loads d = 8c..8c+7 of its 8 tokens) by an 8x8 16-bit
// register transpose, which makes output register (m, r) hold
// O[col][d = 8(4g + r) + m]: 8 consecutive d per (g, r) for one 16-byte store.
// K and V go HBM -> VGPR directly: every byte is used once by one wavefront.
//
// Decode: grid (splits, Hkv, B), 4 wavefronts per workgroup; wavefront w of
// split s takes chunks s*4+w, s*4+w+4*splits, ...; the 4 partial results are
// merged through LDS and, when splits > 1, per-split partials (fp32 O, max,
// sum) go to the workspace for a small combine kernel.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

#define LOG2E 1.4426950408889634f
// Query tiles (of 16) per wavefront in the LDS-fed prefill kernel: two halve the LDS reads per flop and still fit
// the 256 registers that two workgroups per CU leave each wavefront (measured against one tile: 8 x 1024 tokens
// 59 vs 65 us at D = 64, 165 vs 181 us at D = 128; four tiles need one workgroup per CU and lose).
#ifndef NV_PREFILL_QT64
#define NV_PREFILL_QT64 2
#endif
#ifndef NV_PREFILL_QT128
#define NV_PREFILL_QT128 2
#endif

template <int D> struct AttnGeom {
  static constexpr int KS = D / 32;   // k-steps of the S^T product
  static constexpr int NM = D / 16;   // MFMAs of the O^T product == d values per lane load
};

template <typename T, int D>
struct AttnState {
  float m;                                   // running max (log2 domain), same in the 4 lanes of a column
  float l;                                   // this lane's share of the row sum
  f32x4 acc[AttnGeom<D>::NM];
};

// Registers holding one 32-token chunk of K (as MFMA A fragments) and V (row pieces).
template <typename T, int D>
struct ChunkRegs {
  typename VecOf<T>::x8 kf[2][AttnGeom<D>::KS];
  uint32_t vv[8][AttnGeom<D>::NM / 2];       // token slot j: NM consecutive d values starting at NM*c
};

// Non-temporal loads for the decode walk.  A cached key/value row is read once per decode step by one wavefront, and
// a step's rows (every layer's cache) are usually far more than the caches hold; the hint tells L2 and the Infinity
// Cache not to keep the lines.  Measured with 64-wide heads: 128 sequences x 2 kv heads at context 1024 15.99 -> 14.36 us
// per launch (8 wavefronts 15.45 -> 14.08), context 4000 5.3 -> 5.7 TB/s - and, with the LM head streamed the same
// way, the layer weights of a small model stay resident in the Infinity Cache from step to step (Qwen2-0.5B int4 at
// batch 128: 1.42 -> 1.355 ms per step; the KV walk alone evicted them every layer).  With 128-wide heads the hint
// lost 2 % at the kernel (128 x 4 heads: 49.3 -> 50.3 us) and the models that have them do not fit the cache anyway,
// so those keep the default policy.  A launch whose block tables cannot describe more than a few MB takes the default
// policy too: a small cache is better re-read from L2 / the Infinity Cache.  NV_ATTN_NT=0 / 1 (environment) forces it.
template <bool NT, typename V> __device__ __forceinline__ V stream_load(const V* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

// Issue the loads of one chunk.  k_base/v_base point at element [row 0][this kv head][0], row_stride in
// elements; rows[2] = first row of tile 0 / tile 1 (wave-uniform).  TAIL clamps rows to max_row (packed keys).
template <typename T, int D, bool TAIL, bool NT = false>
__device__ __forceinline__ void chunk_load(ChunkRegs<T, D>& r, const uint16_t* k_base, const uint16_t* v_base,
                                           int64_t row_stride, const int64_t (&rows)[2], int64_t max_row) {
  typedef typename VecOf<T>::x8 frag_t;
  constexpr int KS = AttnGeom<D>::KS, NM = AttnGeom<D>::NM;
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  if constexpr (!TAIL) {
    // wave-uniform 64-bit row bases plus one 32-bit per-lane byte offset for K and one for V: the loads take the
    // scalar-base form and the addresses of a whole chunk cost two vector registers instead of two per load
    const uint32_t koff = (uint32_t)(c * (int)row_stride + 8 * g) * 2u;
    const uint32_t voff = (uint32_t)(4 * g * (int)row_stride + NM * c) * 2u;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      const char* kb = reinterpret_cast<const char*>(k_base + rows[tl] * row_stride);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        r.kf[tl][ks] = __builtin_bit_cast(frag_t, stream_load<NT>(reinterpret_cast<const u32x4*>(kb + (koff + 64u * ks))));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const char* vb = reinterpret_cast<const char*>(v_base + (rows[j >> 2] + (j & 3)) * row_stride);
      if constexpr (NM == 8) {
        const u32x4 t = stream_load<NT>(reinterpret_cast<const u32x4*>(vb + voff));
        r.vv[j][0] = t[0]; r.vv[j][1] = t[1]; r.vv[j][2] = t[2]; r.vv[j][3] = t[3];
      } else {
        const u32x2 t = stream_load<NT>(reinterpret_cast<const u32x2*>(vb + voff));
        r.vv[j][0] = t[0]; r.vv[j][1] = t[1];
      }
    }
    return;
  }
#pragma unroll
  for (int tl = 0; tl < 2; ++tl) {
    int64_t krow = rows[tl] + c;
    if (TAIL && krow > max_row) krow = max_row;   // packed keys: never read past the buffer
    const uint16_t* kr = k_base + krow * row_stride + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) r.kf[tl][ks] = *reinterpret_cast<const frag_t*>(kr + 32 * ks);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int off = (j < 4) ? (4 * g + j) : (4 * g + j - 4);
    int64_t vrow = rows[j >> 2] + off;
    if (TAIL && vrow > max_row) vrow = max_row;
    const uint16_t* vr = v_base + vrow * row_stride + NM * c;
    if constexpr (NM == 8) {
      const u32x4 t = *reinterpret_cast<const u32x4*>(vr);
      r.vv[j][0] = t[0]; r.vv[j][1] = t[1]; r.vv[j][2] = t[2]; r.vv[j][3] = t[3];
    } else {
      const u32x2 t = *reinterpret_cast<const u32x2*>(vr);
      r.vv[j][0] = t[0]; r.vv[j][1] = t[1];
    }
  }
}

// One loaded chunk against one 16-column query tile, in three steps so that several query tiles can share the
// chunk's registers: scores + online softmax (per tile), V^T operands (once per chunk), P.V (per tile).
//  limit: number of visible keys for this lane's column; base: first token of the chunk.
template <typename T, int D, bool MASK = true>
__device__ __forceinline__ typename VecOf<T>::x8 chunk_scores(AttnState<T, D>& st, const ChunkRegs<T, D>& r,
                                                              const typename VecOf<T>::x8 (&qf)[AttnGeom<D>::KS],
                                                              int base, int limit, float scale2) {
  typedef typename VecOf<T>::x8 frag_t;
  constexpr int KS = AttnGeom<D>::KS, NM = AttnGeom<D>::NM;
  const int lane = threadIdx.x & 63, g = lane >> 4;

  // ---- S^T = K . Q^T
  f32x4 s[2];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl) {
    s[tl] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s[tl] = mfma16(r.kf[tl][ks], qf[ks], s[tl]);
  }

  // ---- online softmax, column-local.  MASK = false: the caller knows every key of the chunk is visible.
  // scale2 > 0, so the maximum of the scaled scores is the scaled maximum of the raw ones and the scale can ride
  // in the exponent's fma; v_exp_f32 directly (its arguments are <= 0, flushed denormal results are zeros anyway).
  float cmax = -INFINITY;
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (MASK) {
        const int tok = base + 16 * tl + 4 * g + rr;
        s[tl][rr] = (tok < limit) ? s[tl][rr] : -INFINITY;
      }
      cmax = fmaxf(cmax, s[tl][rr]);
    }
  cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
  cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
  const float m_new = fmaxf(st.m, cmax * scale2);
  const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
  const float alpha = (st.m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(st.m - m_use);
  float psum = 0.f;
  frag_t pf;
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float p = __builtin_amdgcn_exp2f(fmaf(s[tl][rr], scale2, -m_use));
      psum += p;
      pf[tl * 4 + rr] = (T)p;
    }
  st.m = m_new;
  st.l = st.l * alpha + psum;
#pragma unroll
  for (int m = 0; m < NM; ++m) st.acc[m] *= alpha;
  return pf;
}

// V^T operands of a chunk: 8 x NM transpose of 16-bit values held in registers.  n_keys: keys at or past it may
// hold junk (TAIL zeroes their V rows).
template <typename T, int D, bool TAIL>
__device__ __forceinline__ void chunk_v_operands(ChunkRegs<T, D>& r, int base, int n_keys,
                                                 typename VecOf<T>::x8 (&vt)[AttnGeom<D>::NM]) {
  typedef typename VecOf<T>::x8 frag_t;
  constexpr int NM = AttnGeom<D>::NM;
  const int lane = threadIdx.x & 63, g = lane >> 4;
  if (TAIL) {  // keys past the end may hold anything (even NaN bit patterns): zero them, 0 * NaN != 0
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tok = base + ((j < 4) ? (4 * g + j) : (16 + 4 * g + j - 4));
      if (tok >= n_keys) {
#pragma unroll
        for (int e = 0; e < NM / 2; ++e) r.vv[j][e] = 0u;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    u32x4 w;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const uint32_t a = r.vv[2 * jj][m >> 1], b = r.vv[2 * jj + 1][m >> 1];
      w[jj] = (m & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
    vt[m] = __builtin_bit_cast(frag_t, w);
  }
}

template <typename T, int D>
__device__ __forceinline__ void chunk_pv(AttnState<T, D>& st, const typename VecOf<T>::x8 (&vt)[AttnGeom<D>::NM],
                                         typename VecOf<T>::x8 pf) {
#pragma unroll
  for (int m = 0; m < AttnGeom<D>::NM; ++m) st.acc[m] = mfma16(vt[m], pf, st.acc[m]);
}

// The two MFMA products and the online softmax for one loaded chunk and one query tile.
template <typename T, int D, bool TAIL, bool MASK = true>
__device__ __forceinline__ void chunk_compute(AttnState<T, D>& st, ChunkRegs<T, D>& r,
                                              const typename VecOf<T>::x8 (&qf)[AttnGeom<D>::KS], int base, int limit,
                                              int n_keys, float scale2) {
  typedef typename VecOf<T>::x8 frag_t;
  constexpr int NM = AttnGeom<D>::NM;
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const frag_t pf = chunk_scores<T, D, MASK>(st, r, qf, base, limit, scale2);
  // same arithmetic as chunk_v_operands + chunk_pv, but each V^T fragment is built right before its MFMA: with a
  // single query tile that keeps eight fewer fragments alive
  if (TAIL) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tok = base + ((j < 4) ? (4 * g + j) : (16 + 4 * g + j - 4));
      if (tok >= n_keys) {
#pragma unroll
        for (int e = 0; e < NM / 2; ++e) r.vv[j][e] = 0u;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    u32x4 w;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const uint32_t a = r.vv[2 * jj][m >> 1], b = r.vv[2 * jj + 1][m >> 1];
      w[jj] = (m & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
    st.acc[m] = mfma16(__builtin_bit_cast(frag_t, w), pf, st.acc[m]);
  }
}

template <typename T, int D, bool TAIL>
__device__ __forceinline__ void attn_chunk(AttnState<T, D>& st, const typename VecOf<T>::x8 (&qf)[AttnGeom<D>::KS],
                                           const uint16_t* k_base, const uint16_t* v_base, int64_t row_stride,
                                           const int64_t (&rows)[2], int64_t max_row, int base, int limit, int n_keys,
                                           float scale2) {
  ChunkRegs<T, D> r;
  chunk_load<T, D, TAIL>(r, k_base, v_base, row_stride, rows, max_row);
  chunk_compute<T, D, TAIL>(st, r, qf, base, limit, n_keys, scale2);
}

// Row index (in units of cache rows) of the first token of the 16-token tile that starts at `tok`.
__device__ __forceinline__ int64_t paged_tile_row(const int32_t* bt_row, int tok, int block_size) {
  const int blk = tok / block_size;              // wave-uniform
  const int off = tok - blk * block_size;
  return (int64_t)bt_row[blk] * block_size + off;
}

// Register sets a decode wavefront rotates through (chunks in flight behind the one being multiplied = sets - 1);
// 1 = a single set ahead, handed over by copy.  A workgroup of up to 4 wavefronts leaves every wavefront a SIMD's
// whole register file, so the depth is a question of memory-level parallelism only: NV_ATTN_DEPTH64 / NV_ATTN_DEPTH128
// (build-time) set it for the measurements recorded with the kernel's launch heuristics.
#ifndef NV_ATTN_DEPTH64
#define NV_ATTN_DEPTH64 2
#endif
#ifndef NV_ATTN_DEPTH128
#define NV_ATTN_DEPTH128 2
#endif
template <int D, int NW> constexpr int decode_depth() {
  if (D == 64) return NW <= 4 ? NV_ATTN_DEPTH64 : 2;
  return NW <= 4 ? NV_ATTN_DEPTH128 : 1;
}
template <typename F, int... Js>
__device__ __forceinline__ bool ring_round(F& body, std::integer_sequence<int, Js...>) {
  return (body(std::integral_constant<int, Js>{}) || ...);
}

// ------------------------------------------------------------------ decode
// NW wavefronts per workgroup share one (sequence, kv head, split); wavefront w takes chunks
// split*NW + w, + NW*splits, ...  Loads run one chunk ahead of the MFMAs and block-table lookups two ahead.
// ADAPT (used where a launch has so many groups that one wavefront each would do, see decode_waves): the workgroup is
// launched with NW wavefronts, and every group decides from the launch's context lengths how many of them it keeps -
// one if it is about as long as the average group, two from 1.5 times the average, four (if launched with four) from
// 3 times - and the others leave at once.  A batch of equal lengths then runs as if launched with one wavefront per
// group (the fastest form there), while a long sequence of a ragged batch is not left to a single wavefront.
template <typename T, int D, int NW, bool NT, bool ADAPT = false>
__global__ __launch_bounds__(NW * 64, (NW < 4 ? 2 : 1)) void paged_decode_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ k_cache,
    const uint16_t* __restrict__ v_cache, const int32_t* __restrict__ block_tables, int bt_stride,
    const int32_t* __restrict__ context_lens, uint16_t* __restrict__ out, int64_t out_stride,
    float* __restrict__ part_o, float* __restrict__ part_ml, int H, int Hkv, int block_size, float scale2,
    int splits) {
  typedef typename VecOf<T>::x8 frag_t;
  constexpr int KS = AttnGeom<D>::KS, NM = AttnGeom<D>::NM;
  __shared__ float s_acc[NW][NM * 4][64];
  __shared__ float s_m[NW][16];
  __shared__ float s_l[NW][16];

  const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: keeps the address math scalar
  const int G = H / Hkv;
  const int n_keys = context_lens[b];
  const int64_t row_stride = (int64_t)Hkv * D;
  const uint16_t* k_base = k_cache + (int64_t)kvh * D;
  const uint16_t* v_base = v_cache + (int64_t)kvh * D;
  const int32_t* bt_row = block_tables + (int64_t)b * bt_stride;

  int nwv = NW;   // wavefronts of this workgroup that take part
  int len_sum = 0;
  if constexpr (ADAPT) {
    for (int i = lane; i < (int)gridDim.z; i += 64) len_sum += context_lens[i];
  }
  // The launch's average length decides; wavefront 0 stays in any case and its first chunk is chunk 0 either way (the
  // adaptive form is launched without KV splits), so it asks only after its first loads are out.
  auto decide = [&]() {
    int total = len_sum;
#pragma unroll
    for (int o = 32; o; o >>= 1) total += __shfl_xor(total, o, 64);
    const int64_t sum = __builtin_amdgcn_readfirstlane(total), mine = (int64_t)n_keys * (int)gridDim.z;
    nwv = mine >= 3 * sum ? 4 : (2 * mine >= 3 * sum ? 2 : 1);
    nwv = nwv > NW ? NW : nwv;
  };
  if constexpr (ADAPT) {
    if (wid != 0) {
      decide();
      if (wid >= nwv) return;     // before any barrier: the hardware barrier counts the wavefronts still alive
    }
  }

  // Q^T as the B operand: lane (c, g) holds q[head c][32 ks + 8 g ..+7]
  frag_t qf[KS];
  {
    const bool live = c < G;
    const uint16_t* qr = q + (int64_t)b * q_stride + (int64_t)(kvh * G + (live ? c : 0)) * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 t = *reinterpret_cast<const u32x4*>(qr + 32 * ks);
      if (!live) t = u32x4{0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(frag_t, t);
    }
  }

  AttnState<T, D> st;
  st.m = -INFINITY;
  st.l = 0.f;
#pragma unroll
  for (int m = 0; m < NM; ++m) st.acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int n_chunks = (n_keys + 31) >> 5;
  int step = splits * nwv;
  const int max_blk = bt_stride - 1;
  // Rows of a chunk's two tiles; a tile that starts past the end re-reads the chunk's first tile (masked anyway).
  // Chunks are dealt round-robin, so at any moment the wavefronts of a workgroup stream one contiguous region of
  // the cache.  The block-table reads of the first chunk do not wait for the context length: the index is clamped
  // to the table's width and the rows are only used once the chunk is known to exist.
  auto lookup = [&](int ch, int64_t (&rows)[2]) {
    const int tok0 = ch << 5, tok1 = tok0 + 16;
    int b0 = tok0 / block_size, b1 = tok1 / block_size;
    b0 = b0 > max_blk ? max_blk : b0;
    b1 = b1 > max_blk ? max_blk : b1;
    const int64_t r0 = (int64_t)bt_row[b0] * block_size + (tok0 - b0 * block_size);
    const int64_t r1 = (int64_t)bt_row[b1] * block_size + (tok1 - b1 * block_size);
    rows[0] = r0;
    rows[1] = tok1 < n_keys ? r1 : r0;
  };
  // Two register sets alternate between "being multiplied" and "being loaded": the loads of the next chunk are
  // issued before the MFMAs of the current one and nothing is ever copied from a set that is still in flight (a
  // copy would wait for its loads), so each wavefront keeps one chunk of HBM traffic in flight behind its compute.
  // Rows are looked up one chunk further ahead than the loads.
  int ch = ADAPT ? wid : split * nwv + wid;
  auto compute = [&](ChunkRegs<T, D>& r, int chunk) {
    const int base = chunk << 5;
    if (base + 32 <= n_keys) chunk_compute<T, D, false, false>(st, r, qf, base, n_keys, n_keys, scale2);
    else chunk_compute<T, D, true>(st, r, qf, base, n_keys, n_keys, scale2);
  };
  // a chunk that has a successor in this wavefront's sequence is not the sequence's last chunk, so all of its 32
  // keys exist: the steady-state loop body carries no masking code
  auto compute_full = [&](ChunkRegs<T, D>& r, int chunk) {
    chunk_compute<T, D, false, false>(st, r, qf, chunk << 5, n_keys, n_keys, scale2);
  };
  constexpr int DEPTH = decode_depth<D, NW>();
  if constexpr (DEPTH == 1) {
    // 128-wide heads in workgroups of 8+ wavefronts (two per SIMD, 256 registers each): two alternating register
    // sets spill there (and run 13 % slower), so one set is loaded ahead and handed over by copy (the 16 KB chunks
    // keep enough bytes in flight per wavefront regardless)
    int64_t rows_cur[2], rows_nxt[2] = {0, 0};
    lookup(ch, rows_cur);
    if (ch < n_chunks) {
      if (ch + step < n_chunks) lookup(ch + step, rows_nxt);
      ChunkRegs<T, D> cur, nxt;
      chunk_load<T, D, false, NT>(cur, k_base, v_base, row_stride, rows_cur, INT64_MAX);
      for (; ch < n_chunks; ch += step) {
        const bool has_next = ch + step < n_chunks;
        int64_t rows_nn[2] = {0, 0};
        if (ch + 2 * step < n_chunks) lookup(ch + 2 * step, rows_nn);
        if (has_next) {
          chunk_load<T, D, false, NT>(nxt, k_base, v_base, row_stride, rows_nxt, INT64_MAX);
          compute_full(cur, ch);
        } else {
          compute(cur, ch);
        }
        cur = nxt;
        rows_nxt[0] = rows_nn[0];
        rows_nxt[1] = rows_nn[1];
      }
    }
  } else {
   int64_t rows0[2];
   lookup(ch, rows0);   // issued before the context length has arrived: clamped index, used only if the chunk exists
   if (ch < n_chunks) {
    // DEPTH register sets in rotation: while set J is multiplied, the loads of the DEPTH - 1 following chunks are
    // in flight in the other sets, and nothing is ever copied out of a set that is still being loaded (a copy would
    // wait for its loads).  The steady-state body is straight-line code - its load is unconditional and the loop is
    // left BEFORE a body whose load would not exist - so the compiler's s_waitcnt vmcnt counts are exact: an MFMA
    // waits for the fragments of its own chunk only.  For the same reason the first DEPTH - 1 loads are issued
    // unconditionally: a wavefront with fewer chunks than that reads its first chunk again (a cache hit).
    ChunkRegs<T, D> rs[DEPTH];
    int64_t rw[2];
    chunk_load<T, D, false, NT>(rs[0], k_base, v_base, row_stride, rows0, INT64_MAX);
    if constexpr (ADAPT) {
      if (wid == 0) decide();
      step = splits * nwv;
    }
#pragma unroll
    for (int j = 1; j < DEPTH - 1; ++j) {
      int64_t rj[2];
      lookup(ch + j * step, rj);
      const bool exists = ch + j * step < n_chunks;
      rj[0] = exists ? rj[0] : rows0[0];
      rj[1] = exists ? rj[1] : rows0[1];
      chunk_load<T, D, false, NT>(rs[j], k_base, v_base, row_stride, rj, INT64_MAX);
    }
    lookup(ch + (DEPTH - 1) * step, rw);   // rows of the next chunk to load (used only if that chunk exists)
    // body J: multiply set J, refill the set that was multiplied last (J - 1) with the chunk DEPTH - 1 ahead; returns
    // true once the remaining chunks (all loaded) have been multiplied
    auto body = [&](auto jc) -> bool {
      constexpr int J = decltype(jc)::value;
      if (ch + (DEPTH - 1) * step >= n_chunks) {
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) {
          if (ch < n_chunks) compute(rs[(J + d) % DEPTH], ch);
          ch += step;
        }
        return true;
      }
      int64_t rn[2];
      lookup(ch + DEPTH * step, rn);
      chunk_load<T, D, false, NT>(rs[(J + DEPTH - 1) % DEPTH], k_base, v_base, row_stride, rw, INT64_MAX);
      compute_full(rs[J], ch);
      ch += step;
      rw[0] = rn[0];
      rw[1] = rn[1];
      return false;
    };
    while (!ring_round(body, std::make_integer_sequence<int, DEPTH>{})) {
    }
   } else if constexpr (ADAPT) {
    if (wid == 0) decide();    // a group without keys (a padding row): the other wavefront has left, know it
   }
  }

  // ---- merge the NW wavefronts of the workgroup.  The first NF = min(NW, 4) wavefronts finish the accumulator
  // register indices r = wid, wid + NF, ...: d = NM * (4g + r) + m, m = 0..NM-1 contiguous.  A single wavefront
  // has nothing to merge and keeps everything in registers.
  constexpr int NF = NW < 4 ? NW : 4;
  float l = st.l;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if constexpr (ADAPT) {
    // the same merge over the nwv (1, 2 or 4) wavefronts that stayed; each finishes 4 / nwv of the register indices
    static_assert(NW == 2 || NW == 4, "the adaptive form is launched with two or four wavefronts");
    float M = st.m, L = l, f[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) f[w] = w == 0 ? 1.f : 0.f;
    if (nwv > 1) {
      if (g == 0) {
        s_m[wid][c] = st.m;
        s_l[wid][c] = l;
      }
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_acc[wid][m * 4 + r][lane] = st.acc[m][r];
      __syncthreads();
      M = -INFINITY;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (w < nwv) M = fmaxf(M, s_m[w][c]);
      L = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        f[w] = 0.f;
        if (w < nwv) {
          const float mw = s_m[w][c];
          f[w] = (mw == -INFINITY) ? 0.f : exp2f(mw - M);
          L += s_l[w][c] * f[w];
        }
      }
    }
    if (c >= G) return;
    const int head = kvh * G + c;
    const float inv = (L > 0.f) ? 1.0f / L : 0.f;
    const int64_t slot = ((int64_t)b * H + head) * splits + split;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (rr >= 4 / nwv) break;
      const int r = wid + rr * nwv;
      const int d0 = NM * (4 * g + r);
      float o[NM];
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        float v = 0.f;
        if (nwv == 1) {
          // r == rr here; a register array cannot be indexed by a run-time value without spilling, so pick by compare
          v = rr == 0 ? st.acc[m][0] : rr == 1 ? st.acc[m][1] : rr == 2 ? st.acc[m][2] : st.acc[m][3];
        } else {
#pragma unroll
          for (int w = 0; w < NW; ++w)
            if (w < nwv) v += s_acc[w][m * 4 + r][lane] * f[w];
        }
        o[m] = v;
      }
      if (splits == 1) {
        uint16_t* dst = out + (int64_t)b * out_stride + (int64_t)head * D + d0;
        if constexpr (NM == 8) {
          u32x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = pack2<T>(o[2 * e] * inv, o[2 * e + 1] * inv);
          *reinterpret_cast<u32x4*>(dst) = w;
        } else {
          u32x2 w;
#pragma unroll
          for (int e = 0; e < 2; ++e) w[e] = pack2<T>(o[2 * e] * inv, o[2 * e + 1] * inv);
          *reinterpret_cast<u32x2*>(dst) = w;
        }
      } else {
        float* dst = part_o + slot * D + d0;
#pragma unroll
        for (int m = 0; m < NM; ++m) dst[m] = o[m];
      }
    }
    if (splits > 1 && g == 0 && wid == 0) {
      part_ml[slot * 2] = M;
      part_ml[slot * 2 + 1] = L;
    }
    return;
  }
  float M, L, f[NW];
  if constexpr (NW == 1) {
    M = st.m;
    L = l;
    f[0] = 1.f;
  } else {
    if (g == 0) {
      s_m[wid][c] = st.m;
      s_l[wid][c] = l;
    }
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_acc[wid][m * 4 + r][lane] = st.acc[m][r];
    __syncthreads();
    if (wid >= NF) return;
    M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, s_m[w][c]);
    L = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float mw = s_m[w][c];
      f[w] = (mw == -INFINITY) ? 0.f : exp2f(mw - M);
      L += s_l[w][c] * f[w];
    }
  }
  if (c >= G) return;
  const int head = kvh * G + c;
  const float inv = (L > 0.f) ? 1.0f / L : 0.f;
  const int64_t slot = ((int64_t)b * H + head) * splits + split;
#pragma unroll
  for (int rr = 0; rr < 4 / NF; ++rr) {
    float o[NM];
    int d0;
    if constexpr (NW == 1) {
      d0 = NM * (4 * g + rr);
#pragma unroll
      for (int m = 0; m < NM; ++m) o[m] = st.acc[m][rr];
    } else {
      const int r = wid + rr * NF;
      d0 = NM * (4 * g + r);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += s_acc[w][m * 4 + r][lane] * f[w];
        o[m] = v;
      }
    }
    if (splits == 1) {
      uint16_t* dst = out + (int64_t)b * out_stride + (int64_t)head * D + d0;
      if constexpr (NM == 8) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack2<T>(o[2 * e] * inv, o[2 * e + 1] * inv);
        *reinterpret_cast<u32x4*>(dst) = w;
      } else {
        u32x2 w;
#pragma unroll
        for (int e = 0; e < 2; ++e) w[e] = pack2<T>(o[2 * e] * inv, o[2 * e + 1] * inv);
        *reinterpret_cast<u32x2*>(dst) = w;
      }
    } else {
      float* dst = part_o + slot * D + d0;
#pragma unroll
      for (int m = 0; m < NM; ++m) dst[m] = o[m];
    }
  }
  if (splits > 1 && g == 0 && wid == 0) {
    part_ml[slot * 2] = M;
    part_ml[slot * 2 + 1] = L;
  }
}

// merge the per-split partials: one wavefront per (sequence, head)
template <typename T, int D>
__global__ __launch_bounds__(256) void decode_combine_kernel(const float* __restrict__ part_o,
                                                             const float* __restrict__ part_ml,
                                                             uint16_t* __restrict__ out, int64_t out_stride, int H,
                                                             int splits, int64_t total) {
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= total) return;
  const int lane = threadIdx.x & 63;
  const int64_t b = item / H;
  const int head = (int)(item - b * H);
  const float* ml = part_ml + item * splits * 2;
  float M = -INFINITY;
  for (int s = 0; s < splits; ++s) M = fmaxf(M, ml[2 * s]);
  float L = 0.f;
  constexpr int PER = D / 64;
  float o[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) o[e] = 0.f;
  for (int s = 0; s < splits; ++s) {
    const float ms = ml[2 * s];
    const float f = (ms == -INFINITY) ? 0.f : exp2f(ms - M);
    L += ml[2 * s + 1] * f;
    const float* po = part_o + (item * splits + s) * D + lane * PER;
#pragma unroll
    for (int e = 0; e < PER; ++e) o[e] += po[e] * f;
  }
  const float inv = (L > 0.f) ? 1.0f / L : 0.f;
  uint16_t* dst = out + b * out_stride + (int64_t)head * D + lane * PER;
#pragma unroll
  for (int e = 0; e < PER; ++e) dst[e] = f32_to_bits<T>(o[e] * inv);
}

// ------------------------------------------------------------------ prefill
// grid (ceil(max_seqlen_q / (64 QT)), H, num_seqs); wavefront w of a workgroup owns QT consecutive 16-query tiles
// of one head of one sequence.  Every K/V chunk it loads is multiplied against all QT tiles, so the bytes fetched per
// flop fall with QT (the chunk's V^T operands are also built once instead of once per tile); the loads of the next
// chunk are issued before the MFMAs of the current one (two register sets in alternation, PP).  Workgroups are
// numbered from the end of the sequence backwards: late query tiles see the most keys and start first.
template <typename T, int D, bool PAGED, int QT, bool PP>
__global__ __launch_bounds__(256) void prefill_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    int64_t kv_stride, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ block_tables, int bt_stride, uint16_t* __restrict__ out, int64_t out_stride, int H,
    int Hkv, int block_size, float scale2) {
  typedef typename VecOf<T>::x8 frag_t;
  constexpr int KS = AttnGeom<D>::KS, NM = AttnGeom<D>::NM;
  const int seq = blockIdx.z, head = blockIdx.y;
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q0 = cu_q[seq], q_len = cu_q[seq + 1] - q0;
  const int k0 = cu_k[seq], k_len = cu_k[seq + 1] - k0;
  const int bx = gridDim.x - 1 - blockIdx.x;
  const int qt = (bx * 4 + wid) * (16 * QT);   // first query of this wavefront
  if (qt >= q_len) return;
  const int kvh = head / (H / Hkv);

  // bottom-right aligned causal mask: query qi sees keys [0, k_len - q_len + qi]
  int limit[QT], tile_keys[QT];
  bool live[QT];
  frag_t qf[QT][KS];
  AttnState<T, D> st[QT];
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    const int t0 = qt + 16 * i, qi = t0 + c;
    live[i] = qi < q_len;
    limit[i] = live[i] ? (k_len - q_len + qi + 1) : 0;
    int last_q = t0 + 15;
    if (last_q > q_len - 1) last_q = q_len - 1;
    tile_keys[i] = t0 < q_len ? (k_len - q_len + last_q + 1) : 0;   // keys any column of the tile can see
    const uint16_t* qr = q + (int64_t)(q0 + (live[i] ? qi : qt)) * q_stride + (int64_t)head * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 t = *reinterpret_cast<const u32x4*>(qr + 32 * ks);
      if (!live[i]) t = u32x4{0, 0, 0, 0};
      qf[i][ks] = __builtin_bit_cast(frag_t, t);
    }
    st[i].m = -INFINITY;
    st[i].l = 0.f;
#pragma unroll
    for (int m = 0; m < NM; ++m) st[i].acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int n_keys = 0;   // keys the wavefront's last existing tile can see: the extent of its chunk loop
#pragma unroll
  for (int i = 0; i < QT; ++i) n_keys = tile_keys[i] > n_keys ? tile_keys[i] : n_keys;

  const int64_t row_stride = PAGED ? (int64_t)Hkv * D : kv_stride;
  const uint16_t* k_base = k + (int64_t)kvh * D;
  const uint16_t* v_base = v + (int64_t)kvh * D;
  const int32_t* bt_row = PAGED ? block_tables + (int64_t)seq * bt_stride : nullptr;
  const int64_t max_row = PAGED ? INT64_MAX : (int64_t)k0 + k_len - 1;
  const int n_chunks = (n_keys + 31) >> 5;

  // a chunk that ends past n_keys takes the TAIL forms: rows clamped (packed keys end with the buffer), junk V zeroed
  auto load = [&](ChunkRegs<T, D>& r, int ch) {
    const int base = ch << 5;
    int64_t rows[2];
    if (PAGED) {
      rows[0] = paged_tile_row(bt_row, base, block_size);
      rows[1] = (base + 16 < n_keys) ? paged_tile_row(bt_row, base + 16, block_size) : rows[0];
    } else {
      rows[0] = k0 + base;
      rows[1] = k0 + base + 16;
    }
    if (base + 32 <= n_keys) chunk_load<T, D, false>(r, k_base, v_base, row_stride, rows, max_row);
    else chunk_load<T, D, true>(r, k_base, v_base, row_stride, rows, max_row);
  };
  auto compute = [&](ChunkRegs<T, D>& r, int ch) {
    const int base = ch << 5;
    frag_t pf[QT], vt[NM];
#pragma unroll
    for (int i = 0; i < QT; ++i)   // tiles that end before this chunk skip it (wave-uniform)
      if (base < tile_keys[i]) pf[i] = chunk_scores<T, D>(st[i], r, qf[i], base, limit[i], scale2);
    if (base + 32 <= n_keys) chunk_v_operands<T, D, false>(r, base, n_keys, vt);
    else chunk_v_operands<T, D, true>(r, base, n_keys, vt);
#pragma unroll
    for (int i = 0; i < QT; ++i)
      if (base < tile_keys[i]) chunk_pv<T, D>(st[i], vt, pf[i]);
  };
  if constexpr (PP) {
    ChunkRegs<T, D> ra, rb;
    load(ra, 0);
    int ch = 0;
    while (true) {
      if (ch + 1 >= n_chunks) {
        compute(ra, ch);
        break;
      }
      load(rb, ch + 1);
      compute(ra, ch);
      ++ch;
      if (ch + 1 >= n_chunks) {
        compute(rb, ch);
        break;
      }
      load(ra, ch + 1);
      compute(rb, ch);
      ++ch;
    }
  } else {
    for (int ch = 0; ch < n_chunks; ++ch) {
      ChunkRegs<T, D> r;
      load(r, ch);
      compute(r, ch);
    }
  }

#pragma unroll
  for (int i = 0; i < QT; ++i) {
    float l = st[i].l;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (!live[i]) continue;
    const float inv = (l > 0.f) ? 1.0f / l : 0.f;
    uint16_t* orow = out + (int64_t)(q0 + qt + 16 * i + c) * out_stride + (int64_t)head * D;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint16_t* dst = orow + NM * (4 * g + r);
      if constexpr (NM == 8) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack2<T>(st[i].acc[2 * e][r] * inv, st[i].acc[2 * e + 1][r] * inv);
        *reinterpret_cast<u32x4*>(dst) = w;
      } else {
        u32x2 w;
#pragma unroll
        for (int e = 0; e < 2; ++e) w[e] = pack2<T>(st[i].acc[2 * e][r] * inv, st[i].acc[2 * e + 1][r] * inv);
        *reinterpret_cast<u32x2*>(dst) = w;
      }
    }
  }
}

// ------------------------------------------------------------------ prefill, K/V shared through LDS
// One workgroup = 4 wavefronts = HW query heads of one KV head x (4 / HW) groups of 16 QT queries, so every K/V
// byte the workgroup fetches serves all four wavefronts (the register-fed kernel above fetches it once per
// wavefront).  Keys are taken in blocks of NC chunks of 32 (128 keys at D = 64, 64 at D = 128) and the online
// softmax advances once per block: its fixed cost per step (column maximum across lanes, rescaling the output
// accumulators, -inf guards) is what bounded the chunk-at-a-time form at ~20 VALU instructions per score.
// A block travels L2 -> LDS by LDS-DMA, 1 KiB per wave-instruction, into one half of a double buffer while the other
// half is multiplied.  The DMA writes its 64 lanes' 16-byte pieces to consecutive LDS addresses, so the LDS image is
// shaped by choosing which global piece each lane fetches:
//   K piece (tile tl, k-step ks): lane (c, g) fetches K[16 tl + c][32 ks + 8 g ..+7] - exactly the MFMA A fragment,
//     read back with one conflict-free ds_read_b128 at lane * 16;
//   V piece: 8 rows x 128 B (D = 64) or 4 rows x 256 B (D = 128), rows whole but with their 16-byte pieces XOR-ed
//     by a function of the row, read back with ds_read_b64_tr_b16: the hardware hands lane c of a 16-lane group the
//     c-th column of a 4-token x 16-d block, i.e. the V^T A-fragment of the P.V product with no transpose on the
//     VALU (MFMA m covers d = 16 m .. 16 m + 15; the XOR spreads the 8 rows a 32-lane half touches over all banks).
// One barrier per block: at the top of iteration i the wavefront waits for its own pieces of block i, the barrier
// then says that all of block i has landed AND that everybody is done with block i-1, whose half the DMA of block
// i+1 may now overwrite.  Every block is fetched whole (rows past the sequence are clamped to real ones), so LDS never
// holds uninitialised values for a token below k_len.
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_void_t;

// Online-softmax step of one 16-query tile over a block of NC chunks whose raw scores sit in s[NC][2].
// MASK: apply the causal limit (only blocks that touch the tile's diagonal need it).  Returns P^T fragments.
template <typename T, int D, int NC, bool MASK>
__device__ __forceinline__ void block_softmax(AttnState<T, D>& st, f32x4 (&s)[NC][2], int base, int limit, float scale2,
                                              typename VecOf<T>::x8 (&pf)[NC]) {
  constexpr int NM = AttnGeom<D>::NM;
  const int g = (threadIdx.x & 63) >> 4;
  float cmax = -INFINITY;
#pragma unroll
  for (int cc = 0; cc < NC; ++cc)
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (MASK) {
          const int tok = base + 32 * cc + 16 * tl + 4 * g + rr;
          s[cc][tl][rr] = (tok < limit) ? s[cc][tl][rr] : -INFINITY;
        }
        cmax = fmaxf(cmax, s[cc][tl][rr]);
      }
  cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
  cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
  // scale2 > 0, so the maximum of the scaled scores is the scaled maximum: the scale rides in the exponent's fma
  const float m_new = fmaxf(st.m, cmax * scale2);
  const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
  const float alpha = (st.m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(st.m - m_use);
  float psum = 0.f;
#pragma unroll
  for (int cc = 0; cc < NC; ++cc)
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[cc][tl][rr], scale2, -m_use));
        psum += p;
        pf[cc][tl * 4 + rr] = (T)p;
      }
  st.m = m_new;
  st.l = st.l * alpha + psum;
#pragma unroll
  for (int m = 0; m < NM; ++m) st.acc[m] *= alpha;
}

// XOR applied to the 16-byte piece index of row t of a chunk's V image (see above)
template <int D> __device__ __forceinline__ int v_swizzle(int t) {
  return D == 64 ? 2 * ((t >> 1) & 3) : 2 * (t & 7);
}

template <typename T, int D, bool PAGED, int QT, int HW>
__global__ __launch_bounds__(256, 2) void prefill_lds_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    int64_t kv_stride, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ block_tables, int bt_stride, uint16_t* __restrict__ out, int64_t out_stride, int H,
    int Hkv, int block_size, float scale2) {
  typedef typename VecOf<T>::x8 frag_t;
  constexpr int KS = AttnGeom<D>::KS, NM = AttnGeom<D>::NM;
  constexpr int QW = 4 / HW;                    // query groups per workgroup
  constexpr int NC = D == 64 ? 4 : 2;           // chunks per key block
  constexpr int BK = 32 * NC;                   // keys per block
  constexpr int K_BYTES = 32 * D * 2;           // one chunk of K (= of V)
  constexpr int PIECES = 2 * K_BYTES / 1024;    // 1-KiB DMA pieces per chunk: 8 (D = 64) or 16
  constexpr int LPW = NC * PIECES / 4;          // pieces each wavefront moves per block
  __shared__ __attribute__((aligned(1024))) unsigned char ring[2][NC][2 * K_BYTES];   // 64 KiB

  const int seq = blockIdx.z;
  const int G = H / Hkv, groups = (G + HW - 1) / HW;
  const int kvh = blockIdx.y / groups, hq = blockIdx.y - kvh * groups;
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q0 = cu_q[seq], q_len = cu_q[seq + 1] - q0;
  const int k0 = cu_k[seq], k_len = cu_k[seq + 1] - k0;
  const int bx = gridDim.x - 1 - blockIdx.x;    // late query blocks see the most keys: start them first
  const int wg_q = bx * (QW * 16 * QT);         // first query of the workgroup
  if (wg_q >= q_len) return;                    // uniform over the workgroup
  const int head_local = hq * HW + (wid % HW);
  const int qt = wg_q + (wid / HW) * (16 * QT); // first query of this wavefront
  const bool active = head_local < G && qt < q_len;
  const int head = kvh * G + (head_local < G ? head_local : 0);

  int limit[QT], tile_keys[QT], tile_clear[QT];
  bool live[QT];
  frag_t qf[QT][KS];
  AttnState<T, D> st[QT];
  int my_keys = 0;            // extent of this wavefront's own loop
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    const int t0 = qt + 16 * i, qi = t0 + c;
    live[i] = active && qi < q_len;
    limit[i] = live[i] ? (k_len - q_len + qi + 1) : 0;   // bottom-right aligned causal mask
    int last_q = t0 + 15;
    if (last_q > q_len - 1) last_q = q_len - 1;
    const bool exists = active && t0 < q_len;
    tile_keys[i] = exists ? (k_len - q_len + last_q + 1) : 0;          // keys any column of the tile can see
    tile_clear[i] = (exists && t0 + 15 < q_len) ? (k_len - q_len + t0 + 1) : 0;   // keys EVERY column sees
    my_keys = tile_keys[i] > my_keys ? tile_keys[i] : my_keys;
    const uint16_t* qr = q + (int64_t)(q0 + (live[i] ? qi : wg_q)) * q_stride + (int64_t)head * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 t = *reinterpret_cast<const u32x4*>(qr + 32 * ks);
      if (!live[i]) t = u32x4{0, 0, 0, 0};
      qf[i][ks] = __builtin_bit_cast(frag_t, t);
    }
    st[i].m = -INFINITY;
    st[i].l = 0.f;
#pragma unroll
    for (int m = 0; m < NM; ++m) st[i].acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int wg_last = wg_q + QW * 16 * QT - 1;
  if (wg_last > q_len - 1) wg_last = q_len - 1;
  const int wg_keys = k_len - q_len + wg_last + 1;     // keys the workgroup's last query sees
  const int n_blocks = (wg_keys + BK - 1) / BK;

  const int64_t row_stride = PAGED ? (int64_t)Hkv * D : kv_stride;
  const uint16_t* k_base = k + (int64_t)kvh * D;
  const uint16_t* v_base = v + (int64_t)kvh * D;
  const int32_t* bt_row = PAGED ? block_tables + (int64_t)seq * bt_stride : nullptr;
  const int64_t max_row = (int64_t)k0 + k_len - 1;     // packed keys: never read past the sequence

  // The wavefront's pieces of a chunk are wid, wid + 4, ...: which operand, which 16-token tile and which lane offset
  // inside that tile a piece takes never changes, so they are worked out once (the offset fits 32 bits: a tile spans
  // 16 cache rows) and the loop adds a wave-uniform tile address, which the DMA takes as its scalar base.
  uint32_t piece_off[PIECES / 4];
#pragma unroll
  for (int n = 0; n < PIECES / 4; ++n) {
    const int piece = wid + 4 * n;
    if (piece < PIECES / 2) {
      const int ks = piece % KS;
      piece_off[n] = (uint32_t)(c * (int)row_stride + 32 * ks + 8 * g) * 2u;
    } else {
      const int vp = piece - PIECES / 2;
      const int t = D == 64 ? 8 * vp + (lane >> 3) : 4 * vp + (lane >> 4);
      const int col = D == 64 ? 8 * ((lane & 7) ^ v_swizzle<D>(t)) : 8 * ((lane & 15) ^ v_swizzle<D>(t));
      piece_off[n] = (uint32_t)((t & 15) * (int)row_stride + col) * 2u;
    }
  }
  // Issue this wavefront's LPW pieces of key block `blk` into ring half `half`.
  auto issue = [&](int blk, int half) {
    const int base0 = blk * BK;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      const int base = base0 + 32 * cc;
      int64_t rows[2];   // first cache row of the chunk's two 16-token tiles (wave-uniform)
      if (PAGED) {       // tiles that start at or past k_len have no table entry: they re-read the block's first tile
        const int64_t r00 = paged_tile_row(bt_row, base0, block_size);
        rows[0] = base < k_len ? paged_tile_row(bt_row, base, block_size) : r00;
        rows[1] = base + 16 < k_len ? paged_tile_row(bt_row, base + 16, block_size) : r00;
      } else {
        rows[0] = k0 + base;
        rows[1] = k0 + base + 16;
      }
      if (PAGED || base + 32 <= k_len) {   // every row of the chunk exists: scalar tile base + constant lane offset
#pragma unroll
        for (int n = 0; n < PIECES / 4; ++n) {
          const int piece = wid + 4 * n;
          const bool is_k = piece < PIECES / 2;
          const int vp = piece - PIECES / 2;
          const int tile = is_k ? piece / KS : (D == 64 ? vp >> 1 : vp >> 2);
          const char* tb = reinterpret_cast<const char*>((is_k ? k_base : v_base) + rows[tile] * row_stride);
          __builtin_amdgcn_global_load_lds((global_void_t*)(tb + piece_off[n]),
                                           (lds_void_t*)&ring[half][cc][piece * 1024], 16, 0, 0);
        }
        continue;
      }
#pragma unroll
      for (int n = 0; n < PIECES / 4; ++n) {
        const int piece = wid + 4 * n;            // wave-uniform
        const uint16_t* src;
        if (piece < PIECES / 2) {                 // K fragment (tl, ks)
          const int tl = piece / KS, ks = piece - tl * KS;
          int64_t row = rows[tl] + c;
          if (!PAGED && row > max_row) row = max_row;
          src = k_base + row * row_stride + 32 * ks + 8 * g;
        } else {
          const int vp = piece - PIECES / 2;
          int t, col;   // LDS slot `lane` of this piece = (row t, swizzled piece); fetch the piece that belongs there
          if constexpr (D == 64) {
            t = 8 * vp + (lane >> 3);
            col = 8 * ((lane & 7) ^ v_swizzle<D>(t));
          } else {
            t = 4 * vp + (lane >> 4);
            col = 8 * ((lane & 15) ^ v_swizzle<D>(t));
          }
          int64_t row = rows[t >> 4] + (t & 15);
          if (!PAGED && row > max_row) row = max_row;
          src = v_base + row * row_stride + col;
        }
        __builtin_amdgcn_global_load_lds((global_void_t*)src, (lds_void_t*)&ring[half][cc][piece * 1024], 16, 0, 0);
      }
    }
  };
  auto fetch_k = [&](const unsigned char* slot, frag_t (&kf)[2][KS]) {
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        kf[tl][ks] = *reinterpret_cast<const frag_t*>(slot + (tl * KS + ks) * 1024 + lane * 16);
  };
  // byte offset, inside a chunk's V image, of this lane's transposed-read address for MFMA m and the chunk's first
  // 16 tokens (the second 16 are 16 rows further): lane 4 q + p of group g points at row 4 g + q, d = 16 m + 4 p
  int vaddr[NM];
  {
    const int qq = (lane & 15) >> 2, pp = lane & 3, t = 4 * g + qq;
#pragma unroll
    for (int m = 0; m < NM; ++m)
      vaddr[m] = K_BYTES + t * (2 * D) + (((2 * m + (pp >> 1)) ^ v_swizzle<D>(t)) << 4) + (pp & 1) * 8;
  }
  // V^T fragments of one chunk; TAIL zeroes the tokens at or past k_len (stale cache rows may hold NaN patterns)
  auto fetch_vt = [&](const unsigned char* slot, int base, auto tail, frag_t (&vt)[NM]) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s lds_v4s_t;
    uint32_t keep[4] = {~0u, ~0u, ~0u, ~0u};
    if constexpr (decltype(tail)::value) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int tok = base + ((e < 4) ? (4 * g + e) : (16 + 4 * g + e - 4));
        if (tok >= k_len) keep[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
      }
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(slot + vaddr[m]));
      const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(slot + vaddr[m] + 16 * 2 * D));
      const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
      if constexpr (decltype(tail)::value)
        vt[m] = __builtin_bit_cast(frag_t, u32x4{l2[0] & keep[0], l2[1] & keep[1], h2[0] & keep[2], h2[1] & keep[3]});
      else
        vt[m] = __builtin_bit_cast(frag_t, u32x4{l2[0], l2[1], h2[0], h2[1]});
    }
  };

  issue(0, 0);
  for (int blk = 0; blk < n_blocks; ++blk) {
    wait_vmcnt<0>();                         // this wavefront's pieces of block blk have landed
    __builtin_amdgcn_s_barrier();
    if (blk + 1 < n_blocks) issue(blk + 1, (blk + 1) & 1);
    const int base = blk * BK;
    if (base >= my_keys) continue;           // wave-uniform: early query groups and idle head slots only move data
    const unsigned char* half = &ring[blk & 1][0][0];

    // ---- S^T = K . Q^T for the whole block, K fragments straight from LDS
    f32x4 s[QT][NC][2];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      frag_t kf[2][KS];
      fetch_k(half + cc * 2 * K_BYTES, kf);
#pragma unroll
      for (int i = 0; i < QT; ++i)
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          s[i][cc][tl] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) s[i][cc][tl] = mfma16(kf[tl][ks], qf[i][ks], s[i][cc][tl]);
        }
    }
    // ---- one softmax step per tile; blocks wholly below the tile's diagonal need no mask
    frag_t pf[QT][NC];
    bool all_clear = true;
#pragma unroll
    for (int i = 0; i < QT; ++i) all_clear = all_clear && (base + BK <= tile_clear[i]);
    if (all_clear) {   // straight-line code for all tiles: their serial softmax chains can be interleaved
#pragma unroll
      for (int i = 0; i < QT; ++i) block_softmax<T, D, NC, false>(st[i], s[i], base, limit[i], scale2, pf[i]);
    } else {
#pragma unroll
      for (int i = 0; i < QT; ++i) block_softmax<T, D, NC, true>(st[i], s[i], base, limit[i], scale2, pf[i]);
    }
    // ---- O^T += V^T . P^T, V^T fragments by transposed LDS reads, shared by all tiles
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      frag_t vt[NM];
      if (base + 32 * cc + 32 <= k_len) fetch_vt(half + cc * 2 * K_BYTES, base + 32 * cc, std::false_type{}, vt);
      else fetch_vt(half + cc * 2 * K_BYTES, base + 32 * cc, std::true_type{}, vt);
#pragma unroll
      for (int i = 0; i < QT; ++i) chunk_pv<T, D>(st[i], vt, pf[i][cc]);
    }
  }

#pragma unroll
  for (int i = 0; i < QT; ++i) {
    float l = st[i].l;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (!live[i]) continue;
    const float inv = (l > 0.f) ? 1.0f / l : 0.f;
    uint16_t* orow = out + (int64_t)(q0 + qt + 16 * i + c) * out_stride + (int64_t)head * D + 4 * g;
#pragma unroll
    for (int m = 0; m < NM; ++m) {   // MFMA m produced d = 16 m + 4 g + r in register r
      u32x2 w;
      w[0] = pack2<T>(st[i].acc[m][0] * inv, st[i].acc[m][1] * inv);
      w[1] = pack2<T>(st[i].acc[m][2] * inv, st[i].acc[m][3] * inv);
      *reinterpret_cast<u32x2*>(orow + 16 * m) = w;
    }
  }
}

// ------------------------------------------------------------------ C ABI
// Wavefronts per workgroup, from the number of (sequence, kv head, split) groups of the launch.  A CU holds 8
// wavefronts of this kernel (two per SIMD), so smaller workgroups put several groups on a CU: one group's dependent
// start (context length -> block table -> first K/V rows) and its merge then overlap the others' streaming, and each
// wavefront walks more chunks with its loads running ahead.  About a thousand wavefronts in total keep the memory
// system busy; beyond that, fewer per group win.  Measured (us per launch with 8 / 4 / 2 / 1 wavefronts):
//   128 seqs x 8 kv heads of 128, context 180:   30.0 / 23.9 / 21.9 / 20.8
//   128 seqs x 4 kv heads of 128, context 1024:  53.0 / 53.0 / 49.5 / 59.7
//   512 seqs x 4 kv heads of 128, context 1024:  199.7 / 198.8 / 197.6 / 193.1
//   128 seqs x 2 kv heads of 64,  context 1024:  15.8 / 15.6 / 16.2 / -
// The choice depends on the launch shape only, never on the context lengths, so a captured launch stays valid for
// every replay.  One wavefront per group only where the block table cannot describe more than 512 tokens per
// sequence: a long straggler of a ragged batch would otherwise stream at a single wavefront's rate.
// NV_ATTN_WAVES (1, 2, 4, 8, or 16 at D=64; 0 = the adaptive form at D=128) selects another instantiation; read on
// every call so one process can compare them.
static int decode_waves(int D, int64_t groups, int64_t max_tokens, int splits) {
  if (const char* e = getenv("NV_ATTN_WAVES")) {
    const int forced = atoi(e);
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
    if (forced == 16) return D == 128 ? 8 : 16;
    if (forced == 0 && D == 128 && splits == 1 && e[0] == '0') return 0;
  }
  if (groups > 512 && max_tokens <= 512) return 1;
  if (D == 64) return groups > 256 ? 2 : 8;       // with non-temporal loads 8 beat 4 at 256 groups (14.08 / 14.36 us)
  // 128-wide heads keep 32 KB in flight per wavefront (two 16 KB chunks), so 512 wavefronts already cover the memory
  // system's latency: with the register-set ring one wavefront per group is the fastest form for a batch of EQUAL
  // lengths from 512 groups on (46.2 / 49.1 / 50.6 / 52.2 us with 1 / 2 / 4 / 8 at 128 x 4 heads, context 1024; 182 /
  // 191 at 4096; 190.6 / 197.3 at 2048 groups), two at 256 groups (24.8 against 26.7 with 4, 36.6 with 1) - and the
  // worst for a ragged one (one sequence of 1280 tokens among 127 of 256: 45.9 / 30.9 / 24.5 us; 16 of 1280 among 112
  // of 640: 52.3 / 43.5).  Up to 1024 groups the launch therefore takes the adaptive form (0: two wavefronts per
  // workgroup, each group keeping one or both by its own length against the launch's average): 47.9 us on the equal
  // batch, 28.9 and 40.4 us on the two ragged ones - ahead of both fixed choices there; four wavefronts per workgroup
  // would need 321 registers each and leave room for one workgroup per CU.  Beyond 1024 groups a CU works through
  // eight or more workgroups, a long one has company while it streams, and the wavefronts that would only leave
  // again cost more than they save (205.8 against 190.6 us at 2048 groups): one per group where the table holds at
  // most 2048 tokens, two otherwise.
  if (splits == 1 && groups >= 512 && groups <= 1024) return 0;
  if (groups > 1024 && max_tokens <= 2048) return 1;
  return groups >= 256 ? 2 : (groups > 128 ? 4 : 8);
}

// Revision of paged_decode_kernel's memory access pattern.  Bumped with every change that can alter the bytes a launch
// moves; counter collections name the revision they were taken on, so a stale one is recognised.
#define NV_PAGED_DECODE_KERNEL_TAG "decode-r03d"
extern "C" const char* nv_paged_attn_decode_kernel_tag(void) { return NV_PAGED_DECODE_KERNEL_TAG; }

extern "C" int nv_paged_attn_decode_pick_splits(int B, int Hkv, int max_context) {
  if (B <= 0 || Hkv <= 0) return 1;
  const int64_t groups = (int64_t)B * Hkv;
  // one workgroup of 8 wavefronts per CU already fills the machine: split only while there are fewer groups than CUs
  int splits = (int)((256 + groups - 1) / groups);
  int by_len = (max_context + 255) / 256;                     // at least one 32-token chunk per wavefront
  if (splits > by_len) splits = by_len;
  if (splits > 32) splits = 32;
  if (splits < 1) splits = 1;
  return splits;
}

extern "C" size_t nv_paged_attn_decode_workspace_bytes(int B, int H, int D, int num_splits) {
  if (num_splits <= 1) return 0;
  return (size_t)B * H * num_splits * (D + 2) * sizeof(float);
}

// Argument checks and launches shared by nv_paged_attn_decode (caller's workspace) and nv_paged_attn_decode_static
// (the library's own per-device array).
static int paged_decode_go(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                           const int32_t* block_tables, int bt_stride, const int32_t* context_lens, void* out,
                           int64_t out_row_stride, void* workspace, size_t workspace_bytes, int B, int H, int Hkv,
                           int D, int block_size, float scale, int num_splits, int dtype, nv_stream_t stream) {
  NV_CHECK_ARG(q && k_cache && v_cache && block_tables && context_lens && out, "nv_paged_attn_decode: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_paged_attn_decode: bad dtype %d", dtype);
  NV_CHECK_ARG(D == 64 || D == 128, "nv_paged_attn_decode: head_dim %d not in {64, 128}", D);
  NV_CHECK_ARG(H > 0 && Hkv > 0 && H % Hkv == 0 && H / Hkv <= 16, "nv_paged_attn_decode: H=%d Hkv=%d (group must be <= 16)", H, Hkv);
  NV_CHECK_ARG(block_size > 0 && block_size % 16 == 0, "nv_paged_attn_decode: block_size=%d must be a multiple of 16", block_size);
  NV_CHECK_ARG(q_row_stride % 8 == 0 && out_row_stride % 8 == 0, "nv_paged_attn_decode: row strides must be multiples of 8");
  NV_CHECK_ARG(((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out) % 16 == 0,
               "nv_paged_attn_decode: pointers must be 16-byte aligned");
  NV_CHECK_ARG(B >= 0 && B <= 65535 && Hkv <= 65535, "nv_paged_attn_decode: B=%d too large for one launch", B);
  if (B == 0) return NV_OK;
  NV_CHECK_ARG(bt_stride >= 1, "nv_paged_attn_decode: block table row stride %d < 1", bt_stride);
  if (num_splits <= 0) num_splits = 1;
  const size_t need = nv_paged_attn_decode_workspace_bytes(B, H, D, num_splits);
  NV_CHECK_ARG(need == 0 || (workspace && workspace_bytes >= need),
               "nv_paged_attn_decode: workspace %zu bytes < %zu needed", workspace_bytes, need);
  float* part_o = (float*)workspace;
  float* part_ml = part_o ? part_o + (size_t)B * H * num_splits * D : nullptr;
  const float scale2 = scale * LOG2E;
  hipStream_t s = (hipStream_t)stream;
  dim3 block(256);
  const int nw = decode_waves(D, (int64_t)B * Hkv * num_splits, (int64_t)bt_stride * block_size, num_splits);
  dim3 grid(num_splits, Hkv, B);
  // non-temporal K/V loads: 64-wide heads, and a cache region behind this launch's block tables of at least 8 MB
  bool nt = D == 64 && (int64_t)B * Hkv * bt_stride * block_size * D * 4 >= ((int64_t)8 << 20);
  if (const char* e = getenv("NV_ATTN_NT")) nt = D == 64 && atoi(e) == 1;
#define LAUNCH_DEC_ARGS                                                                                              \
      (const uint16_t*)q, q_row_stride, (const uint16_t*)k_cache, (const uint16_t*)v_cache, block_tables, bt_stride, \
      context_lens, (uint16_t*)out, out_row_stride, part_o, part_ml, H, Hkv, block_size, scale2, num_splits
#define LAUNCH_DEC(TT, DD, WW)                                                                         \
  do {                                                                                                 \
    if (DD == 64 && nt) paged_decode_kernel<TT, DD, WW, DD == 64><<<grid, dim3(WW * 64), 0, s>>>(LAUNCH_DEC_ARGS); \
    else paged_decode_kernel<TT, DD, WW, false><<<grid, dim3(WW * 64), 0, s>>>(LAUNCH_DEC_ARGS);       \
  } while (0)
#define LAUNCH_DEC_T(TT)                                                         \
  do {                                                                           \
    if (D == 128) {                                                              \
      if (nw == 0) paged_decode_kernel<TT, 128, 2, false, true><<<grid, dim3(128), 0, s>>>(LAUNCH_DEC_ARGS); \
      else if (nw == 1) LAUNCH_DEC(TT, 128, 1);                                  \
      else if (nw == 2) LAUNCH_DEC(TT, 128, 2);                                  \
      else if (nw == 4) LAUNCH_DEC(TT, 128, 4);                                  \
      else LAUNCH_DEC(TT, 128, 8);                                               \
    }                                                                            \
    else if (nw == 1) LAUNCH_DEC(TT, 64, 1);                                     \
    else if (nw == 2) LAUNCH_DEC(TT, 64, 2);                                     \
    else if (nw == 4) LAUNCH_DEC(TT, 64, 4);                                     \
    else if (nw == 8) LAUNCH_DEC(TT, 64, 8);                                     \
    else LAUNCH_DEC(TT, 64, 16);                                                 \
  } while (0)
  if (dtype == NV_BF16) LAUNCH_DEC_T(bf16_t); else LAUNCH_DEC_T(f16_t);
#undef LAUNCH_DEC_T
#undef LAUNCH_DEC
#undef LAUNCH_DEC_ARGS
  NV_CHECK_LAUNCH();
  if (num_splits > 1) {
    const int64_t total = (int64_t)B * H;
    dim3 cgrid((unsigned)((total + 3) / 4));
#define LAUNCH_CMB(TT, DD)                                                                                      \
  decode_combine_kernel<TT, DD><<<cgrid, block, 0, s>>>(part_o, part_ml, (uint16_t*)out, out_row_stride, H,     \
                                                        num_splits, total)
    if (dtype == NV_BF16) { if (D == 128) LAUNCH_CMB(bf16_t, 128); else LAUNCH_CMB(bf16_t, 64); }
    else                  { if (D == 128) LAUNCH_CMB(f16_t, 128); else LAUNCH_CMB(f16_t, 64); }
#undef LAUNCH_CMB
    NV_CHECK_LAUNCH();
  }
  return NV_OK;
}

extern "C" int nv_paged_attn_decode(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                                    const int32_t* block_tables, int bt_stride, const int32_t* context_lens,
                                    void* out, int64_t out_row_stride, void* workspace, size_t workspace_bytes, int B,
                                    int H, int Hkv, int D, int block_size, float scale, int num_splits, int dtype,
                                    nv_stream_t stream) {
  return paged_decode_go(q, q_row_stride, k_cache, v_cache, block_tables, bt_stride, context_lens, out, out_row_stride,
                         workspace, workspace_bytes, B, H, Hkv, D, block_size, scale, num_splits, dtype, stream);
}

// ---- the same launch for callers that may not allocate: the split partials live in an array of the code object.
// Every GPU the library runs on holds its own copy of the array, at its own address, until the process ends; the
// name `static_scratch` is only a host-side handle, the device address comes from hipGetSymbolAddress, once per device.
static __device__ __align__(256) unsigned char static_scratch[NV_PAGED_ATTN_STATIC_SCRATCH_BYTES];
#define NV_MAX_DEVICES 64
static void* static_scratch_ptr[NV_MAX_DEVICES];   // device address of static_scratch per device ordinal, 0 = not yet asked

extern "C" int nv_paged_attn_decode_static(const void* q, int64_t q_row_stride, const void* k_cache,
                                           const void* v_cache, const int32_t* block_tables, int bt_stride,
                                           const int32_t* context_lens, void* out, int64_t out_row_stride, int B, int H,
                                           int Hkv, int D, int block_size, float scale, int num_splits, int dtype,
                                           nv_stream_t stream) {
  if (num_splits <= 0) num_splits = 1;
  NV_CHECK_ARG(B >= 0 && H > 0 && (D == 64 || D == 128), "nv_paged_attn_decode_static: bad shape B=%d H=%d D=%d", B, H, D);
  const size_t need = nv_paged_attn_decode_workspace_bytes(B, H, D, num_splits);
  NV_CHECK_ARG(need <= (size_t)NV_PAGED_ATTN_STATIC_SCRATCH_BYTES,
               "nv_paged_attn_decode_static: B=%d H=%d D=%d splits=%d need %zu bytes of split partials, the built-in "
               "array holds %zu: use fewer splits or nv_paged_attn_decode with a caller workspace",
               B, H, D, num_splits, need, (size_t)NV_PAGED_ATTN_STATIC_SCRATCH_BYTES);
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= NV_MAX_DEVICES) {
    nv_set_error("nv_paged_attn_decode_static: no current device (or ordinal %d >= %d)", device, NV_MAX_DEVICES);
    return NV_ERR_LAUNCH;
  }
  void* scratch = static_scratch_ptr[device];
  if (!scratch) {   // first call on this device (make it outside stream capture): resolve this device's copy
    const hipError_t e = hipGetSymbolAddress(&scratch, HIP_SYMBOL(static_scratch));
    if (e != hipSuccess || !scratch) {
      nv_set_error("nv_paged_attn_decode_static: hipGetSymbolAddress failed on device %d: %s", device, hipGetErrorString(e));
      return NV_ERR_LAUNCH;
    }
    static_scratch_ptr[device] = scratch;
  }
  // nothing in the array has to start at zero: every (sequence, head, split) slot the combine kernel reads is
  // written by this call's decode kernel first
  return paged_decode_go(q, q_row_stride, k_cache, v_cache, block_tables, bt_stride, context_lens, out, out_row_stride,
                         scratch, (size_t)NV_PAGED_ATTN_STATIC_SCRATCH_BYTES, B, H, Hkv, D, block_size, scale,
                         num_splits, dtype, stream);
}

extern "C" int nv_attn_prefill(const void* q, int64_t q_row_stride, const void* k, const void* v,
                               int64_t kv_row_stride, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                               const int32_t* block_tables, int bt_stride, void* out, int64_t out_row_stride,
                               int num_seqs, int max_seqlen_q, int H, int Hkv, int D, int block_size, float scale,
                               int dtype, nv_stream_t stream) {
  NV_CHECK_ARG(q && k && v && cu_seqlens_q && cu_seqlens_k && out, "nv_attn_prefill: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_attn_prefill: bad dtype %d", dtype);
  NV_CHECK_ARG(D == 64 || D == 128, "nv_attn_prefill: head_dim %d not in {64, 128}", D);
  NV_CHECK_ARG(H > 0 && Hkv > 0 && H % Hkv == 0, "nv_attn_prefill: H=%d Hkv=%d", H, Hkv);
  NV_CHECK_ARG(!block_tables || (block_size > 0 && block_size % 16 == 0), "nv_attn_prefill: block_size=%d must be a multiple of 16", block_size);
  NV_CHECK_ARG(q_row_stride % 8 == 0 && out_row_stride % 8 == 0 && kv_row_stride % 8 == 0, "nv_attn_prefill: row strides must be multiples of 8");
  NV_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0, "nv_attn_prefill: pointers must be 16-byte aligned");
  NV_CHECK_ARG(num_seqs >= 0 && num_seqs <= 65535 && H <= 65535, "nv_attn_prefill: too many sequences for one launch");
  if (num_seqs == 0 || max_seqlen_q <= 0) return NV_OK;
  const float scale2 = scale * LOG2E;
  hipStream_t s = (hipStream_t)stream;
  // Query tiles per wavefront.  Measured (8 x 1024 tokens, 14/2 heads, D=64): one tile without prefetch 161 us, one
  // tile with prefetch 168, two tiles with prefetch 126, four tiles 155 (400 registers: one wavefront per SIMD).
  // At D=128 two tiles leave no room for the second register set and lose at long sequences, so that width keeps
  // one tile and no prefetch.
  int qtiles = (D == 64 && max_seqlen_q > 64) ? 2 : 1;
  if (const char* e = getenv("NV_PREFILL_QT")) {   // testing aid: forces the tile count
    const int f = atoi(e);
    if (f == 1 || (f == 2 && D == 64)) qtiles = f;
  }
  dim3 grid((max_seqlen_q + 64 * qtiles - 1) / (64 * qtiles), H, num_seqs), block(256);
#define LAUNCH_PRE_Q(TT, DD, PP, QQ, PING)                                                                            \
  prefill_kernel<TT, DD, PP, QQ, PING><<<grid, block, 0, s>>>(                                                        \
      (const uint16_t*)q, q_row_stride, (const uint16_t*)k, (const uint16_t*)v, kv_row_stride, cu_seqlens_q,          \
      cu_seqlens_k, block_tables, bt_stride, (uint16_t*)out, out_row_stride, H, Hkv, block_size, scale2)
#define LAUNCH_PRE(TT, DD, PP)                                                  \
  do {                                                                          \
    if (DD == 64 && qtiles == 2) LAUNCH_PRE_Q(TT, 64, PP, 2, true);             \
    else LAUNCH_PRE_Q(TT, DD, PP, 1, false);                                    \
  } while (0)
  const bool paged = block_tables != nullptr;
  static int use_lds = -1;
  if (use_lds < 0) {
    const char* e = getenv("NV_PREFILL_LDS");   // measurement aid: 0 selects the register-fed kernel
    use_lds = e ? atoi(e) : 1;
  }
  if (use_lds) {
    const int G = H / Hkv;
    const int hw = G >= 3 ? 4 : G;               // query heads per workgroup; the other wavefronts take query groups
    const int groups = (G + hw - 1) / hw;
    NV_CHECK_ARG((int64_t)Hkv * groups <= 65535, "nv_attn_prefill: too many heads for one launch");
#define LAUNCH_LDS(TT, DD, PP, QQ, HH)                                                                              \
  do {                                                                                                              \
    dim3 lgrid((max_seqlen_q + (4 / HH) * 16 * QQ - 1) / ((4 / HH) * 16 * QQ), Hkv * groups, num_seqs);            \
    prefill_lds_kernel<TT, DD, PP, QQ, HH><<<lgrid, block, 0, s>>>(                                                 \
        (const uint16_t*)q, q_row_stride, (const uint16_t*)k, (const uint16_t*)v, kv_row_stride, cu_seqlens_q,      \
        cu_seqlens_k, block_tables, bt_stride, (uint16_t*)out, out_row_stride, H, Hkv, block_size, scale2);         \
  } while (0)
#define LAUNCH_LDS_H(TT, DD, PP, QQ)                                            \
  do {                                                                          \
    if (hw == 4) LAUNCH_LDS(TT, DD, PP, QQ, 4);                                 \
    else if (hw == 2) LAUNCH_LDS(TT, DD, PP, QQ, 2);                            \
    else LAUNCH_LDS(TT, DD, PP, QQ, 1);                                         \
  } while (0)
#define LAUNCH_LDS_T(TT)                                                                                  \
  do {                                                                                                    \
    if (D == 128) { if (paged) LAUNCH_LDS_H(TT, 128, true, NV_PREFILL_QT128); else LAUNCH_LDS_H(TT, 128, false, NV_PREFILL_QT128); } \
    else          { if (paged) LAUNCH_LDS_H(TT, 64, true, NV_PREFILL_QT64); else LAUNCH_LDS_H(TT, 64, false, NV_PREFILL_QT64); }     \
  } while (0)
    if (dtype == NV_BF16) LAUNCH_LDS_T(bf16_t); else LAUNCH_LDS_T(f16_t);
#undef LAUNCH_LDS_T
#undef LAUNCH_LDS_H
#undef LAUNCH_LDS
    NV_CHECK_LAUNCH();
    return NV_OK;
  }
  if (dtype == NV_BF16) {
    if (D == 128) { if (paged) LAUNCH_PRE(bf16_t, 128, true); else LAUNCH_PRE(bf16_t, 128, false); }
    else          { if (paged) LAUNCH_PRE(bf16_t, 64, true); else LAUNCH_PRE(bf16_t, 64, false); }
  } else {
    if (D == 128) { if (paged) LAUNCH_PRE(f16_t, 128, true); else LAUNCH_PRE(f16_t, 128, false); }
    else          { if (paged) LAUNCH_PRE(f16_t, 64, true); else LAUNCH_PRE(f16_t, 64, false); }
  }
#undef LAUNCH_PRE_Q
#undef LAUNCH_PRE
  NV_CHECK_LAUNCH();
  return NV_OK;
}
