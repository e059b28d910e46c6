// Templates of the decode GEMM (see gemm.hip for the design notes); instantiated per dtype in gemm_bf16.hip / gemm_f16.hip.
#pragma once
#include "common.h"
#include "rope_common.h"
#include "sample_parts.h"

#define GEMM_BK 128

// Diagnostic build only (scripts/micro/gemm_probe.hip): cycle stamps per wavefront, written to the `partial` buffer.
#ifdef NV_STAMP
#define STAMP_DECL uint64_t st_t0 = __builtin_amdgcn_s_memtime(), st_a = 0, st_sumA = 0, st_sumBar = 0, st_sumB = 0, st_t1 = 0, st_t2 = 0
#define STAMP(var) var = __builtin_amdgcn_s_memtime()
#define STAMP_ADD(sum, since) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); sum += now_ - since; since = now_; } while (0)
#else
#define STAMP_DECL
#define STAMP(var)
#define STAMP_ADD(sum, since)
#endif

// 8 weights of one lane's fragment -> model dtype, from 8 int4 nibbles (one dword).
// Two masks spread the nibbles into bytes (even elements / odd elements), v_cvt_f32_ubyteN turns each byte into
// an exact float u, and fma(u, s, -8 s) is the correctly rounded fp32 product (u - 8) * s that the CPU dequantiser
// forms (8 s is exact), rounded once more to the model dtype.
template <typename T>
__device__ __forceinline__ typename VecOf<T>::x8 dequant_w4(uint32_t q, float s, float neg8s) {
  typename VecOf<T>::x8 f;
  uint32_t even = q & 0x0f0f0f0fu, odd = (q >> 4) & 0x0f0f0f0fu;
  // hide the masks from the optimiser, or it narrows the byte extracts below to 4-bit field extracts and
  // misses the single-instruction byte-to-float conversions
  asm volatile("" : "+v"(even), "+v"(odd));
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    f[2 * b] = (T)fmaf((float)((even >> (8 * b)) & 0xffu), s, neg8s);
    f[2 * b + 1] = (T)fmaf((float)((odd >> (8 * b)) & 0xffu), s, neg8s);
  }
  return f;
}

// 8 int8 weights (two dwords) -> model dtype: value = q * s.
template <typename T>
__device__ __forceinline__ typename VecOf<T>::x8 dequant_w8(uint32_t lo, uint32_t hi, float s) {
  typename VecOf<T>::x8 f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[e] = (T)((float)(int8_t)(lo >> (8 * e)) * s);
    f[4 + e] = (T)((float)(int8_t)(hi >> (8 * e)) * s);
  }
  return f;
}


// NTW = 16-column tiles per wavefront (a workgroup covers 64*NTW columns): the x fragment read from LDS is reused
// for NTW MFMAs, which divides the LDS read traffic and the x re-reads from L2 by NTW.
// A workgroup has 8 wavefronts = 2 groups of 4: both groups own the same 4*NTW tiles, group 0 multiplies the first
// two 32-deep k-steps of every K-block and group 1 the last two, so each SIMD holds two wavefronts whose
// dequantisation (VALU) and MFMA streams overlap; the two partial accumulators are added through LDS at the end.
// LDS of one workgroup: the double-buffered x tile, reused for the k-group join and the epilogues' staging.
template <int BM, int NTW, int EPI>
constexpr int gemm_lds_bytes() {
  constexpr int TM = BM / 16, RED_REGS = (NTW * TM + 1) / 2;
  constexpr int LDS_X = 2 * BM * GEMM_BK * 2, LDS_RED = RED_REGS * 256 * 16;
  constexpr int LDS_STAGE = (EPI == 3) ? BM * (64 * NTW + 4) * 4 : 0;   // fp32 rows of one head, padded against conflicts
  constexpr int LDS_XR = LDS_X > LDS_RED ? LDS_X : LDS_RED;
  return LDS_XR > LDS_STAGE ? LDS_XR : LDS_STAGE;
}

// The work of one wavefront of a 512-thread workgroup: column group `wid` (0..3) of k-step group `kg` (0..1), NTW
// tiles starting at tile nt0.  The caller owns the LDS block and decides which wavefront is which; all wavefronts of
// a workgroup must run the same BM and EPI (they meet at the same barriers) but may differ in NTW.
// STREAM (epilogues 0 and 4): the weight matrix is far larger than the caches and every byte of it is read once per
// launch by one wavefront (the LM head in a decode step), so its loads carry the non-temporal hint: they do not
// displace what the rest of the step keeps in the Infinity Cache.  Measured on the metric configuration (Qwen2-0.5B
// int4, 128 rows): with the KV walk and the LM head both non-temporal the 179 MB of layer weights stay cache-resident
// from step to step, 1.42-1.44 -> 1.355 ms per step (o/down 6.5 -> 6.0, gate_up 9.9 -> 9.3, q/k/v 7.6 -> 6.9 us per
// launch); either one alone gains nothing.  The logits are stored with the default policy: non-temporal 8-byte
// stores of four lanes per row took the launch from 92 to 110 us (no write combining), 6 us per step more than the
// 39 MB of cache the logits buffer occupies cost.
template <typename T, int WFMT, int BM, int NTW, int EPI, bool STREAM = false, bool RNT = false>
                                                            // RNT: results written through (common.h out_store; EPI 1-3)
                                                            // EPI: 0 = store (+bias), 1 = silu(gate)*up, 2 = fp32 slab per K-split,
                                                            //      3 = q/k/v projection: +bias, head norm, RoPE, KV-cache store
                                                            //      4 = store + the sampler's per-part statistics (LM head)
__device__ __forceinline__ void gemm_body(const uint16_t* __restrict__ x, int64_t x_stride,
                                          const void* __restrict__ w, const uint16_t* __restrict__ scales,
                                          const uint16_t* __restrict__ bias, uint16_t* __restrict__ out,
                                          int64_t out_stride, float* __restrict__ partial, int64_t M, int N,
                                          int K, int split_k, const RopeArgs& ra, const SampleParts& sp,
                                          uint8_t* lds_raw, int wid, int kg, int nt0) {
  typedef typename VecOf<T>::x8 frag_t;
  constexpr int TM = BM / 16;                       // 16-row tiles of x per workgroup
  constexpr int XCH = BM * 16;                      // 16-byte chunks in one x tile
  constexpr int XV = (XCH + 511) / 512;             // chunks per thread
  constexpr int RED_REGS = (NTW * TM + 1) / 2;      // accumulator tiles exchanged per reduction pass
  uint16_t (*lds_x)[BM * GEMM_BK] = reinterpret_cast<uint16_t (*)[BM * GEMM_BK]>(lds_raw);

  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t m0 = (int64_t)blockIdx.y * BM;
  const int nkb_total = K / GEMM_BK;
  const int per = (nkb_total + split_k - 1) / split_k;   // K-blocks of this split: contiguous range
  const int kb_begin = blockIdx.z * per;
  int kb_end = kb_begin + per;
  if (kb_end > nkb_total) kb_end = nkb_total;
  const int ntiles = N >> 4;

  STAMP_DECL;
  f32x4 acc[NTW][TM];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging helpers ------------------------------------------------------
  // Loads are unconditional (rows past M re-read row M-1, tiles past N re-read the last tile; their results are
  // never stored), so the K loop is straight-line code the scheduler can interleave.
  // Addresses are a wave-uniform 64-bit part (scalar registers) plus a 32-bit per-lane byte offset computed once.
  u32x4 xr[XV];
  uint32_t x_lane_off[XV];
#pragma unroll
  for (int v = 0; v < XV; ++v) {
    const int chunk = (tid + 512 * v) & (XCH - 1);     // row = chunk / 16, 16-byte column = chunk % 16
    const int row = chunk >> 4, col = chunk & 15;
    int64_t m = m0 + row;
    m = m < M ? m : M - 1;
    x_lane_off[v] = (uint32_t)((m - m0) * x_stride * 2 + col * 16);
  }
  const char* x_base = reinterpret_cast<const char*>(x + m0 * x_stride);
  auto load_x_chunk = [&](int kb, int v) {
    xr[v] = *reinterpret_cast<const u32x4*>(x_base + (int64_t)kb * (GEMM_BK * 2) + x_lane_off[v]);
  };
  auto store_x_chunk = [&](int buf, int v) {
    const int chunk = tid + 512 * v;
    const int row = chunk >> 4, col = chunk & 15;
    if (XCH % 512 == 0 || chunk < XCH)   // only the 16-row tile leaves threads without a chunk
      *reinterpret_cast<u32x4*>(&lds_x[buf][row * GEMM_BK + ((col ^ (row & 15)) << 3)]) = xr[v];
  };
  auto load_x = [&](int kb) {
#pragma unroll
    for (int v = 0; v < XV; ++v) load_x_chunk(kb, v);
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int v = 0; v < XV; ++v) store_x_chunk(buf, v);
  };
  // raw weights of this wavefront's two k-steps of one K-block per tile (current and prefetched), and group scales
  constexpr bool UNQUANTISED = WFMT == NV_W_DENSE || WFMT == NV_W_DENSE_TILED;
  constexpr int WREG = UNQUANTISED ? 8 : (WFMT == NV_W_INT8 ? 4 : 2);
  uint32_t wr[NTW][WREG], wn[NTW][WREG];
  uint16_t sc[NTW], sn[NTW];
  const uint32_t w_lane_off = (WFMT == NV_W_DENSE) ? (uint32_t)((c * K + 8 * g) * 2)
                              : (WFMT == NV_W_DENSE_TILED) ? (uint32_t)(lane * 16)
                              : (WFMT == NV_W_INT8 ? (uint32_t)(lane * 32 + kg * 16) : (uint32_t)(lane * 16 + kg * 8));
  const uint32_t s_lane_off = (uint32_t)(c * 2);
  // NV_GEMM_WNT=1 (build-time, measurement aid): every weight load carries the non-temporal hint
#ifndef NV_GEMM_WNT
#define NV_GEMM_WNT 0
#endif
  constexpr bool WNT = STREAM || NV_GEMM_WNT == 1;
  auto wload4 = [](const char* p) -> u32x4 {
    if constexpr (WNT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    else return *reinterpret_cast<const u32x4*>(p);
  };
  auto wload2 = [](const char* p) -> u32x2 {
    if constexpr (WNT) return __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    else return *reinterpret_cast<const u32x2*>(p);
  };
  auto load_w = [&](int kb, uint32_t (&dst)[NTW][WREG], uint16_t (&sdst)[NTW]) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int nt = nt0 + j < ntiles ? nt0 + j : ntiles - 1;
      const char* wb8 = reinterpret_cast<const char*>(w);
      if constexpr (WFMT == NV_W_DENSE) {
        const char* wp = wb8 + ((int64_t)16 * nt * K + (int64_t)kb * GEMM_BK + 64 * kg) * 2 + w_lane_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const u32x4 t = wload4(wp + 64 * ks);
          dst[j][4 * ks] = t[0]; dst[j][4 * ks + 1] = t[1]; dst[j][4 * ks + 2] = t[2]; dst[j][4 * ks + 3] = t[3];
        }
      } else if constexpr (WFMT == NV_W_DENSE_TILED) {   // [N/16][K/128][4 k-steps][64 lanes][8 elements]
        const char* wp = wb8 + (((int64_t)nt * nkb_total + kb) * 4 + 2 * kg) * 1024 + w_lane_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const u32x4 t = wload4(wp + 1024 * ks);
          dst[j][4 * ks] = t[0]; dst[j][4 * ks + 1] = t[1]; dst[j][4 * ks + 2] = t[2]; dst[j][4 * ks + 3] = t[3];
        }
      } else if constexpr (WFMT == NV_W_INT8) {   // tiled [N/16][K/128][64 lanes][8 dwords]: dwords 4 kg .. 4 kg + 3
        const u32x4 a = wload4(wb8 + ((int64_t)nt * nkb_total + kb) * 2048 + w_lane_off);
        dst[j][0] = a[0]; dst[j][1] = a[1]; dst[j][2] = a[2]; dst[j][3] = a[3];
      } else {                                    // tiled [N/16][K/128][64 lanes][4 dwords]: dwords 2 kg, 2 kg + 1
        const u32x2 a = wload2(wb8 + ((int64_t)nt * nkb_total + kb) * 1024 + w_lane_off);
        dst[j][0] = a[0]; dst[j][1] = a[1];
      }
      if constexpr (!UNQUANTISED)
        sdst[j] = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(scales) +
                                                     ((int64_t)kb * N + 16 * nt) * 2 + s_lane_off);
    }
  };

  // ---- main loop ---------------------------------------------------------------
  // A wavefront owns two 32-deep k-steps of every K-block.  The operands of the NEXT k-step (x fragments from LDS,
  // weights dequantised on the VALU) are produced while the MFMAs of the CURRENT one run, so the matrix pipe, the
  // VALU and the LDS work at the same time instead of in turn.  Per K-block t:
  //   phase A: fetch/dequantise (t, step 1);  multiply (t, step 0);  store tile t+1 to the other LDS buffer;  barrier
  //   phase B: fetch/dequantise (t+1, step 0) from that buffer;      multiply (t, step 1)
  // One barrier per K-block is enough: tile t+1 replaces tile t-1, whose last reads were issued in phase A of
  // K-block t-1, before that K-block's barrier.  Global loads run one K-block ahead of their use.
  auto read_x = [&](int buf, int ks, frag_t (&xf)[TM]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = 16 * i + c;
      xf[i] = *reinterpret_cast<const frag_t*>(&lds_x[buf][row * GEMM_BK + (((4 * ks + g) ^ c) << 3)]);
    }
  };
  auto dequant = [&](const uint32_t (&raw)[NTW][WREG], const uint16_t (&scl)[NTW], int kl, frag_t (&wf)[NTW]) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      if constexpr (UNQUANTISED) {
        wf[j] = __builtin_bit_cast(frag_t, u32x4{raw[j][4 * kl], raw[j][4 * kl + 1], raw[j][4 * kl + 2], raw[j][4 * kl + 3]});
      } else if constexpr (WFMT == NV_W_INT8) {
        wf[j] = dequant_w8<T>(raw[j][2 * kl], raw[j][2 * kl + 1], bits_to_f32<T>(scl[j]));
      } else {
#ifdef NV_ABL_NODQ
        wf[j] = __builtin_bit_cast(frag_t, u32x4{raw[j][kl], raw[j][kl], (uint32_t)scl[j], raw[j][kl]});
#else
        const float sj = bits_to_f32<T>(scl[j]);
        wf[j] = dequant_w4<T>(raw[j][kl], sj, -8.0f * sj);
#endif
      }
    }
  };
  // keeps a fragment's dequantisation where it was written (next to the MFMAs of the other k-step) instead of
  // letting the optimiser sink it to its first use behind the barrier
  auto pin = [&](frag_t (&wf)[NTW]) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) asm volatile("" : "+v"(wf[j]));
  };
  // multiply one k-step; as soon as the MFMAs of row tile i are issued, x[i] is refilled from LDS with the
  // fragment of the next k-step, so one set of x registers serves both steps and every read has a whole phase to land.
  // With STAGE the wavefront also moves its share of the next x tile in between: one 16-byte chunk per thread goes
  // from registers to the other LDS buffer and the registers are reloaded with the tile after that, so LDS writes
  // (about a third of the LDS read rate on this chip) trickle in under the MFMAs instead of piling up at the barrier.
  constexpr int STAGE_EVERY = TM / XV > 0 ? TM / XV : 1;
  // with three tiles per wavefront there are no registers to spare for staging inside the MFMA stream: the tile is
  // then stored in one go before the barrier
  constexpr bool SPREAD_STAGING = NTW * TM <= 16;
  auto multiply = [&](const frag_t (&wf)[NTW], frag_t (&xf)[TM], bool refill, int buf, int ks, bool stage, int kb_ahead) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        acc[j][i] = mfma16(wf[j], xf[i], acc[j][i]);   // D[n][m]: lane (c, g) reg r = out[m = 16 i + c][n = 4 g + r]
#ifdef NV_ABL_NOLDSR
      if (false) {
#else
      if (refill) {
#endif
        const int row = 16 * i + c;
        xf[i] = *reinterpret_cast<const frag_t*>(&lds_x[buf][row * GEMM_BK + (((4 * ks + g) ^ c) << 3)]);
      }
      if (stage && i % STAGE_EVERY == 0 && i / STAGE_EVERY < XV) {
        store_x_chunk(buf ^ 1, i / STAGE_EVERY);
#ifndef NV_ABL_NOX
        load_x_chunk(kb_ahead, i / STAGE_EVERY);
#endif
      }
    }
  };
  // scheduling pattern of a phase: per row tile NTW MFMAs, each followed by a few VALU instructions of the
  // dequantisation, then the LDS refill
  constexpr int VALU_PER_MFMA = UNQUANTISED ? 0 : (28 + TM - 1) / TM;
  auto interleave = [&](bool refill, bool stage) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (VALU_PER_MFMA > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
      }
      if (refill) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (stage && i % STAGE_EVERY == 0 && i / STAGE_EVERY < XV) {
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
  };

  const int nkb = kb_end - kb_begin;
  if (nkb > 0) {
    frag_t xa[TM], wa[NTW], wb[NTW];
    load_x(kb_begin);
    load_w(kb_begin, wr, sc);
    store_x(0);
    if (nkb > 1) {
      load_x(kb_begin + 1);
      load_w(kb_begin + 1, wn, sn);
    }
    __syncthreads();
    read_x(0, 2 * kg, xa);
    dequant(wr, sc, 0, wa);
    STAMP(st_t1);
    STAMP(st_a);

    // One K-block with both phases.  wc/sc_c: raw weights of tile t; wx/sc_x: those of tile t+1 (in flight since
    // the previous K-block).  After its last use wc is refilled with tile t+2, so two register sets alternate
    // roles from K-block to K-block and no loaded value is ever copied (a copy would wait for its load).
    auto kblock = [&](int t, uint32_t (&wc)[NTW][WREG], uint16_t (&sc_c)[NTW], uint32_t (&wx)[NTW][WREG],
                      uint16_t (&sc_x)[NTW]) {
      const int buf = t & 1;
      const int ahead = kb_begin + (t + 2 < nkb ? t + 2 : nkb - 1);   // the last tile is simply fetched again
      // phase A: multiply (t, step 0); prepare (t, step 1); x tile t+1 (in registers since K-block t-1) -> LDS,
      // registers <- x tile t+2
      dequant(wc, sc_c, 1, wb);
      multiply(wa, xa, true, buf, 2 * kg + 1, SPREAD_STAGING, ahead);
      pin(wb);
      interleave(true, SPREAD_STAGING);
      if (!SPREAD_STAGING) store_x(buf ^ 1);
      STAMP_ADD(st_sumA, st_a);
#ifndef NV_ABL_NOBAR
      __syncthreads();
#endif
      STAMP_ADD(st_sumBar, st_a);
      // phase B: multiply (t, step 1); prepare (t+1, step 0)
      if (!SPREAD_STAGING) load_x(ahead);
      dequant(wx, sc_x, 0, wa);
#ifndef NV_ABL_NOW
      load_w(ahead, wc, sc_c);
#endif
      multiply(wb, xa, true, buf ^ 1, 2 * kg, false, 0);
      pin(wa);
      interleave(true, false);
      STAMP_ADD(st_sumB, st_a);
    };
    // last K-block: nothing left to fetch
    auto last_kblock = [&](uint32_t (&wc)[NTW][WREG], uint16_t (&sc_c)[NTW]) {
      dequant(wc, sc_c, 1, wb);
      multiply(wa, xa, true, (nkb - 1) & 1, 2 * kg + 1, false, 0);
      interleave(true, false);
      multiply(wb, xa, false, 0, 0, false, 0);
    };
    int t = 0;
    for (; t + 2 <= nkb - 1; t += 2) {
      kblock(t, wr, sc, wn, sn);
      kblock(t + 1, wn, sn, wr, sc);
    }
    if (t < nkb - 1) {
      kblock(t, wr, sc, wn, sn);
      last_kblock(wn, sn);
    } else {
      last_kblock(wr, sc);
    }
  }
  STAMP(st_t2);
  // the rows' sampling temperatures: asked for here so that the loads land during the join below
  float c2[EPI == 4 ? TM : 1];
  if constexpr (EPI == 4) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t m = m0 + 16 * i + c;
      c2[i] = sp.temps[m < M ? m : M - 1];
    }
  }
  __syncthreads();

  // ---- join the two k-step groups' accumulators through LDS, half of the tiles per pass.  In the first pass
  // group 1 hands its half to group 0; in the second the direction is reversed, so each group ends up owning the
  // finished sums of one half of the tiles and both run the epilogue (the RoPE epilogue needs whole rows in one
  // place: there group 0 takes both halves).  a + b == b + a bit for bit, so the owner does not change the result.
#ifndef NV_ABL_NOJOIN   // measurement aid: without the join each k-step group stores its own partial sums (wrong results)
  {
    f32x4* red = reinterpret_cast<f32x4*>(lds_raw);   // [RED_REGS][256 threads of the column group]
    const int t4 = wid * 64 + lane;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int giver = (EPI == 3 || pass == 0) ? 1 : 0;
      if (kg == giver) {
#pragma unroll
        for (int r = 0; r < RED_REGS; ++r) {
          const int idx = pass * RED_REGS + r;
          if (idx < NTW * TM) red[r * 256 + t4] = acc[idx / TM][idx % TM];
        }
      }
      __syncthreads();
      if (kg != giver) {
#pragma unroll
        for (int r = 0; r < RED_REGS; ++r) {
          const int idx = pass * RED_REGS + r;
          if (idx < NTW * TM) acc[idx / TM][idx % TM] += red[r * 256 + t4];
        }
      }
      __syncthreads();
    }
  }
#endif
  if (EPI == 3 && kg == 1) return;
  // tile idx = j * TM + i is finished by group 0 if it travelled in the first pass, else by group 1
  auto owns = [&](int idx) { return EPI == 3 || ((idx < RED_REGS) == (kg == 0)); };

  // ---- RoPE epilogue: the workgroup's 64 NTW columns are exactly one head (blockIdx.x = head slot).  The rotation
  // pairs element d with d + D/2, which live in different wavefronts, so the finished rows (+bias, rounded to the
  // model dtype as a plain store would) go through LDS once and are then handled D/8 lanes per row.
  if constexpr (EPI == 3) {
    constexpr int DD = 64 * NTW, LD = DD + 4;
    float* stage = reinterpret_cast<float*>(lds_raw);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int col = 16 * (wid * NTW + j) + 4 * g;
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if (bias) b4 = unpack4<T>(*reinterpret_cast<const u32x2*>(bias + (int64_t)blockIdx.x * DD + col));
#pragma unroll
      for (int i = 0; i < TM; ++i)
        *reinterpret_cast<f32x4*>(&stage[(16 * i + c) * LD + col]) = round4<T>(acc[j][i] + b4);
    }
    __syncthreads();   // the four wavefronts that are still here
    constexpr int LPH = DD / 8, ROWS = 256 / LPH;
    const int sub = tid & (LPH - 1);
#pragma unroll
    for (int r0 = 0; r0 < BM; r0 += ROWS) {
      const int r = r0 + tid / LPH;
      const bool live = r < BM && m0 + r < M;
      const float* src = &stage[(r < BM ? r : 0) * LD];
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(src + 4 * sub);
      const f32x4 x2 = *reinterpret_cast<const f32x4*>(src + DD / 2 + 4 * sub);
      const int64_t t = live ? m0 + r : 0;
      rope_finish<T, RNT>(x1, x2, ra, (int)blockIdx.x, t, sub, out + t * out_stride + (int64_t)blockIdx.x * DD, true, live);
    }
    return;
  }

  // ---- logits + sampler statistics: the store of EPI 0, and for every row the (maximum, mass) of each 64-column
  // part (sample_parts.h), formed from the values as stored.  A part is the four tiles of column-group wavefronts:
  // tile (wid, j) of this workgroup is slot wid * NTW + j; tile statistics meet in LDS and one thread per (row, part)
  // joins them.
  if constexpr (EPI == 4) {
    constexpr int SLOTS = 4 * NTW;
    float* st_m = reinterpret_cast<float*>(lds_raw);     // [BM][SLOTS]; the reduction's last reads are behind a barrier
    float* st_s = st_m + BM * SLOTS;
    float* st_c = st_s + BM * SLOTS;                     // [BM] log2(e) / T of each row, for the join below
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      c2[i] = sample_exp_scale(c2[i]);
      if (wave == 0 && g == 0) st_c[16 * i + c] = c2[i];
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int nt = nt0 + j;
      const bool tile_live = nt < ntiles;
      const int nb = 16 * (tile_live ? nt : ntiles - 1) + 4 * g;
      float b4[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias) {
        const u32x2 t = *reinterpret_cast<const u32x2*>(bias + nb);
        b4[0] = lo_f32<T>(t[0]); b4[1] = hi_f32<T>(t[0]); b4[2] = lo_f32<T>(t[1]); b4[3] = hi_f32<T>(t[1]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (!owns(j * TM + i)) continue;                  // wave-uniform
        const int64_t m = m0 + 16 * i + c;
        u32x2 o;
        o[0] = pack2<T>(acc[j][i][0] + b4[0], acc[j][i][1] + b4[1]);
        o[1] = pack2<T>(acc[j][i][2] + b4[2], acc[j][i][3] + b4[3]);
#ifndef NV_ABL_NOSTORE
        if (m < M && tile_live) *reinterpret_cast<u32x2*>(out + m * out_stride + nb) = o;
#endif
        float v[4] = {lo_f32<T>(o[0]), hi_f32<T>(o[0]), lo_f32<T>(o[1]), hi_f32<T>(o[1])};
        if (!tile_live) v[0] = v[1] = v[2] = v[3] = -INFINITY;
        float tm, ts;
        sample_tile16(v, c2[i], tm, ts);
        if (g == 0) {
          const int slot = (16 * i + c) * SLOTS + wid * NTW + j;
          st_m[slot] = tm;
          st_s[slot] = ts;
        }
      }
    }
    __syncthreads();
    constexpr int PARTS = NTW;                            // 64-column parts per workgroup
    for (int item = tid; item < BM * PARTS; item += 512) {
      const int r = item / PARTS, q = item - r * PARTS;
      const int64_t m = m0 + r;
      const int part = blockIdx.x * PARTS + q;
      if (m >= M || part >= sp.P) continue;
      float tm[4], ts[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        tm[t] = st_m[r * SLOTS + 4 * q + t];
        ts[t] = st_s[r * SLOTS + 4 * q + t];
      }
      float pm, psum;
      sample_join4(tm, ts, st_c[r], pm, psum);
      const int64_t o = m * sp.P + part;
      sp.m[o] = pm;
      sp.s[o] = psum;
    }
    return;
  }

  // ---- epilogue: per tile, lane holds columns 16 nt + 4g .. +3 of rows m0 + 16 i + c
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int nt = nt0 + j;
    if (nt >= ntiles) continue;
    const int nb = 16 * nt + 4 * g;
    if constexpr (EPI == 0) {
      float b4[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias) {
        const u32x2 t = *reinterpret_cast<const u32x2*>(bias + nb);
        b4[0] = lo_f32<T>(t[0]); b4[1] = hi_f32<T>(t[0]); b4[2] = lo_f32<T>(t[1]); b4[3] = hi_f32<T>(t[1]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int64_t m = m0 + 16 * i + c;
        if (m < M && owns(j * TM + i)) {
          u32x2 o;
          o[0] = pack2<T>(acc[j][i][0] + b4[0], acc[j][i][1] + b4[1]);
          o[1] = pack2<T>(acc[j][i][2] + b4[2], acc[j][i][3] + b4[3]);
#ifdef NV_ABL_NOSTORE
          if (o[0] == 0x12345678u && o[1] == 0x9abcdef0u)   // keeps the arithmetic alive, practically never stores
#endif
          *reinterpret_cast<u32x2*>(out + m * out_stride + nb) = o;
        }
      }
    } else if constexpr (EPI == 1) {
      // rows of W were interleaved so that a lane's 4 columns are (gate j, gate j+1, up j, up j+1):
      // out[m][j] = dtype(dtype(silu(dtype(gate))) * dtype(up)), the roundings eager PyTorch performs
      const int j0 = 8 * nt + 2 * g;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int64_t m = m0 + 16 * i + c;
        if (m < M && owns(j * TM + i)) {
          float r[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float gt = bits_to_f32<T>(f32_to_bits<T>(acc[j][i][e]));
            const float up = bits_to_f32<T>(f32_to_bits<T>(acc[j][i][2 + e]));
            // silu(g) = g / (1 + e^-g) through the hardware exp2 and reciprocal (about 1 ulp of fp32 each, far below
            // the 16-bit rounding that follows); libm expf plus an IEEE division cost ~35 instructions per element
            // and made this epilogue a tenth of the launch
            const float sl = bits_to_f32<T>(f32_to_bits<T>(
                gt * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gt))));
            r[e] = sl * up;
          }
          out_store<RNT>(reinterpret_cast<uint32_t*>(out + m * out_stride + j0), pack2<T>(r[0], r[1]));
        }
      }
    } else {
      float* slab = partial + (int64_t)blockIdx.z * M * N;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int64_t m = m0 + 16 * i + c;
        if (m < M && owns(j * TM + i)) out_store<RNT>(reinterpret_cast<f32x4*>(slab + m * N + nb), acc[j][i]);
      }
    }
  }
#ifdef NV_STAMP
  if (EPI != 2 && partial && lane == 0) {
    uint64_t* st = reinterpret_cast<uint64_t*>(partial) + ((int64_t)blockIdx.x * 8 + wave) * 8;
    st[0] = st_t1 - st_t0; st[1] = st_sumA; st[2] = st_sumBar; st[3] = st_sumB; st[4] = st_t2 - st_t1;
    st[5] = __builtin_amdgcn_s_memtime() - st_t2; st[6] = st_t0; st[7] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// NTW = 16-column tiles per wavefront (a workgroup covers 64*NTW columns): the x fragment read from LDS is reused
// for NTW MFMAs, which divides the LDS read traffic and the x re-reads from L2 by NTW.
// A workgroup has 8 wavefronts = 2 groups of 4: both groups own the same 4*NTW tiles, group 0 multiplies the first
// two 32-deep k-steps of every K-block and group 1 the last two, so each SIMD holds two wavefronts whose
// dequantisation (VALU) and MFMA streams overlap; the two partial accumulators are added through LDS at the end.
template <typename T, int WFMT, int BM, int NTW, int EPI, bool STREAM = false, bool RNT = false>
__global__ __launch_bounds__(512) void gemm_kernel(const uint16_t* __restrict__ x, int64_t x_stride,
                                                   const void* __restrict__ w, const uint16_t* __restrict__ scales,
                                                   const uint16_t* __restrict__ bias, uint16_t* __restrict__ out,
                                                   int64_t out_stride, float* __restrict__ partial, int64_t M, int N,
                                                   int K, int split_k, RopeArgs ra, SampleParts sp) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[gemm_lds_bytes<BM, NTW, EPI>()];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wid = wave & 3, kg = wave >> 2;         // column group of 4 wavefronts, k-step group
  gemm_body<T, WFMT, BM, NTW, EPI, STREAM, RNT>(x, x_stride, w, scales, bias, out, out_stride, partial, M, N, K, split_k, ra,
                                           sp, lds_raw, wid, kg, (blockIdx.x * 4 + wid) * NTW);
}

// 128 rows x 160 columns per workgroup: column groups 0 and 1 take three tiles per wavefront, groups 2 and 3 two.
// Where N / 192 leaves a quarter of the CUs idle and N / 128 needs a second round (Qwen2-7B gate_up: 198 or 296
// workgroups for 256 CUs), ten tiles per workgroup fill the chip in one round (237).  The second k-step group takes
// the column groups in reverse order, so each SIMD (wavefronts w and w + 4) carries 3 + 2 tiles.
#define GEMM_MIXED_TILES 10
template <typename T, int WFMT, int EPI, bool RNT = false>
__global__ __launch_bounds__(512) void gemm_mixed_kernel(const uint16_t* __restrict__ x, int64_t x_stride,
                                                         const void* __restrict__ w, const uint16_t* __restrict__ scales,
                                                         const uint16_t* __restrict__ bias, uint16_t* __restrict__ out,
                                                         int64_t out_stride, float* __restrict__ partial, int64_t M,
                                                         int N, int K, int split_k) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[gemm_lds_bytes<128, 3, EPI>()];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kg = wave >> 2, wid = kg ? 3 - (wave & 3) : (wave & 3);
  const int first = blockIdx.x * GEMM_MIXED_TILES;
  if (wid < 2)
    gemm_body<T, WFMT, 128, 3, EPI, false, RNT>(x, x_stride, w, scales, bias, out, out_stride, partial, M, N, K, split_k, RopeArgs{},
                                    SampleParts{}, lds_raw, wid, kg, first + 3 * wid);
  else
    gemm_body<T, WFMT, 128, 2, EPI, false, RNT>(x, x_stride, w, scales, bias, out, out_stride, partial, M, N, K, split_k, RopeArgs{},
                                    SampleParts{}, lds_raw, wid, kg, first + 6 + 2 * (wid - 2));
}

// Sum the split-K slabs in slab order, add the bias, round to the model dtype.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial,
                                                            const uint16_t* __restrict__ bias,
                                                            uint16_t* __restrict__ out, int64_t out_stride, int64_t M,
                                                            int N, int split_k) {
  const int nvec = N >> 2;
  const int64_t total = M * nvec;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = v / nvec;
    const int n = (int)(v - m * nvec) << 2;
    f32x4 s = *reinterpret_cast<const f32x4*>(partial + m * N + n);
    for (int z = 1; z < split_k; ++z) s += *reinterpret_cast<const f32x4*>(partial + ((int64_t)z * M + m) * N + n);
    if (bias) {
      const u32x2 t = *reinterpret_cast<const u32x2*>(bias + n);
      s[0] += lo_f32<T>(t[0]); s[1] += hi_f32<T>(t[0]); s[2] += lo_f32<T>(t[1]); s[3] += hi_f32<T>(t[1]);
    }
    u32x2 o;
    o[0] = pack2<T>(s[0], s[1]);
    o[1] = pack2<T>(s[2], s[3]);
    *reinterpret_cast<u32x2*>(out + m * out_stride + n) = o;
  }
}


#ifndef NV_GEMM_ARGS
#define NV_GEMM_ARGS
struct GemmArgs {
  const void* x; int64_t xs; const void* w; const void* scales; const void* bias; void* out; int64_t os;
  float* partial; int64_t M; int N; int K; int split_k; int wfmt; int bm; int ntw; int epi; hipStream_t stream;
  RopeArgs rope;   // epi == 3 only
  SampleParts stats;   // epi == 4 only
  int mixed;       // 1: 128 rows x 160 columns per workgroup (gemm_mixed_kernel; quantised weights, epi 0..2)
  int wide;        // 1: the 32x32x16 kernel (int4, 128-row tiles, 192 columns per workgroup), gemm_wide.h
  int stream_once; // 1: non-temporal weight loads (streaming-layout weights, epi 0 / 4, one row tile)
  int result_nt;   // 1: write-through result stores (epi 1 / 2 / 3): the result-store policy
};
#endif

template <typename T, int WFMT, int NTW, int EPI, bool STREAM = false, bool RNT = false>
static void gemm_go_bm(const GemmArgs& a, dim3 grid) {
#define GO(BMV)                                                                                               \
  gemm_kernel<T, WFMT, BMV, NTW, EPI, STREAM, RNT><<<grid, dim3(512), 0, a.stream>>>(                         \
      (const uint16_t*)a.x, a.xs, a.w, (const uint16_t*)a.scales, (const uint16_t*)a.bias, (uint16_t*)a.out,  \
      a.os, a.partial, a.M, a.N, a.K, a.split_k, a.rope, a.stats)
  switch (a.bm) {
    case 16: GO(16); break;
    case 32: GO(32); break;
    case 64: GO(64); break;
    default: GO(128); break;
  }
#undef GO
}

// WT selects which half of the instantiations a translation unit holds: the write-through epilogues of the result-store
// policy (epi 1-3 with a.result_nt set) live in gemm_<dtype>_wt.hip, everything else in gemm_<dtype>.hip, so the two
// halves compile side by side.  gemm_wants_wt() is the routing rule both sides agree on.
static inline bool gemm_wants_wt(const GemmArgs& a) {
  if (!a.result_nt || a.wide) return false;
  if (a.mixed) return a.epi == 1 || a.epi == 2;
  if (a.wfmt == NV_W_DENSE_TILED && a.stream_once && (a.epi == 0 || a.epi == 4)) return false;
  return a.epi == 1 || a.epi == 2 || (a.epi == 3 && a.ntw <= 2);
}

template <typename T, int WFMT, int NTW, bool WT>
static void gemm_go_epi(const GemmArgs& a, dim3 grid) {
  if constexpr (WT) {   // the result-store policy: the same kernels with write-through stores
    if (a.epi == 1) gemm_go_bm<T, WFMT, NTW, 1, false, true>(a, grid);
    else if (a.epi == 2) gemm_go_bm<T, WFMT, NTW, 2, false, true>(a, grid);
    else if constexpr (NTW <= 2) gemm_go_bm<T, WFMT, NTW, 3, false, true>(a, grid);
  } else {
    if (a.epi == 0) gemm_go_bm<T, WFMT, NTW, 0>(a, grid);
    else if (a.epi == 1) gemm_go_bm<T, WFMT, NTW, 1>(a, grid);
    else if (a.epi == 2) gemm_go_bm<T, WFMT, NTW, 2>(a, grid);
    else if (a.epi == 4) {   // the LM head is never quantised
      if constexpr (WFMT == NV_W_DENSE || WFMT == NV_W_DENSE_TILED) gemm_go_bm<T, WFMT, NTW, 4>(a, grid);
    }
    else if constexpr (NTW <= 2) gemm_go_bm<T, WFMT, NTW, 3>(a, grid);   // a head is 64 or 128 columns
  }
}

template <typename T, int WFMT, bool WT>
static void gemm_go_mixed(const GemmArgs& a) {
  const int ntiles = a.N / 16;
  dim3 grid((ntiles + GEMM_MIXED_TILES - 1) / GEMM_MIXED_TILES, (unsigned)((a.M + 127) / 128), a.split_k);
#define GO(EPIV, RNTV)                                                                                           \
  gemm_mixed_kernel<T, WFMT, EPIV, RNTV><<<grid, dim3(512), 0, a.stream>>>(                                      \
      (const uint16_t*)a.x, a.xs, a.w, (const uint16_t*)a.scales, (const uint16_t*)a.bias, (uint16_t*)a.out,     \
      a.os, a.partial, a.M, a.N, a.K, a.split_k)
  if constexpr (WT) {
    if (a.epi == 1) GO(1, true); else GO(2, true);
  } else {
    if (a.epi == 0) GO(0, false); else if (a.epi == 1) GO(1, false); else GO(2, false);
  }
#undef GO
}

// One dtype's instantiations: dense takes 1 or 2 tiles per wavefront (16 raw registers per tile), quantised 1 or 3.
// QUANT selects the other split of a dtype's plain instantiations: the quantised formats (int4 / int8, the 160-column
// kernel) live in gemm_<dtype>_q.hip, the dense ones in gemm_<dtype>.hip.  The write-through unit holds both.
static inline bool gemm_is_quant(const GemmArgs& a) {
  return a.mixed || a.wfmt == NV_W_INT8 || a.wfmt == NV_W_INT4;
}

template <typename T, bool WT, bool DENSE_PART = true, bool QUANT_PART = true>
static void gemm_dispatch(const GemmArgs& a) {
  if constexpr (QUANT_PART) {
    if (a.mixed) {
      if (a.wfmt == NV_W_INT8) gemm_go_mixed<T, NV_W_INT8, WT>(a); else gemm_go_mixed<T, NV_W_INT4, WT>(a);
      return;
    }
  }
  const int ntiles = a.N / 16;
  const int per_block = 4 * a.ntw;
  dim3 grid((ntiles + per_block - 1) / per_block, (unsigned)((a.M + a.bm - 1) / a.bm), a.split_k);
  if constexpr (DENSE_PART) {
    if (a.wfmt == NV_W_DENSE) {
      if (a.ntw == 1) gemm_go_epi<T, NV_W_DENSE, 1, WT>(a, grid); else gemm_go_epi<T, NV_W_DENSE, 2, WT>(a, grid);
      return;
    }
    if constexpr (!WT) {
      if (a.wfmt == NV_W_DENSE_TILED && a.stream_once && (a.epi == 0 || a.epi == 4)) {
        if (a.epi == 4) { if (a.ntw == 1) gemm_go_bm<T, NV_W_DENSE_TILED, 1, 4, true>(a, grid); else gemm_go_bm<T, NV_W_DENSE_TILED, 2, 4, true>(a, grid); }
        else            { if (a.ntw == 1) gemm_go_bm<T, NV_W_DENSE_TILED, 1, 0, true>(a, grid); else gemm_go_bm<T, NV_W_DENSE_TILED, 2, 0, true>(a, grid); }
        return;
      }
    }
    if (a.wfmt == NV_W_DENSE_TILED) {
      if (a.ntw == 1) gemm_go_epi<T, NV_W_DENSE_TILED, 1, WT>(a, grid); else gemm_go_epi<T, NV_W_DENSE_TILED, 2, WT>(a, grid);
      return;
    }
  }
  if constexpr (QUANT_PART) {
    if (a.wfmt == NV_W_INT8) {
      if (a.ntw == 1) gemm_go_epi<T, NV_W_INT8, 1, WT>(a, grid);
      else if (a.ntw == 2) gemm_go_epi<T, NV_W_INT8, 2, WT>(a, grid);
      else gemm_go_epi<T, NV_W_INT8, 3, WT>(a, grid);
    } else if (a.wfmt == NV_W_INT4) {
      if (a.ntw == 1) gemm_go_epi<T, NV_W_INT4, 1, WT>(a, grid);
      else if (a.ntw == 2) gemm_go_epi<T, NV_W_INT4, 2, WT>(a, grid);
      else gemm_go_epi<T, NV_W_INT4, 3, WT>(a, grid);
    }
  }
}

template <typename T>
static void reduce_dispatch(const GemmArgs& a) {
  const int64_t total = a.M * (a.N / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  splitk_reduce_kernel<T><<<dim3((unsigned)blocks), dim3(256), 0, a.stream>>>(
      a.partial, (const uint16_t*)a.bias, (uint16_t*)a.out, a.os, a.M, a.N, a.split_k);
}
