// The int4 GEMM for wide outputs at 65..128 rows per tile, on v_mfma_f32_32x32x16.
//
// Why a second kernel.  With 16x16x32 MFMAs every MFMA keeps the SIMD's VALU port for about 8 of its 16 cycles, and
// the int4 -> 16-bit conversion adds 3 VALU instructions (12 cycles) per MFMA at 128 rows: the VALU port, not the
// matrix pipe, sets the pace (about 21 cycles per MFMA at best, 32 measured).  A 32x32x16 MFMA does twice the work in
// 29 cycles and holds the port for about 3 (scripts/micro/coissue32.hip: with 6 VALU instructions between MFMAs it
// still runs at the matrix pipe's rate), so the same conversion work disappears under the MFMAs.
//
// Shape.  A workgroup has 12 wavefronts = 6 column groups x 2 k-groups (3 wavefronts per SIMD, 170 registers each) and
// covers 128 rows x 192 columns, like the 16x16 kernel with three tiles per wavefront, so the grid and the CU fill are
// the same.  A wavefront owns 32 columns (two adjacent 16-row tiles of the packed weight layout) and all 128 rows:
// four 32x32 accumulators.  Per K-block of 128 it multiplies four 16-deep k-steps (its k-group's half).  Weights
// stream HBM -> VGPR as in the 16x16 kernel (same packed bytes: lane (n, h) takes the dwords that lane (n & 15, g = h)
// and (n & 15, g = 2 + h) of the tile hold), x is staged in LDS in the same swizzled layout, one barrier per K-block.
#pragma once
#include "gemm_impl.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

typedef __attribute__((address_space(3))) void wide_lds_void_t;
typedef __attribute__((address_space(1))) const void wide_global_void_t;

#define GEMM_WIDE_NCG 6
#define GEMM_WIDE_THREADS (GEMM_WIDE_NCG * 2 * 64)

// Diagnostic builds (scripts/micro/gemm_probe.hip): WABL_NOLDS drops the x fragment refills, WABL_NODQ the
// dequantisation arithmetic, WABL_NODMA the x tile fetch.  Results are then wrong; only the timing is of interest.
#ifdef WABL_NODQ
#define WIDE_DQ(T, q, s, n8) __builtin_bit_cast(typename VecOf<T>::x8, u32x4{q, q ^ __float_as_uint(s), q, q})
#else
#define WIDE_DQ(T, q, s, n8) dequant_w4<T>(q, s, n8)
#endif
template <typename T, int EPI>   // EPI: 0 = store (+bias), 1 = silu(gate)*up, 2 = fp32 slab per K-split
__global__ __launch_bounds__(GEMM_WIDE_THREADS) void gemm_wide_w4_kernel(
    const uint16_t* __restrict__ x, int64_t x_stride, const void* __restrict__ w, const uint16_t* __restrict__ scales,
    const uint16_t* __restrict__ bias, uint16_t* __restrict__ out, int64_t out_stride, float* __restrict__ partial,
    int64_t M, int N, int K, int split_k) {
  typedef typename VecOf<T>::x8 frag_t;
  constexpr int BM = 128, NCG = GEMM_WIDE_NCG, MB = BM / 32, GROUP = NCG * 64;
  constexpr int XV = (BM / 4 + 2 * NCG - 1) / (2 * NCG);   // 1 KB pieces of an x tile per wavefront
  constexpr int LDS_X = 2 * BM * GEMM_BK * 2, LDS_RED = (MB / 2) * 4 * GROUP * 16;
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[LDS_X > LDS_RED ? LDS_X : LDS_RED];
  uint16_t (*lds_x)[BM * GEMM_BK] = reinterpret_cast<uint16_t (*)[BM * GEMM_BK]>(lds_raw);

  const int tid = threadIdx.x, lane = tid & 63, j32 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wid = wave % NCG, kg = wave / NCG;
  const int64_t m0 = (int64_t)blockIdx.y * BM;
  const int nkb_total = K / GEMM_BK;
  const int per = (nkb_total + split_k - 1) / split_k;
  const int kb_begin = blockIdx.z * per;
  int kb_end = kb_begin + per;
  if (kb_end > nkb_total) kb_end = nkb_total;
  const int nkb = kb_end - kb_begin;
  const int nblocks = N >> 5;                        // 32-column blocks
  const int nb = blockIdx.x * NCG + wid;             // this wavefront's block
  const int nbc = nb < nblocks ? nb : nblocks - 1;   // loads are unconditional on a clamped block, stores are guarded

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- x staging by LDS-DMA (global_load_lds: no staging registers).  One wave-instruction fills 1 KB of LDS, lane l
  // at byte 16 l: four rows of the tile, 16 slots each.  The swizzled layout (slot p of row r holds the row's 16-byte
  // chunk p ^ (r & 15)) is produced through the per-lane SOURCE address.  The tile has 32 such pieces for 12
  // wavefronts: three rounds, the last of which wraps around and fetches 4 pieces a second time (the same bytes), so
  // every wavefront runs the same straight-line code.
  uint32_t x_lane_off[XV];
  int x_piece[XV];
#pragma unroll
  for (int v = 0; v < XV; ++v) {
    x_piece[v] = (wave + 2 * NCG * v) & (BM / 4 - 1);
    const int row = 4 * x_piece[v] + (lane >> 4), p = lane & 15;
    int64_t m = m0 + row;
    m = m < M ? m : M - 1;
    x_lane_off[v] = (uint32_t)((m - m0) * x_stride * 2 + ((p ^ (row & 15)) << 4));
  }
  const char* x_base = reinterpret_cast<const char*>(x + m0 * x_stride);
  auto fetch_x = [&](int kb, int buf) {
#pragma unroll
    for (int v = 0; v < XV; ++v)
#ifndef WABL_NODMA
      __builtin_amdgcn_global_load_lds((wide_global_void_t*)(x_base + (int64_t)kb * (GEMM_BK * 2) + x_lane_off[v]),
                                       (wide_lds_void_t*)&lds_x[buf][x_piece[v] * 512], 16, 0, 0);
#else
      asm volatile("" ::"v"(x_lane_off[v]), "s"(kb), "s"(buf));
#endif
  };

  // ---- weights: lane (n = j32, h) of the 32x32x16 A operand needs W[32 nb + n][16 s + 8 h .. + 8] for k-step s of the
  // K-block.  In the packed layout [N/16][K/128][64 lanes][4 dwords] (lane (c, g), dword ks = 8 nibbles of
  // W[16 nt + c][128 kb + 32 ks + 8 g ..]) that is tile 2 nb + (n >> 4), lane (n & 15, g = 2 (s & 1) + h), dword s >> 1.
  // This k-group's four steps s = 4 kg .. 4 kg + 3 are dwords 2 kg and 2 kg + 1 of lanes g = h ("lo") and g = 2 + h ("hi").
  const char* w_lane = reinterpret_cast<const char*>(w) +
                       ((int64_t)(2 * nbc + (j32 >> 4)) * nkb_total) * 1024 + ((j32 & 15) + 16 * h) * 16 + kg * 8;
  const char* s_lane = reinterpret_cast<const char*>(scales) + ((int64_t)32 * nbc + j32) * 2;
  auto load_w = [&](int kb, u32x2& lo, u32x2& hi, uint16_t& sc) {
    const char* p = w_lane + (int64_t)kb * 1024;
    lo = *reinterpret_cast<const u32x2*>(p);
    hi = *reinterpret_cast<const u32x2*>(p + 512);
    sc = *reinterpret_cast<const uint16_t*>(s_lane + (int64_t)kb * N * 2);
  };
  // x fragment of the B operand: lane (m = j32, h) holds x[32 mb + m][16 s + 8 h .. + 8] = 16-byte chunk 2 s + h of the row
  auto read_b = [&](int buf, int s, int i) -> frag_t {
    const int row = 32 * i + j32;
    return *reinterpret_cast<const frag_t*>(&lds_x[buf][row * GEMM_BK + (((2 * s + h) ^ (row & 15)) << 3)]);
  };

  if (nkb > 0) {
    u32x2 lo0, hi0, lo1, hi1;
    uint16_t sc0, sc1;
    frag_t xb[MB], wa0, wa1;
    // multiply one k-step; each x fragment register is refilled with the NEXT step's fragment right after the MFMA
    // that read it, so one set of fragment registers serves all steps and every read has a whole step to land
    auto mma_refill = [&](const frag_t& a, int buf, int s) {
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        acc[i] = mfma32(a, xb[i], acc[i]);
#ifndef WABL_NOLDS
        xb[i] = read_b(buf, s, i);
#endif
      }
    };
    fetch_x(kb_begin, 0);
    load_w(kb_begin, lo0, hi0, sc0);
    __syncthreads();   // waits for this wavefront's DMA pieces (vmcnt) before the barrier
    {
      const int second = kb_begin + (nkb > 1 ? 1 : 0);   // a single K-block is simply fetched twice
      fetch_x(second, 1);
      asm volatile("" ::: "memory");
      load_w(second, lo1, hi1, sc1);
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) xb[i] = read_b(0, 4 * kg, i);
    {
      const float s = bits_to_f32<T>(sc0);
      wa0 = WIDE_DQ(T, lo0[0], s, -8.0f * s);
    }
    // One K-block, steps s0..s3 of this k-group.  lo_c/hi_c/sc_c: this K-block's raw weights; lo_n/..: the next one's
    // (in flight since the previous K-block); after their last use the current registers take K-block t + 2.
    // The barrier sits before step 3, whose refills are the first reads of the next tile.
    auto kblock = [&](int t, u32x2& lo_c, u32x2& hi_c, uint16_t& sc_c, u32x2& lo_n, uint16_t& sc_n) {
      const int buf = t & 1;
      const float s = bits_to_f32<T>(sc_c), n8 = -8.0f * s;
      wa1 = WIDE_DQ(T, hi_c[0], s, n8);
      mma_refill(wa0, buf, 4 * kg + 1);
      wa0 = WIDE_DQ(T, lo_c[1], s, n8);
      mma_refill(wa1, buf, 4 * kg + 2);
      wa1 = WIDE_DQ(T, hi_c[1], s, n8);
      mma_refill(wa0, buf, 4 * kg + 3);
      // Barrier: tile t + 1 (DMA issued one K-block ago) must be in place and nobody may still read tile t.  A plain
      // __syncthreads() would also wait for the weight loads issued after that DMA, i.e. every K-block would last at
      // least one HBM latency; the three weight loads are the youngest vector-memory operations of this wavefront, so
      // vmcnt(3) waits for the DMA pieces only and lets the weights keep flying until their registers are read.
#ifndef WABL_NOBAR
      asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
      const int ahead = kb_begin + (t + 2 < nkb ? t + 2 : nkb - 1);   // the last tile is simply fetched again
      fetch_x(ahead, buf);
      asm volatile("" ::: "memory");   // keep the DMA older than the weight loads: the vmcnt above counts on it
      const float s2 = bits_to_f32<T>(sc_n);
      wa0 = WIDE_DQ(T, lo_n[0], s2, -8.0f * s2);
      load_w(ahead, lo_c, hi_c, sc_c);
      asm volatile("" ::: "memory");
      mma_refill(wa1, buf ^ 1, 4 * kg);
    };
    auto last_kblock = [&](u32x2& lo_c, u32x2& hi_c, uint16_t& sc_c) {
      const int buf = (nkb - 1) & 1;
      const float s = bits_to_f32<T>(sc_c), n8 = -8.0f * s;
      wa1 = WIDE_DQ(T, hi_c[0], s, n8);
      mma_refill(wa0, buf, 4 * kg + 1);
      wa0 = WIDE_DQ(T, lo_c[1], s, n8);
      mma_refill(wa1, buf, 4 * kg + 2);
      wa1 = WIDE_DQ(T, hi_c[1], s, n8);
      mma_refill(wa0, buf, 4 * kg + 3);
#pragma unroll
      for (int i = 0; i < MB; ++i) acc[i] = mfma32(wa1, xb[i], acc[i]);
    };
    int t = 0;
    for (; t + 2 <= nkb - 1; t += 2) {
      kblock(t, lo0, hi0, sc0, lo1, sc1);
      kblock(t + 1, lo1, hi1, sc1, lo0, sc0);
    }
    if (t < nkb - 1) {
      kblock(t, lo0, hi0, sc0, lo1, sc1);
      last_kblock(lo1, hi1, sc1);
    } else {
      last_kblock(lo0, hi0, sc0);
    }
  }
  __syncthreads();

  // ---- join the two k-groups through LDS: group 1 hands over its row blocks 0..1 (group 0 finishes those), then
  // group 0 hands over 2..3.  a + b == b + a bit for bit, so which group adds does not change the result.
  {
    f32x4* red = reinterpret_cast<f32x4*>(lds_raw);   // [8 registers quads][GROUP threads]
    const int tg = tid - kg * GROUP;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int giver = pass == 0 ? 1 : 0;
      if (kg == giver) {
#pragma unroll
        for (int i = 0; i < MB / 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x16& a = acc[pass * (MB / 2) + i];
            red[(i * 4 + q) * GROUP + tg] = f32x4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
          }
      }
      __syncthreads();
      if (kg != giver) {
#pragma unroll
        for (int i = 0; i < MB / 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = red[(i * 4 + q) * GROUP + tg];
            f32x16& a = acc[pass * (MB / 2) + i];
            a[4 * q] += v[0]; a[4 * q + 1] += v[1]; a[4 * q + 2] += v[2]; a[4 * q + 3] += v[3];
          }
      }
      __syncthreads();
    }
  }
  if (nb >= nblocks) return;

  // ---- epilogue: group kg owns row blocks 2 kg, 2 kg + 1.  Accumulator register 4 q + e of lane (m = j32, h) is
  // out[32 mb + m][32 nb + 8 q + 4 h + e]: four consecutive columns per q.
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {   // static indices only: a run-time index would move the accumulators to scratch
    const int64_t m = m0 + 32 * mb + j32;
    if (mb / (MB / 2) != kg || m >= M) continue;
    const f32x16& a = acc[mb];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 32 * nb + 8 * q + 4 * h;
      if constexpr (EPI == 0) {
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
          const u32x2 t = *reinterpret_cast<const u32x2*>(bias + n);
          b4[0] = lo_f32<T>(t[0]); b4[1] = hi_f32<T>(t[0]); b4[2] = lo_f32<T>(t[1]); b4[3] = hi_f32<T>(t[1]);
        }
        u32x2 o;
        o[0] = pack2<T>(a[4 * q] + b4[0], a[4 * q + 1] + b4[1]);
        o[1] = pack2<T>(a[4 * q + 2] + b4[2], a[4 * q + 3] + b4[3]);
        *reinterpret_cast<u32x2*>(out + m * out_stride + n) = o;
      } else if constexpr (EPI == 1) {
        // the rows of W were interleaved so that four consecutive columns are (gate j, gate j + 1, up j, up j + 1)
        float r[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float gt = bits_to_f32<T>(f32_to_bits<T>(a[4 * q + e]));
          const float up = bits_to_f32<T>(f32_to_bits<T>(a[4 * q + 2 + e]));
          const float sl = bits_to_f32<T>(f32_to_bits<T>(
              gt * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gt))));
          r[e] = sl * up;
        }
        *reinterpret_cast<uint32_t*>(out + m * out_stride + (n >> 1)) = pack2<T>(r[0], r[1]);
      } else {
        float* slab = partial + (int64_t)blockIdx.z * M * N;
        *reinterpret_cast<f32x4*>(slab + m * N + n) = f32x4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
      }
    }
  }
}

template <typename T>
static void gemm_wide_dispatch(const GemmArgs& a) {
  const int nblocks = a.N / 32;
  dim3 grid((nblocks + GEMM_WIDE_NCG - 1) / GEMM_WIDE_NCG, (unsigned)((a.M + 127) / 128), a.split_k);
#define GO(EPIV)                                                                                                   \
  gemm_wide_w4_kernel<T, EPIV><<<grid, dim3(GEMM_WIDE_THREADS), 0, a.stream>>>(                                    \
      (const uint16_t*)a.x, a.xs, a.w, (const uint16_t*)a.scales, (const uint16_t*)a.bias, (uint16_t*)a.out, a.os, \
      a.partial, a.M, a.N, a.K, a.split_k)
  if (a.epi == 0) GO(0); else if (a.epi == 1) GO(1); else GO(2);
#undef GO
}
