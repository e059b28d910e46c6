// out[M, N] = x[M, K] . W[N, K]^T (+ bias) for the decode step's linears on
// gfx950: dense bf16/fp16 weights, or weight-only int8 / int4 (group 128)
// dequantised in registers, all through MFMA 16x16x32 with fp32 accumulation.
//
// Shape of the problem: M is the number of tokens in the step (1..512), the
// weight matrix is streamed from HBM exactly once, x is small and re-read from
// L2 by every workgroup.  So a workgroup (4 wavefronts) owns a 64-column slab
// of N and up to 128 rows of M, and walks K in blocks of 128 (= one
// quantisation group):
//   * x[BM x 128] is staged in LDS (16-byte chunks XOR-swizzled by row so the
//     fragment reads are bank-conflict free), double-buffered: the next block
//     is loaded to registers while the current one is multiplied;
//   * each wavefront owns 16 of the 64 columns and reads ITS weights straight
//     from HBM to registers (no other wavefront needs them): 16-byte loads,
//     for int4 one load brings a lane's fragments for the whole 128-deep block
//     (the packed layout is tiled for exactly that, see layers/quant.py);
//   * weights are the MFMA A operand, x the B operand, so a lane ends up with
//     4 consecutive output columns of one row and stores them as 8 bytes.
// split_k > 1 gives each K-slice its own workgroup; fp32 partial slabs go to
// the workspace and a second kernel sums them in a fixed order (bit-reproducible).
#include "common.h"

#define GEMM_BN 64
#define GEMM_BK 128

// 8 weights of one lane's fragment -> model dtype, from 8 int4 nibbles (one dword).
// value = (u - 8) * s, evaluated as fma(128 + u, s, -136 s): 136 s is exact in
// fp32, so this is the correctly rounded fp32 product the CPU dequantiser forms.
template <typename T>
__device__ __forceinline__ typename VecOf<T>::x8 dequant_w4(uint32_t q, float s, float neg136s) {
  typename VecOf<T>::x8 f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint32_t bits = 0x43000000u | (((q >> (4 * e)) & 0xfu) << 16);
    f[e] = (T)fmaf(__uint_as_float(bits), s, neg136s);
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

template <typename T, int WFMT, int BM, int EPI>   // EPI: 0 = store (+bias), 1 = silu(gate)*up, 2 = fp32 slab per K-split
__global__ __launch_bounds__(256) void gemm_kernel(const uint16_t* __restrict__ x, int64_t x_stride,
                                                   const void* __restrict__ w, const uint16_t* __restrict__ scales,
                                                   const uint16_t* __restrict__ bias, uint16_t* __restrict__ out,
                                                   int64_t out_stride, float* __restrict__ partial, int64_t M, int N,
                                                   int K, int split_k) {
  typedef typename VecOf<T>::x8 frag_t;
  constexpr int TM = BM / 16;              // 16-row tiles of x per workgroup
  constexpr int XV = BM * 16 / 256;        // 16-byte chunks of the x tile per thread (BM >= 16)
  __shared__ __attribute__((aligned(16))) uint16_t lds_x[2][BM * GEMM_BK];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, c = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * GEMM_BN;
  const int64_t m0 = (int64_t)blockIdx.y * BM;
  const int nkb_total = K / GEMM_BK;
  // K-blocks of this split: contiguous range
  const int per = (nkb_total + split_k - 1) / split_k;
  const int kb_begin = blockIdx.z * per;
  int kb_end = kb_begin + per;
  if (kb_end > nkb_total) kb_end = nkb_total;
  const int nt = (n0 >> 4) + wid;          // this wavefront's 16-column tile
  const int ncol = n0 + 16 * wid;

  f32x4 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging helpers ------------------------------------------------------
  u32x4 xr[XV];
  auto load_x = [&](int kb) {
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int chunk = tid + 256 * v;     // row = chunk / 16, 16-byte column = chunk % 16
      const int row = chunk >> 4, col = chunk & 15;
      const int64_t m = m0 + row;
      if (m < M)
        xr[v] = *reinterpret_cast<const u32x4*>(x + m * x_stride + (int64_t)kb * GEMM_BK + col * 8);
      else
        xr[v] = u32x4{0, 0, 0, 0};
    }
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int chunk = tid + 256 * v;
      const int row = chunk >> 4, col = chunk & 15;
      *reinterpret_cast<u32x4*>(&lds_x[buf][row * GEMM_BK + ((col ^ (row & 15)) << 3)]) = xr[v];
    }
  };
  // weights of one K-block for this lane: raw registers
  constexpr int WREG = (WFMT == NV_W_DENSE) ? 16 : (WFMT == NV_W_INT8 ? 8 : 4);
  uint32_t wr[WREG];
  uint16_t sc_bits = 0;
  auto load_w = [&](int kb) {
    if constexpr (WFMT == NV_W_DENSE) {
      const uint16_t* wp = reinterpret_cast<const uint16_t*>(w) + (int64_t)(ncol + c) * K + (int64_t)kb * GEMM_BK + 8 * g;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(wp + 32 * ks);
        wr[4 * ks] = t[0]; wr[4 * ks + 1] = t[1]; wr[4 * ks + 2] = t[2]; wr[4 * ks + 3] = t[3];
      }
    } else if constexpr (WFMT == NV_W_INT8) {
      // tiled [N/16][K/128][64 lanes][8 dwords]
      const u32x4* wp = reinterpret_cast<const u32x4*>(w) + (((int64_t)nt * nkb_total + kb) * 64 + lane) * 2;
      const u32x4 a = wp[0], b = wp[1];
      wr[0] = a[0]; wr[1] = a[1]; wr[2] = a[2]; wr[3] = a[3];
      wr[4] = b[0]; wr[5] = b[1]; wr[6] = b[2]; wr[7] = b[3];
      sc_bits = scales[(int64_t)kb * N + ncol + c];
    } else {
      // tiled [N/16][K/128][64 lanes][4 dwords]
      const u32x4 a = reinterpret_cast<const u32x4*>(w)[((int64_t)nt * nkb_total + kb) * 64 + lane];
      wr[0] = a[0]; wr[1] = a[1]; wr[2] = a[2]; wr[3] = a[3];
      sc_bits = scales[(int64_t)kb * N + ncol + c];
    }
  };

  if (kb_begin < kb_end) {
    load_x(kb_begin);
    load_w(kb_begin);
    store_x(0);
  }
  __syncthreads();

  for (int kb = kb_begin; kb < kb_end; ++kb) {
    const int buf = (kb - kb_begin) & 1;
    // fragments of the current block's weights (registers are reused by the prefetch below)
    frag_t wf[4];
    if constexpr (WFMT == NV_W_DENSE) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        wf[ks] = __builtin_bit_cast(frag_t, u32x4{wr[4 * ks], wr[4 * ks + 1], wr[4 * ks + 2], wr[4 * ks + 3]});
    } else if constexpr (WFMT == NV_W_INT8) {
      const float s = bits_to_f32<T>(sc_bits);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wf[ks] = dequant_w8<T>(wr[2 * ks], wr[2 * ks + 1], s);
    } else {
      const float s = bits_to_f32<T>(sc_bits);
      const float neg136s = -136.0f * s;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wf[ks] = dequant_w4<T>(wr[ks], s, neg136s);
    }
    const bool more = kb + 1 < kb_end;
    if (more) {
      load_x(kb + 1);
      load_w(kb + 1);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = 16 * i + c;
        const frag_t xf = *reinterpret_cast<const frag_t*>(&lds_x[buf][row * GEMM_BK + (((4 * ks + g) ^ c) << 3)]);
        acc[i] = mfma16(wf[ks], xf, acc[i]);   // D[n][m]: lane (c, g) reg r = out[m = 16 i + c][n = 4 g + r]
      }
    }
    if (more) store_x(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds columns ncol + 4g .. +3 of rows m0 + 16 i + c
  const int nb = ncol + 4 * g;
  if constexpr (EPI == 0) {
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
      const u32x2 t = *reinterpret_cast<const u32x2*>(bias + nb);
      b4[0] = lo_f32<T>(t[0]); b4[1] = hi_f32<T>(t[0]); b4[2] = lo_f32<T>(t[1]); b4[3] = hi_f32<T>(t[1]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t m = m0 + 16 * i + c;
      if (m < M) {
        u32x2 o;
        o[0] = pack2<T>(acc[i][0] + b4[0], acc[i][1] + b4[1]);
        o[1] = pack2<T>(acc[i][2] + b4[2], acc[i][3] + b4[3]);
        *reinterpret_cast<u32x2*>(out + m * out_stride + nb) = o;
      }
    }
  } else if constexpr (EPI == 1) {
    // rows of W were interleaved so that a lane's 4 columns are (gate j, gate j+1, up j, up j+1):
    // out[m][j] = dtype(dtype(silu(dtype(gate))) * dtype(up)), the roundings eager PyTorch performs
    const int j0 = (ncol >> 1) + 2 * g;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t m = m0 + 16 * i + c;
      if (m < M) {
        float r[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float gt = bits_to_f32<T>(f32_to_bits<T>(acc[i][e]));
          const float up = bits_to_f32<T>(f32_to_bits<T>(acc[i][2 + e]));
          const float sl = bits_to_f32<T>(f32_to_bits<T>(gt / (1.0f + expf(-gt))));
          r[e] = sl * up;
        }
        *reinterpret_cast<uint32_t*>(out + m * out_stride + j0) = pack2<T>(r[0], r[1]);
      }
    }
  } else {
    float* slab = partial + (int64_t)blockIdx.z * M * N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t m = m0 + 16 * i + c;
      if (m < M) *reinterpret_cast<f32x4*>(slab + m * N + nb) = acc[i];
    }
  }
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

extern "C" size_t nv_gemm_workspace_bytes(int64_t M, int N, int split_k) {
  if (split_k < 1) split_k = 1;
  return (size_t)split_k * (size_t)M * (size_t)N * sizeof(float);
}

template <typename T, int WFMT, int EPI>
static void launch_gemm(int bm, dim3 grid, hipStream_t s, const void* x, int64_t xs, const void* w, const void* scales,
                        const void* bias, void* out, int64_t os, float* partial, int64_t M, int N, int K, int split_k) {
#define GO(BMV)                                                                                                        \
  gemm_kernel<T, WFMT, BMV, EPI><<<grid, dim3(256), 0, s>>>((const uint16_t*)x, xs, w, (const uint16_t*)scales,      \
                                                            (const uint16_t*)bias, (uint16_t*)out, os, partial, M, N, \
                                                            K, split_k)
  switch (bm) {
    case 16: GO(16); break;
    case 32: GO(32); break;
    case 64: GO(64); break;
    default: GO(128); break;
  }
#undef GO
}

template <typename T, int WFMT>
static void launch_gemm_epi(int epi, int bm, dim3 grid, hipStream_t s, const void* x, int64_t xs, const void* w,
                            const void* scales, const void* bias, void* out, int64_t os, float* partial, int64_t M,
                            int N, int K, int split_k) {
  if (epi == 0) launch_gemm<T, WFMT, 0>(bm, grid, s, x, xs, w, scales, bias, out, os, partial, M, N, K, split_k);
  else if (epi == 1) launch_gemm<T, WFMT, 1>(bm, grid, s, x, xs, w, scales, bias, out, os, partial, M, N, K, split_k);
  else launch_gemm<T, WFMT, 2>(bm, grid, s, x, xs, w, scales, bias, out, os, partial, M, N, K, split_k);
}

extern "C" int nv_gemm(const void* x, int64_t x_row_stride, const void* w, const void* scales, const void* bias,
                       void* out, int64_t out_row_stride, int64_t M, int N, int K, int wfmt, int split_k,
                       void* workspace, size_t workspace_bytes, int flags, int dtype, nv_stream_t stream) {
  const bool silu = (flags & NV_GEMM_SILU_MUL) != 0, slabs_out = (flags & NV_GEMM_PARTIAL_OUT) != 0;
  NV_CHECK_ARG(x && w && (out || slabs_out), "nv_gemm: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_gemm: bad dtype %d", dtype);
  NV_CHECK_ARG(wfmt == NV_W_DENSE || wfmt == NV_W_INT8 || wfmt == NV_W_INT4, "nv_gemm: bad weight format %d", wfmt);
  NV_CHECK_ARG((flags & ~(NV_GEMM_SILU_MUL | NV_GEMM_PARTIAL_OUT)) == 0 && !(silu && slabs_out), "nv_gemm: unsupported flags %d", flags);
  NV_CHECK_ARG(N > 0 && N % GEMM_BN == 0, "nv_gemm: N=%d must be a multiple of %d", N, GEMM_BN);
  NV_CHECK_ARG(K > 0 && K % GEMM_BK == 0, "nv_gemm: K=%d must be a multiple of %d", K, GEMM_BK);
  NV_CHECK_ARG(wfmt == NV_W_DENSE || scales, "nv_gemm: quantised weights need scales");
  NV_CHECK_ARG(x_row_stride % 8 == 0 && out_row_stride % 4 == 0, "nv_gemm: bad row strides");
  NV_CHECK_ARG(((uintptr_t)x | (uintptr_t)w) % 16 == 0 && (uintptr_t)out % 8 == 0 && (!bias || (uintptr_t)bias % 8 == 0),
               "nv_gemm: misaligned pointer");
  NV_CHECK_ARG(M >= 0, "nv_gemm: M=%lld", (long long)M);
  NV_CHECK_ARG(!silu || !bias, "nv_gemm: the silu*up epilogue takes no bias");
  if (M == 0) return NV_OK;
  if (split_k < 1) split_k = 1;
  if (split_k > K / GEMM_BK) split_k = K / GEMM_BK;
  if (silu) split_k = 1;
  const bool use_slabs = slabs_out || split_k > 1;
  const size_t need = use_slabs ? (size_t)split_k * (size_t)M * (size_t)N * sizeof(float) : 0;
  NV_CHECK_ARG(need == 0 || (workspace && workspace_bytes >= need && (uintptr_t)workspace % 16 == 0),
               "nv_gemm: workspace %zu bytes < %zu needed", workspace_bytes, need);
  const int bm = M <= 16 ? 16 : (M <= 32 ? 32 : (M <= 64 ? 64 : 128));
  const int64_t mtiles = (M + bm - 1) / bm;
  NV_CHECK_ARG(mtiles <= 65535 && split_k <= 65535, "nv_gemm: M=%lld too large for one launch", (long long)M);
  dim3 grid(N / GEMM_BN, (unsigned)mtiles, split_k);
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)workspace;
  const int epi = use_slabs ? 2 : (silu ? 1 : 0);
#define DISPATCH(TT)                                                                                                   \
  do {                                                                                                                 \
    if (wfmt == NV_W_DENSE)                                                                                            \
      launch_gemm_epi<TT, NV_W_DENSE>(epi, bm, grid, s, x, x_row_stride, w, scales, bias, out, out_row_stride, partial, M, N, K, split_k); \
    else if (wfmt == NV_W_INT8)                                                                                        \
      launch_gemm_epi<TT, NV_W_INT8>(epi, bm, grid, s, x, x_row_stride, w, scales, bias, out, out_row_stride, partial, M, N, K, split_k);  \
    else                                                                                                               \
      launch_gemm_epi<TT, NV_W_INT4>(epi, bm, grid, s, x, x_row_stride, w, scales, bias, out, out_row_stride, partial, M, N, K, split_k);  \
  } while (0)
  if (dtype == NV_BF16) DISPATCH(bf16_t); else DISPATCH(f16_t);
#undef DISPATCH
  NV_CHECK_LAUNCH();
  if (use_slabs && !slabs_out) {
    const int64_t total = M * (N / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (dtype == NV_BF16)
      splitk_reduce_kernel<bf16_t><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(partial, (const uint16_t*)bias,
                                                                                (uint16_t*)out, out_row_stride, M, N, split_k);
    else
      splitk_reduce_kernel<f16_t><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(partial, (const uint16_t*)bias,
                                                                               (uint16_t*)out, out_row_stride, M, N, split_k);
    NV_CHECK_LAUNCH();
  }
  return NV_OK;
}
