// out[M, N] = x[M, K] . W[N, K]^T (+ bias) for the decode step's linears on
// gfx950: dense bf16/fp16 weights, or weight-only int8 / int4 (group 128)
// dequantised in registers, all through MFMA 16x16x32 with fp32 accumulation.
//
// Shape of the problem: M is the number of tokens in the step (1..512), the
// weight matrix is streamed from HBM exactly once, x is small and re-read from
// L2 by every workgroup.  So a workgroup (4 wavefronts) owns a 64-column slab
// of N (times the tiles-per-wavefront factor, see below) and up to 128 rows of M, and walks K in blocks of 128 (= one
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
//
// Tiles per wavefront.  With one 16-column tile per wavefront every x fragment read from LDS feeds one MFMA and
// the kernel is LDS-bound; with 3 (quantised) or 2 (dense) tiles the same read feeds 3 (2) MFMAs and x is re-read
// from L2 a third (half) as often.  The host picks the factor that leaves the fewest idle CUs.
#include "common.h"
#include <stdlib.h>

#define GEMM_BK 128

#include "rope_common.h"
#include "sample_parts.h"

#define GEMM_MIXED_TILES 10   /* 16-column tiles per workgroup of gemm_mixed_kernel (gemm_impl.h) */

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
void nv_gemm_launch_bf16(const GemmArgs& a);
void nv_gemm_launch_f16(const GemmArgs& a);
void nv_gemm_reduce_bf16(const GemmArgs& a);
void nv_gemm_reduce_f16(const GemmArgs& a);
void nv_gemm_wide_launch_bf16(const GemmArgs& a);
void nv_gemm_wide_launch_f16(const GemmArgs& a);

extern "C" size_t nv_gemm_workspace_bytes(int64_t M, int N, int split_k) {
  if (split_k < 1) split_k = 1;
  return (size_t)split_k * (size_t)M * (size_t)N * sizeof(float);
}

static bool unquantised(int wfmt) { return wfmt == NV_W_DENSE || wfmt == NV_W_DENSE_TILED; }

// ---- launch planning -----------------------------------------------------------
// Estimated duration (microseconds) of one launch with `ntw` tiles per wavefront and `split_k` K-splits: rounds of
// workgroups over the 256 CUs x (K-blocks per workgroup x time per K-block + fixed cost), plus the traffic of the
// fp32 slabs (written here, read by the consumer).  The per-K-block times were measured on MI355X at BM = 128
// (int4: 0.70 / 1.15 / 2.3 us for 1 / 2 / 3 tiles per wavefront) and scale roughly with the rows per workgroup.
static int block_rows(int64_t M) { return M <= 16 ? 16 : (M <= 32 ? 32 : (M <= 64 ? 64 : 128)); }

// Rows per workgroup.  Short K loops (<= 5 K-blocks per workgroup, one tile per wavefront) are dominated by the
// prologue and epilogue: 64-row tiles need under 128 registers, so two workgroups share a CU and hide each other's
// fixed cost (measured at batch 128: o 4.5 -> 3.8 us, down 7.0 -> 6.0 us).  Long loops lose by it (the co-resident
// workgroups repeat the dequantisation: 7B down 27 -> 35 us, LM head 90 -> 126 us) and keep 128 rows.
static int plan_rows(int64_t M, int K, int ntw, int split_k) {
  const int kb_per_block = (K / GEMM_BK + split_k - 1) / split_k;
  return (M > 64 && ntw == 1 && kb_per_block <= 5) ? 64 : block_rows(M);
}

static double plan_cost(int64_t M, int N, int K, int wfmt, int ntw, int split_k, bool slabs) {
  static const double t_quant[4] = {0, 0.70, 1.15, 2.3}, t_dense[3] = {0, 0.60, 1.0};
  const int bm = plan_rows(M, K, ntw, split_k);
  const bool paired = M > 64 && bm == 64;            // two workgroups per CU
  const double t_kb = (unquantised(wfmt) ? t_dense[ntw] : t_quant[ntw]) * (0.4 + 0.6 * bm / 128.0) * (paired ? 1.4 : 1.0);
  const int64_t mtiles = (M + bm - 1) / bm;
  const int ntiles = N / 16, nkb = K / GEMM_BK;
  const int64_t blocks = (int64_t)((ntiles + 4 * ntw - 1) / (4 * ntw)) * mtiles * split_k;
  const int64_t slots = paired ? 512 : 256;
  const int64_t rounds = (blocks + slots - 1) / slots;
  const int kb_per_block = (nkb + split_k - 1) / split_k;
  double us = rounds * (kb_per_block * t_kb + (paired ? 3.0 : 4.0));
  if (slabs) us += 2.0 * split_k * (double)M * N * 4.0 / 4.0e6;   // slab bytes at ~4 TB/s, written and read once
  if (slabs && split_k > 8) us += 1.0;   // the consumer sums slabs in batches of 8 loads: a ninth costs it a round trip
  return us;
}

// 1 where a launch reads a streaming-layout (unquantised) matrix of at least 64 MB exactly once - a single row
// tile, no K split: the LM head of a decode step.  Its weight loads then carry the non-temporal hint (gemm_body,
// STREAM), leaving the Infinity Cache to the layer weights.  With several row tiles (a prefill's LM
// head) the tiles re-read the weights from the caches, which the hint would defeat.  NV_GEMM_STREAM=0 / 1 forces it.
static int plan_stream_once(const GemmArgs& a) {
  if (a.wfmt != NV_W_DENSE_TILED || a.split_k != 1 || (a.epi != 0 && a.epi != 4)) return 0;
  int on = a.M <= a.bm && (int64_t)a.N * a.K * 2 >= ((int64_t)64 << 20);
  if (const char* e = getenv("NV_GEMM_STREAM")) on = atoi(e) == 1;
  return on;
}

// Tiles per wavefront for a given split: dense weights take 1 or 2 (16 raw registers per tile), quantised 1 to 3.
static int pick_ntw(int64_t M, int N, int K, int wfmt, int split_k, bool slabs) {
  const int max_ntw = unquantised(wfmt) ? 2 : 3;
  int best = 1;
  double best_us = plan_cost(M, N, K, wfmt, 1, split_k, slabs);
  for (int ntw = 2; ntw <= max_ntw; ++ntw) {
    const double us = plan_cost(M, N, K, wfmt, ntw, split_k, slabs);
    if (us < best_us) { best_us = us; best = ntw; }
  }
  return best;
}

extern "C" int nv_gemm_pick_split_k(int64_t M, int N, int K, int wfmt) {
  if (M <= 0 || N <= 0 || K < GEMM_BK) return 1;
  const int max_ntw = unquantised(wfmt) ? 2 : 3, nkb = K / GEMM_BK;
  if (const char* e = getenv("NV_GEMM_SPLIT_K")) {   // measurement aid: "K:N:split[,K:N:split...]" forces those shapes' split
    for (const char* p = e; *p;) {
      char* end;
      const long k = strtol(p, &end, 10);
      if (*end != ':') break;
      const long n = strtol(end + 1, &end, 10);
      if (*end != ':') break;
      const long sp = strtol(end + 1, &end, 10);
      if (k == K && n == N && sp >= 1 && sp <= nkb) return (int)sp;
      if (*end != ',') break;
      p = end + 1;
    }
  }
  int best = 1;
  double best_us = 1e30;
  for (int split = 1; split <= nkb && split <= 32; ++split) {
    for (int ntw = 1; ntw <= max_ntw; ++ntw) {
      const double us = plan_cost(M, N, K, wfmt, ntw, split, split > 1);
      if (us < best_us - 1e-9) { best_us = us; best = split; }
    }
  }
  return best;
}

extern "C" int nv_gemm(const void* x, int64_t x_row_stride, const void* w, const void* scales, const void* bias,
                       void* out, int64_t out_row_stride, int64_t M, int N, int K, int wfmt, int split_k,
                       void* workspace, size_t workspace_bytes, int flags, int dtype, nv_stream_t stream) {
  const bool silu = (flags & NV_GEMM_SILU_MUL) != 0, slabs_out = (flags & NV_GEMM_PARTIAL_OUT) != 0;
  NV_CHECK_ARG(x && w && (out || slabs_out), "nv_gemm: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_gemm: bad dtype %d", dtype);
  NV_CHECK_ARG(wfmt >= NV_W_DENSE && wfmt <= NV_W_DENSE_TILED, "nv_gemm: bad weight format %d", wfmt);
  NV_CHECK_ARG((flags & ~(NV_GEMM_SILU_MUL | NV_GEMM_PARTIAL_OUT)) == 0 && !(silu && slabs_out), "nv_gemm: unsupported flags %d", flags);
  NV_CHECK_ARG(N > 0 && N % 16 == 0, "nv_gemm: N=%d must be a multiple of 16", N);
  NV_CHECK_ARG(K > 0 && K % GEMM_BK == 0, "nv_gemm: K=%d must be a multiple of %d", K, GEMM_BK);
  NV_CHECK_ARG(unquantised(wfmt) || scales, "nv_gemm: quantised weights need scales");
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
  GemmArgs a;
  a.x = x; a.xs = x_row_stride; a.w = w; a.scales = scales; a.bias = bias; a.out = out; a.os = out_row_stride;
  a.partial = (float*)workspace; a.M = M; a.N = N; a.K = K; a.split_k = split_k; a.wfmt = wfmt;
  a.ntw = pick_ntw(M, N, K, wfmt, split_k, use_slabs);
  if (const char* e = getenv("NV_GEMM_NTW")) {   // measurement aid
    const int v = atoi(e);
    if (v >= 1 && v <= (unquantised(wfmt) ? 2 : 3)) a.ntw = v;
  }
  a.bm = silu ? block_rows(M) : plan_rows(M, K, a.ntw, split_k);
  // Unquantised weights cost nothing to "decode", so fewer rows per workgroup only repeat L2 reads of the weights:
  // where the launch would leave CUs idle, halve the row tile until the workgroups cover the chip (Qwen3-0.6B bf16
  // gate_up at 128 rows: 96 workgroups of 128 rows -> 384 of 32; decode 62.3 k -> 64.7 k tok/s on the reference's
  // bench).  Quantised weights keep their tile: each row tile repeats the dequantisation (0.5B int4: 8.8 -> 9.1 us).
  if (unquantised(wfmt)) {
    const int64_t col_blocks = (N / 16 + 4 * a.ntw - 1) / (4 * a.ntw);
    while (a.bm > 32 && col_blocks * ((M + a.bm - 1) / a.bm) * split_k < 256) a.bm /= 2;
  }
  if (const char* e = getenv("NV_GEMM_BM")) {   // measurement aid
    const int v = atoi(e);
    if (v == 16 || v == 32 || v == 64 || v == 128) a.bm = v;
  }
  const int64_t mtiles = (M + a.bm - 1) / a.bm;
  NV_CHECK_ARG(mtiles <= 65535 && split_k <= 65535, "nv_gemm: M=%lld too large for one launch", (long long)M);
  a.epi = use_slabs ? 2 : (silu ? 1 : 0);
  a.rope = RopeArgs{};
  a.result_nt = nv_result_policy();
  a.stats = SampleParts{};
  a.stream = (hipStream_t)stream;
  // Where the planner wants 128 rows x 192 columns per workgroup of int4 weights, the 32x32x16 kernel covers the same
  // tile with its conversion work hidden under the MFMAs (gemm_wide.h); NV_GEMM_WIDE=1 / 0 forces it on (where the
  // shape allows) / off.
  a.wide = 0;
  if (wfmt == NV_W_INT4 && M > 64 && N % 32 == 0) {
    a.wide = a.ntw == 3 && a.bm == 128;
    if (const char* e = getenv("NV_GEMM_WIDE")) a.wide = atoi(e) == 1;   // measurement aid
  }
  // Where three tiles per wavefront were chosen and the launch is a single, partly filled round of workgroups,
  // 160-column workgroups (gemm_mixed_kernel) fill more CUs with shorter workgroups: Qwen2-7B gate_up at 128 rows
  // 48.0 us in 237 workgroups against 53.1 / 53.7 us in 198 (32x32x16 / 16x16x32 kernel).  Over several rounds
  // (512 rows: 948 against 792 workgroups) the extra x tile traffic of the additional workgroups loses: 182 against
  // 174 us.  NV_GEMM_MIXED=1 / 0 forces it.
  a.mixed = 0;
  if (!unquantised(wfmt) && M > 64 && a.ntw == 3 && a.bm == 128) {
    const int64_t mt = (M + 127) / 128, ntl = N / 16;
    const int64_t blocks_mixed = (ntl + GEMM_MIXED_TILES - 1) / GEMM_MIXED_TILES * mt * split_k;
    const int64_t blocks_three = (ntl + 11) / 12 * mt * split_k;
    a.mixed = blocks_mixed <= 256 && blocks_mixed > blocks_three;
  }
  if (const char* e = getenv("NV_GEMM_MIXED")) {   // measurement aid
    a.mixed = atoi(e) == 1 && !unquantised(wfmt) && M > 64;
    if (a.mixed) a.bm = 128;
  }
  if (a.mixed) a.wide = 0;
  a.stream_once = plan_stream_once(a);
  if (a.wide) {
    if (dtype == NV_BF16) nv_gemm_wide_launch_bf16(a); else nv_gemm_wide_launch_f16(a);
  } else {
    if (dtype == NV_BF16) nv_gemm_launch_bf16(a); else nv_gemm_launch_f16(a);
  }
  NV_CHECK_LAUNCH();
  if (use_slabs && !slabs_out) {
    if (dtype == NV_BF16) nv_gemm_reduce_bf16(a); else nv_gemm_reduce_f16(a);
    NV_CHECK_LAUNCH();
  }
  return NV_OK;
}


// ---- LM head + sampler: logits = x . W^T (stored, the caller may read them), one token per row.  The GEMM's epilogue
// leaves the sampler's per-part statistics in the workspace, so the sampler's pass over the stored logits (a read of
// rows x vocabulary x 2 bytes and a launch) is not needed; the draw is the one nv_sample makes from the same logits,
// bit for bit (same statistics, same stage 2).
int nv_sample_pick_launch(const void* logits, int64_t logits_row_stride, const float* temperatures, int64_t* tokens,
                          void* workspace, int B, int V, uint64_t seed, const uint64_t* seed_offset, int dtype,
                          hipStream_t s);

extern "C" size_t nv_lm_head_sample_workspace_bytes(int64_t M, int V) { return sample_parts_bytes(M, V); }

extern "C" int nv_lm_head_sample(const void* x, int64_t x_row_stride, const void* w, int wfmt, void* logits,
                                 int64_t logits_row_stride, const float* temperatures, int64_t* tokens,
                                 void* workspace, size_t workspace_bytes, int64_t M, int V, int K, uint64_t seed,
                                 const uint64_t* seed_offset, int dtype, nv_stream_t stream) {
  NV_CHECK_ARG(x && w && logits && temperatures && tokens, "nv_lm_head_sample: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_lm_head_sample: bad dtype %d", dtype);
  NV_CHECK_ARG(unquantised(wfmt), "nv_lm_head_sample: weight format %d (the vocabulary matrix is never quantised)", wfmt);
  NV_CHECK_ARG(V > 0 && V % 16 == 0 && V <= NV_SAMPLE_MAX_VOCAB, "nv_lm_head_sample: V=%d must be a multiple of 16, at most %d", V, NV_SAMPLE_MAX_VOCAB);
  NV_CHECK_ARG(K > 0 && K % GEMM_BK == 0, "nv_lm_head_sample: K=%d must be a multiple of %d", K, GEMM_BK);
  NV_CHECK_ARG(x_row_stride % 8 == 0 && logits_row_stride % 4 == 0 && logits_row_stride >= V, "nv_lm_head_sample: bad row strides");
  NV_CHECK_ARG(((uintptr_t)x | (uintptr_t)w) % 16 == 0 && (uintptr_t)logits % 8 == 0, "nv_lm_head_sample: misaligned pointer");
  NV_CHECK_ARG(M >= 0 && M <= 65535, "nv_lm_head_sample: M=%lld", (long long)M);
  if (M == 0) return NV_OK;
  const size_t need = sample_parts_bytes(M, V);
  NV_CHECK_ARG(workspace && workspace_bytes >= need && (uintptr_t)workspace % 4 == 0,
               "nv_lm_head_sample: workspace %zu bytes < %zu needed", workspace_bytes, need);
  GemmArgs a;
  a.x = x; a.xs = x_row_stride; a.w = w; a.scales = nullptr; a.bias = nullptr; a.out = logits; a.os = logits_row_stride;
  a.partial = nullptr; a.M = M; a.N = V; a.K = K; a.split_k = 1; a.wfmt = wfmt;
  a.ntw = pick_ntw(M, V, K, wfmt, 1, false);
  if (const char* e = getenv("NV_GEMM_NTW")) {   // measurement aid
    const int v = atoi(e);
    if (v >= 1 && v <= 2) a.ntw = v;
  }
  a.bm = plan_rows(M, K, a.ntw, 1);
  a.epi = 4;
  a.rope = RopeArgs{};
  a.result_nt = nv_result_policy();
  a.wide = 0;
  a.mixed = 0;
  a.stats.P = (V + SAMPLE_PART_COLS - 1) / SAMPLE_PART_COLS;
  a.stats.m = (float*)workspace;
  a.stats.s = a.stats.m + M * a.stats.P;
  a.stats.temps = temperatures;
  a.stream = (hipStream_t)stream;
  a.stream_once = plan_stream_once(a);
  if (dtype == NV_BF16) nv_gemm_launch_bf16(a); else nv_gemm_launch_f16(a);
  NV_CHECK_LAUNCH();
  return nv_sample_pick_launch(logits, logits_row_stride, temperatures, tokens, workspace, (int)M, V, seed, seed_offset,
                               dtype, (hipStream_t)stream);
}

// Rows per workgroup for the one-launch q/k/v projection (a workgroup = one head over all of K, so the grid is
// heads x row tiles) and its estimated duration, with the same per-K-block figures as plan_cost.
static int rope_fused_plan(int64_t M, int K, int wfmt, int heads, int D, double* cost_us) {
  static const double t_quant[3] = {0, 0.70, 1.15}, t_dense[3] = {0, 0.60, 1.0};
  const int ntw = D / 64, nkb = K / GEMM_BK;
  int best = 128;
  double best_us = 1e30;
  for (int bm : {128, 64, 32, 16}) {
    if (bm < 128 && bm >= 2 * M) continue;   // a smaller tile would only add empty rows
    const double t_kb = (unquantised(wfmt) ? t_dense[ntw] : t_quant[ntw]) * (0.4 + 0.6 * bm / 128.0);
    const int64_t blocks = (int64_t)heads * ((M + bm - 1) / bm);
    // fewer than 32 workgroups cannot pull the weights fast enough (measured: 18 workgroups at batch 1 lose to split-K)
    const double us = blocks < 32 ? 1e29 : (double)((blocks + 255) / 256) * (nkb * t_kb + 4.0);
    if (us < best_us) { best_us = us; best = bm; }
  }
  if (cost_us) *cost_us = best_us;
  return best;
}

// 1 when the one-launch projection is estimated faster than nv_gemm (best split-K) followed by the RoPE kernel:
// true for short K (the 0.5B model), false where each row tile would repeat a long dequantisation (7B).
extern "C" int nv_gemm_rope_kv_store_is_faster(int64_t M, int K, int wfmt, int H, int Hkv, int D) {
  if (M <= 0 || K < GEMM_BK || (D != 64 && D != 128)) return 0;
  const int N = (H + 2 * Hkv) * D, max_ntw = unquantised(wfmt) ? 2 : 3;
  double fused;
  rope_fused_plan(M, K, wfmt, H + 2 * Hkv, D, &fused);
  double split = 1e30;
  for (int s = 1; s <= K / GEMM_BK && s <= 32; ++s)
    for (int ntw = 1; ntw <= max_ntw; ++ntw) {
      const double us = plan_cost(M, N, K, wfmt, ntw, s, s > 1);
      if (us < split) split = us;
    }
  // Over many rounds of workgroups the RoPE epilogue (a head's rows staged through LDS, then walked row by row)
  // costs more than a plain store plus the memory-bound RoPE kernel.  Measured, one launch against two: Qwen3-0.6B
  // bf16 at 14.6 k rows 311 us against ~225 (prefill 562 k -> 600 k tok/s), Qwen2-0.5B int4 at 32 k rows 2 % slower,
  // equal at 8 k rows, 10 % FASTER at 2 k rows; Qwen2-7B int4 at 512 rows 0.4 % faster.
  if (M > 4096) return 0;
  return fused < split + 4.5 ? 1 : 0;   // + the stand-alone RoPE / cache-store launch
}

// q/k/v projection with everything that follows it before attention in one launch: x . Wqkv^T + bias, per-head
// RMSNorm (Qwen3), RoPE on q and k, q/k/v written to the qkv buffer and k/v to the token's KV-cache row.
// One workgroup owns one head (64 or 128 columns) over the whole K extent, so nothing is left to reduce.
extern "C" int nv_gemm_rope_kv_store(const void* x, int64_t x_row_stride, const void* w, const void* scales,
                                     const void* bias, void* qkv, int64_t qkv_row_stride, int64_t M, int K, int wfmt,
                                     const int64_t* positions, const float* cos_sin, const int32_t* slot_mapping,
                                     void* k_cache, void* v_cache, const void* q_norm_w, const void* k_norm_w,
                                     float eps, int H, int Hkv, int D, int dtype, nv_stream_t stream) {
  NV_CHECK_ARG(x && w && qkv && positions && cos_sin, "nv_gemm_rope_kv_store: null pointer");
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "nv_gemm_rope_kv_store: bad dtype %d", dtype);
  NV_CHECK_ARG(wfmt >= NV_W_DENSE && wfmt <= NV_W_DENSE_TILED, "nv_gemm_rope_kv_store: bad weight format %d", wfmt);
  NV_CHECK_ARG(D == 64 || D == 128, "nv_gemm_rope_kv_store: head_dim %d not in {64, 128}", D);
  NV_CHECK_ARG(H >= 0 && Hkv >= 0 && H + Hkv > 0, "nv_gemm_rope_kv_store: bad head counts");
  NV_CHECK_ARG(K > 0 && K % GEMM_BK == 0, "nv_gemm_rope_kv_store: K=%d must be a multiple of %d", K, GEMM_BK);
  NV_CHECK_ARG(unquantised(wfmt) || scales, "nv_gemm_rope_kv_store: quantised weights need scales");
  NV_CHECK_ARG((k_cache == nullptr) == (v_cache == nullptr) && (!k_cache || slot_mapping),
               "nv_gemm_rope_kv_store: k_cache, v_cache and slot_mapping go together");
  NV_CHECK_ARG((q_norm_w == nullptr) == (k_norm_w == nullptr), "nv_gemm_rope_kv_store: q_norm_w and k_norm_w go together");
  const int N = (H + 2 * Hkv) * D;
  NV_CHECK_ARG(x_row_stride % 8 == 0 && qkv_row_stride % 4 == 0 && qkv_row_stride >= N, "nv_gemm_rope_kv_store: bad row strides");
  NV_CHECK_ARG(((uintptr_t)x | (uintptr_t)w | (uintptr_t)cos_sin) % 16 == 0 &&
               ((uintptr_t)qkv | (uintptr_t)bias | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)q_norm_w |
                (uintptr_t)k_norm_w) % 8 == 0, "nv_gemm_rope_kv_store: misaligned pointer");
  NV_CHECK_ARG(M >= 0, "nv_gemm_rope_kv_store: M=%lld", (long long)M);
  if (M == 0) return NV_OK;
  GemmArgs a;
  a.wide = 0;
  a.mixed = 0;
  a.stream_once = 0;
  a.stats = SampleParts{};
  a.x = x; a.xs = x_row_stride; a.w = w; a.scales = scales; a.bias = bias; a.out = qkv; a.os = qkv_row_stride;
  a.partial = nullptr; a.M = M; a.N = N; a.K = K; a.split_k = 1; a.wfmt = wfmt;
  a.bm = rope_fused_plan(M, K, wfmt, H + 2 * Hkv, D, nullptr);
  if (const char* e = getenv("NV_ROPE_GEMM_BM")) {   // measurement aid
    const int v = atoi(e);
    if (v == 16 || v == 32 || v == 64 || v == 128) a.bm = v;
  }
  NV_CHECK_ARG((M + a.bm - 1) / a.bm <= 65535, "nv_gemm_rope_kv_store: M=%lld too large for one launch", (long long)M);
  a.ntw = D / 64;
  a.epi = 3;
  a.rope.positions = positions; a.rope.cos_sin = cos_sin; a.rope.slots = slot_mapping;
  a.rope.k_cache = (uint16_t*)k_cache; a.rope.v_cache = (uint16_t*)v_cache;
  a.rope.qn_w = (const uint16_t*)q_norm_w; a.rope.kn_w = (const uint16_t*)k_norm_w;
  a.rope.eps = eps; a.rope.H = H; a.rope.Hkv = Hkv; a.rope.D = D; a.result_nt = nv_result_policy();
  a.stream = (hipStream_t)stream;
  if (dtype == NV_BF16) nv_gemm_launch_bf16(a); else nv_gemm_launch_f16(a);
  NV_CHECK_LAUNCH();
  return NV_OK;
}
