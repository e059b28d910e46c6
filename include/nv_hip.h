/*
 * nv_hip.h — C ABI of the MI355X (gfx950) kernels behind nanovllm's paged-KV
 * decode hot path.
 *
 * The reference engine has no FFI of its own: its hot path is Python calling
 * third-party kernels (flash-attn, Triton, inductor, cuBLAS).  Each entry point
 * below replaces one such call site; the comment above it names the reference
 * file:line.  The Python layers in byox_vllm_amd/layers bind these with ctypes.
 *
 * Conventions
 *   - plain pointers to DEVICE memory and integer sizes, no framework types;
 *   - `dtype` is NV_BF16 or NV_F16 (the model dtype of activations/KV cache);
 *   - strides are in ELEMENTS of the tensor's dtype unless the name says bytes;
 *   - `stream` is a hipStream_t passed as void*;
 *   - return 0 on success, a negative NV_ERR_* otherwise (text via nv_last_error);
 *     nothing throws across the boundary;
 *   - no entry point allocates, synchronises or reads device memory on the host,
 *     so every call can be captured into a hipGraph; scratch space is passed in
 *     by the caller (`workspace`), sized by the matching *_workspace_bytes().
 */
#ifndef NV_HIP_H
#define NV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NV_OK 0
#define NV_ERR_INVALID (-1)     /* bad argument (shape, alignment, dtype) */
#define NV_ERR_LAUNCH (-2)      /* the HIP runtime refused the launch */
#define NV_ERR_UNSUPPORTED (-3) /* valid request outside what the kernels cover */

#define NV_BF16 0
#define NV_F16 1

/* weight formats of nv_gemm */
#define NV_W_DENSE 0 /* [N, K] row-major in the model dtype (reference layout, linear.py:24) */
#define NV_W_INT8 1  /* tiled int8, group-128 scales, see byox_vllm_amd/layers/quant.py */
#define NV_W_INT4 2  /* tiled int4, group-128 scales, see byox_vllm_amd/layers/quant.py */
#define NV_W_DENSE_TILED 3 /* model dtype, [N/16][K/128][4 k-steps][64 lanes][8]: every wavefront load is 1 KB contiguous */

typedef void* nv_stream_t;

int nv_version(void);
const char* nv_last_error(void);

/* RMSNorm.rms_forward — nanovllm/layers/layernorm.py:16-26.
 * out[r] = to_dtype(x[r] * rsqrt(mean(x[r]^2) + eps)) * weight, fp32 inside.
 * rows of `hidden` contiguous elements; hidden % 8 == 0. */
int nv_rmsnorm(const void* x, const void* weight, void* out, int64_t rows, int hidden,
               int64_t x_row_stride, int64_t out_row_stride, float eps, int dtype, nv_stream_t stream);

/* RMSNorm.add_rms_forward — nanovllm/layers/layernorm.py:28-40.
 * s = fp32(x) + fp32(residual); residual_out = to_dtype(s); out = norm of the
 * UNROUNDED s, scaled by weight.  out may alias x, residual_out may alias residual. */
int nv_add_rmsnorm(const void* x, const void* residual, const void* weight, void* out, void* residual_out,
                   int64_t rows, int hidden, float eps, int dtype, nv_stream_t stream);

/* nv_add_rmsnorm whose x operand is still the split-K output of nv_gemm
 * (NV_GEMM_PARTIAL_OUT): x = to_dtype(sum of the num_partials fp32 slabs [rows, hidden]),
 * then exactly nv_add_rmsnorm.  Fuses RowParallelLinear's output (linear.py:143-147,
 * single rank) into the next add_rms_forward. */
int nv_add_rmsnorm_splitk(const float* partials, int num_partials, const void* residual, const void* weight,
                          void* out, void* residual_out, int64_t rows, int hidden, float eps, int dtype,
                          nv_stream_t stream);

/* RotaryEmbedding.forward + store_kvcache (+ Qwen3 q_norm/k_norm) fused —
 * nanovllm/layers/rotary_embedding.py:37-48, layers/attention.py:10-49,
 * models/qwen3.py:76-83.
 * qkv: [T, (H + 2*Hkv) * D] rows with stride qkv_row_stride; q and k are rotated
 * IN PLACE (NeoX halves, fp32, table cos_sin = fp32 [max_pos, D] = cat(cos, sin)).
 * If q_norm_w/k_norm_w are non-NULL each head is RMS-normalised first.
 * If k_cache/v_cache are non-NULL, token t's rotated k and its v are copied to
 * cache row slot_mapping[t] (caches are [num_slots, Hkv*D]); slots < 0 are skipped. */
int nv_rope_kv_store(void* qkv, int64_t qkv_row_stride, const int64_t* positions, const float* cos_sin,
                     const int32_t* slot_mapping, void* k_cache, void* v_cache,
                     const void* q_norm_w, const void* k_norm_w, float eps,
                     int64_t T, int H, int Hkv, int D, int dtype, nv_stream_t stream);

/* nv_rope_kv_store whose qkv operand is still the split-K output of the QKV GEMM:
 * qkv = to_dtype(sum of slabs [T, (H+2Hkv)*D] + bias), written to `qkv` (q, k rotated;
 * v as is), then as nv_rope_kv_store.  bias may be NULL. */
int nv_rope_kv_store_splitk(const float* partials, int num_partials, const void* bias, void* qkv,
                            int64_t qkv_row_stride, const int64_t* positions, const float* cos_sin,
                            const int32_t* slot_mapping, void* k_cache, void* v_cache,
                            const void* q_norm_w, const void* k_norm_w, float eps,
                            int64_t T, int H, int Hkv, int D, int dtype, nv_stream_t stream);

/* store_kvcache — nanovllm/layers/attention.py:10-49 (stand-alone form).
 * Copies key[t] / value[t] (row_elems = Hkv*D contiguous elements, row strides
 * given) to cache row slot_mapping[t]; slots < 0 are skipped. */
int nv_kv_store(const void* key, int64_t key_row_stride, const void* value, int64_t value_row_stride,
                void* k_cache, void* v_cache, const int32_t* slot_mapping,
                int64_t T, int row_elems, int dtype, nv_stream_t stream);

/* flash_attn_with_kvcache as called at nanovllm/layers/attention.py:88-96:
 * one query token per sequence against a paged cache.
 * q [B, H, D] (row stride q_row_stride), k_cache/v_cache [num_blocks, block_size, Hkv, D],
 * block_tables int32 [B, bt_stride] (pad -1), context_lens int32 [B] (0 => zero output),
 * out [B, H, D] (row stride out_row_stride).  D in {64, 128}; H/Hkv <= 16.
 * num_splits >= 1 partitions each sequence's keys over that many workgroups
 * (flash-decoding); 0 lets the library choose.  workspace must hold
 * nv_paged_attn_decode_workspace_bytes(B, H, D, num_splits) bytes. */
size_t nv_paged_attn_decode_workspace_bytes(int B, int H, int D, int num_splits);
int nv_paged_attn_decode_pick_splits(int B, int Hkv, int max_context);
/* Cache policy of results that the NEXT launch of a chain reads (no counterpart in the reference, which leaves cache
 * policy to PyTorch's kernels).  1: the normalised and residual rows of nv_add_rmsnorm(_splitk) / nv_rmsnorm at model
 * width, the SiLU(gate)*up rows and split-K slabs of nv_gemm, and the q/k/v rows with their KV-cache rows of
 * nv_rope_kv_store(_splitk) / nv_gemm_rope_kv_store are written through to the memory side (device-scope stores) by
 * every later launch of this process (a captured launch keeps the policy it was captured under); 0, the default,
 * stores normally.  On the 8-XCD MI355X a consumer kernel fetches its producer's results from the memory side
 * anyway; writing them through while the producer still runs shortens that hand-over (decode steps 0.7-2.4 % faster),
 * while the large results of a prefill are better left to the L2 (1-5 % slower with the policy), so the caller sets it
 * per step kind (byox_vllm_amd/engine/model_runner.py).  The results are the same bits either way. */
int nv_set_result_policy(int write_through);
int nv_get_result_policy(void);
/* Revision name of the decode kernel's memory access pattern (no counterpart in the reference: flash-attn is a
 * binary there).  Hardware-counter collections record it so that a measurement taken on another revision of the
 * kernel is not applied to this one. */
const char* nv_paged_attn_decode_kernel_tag(void);
int nv_paged_attn_decode(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                         const int32_t* block_tables, int bt_stride, const int32_t* context_lens,
                         void* out, int64_t out_row_stride, void* workspace, size_t workspace_bytes,
                         int B, int H, int Hkv, int D, int block_size, float scale, int num_splits,
                         int dtype, nv_stream_t stream);

/* nv_paged_attn_decode for callers that cannot allocate device memory (code under hipGraph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments without the workspace:
 * the split partials go to an array built into the library, NV_PAGED_ATTN_STATIC_SCRATCH_BYTES per GPU, held until
 * the process ends.  Requests that need more (nv_paged_attn_decode_workspace_bytes(B, H, D, num_splits)) are
 * rejected with NV_ERR_INVALID.  Calls no allocating routine.  All calls on one GPU share that GPU's array: calls
 * that overlap on one GPU (two streams, two threads) race; calls on one stream are ordered and safe.  The first
 * call on a device looks the array's address up (hipGetSymbolAddress): make it outside stream capture. */
#define NV_PAGED_ATTN_STATIC_SCRATCH_BYTES (16u << 20)
int nv_paged_attn_decode_static(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                                const int32_t* block_tables, int bt_stride, const int32_t* context_lens,
                                void* out, int64_t out_row_stride, int B, int H, int Hkv, int D, int block_size,
                                float scale, int num_splits, int dtype, nv_stream_t stream);

/* flash_attn_varlen_func as called at nanovllm/layers/attention.py:75-86:
 * causal (bottom-right aligned) variable-length prefill.
 * q [Tq, H, D]; without block_tables k/v are packed [Tk, Hkv, D] (row strides
 * given); with block_tables k/v are the paged caches and sequence i reads its
 * cu_seqlens_k[i+1]-cu_seqlens_k[i] keys through block_tables[i]. */
int nv_attn_prefill(const void* q, int64_t q_row_stride, const void* k, const void* v, int64_t kv_row_stride,
                    const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                    const int32_t* block_tables, int bt_stride, void* out, int64_t out_row_stride,
                    int num_seqs, int max_seqlen_q, int H, int Hkv, int D, int block_size, float scale,
                    int dtype, nv_stream_t stream);

/* Many-row products (prefill) through the vendor library, for the same F.linear (nanovllm/layers/linear.py:49,70,144):
 * out[M, N] = x[M, K] * W^T (+ bias), fp32 accumulation, one rounding.  W is ROW-MAJOR [N, K] in the model dtype — a
 * weight kept in one of nv_gemm's formats is written out that way by nv_weight_expand first.  hipBLASLt is loaded on
 * first use (dlopen); nv_gemm_lt_available() says whether that worked, and nv_gemm_lt fails with NV_ERR_INVALID if
 * not.  workspace: device memory the library may use (workspace_bytes of it; the library asks for up to 64 MiB), nothing is allocated
 * here.  Not for captured graphs (the library picks an algorithm per shape on the host at the first call). */
int nv_gemm_lt_available(void);
int nv_gemm_lt(const void* x, int64_t x_row_stride, const void* w_rowmajor, const void* bias, void* out,
               int64_t out_row_stride, int M, int N, int K, int dtype, void* workspace, size_t workspace_bytes,
               nv_stream_t stream);

/* Weight of any nv_gemm format (NV_W_*) -> row-major [N, K] in the model dtype: tiled layouts un-tiled, int4 / int8
 * dequantised as (u - 8) * s resp. q * s in fp32, rounded once (the oracle's dequantisation, so nv_gemm_lt on the result
 * is "dequantise, then F.linear").  gate_up_rows != 0: the rows are stored in the order of nv_gemm's silu(gate)*up
 * epilogue (gate 2i, gate 2i+1, up 2i, up 2i+1) and are written as [all gate rows; all up rows], the order
 * nv_silu_mul reads.  No counterpart in the reference (its weights are row-major already). */
int nv_weight_expand(const void* data, const void* scales, int fmt, int N, int K, int gate_up_rows, int dtype, void* out,
                     nv_stream_t stream);

/* SiluAndMul.forward — nanovllm/layers/activation.py:10-13.
 * out[t, i] = to_dtype(silu(x[t, i])) * x[t, I + i], x [T, 2I]. */
int nv_silu_mul(const void* x, void* out, int64_t T, int I, int dtype, nv_stream_t stream);

/* F.linear at nanovllm/layers/linear.py:49,70,144 and embed_head.py:59:
 * out[M, N] = x[M, K] . W[N, K]^T (+ bias[N]), fp32 accumulation, output in dtype.
 * wfmt selects dense or weight-only int8/int4 (dequantised in registers; the
 * reference has no quantised path, its dense F.linear on the dequantised weight
 * is the oracle).  scales: [K/128, N] in the model dtype for int8/int4, else NULL.
 * K % 128 == 0, N % 64 == 0 for int8/int4; K % 32 == 0, N % 16 == 0 for dense.
 * split_k > 1 spreads K over that many workgroups, each writing an fp32 slab
 * [M, N] into the workspace (nv_gemm_workspace_bytes(M, N, split_k) bytes), summed
 * in slab order by a second kernel.
 * flags:
 *   NV_GEMM_SILU_MUL    W rows are interleaved so that rows 4i..4i+3 are (gate 2i,
 *                       gate 2i+1, up 2i, up 2i+1); the epilogue writes
 *                       silu(gate)*up (SiluAndMul, activation.py:10-13), out is [M, N/2].
 *   NV_GEMM_PARTIAL_OUT leave the split_k fp32 slabs in the workspace (no bias, no
 *                       reduce, `out` unused): the consumer kernel sums them
 *                       (nv_add_rmsnorm_splitk, nv_rope_kv_store_splitk). */
#define NV_GEMM_SILU_MUL 1
#define NV_GEMM_PARTIAL_OUT 2
size_t nv_gemm_workspace_bytes(int64_t M, int N, int split_k);

/* Split-K factor nv_gemm runs fastest with for this shape on MI355X (>= 1): the planner weighs rounds of
 * workgroups over the 256 CUs against the traffic of the fp32 slabs.  A caller-chosen split_k is honoured too;
 * this is the value the host layer uses when the caller has no preference.  No counterpart in the reference
 * (cuBLAS chooses its own algorithm behind F.linear, nanovllm/layers/linear.py:49). */
int nv_gemm_pick_split_k(int64_t M, int N, int K, int wfmt);

/* The q/k/v projection and everything between it and attention in ONE launch:
 *   qkv = to_dtype(x . Wqkv^T + bias)                (QKVParallelLinear, linear.py:92-123, via F.linear :70)
 *   q, k: optional per-head RMSNorm (qwen3.py:80-82), RoPE (rotary_embedding.py:37-48)
 *   k, v of token t -> cache row slot_mapping[t]     (store_kvcache, attention.py:34-49)
 * W holds (H + 2 Hkv) * D rows in any nv_gemm weight format; one workgroup owns one head over the whole K extent,
 * so there is no split-K workspace.  Arguments as in nv_gemm and nv_rope_kv_store; D is 64 or 128. */
int nv_gemm_rope_kv_store_is_faster(int64_t M, int K, int wfmt, int H, int Hkv, int D);  /* planner's estimate: 1 = prefer the
                                                           one-launch form over nv_gemm + nv_rope_kv_store(_splitk) */
int nv_gemm_rope_kv_store(const void* x, int64_t x_row_stride, const void* w, const void* scales, const void* bias,
                          void* qkv, int64_t qkv_row_stride, int64_t M, int K, int wfmt,
                          const int64_t* positions, const float* cos_sin, const int32_t* slot_mapping,
                          void* k_cache, void* v_cache, const void* q_norm_w, const void* k_norm_w, float eps,
                          int H, int Hkv, int D, int dtype, nv_stream_t stream);
int nv_gemm(const void* x, int64_t x_row_stride, const void* w, const void* scales, const void* bias,
            void* out, int64_t out_row_stride, int64_t M, int N, int K, int wfmt, int split_k,
            void* workspace, size_t workspace_bytes, int flags, int dtype, nv_stream_t stream);

/* VocabParallelEmbedding.forward (gather + mask) — nanovllm/layers/embed_head.py:33-41.
 * out[t] = weight[ids[t] - vocab_start] if vocab_start <= ids[t] < vocab_start + vocab_local else 0. */
int nv_embed(const int64_t* ids, const void* weight, void* out, int64_t T, int hidden,
             int64_t vocab_start, int64_t vocab_local, int dtype, nv_stream_t stream);

/* Sampler.forward — nanovllm/layers/sampler.py:11-15.
 * token[b] = argmax_v softmax(logits[b]/T[b])_v / max(noise[b, v], 1e-10); noise is
 * Exp(1): read from `noise` (fp32 [B, V]) when given, else drawn in-kernel from
 * (seed, b, v).  Evaluated as argmax_v (logits/T - log noise), which has the same
 * maximiser.  T[b] == 0 takes argmax(logits[b]) (greedy extension).  Ties resolve
 * to the lowest index.  logits are in `dtype` (NV_BF16/NV_F16) rows of V.
 * Without `noise` the draw uses the inverse CDF of softmax(logits/T) with one uniform per row from (seed, b):
 * the same distribution as the noisy argmax at one exp per logit instead of a hash and two logs. */
size_t nv_sample_workspace_bytes(int B, int V);
int nv_sample(const void* logits, int64_t logits_row_stride, const float* temperatures, const float* noise,
              int64_t* tokens, void* workspace, size_t workspace_bytes, int B, int V, uint64_t seed,
              const uint64_t* seed_offset /* device word mixed into the seed, or NULL: lets a captured
                                             launch draw fresh noise on every replay */,
              int dtype, nv_stream_t stream);

/* ParallelLMHead.forward + Sampler.forward in two launches — nanovllm/layers/embed_head.py:54-66 (decode branch,
 * one rank) followed by sampler.py:11-15, as the decode step chains them (engine/model_runner.py:253-254, 62).
 * logits[M, V] = x[M, K] . w[V, K]^T are stored in `dtype` as nv_gemm stores them; the GEMM's epilogue also leaves,
 * per row and 64-column part, the maximum, the lowest column holding it and the softmax mass at the row's
 * temperature, from which tokens[m] is drawn exactly as nv_sample (noise == NULL) draws it from the stored logits:
 * same token bit for bit, without nv_sample's pass over the logits.  w: NV_W_DENSE or NV_W_DENSE_TILED. */
size_t nv_lm_head_sample_workspace_bytes(int64_t M, int V);
int nv_lm_head_sample(const void* x, int64_t x_row_stride, const void* w, int wfmt, void* logits,
                      int64_t logits_row_stride, const float* temperatures, int64_t* tokens, void* workspace,
                      size_t workspace_bytes, int64_t M, int V, int K, uint64_t seed, const uint64_t* seed_offset,
                      int dtype, nv_stream_t stream);

/* Graph-replay staging — nanovllm/engine/model_runner.py:246-252 as ONE launch:
 * copies the step's ids/positions/slots/context_lens/block_tables into the
 * static buffers a captured decode step reads, padding rows [bs, padded_bs)
 * with slot -1 / context_len 0. */
int nv_stage_decode_inputs(const int64_t* ids, const int64_t* positions, const int32_t* slots,
                           const int32_t* context_lens, const int32_t* block_tables, int bt_cols,
                           int64_t* g_ids, int64_t* g_positions, int32_t* g_slots, int32_t* g_context_lens,
                           int32_t* g_block_tables, int g_bt_cols, int bs, int padded_bs, nv_stream_t stream);

/* The same staging fed by ONE packed host buffer of 32-bit words
 * [ids(bs) | positions(bs) | slots(bs) | context_lens(bs) | temperatures as float bits(bs) | block_tables(bs*bt_cols)]
 * (one host->device copy per step instead of the reference's five, model_runner.py:216-224,232);
 * also fills g_temperatures (padding rows 1.0) and adds 1 to *step_counter when given.
 * Header form, bs < 0: the buffer starts with two extra words {bs, bt_cols} read ON THE DEVICE (clamped to
 * padded_bs / g_bt_cols), so the launch has no per-step host argument and can live inside a captured graph
 * behind a fixed-size copy of the buffer. */
int nv_stage_decode_packed(const int32_t* packed, int bs, int padded_bs, int bt_cols,
                           int64_t* g_ids, int64_t* g_positions, int32_t* g_slots, int32_t* g_context_lens,
                           int32_t* g_block_tables, int g_bt_cols, float* g_temperatures,
                           uint64_t* step_counter, nv_stream_t stream);

/* The header form with device-side bookkeeping.  When header word 1 has NV_STAGE_ADVANCE set, the step continues
 * the previous one for the same rows and nothing else is read from the buffer: for every row i < bs
 *   ids[i] = prev_tokens[i] (the token sampled for the row in the previous step), positions[i] += 1,
 *   context_lens[i] += 1, slots[i] = block_tables[i][positions[i] / block_size] * block_size + positions[i] % block_size
 *   (-1 when that table entry is negative or past g_bt_cols)
 * which is one step of the per-sequence arithmetic of ModelRunner.prepare_decode (model_runner.py:206-226) done on
 * the device from the tables the previous full staging left there; block tables and temperatures stay.  Without the
 * flag the call is nv_stage_decode_packed's header form.  The host decides per step which form applies (same rows, each
 * one token longer by exactly the token returned, no block table changed) by setting the flag in the pinned buffer,
 * so one captured launch serves both.
 * NV_STAGE_STEP_BACK in header word 1 (either form): *step_counter is first reduced by 1.  A host that launched a step
 * before its caller asked for it, and then had to throw that step away, sets it on the next launch so that the
 * sampler's draws are those of a run in which the discarded step never happened. */
#define NV_STAGE_ADVANCE 0x40000000
#define NV_STAGE_STEP_BACK 0x20000000
int nv_stage_decode_packed_advance(const int32_t* packed, int padded_bs, int64_t* g_ids, int64_t* g_positions,
                                   int32_t* g_slots, int32_t* g_context_lens, int32_t* g_block_tables, int g_bt_cols,
                                   float* g_temperatures, uint64_t* step_counter, const int64_t* prev_tokens,
                                   int block_size, nv_stream_t stream);

/* ---- all-reduce over peer-mapped device memory (xGMI), fused with the residual add and RMSNorm that follow it.
 * Replaces dist.all_reduce after RowParallelLinear (nanovllm/layers/linear.py:145-146) and after the vocab-parallel
 * embedding (layers/embed_head.py:38-40) together with the add_rms_forward that consumes the sum
 * (layers/layernorm.py:28-40), and NCCL's process-group set-up for them (engine/model_runner.py:27).
 *
 * nv_comm_init allocates this rank's buffer (nv_comm_buffer_bytes; uncached device memory) and is the only call
 * that allocates.  The caller then exchanges the 64-byte handles of nv_comm_export between the ranks (any host
 * channel: torch.distributed, a pipe) and maps every peer with nv_comm_open_peer (another process: hipIpc) or
 * nv_comm_set_peer (same process).  Every rank must issue the same sequence of nv_allreduce* calls (SPMD, as the
 * reference's ranks do); the calls keep their call counter in device memory, take no per-call host argument and can
 * be captured into a hipGraph.  A rank owns rows [r*R, (r+1)*R), R = ceil(rows / world):
 *   nv_allreduce             out[rows, hidden] = to_dtype(sum over ranks of x), the same bits on every rank (fp32 sum
 *                            in rank order, one rounding);
 *   nv_allreduce_add_rmsnorm out = RMSNorm(to_dtype(sum) + residual) * weight for ALL rows; residual_out only for
 *                            the rows this rank owns: the residual stream stays sharded by rows, each rank reads and
 *                            writes only its own rows of residual / residual_out (they may be the same buffer).
 * x may be NULL when `partials` (num_partials fp32 slabs [rows, hidden] of a split-K nv_gemm) is given: the rank's
 * contribution is then to_dtype(sum of the slabs).  nv_comm_error returns non-zero after a wait for a peer ran out
 * (about two seconds); it reads device memory and synchronises. */
typedef struct nv_comm nv_comm_t;
size_t nv_comm_buffer_bytes(int world, int max_rows, int hidden);
int nv_comm_init(nv_comm_t** comm, int rank, int world, int max_rows, int hidden);
int nv_comm_memory_kind(const nv_comm_t* comm);   /* 2 uncached, 1 fine-grained, 0 plain device memory */
void* nv_comm_local_base(const nv_comm_t* comm);
int nv_comm_export(nv_comm_t* comm, void* handle64);
int nv_comm_open_peer(nv_comm_t* comm, int peer, const void* handle64);
int nv_comm_set_peer(nv_comm_t* comm, int peer, void* base);
int nv_comm_error(nv_comm_t* comm);
int nv_comm_destroy(nv_comm_t* comm);
int nv_allreduce(nv_comm_t* comm, const void* x, const float* partials, int num_partials, void* out, int64_t rows,
                 int hidden, int dtype, nv_stream_t stream);
int nv_allreduce_add_rmsnorm(nv_comm_t* comm, const void* x, const float* partials, int num_partials,
                             const void* residual, const void* weight, void* out, void* residual_out, int64_t rows,
                             int hidden, float eps, int dtype, nv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NV_HIP_H */
