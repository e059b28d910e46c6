// Two-shot all-reduce over peer-mapped device memory (xGMI between the GPUs of one node), fused with the
// residual add and RMSNorm that follow every row-parallel linear of the decoder.
//
// What it replaces: dist.all_reduce(y) after RowParallelLinear (nanovllm/layers/linear.py:145-146) plus the
// add_rms_forward that consumes it (layers/layernorm.py:28-40): 2 per layer on the decode critical path.
//
// Each rank owns a contiguous slice of the ROWS (tokens) of the [rows, hidden] message:
//   push    every rank stores the rows owned by rank p (of its local partial result) into p's staging slot
//           for this sender; one flag per (sender, receiver) says "my rows for you are in place";
//   reduce  the owner sums the world contributions of its rows in fp32, in rank order (so the result does not
//           depend on timing), rounds once to the model dtype, adds the residual, normalises: all of
//           add_rms_forward on whole rows, 1/world of the norm work per rank.  The residual stream therefore
//           stays SHARDED by rows: a rank only ever reads and writes the residual rows it owns;
//   gather  the owner stores the normalised rows into every peer's gather buffer; a second flag; each rank copies
//           the peers' rows from its gather buffer into the ordinary (cacheable) output tensor the next GEMM reads.
// Traffic per rank and call: 2 * (world-1)/world * rows * hidden * 2 bytes, spread over all links at once
// (every rank talks to every peer directly: xGMI is point to point, a ring would serialise the hops).
//
// Memory: one buffer per rank, allocated uncached (so neither side's L2 can hold a stale line of it) and mapped
// into every peer: a header (call counter, workgroup arrival counters, error word), flag lines written by the
// peers, two generations (call parity) of staging slots and of the gather buffer.  Two generations suffice: a rank
// can only start pushing for call k+1 after every peer has raised its push flag for call k... which a peer does
// after finishing call k-1 entirely (stream order), i.e. after its last read of the generation k+1 reuses.
// All waits are bounded (about two seconds of the 100 MHz clock); a wait that runs out sets the header's error
// word and the kernel carries on, so a lost peer cannot hang the GPU.
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "splitk.h"

#define NV_COMM_MAX_WORLD 8
#define NV_COMM_HEADER_BYTES 8192
#define NV_COMM_WAIT_TICKS 200000000ull   // s_memrealtime runs at 100 MHz

struct CommHeader {
  uint32_t seq;         // calls this rank has completed
  uint32_t arrive[2];   // workgroup arrival counters of the push / gather phases (the last arriver resets them)
  uint32_t error;       // non-zero once a bounded wait ran out
  uint32_t pad[28];
  uint32_t flag[2][NV_COMM_MAX_WORLD][32];   // [phase][sender]: one 128-byte line each, written by the sender
};
static_assert(sizeof(CommHeader) <= NV_COMM_HEADER_BYTES, "header region too small");

struct CommArgs {
  char* peer[NV_COMM_MAX_WORLD];   // base of every rank's buffer as mapped into this process (peer[rank] = own)
  int rank, world;
  int64_t stage_off, stage_slot_bytes, gather_off, gather_bytes;   // layout inside a buffer
};

struct nv_comm {
  CommArgs a;
  int max_rows, hidden, device, memory_kind;   // memory_kind: 2 uncached, 1 fine-grained, 0 plain hipMalloc
  size_t bytes;
  bool opened[NV_COMM_MAX_WORLD];
};

__device__ __forceinline__ void store_sys(void* p, u32x4 v) {   // write-through to the system coherence point
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// System-scope loads are issued in batches and waited for once: the compiler cannot see that an asm load is still
// in flight, so every value passes through load_sys_wait() (which ties it to the s_waitcnt) before it is used.
__device__ __forceinline__ void load_sys_issue(u32x4& v, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void load_sys_wait(u32x4 (&v)[N]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// Workgroup arrival: true in the workgroup that arrives last (it then signals the peers).
__device__ __forceinline__ bool arrive_last(uint32_t* counter, int* s_last) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    const uint32_t old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = old == gridDim.x - 1;
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = last;
  }
  __syncthreads();
  return *s_last != 0;
}

__device__ __forceinline__ void signal_peers(const CommArgs& a, int phase, uint32_t value) {
  for (int p = 0; p < a.world; ++p) {
    if (p == a.rank) continue;
    CommHeader* h = reinterpret_cast<CommHeader*>(a.peer[p]);
    __hip_atomic_store(&h->flag[phase][a.rank][0], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Every workgroup waits until all peers have raised flag[phase] to `value`; threads 0 .. world-1 poll one peer each.
__device__ __forceinline__ void wait_peers(const CommArgs& a, int phase, uint32_t value) {
  CommHeader* h = reinterpret_cast<CommHeader*>(a.peer[a.rank]);
  const int p = threadIdx.x;
  if (p < a.world && p != a.rank) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while ((int32_t)(__hip_atomic_load(&h->flag[phase][p][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > NV_COMM_WAIT_TICKS) {
        __hip_atomic_store(&h->error, 1u + (uint32_t)phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
    __threadfence_system();
  }
  __syncthreads();
}

// x (model dtype) or, when part != nullptr, the split-K slabs of the producing GEMM summed in slab order and
// rounded to the model dtype: the value the rank's own F.linear would have produced.
template <typename T>
__device__ __forceinline__ u32x4 local_vec(const uint16_t* x, const float* part, int nparts, int64_t slab, int64_t elem) {
  if (part) {
    f32x4 acc[2];
    // batches of 4: with 8 slabs in flight per call site this kernel ran out of registers and took a stack frame,
    // and a kernel that needs scratch memory is one more thing that can go wrong inside a captured graph
    slab_sum<2, 4>(part + elem, slab, nparts, acc);
    return u32x4{pack2<T>(acc[0][0], acc[0][1]), pack2<T>(acc[0][2], acc[0][3]), pack2<T>(acc[1][0], acc[1][1]),
                 pack2<T>(acc[1][2], acc[1][3])};
  }
  return *reinterpret_cast<const u32x4*>(x + elem);
}

template <typename T, bool NORM>
__global__ __launch_bounds__(256) void allreduce_kernel(CommArgs a, const uint16_t* x,
                                                        const float* __restrict__ part, int nparts,
                                                        const uint16_t* residual, const uint16_t* __restrict__ w,
                                                        uint16_t* out, uint16_t* residual_out, int rows, int hidden,
                                                        float eps) {
  __shared__ int s_last;
  CommHeader* hdr = reinterpret_cast<CommHeader*>(a.peer[a.rank]);
  const uint32_t seq = __hip_atomic_load(&hdr->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int parity = seq & 1;
  const int nvec = hidden >> 3;
  const int R = (rows + a.world - 1) / a.world;            // rows per owner (the last owners may have fewer, or none)
  const int64_t slab = (int64_t)rows * hidden;
  const int64_t row_bytes = (int64_t)hidden * 2;
  const int my_lo = a.rank * R, my_hi = min(rows, my_lo + R);

  // ---- push: rows owned by others, from the local partial result into the owners' staging slots
  if (a.world > 1) {
    const int64_t total = (int64_t)rows * nvec;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
      const int row = (int)(v / nvec), col = (int)(v - (int64_t)row * nvec);
      const int owner = row / R;
      if (owner == a.rank) continue;
      const u32x4 val = local_vec<T>(x, part, nparts, slab, (int64_t)row * hidden + 8 * col);
      char* dst = a.peer[owner] + a.stage_off + ((int64_t)parity * a.world + a.rank) * a.stage_slot_bytes +
                  (int64_t)(row - owner * R) * row_bytes + col * 16;
      store_sys(dst, val);
    }
    if (arrive_last(&hdr->arrive[0], &s_last) && threadIdx.x == 0) signal_peers(a, 0, 2 * seq + 1);
    wait_peers(a, 0, 2 * seq + 1);
  }

  // ---- reduce (+ residual add + RMSNorm) on the rows this rank owns: one wavefront per row
  const int lane = threadIdx.x & 63;
  const char* my_stage = a.peer[a.rank] + a.stage_off + (int64_t)parity * a.world * a.stage_slot_bytes;
  for (int row = my_lo + blockIdx.x * 4 + (threadIdx.x >> 6); row < my_hi; row += gridDim.x * 4) {
    // sum over ranks in rank order, fp32, one rounding: what every rank would get from an exact all-reduce
    auto summed = [&](int i) {
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
      u32x4 v[NV_COMM_MAX_WORLD];
#pragma unroll
      for (int p = 0; p < NV_COMM_MAX_WORLD; ++p) {
        v[p] = u32x4{0, 0, 0, 0};
        if (p < a.world && p != a.rank)
          load_sys_issue(v[p], my_stage + (int64_t)p * a.stage_slot_bytes + (int64_t)(row - my_lo) * row_bytes + i * 16);
      }
      const u32x4 mine = local_vec<T>(x, part, nparts, slab, (int64_t)row * hidden + 8 * i);
      load_sys_wait(v);
#pragma unroll
      for (int p = 0; p < NV_COMM_MAX_WORLD; ++p) {
        if (p >= a.world) break;
        const u32x4 t = (p == a.rank) ? mine : v[p];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += lo_f32<T>(t[j]);
          acc[2 * j + 1] += hi_f32<T>(t[j]);
        }
      }
      return u32x4{pack2<T>(acc[0], acc[1]), pack2<T>(acc[2], acc[3]), pack2<T>(acc[4], acc[5]), pack2<T>(acc[6], acc[7])};
    };
    float inv = 1.f;
    if (NORM) {
      float ss = 0.f;
      for (int i = lane; i < nvec; i += 64) {
        const u32x4 s = summed(i);
        const u32x4 r = *reinterpret_cast<const u32x4*>(residual + (int64_t)row * hidden + 8 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = lo_f32<T>(s[j]) + lo_f32<T>(r[j]), hi = hi_f32<T>(s[j]) + hi_f32<T>(r[j]);
          ss = fmaf(lo, lo, ss);
          ss = fmaf(hi, hi, ss);
        }
      }
      ss = wave_sum(ss);
      inv = 1.0f / sqrtf(ss / (float)hidden + eps);
    }
    for (int i = lane; i < nvec; i += 64) {
      const u32x4 s = summed(i);
      u32x4 o = s;
      if (NORM) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(residual + (int64_t)row * hidden + 8 * i);
        const u32x4 ww = *reinterpret_cast<const u32x4*>(w + 8 * i);
        u32x4 rn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = lo_f32<T>(s[j]) + lo_f32<T>(r[j]), hi = hi_f32<T>(s[j]) + hi_f32<T>(r[j]);
          rn[j] = pack2<T>(lo, hi);
          const float nlo = bits_to_f32<T>(f32_to_bits<T>(lo * inv)), nhi = bits_to_f32<T>(f32_to_bits<T>(hi * inv));
          o[j] = pack2<T>(nlo * lo_f32<T>(ww[j]), nhi * hi_f32<T>(ww[j]));
        }
        *reinterpret_cast<u32x4*>(residual_out + (int64_t)row * hidden + 8 * i) = rn;
      }
      *reinterpret_cast<u32x4*>(out + (int64_t)row * hidden + 8 * i) = o;
      for (int p = 0; p < a.world; ++p) {
        if (p == a.rank) continue;
        store_sys(a.peer[p] + a.gather_off + (int64_t)parity * a.gather_bytes + (int64_t)row * row_bytes + i * 16, o);
      }
    }
  }

  // ---- gather: wait for the peers' rows, copy them from the (uncached) gather buffer into the output tensor
  if (a.world > 1) {
    if (arrive_last(&hdr->arrive[1], &s_last) && threadIdx.x == 0) {
      signal_peers(a, 1, 2 * seq + 2);
      __hip_atomic_store(&hdr->seq, seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wait_peers(a, 1, 2 * seq + 2);
    const char* g = a.peer[a.rank] + a.gather_off + (int64_t)parity * a.gather_bytes;
    const int64_t total = (int64_t)rows * nvec;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v0 < total; v0 += 4 * stride) {
      u32x4 t[4];
      bool take[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t v = v0 + u * stride;
        const int row = v < total ? (int)(v / nvec) : my_lo;
        take[u] = v < total && !(row >= my_lo && row < my_hi);
        t[u] = u32x4{0, 0, 0, 0};
        if (take[u]) load_sys_issue(t[u], g + v * 16);
      }
      load_sys_wait(t);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (take[u]) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out) + (v0 + u * stride) * 16) = t[u];
    }
  } else if (blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_store(&hdr->seq, seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------ C ABI
extern "C" size_t nv_comm_buffer_bytes(int world, int max_rows, int hidden) {
  if (world < 1 || max_rows < 1 || hidden < 8) return 0;
  const size_t R = (size_t)(max_rows + world - 1) / world;
  const size_t slot = (R * hidden * 2 + 255) / 256 * 256, gather = ((size_t)max_rows * hidden * 2 + 255) / 256 * 256;
  return NV_COMM_HEADER_BYTES + 2 * (size_t)world * slot + 2 * gather;
}

extern "C" int nv_comm_init(nv_comm** comm, int rank, int world, int max_rows, int hidden) {
  NV_CHECK_ARG(comm, "nv_comm_init: null pointer");
  NV_CHECK_ARG(world >= 1 && world <= NV_COMM_MAX_WORLD && rank >= 0 && rank < world, "nv_comm_init: rank %d of %d", rank, world);
  NV_CHECK_ARG(max_rows >= 1 && hidden >= 8 && hidden % 8 == 0, "nv_comm_init: max_rows=%d hidden=%d", max_rows, hidden);
  nv_comm* c = new nv_comm();
  c->max_rows = max_rows; c->hidden = hidden;
  c->bytes = nv_comm_buffer_bytes(world, max_rows, hidden);
  c->a.rank = rank; c->a.world = world;
  const size_t R = (size_t)(max_rows + world - 1) / world;
  c->a.stage_slot_bytes = (int64_t)((R * hidden * 2 + 255) / 256 * 256);
  c->a.gather_bytes = (int64_t)(((size_t)max_rows * hidden * 2 + 255) / 256 * 256);
  c->a.stage_off = NV_COMM_HEADER_BYTES;
  c->a.gather_off = c->a.stage_off + 2 * (int64_t)world * c->a.stage_slot_bytes;
  for (int p = 0; p < NV_COMM_MAX_WORLD; ++p) { c->a.peer[p] = nullptr; c->opened[p] = false; }
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; nv_set_error("nv_comm_init: no current device"); return NV_ERR_LAUNCH; }
  void* base = nullptr;
  c->memory_kind = 2;
  if (hipExtMallocWithFlags(&base, c->bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    c->memory_kind = 1;
    if (hipExtMallocWithFlags(&base, c->bytes, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      c->memory_kind = 0;
      if (hipMalloc(&base, c->bytes) != hipSuccess) {
        nv_set_error("nv_comm_init: cannot allocate %zu bytes", c->bytes);
        delete c;
        return NV_ERR_LAUNCH;
      }
    }
  }
  if (hipMemset(base, 0, c->bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(base); delete c;
    nv_set_error("nv_comm_init: cannot clear the buffer");
    return NV_ERR_LAUNCH;
  }
  c->a.peer[rank] = (char*)base;
  *comm = c;
  return NV_OK;
}

extern "C" int nv_comm_memory_kind(const nv_comm* c) { return c ? c->memory_kind : -1; }
extern "C" void* nv_comm_local_base(const nv_comm* c) { return c ? c->a.peer[c->a.rank] : nullptr; }

// 64-byte inter-process handle of this rank's buffer (hipIpcMemHandle_t), to be sent to every peer by the caller.
extern "C" int nv_comm_export(nv_comm* c, void* handle64) {
  NV_CHECK_ARG(c && handle64, "nv_comm_export: null pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  const hipError_t e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), c->a.peer[c->a.rank]);
  if (e != hipSuccess) { nv_set_error("nv_comm_export: hipIpcGetMemHandle: %s", hipGetErrorString(e)); return NV_ERR_LAUNCH; }
  return NV_OK;
}

extern "C" int nv_comm_open_peer(nv_comm* c, int peer, const void* handle64) {
  NV_CHECK_ARG(c && handle64 && peer >= 0 && peer < c->a.world && peer != c->a.rank, "nv_comm_open_peer: bad peer %d", peer);
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  void* base = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { nv_set_error("nv_comm_open_peer: hipIpcOpenMemHandle: %s", hipGetErrorString(e)); return NV_ERR_LAUNCH; }
  c->a.peer[peer] = (char*)base;
  c->opened[peer] = true;
  return NV_OK;
}

// A peer whose buffer is already addressable from this process (several ranks driven by one process).
extern "C" int nv_comm_set_peer(nv_comm* c, int peer, void* base) {
  NV_CHECK_ARG(c && base && peer >= 0 && peer < c->a.world && peer != c->a.rank, "nv_comm_set_peer: bad peer %d", peer);
  c->a.peer[peer] = (char*)base;
  return NV_OK;
}

// Error word of the last calls (0 = every wait was answered).  Reads device memory: synchronises the device.
extern "C" int nv_comm_error(nv_comm* c) {
  if (!c) return -1;
  uint32_t e = 0;
  if (hipMemcpy(&e, c->a.peer[c->a.rank] + offsetof(CommHeader, error), 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)e;
}

extern "C" int nv_comm_destroy(nv_comm* c) {
  if (!c) return NV_OK;
  (void)hipDeviceSynchronize();
  for (int p = 0; p < c->a.world; ++p)
    if (c->opened[p]) (void)hipIpcCloseMemHandle(c->a.peer[p]);
  (void)hipFree(c->a.peer[c->a.rank]);
  delete c;
  return NV_OK;
}

static int allreduce_go(nv_comm* c, const void* x, const float* partials, int num_partials, const void* residual,
                        const void* weight, void* out, void* residual_out, int64_t rows, int hidden, float eps,
                        int dtype, bool norm, nv_stream_t stream, const char* who) {
  NV_CHECK_ARG(c && out && (x || partials), "%s: null pointer", who);
  NV_CHECK_ARG(!partials || num_partials >= 1, "%s: num_partials=%d", who, num_partials);
  NV_CHECK_ARG(!norm || (residual && weight && residual_out), "%s: residual, weight and residual_out are required", who);
  NV_CHECK_ARG(dtype == NV_BF16 || dtype == NV_F16, "%s: bad dtype %d", who, dtype);
  NV_CHECK_ARG(hidden == c->hidden, "%s: hidden=%d, the communicator was built for %d", who, hidden, c->hidden);
  NV_CHECK_ARG(rows >= 0 && rows <= c->max_rows, "%s: rows=%lld exceeds the communicator's %d", who, (long long)rows, c->max_rows);
  NV_CHECK_ARG(((uintptr_t)x | (uintptr_t)partials | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)out |
                (uintptr_t)residual_out) % 16 == 0, "%s: pointers must be 16-byte aligned", who);
  for (int p = 0; p < c->a.world; ++p)
    NV_CHECK_ARG(c->a.peer[p], "%s: the buffer of rank %d has not been connected", who, p);
  if (rows == 0) return NV_OK;
  int64_t blocks = (rows * (hidden / 8) + 1023) / 1024;
  if (blocks < 8) blocks = 8;
  // Every workgroup polls the flags and waits for its peers' workgroups, so ALL workgroups of ALL ranks that share a
  // GPU must be resident at once (a GPU holds 512 of them: 2 per CU at this kernel's register count).  One rank per GPU,
  // the product's configuration, is far inside that with 96.  A rehearsal that puts several ranks on one GPU lowers
  // the cap through NV_COMM_MAX_BLOCKS (8 ranks x 96 would wait for workgroups that cannot start until a wait times out).
  static const int max_blocks = [] {
    const char* e = getenv("NV_COMM_MAX_BLOCKS");
    const int v = e ? atoi(e) : 0;
    return v >= 8 && v <= 96 ? v : 96;
  }();
  if (blocks > max_blocks) blocks = max_blocks;
  dim3 grid((unsigned)blocks), block(256);
  hipStream_t s = (hipStream_t)stream;
#define GO(TT, NN)                                                                                                  \
  allreduce_kernel<TT, NN><<<grid, block, 0, s>>>(c->a, (const uint16_t*)x, partials, num_partials,                 \
                                                  (const uint16_t*)residual, (const uint16_t*)weight, (uint16_t*)out, \
                                                  (uint16_t*)residual_out, (int)rows, hidden, eps)
  if (dtype == NV_BF16) { if (norm) GO(bf16_t, true); else GO(bf16_t, false); }
  else                  { if (norm) GO(f16_t, true); else GO(f16_t, false); }
#undef GO
  NV_CHECK_LAUNCH();
  return NV_OK;
}

extern "C" int nv_allreduce_add_rmsnorm(nv_comm* c, const void* x, const float* partials, int num_partials,
                                        const void* residual, const void* weight, void* out, void* residual_out,
                                        int64_t rows, int hidden, float eps, int dtype, nv_stream_t stream) {
  return allreduce_go(c, x, partials, num_partials, residual, weight, out, residual_out, rows, hidden, eps, dtype, true,
                      stream, "nv_allreduce_add_rmsnorm");
}

extern "C" int nv_allreduce(nv_comm* c, const void* x, const float* partials, int num_partials, void* out, int64_t rows,
                            int hidden, int dtype, nv_stream_t stream) {
  return allreduce_go(c, x, partials, num_partials, nullptr, nullptr, out, nullptr, rows, hidden, 0.f, dtype, false,
                      stream, "nv_allreduce");
}
