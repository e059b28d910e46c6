// Cost of a device-side grid barrier on gfx950: G co-resident workgroups of 512 threads meet `iters` times through one
// monotonically increasing counter in device memory (agent-scope atomics).  Every spin is bounded: a barrier that does
// not complete sets an error flag and all later barriers are skipped, so the grid always drains.
// Build: hipcc -O3 --offload-arch=gfx950 gridbarrier.hip -o gridbarrier
#include <hip/hip_runtime.h>
#include <stdio.h>

// Two-level form: arrivals are counted per group of 8 consecutive workgroups... no: per residue of the workgroup index
// modulo 8 (consecutive workgroups go to different XCDs, so a residue class shares an XCD and its L2), the last of a
// class reports to a global counter, the last of those publishes the phase number in a flag that everybody polls.
__global__ __launch_bounds__(512) void meet2(unsigned* cnt /* [8 * 32] class counters, 128 B apart */, unsigned* top,
                                             unsigned* flag, unsigned* err, int iters, int work, float* sink) {
  float acc = threadIdx.x;
  const unsigned cls = blockIdx.x & 7, per = gridDim.x >> 3;
  for (int it = 0; it < iters; ++it) {
    for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      const unsigned phase = (unsigned)(it + 1);
      const unsigned old = __hip_atomic_fetch_add(&cnt[cls * 32], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old == phase * per - 1) {
        const unsigned old2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old2 == phase * 8 - 1) __hip_atomic_store(flag, phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) {   // no invalidate per poll
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 20)) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// Flag-array form, no read-modify-write at all: workgroup b stores the phase number into its own word flags[b]
// (write-through), and the first wavefront of every workgroup polls ALL words, lane l reading flags[l], flags[l + 64],
// ...: one vector load per 64 workgroups and poll, no serialisation of arrivals at one address.
__global__ __launch_bounds__(512) void meet3(unsigned* flags, unsigned* err, int iters, int work, float* sink) {
  float acc = threadIdx.x;
  const int G = gridDim.x;
  for (int it = 0; it < iters; ++it) {
    for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
    __syncthreads();
    if (threadIdx.x < 64 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      const unsigned phase = (unsigned)(it + 1);
      if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      for (;;) {
        bool ok = true;
        for (int b = threadIdx.x; b < G; b += 64)
          ok = ok && __hip_atomic_load(&flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 18)) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(512) void meet(unsigned* counter, unsigned* err, int iters, int work, float* sink) {
  float acc = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;   // stand-in for a phase of work
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      const unsigned target = (unsigned)(it + 1) * gridDim.x;
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 20)) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  unsigned *counter, *err;
  float* sink;
  hipMalloc(&counter, 4); hipMalloc(&err, 4); hipMalloc(&sink, 1024 * 512 * 4);
  const int grids[] = {64, 256, 512};
  for (int gi = 0; gi < 3; ++gi) {
    const int G = grids[gi];
    for (int work = 0; work <= 1000; work += 1000) {
      float ms[2];
      for (int pass = 0; pass < 2; ++pass) {
        const int iters = pass == 0 ? 100 : 1100;
        hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        meet<<<G, 512>>>(counter, err, iters, work, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms[pass], a, b);
      }
      unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
      printf("grid %3d x 512, %4d fma of work per phase: %.2f us per phase+barrier%s\n", G, work,
             (ms[1] - ms[0]) * 1000.0 / 1000.0, e ? "  (TIMED OUT)" : "");
    }
  }
  unsigned *cnt, *top, *flag;
  hipMalloc(&cnt, 8 * 32 * 4); hipMalloc(&top, 4); hipMalloc(&flag, 4);
  for (int gi = 0; gi < 3; ++gi) {
    const int G = grids[gi];
    float ms[2];
    for (int pass = 0; pass < 2; ++pass) {
      const int iters = pass == 0 ? 100 : 1100;
      hipMemset(cnt, 0, 8 * 32 * 4); hipMemset(top, 0, 4); hipMemset(flag, 0, 4); hipMemset(err, 0, 4);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      meet2<<<G, 512>>>(cnt, top, flag, err, iters, 0, sink);
      hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms[pass], a, b);
    }
    unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
    printf("two-level, grid %3d x 512: %.2f us per barrier%s\n", G, (ms[1] - ms[0]) * 1000.0 / 1000.0, e ? "  (TIMED OUT)" : "");
  }
  unsigned* flags;
  hipMalloc(&flags, 512 * 4);
  const int grids3[] = {16, 64, 112, 256};
  for (int gi = 0; gi < 4; ++gi) {
    const int G = grids3[gi];
    float ms[2];
    for (int pass = 0; pass < 2; ++pass) {
      const int iters = pass == 0 ? 100 : 1100;
      hipMemset(flags, 0, 512 * 4); hipMemset(err, 0, 4);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      meet3<<<G, 512>>>(flags, err, iters, 0, sink);
      hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms[pass], a, b);
    }
    unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
    printf("flag array, grid %3d x 512: %.2f us per barrier%s\n", G, (ms[1] - ms[0]) * 1000.0 / 1000.0, e ? "  (TIMED OUT)" : "");
  }
  return 0;
}
