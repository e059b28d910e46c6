// Reading a row that exists only as the GEMM's split-K slabs (fp32, slab z at part + z * slab): the slabs are
// summed in slab order, exactly as the GEMM's own reduction does, so every consumer rounds the same bits.
// Loads go out in batches of up to 8 slabs before the first add, so a row costs one memory latency per batch,
// not one per slab.
#pragma once
#include "common.h"

// NV consecutive float4 per slab (4 * NV columns)
template <int NV, int NP, bool FIRST>
__device__ __forceinline__ void slab_batch(const float* __restrict__ p, int64_t slab, f32x4 (&acc)[NV]) {
  f32x4 v[NP][NV];
#pragma unroll
  for (int z = 0; z < NP; ++z)
#pragma unroll
    for (int i = 0; i < NV; ++i) v[z][i] = *reinterpret_cast<const f32x4*>(p + z * slab + 4 * i);
#pragma unroll
  for (int z = 0; z < NP; ++z)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (FIRST && z == 0) acc[i] = v[0][i];
      else acc[i] += v[z][i];
    }
}

template <int NV, bool FIRST>
__device__ __forceinline__ void slab_batch_n(const float* __restrict__ p, int64_t slab, int n, f32x4 (&acc)[NV]) {
  switch (n) {  // uniform over the launch
    case 1: slab_batch<NV, 1, FIRST>(p, slab, acc); break;
    case 2: slab_batch<NV, 2, FIRST>(p, slab, acc); break;
    case 3: slab_batch<NV, 3, FIRST>(p, slab, acc); break;
    case 4: slab_batch<NV, 4, FIRST>(p, slab, acc); break;
    case 5: slab_batch<NV, 5, FIRST>(p, slab, acc); break;
    case 6: slab_batch<NV, 6, FIRST>(p, slab, acc); break;
    case 7: slab_batch<NV, 7, FIRST>(p, slab, acc); break;
    default: slab_batch<NV, 8, FIRST>(p, slab, acc); break;
  }
}

// acc[i] = sum over z < nparts of the float4 at p + z * slab + 4 i   (nparts >= 1), loads issued in batches of up to
// BATCH slabs (8, or 4 where the caller is short of registers); the order of the additions is the slab order either way
template <int NV, int BATCH = 8>
__device__ __forceinline__ void slab_sum(const float* __restrict__ p, int64_t slab, int nparts, f32x4 (&acc)[NV]) {
  static_assert(BATCH == 4 || BATCH == 8, "batches of 4 or 8 slabs");
  int n = nparts < BATCH ? nparts : BATCH;
  slab_batch_n<NV, true>(p, slab, n, acc);
  for (int z = n; z < nparts; z += BATCH) {
    n = nparts - z < BATCH ? nparts - z : BATCH;
    slab_batch_n<NV, false>(p + z * slab, slab, n, acc);
  }
}
