// Issue-rate probe for gfx950, part two: v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 with growing numbers
// of independent v_fma_f32 between them.  What it answers: how long one MFMA keeps the SIMD's VALU port, i.e. how many
// plain VALU instructions fit per MFMA before the VALU, not the matrix pipe, sets the pace.
// Build: hipcc -O3 --offload-arch=gfx950 coissue32.hip -o coissue32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define M32(n) "v_mfma_f32_32x32x16_bf16 %" #n ", %8, %9, %" #n "\n"
#define M16(n) "v_mfma_f32_16x16x32_bf16 %" #n ", %8, %9, %" #n "\n"
#define F(n) "v_fma_f32 %" #n ", %" #n ", %10, %10\n"
#define F3 F(4) F(5) F(6)
#define F6 F3 F(7) F(4) F(5)
#define F9 F6 F(6) F(7) F(4)
#define F12 F9 F(5) F(6) F(7)
#define OPS(T) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(a), "v"(b), "v"(c)

template <int BIG, int NF>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  typedef typename std::conditional<BIG, f32x16, f32x4>::type acc_t;
  acc_t a0, a1, a2, a3;
  for (int i = 0; i < (BIG ? 16 : 4); ++i) { a0[i] = 0; a1[i] = 0; a2[i] = 0; a3[i] = 0; }
  float p0 = 1.f, p1 = 1.f, p2 = 1.f, p3 = 1.f, c = 0.5f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#define BODY(MM, FF) asm volatile(MM(0) FF MM(1) FF MM(2) FF MM(3) FF OPS(0))
    if (BIG) {
      if (NF == 0) BODY(M32, ""); else if (NF == 3) BODY(M32, F3); else if (NF == 6) BODY(M32, F6);
      else if (NF == 9) BODY(M32, F9); else BODY(M32, F12);
    } else {
      if (NF == 0) BODY(M16, ""); else if (NF == 3) BODY(M16, F3); else if (NF == 6) BODY(M16, F6);
      else if (NF == 9) BODY(M16, F9); else BODY(M16, F12);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + p0 + p1 + p2 + p3;
}

template <int BIG, int NF>
static void run(float* out) {
  const int iters = 20000;
  for (int threads = 256; threads <= 512; threads *= 2) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<BIG, NF><<<256, threads>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<BIG, NF><<<256, threads>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ns = ms * 1e6 / iters / 4;   // per MFMA of one wavefront
    printf("%s + %2d v_fma each, %d wavefronts per SIMD: %6.2f ns per MFMA per wavefront = %6.2f ns per 16x16x32-equivalent per SIMD\n",
           BIG ? "32x32x16" : "16x16x32", NF, threads / 256, ns, ns / (BIG ? 2 : 1) / (threads / 256));
  }
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0, 0>(out); run<0, 3>(out); run<0, 6>(out); run<0, 9>(out); run<0, 12>(out);
  run<1, 0>(out); run<1, 3>(out); run<1, 6>(out); run<1, 9>(out); run<1, 12>(out);
  return 0;
}
