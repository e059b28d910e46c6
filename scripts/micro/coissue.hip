// Issue-rate probe for gfx950: how many cycles does one 16x16x32 bf16 MFMA cost alone, and with independent VALU /
// LDS instructions placed in its shadow?  Build: hipcc -O3 --offload-arch=gfx950 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(n) "v_mfma_f32_16x16x32_bf16 %" #n ", %9, %10, %" #n "\n"
#define PK(n) "v_pk_fma_f32 %" #n ", %" #n ", %11, %11\n"
#define CVT(n) "v_cvt_f32_ubyte0_e32 %" #n ", %12\n"
#define FMA(n) "v_fma_f32 %" #n ", %" #n ", %12, %12\n"
#define CVTPK(n) "v_cvt_pk_bf16_f32 %" #n ", %" #n ", %12\n"
#define ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %12, %12\n"
#define SHR(n) "v_lshrrev_b32_e32 %" #n ", 4, %" #n "\n"
#define PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %11\n"
#define DSW "ds_write_b128 %13, %8\n"
#define DSR "ds_read_b128 %8, %13\n"


#define WITH_MFMA(OP) MFMA(0) OP(4) OP(5) OP(6) MFMA(1) OP(7) OP(4) OP(5) MFMA(2) OP(6) OP(7) OP(4) MFMA(3) OP(5) OP(6) OP(7)
#define ALONE(OP) OP(4) OP(5) OP(6) OP(7) OP(4) OP(5) OP(6) OP(7) OP(4) OP(5) OP(6) OP(7)
#define SCALAR_OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0[0]), "+v"(p1[0]), "+v"(p2[0]), "+v"(p3[0]), "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(u), "v"(addr)
#define PAIR_OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(u), "v"(addr)

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  __shared__ u32x4 lds[2048];
  lds[threadIdx.x] = u32x4{threadIdx.x, 1, 2, 3};
  __syncthreads();
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  f32x2 p0 = {1.f, 1.f}, p1 = p0, p2 = p0, p3 = p0;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  f32x2 c = {0.5f, 0.5f};
  uint32_t u = threadIdx.x;
  u32x4 d = {0, 0, 0, 0};
  uint32_t addr = threadIdx.x * 16;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0)       // 4 MFMA
      asm volatile(MFMA(0) MFMA(1) MFMA(2) MFMA(3)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(d)
                   : "v"(a), "v"(b), "v"(c), "v"(u), "v"(addr));
    else if (MODE == 1)  // 4 x (MFMA + 3 packed fma)
      asm volatile(MFMA(0) PK(4) PK(5) PK(6) MFMA(1) PK(7) PK(4) PK(5) MFMA(2) PK(6) PK(7) PK(4) MFMA(3) PK(5) PK(6) PK(7)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(d)
                   : "v"(a), "v"(b), "v"(c), "v"(u), "v"(addr));
    else if (MODE == 2)  // 12 packed fma only
      asm volatile(PK(4) PK(5) PK(6) PK(7) PK(4) PK(5) PK(6) PK(7) PK(4) PK(5) PK(6) PK(7)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(d)
                   : "v"(a), "v"(b), "v"(c), "v"(u), "v"(addr));
    else if (MODE == 3)  // 4 x (MFMA + 1 ds_read_b128)
      asm volatile(MFMA(0) DSR MFMA(1) DSR MFMA(2) DSR MFMA(3) DSR "s_waitcnt lgkmcnt(0)\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(d)
                   : "v"(a), "v"(b), "v"(c), "v"(u), "v"(addr));
    else if (MODE == 4)  // 4 x (MFMA + 6 packed fma): VALU-issue bound?
      asm volatile(MFMA(0) PK(4) PK(5) PK(6) PK(7) PK(4) PK(5) MFMA(1) PK(6) PK(7) PK(4) PK(5) PK(6) PK(7)
                   MFMA(2) PK(4) PK(5) PK(6) PK(7) PK(4) PK(5) MFMA(3) PK(6) PK(7) PK(4) PK(5) PK(6) PK(7)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(d)
                   : "v"(a), "v"(b), "v"(c), "v"(u), "v"(addr));
    else if (MODE == 5)  // 4 x (MFMA + 3 cvt)
      asm volatile(MFMA(0) CVT(4) CVT(5) CVT(6) MFMA(1) CVT(7) CVT(4) CVT(5) MFMA(2) CVT(6) CVT(7) CVT(4) MFMA(3) CVT(5) CVT(6) CVT(7)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0[0]), "+v"(p1[0]), "+v"(p2[0]), "+v"(p3[0]), "+v"(d)
                   : "v"(a), "v"(b), "v"(c), "v"(u), "v"(addr));
    else if (MODE == 10) asm volatile(WITH_MFMA(FMA) SCALAR_OPS);
    else if (MODE == 11) asm volatile(ALONE(FMA) SCALAR_OPS);
    else if (MODE == 12) asm volatile(WITH_MFMA(CVTPK) SCALAR_OPS);
    else if (MODE == 13) asm volatile(ALONE(CVTPK) SCALAR_OPS);
    else if (MODE == 14) asm volatile(WITH_MFMA(ANDOR) SCALAR_OPS);
    else if (MODE == 15) asm volatile(ALONE(ANDOR) SCALAR_OPS);
    else if (MODE == 16) asm volatile(WITH_MFMA(SHR) SCALAR_OPS);
    else if (MODE == 17) asm volatile(ALONE(SHR) SCALAR_OPS);
    else if (MODE == 18) asm volatile(WITH_MFMA(PKMUL) PAIR_OPS);
    else if (MODE == 19) asm volatile(ALONE(PKMUL) PAIR_OPS);
    else if (MODE == 30) asm volatile(MFMA(0) DSW MFMA(1) DSW MFMA(2) DSW MFMA(3) DSW SCALAR_OPS);
    else if (MODE == 31) asm volatile(MFMA(0) MFMA(1) MFMA(2) MFMA(3) DSW SCALAR_OPS);
    else if (MODE == 32) asm volatile(MFMA(0) DSR MFMA(1) DSW MFMA(2) DSR MFMA(3) DSR "s_waitcnt lgkmcnt(0)\n" SCALAR_OPS);
    else if (MODE == 33) asm volatile(DSW DSW DSW DSW SCALAR_OPS);
    else if (MODE == 34) asm volatile(DSR DSR DSR DSR "s_waitcnt lgkmcnt(0)\n" SCALAR_OPS);
    else if (MODE == 20) asm volatile(ALONE(CVT) SCALAR_OPS);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + p0[0] + p1[1] + p2[0] + p3[1] + (float)d[0];
}

template <int MODE>
static void run(const char* what, int per_iter_mfma, float* out) {
  const int iters = 20000;
  for (int threads = 256; threads <= 512; threads *= 2) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<256, threads>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<256, threads>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ns_iter = ms * 1e6 / iters;
    printf("%-28s waves/CU %d: %7.2f ns per iteration", what, threads / 64, ns_iter);
    if (per_iter_mfma) printf("  (%.2f ns per MFMA per wave)", ns_iter / per_iter_mfma);
    printf("\n");
  }
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0>("4 MFMA", 4, out);
  run<1>("4 MFMA + 12 pk_fma", 4, out);
  run<2>("12 pk_fma", 0, out);
  run<3>("4 MFMA + 4 ds_read_b128", 4, out);
  run<4>("4 MFMA + 24 pk_fma", 4, out);
  run<5>("4 MFMA + 12 cvt_ubyte", 4, out);
  run<30>("4 MFMA + 4 ds_write_b128", 4, out);
  run<31>("4 MFMA + 1 ds_write_b128", 4, out);
  run<32>("4 MFMA + 3 read + 1 write", 4, out);
  run<33>("4 ds_write_b128", 0, out);
  run<34>("4 ds_read_b128", 0, out);
  run<20>("12 cvt_ubyte", 0, out);
  run<10>("4 MFMA + 12 v_fma_f32", 4, out);
  run<11>("12 v_fma_f32", 0, out);
  run<12>("4 MFMA + 12 cvt_pk_bf16", 4, out);
  run<13>("12 cvt_pk_bf16", 0, out);
  run<14>("4 MFMA + 12 and_or", 4, out);
  run<15>("12 and_or", 0, out);
  run<16>("4 MFMA + 12 lshr", 4, out);
  run<17>("12 lshr", 0, out);
  run<18>("4 MFMA + 12 pk_mul_f32", 4, out);
  run<19>("12 pk_mul_f32", 0, out);
  return 0;
}
