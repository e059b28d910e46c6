// Stand-alone timing and cycle-stamp probe of ONE instantiation of the decode GEMM (no torch, no library):
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 [-DNV_STAMP] [-DPROBE_NTW=3 -DPROBE_EPI=1 ...] \
//         scripts/micro/gemm_probe.hip -o scripts/micro/gemm_probe
//   ./gemm_probe M N K [split_k] [copies]
// Prints the average launch time over `copies` distinct weight buffers (so nothing is re-read from the Infinity
// Cache) and, in a -DNV_STAMP build, where a wavefront's cycles go: prologue, phase A, barrier wait, phase B, epilogue.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
void nv_set_error(const char*, ...) {}
#include "../../byox_vllm_amd/csrc/gemm_impl.h"
#include "../../byox_vllm_amd/csrc/gemm_wide.h"

#ifndef PROBE_NTW
#define PROBE_NTW 3
#endif
#ifndef PROBE_EPI
#define PROBE_EPI 1
#endif
#ifndef PROBE_BM
#define PROBE_BM 128
#endif
#ifndef PROBE_WFMT
#define PROBE_WFMT NV_W_INT4
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 128, N = argc > 2 ? atoi(argv[2]) : 37888, K = argc > 3 ? atoi(argv[3]) : 3584;
  const int split = argc > 4 ? atoi(argv[4]) : 1, copies = argc > 5 ? atoi(argv[5]) : 6;
  const size_t wbytes = PROBE_WFMT == NV_W_INT4 ? (size_t)N * K / 2 : (PROBE_WFMT == NV_W_INT8 ? (size_t)N * K : (size_t)N * K * 2);
  const size_t sbytes = (size_t)N * (K / 128) * 2;
  std::vector<uint32_t> hw(wbytes / 4);
  uint32_t r = 12345;
  for (auto& v : hw) { r = r * 1664525u + 1013904223u; v = r; }
  std::vector<uint16_t> hs(sbytes / 2), hx((size_t)M * K);
  for (auto& v : hs) { r = r * 1664525u + 1013904223u; v = 0x3c00 + (r >> 26); }   // bf16 ~0.0078..0.0156
  for (auto& v : hx) { r = r * 1664525u + 1013904223u; v = (uint16_t)((r >> 16) & 0xbfff) | 0x3000; v &= 0xbf7f; }
  if (getenv("PROBE_ZERO")) {   // all-zero operands: the same instruction stream at the lowest switching activity
    std::fill(hw.begin(), hw.end(), 0x88888888u);
    std::fill(hx.begin(), hx.end(), (uint16_t)0);
  }
  std::vector<void*> dw(copies), ds(copies);
  for (int c = 0; c < copies; ++c) {
    CK(hipMalloc(&dw[c], wbytes)); CK(hipMalloc(&ds[c], sbytes));
    CK(hipMemcpy(dw[c], hw.data(), wbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds[c], hs.data(), sbytes, hipMemcpyHostToDevice));
  }
  void *dx, *dout; float* dpart;
  CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&dout, (size_t)M * N * 2));
  const size_t pbytes = std::max((size_t)split * M * N * 4, (size_t)1 << 22);
  CK(hipMalloc((void**)&dpart, pbytes)); CK(hipMemset(dpart, 0, pbytes));
  const int ntiles = N / 16, per_block = 4 * PROBE_NTW;
  dim3 grid((ntiles + per_block - 1) / per_block, (M + PROBE_BM - 1) / PROBE_BM, split);
#ifdef PROBE_WIDE   // the 32x32x16 kernel: 192 columns per workgroup, 768 threads
  grid = dim3((N / 32 + GEMM_WIDE_NCG - 1) / GEMM_WIDE_NCG, (M + 127) / 128, split);
  auto launch = [&](int c) {
    gemm_wide_w4_kernel<bf16_t, PROBE_EPI><<<grid, dim3(GEMM_WIDE_THREADS)>>>(
        (const uint16_t*)dx, K, dw[c], (const uint16_t*)ds[c], nullptr, (uint16_t*)dout, PROBE_EPI == 1 ? N / 2 : N,
        dpart, M, N, K, split);
  };
#else
  auto launch = [&](int c) {
    gemm_kernel<bf16_t, PROBE_WFMT, PROBE_BM, PROBE_NTW, PROBE_EPI><<<grid, dim3(512)>>>(
        (const uint16_t*)dx, K, dw[c], (const uint16_t*)ds[c], nullptr, (uint16_t*)dout, PROBE_EPI == 1 ? N / 2 : N,
        dpart, M, N, K, split, RopeArgs{}, SampleParts{});
  };
#endif
  for (int c = 0; c < copies; ++c) launch(c);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int reps = 20;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) for (int c = 0; c < copies; ++c) launch(c);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / (reps * copies);
  printf("M=%d N=%d K=%d split=%d grid=%u x %u x %u NTW=%d BM=%d EPI=%d: %.2f us  %.1f TFLOP/s  %.2f TB/s(w)\n", M, N, K, split,
         grid.x, grid.y, grid.z, PROBE_NTW, PROBE_BM, PROBE_EPI, us, 2.0 * M * N * K / us / 1e6, wbytes / us / 1e6);
#ifdef NV_STAMP
  std::vector<uint64_t> st((size_t)grid.x * 64);
  CK(hipMemcpy(st.data(), dpart, st.size() * 8, hipMemcpyDeviceToHost));
  double sum[6] = {0, 0, 0, 0, 0, 0};
  for (unsigned g = 0; g < grid.x; ++g) for (int w = 0; w < 8; ++w) for (int i = 0; i < 6; ++i) sum[i] += st[(g * 8 + w) * 8 + i];
  const double n = grid.x * 8.0;
  printf("cycles per wavefront (avg): prologue %.0f | phaseA %.0f  barrier %.0f  phaseB %.0f (sum over %d K-blocks) | loop total %.0f | epilogue %.0f\n",
         sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, K / 128 / split, sum[4] / n, sum[5] / n);
  for (int w = 0; w < 8; ++w)
    printf("  wg0 wave %d: pro %llu A %llu bar %llu B %llu loop %llu epi %llu\n", w, (unsigned long long)st[w * 8], (unsigned long long)st[w * 8 + 1],
           (unsigned long long)st[w * 8 + 2], (unsigned long long)st[w * 8 + 3], (unsigned long long)st[w * 8 + 4], (unsigned long long)st[w * 8 + 5]);
  // clock: memtime ticks vs memrealtime (100 MHz) across workgroups
  uint64_t t0min = ~0ull, t0max = 0, rmin = ~0ull, rmax = 0;
  for (unsigned g = 0; g < grid.x; ++g) { t0min = std::min(t0min, st[g * 64 + 6]); t0max = std::max(t0max, st[g * 64 + 6]);
    rmin = std::min(rmin, st[g * 64 + 7]); rmax = std::max(rmax, st[g * 64 + 7]); }
  printf("start skew %llu cycles across workgroups\n", (unsigned long long)(t0max - t0min));
#endif
  return 0;
}
