// Instruction-fetch probe: what does a kernel pay per launch for the size of its code?  The same dependent chain of
// N fmas runs (a) as a loop of a few instructions, (b) fully unrolled (8 bytes of code per fma), as dependent nodes of
// a hipGraph; kernels alternate with a different small kernel in between (as the layers of a decode step do).  The
// difference per node between (b) and (a) is the price of fetching the code again after a kernel boundary.
// Build: hipcc -O3 --offload-arch=gfx950 icache.hip -o icache
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int N>
__global__ void chain_loop(float* p, int n) {
  float v = p[threadIdx.x & 63];
#pragma unroll 1
  for (int i = 0; i < n; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
  if (v == 123.456f) p[0] = v;
}

template <int N>
__global__ void chain_unrolled(float* p, int n) {
  float v = p[threadIdx.x & 63];
#pragma unroll
  for (int i = 0; i < N; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
  if (v == 123.456f) p[0] = v + n;
}

__global__ void other(float* p) {
  if (p[threadIdx.x & 63] == 123.456f) p[1] = 1.f;
}

template <typename F>
static double graph_us(hipStream_t s, int nodes, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < nodes; ++i) launch(i);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 20;
  hipEventRecord(a, s);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(b, s); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3 / (reps * nodes);
}

template <int N>
static void run(hipStream_t s, float* buf, int blocks) {
  const int nodes = 200;
  double loop = graph_us(s, nodes, [&](int) { chain_loop<N><<<blocks, 256, 0, s>>>(buf, N); });
  double unr = graph_us(s, nodes, [&](int) { chain_unrolled<N><<<blocks, 256, 0, s>>>(buf, N); });
  double loop_alt = graph_us(s, nodes, [&](int i) { if (i & 1) other<<<blocks, 256, 0, s>>>(buf); else chain_loop<N><<<blocks, 256, 0, s>>>(buf, N); });
  double unr_alt = graph_us(s, nodes, [&](int i) { if (i & 1) other<<<blocks, 256, 0, s>>>(buf); else chain_unrolled<N><<<blocks, 256, 0, s>>>(buf, N); });
  printf("blocks %4d, %5d fmas (%3d KB unrolled): loop %.2f us/node, unrolled %.2f | alternating with a small kernel: "
         "loop %.2f us per pair, unrolled %.2f\n", blocks, N, N * 8 / 1024, loop, unr, 2 * loop_alt, 2 * unr_alt);
}

int main() {
  float* buf; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20);
  hipStream_t s; hipStreamCreate(&s);
  for (int blocks : {128, 256}) {
    run<256>(s, buf, blocks);
    run<1024>(s, buf, blocks);
    run<2048>(s, buf, blocks);
    run<4096>(s, buf, blocks);
  }
  return 0;
}
