// Dispatch-rate probe: how long does one dependent kernel node cost in a hipGraph / in a stream, as a function of the
// kernel's own duration?  Build: hipcc -O3 --offload-arch=gfx950 dispatch.hip -o dispatch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x & 63];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.456f) p[0] = v;
}

static double run_graph(hipStream_t s, float* buf, int nodes, int iters, int blocks, int threads) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < nodes; ++i) spin<<<blocks, threads, 0, s>>>(buf, iters);
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

static double run_stream(hipStream_t s, float* buf, int nodes, int iters, int blocks, int threads) {
  for (int i = 0; i < nodes; ++i) spin<<<blocks, threads, 0, s>>>(buf, iters);
  hipStreamSynchronize(s);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 20;
  hipEventRecord(a, s);
  for (int r = 0; r < reps; ++r)
    for (int i = 0; i < nodes; ++i) spin<<<blocks, threads, 0, s>>>(buf, iters);
  hipEventRecord(b, s); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / (reps * nodes);
}

int main() {
  float* buf; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20);
  hipStream_t s; hipStreamCreate(&s);
  const int nodes = 200;
  int its[] = {0, 200, 1000, 2000, 4000, 8000};
  for (int blocks : {1, 256, 1024}) {
    for (int iters : its) {
      // a lone launch of the same kernel, for its own duration
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      spin<<<blocks, 256, 0, s>>>(buf, iters); hipStreamSynchronize(s);
      double g = run_graph(s, buf, nodes, iters, blocks, 256);
      double st = run_stream(s, buf, nodes, iters, blocks, 256);
      printf("blocks %4d spin %5d: graph %.2f us/node   stream %.2f us/kernel\n", blocks, iters, g, st);
    }
  }
  return 0;
}
