// Overlap probe: do two chains of dependent kernels on two streams run concurrently on this part, launched eagerly
// from two host threads or as two graphs?  Build: hipcc -O3 --offload-arch=gfx950 two_streams.hip -o two_streams -lpthread
// Each kernel spins for a given number of iterations on `blocks` workgroups (half the chip), so two chains fit side by
// side; a chain of N kernels alone takes N x (floor + spin), two chains that overlap take about the same wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>

__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x & 63];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.456f) p[0] = v;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void launch_chain(hipStream_t s, float* buf, int n, int iters, int blocks) {
  for (int i = 0; i < n; ++i) spin<<<blocks, 256, 0, s>>>(buf, iters);
}

int main() {
  float *b0, *b1;
  hipMalloc(&b0, 1 << 20); hipMalloc(&b1, 1 << 20);
  hipMemset(b0, 0, 1 << 20); hipMemset(b1, 0, 1 << 20);
  hipStream_t s0, s1;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  const int n = 350, reps = 5;
  for (int blocks : {64, 128, 256}) {
    for (int iters : {0, 1000, 4000}) {
      launch_chain(s0, b0, n, iters, blocks); launch_chain(s1, b1, n, iters, blocks);
      hipDeviceSynchronize();
      // one chain, one thread
      double t0 = now_us();
      for (int r = 0; r < reps; ++r) { launch_chain(s0, b0, n, iters, blocks); hipStreamSynchronize(s0); }
      double one = (now_us() - t0) / reps;
      // two chains, two host threads
      t0 = now_us();
      for (int r = 0; r < reps; ++r) {
        std::thread th([&] { launch_chain(s1, b1, n, iters, blocks); hipStreamSynchronize(s1); });
        launch_chain(s0, b0, n, iters, blocks); hipStreamSynchronize(s0);
        th.join();
      }
      double two = (now_us() - t0) / reps;
      // two chains, one host thread, launches interleaved
      t0 = now_us();
      for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < n; ++i) { spin<<<blocks, 256, 0, s0>>>(b0, iters); spin<<<blocks, 256, 0, s1>>>(b1, iters); }
        hipStreamSynchronize(s0); hipStreamSynchronize(s1);
      }
      double inter = (now_us() - t0) / reps;
      // the same as graphs: one graph alone, two graphs on two streams
      hipGraph_t g0, g1; hipGraphExec_t e0, e1;
      hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal); launch_chain(s0, b0, n, iters, blocks); hipStreamEndCapture(s0, &g0);
      hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal); launch_chain(s1, b1, n, iters, blocks); hipStreamEndCapture(s1, &g1);
      hipGraphInstantiate(&e0, g0, nullptr, nullptr, 0); hipGraphInstantiate(&e1, g1, nullptr, nullptr, 0);
      hipGraphLaunch(e0, s0); hipGraphLaunch(e1, s1); hipDeviceSynchronize();
      t0 = now_us();
      for (int r = 0; r < reps; ++r) { hipGraphLaunch(e0, s0); hipStreamSynchronize(s0); }
      double gone = (now_us() - t0) / reps;
      t0 = now_us();
      for (int r = 0; r < reps; ++r) { hipGraphLaunch(e0, s0); hipGraphLaunch(e1, s1); hipStreamSynchronize(s0); hipStreamSynchronize(s1); }
      double gtwo = (now_us() - t0) / reps;
      // host cost of launching alone (no sync until the end)
      t0 = now_us();
      launch_chain(s0, b0, n, iters, blocks);
      double host = (now_us() - t0);
      hipDeviceSynchronize();
      printf("blocks %3d spin %4d, %d kernels per chain: eager one %.0f us (%.2f/kernel), two threads %.0f, interleaved %.0f; "
             "graph one %.0f (%.2f/node), two graphs %.0f; host launch cost %.2f us/kernel\n",
             blocks, iters, n, one, one / n, two, inter, gone, gone / n, gtwo, host / n);
      hipGraphExecDestroy(e0); hipGraphExecDestroy(e1); hipGraphDestroy(g0); hipGraphDestroy(g1);
    }
  }
  return 0;
}
