// Which property of a kernel makes its graph node expensive?  200-node chains of one kernel shape; per-node time minus
// the kernel's own (single-launch, event-timed) duration is the dispatch cost.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int LDS>
__global__ void k(float* p, int writes, int reads) {
  __shared__ float s[LDS / 4 + 1];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float v = 0.f;
  for (int i = 0; i < reads; ++i) v += p[(size_t)i * gridDim.x * blockDim.x + gid];
  if (LDS > 0) { s[threadIdx.x % (LDS / 4 + 1)] = v; __syncthreads(); v += s[(threadIdx.x * 7) % (LDS / 4 + 1)]; }
  for (int i = 0; i < writes; ++i) p[(size_t)i * gridDim.x * blockDim.x + gid] = v + i;
}

template <int LDS>
static void probe(const char* what, hipStream_t s, float* buf, int blocks, int threads, int writes, int reads) {
  const int nodes = 200, reps = 20;
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < nodes; ++i) k<LDS><<<blocks, threads, 0, s>>>(buf, writes, reads);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, s);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(b, s); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-44s blocks %4d x %3d thr, LDS %5d, writes %2d reads %2d: %.2f us/node\n", what, blocks, threads, LDS, writes, reads,
         ms * 1e3 / (reps * nodes));
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
}

struct Big { void* p[24]; long n[8]; };
template <int ID>
__global__ void kb(float* p, Big b, int writes) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float v = (float)(ID + b.n[ID & 7]);
  for (int i = 0; i < writes; ++i) p[(size_t)i * gridDim.x * blockDim.x + gid + ID] = v + i;
}

template <int NK>
static void probe_mix(const char* what, hipStream_t s, float* buf, int writes) {
  const int nodes = 200, reps = 20;
  Big b = {};
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < nodes; ++i) {
    switch (i % NK) {
      case 0: kb<0><<<128, 128, 0, s>>>(buf, b, writes); break;
      case 1: kb<1><<<152, 512, 0, s>>>(buf, b, writes); break;
      case 2: kb<2><<<72, 256, 0, s>>>(buf, b, writes); break;
      case 3: kb<3><<<256, 512, 0, s>>>(buf, b, writes); break;
      case 4: kb<4><<<98, 512, 0, s>>>(buf, b, writes); break;
      case 5: kb<5><<<128, 128, 0, s>>>(buf, b, writes); break;
      case 6: kb<6><<<152, 512, 0, s>>>(buf, b, writes); break;
      default: kb<7><<<182, 512, 0, s>>>(buf, b, writes); break;
    }
  }
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEvent_t a, e; hipEventCreate(&a); hipEventCreate(&e);
  hipEventRecord(a, s);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(e, s); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, a, e);
  printf("%-44s %d different kernels, 256-byte kernarg, writes %d: %.2f us/node\n", what, NK, writes, ms * 1e3 / (reps * nodes));
}

int main() {
  float* buf; hipMalloc(&buf, 256u << 20); hipMemset(buf, 0, 256u << 20);
  hipStream_t s; hipStreamCreate(&s);
  probe<0>("empty", s, buf, 1, 64, 0, 0);
  probe<0>("256 small blocks", s, buf, 256, 256, 0, 0);
  probe<0>("256 blocks of 512", s, buf, 256, 512, 0, 0);
  probe<33792>("256 blocks of 512 + 33 KB LDS", s, buf, 256, 512, 0, 0);
  probe<65536>("152 blocks of 512 + 64 KB LDS", s, buf, 152, 512, 0, 0);
  probe<0>("128 blocks of 128, 1 write (64 KB)", s, buf, 128, 128, 1, 0);
  probe<0>("128 blocks of 128, 8 writes (512 KB)", s, buf, 128, 128, 8, 0);
  probe<0>("256 blocks of 256, 16 writes (4 MB)", s, buf, 256, 256, 16, 0);
  probe<0>("256 blocks of 256, 16 reads (4 MB)", s, buf, 256, 256, 0, 16);
  probe<0>("256 blocks of 256, 16 reads + 16 writes", s, buf, 256, 256, 16, 16);
  probe<33792>("256x512 LDS 33K, 8 reads 1 write", s, buf, 256, 512, 1, 8);
  probe_mix<1>("one kernel, big kernarg", s, buf, 0);
  probe_mix<8>("eight kernels in rotation", s, buf, 0);
  probe_mix<8>("eight kernels in rotation", s, buf, 2);
  return 0;
}
