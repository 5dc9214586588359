// Kernel-boundary cost on one in-order stream: N dependent launches of a kernel that runs ~D us, issued (a) one by one,
// (b) as a captured hipGraph.  per-boundary overhead = (wall / N) - D.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_gap.hip -o gpurun_out/launch_gap ; run: ./launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

struct Big { long a[48]; };  // ~384 B of kernel arguments, like a GEMM descriptor

__global__ void spin_kernel(float* p, long cycles, Big b) {
  const long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += (float)b.a[0];
}

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  float* d;
  hipMalloc(&d, 4096);
  Big b{};
  const int N = 2000;
  for (int grid : {1, 256, 1024}) {
    for (long cyc : {0L, 20000L}) {  // clock64 ticks at 100 MHz on gfx9: 20000 -> ~? us, reported below from a long run
      // calibrate the kernel's own duration with one long-running launch sequence of 1
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, st, d, cyc, b);
      hipStreamSynchronize(st);
      hipEventRecord(e0, st);
      hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, st, d, cyc, b);
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float one = 0;
      hipEventElapsedTime(&one, e0, e1);
      // (a) plain launches
      double t0 = now();
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, st, d, cyc, b);
      double t_issue = now() - t0;
      hipStreamSynchronize(st);
      double t_plain = now() - t0;
      // (b) graph of the same N launches
      hipGraph_t g;
      hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, st, d, cyc, b);
      hipStreamEndCapture(st, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, st);
      hipStreamSynchronize(st);
      t0 = now();
      hipGraphLaunch(ge, st);
      hipStreamSynchronize(st);
      double t_graph = now() - t0;
      printf("grid %5d spin %6ld: single launch+events %.1f us | plain: %.2f us per launch (host issue %.2f) | graph: %.2f us per node\n",
             grid, cyc, one * 1e3, t_plain / N * 1e6, t_issue / N * 1e6, t_graph / N * 1e6);
      hipGraphExecDestroy(ge);
      hipGraphDestroy(g);
    }
  }
  return 0;
}
