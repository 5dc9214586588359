// Kernel-boundary cost on one in-order stream: N dependent launches of a kernel that spins ~8 us and then writes
// `mb` MB (every workgroup its own slice), issued one by one and as a captured hipGraph.
// boundary = (wall / N) - (the kernel's own begin -> end, from events attached to single dispatches).
// build + run: hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_gap.hip -o /tmp/lg && /tmp/lg
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

struct Big { long a[48]; };  // ~384 B of kernel arguments, like a GEMM descriptor

__global__ void work_kernel(float4* out, long n4, long cycles, Big b) {
  const long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  const float v = (float)b.a[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    out[i] = make_float4(v, v, v, v);
}

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t st;
  (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  float4* d;
  (void)hipMalloc(&d, 256l << 20);
  Big b{};
  const int N = 1000;
  const long cyc = 20000;
  for (int mb : {0, 4, 16, 64}) {
    const long n4 = (long)mb * (1 << 20) / 16;
    const int grid = 1024;
    auto launch = [&]() { hipLaunchKernelGGL(work_kernel, dim3(grid), dim3(256), 0, st, d, n4, cyc, b); };
    for (int i = 0; i < 5; ++i) launch();
    (void)hipStreamSynchronize(st);
    // the kernel's own duration: events attached to the dispatch, isolated launches
    double own = 0;
    for (int i = 0; i < 20; ++i) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      hipExtLaunchKernelGGL(work_kernel, dim3(grid), dim3(256), 0, st, e0, e1, 0, d, n4, cyc, b);
      (void)hipStreamSynchronize(st);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      own += ms * 1e3 / 20;
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    double t0 = now();
    for (int i = 0; i < N; ++i) launch();
    (void)hipStreamSynchronize(st);
    const double t_plain = (now() - t0) / N * 1e6;
    hipGraph_t g;
    hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) launch();
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    t0 = now();
    (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    const double t_graph = (now() - t0) / N * 1e6;
    printf("writes %3d MB per kernel: own duration %6.2f us | chained launches %6.2f us each (boundary %5.2f) | graph nodes %6.2f us each (boundary %5.2f)\n",
           mb, own, t_plain, t_plain - own, t_graph, t_graph - own);
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
  }
  return 0;
}
