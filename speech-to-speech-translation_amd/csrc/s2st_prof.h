// Per-dispatch timing for bench.py's roofline leg (host side).
//
// With profiling on, a kernel is launched through hipExtLaunchKernelGGL with a start / stop event ATTACHED TO THAT
// DISPATCH: hipEventElapsedTime(start, stop) is then the dispatch's own begin -> end on the GPU clock -- the figure
// rocprofv3 --kernel-trace reports for it -- on whichever stream the kernel was launched, with no marker packets
// between kernels (a hipEventRecord pair around a launch adds its own dispatch latency to every sample and, on two
// overlapping streams, sums to more than the step).  With profiling off this is a plain hipLaunchKernelGGL.
// Records are aggregated per tag (one tag per kernel instantiation) by s2st_profile_report.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

bool s2st_prof_enabled();
void s2st_prof_push(const char* tag, hipEvent_t a, hipEvent_t b, double work, double work2, hipStream_t st);

// work: FLOPs (MFMA-bound kernels) or bytes (HBM-bound kernels) of this launch; work2: free second figure
template <class K, class... A>
inline void s2st_launch(const char* tag, double work, double work2, K kern, dim3 grid, dim3 block, unsigned lds,
                        hipStream_t st, A... args) {
  if (!s2st_prof_enabled()) {
    hipLaunchKernelGGL(kern, grid, block, lds, st, args...);
    return;
  }
  hipEvent_t a = nullptr, b = nullptr;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipExtLaunchKernelGGL(kern, grid, block, lds, st, a, b, 0, args...);
  s2st_prof_push(tag, a, b, work, work2, st);
}

// every other launch site: tag = the kernel expression as written, no work figure (timeline mode of the report)
#define S2ST_LAUNCH(kern, grid, block, lds, st, ...) s2st_launch(#kern, 0.0, 0.0, kern, grid, block, lds, st, __VA_ARGS__)
