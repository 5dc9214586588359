// Cost of a kernel boundary on one in-order stream as a function of the kernel-argument size: N dependent launches of
// a kernel that does nothing but (optionally) spin (one thread reads one argument word, writes one word), wall time / N.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/ubench/arg_size_gap.hip -o /tmp/asg && /tmp/asg
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

template <int WORDS>
struct Args { long a[WORDS]; };

template <int WORDS>
__global__ void tiny_kernel(float* out, long cycles, Args<WORDS> b) {
  const long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)b.a[WORDS - 1];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int WORDS>
void run(hipStream_t st, float* d, int grid, long cycles) {
  Args<WORDS> b{};
  const int N = 4000;
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(tiny_kernel<WORDS>, dim3(grid), dim3(256), 0, st, d, cycles, b);
  (void)hipStreamSynchronize(st);
  double best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    const double t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny_kernel<WORDS>, dim3(grid), dim3(256), 0, st, d, cycles, b);
    (void)hipStreamSynchronize(st);
    const double us = (now() - t0) / N * 1e6;
    if (us < best) best = us;
  }
  printf("arguments %5zu B, grid %4d, spin %5ld clock ticks: %.2f us per launch\n", sizeof(Args<WORDS>) + 16, grid, cycles, best);
}

int main() {
  hipStream_t st;
  (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  float* d;
  (void)hipMalloc(&d, 1 << 20);
  // spin 0: the host's enqueue rate shows (the queue runs dry); spin ~8 us (clock64 ticks at 100 MHz): the device side --
  // wall / N minus the spin is what a dependent launch costs beyond its own work
  for (long cycles : {0l, 800l})
    for (int grid : {256, 2048}) {
      run<1>(st, d, grid, cycles);
      run<8>(st, d, grid, cycles);
      run<39>(st, d, grid, cycles);    // ~ GemmArgs (312 B)
      run<128>(st, d, grid, cycles);
      run<323>(st, d, grid, cycles);   // ~ GemmGroup (2584 B)
    }
  return 0;
}
