// v_permlane16_swap_b32 lane map on gfx950 (used by the GEMM epilogue's 16-byte bf16 stores): a = lane, b = 100 + lane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  p[threadIdx.x] = r[0];
  p[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  (void)hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("vdst':"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]); printf("\nsrc0':"); for (int i = 0; i < 64; ++i) printf(" %u", h[64 + i]); printf("\n");
  return 0;
}
