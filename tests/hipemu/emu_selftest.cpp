#include "hip/hip_runtime.h"
#include <vector>
__global__ void k_reduce(const float* x, float* out, int n) {
  __shared__ float part[4];
  int t = threadIdx.x;
  float v = 0;
  for (int i = blockIdx.x * blockDim.x + t; i < n; i += gridDim.x * blockDim.x) v += x[i];
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if ((t & 63) == 0) part[t >> 6] = v;
  __syncthreads();
  if (t == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(const float* A, const float* B, float* C) {  // A[16][32], B[32][16]
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    float av = A[(l & 15) * 32 + 8 * (l >> 4) + j], bv = B[(8 * (l >> 4) + j) * 16 + (l & 15)];
    unsigned ua, ub; memcpy(&ua, &av, 4); memcpy(&ub, &bv, 4);
    a[j] = (short)(ua >> 16); b[j] = (short)(ub >> 16);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
int main() {
  int n = 100000;
  std::vector<float> x(n, 0.5f);
  float out = 0;
  hipLaunchKernelGGL(k_reduce, dim3(7), dim3(256), 0, 0, (const float*)x.data(), &out, n);
  printf("sum=%f (expect %f)\n", out, 0.5 * n);
  std::vector<float> A(16 * 32), B(32 * 16), C(256), R(256, 0);
  for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7) % 5 - 2); B[i] = (float)((i * 3) % 7 - 3); }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, (const float*)A.data(), (const float*)B.data(), C.data());
  double err = 0; for (int i = 0; i < 256; ++i) err += fabs(C[i] - R[i]);
  printf("mfma err=%g\n", err);
  return (fabs(out - 0.5 * n) < 1 && err == 0) ? 0 : 1;
}
