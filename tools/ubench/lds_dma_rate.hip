// Per-CU fill rate of LDS-DMA (global_load_lds_dwordx4) vs register loads (global_load_dwordx4) from an
// L2-resident buffer, by waves per CU.  hipcc --offload-arch=gfx950 -O3 lds_dma_rate.hip -o lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>  // 0: LDS-DMA, 1: VGPR loads (+ds_write), 2: VGPR loads only
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ src, long n16, int iters, uint4* sink, long long* cyc, int win16, int share) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // each CU streams over its own 256 KB window (L2 resident after the first pass)
  // window of win16 16-byte units per CU; share: the CUs of an XCD (blockIdx % 8) read the same window
  const uint4* base = src + ((long)(share ? (blockIdx.x & 7) : blockIdx.x) * win16) % n16;
  uint4 acc = make_uint4(0, 0, 0, 0);
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll 8
    for (int j = 0; j < 8; ++j) {
      const long off = ((long)(it * 8 + j) * nw + wave) * 64 % win16 + lane;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(smem + (wave * 8 + j) * 1024), 16, 0, 0);
      } else {
        uint4 v = base[off];
        if (MODE == 1) *reinterpret_cast<uint4*>(smem + (wave * 8 + j) * 1024 + lane * 16) = v;
        else { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
      }
    }
    if (MODE == 0) __builtin_amdgcn_s_waitcnt(0x0f70 | 8);  // at most 8 pieces outstanding per wave
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (MODE != 0 && acc.x == 0x12345678u) sink[0] = acc;
  if (MODE == 1 && smem[threadIdx.x] == 77 && iters < 0) sink[1] = acc;
}

int main() {
  const long n16 = 1 << 22;  // 64 MB
  uint4* src; uint4* sink; long long* cyc;
  hipMalloc(&src, n16 * 16); hipMemset(src, 1, n16 * 16);
  hipMalloc(&sink, 64); hipMalloc(&cyc, 8 * 1024);
  const int iters = 2000;
  for (int share : {0, 1}) for (int win16 : {4096, 16384}) for (int ncu : {256, 128, 64}) for (int nw : {4, 8}) for (int mode : {0, 2}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto launch = [&]() {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(ncu), dim3(64 * nw), 65536, 0, src, n16, iters, sink, cyc, win16, share);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(ncu), dim3(64 * nw), 65536, 0, src, n16, iters, sink, cyc, win16, share);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(ncu), dim3(64 * nw), 65536, 0, src, n16, iters, sink, cyc, win16, share);
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(ncu); hipMemcpy(h.data(), cyc, ncu * 8, hipMemcpyDeviceToHost);
    double mc = 0; for (auto c : h) mc += c; mc /= ncu;
    double bytes_cu = (double)iters * 8 * nw * 1024;
    printf("share %d window %4d KB CUs %3d waves/CU %d mode %s: %7.1f us  %6.1f GB/s/CU  %5.1f B/clk/CU (memtime)  aggregate %6.2f TB/s\n", share, win16 * 16 / 1024, ncu, nw,
           mode == 0 ? "lds-dma      " : mode == 1 ? "vgpr+ds_write" : "vgpr only    ", ms * 1e3, bytes_cu / (ms * 1e-3) / 1e9,
           bytes_cu / mc, bytes_cu * ncu / (ms * 1e-3) / 1e12);
  }
  return 0;
}
