// Which CUs does a CU-masked stream run on?  hipExtStreamCreateWithCUMask(stream, words, mask): the mapping of mask bits to
// (XCD, shader engine, CU) on an 8-XCD part is not documented in the guides this project has; this probe launches a
// grid of spinning workgroups on streams with various masks and reports the set of (xcc, se, sh, cu) each one touched.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/ubench/cu_mask_probe.hip -o /tmp/cmp && /tmp/cmp
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned* out, long cycles) {
  const long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}

static void run(const char* name, const std::vector<unsigned>& mask) {
  hipStream_t st;
  if (mask.empty()) {
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { printf("%s: stream failed\n", name); return; }
  } else if (hipExtStreamCreateWithCUMask(&st, (unsigned)mask.size(), mask.data()) != hipSuccess) {
    printf("%s: hipExtStreamCreateWithCUMask failed\n", name);
    return;
  }
  const int grid = 4096;
  unsigned* d;
  (void)hipMalloc(&d, sizeof(unsigned) * 2 * grid);
  (void)hipMemsetAsync(d, 0xff, sizeof(unsigned) * 2 * grid, st);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, st);
  hipLaunchKernelGGL(where_kernel, dim3(grid), dim3(256), 64 * 1024, st, d, 20000L);  // 64 KB of LDS: <= 2 workgroups per CU
  (void)hipEventRecord(e1, st);
  (void)hipStreamSynchronize(st);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned> h(2 * grid);
  (void)hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * grid, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per_xcc;
  for (int i = 0; i < grid; ++i) {
    const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  int total = 0;
  printf("%-28s %7.3f ms  CUs per XCC:", name, ms);
  for (auto& kv : per_xcc) { printf(" %u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  total %d\n", total);
  if (per_xcc.size() && total <= 80) {
    for (auto& kv : per_xcc) {
      printf("    xcc %u:", kv.first);
      for (unsigned v : kv.second) printf(" se%u.sh%u.cu%u", v >> 8, (v >> 4) & 1, v & 0xf);
      printf("\n");
    }
  }
  (void)hipFree(d);
  (void)hipStreamDestroy(st);
}

int main() {
  run("no mask", {});
  run("words=8 all ones", std::vector<unsigned>(8, 0xffffffffu));
  run("bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
  run("bits 0..63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0});
  run("bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0});
  run("bits 0,8,16,24 (stride 8)", {0x01010101u, 0, 0, 0, 0, 0, 0, 0});
  run("every word 0xff (8 of 32)", std::vector<unsigned>(8, 0xffu));
  run("every word 0xffff (16 of 32)", std::vector<unsigned>(8, 0xffffu));
  run("every word 0x00ffffff (24/32)", std::vector<unsigned>(8, 0x00ffffffu));
  run("every word 0x55555555", std::vector<unsigned>(8, 0x55555555u));
  run("words 0-3 ones, 4-7 zero", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
  run("one word only (size 1) 0xffff", {0xffffu});
  return 0;
}
