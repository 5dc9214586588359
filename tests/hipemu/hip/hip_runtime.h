// Functional wave64 emulator of the small HIP subset the s2st kernels use.
//
// TEST INFRASTRUCTURE ONLY: lets tests/ compile the product's .hip sources for the host
// and run them on tiny shapes without a GPU (logic / indexing / schedule debugging).
// It is never loaded by the product path: the product library is built by hipcc for
// gfx950 and fails loudly without a device.  Each GPU thread is a ucontext fiber; a
// workgroup's fibers run on one OS thread, workgroups are spread over OS threads.
// Wave collectives (__shfl*, MFMA) rendezvous the 64 fibers of a wave and apply the
// lane maps documented for gfx950.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIP_KERNEL_NAME(...) __VA_ARGS__

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
typedef void* hipStream_t;
typedef void* hipEvent_t;

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
extern thread_local emu_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }

// ---- scheduler entry points (emu_runtime.cpp) -------------------------------------------
void emu_syncthreads();
void emu_wave_exchange(const void* in, void* all_out, int bytes);  // gathers 64 lanes' data
int emu_lane();
void emu_launch_impl(void (*tramp)(void*), void* args, dim3 grid, dim3 block);

static inline void __syncthreads() { emu_syncthreads(); }

template <class T> static inline T emu_shfl_idx(T v, int src) {
  T all[64];
  emu_wave_exchange(&v, all, sizeof(T));
  return all[src & 63];
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  return emu_shfl_idx(v, emu_lane() ^ mask);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = emu_lane();
  int src = l + (int)d;
  if ((l & (width - 1)) + (int)d >= width) src = l;
  return emu_shfl_idx(v, src);
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
  int l = emu_lane();
  return emu_shfl_idx(v, (l & ~(width - 1)) | (src & (width - 1)));
}
static inline unsigned long long __ballot(int pred) {
  int all[64];
  emu_wave_exchange(&pred, all, sizeof(int));
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) if (all[i]) m |= 1ull << i;
  return m;
}
static inline int __all(int p) { return __ballot(p) == ~0ull; }
static inline int __any(int p) { return __ballot(p) != 0ull; }

// ---- atomics (workgroups run on different OS threads) -------------------------------------
static inline float atomicAdd(float* p, float v) {
  unsigned* up = (unsigned*)p;
  unsigned old = __atomic_load_n(up, __ATOMIC_RELAXED), nw;
  float f;
  do { memcpy(&f, &old, 4); f += v; memcpy(&nw, &f, 4);
  } while (!__atomic_compare_exchange_n(up, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  memcpy(&f, &old, 4);
  return f;
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline double atomicAdd(double* p, double v) {
  unsigned long long* up = (unsigned long long*)p;
  unsigned long long old = __atomic_load_n(up, __ATOMIC_RELAXED), nw;
  double f;
  do { memcpy(&f, &old, 8); f += v; memcpy(&nw, &f, 8);
  } while (!__atomic_compare_exchange_n(up, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  memcpy(&f, &old, 8);
  return f;
}

// ---- math ----------------------------------------------------------------------------------
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
#define __expf(x) expf(x)
#define __frcp_rn(x) (1.0f / (x))
#define __logf(x) logf(x)

// ---- MFMA: v_mfma_f32_16x16x32_bf16 ---------------------------------------------------------
// A: lane l holds A[row l&15][k = 8*(l>>4) + j], B: B[k = 8*(l>>4)+j][col l&15], j = 0..7
// C/D: col = lane&15, row = (lane>>4)*4 + reg   (cdna_hip_programming.md section 3)
typedef short emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
static inline float emu_bf16_to_f32(short s) {
  unsigned u = ((unsigned)(unsigned short)s) << 16;
  float f; memcpy(&f, &u, 4); return f;
}
static inline emu_f32x4 emu_mfma_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c, int, int, int) {
  struct Pack { short a[8]; short b[8]; } mine, all[64];
  for (int j = 0; j < 8; ++j) { mine.a[j] = a[j]; mine.b[j] = b[j]; }
  emu_wave_exchange(&mine, all, sizeof(Pack));
  int l = emu_lane();
  int col = l & 15;
  emu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = d[r];
    for (int k = 0; k < 32; ++k) {
      float av = emu_bf16_to_f32(all[row + 16 * (k >> 3)].a[k & 7]);
      float bv = emu_bf16_to_f32(all[col + 16 * (k >> 3)].b[k & 7]);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emu_mfma_16x16x32_bf16
// ds_read_b64_tr_b16 (see cdna_hip_programming.md T10)
typedef short emu_s16x4 __attribute__((ext_vector_type(4)));
static inline emu_s16x4 emu_ds_read_tr16(const void* p) {
  const unsigned char* mine = (const unsigned char*)p;
  const unsigned char* all[64];
  emu_wave_exchange(&mine, all, sizeof(mine));
  int l = emu_lane(), g = l & ~15, i = l & 15;
  emu_s16x4 r;
  for (int q = 0; q < 4; ++q) {
    const unsigned char* a = all[g + 4 * q + (i >> 2)] + 2 * (i & 3);
    short v; memcpy(&v, a, 2);
    r[q] = v;
  }
  return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu_ds_read_tr16((const void*)(p))
#define __builtin_amdgcn_readfirstlane(x) (emu_shfl_idx((x), 0))
// dynamic LDS: one 160 KiB buffer per workgroup thread (emu_runtime.cpp)
extern thread_local unsigned char emu_dyn_smem[160 * 1024];
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(emu_dyn_smem);
// global_load_lds_dwordx4: LDS destination = wave-uniform base (lane 0's pointer) + lane * size;
// the global source is per lane.  Completion is immediate here (waitcnt / s_barrier are no-op / sync).
static inline void emu_global_load_lds(const void* g, void* l, unsigned size, int offset, unsigned) {
  void* base = emu_shfl_idx(l, 0);
  memcpy((unsigned char*)base + offset + (size_t)emu_lane() * size, (const unsigned char*)g + offset, size);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu_global_load_lds((const void*)(g), (void*)(l), size, off, aux)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
static inline unsigned long long wall_clock64() { static thread_local unsigned long long t = 0; return t += 1ull << 40; }  // (every call is "much later": pacing loops end at once on the emulator)
#define __builtin_amdgcn_s_sleep(x) emu_sleep_us(20)  // (a workgroup waiting for another one: emulated blocks are ~1000x slower)
#include <unistd.h>
static inline void emu_sleep_us(int us) { usleep(us); }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(order)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __builtin_amdgcn_s_barrier() emu_syncthreads()
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)

// ---- launch -------------------------------------------------------------------------------
#include <tuple>
#include <utility>
template <class F, class Tup, size_t... I>
static void emu_call(F f, Tup& t, std::index_sequence<I...>) { f(std::get<I>(t)...); }

template <class... KA, class... A>
static inline void hipLaunchKernelGGL(void (*kernel)(KA...), dim3 grid, dim3 block, size_t, hipStream_t, A... a) {
  struct Ctx { void (*k)(KA...); std::tuple<KA...> args; } ctx{kernel, std::tuple<KA...>(KA(a)...)};
  emu_launch_impl(
      [](void* p) { Ctx* c = (Ctx*)p; emu_call(c->k, c->args, std::index_sequence_for<KA...>{}); },
      &ctx, grid, block);
}

static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
// (every launch of the emulator is synchronous: a second stream is only a label -- enough for the engine's two-chain
//  schedule, S2ST_CHAINS=2, whose partition of rows / seeds / scratch is what the emulator can check; the priority form,
//  which the engine's weight-gradient stream uses, stays unavailable)
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static char tag; *s = (hipStream_t)&tag; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipErrorInvalidValue; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 8; return hipSuccess; }  // 8 'CUs': persistent kernels walk several tiles, stream-K has one workgroup per 'XCD'
using std::min;
using std::max;
