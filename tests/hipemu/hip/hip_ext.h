// emulator counterpart of <hip/hip_ext.h>: the event-carrying launch is a plain launch (events carry no time here)
#pragma once
#include <hip/hip_runtime.h>
template <class K, class... A>
static inline void hipExtLaunchKernelGGL(K kernel, dim3 grid, dim3 block, unsigned lds, hipStream_t st, hipEvent_t,
                                         hipEvent_t, unsigned, A... a) {
  hipLaunchKernelGGL(kernel, grid, block, lds, st, a...);
}
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned, const unsigned*) { *s = nullptr; return hipErrorInvalidValue; }
