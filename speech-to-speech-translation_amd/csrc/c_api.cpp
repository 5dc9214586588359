// extern "C" surface of libs2st_hip.so (see include/s2st_hip.h).
#include "s2st_ops.h"

extern "C" {

int s2st_version(void) { return 100; }

int s2st_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int s2st_gemm_f32(const s2st_gemm_args* a, void* stream) {
  if (!a) return S2ST_ERR_ARG;
  return s2st_gemm(*a, (hipStream_t)stream);
}

}  // extern "C"
