// Shared helpers of libdd3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "dd3d_hip.h"

namespace dd3d {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DD3D_E_LAUNCH;
  }
  return DD3D_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of a kernel ON ONE DEVICE: a process that drives several GPUs must opt
// in on each of them.  `done` = one bit per device ordinal (static storage at the call site); a refusal is reported, not ignored.
inline bool lds_opt_in_needed(unsigned long long (&done)[4]) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 256) return true;
  if ((done[d >> 6] >> (d & 63)) & 1ull) return false;
  done[d >> 6] |= 1ull << (d & 63);
  return true;
}
inline int lds_opt_in(const void* kernel, size_t bytes, const char* what) {
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    set_error("%s: cannot opt in to %zu bytes of dynamic LDS: %s", what, bytes, hipGetErrorString(e));
    return DD3D_E_LAUNCH;
  }
  return DD3D_OK;
}

#define DD3D_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      dd3d::set_error(__VA_ARGS__);      \
      return DD3D_E_INVALID;             \
    }                                    \
  } while (0)

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace dd3d
