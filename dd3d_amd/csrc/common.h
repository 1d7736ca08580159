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

#define DD3D_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      dd3d::set_error(__VA_ARGS__);      \
      return DD3D_E_INVALID;             \
    }                                    \
  } while (0)

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace dd3d
