// Shared helpers of libdd3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "build_flags.h"
#include "dd3d_hip.h"

namespace dd3d {

void set_error(const char* fmt, ...);

// Every translation unit reports the build-time knobs it was compiled with (build_flags.h); dd3d_build_flags() returns their union.
void register_build_flags(const char* file, const char* flags);
struct BuildFlagsNote {
  BuildFlagsNote(const char* file, const char* flags) { register_build_flags(file, flags); }
};
#define DD3D_NOTE_BUILD_FLAGS \
  namespace {                 \
  const dd3d::BuildFlagsNote dd3d_build_flags_note_(__FILE__, DD3D_BUILD_FLAGS); \
  }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DD3D_E_LAUNCH;
  }
  return DD3D_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of a kernel ON ONE DEVICE: a process that drives several GPUs must opt
// in on each of them.  `done` = one bit per device ordinal (static storage at the call site).  lds_opt_in_needed() only ASKS; the call
// site marks the device with lds_opt_in_done() after EVERY opt-in of the site has succeeded, so a refusal is retried by the next launch
// instead of leaving later launches to fail with an opaque launch error (round-3 advisor).  The bit set is updated atomically: two host
// threads may drive different GPUs.
inline int lds_current_device() {
  int d = 0;
  return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 256) ? d : -1;
}
inline bool lds_opt_in_needed(unsigned long long (&done)[4]) {
  const int d = lds_current_device();
  if (d < 0) return true;
  return !((__atomic_load_n(&done[d >> 6], __ATOMIC_ACQUIRE) >> (d & 63)) & 1ull);
}
inline void lds_opt_in_done(unsigned long long (&done)[4]) {
  const int d = lds_current_device();
  if (d >= 0) __atomic_fetch_or(&done[d >> 6], 1ull << (d & 63), __ATOMIC_RELEASE);
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

// inv(K) of a row-major 3x3 by the adjugate, in the operation order dd3d_invert_intrinsics has always used (core.py:93 ImageList.intrinsics.inverse())
__device__ inline void invert3x3(const float* m, float* o) {
  const float a = m[0], bb = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const float A = e * i - f * h, Bc = -(d * i - f * g), Cc = d * h - e * g;
  const float det = a * A + bb * Bc + c * Cc;
  const float r = 1.0f / det;
  o[0] = A * r;
  o[1] = -(bb * i - c * h) * r;
  o[2] = (bb * f - c * e) * r;
  o[3] = Bc * r;
  o[4] = (a * i - c * g) * r;
  o[5] = -(a * f - c * d) * r;
  o[6] = Cc * r;
  o[7] = -(a * h - bb * g) * r;
  o[8] = (a * e - bb * d) * r;
}

}  // namespace dd3d
