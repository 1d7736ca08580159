// BENCH / TEST TOOLING -- not part of libdd3d_hip.so (round-3 verdict: a diagnostic does not belong in the product ABI).  Built by
// __graft_entry__.build() into tests/tools/lib/libdd3d_tools.so; bench.py loads it when present.
// Diagnostic entry point: what does the matrix pipe of THIS chip sustain today on realistic operand bits?
//
// v_mfma_f32_32x32x16_f16 on register-resident operands, nothing else in the loop: every wave holds NSET operand sets of the tower kernel's
// wave tile (A rows 0 / 1 and B columns 0 / 1, each as hi and lo half planes) and issues the three products of the two-half-term arithmetic
// for the 2 x 2 accumulator blocks, set after set.  The rate depends on the DATA (zeros: 0.98 of the nominal 2.5 PFLOP/s; the planes of
// gaussian activations: 0.58-0.71 depending on the chip and its thermal state, profiles/r03_mfma_power_bench.txt, r03l_*): the chip's power
// management caps it.  bench.py times this launch next to the convolution it reports, so that `roofline` can name the ceiling that held on
// the same box in the same minute.  Stand-alone form with more operand fills: tests/tools/src/mfma_power_bench.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dd3d_tools {

typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));
constexpr int PROBE_NSET = 4;

__global__ __launch_bounds__(512) void mfma_probe_kernel(const pf16x8* __restrict__ ops, int iters, float* sink) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  pf16x8 a[PROBE_NSET][4], b[PROBE_NSET][4];  // [set][hi0, hi1, lo0, lo1]
#pragma unroll
  for (int s = 0; s < PROBE_NSET; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[s][q] = ops[(t * PROBE_NSET + s) * 8 + q];
      b[s][q] = ops[(t * PROBE_NSET + s) * 8 + 4 + q];
    }
  pf32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < PROBE_NSET; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][2 + i], b[s][j], acc[i][j], 0, 0, 0);  // lo * hi
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][i], b[s][2 + j], acc[i][j], 0, 0, 0);  // hi * lo
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);  // hi * hi
    }
  }
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) v += acc[i][j][r];
  if (v == 123.456f) sink[0] = v;  // keeps the loop alive
}

}  // namespace dd3d_tools

// `iters` x 48 v_mfma_f32_32x32x16_f16 per wave on register-resident operands (blocks x 8 waves; the two-half-term products of a 2 x 2 wave tile),
// no memory traffic in the loop.  ops: halves [blocks * 512 threads][4 sets][8 fragments: A hi0, hi1, lo0, lo1, B hi0, hi1, lo0, lo1][8]; sink: one
// device float.  FLOP per launch = 2 * 32 * 32 * 16 * 48 * iters * blocks * 8.  Returns 0, -1 (bad arguments) or -2 (launch error).
extern "C" int dd3d_tools_mfma_probe(const void* ops, int32_t blocks, int32_t iters, float* sink, void* stream) {
  using namespace dd3d_tools;
  if (!ops || !sink || blocks <= 0 || iters <= 0) return -1;
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const pf16x8*>(ops), iters, sink);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
