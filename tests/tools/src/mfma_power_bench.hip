// Dev microbenchmark: what does the matrix pipe of an MI355X sustain when the OPERAND DATA is realistic?
//
// The head-tower convolution runs 1.30x slower on real activations than on all-zero operands with the identical instruction stream
// (profiles/r02_tower_data_vs_zeros.txt).  This program takes everything else away: every wave keeps NA A-fragments and NB B-fragments
// in registers and issues v_mfma_f32_32x32x16_f16 back to back (NA x NB independent accumulator blocks, the tower kernel's 2 x 2 wave
// tile when NA = NB = 2 with two planes each), no LDS, no global memory in the loop.  Between MFMA groups the fragments are ROTATED
// among a small set of register-resident values so that consecutive instructions see different operand bits, as a K loop does.
// Operand fills:  zeros | one constant | random halves (uniform bit patterns of normal numbers) | "split": hi = half(x), lo = half(x - hi)
// of gaussian x (the f16x2 arithmetic's planes: lo terms are ~2^-11 of hi with random mantissas).
//   hipcc --offload-arch=gfx950 -O3 tests/tools/src/mfma_power_bench.hip -o build/bin/mfma_power_bench
//   build/bin/mfma_power_bench [waves_per_block=8] [blocks=256] [iters=4000]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NSET = 4;  // operand sets a wave rotates through (register resident)

// ORDER 0: product-major (the conv kernels' order: lo*hi for all four blocks, then hi*lo, then hi*hi; the A operand changes every 2 MFMAs,
//          the B operand every MFMA);  1: A-operand-major (A hi of row i meets B hi 0, B hi 1, B lo 0, B lo 1 back to back, then A lo of row i
//          meets B hi 0, B hi 1: one operand port holds still for 4 / 2 instructions);  2: B-operand-major (the mirror image)
template <int WAVES, int ORDER>
__global__ __launch_bounds__(64 * WAVES) void mfma_loop(const f16x8* __restrict__ ops, int iters, float* sink) {
  // per lane: NSET x (A hi0, A hi1, A lo0, A lo1, B hi0, B hi1, B lo0, B lo1)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a[NSET][4], b[NSET][4];
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[s][q] = ops[((long)t * NSET + s) * 8 + q];
      b[s][q] = ops[((long)t * NSET + s) * 8 + 4 + q];
    }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
      // one "16-k chunk" of the f16x2 tower loop: 3 products x (2 x 2) accumulator blocks = 12 MFMAs
      // planes: a[s][0..1] = hi of rows 0/1, a[s][2..3] = lo;  b likewise
      if constexpr (ORDER == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][i], b[s][2 + j], acc[i][j], 0, 0, 0);  // hi * lo
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);  // hi * hi
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][2 + i], b[s][j], acc[i][j], 0, 0, 0);  // lo * hi
        }
        continue;
      }
      if constexpr (ORDER == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][2 + i], b[s][j], acc[i][j], 0, 0, 0);  // lo * hi
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);  // hi * hi
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][i], b[s][2 + j], acc[i][j], 0, 0, 0);  // hi * lo
        }
        continue;
      }
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
  if (v == 123.456f) sink[0] = v;  // keep the loop
}

static unsigned short f2h(float x) {
  _Float16 h = (_Float16)x;
  unsigned short u;
  memcpy(&u, &h, 2);
  return u;
}
static float h2f(unsigned short u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}
static float gauss() {
  float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

template <int WAVES, int ORDER = 0>
static void run(const char* name, unsigned short* host, long n_half, unsigned short* dev, int blocks, int iters, float* sink) {
  hipMemcpy(dev, host, n_half * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f, worst = 0.f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<WAVES, ORDER>), dim3(blocks), dim3(64 * WAVES), 0, 0, (const f16x8*)dev, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) continue;  // warm-up / clock ramp
    best = fminf(best, ms), worst = fmaxf(worst, ms);
  }
  const double flops = 2.0 * 32 * 32 * 16 * 12.0 * NSET * iters * (double)blocks * WAVES;
  printf("%-44s %8.3f .. %8.3f ms  ->  %7.1f .. %7.1f TFLOP/s of executed f16 MFMA (%.0f%% .. %.0f%% of 2500)\n", name, best, worst, flops / worst / 1e9,
         flops / best / 1e9, flops / worst / 1e9 / 25.0, flops / best / 1e9 / 25.0);
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 8;
  const int blocks = argc > 2 ? atoi(argv[2]) : 256;
  const int iters = argc > 3 ? atoi(argv[3]) : 4000;
  const long threads = (long)blocks * 64 * waves;
  const long n_half = threads * NSET * 8 * 8;
  unsigned short* host = (unsigned short*)malloc(n_half * 2);
  unsigned short* dev;
  float* sink;
  hipMalloc(&dev, n_half * 2);
  hipMalloc(&sink, 4);
  auto go = [&](const char* name) {
    if (waves == 8) run<8>(name, host, n_half, dev, blocks, iters, sink);
    else if (waves == 4) run<4>(name, host, n_half, dev, blocks, iters, sink);
    else run<16>(name, host, n_half, dev, blocks, iters, sink);
  };
  printf("v_mfma_f32_32x32x16_f16, %d blocks x %d waves, %d iterations x %d MFMA per wave, operands in registers\n", blocks, waves, iters, 12 * NSET);
  for (long i = 0; i < n_half; ++i) host[i] = 0;
  go("zeros");
  for (long i = 0; i < n_half; ++i) host[i] = f2h(1.0f);
  go("constant 1.0");
  srand(1);
  for (long i = 0; i < n_half; ++i) host[i] = f2h(gauss());
  go("gaussian halves in every plane");
  // f16x2 planes: per (thread, set): fragments 0,1 / 4,5 = hi, 2,3 / 6,7 = lo of the same gaussian f32 values x 16 (activation scale)
  for (long t = 0; t < threads * NSET; ++t)
    for (int q = 0; q < 8; ++q)
      for (int e = 0; e < 8; ++e) {
        const int is_lo = (q & 2) != 0;
        unsigned short* p = host + (t * 8 + q) * 8 + e;
        if (!is_lo) {
          *p = f2h(gauss() * 4.f);
        }
      }
  for (long t = 0; t < threads * NSET; ++t)
    for (int q = 0; q < 8; ++q)
      for (int e = 0; e < 8; ++e)
        if (q & 2) {
          // lo term of an f32 value whose hi term is fragment q - 2: a random residual within half an ulp of hi
          const float hi = h2f(host[(t * 8 + (q - 2)) * 8 + e]);
          const float ulp = fabsf(hi) * 0.00048828125f;  // 2^-11
          host[(t * 8 + q) * 8 + e] = f2h(((rand() / (float)RAND_MAX) - 0.5f) * ulp);
        }
  go("f16x2 planes (hi = half(x), lo = residual)");
  // relu-like activations: half of the values exactly zero (hi and lo)
  for (long t = 0; t < threads * NSET; ++t)
    for (int q = 0; q < 4; ++q)  // A side only (activations); the filters stay dense
      for (int e = 0; e < 8; ++e)
        if (!(q & 2) && (rand() & 1)) host[(t * 8 + q) * 8 + e] = 0, host[(t * 8 + q + 2) * 8 + e] = 0;
  go("f16x2 planes, 50% of the activations zero");
  if (waves == 8) {  // instruction order: does holding one operand port still between consecutive MFMAs change the power-limited rate?
    run<8, 1>("  same data, A-operand-major order", host, n_half, dev, blocks, iters, sink);
    run<8, 2>("  same data, B-operand-major order", host, n_half, dev, blocks, iters, sink);
    run<8, 0>("  same data, product-major order (again)", host, n_half, dev, blocks, iters, sink);
  }
  return 0;
}
