// Probe (round 6): what one LDS-DMA instruction (global_load_lds_dwordx4, 1 KiB per wave) costs a CU as a function of the CACHE LINES it
// touches.  The split-plane layout [chunk][pixel][plane][32 halves] puts the two 64-byte plane rows of a pixel into one 128-byte line, and a
// piece of the convolution kernels is 16 pixels x ONE plane = 16 half lines; 8 pixels x both planes would be 8 full lines.
//   hipcc --offload-arch=gfx950 -O3 -o build/dma_line_probe tests/tools/dma_line_probe.hip ;  ./build/dma_line_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned char __attribute__((address_space(3))) * ldsbp;
typedef const unsigned char __attribute__((address_space(1))) * gcbp;

template <int PATTERN, int NW>
__global__ __launch_bounds__(64 * NW) void probe(const unsigned char* src, int region, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char* base = src + (size_t)blockIdx.x * region;
  int off;
  if (PATTERN == 0) off = (lane >> 2) * 128 + (lane & 3) * 16;                                   // 16 rows x one plane: 16 half lines (2 KiB span)
  else if (PATTERN == 1) off = (lane >> 3) * 128 + (lane & 7) * 16;                              // 8 full lines = 1 KiB contiguous
  else if (PATTERN == 2) off = (lane >> 2) * 128 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);      // the kernels' swizzled slots (same lines as 0)
  else if (PATTERN == 3) off = (lane >> 2) * 256 + (lane & 3) * 16;                              // 16 quarter-used lines pairs apart (4 KiB span)
  else off = (lane >> 2) * 128 + (lane & 3) * 16;                                                // 4: as 0, but pieces 2 j / 2 j + 1 = the two planes of the same 16 rows
  const int span = (PATTERN == 1 || PATTERN == 4) ? 1024 : (PATTERN == 3 ? 4096 : 2048);
  const int per_iter = 8;
  int pos = wave * span * per_iter;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < per_iter; ++q) {
      const int qo = PATTERN == 4 ? (q >> 1) * 2048 + (q & 1) * 64 : q * span;
      __builtin_amdgcn_global_load_lds((gcbp)(base + pos + qo + off), (ldsbp)(lds + (wave * per_iter + q) * 1024), 16, 0, 0);
    }
    pos += NW * span * per_iter;
    if (pos + span * per_iter > region) pos = wave * span * per_iter;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && threadIdx.x == 0 && lds[5] == 77) sink[blockIdx.x] = 1.f;
}

template <int PATTERN, int NW>
static void run(const unsigned char* src, int region, int blocks, float* sink, const char* what) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t lds = (size_t)NW * 8 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<PATTERN, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<PATTERN, NW>), dim3(blocks), dim3(64 * NW), lds, 0, src, region, iters, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 1) {
      const double pieces_per_cu = (double)iters * 8 * NW * ((blocks + 255) / 256);
      printf("%-44s waves/CU %d  blocks %4d  %8.1f us  %6.1f ns per piece per CU (~%5.1f cycles at 2.4 GHz)  %6.2f TB/s\n", what, NW, blocks, ms * 1e3,
             ms * 1e6 / pieces_per_cu, ms * 1e6 / pieces_per_cu * 2.4, (double)iters * 8 * NW * blocks * 1024 / (ms * 1e-3) / 1e12);
    }
  }
}

int main() {
  const int blocks = 256;
  for (int region : {32 * 1024, 256 * 1024}) {
    unsigned char* src = nullptr;
    float* sink = nullptr;
    CK(hipMalloc(&src, (size_t)blocks * region + 8192));
    CK(hipMalloc(&sink, blocks * sizeof(float)));
    CK(hipMemset(src, 1, (size_t)blocks * region + 8192));
    printf("== %d KiB of source per block (%d MiB in all)\n", region / 1024, (int)((size_t)blocks * region >> 20));
    run<0, 4>(src, region, blocks, sink, "16 rows x one plane (16 half lines)");
    run<1, 4>(src, region, blocks, sink, "8 rows x both planes (8 full lines)");
    run<2, 4>(src, region, blocks, sink, "16 half lines, swizzled slots (shipped)");
    run<3, 4>(src, region, blocks, sink, "16 quarter lines, 256-byte pitch");
    run<4, 4>(src, region, blocks, sink, "16 half lines, both planes in turn (shipped)");
    run<4, 8>(src, region, blocks, sink, "16 half lines, both planes in turn (shipped)");
    run<0, 8>(src, region, blocks, sink, "16 rows x one plane (16 half lines)");
    run<1, 8>(src, region, blocks, sink, "8 rows x both planes (8 full lines)");
    run<0, 1>(src, region, blocks, sink, "16 rows x one plane (16 half lines)");
    run<1, 1>(src, region, blocks, sink, "8 rows x both planes (8 full lines)");
    CK(hipFree(src));
    CK(hipFree(sink));
  }
  return 0;
}
