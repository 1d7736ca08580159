// Dev microbenchmark: how fast can a CU fill its LDS with global_load_lds_dwordx4 from L2-resident data, as a function of the
// per-lane source pattern of one 1-KiB wave instruction?  (Which layout should the split-plane activations have?)
//   hipcc --offload-arch=gfx950 -O3 tests/tools/src/lds_dma_bench.hip -o build/bin/lds_dma_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned char __attribute__((address_space(3))) * ldsbp;

// pattern: lane l of a DMA instruction reads 16 B at  row(l) * row_stride + (l % lanes_per_row) * 16,  row(l) = l / lanes_per_row
template <int INFLIGHT>
__global__ __launch_bounds__(512) void dma_kernel(const unsigned char* __restrict__ src, long ws_per_block, int lanes_per_row, int row_stride, int iters,
                                                  int active_waves, unsigned* sink, int halves) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= active_waves) return;
  const unsigned char* base = src + (long)blockIdx.x * ws_per_block;
  const int rows_per_instr = 64 / lanes_per_row;
  const long instr_span = (long)rows_per_instr * row_stride;  // bytes of source address space one instruction walks
  const long lane_off = (long)(lane / lanes_per_row) * row_stride + (lane % lanes_per_row) * 16;
  const long room = ws_per_block - instr_span;  // start positions wrap inside the block's working set
  long pos = ((long)wave * instr_span) % room;
  const long step = ((long)active_waves * instr_span) % room;
  unsigned char* dst = lds + wave * (INFLIGHT * 1024);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < INFLIGHT; ++d) {
      // halves: instruction pairs (d, d+1) read the lower / upper 64 B of the SAME lines (what plane 0 / plane 1 pieces of a two-term
      // activation tile do); otherwise every instruction walks on
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + pos + lane_off + ((halves && (d & 1)) ? 64 : 0)),
                                       (ldsbp)(dst + d * 1024), 16, 0, 0);
      if (!halves || (d & 1)) {
        pos += step;
        if (pos >= room) pos -= room;
      }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = lds[blockIdx.x & 1023];
}

int main(int argc, char** argv) {
  const int blocks = 256, iters = 2000;
  const long ws = (argc > 1 ? atol(argv[1]) : 96) * 1024;  // per-block working set: 24 MiB in all -> L2 resident (4 MiB per XCD: 32 blocks x 96 KiB = 3 MiB)
  unsigned char* src;
  unsigned* sink;
  hipMalloc(&src, blocks * ws + (4 << 20));
  hipMemset(src, 1, blocks * ws + (4 << 20));
  hipMalloc(&sink, blocks * 4);
  hipFuncSetAttribute((const void*)dma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipFuncSetAttribute((const void*)dma_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  struct Pat { const char* name; int lanes_per_row, row_stride, halves; } pats[] = {
      {"1KiB contiguous (64 lanes x 16 B)", 64, 1024, 0},
      {"8 rows x 128 B, rows contiguous (whole lines)", 8, 128, 0},
      {"8 rows x 128 B, row stride 9216 B (filter rows, [n][k-tile] layout)", 8, 9216, 0},
      {"16 rows x 64 B, row stride 128 B, walking on", 4, 128, 0},
      {"16 rows x 64 B, row stride 128 B, lower then upper halves of the same lines", 4, 128, 1},
      {"16 rows x 64 B, row stride 9216 B, lower then upper halves (filter rows today)", 4, 9216, 1},
      {"16 rows x 64 B, row stride 192 B, walking on", 4, 192, 0},
  };
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int inflight : {8})
    for (int waves : {8})
      for (auto& p : pats) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          if (inflight == 8) hipLaunchKernelGGL(dma_kernel<8>, dim3(blocks), dim3(512), 64 * 1024, 0, src, ws, p.lanes_per_row, p.row_stride, iters, waves, sink, p.halves);
          else hipLaunchKernelGGL(dma_kernel<16>, dim3(blocks), dim3(512), 128 * 1024, 0, src, ws, p.lanes_per_row, p.row_stride, iters, waves, sink, p.halves);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        const double bytes = (double)blocks * waves * iters * inflight * 1024.0;
        printf("inflight/wave %2d waves %d  %-62s %8.3f ms  %7.2f TB/s chip  %6.1f KB/us/CU\n", inflight, waves, p.name, best, bytes / best / 1e9,
               bytes / blocks / best / 1e3 / 1024.0 * 1.024);
      }
  return 0;
}
