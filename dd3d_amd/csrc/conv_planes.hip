// Implicit-GEMM convolution for gfx950 whose INPUT is already split into 16-bit planes (the producer's epilogue wrote them,
// conv_common.h::conv_epilogue): f32-equivalent (three bf16 terms, six cross products) or reduced (two terms / one term) products
// on the bf16 matrix pipe with NO VALU work in the main loop.
//
//   A  activations  [chunk c/32][pixel (b, h, w)][plane][32]   16-bit terms, written by the previous layer
//   B  filters      [n][k-tile][plane][32]                     split once at plan build
// Both operands of a K-tile (= 32 channels of one filter tap) stream HBM/L2 -> LDS with global_load_lds_dwordx4 (1 KiB per wave
// instruction = 16 rows x 64 B of one plane); the im2col gather is the per-lane source address of the A pieces (out-of-image taps
// and rows >= M read a zero page).  LDS rows are 64 B; their four 16-byte slots are XOR-swizzled with (row >> 2) & 3, applied on
// the source side of the DMA and again on the ds_read_b128 address, so that every fragment read is conflict-free.
//
// Pipeline (NS ring stages, NS - 1 tiles in flight): ONE barrier per K-tile, placed between the MFMA groups of its two 16-k chunks,
//   read F1 <- chunk 1 (tile kt) | MFMAs chunk 0 | wait tile kt+1 landed | barrier | DMA tile kt+NS -> the stage just freed |
//   read F0 <- chunk 0 (tile kt+1) | MFMAs chunk 1
// so the fragment reads of the next tile and the DMA issue sit under MFMAs that do not depend on them, and the two waves of a
// SIMD leave the barrier with matrix work already in hand.
#include <algorithm>
#include <cstring>
#include <stdlib.h>

#include "conv_common.h"

DD3D_NOTE_BUILD_FLAGS

namespace dd3d {

// Scheduling pattern of one phase: NMFMA matrix instructions, NDMA LDS-DMA issues and NDS fragment reads in ONE region.  The DMAs go
// first (longest latency), one per MFMA; then one ds_read per MFMA; the remaining MFMAs close the phase.  Without it the machine
// scheduler sinks the reads to just before their first use in the NEXT phase and serialises read -> wait -> MFMA.
template <int NMFMA, int NDS, int NDMA>
__device__ __forceinline__ void sched_interleave() {
  constexpr int NPAIR = NDMA + NDS < NMFMA ? NDMA + NDS : NMFMA;
#pragma unroll
  for (int i = 0; i < NPAIR; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
    if (i < NDMA) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // one VMEM (the LDS-DMA)
    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // one DS read
  }
  if constexpr (NDMA + NDS > NPAIR) {
    __builtin_amdgcn_sched_group_barrier(0x010, NDMA > NPAIR ? NDMA - NPAIR : 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, NDMA > NPAIR ? NDS : NDMA + NDS - NPAIR, 0);
  }
  if constexpr (NMFMA > NPAIR) __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NPAIR, 0);
}

#ifndef DD3D_PREFETCH_DISTANCE
#define DD3D_PREFETCH_DISTANCE 0  // K-tiles between the L2 touch of a tile and its DMA (0: no touch).  Measured: the touches only add VMEM instructions (towers 110 -> 115 us, small convs 22 -> 31 us): the loop is bound by DMA instruction throughput, not by miss latency
#endif
#ifndef DD3D_PRODUCER_WAVES
#define DD3D_PRODUCER_WAVES 0  // loader waves of the warp-specialised form (0: every wave loads and computes)
#endif
#ifndef DD3D_LDS_KIB_8W
#define DD3D_LDS_KIB_8W 144
#endif
#ifndef DD3D_LDS_KIB_4W
#define DD3D_LDS_KIB_4W 72
#endif
#ifndef DD3D_EPI_LDS
#define DD3D_EPI_LDS 1  // 1: the transposed epilogue stages its plane stores through LDS (1 KiB of consecutive bytes per store instruction); 0: straight from the registers (A/B)
#endif
#ifndef DD3D_EPI_T
#define DD3D_EPI_T 1  // 1: transposed accumulators + the 16-bytes-per-lane epilogue (conv_common.h::conv_epilogue_t); 0: round-3 form (A/B)
#endif
#ifndef DD3D_SCHED_VARIANT
#define DD3D_SCHED_VARIANT 1  // 0: DMA burst right after the barrier; 1: evenly spread over the phase; 2: spread over both phases of a step (NS >= 3)
#endif

// PW > 0: warp-specialised form -- PW extra LOADER waves (one per SIMD) issue every LDS-DMA of the block and the WM x WN compute waves
// only read fragments and issue MFMAs.  An LDS-DMA instruction holds its wave's issue slot for ~60-180 cycles (address path), and in
// the unspecialised loop all waves pay that at the same moment (right after the barrier): measured on the head towers, DMA-only loop
// 55 us, MFMA-only loop 65 us, both in every wave 82 us (zero operands).  Loaders and compute waves meet at the same one barrier per K-tile.
template <int TM, int TN, int WM, int WN, int NS, int MODE, bool SK, int PW = 0>
__global__ __launch_bounds__(64 * (WM * WN + PW)) void conv_igemm_planes_kernel(const ConvKArgs a) {
  constexpr int NP = Planes<MODE>::NP;
  constexpr int BM = TM * 32 * WM;
  constexpr int BN = TN * 32 * WN;
  constexpr int NW = WM * WN;
  constexpr int NTHR = 64 * NW;      // compute threads (the accumulator / split-K layouts are theirs)
  constexpr int NL = PW > 0 ? PW : NW;  // waves that issue DMA
  constexpr int PLA = BM * 64, PLB = BN * 64;      // bytes per plane of a stage
  constexpr int A_BYTES = NP * PLA, STAGE = NP * (PLA + PLB);
  constexpr int RA = BM / 16, RB = BN / 16;        // 16-row blocks (one 1-KiB DMA piece per plane)
  constexpr int QN = (RA + RB + NL - 1) / NL;      // row blocks per loading wave (the surplus re-fetches the last block)
  constexpr int P = QN * NP;                       // DMA instructions per wave and K-tile
  constexpr int EV_OFF = NS * STAGE;  // [scale | bias | lo][BN] floats of the epilogue (conv_epilogue_t)
  static_assert(NS >= 2 && EV_OFF + 12 * BN <= 160 * 1024, "LDS ring");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* lds = reinterpret_cast<unsigned char*>(smem);
  typedef unsigned char __attribute__((address_space(3))) * ldsbp;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = PW > 0 && wave >= NW;          // wave-uniform role
  const int lw = PW > 0 ? (wave >= NW ? wave - NW : 0) : wave;  // index among the loading waves
  const int wm = (wave < NW ? wave : 0) / WN;
  const int wn = (wave < NW ? wave : 0) - wm * WN;

  const int bid = remap_block(blockIdx.x, a.ntiles * a.nn);
  const int mt = bid / a.nn;
  const int nt = bid - mt * a.nn;
  int m0 = mt * BM;
  dd3d_conv_seg s = a.seg0;
  if (!a.single) {
    m0 = a.tiles[2 * mt + 1];
    s = a.segs[a.tiles[2 * mt]];
  }
  const int n0 = nt * BN;
  const gcbp g_in = (gcbp)s.in_planes;
  const gcbp g_w = (gcbp)s.w;
  const gcbp g_zero = (gcbp)a.zeros;

  const int nk = a.Kpad / BK;
  int kt_begin = 0, kt_end = nk;
  if (SK) {
    kt_begin = blockIdx.y * a.kt_per_split;
    kt_end = min(nk, kt_begin + a.kt_per_split);
  }
  const int ntile = kt_end - kt_begin;

  // ---- DMA geometry.  Row block r of the stage: r < RA -> A rows 16 r .., else B rows 16 (r - RA) ..; wave w moves blocks
  // w, w + NW, ... (all planes of a block).  lane -> (row = lane >> 2, LDS slot = lane & 3 holding k-slot (lane & 3) ^ ((lane >> 4) & 3)).
  const int slot16 = (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
  const long in_cstride = (long)s.B * s.H * s.W * (NP * 64);  // bytes per 32-channel chunk image of the input
  gcbp q_src[QN];   // A: this lane's pixel for tap (0, 0), chunk 0 / B: this lane's filter row, K-tile 0 (k-slot of the lane included)
  int a_hi0[QN], a_wi0[QN];
  int q_dst[QN];    // wave-uniform: LDS byte offset of the block inside a stage (plane 0)
  int q_pst[QN];    // wave-uniform: LDS bytes between the planes of the block
  bool q_isA[QN];   // wave-uniform
  {
    const int howo = s.Ho * s.Wo;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int r = min(q * NL + lw, RA + RB - 1);
      q_isA[q] = r < RA;
      q_dst[q] = r < RA ? r * 1024 : A_BYTES + (r - RA) * 1024;
      q_pst[q] = r < RA ? PLA : PLB;
      // A geometry
      const int m = m0 + r * 16 + (lane >> 2);
      const bool live = r < RA && m < s.M;
      const int b = m / howo;
      const int rr = m - b * howo;
      const int ho = rr / s.Wo;
      const int wo = rr - ho * s.Wo;
      const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
      const long a_off = (((long)b * s.H + hi0) * s.W + wi0) * (NP * 64) + slot16;
      // B geometry (rows past Npad feed columns >= N, which are never stored)
#if DD3D_EPI_T  // LDS row R of the B rows holds filter row chan_of_row(R) of its 32-row block (conv_common.h::conv_epilogue_t)
      const int brow = (r - RA) * 16 + (lane >> 2);
      const int n = min(n0 + (brow & ~31) + chan_of_row(brow & 31), a.Npad - 1);
#else
      const int n = min(n0 + (r - RA) * 16 + (lane >> 2), a.Npad - 1);
#endif
      const long b_off = (long)n * nk * (NP * 64) + slot16;
      a_hi0[q] = r < RA ? (live ? hi0 : -(1 << 28)) : 0;  // dead A rows are never inside [0, H); B rows always are
      a_wi0[q] = r < RA ? wi0 : 0;
      q_src[q] = r < RA ? g_in + (live ? a_off : 0) : g_w + b_off;
    }
  }

  // The stream walks this block's K-tiles in order; (chunk, tap) are carried instead of divided out of kt.  Tiles past the end
  // re-fetch the last one (into a stage nobody reads any more): every issue_tile() is exactly P DMA instructions, so the counted
  // waits stay exact.  Branch-free (wave-uniform selects), so the whole K loop body is one scheduling region.
  int ld_kt = kt_begin;
  int ld_chunk = kt_begin / a.T;
  int ld_tap = kt_begin - ld_chunk * a.T;
  // prepare(): source addresses of the tile the stream stands on (then the stream advances); emit(): its DMA instructions.  Split in
  // two so that the address arithmetic (scalar tap decode, bounds, 64-bit adds) runs BEFORE the barrier that frees the target stage
  // and only the DMA instructions themselves follow it.
  gcbp nxt_src[QN];
  auto prepare = [&]() {
    const int dh = (ld_tap * a.kw_magic) >> 16;
    const int dw = ld_tap - dh * a.KW;
    const long koff_a = (long)ld_chunk * in_cstride + ((long)dh * s.W + dw) * (NP * 64);
    const long koff_b = (long)ld_kt * (NP * 64);
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      // (bitwise, not short-circuit: no branches in the K loop)
      const bool ok = (int)!q_isA[q] | ((int)((unsigned)(a_hi0[q] + dh) < (unsigned)s.H) & (int)((unsigned)(a_wi0[q] + dw) < (unsigned)s.W));
      nxt_src[q] = ok ? q_src[q] + (q_isA[q] ? koff_a : koff_b) : g_zero + slot16;  // the zero page covers NP planes x 64 B
    }
    const int adv = ld_kt + 1 < kt_end;
    ld_kt += adv;
    ld_tap += adv;
    const int wrap = ld_tap == a.T;
    ld_tap = wrap ? 0 : ld_tap;
    ld_chunk += wrap;
  };
  auto emit = [&](int stage, auto qb_c, auto qe_c) {  // row blocks [qb, qe) of the prepared tile
    constexpr int QB = decltype(qb_c)::value, QE = decltype(qe_c)::value;
    unsigned char* st = lds + stage * STAGE;
#pragma unroll
    for (int q = QB; q < QE; ++q)
#pragma unroll
      for (int p = 0; p < NP; ++p)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(nxt_src[q] + p * 64), (ldsbp)(st + q_dst[q] + p * q_pst[q]), 16, 0, 0);
  };
  // L2 prefetch: the tile the DMA stream reaches PFD steps from now is touched with one 4-byte load per lane at the very addresses
  // its DMA will use.  The first tap of every 32-channel chunk reads activation lines no block has touched yet (the producer wrote
  // them through ANOTHER XCD's L2), and a counted vmcnt wait is only as fast as the slowest line of the tile: without the touch one
  // step in nine waits for an HBM round trip.  The loads share vmcnt with the DMAs (in order), hence one per row block and step in
  // every wave, so the counted waits stay exact; their destination register is never read.
  constexpr int PFD = DD3D_PREFETCH_DISTANCE;
  constexpr int PFN = PFD > 0 ? QN : 0;
  int pf_kt = kt_begin, pf_chunk = kt_begin / a.T, pf_tap = kt_begin - (kt_begin / a.T) * a.T;
  unsigned pf_sink[QN];
#pragma unroll
  for (int q = 0; q < QN; ++q) pf_sink[q] = 0;
  auto pf_advance = [&]() {
    const int adv = pf_kt + 1 < kt_end;
    pf_kt += adv;
    pf_tap += adv;
    const int wrap = pf_tap == a.T;
    pf_tap = wrap ? 0 : pf_tap;
    pf_chunk += wrap;
  };
  auto prefetch = [&]() {
    if constexpr (PFD > 0) {
      const int dh = (pf_tap * a.kw_magic) >> 16;
      const int dw = pf_tap - dh * a.KW;
      const long koff_a = (long)pf_chunk * in_cstride + ((long)dh * s.W + dw) * (NP * 64);
      const long koff_b = (long)pf_kt * (NP * 64);
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const bool ok = (int)!q_isA[q] | ((int)((unsigned)(a_hi0[q] + dh) < (unsigned)s.H) & (int)((unsigned)(a_wi0[q] + dw) < (unsigned)s.W));
        const gcbp src = ok ? q_src[q] + (q_isA[q] ? koff_a : koff_b) : g_zero + slot16;
        // "+v": one register chain for the whole loop -- a plain output would be dead right after each definition, and the allocator
        // could hand the register to something else while the load is still in flight
        asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink[q]) : "v"(src) : "memory");
      }
      pf_advance();
    }
  };
  constexpr std::integral_constant<int, 0> Q0{};
  constexpr std::integral_constant<int, QN> QALL{};
  // variant 2: the first QH row blocks of a tile go out after the barrier that frees their stage, the rest before the next one
  constexpr bool SPLIT = DD3D_SCHED_VARIANT == 2 && NS >= 3 && QN >= 2;
  constexpr int QH = SPLIT ? (QN + 1) / 2 : QN;
  constexpr std::integral_constant<int, QH> QMID{};

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lrow = lane & 31;
  const int kh = lane >> 5;
  const int swz = (lrow >> 2) & 3;
  const int frag_off[2] = {lrow * 64 + (((0 + kh) ^ swz) << 4), lrow * 64 + (((2 + kh) ^ swz) << 4)};  // k-chunk 0 / 1
  const int a_row0 = wm * TM * 32 * 64, b_row0 = A_BYTES + wn * TN * 32 * 64;

  bf16x8 fa[2][TM][NP], fb[2][TN][NP];  // fragment sets of the two 16-k chunks
  auto read_frags = [&](int stage, auto c_c) {
    constexpr int c = decltype(c_c)::value;
    const unsigned char* st = lds + stage * STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) fa[c][i][p] = *reinterpret_cast<const bf16x8*>(st + a_row0 + p * PLA + i * 32 * 64 + frag_off[c]);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p) fb[c][j][p] = *reinterpret_cast<const bf16x8*>(st + b_row0 + p * PLB + j * 32 * 64 + frag_off[c]);
  };
  // cross products, smallest terms first; (i, j) innermost so consecutive MFMAs hit different accumulators
  constexpr int NPROD = NP == 3 ? 6 : (NP == 2 ? 3 : 1);
  constexpr int PA_[6] = {NP == 3 ? 2 : (NP == 2 ? 1 : 0), 0, NP == 3 ? 1 : 0, 1, 0, 0};
  constexpr int PB_[6] = {0, NP == 3 ? 2 : (NP == 2 ? 1 : 0), NP == 3 ? 1 : 0, 0, 1, 0};
  auto mfma_chunk = [&](auto c_c) {
    constexpr int c = decltype(c_c)::value;
#pragma unroll
    for (int t = 0; t < NPROD; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#if DD3D_EPI_T  // filter fragment first: the accumulator block is [channel][pixel] (conv_common.h::conv_epilogue_t)
          if constexpr (Planes<MODE>::F16)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[c][j][PB_[t]]), __builtin_bit_cast(f16x8, fa[c][i][PA_[t]]),
                                                               acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[c][j][PB_[t]], fa[c][i][PA_[t]], acc[i][j], 0, 0, 0);
#else
          if constexpr (Planes<MODE>::F16)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[c][i][PA_[t]]), __builtin_bit_cast(f16x8, fb[c][j][PB_[t]]),
                                                               acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][i][PA_[t]], fb[c][j][PB_[t]], acc[i][j], 0, 0, 0);
#endif
  };
  constexpr std::integral_constant<int, 0> C0{};
  constexpr std::integral_constant<int, 1> C1{};

#if DD3D_EPI_T
  static_assert(PW == 0, "the transposed epilogue stages its vectors in the unspecialised prologue");
  EpiVec<BN, NTHR> evv;
  if (ntile <= 0) {  // an empty K slice: the split-K exchange's barriers publish the vectors
    epi_load_vectors<BN, NTHR>(a, s, n0, tid, evv);
    epi_store_vectors<BN, NTHR>(lds + EV_OFF, tid, evv);
  }
#endif
  if constexpr (PW > 0) {
    if (ntile > 0) {
      if (loader) {
        // ---- loader waves: ring fill, then per K-tile: addresses -> [tile kt+1 landed] -> barrier -> DMA of tile kt+NS
#pragma unroll
        for (int d = 0; d < NS; ++d) {
          prepare();
          emit(d, Q0, QALL);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * P) : "memory");
        __builtin_amdgcn_s_barrier();
        int stage = 0;
        for (int kt = 0; kt < ntile; ++kt) {
          prepare();
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * P) : "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          emit(stage, Q0, QALL);
          stage = stage == NS - 1 ? 0 : stage + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus fetches land before the block may release its LDS
        return;
      }
      // ---- compute waves: fragments + MFMAs only
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      read_frags(0, C0);
      int stage = 0;
      constexpr int NM = TM * TN * NPROD, NDS = (TM + TN) * NP;
      for (int kt = 0; kt < ntile; ++kt) {
        read_frags(stage, C1);
        mfma_chunk(C0);
        sched_uniform<NM, NDS, 0>();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        stage = stage == NS - 1 ? 0 : stage + 1;
        read_frags(stage, C0);
        mfma_chunk(C1);
        sched_uniform<NM, NDS, 0>();
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (loader) {
      return;
    }
  } else
  if (ntile > 0) {
#if DD3D_EPI_T
    epi_load_vectors<BN, NTHR>(a, s, n0, tid, evv);  // (oldest vector-memory operations of the wave: landed by the prologue's counted wait)
#endif
    // prologue: fill the ring (tiles 0 .. NS-1; variant 2 leaves the second part of tile NS-1 to the first step), wait for tile 0
#pragma unroll
    for (int d = 0; d < NS; ++d) {
      prepare();
      if (SPLIT && d == NS - 1) emit(d, Q0, QMID);
      else emit(d, Q0, QALL);
    }
    if constexpr (PFD > 0) {  // the touch stream starts NS + PFD tiles in: tiles NS .. NS+PFD-1 go untouched (one start-up latency)
      for (int d = 0; d < NS + PFD; ++d) pf_advance();
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPLIT ? (NS - 2) * P + QH * NP : (NS - 1) * P) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#if DD3D_EPI_T
    epi_store_vectors<BN, NTHR>(lds + EV_OFF, tid, evv);  // published by the first step's barrier
#endif
    read_frags(0, C0);
    int stage = 0;        // ring stage of tile kt
    int fill = NS - 1;    // variant 2: stage whose tile is half issued
    constexpr int NM = TM * TN * NPROD, NDS = (TM + TN) * NP;
    for (int kt = 0; kt < ntile; ++kt) {
      // ---- phase A: chunk-1 fragment reads of tile kt under the chunk-0 MFMAs (variant 2: + the second part of the tile in flight),
      // and the addresses of the next tile to fetch
      if constexpr (SPLIT) emit(fill, QMID, QALL);
      read_frags(stage, C1);
      mfma_chunk(C0);
      prepare();
      if constexpr (DD3D_SCHED_VARIANT == 0) sched_interleave<NM, NDS, 0>();
      else sched_uniform<NM, NDS, SPLIT ? (QN - QH) * NP : 0>();
      __builtin_amdgcn_sched_barrier(0);
      // my pieces of tile kt+1 have landed once at most NS-2 newer tiles are in flight; my reads of this stage are done
      // (in issue order behind tile kt+1: its step's PFN touches, then NS-2 steps of P DMAs + PFN touches)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (P + PFN) + (SPLIT ? 0 : PFN)) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // everyone: tile kt+1 landed, stage `stage` (tile kt) no longer read
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase B: DMA of tile kt+NS into the stage just freed and chunk-0 fragment reads of tile kt+1 under the chunk-1 MFMAs
      if constexpr (SPLIT) {
        emit(stage, Q0, QMID);
        fill = stage;
      } else {
        emit(stage, Q0, QALL);
      }
      prefetch();
      stage = stage == NS - 1 ? 0 : stage + 1;
      read_frags(stage, C0);  // (past the end: a stage holding surplus data, never used)
      mfma_chunk(C1);
      if constexpr (DD3D_SCHED_VARIANT == 0) sched_interleave<NM, NDS, P>();
      else sched_uniform<NM, NDS, (SPLIT ? QH * NP : P) + PFN>();
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus prefetches must land before the LDS is released
#pragma unroll
    for (int q = 0; q < QN; ++q) asm volatile("" ::"v"(pf_sink[q]));  // the touch registers stay allocated until here
  }

  if constexpr (SK) {
    if (!splitk_exchange<TM, TN, NTHR>(a, acc, bid, tid, blockIdx.y)) return;
  }
#if DD3D_EPI_T
#if DD3D_EPI_LDS
  // plane stores staged through LDS (conv_epilogue_t): the rings are dead, but other waves' surplus LDS-DMAs / fragment reads of the last
  // K steps may still touch them -- every wave has waited for its own (vmcnt(0) above), one barrier makes that true for all of them
  unsigned char* scratch = nullptr;
  if (s.out_planes != nullptr) {
    __syncthreads();
    scratch = lds + wave * (NP * 2048);
  }
  conv_epilogue_t<TM, TN, MODE, WM, WN>(a, s, acc, m0, n0, wm, wn, lane, lds + EV_OFF, scratch);
#else
  conv_epilogue_t<TM, TN, MODE, WM, WN>(a, s, acc, m0, n0, wm, wn, lane, lds + EV_OFF, nullptr);
#endif
#else
  conv_epilogue<TM, TN, MODE, WM, WN>(a, s, acc, m0, n0, wm, wn, lane);
#endif
}

// ------------------------------------------------------------------------------------------------------------------ host
template <int TM, int TN, int WM, int WN, int MODE, bool ALLOW_SK = true>
static int launch_planes_tile(const ConvKArgs& ka, hipStream_t st) {
  constexpr int NP = Planes<MODE>::NP;
  constexpr int BM = TM * 32 * WM, BN = TN * 32 * WN;
  // loader waves (warp-specialised form) for the big 8-wave tiles, whose blocks own a CU for hundreds of K-tiles
  constexpr int PW = (WM * WN == 8 && TM * TN >= 2) ? DD3D_PRODUCER_WAVES : 0;
  constexpr int NTHR = 64 * (WM * WN + PW);
  constexpr int STAGE = NP * (BM + BN) * 64;
  // LDS ring budgets (KiB): what a block may take decides how many blocks -- of this launch or of another stream's -- share a CU
  constexpr int BUDGET = ((WM * WN == 8 || STAGE > 32 * 1024) ? DD3D_LDS_KIB_8W : DD3D_LDS_KIB_4W) * 1024;  // (a big 4-wave tile owns its CU anyway)
  constexpr int NS0 = BUDGET / STAGE;
  constexpr int NS = NS0 > 4 ? 4 : (NS0 < 2 ? 2 : NS0);
  const size_t lds = (size_t)NS * STAGE + 12 * BN;  // rings + the epilogue vectors
  dim3 grid(ka.ntiles * ka.nn, ka.splitk, 1);
  static unsigned long long attr_done[4];
  if (lds_opt_in_needed(attr_done)) {
    if (lds_opt_in(reinterpret_cast<const void*>(conv_igemm_planes_kernel<TM, TN, WM, WN, NS, MODE, false, PW>), (size_t)(lds), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
    if constexpr (ALLOW_SK)
      if (lds_opt_in(reinterpret_cast<const void*>(conv_igemm_planes_kernel<TM, TN, WM, WN, NS, MODE, true, PW>), (size_t)(lds), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
    lds_opt_in_done(attr_done);  // (every opt-in of this call site succeeded on this device)
  }
  if constexpr (!ALLOW_SK) DD3D_REQUIRE(ka.splitk == 1, "dd3d_conv2d_igemm_f32: this tile has no split-K form (its accumulators fill the register file)");
  if constexpr (ALLOW_SK) {
    if (ka.splitk > 1) {
      hipLaunchKernelGGL((conv_igemm_planes_kernel<TM, TN, WM, WN, NS, MODE, true, PW>), grid, dim3(NTHR), lds, st, ka);
      return check_launch("launch_planes_tile split-K kernel");
    }
  }
  hipLaunchKernelGGL((conv_igemm_planes_kernel<TM, TN, WM, WN, NS, MODE, false, PW>), grid, dim3(NTHR), lds, st, ka);
  return check_launch("conv_igemm_planes kernel");
}

template <int MODE>
static int launch_planes_mode(const ConvKArgs& ka, int tile_cfg, hipStream_t st) {
  switch (tile_cfg) {
    case DD3D_TILE_256x128: return launch_planes_tile<2, 2, 4, 2, MODE>(ka, st);
    case DD3D_TILE_128x128: return launch_planes_tile<2, 1, 2, 4, MODE>(ka, st);
    case DD3D_TILE_128x64_K2:
    case DD3D_TILE_128x64: return launch_planes_tile<1, 1, 4, 2, MODE>(ka, st);
    case DD3D_TILE_64x128_K2:
    case DD3D_TILE_64x128: return launch_planes_tile<1, 1, 2, 4, MODE>(ka, st);
    case DD3D_TILE_128x128_W4: return launch_planes_tile<2, 2, 2, 2, MODE>(ka, st);
    case DD3D_TILE_64x64_W4K2:
    case DD3D_TILE_64x64_W4: return launch_planes_tile<1, 1, 2, 2, MODE>(ka, st);
    case DD3D_TILE_128x64_W4: return launch_planes_tile<2, 1, 2, 2, MODE>(ka, st);
    case DD3D_TILE_128x32_W4: return launch_planes_tile<1, 1, 4, 1, MODE>(ka, st);
    case DD3D_TILE_256x128_T42: return launch_planes_tile<4, 2, 2, 2, MODE>(ka, st);
    case DD3D_TILE_128x256_T24: return launch_planes_tile<2, 4, 2, 2, MODE>(ka, st);
    case DD3D_TILE_256x256_W8:
      if constexpr (Planes<MODE>::NP <= 2) return launch_planes_tile<4, 2, 2, 4, MODE, false>(ka, st);
      break;
  }
  DD3D_REQUIRE(false, "dd3d_conv2d_igemm_f32: tile_cfg %d has no split-plane kernel", tile_cfg);
}

int launch_conv_planes(const ConvKArgs& ka, int math_mode, int tile_cfg, hipStream_t st) {
  switch (math_mode) {
    case DD3D_MATH_BF16X3: return launch_planes_mode<DD3D_MATH_BF16X3>(ka, tile_cfg, st);
    case DD3D_MATH_BF16X2: return launch_planes_mode<DD3D_MATH_BF16X2>(ka, tile_cfg, st);
    case DD3D_MATH_BF16: return launch_planes_mode<DD3D_MATH_BF16>(ka, tile_cfg, st);
    case DD3D_MATH_F16X2: return launch_planes_mode<DD3D_MATH_F16X2>(ka, tile_cfg, st);
  }
  DD3D_REQUIRE(false, "dd3d_conv2d_igemm_f32: math mode %d has no split-plane kernel", math_mode);
}

}  // namespace dd3d

// ------------------------------------------------------------------------------------------------------------------
// f32 NHWC -> split planes (for tensors a non-convolution kernel produced).  One thread per (pixel, 8 channels): two 16-byte loads,
// one 16-byte store per plane; HBM-bound.
namespace dd3d {

template <int MODE>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, int M, int C8, int in_pitch,
                                                           int relu, float pscale, int* status) {
  constexpr int NP = Planes<MODE>::NP;
  const long total = (long)M * C8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int m = (int)(t / C8);
    const int c8 = (int)(t - (long)m * C8);
    const float* src = in + (long)m * in_pitch + c8 * 8;
    f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x0[e] = fmaxf(x0[e], 0.f), x1[e] = fmaxf(x1[e], 0.f);
    }
    if constexpr (Planes<MODE>::F16) {
      int ovf = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x0[e] *= pscale, x1[e] *= pscale;
        ovf |= !(fabsf(x0[e]) <= 65504.f) | !(fabsf(x1[e]) <= 65504.f);
      }
      if (ovf && status) atomicOr(status, DD3D_STATUS_F16_OVERFLOW);
    }
    unsigned w[4][NP];
    split_pack<MODE>(x0[0], x0[1], w[0]);
    split_pack<MODE>(x0[2], x0[3], w[1]);
    split_pack<MODE>(x1[0], x1[1], w[2]);
    split_pack<MODE>(x1[2], x1[3], w[3]);
    unsigned char* dst = out + ((long)(c8 >> 2) * M + m) * (NP * 64) + (c8 & 3) * 16;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<u32x4*>(dst + p * 64) = u32x4{w[0][p], w[1][p], w[2][p], w[3][p]};
    }
  }
}

}  // namespace dd3d

extern "C" int dd3d_split_planes(const float* in, void* out, int32_t M, int32_t C, int32_t in_pitch, int32_t math_mode, int32_t relu, float plane_scale,
                                 int32_t* status, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(in && out && M > 0 && C > 0 && C % 32 == 0 && in_pitch >= C && in_pitch % 4 == 0, "dd3d_split_planes: bad shape (M=%d C=%d pitch=%d)", M,
               C, in_pitch);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long total = (long)M * (C / 8);
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  unsigned char* o = reinterpret_cast<unsigned char*>(out);
  const float ps = plane_scale > 0.f ? plane_scale : 1.f;
  switch (math_mode) {
    case DD3D_MATH_BF16X3: hipLaunchKernelGGL(split_planes_kernel<DD3D_MATH_BF16X3>, dim3(blocks), dim3(256), 0, st, in, o, M, C / 8, in_pitch, relu, ps, status); break;
    case DD3D_MATH_BF16X2: hipLaunchKernelGGL(split_planes_kernel<DD3D_MATH_BF16X2>, dim3(blocks), dim3(256), 0, st, in, o, M, C / 8, in_pitch, relu, ps, status); break;
    case DD3D_MATH_BF16: hipLaunchKernelGGL(split_planes_kernel<DD3D_MATH_BF16>, dim3(blocks), dim3(256), 0, st, in, o, M, C / 8, in_pitch, relu, ps, status); break;
    case DD3D_MATH_F16X2: hipLaunchKernelGGL(split_planes_kernel<DD3D_MATH_F16X2>, dim3(blocks), dim3(256), 0, st, in, o, M, C / 8, in_pitch, relu, ps, status); break;
    default: DD3D_REQUIRE(false, "dd3d_split_planes: math mode %d has no planes", math_mode);
  }
  return check_launch("split_planes kernel");
}

// ------------------------------------------------------------------------------------------------------------------
// Pooling / FPN top-down kernels that also write the split planes of their result (one launch instead of kernel + dd3d_split_planes).
namespace dd3d {

template <int MODE>
__device__ __forceinline__ void store_planes8(unsigned char* out_planes, long M, long m, int c8, f32x4 x0, f32x4 x1, float pscale, int* status) {
  constexpr int NP = Planes<MODE>::NP;
  if constexpr (Planes<MODE>::F16) {
    int ovf = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x0[e] *= pscale, x1[e] *= pscale;
      ovf |= !(fabsf(x0[e]) <= 65504.f) | !(fabsf(x1[e]) <= 65504.f);
    }
    if (ovf && status) atomicOr(status, DD3D_STATUS_F16_OVERFLOW);
  }
  unsigned w[4][NP];
  split_pack<MODE>(x0[0], x0[1], w[0]);
  split_pack<MODE>(x0[2], x0[3], w[1]);
  split_pack<MODE>(x1[0], x1[1], w[2]);
  split_pack<MODE>(x1[2], x1[3], w[3]);
  unsigned char* dst = out_planes + ((long)(c8 >> 2) * M + m) * (NP * 64) + (c8 & 3) * 16;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(dst + p * 64) = u32x4{w[0][p], w[1][p], w[2][p], w[3][p]};
}

template <int MODE>
__global__ __launch_bounds__(256) void maxpool2x2_planes_kernel(const float* __restrict__ in, float* __restrict__ out, unsigned char* __restrict__ out_planes,
                                                                int B, int H, int W, int C8, int in_pitch, int out_pitch, float pscale, int* status) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long M = (long)B * Ho * Wo, total = M * C8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long m = t / C8;
    const int c8 = (int)(t - m * C8);
    const int wo = (int)(m % Wo);
    const long r = m / Wo;
    const int ho = (int)(r % Ho), b = (int)(r / Ho);
    const float* p = in + (((long)b * H + 2 * ho) * W + 2 * wo) * in_pitch + c8 * 8;
    f32x4 o[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 v00 = *reinterpret_cast<const f32x4*>(p + 4 * h), v01 = *reinterpret_cast<const f32x4*>(p + in_pitch + 4 * h);
      const f32x4 v10 = *reinterpret_cast<const f32x4*>(p + (long)W * in_pitch + 4 * h), v11 = *reinterpret_cast<const f32x4*>(p + (long)W * in_pitch + in_pitch + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[h][e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
    }
    if (out) {
      float* q = out + m * out_pitch + c8 * 8;
      *reinterpret_cast<f32x4*>(q) = o[0];
      *reinterpret_cast<f32x4*>(q + 4) = o[1];
    }
    store_planes8<MODE>(out_planes, M, m, c8, o[0], o[1], pscale, status);
  }
}

// 2x2 / stride 2 max-pool of a map that exists as split planes only (ABI 4: the DLA data flow keeps no f32 twin of its tensors).  One
// thread per (output pixel, 8 channels): the four candidates' values are rebuilt from their terms (largest term first), and the TERMS of
// the largest value are copied -- the pooled planes are exactly the planes of one of the four inputs, nothing is re-split.
template <int MODE>
__global__ __launch_bounds__(256) void maxpool2x2_planes_in_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int B, int H, int W,
                                                                   int C8) {
  constexpr int NP = Planes<MODE>::NP;
  const int Ho = H >> 1, Wo = W >> 1;
  const long Mi = (long)B * H * W, Mo = (long)B * Ho * Wo, total = Mo * C8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int q = (int)(t & 3);          // 16-byte slot of the 64-byte row: channels 8 q .. 8 q + 7 of the chunk
    const long u = t >> 2;
    const long m = u % Mo;
    const int chunk = (int)(u / Mo);
    const int wo = (int)(m % Wo);
    const long r = m / Wo;
    const int ho = (int)(r % Ho), b = (int)(r / Ho);
    const long p00 = ((long)b * H + 2 * ho) * W + 2 * wo;
    const long pix[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
    u32x4 w[4][NP];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int p = 0; p < NP; ++p) w[k][p] = *reinterpret_cast<const u32x4*>(in + ((long)chunk * Mi + pix[k]) * (NP * 64) + p * 64 + q * 16);
    u32x4 o[NP];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned sel_lo[NP], sel_hi[NP];  // terms of the winners of the dword's two channels
      float best_lo = 0.f, best_hi = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v_lo = 0.f, v_hi = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const unsigned d = w[k][p][e];
          if constexpr (Planes<MODE>::F16) {
            const f16x2 hh = __builtin_bit_cast(f16x2, d);
            v_lo += (float)hh[0], v_hi += (float)hh[1];
          } else {
            v_lo += __uint_as_float(d << 16), v_hi += __uint_as_float(d & 0xffff0000u);
          }
        }
        const bool tl = k == 0 || v_lo > best_lo, th = k == 0 || v_hi > best_hi;
        best_lo = tl ? v_lo : best_lo, best_hi = th ? v_hi : best_hi;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          sel_lo[p] = tl ? (w[k][p][e] & 0xffffu) : sel_lo[p];
          sel_hi[p] = th ? (w[k][p][e] & 0xffff0000u) : sel_hi[p];
        }
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) o[p][e] = sel_lo[p] | sel_hi[p];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(out + ((long)chunk * Mo + m) * (NP * 64) + p * 64 + q * 16) = o[p];
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void upsample2x_add_planes_kernel(float* __restrict__ fine, const float* __restrict__ coarse, unsigned char* __restrict__ fine_planes,
                                                                    int B, int H, int W, int C8, int fine_pitch, int coarse_pitch, float pscale, int* status) {
  const int Hc = H >> 1, Wc = W >> 1;
  const long M = (long)B * H * W, total = M * C8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long m = t / C8;
    const int c8 = (int)(t - m * C8);
    const int x = (int)(m % W);
    const long r = m / W;
    const int y = (int)(r % H), b = (int)(r / H);
    float* f = fine + m * fine_pitch + c8 * 8;
    const float* q = coarse + (((long)b * Hc + (y >> 1)) * Wc + (x >> 1)) * coarse_pitch + c8 * 8;
    const f32x4 o0 = *reinterpret_cast<const f32x4*>(f) + *reinterpret_cast<const f32x4*>(q);
    const f32x4 o1 = *reinterpret_cast<const f32x4*>(f + 4) + *reinterpret_cast<const f32x4*>(q + 4);
    *reinterpret_cast<f32x4*>(f) = o0;
    *reinterpret_cast<f32x4*>(f + 4) = o1;
    store_planes8<MODE>(fine_planes, M, m, c8, o0, o1, pscale, status);
  }
}

// ---- effective squeeze-excitation (vovnet.py:180-185,248-249: x * hsigmoid(fc(avgpool(x))) (+ identity)) in TWO launches
//   gap_mean_kernel        partial[b][rs][c] = sum of the rs-th slice of the H*W rows (two-pass sum, no atomics on the data); the block
//                          that finishes an image last (a counter per image) folds the partials, in slice order, into mean[b][:]
//   ese_gate_scale_kernel  block = (64 channels, a run of pixels, image): its 64 gates = hsigmoid(fc_w[co][:] . mean[b][:] + fc_b[co])
//                          from mean[b][:] staged in LDS, then out = x * gate (+ identity) written as f32 NHWC and / or split planes
// instead of pool / gate / scale / dd3d_split_planes (4 launches, 7 passes over the map -> 2 launches, 5 passes).  Sums are taken in
// the order of the three-launch dd3d_ese_nhwc.  A ONE-launch form (persistent grid of 64 blocks, grid-wide barriers between the
// steps) was built and measured: it cannot use more than a fraction of the chip's blocks without risking a barrier among blocks that
// are not co-resident, and at that size the streaming phases run at a quarter of the HBM rate -- V2-99 B=1 5.40 -> 6.38 ms,
// B=16 48.7 -> 54.1 ms (profiles/r02_ese_one_launch.txt).
constexpr int ESE_MAXC = 4096, ESE_PIXELS = 256;

__global__ __launch_bounds__(256) void gap_mean_kernel(const float* __restrict__ in, float* __restrict__ partial, float* __restrict__ mean,
                                                       int* __restrict__ counters, int HW, int C, int pitch, int RS, float inv_hw) {
  const int cg = blockIdx.x, rs = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int rl = tid >> 4, cl = tid & 15;
  const int c = cg * 64 + cl * 4;
  const int rows_per = (HW + RS - 1) / RS;
  const int r0 = rs * rows_per, r1 = min(HW, r0 + rows_per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 16) acc += *reinterpret_cast<const f32x4*>(in + ((long)b * HW + r) * pitch + c);
  __shared__ f32x4 red[16][16];
  __shared__ int last;
  red[rl][cl] = acc;
  __syncthreads();
  // the exchange follows the split-K one (conv_common.h): write-through stores + s_waitcnt, a device-scope counter, write-through
  // loads in the finishing block -- a release FENCE here would write back the whole L2, which still holds the producer's output
  if (rl == 0 && c < C) {
    f32x4 s = red[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][cl];
    st_sc1(partial + ((long)b * RS + rs) * C + c, s);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int prev = __hip_atomic_fetch_add(counters + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = prev == (int)(gridDim.x * gridDim.y) - 1;
    if (last) __hip_atomic_store(counters + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // left zero for the next launch
  }
  __syncthreads();
  if (!last) return;
  for (int c4 = tid; c4 * 4 < C; c4 += 256) {
    f32x4 m = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < RS; r0 += 8) {  // eight slices in flight, added in slice order
      f32x4 t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = ld_sc1(partial + ((long)b * RS + min(r0 + k, RS - 1)) * C + c4 * 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(t[k]));  // the uses below depend on the wait above
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (r0 + k < RS) m += t[k];
    }
    *reinterpret_cast<f32x4*>(mean + (long)b * C + c4 * 4) = m * inv_hw;
  }
}

struct EseK {
  const float* x;
  const float* identity;
  float* out;
  unsigned char* out_planes;
  const float* fc_w;
  const float* fc_b;
  const float* mean;
  int* status;
  int B, HW, C, x_pitch, id_pitch, out_pitch, pixels;
  float pscale;
};

template <int MODE>
__global__ __launch_bounds__(256) void ese_gate_scale_kernel(const EseK a) {
  const int cg = blockIdx.x, ps = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = a.C, HW = a.HW;
  __shared__ float mean_s[ESE_MAXC];
  __shared__ __attribute__((aligned(16))) float gate_s[64];
  for (int ci = tid; ci < C; ci += 256) mean_s[ci] = a.mean[(long)b * C + ci];
  __syncthreads();
  {
    // wave w: gates of channels cg*64 + 16w .. +15, lanes stride over the input channels (the summation order of ese_gate_kernel).
    // The filter rows come from the L2 / fabric at ~1 us a trip: 16 rows x 4 lane-strides = 64 independent loads are issued per
    // round, so a C = 1024 module needs 4 trips instead of 256 (measured: the dependent form took 126 us on the 2 MB stage-5 map).
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    const int co0 = cg * 64 + wave * 16;
    for (int i0 = 0; i0 < C; i0 += 256) {
      float v[16][4];
#pragma unroll
      for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ci = i0 + u * 64 + lane, co = co0 + k;
          v[k][u] = (ci < C && co < C) ? a.fc_w[(long)co * C + ci] : 0.f;
        }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ci = i0 + u * 64 + lane;
        const float m = ci < C ? mean_s[ci] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (ci < C) acc[k] += v[k][u] * m;
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float t = acc[k];
      for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d, 64);
      const int co = co0 + k;
      if (lane == 0) gate_s[wave * 16 + k] = co < C ? fminf(fmaxf(t + a.fc_b[co] + 3.0f, 0.f), 6.0f) / 6.0f : 0.f;  // F.relu6(x + 3) / 6
    }
  }
  __syncthreads();
  const int q = tid & 7, c8 = cg * 8 + q;  // 8 threads per pixel, 8 channels each
  if (c8 * 8 >= C) return;
  const f32x4 g0 = *reinterpret_cast<const f32x4*>(gate_s + q * 8), g1 = *reinterpret_cast<const f32x4*>(gate_s + q * 8 + 4);
  const long M = (long)a.B * HW;
  const int p0 = ps * a.pixels, p1 = min(HW, p0 + a.pixels);
  auto one = [&](int p, f32x4& o0, f32x4& o1) {
    const long m = (long)b * HW + p;
    const float* px = a.x + m * a.x_pitch + c8 * 8;
    o0 = *reinterpret_cast<const f32x4*>(px) * g0;
    o1 = *reinterpret_cast<const f32x4*>(px + 4) * g1;
    if (a.identity) {
      const float* pi = a.identity + m * a.id_pitch + c8 * 8;
      o0 += *reinterpret_cast<const f32x4*>(pi);
      o1 += *reinterpret_cast<const f32x4*>(pi + 4);
    }
  };
  auto put = [&](int p, f32x4 o0, f32x4 o1) {
    const long m = (long)b * HW + p;
    if (a.out) {
      float* po = a.out + m * a.out_pitch + c8 * 8;
      *reinterpret_cast<f32x4*>(po) = o0;
      *reinterpret_cast<f32x4*>(po + 4) = o1;
    }
    if constexpr (MODE != DD3D_MATH_F32) {
      if (a.out_planes) store_planes8<MODE>(a.out_planes, M, m, c8, o0, o1, a.pscale, a.status);
    }
  };
  for (int p = p0 + (tid >> 3); p < p1; p += 64) {  // two pixels 32 apart per round: their loads are independent
    f32x4 u0, u1, v0, v1;
    const bool two = p + 32 < p1;
    one(p, u0, u1);
    if (two) one(p + 32, v0, v1);
    put(p, u0, u1);
    if (two) put(p + 32, v0, v1);
  }
}

}  // namespace dd3d

#define DD3D_PLANE_MODE_SWITCH(KERNEL, ...)                                                                                          \
  switch (math_mode) {                                                                                                               \
    case DD3D_MATH_BF16X3: hipLaunchKernelGGL(KERNEL<DD3D_MATH_BF16X3>, dim3(blocks), dim3(256), 0, st, __VA_ARGS__); break;         \
    case DD3D_MATH_BF16X2: hipLaunchKernelGGL(KERNEL<DD3D_MATH_BF16X2>, dim3(blocks), dim3(256), 0, st, __VA_ARGS__); break;         \
    case DD3D_MATH_BF16: hipLaunchKernelGGL(KERNEL<DD3D_MATH_BF16>, dim3(blocks), dim3(256), 0, st, __VA_ARGS__); break;             \
    case DD3D_MATH_F16X2: hipLaunchKernelGGL(KERNEL<DD3D_MATH_F16X2>, dim3(blocks), dim3(256), 0, st, __VA_ARGS__); break;           \
    default: DD3D_REQUIRE(false, "math mode %d has no planes", math_mode);                                                          \
  }

extern "C" int dd3d_maxpool2x2_planes(const float* in, float* out, void* out_planes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t in_pitch,
                                      int32_t out_pitch, int32_t math_mode, float plane_scale, int32_t* status, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(in && out_planes && B > 0 && (H % 2) == 0 && (W % 2) == 0 && C > 0 && (C % 32) == 0 && (in_pitch % 4) == 0 && (out_pitch % 4) == 0,
               "dd3d_maxpool2x2_planes: bad arguments (H=%d W=%d C=%d)", H, W, C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  const float ps = plane_scale > 0.f ? plane_scale : 1.f;
  unsigned char* o = reinterpret_cast<unsigned char*>(out_planes);
  DD3D_PLANE_MODE_SWITCH(maxpool2x2_planes_kernel, in, out, o, B, H, W, C / 8, in_pitch, out_pitch, ps, status)
  return check_launch("maxpool2x2_planes kernel");
}

extern "C" int dd3d_maxpool2x2_planes_in(const void* in_planes, void* out_planes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t math_mode,
                                         void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(in_planes && out_planes && B > 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0 && C > 0 && (C % 32) == 0,
               "dd3d_maxpool2x2_planes_in: bad arguments (H=%d W=%d C=%d)", H, W, C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  const unsigned char* i = reinterpret_cast<const unsigned char*>(in_planes);
  unsigned char* o = reinterpret_cast<unsigned char*>(out_planes);
  DD3D_PLANE_MODE_SWITCH(maxpool2x2_planes_in_kernel, i, o, B, H, W, C / 8)
  return check_launch("maxpool2x2_planes_in kernel");
}

extern "C" int dd3d_upsample2x_add_planes(float* fine, const float* coarse, void* fine_planes, int32_t B, int32_t H, int32_t W, int32_t C,
                                          int32_t fine_pitch, int32_t coarse_pitch, int32_t math_mode, float plane_scale, int32_t* status, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(fine && coarse && fine_planes && B > 0 && (H % 2) == 0 && (W % 2) == 0 && C > 0 && (C % 32) == 0 && (fine_pitch % 4) == 0 &&
                   (coarse_pitch % 4) == 0, "dd3d_upsample2x_add_planes: bad arguments (H=%d W=%d C=%d)", H, W, C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long total = (long)B * H * W * (C / 8);
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  const float ps = plane_scale > 0.f ? plane_scale : 1.f;
  unsigned char* o = reinterpret_cast<unsigned char*>(fine_planes);
  DD3D_PLANE_MODE_SWITCH(upsample2x_add_planes_kernel, fine, coarse, o, B, H, W, C / 8, fine_pitch, coarse_pitch, ps, status)
  return check_launch("upsample2x_add_planes kernel");
}

extern "C" int dd3d_ese_fused(const float* x, const float* identity, float* out, void* out_planes, const float* fc_w, const float* fc_b, float* partial,
                              float* mean, int32_t* counters, int32_t B, int32_t HW, int32_t C, int32_t x_pitch, int32_t id_pitch, int32_t out_pitch,
                              int32_t rsplit, int32_t math_mode, float plane_scale, int32_t* status, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(x && (out || out_planes) && fc_w && fc_b && partial && mean && counters && B > 0 && B <= 65535 && HW > 0, "dd3d_ese_fused: bad arguments");
  DD3D_REQUIRE(C > 0 && C <= ESE_MAXC && (C % 8) == 0 && (x_pitch % 4) == 0 && (!out || (out_pitch % 4) == 0) && (!identity || (id_pitch % 4) == 0),
               "dd3d_ese_fused: C=%d must be a multiple of 8 up to %d, pitches multiples of 4", C, ESE_MAXC);
  DD3D_REQUIRE(!out_planes || (C % 32) == 0, "dd3d_ese_fused: split planes need C %% 32 == 0 (C=%d)", C);
  DD3D_REQUIRE(rsplit >= 1 && rsplit <= 1024, "dd3d_ese_fused: rsplit=%d", rsplit);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int CG = (C + 63) / 64;
  hipLaunchKernelGGL(gap_mean_kernel, dim3(CG, rsplit, B), dim3(256), 0, st, x, partial, mean, counters, HW, C, x_pitch, rsplit, 1.0f / (float)HW);
  int rc = check_launch("gap_mean_kernel");
  if (rc != DD3D_OK) return rc;
  EseK a;
  a.x = x, a.identity = identity, a.out = out, a.out_planes = reinterpret_cast<unsigned char*>(out_planes), a.fc_w = fc_w, a.fc_b = fc_b;
  a.mean = mean, a.status = status;
  a.B = B, a.HW = HW, a.C = C, a.x_pitch = x_pitch, a.id_pitch = id_pitch, a.out_pitch = out_pitch, a.pixels = ESE_PIXELS;
  a.pscale = plane_scale > 0.f ? plane_scale : 1.f;
  const dim3 grid(CG, (HW + ESE_PIXELS - 1) / ESE_PIXELS, B);
  DD3D_REQUIRE(grid.y <= 65535, "dd3d_ese_fused: H*W=%d too large", HW);
  if (!out_planes) math_mode = DD3D_MATH_F32;
  switch (math_mode) {
    case DD3D_MATH_F32: hipLaunchKernelGGL(ese_gate_scale_kernel<DD3D_MATH_F32>, grid, dim3(256), 0, st, a); break;
    case DD3D_MATH_BF16X3: hipLaunchKernelGGL(ese_gate_scale_kernel<DD3D_MATH_BF16X3>, grid, dim3(256), 0, st, a); break;
    case DD3D_MATH_BF16X2: hipLaunchKernelGGL(ese_gate_scale_kernel<DD3D_MATH_BF16X2>, grid, dim3(256), 0, st, a); break;
    case DD3D_MATH_BF16: hipLaunchKernelGGL(ese_gate_scale_kernel<DD3D_MATH_BF16>, grid, dim3(256), 0, st, a); break;
    case DD3D_MATH_F16X2: hipLaunchKernelGGL(ese_gate_scale_kernel<DD3D_MATH_F16X2>, grid, dim3(256), 0, st, a); break;
    default: DD3D_REQUIRE(false, "dd3d_ese_fused: math mode %d", math_mode);
  }
  return check_launch("ese_gate_scale_kernel");
}
