// 3x3 / stride 1 / pad 1 implicit-GEMM convolution on split-plane operands in which THE THREE TAPS OF A FILTER ROW SHARE ONE A STAGE.
//
// conv_planes.hip streams, for every K-tile (32 channels of ONE tap), the A rows of the block's output pixels shifted by that tap:
// nine fetches of (nearly) the same pixels per channel chunk.  Its loop is bound by the LDS-DMA instructions a CU can issue (measured:
// step time ~ 57 cycles x DMA instructions per CU, profiles/r02_conv_ablation.txt), not by the matrix pipe.  Here the K order is
// (chunk, dh, dw) with dw innermost, and an A stage holds the BM + 2 consecutive input pixels
//     q = m0 - 1 + j + (dh - 1) W ,   j = 0 .. BM + 1      (flattened (b, h, w) index, stride 1: input and output share it)
// of one (chunk, dh); tap dw of output row r reads LDS row r + dw.  One A fetch serves three K-tiles: 17 row blocks per three steps
// instead of 48 (256-row tile) -- 39 % fewer DMA instructions per MFMA for the two-term modes.
//
// What the per-tap gather gave for free is now explicit: a flattened neighbour is a real neighbour only inside the image, so every lane
// carries a 9-bit validity mask of its output pixels (0 <= ho + dh - 1 < H, 0 <= wo + dw - 1 < W, m < M) and points the fragment read of
// an invalid (pixel, tap) at 16 zero bytes in LDS (one address select per read) -- the same zeros the zero page supplied.
//
// Rings: two A stages (an A stage lives for three steps, so its successor has three steps to land) and NSB B stages (one per K-tile,
// as in conv_planes.hip).  One barrier per K-tile between the MFMA groups of its two 16-k chunks, DMA addresses computed before it.
#include <cstring>
#include <stdlib.h>

#include "conv_common.h"

DD3D_NOTE_BUILD_FLAGS

#ifndef DD3D_ROW_LDS_KIB_8W
#define DD3D_ROW_LDS_KIB_8W 152  // LDS a block of the 8-wave tiles may take (decides NSB = 3 or 2)
#endif
#ifndef DD3D_ROW_LDS_KIB_4W
#define DD3D_ROW_LDS_KIB_4W 76
#endif
#ifndef DD3D_EPI_LDS
#define DD3D_EPI_LDS 1  // 1: the transposed epilogue stages its plane stores through LDS (1 KiB of consecutive bytes per store instruction); 0: straight from the registers (A/B)
#endif
#ifndef DD3D_CHAIN_A_AUX
#define DD3D_CHAIN_A_AUX 16  // cache-policy bits of the activation LDS-DMA of a chain launch (16 = sc1, 2 = nt, 0 = default policy behind an agent acquire)
#endif
#ifndef DD3D_CHAIN_ACQUIRE
#define DD3D_CHAIN_ACQUIRE 0  // 1: a consumer block's polling lane issues an agent-scope acquire (invalidates the CU's L1) before the block reads its producers' rows
#endif
#ifndef DD3D_CHAIN_B_FIRST
#define DD3D_CHAIN_B_FIRST 1  // 1: a chain block issues its filter stages before it waits for its producers
#endif
#ifndef DD3D_EPI_T
#define DD3D_EPI_T 1  // 1: transposed accumulators + the 16-bytes-per-lane epilogue (conv_common.h::conv_epilogue_t); 0: round-3 form (A/B)
#endif

#ifndef DD3D_ROW_B_WAVES
#define DD3D_ROW_B_WAVES 0  // > 0: in the 8-wave tiles only the first DD3D_ROW_B_WAVES waves (one per SIMD: the ones the MFMA arbiter favours) issue the filter stages' LDS-DMA (measured neutral: r06o)
#endif
#ifndef DD3D_ROW_PRIO_SLICE
#define DD3D_ROW_PRIO_SLICE 0  // 8-wave tiles: 1: waves 4-7 run at priority 2 in the first half of a barrier interval and 0 in the second (waves 0-3 stay at 1): the
                               // second-served wave of a SIMD goes FIRST for half its MFMAs, so that nobody runs the end of the interval alone; 2: the roles swapped
#endif
#ifndef DD3D_ROW_B_WAVES_HI
#define DD3D_ROW_B_WAVES_HI 0  // 1: the LAST DD3D_ROW_B_WAVES waves issue the filter pieces (the ones the arbiter serves second: they issue while the first run their MFMAs)
#endif
#ifndef DD3D_ROW_B_SADDR
#define DD3D_ROW_B_SADDR 1  // 1: filter pieces use the scalar-base form of the LDS-DMA (s[base] + a constant 32-bit lane offset: no per-piece address arithmetic, half the address registers; round 6: towers -0.8 %, one image -1.1 %, profiles/r06o_bwaves_ab.txt); 0: the builtin's 64-bit-per-lane form (A/B)
#endif
#ifndef DD3D_ROW_STAMP
#define DD3D_ROW_STAMP 0  // 1: every wave of the first 512 blocks records s_memtime stamps of its phases (timing probe: dd3d_debug_row_stamps; results unchanged)
#endif

namespace dd3d {

#if DD3D_ROW_STAMP
// per wave: [0] entry, [1] prologue issued, [2] first data landed (prologue barrier passed), [3] K loop done, [4] kernel end,
// [5] sum over steps of (after-barrier -> own work done: fragment reads landed, MFMAs issued), [6] of the vmcnt wait, [7] of the barrier wait
__device__ unsigned long long g_row_stamps[512 * 8 * 8];
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#endif

// CHAIN (dd3d_conv_launch.chain): the launch's segments DEPEND on each other -- segment i reads, as its input (and possibly as its
// residual), what segments < i of the same launch write: the stride-1 3 x 3 convolutions of a DLA level (dla.py:50-62, conv1 -> conv2 +
// residual -> the next block's conv1 ...) become ONE launch instead of one per convolution.  Blocks take their items
// in launch order -- item = blockIdx.x = (m-tile, n-tile, K slice), slice fastest, the m-tiles in tiles[] order = dependency order, no
// XCD remap -- so a block's producers always have LOWER block indices.  A block streams its filter stages first, then waits (one lane
// polling, everybody else parked at the barrier) until the producer segment's m-tiles that cover its input rows have all their n-tiles
// finished (chain_sync), and only then issues its activation stages.  The hand-over crosses XCDs whose L2s are not coherent: producers
// store their planes write-through (sc1), wait for the acknowledgements (vmcnt(0)) and only then count their arrival with an agent-scope
// atomic; consumers read activations (LDS-DMA, aux = sc1) and residual planes with sc1 loads (MI355X_MICROARCH.md, inter-workgroup
// visibility: "16 B sc1 stores AND sc1 loads" -- the form the split-K exchange of this file already uses).
// Progress: the hardware dispatches a grid's workgroups in index order per XCD, so the lowest unfinished block is always resident and
// never waits on anything that is not resident or done.  That order is observed behaviour, not a HIP guarantee -- so the poll is
// BOUNDED: a block that has polled ~2 s gives up waiting, sets DD3D_STATUS_CHAIN_TIMEOUT and runs on (its results are then wrong and the
// host raises on the status word): a broken assumption ends in an error, never in a hung GPU.
template <int TM, int TN, int WM, int WN, int NSB, int MODE, bool SK, int NSA, bool CHAIN = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_planes_row_kernel(const ConvKArgs a) {
  constexpr int NP = Planes<MODE>::NP;
  constexpr int BM = TM * 32 * WM;
  constexpr int BN = TN * 32 * WN;
  constexpr int NW = WM * WN;
  constexpr int NTHR = 64 * NW;
  constexpr int AROWS = BM + 16;                   // BM + 2 used; whole 16-row DMA blocks
  constexpr int PLA = AROWS * 64, PLB = BN * 64;   // bytes per plane of an A / B stage
  constexpr int A_STAGE = NP * PLA, B_STAGE = NP * PLB;
  constexpr int B_BASE = NSA * A_STAGE;
  constexpr int NPA = (AROWS / 16) * NP, NPB = (BN / 16) * NP;  // 1-KiB pieces of an A stage / a B stage
  // Filter (B) pieces may be issued by the first BW waves only (DD3D_ROW_B_WAVES): of the two waves that share a SIMD the matrix pipe serves the
  // first-dispatched one first (measured: its 48 MFMAs of a step are out after 2370 cycles, the other's after 3590, profiles/r06n_*), so the
  // second runs the end of every step ALONE and each of its LDS-DMA issue stalls (~60 cycles a piece) idles the pipe; the favoured wave's stalls
  // are covered by the other's MFMAs.  Built, parity-green, and measured NEUTRAL (towers 322 vs 321 us, profiles/r06o_bwaves_ab.txt): a knob, off.
  constexpr int BW = (NW == 8 && DD3D_ROW_B_WAVES > 0 && DD3D_ROW_B_WAVES < NW) ? DD3D_ROW_B_WAVES : NW;
  constexpr int PA = (NPA + NW - 1) / NW, PB = (NPB + BW - 1) / BW;  // per wave (the surplus re-fetches the last piece)
  constexpr int ZERO_OFF = B_BASE + NSB * B_STAGE;  // 16 zero bytes invalid taps read (64 reserved)
  constexpr int EV_OFF = ZERO_OFF + 64;              // [scale | bias | lo][BN] floats of the epilogue (conv_epilogue_t)
  static_assert(NSB >= 2 && NSB <= 6 && NSA >= 2 && NSA <= 4 && EV_OFF + 12 * BN <= 160 * 1024, "LDS rings");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* lds = reinterpret_cast<unsigned char*>(smem);
  typedef unsigned char __attribute__((address_space(3))) * ldsbp;
#if DD3D_ROW_B_SADDR
  const unsigned lds_base32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(ldsbp)lds);  // LDS byte address of the rings (M0 of the scalar-base LDS-DMA)
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave - wm * WN;
#if DD3D_ROW_PRIO_SLICE
  const bool prio_wave = NW == 8 && (DD3D_ROW_PRIO_SLICE == 1 ? wave >= 4 : wave < 4);  // the waves that alternate between priority 2 and 0
  if (NW == 8 && !prio_wave) __builtin_amdgcn_s_setprio(1);
#endif
  const bool bwave = BW == NW || (DD3D_ROW_B_WAVES_HI ? wave >= NW - BW : wave < BW);  // (wave-uniform) this wave issues filter pieces
#if DD3D_ROW_STAMP
  unsigned long long st_t[8] = {stamp_now(), 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = 0;
#endif

  int bid, kslice;
  if constexpr (CHAIN) {  // launch order = dependency order; the K slices of a tile are neighbours (a 2-D grid would dispatch ALL items' slice 1 last)
    kslice = SK ? (int)(blockIdx.x % (unsigned)a.splitk) : 0;
    bid = SK ? (int)(blockIdx.x / (unsigned)a.splitk) : (int)blockIdx.x;
  } else {
    bid = remap_block(blockIdx.x, a.ntiles * a.nn);
    kslice = blockIdx.y;
  }
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

  const int nk = a.Kpad / BK;  // 9 * chunks
  int kt_begin = 0, kt_end = nk;
  if (SK) {  // (the host only picks this kernel when kt_per_split is a multiple of 3: slices start on a filter row)
    kt_begin = kslice * a.kt_per_split;
    kt_end = min(nk, kt_begin + a.kt_per_split);
  }
  const int ngroup = (kt_end - kt_begin) / 3;
  const int g_begin = kt_begin / 3;  // group index = chunk * 3 + dh

  // ---- DMA geometry.  lane -> (row = lane >> 2 of a 16-row piece, LDS slot = lane & 3 holding k-slot (lane & 3) ^ ((lane >> 4) & 3))
  const int slot16 = (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
  const long npix = (long)s.B * s.H * s.W;
  const long in_cstride = npix * (NP * 64);
  // A pieces of this wave: piece = q * NW + wave -> (row block, plane); the lane's LDS row j holds input pixel m0 - 1 + j (+ (dh-1) W)
  int a_pix[PA];   // m0 - 1 + j of this lane
  int a_dst[PA];   // wave-uniform LDS byte offset inside an A stage
  int a_pl[PA];    // wave-uniform plane
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int piece = min(q * NW + wave, NPA - 1);
    const int rb = piece / NP, pl = piece - rb * NP;
    a_pl[q] = pl;
    a_dst[q] = pl * PLA + rb * 1024;
    a_pix[q] = m0 - 1 + rb * 16 + (lane >> 2);
  }
#if DD3D_ROW_B_SADDR
  unsigned b_src[PB];  // byte offset of this lane's filter row from the filter base: K-tile 0, plane and k-slot included
#else
  gcbp b_src[PB];  // this lane's filter row, K-tile 0, plane and k-slot included
#endif
  int b_dst[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int piece = min(q * BW + (BW == NW ? wave : wave % BW), NPB - 1);
    const int rb = piece / NP, pl = piece - rb * NP;
#if DD3D_EPI_T  // LDS row R of the B stage holds filter row chan_of_row(R) of its 32-row block (conv_common.h::conv_epilogue_t)
    const int brow = rb * 16 + (lane >> 2);
    const int n = min(n0 + (brow & ~31) + chan_of_row(brow & 31), a.Npad - 1);  // rows past Npad feed columns >= N, which are never stored
#else
    const int n = min(n0 + rb * 16 + (lane >> 2), a.Npad - 1);  // rows past Npad feed columns >= N, which are never stored
#endif
    b_dst[q] = B_BASE + pl * PLB + rb * 1024;
#if DD3D_ROW_B_SADDR
    b_src[q] = (unsigned)((long)n * nk * (NP * 64) + pl * 64 + slot16);  // (launch_conv_planes_row checks that the filter stays below 4 GiB)
#else
    b_src[q] = g_w + (long)n * nk * (NP * 64) + pl * 64 + slot16;
#endif
  }

  // ---- streams: A walks the groups (chunk, dh), B walks the K-tiles; past the end both re-fetch their last element (exact DMA counts)
  int ld_g = g_begin, ld_kt = kt_begin;
  gcbp nxt_a[PA];
  auto prepare_a = [&]() {  // addresses of group ld_g, then advance
    const int chunk = ld_g / 3, dh = ld_g - chunk * 3;
    const long base = (long)chunk * in_cstride + slot16;
    const int shift = (dh - 1) * s.W;
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const long p = (long)a_pix[q] + shift;
      const bool ok = (unsigned long)p < (unsigned long)npix;
      nxt_a[q] = ok ? g_in + base + p * (NP * 64) + a_pl[q] * 64 : g_zero + slot16;
    }
    ld_g += (ld_g + 1 < g_begin + ngroup);
  };
  auto emit_a = [&](int stage) {
#pragma unroll
    for (int q = 0; q < PA; ++q)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)nxt_a[q], (ldsbp)(lds + stage * A_STAGE + a_dst[q]), 16, 0,
                                       CHAIN ? DD3D_CHAIN_A_AUX : 0);
  };
  auto emit_b = [&](int stage) {  // K-tile ld_kt, then advance
    const long koff = (long)ld_kt * (NP * 64);
#if DD3D_ROW_B_SADDR
    // global_load_lds_dwordx4 vOFFSET, s[BASE:BASE+1]: address = base + zext(lane offset); the LDS destination of the wave instruction = M0 + 16 lane.
    // (The builtin only emits the 64-bit-per-lane form.  The compiler does not see these loads: its own waits only get more conservative --
    // vector-memory loads return in order -- and the K loop's waits are explicit.)
    // (readfirstlane: the operands ARE wave-uniform; this makes the compiler keep them in scalar registers whatever it proves about them)
    const unsigned long kaddr = (unsigned long)(g_w + koff);
    const unsigned long kbase = ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(kaddr >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kaddr);
    const unsigned lds0 = lds_base32 + (unsigned)(stage * B_STAGE);
    // M0 is on the clobber list: the compiler re-materialises it before its own LDS-DMA sequences (the activation pieces).  clang warns that it
    // will not PRESERVE a reserved register across the statement -- nothing here asks it to.  (Saving and restoring M0 by hand instead is WRONG:
    // the compiler may then move its own `s_mov m0` across the unannounced writes; the convolution tests caught that form.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#pragma unroll
    for (int q = 0; q < PB; ++q)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds0 + (unsigned)b_dst[q]), "v"(b_src[q]), "s"(kbase) : "memory", "m0");
#pragma clang diagnostic pop
#else
#pragma unroll
    for (int q = 0; q < PB; ++q)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(b_src[q] + koff), (ldsbp)(lds + stage * B_STAGE + b_dst[q]), 16, 0, 0);
#endif
    ld_kt += (ld_kt + 1 < kt_end);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment addressing.  A: LDS row (wave rows + lrow + dw); B: as conv_planes.hip
  const int lrow = lane & 31;
  const int kh = lane >> 5;
  int fa_off[3][2];  // [dw][chunk]
#pragma unroll
  for (int dw = 0; dw < 3; ++dw) {
    const int row = wm * TM * 32 + lrow + dw;
    const int swz = (row >> 2) & 3;
#pragma unroll
    for (int c = 0; c < 2; ++c) fa_off[dw][c] = row * 64 + (((2 * c + kh) ^ swz) << 4);
  }
  const int swzb = (lrow >> 2) & 3;
  const int fb_off[2] = {(wn * TN * 32 + lrow) * 64 + (((0 + kh) ^ swzb) << 4), (wn * TN * 32 + lrow) * 64 + (((2 + kh) ^ swzb) << 4)};

  // ---- validity of (output pixel, tap): bit dh * 3 + dw
  unsigned vmask[TM];
  {
    const int howo = s.Ho * s.Wo;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + (wm * TM + i) * 32 + lrow;
      unsigned mk = 0;
      if (m < s.M) {
        const int b = m / howo;
        const int rr = m - b * howo;
        const int ho = rr / s.Wo;
        const int wo = rr - ho * s.Wo;
        const unsigned vrow[3] = {(unsigned)(ho > 0), 1u, (unsigned)(ho + 1 < s.H)};
        const unsigned hcol = (unsigned)(wo > 0) | 2u | ((unsigned)(wo + 1 < s.W) << 2);
        mk = (vrow[0] * hcol) | ((vrow[1] * hcol) << 3) | ((vrow[2] * hcol) << 6);
      }
      vmask[i] = mk;
    }
  }

  bf16x8 fa[2][TM][NP], fb[2][TN][NP];
  // An invalid (pixel, tap) reads 16 zero bytes kept behind the rings instead of its LDS row: one address select per fragment read,
  // computed before the read is issued (masking the loaded registers would make every MFMA phase wait for its own prefetch).
  auto read_frags = [&](int sa, int sb, int dw, int tap, auto c_c) {
    constexpr int c = decltype(c_c)::value;
    // One address register per phase (stage + lane part) and the (row block, plane) part as the read's immediate offset; an invalid
    // (pixel, tap) selects a base that the same immediate brings onto the 16 zero bytes: one v_cndmask per read (round 3: add, select, add;
    // towers 305 -> 300 us, level 3 35.4 -> 34.0 us, bench +2.5 % on one box: profiles/r04p_a_read_addressing_ab.txt).  Going further --
    // the vertical validity folded into the LDS-DMA's source select and the three dw masks kept as SGPR lane masks, no v_and / v_cmp in
    // the loop -- passed the parity suite and measured 0.7 % SLOWER (r04q_*): the compiler's loop then needs 250 registers.
    const unsigned char* pa = lds + sa * A_STAGE + fa_off[dw][c];
    const unsigned char* Bs = lds + B_BASE + sb * B_STAGE + fb_off[c];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const bool ok = (vmask[i] >> tap) & 1u;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int C = p * PLA + i * 32 * 64;  // (compile-time: the loops are unrolled; < 64 KiB, <= ZERO_OFF)
        const unsigned char* src = ok ? pa : lds + (ZERO_OFF - C);
        fa[c][i][p] = *reinterpret_cast<const bf16x8*>(src + C);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p) fb[c][j][p] = *reinterpret_cast<const bf16x8*>(Bs + p * PLB + j * 32 * 64);
  };
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

  if (tid < 4) reinterpret_cast<int*>(lds + ZERO_OFF)[tid] = 0;  // (visible to every wave after the prologue's barrier)
#if DD3D_EPI_T
  EpiVec<BN, NTHR> evv;
  epi_load_vectors<BN, NTHR>(a, s, n0, tid, evv);  // (oldest vector-memory operations of the wave: landed by the prologue's counted wait)
  if (ngroup <= 0) epi_store_vectors<BN, NTHR>(lds + EV_OFF, tid, evv);  // (an empty K slice: the split-K exchange's barriers publish them)
#endif
  if (ngroup > 0) {
    // prologue.  The loop's counted waits assume the STEADY-STATE issue order -- step t issues B(t + NSB) and, when its dw == 2, the A group
    // NSA groups ahead behind it -- so the prologue issues in the order the (virtual) steps t = -NSB .. -1 would have: A group NSA - k sits
    // behind the B of virtual step -(3 k - 2), i.e. behind B(NSB + 2 - 3 k); the groups whose virtual step lies before the window come first.
    //   NSB <= 3, NSA = 2 (the product rings):  A(0) | B(0 .. NSB-1) | A(1)
    //   NSB = 5, NSA = 3:                       A(0) | B(0) B(1) | A(1) | B(2) B(3) B(4) | A(2)    (virtual steps -4 and -1 have dw == 2)
    // Then wait for A(0) and B(0): everything issued after the later of the two may stay in flight.
    constexpr int UPFRONT = (1 <= NSA && 1 > NSB) + (2 <= NSA && 4 > NSB) + (3 <= NSA && 7 > NSB) + (4 <= NSA && 10 > NSB);  // k: 3 k - 2 > NSB
    if constexpr (CHAIN) {
      // The filter stages depend on nobody: they stream while the block waits for its producers (DD3D_CHAIN_B_FIRST, default).  The issue
      // order then is B(0 .. NSB-1) | A groups as usual: everything the loop's counted waits need is at least as OLD as in the order
      // they were derived for (the B stages only moved to the front), so the waits stay sufficient.
      const int dep = s.reserved;  // 1 + index of the segment of this launch that writes this segment's input; 0: nobody does
#if DD3D_CHAIN_B_FIRST
      if (bwave) {
#pragma unroll
        for (int d = 0; d < NSB; ++d) emit_b(d);
      }
#endif
      if (dep > 0) {
        if (tid == 0) {
          // rows this block's A stages LOAD (valid taps or not: a row loaded before its producer wrote it could leave a stale line where a
          // later block of this CU / XCD looks for it): q = m0 - 1 + j + (dh - 1) W, j = 0 .. AROWS - 1, clipped to the map
          const long npix_l = (long)s.B * s.H * s.W;
          long lo = (long)m0 - 1 - s.W, hi = (long)m0 - 1 + (AROWS - 1) + s.W;
          lo = lo < 0 ? 0 : lo;
          hi = hi > npix_l - 1 ? npix_l - 1 : hi;
          const int tb = a.seg_tile0[dep - 1];  // the producer has this segment's geometry: its m-tile of row r is tb + r / BM
          const int t0 = tb + (int)(lo / BM), t1 = tb + (int)(hi / BM);
          unsigned spins = 0;
          // newest producer first: tiles finish roughly in launch order, so when the LAST one is there the others need one look each
          for (int t = t1; t >= t0; --t) {
            while (__hip_atomic_load(a.chain_sync + 1 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.nn) {
              // back off: hundreds of parked blocks polling every half microsecond load the fabric the producers store through
              if (spins < 8) __builtin_amdgcn_s_sleep(8);
              else if (spins < 64) __builtin_amdgcn_s_sleep(32);
              else __builtin_amdgcn_s_sleep(127);
              if (++spins > (1u << 19)) {  // ~2 s: the dispatch-order assumption failed (see the kernel's header): error out, do not hang
                if (a.status) atomicOr(a.status, DD3D_STATUS_CHAIN_TIMEOUT);
                break;
              }
            }
          }
#if DD3D_CHAIN_ACQUIRE
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1: lines this CU cached in an earlier launch / replay are not served again
#endif
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    if constexpr (CHAIN && DD3D_CHAIN_B_FIRST) {
#pragma unroll
      for (int u = 0; u < NSA; ++u) {
        prepare_a();
        emit_a(u);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSA - 1) * PA) : "memory");  // A(0) and every B stage landed; the younger A groups may be in flight
    } else {
    int next_a = 0;
#pragma unroll
    for (int u = 0; u < UPFRONT; ++u) {
      prepare_a();
      emit_a(next_a++);
    }
#pragma unroll
    for (int d = 0; d < NSB; ++d) {
      if (bwave) emit_b(d);
      const int k3 = NSB + 2 - d;  // = 3 k of the A group that follows this B, if any
      if (k3 % 3 == 0 && k3 / 3 >= 1 && k3 / 3 <= NSA) {
        prepare_a();
        emit_a(next_a++);
      }
    }
    constexpr int DA0 = NSB + 2 - 3 * NSA;  // (UPFRONT == 0) A(0) follows B(DA0)
    constexpr int PRO_WAIT = UPFRONT > 0 ? (NSB - 1) * PB + (NSA - UPFRONT) * PA : (NSB - 1 - DA0) * PB + (NSA - 1) * PA;
    constexpr int PRO_WAIT_NOB = UPFRONT > 0 ? (NSA - UPFRONT) * PA : (NSA - 1) * PA;  // a wave without filter pieces: only its A groups count
    if (bwave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PRO_WAIT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PRO_WAIT_NOB) : "memory");
    }
#if DD3D_ROW_STAMP
    st_t[1] = stamp_now();
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#if DD3D_ROW_STAMP
    st_t[2] = st_last = stamp_now();
#endif
#if DD3D_EPI_T
    epi_store_vectors<BN, NTHR>(lds + EV_OFF, tid, evv);  // published by the first step's barrier
#endif
    read_frags(0, 0, 0, (g_begin % 3) * 3, C0);
    int sb = 0;
    // One step = one K-tile (filter row dh, column dw).  DMA issue order after the prologue: step s issues B(s + NSB) and, when dw == 2
    // (the group's A stage has just been read for the last time), A(group + NSA) behind it.  "B(s+1) has landed" at step s therefore
    // means: at most the B tiles s+2 .. s+NSB-1 and the A groups issued in steps s+1-NSB .. s-1 are still in flight (in the first steps
    // the prologue's extra A groups sit behind B(1) as well: the counted wait then also waits for them -- conservative, never wrong).
    // NSA A stages = the A stream runs NSA - 1 groups (3 (NSA - 1) K steps) ahead: the activations of a block are read ONCE, so every
    // A group is an HBM / MALL round trip, and the short steps of the small tiles (a few hundred cycles) do not cover one with NSA = 2.
    auto step = [&](int sa, int dh, auto dw_c) {
      constexpr int dw = decltype(dw_c)::value;
#if DD3D_ROW_PRIO_SLICE
      if constexpr (NW == 8) {  // second half of the barrier interval
        if (prio_wave) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      const int tap = dh * 3 + dw;
      // ---- phase A: chunk-1 fragments under the chunk-0 MFMAs; addresses of the next A group (used after the barrier when dw == 2)
      read_frags(sa, sb, dw, tap, C1);
      mfma_chunk(C0);
      if constexpr (dw == 2) prepare_a();
      sched_barrier_phase<TM * TN * NPROD, (TM + TN) * NP, 0>();
      // "B(s+1) has landed": B(s+1) was issued in step s+1-NSB; behind it sit the B tiles s+2 .. s+NSB-1 and the A groups issued in the steps
      // s+1-NSB .. s-1, i.e. one per step j = 1 .. NSB-1 back whose dw was 2.  When dw == 2 the next A group is read right after this barrier:
      // it was issued 3 (NSA - 1) steps ago behind that step's B, so at most the 3 (NSA - 1) - 1 B tiles and NSA - 2 A groups issued since may
      // stay in flight (a no-op tightening for the product rings).
      constexpr int a_in_flight = ((dw + 2) % 3 == 2 && NSB > 1) + ((dw + 1) % 3 == 2 && NSB > 2) + ((dw + 0) % 3 == 2 && NSB > 3) +
                                  ((dw + 2) % 3 == 2 && NSB > 4) + ((dw + 1) % 3 == 2 && NSB > 5);  // j = 1 .. 5: (dw - j) mod 3 == 2
      constexpr int steady = (NSB - 2) * PB + a_in_flight * PA;
      constexpr int a_cap = (3 * (NSA - 1) - 1) * PB + (NSA - 2) * PA;
      constexpr int wait_n = (dw == 2 && steady > a_cap) ? a_cap : steady;
      // (a wave without filter pieces, DD3D_ROW_B_WAVES: the same counts with PB = 0)
      constexpr int steady_nob = a_in_flight * PA, a_cap_nob = (NSA - 2) * PA;
      constexpr int wait_nob = (dw == 2 && steady_nob > a_cap_nob) ? a_cap_nob : steady_nob;
#if DD3D_ROW_STAMP
      const unsigned long long st_a = stamp_now();  // (its lgkmcnt(0) = the fragment reads of this phase have landed)
#endif
      if (BW == NW || bwave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(wait_n) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(wait_nob) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if DD3D_ROW_STAMP
      const unsigned long long st_b = stamp_now();
#endif
      __builtin_amdgcn_s_barrier();  // everyone: K-tile s+1 (and, after dw == 2, the next A group) landed; B stage sb / A stage sa free
      asm volatile("" ::: "memory");
#if DD3D_ROW_STAMP
      {
        const unsigned long long st_c = stamp_now();
        st_t[5] += st_a - st_last, st_t[6] += st_b - st_a, st_t[7] += st_c - st_b;
        st_last = st_c;
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#if DD3D_ROW_PRIO_SLICE
      if constexpr (NW == 8) {  // first half of the barrier interval
        if (prio_wave) __builtin_amdgcn_s_setprio(2);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      // ---- phase B: refill the freed stages; chunk-0 fragments of the next K-tile under the chunk-1 MFMAs
#if DD3D_ROW_STAMP > 1  // (2: slot [6] = the filter pieces' issue instead of the vmcnt wait)
      const unsigned long long st_d = stamp_now();
#endif
      if (BW == NW || bwave) emit_b(sb);
#if DD3D_ROW_STAMP > 1
      st_t[6] += stamp_now() - st_d - (st_b - st_a);
#endif
      if constexpr (dw == 2) emit_a(sa);
      sb = sb == NSB - 1 ? 0 : sb + 1;
      constexpr int ndw = dw == 2 ? 0 : dw + 1;
      const int nsa = dw == 2 ? (sa == NSA - 1 ? 0 : sa + 1) : sa;
      const int ndh = dw == 2 ? (dh == 2 ? 0 : dh + 1) : dh;
      read_frags(nsa, sb, ndw, ndh * 3 + ndw, C0);  // (past the end: surplus data, never used)
      mfma_chunk(C1);
      sched_barrier_phase<TM * TN * NPROD, (TM + TN) * NP, (BW == NW && !DD3D_ROW_B_SADDR ? PB : 0) + (dw == 2 ? PA : 0)>();
    };
    constexpr std::integral_constant<int, 0> D0{};
    constexpr std::integral_constant<int, 1> D1{};
    constexpr std::integral_constant<int, 2> D2{};
    int dh = g_begin % 3;
    int sa = 0;
    for (int g = 0; g < ngroup; ++g) {
      step(sa, dh, D0);
      step(sa, dh, D1);
      step(sa, dh, D2);
      dh = dh == 2 ? 0 : dh + 1;
      sa = sa == NSA - 1 ? 0 : sa + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus fetches land before the LDS is released
#if DD3D_ROW_STAMP
    st_t[3] = stamp_now();
#endif
  }

  if constexpr (SK) {
    if (!splitk_exchange<TM, TN, NTHR>(a, acc, bid, tid, kslice)) return;
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
  conv_epilogue_t<TM, TN, MODE, WM, WN, CHAIN>(a, s, acc, m0, n0, wm, wn, lane, lds + EV_OFF, scratch);
#else
  conv_epilogue_t<TM, TN, MODE, WM, WN, CHAIN>(a, s, acc, m0, n0, wm, wn, lane, lds + EV_OFF, nullptr);
#endif
#else
  static_assert(!CHAIN, "dependent segments need the transposed epilogue (write-through plane stores)");
  conv_epilogue<TM, TN, MODE, WM, WN>(a, s, acc, m0, n0, wm, wn, lane);
#endif
#if DD3D_ROW_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stores acknowledged: what the next launch waits for)
  st_t[4] = stamp_now();
  if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 512) {
#pragma unroll
    for (int i = 0; i < 8; ++i) g_row_stamps[((size_t)blockIdx.x * 8 + (wave & 7)) * 8 + i] = st_t[i];
  }
#endif
  if constexpr (CHAIN) {
    // this tile is in memory: every wave has its write-through stores acknowledged, then ONE agent-scope arrival on the m-tile's counter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ int sh_all_done;
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(a.chain_sync + 1 + mt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int total = a.ntiles * a.nn;  // tiles finished by exactly one block each (with split-K: the slice that arrives last)
      sh_all_done = __hip_atomic_fetch_add(a.chain_sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1;
    }
    __syncthreads();
    if (sh_all_done) {  // the launch's last tile: nobody polls any more -- leave the counters zero for the next launch / graph replay
      for (int i = tid; i < a.ntiles + 1; i += NTHR) __hip_atomic_store(a.chain_sync + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ host
// Ring depths of a tile (also computed by dd3d_amd/engine/tiling.py::kernel_signature for the bench's bookkeeping):
//   NSB  B stages: 3 when two A stages + three B stages fit the block's LDS budget, else 2
//   NSA  A stages: as many (<= 4) as fit -- a block that fits twice into a CU with NSA = 2 (<= 80 KiB) keeps fitting twice, a block that
//        owns its CU anyway may grow to the whole budget
template <int TM, int TN, int WM, int WN, int MODE>
struct RowRings {
  static constexpr int NP = Planes<MODE>::NP;
  static constexpr int BM = TM * 32 * WM, BN = TN * 32 * WN;
  static constexpr int AST = NP * (BM + 16) * 64, BST = NP * BN * 64;
  // (a 4-wave block whose two A stages alone exceed half a CU's LDS owns its CU anyway: it takes the 8-wave budget)
  static constexpr int BUDGET = ((WM * WN == 8 || 2 * AST > 64 * 1024) ? DD3D_ROW_LDS_KIB_8W : DD3D_ROW_LDS_KIB_4W) * 1024;
  // The product rings: two A stages; three B stages when 2 A + 3 B fit the budget, else two.  -DDD3D_ROW_NSA_MAX=3..4 / -DDD3D_ROW_NSB_MAX=4..6 let
  // the rings grow as far as LIMIT allows -- a block that fits twice into a CU with the product rings (<= 80 KiB) keeps fitting twice, a block
  // that owns its CU anyway may take the whole budget -- A first, then B.  Measured (profiles/r04i_a_ring_depth_ab.txt, r05g_b_ring_depth_ab.txt,
  // r05i_ring_depth_ab.txt): see DESIGN.md section 4, round 5; the product keeps 2 + 3.
  static constexpr int NSB0 = (2 * AST + 3 * BST <= BUDGET) ? 3 : 2;
  static constexpr int EXTRA = 64 + 12 * BN;  // the zero bytes invalid taps read + the epilogue vectors
  static constexpr int TOTAL0 = 2 * AST + NSB0 * BST + EXTRA;
  static constexpr int LIMIT = TOTAL0 <= 80 * 1024 ? 80 * 1024 : (BUDGET > TOTAL0 ? BUDGET : TOTAL0);
#ifdef DD3D_ROW_NSA_MAX
  static constexpr int NSA_MAX = DD3D_ROW_NSA_MAX;
#else
  static constexpr int NSA_MAX = 2;
#endif
#ifdef DD3D_ROW_NSB_MAX
  static constexpr int NSB_MAX = DD3D_ROW_NSB_MAX;
#else
  static constexpr int NSB_MAX = 3;
#endif
  static constexpr bool fits(int nsa, int nsb) { return nsa * AST + nsb * BST + EXTRA <= LIMIT; }
  static constexpr int pick_nsa(int n) { return (n > 2 && !fits(n, NSB0)) ? pick_nsa(n - 1) : n; }
  static constexpr int NSA = pick_nsa(NSA_MAX < 2 ? 2 : (NSA_MAX > 4 ? 4 : NSA_MAX));
  static constexpr int pick_nsb(int n) { return (n > NSB0 && !fits(NSA, n)) ? pick_nsb(n - 1) : n; }
  static constexpr int NSB = pick_nsb(NSB_MAX < NSB0 ? NSB0 : (NSB_MAX > 6 ? 6 : NSB_MAX));
  static constexpr int total(int nsa) { return nsa * AST + NSB * BST + EXTRA; }
};

template <int TM, int TN, int WM, int WN, int MODE, bool ALLOW_SK = true>
static int launch_row_tile(const ConvKArgs& ka, hipStream_t st) {
  typedef RowRings<TM, TN, WM, WN, MODE> R;
  constexpr int NTHR = 64 * WM * WN;
  constexpr int NSB = R::NSB, NSA = R::NSA;
  static_assert(R::total(NSA) <= 160 * 1024, "tile does not fit the LDS");
  const size_t lds = (size_t)R::total(NSA);
  dim3 grid(ka.ntiles * ka.nn, ka.splitk, 1);
  static unsigned long long attr_done[4];
  if (lds_opt_in_needed(attr_done)) {
    if (lds_opt_in(reinterpret_cast<const void*>(conv_igemm_planes_row_kernel<TM, TN, WM, WN, NSB, MODE, false, NSA>), (size_t)(lds), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
    if constexpr (ALLOW_SK)
      if (lds_opt_in(reinterpret_cast<const void*>(conv_igemm_planes_row_kernel<TM, TN, WM, WN, NSB, MODE, true, NSA>), (size_t)(lds), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
    lds_opt_in_done(attr_done);  // (every opt-in of this call site succeeded on this device)
  }
  if constexpr (!ALLOW_SK) DD3D_REQUIRE(ka.splitk == 1, "dd3d_conv2d_igemm_f32: this tile has no split-K form (its accumulators fill the register file)");
  if (ka.chain) {
    // dependent segments: items in launch order along x (m-tile, n-tile, K slice; see the kernel's header)
    if constexpr (NSA == 2) {
      static unsigned long long chain_attr_done[4];
      if (lds_opt_in_needed(chain_attr_done)) {
        if (lds_opt_in(reinterpret_cast<const void*>(conv_igemm_planes_row_kernel<TM, TN, WM, WN, NSB, MODE, false, NSA, true>), (size_t)(lds), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
        if constexpr (ALLOW_SK)
          if (lds_opt_in(reinterpret_cast<const void*>(conv_igemm_planes_row_kernel<TM, TN, WM, WN, NSB, MODE, true, NSA, true>), (size_t)(lds), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
        lds_opt_in_done(chain_attr_done);
      }
      const long nblk = (long)ka.ntiles * ka.nn * ka.splitk;
      DD3D_REQUIRE(nblk < (1l << 31), "dd3d_conv2d_igemm_f32: chain launch of %ld blocks", nblk);
      dim3 cgrid((unsigned)nblk, 1, 1);
      if constexpr (ALLOW_SK) {
        if (ka.splitk > 1) {
          hipLaunchKernelGGL((conv_igemm_planes_row_kernel<TM, TN, WM, WN, NSB, MODE, true, NSA, true>), cgrid, dim3(NTHR), lds, st, ka);
          return check_launch("conv_igemm_planes_row chain split-K kernel");
        }
      }
      hipLaunchKernelGGL((conv_igemm_planes_row_kernel<TM, TN, WM, WN, NSB, MODE, false, NSA, true>), cgrid, dim3(NTHR), lds, st, ka);
      return check_launch("conv_igemm_planes_row chain kernel");
    } else {
      DD3D_REQUIRE(false, "dd3d_conv2d_igemm_f32: this tile has no form for dependent segments (chain)");
    }
  }
  if constexpr (ALLOW_SK) {
    if (ka.splitk > 1) {
      hipLaunchKernelGGL((conv_igemm_planes_row_kernel<TM, TN, WM, WN, NSB, MODE, true, NSA>), grid, dim3(NTHR), lds, st, ka);
      return check_launch("launch_row_tile split-K kernel");
    }
  }
  hipLaunchKernelGGL((conv_igemm_planes_row_kernel<TM, TN, WM, WN, NSB, MODE, false, NSA>), grid, dim3(NTHR), lds, st, ka);
  return check_launch("conv_igemm_planes_row kernel");
}

template <int MODE>
static int launch_row_mode(const ConvKArgs& ka, int tile_cfg, hipStream_t st) {
  switch (tile_cfg) {
    case DD3D_TILE_256x128: return launch_row_tile<2, 2, 4, 2, MODE>(ka, st);
    case DD3D_TILE_128x128: return launch_row_tile<2, 1, 2, 4, MODE>(ka, st);
    case DD3D_TILE_128x64_K2:
    case DD3D_TILE_128x64: return launch_row_tile<1, 1, 4, 2, MODE>(ka, st);
    case DD3D_TILE_64x128_K2:
    case DD3D_TILE_64x128: return launch_row_tile<1, 1, 2, 4, MODE>(ka, st);
    case DD3D_TILE_128x128_W4: return launch_row_tile<2, 2, 2, 2, MODE>(ka, st);
    case DD3D_TILE_64x64_W4K2:
    case DD3D_TILE_64x64_W4: return launch_row_tile<1, 1, 2, 2, MODE>(ka, st);
    case DD3D_TILE_128x64_W4: return launch_row_tile<2, 1, 2, 2, MODE>(ka, st);
    case DD3D_TILE_128x32_W4: return launch_row_tile<1, 1, 4, 1, MODE>(ka, st);
    case DD3D_TILE_256x128_T42: return launch_row_tile<4, 2, 2, 2, MODE>(ka, st);
    case DD3D_TILE_128x256_T24: return launch_row_tile<2, 4, 2, 2, MODE>(ka, st);
    case DD3D_TILE_256x256_W8:
      if constexpr (Planes<MODE>::NP <= 2) return launch_row_tile<4, 2, 2, 4, MODE, false>(ka, st);
      break;
    case DD3D_TILE_192x256_W8:
      if constexpr (Planes<MODE>::NP <= 2) return launch_row_tile<3, 2, 2, 4, MODE, false>(ka, st);
      break;
  }
  DD3D_REQUIRE(false, "dd3d_conv2d_igemm_f32: tile_cfg %d has no row-shared split-plane kernel", tile_cfg);
}

// Ring depths of the instantiation a (tile, mode) pair launches -- what a profile reader needs to name the kernel (rocprofv3 prints the
// template arguments); dd3d_amd/engine/tiling.py::kernel_signature restates the rule and a CPU test compares the two for every tile and mode.
template <int MODE>
static int row_rings_mode(int tile_cfg, int* nsb, int* nsa) {
#define DD3D_RR(TM, TN, WM, WN) { typedef RowRings<TM, TN, WM, WN, MODE> R; *nsb = R::NSB; *nsa = R::NSA; return DD3D_OK; }
  switch (tile_cfg) {
    case DD3D_TILE_256x128: DD3D_RR(2, 2, 4, 2)
    case DD3D_TILE_128x128: DD3D_RR(2, 1, 2, 4)
    case DD3D_TILE_128x64_K2:
    case DD3D_TILE_128x64: DD3D_RR(1, 1, 4, 2)
    case DD3D_TILE_64x128_K2:
    case DD3D_TILE_64x128: DD3D_RR(1, 1, 2, 4)
    case DD3D_TILE_128x128_W4: DD3D_RR(2, 2, 2, 2)
    case DD3D_TILE_64x64_W4K2:
    case DD3D_TILE_64x64_W4: DD3D_RR(1, 1, 2, 2)
    case DD3D_TILE_128x64_W4: DD3D_RR(2, 1, 2, 2)
    case DD3D_TILE_128x32_W4: DD3D_RR(1, 1, 4, 1)
    case DD3D_TILE_256x128_T42: DD3D_RR(4, 2, 2, 2)
    case DD3D_TILE_128x256_T24: DD3D_RR(2, 4, 2, 2)
    case DD3D_TILE_256x256_W8:
      if constexpr (Planes<MODE>::NP <= 2) DD3D_RR(4, 2, 2, 4)
      break;
    case DD3D_TILE_192x256_W8:
      if constexpr (Planes<MODE>::NP <= 2) DD3D_RR(3, 2, 2, 4)
      break;
  }
#undef DD3D_RR
  return DD3D_E_UNSUPPORTED;
}

int conv_planes_row_rings(int math_mode, int tile_cfg, int* nsb, int* nsa) {
  switch (math_mode) {
    case DD3D_MATH_BF16X3: return row_rings_mode<DD3D_MATH_BF16X3>(tile_cfg, nsb, nsa);
    case DD3D_MATH_BF16X2: return row_rings_mode<DD3D_MATH_BF16X2>(tile_cfg, nsb, nsa);
    case DD3D_MATH_BF16: return row_rings_mode<DD3D_MATH_BF16>(tile_cfg, nsb, nsa);
    case DD3D_MATH_F16X2: return row_rings_mode<DD3D_MATH_F16X2>(tile_cfg, nsb, nsa);
  }
  return DD3D_E_UNSUPPORTED;
}

// The row-shared form applies to 3x3 / stride 1 / pad 1 convolutions whose K slices start on a filter row.
bool conv_planes_row_applicable(const ConvKArgs& ka) {
  return ka.KH == 3 && ka.KW == 3 && ka.stride == 1 && ka.pad == 1 && (ka.Cin % 32) == 0 && (ka.splitk == 1 || ka.kt_per_split % 3 == 0);
}

int launch_conv_planes_row(const ConvKArgs& ka, int math_mode, int tile_cfg, hipStream_t st) {
#if DD3D_ROW_B_SADDR
  // filter rows are addressed as base + a 32-bit byte offset per lane (at most 3 planes of Kpad halves per filter)
  DD3D_REQUIRE((long)ka.Npad * ka.Kpad * 6 < (1l << 32), "dd3d_conv2d_igemm_f32: filter of %d x %d beyond the 4 GiB the row kernel addresses", ka.Npad, ka.Kpad);
#endif
  switch (math_mode) {
    case DD3D_MATH_BF16X3: return launch_row_mode<DD3D_MATH_BF16X3>(ka, tile_cfg, st);
    case DD3D_MATH_BF16X2: return launch_row_mode<DD3D_MATH_BF16X2>(ka, tile_cfg, st);
    case DD3D_MATH_BF16: return launch_row_mode<DD3D_MATH_BF16>(ka, tile_cfg, st);
    case DD3D_MATH_F16X2: return launch_row_mode<DD3D_MATH_F16X2>(ka, tile_cfg, st);
  }
  DD3D_REQUIRE(false, "dd3d_conv2d_igemm_f32: math mode %d has no split-plane kernel", math_mode);
}

}  // namespace dd3d

#if DD3D_ROW_STAMP
// Timing probe (variant builds only, tests/gpu_row_stamp_probe.py): copies the stamps out and clears them.
extern "C" int dd3d_debug_row_stamps(unsigned long long* out, int n) {
  const size_t bytes = sizeof(unsigned long long) * (size_t)(n < 512 * 8 * 8 ? n : 512 * 8 * 8);
  if (hipDeviceSynchronize() != hipSuccess) return DD3D_E_LAUNCH;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dd3d::g_row_stamps), bytes) != hipSuccess) return DD3D_E_LAUNCH;
  static unsigned long long zeros[512 * 8 * 8];
  if (hipMemcpyToSymbol(HIP_SYMBOL(dd3d::g_row_stamps), zeros, sizeof(zeros)) != hipSuccess) return DD3D_E_LAUNCH;
  return DD3D_OK;
}
#endif
