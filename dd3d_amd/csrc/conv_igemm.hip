// Implicit-GEMM convolution for gfx950 (MI355X) on the exact-f32 matrix pipe
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise an fmaf chain).
//
// GEMM view:  M = B*Ho*Wo output pixels (rows), N = Cout, K = KH*KW*Cin.
//   A[m,k]   gathered on the fly from the NHWC input (im2col never materialised),
//   Wp[n,k]  packed filter, k contiguous,
// both staged per K-tile (BK = 32) through LDS and read back as ds_read_b128 fragments: lane l supplies row (l&31)
// and the 4 consecutive k at 8*s + 4*(l>>5), so one 16-byte LDS read feeds four MFMAs.  Each K-tile is
// (channel-chunk, filter tap), tap fastest, so the nine taps of a 3x3 filter re-read the same input lines
// back-to-back (L1/L2 hits).
//
// One launch covers many *segments* (FPN levels x head towers) through a tile table, so the tiny P6/P7 levels ride
// along with P3 instead of costing their own under-filled launches, and the per-level BatchNorm of the shared towers
// becomes a per-segment (scale, bias) epilogue.
//
// Two kernels share the tiling, the XCD-aware block remap and the epilogue:
//   conv_igemm_f32_kernel      register-staged (global -> VGPR -> padded LDS rows).  Handles every Cin (4, 16, 32k);
//                              used for the Cin < 32 stem layers and, by default, for the 128x128 head-tower tile.
//   conv_igemm_f32_dma_kernel  LDS-DMA (global_load_lds_dwordx4) into a multi-stage swizzled ring with counted vmcnt
//                              waits and raw barriers; Cin % 32 == 0 only; used for every other layer.
#include <cstring>
#include <stdlib.h>

#include "conv_common.h"

DD3D_NOTE_BUILD_FLAGS

namespace dd3d {

constexpr int LDS_ROW = BK + 4;  // register-staged kernel: 144-B rows keep the 16-B slots of 16 rows distinct mod 256 B

// ------------------------------------------------------------------------------------------------------------------
// Register-staged kernel.  Block = 256 threads = 4 wave64; block tile = (TM*32*WM) x (TN*32*WN); double-buffered LDS.
// PIPE 0: one staging register set (load kt+1 -> MFMA kt -> LDS store kt+1 -> barrier).
// PIPE 2: two staging sets, prefetch distance 2, raw barriers (global loads stay in flight across them).
template <int TM, int TN, int WM, int WN, bool SMALLC, int PIPE, bool SK>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvKArgs a) {
  constexpr int BM = TM * 32 * WM;
  constexpr int BN = TN * 32 * WN;
  constexpr int AP = BM / 32;  // A rows per thread (8 threads x 16 B cover one 32-float row)
  constexpr int BP = BN / 32;
  static_assert(WM * WN == 4, "4 waves per block");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave - wm * WN;

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
  const gcfp g_in = as_g(s.in);
  const gcfp g_w = as_g(s.w);

  const int nk = a.Kpad / BK;
  int kt_begin = 0, kt_end = nk;
  if (SK) {
    kt_begin = blockIdx.y * a.kt_per_split;
    kt_end = min(nk, kt_begin + a.kt_per_split);
  }

  // ---- per-thread gather geometry of the A rows this thread stages (fixed over the K loop)
  const int arow = tid >> 3;
  const int avec = tid & 7;
  long a_base[AP];
  int a_hi0[AP], a_wi0[AP];
  {
    const int howo = s.Ho * s.Wo;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int m = m0 + p * 32 + arow;
      if (m < s.M) {
        const int b = m / howo;
        const int r = m - b * howo;
        const int ho = r / s.Wo;
        const int wo = r - ho * s.Wo;
        a_hi0[p] = ho * a.stride - a.pad;
        a_wi0[p] = wo * a.stride - a.pad;
        a_base[p] = (((long)b * s.H + a_hi0[p]) * s.W + a_wi0[p]) * s.in_pitch;
      } else {
        a_hi0[p] = -(1 << 28);  // never inside [0, H)
        a_wi0[p] = 0;
        a_base[p] = 0;
      }
    }
  }

  // Staging registers.  All loads are unconditional (invalid taps / rows read a safe address and are zeroed at LDS-store
  // time) so the compiler can keep exact vmcnt counts instead of draining the queue at every branch.
  constexpr int NSET = PIPE == 2 ? 2 : 1;
  f32x4 ra[NSET][AP], rb[NSET][BP];
  unsigned amask[NSET];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto load_tile = [&](int kt, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    int dh, dw, tap_ok = 1;
    long koff;
    if (SMALLC) {  // Cin in {4,16}: one K-tile spans several taps -> per-thread tap
      const int k = kt * BK + avec * 4;
      const int tap = k >> a.cc_shift;
      const int c = k & ((1 << a.cc_shift) - 1);
      tap_ok = tap < a.T;
      dh = (tap * a.kw_magic) >> 16;
      dw = tap - dh * a.KW;
      koff = ((long)dh * s.W + dw) * s.in_pitch + c;
    } else {  // Cin % 32 == 0: (chunk, tap) uniform over the block
      const int chunk = kt / a.T;
      const int tap = kt - chunk * a.T;
      dh = (tap * a.kw_magic) >> 16;
      dw = tap - dh * a.KW;
      koff = ((long)dh * s.W + dw) * s.in_pitch + chunk * BK + avec * 4;
    }
    unsigned mask = 0;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool ok = tap_ok && (unsigned)(a_hi0[p] + dh) < (unsigned)s.H && (unsigned)(a_wi0[p] + dw) < (unsigned)s.W;
      mask |= (unsigned)ok << p;
      ra[S][p] = *(gcf4p)(g_in + (ok ? a_base[p] + koff : (long)(avec * 4)));
    }
    amask[S] = mask;
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int n = min(n0 + p * 32 + arow, a.Npad - 1);  // rows past Npad feed columns >= N, which are never stored
      rb[S][p] = *(gcf4p)(g_w + (long)n * a.Kpad + kt * BK + avec * 4);
    }
  };

  auto store_tile = [&](int buf, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    float* As = smem + buf * (BM + BN) * LDS_ROW;
    float* Bs = As + BM * LDS_ROW;
#pragma unroll
    for (int p = 0; p < AP; ++p)
      *reinterpret_cast<f32x4*>(As + (p * 32 + arow) * LDS_ROW + avec * 4) = ((amask[S] >> p) & 1u) ? ra[S][p] : zero4;
#pragma unroll
    for (int p = 0; p < BP; ++p) *reinterpret_cast<f32x4*>(Bs + (p * 32 + arow) * LDS_ROW + avec * 4) = rb[S][p];
  };
  constexpr std::integral_constant<int, 0> SET0{};
  constexpr std::integral_constant<int, NSET - 1> SET1{};
  // barrier that leaves global loads in flight: LDS traffic of this wave done, then s_barrier (no vmcnt drain)
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lrow = lane & 31;
  const int lk = (lane >> 5) * 4;

  auto compute_tile = [&](int cur) {
    const float* As = smem + cur * (BM + BN) * LDS_ROW + (wm * TM * 32 + lrow) * LDS_ROW + lk;
    const float* Bs = smem + cur * (BM + BN) * LDS_ROW + BM * LDS_ROW + (wn * TN * 32 + lrow) * LDS_ROW + lk;
#pragma unroll
    for (int s4 = 0; s4 < BK / 8; ++s4) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDS_ROW + s4 * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDS_ROW + s4 * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    }
  };

  if (kt_begin < kt_end) {
    if (PIPE == 0) {
      load_tile(kt_begin, SET0);
      store_tile(0, SET0);
      __syncthreads();
      for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool more = kt + 1 < kt_end;
        if (more) load_tile(kt + 1, SET0);  // global -> registers, in flight under the MFMAs below
        compute_tile(cur);
        if (more) store_tile(cur ^ 1, SET0);
        __syncthreads();
      }
    } else {
      // Every load / LDS store below is unconditional (indices are clamped to the last K-tile, the surplus data lands
      // in a buffer nobody reads) so that the compiler keeps exact vmcnt counts.
      const int kt_last = kt_end - 1;
      load_tile(kt_begin, SET0);
      load_tile(min(kt_begin + 1, kt_last), SET1);
      store_tile(0, SET0);
      load_tile(min(kt_begin + 2, kt_last), SET0);
      lds_barrier();
      for (int kt = kt_begin; kt < kt_end; kt += 2) {
        store_tile(1, SET1);
        load_tile(min(kt + 3, kt_last), SET1);
        compute_tile(0);
        lds_barrier();
        if (kt + 1 >= kt_end) break;
        store_tile(0, SET0);
        load_tile(min(kt + 4, kt_last), SET0);
        compute_tile(1);
        lds_barrier();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus prefetches must not outlive the staging registers
    }
  }

  if constexpr (SK) {  // separate instantiation: the exchange must not cost the plain kernel registers
    if (!splitk_exchange<TM, TN>(a, acc, bid, tid, blockIdx.y)) return;
  }
  conv_epilogue<TM, TN>(a, s, acc, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA kernel (Cin % 32 == 0): the K-tiles are streamed L2/HBM -> LDS with global_load_lds_dwordx4 (no staging VGPRs,
// no ds_write pass) into a ring of NS stages; U tiles are consumed per raw barrier, NS-U tiles are in flight, and the
// wait before the barrier is a COUNTED vmcnt (never a drain inside the loop).
//
// LDS image of a stage: rows of 32 floats (128 B, unpadded because the DMA writes lane-linear: one wave instruction =
// 8 rows x 128 B = 1 KiB), 16-byte slots XOR-swizzled with (row & 7): slot s of row r holds k-chunk s ^ (r & 7).  The
// permutation is applied on the per-lane SOURCE address of the DMA and again on the ds_read_b128 address (same
// involution).  Out-of-image taps and rows >= M read a zero page.
//
// WK = 1: 4 waves, each owns a (TM*32)x(TN*32) output sub-tile.  WK = 2: 8 waves; waves w and w+4 own the SAME sub-tile
// and split every K-tile's four k-steps between them (partial sums merged through LDS at the end), so that the
// small-tile layers (<= 256 blocks, one block per CU) still put two waves on every SIMD.
template <int TM, int TN, int WM, int WN, int NS, int U, int WK, bool SK>
__global__ __launch_bounds__(256 * WK) void conv_igemm_f32_dma_kernel(const ConvKArgs a) {
  constexpr int BM = TM * 32 * WM;
  constexpr int BN = TN * 32 * WN;
  constexpr int STAGE = (BM + BN) * BK;  // floats per stage
  constexpr int NW = 4 * WK;             // waves per block
  constexpr int PA = BM / (8 * NW);      // 1-KiB A pieces (8 rows x 128 B) this wave issues per K-tile
  constexpr int PB = BN / (8 * NW);
  constexpr int P = PA + PB;
  constexpr int D = NS - U;              // prefetch distance in K-tiles
  constexpr int STEPS = (BK / 8) / WK;   // k-steps of a K-tile this wave computes
  static_assert(WM * WN == 4 && NS >= 2 * U && PA >= 1 && PB >= 1, "tile / ring shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef float __attribute__((address_space(3))) * ldsfp;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave & 3;   // which output sub-tile
  const int kg = wave >> 2;  // which half of the k-steps (WK == 2)
  const int wm = wq / WN;
  const int wn = wq - wm * WN;

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
  const gcfp g_in = as_g(s.in);
  const gcfp g_w = as_g(s.w);
  const gcfp g_zero = as_g(a.zeros);

  const int nk = a.Kpad / BK;
  int kt_begin = 0, kt_end = nk;
  if (SK) {
    kt_begin = blockIdx.y * a.kt_per_split;
    kt_end = min(nk, kt_begin + a.kt_per_split);
  }

  // ---- DMA geometry: piece q of this wave = tile rows (q*NW + wave)*8 .. +8; lane -> (row = lane>>3, LDS slot = lane&7)
  const int prow = lane >> 3;
  const int srcchunk = (lane & 7) ^ prow;  // k-chunk this lane fetches (source-side swizzle)
  long a_base[PA];
  int a_hi0[PA], a_wi0[PA];
  gcfp b_src[PB];
  {
    const int howo = s.Ho * s.Wo;
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int m = m0 + (q * NW + wave) * 8 + prow;
      if (m < s.M) {
        const int b = m / howo;
        const int r = m - b * howo;
        const int ho = r / s.Wo;
        const int wo = r - ho * s.Wo;
        a_hi0[q] = ho * a.stride - a.pad;
        a_wi0[q] = wo * a.stride - a.pad;
        a_base[q] = (((long)b * s.H + a_hi0[q]) * s.W + a_wi0[q]) * s.in_pitch + srcchunk * 4;
      } else {
        a_hi0[q] = -(1 << 28);
        a_wi0[q] = 0;
        a_base[q] = 0;
      }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int n = min(n0 + (q * NW + wave) * 8 + prow, a.Npad - 1);
      b_src[q] = g_w + (long)n * a.Kpad + srcchunk * 4;
    }
  }

  // One 1-KiB DMA piece: p < PA -> A rows, else B rows.  (dh, dw, koff) describe the K-tile being fetched.
  auto issue_piece = [&](int p, int kt, int dh, int dw, long koff, float* st) {
    if (p < PA) {
      const bool ok = (unsigned)(a_hi0[p] + dh) < (unsigned)s.H && (unsigned)(a_wi0[p] + dw) < (unsigned)s.W;
      const gcfp src = ok ? g_in + a_base[p] + koff : g_zero + (lane & 7) * 4;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (ldsfp)(st + p * NW * 8 * BK), 16, 0, 0);
    } else {
      const int q = p - PA;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(b_src[q] + kt * BK),
                                       (ldsfp)(st + BM * BK + q * NW * 8 * BK), 16, 0, 0);
    }
  };
  auto issue_tile = [&](int kt, int stage) {
    const int chunk = kt / a.T;
    const int tap = kt - chunk * a.T;
    const int dh = (tap * a.kw_magic) >> 16;
    const int dw = tap - dh * a.KW;
    const long koff = ((long)dh * s.W + dw) * s.in_pitch + chunk * BK;
    float* st = smem + stage * STAGE + wave * 8 * BK;
#pragma unroll
    for (int p = 0; p < P; ++p) issue_piece(p, kt, dh, dw, koff, st);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lrow = lane & 31;
  const int kh = lane >> 5;
  int slot_off[STEPS];  // float offset of this wave's k-step t inside the lane's row (read-side swizzle)
#pragma unroll
  for (int t = 0; t < STEPS; ++t) slot_off[t] = ((2 * (kg * STEPS + t) + kh) ^ (lrow & 7)) * 4;

  // MFMAs of the tile in `stage`, with this wave's DMA pieces of tile `kt_next` (-> stage `fill`) issued right behind
  // an MFMA so that their issue cost hides under the matrix pipe instead of delaying it.
  auto tile_step = [&](int stage, int kt_next, int fill) {
    const float* As = smem + stage * STAGE + (wm * TM * 32 + lrow) * BK;
    const float* Bs = smem + stage * STAGE + BM * BK + (wn * TN * 32 + lrow) * BK;
    const int chunk = kt_next / a.T;
    const int tap = kt_next - chunk * a.T;
    const int dh = (tap * a.kw_magic) >> 16;
    const int dw = tap - dh * a.KW;
    const long koff = ((long)dh * s.W + dw) * s.in_pitch + chunk * BK;
    float* st = smem + fill * STAGE + wave * 8 * BK;
    f32x4 af[2][TM], bf[2][TN];  // fragments double-buffered: the LDS reads of k-step t+1 fly under the MFMAs of t
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * BK + slot_off[0]);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * BK + slot_off[0]);
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
      if (t + 1 < STEPS) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(t + 1) & 1][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * BK + slot_off[t + 1]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[(t + 1) & 1][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * BK + slot_off[t + 1]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t & 1][i][e], bf[t & 1][j][e], acc[i][j], 0, 0, 0);
        if (e == 0) {
#pragma unroll
          for (int p = (t * P) / STEPS; p < ((t + 1) * P) / STEPS; ++p) issue_piece(p, kt_next, dh, dw, koff, st);
        }
      }
    }
  };

  if (kt_begin < kt_end) {
    const int kt_last = kt_end - 1;
#pragma unroll
    for (int d = 0; d < D; ++d) issue_tile(min(kt_begin + d, kt_last), d);
    int base = 0;  // ring stage of tile kt
    for (int kt = kt_begin; kt < kt_end; kt += U) {
      // this wave's pieces of tiles kt .. kt+U-1 have landed once at most D-U newer tiles are still in flight
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * (D - U)) : "memory");
      __builtin_amdgcn_s_barrier();  // everyone's pieces landed; everyone is done with the U stages read last time
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int stage = base + u;
        stage = stage >= NS ? stage - NS : stage;
        int fill = base + u + D;  // == stage of tile kt + u - U, freed by the barrier above
        fill = fill >= NS ? fill - NS : fill;
        // every iteration issues exactly U tiles of DMA (clamped past the end) so the counted wait stays exact
        if (kt + u < kt_end) tile_step(stage, min(kt + D + u, kt_last), fill);
        else issue_tile(kt_last, fill);
      }
      base += U;
      base = base >= NS ? base - NS : base;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus prefetches must land before the LDS is re-used / released
  }

  if (WK == 2) {
    // merge the two K-halves: waves 4..7 park their accumulators in LDS ([sub-tile][block][reg][lane]), waves 0..3 add them
    __syncthreads();  // every wave is done reading the ring and no DMA is in flight
    float* red = smem + (wq * TM * TN) * 16 * 64 + lane;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((i * TN + j) * 16 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * TN + j) * 16 + r) * 64];
  }

  // (WK == 2: the kg == 1 waves have exited, tid < 256 here)
  if constexpr (SK) {  // separate instantiation: the exchange must not cost the plain kernel registers
    if (!splitk_exchange<TM, TN>(a, acc, bid, tid, blockIdx.y)) return;
  }
  conv_epilogue<TM, TN>(a, s, acc, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// Split-operand kernel (Cin % 32 == 0): f32-equivalent products on the BF16 matrix pipe.
//
// Every f32 operand is split EXACTLY into three bf16 terms, x = hi + mid + lo (8 + 8 + 8 significand bits, by truncation),
// and a*b is accumulated in f32 from the six largest cross products (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi);
// the three dropped ones are <= 2^-24 |a*b|, i.e. at the level of one f32 rounding.  v_mfma_f32_32x32x16_bf16 retires
// 16 k-values in 32 cycles where v_mfma_f32_32x32x2_f32 needs 8 x 64, so six of them cost 192 vs 512 cycles per 16 k.
//
// Filters are split once at plan build (Wp3[n][k-tile][plane][32] bf16) and streamed HBM/L2 -> LDS by LDS-DMA into a
// 3-stage ring; activations stay f32 in HBM (no format change for any other kernel): each thread loads 16-byte pieces two
// K-tiles ahead into registers, splits them (5.5 VALU / element, issued in the MFMA shadow) and writes the three planes to
// a double-buffered LDS image.  LDS rows are 64 B (32 bf16) per plane; their four 16-byte slots are XOR-swizzled with
// (row >> 2) & 3, which makes every ds_read_b128 lane group hit 16 distinct 4-bank groups.
// Block = 512 threads = 8 wave64 (two per SIMD), wave tile (TM*32) x (TN*32), WM x WN waves.

template <int TM, int TN, int WM, int WN, int NSA, int NSB, int KT, bool SK>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_bf16x3_kernel(const ConvKArgs a) {
  constexpr int BM = TM * 32 * WM;
  constexpr int BN = TN * 32 * WN;
  constexpr int NW = WM * WN;                // waves per block: 8 (two per SIMD) or 4 (then two blocks share a CU)
  constexpr int NTHR = 64 * NW;
  // A step (one barrier) covers KT consecutive K-tiles of 32 k; KT = 2 halves the barriers / LDS round trips per MFMA of the
  // small tiles, whose 12 MFMAs per wave and K-tile are short next to that fixed cost.
  constexpr int AP1 = BM * 8 / NTHR;        // 16-byte f32 pieces of one K-tile's A rows per thread
  constexpr int AP = KT * AP1;              // ... of a step
  constexpr int PLA = BM * 64;              // bytes per A plane
  constexpr int PLB = BN * 64;              // bytes per B plane
  constexpr int A_SUB = 3 * PLA, B_SUB = 3 * PLB;  // one K-tile's image (3 planes)
  constexpr int A_STAGE = KT * A_SUB, B_STAGE = KT * B_SUB;
  constexpr int NPIECE1 = B_SUB / 1024;     // 1-KiB DMA pieces per K-tile of B
  constexpr int NPIECE = KT * NPIECE1;
  constexpr int PB = (NPIECE + NW - 1) / NW;  // pieces per wave and step (the surplus re-fetches an existing piece)
  // VMEM issue order: A0 B0 A1 B1 A2 [B2] | then per step kt: A(kt+3) before the barrier, B(kt+NSB) after it.  Returns are in
  // order, so "A(kt+1) has landed" (needed at the START of step kt, where its split is interleaved with the MFMAs) is
  // vmcnt <= AP + 2*PB, and "B(kt+1) has landed" (needed before the barrier that ends step kt) is vmcnt <= WAIT_B.
  constexpr int WAIT_A = AP + 2 * PB;
  constexpr int WAIT_B = NSB == 3 ? 2 * AP + PB : AP;
  static_assert((NW == 8 || NW == 4) && AP1 >= 1 && (NSB == 2 || NSB == 3) && (NSA == 1 || NSA == 2) && (KT == 1 || KT == 2), "block shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* lds = reinterpret_cast<unsigned char*>(smem);  // [A stage 0 .. NSA-1 | B stage 0 .. NSB-1]
  typedef unsigned char __attribute__((address_space(3))) * ldsbp;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave - wm * WN;

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
  const gcfp g_in = as_g(s.in);
  const gcfp g_zero = as_g(a.zeros);
  const unsigned char __attribute__((address_space(1)))* g_w3 = (const unsigned char __attribute__((address_space(1)))*)s.w;

  const int nk = a.Kpad / BK;
  int kt_begin = 0, kt_end = nk;
  if (SK) {
    kt_begin = blockIdx.y * a.kt_per_split;
    kt_end = min(nk, kt_begin + a.kt_per_split);
  }

  // ---- A gather geometry: thread -> (row = tid >> 3 (+64 per piece), f32 quad = tid & 7)
  const int arow = tid >> 3;
  const int avec = tid & 7;
  long a_base[AP1];
  int a_hi0[AP1], a_wi0[AP1];
  {
    const int howo = s.Ho * s.Wo;
#pragma unroll
    for (int p = 0; p < AP1; ++p) {
      const int m = m0 + p * (NTHR / 8) + arow;
      if (m < s.M) {
        const int b = m / howo;
        const int r = m - b * howo;
        const int ho = r / s.Wo;
        const int wo = r - ho * s.Wo;
        a_hi0[p] = ho * a.stride - a.pad;
        a_wi0[p] = wo * a.stride - a.pad;
        a_base[p] = (((long)b * s.H + a_hi0[p]) * s.W + a_wi0[p]) * s.in_pitch + avec * 4;
      } else {
        a_hi0[p] = -(1 << 28);
        a_wi0[p] = 0;
        a_base[p] = 0;
      }
    }
  }
  // LDS byte offset (inside a plane) of this thread's 8-byte half slot; piece p adds p * NTHR/8 rows (the swizzle term only
  // depends on (row >> 2) & 3, which 32- or 64-row steps leave unchanged)
  const int a_st0 = arow * 64 + (((avec >> 1) ^ ((arow >> 2) & 3)) << 4) + ((avec & 1) << 3);

  // ---- B DMA geometry: piece = q*8 + wave (mod NPIECE) -> (plane, 16-row block); lane -> (row = lane >> 2, LDS slot = lane & 3)
  long b_src[PB];
  int b_dst[PB], b_sub[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    int piece = q * NW + wave;
    piece = piece >= NPIECE ? piece - NPIECE : piece;
    const int sub = piece / NPIECE1;  // which K-tile of the step
    const int pc = piece - sub * NPIECE1;
    const int plane = pc / (BN / 16);
    const int rb = pc - plane * (BN / 16);
    const int row = rb * 16 + (lane >> 2);
    const int n = min(n0 + row, a.Npad - 1);
    const int slot = (lane & 3) ^ ((lane >> 4) & 3);  // source-side swizzle: LDS slot (lane & 3) of row holds k-slot `slot`
    b_src[q] = ((long)n * nk * 3 + plane) * 64 + slot * 16;
    b_dst[q] = sub * B_SUB + plane * PLB + rb * 1024;  // + lane * 16 implied by the DMA (lane-linear)
    b_sub[q] = sub;
  }

  // The A stream walks the K-tiles in order; (chunk, tap) are carried instead of divided out of kt.  K-tiles past the end of
  // this block's slice read the zero page: the odd tail of a KT = 2 step must contribute nothing.
  int ld_kt = kt_begin;
  int ld_chunk = kt_begin / a.T;
  int ld_tap = kt_begin - ld_chunk * a.T;
  const int nst = (kt_end - kt_begin + KT - 1) / KT;  // steps
  const int st_last = nst - 1;

  f32x4 ra[2][AP];  // two register sets, always indexed by a compile-time constant (runtime indexing would serialise the loads)
  auto load_a = [&](auto set_c) {  // the next KT K-tiles of the stream (always AP loads, so the counted waits stay exact)
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const bool live = ld_kt < kt_end;
      const int dh = (ld_tap * a.kw_magic) >> 16;
      const int dw = ld_tap - dh * a.KW;
      const long koff = ((long)dh * s.W + dw) * s.in_pitch + ld_chunk * BK;
#pragma unroll
      for (int p = 0; p < AP1; ++p) {
        const bool ok = live && (unsigned)(a_hi0[p] + dh) < (unsigned)s.H && (unsigned)(a_wi0[p] + dw) < (unsigned)s.W;
        const gcfp src = ok ? g_in + a_base[p] + koff : g_zero + avec * 4;
        // issued behind the compiler's back: its waitcnt pass drains vmcnt to 0 whenever register loads and LDS-DMA are both
        // pending, which would expose the whole memory latency every other step.  wait_a() below is the matching wait.
        f32x4& dst = ra[set][u * AP1 + p];
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(src) : "memory");
      }
      if (live) {
        ++ld_kt;
        if (++ld_tap == a.T) ld_tap = 0, ++ld_chunk;
      }
    }
  };
  auto wait_a = [&](auto set_c, auto cnt_c) {
    constexpr int set = decltype(set_c)::value;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(cnt_c)::value) : "memory");
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      f32x4& r = ra[set][p];
      asm volatile("" : "+v"(r));  // the split below depends on the wait above
    }
  };
  auto issue_b = [&](int st, int stage) {  // B of step st (K-tiles kt_begin + st*KT ..; clamped inside the filter: the A side is zero there)
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int kt = min(kt_begin + st * KT + b_sub[q], nk - 1);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g_w3 + b_src[q] + (long)kt * 192),
                                       (ldsbp)(lds + NSA * A_STAGE + stage * B_STAGE + b_dst[q]), 16, 0, 0);
    }
  };
  // split four f32 (piece p of register set `set`) into the three bf16 planes (exact, by truncation) and store them
  auto split_store_piece = [&](auto set_c, int p, int stage) {
    constexpr int set = decltype(set_c)::value;
    unsigned char* As = lds + stage * A_STAGE + (p / AP1) * A_SUB + a_st0 + (p % AP1) * (NTHR * 8);
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = a.in_relu ? fmaxf(ra[set][p][e], 0.f) : ra[set][p][e];
      h[e] = __float_as_uint(x) & 0xffff0000u;
      const float r = x - __uint_as_float(h[e]);
      m[e] = __float_as_uint(r) & 0xffff0000u;
      l[e] = __float_as_uint(r - __uint_as_float(m[e]));
    }
    *reinterpret_cast<u32x2*>(As) = u32x2{__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
    *reinterpret_cast<u32x2*>(As + PLA) = u32x2{__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
    *reinterpret_cast<u32x2*>(As + 2 * PLA) = u32x2{__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
  };
  auto split_store = [&](auto set_c, int stage) {
    wait_a(set_c, std::integral_constant<int, WAIT_A>{});
#pragma unroll
    for (int p = 0; p < AP; ++p) split_store_piece(set_c, p, stage);
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

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

  // MFMAs of the tile in (A stage sa, B stage sb).  With two A stages the split + store of the NEXT tile's A pieces (register
  // set `set`, into A stage `next`) is spread between the MFMA groups, so the VALU work issues in the matrix pipe's shadow
  // instead of after it (both waves of a SIMD leave the barrier together: doing all MFMAs, then all VALU, idles the pipe).
  auto compute_tile = [&](int sa, int sb, auto set_c, int next) {
    // smallest terms first; the (i, j) loops are innermost so consecutive MFMAs hit different accumulators
    constexpr int PA_[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB_[6] = {0, 2, 1, 0, 1, 0};
    constexpr int GSTRIDE = 12 / AP1 > 0 ? 12 / AP1 : 1;
    static_assert(AP1 <= 12, "one A piece per MFMA group at most");
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const unsigned char* As = lds + sa * A_STAGE + u * A_SUB + wm * TM * 32 * 64;
      const unsigned char* Bs = lds + NSA * A_STAGE + sb * B_STAGE + u * B_SUB + wn * TN * 32 * 64;
      bf16x8 af[2][TM][3], bf[2][TN][3];  // fragments of both k-chunks: the reads of chunk 1 fly under the MFMAs of chunk 0
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) af[c][i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PLA + i * 32 * 64 + frag_off[c]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) bf[c][j][pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * PLB + j * 32 * 64 + frag_off[c]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 6; ++t) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[c][i][PA_[t]], bf[c][j][PB_[t]], acc[i][j], 0, 0, 0);
          if constexpr (NSA == 2) {
            const int grp = c * 6 + t;
            if (grp % GSTRIDE == 0 && grp / GSTRIDE < AP1) split_store_piece(set_c, u * AP1 + grp / GSTRIDE, next);
          }
        }
    }
  };

  if (kt_begin < kt_end) {
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    // prologue: A(0) B(0) A(1) B(1) | store A(0) | A(2) | barrier | [3 stages: B(2)]
    load_a(S0);
    issue_b(0, 0);
    load_a(S1);
    issue_b(min(1, st_last), 1);
    wait_a(S0, std::integral_constant<int, (NSB == 3 ? AP + PB : 0)>{});  // A(0), B(0) (2 stages: everything) landed
#pragma unroll
    for (int p = 0; p < AP; ++p) split_store_piece(S0, p, 0);
    load_a(S0);
    lds_barrier();
    if (NSB == 3) issue_b(min(2, st_last), 2);
    int sb = 0;
    // Step kt: [A(kt+1) landed] MFMAs of tile kt with the split + store of A(kt+1) in their shadow | A(kt+3) -> the register set
    // just stored | [B(kt+1) landed] barrier | B refill of the ring stage just consumed.  (One A stage: MFMAs | barrier | split +
    // store | ...)  Exactly AP + PB VMEM operations per step (indices clamped past the end).
    auto step = [&](int st_cur, int sa, auto set_c, int next) {
      if constexpr (NSA == 2) {
        wait_a(set_c, std::integral_constant<int, WAIT_A>{});
        compute_tile(sa, sb, set_c, next);
      } else {
        compute_tile(sa, sb, set_c, next);
        lds_barrier();
        split_store(set_c, next);
      }
      load_a(set_c);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_B) : "memory");
      lds_barrier();
      issue_b(min(st_cur + NSB, st_last), sb);
      sb = sb == NSB - 1 ? 0 : sb + 1;
    };
    for (int st = 0; st < nst; st += 2) {
      step(st, 0, S1, NSA - 1);
      if (st + 1 >= nst) break;
      step(st + 1, NSA - 1, S0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus prefetches must land before the LDS is released
  }

  if constexpr (SK) {
    if (!splitk_exchange<TM, TN, NTHR>(a, acc, bid, tid, blockIdx.y)) return;
  }
  conv_epilogue<TM, TN, DD3D_MATH_BF16X3, WM, WN>(a, s, acc, m0, n0, wm, wn, lane);  // f32 NHWC and / or split planes for the next conv
}

// ------------------------------------------------------------------------------------------------------------------ host
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <int TM, int TN, int WM, int WN, bool SMALLC, int PIPE, bool SK>
static void launch_reg_sk(const ConvKArgs& ka, dim3 grid, hipStream_t st) {
  constexpr int BM = TM * 32 * WM, BN = TN * 32 * WN;
  const size_t lds = (size_t)2 * (BM + BN) * LDS_ROW * sizeof(float);
  auto k = conv_igemm_f32_kernel<TM, TN, WM, WN, SMALLC, PIPE, SK>;
  static unsigned long long attr_done[4];
  if (lds_opt_in_needed(attr_done)) {
    if (lds_opt_in(reinterpret_cast<const void*>(k), lds, "conv_igemm_f32 kernel") == DD3D_OK) lds_opt_in_done(attr_done);  // (a refusal is retried by the next launch; this one fails with the launch error below)
  }
  hipLaunchKernelGGL(k, grid, dim3(256, 1, 1), lds, st, ka);
}

template <int TM, int TN, int WM, int WN, bool SMALLC, int PIPE>
static void launch_reg(const ConvKArgs& ka, dim3 grid, hipStream_t st) {
  if (ka.splitk > 1) launch_reg_sk<TM, TN, WM, WN, SMALLC, PIPE, true>(ka, grid, st);
  else launch_reg_sk<TM, TN, WM, WN, SMALLC, PIPE, false>(ka, grid, st);
}

template <int TM, int TN, int WM, int WN, int NS, int U, int WK, bool SK>
static void launch_dma_sk(const ConvKArgs& ka, dim3 grid, hipStream_t st) {
  constexpr int BM = TM * 32 * WM, BN = TN * 32 * WN;
  const size_t lds = (size_t)NS * (BM + BN) * BK * sizeof(float);
  auto k = conv_igemm_f32_dma_kernel<TM, TN, WM, WN, NS, U, WK, SK>;
  static unsigned long long attr_done[4];
  if (lds_opt_in_needed(attr_done)) {
    if (lds_opt_in(reinterpret_cast<const void*>(k), lds, "conv_igemm_f32 kernel") == DD3D_OK) lds_opt_in_done(attr_done);  // (a refusal is retried by the next launch; this one fails with the launch error below)
  }
  hipLaunchKernelGGL(k, grid, dim3(256 * WK, 1, 1), lds, st, ka);
}

template <int TM, int TN, int WM, int WN, int NS, int U, int WK>
static void launch_dma(const ConvKArgs& ka, dim3 grid, hipStream_t st) {
  if (ka.splitk > 1) launch_dma_sk<TM, TN, WM, WN, NS, U, WK, true>(ka, grid, st);
  else launch_dma_sk<TM, TN, WM, WN, NS, U, WK, false>(ka, grid, st);
}

template <int TM, int TN, int WM, int WN, int NSA, int NSB, int KT = 1>
static int launch_x3(const ConvKArgs& ka, hipStream_t st) {
  constexpr int BM = TM * 32 * WM, BN = TN * 32 * WN, NTHR = 64 * WM * WN;
  const size_t lds = (size_t)KT * ((size_t)NSA * 3 * BM * 64 + (size_t)NSB * 3 * BN * 64);
  dim3 grid(ka.ntiles * ka.nn, ka.splitk, 1);
  static unsigned long long attr_done[4];
  if (lds_opt_in_needed(attr_done)) {
    if (lds_opt_in(reinterpret_cast<const void*>(conv_igemm_bf16x3_kernel<TM, TN, WM, WN, NSA, NSB, KT, false>), (size_t)(lds), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
    if (lds_opt_in(reinterpret_cast<const void*>(conv_igemm_bf16x3_kernel<TM, TN, WM, WN, NSA, NSB, KT, true>), (size_t)(lds), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
    lds_opt_in_done(attr_done);  // (every opt-in of this call site succeeded on this device)
  }
  if (ka.splitk > 1) hipLaunchKernelGGL((conv_igemm_bf16x3_kernel<TM, TN, WM, WN, NSA, NSB, KT, true>), grid, dim3(NTHR), lds, st, ka);
  else hipLaunchKernelGGL((conv_igemm_bf16x3_kernel<TM, TN, WM, WN, NSA, NSB, KT, false>), grid, dim3(NTHR), lds, st, ka);
  return check_launch("conv_igemm_bf16x3 kernel");
}

template <int TM, int TN, int WM, int WN>
static int launch_cfg(const ConvKArgs& ka, bool smallc, hipStream_t st) {
  constexpr int BM = TM * 32 * WM, BN = TN * 32 * WN;
  dim3 grid(ka.ntiles * ka.nn, ka.splitk, 1);
  // A/B knobs; the defaults are what measured best on MI355X (profiles/).
  static const int use_dma = env_int("DD3D_CONV_DMA", 1);      // LDS-DMA kernel for Cin % 32 == 0
  static const int dma_big = env_int("DD3D_CONV_DMA_BIG", 0);  // ... also for the 128x128 tile (register-staged is faster)
  static const int wk2 = env_int("DD3D_CONV_DMA_WK2", 1);      // 8 waves (K split inside the block) for the small tiles
  static const int pipe2 = env_int("DD3D_CONV_PIPE", 2);       // register-staged kernel: 2 staging sets unless 0
  if (!smallc && use_dma && ka.zeros != nullptr && (TM * TN != 4 || dma_big)) {
    if constexpr (TM * TN == 4) {
      launch_dma<TM, TN, WM, WN, 2, 1, 1>(ka, grid, st);
    } else if constexpr (BM >= 64 && BN >= 64) {
      if (wk2) launch_dma<TM, TN, WM, WN, 6, 2, 2>(ka, grid, st);
      else launch_dma<TM, TN, WM, WN, 6, 2, 1>(ka, grid, st);
    } else {
      launch_dma<TM, TN, WM, WN, 6, 2, 1>(ka, grid, st);
    }
  } else {
    // the 128x128 tile keeps one staging set (two would push it past 256 VGPRs and halve its occupancy)
    const int pipe = (TM * TN == 4) ? 0 : pipe2;
    if (smallc) {
      if (pipe == 0) launch_reg<TM, TN, WM, WN, true, 0>(ka, grid, st);
      else launch_reg<TM, TN, WM, WN, true, 2>(ka, grid, st);
    } else {
      if (pipe == 0) launch_reg<TM, TN, WM, WN, false, 0>(ka, grid, st);
      else launch_reg<TM, TN, WM, WN, false, 2>(ka, grid, st);
    }
  }
  return check_launch("conv_igemm_f32 kernel");
}

}  // namespace dd3d

extern "C" int dd3d_math_planes(int32_t math_mode) {
  return math_mode == DD3D_MATH_BF16X3 ? 3 : ((math_mode == DD3D_MATH_BF16X2 || math_mode == DD3D_MATH_F16X2) ? 2 : (math_mode == DD3D_MATH_BF16 ? 1 : 0));
}

extern "C" int dd3d_conv_tile_shape(int32_t tile_cfg, int32_t* bm, int32_t* bn) {
  static const int shapes[DD3D_TILE_COUNT][2] = {{128, 128}, {128, 64}, {64, 64}, {128, 32}, {64, 128}, {256, 128}, {128, 128}, {64, 64}, {128, 64}, {128, 64}, {64, 128}, {64, 64}, {256, 128}, {128, 256}, {256, 256}, {128, 32}, {192, 256}};
  DD3D_REQUIRE(tile_cfg >= 0 && tile_cfg < DD3D_TILE_COUNT, "dd3d_conv_tile_shape: unknown tile_cfg %d", tile_cfg);
  *bm = shapes[tile_cfg][0];
  *bn = shapes[tile_cfg][1];
  return DD3D_OK;
}

extern "C" int dd3d_conv_row_rings(int32_t tile_cfg, int32_t math_mode, int32_t* nsb, int32_t* nsa) {
  using namespace dd3d;
  DD3D_REQUIRE(nsb && nsa, "dd3d_conv_row_rings: null output");
  int b = 0, a = 0;
  const int rc = conv_planes_row_rings(math_mode, tile_cfg, &b, &a);
  if (rc != DD3D_OK) {
    set_error("dd3d_conv_row_rings: tile_cfg %d has no row-shared split-plane kernel in math mode %d", tile_cfg, math_mode);
    return rc;
  }
  *nsb = b, *nsa = a;
  return DD3D_OK;
}

extern "C" int dd3d_conv2d_igemm_f32(const dd3d_conv_launch* L, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(L && L->segs && L->tiles, "dd3d_conv2d_igemm_f32: null descriptor");
  DD3D_REQUIRE(L->ntiles > 0 && L->nsegs > 0, "dd3d_conv2d_igemm_f32: empty launch");
  DD3D_REQUIRE(L->Cin == 4 || L->Cin == 16 || (L->Cin % 32) == 0, "dd3d_conv2d_igemm_f32: Cin=%d must be 4, 16 or a multiple of 32", L->Cin);
  DD3D_REQUIRE(L->KH >= 1 && L->KW >= 1 && L->KH * L->KW <= 63, "dd3d_conv2d_igemm_f32: filter %dx%d unsupported", L->KH, L->KW);
  DD3D_REQUIRE(L->Kpad % 32 == 0 && L->Kpad >= L->KH * L->KW * L->Cin, "dd3d_conv2d_igemm_f32: Kpad=%d invalid", L->Kpad);
  DD3D_REQUIRE(L->Npad % 32 == 0 && L->Npad >= L->N && L->N > 0, "dd3d_conv2d_igemm_f32: Npad=%d / N=%d invalid", L->Npad, L->N);
  DD3D_REQUIRE(L->splitk >= 1, "dd3d_conv2d_igemm_f32: splitk=%d", L->splitk);
  DD3D_REQUIRE(L->splitk == 1 || (L->workspace && L->tile_counters),
               "dd3d_conv2d_igemm_f32: split-K needs a workspace of splitk*ntiles*ceil(N/BN)*BM*BN floats and zeroed tile counters");
  int bm, bn;
  if (dd3d_conv_tile_shape(L->tile_cfg, &bm, &bn) != DD3D_OK) return DD3D_E_INVALID;

  ConvKArgs ka;
  ka.segs = L->segs;
  ka.tiles = L->tiles;
  ka.ws = L->workspace;
  ka.ntiles = L->ntiles;
  ka.nn = ceil_div(L->N, bn);
  ka.KH = L->KH, ka.KW = L->KW, ka.stride = L->stride, ka.pad = L->pad;
  ka.Cin = L->Cin, ka.N = L->N, ka.Kpad = L->Kpad, ka.Npad = L->Npad;
  ka.T = L->KH * L->KW;
  ka.cc_shift = L->Cin == 4 ? 2 : (L->Cin == 16 ? 4 : 5);
  ka.kw_magic = 65536 / L->KW + 1;
  ka.relu = L->relu;
  ka.splitk = L->splitk;
  ka.zeros = L->zero_page;
  ka.tile_counters = L->tile_counters;
  ka.in_relu = L->in_relu;
  DD3D_REQUIRE(!L->in_relu || L->math_mode == DD3D_MATH_BF16X3, "dd3d_conv2d_igemm_f32: in_relu needs DD3D_MATH_BF16X3");
  ka.single = (L->nsegs == 1 && L->seg0_host != nullptr);
  if (ka.single) ka.seg0 = *L->seg0_host;
  else memset(&ka.seg0, 0, sizeof(ka.seg0));
  const int nk = L->Kpad / 32;
  ka.kt_per_split = ceil_div(nk, L->splitk);
  const bool smallc = L->Cin < 32;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DD3D_REQUIRE(L->math_mode >= DD3D_MATH_F32 && L->math_mode <= DD3D_MATH_F16X2, "dd3d_conv2d_igemm_f32: unknown math mode %d", L->math_mode);
  ka.out_plane_scale = L->out_plane_scale > 0.f ? L->out_plane_scale : 1.f;
  ka.status = L->status;
  ka.amax = L->amax;
  ka.chain = L->chain ? 1 : 0;
  ka.chain_sync = L->chain_sync;
  ka.seg_tile0 = L->chain_tile0;
  if (L->chain) {
    DD3D_REQUIRE(L->chain_sync && L->chain_tile0 && L->seg0_host && L->nsegs >= 2 && L->in_planes && L->KH == 3 && L->KW == 3 && L->stride == 1 && L->pad == 1,
                 "dd3d_conv2d_igemm_f32: a chain launch needs chain_sync, chain_tile0, seg0_host, >= 2 segments and 3x3 / stride 1 / pad 1 on split-plane input");
    for (int i = 0; i < L->nsegs; ++i) {
      const dd3d_conv_seg& sg = L->seg0_host[i];
      DD3D_REQUIRE(sg.Ho == sg.H && sg.Wo == sg.W && sg.M == sg.B * sg.H * sg.W, "dd3d_conv2d_igemm_f32: chain segment %d is not a same-size convolution", i);
      DD3D_REQUIRE(sg.out == nullptr && sg.out_planes != nullptr && (sg.res_mode == 0 || sg.res_mode == 2) && sg.n_limit == 0,
                   "dd3d_conv2d_igemm_f32: chain segment %d: split-plane outputs only, res_mode 0 or 2, no n_limit", i);
      DD3D_REQUIRE(sg.reserved >= 0 && sg.reserved <= i, "dd3d_conv2d_igemm_f32: chain segment %d names producer %d (must be an earlier segment)", i,
                   sg.reserved - 1);
      if (sg.reserved > 0) {
        const dd3d_conv_seg& pr = L->seg0_host[sg.reserved - 1];
        DD3D_REQUIRE(pr.B == sg.B && pr.H == sg.H && pr.W == sg.W && pr.out_planes == sg.in_planes,
                     "dd3d_conv2d_igemm_f32: chain segment %d does not read what its producer %d writes (geometry / in_planes)", i, sg.reserved - 1);
      }
    }
  }
  if (L->seg0_host != nullptr) {  // a host copy of the descriptors: every segment's residual contract is checked (without one the caller vouches)
    for (int i = 0; i < L->nsegs; ++i) {
      const dd3d_conv_seg& sg = L->seg0_host[i];
      DD3D_REQUIRE(sg.res_mode >= 0 && sg.res_mode <= 3 && (sg.res_mode == 0 || sg.res), "dd3d_conv2d_igemm_f32: segment %d: res_mode=%d / null residual", i,
                   sg.res_mode);
      DD3D_REQUIRE(sg.res_mode <= 1 || L->in_planes, "dd3d_conv2d_igemm_f32: segment %d: a split-plane residual (res_mode %d) needs the split-plane-input kernels",
                   i, sg.res_mode);
      DD3D_REQUIRE(sg.res_mode != 3 || ((sg.Ho % 2) == 0 && (sg.Wo % 2) == 0), "dd3d_conv2d_igemm_f32: segment %d: res_mode 3 needs even Ho, Wo (%d x %d)", i,
                   sg.Ho, sg.Wo);
      DD3D_REQUIRE(sg.res_mode == 0 || !L->in_planes || (L->tile_cfg != DD3D_TILE_256x256_W8 && L->tile_cfg != DD3D_TILE_192x256_W8),
                   "dd3d_conv2d_igemm_f32: segment %d: the 8-wave 256-column tiles carry no residual", i);
      const int stored = sg.n_limit > 0 ? sg.n_limit : L->N;
      DD3D_REQUIRE(sg.res_mode != 1 || !L->in_planes || ((sg.res_pitch % 4) == 0 && sg.res_pitch >= ((stored + 3) & ~3)),
                   "dd3d_conv2d_igemm_f32: segment %d: an f32 residual of the split-plane kernels is read in 16-byte pieces: res_pitch=%d must be a "
                   "multiple of 4 and cover %d channels rounded up to 4", i, sg.res_pitch, stored);
    }
  }
  if (L->in_planes) {
    DD3D_REQUIRE(L->math_mode != DD3D_MATH_F32 && !smallc && L->zero_page && !L->in_relu,
                 "dd3d_conv2d_igemm_f32: split-plane input needs a split-operand math mode, Cin %% 32 == 0, a zero page and no in_relu");
    // 3x3 / stride 1: the three taps of a filter row share one A stage (conv_planes_row.hip); DD3D_CONV_ROW=0 keeps the per-tap gather
    const int use_row = env_int("DD3D_CONV_ROW", 1);  // (read per call: the tests drive both kernels in one process)
    if (use_row && conv_planes_row_applicable(ka)) return launch_conv_planes_row(ka, L->math_mode, L->tile_cfg, st);
    DD3D_REQUIRE(!L->chain, "dd3d_conv2d_igemm_f32: a chain launch runs on the row-shared kernels only (K slices must start on a filter row; DD3D_CONV_ROW=0 disables them)");
    return launch_conv_planes(ka, L->math_mode, L->tile_cfg, st);
  }
  DD3D_REQUIRE(L->math_mode == DD3D_MATH_F32 || L->math_mode == DD3D_MATH_BF16X3,
               "dd3d_conv2d_igemm_f32: math mode %d reads split-plane input only (dd3d_split_planes converts f32 tensors)", L->math_mode);
  if (L->math_mode == DD3D_MATH_BF16X3) {
    DD3D_REQUIRE(!smallc && L->zero_page, "dd3d_conv2d_igemm_f32: the split-bf16 kernel needs Cin %% 32 == 0 and a zero page");
    switch (L->tile_cfg) {
      // 8-wave blocks, one per CU
      case DD3D_TILE_128x128: return launch_x3<2, 1, 2, 4, 2, 3>(ka, st);
      case DD3D_TILE_128x64: return launch_x3<1, 1, 4, 2, 2, 3>(ka, st);
      case DD3D_TILE_64x128: return launch_x3<1, 1, 2, 4, 2, 3>(ka, st);
      case DD3D_TILE_256x128: return launch_x3<2, 2, 4, 2, 2, 2>(ka, st);
      // 4-wave blocks sized so that two (or more) share a CU and hide each other's barriers
      case DD3D_TILE_128x128_W4: return launch_x3<2, 2, 2, 2, 1, 2>(ka, st);  // 24 + 48 = 72 KiB
      case DD3D_TILE_64x64_W4: return launch_x3<1, 1, 2, 2, 2, 3>(ka, st);    // 24 + 36 = 60 KiB
      case DD3D_TILE_128x64_W4: return launch_x3<2, 1, 2, 2, 1, 3>(ka, st);   // 24 + 36 = 60 KiB
      // two K-tiles per barrier
      case DD3D_TILE_128x64_K2: return launch_x3<1, 1, 4, 2, 2, 2, 2>(ka, st);    // 96 + 48 = 144 KiB
      case DD3D_TILE_64x128_K2: return launch_x3<1, 1, 2, 4, 2, 2, 2>(ka, st);    // 48 + 96 = 144 KiB
      case DD3D_TILE_64x64_W4K2: return launch_x3<1, 1, 2, 2, 1, 2, 2>(ka, st);   // 24 + 48 = 72 KiB, two blocks per CU
    }
    DD3D_REQUIRE(false, "dd3d_conv2d_igemm_f32: tile_cfg %d has no split-bf16 kernel", L->tile_cfg);
  }
  switch (L->tile_cfg) {
    case DD3D_TILE_128x128: return launch_cfg<2, 2, 2, 2>(ka, smallc, st);
    case DD3D_TILE_128x64: return launch_cfg<2, 1, 2, 2>(ka, smallc, st);
    case DD3D_TILE_64x64: return launch_cfg<1, 1, 2, 2>(ka, smallc, st);
    case DD3D_TILE_128x32: return launch_cfg<1, 1, 4, 1>(ka, smallc, st);
    case DD3D_TILE_64x128: return launch_cfg<1, 2, 2, 2>(ka, smallc, st);
    default: DD3D_REQUIRE(false, "dd3d_conv2d_igemm_f32: tile_cfg %d exists for DD3D_MATH_BF16X3 only", L->tile_cfg);
  }
  return DD3D_E_INVALID;
}
