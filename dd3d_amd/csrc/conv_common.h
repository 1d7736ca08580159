// Device code shared by the implicit-GEMM convolution kernels of libdd3d_hip.so (gfx950): kernel arguments, the XCD-aware block
// remap, the fused split-K exchange and the epilogue (f32 NHWC store and / or the split-plane store the next convolution streams
// into LDS by LDS-DMA).
#pragma once
#include <type_traits>

#include "common.h"

namespace dd3d {

constexpr int BK = 32;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct ConvKArgs {
  const dd3d_conv_seg* segs;
  const int32_t* tiles;
  float* ws;
  int ntiles, nn;  // m-tiles (all segments), n-tiles
  int KH, KW, stride, pad, Cin, N, Kpad, Npad;
  int T;         // KH*KW
  int cc_shift;  // log2(CC) when Cin < 32
  int kw_magic;  // (65536 / KW) + 1 : tap / KW == (tap * kw_magic) >> 16 for tap < 64
  int relu, splitk, kt_per_split;
  const float* zeros;  // >= 128 B of zeros (source of padded taps for the LDS-DMA kernels)
  int* tile_counters;  // split-K: arrivals per output tile (zero between launches)
  // Single-segment launches (every backbone / FPN conv) carry their descriptor in the kernel arguments: the block then
  // starts its first data loads without the tiles[] -> segs[] -> pointer chain of dependent scalar loads.
  int single;
  int in_relu;  // bf16x3 kernel with f32 input: rectify the input while it is split (conv(relu(x)) without a rectified copy of x)
  float out_plane_scale;  // F16X2: split-plane outputs hold value * this (power of two)
  int* status;            // OR-ed with DD3D_STATUS_* bits, or null
  float* amax;            // F16X2: atomic max of |stored plane value| (scaled), or null
  // Dependent segments in ONE launch (dd3d_conv_launch.chain; conv_planes_row.hip, CHAIN instantiations): chain_sync[0] counts the
  // blocks that have finished a tile, chain_sync[1 + mt] the finished n-tiles of m-tile mt (index into tiles[]); all zero between
  // launches (the last finisher clears them).  seg_tile0[i] = index in tiles[] of segment i's first m-tile (device).
  int chain;
  int* chain_sync;
  const int* seg_tile0;
  dd3d_conv_seg seg0;
};

// Split-operand arithmetic modes (values of dd3d_conv_launch.math_mode) and their plane counts.
//   X3: x = hi + mid + lo, three bf16 terms by truncation (exact, 24 bits); six cross products  -> f32-equivalent
//   X2: x ~ hi + lo, two bf16 terms by round-to-nearest (~17 bits); three cross products
//   X1: x ~ bf16(x) round-to-nearest; one product (plain bf16 inference)
//   F16X2: x * S = hi + lo, two IEEE half terms by round-to-nearest (22+ bits: f32-equivalent inside the half range); three products
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int MODE>
struct Planes {
  static constexpr int NP = MODE == DD3D_MATH_BF16X3 ? 3 : ((MODE == DD3D_MATH_BF16X2 || MODE == DD3D_MATH_F16X2) ? 2 : 1);
  static constexpr bool F16 = MODE == DD3D_MATH_F16X2;
};

// Global-address-space views: the segment descriptor is loaded from memory, so without these casts the compiler
// must emit FLAT loads, which tick lgkmcnt as well and would make every LDS wait also wait for the HBM prefetch.
typedef const float __attribute__((address_space(1))) * gcfp;
typedef float __attribute__((address_space(1))) * gfp;
typedef const f32x4 __attribute__((address_space(1))) * gcf4p;
typedef const unsigned char __attribute__((address_space(1))) * gcbp;
typedef unsigned char __attribute__((address_space(1))) * gbp;
__device__ __forceinline__ gcfp as_g(const float* p) { return (gcfp)p; }
__device__ __forceinline__ gfp as_g(float* p) { return (gfp)p; }

// XCD-aware block remap (bijective): blocks dispatched to the same XCD (bid % 8) get a contiguous range of logical
// tiles, n fastest, so the n-tiles that share an A row band hit the same (private, per-XCD) L2.
__device__ __forceinline__ int remap_block(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Epilogue shared by the kernels.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
//   out = max(lo, acc*scale + bias (+ residual));  one lane owns one output channel per 32-wide column block.
// Split-K fix-up, fused into the conv kernel: every K-slice of an output tile stores its raw accumulators to the workspace
// and counts its arrival; the slice that arrives LAST re-reads all slices in slice order (deterministic, independent of the
// arrival order) into its accumulators and runs the normal epilogue.  No second launch; the counter resets itself.
//
// Coherence across the 8 XCD-private L2s WITHOUT agent-scope fences (on gfx950 a release/acquire fence writes back /
// invalidates the whole L2: measured +30 us per launch): the partial sums are moved with sc1 (agent-coherent: write-through /
// L2-bypassing) 16-byte accesses and ordered by s_waitcnt only.  Workspace layout = accumulator layout,
// ws[slice][tile][quad q of (i,j)][thread][4]: every store / load instruction of a wave covers 1 KiB contiguous.
__device__ __forceinline__ void st_sc1(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 ld_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// returns true when this block has to finish the tile (acc then holds the full sum)
template <int TM, int TN, int NT = 256>
__device__ __forceinline__ bool splitk_exchange(const ConvKArgs& a, f32x16 (&acc)[TM][TN], int tile_id, int tid, int kslice) {
  constexpr int Q = TM * TN * 4;  // 16-byte quads per thread
  constexpr int QS = NT * 4;      // floats between consecutive quads of one thread
  const long slab = (long)a.ntiles * a.nn * Q * QS;  // floats per slice
  float* mine = a.ws + (long)kslice * slab + ((long)tile_id * Q * NT + tid) * 4;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        st_sc1(mine + ((i * TN + j) * 4 + q) * QS, f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial sums have been acknowledged by the coherence point
  __shared__ int sh_last;
  __syncthreads();
  if (tid == 0) {
    const int prev = __hip_atomic_fetch_add(a.tile_counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == a.splitk - 1;
    if (last) __hip_atomic_store(a.tile_counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all slices have arrived
    sh_last = last;
  }
  __syncthreads();
  if (!sh_last) return false;
  constexpr int ZC = Q >= 16 ? 1 : (Q >= 8 ? 2 : 4);  // slices in flight: 64 VGPRs of loads
  const float* base = a.ws + ((long)tile_id * Q * NT + tid) * 4;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int z0 = 0; z0 < a.splitk; z0 += ZC) {
    f32x4 t[ZC][Q];
#pragma unroll
    for (int zz = 0; zz < ZC; ++zz) {
      const int z = min(z0 + zz, a.splitk - 1);  // clamped re-read of the last slice; its value is not added
#pragma unroll
      for (int q = 0; q < Q; ++q) t[zz][q] = ld_sc1(base + (long)z * slab + q * QS);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int zz = 0; zz < ZC; ++zz)
#pragma unroll
      for (int q = 0; q < Q; ++q) asm volatile("" : "+v"(t[zz][q]));  // uses below depend on the wait above
#pragma unroll
    for (int zz = 0; zz < ZC; ++zz) {
      if (z0 + zz >= a.splitk) break;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += t[zz][(i * TN + j) * 4 + q][e];
    }
  }
  return true;
}

// ---- split of f32 values into the 16-bit planes of an arithmetic mode.  `pack(lo_elem, hi_elem, w)`: w[p] = plane p of the
// two values as one dword (lo_elem in bits 0-15: the element with the lower channel index).
template <int MODE>
__device__ __forceinline__ void split_pack(float x0, float x1, unsigned (&w)[Planes<MODE>::NP]) {
  if constexpr (MODE == DD3D_MATH_F16X2) {  // (the caller has applied the plane scale)
    const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(f32x2{x0 - hf[0], x1 - hf[1]}, f16x2);
    w[0] = __builtin_bit_cast(unsigned, h);
    w[1] = __builtin_bit_cast(unsigned, l);
  } else if constexpr (MODE == DD3D_MATH_BF16X3) {  // exact, by truncation (the same split the f32-input kernel applies on the fly)
    const unsigned h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
    const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
    const unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
    const unsigned l0 = __float_as_uint(r0 - __uint_as_float(m0)), l1 = __float_as_uint(r1 - __uint_as_float(m1));
    w[0] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
    w[1] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    w[2] = __builtin_amdgcn_perm(l1, l0, 0x07060302u);
  } else {  // round-to-nearest-even terms (v_cvt_pk_bf16_f32)
    const bf16x2 h = __builtin_convertvector(f32x2{x0, x1}, bf16x2);
    w[0] = __builtin_bit_cast(unsigned, h);
    if constexpr (MODE == DD3D_MATH_BF16X2) {
      const f32x2 hf = __builtin_convertvector(h, f32x2);
      const bf16x2 l = __builtin_convertvector(f32x2{x0 - hf[0], x1 - hf[1]}, bf16x2);
      w[1] = __builtin_bit_cast(unsigned, l);
    }
  }
}

// MODE == DD3D_MATH_F32: f32 NHWC store only (the kernel's math mode has no planes).
// WM x WN: the block's wave grid (only the range-guard sample needs it; 1 x 1 = every wave is "the" wave of its block).
template <int TM, int TN, int MODE = DD3D_MATH_F32, int WM = 1, int WN = 1>
__device__ __forceinline__ void conv_epilogue(const ConvKArgs& a, const dd3d_conv_seg& s, const f32x16 (&acc)[TM][TN], int m0, int n0,
                                              int wm, int wn, int lane) {
  const gcfp g_res = as_g(s.res);
  const gfp g_out = as_g(s.out);
  const int nlim = s.n_limit > 0 ? s.n_limit : a.N;
  if constexpr (MODE == DD3D_MATH_F32) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
      if (n >= nlim) continue;
      const float sc = as_g(s.scale)[n], bi = as_g(s.bias)[n];
      float lo = s.lo ? as_g(s.lo)[n] : -INFINITY;
      if (a.relu) lo = fmaxf(lo, 0.f);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
        float rv[16];  // residuals first, all 16 loads in flight together (they must not queue behind the stores)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          rv[r] = (s.res_mode == 1 && m < s.M) ? g_res[(long)m * s.res_pitch + n] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (m < s.M) g_out[(long)m * s.out_pitch + n] = fmaxf(acc[i][j][r] * sc + bi + rv[r], lo);
        }
      }
    }
  } else {
    // Split-plane output [chunk = n / 32][pixel m][plane][32] (16-bit terms): what the NEXT convolution streams into LDS by LDS-DMA
    // without touching the VALU.  A lane owns ONE channel and 16 pixels of a 32x32 accumulator block; lanes l and l^1 swap one value per
    // pixel pair (DPP quad_perm) so that every lane stores a (channel n, n+1) dword for one of the two pixels: each store
    // instruction of the wave then writes 4 pixels x 64 contiguous bytes.  The f32 NHWC copy is written as well when the segment has
    // one (residual sources, inputs of the pooling / top-down / gating kernels).
    constexpr int NP = Planes<MODE>::NP;
    const bool w32 = s.out != nullptr, wpl = s.out_planes != nullptr;
    const long cstride = (long)s.M * (NP * 64);  // bytes per 32-channel chunk image of the output
    const int odd = lane & 1;
    const float pscale = a.out_plane_scale;
    int ovf = 0;
    float amx = 0.f;  // F16X2: largest |scaled value| this lane stores as planes (tracked by the block's reporting wave only, see below)
    // The reporting wave of the block: chosen from the TILE's coordinates (not the block index: with split-K the block that finishes a tile
    // is whichever slice arrives last), rotating over the wave grid, and moved to wave row / column 0 when the chosen one holds only rows
    // >= M or columns >= N (row m0 and column n0 of a tile are always real) -- wave-uniform.
    const int seed = m0 / (TM * 32 * WM) + n0 / (TN * 32 * WN);
    int wm_sel = seed % WM, wn_sel = (seed / WM) % WN;
    if (m0 + wm_sel * TM * 32 >= s.M) wm_sel = 0;
    if (n0 + wn_sel * TN * 32 >= nlim) wn_sel = 0;
    const bool report = Planes<MODE>::F16 && a.amax != nullptr && wm == wm_sel && wn == wn_sel;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = n0 + (wn * TN + j) * 32;  // wave-uniform: first channel of this column block
      if (nb >= nlim) continue;
      const int n = nb + (lane & 31);
      const bool nv = n < nlim;
      const int nc = nv ? n : nlim - 1;  // clamped index for the per-channel vectors
      const float sc = as_g(s.scale)[nc], bi = as_g(s.bias)[nc];
      float lo = s.lo ? as_g(s.lo)[nc] : -INFINITY;
      if (a.relu) lo = fmaxf(lo, 0.f);
      const gbp pl_base = (gbp)s.out_planes + (long)(nb >> 5) * cstride + ((lane & 30) << 1);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          v[r] = (s.res_mode == 1 && m < s.M && nv) ? g_res[(long)m * s.res_pitch + n] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = fmaxf(acc[i][j][r] * sc + bi + v[r], lo);
          v[r] = nv ? v[r] : 0.f;  // channels past N inside the last chunk are zero planes
        }
        if (w32 && nv) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            if (m < s.M) g_out[(long)m * s.out_pitch + n] = v[r];
          }
        }
        if (wpl) {
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const int ra = 2 * t, rb = 2 * t + 1;
            const float send = odd ? v[ra] : v[rb];
            const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, false));
            const float own = odd ? v[rb] : v[ra];
            const int m = mb + (odd ? (rb & 3) + 8 * (rb >> 2) : (ra & 3) + 8 * (ra >> 2));
            unsigned w[NP];
            float e0 = odd ? recv : own, e1 = odd ? own : recv;
            if constexpr (Planes<MODE>::F16) {
              e0 *= pscale, e1 *= pscale;
              ovf |= !(fabsf(e0) <= 65504.f) | !(fabsf(e1) <= 65504.f);
              if (report && m < s.M) amx = fmaxf(amx, fmaxf(fabsf(e0), fabsf(e1)));
            }
            split_pack<MODE>(e0, e1, w);
            if (m < s.M) {
              const gbp dst = pl_base + (long)m * (NP * 64);
#pragma unroll
              for (int p = 0; p < NP; ++p) *(unsigned __attribute__((address_space(1)))*)(dst + p * 64) = w[p];
            }
          }
        }
      }
    }
    if constexpr (Planes<MODE>::F16) {
      if (ovf && a.status) atomicOr(a.status, DD3D_STATUS_F16_OVERFLOW);  // (NaN / inf inputs trip it as well)
      // Underflow side of the range guard: a SAMPLE of the stored outputs -- one wave tile per block, the wave rotating with the tile's
      // coordinates so that every (row, column) sub-tile position is covered -- is folded into one of 16 per-launch maxima (128 B apart:
      // different L2 lines).  Every wave of every block reporting into ONE address cost 16 % of the whole forward (2016 same-address
      // atomics at the end of a 100 us launch, measured: profiles/r03g_amax_ab.txt); the sample is a lower bound of the true maximum,
      // so a tensor that passes the guard on it passes on all of its entries.
      if (report) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) amx = fmaxf(amx, __shfl_xor(amx, d, 64));
        if (lane == 0 && amx > 0.f) atomicMax(reinterpret_cast<unsigned*>(a.amax + (seed & 15) * 32), __float_as_uint(amx));
      }
    }
  }
}

// ---- Epilogue of the split-plane kernels: TRANSPOSED accumulators.
// The split-plane kernels issue their MFMAs with the FILTER fragment as the first operand (D = W . A^T): a lane of a 32x32 accumulator
// block then owns ONE output pixel (column = lane & 31) and 16 output channels (rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  The filter
// rows of every 32-row block are laid into LDS in the order chan_of_row() -- the LDS-DMA's per-lane source address, nothing else -- so
// that those 16 channels are CONSECUTIVE: half-wave h = lane >> 5 owns channels 16 h .. 16 h + 15 of the block, register r = channel
// 16 h + r.  Every memory instruction of the epilogue is then 16 bytes per lane on consecutive channels of one pixel:
//   f32 NHWC store 4 x dwordx4 per block (the pixel-per-register form: 16 x dword), split-plane store 2 x dwordx4 per plane (16 x dword
//   + 8 DPP swaps for two planes), residual 4 x dwordx4 (16 x dword).
// The epilogue of a short-K convolution is bound by the NUMBER of memory instructions its CU must issue (~70-80 cycles each whatever
// their width, MI355X_MICROARCH.md "store-ISSUE-bound"): DLA level 2 (K = 576) spent 96 of them per wave against a 5.8 us K loop.
// The per-channel vectors (scale, bias, lower clamp with the ReLU folded in) of the block's BN columns are staged in LDS by the
// kernel's prologue (epi_stage_vectors: global loads issued before the first LDS-DMA, written to LDS after the prologue's barrier), so
// the epilogue reads them with ds_read_b128 -- no vector-memory load sits between the stores of consecutive blocks.
// Residual sources (dd3d_conv_seg.res_mode): 1 f32 NHWC same pixel; 2 split planes, same pixel; 3 split planes of the map at HALF the
// resolution (nearest-neighbour x2 upsampling fused into the add: the FPN top-down path).
__device__ __forceinline__ int chan_of_row(int rho) { return ((rho >> 2) & 1) * 16 + (rho >> 3) * 4 + (rho & 3); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// agent-coherent 16-byte accesses of the in-launch hand-over between dependent segments (same forms as the split-K exchange above)
__device__ __forceinline__ void st_sc1_u4(void* p, u32x4 v) {
  // (s_nop 1: a VMEM store of more than 8 bytes must be followed by two wait states before a VALU instruction overwrites its data registers
  // -- gfx940+; the compiler inserts them behind stores it knows, it cannot see inside this statement.  Without them the first dwords of a
  // 16-byte unit were now and then the NEXT unit's address arithmetic: profiles/r06_chain_bringup.txt)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 ld_sc1_u4(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
typedef const u32x4 __attribute__((address_space(1))) * gcu4p;
typedef u32x4 __attribute__((address_space(1))) * gu4p;
typedef f32x4 __attribute__((address_space(1))) * gf4p;

// Prologue half 1: this thread's share of [scale | bias | lo] for columns n0 .. n0 + BN - 1 (clamped to the last real channel).
template <int BN, int NTHR>
struct EpiVec {
  static constexpr int CNT = (3 * BN + NTHR - 1) / NTHR;
  float v[CNT];
};
template <int BN, int NTHR>
__device__ __forceinline__ void epi_load_vectors(const ConvKArgs& a, const dd3d_conv_seg& s, int n0, int tid, EpiVec<BN, NTHR>& e) {
  const int nlim = s.n_limit > 0 ? s.n_limit : a.N;
#pragma unroll
  for (int k = 0; k < EpiVec<BN, NTHR>::CNT; ++k) {
    const int idx = tid + k * NTHR;
    const int which = idx / BN, c = idx - which * BN;
    const int n = min(n0 + c, nlim - 1);
    float x = 0.f;
    if (which == 0) x = as_g(s.scale)[n];
    else if (which == 1) x = as_g(s.bias)[n];
    else if (which == 2) {
      x = s.lo ? as_g(s.lo)[n] : -INFINITY;
      if (a.relu) x = fmaxf(x, 0.f);
    }
    e.v[k] = x;
  }
}
// Prologue half 2 (after the first barrier; any later barrier publishes the values): LDS [3][BN] floats at `dst`.
template <int BN, int NTHR>
__device__ __forceinline__ void epi_store_vectors(unsigned char* dst, int tid, const EpiVec<BN, NTHR>& e) {
#pragma unroll
  for (int k = 0; k < EpiVec<BN, NTHR>::CNT; ++k) {
    const int idx = tid + k * NTHR;
    if (idx < 3 * BN) reinterpret_cast<float*>(dst)[idx] = e.v[k];
  }
}

template <int MODE>
__device__ __forceinline__ void unpack_terms(const u32x4 (&w)[Planes<MODE>::NP][2], float (&x)[16]) {
  constexpr int NP = Planes<MODE>::NP;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float lo_e = 0.f, hi_e = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) {  // largest term first: (hi + mid) + lo
        const unsigned d = w[p][q][e];
        if constexpr (Planes<MODE>::F16) {
          const f16x2 hh = __builtin_bit_cast(f16x2, d);
          lo_e += (float)hh[0], hi_e += (float)hh[1];
        } else {
          lo_e += __uint_as_float(d << 16), hi_e += __uint_as_float(d & 0xffff0000u);
        }
      }
      x[8 * q + 2 * e] = lo_e, x[8 * q + 2 * e + 1] = hi_e;
    }
}

// The 8-wave tile whose waves hold 8 accumulator blocks (256 x 256) has 256 registers per lane: no room for a residual in flight.
template <int TM, int TN, int WM, int WN>
constexpr bool epi_residual_ok() { return !(TM * TN >= 8 && WM * WN >= 8); }

// CHAIN: the launch holds dependent segments (ConvKArgs.chain) -- plane stores are write-through (sc1) and a split-plane residual is read
// with sc1 loads, so that blocks on other XCDs (their L2s are not coherent with this one's) see / read what memory holds.
#ifndef DD3D_CHAIN_RES_SC1
#define DD3D_CHAIN_RES_SC1 1  // 1: a chain launch reads split-plane residuals with sc1 loads (0: default policy; the consumer block's acquire covers them)
#endif
template <int TM, int TN, int MODE, int WM, int WN, bool CHAIN = false>
__device__ __forceinline__ void conv_epilogue_t(const ConvKArgs& a, const dd3d_conv_seg& s, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm,
                                                int wn, int lane, const unsigned char* evec, unsigned char* scratch) {
  constexpr int NP = Planes<MODE>::NP;
  constexpr int BN = TN * 32 * WN;
  constexpr bool RES = epi_residual_ok<TM, TN, WM, WN>();
  // `scratch` (DD3D_EPI_LDS): this wave's private NP * 2 KiB of LDS.  The split planes of a 32 x 32 accumulator block are ONE contiguous
  // run of memory, [pixel][plane][64 B] -- 32 pixels x NP x 64 B -- but a lane holds a pixel's HALF rows (32 bytes per plane).  Stored
  // straight from the registers, an instruction writes 64 scattered 16-byte pieces (two per 64-byte row: twice the write requests, all
  // of them partial lines).  Staged through LDS in the memory layout (16-byte units XOR-swizzled so that both sides are conflict-free)
  // and read back in address order, every store instruction of the wave writes 1 KiB of consecutive bytes.
  constexpr int UP = NP * 4;  // 16-byte units per pixel
  constexpr int IG = (TM * TN >= 8 || TM % 2) ? 1 : 2;  // accumulator blocks whose residuals are in flight together (pairs need an even TM)
  constexpr int NRAW = 2 * NP > 4 ? 2 * NP : 4;          // 16-byte pieces of one block's residual (f32: 4, planes: 2 per plane)
  const int nlim = s.n_limit > 0 ? s.n_limit : a.N;
  const bool w32 = s.out != nullptr, wpl = s.out_planes != nullptr;
  const int h = lane >> 5, px = lane & 31;
  const long cstride = (long)s.M * (NP * 64);  // bytes per 32-channel chunk image of the output
  const float pscale = a.out_plane_scale, inv_pscale = 1.f / pscale;
  const int rmode = RES ? s.res_mode : 0;
  const long res_cstride = rmode == 3 ? (long)s.B * (s.Ho >> 1) * (s.Wo >> 1) * (NP * 64) : cstride;
  int ovf = 0;
  float amx = 0.f;
  // the reporting wave of the block (range-guard sample): see conv_epilogue
  const int seed = m0 / (TM * 32 * WM) + n0 / (TN * 32 * WN);
  int wm_sel = seed % WM, wn_sel = (seed / WM) % WN;
  if (m0 + wm_sel * TM * 32 >= s.M) wm_sel = 0;
  if (n0 + wn_sel * TN * 32 >= nlim) wn_sel = 0;
  const bool report = Planes<MODE>::F16 && a.amax != nullptr && wm == wm_sel && wn == wn_sel;
  const int mbase = m0 + wm * TM * 32 + px;  // this lane's pixel of accumulator row block 0
  auto res_pixel = [&](int m) -> long {      // pixel of the residual map that output pixel m adds
    if (rmode != 3) return m;
    const int howo = s.Ho * s.Wo;
    const int b = m / howo;
    const int rr = m - b * howo;
    const int ho = rr / s.Wo;
    const int wo = rr - ho * s.Wo;
    return ((long)b * (s.Ho >> 1) + (ho >> 1)) * (s.Wo >> 1) + (wo >> 1);
  };
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cb = (wn * TN + j) * 32;   // first column of this block inside the tile
    const int nb = n0 + cb;              // wave-uniform: first channel of this column block
    if (nb >= nlim) continue;
    const int nh = nb + 16 * h;          // this lane's first channel
    const bool partial = nb + 32 > nlim; // wave-uniform: the block holds channels past the last one stored
    const f32x4* ev = reinterpret_cast<const f32x4*>(evec) + (cb + 16 * h) / 4;
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += IG) {
      // ---- residuals of IG blocks, all loads in flight together
      u32x4 raw[IG][NRAW];
      if constexpr (RES) {
        if (rmode == 1) {
#pragma unroll
          for (int ii = 0; ii < IG; ++ii) {
            const int m = mbase + (i0 + ii) * 32;
            const gcu4p p = (gcu4p)(as_g(s.res) + (long)m * s.res_pitch + nh);
#pragma unroll
            for (int q = 0; q < 4; ++q) raw[ii][q] = (m < s.M && nh + 4 * q < nlim) ? p[q] : u32x4{0u, 0u, 0u, 0u};
          }
        } else if (rmode >= 2) {
#pragma unroll
          for (int ii = 0; ii < IG; ++ii) {
            const int m = mbase + (i0 + ii) * 32;
            const bool mv = m < s.M;
            const gcbp p = (gcbp)s.res + (long)(nb >> 5) * res_cstride + (mv ? res_pixel(m) : 0) * (NP * 64) + h * 32;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                if constexpr (CHAIN && DD3D_CHAIN_RES_SC1) raw[ii][2 * pl + q] = ld_sc1_u4((const void*)(p + pl * 64 + q * 16));  // (row 0 when m >= M: read, never used)
                else raw[ii][2 * pl + q] = mv ? *(gcu4p)(p + pl * 64 + q * 16) : u32x4{0u, 0u, 0u, 0u};
              }
          }
          if constexpr (CHAIN && DD3D_CHAIN_RES_SC1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int ii = 0; ii < IG; ++ii) {
              const bool mv = mbase + (i0 + ii) * 32 < s.M;
#pragma unroll
              for (int k = 0; k < 2 * NP; ++k) {
                asm volatile("" : "+v"(raw[ii][k]));  // uses below depend on the wait above
                if (!mv) raw[ii][k] = u32x4{0u, 0u, 0u, 0u};
              }
            }
          }
        }
      }
#pragma unroll
      for (int ii = 0; ii < IG; ++ii) {
        const int i = i0 + ii;
        const int m = mbase + i * 32;
        const bool mv = m < s.M;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 0.f;
        if constexpr (RES) {
          if (rmode == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = __uint_as_float(raw[ii][r >> 2][r & 3]);
          } else if (rmode >= 2) {
            u32x4 wv[NP][2];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) wv[pl][0] = raw[ii][2 * pl], wv[pl][1] = raw[ii][2 * pl + 1];
            unpack_terms<MODE>(wv, v);
            if constexpr (Planes<MODE>::F16) {
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] *= inv_pscale;
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 sc = ev[q], bi = ev[BN / 4 + q], lo = ev[2 * (BN / 4) + q];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * q + e] = fmaxf(acc[i][j][4 * q + e] * sc[e] + bi[e] + v[4 * q + e], lo[e]);
        }
        if (partial) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = nh + r < nlim ? v[r] : 0.f;  // channels past N inside the last chunk are zero planes
        }
        if (w32 && mv) {
          const gf4p o = (gf4p)(as_g(s.out) + (long)m * s.out_pitch + nh);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (!partial || nh + 4 * q + 4 <= nlim) {
              o[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            } else {  // the group that straddles the last stored channel: nothing past it is written (a neighbouring slice may own it)
#pragma unroll
              for (int e = 0; e < 3; ++e)
                if (nh + 4 * q + e < nlim) as_g(s.out)[(long)m * s.out_pitch + nh + 4 * q + e] = v[4 * q + e];
            }
          }
        }
        if (wpl) {
          unsigned w[8][NP];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            float e0 = v[2 * t], e1 = v[2 * t + 1];
            if constexpr (Planes<MODE>::F16) {
              e0 *= pscale, e1 *= pscale;
              ovf |= !(fabsf(e0) <= 65504.f) | !(fabsf(e1) <= 65504.f);
              if (report && mv) amx = fmaxf(amx, fmaxf(fabsf(e0), fabsf(e1)));
            }
            split_pack<MODE>(e0, e1, w[t]);
          }
          if (scratch != nullptr) {
            // unit (pixel px, plane p, 16-byte slot 2 h + jj) -> LDS slot px * UP + swizzled unit index
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
              for (int jj = 0; jj < 2; ++jj) {
                const int u = p * 4 + 2 * h + jj;
                const int us = NP == 2 ? (u ^ (px & 7)) : (p * 4 + ((2 * h + jj) ^ (px & 3)));
                *reinterpret_cast<u32x4*>(scratch + (px * UP + us) * 16) = u32x4{w[4 * jj][p], w[4 * jj + 1][p], w[4 * jj + 2][p], w[4 * jj + 3][p]};
              }
            const int mblk = m0 + (wm * TM + i) * 32;  // first pixel of the block
            const gbp dst = (gbp)s.out_planes + (long)(nb >> 5) * cstride + (long)mblk * (NP * 64);
            if constexpr (CHAIN) {
              // write-through stores are inline assembly: the compiler's wait-count pass does not see them read `vals`, so the LDS reads are
              // waited for explicitly (found the hard way: without it a store instruction now and then wrote 1 KiB of not-yet-landed registers)
              u32x4 vals[2 * NP];
#pragma unroll
              for (int k = 0; k < 2 * NP; ++k) {
                const int U = 64 * k + lane;
                const int pp = U / UP, u = U - pp * UP;
                const int us = NP == 2 ? (u ^ (pp & 7)) : ((u & ~3) + ((u & 3) ^ (pp & 3)));
                vals[k] = *reinterpret_cast<const u32x4*>(scratch + (pp * UP + us) * 16);
              }
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
              for (int k = 0; k < 2 * NP; ++k) asm volatile("" : "+v"(vals[k]));
#pragma unroll
              for (int k = 0; k < 2 * NP; ++k) {
                const int U = 64 * k + lane;
                if (mblk + U / UP < s.M) st_sc1_u4((void*)(dst + (long)U * 16), vals[k]);
              }
            } else {
#pragma unroll
            for (int k = 0; k < 2 * NP; ++k) {
              const int U = 64 * k + lane;  // linear 16-byte unit of the block's run
              const int pp = U / UP, u = U - pp * UP;
              const int us = NP == 2 ? (u ^ (pp & 7)) : ((u & ~3) + ((u & 3) ^ (pp & 3)));
              const u32x4 val = *reinterpret_cast<const u32x4*>(scratch + (pp * UP + us) * 16);
              if (mblk + pp < s.M) *(gu4p)(dst + (long)U * 16) = val;
            }
            }
          } else if (mv) {
            const gbp dst = (gbp)s.out_planes + (long)(nb >> 5) * cstride + (long)m * (NP * 64) + h * 32;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              if constexpr (CHAIN) {
                st_sc1_u4((void*)(dst + p * 64), u32x4{w[0][p], w[1][p], w[2][p], w[3][p]});
                st_sc1_u4((void*)(dst + p * 64 + 16), u32x4{w[4][p], w[5][p], w[6][p], w[7][p]});
              } else {
                *(gu4p)(dst + p * 64) = u32x4{w[0][p], w[1][p], w[2][p], w[3][p]};
                *(gu4p)(dst + p * 64 + 16) = u32x4{w[4][p], w[5][p], w[6][p], w[7][p]};
              }
            }
          }
        }
        if constexpr (TM * TN >= 8) __builtin_amdgcn_sched_barrier(0);  // (keeps the blocks of the 8-block wave tiles from being interleaved: registers)
      }
    }
  }
  if constexpr (Planes<MODE>::F16) {
    if (ovf && a.status) atomicOr(a.status, DD3D_STATUS_F16_OVERFLOW);  // (NaN / inf inputs trip it as well)
    if (report) {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) amx = fmaxf(amx, __shfl_xor(amx, d, 64));
      if (lane == 0 && amx > 0.f) atomicMax(reinterpret_cast<unsigned*>(a.amax + (seed & 15) * 32), __float_as_uint(amx));
    }
  }
}

// ---- instruction-scheduling pattern of one phase of a K step (sched_group_barrier: 0x008 MFMA, 0x010 VMEM, 0x100 DS read)
// NMFMA matrix instructions with NDMA LDS-DMA issues and NDS fragment reads spread EVENLY among them (the DMAs evenly among those): all
// eight waves of a block leave the barrier together, and eight back-to-back bursts of LDS-DMA issues queue up in the CU's one
// address / texture path while the matrix pipe starves; one issue every few MFMAs keeps that path short.
template <int NMFMA, int NDS, int NDMA>
__device__ __forceinline__ void sched_uniform() {
  constexpr int NX = NDS + NDMA;
  int placed = 0, dma = 0;
#pragma unroll
  for (int i = 0; i < NMFMA; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    const int upto = ((i + 1) * NX) / NMFMA;  // other operations due after MFMA i
#pragma unroll
    for (; placed < upto; ++placed) {
      const bool is_dma = NDMA > 0 && ((placed + 1) * NDMA) / NX > (placed * NDMA) / NX;
      if (is_dma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0), ++dma;
      else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  }
}

template <int NMFMA, int NDS, int NDMA>
__device__ __forceinline__ void sched_barrier_phase() {
  sched_uniform<NMFMA, NDS, NDMA>();
  __builtin_amdgcn_sched_barrier(0);
}

// ---- launchers implemented in the kernel translation units
int launch_conv_planes(const ConvKArgs& ka, int math_mode, int tile_cfg, hipStream_t st);  // conv_planes.hip
bool conv_planes_row_applicable(const ConvKArgs& ka);                                          // conv_planes_row.hip
int launch_conv_planes_row(const ConvKArgs& ka, int math_mode, int tile_cfg, hipStream_t st);
int conv_planes_row_rings(int math_mode, int tile_cfg, int* nsb, int* nsa);                     // conv_planes_row.hip

}  // namespace dd3d
