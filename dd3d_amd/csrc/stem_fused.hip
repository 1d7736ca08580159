// The full-resolution stem of DLA-34 in ONE launch (gfx950):  uint8 image -> (x - mean) / std -> base_layer 7x7 3->16 + BN + ReLU ->
// level0 3x3 16->16 + BN + ReLU -> level1 3x3 stride 2 16->32 + BN + ReLU  (tridet/modeling/feature_extractor/dla.py:271-280,327-344,
// core.py:61-66), DD3D_MATH_F16X2 arithmetic (every operand = two IEEE-half terms, three products, f32 accumulation).
//
// Launch by launch these layers are HBM-bound: 230 MB per 384 x 1280 image move through HBM for 2.9 GMAC (preprocess 9 MB, base_layer
// 39, level0 63, level1 47, the plane split 31, ...), 71 us per image at batch 8.  Here a block owns an 8 x 32 tile of level1 outputs and
// keeps everything between the uint8 pixels and those outputs in its LDS:
//
//   image patch  25 x 74 x 4 ch   (region A, 29 KiB)   normalised on the fly, zero outside the real image (ImageList padding)
//   base tile    19 x 67 x 16 ch  (region B, 80 KiB)   zero outside the canvas (= the zero padding level0's filter sees)
//   level0 tile  17 x 65 x 16 ch  (region A, 70 KiB)   overwrites the image patch
//   level1 tile   8 x 32 x 32 ch  (region B, 32 KiB f32) staged for coalesced stores
//
// all as two planes of halves (hi, lo of value x plane scale), pixel-major rows of Cin halves, exactly the operand form of
// v_mfma_f32_16x16x32_f16: lane l holds 8 consecutive k of pixel (l & 15), and with the k orders of stem_conv.hip
//   Cin 4 : k = (dh * 8 + dw) * 4 + c   one 32-k chunk = one filter row, lane quarter q reads pixels dw = 2q, 2q + 1
//   Cin 16: k = (dh * 3 + dw) * 16 + c  one chunk = two taps, quarter q reads tap 2 chunk + (q >> 1), channels 8 (q & 1) .. + 8
// those are 16 contiguous bytes of a tile.  A 16-pixel MFMA row group is 16 CONSECUTIVE pixels of the flattened tile (rows wrap), so no
// tile width is wasted.  HBM traffic per image: 1.5 MB in, 15.7 (+ 15.7 f32) MB out.
#include <stdlib.h>

#include "common.h"

DD3D_NOTE_BUILD_FLAGS

namespace dd3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct StemFusedK {
  dd3d_stem_args a;
};

// (Measured, profiles/r04w_stem_tile_variants.txt: an 8 x 16 tile on 4-wave blocks -- 78 KiB of LDS, two blocks per CU at different phases -- 164 us
// for four images against 160 for this 8 x 32 tile on 8 waves; 8 x 16 on 8 waves 197, 8 x 32 on 4 waves 206.  The parity tests pass for all four.)
#ifndef DD3D_STEM_TW
#define DD3D_STEM_TW 32
#endif
#ifndef DD3D_STEM_WAVES
#define DD3D_STEM_WAVES 8
#endif
constexpr int SF_TH = 8, SF_TW = DD3D_STEM_TW;               // level1 outputs per block
constexpr int SF_R0 = 2 * SF_TH + 1, SF_C0 = 2 * SF_TW + 1;  // level0 tile 17 x 65
constexpr int SF_RB = SF_R0 + 2, SF_CB = SF_C0 + 2;          // base tile 19 x 67
constexpr int SF_RI = SF_RB + 6, SF_CI = SF_CB + 7;          // image patch 25 x 74 (the Cin-4 reads run to tap slot 7)
constexpr int SF_NB = SF_RB * SF_CB, SF_N0 = SF_R0 * SF_C0, SF_N1 = SF_TH * SF_TW;  // 1273, 1105, 256 pixels
constexpr int SF_GB = (SF_NB + 15) / 16, SF_G0 = (SF_N0 + 15) / 16, SF_G1 = SF_N1 / 16;  // 80, 70, 16 row groups
constexpr int SF_IMG_PLANE = SF_RI * SF_CI * 8;   // bytes per plane
constexpr int SF_B_PLANE = SF_GB * 16 * 32;
constexpr int SF_0_PLANE = SF_G0 * 16 * 32;
constexpr int SF_REGION_A = 2 * SF_0_PLANE > 2 * SF_IMG_PLANE ? 2 * SF_0_PLANE : 2 * SF_IMG_PLANE;
constexpr int SF_REGION_B = 2 * SF_B_PLANE;
constexpr int SF_LDS = SF_REGION_A + SF_REGION_B;
constexpr int SF_WAVES = DD3D_STEM_WAVES;
#ifndef DD3D_STEM_CHAINS
#define DD3D_STEM_CHAINS 2  // row groups (independent MFMA accumulation chains) a wave interleaves in the base / level0 stages
#endif
constexpr int SF_U = DD3D_STEM_CHAINS;
static_assert(SF_LDS <= 160 * 1024 && SF_N1 * 32 * 4 <= SF_REGION_B, "stem tile does not fit the LDS");

// value -> (hi, lo) halves of value * plane scale; returns nonzero if the scaled value leaves the half range
__device__ __forceinline__ int sf_split(float v, float pscale, _Float16& hi, _Float16& lo) {
  const float s = v * pscale;
  hi = (_Float16)s;
  lo = (_Float16)(s - (float)hi);
  return !(fabsf(s) <= 65504.f);
}

// One 16-pixel x 16-channel accumulator block -> the (hi, lo) half planes of a tile in LDS, rows of 16 channels (32 bytes) per pixel.
// C/D map of the 16x16 MFMA: column (channel) = lane & 15, row (pixel of the group) = (lane >> 4) * 4 + e.  Lanes n and n ^ 1 swap one value
// per pixel pair (DPP quad_perm [1, 0, 3, 2]) so that every lane stores a (channel n & ~1, n | 1) dword for two of its four pixels: four
// 4-byte LDS stores per lane instead of eight 2-byte ones.  `zero(po)`: the pixel lies outside the canvas (the next filter's zero padding).
template <class Z>
__device__ __forceinline__ int sf_store_tile(unsigned char* plane_hi, int plane_bytes, int g, int lane, f32x4 acc, float sc, float bi, float pscale, Z&& zero) {
  const int q4 = lane >> 4, n_lane = lane & 15, odd = lane & 1;
  int ovf = 0;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = zero(g * 16 + q4 * 4 + e) ? 0.f : fmaxf(acc[e] * sc + bi, 0.f);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float send = odd ? v[2 * t] : v[2 * t + 1];
    const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, false));
    const float own = odd ? v[2 * t + 1] : v[2 * t];
    const int po = g * 16 + q4 * 4 + 2 * t + odd;           // the pixel this lane stores
    const float e0 = odd ? recv : own, e1 = odd ? own : recv;  // channels n & ~1, n | 1
    _Float16 h0, l0, h1, l1;
    ovf |= sf_split(e0, pscale, h0, l0) | sf_split(e1, pscale, h1, l1);
    unsigned char* dst = plane_hi + po * 32 + (n_lane & ~1) * 2;
    *reinterpret_cast<f16x2*>(dst) = f16x2{h0, h1};
    *reinterpret_cast<f16x2*>(dst + plane_bytes) = f16x2{l0, l1};
  }
  return ovf;
}

__global__ __launch_bounds__(64 * SF_WAVES) void stem_fused_f16x2_kernel(const StemFusedK P) {
  const dd3d_stem_args& a = P.a;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const regA = lds;                 // image patch, then the level0 tile
  unsigned char* const regB = lds + SF_REGION_A;   // base tile, then the level1 staging
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4, n_lane = lane & 15;
  const int b = blockIdx.z;
  const int oh0 = blockIdx.y * SF_TH, ow0 = blockIdx.x * SF_TW;  // level1 tile origin
  const int Ho = a.Hp >> 1, Wo = a.Wp >> 1;
  const float pscale = a.plane_scale;
  int ovf = 0;

  // start-of-forward chores one block takes along (dd3d_stem_args.K / inv_K / zero_f32): K^-1 of every image, and the per-forward range-guard
  // maxima back to zero -- every consumer runs in a later launch of the stream
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    if (a.K != nullptr && a.inv_K != nullptr)
      for (int i = tid; i < a.B; i += 64 * SF_WAVES) invert3x3(a.K + 9 * i, a.inv_K + 9 * i);
    if (a.zero_f32 != nullptr)
      for (int i = tid; i < a.zero_count; i += 64 * SF_WAVES) a.zero_f32[i] = 0.f;
  }

  // ------------------------------------------------------------------ stage 0: image patch (normalise, split)
  {
    const int ih0 = 2 * oh0 - 5, iw0 = 2 * ow0 - 5;  // level1 -> level0 (-1) -> base (-1) -> image (-3)
    const int vh = a.sizes[2 * b], vw = a.sizes[2 * b + 1];
    const long plane = (long)a.Hp * a.Wp;
    const uint8_t* src = a.src + (long)b * 3 * plane;
    // Every thread's pixels are LOADED first, then converted: the loop used to wait for each pixel's three byte loads before it issued the
    // next pixel's (four exposed HBM round trips per block on a CU the block owns alone; round 5, profiles/r05k_stem_patch_loads_ab.txt).
    constexpr int NT = 64 * SF_WAVES, NPX = (SF_RI * SF_CI + NT - 1) / NT;
    unsigned char raw[NPX][3];
    bool in_img[NPX];
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int pix = tid + k * NT;
      const int pr = pix / SF_CI, pc = pix - pr * SF_CI;
      const int ih = ih0 + pr, iw = iw0 + pc;
      in_img[k] = pix < SF_RI * SF_CI && (unsigned)ih < (unsigned)vh && (unsigned)iw < (unsigned)vw;
      // UNCONDITIONAL loads (a pixel outside the image reads pixel (0, 0) and is zeroed below): a guarded load compiles to a branch with
      // its own wait, which is the serialisation this loop exists to remove
      const uint8_t* p = src + (long)(in_img[k] ? ih : 0) * a.Wp + (in_img[k] ? iw : 0);
#pragma unroll
      for (int c = 0; c < 3; ++c) raw[k][c] = p[c * plane];
    }
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int pix = tid + k * NT;
      if (pix >= SF_RI * SF_CI) break;
      f16x4 hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
      if (in_img[k]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float x = ((float)raw[k][c] - a.mean[c]) / a.stdv[c];  // the division is kept: torch's (x - mean) / std
          _Float16 h, l;
          sf_split(x, pscale, h, l);  // |x| < 3: always inside the half range
          hi[c] = h, lo[c] = l;
        }
      }
      *reinterpret_cast<f16x4*>(regA + pix * 8) = hi;
      *reinterpret_cast<f16x4*>(regA + SF_IMG_PLANE + pix * 8) = lo;
    }
  }

  // ------------------------------------------------------------------ stage 1: base_layer 7x7 (3 -> 16), Cin padded to 4
  {
    f16x8 w[7][2];
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.w1);
#pragma unroll
    for (int ch = 0; ch < 7; ++ch)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) w[ch][pl] = *reinterpret_cast<const f16x8*>(wp + ((ch * 2 + pl) * 16 + n_lane) * 64 + q4 * 16);
    const float sc = a.scale1[n_lane], bi = a.bias1[n_lane];
    __syncthreads();
    const int rb0 = 2 * oh0 - 2, cb0 = 2 * ow0 - 2;  // canvas position of the base tile's origin
    const auto outside = [&](int po) {
      const int pr = po / SF_CB, pc = po - pr * SF_CB;
      return !((unsigned)(rb0 + pr) < (unsigned)a.Hp && (unsigned)(cb0 + pc) < (unsigned)a.Wp);  // level0's zero padding
    };
    // SF_U row groups per iteration: SF_U independent accumulation chains keep the matrix pipe fed (one chain of 21 dependent MFMAs per
    // group left it idle most of the time: 40 us per image at batch 8; two chains: round 3; SF_U: round 5, profiles/r05j_stem_chains_ab.txt)
    for (int g0 = wave; g0 < SF_GB; g0 += SF_U * SF_WAVES) {
      const unsigned char* base[SF_U];
#pragma unroll
      for (int u = 0; u < SF_U; ++u) {
        const int gu = g0 + u * SF_WAVES;
        const int p = min((gu < SF_GB ? gu : g0) * 16 + n_lane, SF_NB - 1);
        const int r = p / SF_CB, c = p - r * SF_CB;
        base[u] = regA + (r * SF_CI + c + 2 * q4) * 8;
      }
      f32x4 acc[SF_U];
#pragma unroll
      for (int u = 0; u < SF_U; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < 7; ++ch) {
        f16x8 ahi[SF_U], alo[SF_U];
#pragma unroll
        for (int u = 0; u < SF_U; ++u) {
          const unsigned char* src = base[u] + ch * SF_CI * 8;  // filter row ch: pixels (r + ch, c + 2 q4), (.., + 1): 16 bytes, 8-byte aligned
          const u32x2 h0 = *reinterpret_cast<const u32x2*>(src), h1 = *reinterpret_cast<const u32x2*>(src + 8);
          const u32x2 l0 = *reinterpret_cast<const u32x2*>(src + SF_IMG_PLANE), l1 = *reinterpret_cast<const u32x2*>(src + SF_IMG_PLANE + 8);
          ahi[u] = __builtin_bit_cast(f16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
          alo[u] = __builtin_bit_cast(f16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
        }
#pragma unroll
        for (int u = 0; u < SF_U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[u], w[ch][0], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < SF_U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[u], w[ch][1], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < SF_U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[u], w[ch][0], acc[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < SF_U; ++u) {
        const int gu = g0 + u * SF_WAVES;  // (po < SF_GB * 16: the padded tail of the plane absorbs it)
        if (gu < SF_GB) ovf |= sf_store_tile(regB, SF_B_PLANE, gu, lane, acc[u], sc, bi, pscale, outside);
      }
    }
  }

  // ------------------------------------------------------------------ stage 2: level0 3x3 (16 -> 16)
  {
    f16x8 w[5][2];
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.w2);
#pragma unroll
    for (int ch = 0; ch < 5; ++ch)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) w[ch][pl] = *reinterpret_cast<const f16x8*>(wp + ((ch * 2 + pl) * 16 + n_lane) * 64 + q4 * 16);
    const float sc = a.scale2[n_lane], bi = a.bias2[n_lane];
    int aoff[5];
#pragma unroll
    for (int ch = 0; ch < 5; ++ch) {
      const int t = min(2 * ch + (q4 >> 1), 8);  // the padded tenth tap re-reads the ninth (its weights are zero)
      aoff[ch] = ((t / 3) * SF_CB + (t % 3)) * 32 + (q4 & 1) * 16;
    }
    __syncthreads();  // base tile complete; the image patch is dead
    const int r00 = 2 * oh0 - 1, c00 = 2 * ow0 - 1;
    const auto outside = [&](int po) {
      const int pr = po / SF_C0, pc = po - pr * SF_C0;
      return !((unsigned)(r00 + pr) < (unsigned)a.Hp && (unsigned)(c00 + pc) < (unsigned)a.Wp);
    };
    for (int g0 = wave; g0 < SF_G0; g0 += SF_U * SF_WAVES) {
      const unsigned char* base[SF_U];
#pragma unroll
      for (int u = 0; u < SF_U; ++u) {
        const int gu = g0 + u * SF_WAVES;
        const int p = min((gu < SF_G0 ? gu : g0) * 16 + n_lane, SF_N0 - 1);
        const int r = p / SF_C0, c = p - r * SF_C0;
        base[u] = regB + (r * SF_CB + c) * 32;
      }
      f32x4 acc[SF_U];
#pragma unroll
      for (int u = 0; u < SF_U; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < 5; ++ch) {
        f16x8 ahi[SF_U], alo[SF_U];
#pragma unroll
        for (int u = 0; u < SF_U; ++u) {
          ahi[u] = *reinterpret_cast<const f16x8*>(base[u] + aoff[ch]);
          alo[u] = *reinterpret_cast<const f16x8*>(base[u] + SF_B_PLANE + aoff[ch]);
        }
#pragma unroll
        for (int u = 0; u < SF_U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[u], w[ch][0], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < SF_U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[u], w[ch][1], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < SF_U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[u], w[ch][0], acc[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < SF_U; ++u) {
        const int gu = g0 + u * SF_WAVES;
        if (gu < SF_G0) ovf |= sf_store_tile(regA, SF_0_PLANE, gu, lane, acc[u], sc, bi, pscale, outside);
      }
    }
  }

  // ------------------------------------------------------------------ stage 3: level1 3x3 stride 2 (16 -> 32)
  {
    f16x8 w[5][2][2];
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.w3);
#pragma unroll
    for (int ch = 0; ch < 5; ++ch)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) w[ch][pl][nt] = *reinterpret_cast<const f16x8*>(wp + ((ch * 2 + pl) * 32 + nt * 16 + n_lane) * 64 + q4 * 16);
    float sc[2], bi[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) sc[nt] = a.scale3[nt * 16 + n_lane], bi[nt] = a.bias3[nt * 16 + n_lane];
    int aoff[5];
#pragma unroll
    for (int ch = 0; ch < 5; ++ch) {
      const int t = min(2 * ch + (q4 >> 1), 8);
      aoff[ch] = ((t / 3) * SF_C0 + (t % 3)) * 32 + (q4 & 1) * 16;
    }
    __syncthreads();  // level0 tile complete; the base tile is dead
    float* stage = reinterpret_cast<float*>(regB);  // [256 pixels][32 channels] f32
    constexpr int U3 = SF_U >= 4 ? 2 : 1;  // groups per iteration (each has two output-channel halves = two chains)
    for (int g0 = wave; g0 < SF_G1; g0 += U3 * SF_WAVES) {
      const unsigned char* base[U3];
#pragma unroll
      for (int u = 0; u < U3; ++u) {
        const int gu = g0 + u * SF_WAVES;
        const int p = (gu < SF_G1 ? gu : g0) * 16 + n_lane;
        const int r = p / SF_TW, c = p - r * SF_TW;
        base[u] = regA + (2 * r * SF_C0 + 2 * c) * 32;
      }
      f32x4 acc[U3][2];
#pragma unroll
      for (int u = 0; u < U3; ++u) acc[u][0] = acc[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < 5; ++ch) {
        f16x8 ahi[U3], alo[U3];
#pragma unroll
        for (int u = 0; u < U3; ++u) {
          ahi[u] = *reinterpret_cast<const f16x8*>(base[u] + aoff[ch]);
          alo[u] = *reinterpret_cast<const f16x8*>(base[u] + SF_0_PLANE + aoff[ch]);
        }
        // acc += (hi + lo) x (whi + wlo) without the lo x lo term: lo x hi, hi x lo, hi x hi per chain, the chains interleaved
#pragma unroll
        for (int u = 0; u < U3; ++u)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[u], w[ch][0][nt], acc[u][nt], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U3; ++u)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[u], w[ch][1][nt], acc[u][nt], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U3; ++u)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[u], w[ch][0][nt], acc[u][nt], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U3; ++u) {
        const int gu = g0 + u * SF_WAVES;
        if (gu >= SF_G1) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) stage[(gu * 16 + q4 * 4 + e) * 32 + nt * 16 + n_lane] = fmaxf(acc[u][nt][e] * sc[nt] + bi[nt], 0.f);
      }
    }
    __syncthreads();
    // coalesced write-out: thread -> (pixel, 4 channels): 16 bytes of the f32 row, 8 bytes of each plane row
    for (int i = tid; i < SF_N1 * 8; i += 64 * SF_WAVES) {
      const int p = i >> 3, c4 = (i & 7) * 4;
      const int r = p / SF_TW, c = p - r * SF_TW;
      const int oh = oh0 + r, ow = ow0 + c;
      if (oh >= Ho || ow >= Wo) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + p * 32 + c4);
      const long m = ((long)b * Ho + oh) * Wo + ow;
      if (a.out) *reinterpret_cast<f32x4*>(a.out + m * a.out_pitch + c4) = v;
      if (a.out_planes) {
        f16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 h, l;
          ovf |= sf_split(v[e], pscale, h, l);
          hi[e] = h, lo[e] = l;
        }
        unsigned char* dst = reinterpret_cast<unsigned char*>(a.out_planes) + m * 128 + c4 * 2;  // [pixel][plane][32] halves, one chunk
        *reinterpret_cast<f16x4*>(dst) = hi;
        *reinterpret_cast<f16x4*>(dst + 64) = lo;
      }
    }
  }
  if (ovf && a.status) atomicOr(a.status, DD3D_STATUS_F16_OVERFLOW);
}

}  // namespace dd3d

extern "C" int dd3d_stem_fused_f16x2(const dd3d_stem_args* a, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(a && a->src && a->sizes && a->w1 && a->w2 && a->w3 && a->scale1 && a->bias1 && a->scale2 && a->bias2 && a->scale3 && a->bias3,
               "dd3d_stem_fused_f16x2: null pointer");
  DD3D_REQUIRE(a->out || a->out_planes, "dd3d_stem_fused_f16x2: no output");
  DD3D_REQUIRE(a->B > 0 && a->Hp > 0 && a->Wp > 0 && (a->Hp % 2) == 0 && (a->Wp % 2) == 0, "dd3d_stem_fused_f16x2: canvas %dx%d (x%d) must be even", a->Hp,
               a->Wp, a->B);
  DD3D_REQUIRE(!a->out || (a->out_pitch >= 32 && a->out_pitch % 4 == 0), "dd3d_stem_fused_f16x2: out_pitch=%d", a->out_pitch);
  DD3D_REQUIRE(a->plane_scale > 0.f, "dd3d_stem_fused_f16x2: plane_scale=%g", (double)a->plane_scale);
  DD3D_REQUIRE((a->K == nullptr) == (a->inv_K == nullptr) && a->zero_count >= 0 && (a->zero_count == 0 || a->zero_f32),
               "dd3d_stem_fused_f16x2: K and inv_K come together; zero_count=%d needs zero_f32", a->zero_count);
  static unsigned long long attr_done[4];
  if (lds_opt_in_needed(attr_done)) {
    if (lds_opt_in(reinterpret_cast<const void*>(stem_fused_f16x2_kernel), (size_t)SF_LDS, "stem_fused_f16x2_kernel") != DD3D_OK) return DD3D_E_LAUNCH;
    lds_opt_in_done(attr_done);  // (every opt-in of this call site succeeded on this device)
  }
  StemFusedK P;
  P.a = *a;
  const int Ho = a->Hp / 2, Wo = a->Wp / 2;
  hipLaunchKernelGGL(stem_fused_f16x2_kernel, dim3(ceil_div(Wo, SF_TW), ceil_div(Ho, SF_TH), a->B), dim3(64 * SF_WAVES), SF_LDS, reinterpret_cast<hipStream_t>(stream), P);
  return check_launch("stem_fused_f16x2_kernel");
}
