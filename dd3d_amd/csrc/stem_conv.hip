// Small-channel convolutions (Cin = 4 or 16: the full-resolution stem of DLA-34 / V2-99) for gfx950.
//
// These layers are HBM-bound (M = B*H*W is huge, K and N tiny): an im2col K-loop would re-stage every input pixel KH*KW
// times and pay a barrier per 32 k.  Here a block stages the input PATCH of its output tile once -- f32 from HBM, split
// exactly into the three bf16 planes hi/mid/lo (the DD3D_MATH_BF16X3 arithmetic of conv_igemm.hip) -- and every wave then
// builds its MFMA A operands straight out of that patch: for v_mfma_f32_16x16x32_bf16, lane l holds 8 consecutive k of output
// pixel (l & 15), and with the k orders below those 8 values are 16 contiguous bytes of the patch:
//   Cin = 4 : k = (dh * 8 + dw) * 4 + c    (8 tap slots per filter row, slots >= KW carry zero weights): one 32-k chunk is one
//             filter row, lane quarter q = l >> 4 reads the pixels dw = 2q, 2q+1
//   Cin = 16: k = t * 16 + c, t = dh * KW + dw (an odd tap count is padded with a zero tap): one chunk is two taps, quarter q
//             reads tap 2*chunk + (q >> 1), channels 8 * (q & 1) .. +8
// The filters (a few KiB, split on the host into the same planes) stay in registers for the whole block.
// Six bf16 products per f32 product, f32 accumulation, fused scale / bias / lower clamp epilogue as in conv_igemm.hip.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

DD3D_NOTE_BUILD_FLAGS

namespace dd3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct StemK {
  dd3d_smallc_args a;
};

template <int CIN, int KH, int KW, int S, int NT, int TH, int TW>
__global__ __launch_bounds__(256) void stem_conv_bf16x3_kernel(const StemK P) {
  const dd3d_smallc_args& a = P.a;
  constexpr int NCH = CIN == 4 ? KH : (KH * KW + 1) / 2;            // 32-k chunks
  constexpr int PH = (TH - 1) * S + KH;                             // patch rows
  constexpr int PW = (TW - 1) * S + (CIN == 4 ? 8 : KW);            // patch columns (Cin 4: reads run to tap slot 7)
  constexpr int PIXB = CIN * 2;                                     // bytes per patch pixel and plane
  constexpr int PLANE = PH * PW * PIXB;
  constexpr int GROUPS = TH * (TW / 16);                            // 16-pixel MFMA row groups per block
  constexpr int GPW = GROUPS / 4;                                   // per wave
  static_assert(GROUPS % 4 == 0 && TW % 16 == 0 && (CIN == 4 || CIN == 16) && KW <= 8, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char patch[];  // [plane][row][col][CIN] bf16

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z;
  const int oh0 = blockIdx.y * TH, ow0 = blockIdx.x * TW;
  const float* in = a.in + (long)b * a.H * a.W * a.in_pitch;

  // ---- filters -> registers: lane holds column n = lane & 15 (+16 per N tile), k quarter lane >> 4
  bf16x8 bw[NCH][NT][3];
  {
    const unsigned char* w3 = reinterpret_cast<const unsigned char*>(a.w3);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bw[ch][nt][pl] = *reinterpret_cast<const bf16x8*>(w3 + (((long)(ch * 3 + pl) * (16 * NT) + nt * 16 + (lane & 15)) * 64 + (lane >> 4) * 16));
  }

  // ---- stage the patch: f32 -> three bf16 planes (exact split by truncation)
  const int ih0 = oh0 * S - a.pad, iw0 = ow0 * S - a.pad;
  for (int pix = tid; pix < PH * PW; pix += 256) {
    const int pr = pix / PW, pc = pix - pr * PW;
    const int ih = ih0 + pr, iw = iw0 + pc;
    const bool ok = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    const float* src = in + ((long)ih * a.W + iw) * a.in_pitch;
    unsigned char* dst = patch + pix * PIXB;
#pragma unroll
    for (int q = 0; q < CIN / 4; ++q) {
      const f32x4 x = ok ? *reinterpret_cast<const f32x4*>(src + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      unsigned h[4], m[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = __float_as_uint(x[e]) & 0xffff0000u;
        const float r = x[e] - __uint_as_float(h[e]);
        m[e] = __float_as_uint(r) & 0xffff0000u;
        l[e] = __float_as_uint(r - __uint_as_float(m[e]));
      }
      *reinterpret_cast<u32x2*>(dst + 8 * q) = u32x2{__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
      *reinterpret_cast<u32x2*>(dst + PLANE + 8 * q) = u32x2{__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
      *reinterpret_cast<u32x2*>(dst + 2 * PLANE + 8 * q) = u32x2{__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
    }
  }
  __syncthreads();

  // ---- per-lane byte offsets of the 16-byte A read of every chunk, relative to the patch pixel of the lane's output pixel
  const int q4 = lane >> 4;
  int aoff[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (CIN == 4) {
      aoff[ch] = (ch * PW + 2 * q4) * PIXB;  // filter row ch, tap slots 2q, 2q+1
    } else {
      const int t = min(2 * ch + (q4 >> 1), KH * KW - 1);  // the padded tap re-reads the last one (its weights are zero)
      aoff[ch] = ((t / KW) * PW + (t % KW)) * PIXB + (q4 & 1) * 16;
    }
  }

  const int n_lane = lane & 15;
  for (int gi = 0; gi < GPW; ++gi) {
    const int g = wave * GPW + gi;
    const int r = g / (TW / 16), c0 = (g - r * (TW / 16)) * 16;
    const unsigned char* base = patch + ((r * S) * PW + (c0 + n_lane) * S) * PIXB;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      bf16x8 af[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        // Cin 4 rows are 8-byte aligned only: two 8-byte reads; Cin 16 reads one aligned 16-byte slot
        if (CIN == 4) {
          const u32x2 lo = *reinterpret_cast<const u32x2*>(base + pl * PLANE + aoff[ch]);
          const u32x2 hi = *reinterpret_cast<const u32x2*>(base + pl * PLANE + aoff[ch] + 8);
          af[pl] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
        } else {
          af[pl] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE + aoff[ch]);
        }
      }
      constexpr int PA_[6] = {2, 0, 1, 1, 0, 0};  // smallest cross terms first (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi)
      constexpr int PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA_[t]], bw[ch][nt][PB_[t]], acc[nt], 0, 0, 0);
    }
    // ---- epilogue: C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int oh = oh0 + r;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + n_lane;
      if (n >= a.N) continue;
      const float sc = a.scale[n], bi = a.bias[n];
      float lo = a.lo ? a.lo[n] : -INFINITY;
      if (a.relu) lo = fmaxf(lo, 0.f);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ow = ow0 + c0 + q4 * 4 + e;
        if (oh < a.Ho && ow < a.Wo) a.out[(((long)b * a.Ho + oh) * a.Wo + ow) * a.out_pitch + n] = fmaxf(acc[nt][e] * sc + bi, lo);
      }
    }
  }
}

template <int CIN, int KH, int KW, int S, int NT, int TH, int TW>
static int launch_stem(const dd3d_smallc_args* a, hipStream_t st) {
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + (CIN == 4 ? 8 : KW);
  const size_t lds = (size_t)3 * PH * PW * CIN * 2;
  auto k = stem_conv_bf16x3_kernel<CIN, KH, KW, S, NT, TH, TW>;
  static unsigned long long attr_done[4];
  if (lds_opt_in_needed(attr_done)) {
    if (lds_opt_in(reinterpret_cast<const void*>(k), (size_t)(lds), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
    lds_opt_in_done(attr_done);  // (every opt-in of this call site succeeded on this device)
  }
  StemK P;
  P.a = *a;
  hipLaunchKernelGGL(k, dim3(ceil_div(a->Wo, TW), ceil_div(a->Ho, TH), a->B), dim3(256), lds, st, P);
  return check_launch("stem_conv_bf16x3_kernel");
}

}  // namespace dd3d

extern "C" int dd3d_conv2d_smallc_supported(int32_t Cin, int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t N) {
  if (Cin == 4 && KH == 7 && KW == 7 && stride == 1 && pad == 3 && N <= 16) return 1;
  if (Cin == 16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && N <= 16) return 1;
  if (Cin == 16 && KH == 3 && KW == 3 && stride == 2 && pad == 1 && N <= 32) return 1;
  if (Cin == 4 && KH == 3 && KW == 3 && stride == 2 && pad == 1 && N <= 64) return 1;
  return 0;
}

extern "C" int dd3d_conv2d_smallc_bf16x3(const dd3d_smallc_args* a, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(a && a->in && a->out && a->w3 && a->scale && a->bias, "dd3d_conv2d_smallc_bf16x3: null pointer");
  DD3D_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->N > 0, "dd3d_conv2d_smallc_bf16x3: empty problem");
  DD3D_REQUIRE(a->Ho == (a->H + 2 * a->pad - a->KH) / a->stride + 1 && a->Wo == (a->W + 2 * a->pad - a->KW) / a->stride + 1,
               "dd3d_conv2d_smallc_bf16x3: output size %dx%d does not match the geometry", a->Ho, a->Wo);
  DD3D_REQUIRE(dd3d_conv2d_smallc_supported(a->Cin, a->KH, a->KW, a->stride, a->pad, a->N),
               "dd3d_conv2d_smallc_bf16x3: Cin=%d %dx%d stride %d pad %d N=%d is not instantiated", a->Cin, a->KH, a->KW, a->stride, a->pad, a->N);
  DD3D_REQUIRE(a->in_pitch % 4 == 0, "dd3d_conv2d_smallc_bf16x3: in_pitch=%d must be a multiple of 4 floats", a->in_pitch);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->Cin == 4 && a->KH == 7) return launch_stem<4, 7, 7, 1, 1, 8, 64>(a, st);
  if (a->Cin == 16 && a->stride == 1) return launch_stem<16, 3, 3, 1, 1, 4, 64>(a, st);
  if (a->Cin == 16 && a->stride == 2) return launch_stem<16, 3, 3, 2, 2, 4, 32>(a, st);
  return launch_stem<4, 3, 3, 2, 4, 4, 32>(a, st);
}
