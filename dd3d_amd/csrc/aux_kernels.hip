// Memory-bound helper kernels of the forward path (gfx950): input normalisation + padding,
// 2x2 max pooling, intrinsics inverse, error plumbing of the C ABI.
#include <stdarg.h>

#include "common.h"

DD3D_NOTE_BUILD_FLAGS

namespace dd3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Norm3 {
  float mean[3], stdv[3];
};

// One thread per output pixel: three strided byte reads (each plane is read fully coalesced across the wave), one
// 16-byte NHWC4 store.  The division is kept (not folded into a multiply) so the value equals torch's (x-mean)/std.
__global__ __launch_bounds__(256) void preprocess_u8_nhwc4_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ sizes,
                                                                  float* __restrict__ dst, int B, int Hp, int Wp, Norm3 nm) {
  const long npix = (long)B * Hp * Wp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const long t = i / Wp;
    const int y = (int)(t % Hp);
    const int b = (int)(t / Hp);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (y < sizes[2 * b] && x < sizes[2 * b + 1]) {
      const long plane = (long)Hp * Wp;
      const uint8_t* p = src + (long)b * 3 * plane + (long)y * Wp + x;
      o[0] = ((float)p[0] - nm.mean[0]) / nm.stdv[0];
      o[1] = ((float)p[plane] - nm.mean[1]) / nm.stdv[1];
      o[2] = ((float)p[2 * plane] - nm.mean[2]) / nm.stdv[2];
    }
    *reinterpret_cast<f32x4*>(dst + i * 4) = o;
  }
}

// thread per (output pixel, 4-channel group)
__global__ __launch_bounds__(256) void maxpool2x2_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C4,
                                                              int in_pitch, int out_pitch) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float* p = in + (((long)b * H + 2 * ho) * W + 2 * wo) * in_pitch + c;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(p);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(p + in_pitch);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(p + (long)W * in_pitch);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(p + (long)W * in_pitch + in_pitch);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
    *reinterpret_cast<f32x4*>(out + (((long)b * Ho + ho) * Wo + wo) * out_pitch + c) = o;
  }
}

// fine[b, y, x, :] += coarse[b, y/2, x/2, :]   (FPN top-down path: F.interpolate(scale 2, nearest) + add)
__global__ __launch_bounds__(256) void upsample2x_add_nhwc_kernel(float* __restrict__ fine, const float* __restrict__ coarse, int B, int H, int W,
                                                                  int C4, int fine_pitch, int coarse_pitch) {
  const long total = (long)B * H * W * C4;
  const int Hc = H >> 1, Wc = W >> 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float* f = fine + (((long)b * H + y) * W + x) * fine_pitch + c;
    const float* q = coarse + (((long)b * Hc + (y >> 1)) * Wc + (x >> 1)) * coarse_pitch + c;
    *reinterpret_cast<f32x4*>(f) = *reinterpret_cast<const f32x4*>(f) + *reinterpret_cast<const f32x4*>(q);
  }
}

// Aligned bilinear upsampling by an integer factor f (tridet/utils/tensor2d.py:28-47: replicate-pad by one, bilinear with
// align_corners=True to (f*h+1, f*w+1), crop; offset "half" shifts the result by f/2 with edge replication) of channel 0 of an
// NHWC map, fused with the focal-length scaling of DD3DDenseDepth (dense_depth.py:146-151): out /= |(invK00, invK11)| * factor.
__global__ __launch_bounds__(256) void aligned_bilinear_scale_kernel(const float* __restrict__ src, float* __restrict__ out,
                                                                     const float* __restrict__ inv_K, int B, int h, int w, int pitch, int f, int half,
                                                                     float factor) {
  const int H = h * f, W = w * f;
  const long total = (long)B * H * W;
  const float scale = (float)h / (float)(f * h);  // (in - 1) / (out - 1) of the padded (h+1) -> (f*h+1) resize, = 1/f
  const float scale_w = (float)w / (float)(f * w);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long t = i / W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const int ys = half ? max(y - f / 2, 0) : y, xs = half ? max(x - f / 2, 0) : x;
    const float ry = scale * (float)ys, rx = scale_w * (float)xs;
    const int y0 = (int)ry, x0 = (int)rx;
    const float ly = ry - (float)y0, lx = rx - (float)x0;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);  // row / column h, w of the padded map replicate h-1, w-1
    const float* p = src + (long)b * h * w * pitch;
    const float v00 = p[((long)y0 * w + x0) * pitch], v01 = p[((long)y0 * w + x1) * pitch];
    const float v10 = p[((long)y1 * w + x0) * pitch], v11 = p[((long)y1 * w + x1) * pitch];
    float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    if (factor > 0.f) {
      const float k0 = inv_K[9 * b], k4 = inv_K[9 * b + 4];
      v = v / (sqrtf(k0 * k0 + k4 * k4) * factor);
    }
    out[i] = v;
  }
}

// One pass of Pillow's 8-bit resampling (libImaging/Resample.c ImagingResampleHorizontal_8bpc / Vertical_8bpc): out = clip8((2^21 +
// sum_i in[lo + i] * kk[i]) >> 22) with per-output-coordinate bounds (lo, n) and 22-bit fixed-point coefficients computed on the
// host.  `along_w` selects the axis; strides are in bytes (= elements).  The reference resizes its uint8 images with exactly this
// routine (detectron2 ResizeTransform.apply_image -> PIL Image.resize(BILINEAR)), which makes the device result bit-identical.
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int C, int out_h, int out_w,
                                                          long src_plane, int src_row, long dst_plane, int dst_row, const int32_t* __restrict__ lo,
                                                          const int32_t* __restrict__ cnt, const int32_t* __restrict__ kk, int ksize, int along_w) {
  const long total = (long)C * out_h * out_w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % out_w);
    long t = i / out_w;
    const int y = (int)(t % out_h);
    const int c = (int)(t / out_h);
    const int o = along_w ? x : y;
    const int first = lo[o], n = cnt[o];
    const int32_t* k = kk + (long)o * ksize;
    const uint8_t* p = along_w ? src + c * src_plane + (long)y * src_row + first : src + c * src_plane + (long)first * src_row + x;
    const int step = along_w ? 1 : src_row;
    int acc = 1 << 21;
    for (int j = 0; j < n; ++j) acc += (int)p[(long)j * step] * k[j];
    acc >>= 22;
    dst[c * dst_plane + (long)y * dst_row + x] = (uint8_t)min(max(acc, 0), 255);
  }
}

// 3x3 stride-2 max pooling without padding, ceil_mode=True (windows may overhang the bottom / right edge).
__global__ __launch_bounds__(256) void maxpool3x3s2_ceil_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                                                     int Ho, int Wo, int C4, int in_pitch, int out_pitch) {
  const long total = (long)B * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    f32x4 o = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y = 2 * ho + dy;
      if (y >= H) break;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int x = 2 * wo + dx;
        if (x >= W) break;
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + (((long)b * H + y) * W + x) * in_pitch + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], v[e]);
      }
    }
    *reinterpret_cast<f32x4*>(out + (((long)b * Ho + ho) * Wo + wo) * out_pitch + c) = o;
  }
}

// Global average pool, pass 1: partial[b][rs][c] = sum over the rs-th slice of the H*W rows (deterministic: no atomics).
// block = 256 threads = 16 row lanes x 16 float4 columns (64 channels); grid = (C/64, RS, B).
__global__ __launch_bounds__(256) void gap_partial_nhwc_kernel(const float* __restrict__ in, float* __restrict__ partial, int HW, int C,
                                                               int pitch, int RS) {
  const int cg = blockIdx.x, rs = blockIdx.y, b = blockIdx.z;
  const int rl = threadIdx.x >> 4, cl = threadIdx.x & 15;
  const int c = cg * 64 + cl * 4;
  const int rows_per = (HW + RS - 1) / RS;
  const int r0 = rs * rows_per, r1 = min(HW, r0 + rows_per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 16) acc += *reinterpret_cast<const f32x4*>(in + ((long)b * HW + r) * pitch + c);
  __shared__ f32x4 red[16][16];
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < C) {
    f32x4 s = red[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][cl];
    *reinterpret_cast<f32x4*>(partial + ((long)b * RS + rs) * C + c) = s;
  }
}

// eSE gate: g[b][co] = hsigmoid( fc_w[co][:] . mean[b][:] + fc_b[co] ),  mean = sum of the RS partials / HW.
// grid = (C/4, B), block = 256 = 4 waves, one output channel per wave, lanes stride over the input channels.
__global__ __launch_bounds__(256) void ese_gate_kernel(const float* __restrict__ partial, const float* __restrict__ fc_w,
                                                       const float* __restrict__ fc_b, float* __restrict__ gate, int C, int RS, float inv_hw) {
  const int b = blockIdx.y;
  const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  if (co < C) {
    for (int ci = lane; ci < C; ci += 64) {
      float m = 0.f;
      for (int r = 0; r < RS; ++r) m += partial[((long)b * RS + r) * C + ci];
      acc += fc_w[(long)co * C + ci] * (m * inv_hw);
    }
  }
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (lane == 0 && co < C) {
    const float v = acc + fc_b[co];
    gate[(long)b * C + co] = fminf(fmaxf(v + 3.0f, 0.f), 6.0f) / 6.0f;  // F.relu6(x + 3) / 6
  }
}

// out[b, y, x, :] = x[b, y, x, :] * gate[b, :] (+ identity[b, y, x, :])
__global__ __launch_bounds__(256) void scale_add_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                             const float* __restrict__ identity, float* __restrict__ out, int B, int HW, int C4,
                                                             int x_pitch, int id_pitch, int out_pitch) {
  const long total = (long)B * HW * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const long m = i / C4;
    const int b = (int)(m / HW);
    f32x4 v = *reinterpret_cast<const f32x4*>(x + m * x_pitch + c) * *reinterpret_cast<const f32x4*>(gate + (long)b * C4 * 4 + c);
    if (identity) v += *reinterpret_cast<const f32x4*>(identity + m * id_pitch + c);
    *reinterpret_cast<f32x4*>(out + m * out_pitch + c) = v;
  }
}

// General 3x3 inverse by cofactors (the reference calls torch.inverse on the stacked intrinsics).
__global__ void invert3x3_kernel(const float* __restrict__ K, float* __restrict__ invK, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  invert3x3(K + 9 * b, invK + 9 * b);  // common.h
}

// Union of the build-time knobs the translation units report (build_flags.h); zero-initialised storage, so the static registrations of the
// other translation units may run before this one's.
static char g_build_flags[2048];
void register_build_flags(const char* file, const char* flags) {
  if (flags == nullptr || flags[0] == 0) return;
  const char* base = strrchr(file, '/');
  base = base ? base + 1 : file;
  const size_t n = strlen(g_build_flags);
  if (n + 8 < sizeof(g_build_flags)) snprintf(g_build_flags + n, sizeof(g_build_flags) - n, "%s%s:%s", n ? "; " : "", base, flags);
}

}  // namespace dd3d

extern "C" const char* dd3d_build_flags(void) { return dd3d::g_build_flags; }
extern "C" int dd3d_abi_version(void) { return DD3D_ABI_VERSION; }
extern "C" const char* dd3d_last_error(void) { return dd3d::g_err; }
extern "C" const char* dd3d_arch(void) { return "gfx950"; }

extern "C" int dd3d_preprocess_u8_nhwc4(const uint8_t* src, const int32_t* sizes, float* dst, int32_t B, int32_t Hp, int32_t Wp,
                                        const float mean[3], const float stdv[3], void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(src && sizes && dst && B > 0 && Hp > 0 && Wp > 0, "dd3d_preprocess_u8_nhwc4: bad arguments");
  Norm3 nm;
  for (int i = 0; i < 3; ++i) nm.mean[i] = mean[i], nm.stdv[i] = stdv[i];
  const long npix = (long)B * Hp * Wp;
  const int grid = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
  hipLaunchKernelGGL(preprocess_u8_nhwc4_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, sizes, dst, B, Hp,
                     Wp, nm);
  return check_launch("preprocess_u8_nhwc4_kernel");
}

extern "C" int dd3d_maxpool2x2_nhwc(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t in_pitch,
                                    int32_t out_pitch, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(in && out && B > 0, "dd3d_maxpool2x2_nhwc: bad arguments");
  DD3D_REQUIRE((H % 2) == 0 && (W % 2) == 0, "dd3d_maxpool2x2_nhwc: H=%d W=%d must be even", H, W);
  DD3D_REQUIRE((C % 4) == 0 && (in_pitch % 4) == 0 && (out_pitch % 4) == 0, "dd3d_maxpool2x2_nhwc: C / pitches must be multiples of 4");
  const long total = (long)B * (H / 2) * (W / 2) * (C / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(maxpool2x2_nhwc_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out, B, H, W, C / 4,
                     in_pitch, out_pitch);
  return check_launch("maxpool2x2_nhwc_kernel");
}

extern "C" int dd3d_upsample2x_add_nhwc(float* fine, const float* coarse, int32_t B, int32_t H, int32_t W, int32_t C, int32_t fine_pitch,
                                        int32_t coarse_pitch, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(fine && coarse && B > 0, "dd3d_upsample2x_add_nhwc: bad arguments");
  DD3D_REQUIRE((H % 2) == 0 && (W % 2) == 0, "dd3d_upsample2x_add_nhwc: H=%d W=%d must be even", H, W);
  DD3D_REQUIRE((C % 4) == 0 && (fine_pitch % 4) == 0 && (coarse_pitch % 4) == 0, "dd3d_upsample2x_add_nhwc: C / pitches must be multiples of 4");
  const long total = (long)B * H * W * (C / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(upsample2x_add_nhwc_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), fine, coarse, B, H, W, C / 4,
                     fine_pitch, coarse_pitch);
  return check_launch("upsample2x_add_nhwc_kernel");
}

extern "C" int dd3d_maxpool3x3s2_ceil_nhwc(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t in_pitch,
                                           int32_t out_pitch, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(in && out && B > 0 && H >= 3 && W >= 3, "dd3d_maxpool3x3s2_ceil_nhwc: bad arguments");
  DD3D_REQUIRE((C % 4) == 0 && (in_pitch % 4) == 0 && (out_pitch % 4) == 0, "dd3d_maxpool3x3s2_ceil_nhwc: C / pitches must be multiples of 4");
  int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;  // ceil((H-3)/2) + 1
  if ((Ho - 1) * 2 >= H) --Ho;                              // last window must start inside the input (PyTorch rule)
  if ((Wo - 1) * 2 >= W) --Wo;
  const long total = (long)B * Ho * Wo * (C / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(maxpool3x3s2_ceil_nhwc_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out, B, H, W, Ho, Wo,
                     C / 4, in_pitch, out_pitch);
  return check_launch("maxpool3x3s2_ceil_nhwc_kernel");
}

extern "C" int dd3d_ese_nhwc(const float* x, const float* identity, float* out, const float* fc_w, const float* fc_b, float* partial,
                             float* gate, int32_t B, int32_t HW, int32_t C, int32_t x_pitch, int32_t id_pitch, int32_t out_pitch,
                             int32_t rsplit, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(x && out && fc_w && fc_b && partial && gate && B > 0 && HW > 0, "dd3d_ese_nhwc: bad arguments");
  DD3D_REQUIRE((C % 4) == 0 && (x_pitch % 4) == 0 && (out_pitch % 4) == 0 && (!identity || (id_pitch % 4) == 0),
               "dd3d_ese_nhwc: C / pitches must be multiples of 4");
  DD3D_REQUIRE(rsplit >= 1 && rsplit <= 1024, "dd3d_ese_nhwc: rsplit=%d", rsplit);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gap_partial_nhwc_kernel, dim3((C + 63) / 64, rsplit, B), dim3(256), 0, st, x, partial, HW, C, x_pitch, rsplit);
  int rc = check_launch("gap_partial_nhwc_kernel");
  if (rc != DD3D_OK) return rc;
  hipLaunchKernelGGL(ese_gate_kernel, dim3((C + 3) / 4, B), dim3(256), 0, st, partial, fc_w, fc_b, gate, C, rsplit, 1.0f / (float)HW);
  rc = check_launch("ese_gate_kernel");
  if (rc != DD3D_OK) return rc;
  const long total = (long)B * HW * (C / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(scale_add_nhwc_kernel, dim3(grid), dim3(256), 0, st, x, gate, identity, out, B, HW, C / 4, x_pitch, id_pitch, out_pitch);
  return check_launch("scale_add_nhwc_kernel");
}

namespace dd3d {
// The two halves of the f16x2 range guard folded into two words of a rank's exchange record, so that after the all_gather every rank
// sees every rank's verdict and all of them act on the same step:  out[0] = *status (DD3D_STATUS_* bits),  out[1] = 1 when some watched
// launch stored a nonzero sampled maximum below `floor` (maxima: amax[launch][16 sub-maxima, 32 floats apart]).
__global__ __launch_bounds__(256) void fold_range_flags_kernel(const int32_t* status, const float* amax, int n, float floor, int32_t* out) {
  __shared__ int low;
  if (threadIdx.x == 0) low = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, amax[((long)i * 16 + j) * 32]);
    if (m > 0.f && m < floor) low = 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = status ? *status : 0;
    out[1] = low;
  }
}
}  // namespace dd3d

extern "C" int dd3d_fold_range_flags(const int32_t* status, const float* amax, int32_t n_launches, float floor, int32_t* out, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(out && n_launches >= 0 && (n_launches == 0 || amax), "dd3d_fold_range_flags: bad arguments");
  hipLaunchKernelGGL(fold_range_flags_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), status, amax, n_launches, floor, out);
  return check_launch("fold_range_flags_kernel");
}

namespace dd3d {
// Everything the host reads after a forward in ONE contiguous run of 4-byte words (one D2H instead of a blocking read per field):
//   out[0] = *status, out[1] = G, out[2] = n_launches, out[3] = n_flag_recs, out[4 .. 4 + G) = det_count,
//   then n_launches floats = every watched launch's maximum (over its 16 sub-maxima), then 2 words per exchanged record (the ranks'
//   range-guard verdicts as dd3d_fold_range_flags wrote them: flags + r * flag_stride).
__global__ __launch_bounds__(256) void pack_readback_kernel(const int32_t* det_count, int G, const int32_t* status, const float* amax, int n,
                                                            const int32_t* flags, int nrec, long flag_stride, int32_t* out) {
  const int t = threadIdx.x;
  if (t == 0) {
    out[0] = status ? *status : 0;
    out[1] = G, out[2] = n, out[3] = nrec;
  }
  for (int i = t; i < G; i += 256) out[4 + i] = det_count[i];
  float* fo = reinterpret_cast<float*>(out + 4 + G);
  for (int i = t; i < n; i += 256) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, amax[((long)i * 16 + j) * 32]);
    fo[i] = m;
  }
  for (int i = t; i < 2 * nrec; i += 256) out[4 + G + n + i] = flags[(long)(i >> 1) * flag_stride + (i & 1)];
}
}  // namespace dd3d

extern "C" int dd3d_pack_readback(const int32_t* det_count, int32_t G, const int32_t* status, const float* amax, int32_t n_launches,
                                  const int32_t* flags, int32_t n_flag_recs, int64_t flag_stride, int32_t* out, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(out && G >= 0 && n_launches >= 0 && n_flag_recs >= 0 && (G == 0 || det_count) && (n_launches == 0 || amax) && (n_flag_recs == 0 || flags),
               "dd3d_pack_readback: bad arguments");
  hipLaunchKernelGGL(pack_readback_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), det_count, G, status, amax, n_launches, flags,
                     n_flag_recs, (long)flag_stride, out);
  return check_launch("pack_readback_kernel");
}

extern "C" int dd3d_invert_intrinsics(const float* K, float* inv_K, int32_t B, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(K && inv_K && B > 0, "dd3d_invert_intrinsics: bad arguments");
  hipLaunchKernelGGL(invert3x3_kernel, dim3((B + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), K, inv_K, B);
  return check_launch("invert3x3_kernel");
}

extern "C" int dd3d_aligned_bilinear_scale(const float* src, float* out, const float* inv_K, int32_t B, int32_t h, int32_t w, int32_t pitch,
                                           int32_t factor, int32_t offset_half, float focal_factor, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(src && out && B > 0 && h > 0 && w > 0 && factor >= 1 && pitch >= 1, "dd3d_aligned_bilinear_scale: bad arguments");
  DD3D_REQUIRE(focal_factor <= 0.f || inv_K, "dd3d_aligned_bilinear_scale: focal scaling needs inv_K");
  const long total = (long)B * h * w * factor * factor;
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(aligned_bilinear_scale_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, out, inv_K, B, h, w, pitch,
                     factor, offset_half, focal_factor);
  return check_launch("aligned_bilinear_scale_kernel");
}

extern "C" int dd3d_resize_bilinear_u8(const dd3d_resize_args* a, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(a && a->src && a->dst && a->C > 0 && a->H > 0 && a->W > 0 && a->new_h > 0 && a->new_w > 0, "dd3d_resize_bilinear_u8: bad arguments");
  const bool hpass = a->new_w != a->W, vpass = a->new_h != a->H;
  DD3D_REQUIRE(!hpass || (a->lo_w && a->cnt_w && a->kk_w && a->ksize_w > 0), "dd3d_resize_bilinear_u8: horizontal coefficients missing");
  DD3D_REQUIRE(!vpass || (a->lo_h && a->cnt_h && a->kk_h && a->ksize_h > 0), "dd3d_resize_bilinear_u8: vertical coefficients missing");
  DD3D_REQUIRE(!(hpass && vpass) || a->tmp, "dd3d_resize_bilinear_u8: two passes need the C x H x new_w scratch image");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  auto grid_for = [](long total) { return (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384); };
  if (!hpass && !vpass) {  // same size: plain copy into the destination canvas
    for (int c = 0; c < a->C; ++c) {
      hipError_t e = hipMemcpy2DAsync(a->dst + c * a->dst_plane, a->dst_row, a->src + c * a->src_plane, a->src_row, a->W, a->H, hipMemcpyDeviceToDevice, st);
      DD3D_REQUIRE(e == hipSuccess, "dd3d_resize_bilinear_u8: copy failed: %s", hipGetErrorString(e));
    }
    return DD3D_OK;
  }
  const uint8_t* src = a->src;
  long src_plane = a->src_plane;
  int src_row = a->src_row;
  if (hpass) {  // horizontal first, as ImagingResample
    uint8_t* out = vpass ? a->tmp : a->dst;
    const long op = vpass ? (long)a->H * a->new_w : a->dst_plane;
    const int orow = vpass ? a->new_w : a->dst_row;
    hipLaunchKernelGGL(resample_u8_kernel, dim3(grid_for((long)a->C * a->H * a->new_w)), dim3(256), 0, st, src, out, a->C, a->H, a->new_w, src_plane,
                       src_row, op, orow, a->lo_w, a->cnt_w, a->kk_w, a->ksize_w, 1);
    int rc = check_launch("resample_u8_kernel (horizontal)");
    if (rc != DD3D_OK) return rc;
    src = out, src_plane = op, src_row = orow;
  }
  if (vpass) {
    hipLaunchKernelGGL(resample_u8_kernel, dim3(grid_for((long)a->C * a->new_h * a->new_w)), dim3(256), 0, st, src, a->dst, a->C, a->new_h, a->new_w,
                       src_plane, src_row, a->dst_plane, a->dst_row, a->lo_h, a->cnt_h, a->kk_h, a->ksize_h, 0);
    return check_launch("resample_u8_kernel (vertical)");
  }
  return DD3D_OK;
}

