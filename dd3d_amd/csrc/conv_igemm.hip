// Implicit-GEMM convolution for gfx950 (MI355X) on the exact-f32 matrix pipe
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise an fmaf chain).
//
// GEMM view:  M = B*Ho*Wo output pixels (rows), N = Cout, K = KH*KW*Cin.
//   A[m,k]   gathered on the fly from the NHWC input (im2col never materialised),
//   Wp[n,k]  packed filter, k contiguous,
// both staged per K-tile (BK = 32) through LDS in [row][BK+4] images and read back as
// ds_read_b128 fragments: lane l supplies row (l&31) and the 4 consecutive k at 8*s + 4*(l>>5),
// so one 16-byte LDS read feeds four MFMAs.  Each K-tile is (channel-chunk, filter tap), tap
// fastest, so the nine taps of a 3x3 filter re-read the same input lines back-to-back (L1/L2 hits).
//
// One launch covers many *segments* (FPN levels x head towers) through a tile table, so the tiny
// P6/P7 levels ride along with P3 instead of costing their own under-filled launches, and the
// per-level BatchNorm of the shared towers becomes a per-segment (scale, bias) epilogue.
//
// Block = 256 threads = 4 wave64; block tile = (TM*32*WM) x (TN*32*WN); double-buffered LDS with
// register prefetch of the next K-tile; 1-D grid remapped so that blocks that share input rows run
// on the same XCD (private L2).
#include "common.h"

namespace dd3d {

constexpr int BK = 32;
constexpr int LDS_ROW = BK + 4;  // floats per LDS row: 144 B keeps the 16-B slots of 16 rows distinct mod 256 B

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvKArgs {
  const dd3d_conv_seg* segs;
  const int32_t* tiles;
  float* ws;
  int ntiles, nn;  // m-tiles (all segments), n-tiles
  int KH, KW, stride, pad, Cin, N, Kpad, Npad;
  int T;         // KH*KW
  int cc_shift;  // log2(CC) when Cin < 32
  int kw_magic;  // (65536 / KW) + 1 : tap / KW == (tap * kw_magic) >> 16 for tap < 64
  int relu, splitk, ws_rows, kt_per_split;
  int N4;  // round_up(N, 4): row stride of the split-K workspace
};

// Global-address-space views: the segment descriptor is loaded from memory, so without these casts the compiler
// must emit FLAT loads, which tick lgkmcnt as well and would make every LDS wait also wait for the HBM prefetch.
typedef const float __attribute__((address_space(1))) * gcfp;
typedef float __attribute__((address_space(1))) * gfp;
typedef const f32x4 __attribute__((address_space(1))) * gcf4p;
typedef f32x4 __attribute__((address_space(1))) * gf4p;
__device__ __forceinline__ gcfp as_g(const float* p) { return (gcfp)p; }
__device__ __forceinline__ gfp as_g(float* p) { return (gfp)p; }

template <int TM, int TN, int WM, int WN, bool SMALLC>
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

  // ---- XCD-aware block remap (bijective): blocks dispatched to the same XCD (bid % 8) get a contiguous range of
  //      logical tiles, n fastest, so the n-tiles that share an A row band hit the same L2.
  const int nwg = a.ntiles * a.nn;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int mt = bid / a.nn;
  const int nt = bid - mt * a.nn;
  const int seg_id = a.tiles[2 * mt];
  const int m0 = a.tiles[2 * mt + 1];
  const int n0 = nt * BN;
  const dd3d_conv_seg s = a.segs[seg_id];
  const gcfp g_in = as_g(s.in);
  const gcfp g_w = as_g(s.w);

  const int nk = a.Kpad / BK;
  int kt_begin = 0, kt_end = nk;
  if (a.splitk > 1) {
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

  f32x4 ra[AP], rb[BP];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto load_tile = [&](int kt) {
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
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool ok = tap_ok && (unsigned)(a_hi0[p] + dh) < (unsigned)s.H && (unsigned)(a_wi0[p] + dw) < (unsigned)s.W;
      ra[p] = zero4;
      if (ok) ra[p] = *(gcf4p)(g_in + a_base[p] + koff);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int n = n0 + p * 32 + arow;
      rb[p] = zero4;
      if (n < a.Npad) rb[p] = *(gcf4p)(g_w + (long)n * a.Kpad + kt * BK + avec * 4);
    }
  };

  auto store_tile = [&](int buf) {
    float* As = smem + buf * (BM + BN) * LDS_ROW;
    float* Bs = As + BM * LDS_ROW;
#pragma unroll
    for (int p = 0; p < AP; ++p) *reinterpret_cast<f32x4*>(As + (p * 32 + arow) * LDS_ROW + avec * 4) = ra[p];
#pragma unroll
    for (int p = 0; p < BP; ++p) *reinterpret_cast<f32x4*>(Bs + (p * 32 + arow) * LDS_ROW + avec * 4) = rb[p];
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

  if (kt_begin < kt_end) {
    load_tile(kt_begin);
    store_tile(0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int cur = (kt - kt_begin) & 1;
      const bool more = kt + 1 < kt_end;
      if (more) load_tile(kt + 1);  // global -> registers, in flight under the MFMAs below
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
      if (more) store_tile(cur ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  //      out = max(lo, acc*scale + bias (+ residual)); one lane owns one output channel per 32-wide column block.
  const gcfp g_res = as_g(s.res);
  const gfp g_out = as_g(s.out);
  const gfp g_ws = as_g(a.ws);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
    if (n >= a.N) continue;
    float sc = 1.f, bi = 0.f, lo = -INFINITY;
    if (a.splitk == 1) {
      sc = as_g(s.scale)[n];
      bi = as_g(s.bias)[n];
      if (s.lo) lo = as_g(s.lo)[n];
      if (a.relu) lo = fmaxf(lo, 0.f);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m >= s.M) continue;
        if (a.splitk > 1) {
          g_ws[((long)blockIdx.y * a.ws_rows + s.ws_row0 + m) * a.N4 + n] = acc[i][j][r];
        } else {
          float v = acc[i][j][r] * sc + bi;
          if (s.res_mode == 1) v += g_res[(long)m * s.res_pitch + n];
          g_out[(long)m * s.out_pitch + n] = fmaxf(v, lo);
        }
      }
    }
  }
}

// Split-K second pass: sum the partial slabs and apply the epilogue.  grid = (m-tiles, BM / 8): one block per 8 output rows,
// 16 B per lane, so even a 30-tile layer spreads over the whole chip.
constexpr int RED_ROWS = 8;
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvKArgs a) {
  const int seg_id = a.tiles[2 * blockIdx.x];
  const int m0 = a.tiles[2 * blockIdx.x + 1] + blockIdx.y * RED_ROWS;
  const dd3d_conv_seg s = a.segs[seg_id];
  const gcfp g_ws = as_g(a.ws);
  const gcfp g_res = as_g(s.res);
  const gfp g_out = as_g(s.out);
  const int n4 = a.N4 >> 2;  // workspace rows are N4 = round_up(N, 4) floats wide; columns >= N are never written nor used
  for (int idx = threadIdx.x; idx < RED_ROWS * n4; idx += 256) {
    const int r = idx / n4;
    const int c = (idx - r * n4) * 4;
    const int m = m0 + r;
    if (m >= s.M) break;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < a.splitk; ++z) v += *(gcf4p)(g_ws + ((long)z * a.ws_rows + s.ws_row0 + m) * a.N4 + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = c + e;
      if (n >= a.N) break;
      float o = v[e] * as_g(s.scale)[n] + as_g(s.bias)[n];
      if (s.res_mode == 1) o += g_res[(long)m * s.res_pitch + n];
      float lo = s.lo ? as_g(s.lo)[n] : -INFINITY;
      if (a.relu) lo = fmaxf(lo, 0.f);
      g_out[(long)m * s.out_pitch + n] = fmaxf(o, lo);
    }
  }
}

template <int TM, int TN, int WM, int WN>
static int launch_cfg(const ConvKArgs& ka, bool smallc, hipStream_t st) {
  constexpr int BM = TM * 32 * WM, BN = TN * 32 * WN;
  const size_t lds = (size_t)2 * (BM + BN) * LDS_ROW * sizeof(float);
  dim3 grid(ka.ntiles * ka.nn, ka.splitk, 1), block(256, 1, 1);
  if (smallc) {
    auto k = conv_igemm_f32_kernel<TM, TN, WM, WN, true>;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_done = true;
    }
    hipLaunchKernelGGL(k, grid, block, lds, st, ka);
  } else {
    auto k = conv_igemm_f32_kernel<TM, TN, WM, WN, false>;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_done = true;
    }
    hipLaunchKernelGGL(k, grid, block, lds, st, ka);
  }
  int rc = check_launch("conv_igemm_f32_kernel");
  if (rc != DD3D_OK) return rc;
  if (ka.splitk > 1) {
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(ka.ntiles, BM / RED_ROWS), dim3(256), 0, st, ka);
    rc = check_launch("conv_splitk_reduce_kernel");
  }
  return rc;
}

}  // namespace dd3d

extern "C" int dd3d_conv_tile_shape(int32_t tile_cfg, int32_t* bm, int32_t* bn) {
  static const int shapes[DD3D_TILE_COUNT][2] = {{128, 128}, {128, 64}, {64, 64}, {128, 32}, {64, 128}};
  DD3D_REQUIRE(tile_cfg >= 0 && tile_cfg < DD3D_TILE_COUNT, "dd3d_conv_tile_shape: unknown tile_cfg %d", tile_cfg);
  *bm = shapes[tile_cfg][0];
  *bn = shapes[tile_cfg][1];
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
  DD3D_REQUIRE(L->splitk == 1 || L->workspace, "dd3d_conv2d_igemm_f32: split-K needs a workspace of splitk*ws_rows*round_up(N,4) floats");
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
  ka.ws_rows = L->ws_rows;
  ka.N4 = (L->N + 3) / 4 * 4;
  const int nk = L->Kpad / 32;
  ka.kt_per_split = ceil_div(nk, L->splitk);
  const bool smallc = L->Cin < 32;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (L->tile_cfg) {
    case DD3D_TILE_128x128: return launch_cfg<2, 2, 2, 2>(ka, smallc, st);
    case DD3D_TILE_128x64: return launch_cfg<2, 1, 2, 2>(ka, smallc, st);
    case DD3D_TILE_64x64: return launch_cfg<1, 1, 2, 2>(ka, smallc, st);
    case DD3D_TILE_128x32: return launch_cfg<1, 1, 4, 1>(ka, smallc, st);
    case DD3D_TILE_64x128: return launch_cfg<1, 2, 2, 2>(ka, smallc, st);
  }
  return DD3D_E_INVALID;
}
