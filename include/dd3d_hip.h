/*
 * dd3d_hip.h -- C ABI of the MI355X (gfx950) DD3D forward-path library  (libdd3d_hip.so)
 *
 * The reference (TRI-ML/dd3d) has no FFI of its own for this path: it is pure Python and reaches
 * native code only through third-party wheels (cuDNN via torch, torchvision.ops.nms,
 * detectron2._C.nms_rotated, pytorch3d).  The entry points below are what a maintainer would bind
 * INSTEAD of those calls; each one cites the reference call site(s) it replaces.  See
 * INTEGRATION.md for the ctypes stub on the reference side.
 *
 * Conventions
 *   - plain C, no torch types: device pointers, int sizes, an opaque hipStream_t passed as void*.
 *   - every buffer is owned by the caller (PyTorch caching allocator); the library allocates no
 *     persistent device memory and never synchronises the device.
 *   - kernels are enqueued asynchronously on `stream`; calls are safe under hipGraph stream capture.
 *   - return value: 0 = ok, <0 = error (DD3D_E_*); dd3d_last_error() gives a message (thread-local).
 *   - activations are NHWC fp32 with an explicit per-pixel pitch, so a conv can read / write a
 *     channel slice of a wider buffer (this is how torch.cat in dla.py:161 disappears), and / or
 *     "split planes": [channel chunk c/32][pixel (b,h,w)][plane][32] 16-bit terms of the arithmetic
 *     mode (DD3D_MATH_*), the form in which one convolution hands its output to the next so that
 *     the consumer streams it into LDS by LDS-DMA with no conversion work (a channel slice of a
 *     wider buffer = a run of whole chunk images, so the concat-by-placement carries over).
 */
#ifndef DD3D_HIP_H
#define DD3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DD3D_ABI_VERSION 6

#define DD3D_OK 0
#define DD3D_E_INVALID (-1)  /* bad argument (shape / alignment / enum) */
#define DD3D_E_LAUNCH (-2)   /* hip launch error */
#define DD3D_E_UNSUPPORTED (-3)

#define DD3D_MAX_LEVELS 8
#define DD3D_CAND_FIELDS 22 /* SoA fields of a decoded candidate, see dd3d_fcos_select_decode */
#define DD3D_DET_FIELDS 32  /* AoS fields of a final detection, see dd3d_nms_finalize / dd3d_bev_nms_aggregate */

int dd3d_abi_version(void);
const char* dd3d_last_error(void);
/* "gfx950" -- the only architecture the code objects are built for. */
const char* dd3d_arch(void);
/* "" for the product build.  Otherwise the -DDD3D_...=... knobs the library's translation units were compiled with ("file: KNOB=value; ..."):
 * an A/B variant (tests/tools/build_variant.sh; every knob still computes correct results).  dd3d_amd/hip.py refuses such a library unless the
 * caller selected it explicitly. */
const char* dd3d_build_flags(void);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on f32 MFMA (v_mfma_f32_32x32x2_f32), fused epilogue.
 * Replaces: every nn.Conv2d / detectron2 Conv2d(+FrozenBN/BN +ReLU) on the path --
 *   tridet/modeling/feature_extractor/dla.py:50-62,160-167,233-247,346-355 (DLA blocks, roots),
 *   detectron2 FPN lateral/output/top-block convs [ext] (built at dla.py:550-557),
 *   tridet/modeling/dd3d/fcos2d.py:137-152 and fcos3d.py:163-180 (towers, predictors, Scale/Offset),
 *   plus the residual add (dla.py:59-60) and torch.cat before a Root (dla.py:161), which are folded
 *   into the epilogue / the pitch addressing.
 *
 * One launch processes `nsegs` segments that share the filter geometry (KH,KW,stride,pad,Cin,N) but
 * have their own tensors: e.g. the 5 FPN levels x 3 head towers of one tower layer = 15 segments.
 *
 *   out[m, n] = max( lo[n],  relu?( sum_k A[m,k] * Wp[n,k] * scale[n] + bias[n] + res[...] ) )
 *
 * K ordering of the packed filter Wp[Npad][Kpad] (n-major, k contiguous):
 *   k = (c / CC) * (KH*KW*CC) + (kh*KW + kw) * CC + (c % CC),   CC = min(Cin, 32),
 * zero-padded to Kpad (multiple of 32) and Npad (multiple of 32).  Cin must be 4, 16 or a multiple of 32.
 * ------------------------------------------------------------------------------------------------ */
typedef struct dd3d_conv_seg {  /* array lives in DEVICE memory */
  const float* in;     /* NHWC input, already offset to the first input channel                  */
  const float* w;      /* packed filter [Npad][Kpad]                                             */
  const float* scale;  /* [N] folded-norm / Scale multiplier                                     */
  const float* bias;   /* [N] folded-norm shift / conv bias / Offset                             */
  const float* lo;     /* [N] per-channel lower clamp (0 => ReLU on that channel, -inf => none), or NULL */
  const float* res;    /* residual source or NULL: f32 NHWC (res_mode 1) or split planes (res_mode 2 / 3, ABI 4) */
  float* out;          /* NHWC output, already offset to the first output channel                */
  int32_t B, H, W;     /* input batch / height / width                                           */
  int32_t Ho, Wo;      /* output height / width                                                   */
  int32_t in_pitch, out_pitch, res_pitch; /* floats per pixel of the respective buffers          */
  int32_t M;           /* B*Ho*Wo                                                                */
  int32_t res_mode;    /* 0 none; 1 add the f32 value res[m * res_pitch + n] (same pixel) before the clamp;
                          ABI 4, split-plane-input kernels only: the residual is read from SPLIT PLANES of the launch's arithmetic mode,
                          `res` = first chunk of the slice, [ceil(N/32)][pixels][NP][32] terms of value * out_plane_scale:
                          2 same pixel (pixels = M) -- dla.py:59-60 without an f32 twin of the residual source;
                          3 the map at HALF the resolution, pixel (b, ho/2, wo/2) of B x Ho/2 x Wo/2 (Ho, Wo even): the FPN top-down
                            path's F.interpolate(scale 2, nearest) + add [ext d2 FPN.forward] fused into the lateral convolution.
                          DD3D_TILE_256x256_W8 carries no residual (its waves have no registers left for one). */
  const void* in_planes; /* split-plane input [Cin/32][B*H*W][NP][32], first chunk of the slice; read instead of `in` when
                            dd3d_conv_launch.in_planes is set                                    */
  int32_t n_limit;     /* > 0: this segment stores only output channels < n_limit (<= launch N); 0: all N  */
  int32_t reserved;    /* ABI 6, chain launches (dd3d_conv_launch.chain): 1 + index of the segment of THIS launch whose output is this
                          segment's input (in_planes), 0 when the input comes from an earlier launch.  0 in every other launch. */
  void* out_planes;    /* split-plane output [ceil(N/32)][M][NP][32] (first chunk of the slice) or NULL; `out` may then be NULL.
                          Channels N .. 32*ceil(N/32)-1 of the last chunk are written as zeros.  Split-operand modes only. */
} dd3d_conv_seg;

typedef struct dd3d_conv_launch {  /* host memory */
  const dd3d_conv_seg* segs; /* device */
  const int32_t* tiles;      /* device, ntiles x {seg, m0} */
  float* workspace;          /* device, splitk * ntiles * ceil(N/BN) * BM * BN floats (raw accumulator slabs) when splitk > 1, else NULL */
  int32_t nsegs, ntiles;
  int32_t KH, KW, stride, pad;
  int32_t Cin, N, Kpad, Npad;
  int32_t relu;      /* 1: clamp every output channel at 0 */
  int32_t splitk;    /* >= 1 */
  int32_t math_mode; /* DD3D_MATH_* */
  int32_t tile_cfg;  /* DD3D_TILE_* */
  const float* zero_page; /* device, >= 128 B of zeros, 16-B aligned: source of padded taps for the LDS-DMA kernel
                             (NULL selects the register-staged kernel) */
  int32_t* tile_counters; /* device, ceil(N/BN) * ntiles int32, ZERO on entry when splitk > 1 (else NULL).  The slice of an
                             output tile that arrives last sums the partial slabs in slice order and applies the epilogue
                             inside the same launch; the counters are zero again when the launch has completed. */
  const dd3d_conv_seg* seg0_host; /* HOST copy of segs[0 .. nsegs-1], or NULL.  With nsegs == 1 the descriptor then travels in the kernel
                                     arguments (tiles are taken as m0 = i * BM) and the device copies are not read.  With a host copy the
                                     library VALIDATES every segment's residual contract before launching (ABI 5); without one (NULL) the
                                     caller vouches for the device-side descriptors:
                                       res_mode in 0..3 and res != NULL unless 0; res_mode 2 / 3 only with in_planes;
                                       res_mode 3: even Ho, Wo;  DD3D_TILE_256x256_W8: res_mode 0 (its waves hold no residual);
                                       res_mode 1 with in_planes: res_pitch % 4 == 0 and res_pitch >= (channels stored rounded up to 4)
                                       -- the epilogue reads the residual row in 16-byte pieces. */
  int32_t in_relu; /* 1: the input is rectified on the fly, out = epilogue(conv(max(in, 0))) -- LastLevelP6P7: p7 = conv(relu(p6)).
                      DD3D_MATH_BF16X3 with f32 input only */
  int32_t in_planes; /* 1: every segment reads its split-plane input (seg.in_planes) instead of the f32 one; Cin % 32 == 0 and a
                        split-operand math mode */
  float out_plane_scale; /* DD3D_MATH_F16X2: the split-plane outputs hold value * out_plane_scale (a power of two; 0 = 1) */
  int32_t* status;       /* device int32 OR-ed with DD3D_STATUS_* bits by the kernels, or NULL */
  float* amax;           /* DD3D_MATH_F16X2, split-plane outputs: 16 device floats, 32 floats apart (amax[32 * j], j < 16), or NULL.  The
                            launch folds max |value * out_plane_scale| over a sample of its stored outputs (one wave tile per block) into
                            them (atomic max; non-negative floats compare like their bit patterns).  The caller zeroes them before a forward
                            and reads their maximum afterwards: a tensor whose LARGEST scaled entry is below 2^-5 has lost more than four
                            of its 24 bits to the absolute floor 2^-25 of the half pair (the "underflow" side of the range guard). */
  /* ABI 6 -- dependent segments in ONE launch: the stride-1 3 x 3 convolutions of a DLA level (tridet/modeling/feature_extractor/dla.py:50-62:
   * conv1 -> conv2 + residual -> the next block's conv1 ...; dla.py:233-247 chains the blocks) and the four layers of the head towers
   * (tridet/modeling/dd3d/fcos2d.py:137-152, fcos3d.py:163-180: every (tower, level) is a chain of four convolutions) without launch
   * boundaries between them -- a later layer's first tiles start on the CUs the previous layer's last round leaves idle.
   *   chain = 1: segment i may read, as its input (dd3d_conv_seg.reserved names the producing segment) and as a split-plane residual
   *   (res_mode 2), what segments < i of this launch write.  Requirements (validated): 3 x 3 / stride 1 / pad 1 on split-plane input (the
   *   row-shared kernels), split-plane outputs only (out == NULL), res_mode 0 or 2, no n_limit, producers before consumers in `tiles`
   *   (segment-major), a consumer has its producer's B, H, W and reads the producer's out_planes as in_planes, seg0_host given.
   *   chain_sync: device int32 [1 + ntiles], ZERO on entry; the launch leaves it zero.  [0] counts finished tiles, [1 + t] the finished
   *   n-tiles of m-tile t (index into `tiles`).  chain_tile0: device int32 [nsegs], index in `tiles` of every segment's first m-tile.
   *   A consumer block polls its producer's m-tiles that cover the rows it loads (bounded: DD3D_STATUS_CHAIN_TIMEOUT in *status instead
   *   of a hang); the hand-over uses write-through stores and agent-coherent loads, no cache-wide fences. */
  int32_t chain;
  int32_t* chain_sync;
  const int32_t* chain_tile0;
} dd3d_conv_launch;

/* Arithmetic of the implicit GEMM (results agree to f32 rounding level; both accumulate in f32):
 *   DD3D_MATH_F32     v_mfma_f32_32x32x2_f32 on f32 operands; segment.w = Wp[Npad][Kpad] f32
 *   DD3D_MATH_BF16X3  each f32 operand split exactly into 3 bf16 terms, 6 cross products on v_mfma_f32_32x32x16_bf16;
 *                     segment.w = Wp3[Npad][Kpad/32][3][32] bf16 (planes hi, mid, lo of the same k order); Cin % 32 == 0,
 *                     tiles 256x128 / 128x128 / 128x64 / 64x128 */
#define DD3D_MATH_F32 0
#define DD3D_MATH_BF16X3 1
/* Reduced split-operand modes (what BASELINE.json's "bf16 inference" configurations name); split-plane inputs only
 * (dd3d_split_planes converts an f32 tensor), f32 accumulate, filters split the same way, round-to-nearest-even terms:
 *   DD3D_MATH_BF16X2  x ~ hi + lo (two bf16 terms, ~17 significand bits), 3 cross products: half the matrix work of BF16X3
 *   DD3D_MATH_BF16    x ~ bf16(x), one product: plain bf16 operands */
#define DD3D_MATH_BF16X2 2
#define DD3D_MATH_BF16 3
/*   DD3D_MATH_F16X2   x * S = hi + lo, two IEEE half terms by round-to-nearest (2 x 11 significand bits: the pair carries the 24 bits
 *                     of an f32 wherever lo is a normal half), 3 cross products on v_mfma_f32_32x32x16_f16; the dropped lo*lo term is
 *                     <= 2^-24 |a*b| like the dropped terms of DD3D_MATH_BF16X3 -> f32-equivalent at HALF its matrix work, inside the
 *                     half format's exponent range.  S: activations carry the power-of-two `plane_scale` of the launch / of
 *                     dd3d_split_planes (|x * S| <= 65504 or the status word is set; terms below 2^-24 / S are lost); filters are
 *                     scaled per output row by the caller, who divides the products of the scales out of `scale[n]`. */
#define DD3D_MATH_F16X2 4
#define DD3D_STATUS_F16_OVERFLOW 1 /* bit set in *status when a value left the half range while being split */
#define DD3D_STATUS_CHAIN_TIMEOUT 2 /* bit set when a block of a chain launch (dd3d_conv_launch.chain) gave up waiting for its producers */
/* DD3D_MATH_F16X2 range guard, for the multi-GPU exchange: out[0] = *status (may be NULL: 0), out[1] = 1 when one of the n_launches
 * watched launches stored a nonzero sampled maximum below `floor` (amax: [n_launches][16][32] floats, see dd3d_conv_launch.amax).  The
 * two words travel in a rank's record; after the all_gather every rank sees every rank's verdict and all act on the same step. */
int dd3d_fold_range_flags(const int32_t* status, const float* amax, int32_t n_launches, float floor, int32_t* out, void* stream);
/* What the host reads after a forward, packed into ONE contiguous run of 4-byte words so that a single asynchronous device-to-host
 * copy (into pinned memory, enqueued behind the forward) replaces one blocking read per field -- the reference's forward ends by
 * returning Instances (tridet/modeling/dd3d/core.py:153-164), which needs the detection counts on the host:
 *   out[0] = *status (NULL: 0), out[1] = G, out[2] = n_launches, out[3] = n_flag_recs, out[4 .. 4 + G) = det_count[0 .. G),
 *   then n_launches floats: max over the 16 sub-maxima of every watched launch (amax as in dd3d_conv_launch.amax),
 *   then 2 * n_flag_recs words: flags[r * flag_stride + {0, 1}] (every rank's dd3d_fold_range_flags words inside the gathered records).
 * `out` holds 4 + G + n_launches + 2 * n_flag_recs words.  One block; ordered on `stream` behind the forward's last launch. */
int dd3d_pack_readback(const int32_t* det_count, int32_t G, const int32_t* status, const float* amax, int32_t n_launches,
                       const int32_t* flags, int32_t n_flag_recs, int64_t flag_stride, int32_t* out, void* stream);
/* planes per value of a math mode (0 for DD3D_MATH_F32) */
int dd3d_math_planes(int32_t math_mode);

#define DD3D_TILE_128x128 0
#define DD3D_TILE_128x64 1
#define DD3D_TILE_64x64 2
#define DD3D_TILE_128x32 3
#define DD3D_TILE_64x128 4
#define DD3D_TILE_256x128 5    /* DD3D_MATH_BF16X3 only, like the three below */
#define DD3D_TILE_128x128_W4 6 /* 4-wave blocks (256 threads) small enough in LDS for two per CU */
#define DD3D_TILE_64x64_W4 7
#define DD3D_TILE_128x64_W4 8
#define DD3D_TILE_128x64_K2 9  /* 8 waves, two K-tiles (64 k) per barrier */
#define DD3D_TILE_64x128_K2 10
#define DD3D_TILE_64x64_W4K2 11
/* split-plane kernels only: wave tiles of 128 x 64 / 64 x 128 outputs (8 accumulator blocks per wave) -- one ds_read_b128 per two MFMAs
 * in the two-term modes instead of two per three; 4-wave blocks hold one wave per SIMD (up to 512 registers each) */
#define DD3D_TILE_256x128_T42 12 /* 4 waves (2 x 2), wave tile 128 x 64 */
#define DD3D_TILE_128x256_T24 13 /* 4 waves (2 x 2), wave tile 64 x 128 */
#define DD3D_TILE_256x256_W8 14  /* 8 waves (2 x 4), wave tile 128 x 64; one- and two-term modes only */
#define DD3D_TILE_128x32_W4 15   /* split-plane kernels: 4 waves (4 x 1), 32 output columns -- the narrow predictors (N <= 32: fcos2d.py:96-110) */
#define DD3D_TILE_192x256_W8 16  /* row-shared 3 x 3 kernel only: 8 waves (2 x 4), wave tile 96 x 64 -- launches whose 256-row tiles fill only part of the chip
                                  * (the merged FPN output convolutions: 210 blocks instead of 158 on 256 CUs); no residual, no split-K, like DD3D_TILE_256x256_W8 */
#define DD3D_TILE_COUNT 17
/* rows (M) and columns (N) of a block tile for a DD3D_TILE_* id; returns 0 on success */
int dd3d_conv_tile_shape(int32_t tile_cfg, int32_t* bm, int32_t* bn);
/* B / A ring depths (NSB, NSA) of the row-shared split-plane kernel instantiation that `tile_cfg` launches in `math_mode` -- the 5th and
 * 8th template arguments rocprofv3 prints for `conv_igemm_planes_row_kernel<...>` (profile bookkeeping; DD3D_E_UNSUPPORTED when the pair
 * has no such kernel) */
int dd3d_conv_row_rings(int32_t tile_cfg, int32_t math_mode, int32_t* nsb, int32_t* nsa);
int dd3d_conv2d_igemm_f32(const dd3d_conv_launch* launch, void* stream);

/* f32 NHWC -> split planes of `math_mode` (the entry into the plane format for tensors a non-convolution kernel produced: pooled
 * maps, the FPN top-down sums, eSE outputs, the stem), optionally rectified (LastLevelP6P7: p7 = conv(relu(p6)) [ext]).
 *   in [M][in_pitch] f32, channels [0, C), C % 32 == 0;  out [C/32][M][NP][32] 16-bit terms of in * plane_scale (plane_scale: see
 *   DD3D_MATH_F16X2; ignored by the bf16 modes); status: see dd3d_conv_launch */
int dd3d_split_planes(const float* in, void* out, int32_t M, int32_t C, int32_t in_pitch, int32_t math_mode, int32_t relu, float plane_scale,
                      int32_t* status, void* stream);

/* dd3d_maxpool2x2_nhwc / dd3d_upsample2x_add_nhwc that ALSO write the split planes of their result (one launch instead of the kernel
 * followed by dd3d_split_planes): out_planes / fine_planes = [C/32][pixels][NP][32] terms of `math_mode` (first chunk of the slice),
 * C % 32 == 0.  dd3d_maxpool2x2_planes: `out` (f32) may be NULL.  plane_scale / status as in dd3d_split_planes. */
int dd3d_maxpool2x2_planes(const float* in, float* out, void* out_planes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t in_pitch,
                           int32_t out_pitch, int32_t math_mode, float plane_scale, int32_t* status, void* stream);
/* ABI 4: 2x2 / stride 2 max-pool (dla.py:228-231 Tree.downsample) of a map held as split planes ONLY: in_planes [C/32][B*H*W][NP][32] ->
 * out_planes [C/32][B*(H/2)*(W/2)][NP][32]; per channel the terms of the largest of the four values are copied (no re-split). */
int dd3d_maxpool2x2_planes_in(const void* in_planes, void* out_planes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t math_mode, void* stream);
int dd3d_upsample2x_add_planes(float* fine, const float* coarse, void* fine_planes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t fine_pitch,
                               int32_t coarse_pitch, int32_t math_mode, float plane_scale, int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small-channel convolution (the full-resolution stem: dla.py:271-280,327-344 base_layer / level0 / level1,
 * vovnet.py stem_1), DD3D_MATH_BF16X3 arithmetic, + per-channel scale/bias (+ lower clamp / ReLU).
 * The block stages the input patch of its output tile once (f32 -> bf16 hi/mid/lo planes) and feeds
 * v_mfma_f32_16x16x32_bf16 straight from it; no im2col K loop.
 *   in  f32 NHWC [B][H][W] rows of in_pitch floats, channels [0, Cin) used, Cin in {4, 16}
 *   w3  bf16 [chunk][plane][Npad16][32]: 32-k chunks of the k order
 *         Cin 4 : k = (dh*8 + dw)*4 + c      (one chunk per filter row, tap slots >= KW zero)
 *         Cin 16: k = (dh*KW + dw)*16 + c    (two taps per chunk, an odd tap count zero-padded)
 *       planes hi / mid / lo of the exact 3-way bf16 split, Npad16 = round_up(N, 16)
 *   out f32 NHWC rows of out_pitch floats, channels [0, N)
 * Instantiated: (Cin 4, 7x7, s1, p3, N<=16) (Cin 16, 3x3, s1, p1, N<=16) (Cin 16, 3x3, s2, p1, N<=32) (Cin 4, 3x3, s2, p1, N<=64);
 * dd3d_conv2d_smallc_supported() tells.
 * ------------------------------------------------------------------------------------------------ */
typedef struct dd3d_smallc_args {  /* host memory */
  const float* in;
  float* out;
  const void* w3;
  const float* scale;
  const float* bias;
  const float* lo; /* optional per-channel lower clamp */
  int32_t B, H, W, Ho, Wo;
  int32_t in_pitch, out_pitch;
  int32_t Cin, KH, KW, stride, pad, N;
  int32_t relu;
} dd3d_smallc_args;
int dd3d_conv2d_smallc_supported(int32_t Cin, int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t N);

/* ------------------------------------------------------------------------------------------------
 * The DLA stem in one launch (ABI 3): uint8 image -> (x - mean) / std (core.py:61-66, zero outside the real image as
 * ImageList.from_tensors pads, image_list.py:120-142) -> base_layer 7x7 3->16 -> level0 3x3 16->16 -> level1 3x3 stride 2 16->32,
 * each + folded norm + ReLU (dla.py:271-280,327-344), DD3D_MATH_F16X2 arithmetic.  Intermediate maps never leave the LDS.
 *   src    uint8 [B][3][Hp][Wp] planar; sizes int32 [B][2] = real (h, w) of each image; mean / stdv per input channel
 *   wK     halves [chunk][plane hi, lo][Npad16][32] of filter K in the k order of dd3d_conv2d_smallc_bf16x3 (w1: Cin 4, 7 chunks,
 *          16 rows; w2: Cin 16, 5 chunks, 16 rows; w3: Cin 16, 5 chunks, 32 rows), every row n scaled by a power of two s_K[n]
 *   scaleK / biasK  per output channel: out = relu(acc * scaleK + biasK); the caller folds 1 / (s_K[n] * plane_scale) into scaleK
 *   out    f32 NHWC [B][Hp/2][Wp/2] rows of out_pitch floats, channels [0, 32) (may be NULL)
 *   out_planes  [pixel][plane][32] halves of value * plane_scale: the one 32-channel chunk image of the split-plane form (may be NULL)
 *   status: the word dd3d_conv_launch.status names -- DD3D_STATUS_F16_OVERFLOW is set when an intermediate or output value leaves the half range)
 * ------------------------------------------------------------------------------------------------ */
typedef struct dd3d_stem_args {  /* host memory; all pointers device */
  const uint8_t* src;
  const int32_t* sizes;
  float mean[3], stdv[3];
  const void* w1;
  const float* scale1;
  const float* bias1;
  const void* w2;
  const float* scale2;
  const float* bias2;
  const void* w3;
  const float* scale3;
  const float* bias3;
  float* out;
  void* out_planes;
  int32_t B, Hp, Wp, out_pitch;
  float plane_scale;
  int32_t* status;
  /* ABI 5 -- start-of-forward chores the launch can take along (block (0, 0, 0) does them; each may be NULL / 0), so that a DLA plan needs no
   * separate launches for them:  inv_K[b] = inverse(K[b]) for b < B (what dd3d_invert_intrinsics computes: core.py:93), and zero_f32[0 .. zero_count)
   * = 0.f (the per-launch range-guard maxima dd3d_conv_launch.amax points into, which are per forward). */
  const float* K;
  float* inv_K;
  float* zero_f32;
  int32_t zero_count;
} dd3d_stem_args;
int dd3d_stem_fused_f16x2(const dd3d_stem_args* args, void* stream);
int dd3d_conv2d_smallc_bf16x3(const dd3d_smallc_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pre-processing.  Replaces DD3D.preprocess_image + ImageList.from_tensors
 * (tridet/modeling/dd3d/core.py:61-72, tridet/structures/image_list.py:120-142):
 * (u8 - mean)/std per channel inside the (h_i, w_i) image, 0.0 in the right/bottom padding.
 *   src  : uint8 [B][3][Hp][Wp] (CHW planes, canvas already at the padded size)
 *   sizes: int32 [B][2] = (h_i, w_i), device
 *   dst  : fp32 NHWC [B][Hp][Wp][4], channel 3 = 0
 * ------------------------------------------------------------------------------------------------ */
int dd3d_preprocess_u8_nhwc4(const uint8_t* src, const int32_t* sizes, float* dst, int32_t B, int32_t Hp, int32_t Wp,
                             const float mean[3], const float std[3], void* stream);

/* 2x2 stride-2 max pooling, NHWC with pitches.  Replaces nn.MaxPool2d(2, 2) = Tree.downsample
 * (tridet/modeling/feature_extractor/dla.py:224-225,235).  H, W even; C % 4 == 0. */
int dd3d_maxpool2x2_nhwc(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t in_pitch,
                         int32_t out_pitch, void* stream);

/* 3x3 stride-2 max pooling, no padding, ceil_mode=True.  Replaces nn.MaxPool2d(3, 2, ceil_mode=True) in front of
 * VoVNet stages 3-5 (tridet/modeling/feature_extractor/vovnet.py:248-249).  Output size = ceil((H-3)/2)+1 (PyTorch rule). */
int dd3d_maxpool3x3s2_ceil_nhwc(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t in_pitch,
                                int32_t out_pitch, void* stream);

/* effective Squeeze-Excitation + OSA identity: out = x * hsigmoid(fc(mean_hw(x))) (+ identity).
 * Replaces eSEModule.forward and the identity add of _OSA_module.forward (vovnet.py:180-185,233-236).
 *   x [B*HW][x_pitch] (C channels), fc_w [C][C] row-major (= the 1x1 conv weight), fc_b [C],
 *   workspaces: partial [B][rsplit][C], gate [B][C].  Deterministic (two-pass sum, no atomics). */
int dd3d_ese_nhwc(const float* x, const float* identity, float* out, const float* fc_w, const float* fc_b, float* partial,
                  float* gate, int32_t B, int32_t HW, int32_t C, int32_t x_pitch, int32_t id_pitch, int32_t out_pitch,
                  int32_t rsplit, void* stream);

/* The same module in two launches that also write the split planes of the result (pool + per-image mean; gate + scale (+ identity)
 * -> f32 NHWC `out` (may be NULL) and / or the split planes of `math_mode` in `out_planes` (may be NULL; C % 32 == 0; layout and
 * plane_scale as dd3d_split_planes)): 5 passes over the map instead of the 7 of dd3d_ese_nhwc + dd3d_split_planes.  Same summation
 * order as dd3d_ese_nhwc.  C % 8 == 0, C <= 4096.
 *   workspaces: partial [B][rsplit][C], mean [B][C], counters int32 [B] -- zero before the first launch, left zero by the kernel
 *   status OR-ed with DD3D_STATUS_F16_OVERFLOW as dd3d_split_planes; may be NULL */
int dd3d_ese_fused(const float* x, const float* identity, float* out, void* out_planes, const float* fc_w, const float* fc_b, float* partial,
                   float* mean, int32_t* counters, int32_t B, int32_t HW, int32_t C, int32_t x_pitch, int32_t id_pitch, int32_t out_pitch,
                   int32_t rsplit, int32_t math_mode, float plane_scale, int32_t* status, void* stream);

/* fine[b,y,x,:] += coarse[b,y/2,x/2,:].  Replaces the FPN top-down step of detectron2 FPN.forward [ext]:
 * prev = lateral + F.interpolate(prev, scale_factor=2, mode="nearest").  H, W (of `fine`) even; C % 4 == 0. */
int dd3d_upsample2x_add_nhwc(float* fine, const float* coarse, int32_t B, int32_t H, int32_t W, int32_t C,
                             int32_t fine_pitch, int32_t coarse_pitch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused per-(image, level) candidate selection + 2D/3D decode.
 * Replaces FCOS2DInference.forward_for_single_feature_map (fcos2d.py:270-344),
 * FCOS3DInference.forward_for_single_feature_map (fcos3d.py:328-399), predictions_to_boxes3d
 * (fcos3d.py:16-52), allocentric_to_egocentric / unproject_points2d (tridet/utils/geometry.py:15-112),
 * pytorch3d quaternion_to_matrix / matrix_to_quaternion [ext], compute_features_locations
 * (tridet/utils/tensor2d.py:6-25).
 *
 * Head maps are NHWC with these channel layouts (C = num classes, C3 = 1 if class-agnostic else C):
 *   cls  [B*HW][cls_pitch]  : logits 0..C-1
 *   box2d[B*HW][b2d_pitch]  : relu(scale*reg) 0..3, centerness logit 4
 *   box3d[B*HW][b3d_pitch]  : quat 4*C3 (comp*C3+cls), ctr 2*C3, depth C3, size 3*C3, conf C3
 * Output (per image b): cand[b][f][slot_off[level] + j], f < DD3D_CAND_FIELDS:
 *   0-3 box x1,y1,x2,y2 | 4 score=sqrt(cls*ctr) | 5 score_3d | 6 class (int bits) | 7 loc*C+class (int bits)
 *   8-9 location x,y | 10-13 quat wxyz (egocentric) | 14-15 proj_ctr | 16 depth | 17-19 size WLH
 *   20 argmax attribute (int bits) | 21 speed      (nuScenes; NuscenesInference, nuscenes_dd3d.py:268-296)
 * counts[b][level] = number of valid slots (<= topk); npass[b][level] = #scores over threshold.
 * Candidate order inside a level = ascending (loc, class), i.e. torch.nonzero order.
 * ------------------------------------------------------------------------------------------------ */
typedef struct dd3d_select_args {  /* host memory */
  const float* cls[DD3D_MAX_LEVELS];
  const float* box2d[DD3D_MAX_LEVELS];
  const float* box3d[DD3D_MAX_LEVELS]; /* NULL entries => 2D only (MODEL.BOX3D_ON false) */
  int32_t H[DD3D_MAX_LEVELS], W[DD3D_MAX_LEVELS], stride[DD3D_MAX_LEVELS];
  int32_t cls_pitch, b2d_pitch, b3d_pitch;
  int32_t num_levels, B, num_classes;
  int32_t class_agnostic_3d;
  int32_t loc_offset_half;       /* DD3D.FEATURE_LOCATIONS_OFFSET == "half" */
  int32_t thresh_with_ctr;       /* DD3D.FCOS2D.INFERENCE.THRESH_WITH_CTR */
  int32_t topk;                  /* PRE_NMS_TOPK */
  int32_t attr_off, num_attr;    /* nuScenes: cls-map channels [attr_off, attr_off+num_attr) = attribute logits (0 attrs: none) */
  int32_t speed_off;             /* nuScenes: cls-map channel of relu(speed), or -1 */
  float pre_nms_thresh;
  float min_depth, max_depth, focal_factor;
  int32_t scale_depth_by_focal, allocentric, depth_is_distance;
  const float* inv_K;            /* [B][9] row-major inverse intrinsics, device */
  const float* canon_sizes;      /* [>=num_classes][3] (W,L,H), device */
  int32_t* scratch_idx;          /* device, per (b,level) region; offsets below  */
  float* scratch_score;
  int64_t scratch_off[DD3D_MAX_LEVELS]; /* element offset of level l's region for image 0 */
  int64_t scratch_img_stride;           /* elements per image */
  float* cand;                   /* [B][DD3D_CAND_FIELDS][slots per image] */
  int32_t* counts;               /* [B][num_levels] */
  int32_t* npass;                /* [B][num_levels] */
  int32_t slot_off[DD3D_MAX_LEVELS + 1]; /* first slot of level l in an image's candidate row; slot_off[num_levels] = slots per image.
                                    Level l needs min(topk, H*W*C) slots, so the exchanged buffer carries no slot a level can never
                                    fill.  All zero: the dense layout l * topk (num_levels * topk slots per image). */
} dd3d_select_args;
int dd3d_fcos_select_decode(const dd3d_select_args* args, void* stream);

/* Closed-form inverse of the (B,3,3) intrinsics.  Replaces images.intrinsics.inverse() (core.py:93). */
int dd3d_invert_intrinsics(const float* K, float* inv_K, int32_t B, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Class-aware 2D NMS + top-k + resize.  Replaces FCOS2DInference.nms_and_top_k (fcos2d.py:346-367) =
 * detectron2.layers.batched_nms -> torchvision.ops.batched_nms / nms [ext] (rank by score_3d,
 * coordinate trick when 4*n <= 4000 else per-class), torch.kthvalue top-k on the 2D score with >=,
 * and detectron2 detector_postprocess [ext] (core.py:153-160).
 *   cand/counts as written by dd3d_fcos_select_decode for G images (after the RCCL gather: G = all images)
 *   out_size [G][4] = (in_h, in_w, out_h, out_w) float, device
 *   det [G][det_cap][DD3D_DET_FIELDS]:
 *     0-3 box | 4 score | 5 score_3d | 6 class | 7 fpn level | 8-9 location | 10-13 quat | 14-15 proj_ctr
 *     | 16 depth | 17-19 size | 20 attribute | 21 speed | 22-25 global quat | 26-28 global tvec (filled by
 *     dd3d_bev_nms_aggregate) | 29-31 unused       (class / level / attribute stored as float-valued integers)
 *   det_count [G]; order = descending score_3d (torchvision keep order).
 * Workspaces (device): sort_idx int32 [G][ncap], sbox float [G][ncap][4], scls int32 [G][ncap],
 *   mask uint64 [G][ncap][ncap/64], nvalid int32 [G][2], where ncap = round_up(num_levels*topk, 64).
 * ------------------------------------------------------------------------------------------------ */
typedef struct dd3d_nms_args {  /* host memory */
  const float* cand;
  const int32_t* counts;
  int32_t G, num_levels, topk;
  int32_t do_nms;          /* DD3D.INFERENCE.DO_NMS */
  int32_t use_score3d;     /* rank by score_3d (BOX3D_ON) else score */
  float nms_thresh;        /* <= 0 disables suppression (fcos2d.py:349) */
  int32_t post_topk;       /* POST_NMS_TOPK */
  int32_t do_postprocess;  /* DD3D.INFERENCE.DO_POSTPROCESS */
  const float* out_size;
  int32_t* sort_idx;
  float* sbox;
  int32_t* scls;
  uint64_t* mask;
  int32_t* nvalid;
  float* det;
  int32_t* det_count;
  int32_t det_cap;
  int32_t slot_off[DD3D_MAX_LEVELS + 1]; /* same table as the producer of `cand` used (the select args) */
  /* Reading the images out of the all-gathered records of several ranks (ABI 3).  img_per_rec = 0: `cand`, `counts`, `out_size` are
   * dense [G] arrays (above).  img_per_rec = P > 0: they point INTO RECORD 0 of a buffer of records rec_stride 4-byte words apart,
   * each holding P images; image g of this call is global image img_first + g = image (img_first + g) % P of record
   * (img_first + g) / P.  This is how the owner of a nuScenes sample finalises cameras that other ranks decoded. */
  int32_t img_first, img_per_rec;
  int64_t rec_stride;
} dd3d_nms_args;
int dd3d_nms_finalize(const dd3d_nms_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * nuScenes sample aggregation = camera->global box transform + BEV rotated-box NMS.
 * Replaces nuscenes_sample_aggregate / sample_bev_nms (tridet/modeling/dd3d/postprocessing.py:22-108),
 * boxes3d_to_rotated_boxes / bev_nms (tridet/layers/bev_nms.py:51-133), GenericBoxes3D.corners
 * (tridet/structures/boxes3d.py:47-64), pytorch3d Transform3d / rotation conversions [ext] and detectron2
 * batched_nms_rotated -> nms_rotated / box_iou_rotated [ext].
 *   det_in [G][det_cap][DD3D_DET_FIELDS], count_in [G]      as written by dd3d_nms_finalize
 *   inv_K [G][9] inverse intrinsics (as written by dd3d_invert_intrinsics)
 *   pose [G][7] (quat wxyz, tvec) camera->global; group [G] sample index of each image (category id =
 *   class + group*num_classes); out_size as in dd3d_nms_finalize (used when do_postprocess)
 *   max_dets: cap on the batch-global, score-ordered keep list (0 = none; the reference truncates the whole batch,
 *   postprocessing.py:93-94).  det_out / count_out: survivors per image in their original order, fields 22-28 = the
 *   global-frame box when write_global.  count_out = -1 everywhere if more than 8192 boxes arrive.
 * Workspaces (device): work float [G*det_cap][16], sbox float [G*det_cap][8], mask uint64 [mcap][mcap/64]
 *   (mcap = min(round_up(G*det_cap, 64), 8192): rows and columns are positions in the SORTED list, which never holds more than the
 *   8192 boxes of the LDS sorter -- ABI 6; ABI 5 strode the rows by round_up(G*det_cap, 64)/64 words), meta int32 [4].
 * ------------------------------------------------------------------------------------------------ */
typedef struct dd3d_bev_args {  /* host memory */
  const float* det_in;
  const int32_t* count_in;
  const float* inv_K;
  const float* pose;
  const int32_t* group;
  const float* out_size;
  int32_t G, det_cap, num_classes;
  float iou_thresh;
  int32_t max_dets;
  int32_t write_global;
  int32_t do_postprocess;
  float* work;
  float* sbox;
  uint64_t* mask;
  int32_t* meta;
  float* det_out;
  int32_t* count_out;
  /* same record addressing as the NMS arguments above, ABI 3: with img_per_rec > 0, `inv_K`, `pose` and `out_size` point into record 0 of the gathered buffer
   * ([P][9], [P][7], [P][4] blocks of a record) and image g is global image img_first + g; `group`, `det_in`, `count_in` stay dense. */
  int32_t img_first, img_per_rec;
  int64_t rec_stride;
} dd3d_bev_args;
int dd3d_bev_nms_aggregate(const dd3d_bev_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input side (the step right before the path): the test-time resize of the uint8 image.
 * Replaces detectron2 ResizeTransform.apply_image [ext] = PIL Image.resize(BILINEAR) behind ResizeShortestEdge
 * (tridet/data/augmentations/resize_transform.py:85-88, dataset_mapper.py:118-127): Pillow's separable 8-bit resampling --
 * horizontal pass, 8-bit intermediate, vertical pass -- with the per-coordinate bounds and 22-bit fixed-point coefficients
 * computed by the caller (dd3d_amd/inputs.py resample_coeffs).  Bit-identical to PIL.
 *   src  uint8 planar [C][H][W] with strides (src_plane, src_row); dst planar with (dst_plane, dst_row) -- e.g. one image
 *        slot of the forward plan's input canvas; tmp uint8 [C][H][new_w], needed when both sizes change
 *   lo_* / cnt_* int32 [new size], kk_* int32 [new size][ksize_*]
 * ------------------------------------------------------------------------------------------------ */
typedef struct dd3d_resize_args {  /* host memory; all pointers device */
  const uint8_t* src;
  uint8_t* dst;
  uint8_t* tmp;
  int32_t C, H, W, new_h, new_w;
  int64_t src_plane, dst_plane;
  int32_t src_row, dst_row;
  const int32_t *lo_w, *cnt_w, *kk_w;
  const int32_t *lo_h, *cnt_h, *kk_h;
  int32_t ksize_w, ksize_h;
} dd3d_resize_args;
int dd3d_resize_bilinear_u8(const dd3d_resize_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * DD3DDenseDepth tail (tridet/modeling/dd3d/dense_depth.py:140-151): aligned_bilinear(x, factor, offset)
 * (tridet/utils/tensor2d.py:28-47) of channel 0 of an NHWC map [B][h][w] (rows of `pitch` floats) to out [B][factor*h][factor*w],
 * then, when focal_factor > 0, out /= norm(inv_K[b][0][0], inv_K[b][1][1]) * focal_factor  (SCALE_DEPTH_BY_FOCAL_LENGTHS).
 * ------------------------------------------------------------------------------------------------ */
int dd3d_aligned_bilinear_scale(const float* src, float* out, const float* inv_K, int32_t B, int32_t h, int32_t w, int32_t pitch,
                                int32_t factor, int32_t offset_half, float focal_factor, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Evaluator-side overlaps (the step right after the path).  Replace the reference's own numba.cuda / numba
 * kernels in tridet/evaluators/rotate_iou.py (called at kitti_3d_evaluator.py:622-632):
 *   dd3d_rotate_iou_eval    rotate_iou_gpu_eval :292-327: boxes [N][5], qboxes [K][5] = (x, y, x_d, y_d, angle clockwise) ->
 *                           out [N][K]; criterion -1 IoU, 0 / 1 intersection over the query's / box's area, 2 intersection area
 *   dd3d_d3_box_overlap     d3_box_overlap_kernel :330-357: boxes [N][7], qboxes [K][7]; rinc [N][K] holds BEV intersection areas
 *                           on entry (criterion 2 above) and 3D overlaps on return; camera_coordinate picks the vertical axis
 *   dd3d_image_box_overlap  image_box_overlap :360-381: XYXY boxes [N][4], [K][4] -> out [N][K]
 * All pointers device memory, float32, row-major.
 * ------------------------------------------------------------------------------------------------ */
int dd3d_rotate_iou_eval(const float* boxes, const float* qboxes, float* out, int32_t N, int32_t K, int32_t criterion, void* stream);
int dd3d_d3_box_overlap(const float* boxes, const float* qboxes, float* rinc, int32_t N, int32_t K, int32_t criterion,
                        int32_t camera_coordinate, void* stream);
int dd3d_image_box_overlap(const float* boxes, const float* qboxes, float* out, int32_t N, int32_t K, int32_t criterion, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Result formatting for the evaluators.  Replaces the per-box host loops of
 * tridet/evaluators/kitti_3d_evaluator.py:205-264 (convert_3d_box_to_kitti, called once per detection at :120 and per annotation
 * at :145) and tridet/evaluators/nuscenes_evaluator.py:196-198 (global velocity = speed * first column of R(quat_global)).
 *   box3d        [n][10] float32 = Boxes3D.vectorize(): quat (w,x,y,z), tvec (3), size (W,L,H)   (boxes3d.py:142-144)
 *   quat_global  [n][4] float32 and speed [n] float32, or both NULL (KITTI)
 *   out          [n][10] float64 = (W, L, H, x, y, z, rot_y, alpha, vx, vy); alpha rounded to 2 decimals like the reference
 * Arithmetic is float64 (the reference's numpy / pyquaternion path), x / y / z stay float32 values like the reference's in-place add.
 * ------------------------------------------------------------------------------------------------ */
int dd3d_format_boxes3d(const float* box3d, const float* quat_global, const float* speed, double* out, int32_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DD3D_HIP_H */
