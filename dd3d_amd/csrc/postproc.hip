// Post-head kernels of the DD3D forward path for gfx950: everything between the head maps and the final detections
// runs on the device with fixed-capacity buffers and no host synchronisation.
//
//   fcos_select_decode_kernel  one 1024-thread block per (level, image): sigmoid score, threshold, ordered compaction
//                              (= torch.nonzero order), exact top-k by radix select, then the fused per-candidate 2D/3D
//                              decode (quaternion normalise, depth un-normalise, allocentric->egocentric, size scaling).
//   nms_sort_kernel            per image: gather levels, stable bitonic sort by score, coordinate-trick offsets.
//   nms_mask_kernel            64x64 IoU bit-mask tiles (upper triangle).
//   nms_finalize_kernel        per image: greedy reduce of the mask, top-k on the 2D score, resize/clip/filter, output.
//
// These are latency/HBM-bound integer+scalar-float kernels (<= ~150 flop per candidate): no MFMA.
// Floating-point contraction is off so IoU / score comparisons round like the reference's separate mul/add ops.
#pragma clang fp contract(off)
#include <algorithm>

#include "common.h"

DD3D_NOTE_BUILD_FLAGS

namespace dd3d {

constexpr int PT = 1024;        // threads per block for the per-image / per-level kernels
constexpr int TOPK_MAX = 1024;  // PRE_NMS_TOPK capacity of the LDS candidate list
// slot_off[l] = first candidate slot of level l inside an image's row, slot_off[L] = slots per image (engine: level l holds
// min(topk, H*W*C) slots -- the RCCL payload carries no slot a level can never fill); all zero = the dense layout l * topk
__device__ __host__ inline int slot_base(const int32_t* slot_off, int L, int topk, int l) { return slot_off[L] > 0 ? slot_off[l] : l * topk; }
__device__ __host__ inline int level_of_slot(const int32_t* slot_off, int L, int topk, int slot) {
  if (slot_off[L] <= 0) return slot / topk;
  int l = 0;
  while (l + 1 < L && slot >= slot_off[l + 1]) ++l;
  return l;
}

constexpr int NCAP_MAX = 8192;  // max candidates per image (levels * topk) the LDS sorter handles
constexpr float QEPS = 1e-7f;   // tridet/modeling/dd3d/fcos3d.py:13

// ------------------------------------------------------------------------------------------------ block primitives
// exclusive prefix sum over the block's 1024 threads; `total` = block sum.  wsum: 16 ints of LDS.  Ends with a barrier.
__device__ __forceinline__ int block_excl_scan(int v, int* wsum, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < PT / 64; ++w) {
    const int t = wsum[w];
    if (w < wave) base += t;
    tot += t;
  }
  __syncthreads();
  total = tot;
  return base + x - v;
}

// in-LDS bitonic sort into "descending key, ascending val on ties" order.  P = power of two.
__device__ __forceinline__ bool sorts_before(float ka, int va, float kb, int vb) { return ka > kb || (ka == kb && va < vb); }

__device__ void block_bitonic_sort(float* keys, int* vals, int P) {
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += PT) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const float ka = keys[i], kb = keys[ixj];
          const int va = vals[i], vb = vals[ixj];
          const bool up = (i & k) == 0;  // this pair must end in sorted order (up) or reversed
          const bool swap = up ? sorts_before(kb, vb, ka, va) : sorts_before(ka, va, kb, vb);
          if (swap) {
            keys[i] = kb, keys[ixj] = ka;
            vals[i] = vb, vals[ixj] = va;
          }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------ select + decode
struct SelectK {
  dd3d_select_args a;
};

__global__ __launch_bounds__(PT) void fcos_select_decode_kernel(const SelectK P) {
  const dd3d_select_args& a = P.a;
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int C = a.num_classes;
  const int H = a.H[l], W = a.W[l], HW = H * W;
  const int n_el = HW * C;
  const float* cls = a.cls[l];
  const float* b2d = a.box2d[l];
  const float* b3d = a.box3d[l];
  const long pix0 = (long)b * HW;
  int32_t* sidx = a.scratch_idx + a.scratch_off[l] + (long)b * a.scratch_img_stride;
  float* sscore = a.scratch_score + a.scratch_off[l] + (long)b * a.scratch_img_stride;

  __shared__ int wsum[PT / 64];
  __shared__ int hist[256];
  __shared__ int sel_e[TOPK_MAX];
  __shared__ float sel_s[TOPK_MAX];
  __shared__ int sh_bucket, sh_kk, sh_flag;

  // ---- phase 1: score, threshold, ordered compaction  (fcos2d.py:274-300)
  // Round r covers elements r*4096 + tid*4 .. +3 (coalesced).  The pass bits of up to 16 rounds are computed FIRST, with
  // nothing between the rounds' loads (a scan per round used to serialise ten exposed load latencies on the p3 level);
  // then the rounds are compacted four at a time with one 64-bit packed scan (a round passes <= 4096 < 2^16 elements),
  // and the few survivors re-derive their score from the L1/L2-hot maps with the same arithmetic.
  constexpr int VPT = 4, RPS = 16;  // elements per thread per round, rounds per super-round
  auto score_of = [&](int e, float& gate) {
    const int loc = e / C;
    const int c = e - loc * C;
    const float sc = sigmoidf(cls[(pix0 + loc) * a.cls_pitch + c]);
    const float ct = sigmoidf(b2d[(pix0 + loc) * a.b2d_pitch + 4]);
    const float prod = sc * ct;
    gate = a.thresh_with_ctr ? prod : sc;
    return prod;
  };
  __shared__ unsigned long long wsum64[PT / 64];
  // logit(thr) minus a margin that covers any rounding of sigmoidf; thr outside (0, 1) disables the shortcut
  const float thr01 = a.pre_nms_thresh;
  const float skip_below = (thr01 > 0.f && thr01 < 1.f) ? logf(thr01 / (1.f - thr01)) - 0.0625f : -INFINITY;
  int running = 0;
  for (int sbase = 0; sbase < n_el; sbase += PT * VPT * RPS) {
    unsigned long long passbits = 0;  // bit r*4 + v
#pragma unroll
    for (int r = 0; r < RPS; ++r) {
      const int e0 = sbase + r * PT * VPT + tid * VPT;
      if (e0 < n_el) {
        int loc = e0 / C, c = e0 - (e0 / C) * C;  // one division per round, then stepped
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
          if (e0 + v < n_el) {
            // sigmoid(x) * (anything <= 1) > thr needs x > logit(thr): ~99 % of the logits stop here (measured: scoring every
            // element cost 60 k cycles on the p3 block, all VALU)
            if (cls[(pix0 + loc) * a.cls_pitch + c] >= skip_below) {
              float gate;
              score_of(e0 + v, gate);
              passbits |= (unsigned long long)(gate > a.pre_nms_thresh) << (r * VPT + v);
            }
            if (++c == C) c = 0, ++loc;
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RPS / 4; ++q) {
      if (sbase + q * 4 * PT * VPT >= n_el) break;  // block-uniform
      const unsigned bits16 = (unsigned)(passbits >> (q * 16)) & 0xffffu;
      unsigned long long cnt4 = 0;  // four 16-bit counters, one per round
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) cnt4 |= (unsigned long long)__popc((bits16 >> (rr * VPT)) & 0xfu) << (16 * rr);
      // block-wide exclusive scan of the packed counters (no carry between the fields: every field's total is <= 4096)
      unsigned long long incl = cnt4;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long up = __shfl_up(incl, d, 64);
        if ((tid & 63) >= d) incl += up;
      }
      if ((tid & 63) == 63) wsum64[tid >> 6] = incl;
      __syncthreads();
      unsigned long long wave_off = 0, total4 = 0;
#pragma unroll
      for (int w = 0; w < PT / 64; ++w) {
        const unsigned long long t = wsum64[w];
        if (w < (tid >> 6)) wave_off += t;
        total4 += t;
      }
      const unsigned long long excl = wave_off + incl - cnt4;
      int round_base = running;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        int pos = round_base + (int)((excl >> (16 * rr)) & 0xffffu);
        const int e0 = sbase + (q * 4 + rr) * PT * VPT + tid * VPT;
        unsigned nib = (bits16 >> (rr * VPT)) & 0xfu;
        while (nib) {
          const int v = __ffs(nib) - 1;
          nib &= nib - 1;
          float gate;
          sidx[pos] = e0 + v;
          sscore[pos] = score_of(e0 + v, gate);
          ++pos;
        }
        round_base += (int)((total4 >> (16 * rr)) & 0xffffu);
      }
      running = round_base;
      __syncthreads();  // wsum64 is reused by the next packed scan
    }
  }
  const int n = running;
  const int k = min(n, a.topk);
  if (tid == 0) {
    a.npass[b * a.num_levels + l] = n;
    a.counts[b * a.num_levels + l] = k;
    sh_flag = 0;
  }
  __syncthreads();  // scratch writes visible to the whole block

  // ---- phase 2: exact top-k  (fcos2d.py:309-317; order inside the level stays ascending (loc, class))
  if (n <= k) {
    if (tid < n) {
      sel_e[tid] = sidx[tid];
      sel_s[tid] = sscore[tid];
    }
  } else {
    // radix select of the k-th largest score (scores are positive floats => uint order == float order)
    unsigned prefix = 0, mask = 0;
    int kk = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < n; i += PT) {
        const unsigned key = __float_as_uint(sscore[i]);
        if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
      }
      __syncthreads();
      if (tid == 0) {
        int cum = 0, bsel = 0;
        for (int bkt = 255; bkt >= 0; --bkt) {
          const int c = hist[bkt];
          if (cum + c >= kk) {
            bsel = bkt;
            break;
          }
          cum += c;
        }
        sh_bucket = bsel;
        sh_kk = kk - cum;
      }
      __syncthreads();
      prefix |= (unsigned)sh_bucket << shift;
      mask |= 0xFFu << shift;
      kk = sh_kk;
      __syncthreads();
    }
    const unsigned T = prefix;  // key of the k-th largest; take every key > T and the first `kk` keys == T
    int run_gt = 0, run_eq = 0;
    for (int base = 0; base < n; base += PT) {
      const int i = base + tid;
      int gt = 0, eq = 0, e = 0;
      float sc = 0.f;
      if (i < n) {
        sc = sscore[i];
        e = sidx[i];
        const unsigned key = __float_as_uint(sc);
        gt = key > T;
        eq = key == T;
      }
      int total;
      const int ex = block_excl_scan(gt | (eq << 16), wsum, total);
      const int gt_before = run_gt + (ex & 0xFFFF);
      const int eq_before = run_eq + (ex >> 16);
      if (gt || (eq && eq_before < kk)) {
        const int pos = gt_before + min(eq_before, kk);
        sel_e[pos] = e;
        sel_s[pos] = sc;
      }
      run_gt += total & 0xFFFF;
      run_eq += total >> 16;
    }
  }
  __syncthreads();

  // ---- phase 3: decode  (fcos2d.py:319-336, fcos3d.py:343-399, fcos3d.py:16-52, geometry.py:15-55)
  const int NS = slot_base(a.slot_off, a.num_levels, a.topk, a.num_levels);
  float* cand = a.cand + (long)b * DD3D_CAND_FIELDS * NS + slot_base(a.slot_off, a.num_levels, a.topk, l);
  float q0 = 0, q1 = 0, q2 = 0, q3 = 0, qn = 1.f;
  int bad = 0;
  const bool active = tid < k;
  if (active) {
    const int e = sel_e[tid];
    const float s_in = sel_s[tid];
    const int loc = e / C;
    const int c = e - loc * C;
    const int stride = a.stride[l];
    const float off = a.loc_offset_half ? (float)(stride / 2) : 0.f;
    const float lx = (float)((loc % W) * stride) + off;  // tensor2d.py:6-25
    const float ly = (float)((loc / W) * stride) + off;
    const float* r = b2d + (pix0 + loc) * a.b2d_pitch;
    cand[0 * NS + tid] = lx - r[0];
    cand[1 * NS + tid] = ly - r[1];
    cand[2 * NS + tid] = lx + r[2];
    cand[3 * NS + tid] = ly + r[3];
    const float score = sqrtf(s_in);  // fcos2d.py:333
    cand[4 * NS + tid] = score;
    cand[6 * NS + tid] = __int_as_float(c);
    cand[7 * NS + tid] = __int_as_float(e);
    cand[8 * NS + tid] = lx;
    cand[9 * NS + tid] = ly;
    {  // nuScenes extras on the cls tower (nuscenes_dd3d.py:268-296): argmax attribute, speed
      const float* pc = cls + (pix0 + loc) * a.cls_pitch;
      int best = 0;
      if (a.num_attr > 0) {
        float bv = pc[a.attr_off];
        for (int t = 1; t < a.num_attr; ++t) {
          const float v = pc[a.attr_off + t];
          if (v > bv) bv = v, best = t;  // first maximum, as torch.argmax
        }
      }
      cand[20 * NS + tid] = __int_as_float(best);
      cand[21 * NS + tid] = a.speed_off >= 0 ? pc[a.speed_off] : 0.f;
    }
    if (b3d == nullptr) {
      cand[5 * NS + tid] = score;
      for (int f = 10; f < 20; ++f) cand[f * NS + tid] = 0.f;
    } else {
      const int C3 = a.class_agnostic_3d ? 1 : C;
      const int c3 = a.class_agnostic_3d ? 0 : c;
      const float* p = b3d + (pix0 + loc) * a.b3d_pitch;  // channel = component * C3 + class  (fcos3d.py:335-339)
      float qa = p[0 * C3 + c3], qb = p[1 * C3 + c3], qc = p[2 * C3 + c3], qd = p[3 * C3 + c3];
      float cx = p[4 * C3 + c3], cy = p[5 * C3 + c3];
      float depth = p[6 * C3 + c3];
      const float s0 = p[7 * C3 + c3], s1 = p[8 * C3 + c3], s2 = p[9 * C3 + c3];
      const float conf = sigmoidf(p[10 * C3 + c3]);
      cand[5 * NS + tid] = score * conf;  // scores_3d, fcos3d.py:375-376
      const float* K = a.inv_K + 9 * b;
      // quat / max(|quat|, eps), then / |quat| again  (fcos3d.py:31-34)
      float nrm = fmaxf(sqrtf(qa * qa + qb * qb + qc * qc + qd * qd), QEPS);
      qa /= nrm, qb /= nrm, qc /= nrm, qd /= nrm;
      nrm = sqrtf(qa * qa + qb * qb + qc * qc + qd * qd);
      qa /= nrm, qb /= nrm, qc /= nrm, qd /= nrm;
      if (a.scale_depth_by_focal) {  // fcos3d.py:36-38
        const float pixel_size = sqrtf(K[0] * K[0] + K[4] * K[4]);
        depth = depth / (pixel_size * a.focal_factor);
      }
      if (a.depth_is_distance) {  // fcos3d.py:40-41, ray through the *location*
        const float rx = K[0] * lx + K[1] * ly + K[2], ry = K[3] * lx + K[4] * ly + K[5], rz = K[6] * lx + K[7] * ly + K[8];
        depth = depth / fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), QEPS);
      }
      depth = fminf(fmaxf(depth, a.min_depth), a.max_depth);
      cx += lx, cy += ly;  // proj_ctr + locations
      if (a.allocentric) {
        // R_obj_to_local = M(q)  ([ext] pytorch3d quaternion_to_matrix)
        const float two_s = 2.0f / (qa * qa + qb * qb + qc * qc + qd * qd);
        const float o00 = 1 - two_s * (qc * qc + qd * qd), o01 = two_s * (qb * qc - qd * qa), o02 = two_s * (qb * qd + qc * qa);
        const float o10 = two_s * (qb * qc + qd * qa), o11 = 1 - two_s * (qb * qb + qd * qd), o12 = two_s * (qc * qd - qb * qa);
        const float o20 = two_s * (qb * qd - qc * qa), o21 = two_s * (qc * qd + qb * qa), o22 = 1 - two_s * (qb * qb + qc * qc);
        // local frame from the viewing ray through proj_ctr  (geometry.py:30-41)
        float zx = K[0] * cx + K[1] * cy + K[2], zy = K[3] * cx + K[4] * cy + K[5], zz = K[6] * cx + K[7] * cy + K[8];
        const float zn = sqrtf(zx * zx + zy * zy + zz * zz);
        zx /= zn, zy /= zn, zz /= zn;
        float yx = 0.f - zy * zx, yy = 1.f - zy * zy, yz = 0.f - zy * zz;
        const float yn = sqrtf(yx * yx + yy * yy + yz * yz);
        yx /= yn, yy /= yn, yz /= yn;
        const float xx = yy * zz - yz * zy, xy = yz * zx - yx * zz, xz = yx * zy - yy * zx;  // cross(y, z)
        // R = [x y z] (columns) * R_obj
        const float m00 = xx * o00 + yx * o10 + zx * o20, m01 = xx * o01 + yx * o11 + zx * o21, m02 = xx * o02 + yx * o12 + zx * o22;
        const float m10 = xy * o00 + yy * o10 + zy * o20, m11 = xy * o01 + yy * o11 + zy * o21, m12 = xy * o02 + yy * o12 + zy * o22;
        const float m20 = xz * o00 + yz * o10 + zz * o20, m21 = xz * o01 + yz * o11 + zz * o21, m22 = xz * o02 + yz * o12 + zz * o22;
        // [ext] pytorch3d matrix_to_quaternion (0.5.x/0.6.x): candidate of the largest |component|, no sign canonicalisation
        const float t0 = 1.f + m00 + m11 + m22, t1 = 1.f + m00 - m11 - m22, t2 = 1.f - m00 + m11 - m22, t3 = 1.f - m00 - m11 + m22;
        const float a0 = t0 > 0.f ? sqrtf(t0) : 0.f, a1 = t1 > 0.f ? sqrtf(t1) : 0.f;
        const float a2 = t2 > 0.f ? sqrtf(t2) : 0.f, a3 = t3 > 0.f ? sqrtf(t3) : 0.f;
        int best = 0;
        float am = a0;
        if (a1 > am) best = 1, am = a1;
        if (a2 > am) best = 2, am = a2;
        if (a3 > am) best = 3, am = a3;
        const float den = 2.0f * fmaxf(am, 0.1f);
        if (best == 0) q0 = a0 * a0, q1 = m21 - m12, q2 = m02 - m20, q3 = m10 - m01;
        else if (best == 1) q0 = m21 - m12, q1 = a1 * a1, q2 = m10 + m01, q3 = m02 + m20;
        else if (best == 2) q0 = m02 - m20, q1 = m10 + m01, q2 = a2 * a2, q3 = m12 + m21;
        else q0 = m10 - m01, q1 = m20 + m02, q2 = m21 + m12, q3 = a3 * a3;
        q0 /= den, q1 /= den, q2 /= den, q3 /= den;
        qn = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        bad = !(fabsf(qn - 1.0f) <= 1e-3f + 1e-5f);  // torch.allclose(qn, 1, atol=1e-3) with the default rtol
      } else {
        q0 = qa, q1 = qb, q2 = qc, q3 = qd;
      }
      const float* cs = a.canon_sizes + 3 * c;
      cand[14 * NS + tid] = cx;
      cand[15 * NS + tid] = cy;
      cand[16 * NS + tid] = depth;
      cand[17 * NS + tid] = (tanhf(s0) + 1.0f) * cs[0];
      cand[18 * NS + tid] = (tanhf(s1) + 1.0f) * cs[1];
      cand[19 * NS + tid] = (tanhf(s2) + 1.0f) * cs[2];
    }
  }
  if (b3d != nullptr) {
    // the renormalisation is triggered for the whole (level, image) batch if ANY norm is off (geometry.py:48-53)
    if (bad) atomicOr(&sh_flag, 1);
    __syncthreads();
    if (active) {
      if (sh_flag) {
        const float d = fmaxf(qn, QEPS);
        q0 /= d, q1 /= d, q2 /= d, q3 /= d;
      }
      cand[10 * NS + tid] = q0;
      cand[11 * NS + tid] = q1;
      cand[12 * NS + tid] = q2;
      cand[13 * NS + tid] = q3;
    }
  }
}

// ------------------------------------------------------------------------------------------------ NMS
struct NmsK {
  dd3d_nms_args a;
  int ncap;   // round_up(levels*topk, 64)
  int ncap2;  // next power of two >= levels*topk (size of the LDS sort arrays)
  int fin_lds;  // dynamic LDS bytes nms_finalize_kernel was launched with
};

// Word offset of image g's block (`per_img` 4-byte words per image) relative to the pointer of record 0 / image 0: dense arrays when
// img_per_rec == 0, else image (img_first + g) % P of record (img_first + g) / P, records rec_stride words apart (dd3d_hip.h, ABI 3).
__device__ __forceinline__ long rec_off(int g, int img_first, int img_per_rec, long rec_stride, int per_img) {
  if (img_per_rec <= 0) return (long)g * per_img;
  const int gg = img_first + g;
  const int r = gg / img_per_rec;
  return (long)r * rec_stride + (long)(gg - r * img_per_rec) * per_img;
}

// Number of keys[lo, hi) that sort before (ki, position pos) in descending order, ties by position: the initial order of the
// sorters is by increasing value index, so "val_j < val_i" is "j < pos" and the index array need not be read.  One broadcast
// ds_read_b128 feeds four comparisons; lo and hi are multiples of 4 and keys past the last candidate hold -inf.
__device__ __forceinline__ int rank_of(const float* keys, int lo, int hi, float ki, int pos) {
  int rank = 0;
#pragma unroll 4
  for (int j0 = lo; j0 < hi; j0 += 4) {
    const float4 k4 = *reinterpret_cast<const float4*>(keys + j0);
    rank += (k4.x > ki) || (k4.x == ki && j0 + 0 < pos);
    rank += (k4.y > ki) || (k4.y == ki && j0 + 1 < pos);
    rank += (k4.z > ki) || (k4.z == ki && j0 + 2 < pos);
    rank += (k4.w > ki) || (k4.w == ki && j0 + 3 < pos);
  }
  return rank;
}

// mode written to nvalid[g][1]
enum { NMS_TRICK = 0, NMS_PER_CLASS = 1, NMS_NONE = 2 };

constexpr int SORT_SPLIT = 16;  // blocks per image of the rank sort: 64 candidates x 16 threads each

// Per image: gather the levels, order by score (descending, ties by concatenated index), write the sorted work arrays with the
// coordinate-trick offsets.  n <= PT: rank sort spread over SORT_SPLIT blocks -- every block holds all keys in LDS, block s ranks
// candidates s*64 .. s*64+63 with 16 threads per candidate (each compares against 1/16 of the keys) and writes them straight to
// their sorted position (one block with one thread per candidate spent 20 us in the n-step comparison loop, measured).
// n > PT: block 0 runs the in-LDS bitonic sort.
__global__ __launch_bounds__(PT) void nms_sort_kernel(const NmsK P) {
  const dd3d_nms_args& a = P.a;
  const int g = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  const int L = a.num_levels, NS = slot_base(a.slot_off, L, a.topk, L);
  const float* cand = a.cand + rec_off(g, a.img_first, a.img_per_rec, a.rec_stride, DD3D_CAND_FIELDS * NS);
  const int32_t* counts = a.counts + rec_off(g, a.img_first, a.img_per_rec, a.rec_stride, L);
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];  // keys[ncap2] | vals[ncap2]
  float* keys = reinterpret_cast<float*>(dyn_lds);
  int* vals = reinterpret_cast<int*>(dyn_lds) + P.ncap2;
  __shared__ int pref[DD3D_MAX_LEVELS + 1];
  __shared__ float red[PT / 64];
  if (tid == 0) {
    int acc = 0;
    for (int l = 0; l < L; ++l) {
      pref[l] = acc;
      acc += counts[l];
    }
    pref[L] = acc;
  }
  __syncthreads();
  const int n = pref[L];
  const bool suppress = a.do_nms && a.nms_thresh > 0.f;
  const bool split = suppress && n <= PT;
  if (s > 0 && (!split || s * 64 >= n)) return;
  int Pn = 4;  // >= 4: the vectorised rank loop reads keys in quads
  while (Pn < n) Pn <<= 1;
  const float* key_src = cand + (a.use_score3d ? 5 : 4) * NS;
  float mx = -INFINITY;
  for (int i = tid; i < Pn; i += PT) {
    float key = -INFINITY;
    int val = 0x7fffffff;
    if (i < n) {
      int l = 0;
      while (l + 1 < L && i >= pref[l + 1]) ++l;
      const int slot = slot_base(a.slot_off, L, a.topk, l) + (i - pref[l]);
      key = key_src[slot];
      val = slot;  // slots grow with the concatenated (level-major) index => ties keep the Instances.cat order
      mx = fmaxf(mx, fmaxf(fmaxf(cand[0 * NS + slot], cand[1 * NS + slot]), fmaxf(cand[2 * NS + slot], cand[3 * NS + slot])));
    }
    keys[i] = key;
    vals[i] = val;
  }
  // boxes.max() for the coordinate trick
  for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = red[0];
  for (int w = 1; w < PT / 64; ++w) mx = fmaxf(mx, red[w]);
  const int mode = !suppress ? NMS_NONE : (4 * n > 4000 ? NMS_PER_CLASS : NMS_TRICK);  // torchvision 0.10 batched_nms
  if (s == 0 && tid == 0) {
    a.nvalid[2 * g] = n;
    a.nvalid[2 * g + 1] = mode;
  }
  const float off_unit = mx + 1.0f;
  auto emit = [&](int p, int slot) {  // sorted position p <- candidate slot
    const int c = __float_as_int(cand[6 * NS + slot]);
    const float off = mode == NMS_TRICK ? (float)c * off_unit : 0.f;
    a.sort_idx[(long)g * P.ncap + p] = slot;
    a.scls[(long)g * P.ncap + p] = c;
    *reinterpret_cast<float4*>(a.sbox + ((long)g * P.ncap + p) * 4) =
        make_float4(cand[0 * NS + slot] + off, cand[1 * NS + slot] + off, cand[2 * NS + slot] + off, cand[3 * NS + slot] + off);
  };
  if (split) {
    const int e = s * 64 + (tid >> 4), part = tid & 15;
    const int chunk = Pn >= 64 ? Pn >> 4 : 4;
    const int lo = part * chunk, hi = min(Pn, lo + chunk);
    int rank = 0;
    if (e < n && lo < hi) rank = rank_of(keys, lo, hi, keys[e], e);
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) rank += __shfl_xor(rank, d, 64);
    if (part == 0 && e < n) emit(rank, vals[e]);
    return;
  }
  if (suppress) block_bitonic_sort(keys, vals, Pn);
  __syncthreads();
  for (int p = tid; p < n; p += PT) emit(p, vals[p]);
}

// Upper-triangular tile t of an nwords x nwords tile grid -> (rb, cb), cb >= rb; rows first.
__device__ __forceinline__ void tile_of(int t, int nwords, int& rb, int& cb) {
  rb = 0;
  while (t >= nwords - rb) {
    t -= nwords - rb;
    ++rb;
  }
  cb = rb + t;
}

constexpr int MASK_BLOCKS = 256;  // blocks per image of the mask kernels; each walks the tiles t = blockIdx.x, + MASK_BLOCKS, ...
constexpr int MASK_WAVES = 4;     // a 64 x 64 tile is shared by four waves, 16 partners each

// [ext] torchvision nms_kernel.  Word (i, cb) of the mask, cb > i/64: bit j = IoU(i, cb*64+j) > thr (row form, i sorts first).
// The DIAGONAL word (i, i/64) is stored in COLUMN form: bit r = row (i/64)*64 + r, r < i%64, suppresses i when it is kept -- what
// the fixed-point resolution of greedy_reduce consumes.  Both forms evaluate the same expression with the earlier box as `a`.
__global__ __launch_bounds__(64 * MASK_WAVES) void nms_mask_kernel(const NmsK P) {
  const dd3d_nms_args& a = P.a;
  const int g = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = a.nvalid[2 * g], mode = a.nvalid[2 * g + 1];
  if (mode == NMS_NONE) return;
  const int nwords = (n + 63) / 64, ntiles = nwords * (nwords + 1) / 2;
  __shared__ float rbox[64][4], cbox[64][4];
  __shared__ int rcls[64], ccls[64];
  __shared__ unsigned long long part[MASK_WAVES][64];
  const float* sbox = a.sbox + (long)g * P.ncap * 4;
  const int* scls = a.scls + (long)g * P.ncap;
  for (int t = blockIdx.x; t < ntiles; t += MASK_BLOCKS) {
    int rb, cb;
    tile_of(t, nwords, rb, cb);
    if (wave < 2) {  // wave 0 stages the row boxes, wave 1 the column boxes
      const int src = (wave == 0 ? rb : cb) * 64 + lane;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      int c = -1;
      if (src < n) v = *reinterpret_cast<const float4*>(sbox + (long)src * 4), c = scls[src];
      float* dst = wave == 0 ? rbox[lane] : cbox[lane];
      dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
      (wave == 0 ? rcls : ccls)[lane] = c;
    }
    __syncthreads();
    // lane = row i of the tile (off-diagonal) or column j of the tile (diagonal); partners = the other side, 16 per wave
    const bool diag = rb == cb;
    const float* mine = diag ? cbox[lane] : rbox[lane];
    const float m0 = mine[0], m1 = mine[1], m2 = mine[2], m3 = mine[3];
    const int mc = diag ? ccls[lane] : rcls[lane];
    const float sm = (m2 - m0) * (m3 - m1);
    const int me = (diag ? cb : rb) * 64 + lane;
    unsigned long long bits = 0;
    if (me < n) {
      for (int q = 0; q < 64 / MASK_WAVES; ++q) {
        const int k = wave * (64 / MASK_WAVES) + q;
        const float* o = diag ? rbox[k] : cbox[k];
        const int other = (diag ? rb : cb) * 64 + k;
        if (other >= n || (diag && k >= lane)) continue;
        if (mode == NMS_PER_CLASS && (diag ? rcls[k] : ccls[k]) != mc) continue;
        const float left = fmaxf(m0, o[0]), right = fminf(m2, o[2]);
        const float top = fmaxf(m1, o[1]), bottom = fminf(m3, o[3]);
        const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
        const float inter = w * h;
        const float so = (o[2] - o[0]) * (o[3] - o[1]);
        // sa = area of the earlier (row) box, sb = of the later one: a float sum is commutative, so one expression serves both forms
        if (inter / (sm + so - inter) > a.nms_thresh) bits |= 1ull << k;
      }
    }
    part[wave][lane] = bits;
    __syncthreads();
    if (wave == 0 && me < n) {
      unsigned long long v = part[0][lane];
#pragma unroll
      for (int w = 1; w < MASK_WAVES; ++w) v |= part[w][lane];
      a.mask[((long)g * P.ncap + me) * (P.ncap / 64) + cb] = v;
    }
  }
}

// Greedy suppression over a precomputed bit mask, one block of PT threads.  word(i, cw) returns word cw of row i (see
// nms_mask_kernel: row form above the diagonal, column form ON it).  Block row rb = candidates rb*64 .. rb*64+63.
//   * Wave 0 resolves the block row from the diagonal words alone.  keep_j = alive_j and no kept earlier row of the block
//     suppresses j has exactly one solution (induction on j), and the iteration keep <- ballot(alive & !(col & keep)) fixes one
//     more leading candidate per round at least, so it reaches that solution in <= 64 rounds -- in practice in as many rounds as
//     the longest suppression chain of the block (a handful).  The previous form walked all 64 candidates with two v_readlane
//     per step (~2 k cycles per block row, the largest term of the kernel).
//   * Meanwhile wave w holds word rb+w of the 64 rows (one lane per row); after the keep bits are published each wave ORs the
//     words of the kept rows with a 6-step butterfly and folds them into removed[rb+w].
// on_row(rb, keepbits) runs on wave 0 (all 64 lanes) for every block row, in order.  [rb_begin, rb_end): the block rows to resolve
// (callers that stage the mask in pieces walk the rows chunk by chunk).
template <class W, class F>
__device__ __forceinline__ void greedy_reduce(W&& word_of, int n, unsigned long long* removed, F&& on_row, int rb_begin = 0, int rb_end = 1 << 30) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ unsigned long long sh_keepbits;
  const int nwords = (n + 63) / 64;
  for (int rb = rb_begin; rb < min(rb_end, nwords); ++rb) {
    const int i = rb * 64 + lane;
    if (wave == 0) {
      const unsigned long long col = i < n ? word_of(i, rb) : 0ull;
      const bool alive = i < n && !((removed[rb] >> lane) & 1ull);
      unsigned long long keep = __ballot(alive);
      for (;;) {
        const unsigned long long next = __ballot(alive && (col & keep) == 0ull);
        if (next == keep) break;
        keep = next;
      }
      if (lane == 0) sh_keepbits = keep;
      on_row(rb, keep);
    }
    // words of the later columns: wave w -> word rb + w (+16, +32, ...), lane -> row; issued before the barrier so the loads
    // overlap wave 0's part
    unsigned long long word[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int cw = rb + wave + 16 * q + (wave == 0 ? 16 : 0);  // wave 0 owns the diagonal; its first later word is rb + 16
      word[q] = (cw < nwords && i < n) ? word_of(i, cw) : 0ull;
    }
    __syncthreads();
    const unsigned long long kb = sh_keepbits;
    const bool mine = (kb >> lane) & 1ull;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int cw = rb + wave + 16 * q + (wave == 0 ? 16 : 0);
      if (cw >= nwords) break;  // wave-uniform
      unsigned long long v = mine ? word[q] : 0ull;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) v |= __shfl_xor(v, d, 64);
      if (lane == 0) removed[cw] |= v;
    }
    __syncthreads();
  }
}

constexpr int FIN_SMALL = 1024;  // up to this many candidates the whole upper triangle of the mask is staged in LDS (128 KiB)
constexpr int FIN_LDS_BYTES = 156 * 1024;  // dynamic LDS of nms_finalize_kernel (the rest of the CU's 160 KiB: its static arrays)

__global__ __launch_bounds__(PT) void nms_finalize_kernel(const NmsK P) {
  const dd3d_nms_args& a = P.a;
  const int g = blockIdx.x, tid = threadIdx.x;
  const int L = a.num_levels, NS = slot_base(a.slot_off, L, a.topk, L);
  const float* cand = a.cand + rec_off(g, a.img_first, a.img_per_rec, a.rec_stride, DD3D_CAND_FIELDS * NS);
  const int n = a.nvalid[2 * g], mode = a.nvalid[2 * g + 1];
  const int* sort_idx = a.sort_idx + (long)g * P.ncap;
  const int nw = P.ncap / 64;
  const unsigned long long* mask = reinterpret_cast<const unsigned long long*>(a.mask) + (long)g * P.ncap * nw;

  // removed | kept | sidx | stage, the two arrays `cap` entries each (n <= FIN_SMALL: cap = FIN_SMALL).  `stage` holds mask words
  // during the greedy pass and the top-k scratch (tkeys | tvals, `cap` entries each) after it.
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  const bool small = n <= FIN_SMALL && mode != NMS_NONE;
  const int cap = small ? FIN_SMALL : P.ncap2;
  unsigned long long* removed = reinterpret_cast<unsigned long long*>(dyn_lds);  // [NCAP_MAX/64]
  int* kept = reinterpret_cast<int*>(removed + NCAP_MAX / 64);  // sorted positions that survive NMS, in order
  int* sidx = kept + cap;                                       // sort_idx staged once (cuts a level off every gather below)
  unsigned long long* stage = reinterpret_cast<unsigned long long*>(sidx + cap);
  float* tkeys = reinterpret_cast<float*>(stage);               // scratch for the top-k threshold
  int* tvals = reinterpret_cast<int*>(tkeys + cap);
  const int stage_words = (P.fin_lds - NCAP_MAX / 64 * 8 - cap * 8) / 8;
  for (int i = tid; i < n; i += PT) sidx[i] = sort_idx[i];
  __shared__ int wsum[PT / 64];
  __shared__ int sh_nkeep;
  __shared__ float sh_thr;

  int nkeep = 0;
  if (mode == NMS_NONE) {
    for (int i = tid; i < n; i += PT) kept[i] = i;
    nkeep = n;
    __syncthreads();
  } else {
    for (int i = tid; i < NCAP_MAX / 64; i += PT) removed[i] = 0;
    if (tid == 0) sh_nkeep = 0;
    auto on_row = [&](int rb, unsigned long long keepbits) {
      // append the kept positions in order (wave 0, lane = row of the block row)
      const int base = sh_nkeep;
      if ((keepbits >> tid) & 1ull) kept[base + __popcll(keepbits & ((1ull << tid) - 1ull))] = rb * 64 + tid;
      if (tid == 0) sh_nkeep = base + __popcll(keepbits);
    };
    // The mask was written by other CUs, usually of other XCDs: a dependent trip to the L2 / fabric costs ~1.5 us (measured), and one
    // such trip per block row was most of this kernel.  The words are therefore brought into LDS in as few trips as the LDS allows,
    // <= 16 independent loads per thread per trip.
    const int nwords = (n + 63) / 64;
    if (small) {
      // everything at once: one row per thread, its words from the diagonal on; word cw of row i at cw * FIN_SMALL + i
      unsigned long long w[FIN_SMALL / 64];
#pragma unroll
      for (int q = 0; q < FIN_SMALL / 64; ++q) {
        const int cw = (tid >> 6) + q;
        w[q] = (tid < n && cw < nwords) ? mask[(long)tid * nw + cw] : 0ull;
      }
#pragma unroll
      for (int q = 0; q < FIN_SMALL / 64; ++q) {
        const int cw = (tid >> 6) + q;
        if (cw < FIN_SMALL / 64) stage[cw * FIN_SMALL + tid] = w[q];
      }
      __syncthreads();
      greedy_reduce([&](int i, int cw) { return stage[cw * FIN_SMALL + i]; }, n, removed, on_row);
    } else {
      // chunks of R block rows: the R * 64 rows' words from column rb0 on, row-major with an odd pitch (conflict-free both ways)
      __syncthreads();
      for (int rb0 = 0; rb0 < nwords;) {
        const int wrow = nwords - rb0, pitch = wrow | 1;
        const int R = min(wrow, min(stage_words / (64 * pitch), 16 * PT / (64 * wrow)));  // fits the stage and 16 loads per thread
        if (R < 1) {  // (not reachable with NCAP_MAX = 8192: one block row of 128 words needs 8256 stage words)
          greedy_reduce([&](int i, int cw) { return mask[(long)i * nw + cw]; }, n, removed, on_row, rb0, nwords);
          break;
        }
        const int total = R * 64 * wrow;
        unsigned long long w[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int idx = tid + k * PT;
          const int rl = idx / wrow, c = idx - rl * wrow, i = rb0 * 64 + rl;
          w[k] = (idx < total && i < n) ? mask[(long)i * nw + rb0 + c] : 0ull;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int idx = tid + k * PT;
          const int rl = idx / wrow, c = idx - rl * wrow;
          if (idx < total) stage[rl * pitch + c] = w[k];
        }
        __syncthreads();
        greedy_reduce([&](int i, int cw) { return stage[(i - rb0 * 64) * pitch + (cw - rb0)]; }, n, removed, on_row, rb0, rb0 + R);
        rb0 += R;
      }
    }
    nkeep = sh_nkeep;
  }

  // ---- top-k on the 2D score with >= (fcos2d.py:359-365).  kthvalue(n-k+1 smallest) == k-th largest.
  const float* score2d = cand + 4 * NS;
  float thr = -INFINITY;
  if (a.do_nms && a.post_topk > 0 && nkeep > a.post_topk) {
    if (nkeep <= PT) {
      // one kept detection per thread: it is the k-th largest iff (#greater < k <= #greater-or-equal)
      const float my = tid < nkeep ? score2d[sidx[kept[tid]]] : -INFINITY;
      if (tid < ((nkeep + 3) & ~3)) tkeys[tid] = my;  // -inf padding up to a multiple of 4 (nkeep <= PT)
      __syncthreads();
      if (tid < nkeep) {
        int gt = 0, ge = 0;
#pragma unroll 4
        for (int j0 = 0; j0 < nkeep; j0 += 4) {  // one broadcast ds_read_b128 per four candidates
          const float4 v = *reinterpret_cast<const float4*>(tkeys + j0);
          gt += (v.x > my) + (v.y > my) + (v.z > my) + (v.w > my);
          ge += (v.x >= my) + (v.y >= my) + (v.z >= my) + (v.w >= my);
        }
        if (gt < a.post_topk && a.post_topk <= ge) sh_thr = my;  // every thread that gets here holds the same value
      }
      __syncthreads();
    } else {
      int Pn = 1;
      while (Pn < nkeep) Pn <<= 1;
      for (int i = tid; i < Pn; i += PT) {
        tkeys[i] = i < nkeep ? score2d[sidx[kept[i]]] : -INFINITY;
        tvals[i] = i;
      }
      __syncthreads();
      block_bitonic_sort(tkeys, tvals, Pn);
      if (tid == 0) sh_thr = tkeys[a.post_topk - 1];
      __syncthreads();
    }
    thr = sh_thr;
  }

  // ---- resize / clip / drop empty ([ext] detector_postprocess) + ordered write-out
  const float* osz = a.out_size + rec_off(g, a.img_first, a.img_per_rec, a.rec_stride, 4);
  const float in_h = osz[0], in_w = osz[1], out_h = osz[2], out_w = osz[3];
  const float sx = out_w / in_w, sy = out_h / in_h;
  float* det = a.det + (long)g * a.det_cap * DD3D_DET_FIELDS;
  int running = 0;
  for (int base = 0; base < nkeep; base += PT) {
    const int i = base + tid;
    int pass = 0, slot = 0;
    float x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    if (i < nkeep) {
      slot = sidx[kept[i]];
      pass = score2d[slot] >= thr;
      x1 = cand[0 * NS + slot], y1 = cand[1 * NS + slot], x2 = cand[2 * NS + slot], y2 = cand[3 * NS + slot];
      if (a.do_postprocess) {
        x1 = fminf(fmaxf(x1 * sx, 0.f), out_w), x2 = fminf(fmaxf(x2 * sx, 0.f), out_w);
        y1 = fminf(fmaxf(y1 * sy, 0.f), out_h), y2 = fminf(fmaxf(y2 * sy, 0.f), out_h);
        pass = pass && (x2 - x1) > 0.f && (y2 - y1) > 0.f;
      }
    }
    int total;
    const int pos = running + block_excl_scan(pass, wsum, total);
    if (pass && pos < a.det_cap) {
      float* d = det + (long)pos * DD3D_DET_FIELDS;
      d[0] = x1, d[1] = y1, d[2] = x2, d[3] = y2;
      d[4] = cand[4 * NS + slot];
      d[5] = cand[5 * NS + slot];
      d[6] = (float)__float_as_int(cand[6 * NS + slot]);
      d[7] = (float)level_of_slot(a.slot_off, L, a.topk, slot);
#pragma unroll
      for (int f = 8; f < 20; ++f) d[f] = cand[f * NS + slot];
      d[20] = (float)__float_as_int(cand[20 * NS + slot]);
      d[21] = cand[21 * NS + slot];
#pragma unroll
      for (int f = 22; f < DD3D_DET_FIELDS; ++f) d[f] = 0.f;
    }
    running += total;
  }
  if (tid == 0) a.det_count[g] = running;
}


// ================================================================================================ BEV rotated NMS
// nuScenes sample aggregation (tridet/modeling/dd3d/postprocessing.py:22-108, tridet/layers/bev_nms.py:51-133):
// camera-frame boxes -> global frame with the image's pose, top-face corners -> BEV rotated boxes, class- and
// sample-aware rotated-IoU NMS ([ext] detectron2 batched_nms_rotated / box_iou_rotated), optional cap on the number of
// kept boxes, survivors written back per image in their original order.  All images of the batch form ONE problem, as
// in the reference (one batched_nms_rotated call over the concatenated instances).
struct BevK {
  dd3d_bev_args a;
  int ncap;   // round_up(G*det_cap, 64)
  int ncap2;  // LDS sort capacity (power of two)
  int mw;     // 64-bit words per mask row = min(ncap, NCAP_MAX) / 64: rows and columns are SORTED positions, and the sorter holds <= NCAP_MAX boxes
};

struct V2 {
  float x, y;
};
__device__ __forceinline__ float cross2(V2 a, V2 b) { return a.x * b.y - b.x * a.y; }
__device__ __forceinline__ float dot2(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ V2 sub2(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }

// [ext] get_rotated_vertices
__device__ void rotated_vertices(float cx, float cy, float w, float h, float a, V2* p) {
  const double theta = (double)a * 0.01745329251;
  const float c = (float)cos(theta) * 0.5f, s = (float)sin(theta) * 0.5f;
  p[0] = V2{cx - s * h - c * w, cy + c * h - s * w};
  p[1] = V2{cx + s * h - c * w, cy - c * h - s * w};
  p[2] = V2{2 * cx - p[0].x, 2 * cy - p[0].y};
  p[3] = V2{2 * cx - p[1].x, 2 * cy - p[1].y};
}

// [ext] single_box_iou_rotated: intersection polygon = edge crossings + contained vertices, Graham hull, fan area
__device__ float rotated_iou(const float* b1, const float* b2) {
  const float area1 = b1[2] * b1[3], area2 = b2[2] * b2[3];
  if (area1 < 1e-14f || area2 < 1e-14f) return 0.f;
  const float sx = (b1[0] + b2[0]) / 2.0f, sy = (b1[1] + b2[1]) / 2.0f;
  V2 p1[4], p2[4], v1[4], v2[4], q[24];
  rotated_vertices(b1[0] - sx, b1[1] - sy, b1[2], b1[3], b1[4], p1);
  rotated_vertices(b2[0] - sx, b2[1] - sy, b2[2], b2[3], b2[4], p2);
  const float EPS = 1e-5f;
  int num = 0;
  for (int i = 0; i < 4; ++i) {
    v1[i] = sub2(p1[(i + 1) & 3], p1[i]);
    v2[i] = sub2(p2[(i + 1) & 3], p2[i]);
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const float det = cross2(v2[j], v1[i]);
      if (fabsf(det) <= 1e-14f) continue;
      const V2 v12 = sub2(p2[j], p1[i]);
      const float t1 = cross2(v2[j], v12) / det, t2 = cross2(v1[i], v12) / det;
      if (t1 > -EPS && t1 < 1.0f + EPS && t2 > -EPS && t2 < 1.0f + EPS) q[num++] = V2{p1[i].x + v1[i].x * t1, p1[i].y + v1[i].y * t1};
    }
  {
    const V2 AB = v2[0], DA = v2[3];
    const float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
    for (int i = 0; i < 4; ++i) {
      const V2 AP = sub2(p1[i], p2[0]);
      const float APdotAB = dot2(AP, AB), APdotAD = -dot2(AP, DA);
      if (APdotAB > -EPS && APdotAD > -EPS && APdotAB < ABdotAB + EPS && APdotAD < ADdotAD + EPS) q[num++] = p1[i];
    }
  }
  {
    const V2 AB = v1[0], DA = v1[3];
    const float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
    for (int i = 0; i < 4; ++i) {
      const V2 AP = sub2(p2[i], p1[0]);
      const float APdotAB = dot2(AP, AB), APdotAD = -dot2(AP, DA);
      if (APdotAB > -EPS && APdotAD > -EPS && APdotAB < ABdotAB + EPS && APdotAD < ADdotAD + EPS) q[num++] = p2[i];
    }
  }
  if (num <= 2) return 0.f;
  // convex_hull_graham
  int t = 0;
  for (int i = 1; i < num; ++i)
    if (q[i].y < q[t].y || (q[i].y == q[t].y && q[i].x < q[t].x)) t = i;
  const V2 start = q[t];
  for (int i = 0; i < num; ++i) q[i] = sub2(q[i], start);
  {
    const V2 tmp = q[0];
    q[0] = q[t];
    q[t] = tmp;
  }
  float dist[24];
  for (int i = 0; i < num; ++i) dist[i] = dot2(q[i], q[i]);
  for (int i = 1; i < num - 1; ++i)
    for (int j = i + 1; j < num; ++j) {
      const float cp = cross2(q[i], q[j]);
      if (cp < -1e-6f || (fabsf(cp) < 1e-6f && dist[i] > dist[j])) {
        const V2 tq = q[i];
        q[i] = q[j];
        q[j] = tq;
        const float td = dist[i];
        dist[i] = dist[j];
        dist[j] = td;
      }
    }
  int k = 1;
  while (k < num && dist[k] <= 1e-8f) ++k;
  if (k == num) return 0.f;
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < num; ++i) {
    while (m > 1 && cross2(sub2(q[i], q[m - 2]), sub2(q[m - 1], q[m - 2])) >= 0.f) --m;
    q[m++] = q[i];
  }
  if (m <= 2) return 0.f;
  float inter = 0.f;
  for (int i = 1; i < m - 1; ++i) inter += fabsf(cross2(sub2(q[i], q[0]), sub2(q[i + 1], q[0])));
  inter /= 2.0f;
  return inter / (area1 + area2 - inter);
}

__device__ __forceinline__ void quat_to_mat(const float* q, float* R) {  // [ext] pytorch3d quaternion_to_matrix
  const float r = q[0], i = q[1], j = q[2], k = q[3];
  const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
  R[0] = 1 - two_s * (j * j + k * k), R[1] = two_s * (i * j - k * r), R[2] = two_s * (i * k + j * r);
  R[3] = two_s * (i * j + k * r), R[4] = 1 - two_s * (i * i + k * k), R[5] = two_s * (j * k - i * r);
  R[6] = two_s * (i * k - j * r), R[7] = two_s * (j * k + i * r), R[8] = 1 - two_s * (i * i + j * j);
}

__device__ __forceinline__ void mat_to_quat(const float* m, float* q) {  // [ext] pytorch3d 0.5/0.6 matrix_to_quaternion
  const float t0 = 1.f + m[0] + m[4] + m[8], t1 = 1.f + m[0] - m[4] - m[8], t2 = 1.f - m[0] + m[4] - m[8], t3 = 1.f - m[0] - m[4] + m[8];
  const float a0 = t0 > 0.f ? sqrtf(t0) : 0.f, a1 = t1 > 0.f ? sqrtf(t1) : 0.f, a2 = t2 > 0.f ? sqrtf(t2) : 0.f, a3 = t3 > 0.f ? sqrtf(t3) : 0.f;
  int best = 0;
  float am = a0;
  if (a1 > am) best = 1, am = a1;
  if (a2 > am) best = 2, am = a2;
  if (a3 > am) best = 3, am = a3;
  const float den = 2.0f * fmaxf(am, 0.1f);
  if (best == 0) q[0] = a0 * a0, q[1] = m[7] - m[5], q[2] = m[2] - m[6], q[3] = m[3] - m[1];
  else if (best == 1) q[0] = m[7] - m[5], q[1] = a1 * a1, q[2] = m[3] + m[1], q[3] = m[2] + m[6];
  else if (best == 2) q[0] = m[2] - m[6], q[1] = m[3] + m[1], q[2] = a2 * a2, q[3] = m[5] + m[7];
  else q[0] = m[3] - m[1], q[1] = m[6] + m[2], q[2] = m[7] + m[5], q[3] = a3 * a3;
  q[0] /= den, q[1] /= den, q[2] /= den, q[3] /= den;
}

// 1 block: global boxes, BEV rotated boxes, coordinate range, stable sort by score_3d; writes the sorted work arrays.
__global__ __launch_bounds__(PT) void bev_prepare_kernel(const BevK P) {
  const dd3d_bev_args& a = P.a;
  const int tid = threadIdx.x;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  float* keys = reinterpret_cast<float*>(dyn_lds);
  int* vals = reinterpret_cast<int*>(dyn_lds) + P.ncap2;
  __shared__ int offs[1025];
  __shared__ float redmax[PT / 64], redmin[PT / 64];
  if (tid == 0) {
    int acc = 0;
    for (int g = 0; g < a.G; ++g) {
      offs[g] = acc;
      acc += min(a.count_in[g], a.det_cap);
    }
    offs[a.G] = acc;
  }
  __syncthreads();
  const int n = offs[a.G];
  const int ok = n <= P.ncap2;
  if (tid == 0) {
    a.meta[0] = ok ? n : 0;
    a.meta[1] = ok ? 0 : 1;  // overflow flag
  }
  if (!ok) return;
  float mx = -INFINITY, mn = INFINITY;
  for (int i = tid; i < n; i += PT) {
    int g = 0;
    while (g + 1 < a.G && i >= offs[g + 1]) ++g;
    const float* d = a.det_in + ((long)g * a.det_cap + (i - offs[g])) * DD3D_DET_FIELDS;
    const float* K = a.inv_K + rec_off(g, a.img_first, a.img_per_rec, a.rec_stride, 9);
    // camera-frame box: tvec = K^-1 [proj_ctr, 1] * depth  (boxes3d.py:169-173)
    const float u = d[14], v = d[15], dep = d[16];
    const float tS[3] = {(K[0] * u + K[1] * v + K[2]) * dep, (K[3] * u + K[4] * v + K[5]) * dep, (K[6] * u + K[7] * v + K[8]) * dep};
    float R_SO[9], R_WS[9], R_WO[9], qW[4], tW[3], R[9];
    quat_to_mat(d + 10, R_SO);
    const float* ps = a.pose + rec_off(g, a.img_first, a.img_per_rec, a.rec_stride, 7);
    quat_to_mat(ps, R_WS);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) R_WO[3 * r + c] = R_WS[3 * r] * R_SO[c] + R_WS[3 * r + 1] * R_SO[3 + c] + R_WS[3 * r + 2] * R_SO[6 + c];
      tW[r] = R_WS[3 * r] * tS[0] + R_WS[3 * r + 1] * tS[1] + R_WS[3 * r + 2] * tS[2] + ps[4 + r];
    }
    mat_to_quat(R_WO, qW);
    quat_to_mat(qW, R);
    // top-face corners 0 (+,+,+), 1 (+,-,+), 5 (-,-,+), 4 (-,+,+) of (l, w, h) = (size[1], size[0], size[2])  (boxes3d.py:12-16,47-64)
    const float hl = 0.5f * d[18], hw = 0.5f * d[17], hh = 0.5f * d[19];
    const float sgl[4] = {1.f, 1.f, -1.f, -1.f}, sgw[4] = {1.f, -1.f, -1.f, 1.f};
    V2 bev[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float ox = sgl[c] * hl, oy = sgw[c] * hw, oz = hh;
      const float X = R[0] * ox + R[1] * oy + R[2] * oz + tW[0];
      const float Y = R[3] * ox + R[4] * oy + R[5] * oz + tW[1];
      bev[c] = V2{-Y, -X};  // VEHICLE_TO_BEV_ROTATION (bev_nms.py:42-47)
    }
    const V2 fwd = sub2(bev[0], bev[3]), sid = sub2(bev[0], bev[1]);
    const float length = sqrtf(fwd.x * fwd.x + fwd.y * fwd.y), width = sqrtf(sid.x * sid.x + sid.y * sid.y);
    const float cx = (bev[0].x + bev[2].x) / 2.0f, cy = (bev[0].y + bev[2].y) / 2.0f;
    const float ang = 57.29577951308232f * atan2f(fwd.x, fwd.y);
    float* w = a.work + (long)i * 16;  // 0-4 rotated box, 5 category, 6-9 global quat, 10-12 global tvec
    w[0] = cx, w[1] = cy, w[2] = width, w[3] = length, w[4] = ang;
    w[5] = __int_as_float((int)d[6] + a.group[g] * a.num_classes);
    w[6] = qW[0], w[7] = qW[1], w[8] = qW[2], w[9] = qW[3], w[10] = tW[0], w[11] = tW[1], w[12] = tW[2];
    keys[i] = d[5];
    vals[i] = i;
    const float ext = fmaxf(width, length) / 2.0f;
    mx = fmaxf(mx, fmaxf(cx, cy) + ext);
    mn = fminf(mn, fminf(cx, cy) - ext);
  }
  int Pn = 4;
  while (Pn < n) Pn <<= 1;
  for (int i = n + tid; i < Pn; i += PT) keys[i] = -INFINITY, vals[i] = 0x7fffffff;
  for (int dlt = 32; dlt > 0; dlt >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, dlt, 64));
    mn = fminf(mn, __shfl_xor(mn, dlt, 64));
  }
  if ((tid & 63) == 0) redmax[tid >> 6] = mx, redmin[tid >> 6] = mn;
  __syncthreads();
  mx = redmax[0], mn = redmin[0];
  for (int wv = 1; wv < PT / 64; ++wv) mx = fmaxf(mx, redmax[wv]), mn = fminf(mn, redmin[wv]);
  if (n <= PT) {
    const float ki = tid < n ? keys[tid] : 0.f;
    const int vi = tid < n ? vals[tid] : 0;
    int rank = 0;
    if (tid < n) rank = rank_of(keys, 0, Pn, ki, tid);
    __syncthreads();
    if (tid < n) keys[rank] = ki, vals[rank] = vi;
  } else {
    block_bitonic_sort(keys, vals, Pn);
  }
  __syncthreads();
  const float unit = mx - mn + 1.0f;  // [ext] batched_nms_rotated offset unit
  for (int p = tid; p < n; p += PT) {
    const int i = vals[p];
    const float* w = a.work + (long)i * 16;
    const int cat = __float_as_int(w[5]);
    const float off = (float)cat * unit;
    float* sb = a.sbox + (long)p * 8;
    sb[0] = w[0] + off, sb[1] = w[1] + off, sb[2] = w[2], sb[3] = w[3], sb[4] = w[4];
    sb[5] = __int_as_float(cat);
    sb[6] = __int_as_float(i);
  }
}

// Rotated-IoU mask of the aggregated boxes, laid out like nms_mask_kernel's (row form above the diagonal, column form on it;
// rotated_iou is always called with the earlier box first, as [ext] nms_rotated does).
__global__ __launch_bounds__(64 * MASK_WAVES) void bev_mask_kernel(const BevK P) {
  const dd3d_bev_args& a = P.a;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = a.meta[0];
  const int nwords = (n + 63) / 64, ntiles = nwords * (nwords + 1) / 2;
  __shared__ float rbox[64][6], cbox[64][6];
  __shared__ unsigned long long part[MASK_WAVES][64];
  for (int t = blockIdx.x; t < ntiles; t += MASK_BLOCKS) {
    int rb, cb;
    tile_of(t, nwords, rb, cb);
    if (wave < 2) {
      const int src = (wave == 0 ? rb : cb) * 64 + lane;
      float* dst = wave == 0 ? rbox[lane] : cbox[lane];
#pragma unroll
      for (int f = 0; f < 6; ++f) dst[f] = src < n ? a.sbox[(long)src * 8 + f] : 0.f;
    }
    __syncthreads();
    const bool diag = rb == cb;
    float mine[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) mine[f] = diag ? cbox[lane][f] : rbox[lane][f];
    const int me = (diag ? cb : rb) * 64 + lane;
    unsigned long long bits = 0;
    if (me < n) {
      for (int q = 0; q < 64 / MASK_WAVES; ++q) {
        const int k = wave * (64 / MASK_WAVES) + q;
        const float* o = diag ? rbox[k] : cbox[k];
        const int other = (diag ? rb : cb) * 64 + k;
        if (other >= n || (diag && k >= lane)) continue;
        if (__float_as_int(o[5]) != __float_as_int(mine[5])) continue;  // other categories are offset out of reach
        const float iou = diag ? rotated_iou(o, mine) : rotated_iou(mine, o);
        if (iou > a.iou_thresh) bits |= 1ull << k;
      }
    }
    part[wave][lane] = bits;
    __syncthreads();
    if (wave == 0 && me < n) {
      unsigned long long v = part[0][lane];
#pragma unroll
      for (int w = 1; w < MASK_WAVES; ++w) v |= part[w][lane];
      a.mask[(long)me * P.mw + cb] = v;
    }
  }
}

// 1 block: greedy reduce, cap, per-image ordered write-out (optionally with the resize / clip / non-empty filter).
__global__ __launch_bounds__(PT) void bev_finalize_kernel(const BevK P) {
  const dd3d_bev_args& a = P.a;
  const int tid = threadIdx.x;
  const int n = a.meta[0];
  const int nw = P.mw;
  const unsigned long long* mask = reinterpret_cast<const unsigned long long*>(a.mask);
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  unsigned long long* removed = reinterpret_cast<unsigned long long*>(dyn_lds);          // [NCAP_MAX/64]
  int* keepflag = reinterpret_cast<int*>(removed + NCAP_MAX / 64);                       // [ncap2] by ORIGINAL index
  __shared__ int wsum[PT / 64];
  __shared__ int offs[1025];
  __shared__ int sh_nkeep;
  if (tid == 0) {
    int acc = 0;
    for (int g = 0; g < a.G; ++g) {
      offs[g] = acc;
      acc += min(a.count_in[g], a.det_cap);
    }
    offs[a.G] = acc;
    sh_nkeep = 0;
  }
  for (int i = tid; i < NCAP_MAX / 64; i += PT) removed[i] = 0;
  for (int i = tid; i < P.ncap2; i += PT) keepflag[i] = 0;
  __syncthreads();
  if (a.meta[1]) {  // more boxes than the sorter holds: report, keep nothing
    if (tid < a.G) a.count_out[tid] = -1;
    return;
  }
  const int cap = a.max_dets > 0 ? a.max_dets : 0x7fffffff;
  greedy_reduce([&](int i, int cw) { return mask[(long)i * nw + cw]; }, n, removed, [&](int rb, unsigned long long keepbits) {
    const int base = sh_nkeep;
    // keep[:max_dets] truncates the score-ordered keep list of the WHOLE batch (postprocessing.py:93-94)
    if (((keepbits >> tid) & 1ull) && base + __popcll(keepbits & ((1ull << tid) - 1ull)) < cap)
      keepflag[__float_as_int(a.sbox[(long)(rb * 64 + tid) * 8 + 6])] = 1;
    if (tid == 0) sh_nkeep = base + __popcll(keepbits);
  });
  // survivors, image by image, in their original order
  int running = 0;
  for (int g = 0; g < a.G; ++g) {
    const float* osz = a.out_size + rec_off(g, a.img_first, a.img_per_rec, a.rec_stride, 4);
    const float sx = osz[3] / osz[1], sy = osz[2] / osz[0];
    int img_run = 0;
    for (int base = offs[g]; base < offs[g + 1]; base += PT) {
      const int i = base + tid;
      int pass = 0;
      const float* d = nullptr;
      float x1 = 0, y1 = 0, x2 = 0, y2 = 0;
      if (i < offs[g + 1]) {
        d = a.det_in + ((long)g * a.det_cap + (i - offs[g])) * DD3D_DET_FIELDS;
        pass = keepflag[i];
        x1 = d[0], y1 = d[1], x2 = d[2], y2 = d[3];
        if (a.do_postprocess) {
          x1 = fminf(fmaxf(x1 * sx, 0.f), osz[3]), x2 = fminf(fmaxf(x2 * sx, 0.f), osz[3]);
          y1 = fminf(fmaxf(y1 * sy, 0.f), osz[2]), y2 = fminf(fmaxf(y2 * sy, 0.f), osz[2]);
          pass = pass && (x2 - x1) > 0.f && (y2 - y1) > 0.f;
        }
      }
      int total;
      const int pos = img_run + block_excl_scan(pass, wsum, total);
      if (pass) {
        float* o = a.det_out + ((long)g * a.det_cap + pos) * DD3D_DET_FIELDS;
        o[0] = x1, o[1] = y1, o[2] = x2, o[3] = y2;
#pragma unroll
        for (int f = 4; f < 22; ++f) o[f] = d[f];
        const float* w = a.work + (long)i * 16;
#pragma unroll
        for (int f = 0; f < 7; ++f) o[22 + f] = a.write_global ? w[6 + f] : 0.f;
        o[29] = o[30] = o[31] = 0.f;
      }
      img_run += total;
    }
    if (tid == 0) a.count_out[g] = img_run;
    running += img_run;
  }
}

}  // namespace dd3d

extern "C" int dd3d_fcos_select_decode(const dd3d_select_args* args, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(args, "dd3d_fcos_select_decode: null args");
  DD3D_REQUIRE(args->num_levels > 0 && args->num_levels <= DD3D_MAX_LEVELS, "dd3d_fcos_select_decode: num_levels=%d", args->num_levels);
  DD3D_REQUIRE(args->topk > 0 && args->topk <= TOPK_MAX, "dd3d_fcos_select_decode: topk=%d exceeds %d", args->topk, TOPK_MAX);
  DD3D_REQUIRE(args->B > 0 && args->num_classes > 0, "dd3d_fcos_select_decode: B=%d classes=%d", args->B, args->num_classes);
  DD3D_REQUIRE(args->cand && args->counts && args->npass && args->scratch_idx && args->scratch_score, "dd3d_fcos_select_decode: null buffer");
  for (int l = 0; l < args->num_levels; ++l) DD3D_REQUIRE(args->cls[l] && args->box2d[l], "dd3d_fcos_select_decode: null head map at level %d", l);
  DD3D_REQUIRE(args->box3d[0] == nullptr || (args->inv_K && args->canon_sizes), "dd3d_fcos_select_decode: 3D decode needs inv_K and canon_sizes");
  SelectK P;
  P.a = *args;
  hipLaunchKernelGGL(fcos_select_decode_kernel, dim3(args->num_levels, args->B), dim3(PT), 0, reinterpret_cast<hipStream_t>(stream), P);
  return check_launch("fcos_select_decode_kernel");
}

extern "C" int dd3d_nms_finalize(const dd3d_nms_args* args, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(args, "dd3d_nms_finalize: null args");
  const int ns = slot_base(args->slot_off, args->num_levels, args->topk, args->num_levels);
  DD3D_REQUIRE(args->G > 0 && ns > 0 && ns <= NCAP_MAX, "dd3d_nms_finalize: %d candidate slots per image exceed %d", ns, NCAP_MAX);
  DD3D_REQUIRE(args->cand && args->counts && args->out_size && args->sort_idx && args->sbox && args->scls && args->mask && args->nvalid &&
                   args->det && args->det_count,
               "dd3d_nms_finalize: null buffer");
  DD3D_REQUIRE(args->det_cap > 0, "dd3d_nms_finalize: det_cap=%d", args->det_cap);
  DD3D_REQUIRE(args->img_per_rec >= 0 && args->img_first >= 0 && (args->img_per_rec == 0 || args->rec_stride > 0),
               "dd3d_nms_finalize: record addressing img_first=%d img_per_rec=%d rec_stride=%lld", args->img_first, args->img_per_rec, (long long)args->rec_stride);
  NmsK P;
  P.a = *args;
  P.ncap = (ns + 63) / 64 * 64;
  P.ncap2 = 64;
  while (P.ncap2 < ns) P.ncap2 <<= 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t lds_sort = (size_t)P.ncap2 * 8;
  // removed + kept + sidx + the mask stage / top-k scratch.  Images that can never exceed FIN_SMALL candidates take the staged-triangle
  // path only (137 KiB); larger ones get everything the CU has beside the kernel's static arrays (the stage then holds more block rows).
  const size_t lds_fin = P.ncap2 <= FIN_SMALL ? (size_t)NCAP_MAX / 64 * 8 + FIN_SMALL * 8 + (size_t)(FIN_SMALL / 64) * FIN_SMALL * 8 : (size_t)FIN_LDS_BYTES;
  P.fin_lds = (int)lds_fin;
  static unsigned long long attr_done[4];
  if (lds_opt_in_needed(attr_done)) {
    if (lds_opt_in(reinterpret_cast<const void*>(nms_sort_kernel), (size_t)(NCAP_MAX * 8), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
    if (lds_opt_in(reinterpret_cast<const void*>(nms_finalize_kernel), (size_t)(FIN_LDS_BYTES), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
    lds_opt_in_done(attr_done);  // (every opt-in of this call site succeeded on this device)
  }
  hipLaunchKernelGGL(nms_sort_kernel, dim3(args->G, SORT_SPLIT), dim3(PT), lds_sort, st, P);
  int rc = check_launch("nms_sort_kernel");
  if (rc != DD3D_OK) return rc;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(MASK_BLOCKS, args->G), dim3(64 * MASK_WAVES), 0, st, P);
  rc = check_launch("nms_mask_kernel");
  if (rc != DD3D_OK) return rc;
  hipLaunchKernelGGL(nms_finalize_kernel, dim3(args->G), dim3(PT), lds_fin, st, P);
  return check_launch("nms_finalize_kernel");
}

extern "C" int dd3d_bev_nms_aggregate(const dd3d_bev_args* args, void* stream) {
  using namespace dd3d;
  DD3D_REQUIRE(args, "dd3d_bev_nms_aggregate: null args");
  DD3D_REQUIRE(args->G > 0 && args->G <= 1024 && args->det_cap > 0, "dd3d_bev_nms_aggregate: G=%d det_cap=%d", args->G, args->det_cap);
  DD3D_REQUIRE(args->det_in && args->count_in && args->inv_K && args->pose && args->group && args->out_size && args->work && args->sbox &&
                   args->mask && args->meta && args->det_out && args->count_out,
               "dd3d_bev_nms_aggregate: null buffer");
  DD3D_REQUIRE(args->img_per_rec >= 0 && args->img_first >= 0 && (args->img_per_rec == 0 || args->rec_stride > 0),
               "dd3d_bev_nms_aggregate: record addressing img_first=%d img_per_rec=%d rec_stride=%lld", args->img_first, args->img_per_rec, (long long)args->rec_stride);
  BevK P;
  P.a = *args;
  const long ntot = (long)args->G * args->det_cap;
  P.ncap = (int)((ntot + 63) / 64 * 64);
  P.ncap2 = 64;
  while (P.ncap2 < ntot && P.ncap2 < NCAP_MAX) P.ncap2 <<= 1;
  P.mw = (P.ncap < NCAP_MAX ? P.ncap : NCAP_MAX) / 64;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t lds_prep = (size_t)P.ncap2 * 8;
  const size_t lds_fin = (size_t)NCAP_MAX / 64 * 8 + (size_t)P.ncap2 * 4;
  static unsigned long long attr_done[4];
  if (lds_opt_in_needed(attr_done)) {
    if (lds_opt_in(reinterpret_cast<const void*>(bev_prepare_kernel), (size_t)(NCAP_MAX * 8), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
    if (lds_opt_in(reinterpret_cast<const void*>(bev_finalize_kernel), (size_t)(NCAP_MAX / 64 * 8 + NCAP_MAX * 4), "dynamic LDS opt-in") != DD3D_OK) return DD3D_E_LAUNCH;
    lds_opt_in_done(attr_done);  // (every opt-in of this call site succeeded on this device)
  }
  hipLaunchKernelGGL(bev_prepare_kernel, dim3(1), dim3(PT), lds_prep, st, P);
  int rc = check_launch("bev_prepare_kernel");
  if (rc != DD3D_OK) return rc;
  hipLaunchKernelGGL(bev_mask_kernel, dim3(MASK_BLOCKS), dim3(64 * MASK_WAVES), 0, st, P);
  rc = check_launch("bev_mask_kernel");
  if (rc != DD3D_OK) return rc;
  hipLaunchKernelGGL(bev_finalize_kernel, dim3(1), dim3(PT), lds_fin, st, P);
  return check_launch("bev_finalize_kernel");
}
