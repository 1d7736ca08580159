// Evaluator-side overlap kernels for gfx950: the GPU code the reference itself authored (numba.cuda, tridet/evaluators/
// rotate_iou.py) -- rotated-box IoU matrix of the KITTI BEV / 3D AP computation -- plus the 3D and 2D overlaps that follow it.
//   rotate_iou_eval_kernel   rotate_iou_kernel_eval :260-289 -> devRotateIoUEval :254-258 -> inter :239-251 -> rbbox_to_corners
//                            :214-236, quadrilateral_intersection :183-211 (point_in_quadrilateral :161-180,
//                            line_segment_intersection :81-124), sort_vertex_in_convex_polygon :38-78, area :30-35
//   d3_overlap_kernel        d3_box_overlap_kernel :330-357
//   image_overlap_kernel     image_box_overlap :360-381
// One lane per (box, query) pair, 64x4 pairs per block, query boxes of the block's columns in LDS.  float32 throughout, in the
// reference's operation order (fp contraction off), so results agree with it to rounding of cos / sin / sqrt.
#include "common.h"

DD3D_NOTE_BUILD_FLAGS

#pragma clang fp contract(off)

namespace dd3d {

struct P2 {
  float x, y;
};

__device__ __forceinline__ void rbox_corners(const float* b, P2* c) {
  const float a_cos = cosf(b[4]), a_sin = sinf(b[4]);
  const float hx = b[2] / 2.f, hy = b[3] / 2.f;
  const float sx[4] = {-hx, -hx, hx, hx}, sy[4] = {-hy, hy, hy, -hy};
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = P2{a_cos * sx[i] + a_sin * sy[i] + b[0], -a_sin * sx[i] + a_cos * sy[i] + b[1]};
}

__device__ __forceinline__ bool in_quad(P2 p, const P2* q) {
  const float ab0 = q[1].x - q[0].x, ab1 = q[1].y - q[0].y;
  const float ad0 = q[3].x - q[0].x, ad1 = q[3].y - q[0].y;
  const float ap0 = p.x - q[0].x, ap1 = p.y - q[0].y;
  const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
  const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
  const float eps = 0.0001f;
  return abab >= abap - eps && abap >= 0.f - eps && adad >= adap - eps && adap >= 0.f - eps;
}

__device__ __forceinline__ bool seg_cross(P2 A, P2 B, P2 C, P2 D, P2* out) {
  const float BA0 = B.x - A.x, BA1 = B.y - A.y;
  const float DA0 = D.x - A.x, DA1 = D.y - A.y;
  const float CA0 = C.x - A.x, CA1 = C.y - A.y;
  const bool acd = DA1 * CA0 > CA1 * DA0;
  const bool bcd = (D.y - B.y) * (C.x - B.x) > (C.y - B.y) * (D.x - B.x);
  if (acd == bcd) return false;
  const bool abc = CA1 * BA0 > BA1 * CA0;
  const bool abd = DA1 * BA0 > BA1 * DA0;
  if (abc == abd) return false;
  const float DC0 = D.x - C.x, DC1 = D.y - C.y;
  const float ABBA = A.x * B.y - B.x * A.y;
  const float CDDC = C.x * D.y - D.x * C.y;
  const float DH = BA1 * DC0 - BA0 * DC1;
  *out = P2{(ABBA * DC0 - BA0 * CDDC) / DH, (ABBA * DC1 - BA1 * CDDC) / DH};
  return true;
}

__device__ float rbox_intersection(const float* b1, const float* b2) {
  P2 q1[4], q2[4], pts[24];  // the reference reserves 8 points; 24 covers every degenerate double count
  rbox_corners(b1, q1);
  rbox_corners(b2, q2);
  int n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (in_quad(q1[i], q2)) pts[n++] = q1[i];
    if (in_quad(q2[i], q1)) pts[n++] = q2[i];
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 x;
      if (seg_cross(q1[i], q1[(i + 1) & 3], q2[j], q2[(j + 1) & 3], &x)) pts[n++] = x;
    }
  if (n == 0) return 0.f;
  float cx = 0.f, cy = 0.f;
  for (int i = 0; i < n; ++i) cx += pts[i].x, cy += pts[i].y;
  cx /= (float)n, cy /= (float)n;
  float key[24];
  for (int i = 0; i < n; ++i) {
    float vx = pts[i].x - cx, vy = pts[i].y - cy;
    const float d = sqrtf(vx * vx + vy * vy);
    vx = vx / d, vy = vy / d;
    key[i] = vy < 0.f ? -2.f - vx : vx;
  }
  for (int i = 1; i < n; ++i) {  // insertion sort, as the reference
    if (key[i - 1] > key[i]) {
      const float k = key[i];
      const P2 p = pts[i];
      int j = i;
      while (j > 0 && key[j - 1] > k) {
        key[j] = key[j - 1];
        pts[j] = pts[j - 1];
        --j;
      }
      key[j] = k;
      pts[j] = p;
    }
  }
  float total = 0.f;
  for (int i = 0; i < n - 2; ++i)
    total += fabsf(((pts[0].x - pts[i + 2].x) * (pts[i + 1].y - pts[i + 2].y) - (pts[0].y - pts[i + 2].y) * (pts[i + 1].x - pts[i + 2].x)) / 2.0f);
  return total;
}

constexpr int TILE_Q = 64, TILE_B = 4;

__global__ __launch_bounds__(TILE_Q* TILE_B) void rotate_iou_eval_kernel(const float* boxes, const float* qboxes, float* out, int N, int K,
                                                                         int criterion) {
  __shared__ float qs[TILE_Q][5];
  const int tx = threadIdx.x & (TILE_Q - 1), ty = threadIdx.x / TILE_Q;
  const int j = blockIdx.x * TILE_Q + tx, i = blockIdx.y * TILE_B + ty;
  if (ty == 0 && j < K) {
#pragma unroll
    for (int f = 0; f < 5; ++f) qs[tx][f] = qboxes[(long)j * 5 + f];
  }
  __syncthreads();
  if (i >= N || j >= K) return;
  float b[5];
#pragma unroll
  for (int f = 0; f < 5; ++f) b[f] = boxes[(long)i * 5 + f];
  const float a1 = qs[tx][2] * qs[tx][3], a2 = b[2] * b[3];  // rbox1 = query, rbox2 = box (rotate_iou.py:289)
  const float it = rbox_intersection(qs[tx], b);
  float v = it;
  if (criterion == -1) v = it / (a1 + a2 - it);
  else if (criterion == 0) v = it / a1;
  else if (criterion == 1) v = it / a2;
  out[(long)i * K + j] = v;
}

__global__ void d3_overlap_kernel(const float* boxes, const float* qboxes, float* rinc, int N, int K, int criterion, int camera) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * K) return;
  const int i = (int)(idx / K), j = (int)(idx - (long)i * K);
  const float r = rinc[idx];
  if (!(r > 0.f)) return;
  const float* b = boxes + (long)i * 7;
  const float* q = qboxes + (long)j * 7;
  const float iw = camera ? fminf(b[1], q[1]) - fmaxf(b[1] - b[4], q[1] - q[4]) : fminf(b[2] + b[5], q[2] + q[5]) - fmaxf(b[2], q[2]);
  float o = 0.f;
  if (iw > 0.f) {
    const float v1 = b[3] * b[4] * b[5], v2 = q[3] * q[4] * q[5];
    const float inc = iw * r;
    const float ua = criterion == -1 ? (v1 + v2 - inc) : criterion == 0 ? v1 : criterion == 1 ? v2 : inc;
    o = inc / ua;
  }
  rinc[idx] = o;
}

__global__ void image_overlap_kernel(const float* boxes, const float* qboxes, float* out, int N, int K, int criterion) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * K) return;
  const int n = (int)(idx / K), k = (int)(idx - (long)n * K);
  const float* b = boxes + (long)n * 4;
  const float* q = qboxes + (long)k * 4;
  const float qa = (q[2] - q[0]) * (q[3] - q[1]);
  const float iw = fminf(b[2], q[2]) - fmaxf(b[0], q[0]);
  const float ih = fminf(b[3], q[3]) - fmaxf(b[1], q[1]);
  float o = 0.f;
  if (iw > 0.f && ih > 0.f) {
    const float ba = (b[2] - b[0]) * (b[3] - b[1]);
    const float ua = criterion == -1 ? (ba + qa - iw * ih) : criterion == 0 ? ba : criterion == 1 ? qa : 1.0f;
    o = iw * ih / ua;
  }
  out[idx] = o;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Result formatting (kitti_3d_evaluator.py:205-264 convert_3d_box_to_kitti, nuscenes_evaluator.py:196-198 velocity): one thread per
// detection, float64 like the reference's numpy / pyquaternion arithmetic.  The reference converts box by box on the host with one
// device->host copy per field per box; here a whole batch is one launch and one copy.
// out[i] = (W, L, H, x, y, z, rot_y, alpha, vx, vy).
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void quat_unit(double& w, double& x, double& y, double& z) {
  // pyquaternion _normalise(): only when |1 - sum of squares| >= 1e-14
  const double ss = w * w + x * x + y * y + z * z;
  if (!(fabs(1.0 - ss) < 1e-14) && ss > 0.0) {
    const double n = sqrt(ss);
    w /= n; x /= n; y /= n; z /= n;
  }
}

__global__ void __launch_bounds__(256) format_boxes_kernel(const float* __restrict__ box3d, const float* __restrict__ quat_global,
                                                           const float* __restrict__ speed, double* __restrict__ out, int n, double inv_w,
                                                           double inv_x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* b = box3d + (long)i * 10;
  const double PI = 3.141592653589793;
  double w = b[0], x = b[1], y = b[2], z = b[3];
  const float tx = b[4], tz = b[6], sw = b[7], sl = b[8], sh = b[9];
  const float ty = b[5] + sh / 2.0f;  // tvec += [0, H/2, 0] stays float32 (in-place add on a float32 array)
  // inversion * quat with inversion = (inv_w, inv_x, 0, 0): rows of pyquaternion's _q_matrix() times q
  double rw = inv_w * w - inv_x * x, rx = inv_x * w + inv_w * x, ry = inv_w * y - inv_x * z, rz = inv_x * y + inv_w * z;
  quat_unit(rw, rx, ry, rz);
  const double nv = sqrt(rx * rx + ry * ry + rz * rz);
  const bool axis_z_pos = nv < 1e-17 ? false : (rz / nv > 0.0);
  double ang = 2.0 * atan2(nv, rw) + PI;  // _wrap_angle: ((theta + pi) % 2pi) - pi, -pi -> pi
  ang = fmod(ang, 2.0 * PI) - PI;  // the operand is positive, where Python's % is fmod
  if (ang == -PI) ang = PI;
  const double rot_y = axis_z_pos ? -ang : ang;
  const double theta = atan2(fabs((double)tx), fabs((double)tz));
  double alpha = tx < 0.f ? rot_y + theta : rot_y - theta;
  if (alpha > PI) alpha -= 2.0 * PI;
  else if (alpha < -PI) alpha += 2.0 * PI;
  alpha = rint(alpha * 100.0) / 100.0;  // np.around(alpha, 2)
  double* o = out + (long)i * 10;
  o[0] = sw; o[1] = sl; o[2] = sh; o[3] = tx; o[4] = ty; o[5] = tz; o[6] = rot_y; o[7] = alpha;
  double vx = 0.0, vy = 0.0;
  if (quat_global != nullptr && speed != nullptr) {
    // speed * Quaternion(q).rotation_matrix.T[0]: first column of (Q Qbar^T)[1:, 1:]
    double gw = quat_global[i * 4 + 0], gx = quat_global[i * 4 + 1], gy = quat_global[i * 4 + 2], gz = quat_global[i * 4 + 3];
    quat_unit(gw, gx, gy, gz);
    const double s = speed[i];
    vx = s * (gx * gx + gw * gw - gz * gz - gy * gy);
    vy = s * (gy * gx + gz * gw + gw * gz + gx * gy);
  }
  o[8] = vx; o[9] = vy;
}

}  // namespace dd3d

extern "C" int dd3d_rotate_iou_eval(const float* boxes, const float* qboxes, float* out, int32_t N, int32_t K, int32_t criterion, void* stream) {
  using namespace dd3d;
  if (N == 0 || K == 0) return DD3D_OK;
  DD3D_REQUIRE(boxes && qboxes && out && N > 0 && K > 0, "dd3d_rotate_iou_eval: null pointer or negative size");
  hipLaunchKernelGGL(rotate_iou_eval_kernel, dim3(ceil_div(K, TILE_Q), ceil_div(N, TILE_B)), dim3(TILE_Q * TILE_B), 0,
                     reinterpret_cast<hipStream_t>(stream), boxes, qboxes, out, N, K, criterion);
  return check_launch("rotate_iou_eval_kernel");
}

extern "C" int dd3d_d3_box_overlap(const float* boxes, const float* qboxes, float* rinc, int32_t N, int32_t K, int32_t criterion,
                                   int32_t camera_coordinate, void* stream) {
  using namespace dd3d;
  if (N == 0 || K == 0) return DD3D_OK;
  DD3D_REQUIRE(boxes && qboxes && rinc && N > 0 && K > 0, "dd3d_d3_box_overlap: null pointer or negative size");
  const long tot = (long)N * K;
  hipLaunchKernelGGL(d3_overlap_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), boxes, qboxes, rinc,
                     N, K, criterion, camera_coordinate);
  return check_launch("d3_overlap_kernel");
}

extern "C" int dd3d_image_box_overlap(const float* boxes, const float* qboxes, float* out, int32_t N, int32_t K, int32_t criterion, void* stream) {
  using namespace dd3d;
  if (N == 0 || K == 0) return DD3D_OK;
  DD3D_REQUIRE(boxes && qboxes && out && N > 0 && K > 0, "dd3d_image_box_overlap: null pointer or negative size");
  const long tot = (long)N * K;
  hipLaunchKernelGGL(image_overlap_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), boxes, qboxes, out,
                     N, K, criterion);
  return check_launch("image_overlap_kernel");
}

extern "C" int dd3d_format_boxes3d(const float* box3d, const float* quat_global, const float* speed, double* out, int32_t n, void* stream) {
  using namespace dd3d;
  if (n == 0) return DD3D_OK;
  DD3D_REQUIRE(box3d && out && n > 0, "dd3d_format_boxes3d: null pointer or negative size");
  DD3D_REQUIRE((quat_global == nullptr) == (speed == nullptr), "dd3d_format_boxes3d: quat_global and speed go together");
  // Quaternion(axis=[1,0,0], radians=pi/2).inverse = conj(cos(pi/4), sin(pi/4), 0, 0) / sum of squares
  const double c = cos(M_PI / 4.0), sn = sin(M_PI / 4.0), ss = c * c + sn * sn;
  hipLaunchKernelGGL(format_boxes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), box3d, quat_global,
                     speed, out, n, c / ss, -sn / ss);
  return check_launch("format_boxes_kernel");
}
