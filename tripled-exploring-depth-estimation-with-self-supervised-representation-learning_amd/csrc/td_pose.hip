// Pose vector -> camera transform -> projection matrix, one launch for all frame pairs of a step.
// Replaces transformation_from_parameters / rot_from_axisangle / get_translation_matrix
// (reference: mono/model/mono_fm_joint/net.py:225-277; ~30 tiny ATen kernels + 2 host-side zeros().cuda()
// per frame pair) and the per-scale torch.matmul(K, T)[:, :3, :] of Project.forward
// (mono/model/mono_fm_joint/layers.py:73-75), forward and adjoint.
//   axis = v / (|v| + 1e-7);  R = Rodrigues(axis, |v|)  (the reference's product order: x*(x*C)+ca, x*(y*C)-z*sa, ...)
//   invert (frame id < 0):  M = R^T @ Trans(-t)   else   M = Trans(t) @ R
//   P = (K @ M)[:3, :]
// One thread per (pair, sample).  The adjoint re-evaluates the forward on dual numbers carrying the six partial
// derivatives (d/dv, d/dt), so forward and backward cannot drift apart.
#include "td_common.h"

namespace td {

struct D6 {
  float v;
  float d[6];
};

__device__ __forceinline__ D6 cst(float v) {
  D6 r; r.v = v;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.d[i] = 0.f;
  return r;
}
__device__ __forceinline__ D6 var(float v, int k) { D6 r = cst(v); r.d[k] = 1.f; return r; }
__device__ __forceinline__ D6 operator+(const D6& a, const D6& b) {
  D6 r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ __forceinline__ D6 operator-(const D6& a, const D6& b) {
  D6 r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ __forceinline__ D6 operator-(const D6& a) {
  D6 r; r.v = -a.v;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.d[i] = -a.d[i];
  return r;
}
__device__ __forceinline__ D6 operator*(const D6& a, const D6& b) {
  D6 r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ __forceinline__ D6 operator/(const D6& a, const D6& b) {
  D6 r; r.v = a.v / b.v;
  const float ib = 1.f / b.v;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
__device__ __forceinline__ D6 dsqrt(const D6& a) {
  D6 r; r.v = sqrtf(a.v);
  const float k = a.v > 0.f ? 0.5f / r.v : 0.f;      // torch.norm's gradient at the origin is 0
#pragma unroll
  for (int i = 0; i < 6; ++i) r.d[i] = k * a.d[i];
  return r;
}
__device__ __forceinline__ D6 dsin(const D6& a) {
  D6 r; r.v = sinf(a.v);
  const float c = cosf(a.v);
#pragma unroll
  for (int i = 0; i < 6; ++i) r.d[i] = c * a.d[i];
  return r;
}
__device__ __forceinline__ D6 dcos(const D6& a) {
  D6 r; r.v = cosf(a.v);
  const float s = -sinf(a.v);
#pragma unroll
  for (int i = 0; i < 6; ++i) r.d[i] = s * a.d[i];
  return r;
}

// plain-float instantiation of the same expressions (forward)
struct F1 { float v; };
__device__ __forceinline__ F1 operator+(F1 a, F1 b) { return {a.v + b.v}; }
__device__ __forceinline__ F1 operator-(F1 a, F1 b) { return {a.v - b.v}; }
__device__ __forceinline__ F1 operator-(F1 a) { return {-a.v}; }
__device__ __forceinline__ F1 operator*(F1 a, F1 b) { return {a.v * b.v}; }
__device__ __forceinline__ F1 operator/(F1 a, F1 b) { return {a.v / b.v}; }
__device__ __forceinline__ F1 dsqrt(F1 a) { return {sqrtf(a.v)}; }
__device__ __forceinline__ F1 dsin(F1 a) { return {sinf(a.v)}; }
__device__ __forceinline__ F1 dcos(F1 a) { return {cosf(a.v)}; }
__device__ __forceinline__ F1 make_c(F1*, float v) { return {v}; }
__device__ __forceinline__ D6 make_c(D6*, float v) { return cst(v); }

// M (4x4, row-major) from the six pose parameters; S = F1 (values) or D6 (values + partials)
template <typename S>
__device__ __forceinline__ void pose_matrix(const S vx, const S vy, const S vz, const S tx, const S ty, const S tz, bool invert,
                                            S* M) {
  const S zero = make_c((S*)nullptr, 0.f), one = make_c((S*)nullptr, 1.f);
  const S angle = dsqrt(vx * vx + vy * vy + vz * vz);
  const S den = angle + make_c((S*)nullptr, 1e-7f);
  const S x = vx / den, y = vy / den, z = vz / den;
  const S ca = dcos(angle), sa = dsin(angle);
  const S C = one - ca;
  const S xs = x * sa, ys = y * sa, zs = z * sa;
  const S xC = x * C, yC = y * C, zC = z * C;
  const S xyC = x * yC, yzC = y * zC, zxC = z * xC;
  S R[3][3];
  R[0][0] = x * xC + ca; R[0][1] = xyC - zs;    R[0][2] = zxC + ys;
  R[1][0] = xyC + zs;    R[1][1] = y * yC + ca; R[1][2] = yzC - xs;
  R[2][0] = zxC - ys;    R[2][1] = yzC + xs;    R[2][2] = z * zC + ca;
  if (!invert) {                         // Trans(t) @ R = [R t; 0 1]
    const S t[3] = {tx, ty, tz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[i][j];
      M[i * 4 + 3] = t[i];
    }
  } else {                               // R^T @ Trans(-t) = [R^T, R^T (-t); 0 1]
    const S t[3] = {-tx, -ty, -tz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[j][i];
      M[i * 4 + 3] = (R[0][i] * t[0] + R[1][i] * t[1]) + R[2][i] * t[2];
    }
  }
  M[12] = zero; M[13] = zero; M[14] = zero; M[15] = one;
}

struct PoseArgs {
  const float* axisangle;     // [n*B, 3]
  const float* translation;   // [n*B, 3]
  const float* K;             // [B, 4, 4] or null
  const float* gT;            // [n, B, 4, 4] or null (backward)
  const float* gP;            // [n, B, 3, 4] or null (backward)
  float* T;                   // forward out [n, B, 4, 4]
  float* P;                   // forward out [n, B, 3, 4] or null
  float* g_axisangle;         // backward out [n*B, 3]
  float* g_translation;       // backward out [n*B, 3]
  int n, B;
  unsigned invert_mask;       // bit i: pair i is inverted
};

__global__ __launch_bounds__(64) void pose_fwd_kernel(const PoseArgs a) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n * a.B) return;
  const int pair = i / a.B, b = i - pair * a.B;
  const float* v = a.axisangle + (size_t)i * 3;
  const float* t = a.translation + (size_t)i * 3;
  F1 M[16];
  pose_matrix<F1>({v[0]}, {v[1]}, {v[2]}, {t[0]}, {t[1]}, {t[2]}, (a.invert_mask >> pair) & 1u, M);
  float* To = a.T + (size_t)i * 16;
#pragma unroll
  for (int k = 0; k < 16; ++k) To[k] = M[k].v;
  if (a.P) {
    const float* K = a.K + (size_t)b * 16;
    float* Po = a.P + (size_t)i * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        Po[r * 4 + c] = ((K[r * 4 + 0] * M[c].v + K[r * 4 + 1] * M[4 + c].v) + K[r * 4 + 2] * M[8 + c].v) + K[r * 4 + 3] * M[12 + c].v;
  }
}

__global__ __launch_bounds__(64) void pose_bwd_kernel(const PoseArgs a) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n * a.B) return;
  const int pair = i / a.B, b = i - pair * a.B;
  const float* v = a.axisangle + (size_t)i * 3;
  const float* t = a.translation + (size_t)i * 3;
  // upstream gradient of M: gT plus K[:3,:]^T @ gP
  float g[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) g[k] = a.gT ? a.gT[(size_t)i * 16 + k] : 0.f;
  if (a.gP) {
    const float* K = a.K + (size_t)b * 16;
    const float* gp = a.gP + (size_t)i * 12;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        g[m * 4 + c] += (K[0 * 4 + m] * gp[0 * 4 + c] + K[1 * 4 + m] * gp[1 * 4 + c]) + K[2 * 4 + m] * gp[2 * 4 + c];
  }
  D6 M[16];
  pose_matrix<D6>(var(v[0], 0), var(v[1], 1), var(v[2], 2), var(t[0], 3), var(t[1], 4), var(t[2], 5),
                  (a.invert_mask >> pair) & 1u, M);
  float out[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 12; ++k)             // the last row of M is constant
#pragma unroll
    for (int j = 0; j < 6; ++j) out[j] += g[k] * M[k].d[j];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    a.g_axisangle[(size_t)i * 3 + j] = out[j];
    a.g_translation[(size_t)i * 3 + j] = out[3 + j];
  }
}

}  // namespace td

static int pose_args(td::PoseArgs& a, const float* axisangle, const float* translation, const int* invert, const float* K, int n,
                     int B) {
  if (!axisangle || !translation || !invert || n <= 0 || n > 32 || B <= 0) return TD_ERR_BAD_ARG;
  a.axisangle = axisangle; a.translation = translation; a.K = K; a.n = n; a.B = B;
  a.invert_mask = 0;
  for (int i = 0; i < n; ++i) a.invert_mask |= (invert[i] ? 1u : 0u) << i;
  a.gT = a.gP = nullptr; a.T = a.P = a.g_axisangle = a.g_translation = nullptr;
  return TD_OK;
}

extern "C" int td_pose_fwd(const float* axisangle, const float* translation, const int* invert, const float* K, int n_pairs, int B,
                           float* T, float* P, td_stream_t stream) {
  td::PoseArgs a;
  const int rc = pose_args(a, axisangle, translation, invert, K, n_pairs, B);
  if (rc != TD_OK) return rc;
  if (!T || (P && !K)) return TD_ERR_BAD_ARG;
  a.T = T; a.P = P;
  hipLaunchKernelGGL(td::pose_fwd_kernel, dim3((n_pairs * B + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
  return td::record_launch_error(hipGetLastError(), "td_pose_fwd");
}

extern "C" int td_pose_bwd(const float* axisangle, const float* translation, const int* invert, const float* K, int n_pairs, int B,
                           const float* gT, const float* gP, float* g_axisangle, float* g_translation, td_stream_t stream) {
  td::PoseArgs a;
  const int rc = pose_args(a, axisangle, translation, invert, K, n_pairs, B);
  if (rc != TD_OK) return rc;
  if (!g_axisangle || !g_translation || (!gT && !gP) || (gP && !K)) return TD_ERR_BAD_ARG;
  a.gT = gT; a.gP = gP; a.g_axisangle = g_axisangle; a.g_translation = g_translation;
  hipLaunchKernelGGL(td::pose_bwd_kernel, dim3((n_pairs * B + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
  return td::record_launch_error(hipGetLastError(), "td_pose_bwd");
}
