// Shared device helpers for libtripled_hip (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tripled_hip.h"

// Block size of the element-wise / gather kernels (4 waves).  The streaming photometric and reconstruction
// kernels size their own blocks (one wave per column strip, see td_photo_fwd.hip).
#define TD_THREADS 256

#define TD_SSIM_C1 ((float)(0.01 * 0.01))
#define TD_SSIM_C2 ((float)(0.03 * 0.03))
#define TD_L1_EPS2 ((float)(1e-3 * 1e-3))


namespace td {

int record_launch_error(hipError_t e, const char* what);

// the partial rows a consumer prologue sums: rows 0, stride, 2 stride, ... (n of them)
struct BnRows { int n, stride; };

__device__ __forceinline__ int reflect1(int i, int n) {
  // ReflectionPad2d(1) index map (-1 -> 1, n -> n-2), then clamped so that positions
  // outside the padded domain (never consumed) still address valid memory.
  i = i < 0 ? -i : i;
  i = i >= n ? 2 * n - 2 - i : i;
  i = i < 0 ? 0 : i;
  return i >= n ? n - 1 : i;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Sum over a block of NW waves; result valid in thread 0. `scratch` >= NW floats of LDS.
// Fixed association order -> bit-reproducible.
template <int NW = 4>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NW; i += 2) r += scratch[i] + scratch[i + 1];
  }
  __syncthreads();
  return r;
}

// Result of projecting one target pixel into one source frame.
struct Tap {
  int x0, y0, x1, y1;          // clamped tap coordinates
  float nw, ne, sw, se;        // ATen bilinear weights
  bool in_e, in_s;             // east / south taps inside the image (ATen within_bounds)
  float ix, iy;                // clipped sample coordinate
  float gmx, gmy;              // clip-gradient multipliers (0 where the coordinate was clipped)
};

// Bilinear up-sampling of the low-res disparity at full-res pixel (qx,qy), exactly as
// ATen upsample_bilinear2d with align_corners=False (source index clamped below at 0).
struct UpIdx {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ UpIdx up_index(int q, float ratio, int n_in) {
  float s = ratio * ((float)q + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  int i0 = (int)s;
  i0 = i0 > n_in - 1 ? n_in - 1 : i0;
  UpIdx r;
  r.i0 = i0;
  r.i1 = i0 + ((i0 < n_in - 1) ? 1 : 0);
  r.l1 = s - (float)i0;
  r.l0 = 1.f - r.l1;
  return r;
}

__device__ __forceinline__ float upsample_disp(const float* __restrict__ d, int hs, int ws,
                                               float ry, float rx, int qy, int qx) {
  const UpIdx vy = up_index(qy, ry, hs);
  const UpIdx vx = up_index(qx, rx, ws);
  const int o0 = vy.i0 * ws, o1 = vy.i1 * ws;
  return vy.l0 * (vx.l0 * d[o0 + vx.i0] + vx.l1 * d[o0 + vx.i1]) +
         vy.l1 * (vx.l0 * d[o1 + vx.i0] + vx.l1 * d[o1 + vx.i1]);
}

// Camera ray (inv_K . (x,y,1)) scaled by depth -> camera point -> source pixel -> bilinear taps.
// Mirrors the reference's fp32 operation sequence (Backproject/Project in layers.py,
// grid_sampler in ATen) step by step.
__device__ __forceinline__ Tap project_ray(float r0, float r1, float r2, const float* P, float depth,
                                           int W, int H, float* pt /*3: camera point*/,
                                           float* cz /*c0,c1,z*/) {
  const float X = depth * r0, Y = depth * r1, Z = depth * r2;
  pt[0] = X; pt[1] = Y; pt[2] = Z;
  const float c0 = P[0] * X + P[1] * Y + P[2] * Z + P[3];
  const float c1 = P[4] * X + P[5] * Y + P[6] * Z + P[7];
  const float c2 = P[8] * X + P[9] * Y + P[10] * Z + P[11];
  const float z = c2 + 1e-7f;
  cz[0] = c0; cz[1] = c1; cz[2] = z;
  // one correctly-rounded reciprocal per (pixel, source); u = c0 * (1/z) differs from the
  // reference's c0 / z by <= 1 ulp (6e-8 relative: 4e-5 px at x = 640)
  const float iz = 1.f / z;
  const float u = c0 * iz, v = c1 * iz;
  const float gx = (u * (1.f / (float)(W - 1)) - 0.5f) * 2.f;   // reciprocal hoisted by the compiler
  const float gy = (v * (1.f / (float)(H - 1)) - 0.5f) * 2.f;   // (<= 1 ulp vs the reference's division)
  float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
  float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
  Tap t;
  // clip_coordinates_set_grad: gradient is zero at and beyond the borders
  t.gmx = (ix > 0.f && ix < (float)(W - 1)) ? 1.f : 0.f;
  t.gmy = (iy > 0.f && iy < (float)(H - 1)) ? 1.f : 0.f;
  ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  int x0 = (int)fx0, y0 = (int)fy0;
  x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0);
  y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0);
  t.x0 = x0; t.y0 = y0;
  t.in_e = x0 + 1 <= W - 1;
  t.in_s = y0 + 1 <= H - 1;
  t.x1 = t.in_e ? x0 + 1 : x0;
  t.y1 = t.in_s ? y0 + 1 : y0;
  const float ex = fx0 + 1.f - ix, wx = ix - fx0;   // (ix_se - ix), (ix - ix_nw)
  const float ey = fy0 + 1.f - iy, wy = iy - fy0;
  t.nw = ex * ey; t.ne = wx * ey; t.sw = ex * wy; t.se = wx * wy;
  t.ix = ix; t.iy = iy;
  return t;
}

__device__ __forceinline__ Tap project_tap(const float* ik, const float* P, float depth,
                                           int qx, int qy, int W, int H, float* pt, float* cz) {
  const float fx = (float)qx, fy = (float)qy;
  const float r0 = ik[0] * fx + ik[1] * fy + ik[2];
  const float r1 = ik[3] * fx + ik[4] * fy + ik[5];
  const float r2 = ik[6] * fx + ik[7] * fy + ik[8];
  return project_ray(r0, r1, r2, P, depth, W, H, pt, cz);
}

// The four taps of one channel plane.  All four loads are issued unconditionally (the tap
// coordinates are clamped, so the addresses are always valid) -- predicated loads make the
// compiler serialise every gather behind its own s_waitcnt.
struct TapVals {
  float nw, ne, sw, se;
};

// dword load at a 32-bit BYTE offset from a wave-uniform base pointer: the saddr form of global_load (SGPR base + one
// VGPR offset).  An element index makes the compiler widen to a 64-bit per-lane address (index * 4 may not fit 32 bits
// as far as it can tell): a shift and a 64-bit add per load, ~10 % of the photometric kernels' vector instructions.
// Callers address one image plane (< 4 GiB) per base pointer.
__device__ __forceinline__ float ld_at(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

__device__ __forceinline__ TapVals load_taps(const float* __restrict__ plane, int W, const Tap& t) {
  const unsigned o0 = (unsigned)(t.y0 * W), o1 = (unsigned)(t.y1 * W);
  TapVals v;
  v.nw = ld_at(plane, (o0 + (unsigned)t.x0) * 4u);
  v.ne = ld_at(plane, (o0 + (unsigned)t.x1) * 4u);
  v.sw = ld_at(plane, (o1 + (unsigned)t.x0) * 4u);
  v.se = ld_at(plane, (o1 + (unsigned)t.x1) * 4u);
  return v;
}

// ---- packed frames -------------------------------------------------------------------------------------------------
// The photometric FORWARD kernels read the target and source frames as RGBX pixels (float4 per pixel, [B, H, W, 4], x = 0; written
// once per step by td_pack_rgbx): ONE 16-byte load per pixel or bilinear tap instead of three dword loads from the three NCHW planes.
// The forward keeps 46 loads per row in flight two rows deep against a 6-bit vmcnt (63): it is bound by the NUMBER of vector-memory
// instructions, not by bytes (gathering one channel instead of three: 61 -> 45 us; RGBX: 61 -> 49 us at scale 0, 217 -> 191 us over
// the four scales).  Measured and NOT adopted: the same format in the backward (20 instead of 31 loads per row but 126 instead of 106
// cache lines touched: 202.7 -> 209 us over the four scales; it gains 3 us where the disparity has detail and the dword gathers lose
// their coalescing, and loses 3 us where the warp is coherent) and the coefficient field as pixels -- [B,H,W,12] (three 16-byte
// accesses per pixel at a 48-byte stride touch three times the lines of nine coalesced planes: forward 52.0 vs 49.2 us) or
// [B,3,H,W,4] (contiguous 16-byte accesses: forward unchanged, backward 59.6 vs 57.9 us at scale 0, 50.0 vs 46.6 at scale 3).
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld4_at(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(base) + byte_off);
}

struct TapVals4 {
  f4 nw, ne, sw, se;
};

// the four taps of one RGBX image (pixel = 16 bytes)
__device__ __forceinline__ TapVals4 load_taps4(const float* __restrict__ img, int W, const Tap& t) {
  const unsigned o0 = (unsigned)(t.y0 * W), o1 = (unsigned)(t.y1 * W);
  TapVals4 v;
  v.nw = ld4_at(img, (o0 + (unsigned)t.x0) * 16u);
  v.ne = ld4_at(img, (o0 + (unsigned)t.x1) * 16u);
  v.sw = ld4_at(img, (o1 + (unsigned)t.x0) * 16u);
  v.se = ld4_at(img, (o1 + (unsigned)t.x1) * 16u);
  return v;
}

__device__ __forceinline__ TapVals tap_channel(const TapVals4& v, int c) {      // c: compile-time constant after unrolling
  TapVals o;
  o.nw = v.nw[c]; o.ne = v.ne[c]; o.sw = v.sw[c]; o.se = v.se[c];
  return o;
}

// ATen accumulation order nw, ne, sw, se.  A tap outside the image has a clamped address and
// an exactly-zero weight (the clipped coordinate sits on the last row/column), so adding
// value*0 equals ATen's skipping of that tap for finite images.
__device__ __forceinline__ float blend_taps(const TapVals& v, const Tap& t) {
  float o = v.nw * t.nw;
  o += v.ne * t.ne;
  o += v.sw * t.sw;
  o += v.se * t.se;
  return o;
}

// 1/x and sqrt(x) to ~1 ulp (v_rcp_f32 / v_sqrt_f32) for quantities whose conditioning does not
// need correctly rounded results (SSIM ratio, channel means, robust-L1); sampling coordinates
// keep IEEE division.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// SSIM loss value for one channel from 3x3 window sums (sum of x, y, x^2, y^2, xy).
__device__ __forceinline__ float ssim_from_sums(float sx, float sy, float sxx, float syy, float sxy) {
  const float k = 1.f / 9.f;
  const float mx = sx * k, my = sy * k;
  const float vx = sxx * k - mx * mx;
  const float vy = syy * k - my * my;
  const float cxy = sxy * k - mx * my;
  const float n = (2.f * mx * my + TD_SSIM_C1) * (2.f * cxy + TD_SSIM_C2);
  const float d = (mx * mx + my * my + TD_SSIM_C1) * (vx + vy + TD_SSIM_C2);
  const float s = (1.f - n * fast_rcp(d)) * 0.5f;
  return fminf(fmaxf(s, 0.f), 1.f);
}

// SSIM loss value plus the coefficients of its adjoint w.r.t. the prediction x: for every member q
// of the 3x3 window,  d SSIM_loss / d x_q = (alpha + beta * x_q + gamma * y_q) / 9
// (zero outside the clamp's open interval, like torch.clamp's backward on the closed one).
__device__ __forceinline__ float ssim_with_adjoint(float sx, float sy, float sxx, float syy, float sxy,
                                                   float& alpha, float& beta, float& gamma) {
  const float k = 1.f / 9.f;
  const float mx = sx * k, my = sy * k;
  const float vx = sxx * k - mx * mx;
  const float vy = syy * k - my * my;
  const float cxy = sxy * k - mx * my;
  const float A1 = 2.f * mx * my + TD_SSIM_C1, A2 = 2.f * cxy + TD_SSIM_C2;
  const float B1 = mx * mx + my * my + TD_SSIM_C1, B2 = vx + vy + TD_SSIM_C2;
  const float n = A1 * A2, d = B1 * B2;
  const float invd = fast_rcp(d);
  const float q = n * invd;
  const float s = (1.f - q) * 0.5f;
  const float sc = (s >= 0.f && s <= 1.f) ? invd : 0.f;
  alpha = -sc * (my * (A2 - A1) - q * mx * (B2 - B1));
  beta = sc * q * B1;
  gamma = -sc * A1;
  return fminf(fmaxf(s, 0.f), 1.f);
}


// value of the lane to the left / right (wave-wide shift by one lane; the edge lanes receive 0 --
// they only ever feed halo columns whose results are discarded).  old = 0 + bound_ctrl lets the
// compiler fold the shift into the consuming v_add_f32 (v_add_f32_dpp).
__device__ __forceinline__ float lane_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */,
                                                               0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */,
                                                               0xf, 0xf, true));
}
__device__ __forceinline__ float hsum3(float v) { return (lane_left(v) + v) + lane_right(v); }

// ---------------------------------------------------------------------------------------------
// Rows per wave task of the streaming kernels.  A task costs (rows + halo) row iterations and the chip holds
// TD_WAVE_SLOTS such waves at once (MI355X: 256 CUs x 4 SIMDs x 2 waves at ~230 VGPRs), so the launch takes
// ceil(tasks / slots) "rounds" of that length: at B=12, 192x640 eight-row tasks need two rounds of 10 iterations,
// thirteen-row tasks one round of 15 (measured: 116 -> 92 us for the backward).  Pure function of the shape, so
// the *_num_blocks entry points and the launches agree.
constexpr int TD_WAVE_SLOTS = 256 * 4 * 2;
static inline int pick_rows(int units, int H, int halo, int step, int lo, int hi, int slots = TD_WAVE_SLOTS) {
  int best = lo;
  long long best_cost = -1;
  for (int r = lo; r <= hi; r += step) {
    const long long tasks = (long long)units * ((H + r - 1) / r);
    const long long rounds = (tasks + slots - 1) / slots;
    const long long cost = rounds * (r + halo);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; }   // ties keep the smaller tile
  }
  return best;
}

}  // namespace td
