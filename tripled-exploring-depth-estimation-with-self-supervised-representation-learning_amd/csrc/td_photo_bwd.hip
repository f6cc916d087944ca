// Fused photometric backward (recompute-in-kernel).  Per 8x64 tile and per source frame:
//   1. re-warp the source over the tile + 2-pixel halo into LDS,
//   2. for every pixel p of tile + 1-pixel halo whose arg-min selected this frame, form the
//      SSIM window statistics and store the three coefficients (alpha, beta, gamma) per
//      channel such that d SSIM_p / d x_q = alpha_p + beta_p * x_q + gamma_p * y_q for any
//      window member q  (a box-filter adjoint instead of a 9x9 scatter),
//   3. every tile pixel q box-sums the coefficients of its neighbours (with the reflection
//      padding's multiplicities), adds the robust-L1 term, and chains through the bilinear
//      sampler, the projection and the depth to d(loss)/d(upsampled disparity) and dL/dP.
// dL/dP is block-reduced and written per block; d_up is written once per pixel (no atomics,
// bit-reproducible).
#include "td_common.h"

namespace td {

constexpr int BT_H = TD_BWD_TILE_H;
constexpr int BT_W = TD_TILE_W;
constexpr int B2H = BT_H + 4, B2W = BT_W + 4;   // tile + halo 2 (values)
constexpr int B1H = BT_H + 2, B1W = BT_W + 2;   // tile + halo 1 (coefficients)

template <int NS>
struct PhotoBwdArgs {
  const float* tgt;
  const float* src[NS];
  const float* disp;
  const float* P;
  const float* invK;
  const uint8_t* argmin;
  const float* gscale;
  float* d_up;
  float* dP_partial;
  int B, H, W, hs, ws;
  int n_ident;            // candidates preceding the warped ones (n_src when automasking, else 0)
  float inv_count;
  float min_disp, disp_range;
};

template <int NS>
__global__ __launch_bounds__(TD_THREADS) void photo_bwd_kernel(const PhotoBwdArgs<NS> a) {
  __shared__ float s_y[3][B2H][B2W];
  __shared__ float s_x[3][B2H][B2W];
  __shared__ float s_cf[9][B1H][B1W];     // [channel*3 + {alpha,beta,gamma}]
  __shared__ float s_cam[9 + NS * 12];
  __shared__ float s_red[4][12];

  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int ty0 = blockIdx.y * BT_H, tx0 = blockIdx.x * BT_W;
  const int H = a.H, W = a.W;
  const size_t plane = (size_t)H * W;

  if (tid < 9) s_cam[tid] = a.invK[(size_t)b * 16 + (tid / 3) * 4 + (tid % 3)];
  if (tid >= 64 && tid < 64 + NS * 12) {
    const int k = tid - 64, f = k / 12, e = k % 12;
    s_cam[9 + k] = a.P[((size_t)f * a.B + b) * 12 + e];
  }
  const float ry = (float)a.hs / (float)H, rx = (float)a.ws / (float)W;
  const float* dispb = a.disp + (size_t)b * a.hs * a.ws;
  const float g = a.gscale[0] * a.inv_count;          // d total / d (per-pixel min)
  const float g_ssim = g * 0.85f / 3.f / 9.f;          // .. / d (window-mean member), per channel
  const float g_l1 = g * 0.15f / 3.f;

  // target tile (halo 2) once
  for (int pos = tid; pos < B2H * B2W; pos += TD_THREADS) {
    const int py = pos / B2W, px = pos - py * B2W;
    const int qy = reflect1(ty0 + py - 2, H), qx = reflect1(tx0 + px - 2, W);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      s_y[c][py][px] = a.tgt[((size_t)b * 3 + c) * plane + (size_t)qy * W + qx];
  }

  // the two tile pixels this thread owns: rows (tid>>6)*2 + {0,1}, column tid&63
  const int ox = tid & 63, oyb = (tid >> 6) * 2;
  float dup[2] = {0.f, 0.f};

#pragma unroll 1
  for (int f = 0; f < NS; ++f) {
    __syncthreads();   // previous frame's s_x / s_cf fully consumed; s_cam / s_y visible
    const float* Pf = s_cam + 9 + f * 12;
    const float* srcb = a.src[f] + (size_t)b * 3 * plane;
    const int sel = a.n_ident + f;

    // ---- 1. re-warp over tile + halo 2 ----
    for (int pos = tid; pos < B2H * B2W; pos += TD_THREADS) {
      const int py = pos / B2W, px = pos - py * B2W;
      const int qy = reflect1(ty0 + py - 2, H), qx = reflect1(tx0 + px - 2, W);
      const float d = upsample_disp(dispb, a.hs, a.ws, ry, rx, qy, qx);
      const float depth = fast_rcp(a.min_disp + a.disp_range * d);
      float pt[3], cz[3];
      const Tap t = project_tap(s_cam, Pf, depth, qx, qy, W, H, pt, cz);
#pragma unroll
      for (int c = 0; c < 3; ++c) s_x[c][py][px] = sample_tap(srcb + (size_t)c * plane, W, t);
    }
    __syncthreads();

    // ---- 2. SSIM adjoint coefficients over tile + halo 1 ----
    for (int pos = tid; pos < B1H * B1W; pos += TD_THREADS) {
      const int py = pos / B1W, px = pos - py * B1W;
      const int gy = ty0 + py - 1, gx = tx0 + px - 1;
      bool on = gy >= 0 && gy < H && gx >= 0 && gx < W;
      if (on) on = a.argmin[(size_t)b * plane + (size_t)gy * W + gx] == sel;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float al = 0.f, be = 0.f, ga = 0.f;
        if (on) {
          float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const float x = s_x[c][py + dy][px + dx], y = s_y[c][py + dy][px + dx];
              sx += x; sy += y; sxx += x * x; syy += y * y; sxy += x * y;
            }
          const float k = 1.f / 9.f;
          const float mx = sx * k, my = sy * k;
          const float vx = sxx * k - mx * mx, vy = syy * k - my * my, cxy = sxy * k - mx * my;
          const float A1 = 2.f * mx * my + TD_SSIM_C1, A2 = 2.f * cxy + TD_SSIM_C2;
          const float B1 = mx * mx + my * my + TD_SSIM_C1, B2 = vx + vy + TD_SSIM_C2;
          const float n = A1 * A2, d = B1 * B2;
          const float invd = fast_rcp(d);
          const float s = (1.f - n * invd) * 0.5f;
          if (s >= 0.f && s <= 1.f) {     // clamp passes gradient on the closed interval
            const float q = n * invd;
            al = -invd * (my * (A2 - A1) - q * mx * (B2 - B1)) * g_ssim;
            be = invd * q * B1 * g_ssim;
            ga = -invd * A1 * g_ssim;
          }
        }
        s_cf[c * 3 + 0][py][px] = al;
        s_cf[c * 3 + 1][py][px] = be;
        s_cf[c * 3 + 2][py][px] = ga;
      }
    }
    __syncthreads();

    // ---- 3. gather at the owned pixels, chain to depth and P ----
    float dP[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) dP[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int oy = oyb + j;
      const int gy = ty0 + oy, gx = tx0 + ox;
      if (gy < H && gx < W) {
        const int my0 = 1 + (gy == 1 ? 1 : 0), my2 = 1 + (gy == H - 2 ? 1 : 0);
        const int mx0 = 1 + (gx == 1 ? 1 : 0), mx2 = 1 + (gx == W - 2 ? 1 : 0);
        const bool mine = a.argmin[(size_t)b * plane + (size_t)gy * W + gx] == sel;
        float gw[3];
        bool any = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float Ga = 0.f, Gb = 0.f, Gc = 0.f;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const float wy = (float)(dy == 0 ? my0 : (dy == 2 ? my2 : 1));
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const float w = wy * (float)(dx == 0 ? mx0 : (dx == 2 ? mx2 : 1));
              Ga += w * s_cf[c * 3 + 0][oy + dy][ox + dx];
              Gb += w * s_cf[c * 3 + 1][oy + dy][ox + dx];
              Gc += w * s_cf[c * 3 + 2][oy + dy][ox + dx];
            }
          }
          const float x = s_x[c][oy + 2][ox + 2], y = s_y[c][oy + 2][ox + 2];
          float v = Ga + Gb * x + Gc * y;
          if (mine) {
            const float df = x - y;
            v += g_l1 * df * fast_rcp(fast_sqrt(df * df + TD_L1_EPS2));
          }
          gw[c] = v;
          any = any || (v != 0.f);
        }
        if (any) {
          const float d = upsample_disp(dispb, a.hs, a.ws, ry, rx, gy, gx);
          const float depth = fast_rcp(a.min_disp + a.disp_range * d);
          float pt[3], cz[3];
          const Tap t = project_tap(s_cam, Pf, depth, gx, gy, W, H, pt, cz);
          float gix = 0.f, giy = 0.f;
          const float ex = (float)t.x0 + 1.f - t.ix, wx = t.ix - (float)t.x0;
          const float ey = (float)t.y0 + 1.f - t.iy, wy = t.iy - (float)t.y0;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float* r0 = srcb + (size_t)c * plane + (size_t)t.y0 * W;
            const float* r1 = srcb + (size_t)c * plane + (size_t)t.y1 * W;
            const float l_nw = r0[t.x0], l_ne = r0[t.x1], l_sw = r1[t.x0], l_se = r1[t.x1];
            const float vnw = l_nw;
            const float vne = t.in_e ? l_ne : 0.f;
            const float vsw = t.in_s ? l_sw : 0.f;
            const float vse = (t.in_e && t.in_s) ? l_se : 0.f;
            gix += gw[c] * (-vnw * ey + vne * ey - vsw * wy + vse * wy);
            giy += gw[c] * (-vnw * ex - vne * wx + vsw * ex + vse * wx);
          }
          // grid_sampler unnormalise (W/2) * clip multiplier, then (u/(W-1) - 0.5) * 2
          const float du = gix * t.gmx * ((float)W * 0.5f) * 2.f / (float)(W - 1);
          const float dv = giy * t.gmy * ((float)H * 0.5f) * 2.f / (float)(H - 1);
          const float iz = fast_rcp(cz[2]);
          const float dc0 = du * iz, dc1 = dv * iz;
          const float dc2 = -(du * cz[0] + dv * cz[1]) * iz * iz;
          dP[0] += dc0 * pt[0]; dP[1] += dc0 * pt[1]; dP[2] += dc0 * pt[2]; dP[3] += dc0;
          dP[4] += dc1 * pt[0]; dP[5] += dc1 * pt[1]; dP[6] += dc1 * pt[2]; dP[7] += dc1;
          dP[8] += dc2 * pt[0]; dP[9] += dc2 * pt[1]; dP[10] += dc2 * pt[2]; dP[11] += dc2;
          // d/d depth: point = depth * ray, ray = point / depth
          const float dX = dc0 * Pf[0] + dc1 * Pf[4] + dc2 * Pf[8];
          const float dY = dc0 * Pf[1] + dc1 * Pf[5] + dc2 * Pf[9];
          const float dZ = dc0 * Pf[2] + dc1 * Pf[6] + dc2 * Pf[10];
          const float fx = (float)gx, fy = (float)gy;
          const float r0 = s_cam[0] * fx + s_cam[1] * fy + s_cam[2];
          const float r1 = s_cam[3] * fx + s_cam[4] * fy + s_cam[5];
          const float r2 = s_cam[6] * fx + s_cam[7] * fy + s_cam[8];
          const float dD = dX * r0 + dY * r1 + dZ * r2;
          dup[j] += dD * (-a.disp_range * depth * depth);
        }
      }
    }

    // block-reduce dP for this frame
#pragma unroll
    for (int k = 0; k < 12; ++k) dP[k] = wave_sum(dP[k]);
    const int lane = tid & 63, wid = tid >> 6;
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 12; ++k) s_red[wid][k] = dP[k];
    }
    __syncthreads();
    if (tid < 12) {
      const size_t blk = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      a.dP_partial[blk * (NS * 12) + f * 12 + tid] =
          (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
    }
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gy = ty0 + oyb + j, gx = tx0 + ox;
    if (gy < H && gx < W) a.d_up[(size_t)b * plane + (size_t)gy * W + gx] = dup[j];
  }
}

template <int NS>
static int run_bwd(const float* tgt, const float* const* src, const float* disp, const float* P,
                   const float* invK, const uint8_t* argmin, int automask, const float* gscale,
                   float inv_count, int B, int H, int W, int hs, int ws, float min_depth,
                   float max_depth, float* d_up, float* dP_partial, hipStream_t st) {
  PhotoBwdArgs<NS> a;
  a.tgt = tgt;
  for (int i = 0; i < NS; ++i) a.src[i] = src[i];
  a.disp = disp; a.P = P; a.invK = invK; a.argmin = argmin; a.gscale = gscale;
  a.d_up = d_up; a.dP_partial = dP_partial;
  a.B = B; a.H = H; a.W = W; a.hs = hs; a.ws = ws;
  a.n_ident = automask ? NS : 0;
  a.inv_count = inv_count;
  const double lo = 1.0 / (double)max_depth, hi = 1.0 / (double)min_depth;
  a.min_disp = (float)lo;
  a.disp_range = (float)(hi - lo);
  dim3 grid((W + BT_W - 1) / BT_W, (H + BT_H - 1) / BT_H, B);
  hipLaunchKernelGGL((photo_bwd_kernel<NS>), grid, dim3(TD_THREADS), 0, st, a);
  return record_launch_error(hipGetLastError(), "td_photo_bwd");
}

// ---------------------------------------------------------------------------
// dP reduction over the blocks of one sample: one wave per (sample, frame*12+k).
__global__ __launch_bounds__(64) void reduce_dP_kernel(const float* __restrict__ part, int n_src,
                                                       int B, int blocks_per_sample,
                                                       float* __restrict__ dP) {
  const int k = blockIdx.x;             // 0 .. n_src*12-1
  const int b = blockIdx.y;
  const int stride = n_src * 12;
  float acc = 0.f;
  for (int i = threadIdx.x; i < blocks_per_sample; i += 64)
    acc += part[((size_t)b * blocks_per_sample + i) * stride + k];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) {
    const int f = k / 12, e = k % 12;
    dP[((size_t)f * B + b) * 12 + e] = acc;
  }
}

// ---------------------------------------------------------------------------
// Adjoint of the bilinear up-sampling, gather form: G lanes cooperate on one low-res pixel.
template <int G>
__global__ __launch_bounds__(TD_THREADS) void upsample_adjoint_kernel(
    const float* __restrict__ d_up, int B, int H, int W, int hs, int ws,
    float* __restrict__ d_disp, int accumulate) {
  const int gid = (blockIdx.x * TD_THREADS + threadIdx.x) / G;
  const int sub = threadIdx.x % G;
  const int total = B * hs * ws;
  const bool live = gid < total;
  const int id = live ? gid : total - 1;
  const int i = id % ws, j = (id / ws) % hs, b = id / (ws * hs);
  const float ry = (float)hs / (float)H, rx = (float)ws / (float)W;
  const float fy = (float)H / (float)hs, fx = (float)W / (float)ws;
  int ylo = (int)floorf(((float)j - 0.5f) * fy - 0.5f) - 1;
  int yhi = (int)ceilf(((float)j + 1.5f) * fy - 0.5f) + 1;
  int xlo = (int)floorf(((float)i - 0.5f) * fx - 0.5f) - 1;
  int xhi = (int)ceilf(((float)i + 1.5f) * fx - 0.5f) + 1;
  ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
  yhi = yhi > H - 1 ? H - 1 : yhi; xhi = xhi > W - 1 ? W - 1 : xhi;
  const int nx = xhi - xlo + 1, ny = yhi - ylo + 1;
  const float* src = d_up + (size_t)b * H * W;
  float acc = 0.f;
  for (int e = sub; e < nx * ny; e += G) {
    const int yy = ylo + e / nx, xx = xlo + e % nx;
    const UpIdx vy = up_index(yy, ry, hs);
    const UpIdx vx = up_index(xx, rx, ws);
    const float wy = (vy.i0 == j ? vy.l0 : 0.f) + (vy.i1 == j ? vy.l1 : 0.f);
    const float wx = (vx.i0 == i ? vx.l0 : 0.f) + (vx.i1 == i ? vx.l1 : 0.f);
    const float w = wy * wx;
    if (w != 0.f) acc += w * src[(size_t)yy * W + xx];
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, G);
  if (live && sub == 0) {
    if (accumulate) d_disp[id] += acc; else d_disp[id] = acc;
  }
}

}  // namespace td

extern "C" int td_photo_bwd(const float* tgt, const float* const* src, int n_src, const float* disp,
                            const float* P, const float* invK, const uint8_t* argmin, int automask,
                            const float* gscale, float inv_count, int B, int H, int W, int hs, int ws,
                            float min_depth, float max_depth, float* d_up, float* dP_partial,
                            td_stream_t stream) {
  if (!tgt || !src || !disp || !P || !invK || !argmin || !gscale || !d_up || !dP_partial) return TD_ERR_BAD_ARG;
  if (n_src < 1 || n_src > TD_MAX_SRC || B <= 0 || hs <= 0 || ws <= 0 || hs > H || ws > W) return TD_ERR_BAD_ARG;
  if (!(min_depth > 0.f) || !(max_depth > min_depth)) return TD_ERR_BAD_ARG;
  for (int i = 0; i < n_src; ++i) if (!src[i]) return TD_ERR_BAD_ARG;
  if (H < 3 || W < 3) return TD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  switch (n_src) {
    case 1: return td::run_bwd<1>(tgt, src, disp, P, invK, argmin, automask, gscale, inv_count, B, H, W, hs, ws, min_depth, max_depth, d_up, dP_partial, st);
    case 2: return td::run_bwd<2>(tgt, src, disp, P, invK, argmin, automask, gscale, inv_count, B, H, W, hs, ws, min_depth, max_depth, d_up, dP_partial, st);
    case 3: return td::run_bwd<3>(tgt, src, disp, P, invK, argmin, automask, gscale, inv_count, B, H, W, hs, ws, min_depth, max_depth, d_up, dP_partial, st);
    case 4: return td::run_bwd<4>(tgt, src, disp, P, invK, argmin, automask, gscale, inv_count, B, H, W, hs, ws, min_depth, max_depth, d_up, dP_partial, st);
  }
  return TD_ERR_BAD_ARG;
}

extern "C" int td_photo_bwd_num_blocks(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return B * ((H + td::BT_H - 1) / td::BT_H) * ((W + td::BT_W - 1) / td::BT_W);
}

extern "C" int td_reduce_dP(const float* dP_partial, int n_src, int B, int H, int W, float* dP,
                            td_stream_t stream) {
  if (!dP_partial || !dP || n_src < 1 || n_src > TD_MAX_SRC || B <= 0 || H <= 0 || W <= 0) return TD_ERR_BAD_ARG;
  const int bps = ((H + td::BT_H - 1) / td::BT_H) * ((W + td::BT_W - 1) / td::BT_W);
  hipLaunchKernelGGL(td::reduce_dP_kernel, dim3(n_src * 12, B), dim3(64), 0, (hipStream_t)stream,
                     dP_partial, n_src, B, bps, dP);
  return td::record_launch_error(hipGetLastError(), "td_reduce_dP");
}

extern "C" int td_upsample_adjoint(const float* d_up, int B, int H, int W, int hs, int ws,
                                   float* d_disp, int accumulate, td_stream_t stream) {
  if (!d_up || !d_disp || B <= 0 || hs <= 0 || ws <= 0 || hs > H || ws > W) return TD_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int total = B * hs * ws;
  const float f = fmaxf((float)H / (float)hs, (float)W / (float)ws);
  if (f <= 2.5f) {
    const int blocks = (total * 4 + TD_THREADS - 1) / TD_THREADS;
    hipLaunchKernelGGL(td::upsample_adjoint_kernel<4>, dim3(blocks), dim3(TD_THREADS), 0, st, d_up, B, H, W, hs, ws, d_disp, accumulate);
  } else if (f <= 5.f) {
    const int blocks = (total * 16 + TD_THREADS - 1) / TD_THREADS;
    hipLaunchKernelGGL(td::upsample_adjoint_kernel<16>, dim3(blocks), dim3(TD_THREADS), 0, st, d_up, B, H, W, hs, ws, d_disp, accumulate);
  } else {
    const int blocks = (total * 64 + TD_THREADS - 1) / TD_THREADS;
    hipLaunchKernelGGL(td::upsample_adjoint_kernel<64>, dim3(blocks), dim3(TD_THREADS), 0, st, d_up, B, H, W, hs, ws, d_disp, accumulate);
  }
  return td::record_launch_error(hipGetLastError(), "td_upsample_adjoint");
}
