// Fused photometric forward: upsample(disp) -> depth -> back-project -> project -> bilinear
// gather -> 3x3 reflect-padded SSIM + robust L1 -> per-pixel min over {identity, warped}.
// One 256-thread workgroup per 16x64 tile; the warped sources and the target are staged in
// LDS with a 1-pixel halo, each wave then slides a 3-row window down its 4x64 strip so every
// horizontal 3-sum is formed once and reused by the three vertical windows that contain it.
#include "td_common.h"

namespace td {

constexpr int FT_H = TD_FWD_TILE_H;
constexpr int FT_W = TD_TILE_W;
constexpr int FH = FT_H + 2;   // halo rows
constexpr int FW = FT_W + 2;   // halo cols

template <int NS>
struct PhotoFwdArgs {
  const float* tgt;
  const float* src[NS];
  const float* disp;
  const float* P;
  const float* invK;
  const float* idloss;
  const float* noise;
  uint8_t* argmin;
  float* warped;
  float* min_map;
  float* partial;
  float* idloss_out;   // identity mode only
  int B, H, W, hs, ws;
  float min_disp, disp_range;
};

// IDENT = true: "pred" is the raw source frame (auto-mask term); writes idloss_out.
template <int NS, bool IDENT>
__global__ __launch_bounds__(TD_THREADS) void photo_fwd_kernel(const PhotoFwdArgs<NS> a) {
  __shared__ float s_y[3][FH][FW];
  __shared__ float s_x[NS][3][FH][FW];
  __shared__ float s_cam[9 + NS * 12];
  __shared__ float s_red[4];

  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int ty0 = blockIdx.y * FT_H, tx0 = blockIdx.x * FT_W;
  const int H = a.H, W = a.W;
  const size_t plane = (size_t)H * W;

  if (!IDENT) {
    if (tid < 9) s_cam[tid] = a.invK[(size_t)b * 16 + (tid / 3) * 4 + (tid % 3)];
    if (tid >= 64 && tid < 64 + NS * 12) {
      const int k = tid - 64, f = k / 12, e = k % 12;
      s_cam[9 + k] = a.P[((size_t)f * a.B + b) * 12 + e];
    }
    __syncthreads();
  }

  const float ry = (float)a.hs / (float)H, rx = (float)a.ws / (float)W;
  const float* dispb = IDENT ? nullptr : a.disp + (size_t)b * a.hs * a.ws;

  // ---- phase 1: stage target + predictions for the haloed tile ----
  for (int pos = tid; pos < FH * FW; pos += TD_THREADS) {
    const int py = pos / FW, px = pos - py * FW;
    const int qy = reflect1(ty0 + py - 1, H), qx = reflect1(tx0 + px - 1, W);
    const size_t off = (size_t)qy * W + qx;
#pragma unroll
    for (int c = 0; c < 3; ++c) s_y[c][py][px] = a.tgt[((size_t)b * 3 + c) * plane + off];
    if (IDENT) {
#pragma unroll
      for (int f = 0; f < NS; ++f)
#pragma unroll
        for (int c = 0; c < 3; ++c) s_x[f][c][py][px] = a.src[f][((size_t)b * 3 + c) * plane + off];
    } else {
      const float d = upsample_disp(dispb, a.hs, a.ws, ry, rx, qy, qx);
      const float depth = fast_rcp(a.min_disp + a.disp_range * d);
      const bool own = a.warped != nullptr && py >= 1 && py <= FT_H && px >= 1 && px <= FT_W &&
                       (ty0 + py - 1) < H && (tx0 + px - 1) < W;
      Tap taps[NS];
#pragma unroll
      for (int f = 0; f < NS; ++f) {
        float pt[3], cz[3];
        taps[f] = project_tap(s_cam, s_cam + 9 + f * 12, depth, qx, qy, W, H, pt, cz);
      }
      TapVals tv[NS][3];
#pragma unroll
      for (int f = 0; f < NS; ++f)
#pragma unroll
        for (int c = 0; c < 3; ++c) tv[f][c] = load_taps(a.src[f] + ((size_t)b * 3 + c) * plane, W, taps[f]);
#pragma unroll
      for (int f = 0; f < NS; ++f)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float v = blend_taps(tv[f][c], taps[f]);
          s_x[f][c][py][px] = v;
          if (own) a.warped[(((size_t)f * a.B + b) * 3 + c) * plane + off] = v;
        }
    }
  }
  __syncthreads();

  // ---- phase 2: sliding-window SSIM + L1 down a 4-row strip per wave ----
  const int cx = tid & 63, rg = tid >> 6;
  const int gx = tx0 + cx;
  float hy[3][3], hyy[3][3];            // [ring row][channel] horizontal 3-sums
  float hx[3][NS][3], hxx[3][NS][3], hxy[3][NS][3];
  float cy[2][3], cxv[2][NS][3];        // centre values of the two most recent rows
  float acc = 0.f;

#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int r = rg * 4 + k;           // LDS row (tile row r-1)
    const int slot = k % 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float y0 = s_y[c][r][cx], y1 = s_y[c][r][cx + 1], y2 = s_y[c][r][cx + 2];
      hy[slot][c] = y0 + y1 + y2;
      hyy[slot][c] = y0 * y0 + y1 * y1 + y2 * y2;
      cy[k & 1][c] = y1;
#pragma unroll
      for (int f = 0; f < NS; ++f) {
        const float x0 = s_x[f][c][r][cx], x1 = s_x[f][c][r][cx + 1], x2 = s_x[f][c][r][cx + 2];
        hx[slot][f][c] = x0 + x1 + x2;
        hxx[slot][f][c] = x0 * x0 + x1 * x1 + x2 * x2;
        hxy[slot][f][c] = x0 * y0 + x1 * y1 + x2 * y2;
        cxv[k & 1][f][c] = x1;
      }
    }
    if (k >= 2) {
      const int gy = ty0 + rg * 4 + (k - 2);
      const int pc = (k - 1) & 1;       // centre row = previous row
      float loss[NS];
#pragma unroll
      for (int f = 0; f < NS; ++f) {
        float ss = 0.f, l1 = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float sy = hy[0][c] + hy[1][c] + hy[2][c];
          const float syy = hyy[0][c] + hyy[1][c] + hyy[2][c];
          const float sx = hx[0][f][c] + hx[1][f][c] + hx[2][f][c];
          const float sxx = hxx[0][f][c] + hxx[1][f][c] + hxx[2][f][c];
          const float sxy = hxy[0][f][c] + hxy[1][f][c] + hxy[2][f][c];
          ss += ssim_from_sums(sx, sy, sxx, syy, sxy);
          const float df = cy[pc][c] - cxv[pc][f][c];
          l1 += fast_sqrt(df * df + TD_L1_EPS2);
        }
        loss[f] = 0.85f * (ss * (1.f / 3.f)) + 0.15f * (l1 * (1.f / 3.f));
      }
      if (gy < H && gx < W) {
        const size_t pix = (size_t)gy * W + gx;
        if (IDENT) {
#pragma unroll
          for (int f = 0; f < NS; ++f) a.idloss_out[((size_t)b * NS + f) * plane + pix] = loss[f];
        } else {
          float best = 0.f;
          int idx = 0;
          bool have = false;
          if (a.idloss != nullptr) {
#pragma unroll
            for (int f = 0; f < NS; ++f) {
              float v = a.idloss[((size_t)b * NS + f) * plane + pix];
              if (a.noise != nullptr) v += a.noise[((size_t)f * a.B + b) * plane + pix] * 1e-5f;
              if (!have || v < best) { best = v; idx = f; have = true; }
            }
          }
          const int base = (a.idloss != nullptr) ? NS : 0;
#pragma unroll
          for (int f = 0; f < NS; ++f) {
            if (!have || loss[f] < best) { best = loss[f]; idx = base + f; have = true; }
          }
          a.argmin[(size_t)b * plane + pix] = (uint8_t)idx;
          if (a.min_map != nullptr) a.min_map[(size_t)b * plane + pix] = best;
          acc += best;
        }
      }
    }
  }

  if (!IDENT) {
    const float tot = block_sum(acc, s_red);
    if (tid == 0) a.partial[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
  }
}

template <int NS, bool IDENT>
static int launch_fwd(const PhotoFwdArgs<NS>& a, hipStream_t st) {
  dim3 grid((a.W + FT_W - 1) / FT_W, (a.H + FT_H - 1) / FT_H, a.B);
  hipLaunchKernelGGL((photo_fwd_kernel<NS, IDENT>), grid, dim3(TD_THREADS), 0, st, a);
  return record_launch_error(hipGetLastError(), IDENT ? "td_photo_identity" : "td_photo_fwd");
}

template <int NS>
static int run_fwd(const float* tgt, const float* const* src, const float* disp, const float* P,
                   const float* invK, const float* idloss, const float* noise, int B, int H, int W,
                   int hs, int ws, float min_depth, float max_depth, uint8_t* argmin, float* warped,
                   float* min_map, float* partial, float* idloss_out, bool ident, hipStream_t st) {
  PhotoFwdArgs<NS> a;
  a.tgt = tgt;
  for (int i = 0; i < NS; ++i) a.src[i] = src[i];
  a.disp = disp; a.P = P; a.invK = invK; a.idloss = idloss; a.noise = noise;
  a.argmin = argmin; a.warped = warped; a.min_map = min_map; a.partial = partial;
  a.idloss_out = idloss_out;
  a.B = B; a.H = H; a.W = W; a.hs = hs; a.ws = ws;
  const double lo = 1.0 / (double)max_depth, hi = 1.0 / (double)min_depth;
  a.min_disp = (float)lo;
  a.disp_range = (float)(hi - lo);
  return ident ? launch_fwd<NS, true>(a, st) : launch_fwd<NS, false>(a, st);
}

}  // namespace td

extern "C" int td_photo_num_blocks(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return B * ((H + td::FT_H - 1) / td::FT_H) * ((W + td::FT_W - 1) / td::FT_W);
}

static int dispatch_fwd(const float* tgt, const float* const* src, int n_src, const float* disp,
                        const float* P, const float* invK, const float* idloss, const float* noise,
                        int B, int H, int W, int hs, int ws, float min_depth, float max_depth,
                        uint8_t* argmin, float* warped, float* min_map, float* partial,
                        float* idloss_out, bool ident, td_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (n_src) {
    case 1: return td::run_fwd<1>(tgt, src, disp, P, invK, idloss, noise, B, H, W, hs, ws, min_depth, max_depth, argmin, warped, min_map, partial, idloss_out, ident, st);
    case 2: return td::run_fwd<2>(tgt, src, disp, P, invK, idloss, noise, B, H, W, hs, ws, min_depth, max_depth, argmin, warped, min_map, partial, idloss_out, ident, st);
    case 3: return td::run_fwd<3>(tgt, src, disp, P, invK, idloss, noise, B, H, W, hs, ws, min_depth, max_depth, argmin, warped, min_map, partial, idloss_out, ident, st);
    case 4: return td::run_fwd<4>(tgt, src, disp, P, invK, idloss, noise, B, H, W, hs, ws, min_depth, max_depth, argmin, warped, min_map, partial, idloss_out, ident, st);
  }
  return TD_ERR_BAD_ARG;
}

extern "C" int td_photo_identity(const float* tgt, const float* const* src, int n_src, int B, int H,
                                 int W, float* idloss, td_stream_t stream) {
  if (!tgt || !src || !idloss || n_src < 1 || n_src > TD_MAX_SRC || B <= 0) return TD_ERR_BAD_ARG;
  for (int i = 0; i < n_src; ++i) if (!src[i]) return TD_ERR_BAD_ARG;
  if (H < 3 || W < 3) return TD_ERR_UNSUPPORTED;
  return dispatch_fwd(tgt, src, n_src, nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, 1, 1,
                      0.1f, 100.f, nullptr, nullptr, nullptr, nullptr, idloss, true, stream);
}

extern "C" int td_photo_fwd(const float* tgt, const float* const* src, int n_src, const float* disp,
                            const float* P, const float* invK, const float* idloss, const float* noise,
                            int B, int H, int W, int hs, int ws, float min_depth, float max_depth,
                            uint8_t* argmin, float* warped, float* min_map, float* partial,
                            td_stream_t stream) {
  if (!tgt || !src || !disp || !P || !invK || !argmin || !partial) return TD_ERR_BAD_ARG;
  if (n_src < 1 || n_src > TD_MAX_SRC || B <= 0 || hs <= 0 || ws <= 0 || hs > H || ws > W) return TD_ERR_BAD_ARG;
  if (!(min_depth > 0.f) || !(max_depth > min_depth)) return TD_ERR_BAD_ARG;
  for (int i = 0; i < n_src; ++i) if (!src[i]) return TD_ERR_BAD_ARG;
  if (H < 3 || W < 3 || (long long)B * 3 * H * W >= (1ll << 31)) return TD_ERR_UNSUPPORTED;
  return dispatch_fwd(tgt, src, n_src, disp, P, invK, idloss, noise, B, H, W, hs, ws, min_depth,
                      max_depth, argmin, warped, min_map, partial, nullptr, false, stream);
}
