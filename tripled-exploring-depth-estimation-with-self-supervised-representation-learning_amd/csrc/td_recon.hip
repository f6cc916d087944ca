// Masked SSIM + robust-L1 reconstruction loss of the in-painting auto-encoder
// (reference: mono/model/mono_fm_joint_inpaint/net.py:80-91 with compute_reprojection_loss,
// mono/model/mono_fm_joint/net.py:67-71 and SSIM, layers.py:85-107):
//     S = sum_p hole(p) * (0.85 * mean_c SSIM_c(x, y)(p) + 0.15 * mean_c sqrt((y - x)^2 + 1e-6))
// forward and d S / d x, in the same streaming form as the photometric forward: a wave owns a
// column strip and marches down the rows, horizontal 3-sums via DPP lane shifts, vertical ones via
// register rings; the SSIM adjoint is a box filter over per-window coefficients
// (d SSIM_p / d x_q = alpha_p + beta_p x_q + gamma_p y_q), so nothing is scattered.
#include "td_common.h"

namespace td {

constexpr int RC_ROWS = 8;
constexpr int RC_WAVES = 4;

struct ReconArgs {
  const float* x;       // prediction  [B,3,h,w]
  const float* y;       // target      [B,3,h,w]
  const float* hole;    // per-pixel weight [B,h,w]
  const float* gscale;  // backward only
  float* out;           // forward: per-task partial sums; backward: d/dx [B,3,h,w]
  int B, h, w;
  int nstrips, nchunks, ntasks, blocks_per_xcd;
};

template <int HALO>
__device__ __forceinline__ bool recon_task(const ReconArgs& a, int& strip, int& chunk, int& b, int& task) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int bid = (int)(blockIdx.x & 7) * a.blocks_per_xcd + (int)(blockIdx.x >> 3);
  task = bid * RC_WAVES + wave;
  if (task >= a.ntasks) return false;
  strip = task % a.nstrips;
  chunk = (task / a.nstrips) % a.nchunks;
  b = task / (a.nstrips * a.nchunks);
  return true;
}

__global__ __launch_bounds__(RC_WAVES * 64) void recon_fwd_kernel(const ReconArgs a) {
  int strip, chunk, b, task;
  if (!recon_task<1>(a, strip, chunk, b, task)) return;
  const int lane = threadIdx.x & 63;
  const int h = a.h, w = a.w;
  const unsigned plane = (unsigned)(h * w);
  const int x = strip * 62 - 1 + lane;
  const int qx = reflect1(x, w);
  const int y0 = chunk * RC_ROWS;
  const bool col_out = lane >= 1 && lane <= 62 && x < w;
  const float* xb = a.x + (size_t)b * 3 * plane;
  const float* yb = a.y + (size_t)b * 3 * plane;
  const float* hb = a.hole + (size_t)b * plane;
  float p_hy[2][3], p_hyy[2][3], p_hx[2][3], p_hxx[2][3], p_hxy[2][3], c_y[3], c_x[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    p_hy[0][c] = p_hy[1][c] = p_hyy[0][c] = p_hyy[1][c] = p_hx[0][c] = p_hx[1][c] = 0.f;
    p_hxx[0][c] = p_hxx[1][c] = p_hxy[0][c] = p_hxy[1][c] = c_y[c] = c_x[c] = 0.f;
  }
  float acc = 0.f;
  float nx[3], ny[3], nh;
  auto fetch = [&](int k) {
    const int r = y0 - 1 + k;
    const unsigned off = (unsigned)(reflect1(r, h) * w + qx);
#pragma unroll
    for (int c = 0; c < 3; ++c) { nx[c] = xb[c * plane + off]; ny[c] = yb[c * plane + off]; }
    int orow = r - 1;
    orow = orow < 0 ? 0 : (orow > h - 1 ? h - 1 : orow);
    nh = hb[(unsigned)(orow * w + (x < 0 ? 0 : (x > w - 1 ? w - 1 : x)))];
  };
  fetch(0);
#pragma unroll 1
  for (int k = 0; k < RC_ROWS + 2; ++k) {
    float xv[3], yv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { xv[c] = nx[c]; yv[c] = ny[c]; }
    const float hw = nh;
    if (k + 1 < RC_ROWS + 2) fetch(k + 1);
    const int orow = y0 - 2 + k;
    const bool emit = k >= 2 && orow < h;
    float ss = 0.f, l1 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float hy = hsum3(yv[c]), hyy = hsum3(yv[c] * yv[c]);
      const float hx = hsum3(xv[c]), hxx = hsum3(xv[c] * xv[c]), hxy = hsum3(xv[c] * yv[c]);
      if (emit) {
        ss += ssim_from_sums(p_hx[0][c] + p_hx[1][c] + hx, p_hy[0][c] + p_hy[1][c] + hy,
                             p_hxx[0][c] + p_hxx[1][c] + hxx, p_hyy[0][c] + p_hyy[1][c] + hyy,
                             p_hxy[0][c] + p_hxy[1][c] + hxy);
        const float df = c_y[c] - c_x[c];
        l1 += fast_sqrt(df * df + TD_L1_EPS2);
      }
      p_hy[0][c] = p_hy[1][c]; p_hy[1][c] = hy; p_hyy[0][c] = p_hyy[1][c]; p_hyy[1][c] = hyy;
      p_hx[0][c] = p_hx[1][c]; p_hx[1][c] = hx; p_hxx[0][c] = p_hxx[1][c]; p_hxx[1][c] = hxx;
      p_hxy[0][c] = p_hxy[1][c]; p_hxy[1][c] = hxy;
      c_y[c] = yv[c]; c_x[c] = xv[c];
    }
    if (emit && col_out) acc += hw * (0.85f * (ss * (1.f / 3.f)) + 0.15f * (l1 * (1.f / 3.f)));
  }
  const float tot = wave_sum(acc);
  if (lane == 0) a.out[task] = tot;
}

// backward: 60 output columns per wave (2-column halo on both sides), rows y0-2 .. y0+R+1
__global__ __launch_bounds__(RC_WAVES * 64) void recon_bwd_kernel(const ReconArgs a) {
  int strip, chunk, b, task;
  if (!recon_task<2>(a, strip, chunk, b, task)) return;
  const int lane = threadIdx.x & 63;
  const int h = a.h, w = a.w;
  const unsigned plane = (unsigned)(h * w);
  const int x = strip * 60 - 2 + lane;               // padded-domain column
  const int qx = reflect1(x, w);
  const int y0 = chunk * RC_ROWS;
  const bool col_in = x >= 0 && x < w;               // a real pixel column (may carry coefficients)
  const bool col_out = lane >= 2 && lane <= 61 && col_in;
  const float wl = (x == 1) ? 2.f : 1.f;             // reflection multiplicities of the horizontal neighbours
  const float wr = (x == w - 2) ? 2.f : 1.f;
  const float* xb = a.x + (size_t)b * 3 * plane;
  const float* yb = a.y + (size_t)b * 3 * plane;
  const float* hb = a.hole + (size_t)b * plane;
  const float g = a.gscale[0];
  const float g_ssim = g * 0.85f / 3.f / 9.f, g_l1 = g * 0.15f / 3.f;

  float p_hy[2][3], p_hyy[2][3], p_hx[2][3], p_hxx[2][3], p_hxy[2][3];
  float cx[2][3], cy[2][3], ch[2];                   // centres of the two previous rows (x, y, hole)
  float p_hc[2][9];                                  // horizontal coefficient sums of the two previous coef rows
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    p_hy[0][c] = p_hy[1][c] = p_hyy[0][c] = p_hyy[1][c] = p_hx[0][c] = p_hx[1][c] = 0.f;
    p_hxx[0][c] = p_hxx[1][c] = p_hxy[0][c] = p_hxy[1][c] = 0.f;
    cx[0][c] = cx[1][c] = cy[0][c] = cy[1][c] = 0.f;
  }
  ch[0] = ch[1] = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) p_hc[0][i] = p_hc[1][i] = 0.f;

  float nx[3], ny[3], nh;
  auto fetch = [&](int k) {
    const int r = y0 - 2 + k;
    const unsigned off = (unsigned)(reflect1(r, h) * w + qx);
#pragma unroll
    for (int c = 0; c < 3; ++c) { nx[c] = xb[c * plane + off]; ny[c] = yb[c * plane + off]; }
    nh = (r >= 0 && r < h && col_in) ? hb[(unsigned)(r * w + x)] : 0.f;   // weight of the window centred here
  };
  fetch(0);
  constexpr int NK = RC_ROWS + 4;
#pragma unroll 1
  for (int k = 0; k < NK; ++k) {
    float xv[3], yv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { xv[c] = nx[c]; yv[c] = ny[c]; }
    const float hv = nh;
    if (k + 1 < NK) fetch(k + 1);
    const int r = y0 - 2 + k;                        // row just loaded; window row p = r-1; gradient row q = r-2

    // ---- window statistics of row p = r-1 -> coefficients, weighted by hole(p) ----
    float cf[9];
    const float hp = ch[1];                          // hole weight of row r-1 (0 outside the image)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float hy = hsum3(yv[c]), hyy = hsum3(yv[c] * yv[c]);
      const float hx = hsum3(xv[c]), hxx = hsum3(xv[c] * xv[c]), hxy = hsum3(xv[c] * yv[c]);
      float al = 0.f, be = 0.f, ga = 0.f;
      if (k >= 2) {
        const float kk = 1.f / 9.f;
        const float mx = (p_hx[0][c] + p_hx[1][c] + hx) * kk, my = (p_hy[0][c] + p_hy[1][c] + hy) * kk;
        const float vx = (p_hxx[0][c] + p_hxx[1][c] + hxx) * kk - mx * mx;
        const float vy = (p_hyy[0][c] + p_hyy[1][c] + hyy) * kk - my * my;
        const float cxy = (p_hxy[0][c] + p_hxy[1][c] + hxy) * kk - mx * my;
        const float A1 = 2.f * mx * my + TD_SSIM_C1, A2 = 2.f * cxy + TD_SSIM_C2;
        const float B1 = mx * mx + my * my + TD_SSIM_C1, B2 = vx + vy + TD_SSIM_C2;
        const float n = A1 * A2, d = B1 * B2;
        const float invd = fast_rcp(d);
        const float s = (1.f - n * invd) * 0.5f;
        const float sc = (s >= 0.f && s <= 1.f) ? hp * g_ssim * invd : 0.f;
        const float q = n * invd;
        al = -sc * (my * (A2 - A1) - q * mx * (B2 - B1));
        be = sc * q * B1;
        ga = -sc * A1;
      }
      cf[c * 3 + 0] = al; cf[c * 3 + 1] = be; cf[c * 3 + 2] = ga;
      p_hy[0][c] = p_hy[1][c]; p_hy[1][c] = hy; p_hyy[0][c] = p_hyy[1][c]; p_hyy[1][c] = hyy;
      p_hx[0][c] = p_hx[1][c]; p_hx[1][c] = hx; p_hxx[0][c] = p_hxx[1][c]; p_hxx[1][c] = hxx;
      p_hxy[0][c] = p_hxy[1][c]; p_hxy[1][c] = hxy;
    }
    // ---- box filter of the coefficients: horizontal now, vertical over rows r-3, r-2, r-1 ----
    const int q = r - 2;                              // gradient row
    const float wy0 = (q == 1) ? 2.f : 1.f, wy2 = (q == h - 2) ? 2.f : 1.f;
    float gq[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) gq[c] = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float hc = (wl * lane_left(cf[i]) + cf[i]) + wr * lane_right(cf[i]);
      const float G = wy0 * p_hc[0][i] + p_hc[1][i] + wy2 * hc;
      const int c = i / 3, t = i % 3;
      gq[c] += (t == 0) ? G : (t == 1 ? G * cx[0][c] : G * cy[0][c]);
      p_hc[0][i] = p_hc[1][i]; p_hc[1][i] = hc;
    }
    if (k >= 4 && q < h && col_out) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float df = cx[0][c] - cy[0][c];
        const float v = gq[c] + ch[0] * g_l1 * df * fast_rcp(fast_sqrt(df * df + TD_L1_EPS2));
        a.out[((size_t)b * 3 + c) * plane + (unsigned)(q * w + x)] = v;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { cx[0][c] = cx[1][c]; cx[1][c] = xv[c]; cy[0][c] = cy[1][c]; cy[1][c] = yv[c]; }
    ch[0] = ch[1]; ch[1] = hv;
  }
}

static void recon_geometry(ReconArgs& a, int cols) {
  a.nstrips = (a.w + cols - 1) / cols;
  a.nchunks = (a.h + RC_ROWS - 1) / RC_ROWS;
  a.ntasks = a.B * a.nstrips * a.nchunks;
  const int blocks = (a.ntasks + RC_WAVES - 1) / RC_WAVES;
  a.blocks_per_xcd = (blocks + 7) / 8;
}

}  // namespace td

extern "C" int td_recon_num_tasks(int B, int h, int w) {
  if (B <= 0 || h <= 0 || w <= 0) return 0;
  td::ReconArgs a; a.B = B; a.h = h; a.w = w;
  td::recon_geometry(a, 62);
  return a.ntasks;
}

extern "C" int td_recon_fwd(const float* x, const float* y, const float* hole, int B, int h, int w,
                            float* partial, td_stream_t stream) {
  if (!x || !y || !hole || !partial || B <= 0) return TD_ERR_BAD_ARG;
  if (h < 3 || w < 3 || (long long)B * 3 * h * w >= (1ll << 31)) return TD_ERR_UNSUPPORTED;
  td::ReconArgs a; a.x = x; a.y = y; a.hole = hole; a.gscale = nullptr; a.out = partial; a.B = B; a.h = h; a.w = w;
  td::recon_geometry(a, 62);
  hipLaunchKernelGGL(td::recon_fwd_kernel, dim3(a.blocks_per_xcd * 8), dim3(td::RC_WAVES * 64), 0, (hipStream_t)stream, a);
  return td::record_launch_error(hipGetLastError(), "td_recon_fwd");
}

extern "C" int td_recon_bwd(const float* x, const float* y, const float* hole, const float* gscale, int B, int h,
                            int w, float* dx, td_stream_t stream) {
  if (!x || !y || !hole || !gscale || !dx || B <= 0) return TD_ERR_BAD_ARG;
  if (h < 3 || w < 3 || (long long)B * 3 * h * w >= (1ll << 31)) return TD_ERR_UNSUPPORTED;
  td::ReconArgs a; a.x = x; a.y = y; a.hole = hole; a.gscale = gscale; a.out = dx; a.B = B; a.h = h; a.w = w;
  td::recon_geometry(a, 60);
  hipLaunchKernelGGL(td::recon_bwd_kernel, dim3(a.blocks_per_xcd * 8), dim3(td::RC_WAVES * 64), 0, (hipStream_t)stream, a);
  return td::record_launch_error(hipGetLastError(), "td_recon_bwd");
}
