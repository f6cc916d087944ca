// Fused photometric backward, streaming form.  The forward leaves, per pixel, the three SSIM-adjoint
// coefficients per channel of the warped frame its arg-min selected
// (d SSIM_p / d x_q = (alpha_p + beta_p x_q + gamma_p y_q) / 9 for every member q of p's window), so the
// adjoint of the 3x3 windows is a box filter over that field and nothing is scattered.
// A wave owns a strip of 62 columns of ONE source frame and marches down the rows (the frames of a strip are separate wave
// tasks, adjacent in the task order so that they share the coefficient / target / arg-min rows in one XCD's L2: a
// one-frame body needs 164 registers instead of 256, i.e. three waves per SIMD instead of two); per row it
//   1. re-warps the source (two-deep load pipeline as in the forward) and forms d x_c / d(u, v),
//   2. box-filters the coefficient field (horizontal: DPP lane shifts, vertical: two-row register
//      ring; reflection padding = integer multiplicities) -> d loss / d warped pixel one row back,
//      adds the robust-L1 adjoint,
//   3. chains through the bilinear sampler, the projection and the depth.
// No LDS, no barriers, no atomics: every frame task writes its own d_up plane once per pixel (td_upsample_adjoint_planes sums
// the planes while it gathers), dL/dP is accumulated in registers and written once per wave.
// The library is built with -ffp-contract=off for the forward's sake (SSIM variances).  The adjoint has no such
// cancellation, so this translation unit lets the compiler fuse multiply-adds (-4 % time; gradients move by ~1 ulp).
#pragma clang fp contract(fast)
#include "td_common.h"

namespace td {

// gradient rows per wave task: chosen per shape by pick_rows() (PhotoBwdArgs::rows)
constexpr int BS_COLS = 62;       // gradient columns per wave task (1-column halo on both sides)
constexpr int BS_WAVES = 4;

template <int NS>
struct PhotoBwdArgs {
  const float* tgt;          // NCHW frames, or RGBX pixels [B,H,W,4] in the PX form of the kernel
  const float* src[NS];
  const float* disp;
  const float* P;
  const float* invK;
  const uint8_t* argmin;
  const float* coef;
  const float* gscale;
  float* d_up;
  float* dP_partial;
  int B, H, W, hs, ws;
  int n_ident;            // candidates preceding the warped ones (n_src when automasking, else 0)
  int nstrips, nchunks, ntasks, blocks_per_xcd, rows;
  float inv_count;
  float min_disp, disp_range;
};

// PX: the frames are RGBX pixels (one 16-byte load per tap / target pixel instead of three dword loads).  Used at the FINEST scale
// only (hs >= H/2): there the disparity carries per-pixel detail, the dword gathers of neighbouring lanes stop sharing cache lines
// and the pixel-major form is 3 us faster (57.2 -> 54.5 us at B=12 192x640); at the coarser scales the warp is coherent, the
// planar dword loads touch fewer lines and the pixel-major form is 1-3 us SLOWER (csrc/td_common.h, "packed frames").
template <int NS, bool PX>
// second bound: three waves per SIMD (<= 168 registers; the one-frame body takes 164 without spills)
__global__ __launch_bounds__(BS_WAVES * 64, 3) void photo_bwd_kernel(const PhotoBwdArgs<NS> a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int bid = (int)(blockIdx.x & 7) * a.blocks_per_xcd + (int)(blockIdx.x >> 3);
  const int ftask = bid * BS_WAVES + wave;           // (strip task, frame), frame fastest
  if (ftask >= a.ntasks * NS) return;
  const int f = ftask % NS, task = ftask / NS;
  const int strip = task % a.nstrips;
  const int chunk = (task / a.nstrips) % a.nchunks;
  const int b = task / (a.nstrips * a.nchunks);

  const int H = a.H, W = a.W;
  const unsigned plane = (unsigned)(H * W);
  const int x = strip * BS_COLS - 1 + lane;          // padded-domain column of this lane
  const int y0 = chunk * a.rows;
  const bool col_in = x >= 0 && x < W;
  const bool col_out = lane >= 1 && lane <= BS_COLS && col_in;
  const int xc = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
  const float wl = (x == 1) ? 2.f : 1.f, wr = (x == W - 2) ? 2.f : 1.f;

  const float* tgtb = a.tgt + (size_t)b * (PX ? 4 : 3) * plane;
  const float* cfb = a.coef + (size_t)b * 9 * plane;
  const uint8_t* amb = a.argmin + (size_t)b * plane;
  float* dupb = a.d_up + ((size_t)f * a.B + b) * plane;      // this frame's plane of d_up [NS,B,H,W]
  const float* dispb = a.disp + (size_t)b * a.hs * a.ws;
  float ik[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) ik[i] = a.invK[b * 16 + (i / 3) * 4 + (i % 3)];
  const UpIdx ux = up_index(xc, (float)a.ws / (float)W, a.ws);
  const float ratio_y = (float)a.hs / (float)H;
  const float fx = (float)xc;
  const float rx0 = ik[0] * fx, rx1 = ik[3] * fx, rx2 = ik[6] * fx;
  const float g = a.gscale[0] * a.inv_count;
  const float g_ssim = g * 0.85f / 3.f / 9.f, g_l1 = g * 0.15f / 3.f;
  const float sx_scale = (float)W / (float)(W - 1), sy_scale = (float)H / (float)(H - 1);   // d ix / d u, d iy / d v
  const int NK = a.rows + 2;

  {
    const float* srcp = a.src[0];                    // (a select chain: dynamic indexing of the by-value argument struct would go to scratch)
#pragma unroll
    for (int i = 1; i < NS; ++i) srcp = (f == i) ? a.src[i] : srcp;
    const float* srcb = srcp + (size_t)b * (PX ? 4 : 3) * plane;
    float P[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) P[e] = a.P[(f * a.B + b) * 12 + e];
    const int sel = a.n_ident + f;

    float p_hc[2][9];                                // horizontal coefficient sums of the two previous rows
#pragma unroll
    for (int i = 0; i < 9; ++i) p_hc[0][i] = p_hc[1][i] = 0.f;
    float dP[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) dP[e] = 0.f;

    // ---- load pipeline ----
    // coefficient row k (padded row r = y0-1+k) is consumed in iteration k;
    // the warp of gradient row q = r-1 is consumed in the same iteration.
    float dv[4], dv_n[4], ul0, ul1, ul0_n, ul1_n;
    Tap tap = {};
    TapVals tv[3] = {};
    TapVals4 tv4 = {};
    float depth_q = 0.f;
    // field pipeline, THREE rows deep (the backward is memory-latency bound: SQ_WAIT_ANY 47 % with two): slot A holds
    // coefficient row k (loaded two iterations ago), slot B row k+1 (loaded in the previous iteration); the raw arg-min
    // bytes travel with them and the masks are formed at consumption, so no load is waited for when it is issued
    float cfA[9], cfB[9], yqA[3], yqB[3];
    unsigned amA = 0, amB = 0, aqA = 0, aqB = 0;
    auto issue_disp = [&](int q, float* d4, float& l0, float& l1) {        // gradient row q (clamped)
      const int qc = q < 0 ? 0 : (q > H - 1 ? H - 1 : q);
      const UpIdx uy = up_index(qc, ratio_y, a.hs);
      const unsigned o0 = (unsigned)(uy.i0 * a.ws), o1 = (unsigned)(uy.i1 * a.ws);
      d4[0] = ld_at(dispb, (o0 + (unsigned)ux.i0) * 4u); d4[1] = ld_at(dispb, (o0 + (unsigned)ux.i1) * 4u);
      d4[2] = ld_at(dispb, (o1 + (unsigned)ux.i0) * 4u); d4[3] = ld_at(dispb, (o1 + (unsigned)ux.i1) * 4u);
      l0 = uy.l0; l1 = uy.l1;
    };
    auto issue_field = [&](int k, float* cf9, float* yq3, unsigned& am, unsigned& aq) {
      const int r = y0 - 1 + k;                      // coefficient row; its gradient row is q = r - 1
      const int rc = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
      const unsigned offc = (unsigned)(rc * W + xc);
#pragma unroll
      for (int i = 0; i < 9; ++i) cf9[i] = ld_at(cfb + (size_t)i * plane, offc * 4u);
      am = amb[offc];
      const int q = r - 1;
      const int qc = q < 0 ? 0 : (q > H - 1 ? H - 1 : q);
      const unsigned offq = (unsigned)(qc * W + xc);
      if (PX) {
        const f4 t = ld4_at(tgtb, offq * 16u);
        yq3[0] = t[0]; yq3[1] = t[1]; yq3[2] = t[2];
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) yq3[c] = ld_at(tgtb + (size_t)c * plane, offq * 4u);
      }
      aq = amb[offq];
    };
    auto issue_warp = [&](int k, const float* d4, float l0, float l1) {
      // rows k = 0, 1 only feed the vertical window of the coefficient sums: their warp (q above the chunk) is
      // never consumed, so the projection and the 12 gathers are skipped (k is wave-uniform: a scalar branch)
      if (k >= 2) {
        const int q = y0 - 2 + k;
        const int qc = q < 0 ? 0 : (q > H - 1 ? H - 1 : q);
        const float dd = l0 * (ux.l0 * d4[0] + ux.l1 * d4[1]) + l1 * (ux.l0 * d4[2] + ux.l1 * d4[3]);
        depth_q = fast_rcp(a.min_disp + a.disp_range * dd);
        const float fy = (float)qc;
        float pt[3], cz[3];
        tap = project_ray(rx0 + ik[1] * fy + ik[2], rx1 + ik[4] * fy + ik[5], rx2 + ik[7] * fy + ik[8], P, depth_q,
                          W, H, pt, cz);
        if (PX) {
          tv4 = load_taps4(srcb, W, tap);
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) tv[c] = load_taps(srcb + c * plane, W, tap);
        }
      }
    };
    issue_disp(y0 - 2, dv, ul0, ul1);                // q of k = 0
    issue_disp(y0 - 1, dv_n, ul0_n, ul1_n);          // q of k = 1
    issue_field(0, cfA, yqA, amA, aqA);
    issue_field(1, cfB, yqB, amB, aqB);
    issue_warp(0, dv, ul0, ul1);

    // One row of the march.  The pipelines are two-slot rings; instead of shifting them (cfA = cfB, p_hc[0] = p_hc[1],
    // dv = dv_n: 100 of the 420 vector instructions of an iteration were register moves) the row is instantiated
    // twice with the slot roles exchanged, so every load lands in the registers its consumer reads.
    //   F0: field slot holding row k (consumed here, then refilled with row k + 2)   F1: row k + 1
    //   D0: disparity slot of q(k) (dead: its warp was issued one row ago; refilled with q(k + 2))   D1: q(k + 1)
    //   pOld / pNew: horizontal coefficient sums of rows k - 2 / k - 1 (pOld receives row k)
    auto row = [&](const int k, float (&cf0)[9], float (&yq0)[3], unsigned& am0, unsigned& aq0,
                   float (&d0)[4], float& d0l0, float& d0l1, const float (&d1)[4], const float d1l0, const float d1l1,
                   float (&pOld)[9], const float (&pNew)[9]) __attribute__((always_inline)) {
      // ---- consume: coefficient row r, warp of row q = r-1 ----
      const int r_k = y0 - 1 + k;
      const bool inside = r_k >= 0 && r_k < H && col_in;
      const float m_k = (inside && (int)am0 == sel) ? g_ssim : 0.f;
      const int q_k = r_k - 1;
      const float ml1 = (q_k >= 0 && q_k < H && (int)aq0 == sel) ? g_l1 : 0.f;
      float cf[9], xq[3], yv[3], dxi[3], dyi[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) cf[i] = cf0[i] * m_k;
#pragma unroll
      for (int c = 0; c < 3; ++c) { xq[c] = 0.f; yv[c] = yq0[c]; dxi[c] = 0.f; dyi[c] = 0.f; }
      if (k >= 2) {                                   // rows above the chunk have no warp (see issue_warp)
        const float ex = (float)tap.x0 + 1.f - tap.ix, wx = tap.ix - (float)tap.x0;
        const float ey = (float)tap.y0 + 1.f - tap.iy, wy = tap.iy - (float)tap.y0;
        const float mx = tap.gmx * sx_scale, my = tap.gmy * sy_scale;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const TapVals t = PX ? tap_channel(tv4, c) : tv[c];
          // weights re-formed here instead of carried across the row (4 registers); (float)x0 == floor(ix) after the
          // clamp, so these are the forward's ex * ey, ... bit for bit, summed in ATen's order nw, ne, sw, se
          float o = t.nw * (ex * ey);
          o += t.ne * (wx * ey);
          o += t.sw * (ex * wy);
          o += t.se * (wx * wy);
          xq[c] = o;
          const float vnw = t.nw;
          const float vne = tap.in_e ? t.ne : 0.f;
          const float vsw = tap.in_s ? t.sw : 0.f;
          const float vse = (tap.in_e && tap.in_s) ? t.se : 0.f;
          dxi[c] = (-vnw * ey + vne * ey - vsw * wy + vse * wy) * mx;     // d x_c / d u
          dyi[c] = (-vnw * ex - vne * wx + vsw * ex + vse * wx) * my;     // d x_c / d v
        }
      }
      const float dq = depth_q;
      // ---- refill the slots just consumed (rows beyond the chunk are clamped duplicates, never consumed) ----
      issue_field(k + 2, cf0, yq0, am0, aq0);
      issue_disp(y0 + k, d0, d0l0, d0l1);            // q of k + 2
      issue_warp(k + 1, d1, d1l0, d1l1);

      // ---- box filter of the coefficient field -> gradient w.r.t. the warped pixel of row q ----
      const int q = y0 - 2 + k;
      const float wy0 = (q == 1) ? 2.f : 1.f, wy2 = (q == H - 2) ? 2.f : 1.f;
      float gw[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const float hc = (wl * lane_left(cf[i]) + cf[i]) + wr * lane_right(cf[i]);
        const float G = wy0 * pOld[i] + pNew[i] + wy2 * hc;
        const int c = i / 3, t = i % 3;
        gw[c] += (t == 0) ? G : (t == 1 ? G * xq[c] : G * yv[c]);
        pOld[i] = hc;
      }
      if (k >= 2 && q < H && col_out) {
        float du = 0.f, dvv = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float df = xq[c] - yv[c];
          const float v = gw[c] + ml1 * df * fast_rcp(fast_sqrt(df * df + TD_L1_EPS2));
          du += v * dxi[c];
          dvv += v * dyi[c];
        }
        const float fyq = (float)q;
        const float r0 = rx0 + ik[1] * fyq + ik[2], r1 = rx1 + ik[4] * fyq + ik[5], r2 = rx2 + ik[7] * fyq + ik[8];
        const float X = dq * r0, Y = dq * r1, Z = dq * r2;
        const float c0 = P[0] * X + P[1] * Y + P[2] * Z + P[3];
        const float c1 = P[4] * X + P[5] * Y + P[6] * Z + P[7];
        const float z = P[8] * X + P[9] * Y + P[10] * Z + P[11] + 1e-7f;
        const float iz = fast_rcp(z);
        const float dc0 = du * iz, dc1 = dvv * iz, dc2 = -(du * c0 + dvv * c1) * iz * iz;
        dP[0] += dc0 * X; dP[1] += dc0 * Y; dP[2] += dc0 * Z; dP[3] += dc0;
        dP[4] += dc1 * X; dP[5] += dc1 * Y; dP[6] += dc1 * Z; dP[7] += dc1;
        dP[8] += dc2 * X; dP[9] += dc2 * Y; dP[10] += dc2 * Z; dP[11] += dc2;
        const float dX = dc0 * P[0] + dc1 * P[4] + dc2 * P[8];
        const float dY = dc0 * P[1] + dc1 * P[5] + dc2 * P[9];
        const float dZ = dc0 * P[2] + dc1 * P[6] + dc2 * P[10];
        const float dD = dX * r0 + dY * r1 + dZ * r2;
        const float contrib = dD * (-a.disp_range * dq * dq);
        dupb[(unsigned)(q * W + x)] = contrib;
      }
    };
    int k = 0;
#pragma unroll 1
    for (; k + 1 < NK; k += 2) {
      row(k, cfA, yqA, amA, aqA, dv, ul0, ul1, dv_n, ul0_n, ul1_n, p_hc[0], p_hc[1]);
      row(k + 1, cfB, yqB, amB, aqB, dv_n, ul0_n, ul1_n, dv, ul0, ul1, p_hc[1], p_hc[0]);
    }
    if (k < NK) row(k, cfA, yqA, amA, aqA, dv, ul0, ul1, dv_n, ul0_n, ul1_n, p_hc[0], p_hc[1]);
#pragma unroll
    for (int e = 0; e < 12; ++e) {
      const float tot = wave_sum(dP[e]);
      if (lane == 0) a.dP_partial[(size_t)task * (NS * 12) + f * 12 + e] = tot;
    }
  }
}

// strip tasks (each is run once per source frame).  The row tiling is a function of (B, H, W) only, so that
// td_photo_bwd_num_blocks / td_reduce_dP agree with the launch for every frame count: it is sized for two frame tasks per
// strip (the monocular configurations) at three resident waves per SIMD.
static int bwd_tasks(int B, int H, int W, int* nstrips, int* nchunks, int* rows) {
  *nstrips = (W + BS_COLS - 1) / BS_COLS;
  *rows = pick_rows(B * (*nstrips) * 2, H, 2, 1, 8, 64, 256 * 4 * 3);
  *nchunks = (H + *rows - 1) / *rows;
  return B * (*nstrips) * (*nchunks);
}

template <int NS>
static int run_bwd(const float* tgt, const float* const* src, const float* tgt_px, const float* const* src_px, const float* disp, const float* P,
                   const float* invK, const uint8_t* argmin, const float* coef, int automask, const float* gscale,
                   float inv_count, int B, int H, int W, int hs, int ws, float min_depth,
                   float max_depth, float* d_up, float* dP_partial, hipStream_t st) {
  PhotoBwdArgs<NS> a;
  const bool px = tgt_px != nullptr && src_px != nullptr && 2 * hs >= H;      // pixel-major frames at the finest scale only
  a.tgt = px ? tgt_px : tgt;
  for (int i = 0; i < NS; ++i) a.src[i] = px ? src_px[i] : src[i];
  a.disp = disp; a.P = P; a.invK = invK; a.argmin = argmin; a.coef = coef; a.gscale = gscale;
  a.d_up = d_up; a.dP_partial = dP_partial;
  a.B = B; a.H = H; a.W = W; a.hs = hs; a.ws = ws;
  a.n_ident = automask ? NS : 0;
  a.inv_count = inv_count;
  const double lo = 1.0 / (double)max_depth, hi = 1.0 / (double)min_depth;
  a.min_disp = (float)lo;
  a.disp_range = (float)(hi - lo);
  a.ntasks = bwd_tasks(B, H, W, &a.nstrips, &a.nchunks, &a.rows);
  const int blocks = (a.ntasks * NS + BS_WAVES - 1) / BS_WAVES;
  a.blocks_per_xcd = (blocks + 7) / 8;
  if (px)
    hipLaunchKernelGGL((photo_bwd_kernel<NS, true>), dim3(a.blocks_per_xcd * 8), dim3(BS_WAVES * 64), 0, st, a);
  else
    hipLaunchKernelGGL((photo_bwd_kernel<NS, false>), dim3(a.blocks_per_xcd * 8), dim3(BS_WAVES * 64), 0, st, a);
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
// Adjoint of the bilinear up-sampling, separable gather form.  A block owns a tile of TJ x TI low-res pixels:
//   1. row pass: t[y][i] = sum_x wx(x, i) * (sum over the planes of d_up[y][x]) for the full-res rows y the tile's rows can
//      receive from, kept in LDS (at most (TJ + 2) * H/hs + 4 rows of TI floats);
//   2. column pass: d_disp[j][i] = sum_y wy(y, j) * t[y][i].
// Every full-res value is read once per tile row-range (1.25-1.5x in all) instead of once per low-res pixel it can reach
// (the earlier one-launch gather read each value ~9 times at factor 2 and took 35-44 us of the 96x320 level's 60 us
// backward).  Weights are ATen's (up_index), evaluated per tap; fixed summation order: deterministic.
template <int TJ, int TI>
__global__ __launch_bounds__(TD_THREADS) void upsample_adjoint_kernel(
    const float* __restrict__ d_up, int n_planes, int B, int H, int W, int hs, int ws, int tiles_i, int tiles_j,
    float* __restrict__ d_disp, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float t_rows[];
  const int ti = blockIdx.x % tiles_i, tj = (blockIdx.x / tiles_i) % tiles_j, b = blockIdx.x / (tiles_i * tiles_j);
  const int i0 = ti * TI, j0 = tj * TJ;
  const float ry = (float)hs / (float)H, rx = (float)ws / (float)W;
  const float fy = (float)H / (float)hs, fx = (float)W / (float)ws;
  int ylo = (int)floorf(((float)j0 - 0.5f) * fy - 0.5f) - 1;
  int yhi = (int)ceilf(((float)(j0 + TJ - 1) + 1.5f) * fy - 0.5f) + 1;
  ylo = ylo < 0 ? 0 : ylo;
  yhi = yhi > H - 1 ? H - 1 : yhi;
  const int nrows = yhi - ylo + 1;
  const size_t plane = (size_t)B * H * W;
  const float* src = d_up + (size_t)b * H * W;
  for (int item = threadIdx.x; item < nrows * TI; item += TD_THREADS) {
    const int yy = ylo + item / TI, ii = i0 + item % TI;
    float acc = 0.f;
    if (ii < ws) {
      int xlo = (int)floorf(((float)ii - 0.5f) * fx - 0.5f) - 1;
      int xhi = (int)ceilf(((float)ii + 1.5f) * fx - 0.5f) + 1;
      xlo = xlo < 0 ? 0 : xlo;
      xhi = xhi > W - 1 ? W - 1 : xhi;
      const float* row = src + (size_t)yy * W;
      for (int xx = xlo; xx <= xhi; ++xx) {
        const UpIdx vx = up_index(xx, rx, ws);
        const float wx = (vx.i0 == ii ? vx.l0 : 0.f) + (vx.i1 == ii ? vx.l1 : 0.f);
        if (wx != 0.f) {
          float v = row[xx];
          for (int p = 1; p < n_planes; ++p) v += row[(size_t)p * plane + xx];     // per-frame planes, in order
          acc += wx * v;
        }
      }
    }
    t_rows[item] = acc;
  }
  __syncthreads();
  for (int out = threadIdx.x; out < TJ * TI; out += TD_THREADS) {
    const int jj = j0 + out / TI, ii = i0 + out % TI;
    if (jj >= hs || ii >= ws) continue;
    int y0 = (int)floorf(((float)jj - 0.5f) * fy - 0.5f) - 1;
    int y1 = (int)ceilf(((float)jj + 1.5f) * fy - 0.5f) + 1;
    y0 = y0 < ylo ? ylo : y0;
    y1 = y1 > yhi ? yhi : y1;
    float acc = 0.f;
    for (int yy = y0; yy <= y1; ++yy) {
      const UpIdx vy = up_index(yy, ry, hs);
      const float wy = (vy.i0 == jj ? vy.l0 : 0.f) + (vy.i1 == jj ? vy.l1 : 0.f);
      if (wy != 0.f) acc += wy * t_rows[(yy - ylo) * TI + (ii - i0)];
    }
    float* dst = d_disp + ((size_t)b * hs + jj) * ws + ii;
    if (accumulate) *dst += acc; else *dst = acc;
  }
}

template <int TJ, int TI>
static int launch_upsample_adjoint(const float* d_up, int n_planes, int B, int H, int W, int hs, int ws, float* d_disp, int accumulate,
                                   hipStream_t st) {
  const int tiles_i = (ws + TI - 1) / TI, tiles_j = (hs + TJ - 1) / TJ;
  const int max_rows = (int)((TJ + 2) * ((double)H / hs)) + 6;
  const size_t lds = (size_t)max_rows * TI * sizeof(float);
  if (lds > 64 * 1024) return TD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((upsample_adjoint_kernel<TJ, TI>), dim3(tiles_i * tiles_j * B), dim3(TD_THREADS), lds, st, d_up, n_planes, B, H, W, hs,
                     ws, tiles_i, tiles_j, d_disp, accumulate);
  return record_launch_error(hipGetLastError(), "td_upsample_adjoint");
}

// Large factors (> 10: the 12x40 level of a 192x640 image): G lanes cooperate on one low-res pixel's whole window; there the
// tiled form above has too few tiles to cover the chip (30 us against 21 us at factor 16).
template <int G>
__global__ __launch_bounds__(TD_THREADS) void upsample_adjoint_gather_kernel(
    const float* __restrict__ d_up, int n_planes, int B, int H, int W, int hs, int ws,
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
    if (w != 0.f) {
      float v = src[(size_t)yy * W + xx];
      for (int p = 1; p < n_planes; ++p) v += src[(size_t)p * B * H * W + (size_t)yy * W + xx];
      acc += w * v;
    }
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, G);
  if (live && sub == 0) {
    if (accumulate) d_disp[id] += acc; else d_disp[id] = acc;
  }
}

}  // namespace td

extern "C" int td_photo_bwd(const float* tgt, const float* const* src, const float* tgt_rgbx, const float* const* src_rgbx, int n_src, const float* disp,
                            const float* P, const float* invK, const uint8_t* argmin, const float* coef,
                            int automask, const float* gscale, float inv_count, int B, int H, int W, int hs, int ws,
                            float min_depth, float max_depth, float* d_up, float* dP_partial,
                            td_stream_t stream) {
  if (!tgt || !src || !disp || !P || !invK || !argmin || !coef || !gscale || !d_up || !dP_partial) return TD_ERR_BAD_ARG;
  if (n_src < 1 || n_src > TD_MAX_SRC || B <= 0 || hs <= 0 || ws <= 0 || hs > H || ws > W) return TD_ERR_BAD_ARG;
  if (!(min_depth > 0.f) || !(max_depth > min_depth)) return TD_ERR_BAD_ARG;
  for (int i = 0; i < n_src; ++i) if (!src[i]) return TD_ERR_BAD_ARG;
  if ((tgt_rgbx == nullptr) != (src_rgbx == nullptr)) return TD_ERR_BAD_ARG;
  if (src_rgbx) for (int i = 0; i < n_src; ++i) if (!src_rgbx[i]) return TD_ERR_BAD_ARG;
  if (H < 3 || W < 3 || (long long)B * 4 * H * W >= (1ll << 31)) return TD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  switch (n_src) {
    case 1: return td::run_bwd<1>(tgt, src, tgt_rgbx, src_rgbx, disp, P, invK, argmin, coef, automask, gscale, inv_count, B, H, W, hs, ws, min_depth, max_depth, d_up, dP_partial, st);
    case 2: return td::run_bwd<2>(tgt, src, tgt_rgbx, src_rgbx, disp, P, invK, argmin, coef, automask, gscale, inv_count, B, H, W, hs, ws, min_depth, max_depth, d_up, dP_partial, st);
    case 3: return td::run_bwd<3>(tgt, src, tgt_rgbx, src_rgbx, disp, P, invK, argmin, coef, automask, gscale, inv_count, B, H, W, hs, ws, min_depth, max_depth, d_up, dP_partial, st);
    case 4: return td::run_bwd<4>(tgt, src, tgt_rgbx, src_rgbx, disp, P, invK, argmin, coef, automask, gscale, inv_count, B, H, W, hs, ws, min_depth, max_depth, d_up, dP_partial, st);
  }
  return TD_ERR_BAD_ARG;
}

extern "C" int td_photo_bwd_num_blocks(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  int ns, nc, rows;
  return td::bwd_tasks(B, H, W, &ns, &nc, &rows);
}

extern "C" int td_reduce_dP(const float* dP_partial, int n_src, int B, int H, int W, float* dP,
                            td_stream_t stream) {
  if (!dP_partial || !dP || n_src < 1 || n_src > TD_MAX_SRC || B <= 0 || H <= 0 || W <= 0) return TD_ERR_BAD_ARG;
  int ns, nc, rows;
  td::bwd_tasks(B, H, W, &ns, &nc, &rows);        // the task tiling depends on the whole launch shape
  const int bps = ns * nc;
  hipLaunchKernelGGL(td::reduce_dP_kernel, dim3(n_src * 12, B), dim3(64), 0, (hipStream_t)stream,
                     dP_partial, n_src, B, bps, dP);
  return td::record_launch_error(hipGetLastError(), "td_reduce_dP");
}

extern "C" int td_reduce_partials(const float* partial, int n_src, int B, int blocks_per_sample, float* dP,
                                  td_stream_t stream) {
  if (!partial || !dP || n_src < 1 || n_src > TD_MAX_SRC || B <= 0 || blocks_per_sample <= 0) return TD_ERR_BAD_ARG;
  hipLaunchKernelGGL(td::reduce_dP_kernel, dim3(n_src * 12, B), dim3(64), 0, (hipStream_t)stream, partial, n_src, B,
                     blocks_per_sample, dP);
  return td::record_launch_error(hipGetLastError(), "td_reduce_partials");
}

extern "C" int td_upsample_adjoint_planes(const float* d_up, int n_planes, int B, int H, int W, int hs, int ws, float* d_disp,
                                          int accumulate, td_stream_t stream);

extern "C" int td_upsample_adjoint(const float* d_up, int B, int H, int W, int hs, int ws,
                                   float* d_disp, int accumulate, td_stream_t stream) {
  return td_upsample_adjoint_planes(d_up, 1, B, H, W, hs, ws, d_disp, accumulate, stream);
}

extern "C" int td_upsample_adjoint_planes(const float* d_up, int n_planes, int B, int H, int W, int hs, int ws, float* d_disp,
                                          int accumulate, td_stream_t stream) {
  if (!d_up || !d_disp || B <= 0 || hs <= 0 || ws <= 0 || hs > H || ws > W || n_planes < 1 || n_planes > TD_MAX_SRC) return TD_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const float factor = fmaxf((float)H / (float)hs, (float)W / (float)ws);
  if (factor > 10.f) {
    const int blocks = (B * hs * ws * 64 + TD_THREADS - 1) / TD_THREADS;
    hipLaunchKernelGGL(td::upsample_adjoint_gather_kernel<64>, dim3(blocks), dim3(TD_THREADS), 0, st, d_up, n_planes, B, H, W, hs, ws, d_disp,
                       accumulate);
    return td::record_launch_error(hipGetLastError(), "td_upsample_adjoint");
  }
  // tile of low-res pixels per block: large maps take 8 x 32, small ones 4 x 8 so that the grid still covers the chip
  if ((long long)B * hs * ws >= 64 * 1024) return td::launch_upsample_adjoint<8, 32>(d_up, n_planes, B, H, W, hs, ws, d_disp, accumulate, st);
  return td::launch_upsample_adjoint<4, 8>(d_up, n_planes, B, H, W, hs, ws, d_disp, accumulate, st);
}
