// Fused photometric forward: upsample(disp) -> depth -> back-project -> project -> bilinear
// gather -> 3x3 reflect-padded SSIM + robust L1 -> per-pixel min over {identity, warped}.
//
// Streaming formulation, no LDS and no barriers: one wave owns a strip of 62 output columns
// (lanes 1..62; lanes 0 and 63 carry the left/right halo column) and marches down R output
// rows.  Horizontal 3-sums of the SSIM window come from the neighbouring lanes (DPP wave
// shifts), vertical 3-sums from a two-row register ring.  The row loop is software-pipelined
// two rows deep: while row r is reduced, the gathers of row r+1 and the low-res disparity
// taps of row r+2 are in flight, so a wave keeps ~30 loads outstanding without relying on
// occupancy.  Blocks are remapped so that consecutive tasks (neighbouring strips / row chunks
// that share halo lines) land on the same XCD and hit its L2.
//
// SPLIT form (built, parity-tested, NOT dispatched: kSplitFrames below): the frames of a strip as separate waves of one
// block.  A one-frame wave needs 148 registers instead of 215 (three waves per SIMD instead of two); each wave reduces its
// own frame to the per-pixel loss, the losses meet in a 2 KB LDS exchange (one barrier per row, slots alternate by row
// parity), every wave forms the same arg-min, wave 0 writes it, and each wave writes the SSIM-adjoint coefficients where ITS
// frame won (wave 0 also the zeros where an identity term won).  Measured at B=12 192x640 scale 0: 68.8 us against 62.1 us
// for the frames-in-one-wave form -- the target row's window sums, the auto-mask loads and the selection are done once per
// frame wave instead of once per strip (+25 % horizontal sums) and the barrier couples the pair; the extra resident wave
// does not pay for that.  (The backward, whose frames share no arithmetic, gained 11 % from the same split.)
#include <type_traits>

#include "td_common.h"

namespace td {

// output rows per wave task: chosen per shape by pick_rows() (PhotoFwdArgs::rows, always even)
constexpr int FS_COLS = 62;       // output columns per wave task
constexpr int FS_WAVES = 4;       // independent wave tasks per 256-thread block

__device__ __forceinline__ void store_coef_px(float* __restrict__ coef, int b, unsigned plane, unsigned pix, const float* cv) {
#pragma unroll
  for (int i = 0; i < 9; ++i) coef[((size_t)b * 9 + i) * plane + pix] = cv[i];      // nine fully coalesced planes [B,9,H,W]
}

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
  float* pack_tgt;     // identity mode only: RGBX copies of the frames it reads anyway (the per-scale kernels' input format)
  float* pack_src[NS];
  float* coef;         // [B,9,H,W] SSIM-adjoint coefficients of the selected warped frame (nullable)
  int B, H, W, hs, ws;
  int nstrips, nchunks, ntasks, blocks_per_xcd, rows;
  float min_disp, disp_range;
};

// per-row vertical source rows of the bilinear up-sampling (wave-uniform)
struct UpRow {
  int o0, o1;       // element offsets of the two low-res rows
  float l0, l1;
};

// Everything one pipeline row needs, requested one iteration ahead of its use.
template <int NF, int NS>     // NF: frames this wave warps (1 in the split form), NS: all source frames (auto-mask terms)
struct RowLoads {
  Tap taps[NF];
  TapVals4 tv[NF];     // the four RGBX taps of every warped frame
  f4 yv;               // target pixel (RGBX)
  f4 xv[NF];           // identity mode: raw source pixels
  float idv[NS];       // auto-mask term of the output row this pipeline row completes ([B,H,W,NS] interleaved)
  float nz[NS];
};

struct DispTaps {
  float v[4];
  UpRow ur;
};

// MODE: 0 = identity-term kernel, 1 = warped terms only, 2 = + auto-mask, 3 = + auto-mask noise.
// KEEP: also write the warped sources.
template <int NS, int MODE, bool KEEP, bool COEF, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? (FS_WAVES / NS) * NS * 64 : FS_WAVES * 64, SPLIT ? 3 : 1) void photo_fwd_kernel(const PhotoFwdArgs<NS> a) {
  constexpr bool IDENT = MODE == 0;
  static_assert(!SPLIT || (NS >= 2 && MODE >= 1), "the split form is for the warped terms of >= 2 frames");
  constexpr int NF = SPLIT ? 1 : NS;                 // frames this wave reduces
  constexpr int SPB = SPLIT ? FS_WAVES / NS : FS_WAVES;   // strip tasks per block
  __shared__ float xch[SPLIT ? 2 * SPB * NS * 64 : 1];    // [row parity][strip of the block][frame][lane]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int fw = SPLIT ? wave % NS : 0;              // this wave's frame (split form)
  const int slot = SPLIT ? wave / NS : wave;         // strip task of the block
  // XCD-aware remap: hardware deals consecutive blocks round-robin over the 8 XCDs; give every
  // XCD a contiguous range of tasks instead (speed only, any placement is correct)
  const int bid = (int)(blockIdx.x & 7) * a.blocks_per_xcd + (int)(blockIdx.x >> 3);
  int task = bid * SPB + slot;
  bool live = task < a.ntasks;
  if (!SPLIT && !live) return;
  task = live ? task : a.ntasks - 1;                 // split form: every wave keeps marching (barriers); stores are masked
  const int strip = task % a.nstrips;
  const int chunk = (task / a.nstrips) % a.nchunks;
  const int b = task / (a.nstrips * a.nchunks);

  const int H = a.H, W = a.W;
  const unsigned plane = (unsigned)(H * W);
  const int x = strip * FS_COLS - 1 + lane;          // padded-domain column of this lane
  const int qx = reflect1(x, W);
  const int y0 = chunk * a.rows;
  const bool col_out = live && lane >= 1 && lane <= FS_COLS && x < W;
  const int xo = x < W ? (x < 0 ? 0 : x) : W - 1;    // clamped column for prefetching per-pixel inputs

  // the per-scale kernels read RGBX frames [B,H,W,4]; the identity-term kernel reads the NCHW frames and WRITES the RGBX copies
  const float* tgtb = a.tgt + (size_t)b * (IDENT ? 3 : 4) * plane;
  const float* srcb[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const float* p = a.src[SPLIT ? 0 : f];
    if (SPLIT) {
#pragma unroll
      for (int i = 1; i < NS; ++i) p = (fw == i) ? a.src[i] : p;
    }
    srcb[f] = p + (size_t)b * (IDENT ? 3 : 4) * plane;
  }

  // ---- per-wave constants: camera, per-lane horizontal up-sampling taps, x part of the rays ----
  float ik[9], P[NF][12];
  UpIdx ux;
  ux.i0 = ux.i1 = 0; ux.l0 = ux.l1 = 0.f;
  float rx0 = 0.f, rx1 = 0.f, rx2 = 0.f;
  const float* dispb = nullptr;
  float ratio_y = 0.f;
  if (!IDENT) {
#pragma unroll
    for (int i = 0; i < 9; ++i) ik[i] = a.invK[b * 16 + (i / 3) * 4 + (i % 3)];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int e = 0; e < 12; ++e) P[f][e] = a.P[((SPLIT ? fw : f) * a.B + b) * 12 + e];
    ux = up_index(qx, (float)a.ws / (float)W, a.ws);
    ratio_y = (float)a.hs / (float)H;
    const float fx = (float)qx;
    rx0 = ik[0] * fx; rx1 = ik[3] * fx; rx2 = ik[6] * fx;
    dispb = a.disp + (size_t)b * a.hs * a.ws;
  }

  auto stage_a = [&](int k, DispTaps& d) {           // issue the 4 disparity loads of pipeline row k
    const int qy = reflect1(y0 - 1 + k, H);
    const UpIdx uy = up_index(qy, ratio_y, a.hs);
    d.ur.o0 = uy.i0 * a.ws; d.ur.o1 = uy.i1 * a.ws; d.ur.l0 = uy.l0; d.ur.l1 = uy.l1;
    d.v[0] = ld_at(dispb, (unsigned)(d.ur.o0 + ux.i0) * 4u); d.v[1] = ld_at(dispb, (unsigned)(d.ur.o0 + ux.i1) * 4u);
    d.v[2] = ld_at(dispb, (unsigned)(d.ur.o1 + ux.i0) * 4u); d.v[3] = ld_at(dispb, (unsigned)(d.ur.o1 + ux.i1) * 4u);
  };
  auto stage_b = [&](int k, const DispTaps& d, RowLoads<NF, NS>& L) {   // taps + gathers of pipeline row k
    const int qy = reflect1(y0 - 1 + k, H);
    const unsigned off = (unsigned)(qy * W + qx);
    if (IDENT) {
#pragma unroll
      for (int c = 0; c < 3; ++c) L.yv[c] = ld_at(tgtb + (size_t)c * plane, off * 4u);
      L.yv[3] = 0.f;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) L.xv[f][c] = ld_at(srcb[f] + (size_t)c * plane, off * 4u);
        L.xv[f][3] = 0.f;
      }
    } else {
      L.yv = ld4_at(tgtb, off * 16u);
      const float dd = d.ur.l0 * (ux.l0 * d.v[0] + ux.l1 * d.v[1]) + d.ur.l1 * (ux.l0 * d.v[2] + ux.l1 * d.v[3]);
      const float depth = fast_rcp(a.min_disp + a.disp_range * dd);
      const float fy = (float)qy;
      const float r0 = rx0 + ik[1] * fy + ik[2];
      const float r1 = rx1 + ik[4] * fy + ik[5];
      const float r2 = rx2 + ik[7] * fy + ik[8];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        float pt[3], cz[3];
        L.taps[f] = project_ray(r0, r1, r2, P[f], depth, W, H, pt, cz);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) L.tv[f] = load_taps4(srcb[f], W, L.taps[f]);
      if (MODE >= 2) {        // per-pixel inputs of the output row that pipeline row k completes
        int orow = y0 - 2 + k;
        orow = orow < 0 ? 0 : (orow > H - 1 ? H - 1 : orow);
        const unsigned pix = (unsigned)(orow * W + xo);
        // idloss is [B, H, W, NS]: the NS terms of a pixel are adjacent (one 8 / 16-byte load for two / four frames)
        const float* idp = a.idloss + ((size_t)b * plane + pix) * NS;
        if (NS == 2) {
          const float2 v = *reinterpret_cast<const float2*>(idp);
          L.idv[0] = v.x; L.idv[NS > 1 ? 1 : 0] = v.y;
        } else if (NS == 4) {
          const f4 v = *reinterpret_cast<const f4*>(idp);
#pragma unroll
          for (int f = 0; f < NS; ++f) L.idv[f] = v[f & 3];
        } else {
#pragma unroll
          for (int f = 0; f < NS; ++f) L.idv[f] = idp[f];
        }
        if (MODE >= 3) {
#pragma unroll
          for (int f = 0; f < NS; ++f) L.nz[f] = ld_at(a.noise + (size_t)(f * a.B + b) * plane, pix * 4u);
        }
      }
    }
  };

  // two-row ring of horizontal sums + centre values of the previous row
  float p_hy[2][3], p_hyy[2][3], p_hx[2][NF][3], p_hxx[2][NF][3], p_hxy[2][NF][3];
  float c_y[3], c_x[NF][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    p_hy[0][c] = p_hy[1][c] = p_hyy[0][c] = p_hyy[1][c] = c_y[c] = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f)
      p_hx[0][f][c] = p_hx[1][f][c] = p_hxx[0][f][c] = p_hxx[1][f][c] = p_hxy[0][f][c] = p_hxy[1][f][c] = c_x[f][c] = 0.f;
  }
  float acc = 0.f;

  // consume pipeline row k from `cur`; emit_allowed: a full 3-row window is available.  The two-row rings are not
  // shifted: ring slot RO = k & 1 holds the older row and receives this row's sums (the loop is unrolled by two, so
  // RO is a compile-time constant).  In the backward this form removed 40 register moves per row; here the
  // compiler had already coalesced the shifts of the unrolled loop (same instruction count either way)
  auto consume = [&](auto ro_tag, int k, const RowLoads<NF, NS>& cur, bool emit_allowed) {
    constexpr int RO = decltype(ro_tag)::value, RN = 1 - RO;
    float y[3], xw[NF][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = cur.yv[c];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int c = 0; c < 3; ++c) xw[f][c] = IDENT ? cur.xv[f][c] : blend_taps(tap_channel(cur.tv[f], c), cur.taps[f]);
    const int r = y0 - 1 + k;
    if (KEEP && !IDENT && k >= 1 && k <= a.rows && r < H && col_out) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          a.warped[(((size_t)(SPLIT ? fw : f) * a.B + b) * 3 + c) * plane + (unsigned)(r * W + x)] = xw[f][c];
    }
    float ss[NF], l1[NF];
    float own_y[3], own_x[NF][3];                    // the output row's OWN pixels (centre of the window), for the RGBX copies
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      own_y[c] = c_y[c];
#pragma unroll
      for (int f = 0; f < NF; ++f) own_x[f][c] = c_x[f][c];
    }
    float cfs[COEF ? NF : 1][9];                     // (alpha, beta, gamma) per channel of every warped frame of this wave
#pragma unroll
    for (int f = 0; f < NF; ++f) { ss[f] = 0.f; l1[f] = 0.f; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float hy = hsum3(y[c]);
      const float hyy = hsum3(y[c] * y[c]);
      float hx[NF], hxx[NF], hxy[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        hx[f] = hsum3(xw[f][c]);
        hxx[f] = hsum3(xw[f][c] * xw[f][c]);
        hxy[f] = hsum3(xw[f][c] * y[c]);
      }
      if (emit_allowed) {
        const float sy = p_hy[RO][c] + p_hy[RN][c] + hy;
        const float syy = p_hyy[RO][c] + p_hyy[RN][c] + hyy;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const float sx = p_hx[RO][f][c] + p_hx[RN][f][c] + hx[f];
          const float sxx = p_hxx[RO][f][c] + p_hxx[RN][f][c] + hxx[f];
          const float sxy = p_hxy[RO][f][c] + p_hxy[RN][f][c] + hxy[f];
          if (COEF) {
            float al, be, ga;
            ss[f] += ssim_with_adjoint(sx, sy, sxx, syy, sxy, al, be, ga);
            cfs[COEF ? f : 0][c * 3 + 0] = al; cfs[COEF ? f : 0][c * 3 + 1] = be; cfs[COEF ? f : 0][c * 3 + 2] = ga;
          } else {
            ss[f] += ssim_from_sums(sx, sy, sxx, syy, sxy);
          }
          const float df = c_y[c] - c_x[f][c];
          l1[f] += fast_sqrt(df * df + TD_L1_EPS2);
        }
      }
      p_hy[RO][c] = hy;
      p_hyy[RO][c] = hyy;
      c_y[c] = y[c];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        p_hx[RO][f][c] = hx[f];
        p_hxx[RO][f][c] = hxx[f];
        p_hxy[RO][f][c] = hxy[f];
        c_x[f][c] = xw[f][c];
      }
    }
    const int orow = r - 1;
    float loss[NS];                                  // the per-pixel loss of EVERY warped frame
#pragma unroll
    for (int f = 0; f < NF; ++f) loss[SPLIT ? 0 : f] = 0.85f * (ss[f] * (1.f / 3.f)) + 0.15f * (l1[f] * (1.f / 3.f));
    if (SPLIT && emit_allowed) {
      // the frames of this strip meet here: one slot set per row parity, one barrier per row (every wave of the block runs
      // the same row sequence, so the barrier is block-uniform; a slot is rewritten two rows later, behind the next barrier)
      float* xs = xch + ((RO * SPB + slot) * NS) * 64;
      xs[fw * 64 + lane] = loss[0];
      __syncthreads();
#pragma unroll
      for (int f = 0; f < NS; ++f) loss[f] = xs[f * 64 + lane];
    }
    if (emit_allowed && orow < H && col_out) {
      const unsigned pix = (unsigned)(orow * W + x);
      if (IDENT) {
        if (a.pack_tgt != nullptr) {
          // the centre values of the window are this output row's own pixels: the RGBX copies cost no extra read
          f4 v; v[0] = own_y[0]; v[1] = own_y[1]; v[2] = own_y[2]; v[3] = 0.f;
          *reinterpret_cast<f4*>(a.pack_tgt + ((size_t)b * plane + pix) * 4) = v;
#pragma unroll
          for (int f = 0; f < NS; ++f) {
            f4 u; u[0] = own_x[f][0]; u[1] = own_x[f][1]; u[2] = own_x[f][2]; u[3] = 0.f;
            *reinterpret_cast<f4*>(a.pack_src[f] + ((size_t)b * plane + pix) * 4) = u;
          }
        }
        float* idp = a.idloss_out + ((size_t)b * plane + pix) * NS;      // [B, H, W, NS]
        if (NS == 2) {
          *reinterpret_cast<float2*>(idp) = make_float2(loss[0], loss[NS > 1 ? 1 : 0]);
        } else {
#pragma unroll
          for (int f = 0; f < NS; ++f) idp[f] = loss[f];
        }
      } else {
        float best = 0.f;
        int idx = 0;
        bool have = false;
        if (MODE >= 2) {
#pragma unroll
          for (int f = 0; f < NS; ++f) {
            const float v = MODE >= 3 ? cur.idv[f] + cur.nz[f] * 1e-5f : cur.idv[f];
            if (!have || v < best) { best = v; idx = f; have = true; }
          }
        }
        constexpr int base = (MODE >= 2) ? NS : 0;
#pragma unroll
        for (int f = 0; f < NS; ++f)
          if (!have || loss[f] < best) { best = loss[f]; idx = base + f; have = true; }
        if (!SPLIT || fw == 0) {
          a.argmin[(size_t)b * plane + pix] = (uint8_t)idx;
          if (a.min_map != nullptr) a.min_map[(size_t)b * plane + pix] = best;
          acc += best;
        }
        if (COEF) {
          if (SPLIT) {
            // one writer per pixel: the wave whose frame won; wave 0 writes the zeros of an identity win
            const bool won = idx == base + fw;
            if (won || (fw == 0 && idx < base)) {
              float cv[9];
#pragma unroll
              for (int i = 0; i < 9; ++i) cv[i] = won ? cfs[0][i] : 0.f;
              store_coef_px(a.coef, b, plane, pix, cv);
            }
          } else {
            float cv[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
              float v = 0.f;
#pragma unroll
              for (int f = 0; f < NF; ++f) v = (idx == base + f) ? cfs[COEF ? f : 0][i] : v;
              cv[i] = v;
            }
            store_coef_px(a.coef, b, plane, pix, cv);
          }
        }
      }
    }
  };

  // ---- two-deep software pipeline over NK = rows + 2 rows (even, >= 6: the loop is unrolled by two), ping-pong register sets ----
  const int NK = a.rows + 2;
  RowLoads<NF, NS> LA, LB;
  DispTaps DA, DB;
  if (!IDENT) {
    stage_a(0, DA);
    stage_a(1, DB);
  }
  stage_b(0, DA, LA);
  // prologue: rows 0 and 1 only fill the window
  if (!IDENT) stage_a(2, DA);
  stage_b(1, DB, LB);
  constexpr std::integral_constant<int, 0> even{};
  constexpr std::integral_constant<int, 1> odd{};
  consume(even, 0, LA, false);
  if (!IDENT) stage_a(3, DB);
  stage_b(2, DA, LA);
  consume(odd, 1, LB, false);
#pragma unroll 1
  for (int k = 2; k < NK - 2; k += 2) {
    if (!IDENT) stage_a(k + 2, DA);
    stage_b(k + 1, DB, LB);
    consume(even, k, LA, true);
    if (!IDENT) stage_a(k + 3, DB);      // rows >= NK are clamped duplicates, never consumed
    stage_b(k + 2, DA, LA);
    consume(odd, k + 1, LB, true);
  }
  stage_b(NK - 1, DB, LB);
  consume(even, NK - 2, LA, true);
  consume(odd, NK - 1, LB, true);

  if (!IDENT && (!SPLIT || fw == 0)) {
    const float tot = wave_sum(acc);
    if (lane == 0 && live) a.partial[task] = tot;
  }
}

// Strip tasks.  The tiling is a function of (B, H, W) only (td_photo_num_blocks sizes `partial` without knowing the frame
// count); with kSplitFrames the warped-term kernels would be tiled for two frame waves per strip at three waves per SIMD.
constexpr bool kSplitFrames = false;

static int fwd_tasks(int B, int H, int W, int* nstrips, int* nchunks, int* rows, bool ident = false) {
  *nstrips = (W + FS_COLS - 1) / FS_COLS;
  *rows = (ident || !kSplitFrames) ? pick_rows(B * (*nstrips), H, 2, 2, 8, 64)
                                   : pick_rows(B * (*nstrips) * 2, H, 2, 2, 8, 64, 256 * 4 * 3);   // even
  *nchunks = (H + *rows - 1) / *rows;
  return B * (*nstrips) * (*nchunks);
}

template <int NS, int MODE, bool KEEP, bool COEF>
static int launch_fwd(PhotoFwdArgs<NS>& a, hipStream_t st) {
  a.ntasks = fwd_tasks(a.B, a.H, a.W, &a.nstrips, &a.nchunks, &a.rows, MODE == 0);
  constexpr bool SPLIT = kSplitFrames && NS >= 2 && MODE >= 1;
  constexpr int SPB = SPLIT ? FS_WAVES / NS : FS_WAVES;          // strip tasks per block
  const int blocks = (a.ntasks + SPB - 1) / SPB;
  a.blocks_per_xcd = (blocks + 7) / 8;
  hipLaunchKernelGGL((photo_fwd_kernel<NS, MODE, KEEP, COEF, SPLIT>), dim3(a.blocks_per_xcd * 8),
                     dim3(SPLIT ? SPB * NS * 64 : FS_WAVES * 64), 0, st, a);
  return record_launch_error(hipGetLastError(), MODE == 0 ? "td_photo_identity" : "td_photo_fwd");
}

template <int NS>
static int run_fwd(const float* tgt, const float* const* src, const float* disp, const float* P,
                   const float* invK, const float* idloss, const float* noise, int B, int H, int W,
                   int hs, int ws, float min_depth, float max_depth, uint8_t* argmin, float* warped,
                   float* min_map, float* partial, float* idloss_out, float* coef, bool ident, hipStream_t st,
                   float* pack_tgt = nullptr, float* const* pack_src = nullptr) {
  PhotoFwdArgs<NS> a;
  a.tgt = tgt;
  for (int i = 0; i < NS; ++i) a.src[i] = src[i];
  a.disp = disp; a.P = P; a.invK = invK; a.idloss = idloss; a.noise = noise;
  a.argmin = argmin; a.warped = warped; a.min_map = min_map; a.partial = partial;
  a.idloss_out = idloss_out;
  a.pack_tgt = pack_src ? pack_tgt : nullptr;
  for (int i = 0; i < NS; ++i) a.pack_src[i] = pack_src ? pack_src[i] : nullptr;
  a.coef = coef;
  a.B = B; a.H = H; a.W = W; a.hs = hs; a.ws = ws;
  const double lo = 1.0 / (double)max_depth, hi = 1.0 / (double)min_depth;
  a.min_disp = (float)lo;
  a.disp_range = (float)(hi - lo);
  if (ident) return launch_fwd<NS, 0, false, false>(a, st);
  const int mode = idloss == nullptr ? 1 : (noise == nullptr ? 2 : 3);
  const bool keep = warped != nullptr;
  if (a.coef != nullptr && !keep) {        // training step: emit the adjoint coefficients
    switch (mode) {
      case 1: return launch_fwd<NS, 1, false, true>(a, st);
      case 2: return launch_fwd<NS, 2, false, true>(a, st);
      default: return launch_fwd<NS, 3, false, true>(a, st);
    }
  }
  if (a.coef != nullptr) {
    switch (mode) {
      case 1: return launch_fwd<NS, 1, true, true>(a, st);
      case 2: return launch_fwd<NS, 2, true, true>(a, st);
      default: return launch_fwd<NS, 3, true, true>(a, st);
    }
  }
  switch (mode * 2 + (keep ? 1 : 0)) {
    case 2: return launch_fwd<NS, 1, false, false>(a, st);
    case 3: return launch_fwd<NS, 1, true, false>(a, st);
    case 4: return launch_fwd<NS, 2, false, false>(a, st);
    case 5: return launch_fwd<NS, 2, true, false>(a, st);
    case 6: return launch_fwd<NS, 3, false, false>(a, st);
    default: return launch_fwd<NS, 3, true, false>(a, st);
  }
}

}  // namespace td

extern "C" int td_photo_num_blocks(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  int ns, nc, rows;
  return td::fwd_tasks(B, H, W, &ns, &nc, &rows);
}

static int dispatch_fwd(const float* tgt, const float* const* src, int n_src, const float* disp,
                        const float* P, const float* invK, const float* idloss, const float* noise,
                        int B, int H, int W, int hs, int ws, float min_depth, float max_depth,
                        uint8_t* argmin, float* warped, float* min_map, float* partial,
                        float* idloss_out, float* coef, bool ident, td_stream_t stream, float* pack_tgt = nullptr,
                        float* const* pack_src = nullptr) {
  hipStream_t st = (hipStream_t)stream;
  switch (n_src) {
    case 1: return td::run_fwd<1>(tgt, src, disp, P, invK, idloss, noise, B, H, W, hs, ws, min_depth, max_depth, argmin, warped, min_map, partial, idloss_out, coef, ident, st, pack_tgt, pack_src);
    case 2: return td::run_fwd<2>(tgt, src, disp, P, invK, idloss, noise, B, H, W, hs, ws, min_depth, max_depth, argmin, warped, min_map, partial, idloss_out, coef, ident, st, pack_tgt, pack_src);
    case 3: return td::run_fwd<3>(tgt, src, disp, P, invK, idloss, noise, B, H, W, hs, ws, min_depth, max_depth, argmin, warped, min_map, partial, idloss_out, coef, ident, st, pack_tgt, pack_src);
    case 4: return td::run_fwd<4>(tgt, src, disp, P, invK, idloss, noise, B, H, W, hs, ws, min_depth, max_depth, argmin, warped, min_map, partial, idloss_out, coef, ident, st, pack_tgt, pack_src);
  }
  return TD_ERR_BAD_ARG;
}

extern "C" int td_photo_identity(const float* tgt, const float* const* src, int n_src, int B, int H,
                                 int W, float* idloss, float* tgt_rgbx, float* const* src_rgbx, td_stream_t stream) {
  if (!tgt || !src || !idloss || n_src < 1 || n_src > TD_MAX_SRC || B <= 0) return TD_ERR_BAD_ARG;
  for (int i = 0; i < n_src; ++i) if (!src[i]) return TD_ERR_BAD_ARG;
  if ((tgt_rgbx == nullptr) != (src_rgbx == nullptr)) return TD_ERR_BAD_ARG;
  if (src_rgbx) for (int i = 0; i < n_src; ++i) if (!src_rgbx[i]) return TD_ERR_BAD_ARG;
  if (H < 3 || W < 3 || (long long)B * 3 * H * W >= (1ll << 31)) return TD_ERR_UNSUPPORTED;
  return dispatch_fwd(tgt, src, n_src, nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, 1, 1,
                      0.1f, 100.f, nullptr, nullptr, nullptr, nullptr, idloss, nullptr, true, stream, tgt_rgbx, src_rgbx);
}

extern "C" int td_photo_fwd(const float* tgt, const float* const* src, int n_src, const float* disp,
                            const float* P, const float* invK, const float* idloss, const float* noise,
                            int B, int H, int W, int hs, int ws, float min_depth, float max_depth,
                            uint8_t* argmin, float* warped, float* min_map, float* partial, float* coef,
                            td_stream_t stream) {
  if (!tgt || !src || !disp || !P || !invK || !argmin || !partial) return TD_ERR_BAD_ARG;
  if (n_src < 1 || n_src > TD_MAX_SRC || B <= 0 || hs <= 0 || ws <= 0 || hs > H || ws > W) return TD_ERR_BAD_ARG;
  if (!(min_depth > 0.f) || !(max_depth > min_depth)) return TD_ERR_BAD_ARG;
  for (int i = 0; i < n_src; ++i) if (!src[i]) return TD_ERR_BAD_ARG;
  if (H < 3 || W < 3 || (long long)B * 3 * H * W >= (1ll << 31)) return TD_ERR_UNSUPPORTED;
  return dispatch_fwd(tgt, src, n_src, disp, P, invK, idloss, noise, B, H, W, hs, ws, min_depth,
                      max_depth, argmin, warped, min_map, partial, nullptr, coef, false, stream);
}
