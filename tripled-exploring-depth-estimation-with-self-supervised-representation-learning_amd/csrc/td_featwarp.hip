// Feature-metric term (reference: generate_features_pred, mono/model/mono_fm_joint/net.py:196-223, +
// compute_perceptional_loss :63-65 + the min over source frames, mono_fm_joint_inpaint/net.py:58-70):
// warp the C-channel stem features of every source frame into the target view with the predicted
// depth and pose (bilinear, border), take mean_c sqrt((tgt - warped)^2 + 1e-6) per frame and the
// per-pixel minimum over frames.  Fused: the warped feature maps ([B,64,H/2,W/2] per frame, the largest
// tensors of the step) are never materialised.
// Layout: features channels-last (NHWC, bf16 or f32).  8 lanes cooperate on one pixel, 8 channels each
// (16-byte loads), so a wave covers 8 pixels x 64 channels per chunk.
// Backward: d/d(disparity), d/dP, d/d(target features) are written directly; d/d(source features) is a
// true scatter (a source pixel may receive from any target pixel).  It accumulates with INTEGER atomics on a fixed-point image of
// the contributions (round 4: float atomics made this the one hand-written kernel whose result depended on the arrival order):
// every contribution is bounded by |g| = |gscale| * inv_count / C (robust-L1 slope <= 1, bilinear weight <= 1), so it is
// rounded to a multiple of |g| / 2^14 and added as int32 -- integer addition is associative, the sum is bit-reproducible, the
// quantum is 6e-5 of the largest possible term (the result is rounded to bf16, 4e-3, afterwards) and 131072 full-size terms fit
// in one texel: border padding sends EVERY out-of-range sample to the border texels, so a corner can collect a large part of a
// feature map (160 x 512 = 81920 pixels at the largest configuration); 2^20 wrapped there.
// The atomics are issued as 256-byte contiguous wave-instructions (one pixel x 64 channels) through a small LDS transpose;
// td_featwarp_dsrc_finish scales the accumulators back into the feature dtype.
#include <hip/hip_bf16.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

template <typename T, int NS>
struct FeatWarpArgs {
  const T* tgt;
  const T* src[NS];
  const float* disp;
  const float* P;        // [NS,B,3,4] at feature resolution
  const float* invK;     // [B,4,4] at feature resolution
  uint8_t* argmin;       // [B,h,w]
  float* partial;        // fwd: per-block sums of the per-pixel minimum
  // backward
  const float* gscale;
  float inv_count;
  T* d_tgt;              // [B,h,w,C]
  int* d_src[NS];        // [B,h,w,C] int32 fixed point (units of |g| / 2^14), zero-initialised by the caller
  float* d_up;           // [B,h,w]
  float* dP_partial;     // [blocks, NS*12]
  int B, h, w, C, hs, ws;
  float min_disp, disp_range;
};

__device__ __forceinline__ float group8_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}

template <typename T>
__device__ __forceinline__ void tap_vec(const T* base, int C, int w, const Tap& t, int ch, float* nw, float* ne,
                                        float* sw, float* se) {
  load8(base + ((size_t)t.y0 * w + t.x0) * C + ch, nw);
  load8(base + ((size_t)t.y0 * w + t.x1) * C + ch, ne);
  load8(base + ((size_t)t.y1 * w + t.x0) * C + ch, sw);
  load8(base + ((size_t)t.y1 * w + t.x1) * C + ch, se);
}

template <typename T, int NS>
__global__ __launch_bounds__(TD_THREADS) void featwarp_fwd_kernel(const FeatWarpArgs<T, NS> a) {
  __shared__ float s_red[4];
  const int h = a.h, w = a.w, C = a.C;
  // blocks never straddle two samples: blockIdx = b * blocks_per_sample + j
  const int bps = (h * w * 8 + TD_THREADS - 1) / TD_THREADS;
  const int b = (int)blockIdx.x / bps;
  int lp = ((int)blockIdx.x % bps) * (TD_THREADS / 8) + (int)(threadIdx.x >> 3);   // pixel inside the sample
  const int sub = (int)(threadIdx.x & 7);
  const bool live = lp < h * w;
  if (!live) lp = h * w - 1;
  const int x = lp % w, y = lp / w;
  const long long pix = (long long)b * h * w + lp;
  const float d = upsample_disp(a.disp + (size_t)b * a.hs * a.ws, a.hs, a.ws, (float)a.hs / (float)h,
                                (float)a.ws / (float)w, y, x);
  const float depth = fast_rcp(a.min_disp + a.disp_range * d);
  float ik[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) ik[i] = a.invK[b * 16 + (i / 3) * 4 + (i % 3)];
  const T* tb = a.tgt + (size_t)pix * C;
  float best = 0.f;
  int idx = 0;
#pragma unroll
  for (int f = 0; f < NS; ++f) {
    float P[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) P[e] = a.P[(f * a.B + b) * 12 + e];
    float pt[3], cz[3];
    const Tap t = project_tap(ik, P, depth, x, y, w, h, pt, cz);
    const T* sb = a.src[f] + (size_t)b * h * w * C;
    float acc = 0.f;
    for (int ch = sub * 8; ch < C; ch += 64) {
      float tv[8], nw[8], ne[8], sw[8], se[8];
      load8(tb + ch, tv);
      tap_vec(sb, C, w, t, ch, nw, ne, sw, se);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float wv = nw[i] * t.nw;
        wv += ne[i] * t.ne; wv += sw[i] * t.sw; wv += se[i] * t.se;
        const float df = tv[i] - wv;
        acc += sqrtf(df * df + TD_L1_EPS2);
      }
    }
    const float L = group8_sum(acc) / (float)C;
    if (f == 0 || L < best) { best = L; idx = f; }
  }
  float contrib = 0.f;
  if (live && sub == 0) {
    a.argmin[pix] = (uint8_t)idx;
    contrib = best;
  }
  const float tot = block_sum<4>(contrib, s_red);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = tot;
}

template <typename T, int NS>
__global__ __launch_bounds__(TD_THREADS) void featwarp_bwd_kernel(const FeatWarpArgs<T, NS> a) {
  __shared__ float s_gl[4][8][64];          // per wave: d loss / d warped, [pixel][channel of the chunk]
  __shared__ float s_red[4][12];
  const int h = a.h, w = a.w, C = a.C;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int bps = (h * w * 8 + TD_THREADS - 1) / TD_THREADS;
  const int b = (int)blockIdx.x / bps;
  int lp = ((int)blockIdx.x % bps) * (TD_THREADS / 8) + (int)(threadIdx.x >> 3);
  const int sub = (int)(threadIdx.x & 7), pl = lane >> 3;    // pixel slot of this lane inside the wave
  const bool live = lp < h * w;
  if (!live) lp = h * w - 1;
  const int x = lp % w, y = lp / w;
  const long long pix = (long long)b * h * w + lp;
  const float d = upsample_disp(a.disp + (size_t)b * a.hs * a.ws, a.hs, a.ws, (float)a.hs / (float)h,
                                (float)a.ws / (float)w, y, x);
  const float depth = fast_rcp(a.min_disp + a.disp_range * d);
  float ik[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) ik[i] = a.invK[b * 16 + (i / 3) * 4 + (i % 3)];
  const int f_sel = (int)a.argmin[pix];
  float P[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) P[e] = a.P[(f_sel * a.B + b) * 12 + e];
  float pt[3], cz[3];
  const Tap t = project_tap(ik, P, depth, x, y, w, h, pt, cz);
  const float g = live ? a.gscale[0] * a.inv_count / (float)C : 0.f;
  const T* tb = a.tgt + (size_t)pix * C;
  // every lane needs its own frame's base pointers (frames differ between the pixels of a wave)
  const T* sb = a.src[0] + (size_t)b * h * w * C;
  int* db = a.d_src[0] + (size_t)b * h * w * C;
  const float gabs = fabsf(a.gscale[0] * a.inv_count / (float)C);
  const float qscale = gabs > 0.f ? 16384.f / gabs : 0.f;          // launch-uniform
#pragma unroll
  for (int f = 1; f < NS; ++f)
    if (f_sel == f) { sb = a.src[f] + (size_t)b * h * w * C; db = a.d_src[f] + (size_t)b * h * w * C; }
  const float ex = (float)t.x0 + 1.f - t.ix, wx = t.ix - (float)t.x0;
  const float ey = (float)t.y0 + 1.f - t.iy, wy = t.iy - (float)t.y0;
  float gix = 0.f, giy = 0.f;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int ch = c0 + sub * 8;
    float tv[8], nw[8], ne[8], sw[8], se[8], gl[8], dt[8];
    load8(tb + ch, tv);
    tap_vec(sb, C, w, t, ch, nw, ne, sw, se);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float wv = nw[i] * t.nw;
      wv += ne[i] * t.ne; wv += sw[i] * t.sw; wv += se[i] * t.se;
      const float df = tv[i] - wv;
      const float r = g * df / sqrtf(df * df + TD_L1_EPS2);      // d loss / d tgt ; d loss / d warped = -r
      dt[i] = r;
      gl[i] = -r;
      const float vne = t.in_e ? ne[i] : 0.f, vsw = t.in_s ? sw[i] : 0.f, vse = (t.in_e && t.in_s) ? se[i] : 0.f;
      gix += gl[i] * (-nw[i] * ey + vne * ey - vsw * wy + vse * wy);
      giy += gl[i] * (-nw[i] * ex - vne * wx + vsw * ex + vse * wx);
      s_gl[wid][pl][sub * 8 + i] = gl[i];
    }
    if (live) store8(a.d_tgt + (size_t)pix * C + ch, dt);
    // ---- scatter to the source feature gradient: one pixel x 64 channels per atomic wave-instruction ----
    // (wave-private LDS region; a wave executes in order, so no barrier is needed)
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int p = 0; p < 8; ++p) {
      const int src_lane = p * 8;
      const bool p_live = __shfl((int)live, src_lane, 64) != 0;
      if (!p_live) continue;
      const int tx0 = __shfl(t.x0, src_lane, 64), tx1 = __shfl(t.x1, src_lane, 64);
      const int ty0 = __shfl(t.y0, src_lane, 64), ty1 = __shfl(t.y1, src_lane, 64);
      const float wnw = __shfl(t.nw, src_lane, 64), wne = __shfl(t.ne, src_lane, 64);
      const float wsw = __shfl(t.sw, src_lane, 64), wse = __shfl(t.se, src_lane, 64);
      const unsigned long long dbp = __shfl((unsigned long long)db, src_lane, 64);
      int* dbase = reinterpret_cast<int*>(dbp) + c0 + lane;
      const float v = s_gl[wid][p][lane] * qscale;
      // ATen skips taps outside the image; their weight is exactly 0 here, adding 0 is harmless
      atomicAdd(dbase + ((size_t)ty0 * w + tx0) * C, __float2int_rn(v * wnw));
      atomicAdd(dbase + ((size_t)ty0 * w + tx1) * C, __float2int_rn(v * wne));
      atomicAdd(dbase + ((size_t)ty1 * w + tx0) * C, __float2int_rn(v * wsw));
      atomicAdd(dbase + ((size_t)ty1 * w + tx1) * C, __float2int_rn(v * wse));
    }
    __builtin_amdgcn_wave_barrier();
  }
  gix = group8_sum(gix);
  giy = group8_sum(giy);
  // chain: sampling coordinate -> (u, v) -> projection -> P and depth (lane sub == 0 of each pixel)
  float dP[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) dP[e] = 0.f;
  if (live && sub == 0) {
    const float du = gix * t.gmx * ((float)w / (float)(w - 1));
    const float dvv = giy * t.gmy * ((float)h / (float)(h - 1));
    const float iz = fast_rcp(cz[2]);
    const float dc0 = du * iz, dc1 = dvv * iz, dc2 = -(du * cz[0] + dvv * cz[1]) * iz * iz;
    dP[0] = dc0 * pt[0]; dP[1] = dc0 * pt[1]; dP[2] = dc0 * pt[2]; dP[3] = dc0;
    dP[4] = dc1 * pt[0]; dP[5] = dc1 * pt[1]; dP[6] = dc1 * pt[2]; dP[7] = dc1;
    dP[8] = dc2 * pt[0]; dP[9] = dc2 * pt[1]; dP[10] = dc2 * pt[2]; dP[11] = dc2;
    const float dX = dc0 * P[0] + dc1 * P[4] + dc2 * P[8];
    const float dY = dc0 * P[1] + dc1 * P[5] + dc2 * P[9];
    const float dZ = dc0 * P[2] + dc1 * P[6] + dc2 * P[10];
    const float fx = (float)x, fy = (float)y;
    const float r0 = ik[0] * fx + ik[1] * fy + ik[2], r1 = ik[3] * fx + ik[4] * fy + ik[5], r2 = ik[6] * fx + ik[7] * fy + ik[8];
    a.d_up[pix] = (dX * r0 + dY * r1 + dZ * r2) * (-a.disp_range * depth * depth);
  }
  // dP: per block (= one sample), per frame
#pragma unroll 1
  for (int f = 0; f < NS; ++f) {
    float part[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) part[e] = wave_sum((f_sel == f) ? dP[e] : 0.f);
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < 12; ++e) s_red[wid][e] = part[e];
    }
    __syncthreads();
    if (threadIdx.x < 12)
      a.dP_partial[(size_t)blockIdx.x * (NS * 12) + f * 12 + threadIdx.x] =
          (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
    __syncthreads();
  }
}

// accumulators -> gradient in the feature dtype: d_src[i] = acc[i] * |g| / 2^14
template <typename T>
__global__ __launch_bounds__(TD_THREADS) void featwarp_dsrc_finish_kernel(const int* __restrict__ acc, const float* __restrict__ gscale,
                                                                          float inv_count_over_C, long long nvec, T* __restrict__ out) {
  const float unit = fabsf(gscale[0] * inv_count_over_C) * (1.f / 16384.f);
  const long long stride = (long long)gridDim.x * TD_THREADS;
  for (long long i = (long long)blockIdx.x * TD_THREADS + threadIdx.x; i < nvec; i += stride) {
    const int4 a0 = reinterpret_cast<const int4*>(acc)[2 * i], a1 = reinterpret_cast<const int4*>(acc)[2 * i + 1];
    const float v[8] = {(float)a0.x * unit, (float)a0.y * unit, (float)a0.z * unit, (float)a0.w * unit,
                        (float)a1.x * unit, (float)a1.y * unit, (float)a1.z * unit, (float)a1.w * unit};
    store8(out + i * 8, v);
  }
}

// blocks never straddle two samples: the grid is laid out per sample (gridDim = B * blocks_per_sample)
template <typename T, int NS>
static int launch_featwarp(bool fwd, FeatWarpArgs<T, NS>& a, hipStream_t st) {
  const unsigned blocks = (unsigned)(a.B * ((a.h * a.w * 8 + TD_THREADS - 1) / TD_THREADS));
  if (fwd) hipLaunchKernelGGL((featwarp_fwd_kernel<T, NS>), dim3(blocks), dim3(TD_THREADS), 0, st, a);
  else hipLaunchKernelGGL((featwarp_bwd_kernel<T, NS>), dim3(blocks), dim3(TD_THREADS), 0, st, a);
  return record_launch_error(hipGetLastError(), fwd ? "td_featwarp_fwd" : "td_featwarp_bwd");
}

template <typename T, int NS>
static int run_featwarp(bool fwd, const void* tgt, const void* const* src, const float* disp, const float* P,
                        const float* invK, uint8_t* argmin, float* partial, const float* gscale, float inv_count,
                        void* d_tgt, int* const* d_src, float* d_up, float* dP_partial, int B, int h, int w, int C,
                        int hs, int ws, float min_depth, float max_depth, hipStream_t st) {
  FeatWarpArgs<T, NS> a;
  a.tgt = (const T*)tgt;
  for (int i = 0; i < NS; ++i) { a.src[i] = (const T*)src[i]; a.d_src[i] = d_src ? d_src[i] : nullptr; }
  a.disp = disp; a.P = P; a.invK = invK; a.argmin = argmin; a.partial = partial; a.gscale = gscale;
  a.inv_count = inv_count; a.d_tgt = (T*)d_tgt; a.d_up = d_up; a.dP_partial = dP_partial;
  a.B = B; a.h = h; a.w = w; a.C = C; a.hs = hs; a.ws = ws;
  const double lo = 1.0 / (double)max_depth, hi = 1.0 / (double)min_depth;
  a.min_disp = (float)lo; a.disp_range = (float)(hi - lo);
  return launch_featwarp<T, NS>(fwd, a, st);
}

}  // namespace td

extern "C" int td_featwarp_num_blocks(int B, int h, int w) {
  if (B <= 0 || h <= 0 || w <= 0) return 0;
  return B * ((h * w * 8 + TD_THREADS - 1) / TD_THREADS);
}

static int featwarp_dispatch(bool fwd, const void* tgt, const void* const* src, int n_src, int dtype, const float* disp,
                             const float* P, const float* invK, uint8_t* argmin, float* partial, const float* gscale,
                             float inv_count, void* d_tgt, int* const* d_src, float* d_up, float* dP_partial, int B,
                             int h, int w, int C, int hs, int ws, float min_depth, float max_depth, td_stream_t stream) {
  if (!tgt || !src || !disp || !P || !invK || !argmin || n_src < 1 || n_src > 2 || B <= 0) return TD_ERR_BAD_ARG;
  if (C % 64 != 0 || h < 2 || w < 2 || hs > h || ws > w) return TD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
#define TD_FW(T, NS) return td::run_featwarp<T, NS>(fwd, tgt, src, disp, P, invK, argmin, partial, gscale, inv_count, d_tgt, \
                                                    d_src, d_up, dP_partial, B, h, w, C, hs, ws, min_depth, max_depth, st)
  if (dtype == TD_DTYPE_BF16) { if (n_src == 1) TD_FW(__hip_bfloat16, 1); TD_FW(__hip_bfloat16, 2); }
  if (dtype == TD_DTYPE_F32) { if (n_src == 1) TD_FW(float, 1); TD_FW(float, 2); }
#undef TD_FW
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_featwarp_fwd(const void* tgt, const void* const* src, int n_src, int dtype, const float* disp,
                               const float* P, const float* invK, int B, int h, int w, int C, int hs, int ws,
                               float min_depth, float max_depth, uint8_t* argmin, float* partial, td_stream_t stream) {
  if (!partial) return TD_ERR_BAD_ARG;
  return featwarp_dispatch(true, tgt, src, n_src, dtype, disp, P, invK, argmin, partial, nullptr, 0.f, nullptr, nullptr,
                           nullptr, nullptr, B, h, w, C, hs, ws, min_depth, max_depth, stream);
}

extern "C" int td_featwarp_bwd(const void* tgt, const void* const* src, int n_src, int dtype, const float* disp,
                               const float* P, const float* invK, const uint8_t* argmin, const float* gscale,
                               float inv_count, int B, int h, int w, int C, int hs, int ws, float min_depth,
                               float max_depth, void* d_tgt, int* const* d_src, float* d_up, float* dP_partial,
                               td_stream_t stream) {
  if (!gscale || !d_tgt || !d_src || !d_up || !dP_partial) return TD_ERR_BAD_ARG;
  for (int i = 0; i < n_src; ++i) if (!d_src[i]) return TD_ERR_BAD_ARG;
  return featwarp_dispatch(false, tgt, src, n_src, dtype, disp, P, invK, const_cast<uint8_t*>(argmin), nullptr, gscale,
                           inv_count, d_tgt, d_src, d_up, dP_partial, B, h, w, C, hs, ws, min_depth, max_depth, stream);
}

extern "C" int td_featwarp_dsrc_finish(const int* acc, const float* gscale, float inv_count, int C, long long n, int dtype, void* out,
                                       td_stream_t stream) {
  if (!acc || !gscale || !out || n <= 0 || C <= 0) return TD_ERR_BAD_ARG;
  if (n % 8 != 0 || (dtype != TD_DTYPE_BF16 && dtype != TD_DTYPE_F32)) return TD_ERR_UNSUPPORTED;
  const long long nvec = n / 8;
  long long blocks = (nvec + TD_THREADS * 4 - 1) / (TD_THREADS * 4);
  if (blocks > 2048) blocks = 2048;
  if (dtype == TD_DTYPE_BF16)
    hipLaunchKernelGGL((td::featwarp_dsrc_finish_kernel<__hip_bfloat16>), dim3((unsigned)blocks), dim3(TD_THREADS), 0, (hipStream_t)stream, acc, gscale,
                       inv_count / (float)C, nvec, (__hip_bfloat16*)out);
  else
    hipLaunchKernelGGL((td::featwarp_dsrc_finish_kernel<float>), dim3((unsigned)blocks), dim3(TD_THREADS), 0, (hipStream_t)stream, acc, gscale,
                       inv_count / (float)C, nvec, (float*)out);
  return td::record_launch_error(hipGetLastError(), "td_featwarp_dsrc_finish");
}
