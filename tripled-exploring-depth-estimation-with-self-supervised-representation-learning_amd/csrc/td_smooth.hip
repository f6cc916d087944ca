// Edge-aware first + second order smoothness (forward, backward), per-sample mean
// normalisation, area down-sampling, and the small deterministic reductions.
#include "td_common.h"

namespace td {

constexpr int ST_H = 16, ST_W = 64;   // smoothness tile (anchors)

// ---------------------------------------------------------------------------
// per-sample mean of disp: one block per sample
__global__ __launch_bounds__(TD_THREADS) void sample_mean_kernel(const float* __restrict__ disp,
                                                                 int n, float* __restrict__ mean) {
  __shared__ float s_red[4];
  const float* p = disp + (size_t)blockIdx.x * n;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += TD_THREADS) acc += p[i];
  const float tot = block_sum(acc, s_red);
  if (threadIdx.x == 0) mean[blockIdx.x] = tot / (float)n;
}

// The six stencil terms anchored at (y,x) and their edge weights.  t[k] = derivative of the
// (normalised) disparity, w[k] = exp(-0.5 * mean_c |same derivative of the image|); `ok[k]`
// says whether the anchor is inside that term's (shrunken) domain.
struct Stencil {
  float t[6];
  float w[6];
  bool ok[6];
};

__device__ __forceinline__ void stencil_at(const float* __restrict__ D, const float* __restrict__ I,
                                           int h, int w, int y, int x, float inv, Stencil& s) {
  const bool x1 = x + 1 < w, x2 = x + 2 < w, y1 = y + 1 < h, y2 = y + 2 < h;
  s.ok[0] = x1; s.ok[1] = y1; s.ok[2] = x2; s.ok[3] = x1 && y1; s.ok[4] = x1 && y1; s.ok[5] = y2;
  const size_t o = (size_t)y * w + x;
  const int sx1 = x1 ? 1 : 0, sx2 = x2 ? 2 : 0;
  const size_t sy1 = y1 ? (size_t)w : 0, sy2 = y2 ? (size_t)2 * w : 0;
  {
    const float d00 = D[o] * inv, d01 = D[o + sx1] * inv, d02 = D[o + sx2] * inv;
    const float d10 = D[o + sy1] * inv, d11 = D[o + sy1 + sx1] * inv, d20 = D[o + sy2] * inv;
    const float dx0 = d01 - d00, dx1 = d02 - d01, dxr1 = d11 - d10;   // dx at (y,x),(y,x+1),(y+1,x)
    const float dy0 = d10 - d00, dy1 = d20 - d10, dyc1 = d11 - d01;   // dy at (y,x),(y+1,x),(y,x+1)
    s.t[0] = dx0; s.t[1] = dy0; s.t[2] = dx1 - dx0; s.t[3] = dxr1 - dx0; s.t[4] = dyc1 - dy0; s.t[5] = dy1 - dy0;
  }
  float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const size_t plane = (size_t)h * w;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* P = I + c * plane + o;
    const float i00 = P[0], i01 = P[sx1], i02 = P[sx2], i10 = P[sy1], i11 = P[sy1 + sx1], i20 = P[sy2];
    const float dx0 = i01 - i00, dx1 = i02 - i01, dxr1 = i11 - i10;
    const float dy0 = i10 - i00, dy1 = i20 - i10, dyc1 = i11 - i01;
    m[0] += fabsf(dx0); m[1] += fabsf(dy0); m[2] += fabsf(dx1 - dx0);
    m[3] += fabsf(dxr1 - dx0); m[4] += fabsf(dyc1 - dy0); m[5] += fabsf(dy1 - dy0);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) s.w[k] = expf(-0.5f * (m[k] / 3.f));
}

__global__ __launch_bounds__(TD_THREADS) void smooth_fwd_kernel(const float* __restrict__ disp,
                                                                const float* __restrict__ img,
                                                                const float* __restrict__ mean,
                                                                int h, int w, int normalize,
                                                                float* __restrict__ partial) {
  __shared__ float s_red[4];
  const int b = blockIdx.z;
  const float inv = normalize ? 1.f / (mean[b] + 1e-7f) : 1.f;
  const float* D = disp + (size_t)b * h * w;
  const float* I = img + (size_t)b * 3 * h * w;
  const int x = blockIdx.x * ST_W + (threadIdx.x & 63);
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = threadIdx.x >> 6; r < ST_H; r += 4) {
    const int y = blockIdx.y * ST_H + r;
    if (y < h && x < w) {
      Stencil s;
      stencil_at(D, I, h, w, y, x, inv, s);
#pragma unroll
      for (int k = 0; k < 6; ++k) if (s.ok[k]) acc[k] += fabsf(s.t[k]) * s.w[k];
    }
  }
  const size_t blk = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float tot = block_sum(acc[k], s_red);
    if (threadIdx.x == 0) partial[blk * 6 + k] = tot;
  }
}

__global__ __launch_bounds__(TD_THREADS) void smooth_finish_kernel(const float* __restrict__ partial,
                                                                   int nblk, int B, int h, int w,
                                                                   float weight, float* __restrict__ loss) {
  __shared__ float s_red[4];
  const float cnt[6] = {(float)B * h * (w - 1), (float)B * (h - 1) * w, (float)B * h * (w - 2),
                        (float)B * (h - 1) * (w - 1), (float)B * (h - 1) * (w - 1), (float)B * (h - 2) * w};
  float total = 0.f;
  for (int k = 0; k < 6; ++k) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += TD_THREADS) acc += partial[(size_t)i * 6 + k];
    const float tot = block_sum(acc, s_red);
    total += tot / cnt[k];
  }
  if (threadIdx.x == 0) loss[0] = weight * total;
}

// Backward, stage 1: gradient w.r.t. the normalised disparity, gather form through an LDS
// tile of per-anchor signed weights (sign(t_k) * w_k / count_k).
__global__ __launch_bounds__(TD_THREADS) void smooth_bwd_kernel(const float* __restrict__ disp,
                                                                const float* __restrict__ img,
                                                                const float* __restrict__ mean,
                                                                int B, int h, int w, int normalize,
                                                                const float* __restrict__ gscale,
                                                                float weight,
                                                                float* __restrict__ g_hat,
                                                                float* __restrict__ dot_partial) {
  __shared__ float s_s[6][ST_H + 2][ST_W + 2];   // anchors (ty0-2 .. ty0+ST_H-1) x (tx0-2 .. tx0+ST_W-1)
  __shared__ float s_red[4];
  const int b = blockIdx.z;
  const int ty0 = blockIdx.y * ST_H, tx0 = blockIdx.x * ST_W;
  const float inv = normalize ? 1.f / (mean[b] + 1e-7f) : 1.f;
  const float* D = disp + (size_t)b * h * w;
  const float* I = img + (size_t)b * 3 * h * w;
  const float gw = gscale[0] * weight;
  const float icnt[6] = {gw / ((float)B * h * (w - 1)), gw / ((float)B * (h - 1) * w),
                         gw / ((float)B * h * (w - 2)), gw / ((float)B * (h - 1) * (w - 1)),
                         gw / ((float)B * (h - 1) * (w - 1)), gw / ((float)B * (h - 2) * w)};
  for (int pos = threadIdx.x; pos < (ST_H + 2) * (ST_W + 2); pos += TD_THREADS) {
    const int py = pos / (ST_W + 2), px = pos - py * (ST_W + 2);
    const int y = ty0 + py - 2, x = tx0 + px - 2;
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < h && x >= 0 && x < w) {
      Stencil s;
      stencil_at(D, I, h, w, y, x, inv, s);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float sg = s.t[k] > 0.f ? 1.f : (s.t[k] < 0.f ? -1.f : 0.f);
        v[k] = s.ok[k] ? sg * s.w[k] * icnt[k] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) s_s[k][py][px] = v[k];
  }
  __syncthreads();

  const int cx = threadIdx.x & 63;
  const int x = tx0 + cx;
  float dot = 0.f;
  for (int r = threadIdx.x >> 6; r < ST_H; r += 4) {
    const int y = ty0 + r;
    if (y < h && x < w) {
      const int py = r + 2, px = cx + 2;   // LDS position of anchor (y,x)
      float gacc = 0.f;
      // dx: +1 from anchor (y,x-1), -1 from (y,x)
      gacc += s_s[0][py][px - 1] - s_s[0][py][px];
      // dy
      gacc += s_s[1][py - 1][px] - s_s[1][py][px];
      // dxx: D[x+2] - 2 D[x+1] + D[x]
      gacc += s_s[2][py][px - 2] - 2.f * s_s[2][py][px - 1] + s_s[2][py][px];
      // dxy and dyx: D[y+1][x+1] - D[y+1][x] - D[y][x+1] + D[y][x]
#pragma unroll
      for (int k = 3; k <= 4; ++k)
        gacc += s_s[k][py - 1][px - 1] - s_s[k][py - 1][px] - s_s[k][py][px - 1] + s_s[k][py][px];
      // dyy
      gacc += s_s[5][py - 2][px] - 2.f * s_s[5][py - 1][px] + s_s[5][py][px];
      g_hat[(size_t)b * h * w + (size_t)y * w + x] = gacc;
      dot += gacc * D[(size_t)y * w + x];
    }
  }
  const float tot = block_sum(dot, s_red);
  if (threadIdx.x == 0)
    dot_partial[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
}

// Backward, stage 2: through disp_hat = disp / (mean + eps):
//   d disp_i = g_i * inv - (sum_j g_j disp_j) * inv^2 / N
__global__ __launch_bounds__(TD_THREADS) void smooth_bwd_finish_kernel(
    const float* __restrict__ g_hat, const float* __restrict__ dot_partial, const float* __restrict__ mean,
    int n, int blocks_per_sample, int normalize, float* __restrict__ d_disp, int accumulate) {
  __shared__ float s_red[4];
  __shared__ float s_corr;
  const int b = blockIdx.y;
  float inv = 1.f;
  if (normalize) {
    inv = 1.f / (mean[b] + 1e-7f);
    float acc = 0.f;
    for (int i = threadIdx.x; i < blocks_per_sample; i += TD_THREADS)
      acc += dot_partial[(size_t)b * blocks_per_sample + i];
    const float tot = block_sum(acc, s_red);
    if (threadIdx.x == 0) s_corr = tot * inv * inv / (float)n;
    __syncthreads();
  }
  const float corr = normalize ? s_corr : 0.f;
  const int i = blockIdx.x * TD_THREADS + threadIdx.x;
  if (i < n) {
    const size_t o = (size_t)b * n + i;
    const float v = g_hat[o] * inv - corr;
    if (accumulate) d_disp[o] += v; else d_disp[o] = v;
  }
}

// ---------------------------------------------------------------------------
// F.interpolate(mode="area") for integer factors = the mean of an fy x fx window.  G lanes share one output (G = 1, 4, 16 or
// 64 by window size): lane l takes window elements l, l + G, ... (row-major, so a group reads contiguous runs of a row) and
// the group adds its partial sums in a fixed butterfly order.  One thread per output left the 1/16 and 1/32 pyramid levels
// (256 / 1024 strided loads per thread, a few thousand threads) at up to 142 us per call.
template <int G>
__global__ __launch_bounds__(TD_THREADS) void area_downsample_kernel(const float* __restrict__ img,
                                                                     int planes, int H, int W, int h,
                                                                     int w, int fy, int fx,
                                                                     float* __restrict__ out) {
  const size_t tid = (size_t)blockIdx.x * TD_THREADS + threadIdx.x;
  const size_t id = tid / G;
  const int l = (int)(tid % G);
  const size_t total = (size_t)planes * h * w;
  const bool live = id < total;
  const size_t idc = live ? id : 0;
  const int x = idc % w, y = (idc / w) % h;
  const size_t p = idc / ((size_t)w * h);
  const float* src = img + p * H * W + (size_t)y * fy * W + (size_t)x * fx;
  float acc = 0.f;
  const int n = fy * fx;
  for (int e = l; e < n; e += G) {
    const int dy = e / fx, dx = e - dy * fx;
    acc += src[(size_t)dy * W + dx];
  }
#pragma unroll
  for (int m = G >> 1; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if (live && l == 0) out[id] = acc / (float)n;
}

__global__ __launch_bounds__(TD_THREADS) void sum_scaled_kernel(const float* __restrict__ part, int n,
                                                                float scale, float* __restrict__ out) {
  __shared__ float s_red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += TD_THREADS) acc += part[i];
  const float tot = block_sum(acc, s_red);
  if (threadIdx.x == 0) out[0] = tot * scale;
}

}  // namespace td

extern "C" int td_smooth_num_blocks(int B, int h, int w) {
  if (B <= 0 || h <= 0 || w <= 0) return 0;
  return B * ((h + td::ST_H - 1) / td::ST_H) * ((w + td::ST_W - 1) / td::ST_W);
}

extern "C" int td_smooth_fwd(const float* disp, const float* img, int B, int h, int w, int normalize,
                             float* mean, float* partial, td_stream_t stream) {
  if (!disp || !img || !mean || !partial || B <= 0) return TD_ERR_BAD_ARG;
  if (h < 3 || w < 3) return TD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(td::sample_mean_kernel, dim3(B), dim3(TD_THREADS), 0, st, disp, h * w, mean);
  dim3 grid((w + td::ST_W - 1) / td::ST_W, (h + td::ST_H - 1) / td::ST_H, B);
  hipLaunchKernelGGL(td::smooth_fwd_kernel, grid, dim3(TD_THREADS), 0, st, disp, img, mean, h, w, normalize, partial);
  return td::record_launch_error(hipGetLastError(), "td_smooth_fwd");
}

extern "C" int td_smooth_finish(const float* partial, int B, int h, int w, float weight, float* loss,
                                td_stream_t stream) {
  if (!partial || !loss || B <= 0) return TD_ERR_BAD_ARG;
  if (h < 3 || w < 3) return TD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(td::smooth_finish_kernel, dim3(1), dim3(TD_THREADS), 0, (hipStream_t)stream,
                     partial, td_smooth_num_blocks(B, h, w), B, h, w, weight, loss);
  return td::record_launch_error(hipGetLastError(), "td_smooth_finish");
}

extern "C" int td_smooth_bwd(const float* disp, const float* img, const float* mean, int B, int h, int w,
                             int normalize, const float* gscale, float weight, float* g_hat,
                             float* dot_partial, float* d_disp, int accumulate, td_stream_t stream) {
  if (!disp || !img || !mean || !gscale || !g_hat || !dot_partial || !d_disp || B <= 0) return TD_ERR_BAD_ARG;
  if (h < 3 || w < 3) return TD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((w + td::ST_W - 1) / td::ST_W, (h + td::ST_H - 1) / td::ST_H, B);
  hipLaunchKernelGGL(td::smooth_bwd_kernel, grid, dim3(TD_THREADS), 0, st, disp, img, mean, B, h, w,
                     normalize, gscale, weight, g_hat, dot_partial);
  const int n = h * w;
  dim3 grid2((n + TD_THREADS - 1) / TD_THREADS, B);
  hipLaunchKernelGGL(td::smooth_bwd_finish_kernel, grid2, dim3(TD_THREADS), 0, st, g_hat, dot_partial,
                     mean, n, (int)(grid.x * grid.y), normalize, d_disp, accumulate);
  return td::record_launch_error(hipGetLastError(), "td_smooth_bwd");
}

extern "C" int td_area_downsample(const float* img, int B, int C, int H, int W, int h, int w, float* out,
                                  td_stream_t stream) {
  if (!img || !out || B <= 0 || C <= 0 || h <= 0 || w <= 0) return TD_ERR_BAD_ARG;
  if (H % h != 0 || W % w != 0) return TD_ERR_UNSUPPORTED;
  const size_t total = (size_t)B * C * h * w;
  const int fy = H / h, fx = W / w, n = fy * fx;
  hipStream_t st = (hipStream_t)stream;
#define TD_AREA_LAUNCH(G)                                                                                                   \
  hipLaunchKernelGGL(td::area_downsample_kernel<G>, dim3((unsigned)((total * G + TD_THREADS - 1) / TD_THREADS)), dim3(TD_THREADS), 0, st, \
                     img, B * C, H, W, h, w, fy, fx, out)
  if (n >= 256) TD_AREA_LAUNCH(64);
  else if (n >= 64) TD_AREA_LAUNCH(16);
  else if (n >= 16) TD_AREA_LAUNCH(4);
  else TD_AREA_LAUNCH(1);
#undef TD_AREA_LAUNCH
  return td::record_launch_error(hipGetLastError(), "td_area_downsample");
}

extern "C" int td_sum_scaled(const float* partial, int n, float scale, float* out, td_stream_t stream) {
  if (!partial || !out || n <= 0) return TD_ERR_BAD_ARG;
  hipLaunchKernelGGL(td::sum_scaled_kernel, dim3(1), dim3(TD_THREADS), 0, (hipStream_t)stream, partial, n, scale, out);
  return td::record_launch_error(hipGetLastError(), "td_sum_scaled");
}
