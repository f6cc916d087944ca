// Edge-aware first + second order regulariser on C-channel feature maps
// (get_feature_regularization_loss, reference mono/model/mono_fm_joint/net.py:309-330), fused:
//   loss = sum_k coef_k * mean(|d_k F| * exp(-a * mean_c |d_k I|)),   k in {dx, dy, dxx, dxy, dyx, dyy}
// The reference materialises ~40 full-size fp32 temporaries per level (forward + autograd); here
// the feature map (channels-last, bf16 or f32) is read once in the forward and once more in the
// backward.  The per-pixel image weights are shared by all channels and pre-multiplied by
// coef_k / count_k (td_edge_weights), so the whole loss is one weighted sum.
#include <hip/hip_bf16.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

// W[b,k,y,x] = scale[k] * exp(-a * mean_c |d_k I|(y,x)), 0 where term k has no anchor at (y,x)
__global__ __launch_bounds__(TD_THREADS) void edge_weights_kernel(const float* __restrict__ img, int B, int h, int w,
                                                                  float a, float s0, float s1, float s2, float s3,
                                                                  float s4, float s5, float* __restrict__ Wt) {
  const int id = blockIdx.x * TD_THREADS + threadIdx.x;
  const int total = B * h * w;
  if (id >= total) return;
  const int x = id % w, y = (id / w) % h, b = id / (w * h);
  const bool x1 = x + 1 < w, x2 = x + 2 < w, y1 = y + 1 < h, y2 = y + 2 < h;
  const bool ok[6] = {x1, y1, x2, x1 && y1, x1 && y1, y2};
  const float sc[6] = {s0, s1, s2, s3, s4, s5};
  const int sx1 = x1 ? 1 : 0, sx2 = x2 ? 2 : 0, sy1 = y1 ? w : 0, sy2 = y2 ? 2 * w : 0;
  float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const size_t plane = (size_t)h * w;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* P = img + ((size_t)b * 3 + c) * plane + (size_t)y * w + x;
    const float i00 = P[0], i01 = P[sx1], i02 = P[sx2], i10 = P[sy1], i11 = P[sy1 + sx1], i20 = P[sy2];
    const float dx0 = i01 - i00, dx1 = i02 - i01, dxr1 = i11 - i10;
    const float dy0 = i10 - i00, dy1 = i20 - i10, dyc1 = i11 - i01;
    m[0] += fabsf(dx0); m[1] += fabsf(dy0); m[2] += fabsf(dx1 - dx0);
    m[3] += fabsf(dxr1 - dx0); m[4] += fabsf(dyc1 - dy0); m[5] += fabsf(dy1 - dy0);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k)
    Wt[((size_t)(b * 6 + k) * h + y) * w + x] = ok[k] ? sc[k] * expf(-a * (m[k] / 3.f)) : 0.f;
}

template <typename T>
__device__ __forceinline__ void ldf(const T* f, int C, int h, int w, int y, int x, float* v) {
  y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
  x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
  load8(f + ((size_t)y * w + x) * C, v);
}

// one thread = one anchor pixel x 8 channels
template <typename T>
__global__ __launch_bounds__(TD_THREADS) void featreg_fwd_kernel(const T* __restrict__ feat, const float* __restrict__ Wt,
                                                                 int B, int h, int w, int C, float* __restrict__ partial) {
  __shared__ float s_red[4];
  const int c8 = C >> 3;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  const long long total = (long long)B * h * w * c8;
  float acc = 0.f;
  if (gid < total) {
    const int cv = (int)(gid % c8);
    const long long pix = gid / c8;
    const int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((long long)w * h));
    float wk[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) wk[k] = Wt[((size_t)(b * 6 + k) * h + y) * w + x];
    const T* fb = feat + (size_t)b * h * w * C + (size_t)cv * 8;
    float v00[8], v01[8], v02[8], v10[8], v11[8], v20[8];
    ldf(fb, C, h, w, y, x, v00); ldf(fb, C, h, w, y, x + 1, v01); ldf(fb, C, h, w, y, x + 2, v02);
    ldf(fb, C, h, w, y + 1, x, v10); ldf(fb, C, h, w, y + 1, x + 1, v11); ldf(fb, C, h, w, y + 2, x, v20);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float dx0 = v01[i] - v00[i], dx1 = v02[i] - v01[i], dxr1 = v11[i] - v10[i];
      const float dy0 = v10[i] - v00[i], dy1 = v20[i] - v10[i], dyc1 = v11[i] - v01[i];
      acc += fabsf(dx0) * wk[0] + fabsf(dy0) * wk[1] + fabsf(dx1 - dx0) * wk[2] + fabsf(dxr1 - dx0) * wk[3] +
             fabsf(dyc1 - dy0) * wk[4] + fabsf(dy1 - dy0) * wk[5];
    }
  }
  const float tot = block_sum<4>(acc, s_red);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// one thread = one feature pixel x 8 channels; gathers from the <= 18 (anchor, term) pairs touching it
template <typename T>
__global__ __launch_bounds__(TD_THREADS) void featreg_bwd_kernel(const T* __restrict__ feat, const float* __restrict__ Wt,
                                                                 const float* __restrict__ gscale, int B, int h, int w,
                                                                 int C, T* __restrict__ grad) {
  const int c8 = C >> 3;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  const long long total = (long long)B * h * w * c8;
  if (gid >= total) return;
  const int cv = (int)(gid % c8);
  const long long pix = gid / c8;
  const int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((long long)w * h));
  const float g = gscale[0];
  auto W = [&](int k, int ay, int ax) -> float {     // weight of term k anchored at (y+ay, x+ax)
    const int yy = y + ay, xx = x + ax;
    if (yy < 0 || xx < 0) return 0.f;                 // (anchors never exceed the image on the high side here)
    return Wt[((size_t)(b * 6 + k) * h + yy) * w + xx];
  };
  const float w_dx_0 = W(0, 0, 0), w_dx_l = W(0, 0, -1);
  const float w_dy_0 = W(1, 0, 0), w_dy_u = W(1, -1, 0);
  const float w_xx_0 = W(2, 0, 0), w_xx_l = W(2, 0, -1), w_xx_ll = W(2, 0, -2);
  const float w_yy_0 = W(5, 0, 0), w_yy_u = W(5, -1, 0), w_yy_uu = W(5, -2, 0);
  const float w_xy_0 = W(3, 0, 0), w_xy_l = W(3, 0, -1), w_xy_u = W(3, -1, 0), w_xy_ul = W(3, -1, -1);
  const float w_yx_0 = W(4, 0, 0), w_yx_l = W(4, 0, -1), w_yx_u = W(4, -1, 0), w_yx_ul = W(4, -1, -1);
  const T* fb = feat + (size_t)b * h * w * C + (size_t)cv * 8;
  float c0[8], l1[8], l2[8], r1[8], r2[8], u1[8], u2[8], d1[8], d2[8], ul[8], ur[8], dl[8], dr[8];
  ldf(fb, C, h, w, y, x, c0);
  ldf(fb, C, h, w, y, x - 1, l1); ldf(fb, C, h, w, y, x - 2, l2);
  ldf(fb, C, h, w, y, x + 1, r1); ldf(fb, C, h, w, y, x + 2, r2);
  ldf(fb, C, h, w, y - 1, x, u1); ldf(fb, C, h, w, y - 2, x, u2);
  ldf(fb, C, h, w, y + 1, x, d1); ldf(fb, C, h, w, y + 2, x, d2);
  ldf(fb, C, h, w, y - 1, x - 1, ul); ldf(fb, C, h, w, y - 1, x + 1, ur);
  ldf(fb, C, h, w, y + 1, x - 1, dl); ldf(fb, C, h, w, y + 1, x + 1, dr);
  float out[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float dx_l = c0[i] - l1[i], dx_0 = r1[i] - c0[i], dx_ll = l1[i] - l2[i], dx_r = r2[i] - r1[i];
    const float dy_u = c0[i] - u1[i], dy_0 = d1[i] - c0[i], dy_uu = u1[i] - u2[i], dy_d = d2[i] - d1[i];
    float a = 0.f;
    a += sgn(dx_l) * w_dx_l - sgn(dx_0) * w_dx_0;
    a += sgn(dy_u) * w_dy_u - sgn(dy_0) * w_dy_0;
    a += sgn(dx_l - dx_ll) * w_xx_ll - 2.f * sgn(dx_0 - dx_l) * w_xx_l + sgn(dx_r - dx_0) * w_xx_0;
    a += sgn(dy_u - dy_uu) * w_yy_uu - 2.f * sgn(dy_0 - dy_u) * w_yy_u + sgn(dy_d - dy_0) * w_yy_0;
    // dxy(a) = dx(row below a) - dx(a);  dyx(a) = dy(col right of a) - dy(a)
    const float dx_ul = u1[i] - ul[i], dx_u = ur[i] - u1[i], dx_dl = d1[i] - dl[i], dx_d = dr[i] - d1[i];
    const float dy_ul = l1[i] - ul[i], dy_ur = r1[i] - ur[i], dy_l = dl[i] - l1[i], dy_r = dr[i] - r1[i];
    a += sgn(dx_l - dx_ul) * w_xy_ul - sgn(dx_0 - dx_u) * w_xy_u - sgn(dx_dl - dx_l) * w_xy_l + sgn(dx_d - dx_0) * w_xy_0;
    a += sgn(dy_u - dy_ul) * w_yx_ul - sgn(dy_ur - dy_u) * w_yx_u - sgn(dy_0 - dy_l) * w_yx_l + sgn(dy_r - dy_0) * w_yx_0;
    out[i] = a * g;
  }
  store8(grad + (size_t)pix * C + (size_t)cv * 8, out);
}

}  // namespace td

extern "C" int td_edge_weights(const float* img, int B, int h, int w, float a, const float* scale6 /*host*/,
                               float* Wt, td_stream_t stream) {
  if (!img || !scale6 || !Wt || B <= 0 || h <= 0 || w <= 0) return TD_ERR_BAD_ARG;
  const int total = B * h * w;
  hipLaunchKernelGGL(td::edge_weights_kernel, dim3((total + TD_THREADS - 1) / TD_THREADS), dim3(TD_THREADS), 0,
                     (hipStream_t)stream, img, B, h, w, a, scale6[0], scale6[1], scale6[2], scale6[3], scale6[4],
                     scale6[5], Wt);
  return td::record_launch_error(hipGetLastError(), "td_edge_weights");
}

extern "C" int td_featreg_num_blocks(int B, int h, int w, int C) {
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0) return 0;
  const long long total = (long long)B * h * w * (C / 8);
  return (int)((total + TD_THREADS - 1) / TD_THREADS);
}

extern "C" int td_featreg_fwd(const void* feat, int dtype, const float* Wt, int B, int h, int w, int C,
                              float* partial, td_stream_t stream) {
  if (!feat || !Wt || !partial || B <= 0 || h <= 0 || w <= 0 || C <= 0) return TD_ERR_BAD_ARG;
  if (C % 8 != 0) return TD_ERR_UNSUPPORTED;
  const unsigned blocks = (unsigned)td_featreg_num_blocks(B, h, w, C);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_DTYPE_BF16)
    hipLaunchKernelGGL((td::featreg_fwd_kernel<__hip_bfloat16>), dim3(blocks), dim3(TD_THREADS), 0, st,
                       (const __hip_bfloat16*)feat, Wt, B, h, w, C, partial);
  else if (dtype == TD_DTYPE_F32)
    hipLaunchKernelGGL((td::featreg_fwd_kernel<float>), dim3(blocks), dim3(TD_THREADS), 0, st, (const float*)feat, Wt,
                       B, h, w, C, partial);
  else
    return TD_ERR_UNSUPPORTED;
  return td::record_launch_error(hipGetLastError(), "td_featreg_fwd");
}

extern "C" int td_featreg_bwd(const void* feat, int dtype, const float* Wt, const float* gscale, int B, int h, int w,
                              int C, void* grad, td_stream_t stream) {
  if (!feat || !Wt || !gscale || !grad || B <= 0 || h <= 0 || w <= 0 || C <= 0) return TD_ERR_BAD_ARG;
  if (C % 8 != 0) return TD_ERR_UNSUPPORTED;
  const unsigned blocks = (unsigned)td_featreg_num_blocks(B, h, w, C);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_DTYPE_BF16)
    hipLaunchKernelGGL((td::featreg_bwd_kernel<__hip_bfloat16>), dim3(blocks), dim3(TD_THREADS), 0, st,
                       (const __hip_bfloat16*)feat, Wt, gscale, B, h, w, C, (__hip_bfloat16*)grad);
  else if (dtype == TD_DTYPE_F32)
    hipLaunchKernelGGL((td::featreg_bwd_kernel<float>), dim3(blocks), dim3(TD_THREADS), 0, st, (const float*)feat, Wt,
                       gscale, B, h, w, C, (float*)grad);
  else
    return TD_ERR_UNSUPPORTED;
  return td::record_launch_error(hipGetLastError(), "td_featreg_bwd");
}
