// Small fused maps of the auxiliary loss terms:
//   * robust-L1 channel-mean map + adjoint: compute_perceptional_loss on IMAGES, i.e. the
//     auto_res_loss of the disentangled model (reference: mono/model/mono_fm_joint_inpaint/net.py:520-527)
//     and the colourisation distillation term (:310-323); robust_l1: mono/model/mono_fm_joint/net.py:59-65;
//   * sRGB -> normalised CIE Lab of the colourisation input
//     (reference: mono/model/mono_fm_joint_inpaint/color_conversions.py:6-27, 52-75, 106-114).
// One thread per pixel, one read of every operand, one write of the result: the ATen compositions these
// replace are 7 (forward) + ~10 (backward) and ~25 element-wise launches over full-resolution fp32 tensors.
#include <hip/hip_bf16.h>

#include "td_common.h"

namespace td {

struct Strides4 { long long n, c, h, w; };

template <typename T> __device__ __forceinline__ float ld(const T* p) { return (float)*p; }
template <> __device__ __forceinline__ float ld<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v) { *p = (T)v; }
template <> __device__ __forceinline__ void st<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

// out[b,y,x] = weight * mean_c sqrt((pred - target)^2 + 1e-6)              (FWD)
// dpred[b,c,y,x] = gmap[b,y,x] * weight / C * (pred - target) / sqrt(...)  (BWD; dpred has pred's strides)
template <typename T, bool FWD>
__global__ __launch_bounds__(TD_THREADS) void l1map_kernel(const T* __restrict__ pred, Strides4 sp,
                                                           const float* __restrict__ target, int B, int C, int H, int W,
                                                           float weight, const float* __restrict__ gmap,
                                                           float* __restrict__ out, T* __restrict__ dpred) {
  const long long plane = (long long)H * W;
  const long long total = (long long)B * plane;
  const float inv_c = 1.f / (float)C;
  for (long long i = (long long)blockIdx.x * TD_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * TD_THREADS) {
    const int b = (int)(i / plane);
    const long long r = i - (long long)b * plane;
    const int y = (int)(r / W), x = (int)(r - (long long)y * W);
    const T* pp = pred + b * sp.n + y * sp.h + x * sp.w;
    const float* tp = target + (long long)b * C * plane + r;
    if (FWD) {
      float acc = 0.f;
      for (int c = 0; c < C; ++c) {
        const float d = ld<T>(pp + c * sp.c) - tp[c * plane];
        acc += sqrtf(d * d + TD_L1_EPS2);
      }
      out[i] = weight * (acc * inv_c);
    } else {
      const float g = gmap[i] * weight * inv_c;
      T* dp = dpred + b * sp.n + y * sp.h + x * sp.w;
      for (int c = 0; c < C; ++c) {
        const float d = ld<T>(pp + c * sp.c) - tp[c * plane];
        st<T>(dp + c * sp.c, g * d / sqrtf(d * d + TD_L1_EPS2));
      }
    }
  }
}

__global__ __launch_bounds__(TD_THREADS) void rgb2lab_kernel(const float* __restrict__ rgb, int B, long long plane, float l_cent,
                                                             float l_norm, float ab_norm, float* __restrict__ lab) {
  const long long total = (long long)B * plane;
  for (long long i = (long long)blockIdx.x * TD_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * TD_THREADS) {
    const long long b = i / plane, r = i - b * plane;
    const float* p = rgb + b * 3 * plane + r;
    float lin[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = p[c * plane];
      lin[c] = v > 0.04045f ? powf((v + 0.055f) / 1.055f, 2.4f) : v / 12.92f;
    }
    const float X = 0.412453f * lin[0] + 0.357580f * lin[1] + 0.180423f * lin[2];
    const float Y = 0.212671f * lin[0] + 0.715160f * lin[1] + 0.072169f * lin[2];
    const float Z = 0.019334f * lin[0] + 0.119193f * lin[1] + 0.950227f * lin[2];
    const float s[3] = {X / 0.95047f, Y / 1.0f, Z / 1.08883f};
    float f[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) f[c] = s[c] > 0.008856f ? powf(s[c], 1.f / 3.f) : 7.787f * s[c] + 16.f / 116.f;
    float* o = lab + b * 3 * plane + r;
    o[0] = ((116.f * f[1] - 16.f) - l_cent) / l_norm;
    o[plane] = (500.f * (f[0] - f[1])) / ab_norm;
    o[2 * plane] = (200.f * (f[1] - f[2])) / ab_norm;
  }
}

static unsigned grid_for(long long total) {
  long long blocks = (total + TD_THREADS - 1) / TD_THREADS;
  const long long cap = 256 * 16;            // 16 blocks of 4 waves per CU, grid-stride beyond that
  return (unsigned)(blocks < cap ? blocks : cap);
}

template <typename T>
static int run_l1map(bool fwd, const void* pred, Strides4 sp, const float* target, int B, int C, int H, int W, float weight,
                     const float* gmap, float* out, void* dpred, hipStream_t st) {
  const unsigned g = grid_for((long long)B * H * W);
  if (fwd)
    hipLaunchKernelGGL((l1map_kernel<T, true>), dim3(g), dim3(TD_THREADS), 0, st, (const T*)pred, sp, target, B, C, H, W, weight,
                       gmap, out, (T*)dpred);
  else
    hipLaunchKernelGGL((l1map_kernel<T, false>), dim3(g), dim3(TD_THREADS), 0, st, (const T*)pred, sp, target, B, C, H, W, weight,
                       gmap, out, (T*)dpred);
  return record_launch_error(hipGetLastError(), fwd ? "td_l1map_fwd" : "td_l1map_bwd");
}

}  // namespace td

static int l1map_check(const void* pred, const float* target, const long long* strides, int B, int C, int H, int W) {
  if (!pred || !target || !strides || B <= 0 || C <= 0 || H <= 0 || W <= 0) return TD_ERR_BAD_ARG;
  if (C > 64 || (long long)B * C * H * W >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  return TD_OK;
}

extern "C" int td_l1map_fwd(const void* pred, int dtype, const long long* pred_strides, const float* target, int B, int C, int H,
                            int W, float weight, float* out, td_stream_t stream) {
  const int rc = l1map_check(pred, target, pred_strides, B, C, H, W);
  if (rc != TD_OK || !out) return rc != TD_OK ? rc : TD_ERR_BAD_ARG;
  const td::Strides4 sp{pred_strides[0], pred_strides[1], pred_strides[2], pred_strides[3]};
  if (dtype == TD_DTYPE_BF16)
    return td::run_l1map<__hip_bfloat16>(true, pred, sp, target, B, C, H, W, weight, nullptr, out, nullptr, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_l1map<float>(true, pred, sp, target, B, C, H, W, weight, nullptr, out, nullptr, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_l1map_bwd(const void* pred, int dtype, const long long* pred_strides, const float* target, const float* gmap,
                            int B, int C, int H, int W, float weight, void* dpred, td_stream_t stream) {
  const int rc = l1map_check(pred, target, pred_strides, B, C, H, W);
  if (rc != TD_OK || !gmap || !dpred) return rc != TD_OK ? rc : TD_ERR_BAD_ARG;
  const td::Strides4 sp{pred_strides[0], pred_strides[1], pred_strides[2], pred_strides[3]};
  if (dtype == TD_DTYPE_BF16)
    return td::run_l1map<__hip_bfloat16>(false, pred, sp, target, B, C, H, W, weight, gmap, nullptr, dpred, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_l1map<float>(false, pred, sp, target, B, C, H, W, weight, gmap, nullptr, dpred, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_rgb2lab(const float* rgb, int B, int H, int W, float l_cent, float l_norm, float ab_norm, float* lab,
                          td_stream_t stream) {
  if (!rgb || !lab || B <= 0 || H <= 0 || W <= 0 || l_norm == 0.f || ab_norm == 0.f) return TD_ERR_BAD_ARG;
  const long long plane = (long long)H * W;
  hipLaunchKernelGGL(td::rgb2lab_kernel, dim3(td::grid_for((long long)B * plane)), dim3(TD_THREADS), 0, (hipStream_t)stream, rgb, B,
                     plane, l_cent, l_norm, ab_norm, lab);
  return td::record_launch_error(hipGetLastError(), "td_rgb2lab");
}

// ---------------------------------------------------------------------------------------------------------------------
// NCHW [B,3,H,W] colour frame -> RGBX [B,H,W,4] (x = 0): the pixel format the photometric kernels read (td_common.h, "packed
// frames").  Once per frame and step; 12 B read + 16 B written per pixel.
namespace td {
__global__ __launch_bounds__(TD_THREADS) void pack_rgbx_kernel(const float* __restrict__ img, long long plane, long long total,
                                                              float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * TD_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * TD_THREADS) {
    const long long b = i / plane, r = i - b * plane;
    const float* p = img + b * 3 * plane + r;
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(p[0], p[plane], p[2 * plane], 0.f);
  }
}
}  // namespace td

extern "C" int td_pack_rgbx(const float* img, int B, int H, int W, float* out, td_stream_t stream) {
  if (!img || !out || B <= 0 || H <= 0 || W <= 0) return TD_ERR_BAD_ARG;
  const long long plane = (long long)H * W, total = (long long)B * plane;
  long long blocks = (total + TD_THREADS - 1) / TD_THREADS;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(td::pack_rgbx_kernel, dim3((unsigned)blocks), dim3(TD_THREADS), 0, (hipStream_t)stream, img, plane, total, out);
  return td::record_launch_error(hipGetLastError(), "td_pack_rgbx");
}
