// nn.ReflectionPad2d(1) on channels-last activations, forward and adjoint (the pad in front of every
// decoder Conv3x3, reference mono/model/mono_fm_joint/layers.py:171-184).  ATen's backward scatters with
// atomics; here each input element gathers the <= 4 padded positions that mirror onto it.
#include <hip/hip_bf16.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

template <typename T>
__device__ __forceinline__ void copy16(const T* src, T* dst) {      // 8 channels, bit copy
  constexpr int NV = sizeof(T) * 8 / 16;
#pragma unroll
  for (int i = 0; i < NV; ++i) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
}

__device__ __forceinline__ int refl(int i, int n) { i = i < 0 ? -i : i; return i >= n ? 2 * n - 2 - i : i; }

template <typename T>
__global__ __launch_bounds__(TD_THREADS) void reflpad1_fwd_kernel(const T* __restrict__ in, int N, int H, int W, int C,
                                                                  T* __restrict__ out) {
  const int c8 = C >> 3, Ho = H + 2, Wo = W + 2;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  if (gid >= (long long)N * Ho * Wo * c8) return;
  const int cv = (int)(gid % c8);
  const long long pix = gid / c8;
  const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), n = (int)(pix / ((long long)Wo * Ho));
  const int xi = refl(xo - 1, W), yi = refl(yo - 1, H);
  float v[8];
  load8(in + (((size_t)n * H + yi) * W + xi) * C + (size_t)cv * 8, v);
  store8(out + (size_t)pix * C + (size_t)cv * 8, v);
}

template <typename T>
__global__ __launch_bounds__(TD_THREADS) void reflpad1_bwd_kernel(const T* __restrict__ gout, int N, int H, int W, int C,
                                                                  T* __restrict__ gin) {
  const int c8 = C >> 3, Ho = H + 2, Wo = W + 2;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  if (gid >= (long long)N * H * W * c8) return;
  const int cv = (int)(gid % c8);
  const long long pix = gid / c8;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
  // padded coordinates that map onto (y, x): always (y+1, x+1); plus the mirror row/column at the borders
  int ys[2] = {y + 1, -1}, xs[2] = {x + 1, -1};
  if (y == 1) ys[1] = 0; else if (y == H - 2) ys[1] = Ho - 1;
  if (x == 1) xs[1] = 0; else if (x == W - 2) xs[1] = Wo - 1;
  // (H == 3: y == 1 is both 1 and H-2 -> two mirrors)
  const int ys2 = (H == 3 && y == 1) ? Ho - 1 : -1, xs2 = (W == 3 && x == 1) ? Wo - 1 : -1;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const T* base = gout + (size_t)n * Ho * Wo * C + (size_t)cv * 8;
  const int yl[3] = {ys[0], ys[1], ys2}, xl[3] = {xs[0], xs[1], xs2};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (yl[a] < 0) continue;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      if (xl[b] < 0) continue;
      float v[8];
      load8(base + ((size_t)yl[a] * Wo + xl[b]) * C, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
  }
  store8(gin + (size_t)pix * C + (size_t)cv * 8, acc);
}

// ---------------------------------------------------------------------------------------------
// x2 nearest up-sampling fused into the pad: out = ReflectionPad2d(1)(interpolate(in, scale 2, "nearest")), the
// `iconv(upsample(upconv(x)))` step of the image decoders (reference: decoder.py:40-57, layers.py upsample).
// The up-sampled tensor is never materialised (ATen's NHWC nearest kernel alone costs 49 us per call on average).
template <typename T>
__global__ __launch_bounds__(TD_THREADS) void up2_reflpad1_fwd_kernel(const T* __restrict__ in, int N, int H, int W, int C,
                                                                      T* __restrict__ out) {
  const int c8 = C >> 3, Hu = 2 * H, Wu = 2 * W, Ho = Hu + 2, Wo = Wu + 2;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  if (gid >= (long long)N * Ho * Wo * c8) return;
  const int cv = (int)(gid % c8);
  const long long pix = gid / c8;
  const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), n = (int)(pix / ((long long)Wo * Ho));
  const int xi = refl(xo - 1, Wu) >> 1, yi = refl(yo - 1, Hu) >> 1;
  copy16(in + (((size_t)n * H + yi) * W + xi) * C + (size_t)cv * 8, out + (size_t)pix * C + (size_t)cv * 8);
}

template <typename T>
__global__ __launch_bounds__(TD_THREADS) void up2_reflpad1_bwd_kernel(const T* __restrict__ gout, int N, int H, int W, int C,
                                                                      T* __restrict__ gin) {
  const int c8 = C >> 3, Ho = 2 * H + 2, Wo = 2 * W + 2;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  if (gid >= (long long)N * H * W * c8) return;
  const int cv = (int)(gid % c8);
  const long long pix = gid / c8;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
  // padded rows whose source is y: the two up-sampled copies, plus the mirrored border row
  // (padded row 0 mirrors up-row 1 -> source 0; padded row Ho-1 mirrors up-row 2H-2 -> source H-1)
  const int yl[4] = {2 * y + 1, 2 * y + 2, y == 0 ? 0 : -1, y == H - 1 ? Ho - 1 : -1};
  const int xl[4] = {2 * x + 1, 2 * x + 2, x == 0 ? 0 : -1, x == W - 1 ? Wo - 1 : -1};
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const T* base = gout + (size_t)n * Ho * Wo * C + (size_t)cv * 8;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (yl[a] < 0) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (xl[b] < 0) continue;
      float v[8];
      load8(base + ((size_t)yl[a] * Wo + xl[b]) * C, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
  }
  store8(gin + (size_t)pix * C + (size_t)cv * 8, acc);
}

}  // namespace td

extern "C" int td_reflpad1_fwd(const void* in, int dtype, int N, int H, int W, int C, void* out, td_stream_t stream) {
  if (!in || !out || N <= 0 || C <= 0) return TD_ERR_BAD_ARG;
  if (C % 8 != 0 || H < 2 || W < 2) return TD_ERR_UNSUPPORTED;
  const long long total = (long long)N * (H + 2) * (W + 2) * (C / 8);
  const unsigned blocks = (unsigned)((total + TD_THREADS - 1) / TD_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_DTYPE_BF16)
    hipLaunchKernelGGL((td::reflpad1_fwd_kernel<__hip_bfloat16>), dim3(blocks), dim3(TD_THREADS), 0, st,
                       (const __hip_bfloat16*)in, N, H, W, C, (__hip_bfloat16*)out);
  else if (dtype == TD_DTYPE_F32)
    hipLaunchKernelGGL((td::reflpad1_fwd_kernel<float>), dim3(blocks), dim3(TD_THREADS), 0, st, (const float*)in, N, H, W, C, (float*)out);
  else
    return TD_ERR_UNSUPPORTED;
  return td::record_launch_error(hipGetLastError(), "td_reflpad1_fwd");
}

extern "C" int td_reflpad1_bwd(const void* grad_out, int dtype, int N, int H, int W, int C, void* grad_in,
                               td_stream_t stream) {
  if (!grad_out || !grad_in || N <= 0 || C <= 0) return TD_ERR_BAD_ARG;
  if (C % 8 != 0 || H < 2 || W < 2) return TD_ERR_UNSUPPORTED;
  const long long total = (long long)N * H * W * (C / 8);
  const unsigned blocks = (unsigned)((total + TD_THREADS - 1) / TD_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_DTYPE_BF16)
    hipLaunchKernelGGL((td::reflpad1_bwd_kernel<__hip_bfloat16>), dim3(blocks), dim3(TD_THREADS), 0, st,
                       (const __hip_bfloat16*)grad_out, N, H, W, C, (__hip_bfloat16*)grad_in);
  else if (dtype == TD_DTYPE_F32)
    hipLaunchKernelGGL((td::reflpad1_bwd_kernel<float>), dim3(blocks), dim3(TD_THREADS), 0, st, (const float*)grad_out, N, H, W, C, (float*)grad_in);
  else
    return TD_ERR_UNSUPPORTED;
  return td::record_launch_error(hipGetLastError(), "td_reflpad1_bwd");
}

template <bool FWD>
static int run_up2_pad(const void* in, int dtype, int N, int H, int W, int C, void* out, td_stream_t stream, const char* what) {
  if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return TD_ERR_BAD_ARG;
  if (C % 8 != 0) return TD_ERR_UNSUPPORTED;
  const long long total = FWD ? (long long)N * (2 * H + 2) * (2 * W + 2) * (C / 8) : (long long)N * H * W * (C / 8);
  const unsigned blocks = (unsigned)((total + TD_THREADS - 1) / TD_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_DTYPE_BF16) {
    using T = __hip_bfloat16;
    if (FWD) hipLaunchKernelGGL((td::up2_reflpad1_fwd_kernel<T>), dim3(blocks), dim3(TD_THREADS), 0, st, (const T*)in, N, H, W, C, (T*)out);
    else hipLaunchKernelGGL((td::up2_reflpad1_bwd_kernel<T>), dim3(blocks), dim3(TD_THREADS), 0, st, (const T*)in, N, H, W, C, (T*)out);
  } else if (dtype == TD_DTYPE_F32) {
    if (FWD) hipLaunchKernelGGL((td::up2_reflpad1_fwd_kernel<float>), dim3(blocks), dim3(TD_THREADS), 0, st, (const float*)in, N, H, W, C, (float*)out);
    else hipLaunchKernelGGL((td::up2_reflpad1_bwd_kernel<float>), dim3(blocks), dim3(TD_THREADS), 0, st, (const float*)in, N, H, W, C, (float*)out);
  } else {
    return TD_ERR_UNSUPPORTED;
  }
  return td::record_launch_error(hipGetLastError(), what);
}

extern "C" int td_up2_reflpad1_fwd(const void* in, int dtype, int N, int H, int W, int C, void* out, td_stream_t stream) {
  return run_up2_pad<true>(in, dtype, N, H, W, C, out, stream, "td_up2_reflpad1_fwd");
}

extern "C" int td_up2_reflpad1_bwd(const void* grad_out, int dtype, int N, int H, int W, int C, void* grad_in,
                                   td_stream_t stream) {
  return run_up2_pad<false>(grad_out, dtype, N, H, W, C, grad_in, stream, "td_up2_reflpad1_bwd");
}
