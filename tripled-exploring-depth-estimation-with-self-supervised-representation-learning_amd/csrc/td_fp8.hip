// Per-tensor fp8 (OCP e4m3fn, the MI355X MFMA fp8 input format) quantisation for the fp8 1x1-convolution path of the
// all-aux-heads configuration (BASELINE config 5: cfg_kitti_fm_joint_inpaint_disentangle_distill_full_colorize,
// "fp8 MFMA conv path"; the reference itself trains in fp32, mono/apis/trainer.py).  A 1x1 stride-1 convolution on
// channels-last activations is the GEMM [N*H*W, Cin] x [Cin, Cout]; it runs on the fp8 MFMA through hipBLASLt
// (torch._scaled_mm, a plain library GEMM) and this file supplies what surrounds it:
//   td_fp8_amax_partials   per-block max |x|                       (one read of x)
//   td_fp8_quantize        scale = 448 / amax (current scaling), q = sat_e4m3(x * scale), inv_scale = amax / 448
//                          (one read of x, a half-size write); the partials are re-reduced by every block, so there is
//                          no host round trip and no third launch
#include <hip/hip_bf16.h>
#include <hip/hip_fp8.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

constexpr int FP8_ELEMS_PER_BLOCK = TD_THREADS * 8 * 4;      // 8 elements per thread, 4 iterations
constexpr float FP8_E4M3_MAX = 448.f;

template <typename T>
__global__ __launch_bounds__(TD_THREADS) void fp8_amax_kernel(const T* __restrict__ x, long long n, float* __restrict__ partials) {
  __shared__ float red[4];
  const long long base = (long long)blockIdx.x * FP8_ELEMS_PER_BLOCK;
  float m = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long long i = base + ((long long)it * TD_THREADS + threadIdx.x) * 8;
    if (i + 8 <= n) {
      float v[8];
      load8(x + i, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(v[k]));
    } else {
      for (long long j = i; j < n; ++j) m = fmaxf(m, fabsf((float)x[j]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <typename T>
__global__ __launch_bounds__(TD_THREADS) void fp8_quant_kernel(const T* __restrict__ x, long long n, const float* __restrict__ partials,
                                                               int nblk, unsigned char* __restrict__ q, float* __restrict__ inv_scale) {
  __shared__ float red[4];
  __shared__ float s_scale;
  float m = 0.f;
  for (int i = threadIdx.x; i < nblk; i += TD_THREADS) m = fmaxf(m, partials[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float amax = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-12f);
    s_scale = FP8_E4M3_MAX / amax;
    if (blockIdx.x == 0) inv_scale[0] = amax / FP8_E4M3_MAX;
  }
  __syncthreads();
  const float scale = s_scale;
  const long long base = (long long)blockIdx.x * FP8_ELEMS_PER_BLOCK;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long long i = base + ((long long)it * TD_THREADS + threadIdx.x) * 8;
    if (i + 8 <= n) {
      float v[8];
      load8(x + i, v);
      unsigned long long packed = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        packed |= (unsigned long long)__hip_cvt_float_to_fp8(v[k] * scale, __HIP_SATFINITE, __HIP_E4M3) << (8 * k);
      *reinterpret_cast<unsigned long long*>(q + i) = packed;
    } else {
      for (long long j = i; j < n; ++j) q[j] = __hip_cvt_float_to_fp8((float)x[j] * scale, __HIP_SATFINITE, __HIP_E4M3);
    }
  }
}

}  // namespace td

extern "C" int td_fp8_num_blocks(long long n) {
  if (n <= 0) return 0;
  return (int)((n + td::FP8_ELEMS_PER_BLOCK - 1) / td::FP8_ELEMS_PER_BLOCK);
}

extern "C" int td_fp8_amax_partials(const void* x, int dtype, long long n, float* partials, td_stream_t stream) {
  if (!x || !partials || n <= 0) return TD_ERR_BAD_ARG;
  if (n >= (1ll << 40) || n % 8 != 0) return TD_ERR_UNSUPPORTED;       // (8-element vectors: the GEMM needs K % 16 == 0 anyway)
  const int nblk = td_fp8_num_blocks(n);
  if (dtype == TD_DTYPE_BF16)
    hipLaunchKernelGGL((td::fp8_amax_kernel<__hip_bfloat16>), dim3(nblk), dim3(TD_THREADS), 0, (hipStream_t)stream,
                       (const __hip_bfloat16*)x, n, partials);
  else if (dtype == TD_DTYPE_F32)
    hipLaunchKernelGGL((td::fp8_amax_kernel<float>), dim3(nblk), dim3(TD_THREADS), 0, (hipStream_t)stream, (const float*)x, n, partials);
  else
    return TD_ERR_UNSUPPORTED;
  return td::record_launch_error(hipGetLastError(), "td_fp8_amax_partials");
}

extern "C" int td_fp8_quantize(const void* x, int dtype, long long n, const float* partials, uint8_t* q, float* inv_scale,
                               td_stream_t stream) {
  if (!x || !partials || !q || !inv_scale || n <= 0) return TD_ERR_BAD_ARG;
  if (n >= (1ll << 40) || n % 8 != 0) return TD_ERR_UNSUPPORTED;
  const int nblk = td_fp8_num_blocks(n);
  if (dtype == TD_DTYPE_BF16)
    hipLaunchKernelGGL((td::fp8_quant_kernel<__hip_bfloat16>), dim3(nblk), dim3(TD_THREADS), 0, (hipStream_t)stream,
                       (const __hip_bfloat16*)x, n, partials, nblk, q, inv_scale);
  else if (dtype == TD_DTYPE_F32)
    hipLaunchKernelGGL((td::fp8_quant_kernel<float>), dim3(nblk), dim3(TD_THREADS), 0, (hipStream_t)stream, (const float*)x, n, partials,
                       nblk, q, inv_scale);
  else
    return TD_ERR_UNSUPPORTED;
  return td::record_launch_error(hipGetLastError(), "td_fp8_quantize");
}
