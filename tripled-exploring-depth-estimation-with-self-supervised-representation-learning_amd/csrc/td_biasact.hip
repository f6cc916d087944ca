// Bias + activation behind the decoders' convolutions and their adjoint with the bias gradient, on channels-last activations
// (reference: ConvBlock = Conv3x3 -> ELU, mono/model/mono_fm_joint/layers.py:143-155; the CRP DepthDecoder's
// F.leaky_relu(iconv / merge), mono/model/mono_fm_joint/depth_decoder.py:89-103):
//
//   forward   a = act(y + bias[c])                      one pass (MIOpen adds a bias with a tensor-op kernel of its own and ATen
//                                                       runs the activation as another)
//   backward  gy = g * act'(a)  and  dbias[c] = sum gy  one pass + a finish launch (ATen: the activation's backward pass plus a
//                                                       15 us column reduction per layer for the bias gradient)
//
// act: 0 none, 1 ELU (alpha 1; derivative from the result: a > 0 ? 1 : a + 1, torch's in-place form), 2 leaky ReLU (slope 0.01;
// the result has the sign of the input).  Tensors are [M, C] rows, C % 8 == 0, 256 % (C / 8) == 0; a thread always meets the same
// eight channels (its 16-byte vector index stays congruent mod C / 8), so the bias gradient is accumulated in registers, reduced
// per block through LDS in a fixed order and finished by one block per 16 channels: deterministic.
#include <hip/hip_bf16.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

template <int ACT>
__device__ __forceinline__ float ba_fwd(float z) {
  if (ACT == 1) return z > 0.f ? z : expm1f(z);
  if (ACT == 2) return z > 0.f ? z : 0.01f * z;
  return z;
}
template <int ACT>
__device__ __forceinline__ float ba_bwd(float g, float a) {
  if (ACT == 1) return a > 0.f ? g : g * (a + 1.f);
  if (ACT == 2) return a > 0.f ? g : 0.01f * g;
  return g;
}

template <typename T, int ACT>
__global__ __launch_bounds__(TD_THREADS) void bias_act_fwd_kernel(const T* __restrict__ y, const void* __restrict__ bias, int bias_bf16,
                                                                  long long nvec, int cvec, T* __restrict__ a) {
  const long long stride = (long long)gridDim.x * TD_THREADS;
  long long i = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  const int cg = (int)(i % cvec);
  float b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e)
    b[e] = !bias ? 0.f : bias_bf16 ? bf2f(reinterpret_cast<const unsigned short*>(bias)[cg * 8 + e]) : reinterpret_cast<const float*>(bias)[cg * 8 + e];
  for (; i < nvec; i += stride) {
    float v[8];
    load8(y + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ba_fwd<ACT>(v[e] + b[e]);
    store8(a + i * 8, v);
  }
}

template <typename T, int ACT>
__global__ __launch_bounds__(TD_THREADS) void bias_act_bwd_kernel(const T* __restrict__ g, const T* __restrict__ a, long long nvec, int cvec,
                                                                  T* __restrict__ gy, float* __restrict__ partial) {
  __shared__ float red[TD_THREADS][9];
  const long long stride = (long long)gridDim.x * TD_THREADS;
  long long i = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  for (; i < nvec; i += stride) {
    float vg[8], va[8];
    load8(g + i * 8, vg);
    if (ACT != 0) load8(a + i * 8, va);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      vg[e] = ba_bwd<ACT>(vg[e], ACT != 0 ? va[e] : 0.f);
      s[e] += vg[e];
    }
    if (ACT != 0) store8(gy + i * 8, vg);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  const int C = cvec * 8;
  if ((int)threadIdx.x < C) {                 // channel c: threads cg, cg + cvec, ... hold its partial sums
    const int c = threadIdx.x, cg = c >> 3, e = c & 7;
    float t = 0.f;
    for (int j = cg; j < TD_THREADS; j += cvec) t += red[j][e];
    partial[(size_t)blockIdx.x * C + c] = t;
  }
}

// dbias[c] = ordered sum of the per-block partial rows: 16 channels x 16 row slices per block, the slices' loads issued eight at a
// time (a one-thread-per-channel loop over 512 rows took 42 us: one L2 round trip per row)
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_finish_kernel(const float* __restrict__ partial, int blocks, int C, T* __restrict__ db) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, part = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  float t = 0.f;
  if (c < C) {
    int b = part;
    for (; b + 7 * 16 < blocks; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(b + 16 * u) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; b < blocks; b += 16) t += partial[(size_t)b * C + c];
  }
  red[part][cl] = t;
  __syncthreads();
  if (part == 0 && c < C) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += red[j][cl];
    if constexpr (sizeof(T) == 2) db[c] = __float2bfloat16(s);
    else db[c] = s;
  }
}

static inline int ba_blocks(long long nvec) {
  long long b = (nvec + TD_THREADS * 8 - 1) / (TD_THREADS * 8);      // >= 8 vectors per thread
  if (b > 512) b = 512;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace td

static bool ba_shape_ok(long long M, int C, int act) {
  return M > 0 && C >= 8 && C <= 256 && C % 8 == 0 && TD_THREADS % (C / 8) == 0 && act >= 0 && act <= 2 && M * (long long)C < (1ll << 40);
}

extern "C" long long td_bias_act_workspace_floats(long long M, int C) {
  if (!ba_shape_ok(M, C, 0)) return 0;
  return (long long)td::ba_blocks(M * (C / 8)) * C;
}

extern "C" int td_bias_act_fwd(const void* y, const void* bias, int bias_dtype, int dtype, long long M, int C, int act, void* a,
                               td_stream_t stream) {
  if (!y || !a) return TD_ERR_BAD_ARG;
  if (!ba_shape_ok(M, C, act) || (dtype != TD_DTYPE_BF16 && dtype != TD_DTYPE_F32)) return TD_ERR_UNSUPPORTED;
  const long long nvec = M * (C / 8);
  const int blocks = td::ba_blocks(nvec) * 4 > 2048 ? 2048 : td::ba_blocks(nvec) * 4, cvec = C / 8;
  hipStream_t st = (hipStream_t)stream;
  const int bb = bias_dtype == TD_DTYPE_BF16;
#define TD_BA_F(T, ACT) hipLaunchKernelGGL((td::bias_act_fwd_kernel<T, ACT>), dim3(blocks), dim3(TD_THREADS), 0, st, (const T*)y, bias, bb, nvec, cvec, (T*)a)
  if (dtype == TD_DTYPE_BF16) { if (act == 0) TD_BA_F(__hip_bfloat16, 0); else if (act == 1) TD_BA_F(__hip_bfloat16, 1); else TD_BA_F(__hip_bfloat16, 2); }
  else { if (act == 0) TD_BA_F(float, 0); else if (act == 1) TD_BA_F(float, 1); else TD_BA_F(float, 2); }
#undef TD_BA_F
  return td::record_launch_error(hipGetLastError(), "td_bias_act_fwd");
}

extern "C" int td_bias_act_bwd(const void* g, const void* a, int dtype, long long M, int C, int act, void* gy, void* dbias, int dbias_dtype,
                               float* workspace, td_stream_t stream) {
  if (!g || !workspace || (act != 0 && (!a || !gy))) return TD_ERR_BAD_ARG;
  if (!ba_shape_ok(M, C, act) || (dtype != TD_DTYPE_BF16 && dtype != TD_DTYPE_F32)) return TD_ERR_UNSUPPORTED;
  if (dbias && dbias_dtype != TD_DTYPE_BF16 && dbias_dtype != TD_DTYPE_F32) return TD_ERR_UNSUPPORTED;
  const long long nvec = M * (C / 8);
  const int blocks = td::ba_blocks(nvec), cvec = C / 8;
  hipStream_t st = (hipStream_t)stream;
#define TD_BA_B(T, ACT) hipLaunchKernelGGL((td::bias_act_bwd_kernel<T, ACT>), dim3(blocks), dim3(TD_THREADS), 0, st, (const T*)g, (const T*)a, nvec, cvec, (T*)gy, workspace)
  if (dtype == TD_DTYPE_BF16) { if (act == 0) TD_BA_B(__hip_bfloat16, 0); else if (act == 1) TD_BA_B(__hip_bfloat16, 1); else TD_BA_B(__hip_bfloat16, 2); }
  else { if (act == 0) TD_BA_B(float, 0); else if (act == 1) TD_BA_B(float, 1); else TD_BA_B(float, 2); }
#undef TD_BA_B
  if (dbias) {
    if (dbias_dtype == TD_DTYPE_BF16)
      hipLaunchKernelGGL((td::bias_grad_finish_kernel<__hip_bfloat16>), dim3((C + 15) / 16), dim3(256), 0, st, (const float*)workspace, blocks, C, (__hip_bfloat16*)dbias);
    else
      hipLaunchKernelGGL((td::bias_grad_finish_kernel<float>), dim3((C + 15) / 16), dim3(256), 0, st, (const float*)workspace, blocks, C, (float*)dbias);
  }
  return td::record_launch_error(hipGetLastError(), "td_bias_act_bwd");
}
