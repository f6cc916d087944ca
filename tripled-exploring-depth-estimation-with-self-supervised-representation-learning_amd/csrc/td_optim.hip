// One pass for the optimiser step of the flat parameter store (tripled_amd/flat_amp.py): gradient clipping (the scale of
// torch.nn.utils.clip_grad_norm_), Adam (torch.optim.Adam, weight_decay = 0, no amsgrad) and the refresh of the bf16 working copy
// of the convolution weights.  Reference: the optimiser hook of the training loop, mono/core/utils/dist_utils.py:54-60
// (clip_grad_norm_ + optimizer.step()) with the config's Adam (config/cfg_kitti_tripleD.py: optimizer = dict(type='Adam', ...)).
//
//   c      = min(1, max_norm / (total_norm + 1e-6))          (clip_grad_norm_; 1 when total_norm is NULL)
//   g      = c * grad
//   m      = m + (1 - beta1) * (g - m)                        (torch's lerp form)
//   v      = beta2 * v + (1 - beta2) * g * g
//   w     -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
//   lp[i]  = bf16(w[i])  for i < n_lp
//
// ATen runs this as four passes (scale the gradients, multi-tensor Adam, cast): 2.9 GB of traffic for the 57 M parameters of
// cfg_kitti_tripleD against the 1.7 GB of one pass (30 B per parameter: HBM-bound).
#include <hip/hip_bf16.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

__global__ __launch_bounds__(TD_THREADS) void adam_flat_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                               float* __restrict__ v, __hip_bfloat16* __restrict__ lp, long long n4,
                                                               long long n_lp4, const float* __restrict__ step, const float* __restrict__ lr_dev,
                                                               float lr_host, float beta1, float beta2, float one_m_b1, float one_m_b2, float eps,
                                                               const float* __restrict__ total_norm, float max_norm) {
  const float t = step[0];
  const float lr = lr_dev ? lr_dev[0] : lr_host;
  const float bc1 = 1.f - powf(beta1, t), bc2 = 1.f - powf(beta2, t);
  const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
  float clip = 1.f;
  if (total_norm) {
    const float c = max_norm / (total_norm[0] + 1e-6f);
    clip = c < 1.f ? c : 1.f;
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i], wv = reinterpret_cast<float4*>(w)[i];
    float gg[4] = {gv.x * clip, gv.y * clip, gv.z * clip, gv.w * clip};
    float mm[4] = {mv.x, mv.y, mv.z, mv.w}, vq[4] = {vv.x, vv.y, vv.z, vv.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mm[e] = mm[e] + one_m_b1 * (gg[e] - mm[e]);
      vq[e] = beta2 * vq[e] + one_m_b2 * gg[e] * gg[e];
      const float denom = sqrtf(vq[e]) / bc2_sqrt + eps;
      ww[e] -= step_size * mm[e] / denom;
    }
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(vq[0], vq[1], vq[2], vq[3]);
    reinterpret_cast<float4*>(w)[i] = make_float4(ww[0], ww[1], ww[2], ww[3]);
    if (i < n_lp4) {
      uint2 o;
      o.x = (unsigned)f2bf(ww[0]) | ((unsigned)f2bf(ww[1]) << 16);
      o.y = (unsigned)f2bf(ww[2]) | ((unsigned)f2bf(ww[3]) << 16);
      reinterpret_cast<uint2*>(lp)[i] = o;
    }
  }
}

}  // namespace td

extern "C" int td_adam_flat(float* w, const float* grad, float* exp_avg, float* exp_avg_sq, void* lowp, long long n, long long n_lowp,
                            const float* step, const float* lr_dev, float lr_host, double beta1_d, double beta2_d, float eps,
                            const float* total_norm, float max_norm, td_stream_t stream) {
  // betas travel as doubles: torch forms 1 - beta in double (1 - 0.999 = 1e-3, not the 1.00005e-3 of float(0.999))
  const float beta1 = (float)beta1_d, beta2 = (float)beta2_d;
  if (!w || !grad || !exp_avg || !exp_avg_sq || !step || n <= 0 || n_lowp < 0 || n_lowp > n || (n_lowp > 0 && !lowp)) return TD_ERR_BAD_ARG;
  if (n % 4 != 0 || n_lowp % 4 != 0) return TD_ERR_UNSUPPORTED;      // the flat store aligns every parameter to 8 elements
  const long long n4 = n / 4;
  long long blocks = (n4 + TD_THREADS - 1) / TD_THREADS;
  if (blocks > 4096) blocks = 4096;                                   // 16 x 256 CUs: grid-stride over the rest
  hipLaunchKernelGGL(td::adam_flat_kernel, dim3((unsigned)blocks), dim3(TD_THREADS), 0, (hipStream_t)stream, w, grad, exp_avg, exp_avg_sq,
                     (__hip_bfloat16*)lowp, n4, n_lowp / 4, step, lr_dev, lr_host, beta1, beta2, (float)(1.0 - (double)beta1_d), (float)(1.0 - (double)beta2_d), eps,
                     total_norm, max_norm);
  return td::record_launch_error(hipGetLastError(), "td_adam_flat");
}

// ---------------------------------------------------------------------------------------------------------------------
// Gradient gather of the flat parameter store: the per-parameter gradients autograd allocated (bf16 for the convolutions, fp32
// for the rest, each in its parameter's memory order) -> the flat fp32 gradient buffer, up to 64 tensors per launch with the
// pointers in the kernel arguments (captured by value in a HIP graph).  Replaces torch.cat over ~320 tensors (four batched-copy
// kernels, 405 us per step) + the bf16 -> fp32 pass over the result (reference: nothing -- the reference's optimiser walks the
// per-parameter gradients; this is the flat store's own plumbing, tripled_amd/flat_amp.py::collect).
namespace td {

constexpr int GG_MAX = 64;                 // tensors per launch
constexpr int GG_SEG = 8192;               // elements per block
struct GatherArgs {
  const void* src[GG_MAX];                 // NULL: the slot is zero-filled (a parameter no gradient reached)
  long long dst[GG_MAX];                   // element offset in the flat buffer
  long long numel[GG_MAX];
  int first_block[GG_MAX + 1];             // prefix sums of ceil(numel / GG_SEG)
  int n;
};

template <typename T>
__global__ __launch_bounds__(TD_THREADS) void gather_flat_kernel(const GatherArgs a, float* __restrict__ flat) {
  // which tensor does this block belong to: binary search over <= 65 prefix sums
  int lo = 0, hi = a.n;
  const int b = (int)blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.first_block[mid] <= b) lo = mid; else hi = mid;
  }
  const int t = lo;
  const long long e0 = (long long)(b - a.first_block[t]) * GG_SEG;
  const long long n = a.numel[t];
  const long long e1 = e0 + GG_SEG < n ? e0 + GG_SEG : n;
  const T* __restrict__ s = reinterpret_cast<const T*>(a.src[t]);
  float* __restrict__ d = flat + a.dst[t];
  // 8 elements per thread and iteration where both sides are 16-byte aligned, scalar otherwise (tiny tensors, odd offsets)
  const bool vec = s && (reinterpret_cast<size_t>(s) % 16 == 0) && (a.dst[t] % 8 == 0) && (e0 % 8 == 0);
  if (vec) {
    const long long ev = e0 + ((e1 - e0) / 8) * 8;
    for (long long i = e0 + (long long)threadIdx.x * 8; i < ev; i += TD_THREADS * 8) {
      float v[8];
      load8(s + i, v);
      store8(d + i, v);
    }
    for (long long i = ev + threadIdx.x; i < e1; i += TD_THREADS) d[i] = sizeof(T) == 2 ? bf2f(reinterpret_cast<const unsigned short*>(s)[i]) : reinterpret_cast<const float*>(s)[i];
  } else {
    for (long long i = e0 + threadIdx.x; i < e1; i += TD_THREADS)
      d[i] = !s ? 0.f : (sizeof(T) == 2 ? bf2f(reinterpret_cast<const unsigned short*>(s)[i]) : reinterpret_cast<const float*>(s)[i]);
  }
}

}  // namespace td

extern "C" int td_gather_flat(const void* const* srcs, const long long* dst_offsets, const long long* numels, int n, int src_dtype,
                              float* flat, td_stream_t stream) {
  if (!dst_offsets || !numels || !flat || n < 0 || (n > 0 && !srcs)) return TD_ERR_BAD_ARG;
  if (src_dtype != TD_DTYPE_BF16 && src_dtype != TD_DTYPE_F32) return TD_ERR_UNSUPPORTED;
  for (int base = 0; base < n; base += td::GG_MAX) {
    td::GatherArgs a;
    a.n = n - base < td::GG_MAX ? n - base : td::GG_MAX;
    int blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      if (numels[base + i] < 0 || dst_offsets[base + i] < 0) return TD_ERR_BAD_ARG;
      a.src[i] = srcs[base + i];
      a.dst[i] = dst_offsets[base + i];
      a.numel[i] = numels[base + i];
      a.first_block[i] = blocks;
      blocks += (int)((numels[base + i] + td::GG_SEG - 1) / td::GG_SEG);
    }
    for (int i = a.n; i <= td::GG_MAX; ++i) a.first_block[i] = blocks;
    if (blocks == 0) continue;
    if (src_dtype == TD_DTYPE_BF16)
      hipLaunchKernelGGL((td::gather_flat_kernel<__hip_bfloat16>), dim3((unsigned)blocks), dim3(TD_THREADS), 0, (hipStream_t)stream, a, flat);
    else
      hipLaunchKernelGGL((td::gather_flat_kernel<float>), dim3((unsigned)blocks), dim3(TD_THREADS), 0, (hipStream_t)stream, a, flat);
  }
  return td::record_launch_error(hipGetLastError(), "td_gather_flat");
}
