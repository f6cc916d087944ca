// Channel concatenation of the DepthDecoder stages on channels-last activations
// (reference: mono/model/mono_fm_joint/depth_decoder.py:89-103, torch.cat((reduce_k(l_k), x, disp), 1)):
// two wide parts (C0, C1 multiples of 8) plus a narrow tail (C2 <= 8 channels, the 1-channel disparity),
// zero-padded to a multiple of 8 output channels so that the following 3x3 convolution takes MIOpen's
// NHWC fast path.  ATen's cat spends 528 us on the 48x160 stage (the 1- and 7-channel pieces push it onto
// an element-wise path); this is one 16-byte-per-lane pass each way.
#include <hip/hip_bf16.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

template <typename T>
__device__ __forceinline__ void copy8(const T* src, T* dst) {
  if (sizeof(T) == 2) {
    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
  } else {
    reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(src)[0];
    reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(src)[1];
  }
}

// FWD: out[pix, :] = [a[pix, :C0], b[pix, :C1], t[pix, :C2], 0...]      BWD: the three slices of gout
template <typename T, bool FWD>
__global__ __launch_bounds__(TD_THREADS) void join_kernel(T* __restrict__ a, T* __restrict__ b, T* __restrict__ t,
                                                          long long npix, int C0, int C1, int C2, T* __restrict__ out) {
  const int n0 = C0 >> 3, n1 = C1 >> 3, nc = n0 + n1 + 1;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  if (gid >= npix * nc) return;
  const int chunk = (int)(gid % nc);
  const long long pix = gid / nc;
  T* o = out + (size_t)pix * ((size_t)nc * 8) + (size_t)chunk * 8;
  if (chunk < n0) {
    T* p = a + (size_t)pix * C0 + (size_t)chunk * 8;
    if (FWD) copy8<T>(p, o); else copy8<T>(o, p);
  } else if (chunk < n0 + n1) {
    T* p = b + (size_t)pix * C1 + (size_t)(chunk - n0) * 8;
    if (FWD) copy8<T>(p, o); else copy8<T>(o, p);
  } else {
    T* p = t + (size_t)pix * C2;
    if (FWD) {
      alignas(16) T tail[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) tail[i] = i < C2 ? p[i < C2 ? i : 0] : T(0.f);
      copy8<T>(tail, o);
    } else {
      for (int i = 0; i < C2; ++i) p[i] = o[i];
    }
  }
}

// The same concatenation with the middle operand given at HALF resolution and up-sampled x2 (nearest) on the fly:
// torch.cat((reduce_k(l_k), upsample(x), disp), 1) of depth_decoder.py:89-103 without materialising upsample(x)
// (12 x 256 x 96 x 320 bf16 = 189 MB at the last stage of C2).  Backward: a and tail are slices of the gradient; the
// gradient of the low-resolution operand is the sum of the slices of its four output pixels (gather form).
template <typename T, bool FWD>
__global__ __launch_bounds__(TD_THREADS) void join_up2_kernel(T* __restrict__ a, T* __restrict__ b, T* __restrict__ t, int N, int H, int W,
                                                              int C0, int C1, int C2, T* __restrict__ out) {
  const int n0 = C0 >> 3, n1 = C1 >> 3, nc = n0 + n1 + 1;
  const long long npix = (long long)N * H * W;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  const int h2 = H >> 1, w2 = W >> 1;
  if (FWD) {
    if (gid >= npix * nc) return;
    const int chunk = (int)(gid % nc);
    const long long pix = gid / nc;
    T* o = out + (size_t)pix * ((size_t)nc * 8) + (size_t)chunk * 8;
    if (chunk < n0) {
      copy8<T>(a + (size_t)pix * C0 + (size_t)chunk * 8, o);
    } else if (chunk < n0 + n1) {
      const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
      const size_t src = ((size_t)n * h2 + (y >> 1)) * w2 + (x >> 1);
      copy8<T>(b + src * C1 + (size_t)(chunk - n0) * 8, o);
    } else {
      T* p = t + (size_t)pix * C2;
      alignas(16) T tail[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) tail[i] = i < C2 ? p[i < C2 ? i : 0] : T(0.f);
      copy8<T>(tail, o);
    }
    return;
  }
  // backward: [0, npix * (n0 + 1)) -> slices for a and tail;  then npix / 4 * n1 threads -> sums for b
  const long long first = npix * (n0 + 1);
  if (gid < first) {
    const int chunk = (int)(gid % (n0 + 1));
    const long long pix = gid / (n0 + 1);
    if (chunk < n0) {
      copy8<T>(out + (size_t)pix * ((size_t)nc * 8) + (size_t)chunk * 8, a + (size_t)pix * C0 + (size_t)chunk * 8);
    } else {
      const T* o = out + (size_t)pix * ((size_t)nc * 8) + (size_t)(n0 + n1) * 8;
      T* p = t + (size_t)pix * C2;
      for (int i = 0; i < C2; ++i) p[i] = o[i];
    }
    return;
  }
  const long long g2 = gid - first;
  const long long nlow = (long long)N * h2 * w2;
  if (g2 >= nlow * n1) return;
  const int chunk = (int)(g2 % n1);
  const long long lp = g2 / n1;
  const int x = (int)(lp % w2), y = (int)((lp / w2) % h2), n = (int)(lp / ((long long)w2 * h2));
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const size_t pix = ((size_t)n * H + (2 * y + dy)) * W + (2 * x + dx);
      float v[8];
      load8(out + pix * ((size_t)nc * 8) + (size_t)(n0 + chunk) * 8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
  store8(b + (size_t)lp * C1 + (size_t)chunk * 8, acc);
}

template <typename T>
static int run_join_up2(bool fwd, void* a, void* b, void* t, int N, int H, int W, int C0, int C1, int C2, void* out, hipStream_t st) {
  const long long npix = (long long)N * H * W;
  const int n0 = C0 / 8, n1 = C1 / 8;
  const long long total = fwd ? npix * (n0 + n1 + 1) : npix * (n0 + 1) + npix / 4 * n1;
  const unsigned blocks = (unsigned)((total + TD_THREADS - 1) / TD_THREADS);
  if (fwd)
    hipLaunchKernelGGL((join_up2_kernel<T, true>), dim3(blocks), dim3(TD_THREADS), 0, st, (T*)a, (T*)b, (T*)t, N, H, W, C0, C1, C2, (T*)out);
  else
    hipLaunchKernelGGL((join_up2_kernel<T, false>), dim3(blocks), dim3(TD_THREADS), 0, st, (T*)a, (T*)b, (T*)t, N, H, W, C0, C1, C2, (T*)out);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

template <typename T>
static int run_join(bool fwd, void* a, void* b, void* t, long long npix, int C0, int C1, int C2, void* out, hipStream_t st) {
  const long long total = npix * (C0 / 8 + C1 / 8 + 1);
  const unsigned blocks = (unsigned)((total + TD_THREADS - 1) / TD_THREADS);
  if (fwd)
    hipLaunchKernelGGL((join_kernel<T, true>), dim3(blocks), dim3(TD_THREADS), 0, st, (T*)a, (T*)b, (T*)t, npix, C0, C1, C2, (T*)out);
  else
    hipLaunchKernelGGL((join_kernel<T, false>), dim3(blocks), dim3(TD_THREADS), 0, st, (T*)a, (T*)b, (T*)t, npix, C0, C1, C2, (T*)out);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

}  // namespace td

static int join_check(const void* a, const void* b, const void* t, const void* out, long long npix, int C0, int C1, int C2) {
  if (!a || !b || !t || !out || npix <= 0 || C0 <= 0 || C1 <= 0 || C2 <= 0) return TD_ERR_BAD_ARG;
  if (C0 % 8 || C1 % 8 || C2 > 8 || npix * (long long)(C0 + C1 + 8) >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  return TD_OK;
}

extern "C" int td_join_fwd(const void* a, const void* b, const void* tail, int dtype, long long npix, int C0, int C1, int C2,
                           void* out, td_stream_t stream) {
  const int rc = join_check(a, b, tail, out, npix, C0, C1, C2);
  if (rc != TD_OK) return rc;
  if (dtype == TD_DTYPE_BF16)
    return td::run_join<__hip_bfloat16>(true, (void*)a, (void*)b, (void*)tail, npix, C0, C1, C2, out, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32) return td::run_join<float>(true, (void*)a, (void*)b, (void*)tail, npix, C0, C1, C2, out, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_join_bwd(const void* grad_out, int dtype, long long npix, int C0, int C1, int C2, void* grad_a, void* grad_b,
                           void* grad_tail, td_stream_t stream) {
  const int rc = join_check(grad_a, grad_b, grad_tail, grad_out, npix, C0, C1, C2);
  if (rc != TD_OK) return rc;
  if (dtype == TD_DTYPE_BF16)
    return td::run_join<__hip_bfloat16>(false, grad_a, grad_b, grad_tail, npix, C0, C1, C2, (void*)grad_out, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32) return td::run_join<float>(false, grad_a, grad_b, grad_tail, npix, C0, C1, C2, (void*)grad_out, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

static int join_up2_check(const void* a, const void* b, const void* t, const void* out, int N, int H, int W, int C0, int C1, int C2) {
  if (!a || !b || !t || !out || N <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C1 <= 0 || C2 <= 0) return TD_ERR_BAD_ARG;
  if (C0 % 8 || C1 % 8 || C2 > 8 || (H & 1) || (W & 1) || (long long)N * H * W * (C0 + C1 + 8) >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  return TD_OK;
}

extern "C" int td_join_up2_fwd(const void* a, const void* b_half, const void* tail, int dtype, int N, int H, int W, int C0, int C1, int C2,
                               void* out, td_stream_t stream) {
  const int rc = join_up2_check(a, b_half, tail, out, N, H, W, C0, C1, C2);
  if (rc != TD_OK) return rc;
  if (dtype == TD_DTYPE_BF16)
    return td::run_join_up2<__hip_bfloat16>(true, (void*)a, (void*)b_half, (void*)tail, N, H, W, C0, C1, C2, out, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_join_up2<float>(true, (void*)a, (void*)b_half, (void*)tail, N, H, W, C0, C1, C2, out, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_join_up2_bwd(const void* grad_out, int dtype, int N, int H, int W, int C0, int C1, int C2, void* grad_a,
                               void* grad_b_half, void* grad_tail, td_stream_t stream) {
  const int rc = join_up2_check(grad_a, grad_b_half, grad_tail, grad_out, N, H, W, C0, C1, C2);
  if (rc != TD_OK) return rc;
  if (dtype == TD_DTYPE_BF16)
    return td::run_join_up2<__hip_bfloat16>(false, grad_a, grad_b_half, grad_tail, N, H, W, C0, C1, C2, (void*)grad_out, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_join_up2<float>(false, grad_a, grad_b_half, grad_tail, N, H, W, C0, C1, C2, (void*)grad_out, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}
