// 5x5 / stride 1 / pad 2 max-pooling for channels-last (NHWC) activations -- the 16 pools of the
// CRP blocks (reference: mono/model/mono_fm_joint/layers.py:208, nn.MaxPool2d(5, 1, 2)).
// Forward keeps a 1-byte window offset per element (ATen keeps an int64 index); backward is a
// gather over the 25 outputs whose window contains the input element -- no atomics.
// Tie-break and NaN handling follow ATen's max_pool2d: row-major scan, strict '>', NaN wins.
#include <hip/hip_bf16.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

// one thread = one output pixel x 8 consecutive channels
template <typename T>
__global__ __launch_bounds__(TD_THREADS) void maxpool5_fwd_kernel(const T* __restrict__ in, int N, int H, int W, int C,
                                                                  T* __restrict__ out, uint8_t* __restrict__ idx) {
  const int c8 = C >> 3;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  const long long total = (long long)N * H * W * c8;
  if (gid >= total) return;
  const int cv = (int)(gid % c8);
  const long long pix = gid / c8;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
  float best[8];
  unsigned char arg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; arg[i] = 12; }
  const T* base = in + (size_t)n * H * W * C + (size_t)cv * 8;
#pragma unroll
  for (int dy = 0; dy < 5; ++dy) {
    const int yy = y + dy - 2;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const int xx = x + dx - 2;
      if (xx < 0 || xx >= W) continue;
      float v[8];
      load8(base + ((size_t)yy * W + xx) * C, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (v[i] > best[i] || v[i] != v[i]) { best[i] = v[i]; arg[i] = (unsigned char)(dy * 5 + dx); }
      }
    }
  }
  const size_t o = ((size_t)pix) * C + (size_t)cv * 8;
  store8(out + o, best);
  uint2 packed;
  packed.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | ((unsigned)arg[3] << 24);
  packed.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | ((unsigned)arg[7] << 24);
  *reinterpret_cast<uint2*>(idx + o) = packed;
}

// LDS-tiled forward.  The one-output-per-thread form above reads every input vector 25 times through L1/L2 (1.2 GB of
// cache traffic for the 47 MB map of the last decoder stage: 200 us, 0.6 TB/s of useful bytes); here a block stages an
// (8+4) x (16+4) pixel tile of a 64-channel slab in LDS once (1.9x the tile's bytes from global memory, 30 KB) and the 25
// window reads come from LDS with conflict-free 16-byte accesses (consecutive threads = consecutive channel vectors).
// Same scan order and tie-break as above; positions outside the image hold -inf and can never be selected.
constexpr int MP_TH = 8, MP_TW = 16, MP_CV = 8;      // tile rows, tile columns, 8-channel vectors per slab (64 channels)
template <typename T>
__global__ __launch_bounds__(TD_THREADS) void maxpool5_fwd_lds_kernel(const T* __restrict__ in, int N, int H, int W, int C,
                                                                      T* __restrict__ out, uint8_t* __restrict__ idx) {
  struct alignas(16) Vec { T v[8]; };
  __shared__ Vec tile[MP_TH + 4][MP_TW + 4][MP_CV];
  const int slabs = C / (8 * MP_CV);
  const int n = blockIdx.z / slabs, slab = blockIdx.z % slabs;
  const int ty0 = blockIdx.y * MP_TH, tx0 = blockIdx.x * MP_TW;
  const T* base = in + (size_t)n * H * W * C + (size_t)slab * (8 * MP_CV);
  // stage the tile + 2-pixel halo: consecutive threads take consecutive channel vectors of a pixel (128 contiguous bytes)
  for (int e = threadIdx.x; e < (MP_TH + 4) * (MP_TW + 4) * MP_CV; e += TD_THREADS) {
    const int cv = e % MP_CV, p = e / MP_CV;
    const int px = p % (MP_TW + 4), py = p / (MP_TW + 4);
    const int yy = ty0 + py - 2, xx = tx0 + px - 2;
    Vec val;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      val = *reinterpret_cast<const Vec*>(base + ((size_t)yy * W + xx) * C + (size_t)cv * 8);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) val.v[i] = T(-INFINITY);
    }
    tile[py][px][cv] = val;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < MP_TH * MP_TW * MP_CV; o += TD_THREADS) {
    const int cv = o % MP_CV, p = o / MP_CV;
    const int lx = p % MP_TW, ly = p / MP_TW;
    const int y = ty0 + ly, x = tx0 + lx;
    if (y >= H || x >= W) continue;
    float best[8];
    unsigned char arg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; arg[i] = 12; }
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        float v[8];
        load8(tile[ly + dy][lx + dx][cv].v, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (v[i] > best[i] || v[i] != v[i]) { best[i] = v[i]; arg[i] = (unsigned char)(dy * 5 + dx); }
        }
      }
    }
    const size_t off = (((size_t)n * H + y) * W + x) * C + (size_t)slab * (8 * MP_CV) + (size_t)cv * 8;
    store8(out + off, best);
    uint2 packed;
    packed.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | ((unsigned)arg[3] << 24);
    packed.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | ((unsigned)arg[7] << 24);
    *reinterpret_cast<uint2*>(idx + off) = packed;
  }
}

// one thread = one input pixel x 8 channels: sum the gradients of the outputs that selected it
template <typename T>
__global__ __launch_bounds__(TD_THREADS) void maxpool5_bwd_kernel(const T* __restrict__ gout,
                                                                  const uint8_t* __restrict__ idx, int N, int H,
                                                                  int W, int C, T* __restrict__ gin) {
  const int c8 = C >> 3;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  const long long total = (long long)N * H * W * c8;
  if (gid >= total) return;
  const int cv = (int)(gid % c8);
  const long long pix = gid / c8;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const size_t nb = (size_t)n * H * W * C + (size_t)cv * 8;
#pragma unroll
  for (int dy = 0; dy < 5; ++dy) {
    const int oy = y - (dy - 2);                       // output whose window offset dy lands on y
    if (oy < 0 || oy >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const int ox = x - (dx - 2);
      if (ox < 0 || ox >= W) continue;
      const size_t o = nb + ((size_t)oy * W + ox) * C;
      const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
      const unsigned want = (unsigned)(dy * 5 + dx);
      const unsigned w0 = pk.x, w1 = pk.y;
      const bool hit0 = ((w0 & 0xff) == want) | (((w0 >> 8) & 0xff) == want) | (((w0 >> 16) & 0xff) == want) | ((w0 >> 24) == want);
      const bool hit1 = ((w1 & 0xff) == want) | (((w1 >> 8) & 0xff) == want) | (((w1 >> 16) & 0xff) == want) | ((w1 >> 24) == want);
      if (!(hit0 | hit1)) continue;
      float g[8];
      load8(gout + o, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (((w0 >> (8 * i)) & 0xff) == want) acc[i] += g[i];
        if (((w1 >> (8 * i)) & 0xff) == want) acc[4 + i] += g[4 + i];
      }
    }
  }
  store8(gin + (size_t)pix * C + (size_t)cv * 8, acc);
}

// ---------------------------------------------------------------------------------------------
// ResNet stem pool: MaxPool2d(kernel 3, stride 2, padding 1) (reference: mono/model/mono_fm_joint/resnet.py:101),
// same conventions: one thread = one output pixel x 8 channels, 1-byte window offset dy*3+dx, gather-form backward
// (each input pixel lies in at most 2 x 2 windows).  ATen: 49 us forward / 105 us backward per call at C2.
template <typename T>
__global__ __launch_bounds__(TD_THREADS) void maxpool3s2_fwd_kernel(const T* __restrict__ in, int N, int H, int W, int C, int Ho, int Wo,
                                                                    T* __restrict__ out, uint8_t* __restrict__ idx) {
  const int c8 = C >> 3;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  if (gid >= (long long)N * Ho * Wo * c8) return;
  const int cv = (int)(gid % c8);
  const long long pix = gid / c8;
  const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), n = (int)(pix / ((long long)Wo * Ho));
  float best[8];
  unsigned char arg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; arg[i] = 4; }
  const T* base = in + (size_t)n * H * W * C + (size_t)cv * 8;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = 2 * yo - 1 + dy;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int xx = 2 * xo - 1 + dx;
      if (xx < 0 || xx >= W) continue;
      float v[8];
      load8(base + ((size_t)yy * W + xx) * C, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (v[i] > best[i] || v[i] != v[i]) { best[i] = v[i]; arg[i] = (unsigned char)(dy * 3 + dx); }
      }
    }
  }
  const size_t o = ((size_t)pix) * C + (size_t)cv * 8;
  store8(out + o, best);
  uint2 packed;
  packed.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | ((unsigned)arg[3] << 24);
  packed.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | ((unsigned)arg[7] << 24);
  *reinterpret_cast<uint2*>(idx + o) = packed;
}

template <typename T>
__global__ __launch_bounds__(TD_THREADS) void maxpool3s2_bwd_kernel(const T* __restrict__ gout, const uint8_t* __restrict__ idx, int N,
                                                                    int H, int W, int C, int Ho, int Wo, T* __restrict__ gin) {
  const int c8 = C >> 3;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  if (gid >= (long long)N * H * W * c8) return;
  const int cv = (int)(gid % c8);
  const long long pix = gid / c8;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const size_t nb = (size_t)n * Ho * Wo * C + (size_t)cv * 8;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int t = y + 1 - dy;                     // 2 * oy = y + 1 - dy
    if (t < 0 || (t & 1)) continue;
    const int oy = t >> 1;
    if (oy >= Ho) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int u = x + 1 - dx;
      if (u < 0 || (u & 1)) continue;
      const int ox = u >> 1;
      if (ox >= Wo) continue;
      const size_t o = nb + ((size_t)oy * Wo + ox) * C;
      const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
      const unsigned want = (unsigned)(dy * 3 + dx);
      float g[8];
      load8(gout + o, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (((pk.x >> (8 * i)) & 0xff) == want) acc[i] += g[i];
        if (((pk.y >> (8 * i)) & 0xff) == want) acc[4 + i] += g[4 + i];
      }
    }
  }
  store8(gin + (size_t)pix * C + (size_t)cv * 8, acc);
}

template <typename T>
static int run_maxpool3s2(bool fwd, const void* a, const void* aux, int N, int H, int W, int C, void* o, void* o2, hipStream_t st) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = fwd ? (long long)N * Ho * Wo * (C / 8) : (long long)N * H * W * (C / 8);
  const unsigned blocks = (unsigned)((total + TD_THREADS - 1) / TD_THREADS);
  if (fwd)
    hipLaunchKernelGGL((maxpool3s2_fwd_kernel<T>), dim3(blocks), dim3(TD_THREADS), 0, st, (const T*)a, N, H, W, C, Ho, Wo, (T*)o, (uint8_t*)o2);
  else
    hipLaunchKernelGGL((maxpool3s2_bwd_kernel<T>), dim3(blocks), dim3(TD_THREADS), 0, st, (const T*)a, (const uint8_t*)aux, N, H, W, C, Ho, Wo, (T*)o);
  return record_launch_error(hipGetLastError(), fwd ? "td_maxpool3s2_fwd" : "td_maxpool3s2_bwd");
}

template <typename T>
static int run_maxpool(bool fwd, const void* a, const void* aux, int N, int H, int W, int C, void* o, void* o2, hipStream_t st) {
  const long long total = (long long)N * H * W * (C / 8);
  const unsigned blocks = (unsigned)((total + TD_THREADS - 1) / TD_THREADS);
  // LDS tiles pay on the large maps (163 vs 202 us at 12x256x48x160, 47 vs 54 us at 24x80); on the small ones the few
  // tiles leave most of the chip idle (31 vs 17 us at 12x40) and the one-output-per-thread form stays
  if (fwd && C % (8 * MP_CV) == 0 && (long long)H * W >= 1536 && (long long)N * (C / (8 * MP_CV)) <= 65535) {
    const dim3 grid((W + MP_TW - 1) / MP_TW, (H + MP_TH - 1) / MP_TH, N * (C / (8 * MP_CV)));
    hipLaunchKernelGGL((maxpool5_fwd_lds_kernel<T>), grid, dim3(TD_THREADS), 0, st, (const T*)a, N, H, W, C, (T*)o, (uint8_t*)o2);
  } else if (fwd)
    hipLaunchKernelGGL((maxpool5_fwd_kernel<T>), dim3(blocks), dim3(TD_THREADS), 0, st, (const T*)a, N, H, W, C, (T*)o, (uint8_t*)o2);
  else
    hipLaunchKernelGGL((maxpool5_bwd_kernel<T>), dim3(blocks), dim3(TD_THREADS), 0, st, (const T*)a, (const uint8_t*)aux, N, H, W, C, (T*)o);
  return record_launch_error(hipGetLastError(), fwd ? "td_maxpool5_fwd" : "td_maxpool5_bwd");
}

}  // namespace td

extern "C" int td_maxpool5_fwd(const void* in, int dtype, int N, int H, int W, int C, void* out, uint8_t* idx,
                               td_stream_t stream) {
  if (!in || !out || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return TD_ERR_BAD_ARG;
  if (C % 8 != 0) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16) return td::run_maxpool<__hip_bfloat16>(true, in, nullptr, N, H, W, C, out, idx, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32) return td::run_maxpool<float>(true, in, nullptr, N, H, W, C, out, idx, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_maxpool5_bwd(const void* grad_out, const uint8_t* idx, int dtype, int N, int H, int W, int C,
                               void* grad_in, td_stream_t stream) {
  if (!grad_out || !idx || !grad_in || N <= 0 || H <= 0 || W <= 0 || C <= 0) return TD_ERR_BAD_ARG;
  if (C % 8 != 0) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16) return td::run_maxpool<__hip_bfloat16>(false, grad_out, idx, N, H, W, C, grad_in, nullptr, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32) return td::run_maxpool<float>(false, grad_out, idx, N, H, W, C, grad_in, nullptr, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_maxpool3s2_fwd(const void* in, int dtype, int N, int H, int W, int C, void* out, uint8_t* idx,
                                 td_stream_t stream) {
  if (!in || !out || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return TD_ERR_BAD_ARG;
  if (C % 8 != 0) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16) return td::run_maxpool3s2<__hip_bfloat16>(true, in, nullptr, N, H, W, C, out, idx, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32) return td::run_maxpool3s2<float>(true, in, nullptr, N, H, W, C, out, idx, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_maxpool3s2_bwd(const void* grad_out, const uint8_t* idx, int dtype, int N, int H, int W, int C,
                                 void* grad_in, td_stream_t stream) {
  if (!grad_out || !idx || !grad_in || N <= 0 || H <= 0 || W <= 0 || C <= 0) return TD_ERR_BAD_ARG;
  if (C % 8 != 0) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16) return td::run_maxpool3s2<__hip_bfloat16>(false, grad_out, idx, N, H, W, C, grad_in, nullptr, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32) return td::run_maxpool3s2<float>(false, grad_out, idx, N, H, W, C, grad_in, nullptr, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}
