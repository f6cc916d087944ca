// Input expansion on the device: uint8 frames -> ("color", f, 0) float images and their colour-jittered
// ("color_aug", f, 0) copies.  Replaces, on the host side of the reference, ToTensor + ColorJitter per frame in
// MonoDataset.preprocess (mono/datasets/mono_dataset.py:83-101, jitter parameters drawn at :146-152) and the float32
// upload of BOTH copies in change_input_variable (mono/apis/trainer.py:19-29): the wire carries 3 bytes per pixel
// and frame instead of 24.
//
// The jitter is the torchvision ColorJitter sequence (brightness, contrast, saturation, hue in a per-sample random
// order) evaluated in float32 on [0,1] values, i.e. torchvision's tensor formulas:
//   brightness  clamp(f * x)                     contrast    clamp(f * x + (1 - f) * mean(gray(x)))   (mean over the image)
//   saturation  clamp(f * x + (1 - f) * gray(x)) hue         RGB -> HSV, h = frac(h + f), HSV -> RGB
//   gray = 0.299 r + 0.587 g + 0.114 b
// (the reference applies the same operations to 8-bit PIL images, which quantise after every stage: parity with it is
// to within that quantisation, oracle/augment.py and tests/test_augment_cpu.py).
// aug[n, 0..8] = (enabled, op0, op1, op2, op3, brightness, contrast, saturation, hue); op codes 0..3 in that order.
#include "td_common.h"

namespace td {

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return (0.299f * r + 0.587f * g) + 0.114f * b; }

__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float f) {
  // torchvision _rgb2hsv / _hsv2rgb
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float ones = 1.f;
  const float s = cr / (eqc ? ones : maxc);
  const float crd = eqc ? ones : cr;
  const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
  const float hr = (maxc == r) ? (bc - gc) : 0.f;
  const float hg = ((maxc == g) && (maxc != r)) ? (2.f + rc - bc) : 0.f;
  const float hb = ((maxc != g) && (maxc != r)) ? (4.f + gc - rc) : 0.f;
  float h = (hr + hg + hb) / 6.f + 1.f;
  h = h - floorf(h);                       // fmod(., 1) of a non-negative value
  h = h + f;
  h = h - floorf(h);                       // (h + f) % 1.0 with python semantics
  const float v = maxc;
  const float i6 = floorf(h * 6.f);
  const float fr = h * 6.f - i6;
  const int i = ((int)i6) % 6;
  const float p = clamp01(v * (1.f - s)), q = clamp01(v * (1.f - s * fr)), t = clamp01(v * (1.f - s * (1.f - fr)));
  switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// applies ops [first, last) of the sample's order; `mean` is used by the contrast stage
__device__ __forceinline__ void apply_ops(const float* __restrict__ a, int first, int last, float mean, float& r, float& g, float& b) {
  for (int k = first; k < last; ++k) {
    const int op = (int)a[1 + k];
    if (op == 0) {
      const float f = a[5];
      r = clamp01(f * r); g = clamp01(f * g); b = clamp01(f * b);
    } else if (op == 1) {
      const float f = a[6], m = (1.f - f) * mean;
      r = clamp01(f * r + m); g = clamp01(f * g + m); b = clamp01(f * b + m);
    } else if (op == 2) {
      const float f = a[7], m = (1.f - f) * gray_of(r, g, b);
      r = clamp01(f * r + m); g = clamp01(f * g + m); b = clamp01(f * b + m);
    } else {
      hue_shift(r, g, b, a[8]);
    }
  }
}

__device__ __forceinline__ int contrast_pos(const float* a) {
  for (int k = 0; k < 4; ++k) if ((int)a[1 + k] == 1) return k;
  return 4;
}

// one block per image: mean of the grayscale image as the contrast stage sees it (after the stages before it)
__global__ __launch_bounds__(1024) void jitter_mean_kernel(const unsigned char* __restrict__ u8, const float* __restrict__ aug,
                                                           int plane, float* __restrict__ means) {
  __shared__ float red[16];
  const int n = blockIdx.x;
  const float* a = aug + (size_t)n * 9;
  const unsigned char* p = u8 + (size_t)n * 3 * plane;
  float acc = 0.f;
  if (a[0] != 0.f) {
    const int cpos = contrast_pos(a);
    for (int i = threadIdx.x; i < plane; i += 1024) {
      float r = p[i] / 255.f, g = p[plane + i] / 255.f, b = p[2 * plane + i] / 255.f;
      apply_ops(a, 0, cpos, 0.f, r, g, b);
      acc += gray_of(r, g, b);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[i];
    means[n] = t / (float)plane;
  }
}

__global__ __launch_bounds__(TD_THREADS) void jitter_apply_kernel(const unsigned char* __restrict__ u8, const float* __restrict__ aug,
                                                                  const float* __restrict__ means, int N, int plane,
                                                                  float* __restrict__ color, float* __restrict__ color_aug) {
  const long long total = (long long)N * plane;
  for (long long id = (long long)blockIdx.x * TD_THREADS + threadIdx.x; id < total; id += (long long)gridDim.x * TD_THREADS) {
    const int n = (int)(id / plane);
    const int i = (int)(id - (long long)n * plane);
    const unsigned char* p = u8 + (size_t)n * 3 * plane + i;
    float r = p[0] / 255.f, g = p[plane] / 255.f, b = p[2 * plane] / 255.f;   // ToTensor: exact division
    float* c = color + (size_t)n * 3 * plane + i;
    c[0] = r; c[plane] = g; c[2 * plane] = b;
    const float* a = aug + (size_t)n * 9;
    if (a[0] != 0.f) apply_ops(a, 0, 4, means[n], r, g, b);
    float* o = color_aug + (size_t)n * 3 * plane + i;
    o[0] = r; o[plane] = g; o[2 * plane] = b;
  }
}

}  // namespace td

extern "C" int td_color_jitter(const uint8_t* frames_u8, const float* aug, int N, int H, int W, float* means_scratch, float* color,
                               float* color_aug, td_stream_t stream) {
  if (!frames_u8 || !aug || !means_scratch || !color || !color_aug || N <= 0 || H <= 0 || W <= 0) return TD_ERR_BAD_ARG;
  if ((long long)H * W >= (1ll << 30)) return TD_ERR_UNSUPPORTED;
  const int plane = H * W;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(td::jitter_mean_kernel, dim3(N), dim3(1024), 0, st, frames_u8, aug, plane, means_scratch);
  long long blocks = ((long long)N * plane + TD_THREADS - 1) / TD_THREADS;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(td::jitter_apply_kernel, dim3((unsigned)blocks), dim3(TD_THREADS), 0, st, frames_u8, aug, (const float*)means_scratch,
                     N, plane, color, color_aug);
  return td::record_launch_error(hipGetLastError(), "td_color_jitter");
}
