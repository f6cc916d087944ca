// 8-element (16-byte bf16 / 32-byte f32) channel-vector loads and stores for channels-last kernels.
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

namespace td {

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {      // round-to-nearest-even, NaN kept
  return __builtin_bit_cast(unsigned short, __float2bfloat16(f));
}

__device__ __forceinline__ void load8(const __hip_bfloat16* p, float* v) {
  const uint4 r = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = bf2f((unsigned short)(w[i] & 0xffff)); v[2 * i + 1] = bf2f((unsigned short)(w[i] >> 16)); }
}
__device__ __forceinline__ void load8(const float* p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(__hip_bfloat16* p, const float* v) {
  uint4 r;
  r.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
  r.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
  r.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
  r.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = r;
}
__device__ __forceinline__ void store8(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// 8 channels as they lie in memory: lets a kernel request rows well before it converts them (the conversion of load8 would
// otherwise sit right behind the load and wait for it)
template <typename T>
struct Raw8;
template <>
struct Raw8<__hip_bfloat16> {
  unsigned w[4];
  __device__ __forceinline__ void load(const __hip_bfloat16* p) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    w[0] = r.x; w[1] = r.y; w[2] = r.z; w[3] = r.w;
  }
  __device__ __forceinline__ void neg_inf_unless(bool ok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = ok ? w[i] : 0xff80ff80u;
  }
  __device__ __forceinline__ void unpack(float* v) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
};
template <>
struct Raw8<float> {
  float w[8];
  __device__ __forceinline__ void load(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  }
  __device__ __forceinline__ void neg_inf_unless(bool ok) {
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = ok ? w[i] : -INFINITY;
  }
  __device__ __forceinline__ void unpack(float* v) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = w[i];
  }
};

}  // namespace td
