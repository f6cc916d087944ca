// Shared pieces of the hand-written MFMA convolution kernels (td_conv1x1.hip, td_conv3x3.hip): the LDS stage layout, the
// per-stage MFMA block and the epilogue (accumulators -> bf16 image in LDS -> whole pixel rows + BatchNorm partial sums).
#pragma once
#include <hip/hip_bf16.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

typedef __attribute__((ext_vector_type(8))) __bf16 cv_bf16x8;
typedef __attribute__((ext_vector_type(16))) float cv_f32x16;

constexpr int CV_BK = 64;            // K elements per stage (128 bytes per row)
constexpr int CV_THREADS = 256;

// byte offset of 16-byte chunk c (0..7) of row r in a [rows][64 bf16] stage
__device__ __forceinline__ int cv_swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

template <int BM, int BN>
struct CvTile {
  static constexpr int WM = BM / 2, WN = BN / 2;          // wave tile: WM pixels x WN channels
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int PITCH = BN * 2 + 8;                // epilogue image row pitch (bytes): 8-byte stores conflict-free
  static constexpr int CX = BN / 8;                       // 8-channel vectors per pixel row of the tile
  static constexpr int RSTEP = CV_THREADS / CX;           // pixel rows covered per pass of the store loop
  static constexpr int RED_BYTES = 2 * RSTEP * (BN + 1) * 4;
  static constexpr int IMG_BYTES = BM * PITCH;
  static constexpr int LDS_BYTES = (2 * STAGE > IMG_BYTES + RED_BYTES) ? 2 * STAGE : IMG_BYTES + RED_BYTES;
};

template <int BM, int BN>
struct CvAcc {                      // the accumulators of one wave: [channel tile][pixel tile] of 32x32 f32
  cv_f32x16 v[CvTile<BM, BN>::TN][CvTile<BM, BN>::TM];
};

// XCD-aware tile order: blocks b and b + 8 share an XCD (round-robin dispatch), so XCD x walks a contiguous run of tiles
__device__ __forceinline__ int cv_xcd_tile(int b, int total_blocks) {
  const int xcd = b & 7, q = total_blocks >> 3, r = total_blocks & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// one 64-deep K stage: D[n][pixel] += W-tile (sb: [BN][64] swizzled) x X-tile (sa: [BM][64] swizzled)
template <int BM, int BN>
__device__ __forceinline__ void cv_stage_mfma(CvAcc<BM, BN>& acc, const unsigned char* sa,
                                              const unsigned char* sb, int wm, int wn, int l31, int h) {
  using T = CvTile<BM, BN>;
#pragma unroll
  for (int kk = 0; kk < CV_BK / 16; ++kk) {
    cv_bf16x8 fw[T::TN], fx[T::TM];
#pragma unroll
    for (int i = 0; i < T::TN; ++i)
      fw[i] = __builtin_bit_cast(cv_bf16x8, *reinterpret_cast<const uint4*>(sb + cv_swz(wn * T::WN + i * 32 + l31, 2 * kk + h)));
#pragma unroll
    for (int j = 0; j < T::TM; ++j)
      fx[j] = __builtin_bit_cast(cv_bf16x8, *reinterpret_cast<const uint4*>(sa + cv_swz(wm * T::WM + j * 32 + l31, 2 * kk + h)));
#pragma unroll
    for (int i = 0; i < T::TN; ++i)
#pragma unroll
      for (int j = 0; j < T::TM; ++j) acc.v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fx[j], acc.v[i][j], 0, 0, 0);
  }
}

// accumulators -> bf16 image [pixel][channel] in LDS (the stages must be free: call behind the K loop's last barrier) -> global in
// whole pixel rows (16 bytes per lane) [+ bias] and, with ws, the BatchNorm partial sums of this tile from the bf16-rounded values
template <int BM, int BN>
__device__ __forceinline__ void cv_epilogue(CvAcc<BM, BN>& acc, unsigned char* lds,
                                            __hip_bfloat16* __restrict__ y, float* __restrict__ ws, long long row0, int rows_valid, int n0,
                                            int N, long long stat_row, int tid, int wm, int wn, int l31, int h) {
  using T = CvTile<BM, BN>;
  unsigned char* img = lds;
#pragma unroll
  for (int i = 0; i < T::TN; ++i)
#pragma unroll
    for (int j = 0; j < T::TM; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // registers 4g..4g+3 of lane (l31, h): channels wn*WN + 32 i + 8 g + 4 h + {0..3} of pixel wm*WM + 32 j + l31
        uint2 p;
        p.x = (unsigned)f2bf(acc.v[i][j][4 * g + 0]) | ((unsigned)f2bf(acc.v[i][j][4 * g + 1]) << 16);
        p.y = (unsigned)f2bf(acc.v[i][j][4 * g + 2]) | ((unsigned)f2bf(acc.v[i][j][4 * g + 3]) << 16);
        const int pix = wm * T::WM + j * 32 + l31, ch = wn * T::WN + i * 32 + 8 * g + 4 * h;
        *reinterpret_cast<uint2*>(img + pix * T::PITCH + ch * 2) = p;
      }
  __syncthreads();
  const int cx = tid % T::CX, ry = tid / T::CX;
  float sa8[8], sq8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sa8[e] = 0.f; sq8[e] = 0.f; }
#pragma unroll 4
  for (int r = ry; r < BM; r += T::RSTEP) {
    const uint2 lo = *reinterpret_cast<const uint2*>(img + r * T::PITCH + cx * 16);
    const uint2 hi = *reinterpret_cast<const uint2*>(img + r * T::PITCH + cx * 16 + 8);
    if (r < rows_valid) {
      *reinterpret_cast<uint4*>(y + (row0 + r) * (long long)N + n0 + cx * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      if (ws) {
        const unsigned wv[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v0 = bf2f((unsigned short)(wv[e] & 0xffff)), v1 = bf2f((unsigned short)(wv[e] >> 16));
          sa8[2 * e] += v0;
          sq8[2 * e] = fmaf(v0, v0, sq8[2 * e]);
          sa8[2 * e + 1] += v1;
          sq8[2 * e + 1] = fmaf(v1, v1, sq8[2 * e + 1]);
        }
      }
    }
  }
  if (ws) {
    float* red = reinterpret_cast<float*>(lds + T::IMG_BYTES);        // [2][RSTEP][BN + 1], behind the image
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(0 * T::RSTEP + ry) * (BN + 1) + cx * 8 + e] = sa8[e];
      red[(1 * T::RSTEP + ry) * (BN + 1) + cx * 8 + e] = sq8[e];
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid - which * BN;
      float t = 0.f;
#pragma unroll 8
      for (int j = 0; j < T::RSTEP; ++j) t += red[(which * T::RSTEP + j) * (BN + 1) + c];
      ws[(stat_row * N + n0 + c) * 2 + which] = t;
    }
  }
}

struct ConvTile { int bm, bn; };

static inline ConvTile cv_pick_tile(long long Mg, int G, int N) {
  auto blocks = [&](int bm, int bn) { return (long long)G * ((Mg + bm - 1) / bm) * (N / bn); };
  if (N % 128 == 0 && blocks(128, 128) >= 448) return {128, 128};
  if (blocks(128, 64) >= 448) return {128, 64};
  return {64, 64};
}

}  // namespace td
