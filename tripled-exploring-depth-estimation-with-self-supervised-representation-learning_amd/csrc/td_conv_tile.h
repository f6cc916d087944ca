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

// BT: the weight operand is stored [reduction][N] (N contiguous) -- the data gradient dX = dY . W reads the forward's [Cout][Cin]
// weight as it lies.  Its stage tile is kept in LDS in that natural layout ([64 reduction rows][BN columns], pitch BN * 2 + 64
// bytes: pitch mod 256 in {64, 192} makes the four rows a ds_read_b64_tr_b16 half-wave touches hit disjoint banks) and the MFMA
// fragments come back through the hardware transpose.
template <int BM, int BN, bool BT = false>
struct CvTile {
  static constexpr int WM = BM / 2, WN = BN / 2;          // wave tile: WM pixels x WN channels
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int PITCH_T = BN * 2 + 64;             // BT: bytes per reduction row of the weight tile
  static constexpr int A_BYTES = BM * 128, B_BYTES = BT ? 64 * PITCH_T : BN * 128, STAGE = A_BYTES + B_BYTES;
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

// MFMA operand fragment out of a [rows][cols] bf16 tile kept as it lies in memory (pitch bytes per row), for the 32 columns
// col0.. and the reduction rows 16 ks .. 16 ks + 15: lane l -> column col0 + (l & 31), rows 16 ks + 8 (l >> 5) + 0..7, through
// two ds_read_b64_tr_b16 (EXEC must be all ones)
template <int PITCH>
__device__ __forceinline__ cv_bf16x8 cv_frag_tr(const unsigned char* tile, int col0, int ks, int lane) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const int g = lane >> 4, i = lane & 15;
  const int row = 16 * ks + 8 * (g >> 1) + (i >> 2);
  const unsigned char* p = tile + row * PITCH + (col0 + 16 * (g & 1) + 4 * (i & 3)) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * PITCH));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(cv_bf16x8, v);
}

// one 64-deep K stage: D[n][pixel] += W-tile (sb: [BN][64] swizzled, or [64][BN] natural with BT) x X-tile (sa: [BM][64] swizzled)
template <int BM, int BN, bool BT = false>
__device__ __forceinline__ void cv_stage_mfma(CvAcc<BM, BN>& acc, const unsigned char* sa,
                                              const unsigned char* sb, int wm, int wn, int l31, int h) {
  using T = CvTile<BM, BN, BT>;
#pragma unroll
  for (int kk = 0; kk < CV_BK / 16; ++kk) {
    cv_bf16x8 fw[T::TN], fx[T::TM];
#pragma unroll
    for (int i = 0; i < T::TN; ++i) {
      if constexpr (BT)
        fw[i] = cv_frag_tr<T::PITCH_T>(sb, wn * T::WN + i * 32, kk, l31 + 32 * h);
      else
        fw[i] = __builtin_bit_cast(cv_bf16x8, *reinterpret_cast<const uint4*>(sb + cv_swz(wn * T::WN + i * 32 + l31, 2 * kk + h)));
    }
#pragma unroll
    for (int j = 0; j < T::TM; ++j)
      fx[j] = __builtin_bit_cast(cv_bf16x8, *reinterpret_cast<const uint4*>(sa + cv_swz(wm * T::WM + j * 32 + l31, 2 * kk + h)));
#pragma unroll
    for (int i = 0; i < T::TN; ++i)
#pragma unroll
      for (int j = 0; j < T::TM; ++j) acc.v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fx[j], acc.v[i][j], 0, 0, 0);
  }
}

// What the epilogue does besides storing the tile (EPI):
//   0  nothing, or with ws: the FORWARD statistics of the BatchNorm that follows (sum y, sum y^2 of the bf16-rounded outputs)
//   1  adds `res` (same [M, N] layout) to the rounded outputs and rounds again -- exactly the bf16 tensor add autograd would run
//      behind the convolution's data gradient (the residual branch of a ResNet block: dx = dgrad(conv1) + d(shortcut))
//   2  the BACKWARD sums of the BatchNorm whose output this tensor is the gradient of (data gradient of the convolution that
//      consumed relu(bn(z))): ws <- (sum g, sum g (z - mean)) with g = y * [fma(z, gamma invstd, beta - mean gamma invstd) > 0],
//      i.e. td_bn_bwd's statistics pass (relu mask recomputed from z) without reading the gradient back
//   3  two outputs: y as computed and y2 = y + res (rounded again): a convolution whose result also joins a running sum
//      (the CRP block's top_i = conv(pool(top_{i-1})), x_i = x_{i-1} + top_i, mono/model/mono_fm_joint/layers.py:200-215)
struct CvEpi {
  __hip_bfloat16* y2;             // EPI 3: [M, N] second output
  const __hip_bfloat16* res;      // EPI 1 / 3: [M, N]
  const __hip_bfloat16* z;        // EPI 2: [M, N] input of the BatchNorm
  const float* gamma;             // EPI 2: [N]
  const float* beta;              // EPI 2: [N]
  const float* mean;              // EPI 2: [G, N] (this block's group selected by the caller)
  const float* invstd;            // EPI 2: [G, N]
};

// accumulators -> bf16 image [pixel][channel] in LDS (the stages must be free: call behind the K loop's last barrier) -> global in
// whole pixel rows (16 bytes per lane) and, with ws, per-channel partial sums of this tile (EPI 0 / 2 above)
template <int BM, int BN, int EPI = 0>
__device__ __forceinline__ void cv_epilogue(CvAcc<BM, BN>& acc, unsigned char* lds,
                                            __hip_bfloat16* __restrict__ y, float* __restrict__ ws, long long row0, int rows_valid, int n0,
                                            int N, long long stat_row, int tid, int wm, int wn, int l31, int h, const CvEpi& ep = CvEpi{}) {
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
  float esc[8], esh[8], emu[8];
  if constexpr (EPI == 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = n0 + cx * 8 + e;
      emu[e] = ep.mean[c];
      esc[e] = ep.gamma[c] * ep.invstd[c];
      esh[e] = ep.beta[c] - emu[e] * esc[e];
    }
  }
  const bool stats = ws != nullptr;
#pragma unroll 4
  for (int r = ry; r < BM; r += T::RSTEP) {
    const uint2 lo = *reinterpret_cast<const uint2*>(img + r * T::PITCH + cx * 16);
    const uint2 hi = *reinterpret_cast<const uint2*>(img + r * T::PITCH + cx * 16 + 8);
    if (r < rows_valid) {
      unsigned wv[4] = {lo.x, lo.y, hi.x, hi.y};
      const long long off = (row0 + r) * (long long)N + n0 + cx * 8;
      if constexpr (EPI == 3) *reinterpret_cast<uint4*>(y + off) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
      if constexpr (EPI == 1 || EPI == 3) {
        const uint4 rr = *reinterpret_cast<const uint4*>(ep.res + off);
        const unsigned rv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v0 = bf2f((unsigned short)(wv[e] & 0xffff)) + bf2f((unsigned short)(rv[e] & 0xffff));
          const float v1 = bf2f((unsigned short)(wv[e] >> 16)) + bf2f((unsigned short)(rv[e] >> 16));
          wv[e] = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
        }
      }
      *reinterpret_cast<uint4*>((EPI == 3 ? ep.y2 : y) + off) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
      if constexpr (EPI == 0) {
        if (stats) {
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
      if constexpr (EPI == 2) {
        const uint4 zz = *reinterpret_cast<const uint4*>(ep.z + off);
        const unsigned zv[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z0 = bf2f((unsigned short)(zv[e] & 0xffff)), z1 = bf2f((unsigned short)(zv[e] >> 16));
          float g0 = bf2f((unsigned short)(wv[e] & 0xffff)), g1 = bf2f((unsigned short)(wv[e] >> 16));
          g0 = fmaf(z0, esc[2 * e], esh[2 * e]) <= 0.f ? 0.f : g0;
          g1 = fmaf(z1, esc[2 * e + 1], esh[2 * e + 1]) <= 0.f ? 0.f : g1;
          sa8[2 * e] += g0;
          sq8[2 * e] = fmaf(g0, z0 - emu[2 * e], sq8[2 * e]);
          sa8[2 * e + 1] += g1;
          sq8[2 * e + 1] = fmaf(g1, z1 - emu[2 * e + 1], sq8[2 * e + 1]);
        }
      }
    }
  }
  if (stats) {
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
