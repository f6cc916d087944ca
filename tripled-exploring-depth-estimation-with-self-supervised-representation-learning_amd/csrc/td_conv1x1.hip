// 1x1 convolutions of the ResNet bottlenecks as a bf16 MFMA GEMM with the following BatchNorm's batch statistics
// formed in the epilogue (reference: mono/model/mono_fm_joint/resnet.py:51-86 -- conv1x1 -> bn -> relu,
// conv1x1 -> bn -> (+identity) -> relu, and the strided 1x1 down-sample branch :119-127).
//
//   Y[M, N] = X[src(m), K] . W[N, K]^T          bf16 operands, fp32 accumulation on v_mfma_f32_32x32x16_bf16
//   ws[(g * S + s), n, :] = (sum y, sum y^2) over the valid rows of m-tile s of statistics group g,
//                           taken from the bf16-ROUNDED outputs, i.e. exactly what td_bn_fwd's own partial-sum pass
//                           over the stored tensor would see: the pass itself (one full read of Y) disappears.
//
// Channels-last activations make a 1x1 convolution a plain row-major GEMM: X is [rows, K] with K contiguous, W is
// [N, K] with K contiguous, so BOTH MFMA operands are K-contiguous 16-byte fragments.  The operands are taken in the
// order (W, X): D[i = n][j = m], which leaves every lane with FOUR CONSECUTIVE CHANNELS of one output pixel per
// register quad -- the accumulators go to an LDS image [pixel][channel] with 8-byte stores and leave as whole
// 128..256-byte pixel rows.
//
// Tiling (wave64): 256 threads = 2 x 2 waves; block tile BM pixels x BN channels (128x128, 128x64 or 64x64, chosen per
// shape so that the grid fills 256 CUs at two blocks per CU); K is walked in 64-element steps through a double-buffered,
// XOR-swizzled LDS stage (chunk ^= (row >> 1) & 7: conflict-free ds_read_b128 on 128-byte rows); one barrier per
// K-step.  Consecutive block ids are remapped so that the blocks of one XCD walk neighbouring tiles (they share the
// pixel rows between channel tiles and always share W in that XCD's L2).
// Roofline: these GEMMs are HBM-bound (K = 64..2048, N = 64..2048 against M up to 276 480): bytes = 2 (M K + N K + M N).
#include <hip/hip_bf16.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

typedef __attribute__((ext_vector_type(8))) __bf16 cv_bf16x8;
typedef __attribute__((ext_vector_type(16))) float cv_f32x16;

constexpr int CV_BK = 64;            // K elements per stage (128 bytes per row)
constexpr int CV_THREADS = 256;

struct ConvRows {                    // output pixel -> input pixel of a strided 1x1 convolution (stride 1: identity)
  int Wo, HoWo, Wi, HiWi, stride;
};

__device__ __forceinline__ long long cv_src_row(long long m, const ConvRows& g) {
  if (g.stride == 1) return m;
  const long long b = m / g.HoWo;
  const int rem = (int)(m - b * g.HoWo);
  const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
  return b * g.HiWi + (long long)ho * g.stride * g.Wi + (long long)wo * g.stride;
}

// byte offset of 16-byte chunk c (0..7) of row r in a [rows][64 bf16] stage
__device__ __forceinline__ int cv_swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

template <int BM, int BN>
__global__ __launch_bounds__(CV_THREADS, 2) void conv1x1_mfma_kernel(
    const __hip_bfloat16* __restrict__ x, const __hip_bfloat16* __restrict__ w, __hip_bfloat16* __restrict__ y,
    float* __restrict__ ws, long long Mg, int K, int N, int tiles_per_group, int total_blocks, ConvRows geom) {
  constexpr int WM = BM / 2, WN = BN / 2;          // wave tile: WM pixels x WN channels
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int PITCH = BN * 2 + 8;                // epilogue image row pitch (bytes): 8-byte stores conflict-free
  constexpr int CX = BN / 8;                       // 8-channel vectors per pixel row of the tile
  constexpr int RSTEP = CV_THREADS / CX;           // pixel rows covered per pass of the store loop
  constexpr int RED_BYTES = 2 * RSTEP * (BN + 1) * 4;
  constexpr int IMG_BYTES = BM * PITCH;
  constexpr int LDS_BYTES = (2 * STAGE > IMG_BYTES + RED_BYTES) ? 2 * STAGE : IMG_BYTES + RED_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, h = lane >> 5;

  // XCD-aware tile order: blocks b and b + 8 share an XCD (round-robin dispatch), so XCD x walks a contiguous run of tiles
  int L;
  {
    const int b = blockIdx.x, xcd = b & 7, q = total_blocks >> 3, r = total_blocks & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int NT = N / BN;
  const int nt = L % NT, mt = L / NT;
  const int grp = mt / tiles_per_group, s = mt - grp * tiles_per_group;
  const long long row0 = (long long)grp * Mg + (long long)s * BM;
  const int rows_valid = (int)((Mg - (long long)s * BM) < BM ? (Mg - (long long)s * BM) : BM);
  const int n0 = nt * BN;

  // ---- global -> register -> LDS staging: thread t moves chunk (t & 7) of rows (t >> 3) + 32 i.  Named registers, not
  // arrays: hipcc otherwise parks the staging registers in scratch / promotes them to LDS.
  const int lc = tid & 7, lr = tid >> 3;
  constexpr int NA = BM / 32, NB = BN / 32;
  auto a_ptr = [&](int i) {
    int r = lr + 32 * i;
    r = r < rows_valid ? r : rows_valid - 1;       // rows past the group's end: a valid row, never stored or summed
    return x + cv_src_row(row0 + r, geom) * (long long)K + lc * 8;
  };
  const __hip_bfloat16 *pa0 = a_ptr(0), *pa1 = a_ptr(1), *pa2 = NA > 2 ? a_ptr(2) : pa0, *pa3 = NA > 2 ? a_ptr(3) : pa0;
  const __hip_bfloat16* pb0 = w + (long long)(n0 + lr) * K + lc * 8;
  const __hip_bfloat16 *pb1 = pb0 + 32ll * K, *pb2 = pb0 + (NB > 2 ? 64ll : 0ll) * K, *pb3 = pb0 + (NB > 2 ? 96ll : 0ll) * K;
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  ra2 = ra3 = rb2 = rb3 = make_uint4(0, 0, 0, 0);
#define CV_LD(p, kt) (*reinterpret_cast<const uint4*>((p) + (kt) * CV_BK))
#define CV_LOAD_GLOBAL(kt)                                     \
  {                                                            \
    ra0 = CV_LD(pa0, kt);                                      \
    ra1 = CV_LD(pa1, kt);                                      \
    if (NA > 2) { ra2 = CV_LD(pa2, kt); ra3 = CV_LD(pa3, kt); } \
    rb0 = CV_LD(pb0, kt);                                      \
    rb1 = CV_LD(pb1, kt);                                      \
    if (NB > 2) { rb2 = CV_LD(pb2, kt); rb3 = CV_LD(pb3, kt); } \
  }
#define CV_ST(base, i, v) (*reinterpret_cast<uint4*>((base) + cv_swz(lr + 32 * (i), lc)) = (v))
#define CV_WRITE_LDS(stage)                                              \
  {                                                                      \
    unsigned char* base_ = lds + (stage) * STAGE;                        \
    CV_ST(base_, 0, ra0);                                                \
    CV_ST(base_, 1, ra1);                                                \
    if (NA > 2) { CV_ST(base_, 2, ra2); CV_ST(base_, 3, ra3); }          \
    CV_ST(base_ + A_BYTES, 0, rb0);                                      \
    CV_ST(base_ + A_BYTES, 1, rb1);                                      \
    if (NB > 2) { CV_ST(base_ + A_BYTES, 2, rb2); CV_ST(base_ + A_BYTES, 3, rb3); } \
  }

  cv_f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / CV_BK;
  CV_LOAD_GLOBAL(0)
  CV_WRITE_LDS(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) CV_LOAD_GLOBAL(kt + 1)
    const unsigned char* sa = lds + (kt & 1) * STAGE;
    const unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < CV_BK / 16; ++kk) {
      cv_bf16x8 fw[TN], fx[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i)
        fw[i] = __builtin_bit_cast(cv_bf16x8, *reinterpret_cast<const uint4*>(sb + cv_swz(wn * WN + i * 32 + l31, 2 * kk + h)));
#pragma unroll
      for (int j = 0; j < TM; ++j)
        fx[j] = __builtin_bit_cast(cv_bf16x8, *reinterpret_cast<const uint4*>(sa + cv_swz(wm * WM + j * 32 + l31, 2 * kk + h)));
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
    }
    if (more) CV_WRITE_LDS((kt + 1) & 1)
    __syncthreads();
  }

  // ---- epilogue: accumulators -> bf16 image [pixel][channel] in LDS (the stages are free after the last barrier)
  unsigned char* img = lds;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // registers 4g..4g+3 of lane (l31, h): channels wn*WN + 32 i + 8 g + 4 h + {0..3} of pixel wm*WM + 32 j + l31
        uint2 p;
        p.x = (unsigned)f2bf(acc[i][j][4 * g + 0]) | ((unsigned)f2bf(acc[i][j][4 * g + 1]) << 16);
        p.y = (unsigned)f2bf(acc[i][j][4 * g + 2]) | ((unsigned)f2bf(acc[i][j][4 * g + 3]) << 16);
        const int pix = wm * WM + j * 32 + l31, ch = wn * WN + i * 32 + 8 * g + 4 * h;
        *reinterpret_cast<uint2*>(img + pix * PITCH + ch * 2) = p;
      }
  __syncthreads();

  // ---- image -> global in whole pixel rows (16 bytes per lane), and the BatchNorm partial sums of this tile
  const int cx = tid % CX, ry = tid / CX;
  float sa8[8], sq8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sa8[e] = 0.f; sq8[e] = 0.f; }
#pragma unroll 4
  for (int r = ry; r < BM; r += RSTEP) {
    const uint2 lo = *reinterpret_cast<const uint2*>(img + r * PITCH + cx * 16);
    const uint2 hi = *reinterpret_cast<const uint2*>(img + r * PITCH + cx * 16 + 8);
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
    float* red = reinterpret_cast<float*>(lds + IMG_BYTES);        // [2][RSTEP][BN + 1], behind the image
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(0 * RSTEP + ry) * (BN + 1) + cx * 8 + e] = sa8[e];
      red[(1 * RSTEP + ry) * (BN + 1) + cx * 8 + e] = sq8[e];
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid - which * BN;
      float t = 0.f;
#pragma unroll 8
      for (int j = 0; j < RSTEP; ++j) t += red[(which * RSTEP + j) * (BN + 1) + c];
      ws[(((long long)grp * tiles_per_group + s) * N + n0 + c) * 2 + which] = t;
    }
  }
}

#undef CV_LOAD_GLOBAL
#undef CV_WRITE_LDS
#undef CV_LD
#undef CV_ST

struct ConvTile { int bm, bn; };

static inline ConvTile cv_pick_tile(long long Mg, int G, int N) {
  auto blocks = [&](int bm, int bn) { return (long long)G * ((Mg + bm - 1) / bm) * (N / bn); };
  if (N % 128 == 0 && blocks(128, 128) >= 448) return {128, 128};
  if (blocks(128, 64) >= 448) return {128, 64};
  return {64, 64};
}

template <int BM, int BN>
static int cv_launch(const void* x, const void* w, void* y, float* ws, long long Mg, int G, int K, int N, ConvRows geom, hipStream_t st) {
  const int tpg = (int)((Mg + BM - 1) / BM);
  const long long nblk = (long long)G * tpg * (N / BN);
  if (nblk > 0x7fffffffll) return TD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((conv1x1_mfma_kernel<BM, BN>), dim3((unsigned)nblk), dim3(CV_THREADS), 0, st, (const __hip_bfloat16*)x,
                     (const __hip_bfloat16*)w, (__hip_bfloat16*)y, ws, Mg, K, N, tpg, (int)nblk, geom);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

}  // namespace td

static bool cv_shape_ok(long long M, int G, int K, int N) {
  return M > 0 && G >= 1 && G <= 64 && M % G == 0 && K > 0 && N > 0 && K % 64 == 0 && N % 64 == 0;
}

extern "C" int td_conv1x1_stat_rows(long long M, int groups, int N) {
  if (M <= 0 || groups < 1 || M % groups != 0 || N <= 0 || N % 64 != 0) return 0;
  const td::ConvTile t = td::cv_pick_tile(M / groups, groups, N);
  return (int)((M / groups + t.bm - 1) / t.bm);
}

extern "C" int td_conv1x1_fwd(const void* x, const void* w, long long M, int groups, int K, int N, int Hi, int Wi, int stride, void* y,
                              float* stat_partials, td_stream_t stream) {
  if (!x || !w || !y || !cv_shape_ok(M, groups, K, N) || stride < 1) return TD_ERR_BAD_ARG;
  td::ConvRows geom = {0, 0, 0, 0, 1};
  if (stride > 1) {
    if (Hi <= 0 || Wi <= 0) return TD_ERR_BAD_ARG;
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    if (M % ((long long)Ho * Wo) != 0) return TD_ERR_BAD_ARG;
    geom = {Wo, Ho * Wo, Wi, Hi * Wi, stride};
  }
  if (M * (long long)(K > N ? K : N) >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  const long long Mg = M / groups;
  const td::ConvTile t = td::cv_pick_tile(Mg, groups, N);
  hipStream_t st = (hipStream_t)stream;
  if (t.bm == 128 && t.bn == 128) return td::cv_launch<128, 128>(x, w, y, stat_partials, Mg, groups, K, N, geom, st);
  if (t.bm == 128 && t.bn == 64) return td::cv_launch<128, 64>(x, w, y, stat_partials, Mg, groups, K, N, geom, st);
  return td::cv_launch<64, 64>(x, w, y, stat_partials, Mg, groups, K, N, geom, st);
}
