// 3x3 convolutions (stride 1 or 2, padding 0 or 1) on channels-last bf16 activations as an implicit bf16 MFMA GEMM with the
// following BatchNorm's batch statistics in the epilogue (reference: conv2 -> bn2 of Bottleneck.forward and conv1/conv2 of
// BasicBlock.forward, mono/model/mono_fm_joint/resnet.py:30-49, 66-86; the padded Conv3x3 of the decoders, layers.py:176-190).
//
//   Y[m, n] = sum over taps (r, s) and channels c of X[pixel(m) + (r, s), c] * W[n, r, s, c]
//
// The GEMM is the one of td_conv1x1.hip (same tiles, LDS stages, MFMA block and epilogue: td_conv_tile.h) with K = 9 * Cin
// walked tap by tap: a K stage is 64 channels of ONE tap, so an A-tile row is still 128 contiguous bytes of one input pixel --
// the im2col matrix is never formed, a stage only shifts the pixel each row reads from and zero-fills what falls into the padding.
// W is read as it lies in memory for a channels-last [N, Cin, 3, 3] weight ([N][r][s][c]): K-contiguous like X, no re-layout.
// Cin needs to be a multiple of 8 (16-byte loads); a last partial 64-channel chunk is zero-filled in both operands.
#include "td_conv_tile.h"

namespace td {

struct Conv3Geom {
  int Hi, Wi, Ho, Wo, Cin, stride, pad;
};

template <int BM, int BN>
__global__ __launch_bounds__(CV_THREADS, 2) void conv3x3_mfma_kernel(
    const __hip_bfloat16* __restrict__ x, const __hip_bfloat16* __restrict__ w, __hip_bfloat16* __restrict__ y,
    float* __restrict__ ws, long long Mg, int N, int tiles_per_group, int total_blocks, Conv3Geom g) {
  using T = CvTile<BM, BN>;
  constexpr int A_BYTES = T::A_BYTES, STAGE = T::STAGE, TN = T::TN, TM = T::TM;
  __shared__ __attribute__((aligned(16))) unsigned char lds[T::LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, h = lane >> 5;
  const int L = cv_xcd_tile(blockIdx.x, total_blocks);
  const int NT = N / BN;
  const int nt = L % NT, mt = L / NT;
  const int grp = mt / tiles_per_group, s = mt - grp * tiles_per_group;
  const long long row0 = (long long)grp * Mg + (long long)s * BM;
  const int rows_valid = (int)((Mg - (long long)s * BM) < BM ? (Mg - (long long)s * BM) : BM);
  const int n0 = nt * BN;

  // ---- staging: thread t moves 16-byte chunk (t & 7) of rows (t >> 3) + 32 i; per row the top-left input pixel of its window
  const int lc = tid & 7, lr = tid >> 3;
  constexpr int NA = BM / 32, NB = BN / 32;
  const int HoWo = g.Ho * g.Wo;
  int pb[4], ph[4], pw[4];                          // image base (in pixels), window origin row / column (may be -1: padding)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = lr + 32 * (i < NA ? i : 0);
    r = r < rows_valid ? r : rows_valid - 1;        // rows past the group's end: a valid row, never stored or summed
    const long long m = row0 + r;
    const int b = (int)(m / HoWo);
    const int rem = (int)(m - (long long)b * HoWo);
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    pb[i] = b * g.Hi * g.Wi;
    ph[i] = ho * g.stride - g.pad;
    pw[i] = wo * g.stride - g.pad;
  }
  const int KC = (g.Cin + CV_BK - 1) / CV_BK;       // 64-channel chunks per tap
  const long long wrow = 9ll * g.Cin;               // elements per output channel of W
  const __hip_bfloat16* wb = w + (long long)(n0 + lr) * wrow + lc * 8;
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  ra2 = ra3 = rb2 = rb3 = make_uint4(0, 0, 0, 0);
#define C3_LDA(i, dst)                                                                                            \
  {                                                                                                               \
    const int hi_ = ph[i] + tr_, wi_ = pw[i] + ts_;                                                               \
    const bool ok_ = cok_ && (unsigned)hi_ < (unsigned)g.Hi && (unsigned)wi_ < (unsigned)g.Wi;                    \
    dst = ok_ ? *reinterpret_cast<const uint4*>(x + (long long)(pb[i] + hi_ * g.Wi + wi_) * g.Cin + coff_) : z_;  \
  }
#define C3_LDB(j, dst) dst = cok_ ? *reinterpret_cast<const uint4*>(wb + (long long)(32 * (j)) * wrow + woff_) : z_;
#define C3_LOAD_GLOBAL(kt)                                                       \
  {                                                                              \
    const int tap_ = (kt) / KC, ch_ = (kt) - tap_ * KC;                          \
    const int tr_ = tap_ / 3, ts_ = tap_ - 3 * tr_;                              \
    const int coff_ = ch_ * CV_BK + lc * 8;                                      \
    const bool cok_ = coff_ < g.Cin;                                             \
    const long long woff_ = (long long)tap_ * g.Cin + ch_ * CV_BK;               \
    const uint4 z_ = make_uint4(0, 0, 0, 0);                                     \
    C3_LDA(0, ra0) C3_LDA(1, ra1)                                                \
    if (NA > 2) { C3_LDA(2, ra2) C3_LDA(3, ra3) }                                \
    C3_LDB(0, rb0) C3_LDB(1, rb1)                                                \
    if (NB > 2) { C3_LDB(2, rb2) C3_LDB(3, rb3) }                                \
  }
#define C3_ST(base, i, v) (*reinterpret_cast<uint4*>((base) + cv_swz(lr + 32 * (i), lc)) = (v))
#define C3_WRITE_LDS(stage)                                                       \
  {                                                                               \
    unsigned char* base_ = lds + (stage) * STAGE;                                 \
    C3_ST(base_, 0, ra0);                                                         \
    C3_ST(base_, 1, ra1);                                                         \
    if (NA > 2) { C3_ST(base_, 2, ra2); C3_ST(base_, 3, ra3); }                   \
    C3_ST(base_ + A_BYTES, 0, rb0);                                               \
    C3_ST(base_ + A_BYTES, 1, rb1);                                               \
    if (NB > 2) { C3_ST(base_ + A_BYTES, 2, rb2); C3_ST(base_ + A_BYTES, 3, rb3); } \
  }

  CvAcc<BM, BN> acc;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc.v[i][j][e] = 0.f;

  const int nk = 9 * KC;
  C3_LOAD_GLOBAL(0)
  C3_WRITE_LDS(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) C3_LOAD_GLOBAL(kt + 1)
    const unsigned char* sa = lds + (kt & 1) * STAGE;
    cv_stage_mfma<BM, BN>(acc, sa, sa + A_BYTES, wm, wn, l31, h);
    if (more) C3_WRITE_LDS((kt + 1) & 1)
    __syncthreads();
  }
#undef C3_LDA
#undef C3_LDB
#undef C3_LOAD_GLOBAL
#undef C3_ST
#undef C3_WRITE_LDS
  cv_epilogue<BM, BN>(acc, lds, y, ws, row0, rows_valid, n0, N, (long long)grp * tiles_per_group + s, tid, wm, wn, l31, h);
}

template <int BM, int BN>
static int c3_launch(const void* x, const void* w, void* y, float* ws, long long Mg, int G, int N, Conv3Geom g, hipStream_t st) {
  const int tpg = (int)((Mg + BM - 1) / BM);
  const long long nblk = (long long)G * tpg * (N / BN);
  if (nblk > 0x7fffffffll) return TD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((conv3x3_mfma_kernel<BM, BN>), dim3((unsigned)nblk), dim3(CV_THREADS), 0, st, (const __hip_bfloat16*)x,
                     (const __hip_bfloat16*)w, (__hip_bfloat16*)y, ws, Mg, N, tpg, (int)nblk, g);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

}  // namespace td

extern "C" int td_conv3x3_fwd(const void* x, const void* w, int B, int groups, int Hi, int Wi, int Cin, int N, int stride, int pad, void* y,
                              float* stat_partials, td_stream_t stream) {
  if (!x || !w || !y || B <= 0 || groups < 1 || groups > 64 || B % groups != 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || N <= 0) return TD_ERR_BAD_ARG;
  if (stride < 1 || stride > 2 || pad < 0 || pad > 1) return TD_ERR_BAD_ARG;
  if (Cin % 8 != 0 || N % 64 != 0) return TD_ERR_UNSUPPORTED;
  const int Ho = (Hi + 2 * pad - 3) / stride + 1, Wo = (Wi + 2 * pad - 3) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return TD_ERR_BAD_ARG;
  if ((long long)B * Hi * Wi >= (1ll << 31) || (long long)B * Ho * Wo * (long long)(9 * Cin > N ? 9 * Cin : N) >= (1ll << 42)) return TD_ERR_UNSUPPORTED;
  const long long M = (long long)B * Ho * Wo, Mg = M / groups;
  const td::Conv3Geom g = {Hi, Wi, Ho, Wo, Cin, stride, pad};
  const td::ConvTile t = td::cv_pick_tile(Mg, groups, N);
  hipStream_t st = (hipStream_t)stream;
  if (t.bm == 128 && t.bn == 128) return td::c3_launch<128, 128>(x, w, y, stat_partials, Mg, groups, N, g, st);
  if (t.bm == 128 && t.bn == 64) return td::c3_launch<128, 64>(x, w, y, stat_partials, Mg, groups, N, g, st);
  return td::c3_launch<64, 64>(x, w, y, stat_partials, Mg, groups, N, g, st);
}
