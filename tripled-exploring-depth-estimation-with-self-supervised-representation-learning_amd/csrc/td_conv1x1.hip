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
#include "td_conv_tile.h"

namespace td {

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

// ---- fused forms (round 4) -------------------------------------------------------------------------------------------
// The same GEMM serves the data gradient (BT: the weight is read as it lies, [reduction][N], through transposed LDS reads) and
// takes BatchNorm work of the NEIGHBOURING layers on both ends, so that the separate passes over the activations disappear
// (reference: Bottleneck.forward, mono/model/mono_fm_joint/resnet.py:66-86, and its autograd):
//   PRO 1  A = relu(bn(z)): conv3's input straight from conv2's raw output; the statistics are finished from <= 16 partial rows in
//          the prologue of every block (same ordered sum everywhere), block 0 writes save_mean / save_invstd / running statistics
//   PRO 2  A = the BatchNorm-backward of (g, z): dz = k0 g [fma(z, sc, sh) > 0] + k1 z + k2 -- conv1's data gradient straight
//          from conv2's data gradient; coefficients finished from the backward partial rows in the prologue, block 0 writes
//          dgamma / dbeta
//   a_side the transformed A operand is also written out by the n-tile-0 blocks (the weight gradient needs it)
//   EPI    td_conv_tile.h: residual add / backward sums of the next BatchNorm in backward order
struct CvFuse {
  const __hip_bfloat16* a2;       // PRO 2: z (x is g)
  const float* part;              // PRO 1: [G, part_rows, K, 2] (sum z, sum z^2); PRO 2: (sum g, sum g (z - mean)); rows 0, stride, ...
  int part_S, part_stride, part_rows;
  const float* gamma;             // [K]
  const float* beta;              // [K]
  float* mean;                    // [G, K]  PRO 1: written; PRO 2: read
  float* invstd;                  // [G, K]
  float* rmean;                   // PRO 1: running statistics (nullable)
  float* rvar;
  float momentum, eps;
  float* dgamma;                  // PRO 2: [K]
  float* dbeta;
  __hip_bfloat16* a_side;         // [M, K] (nullable)
  int G;
  CvEpi ep;
};

constexpr int CV_PRO_MAX_K = 512;        // prologue coefficients live in LDS: 5 floats per reduction channel
constexpr int CV_PRO_MAX_S = 16;         // partial rows a block finishes itself

__device__ __forceinline__ uint4 cv_pro1(uint4 x, const float* sc, const float* sh) {
  const unsigned w[4] = {x.x, x.y, x.z, x.w};
  unsigned o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v0 = fmaf(bf2f((unsigned short)(w[e] & 0xffff)), sc[2 * e], sh[2 * e]);
    float v1 = fmaf(bf2f((unsigned short)(w[e] >> 16)), sc[2 * e + 1], sh[2 * e + 1]);
    v0 = v0 < 0.f ? 0.f : v0;
    v1 = v1 < 0.f ? 0.f : v1;
    o[e] = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ uint4 cv_pro2(uint4 g, uint4 z, const float* k0, const float* k1, const float* k2, const float* sc,
                                         const float* sh) {
  const unsigned gw[4] = {g.x, g.y, g.z, g.w}, zw[4] = {z.x, z.y, z.z, z.w};
  unsigned o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float z0 = bf2f((unsigned short)(zw[e] & 0xffff)), z1 = bf2f((unsigned short)(zw[e] >> 16));
    float g0 = bf2f((unsigned short)(gw[e] & 0xffff)), g1 = bf2f((unsigned short)(gw[e] >> 16));
    g0 = fmaf(z0, sc[2 * e], sh[2 * e]) <= 0.f ? 0.f : g0;                 // threshold_backward on the recomputed pre-activation
    g1 = fmaf(z1, sc[2 * e + 1], sh[2 * e + 1]) <= 0.f ? 0.f : g1;
    const float v0 = fmaf(k0[2 * e], g0, fmaf(k1[2 * e], z0, k2[2 * e]));
    const float v1 = fmaf(k0[2 * e + 1], g1, fmaf(k1[2 * e + 1], z1, k2[2 * e + 1]));
    o[e] = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// (A, B) = ordered sum of the partial rows 0, stride, ... (S of them) of channel c
__device__ __forceinline__ void cv_sum_rows(const float* __restrict__ part, int S, int stride, int K, int c, float& A, float& B) {
  float2 v[CV_PRO_MAX_S];
#pragma unroll
  for (int u = 0; u < CV_PRO_MAX_S; ++u)
    v[u] = *reinterpret_cast<const float2*>(part + ((size_t)(u < S ? u : 0) * stride * K + c) * 2);
  A = 0.f;
  B = 0.f;
#pragma unroll
  for (int u = 0; u < CV_PRO_MAX_S; ++u) {
    A += u < S ? v[u].x : 0.f;
    B += u < S ? v[u].y : 0.f;
  }
}

template <int BM, int BN, bool BT, int PRO, int EPI, bool DEEP>
__global__ __launch_bounds__(CV_THREADS, 2) void conv1x1_mfma_kernel(
    const __hip_bfloat16* __restrict__ x, const __hip_bfloat16* __restrict__ w, __hip_bfloat16* __restrict__ y,
    float* __restrict__ ws, long long Mg, int K, int N, int tiles_per_group, int total_blocks, ConvRows geom, CvFuse fz) {
  using T = CvTile<BM, BN, BT>;
  constexpr int A_BYTES = T::A_BYTES, STAGE = T::STAGE, TN = T::TN, TM = T::TM;
  __shared__ __attribute__((aligned(16))) unsigned char lds[T::LDS_BYTES];
  __shared__ __attribute__((aligned(16))) float s_pro[PRO == 0 ? 4 : (PRO == 1 ? 2 : 5) * CV_PRO_MAX_K];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, h = lane >> 5;

  const int L = cv_xcd_tile(blockIdx.x, total_blocks);
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
  auto a_off = [&](int i) {
    int r = lr + 32 * i;
    r = r < rows_valid ? r : rows_valid - 1;       // rows past the group's end: a valid row, never stored or summed
    return cv_src_row(row0 + r, geom) * (long long)K + lc * 8;
  };
  const long long oa0 = a_off(0), oa1 = a_off(1), oa2 = NA > 2 ? a_off(2) : oa0, oa3 = NA > 2 ? a_off(3) : oa0;
  const __hip_bfloat16 *pa0 = x + oa0, *pa1 = x + oa1, *pa2 = x + oa2, *pa3 = x + oa3;
  // weight tile: [BN rows][64 k] chunks (t & 7) of rows (t >> 3) + 32 i -- or, BT, [64 reduction rows][BN columns]: chunk t % (BN / 8)
  // of rows t / (BN / 8) + (256 / (BN / 8)) i; either way NB 16-byte chunks per thread and stage
  constexpr int CPR = BN / 8, RPP = CV_THREADS / CPR;
  const int bc = tid % CPR, br = tid / CPR;
  const __hip_bfloat16* pb0 = BT ? w + (long long)br * N + n0 + bc * 8 : w + (long long)(n0 + lr) * K + lc * 8;
  const long long bstep = BT ? (long long)RPP * N : 32ll * K;
  const __hip_bfloat16 *pb1 = pb0 + bstep, *pb2 = pb0 + (NB > 2 ? 2 : 0) * bstep, *pb3 = pb0 + (NB > 2 ? 3 : 0) * bstep;
  const long long bkt = BT ? (long long)CV_BK * N : (long long)CV_BK;     // weight pointer advance per K stage
  // DEEP: two register sets (A: even stages, B: odd stages), the loads run TWO stages ahead of the arithmetic.  A K stage is a few
  // MFMAs (~0.15 us) against a global-load round trip of 1-2 us, and the deep layers have at most one block per CU, so with the
  // loads one stage ahead every stage cost a full memory latency (K = 2048: 25 us for a 6 MB problem).  Instantiated where the
  // second set fits without spilling (64 x 64 tiles; 128 x 64 without a prologue) and K has at least three stages.
#define CV_DECL(S)                                                                          \
  uint4 ra0##S, ra1##S, ra2##S, ra3##S, rb0##S, rb1##S, rb2##S, rb3##S, rz0##S, rz1##S, rz2##S, rz3##S; \
  ra0##S = ra1##S = ra2##S = ra3##S = rb0##S = rb1##S = rb2##S = rb3##S = make_uint4(0, 0, 0, 0);       \
  rz0##S = rz1##S = rz2##S = rz3##S = make_uint4(0, 0, 0, 0);
  CV_DECL(A)
  CV_DECL(B)
#define CV_LD(p, kt) (*reinterpret_cast<const uint4*>((p) + (kt) * CV_BK))
#define CV_LDB(p, kt) (*reinterpret_cast<const uint4*>((p) + (kt) * bkt))
#define CV_LOAD_GLOBAL(kt, S)                                  \
  {                                                            \
    ra0##S = CV_LD(pa0, kt);                                   \
    ra1##S = CV_LD(pa1, kt);                                   \
    if (NA > 2) { ra2##S = CV_LD(pa2, kt); ra3##S = CV_LD(pa3, kt); } \
    if constexpr (PRO == 2) {                                  \
      rz0##S = CV_LD(fz.a2 + oa0, kt);                         \
      rz1##S = CV_LD(fz.a2 + oa1, kt);                         \
      if (NA > 2) { rz2##S = CV_LD(fz.a2 + oa2, kt); rz3##S = CV_LD(fz.a2 + oa3, kt); } \
    }                                                          \
    rb0##S = CV_LDB(pb0, kt);                                  \
    rb1##S = CV_LDB(pb1, kt);                                  \
    if (NB > 2) { rb2##S = CV_LDB(pb2, kt); rb3##S = CV_LDB(pb3, kt); } \
  }
#define CV_ST(base, i, v) (*reinterpret_cast<uint4*>((base) + cv_swz(lr + 32 * (i), lc)) = (v))
#define CV_STB(base, i, v)                                                                              \
  {                                                                                                     \
    if constexpr (BT) *reinterpret_cast<uint4*>((base) + (br + RPP * (i)) * T::PITCH_T + bc * 16) = (v); \
    else CV_ST(base, i, v);                                                                             \
  }
  // the A operand passes through the prologue transform on its way into LDS (and, n-tile 0, out to a_side)
#define CV_PRO_A(kt, S)                                                                                 \
  {                                                                                                     \
    if constexpr (PRO != 0) {                                                                           \
      const int c0_ = (kt) * CV_BK + lc * 8;                                                            \
      float q0[8], q1[8], q2[8], q3[8], q4[8];                                                          \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                   \
        q0[e] = s_pro[0 * CV_PRO_MAX_K + c0_ + e];                                                      \
        q1[e] = s_pro[1 * CV_PRO_MAX_K + c0_ + e];                                                      \
        if constexpr (PRO == 2) {                                                                       \
          q2[e] = s_pro[2 * CV_PRO_MAX_K + c0_ + e];                                                    \
          q3[e] = s_pro[3 * CV_PRO_MAX_K + c0_ + e];                                                    \
          q4[e] = s_pro[4 * CV_PRO_MAX_K + c0_ + e];                                                    \
        }                                                                                               \
      }                                                                                                 \
      if constexpr (PRO == 1) {                                                                         \
        ra0##S = cv_pro1(ra0##S, q0, q1);                                                               \
        ra1##S = cv_pro1(ra1##S, q0, q1);                                                               \
        if (NA > 2) { ra2##S = cv_pro1(ra2##S, q0, q1); ra3##S = cv_pro1(ra3##S, q0, q1); }             \
      } else {                                                                                          \
        ra0##S = cv_pro2(ra0##S, rz0##S, q0, q1, q2, q3, q4);                                           \
        ra1##S = cv_pro2(ra1##S, rz1##S, q0, q1, q2, q3, q4);                                           \
        if (NA > 2) { ra2##S = cv_pro2(ra2##S, rz2##S, q0, q1, q2, q3, q4); ra3##S = cv_pro2(ra3##S, rz3##S, q0, q1, q2, q3, q4); } \
      }                                                                                                 \
      if (fz.a_side && nt == 0) {                                                                       \
        __hip_bfloat16* sd_ = fz.a_side + (row0 + lr) * (long long)K + c0_;                             \
        if (lr < rows_valid) *reinterpret_cast<uint4*>(sd_) = ra0##S;                                   \
        if (lr + 32 < rows_valid) *reinterpret_cast<uint4*>(sd_ + 32ll * K) = ra1##S;                   \
        if (NA > 2) {                                                                                   \
          if (lr + 64 < rows_valid) *reinterpret_cast<uint4*>(sd_ + 64ll * K) = ra2##S;                 \
          if (lr + 96 < rows_valid) *reinterpret_cast<uint4*>(sd_ + 96ll * K) = ra3##S;                 \
        }                                                                                               \
      }                                                                                                 \
    }                                                                                                   \
  }
#define CV_WRITE_LDS(stage, kt, S)                                       \
  {                                                                      \
    unsigned char* base_ = lds + (stage) * STAGE;                        \
    CV_PRO_A(kt, S)                                                      \
    CV_ST(base_, 0, ra0##S);                                             \
    CV_ST(base_, 1, ra1##S);                                             \
    if (NA > 2) { CV_ST(base_, 2, ra2##S); CV_ST(base_, 3, ra3##S); }    \
    CV_STB(base_ + A_BYTES, 0, rb0##S);                                  \
    CV_STB(base_ + A_BYTES, 1, rb1##S);                                  \
    if (NB > 2) { CV_STB(base_ + A_BYTES, 2, rb2##S); CV_STB(base_ + A_BYTES, 3, rb3##S); } \
  }

  CvAcc<BM, BN> acc;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc.v[i][j][e] = 0.f;

  const int nk = K / CV_BK;
  CV_LOAD_GLOBAL(0, A)                    // the first tiles are on their way while the prologue finishes the statistics
  if constexpr (DEEP) CV_LOAD_GLOBAL(nk > 1 ? 1 : 0, B)

  if constexpr (PRO == 1) {
    const float* part = fz.part + (size_t)grp * fz.part_rows * K * 2;
    const bool first = (s == 0 && nt == 0);
    for (int c = tid; c < K; c += CV_THREADS) {
      float A, B;
      cv_sum_rows(part, fz.part_S, fz.part_stride, K, c, A, B);
      const double m = (double)A / (double)Mg;
      double var = (double)B / (double)Mg - m * m;      // biased variance, formed in double (as bn_apply_kernel)
      var = var > 0.0 ? var : 0.0;
      const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)fz.eps));
      const float scv = fz.gamma[c] * isf;
      s_pro[c] = scv;
      s_pro[CV_PRO_MAX_K + c] = fz.beta[c] - mf * scv;
      if (first) {
        fz.mean[(size_t)grp * K + c] = mf;
        fz.invstd[(size_t)grp * K + c] = isf;
      }
      if (fz.rmean && first && grp == 0) {             // one momentum update per group, in order (G separate calls)
        double rm = fz.rmean[c], rv = fz.rvar[c];
        for (int g2 = 0; g2 < fz.G; ++g2) {
          float A2 = A, B2 = B;
          if (g2 != 0) cv_sum_rows(fz.part + (size_t)g2 * fz.part_rows * K * 2, fz.part_S, fz.part_stride, K, c, A2, B2);
          const double m2 = (double)A2 / (double)Mg;
          double v2 = (double)B2 / (double)Mg - m2 * m2;
          v2 = v2 > 0.0 ? v2 : 0.0;
          const double unbiased = Mg > 1 ? v2 * ((double)Mg / (double)(Mg - 1)) : v2;
          rm = (double)(float)((1.0 - fz.momentum) * rm + fz.momentum * m2);
          rv = (double)(float)((1.0 - fz.momentum) * rv + fz.momentum * unbiased);
        }
        fz.rmean[c] = (float)rm;
        fz.rvar[c] = (float)rv;
      }
    }
    __syncthreads();
  }
  if constexpr (PRO == 2) {
    const float* part = fz.part + (size_t)grp * fz.part_rows * K * 2;
    const float inv_m = 1.f / (float)Mg;
    const bool first = (s == 0 && nt == 0 && grp == 0);
    for (int c = tid; c < K; c += CV_THREADS) {
      float A, B;
      cv_sum_rows(part, fz.part_S, fz.part_stride, K, c, A, B);
      const float is = fz.invstd[(size_t)grp * K + c], mu = fz.mean[(size_t)grp * K + c];
      const float dg = B * is;                     // sum g * xhat
      const float c0 = fz.gamma[c] * is;
      const float c1 = -c0 * is * dg * inv_m;      // multiplies (z - mean)
      s_pro[0 * CV_PRO_MAX_K + c] = c0;
      s_pro[1 * CV_PRO_MAX_K + c] = c1;
      s_pro[2 * CV_PRO_MAX_K + c] = -c0 * A * inv_m - c1 * mu;
      s_pro[3 * CV_PRO_MAX_K + c] = c0;            // gamma * invstd: the forward's scale
      s_pro[4 * CV_PRO_MAX_K + c] = fz.beta[c] - mu * c0;
      if (first) {
        float dg_tot = 0.f, db_tot = 0.f;
        for (int g2 = 0; g2 < fz.G; ++g2) {
          float A2 = A, B2 = B;
          if (g2 != 0) cv_sum_rows(fz.part + (size_t)g2 * fz.part_rows * K * 2, fz.part_S, fz.part_stride, K, c, A2, B2);
          dg_tot += B2 * fz.invstd[(size_t)g2 * K + c];
          db_tot += A2;
        }
        fz.dgamma[c] = dg_tot;
        fz.dbeta[c] = db_tot;
      }
    }
    __syncthreads();
  }

  CV_WRITE_LDS(0, 0, A)
  __syncthreads();
  if constexpr (!DEEP) {
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more) CV_LOAD_GLOBAL(kt + 1, A)
      const unsigned char* sa = lds + (kt & 1) * STAGE;
      cv_stage_mfma<BM, BN, BT>(acc, sa, sa + A_BYTES, wm, wn, l31, h);
      if (more) CV_WRITE_LDS((kt + 1) & 1, kt + 1, A)
      __syncthreads();
    }
  } else {
    // branch-free inside the stage pair: a conditional load makes the compiler's waitcnt pass drain the whole queue at the join
    // (vmcnt(0) in front of the LDS writes, i.e. no prefetch at all); instead the stage index is clamped -- past the end the last
    // stage is re-read and re-written into the idle buffer, which nobody reads
    const int last = nk - 1;
    for (int kt = 0; kt < nk; kt += 2) {
      const int k2 = kt + 2 < last ? kt + 2 : last, k1 = kt + 1 < last ? kt + 1 : last, k3 = kt + 3 < last ? kt + 3 : last;
      CV_LOAD_GLOBAL(k2, A)
      __builtin_amdgcn_sched_barrier(0);       // keep the requests in front of the stage's arithmetic (the scheduler sinks them)
      cv_stage_mfma<BM, BN, BT>(acc, lds, lds + A_BYTES, wm, wn, l31, h);
      CV_WRITE_LDS(1, k1, B)
      __syncthreads();
      if (kt + 1 >= nk) break;
      CV_LOAD_GLOBAL(k3, B)
      __builtin_amdgcn_sched_barrier(0);
      cv_stage_mfma<BM, BN, BT>(acc, lds + STAGE, lds + STAGE + A_BYTES, wm, wn, l31, h);
      CV_WRITE_LDS(0, k2, A)
      __syncthreads();
    }
  }

  CvEpi ep = fz.ep;
  if constexpr (EPI == 2) { ep.mean += (size_t)grp * N; ep.invstd += (size_t)grp * N; }
  cv_epilogue<BM, BN, EPI>(acc, lds, y, ws, row0, rows_valid, n0, N, (long long)grp * tiles_per_group + s, tid, wm, wn, l31, h, ep);
}

#undef CV_LOAD_GLOBAL
#undef CV_DECL
#undef CV_WRITE_LDS
#undef CV_PRO_A
#undef CV_LD
#undef CV_LDB
#undef CV_ST
#undef CV_STB

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolution:  dW[n, k] = sum_m dY[m, n] * X[src(m), k]   (reference: autograd of
// nn.Conv2d(kernel_size=1) in resnet.py:51-86).  The reduction runs over the pixel index m, i.e. over the ROWS of both
// channels-last matrices, so both MFMA operands are needed "8 consecutive rows of one column" per lane: the tiles go to
// LDS as they lie in memory (rows of 64 channels, 16-byte stores, pitch = 192 bytes) and are read back through the
// hardware transpose ds_read_b64_tr_b16 (two reads per 32x32x16 operand; pitch = 192 mod 256 makes the 32 lanes of a half hit
// 64 distinct banks).  M is cut into P row ranges (the grid is (N/64) x (K/64) x P workgroups), every workgroup writes its
// 64 x 64 fp32 partial tile and conv1x1_wgrad_reduce_kernel adds the P slabs in order: deterministic, no atomics, no
// zero-fill / cast passes around it (MIOpen's split-K weight-gradient kernels need both).
constexpr int WG_T = 64;                         // output tile (n x k) and the reduction rows per stage
constexpr int WG_PITCH = WG_T * 2 + 64;          // bytes per LDS row

typedef __attribute__((ext_vector_type(4))) short cv_s16x4;

__device__ __forceinline__ cv_bf16x8 wg_frag(const unsigned char* tile, int col0, int ks, int lane) {
  // operand fragment of the 32 columns col0.. of a [64 rows][64 cols] bf16 tile for reduction rows 16 ks .. 16 ks + 15:
  // lane l -> column col0 + (l & 31), rows 16 ks + 8 (l >> 5) + 0..7
  const int g = lane >> 4, i = lane & 15;
  const int row = 16 * ks + 8 * (g >> 1) + (i >> 2);
  const unsigned char* p = tile + row * WG_PITCH + (col0 + 16 * (g & 1) + 4 * (i & 3)) * 2;
  typedef __attribute__((address_space(3))) cv_s16x4* lds_ptr;
  const cv_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
  const cv_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * WG_PITCH));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(cv_bf16x8, v);
}

constexpr int WG_TILE_BYTES = WG_T * WG_PITCH, WG_STAGE = 2 * WG_TILE_BYTES, WG_LDS_BYTES = 2 * WG_STAGE;

// one workgroup of the weight gradient: tile / row range L of the problem (dy, x) -> its fp32 partial tile in the slab buffer
__device__ __forceinline__ void wgrad_block(
    const __hip_bfloat16* __restrict__ dy, const __hip_bfloat16* __restrict__ x, float* __restrict__ part, long long M, int K, int N,
    int rows_per_split, int tiles_n, int tiles_k, ConvRows geom, int L, unsigned char* lds) {
  constexpr int TILE_BYTES = WG_TILE_BYTES, STAGE = WG_STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid >> 1, wk = wid & 1;
  const int nt = L % tiles_n, kt = (L / tiles_n) % tiles_k, sp = L / (tiles_n * tiles_k);
  const int n0 = nt * WG_T, k0 = kt * WG_T;
  const long long m_lo = (long long)sp * rows_per_split;
  const long long m_hi = (m_lo + rows_per_split < M) ? m_lo + rows_per_split : M;
  const int nstage = (int)((m_hi - m_lo + WG_T - 1) / WG_T);

  // staging: thread t moves 16-byte chunk (t & 7) of rows (t >> 3) and (t >> 3) + 32 of both tiles.  A stage is four MFMAs per
  // wave (~0.15 us) against a global-load round trip of 1-2 us and a workgroup has only a handful of stages, so the loads run
  // TWO stages ahead of the arithmetic: two register sets (a: even stages, b: odd stages), each written to its LDS buffer one
  // iteration after it was requested (one stage ahead left the kernel waiting on memory at 21 us per layer).
  const int lc = tid & 7, lr = tid >> 3;
  uint4 ry0a, ry1a, rx0a, rx1a, ry0b, ry1b, rx0b, rx1b;
#define WG_LOAD(st, S)                                                                                             \
  {                                                                                                                \
    const long long r0 = m_lo + (long long)(st) * WG_T + lr, r1 = r0 + 32;                                         \
    const uint4 z = make_uint4(0, 0, 0, 0);                                                                        \
    ry0##S = r0 < m_hi ? *reinterpret_cast<const uint4*>(dy + r0 * N + n0 + lc * 8) : z;                           \
    ry1##S = r1 < m_hi ? *reinterpret_cast<const uint4*>(dy + r1 * N + n0 + lc * 8) : z;                           \
    rx0##S = r0 < m_hi ? *reinterpret_cast<const uint4*>(x + cv_src_row(r0, geom) * (long long)K + k0 + lc * 8) : z; \
    rx1##S = r1 < m_hi ? *reinterpret_cast<const uint4*>(x + cv_src_row(r1, geom) * (long long)K + k0 + lc * 8) : z; \
  }
#define WG_WRITE(buf, S)                                                                         \
  {                                                                                              \
    unsigned char* b_ = lds + (buf) * STAGE;                                                     \
    *reinterpret_cast<uint4*>(b_ + lr * WG_PITCH + lc * 16) = ry0##S;                            \
    *reinterpret_cast<uint4*>(b_ + (lr + 32) * WG_PITCH + lc * 16) = ry1##S;                     \
    *reinterpret_cast<uint4*>(b_ + TILE_BYTES + lr * WG_PITCH + lc * 16) = rx0##S;               \
    *reinterpret_cast<uint4*>(b_ + TILE_BYTES + (lr + 32) * WG_PITCH + lc * 16) = rx1##S;        \
  }
#define WG_MFMA(buf)                                                                             \
  {                                                                                              \
    const unsigned char* ty = lds + (buf) * STAGE;                                               \
    const unsigned char* tx = ty + TILE_BYTES;                                                   \
    _Pragma("unroll") for (int ks = 0; ks < WG_T / 16; ++ks) {                                   \
      const cv_bf16x8 fa = wg_frag(ty, wn * 32, ks, lane); /* A[i = n][r = m] */                 \
      const cv_bf16x8 fb = wg_frag(tx, wk * 32, ks, lane); /* B[r = m][j = k] */                 \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);                       \
    }                                                                                            \
  }
  cv_f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  WG_LOAD(0, a)
  if (nstage > 1) WG_LOAD(1, b)
  WG_WRITE(0, a)
  __syncthreads();
  for (int st = 0; st < nstage; st += 2) {
    if (st + 2 < nstage) WG_LOAD(st + 2, a)
    WG_MFMA(0)
    if (st + 1 < nstage) WG_WRITE(1, b)
    __syncthreads();
    if (st + 1 >= nstage) break;
    if (st + 3 < nstage) WG_LOAD(st + 3, b)
    WG_MFMA(1)
    if (st + 2 < nstage) WG_WRITE(0, a)
    __syncthreads();
  }
#undef WG_MFMA
#undef WG_LOAD
#undef WG_WRITE
  // D[i = n][j = k]: lane -> k = k0 + 32 wk + (lane & 31), register r -> n = n0 + 32 wn + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* out = part + ((size_t)sp * N + n0 + wn * 32 + 4 * (lane >> 5)) * K + k0 + wk * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) out[(size_t)((r & 3) + 8 * (r >> 2)) * K] = acc[r];
}

__global__ __launch_bounds__(CV_THREADS, 2) void conv1x1_wgrad_kernel(
    const __hip_bfloat16* __restrict__ dy, const __hip_bfloat16* __restrict__ x, float* __restrict__ part, long long M, int K, int N,
    int rows_per_split, int tiles_n, int tiles_k, ConvRows geom) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[WG_LDS_BYTES];
  // all (n, k) tiles of one row range read the same rows of dY and X: with the round-robin dispatch every XCD would pull every
  // row range through its own L2 (8 x the operand bytes over the fabric); the remap gives each XCD a contiguous run of ranges
  wgrad_block(dy, x, part, M, K, N, rows_per_split, tiles_n, tiles_k, geom, cv_xcd_tile(blockIdx.x, gridDim.x), lds);
}

// ---- grouped form: the weight gradients of MANY 1x1 convolutions in one launch (round 4).  A training step has ~90 of them; each
// is a latency-bound launch of 13-35 us plus its slab sum (5.6 us), and nothing downstream needs any of them before the optimiser
// runs, so the autograd nodes only ENQUEUE their problem (tripled_amd.ops.deferred_wgrads) and one grouped launch per <= 40 problems
// runs them side by side: the per-problem tiling, row ranges and summation order are exactly those of the single form (bit-equal).
constexpr int WG_GROUP_MAX = 40;
struct WgProblem {
  const __hip_bfloat16* dy;
  const __hip_bfloat16* x;
  float* part;
  long long M;
  int K, N, rows_per_split, tiles_n, tiles_k;
  ConvRows geom;
};
struct WgGroupArgs {
  WgProblem p[WG_GROUP_MAX];
  int first_block[WG_GROUP_MAX + 1];
  int n;
};
__global__ __launch_bounds__(CV_THREADS, 2) void conv1x1_wgrad_group_kernel(const WgGroupArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[WG_LDS_BYTES];
  const int L = cv_xcd_tile(blockIdx.x, gridDim.x);       // XCD-contiguous runs of blocks, i.e. of whole row ranges of a problem
  int lo = 0, hi = a.n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.first_block[mid] <= L) lo = mid; else hi = mid;
  }
  const WgProblem& q = a.p[lo];
  wgrad_block(q.dy, q.x, q.part, q.M, q.K, q.N, q.rows_per_split, q.tiles_n, q.tiles_k, q.geom, L - a.first_block[lo], lds);
}

// dW = sum over the P slabs, in a fixed order: a block owns 64 consecutive elements (16 float4 columns) and splits the slabs
// over 16 lanes groups (p = g, g + 16, ...), combined through LDS -- a 64 x 64 weight with 360 slabs is 64 blocks of 23-deep
// loops instead of 4 blocks of 360-deep ones.
template <typename T>
__global__ __launch_bounds__(TD_THREADS) void conv1x1_wgrad_reduce_kernel(const float* __restrict__ part, int P, long long NK, T* __restrict__ dw) {
  __shared__ float4 red[16][16];
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long long i = ((long long)blockIdx.x * 16 + c) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < NK) {
    for (int p = g; p < P; p += 16) {
      const float4 b = *reinterpret_cast<const float4*>(part + (size_t)p * NK + i);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
  }
  red[g][c] = a;
  __syncthreads();
  if (g == 0 && i < NK) {
#pragma unroll
    for (int j = 1; j < 16; ++j) {
      const float4 b = red[j][c];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if constexpr (sizeof(T) == 2) {
      uint2 o;
      o.x = (unsigned)f2bf(a.x) | ((unsigned)f2bf(a.y) << 16);
      o.y = (unsigned)f2bf(a.z) | ((unsigned)f2bf(a.w) << 16);
      *reinterpret_cast<uint2*>(dw + i) = o;
    } else {
      *reinterpret_cast<float4*>(dw + i) = a;
    }
  }
}

// grouped slab sum: block b of problem p handles 64 consecutive elements exactly like conv1x1_wgrad_reduce_kernel
constexpr int RD_GROUP_MAX = 96;
struct RdProblem {
  const float* part;
  void* dw;
  long long NK;
  int P, bf16;
};
struct RdGroupArgs {
  RdProblem p[RD_GROUP_MAX];
  int first_block[RD_GROUP_MAX + 1];
  int n;
};
__global__ __launch_bounds__(TD_THREADS) void conv1x1_wgrad_reduce_group_kernel(const RdGroupArgs a) {
  __shared__ float4 red[16][16];
  int lo = 0, hi = a.n;
  const int b = (int)blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.first_block[mid] <= b) lo = mid; else hi = mid;
  }
  const RdProblem& q = a.p[lo];
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long long i = ((long long)(b - a.first_block[lo]) * 16 + c) * 4;
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < q.NK) {
    for (int p = g; p < q.P; p += 16) {
      const float4 v = *reinterpret_cast<const float4*>(q.part + (size_t)p * q.NK + i);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
  }
  red[g][c] = s4;
  __syncthreads();
  if (g == 0 && i < q.NK) {
#pragma unroll
    for (int j = 1; j < 16; ++j) {
      const float4 v = red[j][c];
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    if (q.bf16) {
      uint2 o;
      o.x = (unsigned)f2bf(s4.x) | ((unsigned)f2bf(s4.y) << 16);
      o.y = (unsigned)f2bf(s4.z) | ((unsigned)f2bf(s4.w) << 16);
      *reinterpret_cast<uint2*>(reinterpret_cast<__hip_bfloat16*>(q.dw) + i) = o;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(q.dw) + i) = s4;
    }
  }
}

// dW = ordered sum of P fp32 slabs of NK elements, rounded once to the weight's dtype (shared with td_conv3x3_wgrad.hip)
int cv_wgrad_reduce(const float* part, int P, long long NK, int dw_dtype, void* dw, hipStream_t st) {
  const unsigned blocks = (unsigned)((NK / 4 + 15) / 16);
  if (dw_dtype == TD_DTYPE_BF16)
    hipLaunchKernelGGL((conv1x1_wgrad_reduce_kernel<__hip_bfloat16>), dim3(blocks), dim3(TD_THREADS), 0, st, part, P, NK, (__hip_bfloat16*)dw);
  else
    hipLaunchKernelGGL((conv1x1_wgrad_reduce_kernel<float>), dim3(blocks), dim3(TD_THREADS), 0, st, part, P, NK, (float*)dw);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

static inline int wg_splits(long long M, int K, int N) {
  const long long tiles = (long long)(N / WG_T) * (K / WG_T);
  long long p = (768 + tiles - 1) / tiles;                // 3 workgroups per CU = what its LDS holds at once (768 / 512 / 1024: 355 / 377 / 396 us over the probe's shapes)
  const long long by_rows = (M + 4 * WG_T - 1) / (4 * WG_T);   // at least four stages per workgroup
  if (p > by_rows) p = by_rows;
  if (p > 512) p = 512;
  return (int)(p < 1 ? 1 : p);
}

template <int BM, int BN, bool BT, int PRO, int EPI, bool DEEP>
static int cv_launch(const void* x, const void* w, void* y, float* ws, long long Mg, int G, int K, int N, ConvRows geom, hipStream_t st,
                     const CvFuse& fz) {
  const int tpg = (int)((Mg + BM - 1) / BM);
  const long long nblk = (long long)G * tpg * (N / BN);
  if (nblk > 0x7fffffffll) return TD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((conv1x1_mfma_kernel<BM, BN, BT, PRO, EPI, DEEP>), dim3((unsigned)nblk), dim3(CV_THREADS), 0, st, (const __hip_bfloat16*)x,
                     (const __hip_bfloat16*)w, (__hip_bfloat16*)y, ws, Mg, K, N, tpg, (int)nblk, geom, fz);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

// tile choice as the plain forward; one instantiation set per fused form
template <bool BT, int PRO, int EPI>
static int cv_dispatch(const void* x, const void* w, void* y, float* ws, long long M, int G, int K, int N, ConvRows geom, hipStream_t st,
                       const CvFuse& fz) {
  const long long Mg = M / G;
  ConvTile t = cv_pick_tile(Mg, G, N);
  const bool deep = K >= 3 * CV_BK;
  if constexpr (PRO == 2) {      // its prologue coefficients + the transposed weight stage leave no room for two 128 x 128 blocks per CU
    if (t.bm == 128 && t.bn == 128) t.bn = 64;
  } else
    if (t.bm == 128 && t.bn == 128) return cv_launch<128, 128, BT, PRO, EPI, false>(x, w, y, ws, Mg, G, K, N, geom, st, fz);
  if (t.bm == 128 && t.bn == 64) {
    if constexpr (PRO == 0)
      if (deep) return cv_launch<128, 64, BT, PRO, EPI, true>(x, w, y, ws, Mg, G, K, N, geom, st, fz);
    return cv_launch<128, 64, BT, PRO, EPI, false>(x, w, y, ws, Mg, G, K, N, geom, st, fz);
  }
  if (deep) return cv_launch<64, 64, BT, PRO, EPI, true>(x, w, y, ws, Mg, G, K, N, geom, st, fz);
  return cv_launch<64, 64, BT, PRO, EPI, false>(x, w, y, ws, Mg, G, K, N, geom, st, fz);
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
  return td::cv_dispatch<false, 0, 0>(x, w, y, stat_partials, M, groups, K, N, geom, (hipStream_t)stream, td::CvFuse{});
}

// partial rows (any producer) reduced in place to at most CV_PRO_MAX_S rows for a GEMM prologue; returns (rows, stride)
namespace td {
BnRows bn_shrink_partials_to(float* partials, int S, int G, int C, int max_rows, hipStream_t st);   // td_bn.hip
}

extern "C" int td_conv1x1_fwd_bnrelu(const void* z, const void* w, long long M, int groups, int K, int N, float* in_partials,
                                     int in_rows, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     float momentum, float eps, float* save_mean, float* save_invstd, void* a_side, void* y,
                                     float* stat_partials, td_stream_t stream) {
  if (!z || !w || !y || !in_partials || in_rows < 1 || !gamma || !beta || !save_mean || !save_invstd || !cv_shape_ok(M, groups, K, N))
    return TD_ERR_BAD_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return TD_ERR_BAD_ARG;
  if (K > td::CV_PRO_MAX_K || M * (long long)(K > N ? K : N) >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const td::BnRows rows = td::bn_shrink_partials_to(in_partials, in_rows, groups, K, td::CV_PRO_MAX_S, st);
  td::CvFuse fz = {};
  fz.part = in_partials; fz.part_S = rows.n; fz.part_stride = rows.stride; fz.part_rows = in_rows;
  fz.gamma = gamma; fz.beta = beta; fz.mean = save_mean; fz.invstd = save_invstd; fz.rmean = running_mean; fz.rvar = running_var;
  fz.momentum = momentum; fz.eps = eps; fz.a_side = (__hip_bfloat16*)a_side; fz.G = groups;
  return td::cv_dispatch<false, 1, 0>(z, w, y, stat_partials, M, groups, K, N, td::ConvRows{0, 0, 0, 0, 1}, st, fz);
}

// y = conv1x1(x, w) and run_out = run_in + y in one launch (the CRP block's pointwise convolution + running sum)
extern "C" int td_conv1x1_fwd_sum(const void* x, const void* w, long long M, int K, int N, const void* run_in, void* y, void* run_out,
                                  td_stream_t stream) {
  if (!x || !w || !y || !run_in || !run_out || !cv_shape_ok(M, 1, K, N)) return TD_ERR_BAD_ARG;
  if (M * (long long)(K > N ? K : N) >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  td::CvFuse fz = {};
  fz.ep.res = (const __hip_bfloat16*)run_in;
  fz.ep.y2 = (__hip_bfloat16*)run_out;
  return td::cv_dispatch<false, 0, 3>(x, w, y, nullptr, M, 1, K, N, td::ConvRows{0, 0, 0, 0, 1}, (hipStream_t)stream, fz);
}

// Data gradient dX[M, Cin] = dY[M, Cout] . W[Cout, Cin] (stride 1).
extern "C" int td_conv1x1_dgrad(const void* dy, const void* w, long long M, int groups, int Cout, int Cin, const void* residual,
                                void* dx, td_stream_t stream) {
  if (!dy || !w || !dx || !cv_shape_ok(M, groups, Cout, Cin)) return TD_ERR_BAD_ARG;
  if (M * (long long)(Cout > Cin ? Cout : Cin) >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  td::CvFuse fz = {};
  fz.ep.res = (const __hip_bfloat16*)residual;
  const td::ConvRows geom = {0, 0, 0, 0, 1};
  if (residual) return td::cv_dispatch<true, 0, 1>(dy, w, dx, nullptr, M, groups, Cout, Cin, geom, (hipStream_t)stream, fz);
  return td::cv_dispatch<true, 0, 0>(dy, w, dx, nullptr, M, groups, Cout, Cin, geom, (hipStream_t)stream, fz);
}

extern "C" int td_conv1x1_dgrad_bnsums(const void* dy, const void* w, long long M, int groups, int Cout, int Cin, const void* z,
                                       const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                                       void* dx, float* out_partials, td_stream_t stream) {
  if (!dy || !w || !dx || !z || !gamma || !beta || !save_mean || !save_invstd || !out_partials || !cv_shape_ok(M, groups, Cout, Cin))
    return TD_ERR_BAD_ARG;
  if (M * (long long)(Cout > Cin ? Cout : Cin) >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  td::CvFuse fz = {};
  fz.ep.z = (const __hip_bfloat16*)z; fz.ep.gamma = gamma; fz.ep.beta = beta; fz.ep.mean = save_mean; fz.ep.invstd = save_invstd;
  return td::cv_dispatch<true, 0, 2>(dy, w, dx, out_partials, M, groups, Cout, Cin, td::ConvRows{0, 0, 0, 0, 1}, (hipStream_t)stream, fz);
}

extern "C" int td_conv1x1_dgrad_bnbwd(const void* g, const void* z, const void* w, long long M, int groups, int Cout, int Cin,
                                      float* in_partials, int in_rows, const float* gamma, const float* beta, const float* save_mean,
                                      const float* save_invstd, float* dgamma, float* dbeta, void* dz_side, const void* residual,
                                      void* dx, td_stream_t stream) {
  if (!g || !z || !w || !dx || !in_partials || in_rows < 1 || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta ||
      !cv_shape_ok(M, groups, Cout, Cin))
    return TD_ERR_BAD_ARG;
  if (Cout > td::CV_PRO_MAX_K || M * (long long)(Cout > Cin ? Cout : Cin) >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const td::BnRows rows = td::bn_shrink_partials_to(in_partials, in_rows, groups, Cout, td::CV_PRO_MAX_S, st);
  td::CvFuse fz = {};
  fz.a2 = (const __hip_bfloat16*)z; fz.part = in_partials; fz.part_S = rows.n; fz.part_stride = rows.stride; fz.part_rows = in_rows;
  fz.gamma = gamma; fz.beta = beta; fz.mean = const_cast<float*>(save_mean); fz.invstd = const_cast<float*>(save_invstd);
  fz.dgamma = dgamma; fz.dbeta = dbeta; fz.a_side = (__hip_bfloat16*)dz_side; fz.G = groups;
  fz.ep.res = (const __hip_bfloat16*)residual;
  const td::ConvRows geom = {0, 0, 0, 0, 1};
  if (residual) return td::cv_dispatch<true, 2, 1>(g, w, dx, nullptr, M, groups, Cout, Cin, geom, st, fz);
  return td::cv_dispatch<true, 2, 0>(g, w, dx, nullptr, M, groups, Cout, Cin, geom, st, fz);
}

extern "C" long long td_conv1x1_wgrad_workspace_floats(long long M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0 || K % 64 != 0 || N % 64 != 0) return 0;
  return (long long)td::wg_splits(M, K, N) * N * K;
}

extern "C" int td_conv1x1_wgrad(const void* dy, const void* x, long long M, int K, int N, int Hi, int Wi, int stride, int dw_dtype,
                                void* dw, float* workspace, td_stream_t stream) {
  if (!dy || !x || !dw || !workspace || M <= 0 || K <= 0 || N <= 0 || K % 64 != 0 || N % 64 != 0 || stride < 1) return TD_ERR_BAD_ARG;
  if (dw_dtype != TD_DTYPE_BF16 && dw_dtype != TD_DTYPE_F32) return TD_ERR_UNSUPPORTED;
  td::ConvRows geom = {0, 0, 0, 0, 1};
  if (stride > 1) {
    if (Hi <= 0 || Wi <= 0) return TD_ERR_BAD_ARG;
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    if (M % ((long long)Ho * Wo) != 0) return TD_ERR_BAD_ARG;
    geom = {Wo, Ho * Wo, Wi, Hi * Wi, stride};
  }
  if (M * (long long)(K > N ? K : N) >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int P = td::wg_splits(M, K, N);
  long long rps = (M + P - 1) / P;
  rps = (rps + td::WG_T - 1) / td::WG_T * td::WG_T;
  const int P_eff = (int)((M + rps - 1) / rps);          // (<= P: trailing empty ranges are not launched; the reduce reads P_eff slabs)
  const int tn = N / td::WG_T, tk = K / td::WG_T;
  hipLaunchKernelGGL(td::conv1x1_wgrad_kernel, dim3((unsigned)(tn * tk * P_eff)), dim3(td::CV_THREADS), 0, st, (const __hip_bfloat16*)dy,
                     (const __hip_bfloat16*)x, workspace, M, K, N, (int)rps, tn, tk, geom);
  const long long NK = (long long)N * K;
  const unsigned blocks = (unsigned)((NK / 4 + 15) / 16);
  if (dw_dtype == TD_DTYPE_BF16)
    hipLaunchKernelGGL((td::conv1x1_wgrad_reduce_kernel<__hip_bfloat16>), dim3(blocks), dim3(TD_THREADS), 0, st, (const float*)workspace, P_eff,
                       NK, (__hip_bfloat16*)dw);
  else
    hipLaunchKernelGGL((td::conv1x1_wgrad_reduce_kernel<float>), dim3(blocks), dim3(TD_THREADS), 0, st, (const float*)workspace, P_eff, NK,
                       (float*)dw);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

// Grouped weight gradients: problem i is exactly td_conv1x1_wgrad(dy[i], x[i], M[i], K[i], N[i], Hi[i], Wi[i], stride[i], dw_dtype[i],
// dw[i], workspace[i]) -- same tiling, same row ranges, same summation order, bit-equal results -- but <= 40 problems share a launch
// (and <= 96 slab sums).  All array arguments are HOST arrays of length n.
extern "C" int td_conv1x1_wgrad_group(int n, const void* const* dy, const void* const* x, const long long* M, const int* K, const int* N,
                                      const int* Hi, const int* Wi, const int* stride, const int* dw_dtype, void* const* dw,
                                      float* const* workspace, td_stream_t stream) {
  if (n < 0 || (n > 0 && (!dy || !x || !M || !K || !N || !Hi || !Wi || !stride || !dw_dtype || !dw || !workspace))) return TD_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n; ++i) {
    if (!dy[i] || !x[i] || !dw[i] || !workspace[i] || M[i] <= 0 || K[i] <= 0 || N[i] <= 0 || K[i] % 64 != 0 || N[i] % 64 != 0 || stride[i] < 1)
      return TD_ERR_BAD_ARG;
    if (dw_dtype[i] != TD_DTYPE_BF16 && dw_dtype[i] != TD_DTYPE_F32) return TD_ERR_UNSUPPORTED;
    if (M[i] * (long long)(K[i] > N[i] ? K[i] : N[i]) >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
    if (stride[i] > 1) {
      if (Hi[i] <= 0 || Wi[i] <= 0) return TD_ERR_BAD_ARG;
      const int Ho = (Hi[i] - 1) / stride[i] + 1, Wo = (Wi[i] - 1) / stride[i] + 1;
      if (M[i] % ((long long)Ho * Wo) != 0) return TD_ERR_BAD_ARG;
    }
  }
  int P_eff[4096];
  if (n > 4096) return TD_ERR_UNSUPPORTED;
  for (int base = 0; base < n; base += td::WG_GROUP_MAX) {
    td::WgGroupArgs a;
    a.n = n - base < td::WG_GROUP_MAX ? n - base : td::WG_GROUP_MAX;
    long long blocks = 0;
    for (int j = 0; j < a.n; ++j) {
      const int i = base + j;
      td::ConvRows geom = {0, 0, 0, 0, 1};
      if (stride[i] > 1) {
        const int Ho = (Hi[i] - 1) / stride[i] + 1, Wo = (Wi[i] - 1) / stride[i] + 1;
        geom = {Wo, Ho * Wo, Wi[i], Hi[i] * Wi[i], stride[i]};
      }
      const int P = td::wg_splits(M[i], K[i], N[i]);
      long long rps = (M[i] + P - 1) / P;
      rps = (rps + td::WG_T - 1) / td::WG_T * td::WG_T;
      P_eff[i] = (int)((M[i] + rps - 1) / rps);
      td::WgProblem& q = a.p[j];
      q.dy = (const __hip_bfloat16*)dy[i]; q.x = (const __hip_bfloat16*)x[i]; q.part = workspace[i]; q.M = M[i]; q.K = K[i]; q.N = N[i];
      q.rows_per_split = (int)rps; q.tiles_n = N[i] / td::WG_T; q.tiles_k = K[i] / td::WG_T; q.geom = geom;
      a.first_block[j] = (int)blocks;
      blocks += (long long)q.tiles_n * q.tiles_k * P_eff[i];
      if (blocks > 0x7fffffffll) return TD_ERR_UNSUPPORTED;
    }
    for (int j = a.n; j <= td::WG_GROUP_MAX; ++j) a.first_block[j] = (int)blocks;
    hipLaunchKernelGGL(td::conv1x1_wgrad_group_kernel, dim3((unsigned)blocks), dim3(td::CV_THREADS), 0, st, a);
  }
  for (int base = 0; base < n; base += td::RD_GROUP_MAX) {
    td::RdGroupArgs a;
    a.n = n - base < td::RD_GROUP_MAX ? n - base : td::RD_GROUP_MAX;
    long long blocks = 0;
    for (int j = 0; j < a.n; ++j) {
      const int i = base + j;
      td::RdProblem& q = a.p[j];
      q.part = workspace[i]; q.dw = dw[i]; q.NK = (long long)N[i] * K[i]; q.P = P_eff[i]; q.bf16 = dw_dtype[i] == TD_DTYPE_BF16;
      a.first_block[j] = (int)blocks;
      blocks += (q.NK / 4 + 15) / 16;
    }
    for (int j = a.n; j <= td::RD_GROUP_MAX; ++j) a.first_block[j] = (int)blocks;
    hipLaunchKernelGGL(td::conv1x1_wgrad_reduce_group_kernel, dim3((unsigned)blocks), dim3(TD_THREADS), 0, st, a);
  }
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}
