// Training-mode batch normalisation of channels-last activations with the residual add and ReLU that
// follow it in every ResNet block of the encoders fused in (reference: mono/model/mono_fm_joint/resnet.py:
// 30-49 BasicBlock.forward, 66-86 Bottleneck.forward: conv -> bn -> [+ identity] -> relu).
//
// The activation tensor is a row-major [M, C] matrix (M = N*H*W pixels, C channels, C % 64 == 0).  A
// block is 8 x 32 threads: 8 threads x 8 channels cover one 64-channel group of a row with 16-byte
// loads, 32 rows per iteration, four iterations in flight.  Per-channel sums are staged and
// deterministic: per-block partial rows -> (more than 96 rows: shrunk in place to <= 96) -> finished in the
// prologue of every block of the consuming kernel, all blocks forming the same ordered sum (kernel boundaries
// are the only inter-block synchronisation; no atomics, no fences, no finalize launch).
//
//   forward : stats (sum x, sum x^2 partials; or the producing convolution's epilogue) [-> shrink]
//             -> apply   prologue: mean, 1/std, running stats;  y = relu?( (x - mean) * invstd * gamma + beta [+ residual] )
//   backward: reduce (sum g, sum g*(x-mean) partials; g = dy * [y > 0]) [-> shrink]
//             -> dx      prologue: dgamma, dbeta, coefficients;  dx = gamma*invstd * (g - dbeta/M - xhat * dgamma/M),  dresidual = g
#include <hip/hip_bf16.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

constexpr int BN_RY = 32;        // rows per block iteration
constexpr int BN_UNROLL = 4;     // row iterations in flight
constexpr int BN_THREADS = 256;  // 8 channel-vector lanes x 32 rows

// ---------------------------------------------------------------------------------------------
// per-channel partial sums of a [rows, 64-channel] slab: MODE 0: (x, x^2)   MODE 1: (g, g * (x - mean))
// ---------------------------------------------------------------------------------------------
// relu: 0 none, 1 mask = [y > 0] read from y, 2 mask recomputed from x as [fma(x, gamma*invstd, beta - mean*gamma*invstd) > 0]
// (exactly the forward's pre-activation value; only valid without a residual)
template <typename T, int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_partials_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                 const T* __restrict__ y, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 long long M, int C, int rows_per_split, int relu,
                                                                 float* __restrict__ ws) {
  // M = rows of ONE statistics group; blockIdx.z = group (consecutive row ranges of the tensor)
  __shared__ float lds[2][BN_RY][64 + 1];
  const int tid = threadIdx.x, cx = tid & 7, ry = tid >> 3;
  const int cg = blockIdx.x, s = blockIdx.y, grp = blockIdx.z;
  const long long base = (long long)grp * M;
  const long long r0 = base + (long long)s * rows_per_split;
  const long long r1 = (r0 + rows_per_split < base + M) ? r0 + rows_per_split : base + M;
  const size_t col = (size_t)cg * 64 + (size_t)cx * 8;
  ws += (size_t)grp * gridDim.y * C * 2;
  if (MODE == 1) { mean += (size_t)grp * C; invstd += (size_t)grp * C; }
  float a[8], b[8], mu[8], sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = 0.f; b[i] = 0.f; mu[i] = 0.f; sc[i] = 0.f; sh[i] = 0.f; }
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) mu[i] = mean[col + i];
    if (relu == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sc[i] = gamma[col + i] * invstd[col + i];
        sh[i] = beta[col + i] - mu[i] * sc[i];
      }
    }
  }
  for (long long r = r0 + ry; r < r1; r += BN_UNROLL * BN_RY) {
    float vx[BN_UNROLL][8], vg[BN_UNROLL][8], vy[BN_UNROLL][8];
    bool ok[BN_UNROLL];
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {       // unconditional loads from clamped (valid) rows
      const long long rr = r + (long long)u * BN_RY;
      ok[u] = rr < r1;
      const size_t off = (size_t)(ok[u] ? rr : r) * C + col;
      load8(x + off, vx[u]);
      if (MODE == 1) {
        load8(dy + off, vg[u]);
        if (relu == 1) load8(y + off, vy[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) {
          const float t = ok[u] ? vx[u][i] : 0.f;
          a[i] += t;
          b[i] = fmaf(t, t, b[i]);
        } else {
          float g = vg[u][i];
          if (relu == 1) g = vy[u][i] <= 0.f ? 0.f : g;   // threshold_backward
          if (relu == 2) g = fmaf(vx[u][i], sc[i], sh[i]) <= 0.f ? 0.f : g;
          g = ok[u] ? g : 0.f;
          a[i] += g;
          b[i] = fmaf(g, vx[u][i] - mu[i], b[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { lds[0][ry][cx * 8 + i] = a[i]; lds[1][ry][cx * 8 + i] = b[i]; }
  __syncthreads();
  if (tid < 128) {
    const int which = tid >> 6, c = tid & 63;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < BN_RY; ++j) acc += lds[which][j][c];
    ws[((size_t)s * C + (size_t)cg * 64 + c) * 2 + which] = acc;
  }
}

// sums the S partials of one BN_FIN_CH-channel group; result (sumA, sumB) valid in threads 0..BN_FIN_CH-1.
// 16 channels per block (128 contiguous bytes per partial row): a 64-channel layer still gets 4 blocks and a 2048-channel one
// 128, where one block per 64 channels left the finalize of the wide layers on 1..32 CUs, latency-bound at ~6 us.
constexpr int BN_FIN_CH = 16;
constexpr int BN_FIN_PARTS = 16;                      // finalize block: 16 channels x 16 slices of the S partials
constexpr int BN_FIN_THREADS = BN_FIN_CH * BN_FIN_PARTS;
__device__ __forceinline__ void bn_sum_partials(const float* __restrict__ ws, int S, int C, int cg, float& A, float& B) {
  __shared__ float red[2][BN_FIN_PARTS][BN_FIN_CH];
  const int tid = threadIdx.x, c = tid % BN_FIN_CH, part = tid / BN_FIN_CH;
  float a = 0.f, b = 0.f;
  // the partial rows were written by other CUs a kernel ago (L2 / fabric latency per load): issue 12 loads before the first add;
  // the adds keep their order, so the sums are the same bit for bit
  int s = part;
  for (; s + 11 * BN_FIN_PARTS < S; s += 12 * BN_FIN_PARTS) {
    float2 p[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) p[u] = *reinterpret_cast<const float2*>(ws + ((size_t)(s + u * BN_FIN_PARTS) * C + (size_t)cg * BN_FIN_CH + c) * 2);
#pragma unroll
    for (int u = 0; u < 12; ++u) { a += p[u].x; b += p[u].y; }
  }
  for (; s < S; s += BN_FIN_PARTS) {
    const float2 p = *reinterpret_cast<const float2*>(ws + ((size_t)s * C + (size_t)cg * BN_FIN_CH + c) * 2);
    a += p.x;
    b += p.y;
  }
  red[0][part][c] = a;
  red[1][part][c] = b;
  __syncthreads();
  A = 0.f;
  B = 0.f;
  if (tid < BN_FIN_CH) {
#pragma unroll
    for (int j = 0; j < BN_FIN_PARTS; ++j) { A += red[0][j][c]; B += red[1][j][c]; }
  }
}

// More than BN_PRO_MAX_S partial rows (the two shallow ResNet stages: up to 1440 row tiles per group): shrink them first, in
// place -- block (cg, q) sums rows [q Sq, (q + 1) Sq) of its 16 channels in order and leaves the result in row q Sq, so the
// consumer's prologue (bn_tile_sums with row stride Sq) sees at most BN_PRO_MAX_S rows.  Q times more blocks than a one-stage
// finalize with 1/Q of its dependent loads each: ~3 us where the finalize kernels took 7-8 us, and no finalize kernel at all
// for the layers with few rows.
__global__ __launch_bounds__(BN_FIN_THREADS) void bn_shrink_kernel(float* __restrict__ ws, int S, int C, int Sq) {
  const int q = blockIdx.y, grp = blockIdx.z;
  float* w = ws + ((size_t)grp * S + (size_t)q * Sq) * C * 2;
  const int rows = S - q * Sq < Sq ? S - q * Sq : Sq;
  float A, B;
  bn_sum_partials(w, rows, C, blockIdx.x, A, B);       // its barrier orders every read of row 0 before the write below
  if (threadIdx.x < BN_FIN_CH) {
    float* o = w + ((size_t)blockIdx.x * BN_FIN_CH + threadIdx.x) * 2;
    o[0] = A;
    o[1] = B;
  }
}

// ---------------------------------------------------------------------------------------------
// Cross-rank (synchronised) statistics: the partial sums of this rank are reduced to per-channel sums, the
// caller all-reduces them over the data-parallel group (2*G*C floats), and the finalize runs on the GLOBAL sums
// with the global row count.  Same kernels either side (bn_partials / bn_apply / bn_dx).
// ---------------------------------------------------------------------------------------------
// mean / invstd / running statistics from global (sum x, sum x^2) over `count` rows (count: device scalar, the
// all-reduced row count, so that ranks with different batch sizes stay correct)
__global__ __launch_bounds__(64) void bn_stats_from_sums_kernel(const float* __restrict__ sums, const float* __restrict__ count,
                                                               int C, int G, float eps, float momentum,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const double M = (double)count[0];
  for (int grp = 0; grp < G; ++grp) {
    const size_t i = (size_t)grp * C + c;
    const double m = (double)sums[i * 2] / M;
    double var = (double)sums[i * 2 + 1] / M - m * m;
    var = var > 0.0 ? var : 0.0;
    save_mean[i] = (float)m;
    save_invstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
      const double unbiased = M > 1.0 ? var * (M / (M - 1.0)) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
  }
}

// dgamma / dbeta from the LOCAL sums (the parameter gradients are averaged by the gradient all-reduce like every
// other parameter), dx coefficients from the GLOBAL sums and the global row count
__global__ __launch_bounds__(64) void bn_coef_from_sums_kernel(const float* __restrict__ local, const float* __restrict__ global,
                                                              const float* __restrict__ count, int C, int G,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ coef) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const float inv_m = 1.f / count[0];
  float dg_tot = 0.f, db_tot = 0.f;
  for (int grp = 0; grp < G; ++grp) {
    const size_t i = (size_t)grp * C + c;
    const float is = invstd[i], mu = mean[i];
    dg_tot += local[i * 2 + 1] * is;
    db_tot += local[i * 2];
    const float A = global[i * 2], dg = global[i * 2 + 1] * is;
    const float k0 = gamma[c] * is;
    const float k1 = -k0 * is * dg * inv_m;
    coef[i * 3 + 0] = k0;
    coef[i * 3 + 1] = k1;
    coef[i * 3 + 2] = -k0 * A * inv_m - k1 * mu;
  }
  dgamma[c] = dg_tot;
  dbeta[c] = db_tot;
}

// ---------------------------------------------------------------------------------------------
// Finalize inside the consumer.  A finalize launch is 3-8 us of latency for a few KB of sums and there are ~300 of them in a
// training step; when a statistics group has at most BN_PRO_MAX_S partial rows (every layer of the two deepest ResNet stages)
// each 64-channel block of bn_apply / bn_dx re-reduces the partials of ITS channels in its prologue instead: S x 64 float2,
// all loads issued at once, the same ordered sum in every block, so all blocks of a launch normalise with bit-identical
// statistics.  The blocks with blockIdx.y == 0 write save_mean / save_invstd (dgamma / dbeta), block (y, z) == (0, 0) updates
// the running statistics group by group.  More than BN_PRO_MAX_S partial rows are first shrunk in place (bn_shrink_kernel above);
// there is no finalize kernel any more.
// ---------------------------------------------------------------------------------------------
constexpr int BN_PRO_MAX_S = 96;
constexpr int BN_PRO_PARTS = BN_THREADS / 64;
constexpr int BN_PRO_DEPTH = BN_PRO_MAX_S / BN_PRO_PARTS;
// (A, B) = sums over the S partial rows for channel 64 * ct + threadIdx.x, valid in threads 0..63; block-uniform call
__device__ __forceinline__ void bn_tile_sums(const float* __restrict__ ws, int S, int stride, int C, int ct, float& A, float& B) {
  __shared__ float red[2][BN_PRO_PARTS][64];
  const int tid = threadIdx.x, c = tid & 63, part = tid >> 6;
  const float* p = ws + ((size_t)ct * 64 + c) * 2;
  float2 v[BN_PRO_DEPTH];
#pragma unroll
  for (int u = 0; u < BN_PRO_DEPTH; ++u) {
    const int row = part + BN_PRO_PARTS * u;
    v[u] = *reinterpret_cast<const float2*>(p + (size_t)(row < S ? row : 0) * stride * C * 2);
  }
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int u = 0; u < BN_PRO_DEPTH; ++u) {
    const bool ok = part + BN_PRO_PARTS * u < S;
    a += ok ? v[u].x : 0.f;
    b += ok ? v[u].y : 0.f;
  }
  red[0][part][c] = a;
  red[1][part][c] = b;
  __syncthreads();
  A = 0.f;
  B = 0.f;
  if (tid < 64) {
#pragma unroll
    for (int j = 0; j < BN_PRO_PARTS; ++j) { A += red[0][j][c]; B += red[1][j][c]; }
  }
  __syncthreads();      // red is reused by the next call
}

struct BnFinFwd {       // forward statistics formed in bn_apply's prologue (FIN = true)
  const float* ws;      // [G, rows, C, 2] partial (sum x, sum x^2); rows 0, stride, 2 stride, ... (S of them) are summed
  int S, stride, rows, G;
  float eps, momentum;
  float* running_mean;
  float* running_var;
  float* save_mean;
  float* save_invstd;
};
struct BnFinBwd {       // backward coefficients formed in bn_dx's prologue (FIN = true)
  const float* ws;      // [G, rows, C, 2] partial (sum g, sum g (x - mean)), read like BnFinFwd::ws
  int S, stride, rows, G;
  float* dgamma;
  float* dbeta;
};

// per-channel sums of this rank for the synchronised form: the SAME ordered sum as the prologues above (so a one-rank
// synchronised BatchNorm equals the local one bit for bit), written out as [G, C, 2] for the all-reduce
__global__ __launch_bounds__(BN_THREADS) void bn_tile_reduce_kernel(const float* __restrict__ ws, int S, int stride, int rows, int C, int G,
                                                                    float* __restrict__ sums) {
  for (int grp = 0; grp < G; ++grp) {
    float A, B;
    bn_tile_sums(ws + (size_t)grp * rows * C * 2, S, stride, C, blockIdx.x, A, B);
    if (threadIdx.x < 64) {
      const size_t c = (size_t)grp * C + (size_t)blockIdx.x * 64 + threadIdx.x;
      sums[c * 2] = A;
      sums[c * 2 + 1] = B;
    }
  }
}

template <typename T, bool FIN = false>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              long long M, int C, int rows_per_split, int relu, T* __restrict__ y,
                                                              BnFinFwd fin) {
  const int tid = threadIdx.x, cx = tid & 7, ry = tid >> 3;
  const int grp = blockIdx.z;
  const long long base = (long long)grp * M;
  const long long r0 = base + (long long)blockIdx.y * rows_per_split;
  const long long r1 = (r0 + rows_per_split < base + M) ? r0 + rows_per_split : base + M;
  const size_t col = (size_t)blockIdx.x * 64 + (size_t)cx * 8;
  // the first rows are requested BEFORE the statistics prologue: its dependent chain (partials -> LDS -> mean / invstd) is a
  // 2-3 us round trip that would otherwise stand in front of every block's first load
  Raw8<T> qx[BN_UNROLL], qr[BN_UNROLL];
  bool ok[BN_UNROLL];
#define BN_APPLY_ISSUE(rr0)                                                       \
  {                                                                               \
    _Pragma("unroll") for (int u = 0; u < BN_UNROLL; ++u) {                       \
      const long long rr = (rr0) + (long long)u * BN_RY;                          \
      ok[u] = rr < r1;                                                            \
      const size_t off = (size_t)(ok[u] ? rr : (rr0)) * C + col;                  \
      qx[u].load(x + off);                                                        \
      if (res) qr[u].load(res + off);                                             \
    }                                                                             \
  }
  long long r = r0 + ry;
  if (r < r1) BN_APPLY_ISSUE(r)
  float sc[8], sh[8];
  if constexpr (FIN) {
    __shared__ float s_sc[64], s_sh[64];
    const int c = blockIdx.x * 64 + (tid & 63);
    float A, B;
    bn_tile_sums(fin.ws + (size_t)grp * fin.rows * C * 2, fin.S, fin.stride, C, blockIdx.x, A, B);
    if (tid < 64) {
      const double m = (double)A / (double)M;
      double var = (double)B / (double)M - m * m;      // biased variance, formed in double
      var = var > 0.0 ? var : 0.0;
      const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)fin.eps));
      const float scv = gamma[c] * isf;
      s_sc[tid] = scv;
      s_sh[tid] = beta[c] - mf * scv;
      if (blockIdx.y == 0) {
        fin.save_mean[(size_t)grp * C + c] = mf;
        fin.save_invstd[(size_t)grp * C + c] = isf;
      }
    }
    if (fin.running_mean && blockIdx.y == 0 && blockIdx.z == 0) {
      // groups in order: one momentum update per group, exactly like G separate calls
      double rm = 0.0, rv = 0.0;
      if (tid < 64) { rm = fin.running_mean[c]; rv = fin.running_var[c]; }
      for (int g2 = 0; g2 < fin.G; ++g2) {
        float A2 = A, B2 = B;
        if (g2 != 0) bn_tile_sums(fin.ws + (size_t)g2 * fin.rows * C * 2, fin.S, fin.stride, C, blockIdx.x, A2, B2);
        if (tid < 64) {
          const double m = (double)A2 / (double)M;
          double var = (double)B2 / (double)M - m * m;
          var = var > 0.0 ? var : 0.0;
          const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
          rm = (double)(float)((1.0 - fin.momentum) * rm + fin.momentum * m);
          rv = (double)(float)((1.0 - fin.momentum) * rv + fin.momentum * unbiased);
        }
      }
      if (tid < 64) { fin.running_mean[c] = (float)rm; fin.running_var[c] = (float)rv; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = s_sc[cx * 8 + i]; sh[i] = s_sh[cx * 8 + i]; }
  } else {
    mean += (size_t)grp * C;
    invstd += (size_t)grp * C;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sc[i] = gamma[col + i] * invstd[col + i];
      sh[i] = beta[col + i] - mean[col + i] * sc[i];
    }
  }
  for (; r < r1; r += BN_UNROLL * BN_RY) {
    float vx[BN_UNROLL][8], vr[BN_UNROLL][8];
    bool okc[BN_UNROLL];
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      qx[u].unpack(vx[u]);
      if (res) qr[u].unpack(vr[u]);
      okc[u] = ok[u];
    }
    const long long rn = r + BN_UNROLL * BN_RY;           // the next rows are on their way while these are written
    if (rn < r1) BN_APPLY_ISSUE(rn)
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float v = fmaf(vx[u][i], sc[i], sh[i]);
        if (res) v += vr[u][i];
        if (relu) v = v < 0.f ? 0.f : v;       // NaN propagates, like ATen's relu
        o[i] = v;
      }
      if (okc[u]) store8(y + (size_t)(r + (long long)u * BN_RY) * C + col, o);
    }
  }
#undef BN_APPLY_ISSUE
}

template <typename T, bool FIN = false>
__global__ __launch_bounds__(BN_THREADS) void bn_dx_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                           const float* __restrict__ coef, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, long long M, int C, int rows_per_split,
                                                           int relu, T* __restrict__ dx, T* __restrict__ dres, BnFinBwd fin) {
  const int tid = threadIdx.x, cx = tid & 7, ry = tid >> 3;
  const int grp = blockIdx.z;
  const long long base = (long long)grp * M;
  const long long r0 = base + (long long)blockIdx.y * rows_per_split;
  const long long r1 = (r0 + rows_per_split < base + M) ? r0 + rows_per_split : base + M;
  const size_t col = (size_t)blockIdx.x * 64 + (size_t)cx * 8;
  // first rows requested before the coefficient prologue (see bn_apply_kernel)
  Raw8<T> qx[BN_UNROLL], qg[BN_UNROLL], qy[BN_UNROLL];
  bool ok[BN_UNROLL];
#define BN_DX_ISSUE(rr0)                                                          \
  {                                                                               \
    _Pragma("unroll") for (int u = 0; u < BN_UNROLL; ++u) {                       \
      const long long rr = (rr0) + (long long)u * BN_RY;                          \
      ok[u] = rr < r1;                                                            \
      const size_t off = (size_t)(ok[u] ? rr : (rr0)) * C + col;                  \
      qx[u].load(x + off);                                                        \
      qg[u].load(dy + off);                                                       \
      if (relu == 1) qy[u].load(y + off);                                         \
    }                                                                             \
  }
  long long r = r0 + ry;
  if (r < r1) BN_DX_ISSUE(r)
  float k0[8], k1[8], k2[8];
  if constexpr (FIN) {
    // dgamma, dbeta (summed over the statistics groups) and this group's dx coefficients  dx = k0 * g + k1 * x + k2
    __shared__ float s_k[3][64];
    const int c = blockIdx.x * 64 + (tid & 63);
    const float inv_m = 1.f / (float)M;
    float A, B;
    bn_tile_sums(fin.ws + (size_t)grp * fin.rows * C * 2, fin.S, fin.stride, C, blockIdx.x, A, B);
    if (tid < 64) {
      const float is = invstd[(size_t)grp * C + c], mu = mean[(size_t)grp * C + c];
      const float dg = B * is;                     // sum g * xhat
      const float c0 = gamma[c] * is;
      const float c1 = -c0 * is * dg * inv_m;      // multiplies (x - mean)
      s_k[0][tid] = c0;
      s_k[1][tid] = c1;
      s_k[2][tid] = -c0 * A * inv_m - c1 * mu;
    }
    if (blockIdx.y == 0 && blockIdx.z == 0) {
      float dg_tot = 0.f, db_tot = 0.f;
      for (int g2 = 0; g2 < fin.G; ++g2) {
        float A2 = A, B2 = B;
        if (g2 != 0) bn_tile_sums(fin.ws + (size_t)g2 * fin.rows * C * 2, fin.S, fin.stride, C, blockIdx.x, A2, B2);
        if (tid < 64) {
          dg_tot += B2 * invstd[(size_t)g2 * C + c];
          db_tot += A2;
        }
      }
      if (tid < 64) { fin.dgamma[c] = dg_tot; fin.dbeta[c] = db_tot; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) { k0[i] = s_k[0][cx * 8 + i]; k1[i] = s_k[1][cx * 8 + i]; k2[i] = s_k[2][cx * 8 + i]; }
  } else {
    coef += (size_t)grp * C * 3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      k0[i] = coef[(col + i) * 3 + 0];
      k1[i] = coef[(col + i) * 3 + 1];
      k2[i] = coef[(col + i) * 3 + 2];
    }
  }
  mean += (size_t)grp * C;
  invstd += (size_t)grp * C;
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { sc[i] = 0.f; sh[i] = 0.f; }
  if (relu == 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sc[i] = gamma[col + i] * invstd[col + i];
      sh[i] = beta[col + i] - mean[col + i] * sc[i];
    }
  }
  for (; r < r1; r += BN_UNROLL * BN_RY) {
    float vx[BN_UNROLL][8], vg[BN_UNROLL][8], vy[BN_UNROLL][8];
    bool okc[BN_UNROLL];
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      qx[u].unpack(vx[u]);
      qg[u].unpack(vg[u]);
      if (relu == 1) qy[u].unpack(vy[u]);
      okc[u] = ok[u];
    }
    const long long rn = r + BN_UNROLL * BN_RY;
    if (rn < r1) BN_DX_ISSUE(rn)
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      float o[8], g[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        g[i] = vg[u][i];
        if (relu == 1) g[i] = vy[u][i] <= 0.f ? 0.f : g[i];
        if (relu == 2) g[i] = fmaf(vx[u][i], sc[i], sh[i]) <= 0.f ? 0.f : g[i];
        o[i] = fmaf(k0[i], g[i], fmaf(k1[i], vx[u][i], k2[i]));
      }
      if (okc[u]) {
        const size_t off = (size_t)(r + (long long)u * BN_RY) * C + col;
        store8(dx + off, o);
        if (dres) store8(dres + off, g);
      }
    }
  }
#undef BN_DX_ISSUE
}

// workgroups per statistics group of a pass (bn_partials, bn_apply, bn_dx): 2048 / 1024 / 512 / 256 were measured on the training
// step (35.7 / 35.3 / 34.9 / 35.2 ms).  Per GROUP, so a grouped call cuts its rows exactly like separate calls would (bit-equal).
constexpr int BN_STAT_BLOCKS = 512;
static inline int bn_splits(long long M, int C, int target_blocks, int cap) {
  const int groups = C / 64;
  long long s = (target_blocks + groups - 1) / groups;
  const long long by_rows = (M + 63) / 64;          // at least 64 rows per block
  if (s > by_rows) s = by_rows;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  return (int)s;
}
static inline int bn_rows_per_split(long long M, int S) {
  const long long r = (M + S - 1) / S;
  return (int)((r + BN_RY - 1) / BN_RY * BN_RY);
}

// the partial rows a consumer prologue sums: all S of them, or -- after bn_shrink_kernel -- every Sq-th
BnRows bn_shrink_partials_to(float* partials, int S, int G, int C, int max_rows, hipStream_t st) {
  if (S <= max_rows) return {S, 1};
  const int Sq = (S + max_rows - 1) / max_rows, Q = (S + Sq - 1) / Sq;
  hipLaunchKernelGGL(bn_shrink_kernel, dim3(C / BN_FIN_CH, Q, G), dim3(BN_FIN_THREADS), 0, st, partials, S, C, Sq);
  return {Q, Sq};
}
static BnRows shrink_partials(float* partials, int S, int G, int C, hipStream_t st) {
  return bn_shrink_partials_to(partials, S, G, C, BN_PRO_MAX_S, st);
}

// statistics (in the apply kernel's prologue) + apply, from partial sums; `partials` is reduced in place when S > BN_PRO_MAX_S
template <typename T>
static void launch_bn_finalize_apply(const void* x, const void* res, const float* gamma, const float* beta, float* rmean, float* rvar,
                                     float momentum, float eps, int relu, long long Mg, int G, int C, float* partials, int S, void* y,
                                     float* save_mean, float* save_invstd, hipStream_t st) {
  const BnRows rows = shrink_partials(partials, S, G, C, st);
  const int S2 = bn_splits(Mg, C, BN_STAT_BLOCKS, 4096), rps2 = bn_rows_per_split(Mg, S2);
  const dim3 grid(C / 64, (unsigned)((Mg + rps2 - 1) / rps2), G);
  const BnFinFwd fin = {partials, rows.n, rows.stride, S, G, eps, momentum, rmean, rvar, save_mean, save_invstd};
  hipLaunchKernelGGL((bn_apply_kernel<T, true>), grid, dim3(BN_THREADS), 0, st, (const T*)x, (const T*)res, gamma, beta,
                     (const float*)nullptr, (const float*)nullptr, Mg, C, rps2, relu, (T*)y, fin);
}

// row splits of a statistics pass whose partial rows a GEMM prologue finishes (td_conv1x1_fwd_bnrelu / td_conv1x1_dgrad_bnbwd: at
// most 16 rows per block there): wide, short layers are cut into 16 ranges directly (no shrink launch); the others as every pass
static inline int bn_stat_splits(long long Mg, int C) {
  const int S = bn_splits(Mg, C, BN_STAT_BLOCKS, 512);
  if (S > 16 && (C / 64) * 16 >= 128) return bn_splits(Mg, C, (C / 64) * 16, 16);
  return S;
}
static inline int bn_stat_rows(long long Mg, int C) {
  const int S = bn_stat_splits(Mg, C), rps = bn_rows_per_split(Mg, S);
  return (int)((Mg + rps - 1) / rps);
}

template <typename T>
static int run_bn_partials(int mode, const void* x, const void* dy, const void* y, const float* gamma, const float* beta, const float* mean,
                           const float* invstd, int relu, long long M, int G, int C, float* ws, hipStream_t st) {
  const long long Mg = M / G;
  const int S = bn_stat_splits(Mg, C), rps = bn_rows_per_split(Mg, S);
  const int S_eff = (int)((Mg + rps - 1) / rps);
  if (mode == 0)
    hipLaunchKernelGGL((bn_partials_kernel<T, 0>), dim3(C / 64, S_eff, G), dim3(BN_THREADS), 0, st, (const T*)x, (const T*)nullptr,
                       (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, Mg, C,
                       rps, 0, ws);
  else
    hipLaunchKernelGGL((bn_partials_kernel<T, 1>), dim3(C / 64, S_eff, G), dim3(BN_THREADS), 0, st, (const T*)x, (const T*)dy,
                       (const T*)y, mean, invstd, gamma, beta, Mg, C, rps, relu, ws);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

// dx (+ dgamma, dbeta) from backward partial sums formed elsewhere (a GEMM epilogue): shrink when needed, finish in bn_dx's prologue
template <typename T>
static int run_bn_dx_from_partials(const void* dy, const void* x, const void* y, const float* gamma, const float* beta, const float* mean,
                                   const float* invstd, int relu, long long M, int G, int C, float* partials, int S, void* dx, void* dres,
                                   float* dgamma, float* dbeta, hipStream_t st) {
  const long long Mg = M / G;
  const BnRows rows = shrink_partials(partials, S, G, C, st);
  const int S2 = bn_splits(Mg, C, BN_STAT_BLOCKS, 4096), rps2 = bn_rows_per_split(Mg, S2);
  const dim3 grid(C / 64, (unsigned)((Mg + rps2 - 1) / rps2), G);
  const BnFinBwd fin = {partials, rows.n, rows.stride, S, G, dgamma, dbeta};
  hipLaunchKernelGGL((bn_dx_kernel<T, true>), grid, dim3(BN_THREADS), 0, st, (const T*)dy, (const T*)x, (const T*)y, (const float*)nullptr,
                     mean, invstd, gamma, beta, Mg, C, rps2, relu, (T*)dx, (T*)dres, fin);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

template <typename T>
static int run_bn_fwd(const void* x, const void* res, const float* gamma, const float* beta, float* rmean, float* rvar,
                      float momentum, float eps, int relu, long long M, int G, int C, void* y, float* save_mean, float* save_invstd,
                      float* ws, hipStream_t st) {
  const long long Mg = M / G;                      // rows per statistics group
  const int S = bn_splits(Mg, C, BN_STAT_BLOCKS, 512), rps = bn_rows_per_split(Mg, S);
  const int S_eff = (int)((Mg + rps - 1) / rps);
  hipLaunchKernelGGL((bn_partials_kernel<T, 0>), dim3(C / 64, S_eff, G), dim3(BN_THREADS), 0, st, (const T*)x, (const T*)nullptr,
                     (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, Mg, C, rps, 0, ws);
  launch_bn_finalize_apply<T>(x, res, gamma, beta, rmean, rvar, momentum, eps, relu, Mg, G, C, ws, S_eff, y, save_mean, save_invstd, st);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

template <typename T>
static int run_bn_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* beta, const float* mean, const float* invstd,
                      int relu, long long M, int G, int C, void* dx, void* dres, float* dgamma, float* dbeta, float* ws, hipStream_t st) {
  const long long Mg = M / G;
  const int S = bn_splits(Mg, C, BN_STAT_BLOCKS, 512), rps = bn_rows_per_split(Mg, S);
  const int S_eff = (int)((Mg + rps - 1) / rps);
  hipLaunchKernelGGL((bn_partials_kernel<T, 1>), dim3(C / 64, S_eff, G), dim3(BN_THREADS), 0, st, (const T*)x, (const T*)dy,
                     (const T*)y, mean, invstd, gamma, beta, Mg, C, rps, relu, ws);
  const BnRows rows = shrink_partials(ws, S_eff, G, C, st);
  const int S2 = bn_splits(Mg, C, BN_STAT_BLOCKS, 4096), rps2 = bn_rows_per_split(Mg, S2);
  const dim3 grid(C / 64, (unsigned)((Mg + rps2 - 1) / rps2), G);
  const BnFinBwd fin = {ws, rows.n, rows.stride, S_eff, G, dgamma, dbeta};
  hipLaunchKernelGGL((bn_dx_kernel<T, true>), grid, dim3(BN_THREADS), 0, st, (const T*)dy, (const T*)x, (const T*)y, (const float*)nullptr,
                     mean, invstd, gamma, beta, Mg, C, rps2, relu, (T*)dx, (T*)dres, fin);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

// ---- staged entry points for synchronised statistics ----
template <typename T>
static int run_bn_local_sums(int mode, const void* x, const void* dy, const void* y, const float* gamma, const float* beta,
                             const float* mean, const float* invstd, int relu, long long M, int G, int C, float* sums, float* ws,
                             hipStream_t st) {
  const long long Mg = M / G;
  const int S = bn_splits(Mg, C, BN_STAT_BLOCKS, 512), rps = bn_rows_per_split(Mg, S);
  const int S_eff = (int)((Mg + rps - 1) / rps);
  if (mode == 0)
    hipLaunchKernelGGL((bn_partials_kernel<T, 0>), dim3(C / 64, S_eff, G), dim3(BN_THREADS), 0, st, (const T*)x, (const T*)nullptr,
                       (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, Mg, C,
                       rps, 0, ws);
  else
    hipLaunchKernelGGL((bn_partials_kernel<T, 1>), dim3(C / 64, S_eff, G), dim3(BN_THREADS), 0, st, (const T*)x, (const T*)dy,
                       (const T*)y, mean, invstd, gamma, beta, Mg, C, rps, relu, ws);
  const BnRows rows = shrink_partials(ws, S_eff, G, C, st);
  hipLaunchKernelGGL(bn_tile_reduce_kernel, dim3(C / 64), dim3(BN_THREADS), 0, st, (const float*)ws, rows.n, rows.stride, S_eff, C, G, sums);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

template <typename T>
static int run_bn_sync_apply(const void* x, const void* res, const float* sums, const float* count, const float* gamma, const float* beta,
                             float* rmean, float* rvar, float momentum, float eps, int relu, long long M, int G, int C, void* y,
                             float* save_mean, float* save_invstd, hipStream_t st) {
  const long long Mg = M / G;
  hipLaunchKernelGGL(bn_stats_from_sums_kernel, dim3(C / 64), dim3(64), 0, st, sums, count, C, G, eps, momentum, rmean, rvar, save_mean,
                     save_invstd);
  const int S2 = bn_splits(Mg, C, 2048 / G, 4096), rps2 = bn_rows_per_split(Mg, S2);
  hipLaunchKernelGGL((bn_apply_kernel<T, false>), dim3(C / 64, (unsigned)((Mg + rps2 - 1) / rps2), G), dim3(BN_THREADS), 0, st, (const T*)x,
                     (const T*)res, gamma, beta, (const float*)save_mean, (const float*)save_invstd, Mg, C, rps2, relu, (T*)y, BnFinFwd{});
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

template <typename T>
static int run_bn_sync_dx(const void* dy, const void* x, const void* y, const float* local, const float* global, const float* count,
                          const float* gamma, const float* beta, const float* mean, const float* invstd, int relu, long long M, int G,
                          int C, void* dx, void* dres, float* dgamma, float* dbeta, float* coef, hipStream_t st) {
  const long long Mg = M / G;
  hipLaunchKernelGGL(bn_coef_from_sums_kernel, dim3(C / 64), dim3(64), 0, st, local, global, count, C, G, gamma, mean, invstd, dgamma, dbeta,
                     coef);
  const int S2 = bn_splits(Mg, C, 2048 / G, 4096), rps2 = bn_rows_per_split(Mg, S2);
  hipLaunchKernelGGL((bn_dx_kernel<T, false>), dim3(C / 64, (unsigned)((Mg + rps2 - 1) / rps2), G), dim3(BN_THREADS), 0, st, (const T*)dy,
                     (const T*)x, (const T*)y, (const float*)coef, mean, invstd, gamma, beta, Mg, C, rps2, relu, (T*)dx, (T*)dres, BnFinBwd{});
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

}  // namespace td

static bool bn_shape_ok(long long M, int G, int C) { return G >= 1 && G <= 64 && M > 0 && C > 0 && M % G == 0; }

extern "C" long long td_bn_workspace_floats(long long M, int groups, int C) {
  if (!bn_shape_ok(M, groups, C) || C % 64 != 0) return 0;
  const int S = td::bn_splits(M / groups, C, td::BN_STAT_BLOCKS, 512);
  return ((long long)2 * S * C + (long long)3 * C) * groups;
}

extern "C" int td_bn_fwd(const void* x, const void* residual, int dtype, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, int relu, long long M, int groups,
                         int C, void* y, float* save_mean, float* save_invstd, float* workspace, td_stream_t stream) {
  if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace || !bn_shape_ok(M, groups, C)) return TD_ERR_BAD_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return TD_ERR_BAD_ARG;
  if (C % 64 != 0 || M * (long long)C >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16)
    return td::run_bn_fwd<__hip_bfloat16>(x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu, M, groups, C, y,
                                          save_mean, save_invstd, workspace, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_bn_fwd<float>(x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu, M, groups, C, y, save_mean,
                                 save_invstd, workspace, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_bn_bwd(const void* dy, const void* x, const void* y, int dtype, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_invstd, int relu, long long M, int groups, int C, void* dx, void* dresidual,
                         float* dgamma, float* dbeta, float* workspace, td_stream_t stream) {
  if (!dy || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace || !bn_shape_ok(M, groups, C))
    return TD_ERR_BAD_ARG;
  if (relu < 0 || relu > 1) return TD_ERR_BAD_ARG;
  if (relu && !y && (!beta || dresidual)) return TD_ERR_BAD_ARG;   // the mask comes from y, or is recomputed from x (no residual)
  if (relu && !y) relu = 2;
  if (C % 64 != 0 || M * (long long)C >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16)
    return td::run_bn_bwd<__hip_bfloat16>(dy, x, y, gamma, beta, save_mean, save_invstd, relu, M, groups, C, dx, dresidual, dgamma, dbeta,
                                          workspace, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_bn_bwd<float>(dy, x, y, gamma, beta, save_mean, save_invstd, relu, M, groups, C, dx, dresidual, dgamma, dbeta, workspace,
                                 (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

// Forward with the partial sums already formed by the producing kernel (td_conv1x1_fwd's epilogue): the statistics are finished
// in the apply kernel's prologue (one launch when stat_rows <= 96; above that a shrink launch first reduces `partials` IN PLACE)
// and there is no statistics pass over x.  partials: [groups, stat_rows, C, 2] f32, scratch after the call.
extern "C" int td_bn_fwd_from_partials(const void* x, const void* residual, int dtype, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, float momentum, float eps, int relu, long long M,
                                       int groups, int C, float* partials, int stat_rows, void* y, float* save_mean,
                                       float* save_invstd, td_stream_t stream) {
  if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !partials || stat_rows < 1 || !bn_shape_ok(M, groups, C))
    return TD_ERR_BAD_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return TD_ERR_BAD_ARG;
  if (C % 64 != 0 || M * (long long)C >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  if (dtype != TD_DTYPE_BF16 && dtype != TD_DTYPE_F32) return TD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long long Mg = M / groups;
  if (dtype == TD_DTYPE_BF16)
    td::launch_bn_finalize_apply<__hip_bfloat16>(x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu, Mg, groups, C, partials,
                                                 stat_rows, y, save_mean, save_invstd, st);
  else
    td::launch_bn_finalize_apply<float>(x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu, Mg, groups, C, partials, stat_rows,
                                        y, save_mean, save_invstd, st);
  return hipGetLastError() == hipSuccess ? TD_OK : TD_ERR_LAUNCH;
}

// ---- the statistics passes on their own (their partial rows are finished by a GEMM prologue or by td_bn_bwd_from_partials) ----
extern "C" int td_bn_partial_rows(long long M, int groups, int C) {
  if (!bn_shape_ok(M, groups, C) || C % 64 != 0) return 0;
  return td::bn_stat_rows(M / groups, C);
}

extern "C" int td_bn_fwd_partials(const void* x, int dtype, long long M, int groups, int C, float* partials, td_stream_t stream) {
  if (!x || !partials || !bn_shape_ok(M, groups, C)) return TD_ERR_BAD_ARG;
  if (C % 64 != 0 || M * (long long)C >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16)
    return td::run_bn_partials<__hip_bfloat16>(0, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, M, groups, C, partials, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_bn_partials<float>(0, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, M, groups, C, partials, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_bn_bwd_partials(const void* dy, const void* x, const void* y, int dtype, const float* gamma, const float* beta,
                                  const float* save_mean, const float* save_invstd, int relu, long long M, int groups, int C,
                                  float* partials, td_stream_t stream) {
  if (!dy || !x || !gamma || !save_mean || !save_invstd || !partials || !bn_shape_ok(M, groups, C)) return TD_ERR_BAD_ARG;
  if (relu < 0 || relu > 1 || (relu && !y && !beta)) return TD_ERR_BAD_ARG;
  if (relu && !y) relu = 2;
  if (C % 64 != 0 || M * (long long)C >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16)
    return td::run_bn_partials<__hip_bfloat16>(1, x, dy, y, gamma, beta, save_mean, save_invstd, relu, M, groups, C, partials, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_bn_partials<float>(1, x, dy, y, gamma, beta, save_mean, save_invstd, relu, M, groups, C, partials, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_bn_bwd_from_partials(const void* dy, const void* x, const void* y, int dtype, const float* gamma, const float* beta,
                                       const float* save_mean, const float* save_invstd, int relu, long long M, int groups, int C,
                                       float* partials, int stat_rows, void* dx, void* dresidual, float* dgamma, float* dbeta,
                                       td_stream_t stream) {
  if (!dy || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !partials || stat_rows < 1 || !bn_shape_ok(M, groups, C))
    return TD_ERR_BAD_ARG;
  if (relu < 0 || relu > 1) return TD_ERR_BAD_ARG;
  if (relu && !y && (!beta || dresidual)) return TD_ERR_BAD_ARG;
  if (relu && !y) relu = 2;
  if (C % 64 != 0 || M * (long long)C >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16)
    return td::run_bn_dx_from_partials<__hip_bfloat16>(dy, x, y, gamma, beta, save_mean, save_invstd, relu, M, groups, C, partials, stat_rows,
                                                       dx, dresidual, dgamma, dbeta, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_bn_dx_from_partials<float>(dy, x, y, gamma, beta, save_mean, save_invstd, relu, M, groups, C, partials, stat_rows, dx,
                                              dresidual, dgamma, dbeta, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

// ---- synchronised (cross-rank) statistics: local sums -> [caller: all-reduce] -> apply / dx --------------------

extern "C" int td_bn_sync_fwd_sums(const void* x, int dtype, long long M, int groups, int C, float* sums, float* workspace,
                                   td_stream_t stream) {
  if (!x || !sums || !workspace || !bn_shape_ok(M, groups, C)) return TD_ERR_BAD_ARG;
  if (C % 64 != 0 || M * (long long)C >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16)
    return td::run_bn_local_sums<__hip_bfloat16>(0, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, M, groups, C, sums, workspace,
                                                 (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_bn_local_sums<float>(0, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, M, groups, C, sums, workspace,
                                        (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_bn_sync_fwd_apply(const void* x, const void* residual, int dtype, const float* sums, const float* count,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                    float eps, int relu, long long M, int groups, int C, void* y, float* save_mean, float* save_invstd,
                                    td_stream_t stream) {
  if (!x || !sums || !count || !gamma || !beta || !y || !save_mean || !save_invstd || !bn_shape_ok(M, groups, C)) return TD_ERR_BAD_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return TD_ERR_BAD_ARG;
  if (C % 64 != 0 || M * (long long)C >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16)
    return td::run_bn_sync_apply<__hip_bfloat16>(x, residual, sums, count, gamma, beta, running_mean, running_var, momentum, eps, relu, M,
                                                 groups, C, y, save_mean, save_invstd, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_bn_sync_apply<float>(x, residual, sums, count, gamma, beta, running_mean, running_var, momentum, eps, relu, M, groups, C,
                                        y, save_mean, save_invstd, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_bn_sync_bwd_sums(const void* dy, const void* x, const void* y, int dtype, const float* gamma, const float* beta,
                                   const float* save_mean, const float* save_invstd, int relu, long long M, int groups, int C,
                                   float* sums, float* workspace, td_stream_t stream) {
  if (!dy || !x || !gamma || !save_mean || !save_invstd || !sums || !workspace || !bn_shape_ok(M, groups, C)) return TD_ERR_BAD_ARG;
  if (relu < 0 || relu > 1 || (relu && !y && !beta)) return TD_ERR_BAD_ARG;
  if (relu && !y) relu = 2;
  if (C % 64 != 0 || M * (long long)C >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16)
    return td::run_bn_local_sums<__hip_bfloat16>(1, x, dy, y, gamma, beta, save_mean, save_invstd, relu, M, groups, C, sums, workspace,
                                                 (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_bn_local_sums<float>(1, x, dy, y, gamma, beta, save_mean, save_invstd, relu, M, groups, C, sums, workspace,
                                        (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}

extern "C" int td_bn_sync_bwd_dx(const void* dy, const void* x, const void* y, int dtype, const float* local_sums, const float* global_sums,
                                 const float* count, const float* gamma, const float* beta, const float* save_mean,
                                 const float* save_invstd, int relu, long long M, int groups, int C, void* dx, void* dresidual,
                                 float* dgamma, float* dbeta, float* coef, td_stream_t stream) {
  if (!dy || !x || !local_sums || !global_sums || !count || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !coef ||
      !bn_shape_ok(M, groups, C))
    return TD_ERR_BAD_ARG;
  if (relu < 0 || relu > 1 || (relu && !y && (!beta || dresidual))) return TD_ERR_BAD_ARG;
  if (relu && !y) relu = 2;
  if (C % 64 != 0 || M * (long long)C >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16)
    return td::run_bn_sync_dx<__hip_bfloat16>(dy, x, y, local_sums, global_sums, count, gamma, beta, save_mean, save_invstd, relu, M, groups,
                                              C, dx, dresidual, dgamma, dbeta, coef, (hipStream_t)stream);
  if (dtype == TD_DTYPE_F32)
    return td::run_bn_sync_dx<float>(dy, x, y, local_sums, global_sums, count, gamma, beta, save_mean, save_invstd, relu, M, groups, C, dx,
                                     dresidual, dgamma, dbeta, coef, (hipStream_t)stream);
  return TD_ERR_UNSUPPORTED;
}
