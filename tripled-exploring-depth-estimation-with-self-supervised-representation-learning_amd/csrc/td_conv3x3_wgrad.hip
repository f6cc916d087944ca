// Weight gradient of the 3x3 stride-1 convolutions (zero padding 1: the ResNet blocks, reference
// mono/model/mono_fm_joint/resnet.py:30-49, 57-58; padding 0 on a pre-padded input: the decoders' Conv3x3 = ReflectionPad2d(1) +
// conv, mono/model/mono_fm_joint/layers.py:171-184) on channels-last bf16 activations:
//
//     dW[n, ky, kx, c] = sum over output pixels (b, ho, wo) of  dY[b, ho, wo, n] * X[b, ho + ky - pad, wo + kx - pad, c]
//
// i.e. nine weight gradients of 1x1 convolutions that share dY and read X shifted by a tap.  The kernel is the 1x1 weight
// gradient (td_conv1x1.hip: reduction over the pixel index, operands through ds_read_b64_tr_b16, M cut into row ranges whose
// fp32 slabs a second kernel adds in a fixed order -- deterministic, and none of the zero-fill / cast passes MIOpen's split-K
// kernels run around themselves) with NINE accumulator tiles per wave and the X operand staged ONCE per 64-pixel stage for all
// taps: three bands (one per ky) of the input pixels the stage touches, laid out in "padded row coordinates" so that a tap is a
// constant row offset:
//
//     output pixel d of the stage (row r0 + d, image-row crossings k = (wo0 + d) / Wo)  ->  band position d + G k + kx,
//     G = 2 - pad: pad 0: the two extra input columns per image row; pad 1: ONE zero slot between image rows that serves as the
//     right padding of one row and the left padding of the next.  Rows above / below the image are zero-filled while the band is
//     loaded.  No masks in the MFMA loop.
//
// Block = 512 threads = 8 waves on a 64 (n) x 64 (c) tile: 2 x 2 wave tiles of 32 x 32, each held by TWO waves that split the nine
// taps 5 + 4 (nine accumulators in one wave are 144 registers: with the staging registers that spilled; 80 leave room for two
// blocks per CU at 4 waves per SIMD).  Per stage 36 v_mfma_f32_32x32x16_bf16 per wave pair against one staging round (dY tile 8 KB
// + bands <= 28 KB, from L2: neighbouring (n, c) tiles of a row range sit on one XCD); the next stage's operands travel in
// registers while the current one is multiplied (single LDS buffer, two barriers per stage).
#include <hip/hip_bf16.h>

#include "td_conv_tile.h"

namespace td {

constexpr int W3_T = 64;                    // tile edge and pixels per stage
constexpr int W3_PITCH = W3_T * 2 + 64;     // bytes per LDS row (192: conflict-free transposed reads, see td_conv1x1.hip)
constexpr int W3_SPAN = 74;                 // band positions per stage: 64 + 2 + G * (row crossings <= 4 for Wo >= 20)
constexpr int W3_DY_BYTES = W3_T * W3_PITCH;
constexpr int W3_BAND_BYTES = W3_SPAN * W3_PITCH;
constexpr int W3_XCHUNKS = 3 * W3_SPAN * 8;                       // 16-byte chunks of the three bands
constexpr int W3_THREADS = 512;
static_assert(W3_XCHUNKS <= 4 * W3_THREADS, "four band chunks per thread");

typedef __attribute__((ext_vector_type(4))) short w3_s16x4;

// v or zeros, component by component (a ?: between two uint4 objects is lowered through scratch memory)
__device__ __forceinline__ uint4 w3_keep(bool ok, uint4 v) { return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u); }

__device__ __forceinline__ cv_bf16x8 w3_frag(const unsigned char* p_lo, const unsigned char* p_hi) {
  typedef __attribute__((address_space(3))) w3_s16x4* lds_ptr;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const w3_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p_lo));
  const w3_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p_hi));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(cv_bf16x8, v);
}

template <int PAD>
__global__ __launch_bounds__(W3_THREADS, 2) void conv3x3_wgrad_kernel(
    const __hip_bfloat16* __restrict__ dy, const __hip_bfloat16* __restrict__ x, float* __restrict__ part, long long M, int Ho, int Wo,
    int C, int N, int rows_per_split, int tiles_n, int tiles_c) {
  constexpr int G = 2 - PAD;
  __shared__ __attribute__((aligned(16))) unsigned char lds[W3_DY_BYTES + 3 * W3_BAND_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = (wid >> 1) & 1, wc = wid & 1, t_lo = (wid >> 2) * 5;      // this wave's taps: t_lo .. min(t_lo + 5, 9)
  const int Hi = Ho + 2 - 2 * PAD, Wi = Wo + 2 - 2 * PAD, L = Wo + G;
  // all (n, c) tiles of one row range read the same rows: keep them on one XCD (td_conv1x1.hip)
  const int Lb = cv_xcd_tile(blockIdx.x, gridDim.x);
  const int nt = Lb % tiles_n, ct = (Lb / tiles_n) % tiles_c, sp = Lb / (tiles_n * tiles_c);
  const int n0 = nt * W3_T, c0 = ct * W3_T;
  // pixel indices fit 32 bits (checked by the entry point); only the final element offsets are 64-bit
  const int Mi = (int)M;
  const int m_lo = sp * rows_per_split;
  const int m_hi = (m_lo + rows_per_split < Mi) ? m_lo + rows_per_split : Mi;
  const int nstage = (m_hi - m_lo + W3_T - 1) / W3_T, last = nstage - 1;
  const int HoWo = Ho * Wo;

  // ---- staging registers (named: see td_conv1x1.hip); two sets: the loads run TWO stages ahead of the arithmetic (a stage's
  // MFMAs take ~0.3 us, a load round trip 1.5-2.5 us, and the 166 registers leave one block per CU to hide it) ----
  const int lc = tid & 7, lr = tid >> 3;
  // this thread's (up to) four band chunks: entry e = tid + 512 i -> (ky, position j, 16-byte chunk), fixed for the whole kernel
  auto load_x = [&](int e, int b0, int ho0, int wo0) -> uint4 {
    if (e >= W3_XCHUNKS) return make_uint4(0, 0, 0, 0);
    const int ky = e / (W3_SPAN * 8), rem = e - ky * (W3_SPAN * 8), j = rem >> 3, ch = rem & 7;
    const int Pj = wo0 + j;
    // k = Pj / L by comparison (Pj < Wo + 74, L >= 21: k <= 4)
    const int k = (Pj >= L) + (Pj >= 2 * L) + (Pj >= 3 * L) + (Pj >= 4 * L);
    const int v = Pj - k * L;
    // output image row ho0 + k of image b0 (carried over image boundaries)
    int ho = ho0 + k, b = b0;
    if (ho >= Ho) { ho -= Ho; b += 1; }
    if (ho >= Ho) { ho -= Ho; b += 1; }
    const int hi = ho + ky - PAD, wi = v - PAD;         // PAD 1: v == 0 is the zero slot (wi = -1)
    const bool ok = hi >= 0 && hi < Hi && wi >= 0 && wi < Wi && (b * HoWo + ho * Wo) < Mi;
    const int q = ok ? (b * Hi + hi) * Wi + wi : 0;
    const uint4 val = *reinterpret_cast<const uint4*>(x + (long long)q * C + c0 + ch * 8);
    return w3_keep(ok, val);
  };
#define W3_DECL(S) uint4 ry0##S, rx0##S, rx1##S, rx2##S, rx3##S;
  W3_DECL(A)
  W3_DECL(B)
#define W3_LOAD(st, S)                                                                                               \
  {                                                                                                                  \
    const int r0_ = m_lo + (st) * W3_T;                                                                              \
    const int ra_ = r0_ + lr;                                                                                        \
    ry0##S = *reinterpret_cast<const uint4*>(dy + (long long)(ra_ < Mi ? ra_ : Mi - 1) * N + n0 + lc * 8);           \
    ry0##S = w3_keep(ra_ < m_hi, ry0##S);                                                                            \
    const int rq_ = r0_ < Mi ? r0_ : Mi - 1;                                                                         \
    const int b0_ = rq_ / HoWo, rem_ = rq_ - b0_ * HoWo, ho0_ = rem_ / Wo, wo0_ = rem_ - ho0_ * Wo;                  \
    rx0##S = load_x(tid, b0_, ho0_, wo0_);                                                                           \
    rx1##S = load_x(tid + 512, b0_, ho0_, wo0_);                                                                     \
    rx2##S = load_x(tid + 1024, b0_, ho0_, wo0_);                                                                    \
    rx3##S = load_x(tid + 1536, b0_, ho0_, wo0_);                                                                    \
  }
  auto store_x = [&](int e, uint4 v) {
    if (e < W3_XCHUNKS) {
      const int ky = e / (W3_SPAN * 8), rem = e - ky * (W3_SPAN * 8), j = rem >> 3, ch = rem & 7;
      *reinterpret_cast<uint4*>(lds + W3_DY_BYTES + ky * W3_BAND_BYTES + j * W3_PITCH + ch * 16) = v;
    }
  };
#define W3_WRITE(S)                                                                      \
  {                                                                                      \
    *reinterpret_cast<uint4*>(lds + lr * W3_PITCH + lc * 16) = ry0##S;                   \
    store_x(tid, rx0##S); store_x(tid + 512, rx1##S); store_x(tid + 1024, rx2##S); store_x(tid + 1536, rx3##S); \
  }

  cv_f32x16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // fragment geometry of this lane (td_conv1x1.hip wg_frag): supplier row 8 (g >> 1) + (i >> 2) (+4), column chunk 16 (g & 1) + 4 (i & 3)
  const int fg = lane >> 4, fi = lane & 15;
  const int frow = 8 * (fg >> 1) + (fi >> 2), fcol = (16 * (fg & 1) + 4 * (fi & 3)) * 2;

  // one stage out of LDS: k-step outermost (one dY fragment live at a time), this wave's (up to) five independent accumulators
  auto stage_mfma = [&](int st) {
    const int wo0 = (m_lo + st * W3_T) % Wo;
    auto band_off = [&](int d) {          // byte offset of pixel row d of the stage inside a band: position d + G k(d)
      const int t = wo0 + d;
      const int k = (t >= Wo) + (t >= 2 * Wo) + (t >= 3 * Wo) + (t >= 4 * Wo);
      return (d + G * k) * W3_PITCH + fcol;
    };
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned char* p = lds + (16 * ks + frow) * W3_PITCH + (wn * 32) * 2 + fcol;
      const cv_bf16x8 fa = w3_frag(p, p + 4 * W3_PITCH);
      const int off_lo = band_off(16 * ks + frow), off_hi = band_off(16 * ks + frow + 4);
#pragma unroll
      for (int ti = 0; ti < 5; ++ti) {
        const int tap = t_lo + ti;
        if (tap < 9) {                       // wave-uniform (EXEC stays full for the transposed reads)
          const int ky = tap / 3, kx = tap - 3 * ky;
          const unsigned char* band = lds + W3_DY_BYTES + ky * W3_BAND_BYTES + kx * W3_PITCH + (wc * 32) * 2;
          const cv_bf16x8 fb = w3_frag(band + off_lo, band + off_hi);
          acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[ti], 0, 0, 0);
        }
      }
    }
  };

  // branch-free prefetch (a conditional load makes the waitcnt pass drain the queue at the join): past the end the last stage is
  // re-read and never used
  W3_LOAD(0, A)
  W3_LOAD(last < 1 ? last : 1, B)
  W3_WRITE(A)
  __syncthreads();
  for (int st = 0; st < nstage; st += 2) {
    W3_LOAD(st + 2 < last ? st + 2 : last, A)
    __builtin_amdgcn_sched_barrier(0);
    stage_mfma(st);
    __syncthreads();            // every wave has read this stage
    W3_WRITE(B)
    __syncthreads();
    if (st + 1 >= nstage) break;
    W3_LOAD(st + 3 < last ? st + 3 : last, B)
    __builtin_amdgcn_sched_barrier(0);
    stage_mfma(st + 1);
    __syncthreads();
    W3_WRITE(A)
    __syncthreads();
  }
#undef W3_LOAD
#undef W3_WRITE
#undef W3_DECL
  // D[i = n][j = c]: lane -> c = c0 + 32 wc + (lane & 31), register r -> n = n0 + 32 wn + (r & 3) + 8 (r >> 2) + 4 (lane >> 5);
  // slab layout = the weight's memory [N][3][3][C]
  float* out = part + (size_t)sp * N * 9 * C;
  const int cc = c0 + wc * 32 + (lane & 31);
#pragma unroll
  for (int ti = 0; ti < 5; ++ti) {
    const int t = t_lo + ti;
    if (t < 9) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[((size_t)n * 9 + t) * C + cc] = acc[ti][r];
      }
    }
  }
}

// defined in td_conv1x1.hip: dW = ordered sum of P fp32 slabs, rounded once
int cv_wgrad_reduce(const float* part, int P, long long NK, int dw_dtype, void* dw, hipStream_t st);

static inline int w3_splits(long long M, int C, int N) {
  const long long tiles = (long long)(N / W3_T) * (C / W3_T);
  // one workgroup (8 waves) per CU: a stage is 36 MFMAs per wave pair and every split costs a 4 N 9 C-byte slab written and
  // re-read (measured shapes: profiles/r04/conv3x3_wgrad_bench_*.txt)
  long long p = (256 + tiles - 1) / tiles;
  const long long by_rows = (M + 4 * W3_T - 1) / (4 * W3_T);
  if (p > by_rows) p = by_rows;
  if (p > 512) p = 512;
  return (int)(p < 1 ? 1 : p);
}

}  // namespace td

static bool w3_shape_ok(int B, int Ho, int Wo, int C, int N, int pad) {
  return B > 0 && Ho >= 4 && Wo >= 20 && C > 0 && N > 0 && C % 64 == 0 && N % 64 == 0 && (pad == 0 || pad == 1);
}

extern "C" long long td_conv3x3_wgrad_workspace_floats(int B, int Ho, int Wo, int C, int N) {
  if (!w3_shape_ok(B, Ho, Wo, C, N, 1)) return 0;
  return (long long)td::w3_splits((long long)B * Ho * Wo, C, N) * N * 9 * C;
}

extern "C" int td_conv3x3_wgrad(const void* dy, const void* x, int B, int Ho, int Wo, int C, int N, int pad, int dw_dtype, void* dw,
                                float* workspace, td_stream_t stream) {
  if (!dy || !x || !dw || !workspace) return TD_ERR_BAD_ARG;
  if (!w3_shape_ok(B, Ho, Wo, C, N, pad)) return TD_ERR_UNSUPPORTED;
  if (dw_dtype != TD_DTYPE_BF16 && dw_dtype != TD_DTYPE_F32) return TD_ERR_UNSUPPORTED;
  const long long M = (long long)B * Ho * Wo;
  if (M >= (1ll << 31) - 128 || M * (long long)(C > N ? C : N) >= (1ll << 40)) return TD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int P = td::w3_splits(M, C, N);
  long long rps = (M + P - 1) / P;
  rps = (rps + td::W3_T - 1) / td::W3_T * td::W3_T;
  const int P_eff = (int)((M + rps - 1) / rps);
  const int tn = N / td::W3_T, tc = C / td::W3_T;
  const dim3 grid((unsigned)(tn * tc * P_eff));
  if (pad == 1)
    hipLaunchKernelGGL((td::conv3x3_wgrad_kernel<1>), grid, dim3(td::W3_THREADS), 0, st, (const __hip_bfloat16*)dy, (const __hip_bfloat16*)x,
                       workspace, M, Ho, Wo, C, N, (int)rps, tn, tc);
  else
    hipLaunchKernelGGL((td::conv3x3_wgrad_kernel<0>), grid, dim3(td::W3_THREADS), 0, st, (const __hip_bfloat16*)dy, (const __hip_bfloat16*)x,
                       workspace, M, Ho, Wo, C, N, (int)rps, tn, tc);
  if (hipGetLastError() != hipSuccess) return TD_ERR_LAUNCH;
  return td::cv_wgrad_reduce(workspace, P_eff, (long long)N * 9 * C, dw_dtype, dw, st);
}
