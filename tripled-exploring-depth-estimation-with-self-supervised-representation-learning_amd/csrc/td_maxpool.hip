// 5x5 / stride 1 / pad 2 max-pooling for channels-last (NHWC) activations -- the 16 pools of the
// CRP blocks (reference: mono/model/mono_fm_joint/layers.py:208, nn.MaxPool2d(5, 1, 2)).
// Forward (column march, below) keeps a 1-byte window offset dy * 5 + dx per element (ATen keeps an int64 index).
// Backward: on the large maps an LDS scatter with column turns, on the small ones a gather over the 25 outputs whose
// window contains the input element -- no atomics in either.
// Tie-break and NaN handling follow ATen's max_pool2d: row-major scan, strict '>', NaN wins.  A window in which nothing
// beats -inf sends its gradient to the centre (ATen's choice there differs between its NCHW and NHWC kernels).
#include <hip/hip_bf16.h>

#include "td_common.h"
#include "td_vec8.h"

namespace td {

// one thread = one input pixel x 8 channels: sum the gradients of the outputs that selected it
template <typename T>
__global__ __launch_bounds__(TD_THREADS) void maxpool5_bwd_kernel(const T* __restrict__ gout,
                                                                  const uint8_t* __restrict__ idx, int N, int H,
                                                                  int W, int C, T* __restrict__ gin, const T* __restrict__ add) {
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
  if (add) {                                  // + a gradient the input also receives directly (the CRP block's running sum)
    float r[8];
    load8(add + (size_t)pix * C + (size_t)cv * 8, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += r[i];
  }
  store8(gin + (size_t)pix * C + (size_t)cv * 8, acc);
}

// ---------------------------------------------------------------------------------------------
// Column-march forward.  A tiled forward (LDS tile staged by 7-8 dependent global loads per thread, then 25 LDS reads per
// output) ran at 163 us on the 47 MB map of the finest CRP stage, 0.7 TB/s of the 118 MB it has to move.  Here one thread owns
// a pixel column x 8 channels of a strip of rows and walks down it: every row costs 5 unconditional 16-byte loads (x-2..x+2,
// the horizontal neighbours come out of L1), the next row's loads are issued before this row's arithmetic, and the vertical
// window lives in a 5-deep register ring -- no LDS, no barrier, no branch on data: 65 us, now bound by the vector ALU (a
// wave64 instruction occupies its SIMD for 4 cycles; ~600 of them per 8 outputs).
//   row maximum + its first column (ATen's scan order is row-major, so "first maximum" = first row holding the maximum, first
//   column inside that row; "NaN wins" = last NaN = last row holding one, last NaN inside it: the two-stage scan keeps both
//   rules), then the same scan down the 5 ring rows.
// Rows/columns outside the map load a clamped address and are replaced by -inf.
// The backward is NOT this march turned around: that needs a 5-way compare-select per (column, channel) element to find the
// input row an output's offset points at (136 us with a register ring, 167 us with a thread-private LDS ring and ds_add_f32
// -- both more vector-ALU work than the gather's 124 us).  Maps of at least 1536 pixels with C % 64 == 0 take the LDS scatter
// with column turns below (maxpool5_bwd_scatter_kernel, 66 us); smaller maps keep the 25-candidate gather above.
struct MarchCoord {
  int cv, x, n, y0, y1;
  bool live;
};
// thread -> (image, strip of TH rows, column, 8-channel vector); channel vectors fastest: a wave reads whole pixels
__device__ __forceinline__ MarchCoord march_coord(int N, int H, int W, int C, int TH, int strips) {
  MarchCoord m;
  const int c8 = C >> 3;
  const long long gid = (long long)blockIdx.x * TD_THREADS + threadIdx.x;
  m.live = gid < (long long)N * strips * W * c8;
  long long t = m.live ? gid : 0;
  m.cv = (int)(t % c8); t /= c8;
  m.x = (int)(t % W); t /= W;
  const int strip = (int)(t % strips);
  m.n = (int)(t / strips);
  m.y0 = strip * TH;
  m.y1 = m.y0 + TH < H ? m.y0 + TH : H;
  return m;
}

template <typename T>
__global__ __launch_bounds__(TD_THREADS) void maxpool5_fwd_march_kernel(const T* __restrict__ in, int N, int H, int W, int C, int TH,
                                                                        int strips, T* __restrict__ out, uint8_t* __restrict__ idx) {
  const MarchCoord m = march_coord(N, H, W, C, TH, strips);
  if (!m.live) return;
  const T* base = in + (size_t)m.n * H * W * C + (size_t)m.cv * 8;
  bool okx[5];
  size_t xo[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int xx = m.x + k - 2;
    okx[k] = xx >= 0 && xx < W;
    xo[k] = (size_t)(okx[k] ? xx : m.x) * C;
  }
  const unsigned col0 = 0x22222222u;        // a window in which nothing beats -inf records its centre (offset 12)
  float hm[5][8];        // ring of row maxima, slot = (row - first row) % 5
  unsigned hd[5];        // their columns, 4 bits per channel
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    hd[s] = col0;
#pragma unroll
    for (int i = 0; i < 8; ++i) hm[s][i] = -INFINITY;
  }
  const int rstart = m.y0 - 2, rend = m.y1 + 1;
  Raw8<T> cur[5], nxt[5];
  {
    const int rc = rstart < 0 ? 0 : rstart;
#pragma unroll
    for (int k = 0; k < 5; ++k) cur[k].load(base + (size_t)rc * W * C + xo[k]);
  }
  for (int rbase = rstart; rbase <= rend; rbase += 5) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int r = rbase + j;
      if (r <= rend) {
        {   // next row's loads first
          int rn = r + 1;
          rn = rn < 0 ? 0 : (rn >= H ? H - 1 : rn);
#pragma unroll
          for (int k = 0; k < 5; ++k) nxt[k].load(base + (size_t)rn * W * C + xo[k]);
        }
        const bool oky = r >= 0 && r < H;
        float best[8];
        unsigned col = col0;
#pragma unroll
        for (int i = 0; i < 8; ++i) best[i] = -INFINITY;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          cur[k].neg_inf_unless(oky && okx[k]);
          float v[8];
          cur[k].unpack(v);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const bool take = (v[i] > best[i]) | (v[i] != v[i]);
            best[i] = take ? v[i] : best[i];
            col = take ? ((col & ~(0xfu << (4 * i))) | ((unsigned)k << (4 * i))) : col;
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) hm[j][i] = best[i];
        hd[j] = col;
        const int y = r - 2;                      // the window of output row y is complete: ring rows y-2..y+2
        if (y >= m.y0) {
          float o[8];
          unsigned sel[8], sdy[8];
          const unsigned dy0 = 2;
#pragma unroll
          for (int i = 0; i < 8; ++i) { o[i] = -INFINITY; sel[i] = col0; sdy[i] = dy0; }
#pragma unroll
          for (int dy = 0; dy < 5; ++dy) {
            const int s = (j + 1 + dy) % 5;       // slot of row y - 2 + dy (slot j holds row y + 2)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float v = hm[s][i];
              const bool take = (v > o[i]) | (v != v);
              o[i] = take ? v : o[i];
              sel[i] = take ? hd[s] : sel[i];
              sdy[i] = take ? (unsigned)dy : sdy[i];
            }
          }
          unsigned a[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] = sdy[i] * 5u + ((sel[i] >> (4 * i)) & 0xfu);
          const size_t off = (((size_t)m.n * H + y) * W + m.x) * C + (size_t)m.cv * 8;
          store8(out + off, o);
          uint2 packed;
          packed.x = a[0] | (a[1] << 8) | (a[2] << 16) | (a[3] << 24);
          packed.y = a[4] | (a[5] << 8) | (a[6] << 16) | (a[7] << 24);
          *reinterpret_cast<uint2*>(idx + off) = packed;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) cur[k] = nxt[k];
      }
    }
  }
}

// Scatter backward for the large maps.  The gather above tests 25 recorded offsets per input element (~800 vector-ALU
// instructions per 8 channels: 124 us on the finest CRP map, ALU-bound); an OUTPUT element knows its one target.  A workgroup
// takes 32 output columns x 64 channels (8 channel vectors fastest: a wave reads whole 128-byte pixel slabs) of a strip of
// rows and walks down it, rows requested 4 iterations ahead.  The accumulators are an LDS ring of 6 input rows x 32 columns,
// one array per channel-in-vector i (the compiler then knows the 8 read-add-writes of a thread are independent and
// pipelines them).  Two outputs of one row reach the same input element only from columns less than 5 apart, so the threads
// take turns by column % 5, a barrier between turns: inside a turn every address has one writer -- plain ds_read / add /
// ds_write, a fixed summation order, no atomics.  After the turns of output row r input row r - 2 has seen rows r-4..r:
// its slot is written out for the 28 interior columns the workgroup owns (tiles overlap by 2 + 2 output columns) and zeroed;
// with 6 slots the slot being read out is never a target of the next row's adds.  65.7 us on the finest map (24.3 at 24x80).
// Measured and dropped on the way: ds_add_f32 instead of turns (~0.3 lanes per clock: 235 us; integer ds_add_u32 runs the same
// kernel in 43 us); turns by recorded dx with per-element predicates (93 us); wave-owned channel vectors so that no barrier
// is needed at all (lane = column: every lane another 128-byte line, 112 us).
constexpr int MS_COLS = 32, MS_OWN = MS_COLS - 4, MS_SLOTS = 6;
constexpr int MS_PLANE = MS_SLOTS * MS_COLS * 8;             // floats per channel-in-vector array: [slot][column][vector]
template <typename T>
__global__ __launch_bounds__(TD_THREADS) void maxpool5_bwd_scatter_kernel(const T* __restrict__ gout, const uint8_t* __restrict__ idx, int N,
                                                                          int H, int W, int C, int TH, T* __restrict__ gin,
                                                                          const T* __restrict__ add) {
  __shared__ float t0[MS_PLANE], t1[MS_PLANE], t2[MS_PLANE], t3[MS_PLANE], t4[MS_PLANE], t5[MS_PLANE], t6[MS_PLANE], t7[MS_PLANE];
  float* const plane[8] = {t0, t1, t2, t3, t4, t5, t6, t7};
  const int tid = threadIdx.x, cv = tid & 7, lc = tid >> 3;   // channel vectors fastest: a wave reads 8 whole 128-byte pixel slabs
  const int slabs = C >> 6;
  const int n = blockIdx.z / slabs, slab = blockIdx.z - n * slabs;
  const int ox = (int)blockIdx.x * MS_OWN - 2 + lc;          // this lane's output column (and, for 2 <= lc < 30, its input column)
  const int y0 = (int)blockIdx.y * TH, y1 = y0 + TH < H ? y0 + TH : H;
  const bool okx = ox >= 0 && ox < W;
  const bool own = okx && lc >= 2 && lc < 2 + MS_OWN;
  const int turn = lc % 5;
  const int me = lc * 8 + cv;                                // (slot 0, own column, own vector); a slot is MS_COLS * 8 floats, a column 8
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int sl = 0; sl < MS_SLOTS; ++sl) plane[i][sl * (MS_COLS * 8) + me] = 0.f;
  __syncthreads();                                           // the first turn already adds into other threads' entries
  const size_t nb = (size_t)n * H * W * C + (size_t)slab * 64 + (size_t)cv * 8;
  const size_t xoff = (size_t)(okx ? ox : 0) * C;
  const int rstart = y0 - 2, rend = y1 + 1;
  constexpr int MS_AHEAD = 4;                                // rows requested ahead of use
  Raw8<T> gq[MS_AHEAD];
  uint2 iq[MS_AHEAD];
#pragma unroll
  for (int j = 0; j < MS_AHEAD; ++j) {
    int rc = rstart + j;
    rc = rc < 0 ? 0 : (rc >= H ? H - 1 : rc);
    const size_t o = nb + (size_t)rc * W * C + xoff;
    iq[j] = *reinterpret_cast<const uint2*>(idx + o);
    gq[j].load(gout + o);
  }
  int b = 4;                                                 // slot of input row r - 2 = (r - rstart + 4) % 6
  for (int rb = rstart; rb <= rend; rb += MS_AHEAD) {
#pragma unroll
    for (int j = 0; j < MS_AHEAD; ++j) {
      const int r = rb + j;                                  // rows past rend (the group of MS_AHEAD is always completed): no adds, no store
      const bool ok = okx && r >= 0 && r < H && r <= rend;
      const uint2 ci = iq[j];
      float g[8];
      gq[j].unpack(g);
      {
        int rn = r + MS_AHEAD;
        rn = rn < 0 ? 0 : (rn >= H ? H - 1 : rn);
        const size_t o = nb + (size_t)rn * W * C + xoff;
        iq[j] = *reinterpret_cast<const uint2*>(idx + o);
        gq[j].load(gout + o);
      }
      int off[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned code = ((i < 4 ? ci.x : ci.y) >> (8 * (i & 3))) & 0xffu;
        const unsigned dy = (code * 13u) >> 6;               // code / 5 for code <= 24
        const int dxs = (int)(code - 5u * dy) - 2;           // target column - own column
        const bool hit = ok && (unsigned)(lc + dxs) < (unsigned)MS_COLS;
        unsigned sl = (unsigned)b + dy;                      // slot of input row r + dy - 2
        sl = sl >= (unsigned)MS_SLOTS ? sl - MS_SLOTS : sl;
        off[i] = hit ? (int)sl * (MS_COLS * 8) + dxs * 8 + me : b * (MS_COLS * 8) + me;
        g[i] = hit ? g[i] : 0.f;
      }
#pragma unroll
      for (int p = 0; p < 5; ++p) {
        if (turn == p) {
#pragma unroll
          for (int i = 0; i < 8; ++i) plane[i][off[i]] += g[i];
        }
        __syncthreads();       // the next turn (and the read-out below) reads what other threads just wrote
      }
      const int y = r - 2;
      float a[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a[i] = plane[i][b * (MS_COLS * 8) + me];
        plane[i][b * (MS_COLS * 8) + me] = 0.f;
      }
      if (own && y >= y0 && y < y1) {
        const size_t o = nb + ((size_t)y * W + ox) * C;
        if (add) {
          float r8[8];
          load8(add + o, r8);
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] += r8[i];
        }
        store8(gin + o, a);
      }
      b = b + 1 == MS_SLOTS ? 0 : b + 1;
    }
  }
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
static int run_maxpool(bool fwd, const void* a, const void* aux, int N, int H, int W, int C, void* o, void* o2, hipStream_t st,
                       const void* add = nullptr) {
  if (fwd) {
    // strips of TH rows: enough threads to fill the chip on the small maps, at most (TH + 4) / TH re-read on the large ones
    int TH = H / 6;
    TH = TH < 1 ? 1 : (TH > 8 ? 8 : TH);
    const int strips = (H + TH - 1) / TH;
    const long long threads = (long long)N * strips * W * (C / 8);
    const long long nblk = (threads + TD_THREADS - 1) / TD_THREADS;
    if (nblk > 0x7fffffffll) return TD_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((maxpool5_fwd_march_kernel<T>), dim3((unsigned)nblk), dim3(TD_THREADS), 0, st, (const T*)a, N, H, W, C, TH, strips, (T*)o, (uint8_t*)o2);
    return record_launch_error(hipGetLastError(), "td_maxpool5_fwd");
  }
  if (C % 64 == 0 && (long long)H * W >= 1536 && (long long)N * (C / 64) <= 65535 && (H + 3) / 4 <= 65535) {
    // strips of TH rows ((TH + 4) / TH of the rows are read): as many as keep the launch within the 3 workgroups per CU the
    // LDS ring allows (a second, mostly empty round of workgroups would double the time), at least 8 rows each
    const long long per_strip = (long long)((W + MS_OWN - 1) / MS_OWN) * N * (C / 64);
    long long strips = (256 * 3) / per_strip;
    strips = strips < 1 ? 1 : strips;
    int TH = (int)((H + strips - 1) / strips);
    TH = TH < 8 ? 8 : TH;
    const dim3 grid((W + MS_OWN - 1) / MS_OWN, (H + TH - 1) / TH, N * (C / 64));
    hipLaunchKernelGGL((maxpool5_bwd_scatter_kernel<T>), grid, dim3(TD_THREADS), 0, st, (const T*)a, (const uint8_t*)aux, N, H, W, C, TH, (T*)o, (const T*)add);
    return record_launch_error(hipGetLastError(), "td_maxpool5_bwd");
  }
  const long long total = (long long)N * H * W * (C / 8);
  const unsigned blocks = (unsigned)((total + TD_THREADS - 1) / TD_THREADS);
  hipLaunchKernelGGL((maxpool5_bwd_kernel<T>), dim3(blocks), dim3(TD_THREADS), 0, st, (const T*)a, (const uint8_t*)aux, N, H, W, C, (T*)o, (const T*)add);
  return record_launch_error(hipGetLastError(), "td_maxpool5_bwd");
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

extern "C" int td_maxpool5_bwd_add(const void* grad_out, const uint8_t* idx, const void* add, int dtype, int N, int H, int W, int C,
                                   void* grad_in, td_stream_t stream) {
  if (!grad_out || !idx || !grad_in || N <= 0 || H <= 0 || W <= 0 || C <= 0) return TD_ERR_BAD_ARG;
  if (C % 8 != 0) return TD_ERR_UNSUPPORTED;
  if (dtype == TD_DTYPE_BF16) return td::run_maxpool<__hip_bfloat16>(false, grad_out, idx, N, H, W, C, grad_in, nullptr, (hipStream_t)stream, add);
  if (dtype == TD_DTYPE_F32) return td::run_maxpool<float>(false, grad_out, idx, N, H, W, C, grad_in, nullptr, (hipStream_t)stream, add);
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
