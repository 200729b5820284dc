// ttrnn_fast_gemm.hip — the batched input projection as ONE dense GEMM in split-bf16 fp32 arithmetic (gfx950).
//
// K-in computes gin[n] = W_in x[n] for all B*T rows at once: nothing is sequential, so the TT chain buys nothing but
// FLOPs there (cfg4: 2.4 MFLOP per row through the chain against 0.5 dense-equivalent; cfg5 with merged cores: equal).
// Per launch the dense matrix is produced by the chain kernel itself, applied to the `in` unit rows (WG[j][m'] in the
// gate-interleaved output order m' = 4*hid + slot that the recurrent kernels read), split into three bf16 planes
// (ttrnn_split.h) and then
//     gin[n][m'] = sum_j x[n][j] WG[j][m']       — six bf16-MFMA terms per product, fp32 accumulate
// runs as a 128 x 256-tile GEMM: x rows are split while they are staged into LDS (18 VALU per four elements, once per
// workgroup tile), both operands sit in LDS as [plane][row][32 k] with the 16-byte slots XOR-swizzled per row group
// (x_off: conflict-free ds_read_b128 fragment reads), chunks of 32 k double-buffered, one barrier per chunk.  MFMA rows are the output
// features, so a lane's four accumulators are the four gate slots of ONE hidden unit: one 16-byte store per lane.
// Tile order is XCD-aware: the 32 workgroups an XCD runs side by side form an 8 (feature tiles) x 4 (row tiles) block.
// Replaces t3nsor/layers.py:121-127 -> ops.py:54-93 for the input_weights of a whole sequence (lstm.py:25).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <type_traits>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"

namespace ttrnn {

namespace {
constexpr int GF = 128;                       // features per workgroup tile
constexpr int GR = 256;                       // rows per workgroup tile
constexpr int GT = GF;
constexpr int GK = 32;                        // k per chunk = one bf16 MFMA
constexpr int GPA = GR * GK, GPB = GF * GK;   // bf16 elements per plane tile (rows / features)
constexpr int G_BUF = 3 * GPA + 3 * GPB;      // one buffer: three planes of each operand
constexpr size_t G_LDS = (size_t)2 * G_BUF * sizeof(__bf16);
// HALF scratch: [one fp32 scale per COLUMN of W (output feature): M floats, padded to 256 bytes | one per row of x]
__host__ __device__ constexpr size_t g_rs_off(int M) { return (((size_t)M * sizeof(float) + 255) & ~(size_t)255) / sizeof(float); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Two-piece fp16 variant (HALF): every row of x is multiplied by its own power of two (max |x[n][:]| 2^e <= 2^14), every
// COLUMN of W (output feature) by its own (max_k |W[k][m]| 2^ew <= 2^14), both exact; product = x0w0 + x0w1 + x1w0
// (ttrnn_split.h), the epilogue undoes the two scales of its (row, feature).  An output's error is relative to (row
// maximum) x (its own column's maximum) — what matters for a gate pre-activation — and equals the bf16 variant's wherever
// the entries lie within 2^17 of those maxima; a large entry of W costs only its own output feature bits (round 2: one
// scale for the whole matrix).
__device__ __forceinline__ int g_expo(float x) {          // x < 2^e; zero / non-finite: neutral; clamped so that a scale
  if (!(x > 0.f) || !(x <= 3.4028235e38f)) return 0;       // 2^(14 - e) and its inverse stay normal fp32 numbers: every
  int e;                                                   // finite row / column is brought to < 2^15 (fp16 range); the
  frexpf(x, &e);                                           // two inverses are applied one after the other, the smaller
  return e < -100 ? -100 : e;                              // first (unscale2), never as their product
}
// acc / (row scale) / (column scale) without forming the product of the two inverses (2^228 apart at worst): the smaller
// factor first, so the intermediate overflows only where the result does
__device__ __forceinline__ float unscale2(float a, float u, float v) { return a * fminf(u, v) * fmaxf(u, v); }
__device__ __forceinline__ f32x4 unscale2(f32x4 a, f32x4 u, float v) {
  return f32x4{unscale2(a[0], u[0], v), unscale2(a[1], u[1], v), unscale2(a[2], u[2], v), unscale2(a[3], u[3], v)};
}
}  // namespace

// wsc[m] = 2^(14 - e), max_k |W(k, m)| < 2^e with W(k, m) = WG[k*sk + m*sm]: a workgroup takes 32 columns, its 8 thread
// groups stride over k with 8 independent loads in flight each (one thread per column walking all of k was a chain of
// 1 024 dependent-latency loads on 16 workgroups: 0.3 ms at cfg5's size)
__global__ void __launch_bounds__(256) k_gemm_col_scales(const float* __restrict__ WG, int K, int M, int64_t sk, int64_t sm,
                                                         float* __restrict__ wsc) {
  __shared__ float red[8][32];
  const int col = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int m = blockIdx.x * 32 + col;
  float mx = 0.f;
  if (m < M) {
    const float* base = WG + (size_t)m * sm;
    int k = kg;
    for (; k + 56 < K; k += 64) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fabsf(base[(size_t)(k + 8 * i) * sk]);
#pragma unroll
      for (int i = 0; i < 8; ++i) mx = fmaxf(mx, v[i]);
    }
    for (; k < K; k += 8) mx = fmaxf(mx, fabsf(base[(size_t)k * sk]));
  }
  red[kg][col] = mx;
  __syncthreads();
  if (kg == 0 && m < M) {
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, red[i][col]);
    wsc[m] = ldexpf(1.f, 14 - g_expo(mx));
  }
}

// rs[n] = 2^(14 - e), max_k |x[n][k]| < 2^e: one wave per row, 8 elements per lane and pass (K % 8 == 0)
template <typename TS>
__global__ void __launch_bounds__(256) k_row_scales(const TS* __restrict__ x, int64_t n_rows, int K,
                                                    float* __restrict__ rs);


template <typename TS>
__global__ void __launch_bounds__(256) k_fill_identity(TS* __restrict__ id, int K) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (size_t)K * K) st(id, e, (e / K == e % K) ? 1.0f : 0.0f);
}

// planes[p][kc][m][32] (bf16) <- W(k, m) = WG[k*sk + m*sm] (fp32), k zero-padded up to 32*KCn
__global__ void __launch_bounds__(256) k_gemm_prep(const float* __restrict__ WG, int K, int KCn, int M, int64_t sk,
                                                   int64_t sm, __bf16* __restrict__ planes) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // (k quad, m)
  const int m = (int)(e % M);
  const int k4 = (int)(e / M) * 4;
  if (k4 >= 32 * KCn) return;
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = k4 + i < K ? WG[(size_t)(k4 + i) * sk + (size_t)m * sm] : 0.f;
  unsigned a0, b0, c0, a1, b1, c1;
  split_pair(v[0], v[1], a0, b0, c0);
  split_pair(v[2], v[3], a1, b1, c1);
  const size_t pl = (size_t)KCn * M * 32;
  const size_t off = ((size_t)(k4 >> 5) * M + m) * 32 + (k4 & 31);
  *reinterpret_cast<u32x2*>(planes + off) = u32x2{a0, a1};
  *reinterpret_cast<u32x2*>(planes + pl + off) = u32x2{b0, b1};
  *reinterpret_cast<u32x2*>(planes + 2 * pl + off) = u32x2{c0, c1};
}

__global__ void __launch_bounds__(256) k_gemm_prep_h(const float* __restrict__ WG, int K, int KCn, int M, int64_t sk,
                                                     int64_t sm, const float* __restrict__ wsc,
                                                     _Float16* __restrict__ planes) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // (k quad, m)
  const int m = (int)(e % M);
  const int k4 = (int)(e / M) * 4;
  if (k4 >= 32 * KCn) return;
  const float ws = wsc[m];
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = k4 + i < K ? WG[(size_t)(k4 + i) * sk + (size_t)m * sm] * ws : 0.f;
  unsigned a0, b0, a1, b1;
  split_pair_h(v[0], v[1], a0, b0);
  split_pair_h(v[2], v[3], a1, b1);
  const size_t pl = (size_t)KCn * M * 32;
  const size_t off = ((size_t)(k4 >> 5) * M + m) * 32 + (k4 & 31);
  *reinterpret_cast<u32x2*>(planes + off) = u32x2{a0, a1};
  *reinterpret_cast<u32x2*>(planes + pl + off) = u32x2{b0, b1};
}

__device__ __forceinline__ void ld8(const float* p, size_t i, f32x4& a, f32x4& b) {
  a = *reinterpret_cast<const f32x4*>(p + i);
  b = *reinterpret_cast<const f32x4*>(p + i + 4);
}
__device__ __forceinline__ void ld8(const bf16_t* p, size_t i, f32x4& a, f32x4& b) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(p + i);
  a = f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xFFFF0000u), __uint_as_float(v[1] << 16),
            __uint_as_float(v[1] & 0xFFFF0000u)};
  b = f32x4{__uint_as_float(v[2] << 16), __uint_as_float(v[2] & 0xFFFF0000u), __uint_as_float(v[3] << 16),
            __uint_as_float(v[3] & 0xFFFF0000u)};
}

// y[n][m'] (fp32, row stride M) = sum_k x[n][k] W[k][m'] (+ bias of hidden unit m'/4, slots i,g,f,o when bias != NULL)
// Workgroup tile: 128 features x 256 rows, wave tile 64 x 64 (4 x 4 MFMA tiles: 24 fragment reads feed 96 MFMAs per
// chunk — the 128 x 128 tile with 32 x 64 wave tiles spent as long in LDS traffic, splitting and barriers as in MFMAs).
// FW = 2 (HALF only): 256 features x 256 rows per workgroup, wave tile 128 x 64 — the row operand, whose staging carries the
// split, is then staged once per 96 instead of 48 MFMAs of a wave, and there is one barrier per 96
template <typename TS, bool HALF, int FW = 1>
__global__ void __launch_bounds__(FAST_NT) k_gemm_split(int64_t n_rows, int K, int KCn, int M,
                                                        const TS* __restrict__ x, const void* __restrict__ planes_v,
                                                        const TS* __restrict__ bias, int Hb, float* __restrict__ y,
                                                        const float* __restrict__ bias_ilv,
                                                        const float* __restrict__ scratch) {
  // HALF: two fp16 pieces, three terms, scales in `scratch` (one per column of W, then one per row of x: g_rs_off)
  const float* rowsc = scratch + g_rs_off(M);
  using E = typename std::conditional<HALF, _Float16, __bf16>::type;
  using X8 = typename std::conditional<HALF, xh8, xbf8>::type;
  constexpr int NP = HALF ? 2 : 3;
  constexpr int GPBW = FW * GPB, GFW = FW * GF, MI = 4 * FW;
  constexpr int BUF = NP * (GPA + GPBW);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  E* lds = reinterpret_cast<E*>(smem);                     // [buf][A planes NP][256][32], [B planes NP][128][32]
  const E* planes = reinterpret_cast<const E*>(planes_v);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  // ---- tile of this workgroup (XCD-aware when the feature tiles come in groups of 8) ----------------------------------
  const int MT = M / GFW;
  const int64_t RT = (n_rows + GR - 1) / GR;
  int mt_tile;
  int64_t rt_tile;
  if (MT % 8 == 0) {
    const int xcd = blockIdx.x & 7;
    const int64_t i = blockIdx.x >> 3;
    const int64_t g = (i >> 5) * 8 + xcd;                 // super-tile: 8 feature tiles x 4 row tiles
    const int within = (int)(i & 31);
    const int smc = MT / 8;
    mt_tile = (int)(g % smc) * 8 + (within & 7);
    rt_tile = (g / smc) * 4 + (within >> 3);
  } else {
    mt_tile = (int)(blockIdx.x % MT);
    rt_tile = blockIdx.x / MT;
  }
  if (rt_tile >= RT) return;
  const int m0 = mt_tile * GFW;
  const int64_t n0 = rt_tile * GR;
  const int wm = wave & 1, wr = wave >> 1;                 // wave tile: features [64 FW wm, +64 FW) x rows [64 wr, +64)

  // ---- staging: thread -> (rows tid >> 2 and 128 + (tid >> 2) / feature tid >> 2, k group tid & 3) ------------------------
  const int srow = tid >> 2, skq = tid & 3;
  const TS* xrow[2];
  float xsc[2] = {1.f, 1.f};                               // HALF: the staged rows' scales
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int64_t an = n0 + srow + 128 * e < n_rows ? n0 + srow + 128 * e : n_rows - 1;
    xrow[e] = x + (size_t)an * K;
    if constexpr (HALF) xsc[e] = rowsc[an];
  }
  const size_t plane_elems = (size_t)KCn * M * 32;
  const E* wrow = planes + (size_t)(m0 + srow) * 32 + 8 * skq;
  f32x4 xa[2], xb[2];
  X8 wb[FW][NP];
  auto stage_load = [&](int kc) {
    const int k = kc * GK + 8 * skq;
    const int kcl = k + 8 <= K ? k : (K >= 8 ? K - 8 : 0);      // unconditional loads; out-of-range groups are zeroed below
#pragma unroll
    for (int e = 0; e < 2; ++e) ld8(xrow[e], (size_t)kcl, xa[e], xb[e]);
#pragma unroll
    for (int f = 0; f < FW; ++f)
#pragma unroll
      for (int p = 0; p < NP; ++p)
        wb[f][p] = *reinterpret_cast<const X8*>(wrow + (size_t)f * GF * 32 + p * plane_elems + (size_t)kc * M * 32);
  };
  auto stage_store = [&](int buf, int kc) {
    E* As = lds + buf * BUF;
    E* Bs = As + NP * GPA;
    const int k = kc * GK + 8 * skq;
    const float keep = k + 8 <= K ? 1.0f : 0.0f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const f32x4 va = xa[e] * (keep * xsc[e]), vb = xb[e] * (keep * xsc[e]);
      const int off = x_off<GK>(srow + 128 * e, 8 * skq);   // 16-byte slots XOR-swizzled per row group (ttrnn_split.h)
      if constexpr (HALF) {
        unsigned p0[4], p1[4];
        split_pair_h(va[0], va[1], p0[0], p1[0]);
        split_pair_h(va[2], va[3], p0[1], p1[1]);
        split_pair_h(vb[0], vb[1], p0[2], p1[2]);
        split_pair_h(vb[2], vb[3], p0[3], p1[3]);
        *reinterpret_cast<u32x4*>(As + off) = u32x4{p0[0], p0[1], p0[2], p0[3]};
        *reinterpret_cast<u32x4*>(As + GPA + off) = u32x4{p1[0], p1[1], p1[2], p1[3]};
      } else {
        unsigned p0[4], p1[4], p2[4];
        split_pair(va[0], va[1], p0[0], p1[0], p2[0]);
        split_pair(va[2], va[3], p0[1], p1[1], p2[1]);
        split_pair(vb[0], vb[1], p0[2], p1[2], p2[2]);
        split_pair(vb[2], vb[3], p0[3], p1[3], p2[3]);
        *reinterpret_cast<u32x4*>(As + off) = u32x4{p0[0], p0[1], p0[2], p0[3]};
        *reinterpret_cast<u32x4*>(As + GPA + off) = u32x4{p1[0], p1[1], p1[2], p1[3]};
        *reinterpret_cast<u32x4*>(As + 2 * GPA + off) = u32x4{p2[0], p2[1], p2[2], p2[3]};
      }
    }
#pragma unroll
    for (int f = 0; f < FW; ++f) {
      const int offb = x_off<GK>(srow + GF * f, 8 * skq);
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<X8*>(Bs + p * GPBW + offb) = wb[f][p];
    }
  };

  // one fp32 accumulator per tile: the six terms of a chunk are added smallest first (SPLIT_TW / SPLIT_TX order)
  f32x4 acc[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ri = 0; ri < 4; ++ri) acc[mi][ri] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_load(0);
  stage_store(0, 0);
  stage_load(1 < KCn ? 1 : 0);
  for (int kc = 0; kc < KCn; ++kc) {
    const int buf = kc & 1;
    lds_barrier();                                         // chunk kc is in `buf`; nobody reads the other buffer any more
    const E* As = lds + buf * BUF;
    const E* Bs = As + NP * GPA;
    // feature tiles in groups of four (FW = 2: two passes over the row tiles, so that only 32 fragment registers are live)
#pragma unroll
    for (int h = 0; h < FW; ++h) {
      X8 wf[4][NP];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int p = 0; p < NP; ++p)
          wf[mi][p] = *reinterpret_cast<const X8*>(Bs + p * GPBW + x_off<GK>(wm * 64 * FW + 64 * h + 16 * mi + c, 8 * q));
#pragma unroll
      for (int ri = 0; ri < 4; ++ri) {
        X8 af[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
          af[p] = *reinterpret_cast<const X8*>(As + p * GPA + x_off<GK>(wr * 64 + 16 * ri + c, 8 * q));
        // the next chunk's split + LDS stores (other buffer; its data was requested a chunk ago) ride inside the MFMA stream
        if (h == 0 && ri == 1) {                            // unconditional: no branch inside the chunk
          stage_store(buf ^ 1, kc + 1);                     // past the end it fills the idle buffer with a masked chunk
          stage_load(kc + 2 < KCn ? kc + 2 : KCn - 1);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          f32x4& a = acc[4 * h + mi][ri];
          if constexpr (HALF) {
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mi][1], af[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mi][0], af[1], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mi][0], af[0], a, 0, 0, 0);
          } else {
#pragma unroll
            for (int s = 0; s < 6; ++s)
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mi][SPLIT_TW[s]], af[SPLIT_TX[s]], a, 0, 0, 0);
          }
        }
      }
    }
  }
  // HALF: undo the scales (powers of two: exact), one factor per output row and one per output feature
  float unsc[4] = {1.f, 1.f, 1.f, 1.f};
  if constexpr (HALF) {
#pragma unroll
    for (int ri = 0; ri < 4; ++ri) {
      const int64_t n = n0 + wr * 64 + 16 * ri + c;
      unsc[ri] = 1.0f / rowsc[n < n_rows ? n : n_rows - 1];
    }
  }
  // ---- epilogue: lane (c, q) of tile (mi, ri) holds features m0 + 64wm + 16mi + 4q .. +3 of row n0 + 64wr + 16ri + c ------
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int mf = m0 + wm * 64 * FW + 16 * mi + 4 * q;
    f32x4 bh = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) {
      const int hd = mf >> 2;
      bh = f32x4{ld(bias, hd), ld(bias, 2 * Hb + hd), ld(bias, Hb + hd), ld(bias, 3 * Hb + hd)};     // slots i,g,f,o
    } else if (bias_ilv) {
      bh = *reinterpret_cast<const f32x4*>(bias_ilv + mf);          // fp32 row already in column order (ttrnn_g2.hip)
    }
    f32x4 unf = f32x4{1.f, 1.f, 1.f, 1.f};
    if constexpr (HALF) {
      const f32x4 ws4 = *reinterpret_cast<const f32x4*>(scratch + mf);
      unf = f32x4{1.0f / ws4[0], 1.0f / ws4[1], 1.0f / ws4[2], 1.0f / ws4[3]};
    }
#pragma unroll
    for (int ri = 0; ri < 4; ++ri) {
      const int64_t n = n0 + wr * 64 + 16 * ri + c;
      if (n < n_rows) *reinterpret_cast<f32x4*>(y + (size_t)n * M + mf) = (HALF ? unscale2(acc[mi][ri], unf, unsc[ri]) : acc[mi][ri]) + bh;
    }
  }
}

template <typename TS>
__global__ void __launch_bounds__(256) k_row_scales(const TS* __restrict__ x, int64_t n_rows, int K,
                                                    float* __restrict__ rs) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  for (int64_t n = w; n < n_rows; n += nw) {
    float m = 0.f;
    for (int k = 8 * lane; k < K; k += 512) {
      f32x4 a, b;
      ld8(x + (size_t)n * K, (size_t)k, a, b);
      m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))),
                         fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3])))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) rs[n] = ldexpf(1.f, 14 - g_expo(m));
  }
}

// ---- dense weight gradient dW[j][o] (+)= sum_n x[n][j] dy[n][o] on the fp32 MFMA ----------------------------------------------
// Both operands have the contraction index n as their SLOW index, which the k-packed bf16 MFMA operands cannot take
// without a transpose; the fp32 16x16x4 MFMA takes one value per lane and reads them straight out of row-major LDS tiles
// (row stride 144 floats: the four k rows of an operand fall on four different 16-bank groups).  128 x 128 tile per
// workgroup, rows staged 64 at a time, double-buffered, LDS-only barrier.  KS > 1 splits the rows over KS workgroups per
// tile (atomic flush into a zeroed dW) when there are fewer tiles than CUs; all tiles of one row range run on one XCD
// (blockIdx % 8), so its L2 serves every x / dy line to all of its users from one HBM read.
struct DenseG {
  static constexpr int TM = 128, TN = 128, KB = 64, LS = 144;
  static constexpr size_t LDS_BYTES = (size_t)2 * 2 * KB * LS * sizeof(float);
};

__device__ __forceinline__ f32x4 ld4(const float* p, size_t i) { return *reinterpret_cast<const f32x4*>(p + i); }
__device__ __forceinline__ f32x4 ld4(const bf16_t* p, size_t i) {
  const uint2 v = *reinterpret_cast<const uint2*>(p + i);
  return f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xFFFF0000u), __uint_as_float(v.y << 16),
               __uint_as_float(v.y & 0xFFFF0000u)};
}

// Operand rows that are the PREVIOUS step's hidden states (the h_{t-1} rows of a hidden matrix's weight gradient) read in
// place: x = out[B][T][IN] of the layer, row n of the operand = out row n - 1, or row n / T of `first` (h_0; zeros if
// NULL) where n % T == 0.  T == 0: plain rows.  (Row counts below 2^31: a 32-bit division per staged quad.)
struct RowShift {
  int T = 0;
  const void* first = nullptr;
};
template <typename TS>
__device__ __forceinline__ f32x4 ld4_rows(const TS* __restrict__ x, const RowShift& rs, int64_t n, int IN, int col) {
  if (rs.T <= 0) return ld4(x, (size_t)n * IN + col);
  const unsigned b = (unsigned)n / (unsigned)rs.T;
  const bool head = (unsigned)n == b * (unsigned)rs.T;
  const TS* fp = reinterpret_cast<const TS*>(rs.first);
  const TS* src = head ? (fp ? fp + (size_t)b * IN : x) : x + (size_t)(n - 1) * IN;
  const f32x4 v = ld4(src, (size_t)col);
  return (head && !fp) ? f32x4{0.f, 0.f, 0.f, 0.f} : v;
}

// ---- bias gradient = column sums of dy, WITHOUT atomics ------------------------------------------------------------------
// Round 3 flushed every staging thread's running column sums with atomicAdd: 8 192 atomics per workgroup onto 256 addresses
// (16 per thread), 4 096 per address and launch at KS = 128 — same-address atomics retire one after the other at L2, and on
// small problems (B*T = 4 096 rows: one chunk per workgroup) that queue WAS the kernel: 400 us for 5 us of work
// (profiles/r4/seq_h64_before.txt), and the sums' last bits depended on the order of arrival.  Now: the G staging groups of a
// workgroup (threads that stage the same column quad) are summed through LDS in a fixed order, one thread per column
// stores the workgroup's sum to bpart[ks][column], and k_dense_bias_reduce adds the row ranges in a fixed order:
// repeatable bit for bit, no atomics.
template <int W, int G>
__device__ __forceinline__ void bias_partial(float* red, const f32x4& v, int group, int col4, float* __restrict__ dst, int tid,
                                             int ncols = W) {      // ncols < W: a last, partly filled column tile
  lds_barrier();                                           // every wave is done with the staging buffers `red` overlays
  *reinterpret_cast<f32x4*>(red + group * W + col4) = v;
  lds_barrier();
  if (tid < ncols) {
    float sum = red[tid];
#pragma unroll
    for (int g = 1; g < G; ++g) sum += red[g * W + tid];
    dst[tid] = sum;
  }
}

// d_bias[o] += sum over the KS row ranges of bpart[ks][o] (d_bias is accumulated into by contract).  One thread per column
// walking KS = 128 ranges was a chain of 128 dependent-issue loads (30 us); 32 columns x 8 range groups per workgroup, each
// thread its group's ranges in order, the eight group sums added in order through LDS: the same fixed order on every launch.
__global__ void __launch_bounds__(256) k_dense_bias_reduce(const float* __restrict__ bpart, int KS, int OUT,
                                                           float* __restrict__ d_bias) {
  __shared__ float red[8][32];
  const int c = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int o = blockIdx.x * 32 + c;
  const int per = (KS + 7) / 8, k0 = kg * per, k1 = k0 + per < KS ? k0 + per : KS;
  float v = 0.f;
  if (o < OUT)
    for (int k = k0; k < k1; ++k) v += bpart[(size_t)k * OUT + o];
  red[kg][c] = v;
  __syncthreads();
  if (kg == 0 && o < OUT) {
    float sum = red[0][c];
#pragma unroll
    for (int g = 1; g < 8; ++g) sum += red[g][c];
    d_bias[o] += sum;
  }
}

template <typename TS>
__global__ void __launch_bounds__(FAST_NT) k_dense_wgrad(int64_t n_rows, int IN, int OUT, int KS, int64_t rows_per,
                                                         const TS* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dW, float* __restrict__ bpart, float* __restrict__ part,
                                                         RowShift rsh) {
  constexpr int KB = DenseG::KB, LS = DenseG::LS;
  extern __shared__ __attribute__((aligned(16))) float ldsf[];
  float* xs = ldsf;                      // [2][KB][LS]
  float* ds = ldsf + 2 * KB * LS;        // [2][KB][LS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  // a last, partly filled j tile is masked; so is a last column tile (round 5: gate widths that are not a multiple of 128 — a TT-GRU
  // with H = 64 has 192 — used to leave this route for the per-row kernels and their atomics: 12 of the 16 grid shapes whose
  // gradients were not repeatable in round 4)
  const int TJ = (IN + DenseG::TM - 1) / DenseG::TM, TO = (OUT + DenseG::TN - 1) / DenseG::TN;
  int tj, to, ks;
  if (KS == 1 && TJ == 8 && TO == 32) {
    // one workgroup per tile: the 32 workgroups of an XCD form a 4 x 8 block of tiles
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    tj = (xcd >> 2) * 4 + (local >> 3);
    to = (xcd & 3) * 8 + (local & 7);
    ks = 0;
  } else if (KS % 8 == 0) {
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3, tile = i % (TJ * TO);
    ks = (i / (TJ * TO)) * 8 + xcd;
    tj = tile / TO;
    to = tile % TO;
  } else {
    const int tile = blockIdx.x % (TJ * TO);
    ks = blockIdx.x / (TJ * TO);
    tj = tile / TO;
    to = tile % TO;
  }
  const int j0 = tj * DenseG::TM, o0 = to * DenseG::TN;
  const int64_t r0 = (int64_t)ks * rows_per;
  const int64_t r1 = r0 + rows_per < n_rows ? r0 + rows_per : n_rows;
  if (r0 >= r1) return;
  const int wm = wave & 1, wn = wave >> 1;               // wave tile: 64 (j) x 32 (o)

  f32x4 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) { acc[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mi][1] = acc[mi][0]; }
  f32x4 dbs = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool want_bias = bpart != nullptr && tj == 0;      // (workgroup-uniform)

  // staging: thread -> rows (tid / 32) + 16e of the chunk, four consecutive columns.  The loads of chunk ch+1 are issued
  // right after chunk ch went to LDS and are not touched (not even scaled) before the next iteration
  constexpr int SR = KB / 16;
  const int srow = tid >> 5, scol = (tid & 31) * 4;
  const bool jin = j0 + scol + 4 <= IN;                   // IN % 4 == 0: a column quad is in or out as a whole
  const int jc = jin ? j0 + scol : 0;
  const bool oin = o0 + scol + 4 <= OUT;                  // OUT % 4 == 0 likewise
  const int oc = oin ? o0 + scol : 0;
  f32x4 sx[SR], sd[SR];
  auto stage_load = [&](int64_t nb) {
#pragma unroll
    for (int e = 0; e < SR; ++e) {
      const int64_t n = nb + srow + 16 * e;
      const int64_t nc = n < r1 ? n : r1 - 1;               // unconditional loads; rows past the end are zeroed at the store
      sx[e] = ld4_rows(x, rsh, nc, IN, jc);
      sd[e] = ld4(dy, (size_t)nc * OUT + oc);
    }
  };
  stage_load(r0);
  const int64_t chunks = (r1 - r0 + KB - 1) / KB;
  for (int64_t ch = 0; ch < chunks; ++ch) {
    const int buf = (int)(ch & 1);
    float* xb = xs + buf * KB * LS;
    float* db = ds + buf * KB * LS;
#pragma unroll
    for (int e = 0; e < SR; ++e) {
      const float keep = r0 + ch * KB + srow + 16 * e < r1 ? 1.0f : 0.0f;
      const f32x4 vd = sd[e] * (oin ? keep : 0.0f);
      *reinterpret_cast<f32x4*>(xb + (srow + 16 * e) * LS + scol) = sx[e] * (jin ? keep : 0.0f);
      *reinterpret_cast<f32x4*>(db + (srow + 16 * e) * LS + scol) = vd;
      dbs += vd;
    }
    stage_load(r0 + (ch + 1 < chunks ? ch + 1 : ch) * KB);
    lds_barrier();                                         // LDS hand-off only: the next chunk's global loads stay in flight
#pragma unroll
    for (int sp = 0; sp < KB / 4; ++sp) {
      float a[4], bv[2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a[mi] = xb[(4 * sp + q) * LS + wm * 64 + 16 * mi + c];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) bv[ni] = db[(4 * sp + q) * LS + wn * 32 + 16 * ni + c];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], bv[ni], acc[mi][ni], 0, 0, 0);
    }
    // the next chunk goes to the other buffer; its barrier orders these reads before the stores of the chunk after it
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int jj = j0 + wm * 64 + 16 * mi + 4 * q + j;
        if (jj < IN && o0 + wn * 32 + 16 * ni + c < OUT) {
          const size_t e = (size_t)jj * OUT + o0 + wn * 32 + 16 * ni + c;
          if (KS == 1) dW[e] = acc[mi][ni][j];
          else if (part) part[(size_t)ks * IN * OUT + e] = acc[mi][ni][j];      // summed by k_dense_reduce
          else atomicAdd(dW + e, acc[mi][ni][j]);
        }
      }
  if (want_bias)      // 16 staging rows x 32 column quads: sixteen groups over the tile's 128 columns
    bias_partial<DenseG::TN, 16>(ldsf, dbs, srow, scol, bpart + (size_t)ks * OUT + o0, tid, OUT - o0 < DenseG::TN ? OUT - o0 : DenseG::TN);
}

// ---- the same dense gradient on the bf16 MFMA (split arithmetic) ---------------------------------------------------------------
// The contraction index (row n) is the slow index of both operands, so the k-packed MFMA operands are columns of the
// staged tiles: gfx950's transposing LDS read (ds_read_b64_tr_b16: a 4-row x 16-column block of 16-bit elements,
// delivered column-major) fetches them from ROW-major bf16 planes, so staging stays a plain coalesced copy + split
// (three 8-byte stores per four elements).  Plane layout: [32 rows][128 columns] with the 16-byte chunks XOR-swizzled by
// the row (conflict-free for the row stores and the transposed reads); 128 (j) x 256 (o) tile per workgroup, 64 x 64 per
// wave, 32 rows per chunk = one MFMA k-step, both operands split on the fly.
struct DenseS {
  static constexpr int TJ = 128, TO = 256, KB = 32;
  static constexpr int PLANE = KB * 128;                   // bf16 elements of one [32][128] plane
  static constexpr int BUF = 3 * PLANE * 3;                // x: 3 planes; dy: 3 planes x 2 column halves
  static constexpr size_t LDS_BYTES = (size_t)2 * BUF * sizeof(__bf16);
  static constexpr int BUF_H = 2 * PLANE * 3;              // two-piece fp16 operands: two planes each
  static constexpr size_t LDS_BYTES_H = (size_t)2 * BUF_H * sizeof(_Float16);
};
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Two-piece fp16 variant (HALF) of the dense gradient: the contraction runs over the ROWS, so a scale may differ per
// COLUMN of x and per column of dy (it factors out of every sum): colmax = [IN + OUT] column maxima as fp32 bit patterns
// (non-negative floats order like unsigned integers: atomicMax), scale = 2^(14 - e) with max < 2^e, exact; the epilogue
// multiplies dW[j][o] by the two inverse powers of two.  Entries within 2^17 of their column's maximum keep 22+ bits.
__device__ __forceinline__ float col_scale(unsigned bits) { return ldexpf(1.f, 14 - g_expo(__uint_as_float(bits))); }
__device__ __forceinline__ float col_unscale(unsigned bits) { return ldexpf(1.f, g_expo(__uint_as_float(bits)) - 14); }

// colmax[c] = max over the rows of |a[n][c]| (bit pattern).  A thread per four columns and row group; narrow matrices
// (C / 4 < 256 column quads) spread their rows over the otherwise idle threads, take up to eight times the rows per workgroup
// and reduce the row groups in LDS before ONE atomic per column and workgroup: at one thread per quad the 40-column input of
// cfg4's first layer kept 10 of 256 threads busy and sent 26 k atomics to 40 addresses (83 us for 13 MB).
__host__ __device__ inline int colmax_groups(int C) { return 256 / (C / 4 < 256 ? C / 4 : 256); }
__host__ __device__ inline int colmax_rows(int C) { const int g = colmax_groups(C); return 128 * (g < 8 ? g : 8); }
template <typename TS>
__global__ void __launch_bounds__(256) k_col_absmax(const TS* __restrict__ a, int64_t n_rows, int C,
                                                    unsigned* __restrict__ colmax) {
  __shared__ f32x4 red[256];
  const int nq = C / 4 < 256 ? C / 4 : 256;                  // column quads of this workgroup's x-slice
  const int ng = 256 / nq;                                   // row groups
  const int cq = threadIdx.x % nq, rg = threadIdx.x / nq;
  const int c4 = (blockIdx.x * 256 + cq) * 4;
  const bool live = c4 < C && rg < ng;
  const int RB = colmax_rows(C);
  const int64_t r0 = (int64_t)blockIdx.y * RB, r1 = r0 + RB < n_rows ? r0 + RB : n_rows;
  f32x4 m = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live)
    // four rows in flight per thread (one load per iteration left the pass latency-bound: 72 us for the 84 MB of a 81 920 x 256
    // input); rows past the end repeat the last one (max is idempotent)
    for (int64_t n = r0 + rg; n < r1; n += 4 * ng) {
      f32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t nk = n + (int64_t)k * ng < r1 ? n + (int64_t)k * ng : n;
        v[k] = ld4(a, (size_t)nk * C + c4);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], fabsf(v[k][i]));
    }
  if (ng == 1) {
    if (live)
#pragma unroll
      for (int i = 0; i < 4; ++i) atomicMax(colmax + c4 + i, __float_as_uint(m[i]));
    return;
  }
  red[threadIdx.x] = m;
  __syncthreads();
  if (live && rg == 0) {
    for (int g = 1; g < ng; ++g) {
      const f32x4 o = red[g * nq + cq];
#pragma unroll
      for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], o[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) atomicMax(colmax + c4 + i, __float_as_uint(m[i]));
  }
}

// byte offset of 16-byte chunk ch (0..15) of row `row` in a [rows][128 x 16-bit] plane
__device__ __forceinline__ int tr_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// the 8 consecutive rows 8g .. 8g+7 (g = lane >> 4) of column 16*cb + (lane & 15): the k-packed MFMA operand of a lane
__device__ __forceinline__ xbf8 tr_frag(const __bf16* plane, int cb, int lane) {
  const int g = lane >> 4, qq = (lane & 15) >> 2, p = lane & 3;
  const char* base = reinterpret_cast<const char*>(plane);
  const int o0 = tr_off(8 * g + qq, 2 * cb + (p >> 1)) + 8 * (p & 1);
  const int o1 = tr_off(8 * g + 4 + qq, 2 * cb + (p >> 1)) + 8 * (p & 1);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(xbf8, v);
}

template <typename TS, bool HALF, bool SHIFT, bool ABL = false>
__global__ void __launch_bounds__(FAST_NT) k_dense_wgrad_split(int64_t n_rows, int IN, int OUT, int KS, int64_t rows_per,
                                                               const TS* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dW, float* __restrict__ bpart, float* __restrict__ part,
                                                               const unsigned* __restrict__ colmax_x,
                                                               const unsigned* __restrict__ colmax_dy, RowShift rsh, int dev) {
  constexpr int KB = DenseS::KB, PL = DenseS::PLANE;
  constexpr int NP = HALF ? 2 : 3, BUFE = HALF ? DenseS::BUF_H : DenseS::BUF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
  __bf16* ldsb = reinterpret_cast<__bf16*>(smem2);      // (fp16 pieces in the HALF variant: same 16-bit layout)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int TJn = (IN + DenseS::TJ - 1) / DenseS::TJ, TOn = OUT / DenseS::TO;
  int tj, to, ks;
  if (KS % 8 == 0) {
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3, tile = i % (TJn * TOn);
    ks = (i / (TJn * TOn)) * 8 + xcd;                      // all tiles of one row range on one XCD ...
    if (TJn % 4 == 0 && TOn % 8 == 0) {                    // ... in blocks of 4 x 8 tiles (the 32 that run side by side)
      const int blk = tile >> 5, within = tile & 31, nbo = TOn / 8;
      tj = (blk / nbo) * 4 + (within >> 3);
      to = (blk % nbo) * 8 + (within & 7);
    } else {
      tj = tile / TOn;
      to = tile % TOn;
    }
  } else {
    const int tile = blockIdx.x % (TJn * TOn);
    ks = blockIdx.x / (TJn * TOn);
    tj = tile / TOn;
    to = tile % TOn;
  }
  const int j0 = tj * DenseS::TJ, o0 = to * DenseS::TO;
  const int64_t r0 = (int64_t)ks * rows_per;
  const int64_t r1 = r0 + rows_per < n_rows ? r0 + rows_per : n_rows;
  if (r0 >= r1) return;
  const int wm = wave & 1, wn = wave >> 1;                 // wave tile: j [64 wm, +64) x o [64 wn, +64)

  f32x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dbs[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) dbs[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool want_bias = bpart != nullptr && tj == 0;      // (workgroup-uniform)

  // staging: x quads tid + 512e (row = id >> 5, columns 4 (id & 31)); dy quads tid + 512e (row = id >> 6, columns 4 (id & 63))
  f32x4 sx[2], sd[4];
  int xr[2], xc[2], dr[4], dc[4];
  bool xin[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int id = tid + FAST_NT * e;
    xr[e] = id >> 5;
    xc[e] = 4 * (id & 31);
    xin[e] = j0 + xc[e] + 4 <= IN;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int id = tid + FAST_NT * e;
    dr[e] = id >> 6;
    dc[e] = 4 * (id & 63);
  }
  // HALF: the staged columns' scales (a thread always stages the same columns)
  f32x4 scx[2], scd[4];
  if constexpr (HALF) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) scx[e][i] = xin[e] ? col_scale(colmax_x[j0 + xc[e] + i]) : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) scd[e][i] = col_scale(colmax_dy[o0 + dc[e] + i]);
  }
  // SHIFT (RowShift, T >= KB): the staged row's (sample, step) is carried from chunk to chunk — a division per staged quad
  // cost cfg5's gradient 1.7 ms, and any branch in here splits the loop body the MFMA stream is scheduled in (1.3 ms): selects
  // only.  Call k of stage_load stages rows r0 + k KB + xr[e] (the calls past the last chunk repeat it and are discarded: any
  // valid address)
  unsigned sh_n[2] = {0, 0}, sh_t[2] = {0, 0}, sh_b[2] = {0, 0};
  if constexpr (SHIFT) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      sh_n[e] = (unsigned)(r0 + xr[e]);
      sh_b[e] = sh_n[e] / (unsigned)rsh.T;
      sh_t[e] = sh_n[e] - sh_b[e] * (unsigned)rsh.T;
    }
  }
  // (ABL: the harness instantiation, tools/wgrad_bench — option `dev` 8 = no MFMAs, 16 = no split / LDS stores, 32 = no loads,
  // 64 = no fragment reads; the product instantiation has none of these branches)
  const int dv = ABL ? dev : 0;
  auto stage_load = [&](int64_t nb) {
    if constexpr (ABL) { if (dv & 32) return; }
    if constexpr (SHIFT) {
      const TS* fp = reinterpret_cast<const TS*>(rsh.first);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int col = xin[e] ? j0 + xc[e] : 0;
        const bool inr = (int64_t)sh_n[e] < r1, head = inr && sh_t[e] == 0;
        // row n - 1 of x; a head row takes its sample's initial state, or reads row n (any valid address) and is zeroed
        const unsigned row = !inr ? (unsigned)(r1 - 1) : head ? sh_n[e] : sh_n[e] - 1;
        const TS* src = (head && fp) ? fp + (size_t)sh_b[e] * IN : x + (size_t)row * IN;
        const f32x4 v = ld4(src, (size_t)col);
        sx[e] = (head && !fp) ? f32x4{0.f, 0.f, 0.f, 0.f} : v;
        sh_n[e] += KB;
        sh_t[e] += KB;
        const bool wrap = sh_t[e] >= (unsigned)rsh.T;
        sh_t[e] -= wrap ? (unsigned)rsh.T : 0u;
        sh_b[e] += wrap ? 1u : 0u;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int64_t n = nb + xr[e] < r1 ? nb + xr[e] : r1 - 1;          // unconditional loads; masked at the store
        sx[e] = ld4(x, (size_t)n * IN + (xin[e] ? j0 + xc[e] : 0));
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t n = nb + dr[e] < r1 ? nb + dr[e] : r1 - 1;
      sd[e] = ld4(dy, (size_t)n * OUT + o0 + dc[e]);
    }
  };
  auto store4 = [&](__bf16* plane0, int row, int col, const f32x4& v) {   // col: 0..127, multiple of 4
    char* base = reinterpret_cast<char*>(plane0) + tr_off(row, col >> 3) + 8 * ((col >> 2) & 1);
    if constexpr (HALF) {
      unsigned a0, b0, a1, b1;
      split_pair_h(v[0], v[1], a0, b0);
      split_pair_h(v[2], v[3], a1, b1);
      *reinterpret_cast<u32x2*>(base) = u32x2{a0, a1};
      *reinterpret_cast<u32x2*>(base + PL * 2) = u32x2{b0, b1};
    } else {
      unsigned a0, b0, c0, a1, b1, c1;
      split_pair(v[0], v[1], a0, b0, c0);
      split_pair(v[2], v[3], a1, b1, c1);
      *reinterpret_cast<u32x2*>(base) = u32x2{a0, a1};
      *reinterpret_cast<u32x2*>(base + PL * 2) = u32x2{b0, b1};
      *reinterpret_cast<u32x2*>(base + 2 * PL * 2) = u32x2{c0, c1};
    }
  };
  auto stage_store = [&](int buf, int64_t nb) {
    if (ABL && (dv & 16)) {
      // keep the loaded registers alive (a dependence on every one of them), skip the split and the LDS stores
      float t = 0.f;
#pragma unroll
      for (int e = 0; e < 2; ++e) t += sx[e][0] + sx[e][3];
#pragma unroll
      for (int e = 0; e < 4; ++e) t += sd[e][0] + sd[e][3];
      dbs[0][0] += t;
      return;
    }
    __bf16* xs = ldsb + buf * BUFE;                        // [NP][32][128]
    __bf16* ds0 = xs + NP * PL;                            // columns 0..127:   [NP][32][128]
    __bf16* ds1 = ds0 + NP * PL;                           // columns 128..255: [NP][32][128]
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float keep = (nb + xr[e] < r1 && xin[e]) ? 1.0f : 0.0f;
      if constexpr (HALF) store4(xs, xr[e], xc[e], sx[e] * scx[e] * keep);
      else store4(xs, xr[e], xc[e], sx[e] * keep);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float keep = nb + dr[e] < r1 ? 1.0f : 0.0f;
      const f32x4 vd = sd[e] * keep;
      if constexpr (HALF) store4(dc[e] < 128 ? ds0 : ds1, dr[e], dc[e] & 127, vd * scd[e]);
      else store4(dc[e] < 128 ? ds0 : ds1, dr[e], dc[e] & 127, vd);
      dbs[e] += vd;
    }
  };

  const int64_t chunks = (r1 - r0 + KB - 1) / KB;
  stage_load(r0);
  stage_store(0, r0);
  stage_load(r0 + (1 < chunks ? 1 : 0) * KB);
  for (int64_t ch = 0; ch < chunks; ++ch) {
    const int buf = (int)(ch & 1);
    lds_barrier();                                         // chunk ch is in `buf`; nobody reads the other buffer any more
    const __bf16* xs = ldsb + buf * BUFE;
    const __bf16* dsw = xs + NP * PL + (wn >> 1) * NP * PL;  // this wave's column half of dy
    xbf8 af[4][NP];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int p = 0; p < NP; ++p) af[mi][p] = (ABL && (dv & 64)) ? xbf8{} : tr_frag(xs + p * PL, wm * 4 + mi, lane);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      xbf8 bf[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) bf[p] = (ABL && (dv & 64)) ? xbf8{} : tr_frag(dsw + p * PL, (wn & 1) * 4 + ni, lane);
      if (ni == 1) {                                        // the next chunk's split + stores ride inside the MFMA stream
        stage_store(buf ^ 1, r0 + (ch + 1) * KB);           // (past the end: a fully masked chunk into the idle buffer)
        stage_load(r0 + (ch + 2 < chunks ? ch + 2 : ch + 1 < chunks ? ch + 1 : ch) * KB);
      }
      if constexpr (ABL) { if (dv & 8) continue; }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        if constexpr (HALF) {
          const xh8 a0 = __builtin_bit_cast(xh8, af[mi][0]), a1 = __builtin_bit_cast(xh8, af[mi][1]);
          const xh8 b0 = __builtin_bit_cast(xh8, bf[0]), b1 = __builtin_bit_cast(xh8, bf[1]);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[mi][ni], 0, 0, 0);
        } else {
#pragma unroll
          for (int s = 0; s < 6; ++s)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][SPLIT_TW[s]], bf[SPLIT_TX[s]], acc[mi][ni], 0, 0, 0);
        }
      }
    }
  }
  float uny[4] = {1.f, 1.f, 1.f, 1.f};
  if constexpr (HALF) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) uny[ni] = col_unscale(colmax_dy[o0 + wn * 64 + 16 * ni + c]);
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int jj = j0 + wm * 64 + 16 * mi + 4 * q + j;
      if (jj < IN) {
        float unx = 1.f;
        if constexpr (HALF) unx = col_unscale(colmax_x[jj]);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const size_t e = (size_t)jj * OUT + o0 + wn * 64 + 16 * ni + c;
          const float v = HALF ? unscale2(acc[mi][ni][j], unx, uny[ni]) : acc[mi][ni][j];
          if (KS == 1) dW[e] = v;
          else if (part) part[(size_t)ks * IN * OUT + e] = v;      // summed by k_dense_reduce
          else atomicAdd(dW + e, v);
        }
      }
    }
  if (want_bias)      // a thread stages the same column quad in all four of its rows; eight waves = eight groups over 256 columns
    bias_partial<DenseS::TO, 8>(reinterpret_cast<float*>(smem2), (dbs[0] + dbs[1]) + (dbs[2] + dbs[3]), wave, dc[0],
                                bpart + (size_t)ks * OUT + o0, tid);
}

// dW[e] = sum over the KS row ranges of part[ks][e] (row ranges past the end of the rows are not summed).  32 element quads x 8
// range groups per workgroup, the groups' sums added in a fixed order through LDS (one thread per quad walking all KS ranges
// left a small matrix — in = 64: 16 workgroups — at 0.4 TB/s: 37 us for 16 MB)
__global__ void __launch_bounds__(256) k_dense_reduce(const float* __restrict__ part, int KS, size_t n4,
                                                      float* __restrict__ dW) {
  __shared__ f32x4 red[8][32];
  const int c = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const size_t e = (size_t)blockIdx.x * 32 + c;
  const int per = (KS + 7) / 8, k0 = kg * per, k1 = k0 + per < KS ? k0 + per : KS;
  const f32x4* p = reinterpret_cast<const f32x4*>(part);
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e < n4)
    for (int k = k0; k < k1; ++k) v += p[(size_t)k * n4 + e];
  red[kg][c] = v;
  __syncthreads();
  if (kg == 0 && e < n4) {
    f32x4 sum = red[0][c];
#pragma unroll
    for (int g = 1; g < 8; ++g) sum += red[g][c];
    reinterpret_cast<f32x4*>(dW)[e] = sum;
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
static size_t al256g(size_t v) { return (v + 255) & ~(size_t)255; }

bool gemm_split_ok(int K, int M) { return K >= 8 && K % 8 == 0 && M % GT == 0; }

size_t gemm_split_identity_bytes(int K) { return al256g((size_t)K * K * sizeof(float)); }
size_t gemm_split_dense_bytes(int K, int M) { return al256g((size_t)K * M * sizeof(float)); }
static int gemm_chunks(int K) { return ((K + 2 * GK - 1) / (2 * GK)) * 2; }      // even: the k loop is unrolled by two
size_t gemm_split_plane_bytes(int K, int M) { return al256g((size_t)3 * gemm_chunks(K) * GK * M * sizeof(__bf16)); }

int launch_fill_identity(int dtype, int K, void* id, hipStream_t stream) {
  const int grid = (int)(((size_t)K * K + 255) / 256);
  if (dtype == TTRNN_F32) hipLaunchKernelGGL(k_fill_identity<float>, dim3(grid), dim3(256), 0, stream, (float*)id, K);
  else hipLaunchKernelGGL(k_fill_identity<bf16_t>, dim3(grid), dim3(256), 0, stream, (bf16_t*)id, K);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// transposed: W(k, m) = WG[m][k] (WG is [M][K] row-major) instead of WG[k][m]
int launch_gemm_split_prep(const float* WG, int K, int M, void* planes, hipStream_t stream, bool transposed) {
  const int KCn = gemm_chunks(K);
  const size_t threads = (size_t)KCn * 8 * M;
  hipLaunchKernelGGL(k_gemm_prep, dim3((int)((threads + 255) / 256)), dim3(256), 0, stream, WG, K, KCn, M,
                     (int64_t)(transposed ? 1 : M), (int64_t)(transposed ? K : 1), (__bf16*)planes);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

bool gemm_use_half(int64_t n_rows, int K, int M) {
  const int pieces = opt(OPT_GEMM_PIECES);
  if (pieces == 2) return true;
  if (pieces == 3) return false;
  return (double)n_rows * K * M >= 4294967296.0;
}

// two-piece fp16 variant: scratch = one fp32 scale per column of W (M, padded) + one per row of x (+ the pre-split pieces of
// x where the pre-split GEMM of ttrnn_fast_gemm3.hip takes the launch)
static size_t half_scales_bytes(int64_t n_rows, int M) {
  return al256g(g_rs_off(M) * sizeof(float) + (size_t)(n_rows > 0 ? n_rows : 0) * sizeof(float));
}
size_t gemm_half_scratch_bytes(int64_t n_rows, int K, int M) {
  return half_scales_bytes(n_rows, M) + (gemm3_ok(n_rows, K, M) ? gemm3_xplane_bytes(n_rows, K) : 0);
}

int launch_gemm_half_prep(const float* WG, int K, int M, void* planes, void* scratch, hipStream_t stream, bool transposed) {
  const int KCn = gemm_chunks(K);
  const size_t threads = (size_t)KCn * 8 * M;
  hipLaunchKernelGGL(k_gemm_col_scales, dim3((M + 31) / 32), dim3(256), 0, stream, WG, K, M,
                     (int64_t)(transposed ? 1 : M), (int64_t)(transposed ? K : 1), (float*)scratch);
  hipLaunchKernelGGL(k_gemm_prep_h, dim3((int)((threads + 255) / 256)), dim3(256), 0, stream, WG, K, KCn, M,
                     (int64_t)(transposed ? 1 : M), (int64_t)(transposed ? K : 1), (const float*)scratch,
                     (_Float16*)planes);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

template <typename TS, bool HALF = false, int FW = 1>
static int launch_gemm_t(int64_t n_rows, int K, int M, const void* x, const void* planes, const void* bias, int Hb,
                         float* y, hipStream_t stream, const float* bias_ilv, const float* scratch = nullptr) {
  constexpr size_t lds = HALF ? (size_t)2 * 2 * (GPA + FW * GPB) * sizeof(_Float16) : G_LDS;
  {
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_gemm_split<TS, HALF, FW>), lds) != TTRNN_OK)
      return TTRNN_ERR_LAUNCH;
  }
  const int KCn = gemm_chunks(K);
  const int MT = M / (GT * FW);
  const int64_t RT = (n_rows + GR - 1) / GR;
  int64_t grid;
  if (MT % 8 == 0) {
    const int64_t supers = (int64_t)(MT / 8) * ((RT + 3) / 4);
    grid = ((supers + 7) / 8) * 8 * 32;
  } else {
    grid = (int64_t)MT * RT;
  }
  hipLaunchKernelGGL((k_gemm_split<TS, HALF, FW>), dim3((unsigned)grid), dim3(FAST_NT), lds, stream, n_rows,
                     K, KCn, M, (const TS*)x, planes, (const TS*)bias, Hb, y, bias_ilv, scratch);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// 256-feature workgroup tiles where the matrix is wide enough to keep every CU busy with them
static bool gemm_wide_tiles(int64_t n_rows, int M) { return M % (2 * GT) == 0 && (M / (2 * GT)) * ((n_rows + GR - 1) / GR) >= 1024; }

// rs[n] = 2^(14 - e), bound[n] < 2^e: the row scales from maxima somebody else measured (the reverse-time kernel's by-product)
__global__ void __launch_bounds__(256) k_rowmax_scales(const float* __restrict__ bound, int64_t n_rows, float* __restrict__ rs) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < n_rows) rs[n] = ldexpf(1.f, 14 - g_expo(bound[n]));
}

int launch_gemm_half(int dtype, int64_t n_rows, int K, int M, const void* x, const void* planes, void* scratch,
                     const void* bias, int Hb, float* y, hipStream_t stream, const float* bias_ilv, const float* rowmax) {
  if (n_rows <= 0) return TTRNN_OK;
  if (gemm3_ok(n_rows, K, M))       // x split once into fp16 planes, both operands by LDS-DMA (ttrnn_fast_gemm3.hip)
    return launch_gemm3h(dtype, n_rows, K, M, x, planes, scratch, (char*)scratch + half_scales_bytes(n_rows, M), bias, Hb, y,
                         stream, bias_ilv);
  float* rs = (float*)scratch + g_rs_off(M);
  const int grid = (int)((n_rows + 3) / 4 < 2048 ? (n_rows + 3) / 4 : 2048);
  if (rowmax && dtype == TTRNN_F32) {
    hipLaunchKernelGGL(k_rowmax_scales, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream, rowmax, n_rows, rs);
    return gemm_wide_tiles(n_rows, M)
               ? launch_gemm_t<float, true, 2>(n_rows, K, M, x, planes, bias, Hb, y, stream, bias_ilv, (const float*)scratch)
               : launch_gemm_t<float, true>(n_rows, K, M, x, planes, bias, Hb, y, stream, bias_ilv, (const float*)scratch);
  }
  if (dtype == TTRNN_F32) {
    hipLaunchKernelGGL(k_row_scales<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, n_rows, K, rs);
    return gemm_wide_tiles(n_rows, M)
               ? launch_gemm_t<float, true, 2>(n_rows, K, M, x, planes, bias, Hb, y, stream, bias_ilv, (const float*)scratch)
               : launch_gemm_t<float, true>(n_rows, K, M, x, planes, bias, Hb, y, stream, bias_ilv, (const float*)scratch);
  }
  hipLaunchKernelGGL(k_row_scales<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, n_rows, K, rs);
  return gemm_wide_tiles(n_rows, M)
             ? launch_gemm_t<bf16_t, true, 2>(n_rows, K, M, x, planes, bias, Hb, y, stream, bias_ilv, (const float*)scratch)
             : launch_gemm_t<bf16_t, true>(n_rows, K, M, x, planes, bias, Hb, y, stream, bias_ilv, (const float*)scratch);
}

int launch_gemm_split(int dtype, int64_t n_rows, int K, int M, const void* x, const void* planes, const void* bias,
                      int Hb, float* y, hipStream_t stream, const float* bias_ilv) {
  if (n_rows <= 0) return TTRNN_OK;
  return dtype == TTRNN_F32 ? launch_gemm_t<float>(n_rows, K, M, x, planes, bias, Hb, y, stream, bias_ilv)
                            : launch_gemm_t<bf16_t>(n_rows, K, M, x, planes, bias, Hb, y, stream, bias_ilv);
}


// can launch_dense_wgrad read shifted rows (shift_T) for this many rows?
bool dense_wgrad_shift_ok(int64_t n_rows, int64_t shift_T) {
  return n_rows < ((int64_t)1 << 31) && shift_T >= DenseS::KB && shift_T < ((int64_t)1 << 31);
}
bool dense_wgrad_ok(int in, int out) { return in >= 4 && in % 4 == 0 && out >= 4 && out % 4 == 0; }      // (fp32-MFMA kernel: any such width)

// dW (fp32 [in][out]) = x^T dy over n_rows rows (overwritten); d_bias (may be NULL) is accumulated into.
// split: three-way bf16 splits on the bf16 MFMA (needs out % 256 == 0); otherwise the fp32 MFMA.
static size_t dense_colmax_bytes(int in, int out) { return al256g((size_t)(in + out) * sizeof(unsigned)); }
// partial column sums of dy per row range (at most one range per CU, rounded up to the XCD multiple)
// Row ranges (split-K) of the dense weight gradient for `tiles` output tiles: a multiple of 8 (one row range per XCD at a time, its
// tiles side by side on that XCD's L2).  Where the chip takes the tiles x ranges in ONE round, the largest such count; where it
// does not — 72 tiles (768 x 3072: H = 768 hidden matrices) x 8 ranges = 576 workgroups ran as three rounds, the third a quarter
// full — the count with the fewest rounds per range, a small charge per range for its partial tile and the reduction: 32 ranges
// there (nine rounds of a quarter of the work: 2.16 -> ~1.7 ms), 24 for 18 tiles (384 x 1536).  (dev bit 28: the old rule.)
static int dense_ks(int tiles, int cus) {
  const int up = ((cus / tiles + 7) / 8) * 8, dn = (cus / tiles) / 8 * 8;
  const int old_rule = (tiles * up <= cus || dn < 8) ? up : dn;
  if (tiles * up <= cus || (opt(OPT_DEV) & (1 << 28))) return old_rule;
  const int per_xcd = cus / 8 > 0 ? cus / 8 : 1;
  int best = old_rule;
  double bc = 1e30;
  for (int ks = 8; ks <= 64; ks += 8) {
    const int rounds = (tiles * (ks / 8) + per_xcd - 1) / per_xcd;
    const double cst = (double)rounds / ks + 0.0005 * ks;
    if (cst < bc) { bc = cst; best = ks; }
  }
  return best;
}

static size_t dense_bias_part_bytes(int out) { return al256g((size_t)((device_cu_count() + 7) / 8 * 8) * out * sizeof(float)); }

// two fp16 pieces (three terms) where the column-maximum passes over x and dy pay off, else three bf16 pieces
static bool dense_wgrad_use_half(int64_t n_rows, int in, int out) {
  const int pieces = opt(OPT_GEMM_PIECES);
  if (pieces == 2) return true;
  if (pieces == 3) return false;
  return (double)n_rows * in * out >= 68719476736.0;      // 2^36 multiply-adds
}

int launch_dense_wgrad(int dtype, int64_t n_rows, int in, int out, const void* x, const float* dy, float* dW,
                       float* d_bias, hipStream_t stream, bool split, float* scratch_all, const unsigned* x_colmax,
                       const unsigned* dy_colmax, int shift_T, const void* shift_first) {
  const int cus = device_cu_count();
  if (shift_T > 0 && !dense_wgrad_shift_ok(n_rows, shift_T)) return TTRNN_ERR_UNSUPPORTED;
  RowShift rsh;
  rsh.T = shift_T > 0 ? shift_T : 0;
  rsh.first = shift_T > 0 ? shift_first : nullptr;
  split = split && out % DenseS::TO == 0 && !opt(OPT_DENSE_FP32);      // A/B switch: dense gradient on the fp32 MFMA
  // scratch: [column maxima of x and dy (HALF) | partial tiles of the row ranges | partial column sums (bias) of the row ranges]
  if (d_bias && !scratch_all) return TTRNN_ERR_WORKSPACE;
  float* bpart = d_bias ? (float*)((char*)scratch_all + dense_wgrad_scratch_bytes(in, out) - dense_bias_part_bytes(out)) : nullptr;
  unsigned* colmax = (unsigned*)scratch_all;
  float* scratch = scratch_all ? (float*)((char*)scratch_all + dense_colmax_bytes(in, out)) : nullptr;
  // the producer's bounds for dy's columns make the two-piece variant free of its 4-bytes-per-element pass over dy: taken
  // at every size then (x's pass, where its bounds are missing, reads in / out of that)
  const bool half = split && colmax && n_rows > 0 &&
                    (dense_wgrad_use_half(n_rows, in, out) || (dy_colmax && opt(OPT_GEMM_PIECES) != 3));
  const unsigned* cx = colmax;
  const unsigned* cd = colmax ? colmax + in : nullptr;
  if (half) {
    const unsigned gy = (unsigned)((n_rows + colmax_rows(in) - 1) / colmax_rows(in));
    const unsigned gyo = (unsigned)((n_rows + colmax_rows(out) - 1) / colmax_rows(out));
    if (x_colmax) {
      cx = x_colmax;
    } else {
      if (hipMemsetAsync(colmax, 0, (size_t)in * sizeof(unsigned), stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
      if (dtype == TTRNN_F32)
        hipLaunchKernelGGL(k_col_absmax<float>, dim3((in / 4 + 255) / 256, gy), dim3(256), 0, stream, (const float*)x, n_rows, in,
                           colmax);
      else
        hipLaunchKernelGGL(k_col_absmax<bf16_t>, dim3((in / 4 + 255) / 256, gy), dim3(256), 0, stream, (const bf16_t*)x, n_rows,
                           in, colmax);
      if (rsh.T > 0 && rsh.first) {      // shifted rows: the maxima over `out` bound rows 1.., the initial states are rows 0
        const int64_t nb = n_rows / rsh.T;
        const unsigned gb = (unsigned)((nb + colmax_rows(in) - 1) / colmax_rows(in));
        if (dtype == TTRNN_F32)
          hipLaunchKernelGGL(k_col_absmax<float>, dim3((in / 4 + 255) / 256, gb), dim3(256), 0, stream, (const float*)rsh.first,
                             nb, in, colmax);
        else
          hipLaunchKernelGGL(k_col_absmax<bf16_t>, dim3((in / 4 + 255) / 256, gb), dim3(256), 0, stream,
                             (const bf16_t*)rsh.first, nb, in, colmax);
      }
    }
    if (dy_colmax) {
      cd = dy_colmax;
    } else {
      if (hipMemsetAsync(colmax + in, 0, (size_t)out * sizeof(unsigned), stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
      hipLaunchKernelGGL(k_col_absmax<float>, dim3((out / 4 + 255) / 256, gyo), dim3(256), 0, stream, dy, n_rows, out,
                         colmax + in);
    }
  }
  const int KBc = split ? DenseS::KB : DenseG::KB;
  const int tiles = split ? ((in + DenseS::TJ - 1) / DenseS::TJ) * (out / DenseS::TO)
                          : ((in + DenseG::TM - 1) / DenseG::TM) * ((out + DenseG::TN - 1) / DenseG::TN);
  int KS = 1;
  if (tiles < cus) {
    // multiple of 8: one row range per XCD at a time — rounded DOWN where rounding up would not fit the chip in one go
    // (cfg3: 6 tiles x 48 ranges = 288 workgroups on 256 CUs ran as two rounds, 658 us; 40 ranges: one round)
    KS = dense_ks(tiles, cus);
    const int64_t max_ks = (n_rows + KBc - 1) / KBc;                // at least one chunk per split
    if (KS > max_ks) KS = max_ks < 1 ? 1 : (int)max_ks;
  }
  int64_t rows_per = (n_rows + KS - 1) / KS;
  rows_per = (rows_per + KBc - 1) / KBc * KBc;
  KS = (int)((n_rows + rows_per - 1) / rows_per);                   // no empty row range
  // KS > 1: every row range writes its own partial tile into `scratch` (summed afterwards); atomics without scratch
  float* part = KS > 1 ? scratch : nullptr;
  if (KS > 1 && !part && hipMemsetAsync(dW, 0, (size_t)in * out * sizeof(float), stream) != hipSuccess)
    return TTRNN_ERR_LAUNCH;
  const int di = (dtype == TTRNN_F32 ? 0 : 1) + (split ? 2 : 0) + (half ? 2 : 0);
  const size_t lds = half ? DenseS::LDS_BYTES_H : split ? DenseS::LDS_BYTES : DenseG::LDS_BYTES;
  const unsigned grid = (unsigned)(tiles * KS);
#define TT_WG(TSV, HV, SV)                                                                                                    \
  do {                                                                                                                        \
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_dense_wgrad_split<TSV, HV, SV>), lds) != TTRNN_OK)                  \
      return TTRNN_ERR_LAUNCH;                                                                                                \
    hipLaunchKernelGGL((k_dense_wgrad_split<TSV, HV, SV>), dim3(grid), dim3(FAST_NT), lds, stream, n_rows, in, out, KS,       \
                       rows_per, (const TSV*)x, dy, dW, bpart, part, cx, cd, rsh, 0);                                         \
  } while (0)
#define TT_WG2(TSV, HV) do { if (rsh.T > 0) TT_WG(TSV, HV, true); else TT_WG(TSV, HV, false); } while (0)
#ifdef TTRNN_ABLATIONS      // harness build only (`make ablation`): the no-MFMA / no-split / no-load / no-fragment-read instantiation
  if (di == 4 && rsh.T == 0 && (opt(OPT_DEV) & (8 | 16 | 32 | 64))) {      // tools/wgrad_bench
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_dense_wgrad_split<float, true, false, true>), lds) != TTRNN_OK)
      return TTRNN_ERR_LAUNCH;
    hipLaunchKernelGGL((k_dense_wgrad_split<float, true, false, true>), dim3(grid), dim3(FAST_NT), lds, stream, n_rows, in, out,
                       KS, rows_per, (const float*)x, dy, dW, bpart, part, cx, cd, rsh, opt(OPT_DEV));
  } else
#endif
  switch (di) {
    case 0:
      if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_dense_wgrad<float>), lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
      hipLaunchKernelGGL(k_dense_wgrad<float>, dim3(grid), dim3(FAST_NT), lds, stream, n_rows, in, out, KS, rows_per,
                         (const float*)x, dy, dW, bpart, part, rsh);
      break;
    case 1:
      if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_dense_wgrad<bf16_t>), lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
      hipLaunchKernelGGL(k_dense_wgrad<bf16_t>, dim3(grid), dim3(FAST_NT), lds, stream, n_rows, in, out, KS, rows_per,
                         (const bf16_t*)x, dy, dW, bpart, part, rsh);
      break;
    case 2: TT_WG2(float, false); break;
    case 3: TT_WG2(bf16_t, false); break;
    case 4: TT_WG2(float, true); break;
    default: TT_WG2(bf16_t, true);
  }
#undef TT_WG2
#undef TT_WG
  if (part) {
    const size_t n4 = (size_t)in * out / 4;
    hipLaunchKernelGGL(k_dense_reduce, dim3((unsigned)((n4 + 31) / 32)), dim3(256), 0, stream, part, KS, n4, dW);
  }
  if (bpart)
    hipLaunchKernelGGL(k_dense_bias_reduce, dim3((unsigned)((out + 31) / 32)), dim3(256), 0, stream, bpart, KS, out, d_bias);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// bytes of the scratch launch_dense_wgrad wants for this shape: column maxima (two-piece fp16 variant) + the partial tiles
// of the row ranges (none when one workgroup per tile covers all rows)
size_t dense_wgrad_scratch_bytes(int in, int out) {
  const int cus = device_cu_count();
  const int tiles_s = ((in + DenseS::TJ - 1) / DenseS::TJ) * (out / DenseS::TO > 0 ? out / DenseS::TO : 1);
  const int tiles_g = ((in + DenseG::TM - 1) / DenseG::TM) * ((out + DenseG::TN - 1) / DenseG::TN);
  const int tiles = tiles_s < tiles_g ? tiles_s : tiles_g;
  if (tiles >= cus) return dense_colmax_bytes(in, out) + dense_bias_part_bytes(out);
  int KS = ((cus / tiles + 7) / 8) * 8;
  if (tiles_s < cus && dense_ks(tiles_s, cus) > KS) KS = dense_ks(tiles_s, cus);
  if (tiles_g < cus && dense_ks(tiles_g, cus) > KS) KS = dense_ks(tiles_g, cus);
  return dense_colmax_bytes(in, out) + al256g((size_t)KS * in * out * sizeof(float)) + dense_bias_part_bytes(out);
}

}  // namespace ttrnn
