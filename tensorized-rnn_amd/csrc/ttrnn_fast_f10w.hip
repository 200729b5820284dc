// ttrnn_fast_f10w.hip — batched weight gradients of a hidden-shaped TT-matrix through the fused core (gfx950, fp32).
//
// dL/dG_k of y_n = TT(G) x_n over n = B*T rows (autograd of t3nsor/ops.py:78-93 for `hidden_weights`: x_n = h_{t-1},
// dy_n = the gate gradients of the reverse-time kernel; dx is NOT produced here — dh travels inside that kernel).
// The stage-wise kernel (ttrnn_fast_bwd.hip) recomputes the three forward stages, runs two transposed stages and three
// weight-gradient GEMMs per row: 1.83 MFLOP per row on the fp32 MFMA.  With cores 1 and 0 contracted (ttrnn_f10.h)
//     y[m][i2] = sum_k W10[k][m] * C2[i2][k],      C2 = S2(x)        (k = (row2, r2), m = (i0, i1))
// the gradients factor as
//     dW10[k][m]        += sum_i2   C2[i2][k] * dy[m][i2]                       (256 x 64, contraction 16 per row)
//     dW2[j2][(i2,r2)]  += sum_row2 x[row2][j2] * dC2[row2][(i2,r2)],   dC2 = W10 dy   (8 x 128, contraction 32 per row)
// accumulated over all rows in MFMA accumulators that stay in registers for the whole launch (32 + 4 VGPRs per lane),
// and ONE small kernel at the end turns dW10 into dG0 and dG1 by the product rule (W10 = sum_r1 G0 * G1).
// Per row: S2 (fp32 MFMA, 4 per wave), dC2 = T01 on split-bf16 MFMAs with the fragments of the reverse-time kernel
// (24 per wave), dW10 (32 fp32 MFMAs per wave), dW2 (8 per wave): 0.66 MFLOP on the fp32 pipe + 0.52 on the bf16 pipe.
// The weight-gradient GEMMs contract over the index that is the MFMA *column* of their producers, which the fp32
// MFMA's one-value-per-lane operands read straight from (padded, conflict-free) fp32 LDS images; their operands are
// therefore not split.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_f10.h"

namespace ttrnn {

template <class S>
struct F10W {
  using F = F10<S>;
  static constexpr int IN = F::H, OUT = out_size_of<S>();
  static constexpr int H = 256;                            // hidden units of the cell the matrix feeds
  static constexpr int NG = OUT / H;                       // gates: 4 (LSTM) or 3 (GRU)
  static constexpr int XI = F::ROWS2 * 8;                  // x image [ROWS2][8] (j2 zero-padded to the fp32 MFMA's k)
  static constexpr int K1 = F::M, NM1 = K1 / 32;          // T01 contraction (m) and its k-blocks
  static constexpr int FT = F::K / 16, XF = FT / FAST_NW;  // T01 / dW10 feature tiles (k = (row2, r2)), per wave
  static constexpr int MT = F::M / 16;                     // dW10 column tiles (m)
  static constexpr int K2 = F::I2 * F::R2;                 // (i2, r2): columns of dC2 / dW2
  static constexpr int CT2 = K2 / 16, XC = (CT2 + FAST_NW - 1) / FAST_NW;  // dW2 column tiles, per wave
  static constexpr int PL1 = F::I2 * K1;                   // bf16 elements per plane of the T01 operand
  static constexpr int DGS = K1 + 16;                      // row stride of dgT  [I2][m]       (fp32, +16: bank shift)
  static constexpr int C2S = F::K + 16;                    // row stride of C2   [I2][k]
  // row stride of dC2 [ROWS2][(i2,r2)]: read with lanes along the columns (dW2) AND with lanes along the rows (dx);
  // +4 keeps 16-byte alignment and leaves both patterns at most 2-way conflicted
  static constexpr int DCS = K2 + 4;
  // dx = G2^T dC2 (only for matrices whose input is another layer's output): wave = (column tile, quarter of the
  // (i2,r2) contraction), partial sums through LDS
  static constexpr int KQ = K2 / 4 / 4;                    // k-steps per quarter
};

// LSTM shapes (gates aligned with the m index) load a thread's four gate gradients as 4 consecutive k of the T01
// operand; any other cell (GRU: I2 = 12) goes element by element with the natural k order (NATK)
template <class S>
constexpr bool f10w_natk() { return !(f10_ok<S>() || f10_in_ok<S>()); }
template <class S>
constexpr bool NEEDS_HIDDEN_DX() { return F10<S>::J2 == 8 && F10<S>::ROWS2 == 32; }   // dx only for hidden-shaped inputs

template <class S>
constexpr bool f10w_ok() {
  using F = F10<S>;
  using W = F10W<S>;
  return S::D == 3 && S::R[0] == 1 && S::R[3] == 1 && shape_ok<S>() && F::I2 % 4 == 0 && F::I2 <= 16 &&
         F::K % 32 == 0 && W::K1 % 32 == 0 && W::FT % FAST_NW == 0 && W::MT == 4 && F::J2 <= 8 &&
         St<S, 2>::KP == 8 && !St<S, 2>::SPLIT && F::ROWS2 % 4 == 0 && F::ROWS2 <= 32 && W::K2 % 16 == 0 &&
         W::OUT % W::H == 0 && (f10_ok<S>() || f10_in_ok<S>() || W::OUT <= 2 * FAST_NT) && S::R[2] % 4 == 0 &&
         W::XI <= 4 * FAST_NT;
}

template <class S>
constexpr size_t f10w_lds_bytes() {
  using F = F10<S>;
  using W = F10W<S>;
  return sizeof(float) * (W::XI + F::I2 * W::DGS + F::I2 * W::C2S + F::ROWS2 * W::DCS + 4 * W::H) +
         2 * 3 * (size_t)W::PL1;
}

// wfrag: the T01 fragments of k_f10b_prep (ttrnn_fast_f10b.hip);  dW10: fp32 [K][M] accumulation buffer (zeroed);
// NEED_DX: also dx[n] = W^T dy[n] (fp32 rows)
template <class S, typename TI, bool NEED_DX>
__global__ void __launch_bounds__(FAST_NT) k_ttlinear_wgrad_f10(long n_rows, const float* __restrict__ packed,
                                                                const xbf8* __restrict__ wfrag,
                                                                const TI* __restrict__ x,
                                                                const float* __restrict__ dy, TI* __restrict__ dx,
                                                                float* __restrict__ dW10,
                                                                float* __restrict__ d_packed,
                                                                float* __restrict__ d_bias) {
  static_assert(f10w_ok<S>(), "shape not supported by the fused-core weight-gradient kernel");
  using F = F10<S>;
  using W = F10W<S>;
  constexpr int H = W::H, IN = W::IN, OUT = W::OUT, XI = W::XI;
  constexpr bool DENSE = F::J2 == 8;                          // x row == the [ROWS2][8] image (hidden-shaped input)
  static_assert(!NEED_DX || NEEDS_HIDDEN_DX<S>(), "dx is only produced for hidden-shaped inputs");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* ximg = reinterpret_cast<float*>(smem);               // x row as [ROWS2][8] (j2 zero-padded)
  float* dgT = ximg + XI;                                     // dy transposed: [i2][m]
  float* c2i = dgT + F::I2 * W::DGS;                          // C2 = S2(x): [i2][k]
  float* dci = c2i + F::I2 * W::C2S;                          // dC2 = W10 dy: [row2][(i2,r2)]
  float* dxs = dci + F::ROWS2 * W::DCS;                       // NEED_DX: four partial-sum slices of dx [4][H]
  __bf16* img1 = reinterpret_cast<__bf16*>(dxs + 4 * H);      // split dy, T01 operand: 3 planes [i2][k1]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;

  float w2[nwreg<S, 2>()];
  load_wfrag<S, 2>(w2, packed, wave, lane);
  xbf8 w01[W::XF][W::NM1][3];
#pragma unroll
  for (int xx = 0; xx < W::XF; ++xx)
#pragma unroll
    for (int u = 0; u < W::NM1; ++u)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        w01[xx][u][p] = wfrag[(size_t)(((wave + FAST_NW * xx) * W::NM1 + u) * 3 + p) * 64 + lane];

  // accumulators: dW10 tiles (k-tile wave + 8*xx, m-tile mt) and the dW2 tiles of the column tiles wave + 8*xc
  f32x4 g10[W::XF][W::MT], g2[W::XC];
#pragma unroll
  for (int xx = 0; xx < W::XF; ++xx)
#pragma unroll
    for (int mt = 0; mt < W::MT; ++mt) g10[xx][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int xc = 0; xc < W::XC; ++xc) g2[xc] = f32x4{0.f, 0.f, 0.f, 0.f};
  // NEED_DX: G2 fragments of this wave's quarter of the (i2,r2) contraction: row j2 = c, one value per k-step
  float wdx[NEED_DX ? W::KQ : 1];
  const int dct = wave & 1, dkq = wave >> 1;                 // dx: column tile (row2 half), contraction quarter
  if constexpr (NEED_DX) {
    const float* W2 = packed + woff_of<S>(2);                 // [J2][M2 = (i2,r2)]
#pragma unroll
    for (int s = 0; s < W::KQ; ++s) wdx[s] = c < F::J2 ? W2[c * F::M2 + 4 * (dkq * W::KQ + s) + q] : 0.f;
  }
  constexpr bool NATK = f10w_natk<S>();
  float dbias[4] = {0.f, 0.f, 0.f, 0.f};

  // loads.  LSTM: thread tid < H owns hidden unit tid (its 4 gate gradients, one strided dword each).  NATK: thread
  // tid owns the flat outputs tid and tid + 512.  Threads tid < H/4 also carry 4 values of x.
  const bool own = tid < H;
  const int hid = own ? tid : 0;
  const long G = gridDim.x;
  long n = blockIdx.x;
  f32x4 dyv = f32x4{0.f, 0.f, 0.f, 0.f}, xv = dyv;
  auto load_row = [&](long r) {
    if constexpr (NATK) {
      dyv[0] = tid < OUT ? dy[r * OUT + tid] : 0.f;
      dyv[1] = tid + FAST_NT < OUT ? dy[r * OUT + tid + FAST_NT] : 0.f;
    } else {
      if (own) dyv = f32x4{dy[r * OUT + hid], dy[r * OUT + H + hid], dy[r * OUT + 2 * H + hid], dy[r * OUT + 3 * H + hid]};
    }
    if (tid < XI / 4) {       // four consecutive positions p = 4*tid + j of the [ROWS2][8] image: (row2, j2) <- x[row2*J2 + j2]
      if constexpr (DENSE && sizeof(TI) == 4) {
        xv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + r * IN + 4 * tid);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int p = 4 * tid + j, row2 = p >> 3, j2 = p & 7;
          const float v = ld(x, (size_t)(r * IN + (j2 < F::J2 ? row2 * F::J2 + j2 : 0)));
          xv[j] = j2 < F::J2 ? v : 0.f;
        }
      }
    }
  };
  if (n < n_rows) load_row(n);
  for (; n < n_rows; n += G) {
    // ---- phase 1: this row into LDS; the next row's loads take off ---------------------------------------------
    if constexpr (NATK) {
      // flat output o = m*I2 + i2: dgT[i2][m], T01 operand plane[i2][k1 = m], one element at a time
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int o = tid + e * FAST_NT;
        if (o < OUT) {
          const int m = o / F::I2, i2 = o % F::I2;
          dgT[i2 * W::DGS + m] = dyv[e];
          dbias[e] += dyv[e];
          __bf16 p0, p1, p2;
          split3(dyv[e], p0, p1, p2);
          const int off = x_off<W::K1>(i2, m);
          img1[off] = p0; img1[W::PL1 + off] = p1; img1[2 * W::PL1 + off] = p2;
        }
      }
    } else if (own) {
      // o = gate*H + hid = m*I2 + i2  ->  m = MPG*gate + hid/I2, i2 = hid % I2
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        dgT[(hid % F::I2) * W::DGS + F::MPG * g + hid / F::I2] = dyv[g];
        dbias[g] += dyv[g];
      }
      store_split4(img1, W::PL1, x_off<W::K1>(hid % F::I2, 4 * (hid / F::I2)), dyv);     // k1 = 4*(hid/I2) + gate
    }
    if (tid < XI / 4) *reinterpret_cast<f32x4*>(ximg + 4 * tid) = xv;
    if (n + G < n_rows) load_row(n + G);
    lds_barrier();
    // ---- phase 2: C2 = S2(x) (fp32 MFMA) and dC2 = W10 dy (split-bf16 MFMAs), both into fp32 images ----------------
    {
      using T2 = St<S, 2>;
      f32x4 acc[T2::XM][T2::YR];
      stage_mma<S, 2>(w2, ximg, acc, wave, lane);
#pragma unroll
      for (int xm = 0; xm < T2::XM; ++xm)
#pragma unroll
        for (int y = 0; y < T2::YR; ++y) {
          const int row2 = 16 * y + c, m0 = 16 * (wave + FAST_NW * xm) + 4 * q;   // feature m0 = (i2, r2 .. r2+3)
          if (m0 < F::M2 && row2 < F::ROWS2)
            *reinterpret_cast<f32x4*>(c2i + (m0 / F::R2) * W::C2S + row2 * F::R2 + m0 % F::R2) = acc[xm][y];
        }
      xbf8 bf[W::NM1][3];
#pragma unroll
      for (int u = 0; u < W::NM1; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bf[u][p] = *reinterpret_cast<const xbf8*>(img1 + p * W::PL1 +
                                                    x_off<W::K1>(c < F::I2 ? c : F::I2 - 1, 32 * u + 8 * q));
#pragma unroll
      for (int xx = 0; xx < W::XF; ++xx) {
        f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
#pragma unroll
        for (int u = 0; u < W::NM1; ++u) {
#pragma unroll
          for (int s = 0; s < 5; ++s)
            acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w01[xx][u][SPLIT_TW[s]], bf[u][SPLIT_TX[s]], acc_lo, 0, 0, 0);
          acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w01[xx][u][0], bf[u][0], acc_hi, 0, 0, 0);
        }
        // lane (c = i2, q), registers j: features 16ft + 4q + j = (row2, r2 = 4*(q&1) + j)
        const int f0 = 16 * (wave + FAST_NW * xx) + 4 * q;
        if (c < F::I2) *reinterpret_cast<f32x4*>(dci + (f0 / F::R2) * W::DCS + c * F::R2 + f0 % F::R2) = acc_hi + acc_lo;
      }
    }
    lds_barrier();
    // ---- phase 3: weight-gradient MFMAs (fp32, contraction on the k slots) -----------------------------------------
#pragma unroll
    for (int s = 0; s < F::I2 / 4; ++s) {                    // dW10[k][m] += C2[i2][k] * dy[m][i2], i2 = 4s + q
      float a[W::XF], bm[W::MT];
#pragma unroll
      for (int xx = 0; xx < W::XF; ++xx) a[xx] = c2i[(4 * s + q) * W::C2S + 16 * (wave + FAST_NW * xx) + c];
#pragma unroll
      for (int mt = 0; mt < W::MT; ++mt) bm[mt] = dgT[(4 * s + q) * W::DGS + 16 * mt + c];
#pragma unroll
      for (int xx = 0; xx < W::XF; ++xx)
#pragma unroll
        for (int mt = 0; mt < W::MT; ++mt)
          g10[xx][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[xx], bm[mt], g10[xx][mt], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < F::ROWS2 / 4; ++s) {                 // dW2[j2][col] += x[row2][j2] * dC2[row2][col], row2 = 4s + q
      const float a = c < F::J2 ? ximg[(4 * s + q) * 8 + c] : 0.f;
#pragma unroll
      for (int xc = 0; xc < W::XC; ++xc) {
        const int ctile = wave + FAST_NW * xc;                  // wave-uniform; tiles beyond CT2 do not exist
        const float bcol = dci[(4 * s + q) * W::DCS + 16 * (ctile < W::CT2 ? ctile : 0) + c];
        g2[xc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bcol, g2[xc], 0, 0, 0);
      }
    }
    if constexpr (NEED_DX) {
      // dx[row2][j2] = sum_(i2,r2) G2[j2; i2,r2] * dC2[row2][(i2,r2)]: this wave's quarter of the contraction for the
      // 16 chain rows of column tile dct
      f32x4 ax = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < W::KQ; ++s)
        ax = __builtin_amdgcn_mfma_f32_16x16x4f32(wdx[s], dci[(16 * dct + c) * W::DCS + 4 * (dkq * W::KQ + s) + q], ax, 0, 0, 0);
      // lane (c = row2 in the tile, q), registers j: j2 = 4q + j (q < 2)
      if (q < 2) *reinterpret_cast<f32x4*>(dxs + dkq * H + (16 * dct + c) * F::J2 + 4 * q) = ax;
    }
    lds_barrier();
    if constexpr (NEED_DX) {
      if (tid < H / 4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(dxs) + tid;
        const f32x4 v = p[0] + p[H / 4] + p[2 * (H / 4)] + p[3 * (H / 4)];
#pragma unroll
        for (int j = 0; j < 4; ++j) st(dx, (size_t)(n * H + 4 * tid + j), v[j]);
      }
    }
  }
  // ---- flush: one atomic per accumulator element and workgroup ---------------------------------------------------------
#pragma unroll
  for (int xx = 0; xx < W::XF; ++xx)
#pragma unroll
    for (int mt = 0; mt < W::MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j)      // D row 4q + j = feature k, column c = m
        atomicAdd(dW10 + (size_t)(16 * (wave + FAST_NW * xx) + 4 * q + j) * F::M + 16 * mt + c, g10[xx][mt][j]);
  if (q < 2) {
    float* dW2 = d_packed + woff_of<S>(2);                    // [J2][M2 = (i2, r2)]
#pragma unroll
    for (int xc = 0; xc < W::XC; ++xc)
      if (wave + FAST_NW * xc < W::CT2) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * q + j < F::J2) atomicAdd(dW2 + (size_t)(4 * q + j) * F::M2 + 16 * (wave + FAST_NW * xc) + c, g2[xc][j]);
      }
  }
  if (d_bias) {
    if constexpr (NATK) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (tid + e * FAST_NT < OUT) atomicAdd(d_bias + tid + e * FAST_NT, dbias[e]);
    } else if (own) {
#pragma unroll
      for (int g = 0; g < 4; ++g) atomicAdd(d_bias + g * H + hid, dbias[g]);
    }
  }
}

// dG0, dG1 from dW10 by the product rule; one thread per core element, accumulating into d_packed
template <class S>
__global__ void __launch_bounds__(256) k_f10w_finish(const float* __restrict__ packed, const float* __restrict__ dW10,
                                                     float* __restrict__ d_packed) {
  using F = F10<S>;
  constexpr int N0 = F::J0 * F::R1 * F::I0;                  // elements of W_0 [J0*R1][I0]
  constexpr int N1 = F::J1 * F::R2 * F::I1 * F::R1;          // elements of W_1 [J1*R2][I1*R1]
  const float* W0 = packed + woff_of<S>(0);
  const float* W1 = packed + woff_of<S>(1);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < N0) {
    const int i0 = e % F::I0, r1 = (e / F::I0) % F::R1, j0 = e / (F::I0 * F::R1);
    float v = 0.f;
    for (int j1 = 0; j1 < F::J1; ++j1)
      for (int r2 = 0; r2 < F::R2; ++r2)
        for (int i1 = 0; i1 < F::I1; ++i1)
          v = fmaf(W1[(j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1 + r1],
                   dW10[(size_t)((j0 * F::J1 + j1) * F::R2 + r2) * F::M + i0 * F::I1 + i1], v);
    d_packed[woff_of<S>(0) + e] += v;
  } else if (e < N0 + N1) {
    const int f = e - N0;
    const int r1 = f % F::R1, i1 = (f / F::R1) % F::I1, r2 = (f / (F::R1 * F::I1)) % F::R2, j1 = f / (F::R1 * F::I1 * F::R2);
    float v = 0.f;
    for (int j0 = 0; j0 < F::J0; ++j0)
      for (int i0 = 0; i0 < F::I0; ++i0)
        v = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0],
                 dW10[(size_t)((j0 * F::J1 + j1) * F::R2 + r2) * F::M + i0 * F::I1 + i1], v);
    d_packed[woff_of<S>(1) + f] += v;
  }
}

// ---- dispatch ------------------------------------------------------------------------------------------
template <class S, typename TI>
static int launch_wgrad_f10(const TtShape& ts, long n_rows, const float* packed, const void* x, const void* dy,
                            void* dx, float* d_packed, float* d_bias, void* ws, hipStream_t stream) {
  using F = F10<S>;
  if (!ws) return TTRNN_ERR_WORKSPACE;
  float* dW10 = reinterpret_cast<float*>(ws);
  const size_t dw_bytes = (size_t)F::K * F::M * sizeof(float);
  void* wfrag = (char*)ws + dw_bytes;
  int st = launch_f10b_prep(ts, packed, wfrag, stream, dW10, (int)(dw_bytes / sizeof(float)));      // + dW10 = 0
  if (st != TTRNN_OK) return st;
  constexpr size_t lds = f10w_lds_bytes<S>();
  static_assert(lds <= 160 * 1024, "LDS image set too large");
  auto kern = k_ttlinear_wgrad_f10<S, TI, false>;
  if constexpr (NEEDS_HIDDEN_DX<S>()) {
    if (dx) kern = k_ttlinear_wgrad_f10<S, TI, true>;
  }
  if (lds > 64 * 1024) {
    {
      if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != TTRNN_OK)
        return TTRNN_ERR_LAUNCH;
    }
  }
  const int cus = device_cu_count();
  const long grid = n_rows < (long)cus ? n_rows : (long)cus;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(FAST_NT), lds, stream, n_rows, packed, (const xbf8*)wfrag,
                     (const TI*)x, (const float*)dy, (TI*)dx, dW10, d_packed, d_bias);
  constexpr int NE = F::J0 * F::R1 * F::I0 + F::J1 * F::R2 * F::I1 * F::R1;
  hipLaunchKernelGGL((k_f10w_finish<S>), dim3((NE + 255) / 256), dim3(256), 0, stream, packed, dW10, d_packed);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

bool f10_ttlinear_wgrad_available(const TtShape& s, int dtype, int dy_dtype) {
  if (opt(OPT_NO_F10) || dy_dtype != TTRNN_F32) return false;
  if (dtype == TTRNN_F32)
    return shape_matches<ShpH256R8L>(s) || shape_matches<ShpH256R16L>(s) || shape_matches<ShpH256R8G>(s) ||
           shape_matches<ShpI40R16L>(s);
  return dtype == TTRNN_BF16 && shape_matches<ShpH256R8G>(s);
}

size_t f10_ttlinear_wgrad_workspace_bytes(const TtShape& s) {
  if (shape_matches<ShpH256R8L>(s))
    return (size_t)F10<ShpH256R8L>::K * F10<ShpH256R8L>::M * sizeof(float) + f10b_fragment_bytes(s);
  if (shape_matches<ShpH256R16L>(s))
    return (size_t)F10<ShpH256R16L>::K * F10<ShpH256R16L>::M * sizeof(float) + f10b_fragment_bytes(s);
  if (shape_matches<ShpH256R8G>(s))
    return (size_t)F10<ShpH256R8G>::K * F10<ShpH256R8G>::M * sizeof(float) + f10b_fragment_bytes(s);
  if (shape_matches<ShpI40R16L>(s))
    return (size_t)F10<ShpI40R16L>::K * F10<ShpI40R16L>::M * sizeof(float) + f10b_fragment_bytes(s);
  return 0;
}

// whether the kernel can also produce dx for this shape (hidden-shaped inputs only)
bool f10_ttlinear_wgrad_has_dx(const TtShape& s) { return !shape_matches<ShpI40R16L>(s); }

// dx may be NULL (hidden-to-hidden matrices: dh travels inside the reverse-time kernel)
// dtype: storage type of x / dx (fp32; bf16 for the GRU shape); dy is fp32
int launch_ttlinear_wgrad_f10(const TtShape& s, int dtype, int64_t n_rows, const float* packed, const void* x,
                              const void* dy, void* dx, float* d_packed, float* d_bias, void* ws, hipStream_t stream) {
  if (n_rows <= 0) return TTRNN_OK;
  if (dtype == TTRNN_F32) {
    if (shape_matches<ShpH256R8L>(s))
      return launch_wgrad_f10<ShpH256R8L, float>(s, (long)n_rows, packed, x, dy, dx, d_packed, d_bias, ws, stream);
    if (shape_matches<ShpH256R16L>(s))
      return launch_wgrad_f10<ShpH256R16L, float>(s, (long)n_rows, packed, x, dy, dx, d_packed, d_bias, ws, stream);
    if (shape_matches<ShpH256R8G>(s))
      return launch_wgrad_f10<ShpH256R8G, float>(s, (long)n_rows, packed, x, dy, dx, d_packed, d_bias, ws, stream);
    if (shape_matches<ShpI40R16L>(s) && !dx)
      return launch_wgrad_f10<ShpI40R16L, float>(s, (long)n_rows, packed, x, dy, dx, d_packed, d_bias, ws, stream);
  } else if (dtype == TTRNN_BF16 && shape_matches<ShpH256R8G>(s)) {
    return launch_wgrad_f10<ShpH256R8G, bf16_t>(s, (long)n_rows, packed, x, dy, dx, d_packed, d_bias, ws, stream);
  }
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
