// ttrnn_fast_f10bh.hip — reverse-time TT-LSTM kernel on the fused core with TWO fp16 pieces per operand (gfx950).
//
// Same two transposed stages as ttrnn_fast_f10b.hip (T01: dC2 = W10 dg, T2: dh = G2 dC2; reference: torch autograd through
// tensorized_rnn/lstm.py:23-32,123-133 and t3nsor/ops.py:78-93), three f16 MFMA terms (hi*hi, hi*lo, lo*hi) instead of six bf16
// ones.  fp16's narrow exponent is handled per operand:
//   * the weights (fixed per launch): one power-of-two scale per ROW of W10 (feature (row2, r2)) and of G2 (j2), row maximum
//     -> [2^13, 2^14); a row is an OUTPUT row of its stage, so the scale is undone on the accumulators, per register.  The
//     pieces' representation error is then at most 2^-22 of each entry, or 2^-25 absolute (scaled) for entries more than
//     2^17 below their row's maximum: summed over a row never more than 2^-21 of the row's L1 norm — the error bound of an
//     fp32 dot product with that row.  No guard and no fallback are needed (with ONE scale per matrix, as in the cfg5-class
//     kernel, a whole row can sit in fp16's subnormal range: there they are);
//   * the gradients (new every step, magnitudes falling over many decades along a sequence): one scale per step from the
//     EXACT maximum of |dg_t| — every gate wave reduces its values with four DPP row rotations + four readlanes (a
//     ds_bpermute butterfly would cost the stage it saves) and the four maxima cross the barrier behind the gate phase.  The
//     1 024 values are split ONCE, two per thread, in a phase of its own behind that barrier (splitting them in every one of
//     the eight waves where the fragments are read made T01 VALU-issue bound: 3 416 cycles per step against 2 956 of the
//     three-bf16-piece kernel).  T01's results are rescaled for T2 by the bound |dC2| <= max_f L1(W10 row f) max|dg_t| — known
//     at the same moment, so they go to LDS split already — which costs T2's operand the 3 - 5 bits between the bound and the
//     typical entry (of the 6 it has to spare before the second piece thins out) and cannot overflow.
// Per step: G (waves 0-3; fp32 image + row for HBM + wave maximum) | barrier | split | barrier | T01 (12 MFMAs per wave at
// r = 8) | barrier | T2 (3 MFMAs per wave) | barrier.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_f10.h"

namespace ttrnn {
namespace {

using ShpH512R8G = Shp<3, 8, 8, 8, 1, 8, 12, 16, 1, 8, 8, 1>;      // benchmarking.py defaults with --gru (ttrnn_fast_f10g5.hip has the forward)
typedef _Float16 xh4 __attribute__((ext_vector_type(4)));

template <class S>
struct F10BH {
  using F = F10<S>;
  static constexpr int H = F::H;
  static constexpr int K1 = F::M, NM1 = K1 / 32;            // T01 contraction (64), its k-blocks
  static constexpr int FT = F::K / 16, XF = FT / FAST_NW;   // T01 feature tiles, per wave
  static constexpr int K2 = F::I2 * F::R2, NM2 = K2 / 32;   // T2 contraction, its k-blocks = partial-sum slices
  static constexpr int CT2 = F::ROWS2 / 16;                 // T2 column tiles (2)
  static constexpr int XT2 = NM2 * CT2 / FAST_NW;           // T2 (column tile, k-block) pairs per wave
  static constexpr int HI = F::I2 / 2;
  static constexpr int NWG = H / 64;                        // gate waves: one hidden unit per thread (H = 512: all eight)
  static constexpr int ROWS = F::K + 16;                    // rows: W10's, then G2's (padded to a tile)
  // header (floats): [un1: F::K | un2: 16 | L1 norms of the scaled rows: ROWS], padded to 256 bytes
  static constexpr int UN1 = 0, UN2 = F::K, L1N = F::K + 16;
  static constexpr int HDR_FLOATS = (F::K + 16 + ROWS + 63) / 64 * 64;
  static constexpr size_t FRAGS_T = (size_t)(FT * NM1 + NM2) * 2 * 64;    // xh8 fragments of T01 and T2
  static constexpr size_t FRAGS = FRAGS_T + (size_t)F::R2 * 64;           // + (wave-local kernel) T2's B operand per r2: two planes of xh4
  __device__ static constexpr int m_of_k1(int k) { return (k & 3) * F::MPG + (k >> 2); }
  __device__ static constexpr int k2_of(int i2, int r2) { return ((r2 >> 2) * HI + (i2 >> 1)) * 8 + (i2 & 1) * 4 + (r2 & 3); }
};

template <class S>
constexpr bool f10bh_ok() {
  using F = F10<S>;
  using B = F10BH<S>;
  return f10_ok<S>() && F::I2 == 16 && (B::K1 == 64 || B::K1 == 128) && B::FT % FAST_NW == 0 && B::K2 % 32 == 0 &&
         (B::NM2 * B::CT2) % FAST_NW == 0 && (B::CT2 == 2 || B::CT2 == 4) && F::J2 == 8 && (F::H == 256 || F::H == 512) &&
         B::NM2 <= FAST_NW && B::NM1 <= FAST_NW && B::NWG <= FAST_NW;
}

// x < 2^e, clamped so that 2^(14 - e) and its inverse stay normal floats (gradients deep in a sequence are tiny)
__device__ __forceinline__ int expo_wide(float x) {
  if (!(x > 0.f)) return 0;
  if (!(x < 3e38f)) return 100;
  int e;
  frexpf(x, &e);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

// The step scales inside the time loop, from the bits of the maximum (frexpf / ldexpf of the device library cost the kernel
// 300 cycles per stage): mx >= 0 has the biased exponent eb = bits >> 23, mx < 2^(eb - 126); eb clamped to [27, 227] keeps
// 2^(14 - e) and its inverse normal (a zero maximum scales by 2^114: zeros stay zeros).  Returns 2^(14 - e), un = 2^(e - 14).
__device__ __forceinline__ float step_scale(float mx, float& un) {
  int eb = (int)(__float_as_uint(mx) >> 23);
  eb = eb < 27 ? 27 : (eb > 227 ? 227 : eb);
  un = __uint_as_float((unsigned)(eb - 13) << 23);
  return __uint_as_float((unsigned)(267 - eb) << 23);
}

template <int N>
__device__ __forceinline__ float row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}
// maximum over the 64 lanes, in every lane (max is idempotent: rotations inside the 16-lane rows, then one lane of each row)
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, row_ror<1>(v));
  v = fmaxf(v, row_ror<2>(v));
  v = fmaxf(v, row_ror<4>(v));
  v = fmaxf(v, row_ror<8>(v));
  const int i = __float_as_int(v);
  const float a = __int_as_float(__builtin_amdgcn_readlane(i, 0)), b = __int_as_float(__builtin_amdgcn_readlane(i, 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(i, 32)), d = __int_as_float(__builtin_amdgcn_readlane(i, 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// eight consecutive k of an fp32 operand -> the MFMA fragments of its two fp16 pieces
__device__ __forceinline__ void split8h(const f32x4& va, const f32x4& vb, xh8& hi8, xh8& lo8) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  split_pair_h(va[0], va[1], h0, l0);
  split_pair_h(va[2], va[3], h1, l1);
  split_pair_h(vb[0], vb[1], h2, l2);
  split_pair_h(vb[2], vb[3], h3, l3);
  hi8 = __builtin_bit_cast(xh8, u32x4{h0, h1, h2, h3});
  lo8 = __builtin_bit_cast(xh8, u32x4{l0, l1, l2, l3});
}

// Fragments + row scales + the scaled rows' L1 norms.  Block ft < FT: rows 16 ft + r of W10 (wave u = k-block u); block FT: the rows of G2.
// Fragment order: T01 wf[((ft*NM1 + u)*2 + p)*64 + lane], lane (r, q): k = 32u + 8q + e (gate-interleaved: m_of_k1);
//                 T2  wf[T01 part + (u*2 + p)*64 + lane], lane (r, q): row j2 = r (< J2, else 0), k2 slot 4u + q
// WL (round 6, k_lstm_bwd_f10h<S, DIAG, true>): feature tile t = (r2 = t / CT2, sixteen chain rows 16 (t % CT2) ..) instead of sixteen
// consecutive (row2, r2) — a tile then shares ONE r2, which is what lets T2 take T01's accumulator tiles as its operand — and, behind
// the fragments above, T2's B operand for v_mfma_f32_16x16x16_f16 per r2: lane (c = j2, q) holds G2[r2][j2][i2 = 4 q ..] (two planes)
template <class S, bool NATK = false, bool WL = false>
__global__ void __launch_bounds__(FAST_NT) k_f10bh_prep(const float* __restrict__ packed, float* __restrict__ hdr,
                                                        xh8* __restrict__ wfrag, unsigned* __restrict__ zero, int zero_n) {
  // (the by-products' column maxima start from zero: cleared here instead of by a launch of their own)
  if (zero)
    for (int e = blockIdx.x * FAST_NT + threadIdx.x; e < zero_n; e += gridDim.x * FAST_NT) zero[e] = 0u;
  using F = F10<S>;
  using B = F10BH<S>;
  __shared__ float red[2][FAST_NW][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const bool g2 = blockIdx.x == B::FT;
  const int nu = g2 ? B::NM2 : B::NM1;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (wave < nu) {
    if (!g2) {
      const int f = 16 * blockIdx.x + r;
      const int row2 = WL ? 16 * ((int)blockIdx.x % B::CT2) + r : f / F::R2, r2 = WL ? (int)blockIdx.x / B::CT2 : f % F::R2;
      const int j1 = row2 % F::J1, j0 = row2 / F::J1;
      const float* W0 = packed + woff_of<S>(0);               // [J0*R1][I0]
      const float* W1 = packed + woff_of<S>(1);               // [J1*R2][I1*R1]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // LSTM: gate-interleaved k order (m_of_k1); GRU: natural order (its gate index does not align with the tiles)
        const int m = NATK ? 32 * wave + 8 * q + e : B::m_of_k1(32 * wave + 8 * q + e);
        const int i0 = m / F::I1, i1 = m % F::I1;
        const float* w1p = W1 + (j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1;
        float a = 0.f;
        for (int r1 = 0; r1 < F::R1; ++r1) a = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0], w1p[r1], a);
        v[e] = a;
      }
    } else {
      const float* W2 = packed + woff_of<S>(2);               // [J2][M2 = I2*R2]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int slot = 4 * wave + q;
        const int i2 = 2 * (slot % B::HI) + (e >> 2), r2 = (slot / B::HI) * 4 + (e & 3);
        v[e] = (r < F::J2 && i2 < F::I2) ? W2[r * F::M2 + i2 * F::R2 + r2] : 0.f;
      }
    }
  }
  // row maximum: the 8 values of the lane, the four q groups of the wave, the waves of the block
  float mx = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[e]));
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  if (q == 0) red[0][wave][r] = mx;
  __syncthreads();
  float rm = 0.f;
  for (int w = 0; w < nu; ++w) rm = fmaxf(rm, red[0][w][r]);
  const int ex = expo_wide(rm);
  const float sc = ldexpf(1.f, 14 - ex);
  xh8 p0, p1;
  float as = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    _Float16 a, b;
    const float w = v[e] * sc;
    split2h(w, a, b);
    p0[e] = a; p1[e] = b;
    as += fabsf(w);
  }
  as += __shfl_xor(as, 16); as += __shfl_xor(as, 32);
  if (q == 0) red[1][wave][r] = as;
  if (wave < nu) {
    xh8* dst = wfrag + (g2 ? (size_t)(B::FT * B::NM1 + wave) * 2 * 64 : (size_t)(blockIdx.x * B::NM1 + wave) * 2 * 64) + lane;
    dst[0] = p0;
    dst[64] = p1;
  }
  if constexpr (WL) {
    if (g2 && wave < F::R2) {                                 // wave = r2: the 16 x 16 tile B[k = i2][n = j2] of G2[r2], row j2 under its scale
      static_assert(F::R2 <= FAST_NW && F::I2 == 16, "one wave per r2");
      const float* W2 = packed + woff_of<S>(2);
      xh4* d4 = reinterpret_cast<xh4*>(wfrag + B::FRAGS_T);
      xh4 q0, q1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float w = (r < F::J2 ? W2[r * F::M2 + (4 * q + j) * F::R2 + wave] : 0.f) * sc;
        _Float16 a, b;
        split2h(w, a, b);
        q0[j] = a; q1[j] = b;
      }
      d4[(wave * 2 + 0) * 64 + lane] = q0;
      d4[(wave * 2 + 1) * 64 + lane] = q1;
    }
  }
  __syncthreads();
  if (tid < 16) {
    float a2 = 0.f;
    for (int w = 0; w < nu; ++w) a2 += red[1][w][tid];
    const int row = g2 ? F::K + tid : 16 * blockIdx.x + tid;
    hdr[row] = ldexpf(1.f, ex - 14);                      // (tid < 16: r == tid, rm / ex are this row's)
    hdr[B::L1N + row] = a2;
  }
}

template <class S>
constexpr size_t f10bh_lds_bytes() {
  using B = F10BH<S>;
  using F = F10<S>;
  return sizeof(float) * ((size_t)4 * B::H + (size_t)B::NM2 * B::H) +
         sizeof(_Float16) * 2 * ((size_t)F::I2 * B::K1 + (size_t)F::ROWS2 * B::K2) + (B::XF >= 4 ? sizeof(float) * F::K : 0);
}

template <class S>
constexpr size_t f10bp_lds_bytes() { return f10bh_lds_bytes<S>() + (size_t)2 * 2 * F10<S>::H * sizeof(f32x4); }      // + the factor vectors

// WL (round 6): T01 and T2 WAVE-LOCAL.  Wave w takes chain-row block rb = w % CT2 and the ranks r2 = 2 (w / CT2),
// + 1: T01 as D[i2][row2] = dz^T[i2][m] W10[(row2, r2)][m] — the gate-gradient image is the A operand, the fused core's rows of ONE r2
// the B operand — leaves a tile with the chain row on the lane and four consecutive i2 in the registers, which IS the A operand of
// v_mfma_f32_16x16x16_f16 for T2's contraction over i2 (B = G2[r2][j2][i2], per r2, from registers); the wave sums its two r2 and
// leaves one partial dh slice.  No dC2 image, no barrier between the transposed stages: three barriers per step instead of four.
template <class S, bool DIAG, bool WL = false>
__global__ void __launch_bounds__(FAST_NT) k_lstm_bwd_f10h(int Bn, int T, const float* __restrict__ c0,
                                                           const float* __restrict__ hdr, const xh8* __restrict__ wfrag,
                                                           const float* __restrict__ reserve,
                                                           const float* __restrict__ d_out,
                                                           const float* __restrict__ d_hT,
                                                           const float* __restrict__ d_cT, float* __restrict__ dg_in,
                                                           float* __restrict__ dg_hid, float* __restrict__ d_h0,
                                                           float* __restrict__ d_c0,
                                                           unsigned long long* __restrict__ diag, BwdStats bs) {
  static_assert(f10bh_ok<S>(), "shape not supported by the two-piece fused-core reverse-time kernel");
  using F = F10<S>;
  using B = F10BH<S>;
  constexpr int H = F::H, GH = 4 * H;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ __attribute__((aligned(16))) float smax1[B::NWG];               // per-wave maxima of |dg| (the gate waves)
  __shared__ float sl1[FAST_NW];                                             // (prologue) per-wave maxima of the rows' L1 norms
  constexpr int PL1 = F::I2 * B::K1, PL2 = F::ROWS2 * B::K2;                 // elements per fp16 plane
  // the fp32 row goes out in 16-byte pieces: H = 256 from waves 4-7 (one each) while the gate waves split, H = 512 (every wave
  // is a gate wave) one piece per thread behind its split
  static_assert(GH / 4 == FAST_NT - H || GH / 4 == FAST_NT, "one 16-byte piece of the fp32 row per storing thread");
  constexpr bool ALLG = H == FAST_NT;
  float* dgf = reinterpret_cast<float*>(smem);                               // [4H] in HBM row order
  float* dhs = dgf + GH;                                                     // [NM2][H]
  _Float16* img1h = reinterpret_cast<_Float16*>(dhs + B::NM2 * H);           // dg's two fp16 pieces [2][I2][K1] (x_off)
  _Float16* img2h = img1h + 2 * PL1;                                         // dC2's two fp16 pieces [2][ROWS2][K2] (x_off)
  float* un1s = reinterpret_cast<float*>(img2h + 2 * PL2);                   // (XF >= 4) T01's inverse row scales [F::K]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  // resident fragments (two pieces) and the inverse row scales of the accumulator registers
  static_assert(!WL || (B::XF == 2 && F::R2 == 8 && B::NM2 * B::CT2 == FAST_NW && F::I2 == 16), "wave-local stages: rank 8, two tiles per wave");
  xh8 w01[B::XF][B::NM1][2], w2t[B::XT2][2];
  f32x4 un1[B::XF], un2;
  xh4 g2x[WL ? 2 : 1][2];                                  // (WL) T2's B operand of the wave's two r2
  float un1w[2] = {0.f, 0.f}, un2w = 0.f;                   // (WL) inverse row scales: per lane column
#pragma unroll
  for (int x = 0; x < B::XF; ++x) {
    // (WL) tile (r2 = 2 (wave / CT2) + x, row block wave % CT2)
    const int tile = WL ? (2 * (wave / B::CT2) + x) * B::CT2 + wave % B::CT2 : wave + FAST_NW * x;
#pragma unroll
    for (int u = 0; u < B::NM1; ++u)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        w01[x][u][p] = wfrag[(size_t)((tile * B::NM1 + u) * 2 + p) * 64 + lane];
    if constexpr (B::XF < 4) un1[x] = *reinterpret_cast<const f32x4*>(hdr + B::UN1 + 16 * tile + 4 * q);
    if constexpr (WL) {
      un1w[x] = hdr[B::UN1 + 16 * tile + (lane & 15)];
#pragma unroll
      for (int p = 0; p < 2; ++p)
        g2x[x][p] = reinterpret_cast<const xh4*>(wfrag + B::FRAGS_T)[((2 * (wave / B::CT2) + x) * 2 + p) * 64 + lane];
    }
  }
  if constexpr (WL) un2w = hdr[B::UN2 + (lane & 15)];
  if constexpr (B::XF >= 4)
    for (int f = tid; f < F::K; f += FAST_NT) un1s[f] = hdr[B::UN1 + f];
  static_assert(FAST_NW % B::CT2 == 0, "a wave's T2 pairs share the column tile");
  const int ct = wave % B::CT2;
#pragma unroll
  for (int x = 0; x < B::XT2; ++x)
#pragma unroll
    for (int p = 0; p < 2; ++p)
      w2t[x][p] = wfrag[(size_t)(B::FT * B::NM1 * 2 + ((wave + FAST_NW * x) / B::CT2) * 2 + p) * 64 + lane];
  un2 = *reinterpret_cast<const f32x4*>(hdr + B::UN2 + 4 * q);
  // bound of T01's results: |dC2[f][.]| <= L1(row f of W10) max|dg|.  The largest row norm, once per launch
  float maxl1;
  {
    float l = 0.f;
    for (int f = tid; f < F::K; f += FAST_NT) l = fmaxf(l, hdr[B::L1N + f] * hdr[B::UN1 + f]);
    l = wave_max(l);
    if (lane == 0) sl1[wave] = l;
    __syncthreads();
    maxl1 = sl1[0];
#pragma unroll
    for (int w = 1; w < FAST_NW; ++w) maxl1 = fmaxf(maxl1, sl1[w]);
  }

  // gate phase: thread tid < H owns hidden unit tid; three rotating register sets (see k_lstm_bwd_f10)
  const bool own = tid < H;
  const int hid = own ? tid : 0;
  float dcs = (own && d_cT) ? d_cT[b * H + hid] : 0.f;
  const float c0v = (own && c0) ? c0[b * H + hid] : 0.f;
  const float* dptr = d_out ? d_out : reserve;
  const float dscale = d_out ? 1.0f : 0.0f;
  f32x4 ra0 = f32x4{0.f, 0.f, 0.f, 0.f}, ra1 = ra0, ra2 = ra0;
  float rb0 = 0.f, rb1 = 0.f, rb2 = 0.f;
  float do0 = 0.f, do1 = 0.f, do2 = 0.f;
  const float* xptr = bs.x ? reinterpret_cast<const float*>(bs.x) : reserve;
  const float xscale = bs.x ? 1.0f : 0.0f;
  float xq0 = 0.f, xq1 = 0.f, xq2 = 0.f;
  f32x4 cmx = f32x4{0.f, 0.f, 0.f, 0.f}, sxd = cmx, sdg = cmx;
  if (own) {
    dhs[hid] = d_hT ? d_hT[b * H + hid] : 0.f;
#pragma unroll
    for (int sl = 1; sl < B::NM2; ++sl) dhs[sl * H + hid] = 0.f;
    if (T > 0) {
      const size_t bt = b * T + (T - 1);
      const float* rv = reserve + res_gate(bt, H, hid);
      const float* rc = reserve + res_cell((size_t)Bn * T, bt, H, hid);
      ra0 = *reinterpret_cast<const f32x4*>(rv);
      rb0 = rc[0];
      do0 = dptr[bt * H + hid];
      xq0 = xptr[bt];
      if (T > 1) {
        ra1 = *reinterpret_cast<const f32x4*>(rv - H * 4);
        rb1 = rc[-H];
        do1 = dptr[(bt - 1) * H + hid];
        xq1 = xptr[bt - 1];
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): keep weight-register waits out of the time loop
  lds_barrier();
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();

  auto step = [&](const int t, const f32x4& ra, const float& rb, const float& dout_c, const float& nb, f32x4& fa,
                  float& fb, float& dout_f, const float& x_c, float& x_f) {
    const size_t bt = b * T + t;
    f32x4 pkeep = f32x4{0.f, 0.f, 0.f, 0.f};                      // the gate waves' values stay in registers for the split
    // ---- G: gate gradients (lstm.py:26-32 differentiated) ---------------------------------------------------------
    if (own) {
      const f32x4 qa = ra;
      float dht = dout_c * dscale;
#pragma unroll
      for (int sl = 0; sl < B::NM2; ++sl) dht += dhs[sl * H + hid];
      const float ig = qa[0], gg = qa[1], fg = qa[2], og = qa[3], cy = rb;
      const float cprev = t > 0 ? nb : c0v;
      const float tc = ftanh(cy);
      const float dct = dcs + dht * og * (1.0f - tc * tc);
      const float p0 = dct * gg * ig * (1.0f - ig);             // d pre-activation of i
      const float p1 = dct * cprev * fg * (1.0f - fg);          //                     f
      const float p2 = dct * ig * (1.0f - gg * gg);             //                     g
      const float p3 = dht * tc * og * (1.0f - og);             //                     o
      dcs = dct * fg;
      const f32x4 pv = f32x4{p0, p1, p2, p3};
      const float xv = x_c * xscale;
      float mx = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float a = fabsf(pv[g]);
        mx = fmaxf(mx, a);
        cmx[g] = fmaxf(cmx[g], a);
        sxd[g] = fmaf(xv, pv[g], sxd[g]);
        sdg[g] += pv[g];
      }
      dgf[hid] = p0; dgf[H + hid] = p1; dgf[2 * H + hid] = p2; dgf[3 * H + hid] = p3;
      pkeep = pv;
      mx = wave_max(mx);
      if (lane == 0) smax1[wave] = mx;
    }
    TT_STAMP(0)
    lds_barrier();
    TT_STAMP(1)
    // ---- split: the step's scale from the exact maximum; the gate waves split their own four values (still in registers) ----
    float u2;                                                     // inverse of T2's operand scale, kept for T2's epilogue
    float t01f;                                                   // un-scale of T01's accumulators times T2's operand scale
    {
      f32x4 m4 = *reinterpret_cast<const f32x4*>(smax1);
      if constexpr (B::NWG == 8) {
        const f32x4 m5 = *reinterpret_cast<const f32x4*>(smax1 + 4);
        m4 = f32x4{fmaxf(m4[0], m5[0]), fmaxf(m4[1], m5[1]), fmaxf(m4[2], m5[2]), fmaxf(m4[3], m5[3])};
      }
      const float mxg = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      if (bs.rowmax && tid == FAST_NT - 1) bs.rowmax[bt] = mxg;  // by-product: the row maximum of d_gates (dy_rowmax hint of dx)
      float ug;
      const float sg = step_scale(mxg, ug);
      const float s2 = step_scale(mxg * maxl1, u2);              // |dC2| <= maxl1 * mxg: no overflow, whatever the signs
      t01f = ug * s2;
      if (own) {
        // o = gate*H + hid = m*I2 + i2  ->  m = MPG*gate + hid/I2, i2 = hid%I2: the 4 gates are k = 4*(hid/I2) .. +3
        store_split4_h(img1h, PL1, x_off<B::K1>(hid % F::I2, 4 * (hid / F::I2)), pkeep * sg);
      }
      if (ALLG || !own) {                                         // (H = 256: waves 4-7) the fp32 row goes out to HBM meanwhile
        const int i4 = ALLG ? tid : tid - H;
        const f32x4 v = reinterpret_cast<const f32x4*>(dgf)[i4];
        reinterpret_cast<f32x4*>(dg_in + bt * GH)[i4] = v;
        if (dg_hid && dg_hid != dg_in) reinterpret_cast<f32x4*>(dg_hid + bt * GH)[i4] = v;
      }
    }
    TT_STAMP(2)
    lds_barrier();
    TT_STAMP(3)
    if (own) {
      // record(t-2), d_out(t-2), x(t-2): requested HERE, under T01's MFMAs (the gate waves reach T01's barrier 180 cycles before
      // waves 4-7: their thirty instructions of address arithmetic cost the split phase 150 cycles and T01 nothing), consumed two
      // steps from now.  Always four loads, no branch (index clamped; a null d_out / x reads the reserve and is scaled by zero)
      const size_t b2 = t > 1 ? bt - 2 : b * T;
      fa = *reinterpret_cast<const f32x4*>(reserve + res_gate(b2, H, hid));
      fb = reserve[res_cell((size_t)Bn * T, b2, H, hid)];
      dout_f = dptr[b2 * H + hid];
      x_f = xptr[b2];
    }
    // ---- T01: dC2 = W10 dg, rescaled for T2 and split into its operand image ------------------------------------------------
    {
      if constexpr (B::XF >= 4) {
        // many feature tiles per wave (H = 512): k-block outermost — one pair of operand fragments live at a time, the tiles'
        // accumulators are the independent chains
        f32x4 acc[B::XF];
#pragma unroll
        for (int x = 0; x < B::XF; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < B::NM1; ++u) {
          const xh8 b0 = *reinterpret_cast<const xh8*>(img1h + x_off<B::K1>(c, 32 * u + 8 * q));
          const xh8 b1 = *reinterpret_cast<const xh8*>(img1h + PL1 + x_off<B::K1>(c, 32 * u + 8 * q));
#pragma unroll
          for (int x = 0; x < B::XF; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][1], b0, acc[x], 0, 0, 0);
#pragma unroll
          for (int x = 0; x < B::XF; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][0], b1, acc[x], 0, 0, 0);
#pragma unroll
          for (int x = 0; x < B::XF; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][0], b0, acc[x], 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < B::XF; ++x) {
          const int f0 = 16 * (wave + FAST_NW * x) + 4 * q;
          const int row2 = f0 / F::R2, r20 = f0 % F::R2;
          const f32x4 u1 = *reinterpret_cast<const f32x4*>(un1s + f0);
          store_split4_h(img2h, PL2, x_off<B::K2>(row2, B::k2_of(c, r20)), acc[x] * (u1 * t01f));
        }
      } else if constexpr (WL) {
        xh8 bf[B::NM1][2];
#pragma unroll
        for (int u = 0; u < B::NM1; ++u)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            bf[u][p] = *reinterpret_cast<const xh8*>(img1h + p * PL1 + x_off<B::K1>(c, 32 * u + 8 * q));
        f32x4 au[2][B::NM1];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int u = 0; u < B::NM1; ++u) {                        // A = the gate-gradient image (rows i2), B = the fused core's rows of one r2
            au[x][u] = f32x4{0.f, 0.f, 0.f, 0.f};
            au[x][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[u][0], w01[x][u][1], au[x][u], 0, 0, 0);
            au[x][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[u][1], w01[x][u][0], au[x][u], 0, 0, 0);
            au[x][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[u][0], w01[x][u][0], au[x][u], 0, 0, 0);
          }
        TT_STAMP(4)
        TT_STAMP(5)
        // lane (c = chain row in the block, q), registers j: i2 = 4 q + j — the k group of the 16x16x16 MFMA, for T2 as they are
        // (per r2: the two small terms on one accumulator, the large one on its own, the two r2 independent: two MFMA latencies on
        // the path instead of six)
        f32x4 alo[2], ahi[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          f32x4 acc = au[x][0];
#pragma unroll
          for (int u = 1; u < B::NM1; ++u) acc += au[x][u];
          acc = acc * (un1w[x] * t01f);
          unsigned a0lo, a1lo, a0hi, a1hi;
          split_pair_h(acc[0], acc[1], a0lo, a1lo);
          split_pair_h(acc[2], acc[3], a0hi, a1hi);
          const xh4 a0 = __builtin_bit_cast(xh4, u32x2{a0lo, a0hi}), a1 = __builtin_bit_cast(xh4, u32x2{a1lo, a1hi});
          const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
          alo[x] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, g2x[x][0], z4, 0, 0, 0);
          ahi[x] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, g2x[x][0], z4, 0, 0, 0);
          alo[x] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, g2x[x][1], alo[x], 0, 0, 0);
        }
        const f32x4 acc2 = (ahi[0] + alo[0]) + (ahi[1] + alo[1]);
        // lane (c = j2, q), registers j: chain row 16 rb + 4 q + j -> hidden unit row * J2 + j2; slice = the wave's r2 pair
        if (c < F::J2) {
          float* dst = dhs + (wave / B::CT2) * H + (16 * (wave % B::CT2) + 4 * q) * F::J2 + c;
          const float us = un2w * u2;
#pragma unroll
          for (int j = 0; j < 4; ++j) dst[j * F::J2] = acc2[j] * us;
        }
      } else {
        xh8 bf[B::NM1][2];
#pragma unroll
        for (int u = 0; u < B::NM1; ++u)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            bf[u][p] = *reinterpret_cast<const xh8*>(img1h + p * PL1 + x_off<B::K1>(c, 32 * u + 8 * q));
#pragma unroll
        for (int x = 0; x < B::XF; ++x) {
          f32x4 au[B::NM1];                                        // one chain per k-block: three dependent MFMAs, not six
#pragma unroll
          for (int u = 0; u < B::NM1; ++u) {
            au[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            au[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][1], bf[u][0], au[u], 0, 0, 0);
            au[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][0], bf[u][1], au[u], 0, 0, 0);
            au[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][0], bf[u][0], au[u], 0, 0, 0);
          }
          f32x4 acc = au[0];
#pragma unroll
          for (int u = 1; u < B::NM1; ++u) acc += au[u];
          // lane (c = i2, q), registers j: features 16ft + 4q + j = (row2, r2 = r20 + j): four consecutive k of T2
          const int f0 = 16 * (wave + FAST_NW * x) + 4 * q;
          const int row2 = f0 / F::R2, r20 = f0 % F::R2;
          store_split4_h(img2h, PL2, x_off<B::K2>(row2, B::k2_of(c, r20)), acc * (un1[x] * t01f));
        }
      }
    }
    if constexpr (!WL) {
    TT_STAMP(4)
    lds_barrier();
    TT_STAMP(5)
    }
    // ---- T2: dh_{t-1}[row2][j2], pair = (column tile ct, k-block ub): one partial-sum slice per k-block -------------
    if constexpr (!WL) {
      // (r = 16: two pairs per wave — both pairs' products first, ONE guarded block of stores behind them: lesson 40)
      f32x4 acc2[B::XT2];
#pragma unroll
      for (int x = 0; x < B::XT2; ++x) {
        const int ub = (wave + FAST_NW * x) / B::CT2;
        const int row = 16 * ct + c;
        xh8 b2[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          b2[p] = *reinterpret_cast<const xh8*>(img2h + p * PL2 + x_off<B::K2>(row, 32 * ub + 8 * q));
        // the two small terms on one accumulator, the large one on its own: two MFMA latencies on the path instead of three
        const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[x][1], b2[0], z4, 0, 0, 0);
        const f32x4 ahi = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[x][0], b2[0], z4, 0, 0, 0);
        alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[x][0], b2[1], alo, 0, 0, 0);
        acc2[x] = ahi + alo;
      }
      // lane (c = row2 in the column tile, q), registers j: j2 = 4q + j (valid for q < 2): hidden = row2*J2 + j2
      if (q < 2) {
#pragma unroll
        for (int x = 0; x < B::XT2; ++x) {
          const int ub = (wave + FAST_NW * x) / B::CT2;
          *reinterpret_cast<f32x4*>(dhs + ub * H + (16 * ct + c) * F::J2 + 4 * q) = acc2[x] * (un2 * u2);
        }
      }
    }
    TT_STAMP(6)
    lds_barrier();
    TT_STAMP(7)
  };
  for (int t = T - 1; t >= 0; t -= 3) {
    step(t, ra0, rb0, do0, rb1, ra2, rb2, do2, xq0, xq2);
    if (t >= 1) step(t - 1, ra1, rb1, do1, rb2, ra0, rb0, do0, xq1, xq0);
    if (t >= 2) step(t - 2, ra2, rb2, do2, rb0, ra1, rb1, do1, xq2, xq1);
  }
  if (own) {
    if (bs.colmax) {
#pragma unroll
      for (int g = 0; g < 4; ++g) atomicMax(bs.colmax + g * H + hid, __float_as_uint(cmx[g]));
    }
    if (bs.part) {
      float* pp = bs.part + b * 2 * GH;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        pp[g * H + hid] = sxd[g];
        pp[GH + g * H + hid] = sdg[g];
      }
    }
  }
  if constexpr (DIAG) {
    if (lane == 0 && diag && b < 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) diag[(b * FAST_NW + wave) * 8 + i] = seg[i];
    }
  }
  if (own) {
    if (d_h0) {
      float v = 0.f;
#pragma unroll
      for (int sl = 0; sl < B::NM2; ++sl) v += dhs[sl * H + hid];
      d_h0[b * H + hid] = v;
    }
    if (d_c0) d_c0[b * H + hid] = dcs;
  }
}

// ---- LSTM, H = 256: the record-dependent half of the gate phase on the helper waves, one step AHEAD (round 5) ------------------
// In k_lstm_bwd_f10h waves 4-7 sit 540 of the step's 2 851 stamped cycles in the barrier behind the gate phase while waves 0-3
// walk ~75 dependent instructions per unit (profiles/r4/stamps_cfg2_f10bh.txt).  Half of that chain does not depend on the
// recurrence at all: with the forward pass's record (i, g, f, o, c_t, c_{t-1}) the gate gradients are LINEAR in (dh_t, dc_t):
//     dct = dcs + dht A            A  = o (1 - tanh^2 c_t)
//     p_i = dct P0                 P0 = g i (1 - i)            p_f = dct P1      P1 = c_{t-1} f (1 - f)
//     p_g = dct P2                 P2 = i (1 - g^2)            p_o = dht P3      P3 = tanh(c_t) o (1 - o)
//     dcs' = dct f
// Here waves 4-7 (thread 256 + u <-> unit u) own the records: during the gate phase of step t they turn record t-1 into
// (A, P0, P1, P2 | P3, f, d_out) — the tanh, the derivative factors, every global load of the kernel — and leave them in LDS;
// waves 0-3 read two 16-byte vectors and run seven multiply-adds, the wave maximum and the image stores.  The helper waves also
// take the by-products (column maxima, input_size == 1 sums: from the fp32 image they already read for the HBM row).  T01, T2,
// scales and barriers are k_lstm_bwd_f10h's.  Products are associated differently (dct (g i (1 - i)) instead of
// ((dct g) i) (1 - i)): gradients agree with the other reverse kernels to an fp32 ulp of a product, not bit for bit.
template <class S, bool DIAG>
__global__ void __launch_bounds__(FAST_NT) k_lstm_bwd_f10p(int Bn, int T, const float* __restrict__ c0,
                                                           const float* __restrict__ hdr, const xh8* __restrict__ wfrag,
                                                           const float* __restrict__ reserve,
                                                           const float* __restrict__ d_out,
                                                           const float* __restrict__ d_hT,
                                                           const float* __restrict__ d_cT, float* __restrict__ dg_in,
                                                           float* __restrict__ dg_hid, float* __restrict__ d_h0,
                                                           float* __restrict__ d_c0,
                                                           unsigned long long* __restrict__ diag, BwdStats bs) {
  static_assert(f10bh_ok<S>() && F10<S>::H == 256 && F10BH<S>::XF < 4, "H = 256 only: waves 4-7 are the helper waves");
  using F = F10<S>;
  using B = F10BH<S>;
  constexpr int H = F::H, GH = 4 * H;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ __attribute__((aligned(16))) float smax1[4];
  __shared__ float sl1[FAST_NW];
  constexpr int PL1 = F::I2 * B::K1, PL2 = F::ROWS2 * B::K2;
  float* dgf = reinterpret_cast<float*>(smem);                               // [4H] in HBM row order
  float* dhs = dgf + GH;                                                     // [NM2][H]
  _Float16* img1h = reinterpret_cast<_Float16*>(dhs + B::NM2 * H);           // dg's two fp16 pieces [2][I2][K1] (x_off)
  _Float16* img2h = img1h + 2 * PL1;                                         // dC2's two fp16 pieces [2][ROWS2][K2] (x_off)
  f32x4* facA = reinterpret_cast<f32x4*>(img2h + 2 * PL2);                   // [2 parities][H]: A, P0, P1, P2
  f32x4* facB = facA + 2 * H;                                                // [2 parities][H]: P3, f, d_out, -

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  xh8 w01[B::XF][B::NM1][2], w2t[B::XT2][2];
  f32x4 un1[B::XF], un2;
#pragma unroll
  for (int x = 0; x < B::XF; ++x) {
#pragma unroll
    for (int u = 0; u < B::NM1; ++u)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        w01[x][u][p] = wfrag[(size_t)(((wave + FAST_NW * x) * B::NM1 + u) * 2 + p) * 64 + lane];
    un1[x] = *reinterpret_cast<const f32x4*>(hdr + B::UN1 + 16 * (wave + FAST_NW * x) + 4 * q);
  }
  const int ct = wave % B::CT2;
#pragma unroll
  for (int x = 0; x < B::XT2; ++x)
#pragma unroll
    for (int p = 0; p < 2; ++p)
      w2t[x][p] = wfrag[(size_t)(B::FT * B::NM1 * 2 + ((wave + FAST_NW * x) / B::CT2) * 2 + p) * 64 + lane];
  un2 = *reinterpret_cast<const f32x4*>(hdr + B::UN2 + 4 * q);
  float maxl1;
  {
    float l = 0.f;
    for (int f = tid; f < F::K; f += FAST_NT) l = fmaxf(l, hdr[B::L1N + f] * hdr[B::UN1 + f]);
    l = wave_max(l);
    if (lane == 0) sl1[wave] = l;
    __syncthreads();
    maxl1 = sl1[0];
#pragma unroll
    for (int w = 1; w < FAST_NW; ++w) maxl1 = fmaxf(maxl1, sl1[w]);
  }

  const bool own = tid < H;                    // gate thread of unit tid; the others: helper thread of unit tid - H
  const int hid = own ? tid : tid - H;
  float dcs = (own && d_cT) ? d_cT[b * H + hid] : 0.f;
  const float c0v = (!own && c0) ? c0[b * H + hid] : 0.f;
  const float* dptr = d_out ? d_out : reserve;
  const float dscale = d_out ? 1.0f : 0.0f;
  const float* xptr = bs.x ? reinterpret_cast<const float*>(bs.x) : reserve;
  const float xscale = bs.x ? 1.0f : 0.0f;
  // helper threads: three rotating record sets (gates i,g,f,o | c_t | d_out | x), loads issued two steps ahead
  f32x4 ra0 = f32x4{0.f, 0.f, 0.f, 0.f}, ra1 = ra0, ra2 = ra0;
  float rb0 = 0.f, rb1 = 0.f, rb2 = 0.f, do0 = 0.f, do1 = 0.f, do2 = 0.f, xq0 = 0.f, xq1 = 0.f, xq2 = 0.f;
  f32x4 cmx = f32x4{0.f, 0.f, 0.f, 0.f}, sxd = cmx, sdg = cmx;      // helper thread: by-products of its unit (gates i, f, g, o)
  auto issue = [&](int t, f32x4& ra, float& rb, float& dq, float& xq) {      // loads of record t; clamped, unconditional
    const size_t bt = b * T + (t > 0 ? t : 0);
    ra = *reinterpret_cast<const f32x4*>(reserve + res_gate(bt, H, hid));
    rb = reserve[res_cell((size_t)Bn * T, bt, H, hid)];
    dq = dptr[bt * H + hid];
    xq = xptr[bt];
  };
  // record s -> the factors of step s (cn: c_{s-1}, the NEXT record's cell state), parity s & 1
  auto factors = [&](int s, const f32x4& ra, float cy, float cn, float dq) {
    const float ig = ra[0], gg = ra[1], fg = ra[2], og = ra[3];
    const float cprev = s > 0 ? cn : c0v;
    const float tc = ftanh(cy);
    const f32x4 fa = f32x4{og * (1.0f - tc * tc), gg * ig * (1.0f - ig), cprev * fg * (1.0f - fg), ig * (1.0f - gg * gg)};
    const f32x4 fb = f32x4{tc * og * (1.0f - og), fg, dq * dscale, 0.f};
    facA[(s & 1) * H + hid] = fa;
    facB[(s & 1) * H + hid] = fb;
  };
  if (own) {
    dhs[hid] = d_hT ? d_hT[b * H + hid] : 0.f;
#pragma unroll
    for (int sl = 1; sl < B::NM2; ++sl) dhs[sl * H + hid] = 0.f;
  } else if (T > 0) {
    issue(T - 1, ra0, rb0, do0, xq0);
    issue(T - 2, ra1, rb1, do1, xq1);
    issue(T - 3, ra2, rb2, do2, xq2);
    factors(T - 1, ra0, rb0, rb1, do0);          // step T-1's factors: the loop produces step t-1's during step t
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  lds_barrier();
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();

  // step t.  Helper threads: (rn, cyn, dqn) = record t-1, cnn = c_{t-2} (record t-2's cell state), xc = x_t (record t);
  // (fa, fb, fd, fx) = record t's own set, dead behind this step's by-products: refilled with record t-3
  auto step = [&](const int t, const f32x4& rn, const float& cyn, const float& cnn, const float& dqn, const float& xc,
                  f32x4& fa, float& fb, float& fd, float& fx) {
    const size_t bt = b * T + t;
    f32x4 pkeep = f32x4{0.f, 0.f, 0.f, 0.f};
    if (own) {
      // ---- G: seven multiply-adds per unit on the factors the helper waves left for this step -------------------------------
      const f32x4 qa = facA[(t & 1) * H + hid], qb = facB[(t & 1) * H + hid];
      float dht = qb[2];
#pragma unroll
      for (int sl = 0; sl < B::NM2; ++sl) dht += dhs[sl * H + hid];
      const float dct = fmaf(dht, qa[0], dcs);
      const f32x4 pv = f32x4{dct * qa[1], dct * qa[2], dct * qa[3], dht * qb[0]};
      dcs = dct * qb[1];
      dgf[hid] = pv[0]; dgf[H + hid] = pv[1]; dgf[2 * H + hid] = pv[2]; dgf[3 * H + hid] = pv[3];
      pkeep = pv;
      float mx = fmaxf(fmaxf(fabsf(pv[0]), fabsf(pv[1])), fmaxf(fabsf(pv[2]), fabsf(pv[3])));
      mx = wave_max(mx);
      if (lane == 0) smax1[wave] = mx;
    } else if (t > 0) {
      factors(t - 1, rn, cyn, cnn, dqn);
    }
    TT_STAMP(0)
    lds_barrier();
    TT_STAMP(1)
    float u2, t01f;
    {
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(smax1);
      const float mxg = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      if (bs.rowmax && tid == FAST_NT - 1) bs.rowmax[bt] = mxg;
      float ug;
      const float sg = step_scale(mxg, ug);
      const float s2 = step_scale(mxg * maxl1, u2);
      t01f = ug * s2;
      if (own) {
        store_split4_h(img1h, PL1, x_off<B::K1>(hid % F::I2, 4 * (hid / F::I2)), pkeep * sg);
      } else {
        // the fp32 row goes out to HBM; the unit's by-products from the same image
        const int i4 = tid - H;
        const f32x4 v = reinterpret_cast<const f32x4*>(dgf)[i4];
        reinterpret_cast<f32x4*>(dg_in + bt * GH)[i4] = v;
        if (dg_hid && dg_hid != dg_in) reinterpret_cast<f32x4*>(dg_hid + bt * GH)[i4] = v;
        const f32x4 pu = f32x4{dgf[hid], dgf[H + hid], dgf[2 * H + hid], dgf[3 * H + hid]};
        const float xv = xc * xscale;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          cmx[g] = fmaxf(cmx[g], fabsf(pu[g]));
          sxd[g] = fmaf(xv, pu[g], sxd[g]);
          sdg[g] += pu[g];
        }
      }
    }
    TT_STAMP(2)
    lds_barrier();
    TT_STAMP(3)
    if (!own) issue(t - 3, fa, fb, fd, fx);      // under T01's MFMAs, into the set record t just left; turned into factors during step t-2
    // ---- T01 (k_lstm_bwd_f10h) --------------------------------------------------------------------------------------------------
    {
      xh8 bf[B::NM1][2];
#pragma unroll
      for (int u = 0; u < B::NM1; ++u)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          bf[u][p] = *reinterpret_cast<const xh8*>(img1h + p * PL1 + x_off<B::K1>(c, 32 * u + 8 * q));
#pragma unroll
      for (int x = 0; x < B::XF; ++x) {
        f32x4 au[B::NM1];
#pragma unroll
        for (int u = 0; u < B::NM1; ++u) {
          au[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          au[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][1], bf[u][0], au[u], 0, 0, 0);
          au[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][0], bf[u][1], au[u], 0, 0, 0);
          au[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][0], bf[u][0], au[u], 0, 0, 0);
        }
        f32x4 acc = au[0];
#pragma unroll
        for (int u = 1; u < B::NM1; ++u) acc += au[u];
        const int f0 = 16 * (wave + FAST_NW * x) + 4 * q;
        const int row2 = f0 / F::R2, r20 = f0 % F::R2;
        store_split4_h(img2h, PL2, x_off<B::K2>(row2, B::k2_of(c, r20)), acc * (un1[x] * t01f));
      }
    }
    TT_STAMP(4)
    lds_barrier();
    TT_STAMP(5)
    // ---- T2 (k_lstm_bwd_f10h) ---------------------------------------------------------------------------------------------------
    {
      f32x4 acc2[B::XT2];
#pragma unroll
      for (int x = 0; x < B::XT2; ++x) {
        const int ub = (wave + FAST_NW * x) / B::CT2;
        const int row = 16 * ct + c;
        xh8 b2[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          b2[p] = *reinterpret_cast<const xh8*>(img2h + p * PL2 + x_off<B::K2>(row, 32 * ub + 8 * q));
        const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[x][1], b2[0], z4, 0, 0, 0);
        const f32x4 ahi = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[x][0], b2[0], z4, 0, 0, 0);
        alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[x][0], b2[1], alo, 0, 0, 0);
        acc2[x] = ahi + alo;
      }
      if (q < 2) {
#pragma unroll
        for (int x = 0; x < B::XT2; ++x) {
          const int ub = (wave + FAST_NW * x) / B::CT2;
          *reinterpret_cast<f32x4*>(dhs + ub * H + (16 * ct + c) * F::J2 + 4 * q) = acc2[x] * (un2 * u2);
        }
      }
    }
    TT_STAMP(6)
    lds_barrier();
    TT_STAMP(7)
  };
  // three rotating record sets: record t lives in set (T - 1 - t) % 3
  for (int t = T - 1; t >= 0; t -= 3) {
    step(t, ra1, rb1, rb2, do1, xq0, ra0, rb0, do0, xq0);
    if (t >= 1) step(t - 1, ra2, rb2, rb0, do2, xq1, ra1, rb1, do1, xq1);
    if (t >= 2) step(t - 2, ra0, rb0, rb1, do0, xq2, ra2, rb2, do2, xq2);
  }
  if (!own) {
    if (bs.colmax) {
#pragma unroll
      for (int g = 0; g < 4; ++g) atomicMax(bs.colmax + g * H + hid, __float_as_uint(cmx[g]));
    }
    if (bs.part) {
      float* pp = bs.part + b * 2 * GH;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        pp[g * H + hid] = sxd[g];
        pp[GH + g * H + hid] = sdg[g];
      }
    }
  }
  if constexpr (DIAG) {
    if (lane == 0 && diag && b < 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) diag[(b * FAST_NW + wave) * 8 + i] = seg[i];
    }
  }
  if (own) {
    if (d_h0) {
      float v = 0.f;
#pragma unroll
      for (int sl = 0; sl < B::NM2; ++sl) v += dhs[sl * H + hid];
      d_h0[b * H + hid] = v;
    }
    if (d_c0) d_c0[b * H + hid] = dcs;
  }
}

// ---- LSTM, one wave per 64 hidden units (wave-local T2) ----------------------------------------------------------------------
// The same arithmetic as k_lstm_bwd_f10h with the work laid out so that TWO of its four barriers disappear: H / 64 waves (four at
// H = 256, eight at H = 512), every wave a gate wave for 64 CONSECUTIVE hidden units = eight rows row2 of dh.  T01's feature tiles
// are dealt out contiguously — wave w computes dC2 for the features (row2, r2) of exactly its own eight rows — so T2 (whose
// contraction runs over (i2, r2) of ONE row2) reads only what the same wave wrote, and its result dh is what the same wave's
// gate phase consumes next step: both hand-offs are in-wave LDS round trips (in order per wave, one lgkmcnt wait).  T2's column
// tile is half empty (eight rows of sixteen): 3 NM2 MFMAs per wave and step, G2's fragments read from LDS.  What still crosses
// the waves: the step's exact maximum (the operand scale) and the dg image, T01's operand — two barriers with the split of the
// thread's own four values between them.  The fp32 row of d_gates goes out from the gate threads directly (four coalesced dword
// stores); the record prefetch is issued behind the second barrier, under T01's MFMAs.  Four-wave workgroups run two per CU.
template <class S>
struct F10BL {
  using F = F10<S>;
  using B = F10BH<S>;
  static constexpr int NWV = F::H / 64, NT = NWV * 64;
  static constexpr int XF = B::FT / NWV;                    // feature tiles per wave (contiguous)
  static constexpr int RW = 16 * XF / F::R2;                // rows row2 per wave
};

template <class S>
constexpr bool f10bl_ok() {
  using F = F10<S>;
  using B = F10BH<S>;
  using L = F10BL<S>;
  return f10_ok<S>() && F::I2 == 16 && F::J2 == 8 && F::H % 64 == 0 && (L::NWV == 4 || L::NWV == 6 || L::NWV == 8) && B::FT % L::NWV == 0 &&
         L::RW == 8 && L::RW * L::NWV == F::ROWS2 && B::K1 % 32 == 0 && B::K2 % 32 == 0 && (16 * L::XF) % F::R2 == 0 && F::R2 % 4 == 0;
}

template <class S>
constexpr size_t f10bl_lds_bytes() {
  using B = F10BH<S>;
  using F = F10<S>;
  return sizeof(float) * ((size_t)B::H + F::K) + sizeof(_Float16) * 2 * ((size_t)F::I2 * B::K1 + (size_t)F::ROWS2 * B::K2) +
         sizeof(xh8) * (size_t)B::NM2 * 2 * 64;
}

template <class S, bool DIAG>
__global__ void __launch_bounds__(F10BL<S>::NT, F10BL<S>::NWV == 4 ? 2 : 1)
    k_lstm_bwd_f10l(int Bn, int T, const float* __restrict__ c0, const float* __restrict__ hdr, const xh8* __restrict__ wfrag,
                    const float* __restrict__ reserve, const float* __restrict__ d_out, const float* __restrict__ d_hT,
                    const float* __restrict__ d_cT, float* __restrict__ dg_in, float* __restrict__ dg_hid,
                    float* __restrict__ d_h0, float* __restrict__ d_c0, unsigned long long* __restrict__ diag, BwdStats bs) {
  static_assert(f10bl_ok<S>(), "shape not supported by the wave-local fused-core reverse-time kernel");
  using F = F10<S>;
  using B = F10BH<S>;
  using L = F10BL<S>;
  constexpr int H = F::H, GH = 4 * H, NWV = L::NWV, NT = L::NT, XF = L::XF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ __attribute__((aligned(16))) float smax1[(NWV + 3) / 4 * 4];      // (six waves: two padding entries, zero)
  __shared__ float sl1[NWV];
  constexpr int PL1 = F::I2 * B::K1, PL2 = F::ROWS2 * B::K2;
  if (threadIdx.x < (NWV + 3) / 4 * 4) smax1[threadIdx.x] = 0.f;
  float* dhs = reinterpret_cast<float*>(smem);                               // [H]: dh_{t-1}, written and read by the owning wave
  float* un1s = dhs + H;                                                     // [F::K] T01's inverse row scales
  _Float16* img1h = reinterpret_cast<_Float16*>(un1s + F::K);                // dg's two fp16 pieces [2][I2][K1] (x_off)
  _Float16* img2h = img1h + 2 * PL1;                                         // dC2's two fp16 pieces [2][ROWS2][K2] (x_off)
  xh8* w2s = reinterpret_cast<xh8*>(img2h + 2 * PL2);                        // G2's fragments [NM2][2][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  xh8 w01[XF][B::NM1][2];
#pragma unroll
  for (int x = 0; x < XF; ++x)
#pragma unroll
    for (int u = 0; u < B::NM1; ++u)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        w01[x][u][p] = wfrag[(size_t)(((XF * wave + x) * B::NM1 + u) * 2 + p) * 64 + lane];
  for (int e = tid; e < B::NM2 * 2 * 64; e += NT) w2s[e] = wfrag[(size_t)B::FT * B::NM1 * 2 * 64 + e];
  for (int f = tid; f < F::K; f += NT) un1s[f] = hdr[B::UN1 + f];
  const f32x4 un2 = *reinterpret_cast<const f32x4*>(hdr + B::UN2 + 4 * (q & 1));
  float maxl1;
  {
    float l = 0.f;
    for (int f = tid; f < F::K; f += NT) l = fmaxf(l, hdr[B::L1N + f] * hdr[B::UN1 + f]);
    l = wave_max(l);
    if (lane == 0) sl1[wave] = l;
    __syncthreads();
    maxl1 = sl1[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) maxl1 = fmaxf(maxl1, sl1[w]);
  }

  // gate phase: thread tid owns hidden unit tid; three rotating register sets (see k_lstm_bwd_f10)
  const int hid = tid;
  float dcs = d_cT ? d_cT[b * H + hid] : 0.f;
  const float c0v = c0 ? c0[b * H + hid] : 0.f;
  const float* dptr = d_out ? d_out : reserve;
  const float dscale = d_out ? 1.0f : 0.0f;
  f32x4 ra0 = f32x4{0.f, 0.f, 0.f, 0.f}, ra1 = ra0, ra2 = ra0;
  float rb0 = 0.f, rb1 = 0.f, rb2 = 0.f;
  float do0 = 0.f, do1 = 0.f, do2 = 0.f;
  const float* xptr = bs.x ? reinterpret_cast<const float*>(bs.x) : reserve;
  const float xscale = bs.x ? 1.0f : 0.0f;
  float xq0 = 0.f, xq1 = 0.f, xq2 = 0.f;
  f32x4 cmx = f32x4{0.f, 0.f, 0.f, 0.f}, sxd = cmx, sdg = cmx;
  dhs[hid] = d_hT ? d_hT[b * H + hid] : 0.f;
  if (T > 0) {
    const size_t bt = b * T + (T - 1);
    const float* rv = reserve + res_gate(bt, H, hid);
    const float* rc = reserve + res_cell((size_t)Bn * T, bt, H, hid);
    ra0 = *reinterpret_cast<const f32x4*>(rv);
    rb0 = rc[0];
    do0 = dptr[bt * H + hid];
    xq0 = xptr[bt];
    if (T > 1) {
      ra1 = *reinterpret_cast<const f32x4*>(rv - H * 4);
      rb1 = rc[-H];
      do1 = dptr[(bt - 1) * H + hid];
      xq1 = xptr[bt - 1];
    }
  }
  const bool two_rows = dg_hid && dg_hid != dg_in;
  // this thread's places: its four gate values in the dg image, its T2 operand row and result slot
  const int off1 = x_off<B::K1>(hid % F::I2, 4 * (hid / F::I2));
  const int row2r = L::RW * wave + (c & 7);
  float* dh_dst = dhs + (L::RW * wave + (c & 7)) * F::J2 + 4 * (q & 1);
  const bool t2_store = c < 8 && q < 2;
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): keep weight-register waits out of the time loop
  lds_barrier();
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();

  auto step = [&](const int t, const f32x4& ra, const float& rb, const float& dout_c, const float& nb, f32x4& fa,
                  float& fb, float& dout_f, const float& x_c, float& x_f) {
    const size_t bt = b * T + t;
    // ---- G: gate gradients (lstm.py:26-32 differentiated); dh_{t-1} of this unit came from this wave's own T2 --------
    const f32x4 qa = ra;
    const float dht = fmaf(dout_c, dscale, dhs[hid]);
    const float ig = qa[0], gg = qa[1], fg = qa[2], og = qa[3], cy = rb;
    const float cprev = t > 0 ? nb : c0v;
    const float tc = ftanh(cy);
    const float dct = dcs + dht * og * (1.0f - tc * tc);
    const float p0 = dct * gg * ig * (1.0f - ig);             // d pre-activation of i
    const float p1 = dct * cprev * fg * (1.0f - fg);          //                     f
    const float p2 = dct * ig * (1.0f - gg * gg);             //                     g
    const float p3 = dht * tc * og * (1.0f - og);             //                     o
    dcs = dct * fg;
    const f32x4 pv = f32x4{p0, p1, p2, p3};
    {
      float* row = dg_in + bt * GH + hid;
      row[0] = p0; row[H] = p1; row[2 * H] = p2; row[3 * H] = p3;
      if (two_rows) {
        float* row2 = dg_hid + bt * GH + hid;
        row2[0] = p0; row2[H] = p1; row2[2 * H] = p2; row2[3 * H] = p3;
      }
    }
    const float xv = x_c * xscale;
    float mx = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float a = fabsf(pv[g]);
      mx = fmaxf(mx, a);
      cmx[g] = fmaxf(cmx[g], a);
      sxd[g] = fmaf(xv, pv[g], sxd[g]);
      sdg[g] += pv[g];
    }
    mx = wave_max(mx);
    if (lane == 0) smax1[wave] = mx;
    TT_STAMP(0)
    lds_barrier();
    TT_STAMP(1)
    // ---- split: the step's scale from the exact maximum; every thread splits its own four values -----------------------
    float u2, t01f;
    {
      f32x4 m4 = *reinterpret_cast<const f32x4*>(smax1);
      if constexpr (NWV > 4) {
        const f32x4 m5 = *reinterpret_cast<const f32x4*>(smax1 + 4);
        m4 = f32x4{fmaxf(m4[0], m5[0]), fmaxf(m4[1], m5[1]), fmaxf(m4[2], m5[2]), fmaxf(m4[3], m5[3])};
      }
      const float mxg = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      if (bs.rowmax && tid == NT - 1) bs.rowmax[bt] = mxg;
      float ug;
      const float sg = step_scale(mxg, ug);
      const float s2 = step_scale(mxg * maxl1, u2);              // |dC2| <= maxl1 * mxg: no overflow, whatever the signs
      t01f = ug * s2;
      store_split4_h(img1h, PL1, off1, pv * sg);
    }
    TT_STAMP(2)
    lds_barrier();
    TT_STAMP(3)
    {
      // record(t-2), d_out(t-2), x(t-2): requested here, under T01's MFMAs, consumed two steps from now.  Always four loads, no
      // branch (index clamped; a null d_out / x reads the reserve and is scaled by zero)
      const size_t b2 = t > 1 ? bt - 2 : b * T;
      fa = *reinterpret_cast<const f32x4*>(reserve + res_gate(b2, H, hid));
      fb = reserve[res_cell((size_t)Bn * T, b2, H, hid)];
      dout_f = dptr[b2 * H + hid];
      x_f = xptr[b2];
    }
    // ---- T01: dC2 = W10 dg for this wave's own rows, rescaled for T2 and split into its operand image ------------------------
    // (k-block outermost inside a group of four tiles: one pair of operand fragments and four accumulators live at a time)
    {
      constexpr int XG = XF > 4 ? 4 : XF;
      static_assert(XF % XG == 0, "tile groups");
#pragma unroll
      for (int x0 = 0; x0 < XF; x0 += XG) {
        f32x4 acc[XG];
#pragma unroll
        for (int x = 0; x < XG; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < B::NM1; ++u) {
          const xh8 b0 = *reinterpret_cast<const xh8*>(img1h + x_off<B::K1>(c, 32 * u + 8 * q));
          const xh8 b1 = *reinterpret_cast<const xh8*>(img1h + PL1 + x_off<B::K1>(c, 32 * u + 8 * q));
#pragma unroll
          for (int x = 0; x < XG; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x0 + x][u][1], b0, acc[x], 0, 0, 0);
#pragma unroll
          for (int x = 0; x < XG; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x0 + x][u][0], b1, acc[x], 0, 0, 0);
#pragma unroll
          for (int x = 0; x < XG; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x0 + x][u][0], b0, acc[x], 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < XG; ++x) {
          // lane (c = i2, q), registers j: features 16ft + 4q + j = (row2, r2 = r20 + j): four consecutive k of T2
          const int f0 = 16 * (XF * wave + x0 + x) + 4 * q;
          const int row2 = f0 / F::R2, r20 = f0 % F::R2;
          const f32x4 u1 = *reinterpret_cast<const f32x4*>(un1s + f0);
          store_split4_h(img2h, PL2, x_off<B::K2>(row2, B::k2_of(c, r20)), acc[x] * (u1 * t01f));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's own rows of the dC2 image; nobody else reads them
    TT_STAMP(4)
    // ---- T2: dh_{t-1}[row2][j2] for the wave's eight rows (columns 8-15 of the tile repeat them, unused) ----------------------
    {
      f32x4 alo = f32x4{0.f, 0.f, 0.f, 0.f}, ahi = alo;
#pragma unroll
      for (int u = 0; u < B::NM2; ++u) {
        const xh8 b0 = *reinterpret_cast<const xh8*>(img2h + x_off<B::K2>(row2r, 32 * u + 8 * q));
        const xh8 b1 = *reinterpret_cast<const xh8*>(img2h + PL2 + x_off<B::K2>(row2r, 32 * u + 8 * q));
        const xh8 a0 = w2s[(u * 2 + 0) * 64 + lane], a1 = w2s[(u * 2 + 1) * 64 + lane];
        alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, alo, 0, 0, 0);
        ahi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, ahi, 0, 0, 0);
        alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, alo, 0, 0, 0);
      }
      // lane (c = row2 of the wave, q), registers j: j2 = 4q + j (q < 2): hidden = row2*J2 + j2
      if (t2_store) *reinterpret_cast<f32x4*>(dh_dst) = (ahi + alo) * (un2 * u2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // ... read back by this wave's gate threads
    TT_STAMP(5)
  };
  for (int t = T - 1; t >= 0; t -= 3) {
    step(t, ra0, rb0, do0, rb1, ra2, rb2, do2, xq0, xq2);
    if (t >= 1) step(t - 1, ra1, rb1, do1, rb2, ra0, rb0, do0, xq1, xq0);
    if (t >= 2) step(t - 2, ra2, rb2, do2, rb0, ra1, rb1, do1, xq2, xq1);
  }
  if (bs.colmax) {
#pragma unroll
    for (int g = 0; g < 4; ++g) atomicMax(bs.colmax + g * H + hid, __float_as_uint(cmx[g]));
  }
  if (bs.part) {
    float* pp = bs.part + b * 2 * GH;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      pp[g * H + hid] = sxd[g];
      pp[GH + g * H + hid] = sdg[g];
    }
  }
  if constexpr (DIAG) {
    if (lane == 0 && diag && b < 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) diag[(b * FAST_NW + wave) * 8 + i] = seg[i];
    }
  }
  if (d_h0) d_h0[b * H + hid] = dhs[hid];
  if (d_c0) d_c0[b * H + hid] = dcs;
}

// ---- GRU -----------------------------------------------------------------------------------------------------------
// The same scheme for the TT-GRU (fp32 or bf16 storage; reserve, gate gradients and all arithmetic fp32; gru.py:38-44
// differentiated as in k_gru_bwd_f10): three gates, I2 = 12 columns (padded to a tile), natural k order of T01's operand —
// a unit's gates are single elements of the image (six 2-byte stores per thread), d_gates_in differs from d_gates_hid in the
// n block (written by the gate threads directly), the direct path dh_{t-1} += dh_t z stays in the thread's register.
template <class S>
constexpr bool f10bh_gru_ok() {
  using F = F10<S>;
  using B = F10BH<S>;
  return S::D == 3 && S::R[0] == 1 && S::R[3] == 1 && shape_ok_recurrent<S>() && (F::H == 256 || F::H == 512) &&
         out_size_of<S>() == 3 * F::H && F::I2 % 2 == 0 && F::I2 <= 16 && B::K1 % 32 == 0 && B::K1 <= 128 && B::FT % FAST_NW == 0 &&
         B::K2 % 32 == 0 && (B::CT2 == 2 || B::CT2 == 4) && F::J2 == 8 && S::R[2] % 4 == 0 && B::NM2 * B::CT2 <= 2 * FAST_NW;
}

template <class S>
constexpr size_t f10bh_gru_lds_bytes() {
  using B = F10BH<S>;
  using F = F10<S>;
  return sizeof(float) * ((size_t)3 * B::H + (size_t)B::NM2 * B::H) +
         sizeof(_Float16) * 2 * ((size_t)16 * B::K1 + (size_t)F::ROWS2 * B::K2);
}

template <class S, typename TS>
__global__ void __launch_bounds__(FAST_NT) k_gru_bwd_f10h(int Bn, int T, const TS* __restrict__ out,
                                                          const TS* __restrict__ h0, const float* __restrict__ hdr,
                                                          const xh8* __restrict__ wfrag, const float* __restrict__ reserve,
                                                          const TS* __restrict__ d_out, const TS* __restrict__ d_hT,
                                                          float* __restrict__ dg_in, float* __restrict__ dg_hid,
                                                          TS* __restrict__ d_h0, BwdStats bs) {
  static_assert(f10bh_gru_ok<S>(), "shape not supported by the two-piece fused-core GRU reverse-time kernel");
  using F = F10<S>;
  using B = F10BH<S>;
  constexpr int H = F::H, GH = 3 * H;
  constexpr int NP = B::NM2 * B::CT2;                                         // T2 (column tile, k-block) pairs: pair p = wave + 8 i
  constexpr int NPW = (NP + FAST_NW - 1) / FAST_NW;                           // ... one per wave at r = 8 (six pairs), up to two at r = 16 (twelve)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ __attribute__((aligned(16))) float smax1[8];
  __shared__ float sl1[FAST_NW];
  constexpr int PL1 = 16 * B::K1, PL2 = F::ROWS2 * B::K2;
  constexpr bool HELPERS = GH / 4 <= FAST_NT - H;      // H = 256: waves 4-7 store the fp32 rows; H = 512: every wave is a gate wave
  constexpr int NGW = H / 64;                           // gate waves
  float* dgf = reinterpret_cast<float*>(smem);                               // [3H]: dr, dz, dn*r (hidden chain)
  float* dhs = dgf + GH;                                                     // [NM2][H]
  _Float16* img1h = reinterpret_cast<_Float16*>(dhs + B::NM2 * H);           // [2][16][K1], natural k
  _Float16* img2h = img1h + 2 * PL1;                                         // [2][ROWS2][K2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  xh8 w01[B::XF][B::NM1][2], w2t[NPW][2];
  f32x4 un1[B::XF], un2;
#pragma unroll
  for (int x = 0; x < B::XF; ++x) {
#pragma unroll
    for (int u = 0; u < B::NM1; ++u)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        w01[x][u][p] = wfrag[(size_t)(((wave + FAST_NW * x) * B::NM1 + u) * 2 + p) * 64 + lane];
    un1[x] = *reinterpret_cast<const f32x4*>(hdr + B::UN1 + 16 * (wave + FAST_NW * x) + 4 * q);
  }
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pid = wave + FAST_NW * i < NP ? wave + FAST_NW * i : 0, ub = pid / B::CT2;
#pragma unroll
    for (int p = 0; p < 2; ++p) w2t[i][p] = wfrag[(size_t)(B::FT * B::NM1 * 2 + ub * 2 + p) * 64 + lane];
  }
  un2 = *reinterpret_cast<const f32x4*>(hdr + B::UN2 + 4 * q);
  float maxl1;
  {
    float l = 0.f;
    for (int f = tid; f < F::K; f += FAST_NT) l = fmaxf(l, hdr[B::L1N + f] * hdr[B::UN1 + f]);
    l = wave_max(l);
    if (lane == 0) sl1[wave] = l;
    __syncthreads();
    maxl1 = sl1[0];
#pragma unroll
    for (int w = 1; w < FAST_NW; ++w) maxl1 = fmaxf(maxl1, sl1[w]);
  }
  // columns 12..15 of T01's operand are never written by the gate threads: zeros, once (their results are never stored)
  for (int e = tid; e < 2 * PL1; e += FAST_NT) img1h[e] = (_Float16)0.f;

  const bool own = tid < H;
  const int hid = own ? tid : 0;
  float dhd = 0.f;
  const TS* dptr = d_out ? d_out : out;
  const float dscale = d_out ? 1.0f : 0.0f;
  f32x4 ra0 = f32x4{0.f, 0.f, 0.f, 0.f}, ra1 = ra0, ra2 = ra0;
  TS do0 = TS{}, do1 = TS{}, do2 = TS{}, hp0 = TS{}, hp1 = TS{}, hp2 = TS{};
  const TS* xptr = bs.x ? reinterpret_cast<const TS*>(bs.x) : out;
  const float xscale = bs.x ? 1.0f : 0.0f;
  TS xq0 = TS{}, xq1 = TS{}, xq2 = TS{};
  float cmi[3] = {0.f, 0.f, 0.f}, sxd[3] = {0.f, 0.f, 0.f}, sdg[3] = {0.f, 0.f, 0.f}, cmh = 0.f;
  auto issue = [&](int t, f32x4& ra, TS& dq, TS& hq, TS& xq) {       // loads of set(t); clamped, unconditional
    const size_t bt = b * T + (t > 0 ? t : 0);
    ra = *reinterpret_cast<const f32x4*>(reserve + (bt * H + hid) * 4);
    dq = dptr[bt * H + hid];
    const TS* hp = t >= 1 ? out + (bt - 1) * H : (h0 ? h0 + b * H : out + bt * H);
    hq = hp[hid];
    xq = xptr[bt];
  };
  if (own) {
    dhs[hid] = d_hT ? ld(d_hT, b * H + hid) : 0.f;
#pragma unroll
    for (int sl = 1; sl < B::NM2; ++sl) dhs[sl * H + hid] = 0.f;
    if (T > 0) {
      issue(T - 1, ra0, do0, hp0, xq0);
      issue(T - 2, ra1, do1, hp1, xq1);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  lds_barrier();

  auto step = [&](const int t, const f32x4& ra, const TS& dq, const TS& hq, const TS& xq, f32x4& fa, TS& fd, TS& fh,
                  TS& fx) {
    const size_t bt = b * T + t;
    float pk[3] = {0.f, 0.f, 0.f};
    // ---- G: gate gradients (gru.py:38-44 differentiated) -----------------------------------------------------------
    if (own) {
      issue(t - 2, fa, fd, fh, fx);
      float dht = dhd + to_f32(dq) * dscale;
#pragma unroll
      for (int sl = 0; sl < B::NM2; ++sl) dht += dhs[sl * H + hid];
      const float rg = ra[0], zg = ra[1], ng = ra[2], hn = ra[3];
      const float hprev = (t > 0 || h0) ? to_f32(hq) : 0.f;
      const float dn_pre = dht * (1.0f - zg) * (1.0f - ng * ng);
      const float dz_pre = dht * (hprev - ng) * zg * (1.0f - zg);
      const float dr_pre = dn_pre * hn * rg * (1.0f - rg);
      dhd = dht * zg;
      pk[0] = dr_pre; pk[1] = dz_pre; pk[2] = dn_pre * rg;
      const float pin[3] = {dr_pre, dz_pre, dn_pre};
      const float xv = to_f32(xq) * xscale;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        cmi[g] = fmaxf(cmi[g], fabsf(pin[g]));
        sxd[g] = fmaf(xv, pin[g], sxd[g]);
        sdg[g] += pin[g];
        dgf[g * H + hid] = pk[g];
      }
      cmh = fmaxf(cmh, fabsf(pk[2]));
      dg_in[bt * GH + 2 * H + hid] = dn_pre;                // the only block where d_gates_in != d_gates_hid
      const float mx = wave_max(fmaxf(fmaxf(fabsf(pk[0]), fabsf(pk[1])), fabsf(pk[2])));
      if (lane == 0) smax1[wave] = mx;
    }
    lds_barrier();
    // ---- split: the step's scale; the gate waves split their own three values; waves 4-7 send the fp32 rows to HBM ------
    float u2, t01f;
    {
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(smax1);
      float mxg = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      if constexpr (NGW == 8) {
        const f32x4 m5 = *reinterpret_cast<const f32x4*>(smax1 + 4);
        mxg = fmaxf(mxg, fmaxf(fmaxf(m5[0], m5[1]), fmaxf(m5[2], m5[3])));
      }
      float ug;
      const float sg = step_scale(mxg, ug);
      const float s2 = step_scale(mxg * maxl1, u2);
      t01f = ug * s2;
      if (own) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const int o = g * H + hid, m = o / F::I2, i2 = o % F::I2;     // flat gate index -> (m, i2), natural k = m
          _Float16 p0, p1;
          split2h(pk[g] * sg, p0, p1);
          const int off = x_off<B::K1>(i2, m);
          img1h[off] = p0;
          img1h[PL1 + off] = p1;
        }
      }
      {
        // the fp32 rows to HBM: the helper waves (H = 256), or — every wave being a gate wave — threads 0 .. 3H/4 - 1 behind their split
        const int i4 = HELPERS ? tid - H : tid;
        if (i4 >= 0 && i4 < GH / 4) {
          const f32x4 v = reinterpret_cast<const f32x4*>(dgf)[i4];
          reinterpret_cast<f32x4*>(dg_hid + bt * GH)[i4] = v;
          if (i4 < 2 * H / 4) reinterpret_cast<f32x4*>(dg_in + bt * GH)[i4] = v;
        }
      }
    }
    lds_barrier();
    // ---- T01 ---------------------------------------------------------------------------------------------------------
    {
      const int rowc = c < F::I2 ? c : F::I2;                // (rows I2..15 of the image are zeros)
      xh8 bf[B::NM1][2];
#pragma unroll
      for (int u = 0; u < B::NM1; ++u)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          bf[u][p] = *reinterpret_cast<const xh8*>(img1h + p * PL1 + x_off<B::K1>(rowc, 32 * u + 8 * q));
      // all products first, ONE guarded block of stores behind them (a guard per tile puts a branch — and the wait for that
      // tile's MFMAs — between the tiles: lesson 40)
      // (K1 <= 64: one accumulator per k-block, added behind the MFMAs — the order the rank-8 kernel has always summed in; K1 = 96
      // (H = 512): the k-blocks accumulate in place, four accumulators live instead of twelve)
      constexpr int NAU = B::NM1 <= 2 ? B::NM1 : 1;
      f32x4 au[B::XF][NAU];
#pragma unroll
      for (int x = 0; x < B::XF; ++x)
#pragma unroll
        for (int u = 0; u < B::NM1; ++u) {
          const int ua = NAU == 1 ? 0 : u;
          if (NAU > 1 || u == 0) au[x][ua] = f32x4{0.f, 0.f, 0.f, 0.f};
          au[x][ua] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][1], bf[u][0], au[x][ua], 0, 0, 0);
          au[x][ua] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][0], bf[u][1], au[x][ua], 0, 0, 0);
          au[x][ua] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[x][u][0], bf[u][0], au[x][ua], 0, 0, 0);
        }
      if (c < F::I2) {
#pragma unroll
        for (int x = 0; x < B::XF; ++x) {
          f32x4 acc = au[x][0];
#pragma unroll
          for (int u = 1; u < NAU; ++u) acc += au[x][u];
          const int f0 = 16 * (wave + FAST_NW * x) + 4 * q;
          const int row2 = f0 / F::R2, r20 = f0 % F::R2;
          store_split4_h(img2h, PL2, x_off<B::K2>(row2, B::k2_of(c, r20)), acc * (un1[x] * t01f));
        }
      }
    }
    lds_barrier();
    // ---- T2: pairs (column tile ct, k-block ub) = wave, wave + 8 (< NP) ----------------------------------------------------
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pid = wave + FAST_NW * i;
      if (pid < NP) {
        const int ub = pid / B::CT2, ct = pid % B::CT2;
        const int row = 16 * ct + c;
        xh8 b2[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          b2[p] = *reinterpret_cast<const xh8*>(img2h + p * PL2 + x_off<B::K2>(row, 32 * ub + 8 * q));
        const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[i][1], b2[0], z4, 0, 0, 0);
        const f32x4 ahi = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[i][0], b2[0], z4, 0, 0, 0);
        alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[i][0], b2[1], alo, 0, 0, 0);
        const f32x4 acc = ahi + alo;
        if (q < 2) *reinterpret_cast<f32x4*>(dhs + ub * H + row * F::J2 + 4 * q) = acc * (un2 * u2);
      }
    }
    lds_barrier();
  };
  for (int t = T - 1; t >= 0; t -= 3) {
    step(t, ra0, do0, hp0, xq0, ra2, do2, hp2, xq2);
    if (t >= 1) step(t - 1, ra1, do1, hp1, xq1, ra0, do0, hp0, xq0);
    if (t >= 2) step(t - 2, ra2, do2, hp2, xq2, ra1, do1, hp1, xq1);
  }
  if (own) {
    if (bs.colmax) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        atomicMax(bs.colmax + g * H + hid, __float_as_uint(cmi[g]));
        atomicMax(bs.colmax + GH + g * H + hid, __float_as_uint(g < 2 ? cmi[g] : cmh));
      }
    }
    if (bs.part) {
      float* pp = bs.part + b * 2 * GH;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        pp[g * H + hid] = sxd[g];
        pp[GH + g * H + hid] = sdg[g];
      }
    }
  }
  if (own && d_h0) {
    float v = dhd;
#pragma unroll
    for (int sl = 0; sl < B::NM2; ++sl) v += dhs[sl * H + hid];
    st(d_h0, b * H + hid, v);
  }
}

// ---- naive per-gate sets (tt_linearset.py:5-38; `--naive_tt [--gru]`): G independent TT-matrices H -> H, one per gate ------------
// The reverse recurrence dh_{t-1} = sum_g W_g^T dz_g on the per-gate fused cores (round 6; until then the tier's k_g2_bwd on the
// joint matrix): S = ONE gate's shape.  T01 runs per gate — feature tile ft of gate g against that gate's sixteen-row image
// [i2][m] (K1 = 32: one k-block) — and writes its results as k = (g, i2, r2) of chain row (j0, j1); T2 contracts over ALL gates'
// (i2, r2) in G * NM2 k-blocks, one (column tile, k-block) pair or two per wave, each k-block un-scaled by ITS gate's row scales and
// kept as a partial sum the gate threads add.  Per step: G | barrier | split | barrier | T01 | barrier | T2 | barrier, as the
// single-matrix kernels.  Fragments and scale headers: k_f10bh_prep<S, true> once per gate on the un-joined cores.
template <class S, int CELL>
__global__ void __launch_bounds__(FAST_NT) k_rnn_bwd_f10n(int Bn, int T, const float* __restrict__ out, const float* __restrict__ h0,
                                                          const float* __restrict__ c0, const float* __restrict__ hdrs,
                                                          const xh8* __restrict__ wfrags, const float* __restrict__ reserve,
                                                          const float* __restrict__ d_out, const float* __restrict__ d_hT,
                                                          const float* __restrict__ d_cT, float* __restrict__ dg_in,
                                                          float* __restrict__ dg_hid, float* __restrict__ d_h0,
                                                          float* __restrict__ d_c0, BwdStats bs) {
  using F = F10<S>;
  using B = F10BH<S>;
  constexpr bool LSTM = CELL == TTRNN_LSTM;
  constexpr int G = LSTM ? 4 : 3;
  constexpr int H = F::H, GH = G * H;
  constexpr int K2T = G * B::K2;                                              // T2's contraction over all gates
  constexpr int NS = G * B::NM2;                                              // its k-blocks = partial-sum slices
  constexpr int NP = NS * B::CT2, NPW = (NP + FAST_NW - 1) / FAST_NW;        // (column tile, k-block) pairs; per wave
  static_assert(B::NM1 == 1 && B::CT2 == 2 && B::XF == 2 && H == 256 && F::I2 == 8 && GH / 4 <= FAST_NT - H, "one gate of the naive sets of H = 256");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ __attribute__((aligned(16))) float smax1[4];
  __shared__ float sl1[FAST_NW];
  constexpr int PG1 = 16 * B::K1;                                             // a gate's T01 image (halves per plane)
  constexpr int PL1 = G * PG1, PL2 = F::ROWS2 * K2T;
  float* dgf = reinterpret_cast<float*>(smem);                               // [G H]: the gate gradients of the hidden chain, HBM row order
  float* dhs = dgf + GH;                                                     // [NS][H]
  _Float16* img1h = reinterpret_cast<_Float16*>(dhs + NS * H);               // [2][G][16][K1]
  _Float16* img2h = img1h + 2 * PL1;                                         // [2][ROWS2][K2T]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;
  constexpr int HF = B::HDR_FLOATS;
  constexpr size_t FR = B::FRAGS;

  xh8 w01[G][B::XF][2], w2t[NPW][2];
  f32x4 un1[G][B::XF], un2[NPW];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int x = 0; x < B::XF; ++x) {
#pragma unroll
      for (int p = 0; p < 2; ++p) w01[g][x][p] = wfrags[g * FR + (size_t)((wave + FAST_NW * x) * 2 + p) * 64 + lane];
      un1[g][x] = *reinterpret_cast<const f32x4*>(hdrs + g * HF + B::UN1 + 16 * (wave + FAST_NW * x) + 4 * q);
    }
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pid = wave + FAST_NW * i < NP ? wave + FAST_NW * i : 0;
    const int ubt = pid / B::CT2, g = ubt / B::NM2, ub = ubt % B::NM2;
#pragma unroll
    for (int p = 0; p < 2; ++p) w2t[i][p] = wfrags[g * FR + (size_t)(B::FT * B::NM1 * 2 + ub * 2 + p) * 64 + lane];
    un2[i] = *reinterpret_cast<const f32x4*>(hdrs + g * HF + B::UN2 + 4 * q);
  }
  float maxl1;
  {
    float l = 0.f;
    for (int f = tid; f < G * F::K; f += FAST_NT) {
      const int g = f / F::K, ff = f % F::K;
      l = fmaxf(l, hdrs[g * HF + B::L1N + ff] * hdrs[g * HF + B::UN1 + ff]);
    }
    l = wave_max(l);
    if (lane == 0) sl1[wave] = l;
    __syncthreads();
    maxl1 = sl1[0];
#pragma unroll
    for (int w = 1; w < FAST_NW; ++w) maxl1 = fmaxf(maxl1, sl1[w]);
  }
  // rows 8..15 of every gate's T01 image are never written by the gate threads: zeros, once
  for (int e = tid; e < 2 * PL1; e += FAST_NT) img1h[e] = (_Float16)0.f;

  const bool own = tid < H;
  const int hid = own ? tid : 0;
  float dhd = 0.f;                                                            // GRU: dh_t z_t; LSTM: unused
  float dcs = (LSTM && own && d_cT) ? d_cT[b * H + hid] : 0.f;
  const float c0v = (LSTM && own && c0) ? c0[b * H + hid] : 0.f;
  const float* dptr = d_out ? d_out : reserve;
  const float dscale = d_out ? 1.0f : 0.0f;
  f32x4 ra0 = f32x4{0.f, 0.f, 0.f, 0.f}, ra1 = ra0, ra2 = ra0;
  float rb0 = 0.f, rb1 = 0.f, rb2 = 0.f, do0 = 0.f, do1 = 0.f, do2 = 0.f;     // rb: LSTM c_t / GRU h_{t-1}
  float cmi[G], cmh = 0.f;
#pragma unroll
  for (int g = 0; g < G; ++g) cmi[g] = 0.f;
  auto issue = [&](int t, f32x4& ra, float& rb, float& dq) {                   // loads of set(t); clamped, unconditional
    const size_t bt = b * T + (t > 0 ? t : 0);
    ra = *reinterpret_cast<const f32x4*>(reserve + res_gate(bt, H, hid));
    dq = dptr[bt * H + hid];
    if (LSTM) {
      rb = reserve[res_cell((size_t)Bn * T, bt, H, hid)];
    } else {
      const float* hp = t >= 1 ? out + (bt - 1) * H : (h0 ? h0 + b * H : out + bt * H);
      rb = hp[hid];
    }
  };
  if (own) {
    dhs[hid] = d_hT ? d_hT[b * H + hid] : 0.f;
#pragma unroll
    for (int sl = 1; sl < NS; ++sl) dhs[sl * H + hid] = 0.f;
    if (T > 0) {
      issue(T - 1, ra0, rb0, do0);
      issue(T - 2, ra1, rb1, do1);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  lds_barrier();

  // nb: the record one step back (LSTM: c_{t-1} = set(t-1)'s cell value)
  auto step = [&](const int t, const f32x4& ra, const float& rb, const float& dq, const float& nb, f32x4& fa, float& fb, float& fd) {
    const size_t bt = b * T + t;
    float pk[G];
#pragma unroll
    for (int g = 0; g < G; ++g) pk[g] = 0.f;
    // ---- G: gate gradients (lstm.py:26-32 / gru.py:38-44 differentiated) ------------------------------------------------
    if (own) {
      issue(t - 2, fa, fb, fd);
      float dht = dhd + dq * dscale;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) dht += dhs[sl * H + hid];
      if constexpr (LSTM) {
        const float ig = ra[0], gg = ra[1], fg = ra[2], og = ra[3], cy = rb;     // record slots i, g, f, o
        const float cprev = t > 0 ? nb : c0v;
        const float tc = ftanh(cy);
        const float dct = dcs + dht * og * (1.0f - tc * tc);
        pk[0] = dct * gg * ig * (1.0f - ig);                                     // gate order of the rows: i, f, g, o
        pk[1] = dct * cprev * fg * (1.0f - fg);
        pk[2] = dct * ig * (1.0f - gg * gg);
        pk[3] = dht * tc * og * (1.0f - og);
        dcs = dct * fg;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          cmi[g] = fmaxf(cmi[g], fabsf(pk[g]));
          dgf[g * H + hid] = pk[g];
        }
      } else {
        const float rg = ra[0], zg = ra[1], ng = ra[2], hn = ra[3];
        const float hprev = (t > 0 || h0) ? rb : 0.f;
        const float dn_pre = dht * (1.0f - zg) * (1.0f - ng * ng);
        const float dz_pre = dht * (hprev - ng) * zg * (1.0f - zg);
        const float dr_pre = dn_pre * hn * rg * (1.0f - rg);
        dhd = dht * zg;
        pk[0] = dr_pre; pk[1] = dz_pre; pk[2] = dn_pre * rg;
        const float pin[3] = {dr_pre, dz_pre, dn_pre};
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          cmi[g] = fmaxf(cmi[g], fabsf(pin[g]));
          dgf[g * H + hid] = pk[g];
        }
        cmh = fmaxf(cmh, fabsf(pk[2]));
        dg_in[bt * GH + 2 * H + hid] = dn_pre;                // the only block where d_gates_in != d_gates_hid
      }
      float mx = fabsf(pk[0]);
#pragma unroll
      for (int g = 1; g < G; ++g) mx = fmaxf(mx, fabsf(pk[g]));
      mx = wave_max(mx);
      if (lane == 0) smax1[wave] = mx;
    }
    lds_barrier();
    // ---- split: the step's scale; the gate threads split their own values; waves 4-7 send the fp32 rows to HBM ------------
    float u2, t01f;
    {
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(smax1);
      const float mxg = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      float ug;
      const float sg = step_scale(mxg, ug);
      const float s2 = step_scale(mxg * maxl1, u2);
      t01f = ug * s2;
      if (own) {
        const int m = hid / F::I2, i2 = hid % F::I2;           // the unit's place in every gate's matrix: (m, i2), natural k = m
        const int off = x_off<B::K1>(i2, m);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          _Float16 p0, p1;
          split2h(pk[g] * sg, p0, p1);
          img1h[g * PG1 + off] = p0;
          img1h[PL1 + g * PG1 + off] = p1;
        }
      } else {
        const int i4 = tid - H;
        if (i4 < GH / 4) {
          const f32x4 v = reinterpret_cast<const f32x4*>(dgf)[i4];
          if (LSTM) {
            reinterpret_cast<f32x4*>(dg_in + bt * GH)[i4] = v;  // (an LSTM's d_gates_hid IS d_gates_in)
          } else {
            reinterpret_cast<f32x4*>(dg_hid + bt * GH)[i4] = v;
            if (i4 < 2 * H / 4) reinterpret_cast<f32x4*>(dg_in + bt * GH)[i4] = v;
          }
        }
      }
    }
    lds_barrier();
    // ---- T01: per gate, the wave's two feature tiles against the gate's image ------------------------------------------
    {
      const int rowc = c < F::I2 ? c : F::I2;                  // (rows 8..15 of a gate's image are zeros)
      f32x4 au[G][B::XF];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        xh8 bf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) bf[p] = *reinterpret_cast<const xh8*>(img1h + p * PL1 + g * PG1 + x_off<B::K1>(rowc, 8 * q));
#pragma unroll
        for (int x = 0; x < B::XF; ++x) {
          au[g][x] = f32x4{0.f, 0.f, 0.f, 0.f};
          au[g][x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[g][x][1], bf[0], au[g][x], 0, 0, 0);
          au[g][x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[g][x][0], bf[1], au[g][x], 0, 0, 0);
          au[g][x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01[g][x][0], bf[0], au[g][x], 0, 0, 0);
        }
      }
      if (c < F::I2) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int x = 0; x < B::XF; ++x) {
            const int f0 = 16 * (wave + FAST_NW * x) + 4 * q;
            const int row2 = f0 / F::R2, r20 = f0 % F::R2;
            store_split4_h(img2h, PL2, x_off<K2T>(row2, g * B::K2 + B::k2_of(c, r20)), au[g][x] * (un1[g][x] * t01f));
          }
      }
    }
    lds_barrier();
    // ---- T2: pairs (column tile ct, k-block ubt of all gates) = wave, wave + 8 (< NP) -----------------------------------
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pid = wave + FAST_NW * i;
      if (pid < NP) {
        const int ubt = pid / B::CT2, ct = pid % B::CT2;
        const int row = 16 * ct + c;
        xh8 b2[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          b2[p] = *reinterpret_cast<const xh8*>(img2h + p * PL2 + x_off<K2T>(row, 32 * ubt + 8 * q));
        const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[i][1], b2[0], z4, 0, 0, 0);
        const f32x4 ahi = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[i][0], b2[0], z4, 0, 0, 0);
        alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2t[i][0], b2[1], alo, 0, 0, 0);
        const f32x4 acc = ahi + alo;
        if (q < 2) *reinterpret_cast<f32x4*>(dhs + ubt * H + row * F::J2 + 4 * q) = acc * (un2[i] * u2);
      }
    }
    lds_barrier();
  };
  for (int t = T - 1; t >= 0; t -= 3) {
    step(t, ra0, rb0, do0, rb1, ra2, rb2, do2);
    if (t >= 1) step(t - 1, ra1, rb1, do1, rb2, ra0, rb0, do0);
    if (t >= 2) step(t - 2, ra2, rb2, do2, rb0, ra1, rb1, do1);
  }
  if (own) {
    if (bs.colmax) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        atomicMax(bs.colmax + g * H + hid, __float_as_uint(cmi[g]));
        atomicMax(bs.colmax + GH + g * H + hid, __float_as_uint((LSTM || g < 2) ? cmi[g] : cmh));
      }
    }
    if (d_h0) {
      float v = dhd;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) v += dhs[sl * H + hid];
      d_h0[b * H + hid] = v;
    }
    if (LSTM && d_c0) d_c0[b * H + hid] = dcs;
  }
}

template <class S, int CELL>
constexpr size_t f10n_bwd_lds_bytes() {
  using B = F10BH<S>;
  using F = F10<S>;
  constexpr int G = CELL == TTRNN_LSTM ? 4 : 3;
  return sizeof(float) * ((size_t)G * F::H + (size_t)G * B::NM2 * F::H) +
         sizeof(_Float16) * 2 * ((size_t)G * 16 * B::K1 + (size_t)F::ROWS2 * G * B::K2);
}

template <class S, int CELL>
int launch_n_bwd(const RnnShape& rs, const void* out, const void* h0, const void* c0, const float* packed_hid, const float* reserve,
                 const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid, void* d_h0, void* d_c0, void* ws,
                 hipStream_t stream, const BwdStats& bs) {
  using B = F10BH<S>;
  constexpr int G = CELL == TTRNN_LSTM ? 4 : 3;
  const size_t npg = f10n_unjoined_floats(CELL);
  float* pg = reinterpret_cast<float*>(ws);
  float* hdrs = pg + ((npg + 63) & ~(size_t)63);
  xh8* wfrags = reinterpret_cast<xh8*>(hdrs + (size_t)G * B::HDR_FLOATS);
  int st = launch_f10n_unjoin(rs, packed_hid, pg, stream);
  if (st != TTRNN_OK) return st;
  for (int g = 0; g < G; ++g)      // (the first launch clears the by-products' column maxima)
    hipLaunchKernelGGL((k_f10bh_prep<S, true>), dim3(B::FT + 1), dim3(FAST_NT), 0, stream, (const float*)(pg + (size_t)g * (npg / G)),
                       hdrs + (size_t)g * B::HDR_FLOATS, wfrags + (size_t)g * B::FRAGS, g == 0 ? bs.colmax : nullptr,
                       (g == 0 && bs.colmax) ? 2 * G * B::H : 0);
  constexpr size_t lds = f10n_bwd_lds_bytes<S, CELL>();
  static_assert(lds <= 64 * 1024, "raise the dynamic LDS limit for this shape");
  hipLaunchKernelGGL((k_rnn_bwd_f10n<S, CELL>), dim3(rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T, (const float*)out, (const float*)h0,
                     (const float*)c0, (const float*)hdrs, (const xh8*)wfrags, reserve, (const float*)d_out, (const float*)d_hT,
                     (const float*)d_cT, dg_in, dg_hid, (float*)d_h0, (float*)d_c0, bs);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

template <class S, typename TS>
int launch_gru_t(const RnnShape& rs, const void* out, const void* h0, const float* packed_hid, const float* reserve,
                 const void* d_out, const void* d_hT, float* dg_in, float* dg_hid, void* d_h0, void* ws,
                 hipStream_t stream, const BwdStats& bs) {
  using B = F10BH<S>;
  float* hdr = reinterpret_cast<float*>(ws);
  xh8* wfrag = reinterpret_cast<xh8*>(hdr + B::HDR_FLOATS);
  hipLaunchKernelGGL((k_f10bh_prep<S, true>), dim3(B::FT + 1), dim3(FAST_NT), 0, stream, packed_hid, hdr, wfrag, bs.colmax,
                     bs.colmax ? 2 * 3 * B::H : 0);
  constexpr size_t lds = f10bh_gru_lds_bytes<S>();
  static_assert(lds <= 64 * 1024, "raise the dynamic LDS limit for this shape");
  hipLaunchKernelGGL((k_gru_bwd_f10h<S, TS>), dim3(rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T, (const TS*)out,
                     (const TS*)h0, hdr, wfrag, reserve, (const TS*)d_out, (const TS*)d_hT, dg_in, dg_hid, (TS*)d_h0, bs);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

template <class S>
int launch_t(const RnnShape& rs, const void* c0, const float* packed_hid, const float* reserve, const void* d_out,
             const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid, void* d_h0, void* d_c0, void* ws,
             unsigned long long* diag, hipStream_t stream, const BwdStats& bs) {
  using B = F10BH<S>;
  float* hdr = reinterpret_cast<float*>(ws);
  xh8* wfrag = reinterpret_cast<xh8*>(hdr + B::HDR_FLOATS);
  constexpr size_t lds = f10bh_lds_bytes<S>();
  static_assert(lds <= 150 * 1024, "LDS image set too large");
  const bool dg = opt(OPT_DIAG) != 0;
  // (round 6) rank 8 at H = 256, one sample per CU: T01 and T2 wave-local, three barriers per step (option dev2 bit 10: the
  // four-barrier kernel, A/B)
  constexpr bool WL_OK = f10bh_ok<S>() && B::XF == 2 && F10<S>::R2 == 8 && B::NM2 * B::CT2 == FAST_NW && F10<S>::H == 256;
  if constexpr (WL_OK) {
    const bool other = (f10bl_ok<S>() && (rs.B > device_cu_count() || (opt(OPT_DEV) & 128)) && !(opt(OPT_DEV) & 32768)) ||
                       (opt(OPT_DEV) & (1 << 17)) || (opt(OPT_DEV2) & 1024);
    if (!other) {
      hipLaunchKernelGGL((k_f10bh_prep<S, false, true>), dim3(B::FT + 1), dim3(FAST_NT), 0, stream, packed_hid, hdr, wfrag, bs.colmax,
                         bs.colmax ? 2 * 4 * B::H : 0);
      auto kern = dg ? k_lstm_bwd_f10h<S, true, true> : k_lstm_bwd_f10h<S, false, true>;
      if (lds > 64 * 1024 && ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
      hipLaunchKernelGGL(kern, dim3(rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T, (const float*)c0, hdr, wfrag, reserve,
                         (const float*)d_out, (const float*)d_hT, (const float*)d_cT, dg_in, dg_hid, (float*)d_h0,
                         (float*)d_c0, dg ? diag : nullptr, bs);
      return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL((k_f10bh_prep<S>), dim3(B::FT + 1), dim3(FAST_NT), 0, stream, packed_hid, hdr, wfrag, bs.colmax,
                     bs.colmax ? 2 * 4 * B::H : 0);
  if constexpr (f10bl_ok<S>()) {
    // one wave per 64 hidden units, T2 wave-local, two barriers per step: where two of its four-wave workgroups share a CU
    // (B > #CUs; cfg4: 586 -> 526 us per layer) and at H = 512 (eight waves either way).  One sample per CU at H = 256 stays on
    // the eight-wave kernel, whose phases are shorter than the four barriers cost (cfg2: 2 851 against 3 301 cycles per step).
    // Option dev bit 15: the eight-wave kernel everywhere; bit 7: this kernel everywhere (A/B, the stamps of lesson 51)
    const bool local = F10BL<S>::NWV != 4 || rs.B > device_cu_count() || (opt(OPT_DEV) & 128);
    if (local && !(opt(OPT_DEV) & 32768)) {
      constexpr size_t ldsl = f10bl_lds_bytes<S>();
      static_assert(ldsl <= 64 * 1024, "raise the dynamic LDS limit for this shape");
      auto kl = dg ? k_lstm_bwd_f10l<S, true> : k_lstm_bwd_f10l<S, false>;
      hipLaunchKernelGGL(kl, dim3(rs.B), dim3(F10BL<S>::NT), ldsl, stream, rs.B, rs.T, (const float*)c0, hdr, wfrag, reserve,
                         (const float*)d_out, (const float*)d_hT, (const float*)d_cT, dg_in, dg_hid, (float*)d_h0, (float*)d_c0,
                         dg ? diag : nullptr, bs);
      return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
    }
  }
  if constexpr (F10<S>::H == 256 && F10BH<S>::XF < 4) {
    // the record-dependent half of the gate phase on the helper waves, one step ahead (k_lstm_bwd_f10p, round 5): MEASURED SLOWER
    // (cfg2 training step 1.689 against 1.642 ms; profiles/r5/stamps_cfg2_f10p.txt: the gate waves' phase stays at 670 cycles with
    // 45 instead of 75 instructions in it — it is made of the LDS round trips on either side of the barrier and the wave-maximum
    // chain, not of the gate arithmetic — while the helper waves' split phase grows from 430 to 610).  Option dev bit 17 selects it.
    if (opt(OPT_DEV) & (1 << 17)) {
      constexpr size_t ldsp = f10bp_lds_bytes<S>();
      static_assert(ldsp <= 150 * 1024, "LDS image set too large");
      auto kp = dg ? k_lstm_bwd_f10p<S, true> : k_lstm_bwd_f10p<S, false>;
      if (ldsp > 64 * 1024 && ensure_dynamic_lds(reinterpret_cast<const void*>(kp), ldsp) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
      hipLaunchKernelGGL(kp, dim3(rs.B), dim3(FAST_NT), ldsp, stream, rs.B, rs.T, (const float*)c0, hdr, wfrag, reserve,
                         (const float*)d_out, (const float*)d_hT, (const float*)d_cT, dg_in, dg_hid, (float*)d_h0,
                         (float*)d_c0, dg ? diag : nullptr, bs);
      return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
    }
  }
  if constexpr (f10bh_ok<S>()) {
    auto kern = dg ? k_lstm_bwd_f10h<S, true> : k_lstm_bwd_f10h<S, false>;
    if (lds > 64 * 1024 && ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T, (const float*)c0, hdr, wfrag, reserve,
                       (const float*)d_out, (const float*)d_hT, (const float*)d_cT, dg_in, dg_hid, (float*)d_h0,
                       (float*)d_c0, dg ? diag : nullptr, bs);
    return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  } else {
    return TTRNN_ERR_UNSUPPORTED;      // (H = 384: the wave-local kernel only; dev bit 15 has no eight-wave kernel to select)
  }
}

}  // namespace

bool f10bh_available(const RnnShape& rs, int dtype) {
  if (opt(OPT_GEMM_PIECES) == 3) return false;
  if (rs.cell == TTRNN_GRU)      // (bf16 storage: in every math mode, as the three-piece kernel; fp32 storage: split mode)
    return (dtype == TTRNN_BF16 || (dtype == TTRNN_F32 && opt(OPT_FP32_MATH) == TTRNN_MATH_SPLIT)) &&
           (shape_matches<ShpH256R8G>(rs.hid_s) || (shape_matches<ShpH256R16G>(rs.hid_s) && !(opt(OPT_DEV2) & 128)));
  if (dtype != TTRNN_F32 || rs.cell != TTRNN_LSTM || opt(OPT_FP32_MATH) != TTRNN_MATH_SPLIT) return false;
  return shape_matches<ShpH256R8L>(rs.hid_s) || shape_matches<ShpH256R16L>(rs.hid_s);
}

// the reference's default benchmark shape (H = 512, r = 8): a runtime-tier shape whose reverse-time recurrence runs here in split
// mode — eight gate waves, four feature tiles of T01 (K = 128) and two T2 pairs per wave (dev bit 16384 keeps the tier's kernel)
// (round 5: H = 384, r = 8 — benchmarking.py --hidden_size 384 — on the wave-local kernel as six waves)
// (round 6: the TT-GRU of the same benchmark — `--gru`: (8, 8, 8) x (8, 12, 16), K1 = 96 — on k_gru_bwd_f10h as eight gate waves;
// option dev2 bit 8 keeps the tier's kernel)
bool f10bh_h512_available(const RnnShape& rs, int dtype) {
  if (dtype != TTRNN_F32 || opt(OPT_FP32_MATH) != TTRNN_MATH_SPLIT || opt(OPT_GEMM_PIECES) == 3 || (opt(OPT_DEV) & 16384) || rs.T <= 0)
    return false;
  if (rs.cell == TTRNN_GRU) return rs.hid_blocks <= 1 && shape_matches<ShpH512R8G>(rs.hid_s) && !(opt(OPT_DEV2) & 256);
  return rs.cell == TTRNN_LSTM && (shape_matches<ShpH512R8L>(rs.hid_s) || shape_matches<ShpH384R8L>(rs.hid_s));
}
size_t f10bh_h512_workspace_bytes() {
  constexpr size_t a = F10BH<ShpH512R8L>::HDR_FLOATS * sizeof(float) + F10BH<ShpH512R8L>::FRAGS * sizeof(xh8) + 4096;
  constexpr size_t b = F10BH<ShpH384R8L>::HDR_FLOATS * sizeof(float) + F10BH<ShpH384R8L>::FRAGS * sizeof(xh8) + 4096;
  constexpr size_t g = F10BH<ShpH512R8G>::HDR_FLOATS * sizeof(float) + F10BH<ShpH512R8G>::FRAGS * sizeof(xh8) + 4096;
  return (a > b ? a : b) > g ? (a > b ? a : b) : g;
}
int launch_rnn_bwd_f10_h512(const RnnShape& rs, const void* out, const void* h0, const void* c0, const float* packed_hid,
                            const float* reserve, const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid,
                            void* d_h0, void* d_c0, void* ws, hipStream_t stream, float* stats) {
  BwdStats bs;
  if (stats) bs.colmax = reinterpret_cast<unsigned*>(stats);      // (cleared by the prep launch)
  if (rs.cell == TTRNN_GRU) {
    int st = launch_gru_t<ShpH512R8G, float>(rs, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws, stream, bs);
    if (st == TTRNN_OK && stats) st = launch_bwd_stats_finish(rs.cell, rs.B, rs.G * rs.H, nullptr, stats, stream);
    return st;
  }
  unsigned long long* diag = reinterpret_cast<unsigned long long*>((char*)ws + f10bh_h512_workspace_bytes() - 4096);
  int st = shape_matches<ShpH384R8L>(rs.hid_s)
               ? launch_t<ShpH384R8L>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, dg_hid, d_h0, d_c0, ws, diag, stream, bs)
               : launch_t<ShpH512R8L>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, dg_hid, d_h0, d_c0, ws, diag, stream, bs);
  if (st == TTRNN_OK && stats) st = launch_bwd_stats_finish(rs.cell, rs.B, rs.G * rs.H, nullptr, stats, stream);
  return st;
}

size_t f10bh_workspace_bytes(const RnnShape& rs) {
  if (shape_matches<ShpH256R8G>(rs.hid_s))
    return F10BH<ShpH256R8G>::HDR_FLOATS * sizeof(float) + F10BH<ShpH256R8G>::FRAGS * sizeof(xh8);
  if (shape_matches<ShpH256R16G>(rs.hid_s))
    return F10BH<ShpH256R16G>::HDR_FLOATS * sizeof(float) + F10BH<ShpH256R16G>::FRAGS * sizeof(xh8);
  if (shape_matches<ShpH256R8L>(rs.hid_s))
    return F10BH<ShpH256R8L>::HDR_FLOATS * sizeof(float) + F10BH<ShpH256R8L>::FRAGS * sizeof(xh8);
  if (shape_matches<ShpH256R16L>(rs.hid_s))
    return F10BH<ShpH256R16L>::HDR_FLOATS * sizeof(float) + F10BH<ShpH256R16L>::FRAGS * sizeof(xh8);
  return 0;
}

int launch_lstm_bwd_f10h(const RnnShape& rs, const void* c0, const float* packed_hid, const float* reserve,
                         const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid, void* d_h0,
                         void* d_c0, void* ws, unsigned long long* diag, hipStream_t stream, const BwdStats& bs) {
  if (shape_matches<ShpH256R8L>(rs.hid_s))
    return launch_t<ShpH256R8L>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, dg_hid, d_h0, d_c0, ws, diag, stream,
                                bs);
  if (shape_matches<ShpH256R16L>(rs.hid_s))
    return launch_t<ShpH256R16L>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, dg_hid, d_h0, d_c0, ws, diag, stream,
                                 bs);
  return TTRNN_ERR_UNSUPPORTED;
}

// naive per-gate sets of H = 256, d = 3, r = 8 (fp32 storage, split mode; option dev2 bit 9 keeps the tier's kernel)
bool f10n_bwd_available(const RnnShape& rs, int dtype) {
  return dtype == TTRNN_F32 && opt(OPT_FP32_MATH) == TTRNN_MATH_SPLIT && opt(OPT_GEMM_PIECES) != 3 && !opt(OPT_NO_F10) &&
         !(opt(OPT_DEV2) & 512) && rs.B >= 1 && rs.T >= 1 && f10n_shape_matches(rs);
}
size_t f10n_bwd_workspace_bytes(const RnnShape& rs) {
  using B = F10BH<ShpH256N>;
  const int G = rs.cell == TTRNN_LSTM ? 4 : 3;
  const size_t npg = (f10n_unjoined_floats(rs.cell) + 63) & ~(size_t)63;
  return npg * sizeof(float) + (size_t)G * (B::HDR_FLOATS * sizeof(float) + B::FRAGS * sizeof(xh8)) + 256;
}
int launch_rnn_bwd_f10n(const RnnShape& rs, const void* out, const void* h0, const void* c0, const float* packed_hid,
                        const float* reserve, const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid,
                        void* d_h0, void* d_c0, void* ws, hipStream_t stream, float* stats) {
  BwdStats bs;
  if (stats) bs.colmax = reinterpret_cast<unsigned*>(stats);      // (cleared by the first prep launch)
  int st = rs.cell == TTRNN_LSTM
               ? launch_n_bwd<ShpH256N, TTRNN_LSTM>(rs, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, dg_hid, d_h0, d_c0, ws, stream, bs)
               : launch_n_bwd<ShpH256N, TTRNN_GRU>(rs, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, dg_hid, d_h0, d_c0, ws, stream, bs);
  if (st == TTRNN_OK && stats) st = launch_bwd_stats_finish(rs.cell, rs.B, rs.G * rs.H, nullptr, stats, stream);
  return st;
}

int launch_gru_bwd_f10h(const RnnShape& rs, int dtype, const void* out, const void* h0, const float* packed_hid,
                        const float* reserve, const void* d_out, const void* d_hT, float* dg_in, float* dg_hid, void* d_h0,
                        void* ws, hipStream_t stream, const BwdStats& bs) {
  if (shape_matches<ShpH256R16G>(rs.hid_s)) {      // (round 6: `benchmarking.py --hidden_size 256 --gru --ttrank 16`)
    if (dtype == TTRNN_F32)
      return launch_gru_t<ShpH256R16G, float>(rs, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws, stream, bs);
    return launch_gru_t<ShpH256R16G, bf16_t>(rs, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws, stream, bs);
  }
  if (!shape_matches<ShpH256R8G>(rs.hid_s)) return TTRNN_ERR_UNSUPPORTED;
  if (dtype == TTRNN_F32)
    return launch_gru_t<ShpH256R8G, float>(rs, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws, stream, bs);
  return launch_gru_t<ShpH256R8G, bf16_t>(rs, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws, stream, bs);
}

}  // namespace ttrnn
