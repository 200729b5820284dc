// ttrnn_f10_dev.h — device pieces of the fused-core forward step shared by ttrnn_fast_f10.hip (one sample per workgroup)
// and ttrnn_fast_f10nb.hip (two samples per workgroup).  Device-only (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include "ttrnn_core.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_f10.h"

namespace ttrnn {

template <class S, int KS>
constexpr size_t f10_lds_bytes() {
  // fp32 h (two parities, for the output store) + bf16 h planes (two parities) + the three planes of the S10 operand
  // + (KS == 2) the partial accumulators handed from the second k-half's waves to the gate waves
  return 2 * sizeof(float) * F10<S>::H + 2 * 3 * 2 * (size_t)F10<S>::H + 2 * 3 * (size_t)F10<S>::PLANE +
         (KS == 2 ? F10<S>::MT * 64 * sizeof(f32x4) : 0);
}

// term-packed fragments of core 2 for m-tile mt: lane (r, q) holds feature 16mt + r, k-group q (8 values of j2)
template <class S>
__device__ __forceinline__ void f10_load_w2(xbf8& a1, xbf8& a2, const float* packed, int mt, int lane) {
  using F = F10<S>;
  const int r = lane & 15, q = lane >> 4;
  const float* W2 = packed + woff_of<S>(2);               // [J2][M2]
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    __bf16 p0, p1, p2;
    split3(e < F::J2 ? W2[e * F::M2 + 16 * mt + r] : 0.f, p0, p1, p2);
    a1[e] = (q & 1) ? p1 : p0;                            // groups w0 | w1 | w0 | w1
    a2[e] = q == 0 ? p2 : (q == 1 ? p0 : (__bf16)0.f);    // groups w2 | w0 | 0 | 0
  }
}

// one S2 tile (m-tile mt, chain-row tile rt): hp = the three bf16 planes [3][XPL] of the input; the MFMA part ...
template <class S>
__device__ __forceinline__ f32x4 f10_s2_mma(const xbf8& a1, const xbf8& a2, const __bf16* hp, int rt, int lane) {
  using F = F10<S>;
  const int c = lane & 15, q = lane >> 4;
  const int row = 16 * rt + c;
  const int pl1 = q >> 1;                                 // groups x0 | x0 | x1 | x1
  const int pl2 = q == 1 ? 2 : 0;                         // groups x0 | x2 | (x0 against zero core groups)
  const xbf8 b1 = *reinterpret_cast<const xbf8*>(hp + pl1 * F::XPL + row * 8);
  const xbf8 b2 = *reinterpret_cast<const xbf8*>(hp + pl2 * F::XPL + row * 8);
  f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc, 0, 0, 0);
}
// ... and the split of its result into the S10 operand
template <class S>
__device__ __forceinline__ void f10_s2_store(f32x4 acc, __bf16* img, int mt, int rt, int lane) {
  using F = F10<S>;
  const int c = lane & 15, q = lane >> 4;
  const int row = 16 * rt + c;
  const int m0 = 16 * mt + 4 * q;
  const int i = m0 / F::R2, a0 = m0 % F::R2;
  // C2[i][row][a0..a0+3] (ops.py:89-90: C2 flat == the [I2][K10] operand of the next stage), k order: F10::kperm
  if (row < F::ROWS2) store_split4(img, F::PLANE, x_off<F::K>(i, F::kperm(row, a0)), acc);   // padding rows: no store
}
// S10 k-blocks [u0, u0 + NU) (w10 holds exactly those): reads run PD blocks ahead of the MFMAs (at most two waves per
// SIMD do this: little else hides the LDS latency; sched_barrier keeps the compiler from sinking the reads back next
// to their use)
template <class S, int NU>
__device__ __forceinline__ void f10_s10_part(const xbf8 (&w10)[3][NU], const __bf16* img, int row, int q, int u0,
                                             f32x4& acc_lo, f32x4& acc_hi) {
  using F = F10<S>;
  constexpr int PD = NU < 3 ? NU : 3;
  xbf8 af[NU][3];
#pragma unroll
  for (int u = 0; u < PD; ++u) {
    const int off = x_off<F::K>(row, 32 * (u0 + u) + 8 * q);
#pragma unroll
    for (int p = 0; p < 3; ++p) af[u][p] = *reinterpret_cast<const xbf8*>(img + p * F::PLANE + off);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (u + PD < NU) {
      const int off = x_off<F::K>(row, 32 * (u0 + u + PD) + 8 * q);
#pragma unroll
      for (int p = 0; p < 3; ++p) af[u + PD][p] = *reinterpret_cast<const xbf8*>(img + p * F::PLANE + off);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 5; ++s)      // the five low-order terms, then the leading one into its own accumulator
      acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10[SPLIT_TW[s]][u], af[u][SPLIT_TX[s]], acc_lo, 0, 0, 0);
    acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10[0][u], af[u][0], acc_hi, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- the same step on two-piece fp16 operands (ttrnn_split.h): LSTM forward kernels ---------------------------------
// Scales: powers of two (exact), TWO-SIDED DIAGONAL — one exponent per output mode index i2 and per rank index r2 of core 2,
// one per row of the fused core — so that a single large entry of a core moves only the scale of its own row / column and
// every other entry keeps its 22 bits (round 2 had one scale per operand: one entry x 1e5 cost all the others 5-12 bits):
//   core 2 (S2's constant operand)   G2'[r2,i2,j2] = G2 2^(6 + eu[i2] + ev[r2])      max over each i2 and each r2 slice < 2^6
//   h (S2's dynamic operand)         h' = 2^6 h                                        (|h_t| < 1; a caller's h_0: f10h_h0_expo)
//   S2's result = the S10 operand    X'[(j0,j1,r2)][i2] < J2 2^12 = 2^15: inside fp16
//   the fused core                   W10'[m][(j0,j1,r2)] = W10 2^(ep[m] - ev[r2]) x gate factor   max over each row < 2^13.6
//   accumulators of S10              acc[m][i2] = 2^(ep[m] + eu[i2] + 12) x gate factor x pre-activation
// eu, ev, ep are computed once per launch by k_f10h_scale from the cores themselves (eu from the maxima of core 2 over
// (r2, j2), ev from the maxima of 2^eu G2 over (i2, j2), ep from the row maxima of 2^-ev W10) and sit as int32 at the start of
// the fragment workspace: [eu: 16][ev: 16][ep: 64].  Second pieces stay normal over >= 9 binades below each row's / slice's
// maximum; smaller entries keep an ABSOLUTE error of 2^-31 of THAT maximum or better.
static constexpr int F10H_HDR_BYTES = 1024;      // eu[16] | ev[16] | ep[M <= 224] as int32
static constexpr int F10H_PARTS = 64;          // (historic) workgroups of k_f10h_scale for M = 64; launched with F10<S>::M, one row of the fused core each
static constexpr int F10H_EU = 0, F10H_EV = 16, F10H_EP = 32;
static constexpr float F10H_HSC = 64.0f;       // 2^6: the scale of h
struct F10hScales { f32x4 pre, un; };          // per lane: 2^(ep[m_j] + eu[c] + 12) for the lane's four rows m_j, and the inverse
__device__ __forceinline__ int f10h_expo(float x) {
  // x < 2^e (frexp: x = f 2^e, f in [0.5, 1)); zero / non-finite maxima fall back to a neutral exponent
  if (!(x > 0.f)) return 0;
  if (!(x < 3e38f)) return 40;
  int e;
  frexpf(x, &e);
  return e < -40 ? -40 : (e > 40 ? 40 : e);
}
// lane (c, q) of S10 feature tile `tile`: accumulator register j is row m_j = MPG j + 4 tile + q (gate j), column i2 = c
template <class S>
__device__ __forceinline__ F10hScales f10h_scales(const float* __restrict__ hdr, int tile, int lane) {
  using F = F10<S>;
  const int* e = reinterpret_cast<const int*>(hdr);
  const int c = lane & 15, q = lane >> 4;
  const int eu = e[F10H_EU + (c < F::I2 ? c : F::I2 - 1)];
  F10hScales r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int s = e[F10H_EP + F::MPG * j + 4 * tile + q] + eu + 12;
    r.pre[j] = ldexpf(1.f, s);
    r.un[j] = ldexpf(1.f, -s);
  }
  return r;
}
// scale of row m2 = i2 R2 + r2 of core 2 / of entry (m, r2) of the fused core
template <class S>
__device__ __forceinline__ float f10h_g2_scale(const float* __restrict__ hdr, int m2) {
  const int* e = reinterpret_cast<const int*>(hdr);
  return ldexpf(1.f, 6 + e[F10H_EU + m2 / F10<S>::R2] + e[F10H_EV + m2 % F10<S>::R2]);
}
template <class S>
__device__ __forceinline__ float f10h_w_scale(const float* __restrict__ hdr, int m, int r2) {
  const int* e = reinterpret_cast<const int*>(hdr);
  return ldexpf(1.f, e[F10H_EP + m] - e[F10H_EV + r2]);
}

template <class S, int KS>
constexpr size_t f10h_lds_bytes() {
  // fp32 h (two parities, for the output store) + fp16 h planes (two parities x two planes) + the two planes of the S10
  // operand + (KS == 2) the partial accumulators handed from the second k-half's waves to the gate waves
  return 2 * sizeof(float) * F10<S>::H + 2 * 2 * 2 * (size_t)F10<S>::H + 2 * 2 * (size_t)F10<S>::PLANE +
         (KS == 2 ? F10<S>::MT * 64 * sizeof(f32x4) : 0);
}

// A caller's h_0 may lie outside (-1, 1), the range the scale of h assumes (|h_t| < 1 for every t >= 1): per SAMPLE the
// pieces of h_0 are those of 2^-e0 h_0 (e0 >= 0, max |h_0[b]| < 2^e0) and the first step's accumulators — whose initial
// value is scaled alike — are multiplied back by 2^e0.  Per sample, so that a sample's result never depends on which other
// samples share its batch.  Every thread of the workgroup calls this; `scratch`: NWV floats of LDS nothing else uses yet.
template <int NWV>
__device__ __forceinline__ int f10h_h0_expo(float hmine, float* scratch, int wave, int lane) {
  float mx = fabsf(hmine);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) scratch[wave] = mx;
  lds_barrier();
  mx = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) mx = fmaxf(mx, scratch[w]);
  lds_barrier();
  const int e = f10h_expo(mx);
  return e < 0 ? 0 : e;
}

// term-packed fragment of core 2 for m-tile mt: k-groups w0 | w1 | w0 | w1 against activation groups x0 | x0 | x1 | x1:
// ONE MFMA = x0 w0 + x0 w1 + x1 w0 + x1 w1
template <class S>
__device__ __forceinline__ void f10h_load_w2(xh8& a1, const float* packed, int mt, int lane, const float* __restrict__ hdr) {
  using F = F10<S>;
  const int r = lane & 15, q = lane >> 4;
  const float* W2 = packed + woff_of<S>(2);               // [J2][M2]
  const float g2scale = f10h_g2_scale<S>(hdr, 16 * mt + r);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    _Float16 p0, p1;
    split2h(e < F::J2 ? W2[e * F::M2 + 16 * mt + r] * g2scale : 0.f, p0, p1);
    a1[e] = (q & 1) ? p1 : p0;
  }
}
template <class S>
__device__ __forceinline__ f32x4 f10h_s2_mma(const xh8& a1, const _Float16* hp, int rt, int lane) {
  using F = F10<S>;
  const int c = lane & 15, q = lane >> 4;
  const xh8 b1 = *reinterpret_cast<const xh8*>(hp + (q >> 1) * F::XPL + (16 * rt + c) * 8);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}
template <class S>
__device__ __forceinline__ void f10h_s2_store(f32x4 acc, _Float16* img, int mt, int rt, int lane) {
  using F = F10<S>;
  const int c = lane & 15, q = lane >> 4;
  const int row = 16 * rt + c;
  const int m0 = 16 * mt + 4 * q;
  const int i = m0 / F::R2, a0 = m0 % F::R2;
  if (row < F::ROWS2) store_split4_h(img, F::PLANE, x_off<F::K>(i, F::kperm(row, a0)), acc);   // padding rows: no store
}
// S10 k-blocks [u0, u0 + NU): (w1, t0) and (w0, t1) into acc_lo, the leading term into acc_hi
template <class S, int NU>
__device__ __forceinline__ void f10h_s10_part(const xh8 (&w10)[2][NU], const _Float16* img, int row, int q, int u0,
                                              f32x4& acc_lo, f32x4& acc_hi) {
  using F = F10<S>;
  constexpr int PD = NU < 4 ? NU : 4;
  xh8 af[NU][2];
#pragma unroll
  for (int u = 0; u < PD; ++u) {
    const int off = x_off<F::K>(row, 32 * (u0 + u) + 8 * q);
#pragma unroll
    for (int p = 0; p < 2; ++p) af[u][p] = *reinterpret_cast<const xh8*>(img + p * F::PLANE + off);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (u + PD < NU) {
      const int off = x_off<F::K>(row, 32 * (u0 + u + PD) + 8 * q);
#pragma unroll
      for (int p = 0; p < 2; ++p) af[u + PD][p] = *reinterpret_cast<const xh8*>(img + p * F::PLANE + off);
    }
    __builtin_amdgcn_sched_barrier(0);
    acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w10[1][u], af[u][0], acc_lo, 0, 0, 0);
    acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w10[0][u], af[u][1], acc_lo, 0, 0, 0);
    acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(w10[0][u], af[u][0], acc_hi, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- S2 INSIDE the gate waves, on tile PAIRS (round 5: k_lstm_fwd_f10s, k_gru_fwd_f10vh) --------------------------------------
// Wave w of the fused-core kernels owns hidden units 64w .. 64w+63 (lane l <-> unit 64w + l) = chain rows 8w .. 8w+7 of S2, eight
// consecutive units (j2 = l & 7) per row (l >> 3): S2 of the NEW state needs no other wave's data.  Every lane packs its unit as
// one dword (p0 | p1 << 16, the two fp16 pieces of 2^sH h) and the B operand is gathered through the LDS crossbar (ds_bpermute:
// no memory, no barrier).  A wave has only EIGHT chain rows for the MFMA's sixteen columns, so two m-tiles share one MFMA:
//     columns 0-7  = chain rows 0-7 against m-tile X = p,        live k-slots 0-15  (B is zero in 16-31)
//     columns 8-15 = chain rows 0-7 against m-tile Y = p + NP,   live k-slots 16-31 (B is zero in 0-15)
//     k-slot pairs (2j, 2j+1) of a half = (x0, x1) of j2 = j;   A1 = (w0, w0),  A2 = (w1, w1)  ->  two MFMAs = all four terms
// — every lane of the result is a live (feature, chain row) pair: MT2 MFMAs and MT2 / 2 splitting passes per wave and step where
// one m-tile per MFMA costs MT2 passes with half of the lanes masked.
template <class S>
struct F10P {
  using F = F10<S>;
  static constexpr int NP = F::MT2 / 2;                    // tile pairs
  static_assert(F::MT2 % 2 == 0 && F::J2 == 8, "tile pairs");
};
// lane (r, q) of A: k = 8q .. 8q+7 <-> m-tile (q < 2 ? X : Y), j2 = 4 (q & 1) + e / 2, both slots of a pair the same piece
template <class S>
__device__ __forceinline__ void f10p_load_w2(xh8 (&a1)[F10P<S>::NP], xh8 (&a2)[F10P<S>::NP], const float* packed, int lane,
                                             const float* __restrict__ hdr) {
  using F = F10<S>;
  constexpr int NP = F10P<S>::NP;
  const int r = lane & 15, q = lane >> 4;
  const float* W2 = packed + woff_of<S>(2);               // [J2][M2]
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int m2 = 16 * (q < 2 ? p : p + NP) + r;
    const float sc = f10h_g2_scale<S>(hdr, m2);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 p0, p1;
      split2h(W2[(4 * (q & 1) + e) * F::M2 + m2] * sc, p0, p1);
      a1[p][2 * e] = p0; a1[p][2 * e + 1] = p0;
      a2[p][2 * e] = p1; a2[p][2 * e + 1] = p1;
    }
  }
}
// per-lane loop invariants: the gather address, whether the lane's B slots are live, the store offsets of the NP results
template <class S>
struct F10pLane {
  int gsrc;
  bool live;
  int soff[F10P<S>::NP];
  __device__ __forceinline__ void init(int wave, int lane) {
    using F = F10<S>;
    const int c = lane & 15, q = lane >> 4;
    gsrc = 4 * (8 * (c & 7) + 4 * (q & 1));               // byte address of lane 8 rho + 4 (q & 1) for ds_bpermute
    live = (c < 8) == (q < 2);
#pragma unroll
    for (int p = 0; p < F10P<S>::NP; ++p) {
      const int m0 = 16 * (c < 8 ? p : p + F10P<S>::NP) + 4 * q;     // registers j: features m0 + j, chain row 8 wave + (c & 7)
      soff[p] = x_off<F::K>(m0 / F::R2, F::kperm(8 * wave + (c & 7), m0 % F::R2));
    }
  }
};
// pk: this lane's unit as (p0 | p1 << 16); img: the two fp16 planes the NEXT S10 reads
template <class S>
__device__ __forceinline__ void f10p_s2(const xh8 (&a1)[F10P<S>::NP], const xh8 (&a2)[F10P<S>::NP], const F10pLane<S>& ln,
                                        unsigned pk, _Float16* img) {
  using F = F10<S>;
  constexpr int NP = F10P<S>::NP;
  unsigned d[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) d[e] = (unsigned)__builtin_amdgcn_ds_bpermute(ln.gsrc + 4 * e, (int)pk);
#pragma unroll
  for (int e = 0; e < 4; ++e) d[e] = ln.live ? d[e] : 0u;
  const xh8 bfrag = __builtin_bit_cast(xh8, u32x4{d[0], d[1], d[2], d[3]});
  f32x4 acc[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[p], bfrag, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
  for (int p = 0; p < NP; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[p], bfrag, acc[p], 0, 0, 0);
#pragma unroll
  for (int p = 0; p < NP; ++p) store_split4_h(img, F::PLANE, ln.soff[p], acc[p]);
}
// the same with core 2's fragments in LDS (afr[(2 p + piece) * 64 + lane], written once per launch by every wave for itself or by
// the workgroup): r = 16 shapes, whose twelve m-tiles would take 48 VGPRs next to 128 of fused-core fragments
template <class S>
__device__ __forceinline__ void f10p_s2_lds(const xh8* afr, int lane, const F10pLane<S>& ln, unsigned pk, _Float16* img) {
  using F = F10<S>;
  constexpr int NP = F10P<S>::NP;
  unsigned d[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) d[e] = (unsigned)__builtin_amdgcn_ds_bpermute(ln.gsrc + 4 * e, (int)pk);
#pragma unroll
  for (int e = 0; e < 4; ++e) d[e] = ln.live ? d[e] : 0u;
  const xh8 bfrag = __builtin_bit_cast(xh8, u32x4{d[0], d[1], d[2], d[3]});
  constexpr int GP = NP < 3 ? NP : 3;      // pairs per group: three accumulators and six fragments live at a time
  static_assert(NP % GP == 0, "pair groups");
#pragma unroll
  for (int p0 = 0; p0 < NP; p0 += GP) {
    xh8 fa[GP], fb[GP];
#pragma unroll
    for (int p = 0; p < GP; ++p) {
      fa[p] = afr[(2 * (p0 + p)) * 64 + lane];
      fb[p] = afr[(2 * (p0 + p) + 1) * 64 + lane];
    }
    f32x4 acc[GP];
#pragma unroll
    for (int p = 0; p < GP; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[p], bfrag, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < GP; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[p], bfrag, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < GP; ++p) store_split4_h(img, F::PLANE, ln.soff[p0 + p], acc[p]);
  }
}
__device__ __forceinline__ unsigned f10p_pack(float hscaled) {
  _Float16 p0, p1;
  split2h(hscaled, p0, p1);
  return (unsigned)__builtin_bit_cast(unsigned short, p0) | ((unsigned)__builtin_bit_cast(unsigned short, p1) << 16);
}

// ---- the scale header of the two-piece fp16 kernels (LSTM: ttrnn_fast_f10.hip; fp32 GRU: ttrnn_fast_f10gh.hip) ---------------
// The diagonal power-of-two scales of ttrnn_f10_dev.h, from the cores themselves.  F10H_PARTS = M workgroups; every one derives
// eu / ev from core 2 (one or two thousand entries) and then takes ONE row of the fused core (16 workgroups of four rows each
// took 11 us for r = 16: a row is 512 dot products of 16 terms):
//   eu[i2] = -expo(max_{r2,j2} |G2|),  ev[r2] = -expo(max_{i2,j2} 2^eu |G2|),  ep[m] = 12 - expo(max_k 2^-ev |W10[m][k]|)
template <class S>
__global__ void __launch_bounds__(256) k_f10h_scale(const float* __restrict__ packed, int* __restrict__ hdr) {
  using F = F10<S>;
  static_assert(F::I2 <= 16 && F::R2 <= 16 && F::K % 64 == 0, "one workgroup (blockIdx.x) per row of the fused core: launch F::M of them");
  __shared__ unsigned mx[32];
  __shared__ int eu[16], ev[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* W2 = packed + woff_of<S>(2);               // [J2][M2], m2 = i2 R2 + r2
  if (tid < 32) mx[tid] = 0u;
  __syncthreads();
  for (int i = tid; i < F::J2 * F::M2; i += 256)          // non-negative floats order like their bit patterns
    atomicMax(&mx[(i % F::M2) / F::R2], __float_as_uint(fabsf(W2[i])));
  __syncthreads();
  if (tid < 16) eu[tid] = tid < F::I2 ? -f10h_expo(__uint_as_float(mx[tid])) : 0;
  __syncthreads();
  for (int i = tid; i < F::J2 * F::M2; i += 256) {
    const int m2 = i % F::M2;
    atomicMax(&mx[16 + m2 % F::R2], __float_as_uint(fabsf(W2[i]) * ldexpf(1.f, eu[m2 / F::R2])));
  }
  __syncthreads();
  if (tid < 16) ev[tid] = tid < F::R2 ? -f10h_expo(__uint_as_float(mx[16 + tid])) : 0;
  __syncthreads();
  // row m = blockIdx.x of the fused core (the same fmaf chain as k_f10h_prep): the workgroup's threads over k = (row2, r2)
  __shared__ float red[4];
  const int m = blockIdx.x;
  const int i0 = m / F::I1, i1 = m % F::I1;
  const float* W0 = packed + woff_of<S>(0);               // [J0*R1][I0]
  const float* W1 = packed + woff_of<S>(1);               // [J1*R2][I1*R1]
  float best = 0.f;
  for (int k = tid; k < F::K; k += 256) {
    const int r2 = k % F::R2, row2 = k / F::R2;
    const int j1 = row2 % F::J1, j0 = row2 / F::J1;
    const float* w1p = W1 + (j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1;
    float v = 0.f;
    for (int r1 = 0; r1 < F::R1; ++r1) v = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0], w1p[r1], v);
    best = fmaxf(best, fabsf(v) * ldexpf(1.f, -ev[r2]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
  if (lane == 0) red[wave] = best;
  __syncthreads();
  if (tid == 0) hdr[F10H_EP + m] = 12 - f10h_expo(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
  if (blockIdx.x == 0 && tid < 16) {
    hdr[F10H_EU + tid] = eu[tid];
    hdr[F10H_EV + tid] = ev[tid];
  }
}

}  // namespace ttrnn
