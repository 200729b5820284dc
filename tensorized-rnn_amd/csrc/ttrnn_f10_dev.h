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
// Scales (powers of two, exact) derived from the maxima k_f10h_scale leaves at the start of the fragment workspace:
//   2^a   core 2 (S2's constant operand)      max |G2| 2^a  <= 2^6
//   2^sH  h (S2's dynamic operand)            |h| 2^sH  <= 2^6       (|h_t| < 1; a caller's h_0: f10h_h0_expo, per sample)
//   2^sw  the fused core W10                  max |W10| 2^sw <= 2^12 (bound R1 max|G0| max|G1|)
//   2^S, 2^-S with S = a + sH + sw: accumulators of S10 are 2^S times the pre-activations
// so that |T| 2^(a+sH) <= J2 2^12 = 2^15 stays inside fp16 and the second pieces stay normal over >= 9 binades below each
// operand's maximum (smaller entries keep an ABSOLUTE error of 2^-31 of the maximum or better).
static constexpr int F10H_HDR_BYTES = 256;
// The header holds F10H_PARTS partial maxima per quantity ([part][4]: |core 0|, |core 1|, |core 2|, |h_0|), one per
// workgroup of k_f10h_scale (the h_0 column is always 0 now: f10h_h0_expo); every consumer reduces them itself — 16 loads — instead of
// waiting for one more dependent launch to do it.
static constexpr int F10H_PARTS = 16;
struct F10hScales { float g2, h, w, pre, un; };
__device__ __forceinline__ int f10h_expo(float x) {
  // x < 2^e (frexp: x = f 2^e, f in [0.5, 1)); zero / non-finite maxima fall back to a neutral exponent
  if (!(x > 0.f)) return 0;
  if (!(x < 3e38f)) return 40;
  int e;
  frexpf(x, &e);
  return e < -40 ? -40 : (e > 40 ? 40 : e);
}
template <class S>
__device__ __forceinline__ F10hScales f10h_scales(const float* __restrict__ hdr) {
  float m[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < F10H_PARTS; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(hdr + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
  }
  const int eg = f10h_expo(m[2]);
  int eh = f10h_expo(m[3]);
  if (eh < 0) eh = 0;                                              // |h_t| < 1 for every t >= 1
  const int ew = f10h_expo((float)F10<S>::R1 * m[0] * m[1]);
  const int a = 6 - eg, sh = 6 - eh, sw = 12 - ew;
  F10hScales r;
  r.g2 = ldexpf(1.f, a);
  r.h = ldexpf(1.f, sh);
  r.w = ldexpf(1.f, sw);
  r.pre = ldexpf(1.f, a + sh + sw);
  r.un = ldexpf(1.f, -(a + sh + sw));
  return r;
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
__device__ __forceinline__ void f10h_load_w2(xh8& a1, const float* packed, int mt, int lane, float g2scale) {
  using F = F10<S>;
  const int r = lane & 15, q = lane >> 4;
  const float* W2 = packed + woff_of<S>(2);               // [J2][M2]
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

}  // namespace ttrnn
