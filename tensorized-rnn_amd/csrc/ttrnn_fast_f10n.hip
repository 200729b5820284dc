// ttrnn_fast_f10n.hip — the fused-core forward kernel for the NAIVE per-gate TT-LSTM / TT-GRU (gfx950).
//
// `is_naive=True` (tensorized_rnn/tt_linearset.py:5-38, tt_lstm.py:17-21; pmnist_test.py --naive_tt) gives every gate its own
// TTLinear; the host presents the set as ONE block-diagonal TT-matrix with a gate-selector core (ttrnn_rnn_desc::hid_blocks = 4).
// Until round 5 these matrices ran on the runtime-shape tier (ttrnn_g2.hip) at every size: resident at H = 256, but three barriers
// and 4 850 cycles per step against the concatenated matrix's 1 460.  Per gate the matrix is a d = 3 TT-matrix of its own,
// (4, 8, 8) x (4, 8, 8), r = 8 — small enough for the fused-core scheme with ONE GATE PER WAVE:
//     wave g holds gate g's fused core W10_g = G1_g G0_g (32 rows = two 16-feature tiles x 8 k-blocks, two fp16 pieces: 128 VGPRs),
//     S10  (2 x 24 MFMAs) of gate g against gate g's operand image, raw accumulators into the fp32 gate vector  | barrier |
//     gates (lstm.py:26-32), one hidden unit per thread, then S2 of the NEW state inside the wave — its eight chain rows against
//     the core-2 fragments of ALL FOUR gates (tile pairs, ttrnn_f10_dev.h: f10p_*; fragments in LDS) into the four images | barrier.
// The structure of k_gru_fwd_f10vh (ttrnn_fast_f10gh.hip), whose gates also meet through LDS; the arithmetic of every fp32 kernel of
// the family: two fp16 pieces per operand under the diagonal power-of-two scales of k_f10h_scale, computed per gate.
// Input side: the runtime tier's K-in, in its conventions (gin slots i, g, f, o with both biases folded in; input_size == 1: the
// unit row's projection + the bias row, scaled by x_t here).  The reverse recurrence stays on the tier's kernel (same reserve).
// The naive TT-GRU (gru.py:150-153: three gates, per-gate biases) runs the same kernel with three gate waves — the fourth does
// gates and S2 only — and the GRU's per-step tracking of a large state's exponent (k_gru_fwd_f10vh).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_f10.h"
#include "ttrnn_f10_dev.h"

namespace ttrnn {

namespace {

constexpr int f10n_gates(int cell) { return cell == TTRNN_LSTM ? 4 : 3; }

template <class S>
constexpr bool f10n_ok() {
  using F = F10<S>;
  return S::D == 3 && S::R[0] == 1 && S::R[3] == 1 && shape_ok_recurrent<S>() && F::K % 64 == 0 && F::M == 32 && F::I2 == 8 &&
         out_size_of<S>() == F::H && F::J2 == 8 && F::ROWS2 == 32 && F::M2 % 32 == 0 && F::R2 % 4 == 0 && F::R2 <= 16 && F::H == 256;
}
template <class S>
constexpr int f10n_packed_elems() { return woff_of<S>(S::D); }

// Gate g's three cores out of the joint packed cores (t3nsor/ops.py:47-51 layout, include/ttrnn.h: W_k[(j R_{k+1} + b) M_k + i R_k + a]):
// joint core k + 1 holds gate g's core k in rank block g (left rank index g for the first core: the selector's output).  The same
// launch CHECKS the promise: a non-zero joint entry outside the blocks, or off the selector's diagonal, is counted in
// TTRNN_STAT_BLOCK_VIOLATIONS (what k_g2_merge does for the tier).
template <class S, int G>
__global__ void __launch_bounds__(256) k_f10n_unjoin(TtShape js, const float* __restrict__ pj, float* __restrict__ pg, unsigned* status) {
  constexpr int TOT = f10n_packed_elems<S>();
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (long)G * TOT) {
    const int g = (int)(t / TOT), e = (int)(t - (long)g * TOT);
    const int k = e < woff_of<S>(1) ? 0 : (e < woff_of<S>(2) ? 1 : 2);
    const int Rp = S::R[k], Rn = S::R[k + 1], M = S::I[k] * Rp;
    const int local = e - woff_of<S>(k);
    const int row = local / M, col = local - row * M;
    const int j = row / Rn, b = row - j * Rn, i = col / Rp, a = col - i * Rp;
    const int kk = k + 1;
    const int aJ = k == 0 ? g : g * Rp + a, bJ = k == S::D - 1 ? 0 : g * Rn + b;
    pg[t] = pj[js.woff[kk] + (long)(j * js.R[kk + 1] + bJ) * js.M[kk] + i * js.R[kk] + aJ];
  }
  if (status && t < js.wtotal) {
    int k = 0;
    while (k + 1 < js.d && t >= js.woff[k + 1]) ++k;
    const int local = (int)t - js.woff[k];
    const int row = local / js.M[k], col = local - row * js.M[k];
    const int b = row % js.R[k + 1], i = col / js.R[k], a = col - i * js.R[k];
    bool inside;
    if (k == 0) inside = i == b;                                                      // selector (1, G, 1, G): the identity
    else if (k == 1) inside = a == b / (js.R[2] / G);
    else if (k == js.d - 1) inside = true;
    else inside = a / (js.R[k] / G) == b / (js.R[k + 1] / G);
    if (!inside && pj[t] != 0.f) atomicAdd(status + TTRNN_STAT_BLOCK_VIOLATIONS, 1u);
  }
}

// The four fused cores in fragment order, natural feature order, rows under their gate's 2^(ep[m] - ev[r2]), two fp16 pieces:
//   wfrag[g][((t * NM + u) * 2 + plane) * 64 + lane]      (k_f10gh_prep for every gate)
template <class S>
__global__ void __launch_bounds__(64) k_f10n_prep(const float* __restrict__ pg, const float* __restrict__ hdrs, xh8* __restrict__ wfrag) {
  using F = F10<S>;
  constexpr int TOT = f10n_packed_elems<S>(), NT = F::MT * F::NM;
  const int lane = threadIdx.x, g = blockIdx.x / NT, rest = blockIdx.x % NT;
  const int u = rest % F::NM, t = rest / F::NM;
  const float* packed = pg + (size_t)g * TOT;
  const float* hdr = hdrs + g * (F10H_HDR_BYTES / 4);
  const int r = lane & 15, q = lane >> 4;
  const int m = 16 * t + r;
  const int i0 = m / F::I1, i1 = m % F::I1;
  const float* W0 = packed + woff_of<S>(0);               // [J0*R1][I0]
  const float* W1 = packed + woff_of<S>(1);               // [J1*R2][I1*R1]
  xh8 f0, f1;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int slot = 4 * u + q;                           // k = 8*slot + e in F10::kperm order
    const int r2 = (slot / F::HR) * 4 + (e & 3), row2 = 2 * (slot % F::HR) + (e >> 2);
    const int j1 = row2 % F::J1, j0 = row2 / F::J1;
    const float* w1p = W1 + (j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1;
    float v = 0.f;
    for (int r1 = 0; r1 < F::R1; ++r1) v = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0], w1p[r1], v);
    _Float16 p0, p1;
    split2h(v * f10h_w_scale<S>(hdr, m, r2), p0, p1);
    f0[e] = p0; f1[e] = p1;
  }
  xh8* dst = wfrag + (size_t)g * NT * 2 * 64 + (size_t)((t * F::NM + u) * 2) * 64 + lane;
  dst[0] = f0; dst[64] = f1;
}

// workspace: [G headers][G fragment sets][G per-gate packed cores]
template <class S>
constexpr size_t f10n_frag_elems() { return (size_t)F10<S>::MT * F10<S>::NM * 2 * 64; }       // xh8 per gate
template <class S>
constexpr size_t f10n_ws_bytes(int G) {
  return (size_t)G * (F10H_HDR_BYTES + f10n_frag_elems<S>() * sizeof(xh8) + (size_t)f10n_packed_elems<S>() * sizeof(float));
}

// H0: the caller passed an initial state; OUT = false: only the final state is consumed; IN1: input_size == 1; DIAG: stamps
template <class S, int CELL, bool H0, bool OUT, bool IN1, bool DIAG = false>
__global__ void __launch_bounds__(256, 2) k_rnn_fwd_f10n(int B, int T, GinSrc gs, const float* __restrict__ bilv,
                                                          const float* __restrict__ h0, const float* __restrict__ c0,
                                                          const float* __restrict__ pg, const float* __restrict__ hdrs,
                                                          const xh8* __restrict__ wfrag, float* __restrict__ out,
                                                          float* __restrict__ hT, float* __restrict__ cT, float* __restrict__ reserve) {
  static_assert(f10n_ok<S>(), "shape not supported by the per-gate fused-core LSTM kernel");
  using F = F10<S>;
  constexpr bool LSTM = CELL == TTRNN_LSTM;
  constexpr int H = F::H, G = f10n_gates(CELL), NP = F10P<S>::NP;
  __shared__ __attribute__((aligned(16))) _Float16 img[G][2 * F::PLANE];        // S10 operands, two fp16 planes [I2][K10] per gate
  __shared__ __attribute__((aligned(16))) float gbuf[G * H + 128];              // SCALED gate sums (+ a dump for the padding columns)
  __shared__ __attribute__((aligned(16))) xh8 afr[G][2 * NP * 64];              // core-2 fragments of the gates (tile pairs)
  __shared__ float hmax[4];                                                      // GRU, H0: the waves' maxima of |h_t|

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                     // = the gate this wave multiplies
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;
  const bool gw = wave < G;                                                      // (a GRU's fourth wave multiplies no gate: gates + S2 only)
  const int wg = gw ? wave : 0;
  const float* hdr_g = hdrs + wg * (F10H_HDR_BYTES / 4);

  if (gw) {
    xh8 t1[NP], t2[NP];
    f10p_load_w2<S>(t1, t2, pg + (size_t)wg * f10n_packed_elems<S>(), lane, hdr_g);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      afr[wave][(2 * p) * 64 + lane] = t1[p];
      afr[wave][(2 * p + 1) * 64 + lane] = t2[p];
    }
  }
  F10pLane<S> ln;
  ln.init(wave, lane);
  xh8 w10a[2][F::NM], w10b[2][F::NM];                                            // feature tiles 0 and 1 of gate `wave`
  {
    const xh8* wf = wfrag + (size_t)wg * f10n_frag_elems<S>() + lane;
#pragma unroll
    for (int u = 0; u < F::NM; ++u)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        w10a[p][u] = wf[(size_t)(u * 2 + p) * 64];
        w10b[p][u] = wf[(size_t)((F::NM + u) * 2 + p) * 64];
      }
  }
  // the accumulators of S10 carry 2^(ep[m] + eu[i2] + 12) of their gate; the GATE thread multiplies its four sums back
  const int hid = tid;
  f32x4 usc = f32x4{0.f, 0.f, 0.f, 0.f};
  {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int* e = reinterpret_cast<const int*>(hdrs + g * (F10H_HDR_BYTES / 4));
      usc[g] = ldexpf(1.f, -(e[F10H_EP + hid / F::I2] + e[F10H_EU + hid % F::I2] + 12));
    }
  }
  // where lane (c, q) puts accumulator register j of tile t2: unit (16 t2 + 4 q + j) I2 + c of gate `wave` (columns >= I2: the dump)
  const int gdst = c < F::I2 ? wg * H + (4 * q) * F::I2 + c : G * H + lane;
  const int gstep = c < F::I2 ? F::I2 : 0;

  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gs.gin);
  const f32x4* bil4 = reinterpret_cast<const f32x4*>(bilv);
  const float* __restrict__ xs = reinterpret_cast<const float*>(gs.x);
  float hst = (H0 && h0) ? h0[b * H + hid] : 0.f;      // (H0: the caller passed h_0 or c_0)
  float cst = (H0 && c0) ? c0[b * H + hid] : 0.f;
  f32x4 gi = f32x4{0.f, 0.f, 0.f, 0.f}, bb = gi;
  XChunk<float> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (IN1) xq.init(xs, b * T, T, lane);
  if (T > 0) {
    if (IN1) { gi = gin4[hid]; bb = bil4[hid]; }
    else gi = gin4[(b * T) * H + hid];
  }
  // a caller's h_0 outside (-1, 1): the operand is 2^-e h, the sums are multiplied back.  LSTM: the first step only (|h_t| < 1 from
  // then on); GRU: |h_t| <= max(1, |h_{t-1}|) only, so the exponent is re-derived every step WHILE it is positive (k_gru_fwd_f10vh)
  int e_cur = 0;
  if constexpr (H0) e_cur = __builtin_amdgcn_readfirstlane(f10h_h0_expo<4>(hst, gbuf, wave, lane));
  else __syncthreads();                                                          // afr is complete
  f32x4 un_t = usc * ldexpf(1.f, e_cur);
  float hsc = ldexpf(F10H_HSC, -e_cur);

  // S2 of this wave's eight chain rows (the state in its lanes: four crossbar gathers) against every gate's core 2
  auto s2_from_lanes = [&](float hscaled) {
    const unsigned pk = f10p_pack(hscaled);
    unsigned d[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = (unsigned)__builtin_amdgcn_ds_bpermute(ln.gsrc + 4 * e, (int)pk);
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = ln.live ? d[e] : 0u;
    const xh8 bfrag = __builtin_bit_cast(xh8, u32x4{d[0], d[1], d[2], d[3]});
#pragma unroll
    for (int g = 0; g < G; ++g) {
      xh8 fa[NP], fb[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) { fa[p] = afr[g][(2 * p) * 64 + lane]; fb[p] = afr[g][(2 * p + 1) * 64 + lane]; }
      f32x4 acc[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[p], bfrag, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int p = 0; p < NP; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[p], bfrag, acc[p], 0, 0, 0);
#pragma unroll
      for (int p = 0; p < NP; ++p) store_split4_h(img[g], F::PLANE, ln.soff[p], acc[p]);
    }
  };
  s2_from_lanes(hst * hsc);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): keep weight-register waits out of the time loop
  lds_barrier();

  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();
  const int row10 = c < F::I2 ? c : F::I2 - 1;
  for (int t = 0; t < T; ++t) {
    // ---- S10 of this wave's gate: two feature tiles ------------------------------------------------------------------------------
    if (gw) {
      // both tiles against ONE read of the operand rows (f10h_s10_part twice read them twice and ran two dependent chains one after
      // the other: 1 500 stamped cycles for the 48 MFMAs); per accumulator the same products in the same order
      f32x4 lo0 = f32x4{0.f, 0.f, 0.f, 0.f}, hi0 = lo0, lo1 = lo0, hi1 = lo0;
      {
        constexpr int NU = F::NM, PD = 4;
        const _Float16* im = img[wg];
        xh8 af[NU][2];
#pragma unroll
        for (int u = 0; u < PD; ++u) {
          const int off = x_off<F::K>(row10, 32 * u + 8 * q);
#pragma unroll
          for (int p = 0; p < 2; ++p) af[u][p] = *reinterpret_cast<const xh8*>(im + p * F::PLANE + off);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (u + PD < NU) {
            const int off = x_off<F::K>(row10, 32 * (u + PD) + 8 * q);
#pragma unroll
            for (int p = 0; p < 2; ++p) af[u + PD][p] = *reinterpret_cast<const xh8*>(im + p * F::PLANE + off);
          }
          __builtin_amdgcn_sched_barrier(0);
          lo0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w10a[1][u], af[u][0], lo0, 0, 0, 0);
          lo1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w10b[1][u], af[u][0], lo1, 0, 0, 0);
          lo0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w10a[0][u], af[u][1], lo0, 0, 0, 0);
          lo1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w10b[0][u], af[u][1], lo1, 0, 0, 0);
          hi0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w10a[0][u], af[u][0], hi0, 0, 0, 0);
          hi1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w10b[0][u], af[u][0], hi1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      const f32x4 a0 = hi0 + lo0, a1 = hi1 + lo1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gbuf[gdst + j * gstep] = a0[j];
        gbuf[gdst + (16 + j) * gstep] = a1[j];
      }
    }
    TT_STAMP(0)
    lds_barrier();
    TT_STAMP(1)
    // ---- gates + state (lstm.py:26-32), then S2 of the new state from the lanes, for every gate's core 2 ----------------------------
    const size_t bt = b * T + t;
    {
      f32x4 g4 = gi;
      if (IN1) g4 = bb + xq.at(t) * gi;
      const f32x4 un = H0 ? un_t : usc;
      float hy;
      if constexpr (LSTM) {
        const float ig = fsigmoid(fmaf(gbuf[hid], un[0], g4[0]));                  // gin slots i, g, f, o
        const float fg = fsigmoid(fmaf(gbuf[H + hid], un[1], g4[2]));
        const float gg = ftanh(fmaf(gbuf[2 * H + hid], un[2], g4[1]));
        const float og = fsigmoid(fmaf(gbuf[3 * H + hid], un[3], g4[3]));
        const float cy = fg * cst + ig * gg;
        hy = og * ftanh(cy);
        cst = cy;
        if (reserve) {
          *reinterpret_cast<f32x4*>(reserve + res_gate(bt, H, hid)) = f32x4{ig, gg, fg, og};
          reserve[res_cell((size_t)B * T, bt, H, hid)] = cy;
        }
        if constexpr (H0) { un_t = usc; hsc = F10H_HSC; }                        // |h_t| < 1 from here on
      } else {                                                                   // gin slots r, z, n, b_hid of n (the tier's convention)
        const float hn = fmaf(gbuf[2 * H + hid], un[2], g4[3]);
        const float rg = fsigmoid(fmaf(gbuf[hid], un[0], g4[0]));                 // gru.py:38-39
        const float zg = fsigmoid(fmaf(gbuf[H + hid], un[1], g4[1]));             // gru.py:40-41
        const float ng = ftanh(g4[2] + rg * hn);                                  // gru.py:42-43
        hy = (1.0f - zg) * ng + zg * hst;                                         // gru.py:44
        if (reserve) *reinterpret_cast<f32x4*>(reserve + (bt * H + hid) * 4) = f32x4{rg, zg, ng, hn};
        if constexpr (H0) {
          if (e_cur > 0) {                 // workgroup-uniform: the state was outside (-1, 1) — re-derive its exponent
            float mx = fabsf(hy);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (lane == 0) hmax[wave] = mx;
            lds_barrier();
            mx = fmaxf(fmaxf(hmax[0], hmax[1]), fmaxf(hmax[2], hmax[3]));
            const int e = f10h_expo(mx);
            e_cur = __builtin_amdgcn_readfirstlane(e < 0 ? 0 : e);
            hsc = ldexpf(F10H_HSC, -e_cur);
            un_t = usc * ldexpf(1.f, e_cur);
          }
        }
      }
      if constexpr (OUT) out[bt * H + hid] = hy;
      hst = hy;
      TT_STAMP(2)
      s2_from_lanes(hy * hsc);
      if (!IN1 && t + 1 < T) gi = gin4[(bt + 1) * H + hid];
    }
    if (IN1) xq.advance(xs, b * T, T, t, lane);
    TT_STAMP(3)
    lds_barrier();
    TT_STAMP(4)
  }
  if (hT) hT[b * H + hid] = hst;
  if (LSTM && cT) cT[b * H + hid] = cst;
  if constexpr (DIAG) {
    if (lane == 0 && reserve && b < 8) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(reserve) + (b * 8 + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = seg[i];
    }
  }
}

// the joint matrix of G per-gate matrices of shape S: cores (1,G,1,G), then S's cores in rank blocks
template <class S>
bool f10n_joint_matches(const TtShape& s, int G) {
  if (s.d != S::D + 1 || s.I[0] != G || s.J[0] != 1 || s.R[0] != 1 || s.R[1] != G) return false;
  for (int k = 0; k < S::D; ++k) {
    if (s.I[k + 1] != S::I[k] || s.J[k + 1] != S::J[k]) return false;
    if (s.R[k + 2] != (k == S::D - 1 ? 1 : G * S::R[k + 1])) return false;
  }
  return true;
}

template <class S, int CELL>
int launch_n(const RnnShape& rs, GinSrc gin, const float* bilv, const void* h0, const void* c0, const float* packed_hid, void* out,
             void* hT, void* cT, float* reserve, void* ws, hipStream_t stream) {
  if (!ws) return TTRNN_ERR_WORKSPACE;
  static_assert((F10H_EP + F10<S>::M) * sizeof(int) <= F10H_HDR_BYTES, "header");
  constexpr int G = f10n_gates(CELL);
  unsigned char* p = reinterpret_cast<unsigned char*>(ws);
  float* hdrs = reinterpret_cast<float*>(p); p += (size_t)G * F10H_HDR_BYTES;
  xh8* wfrag = reinterpret_cast<xh8*>(p); p += (size_t)G * f10n_frag_elems<S>() * sizeof(xh8);
  float* pg = reinterpret_cast<float*>(p);
  constexpr int TOT = f10n_packed_elems<S>();
  const long n = (long)G * TOT > rs.hid_s.wtotal ? (long)G * TOT : rs.hid_s.wtotal;
  hipLaunchKernelGGL((k_f10n_unjoin<S, G>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rs.hid_s, packed_hid, pg,
                     device_status_ptr());
  for (int g = 0; g < G; ++g)
    hipLaunchKernelGGL((k_f10h_scale<S>), dim3(F10<S>::M), dim3(256), 0, stream, (const float*)(pg + (size_t)g * TOT),
                       reinterpret_cast<int*>(hdrs + g * (F10H_HDR_BYTES / 4)));
  hipLaunchKernelGGL((k_f10n_prep<S>), dim3(G * F10<S>::MT * F10<S>::NM), dim3(64), 0, stream, (const float*)pg,
                     (const float*)hdrs, wfrag);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  const bool in1 = gin.in1 != 0;
  const bool hs = h0 || c0;
  auto kern = in1 ? (out ? (hs ? k_rnn_fwd_f10n<S, CELL, true, true, true> : k_rnn_fwd_f10n<S, CELL, false, true, true>)
                         : (hs ? k_rnn_fwd_f10n<S, CELL, true, false, true> : k_rnn_fwd_f10n<S, CELL, false, false, true>))
                  : (out ? (hs ? k_rnn_fwd_f10n<S, CELL, true, true, false> : k_rnn_fwd_f10n<S, CELL, false, true, false>)
                         : (hs ? k_rnn_fwd_f10n<S, CELL, true, false, false> : k_rnn_fwd_f10n<S, CELL, false, false, false>));
  if (opt(OPT_DIAG) && reserve && out && !hs)      // stamped build (diagnostics)
    kern = in1 ? k_rnn_fwd_f10n<S, CELL, false, true, true, true> : k_rnn_fwd_f10n<S, CELL, false, true, false, true>;
  hipLaunchKernelGGL(kern, dim3(rs.B), dim3(256), 0, stream, rs.B, rs.T, gin, bilv, (const float*)h0, (const float*)c0, (const float*)pg,
                     (const float*)hdrs, (const xh8*)wfrag, (float*)out, (float*)hT, (float*)cT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

}  // namespace

// the per-gate packed cores of a joint matrix, for the reverse-time kernel of the same sets (ttrnn_fast_f10bh.hip: k_rnn_bwd_f10n)
size_t f10n_unjoined_floats(int cell) { return (size_t)f10n_gates(cell) * f10n_packed_elems<ShpH256N>(); }
int launch_f10n_unjoin(const RnnShape& rs, const float* packed_hid, float* pg, hipStream_t stream) {
  constexpr int TOT = f10n_packed_elems<ShpH256N>();
  const int G = f10n_gates(rs.cell);
  if (!f10n_joint_matches<ShpH256N>(rs.hid_s, G)) return TTRNN_ERR_UNSUPPORTED;
  const long n = (long)G * TOT > rs.hid_s.wtotal ? (long)G * TOT : rs.hid_s.wtotal;
  if (G == 4)
    hipLaunchKernelGGL((k_f10n_unjoin<ShpH256N, 4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rs.hid_s, packed_hid, pg,
                       (unsigned*)nullptr);
  else
    hipLaunchKernelGGL((k_f10n_unjoin<ShpH256N, 3>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rs.hid_s, packed_hid, pg,
                       (unsigned*)nullptr);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}
bool f10n_shape_matches(const RnnShape& rs) {
  return (rs.cell == TTRNN_LSTM || rs.cell == TTRNN_GRU) && rs.hid_blocks == f10n_gates(rs.cell) &&
         f10n_joint_matches<ShpH256N>(rs.hid_s, f10n_gates(rs.cell));
}

// fp32-storage naive TT-LSTM / TT-GRU of H = 256, d = 3, r = 8 per gate, split math mode (dev bit 25: the runtime-shape tier's kernel, A/B)
bool f10n_available(const RnnShape& rs, int dtype) {
  const int G = f10n_gates(rs.cell);
  return !opt(OPT_NO_F10) && !(opt(OPT_DEV) & (1 << 25)) && rs.B >= 1 && rs.T >= 1 && dtype == TTRNN_F32 &&
         (rs.cell == TTRNN_LSTM || rs.cell == TTRNN_GRU) && rs.hid_blocks == G && opt(OPT_FP32_MATH) == TTRNN_MATH_SPLIT &&
         f10n_joint_matches<ShpH256N>(rs.hid_s, G);
}
size_t f10n_workspace_bytes(const RnnShape& rs) {
  const int G = f10n_gates(rs.cell);
  return f10n_joint_matches<ShpH256N>(rs.hid_s, G) && rs.hid_blocks == G ? f10n_ws_bytes<ShpH256N>(G) : 0;
}
int launch_lstm_fwd_f10n(const RnnShape& rs, GinSrc gin, const float* bilv, const void* h0, const void* c0, const float* packed_hid,
                         void* out, void* hT, void* cT, float* reserve, void* ws, hipStream_t stream) {
  if (!f10n_joint_matches<ShpH256N>(rs.hid_s, f10n_gates(rs.cell))) return TTRNN_ERR_UNSUPPORTED;
  return rs.cell == TTRNN_LSTM ? launch_n<ShpH256N, TTRNN_LSTM>(rs, gin, bilv, h0, c0, packed_hid, out, hT, cT, reserve, ws, stream)
                               : launch_n<ShpH256N, TTRNN_GRU>(rs, gin, bilv, h0, c0, packed_hid, out, hT, cT, reserve, ws, stream);
}

}  // namespace ttrnn
