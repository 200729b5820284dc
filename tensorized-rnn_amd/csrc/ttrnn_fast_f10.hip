// ttrnn_fast_f10.hip — forward kernels with cores 1 and 0 contracted ahead of the time loop (gfx950):
//   k_lstm_fwd_f10      fp32 TT-LSTM recurrent kernel (described below; KS = 2: k-split variant for r = 16)
//   k_ttlinear_fwd_f10  the batched input projection of stacked layers on the same two stages
//   k_gru_fwd_f10       bf16-storage TT-GRU recurrent kernel on the same fused core
//   k_f10_prep / k_f10g_prep  build the fused core's MFMA fragments once per launch
//
// For a d = 3 TT-matrix the per-timestep chain of t3nsor/ops.py:78-93 is three GEMM stages
//     S2: [J0*J1][J2]       x core2 -> [I2][J0*J1][R2]
//     S1: [I2*J0][J1*R2]    x core1 -> [I1][I2*J0][R1]
//     S0: [I2*I1][J0*R1]    x core0 -> [I0][I2*I1]            (gate pre-activations, o = (i0*I1 + i1)*I2 + i2)
// The cores never change inside a launch, so the contraction over the inner rank R1 can be done ONCE per launch:
//     W10[(j0,j1,r2)][(i0,i1)] = sum_r1 G0[i0,j0,r1] * G1[r1,i1,j1,r2]                 (K10 = J0*J1*R2, M10 = I0*I1)
//     S10: [I2][K10] x W10 -> [M10][I2]
// and the S2 output, viewed flat, IS the [I2][K10] operand.  For cfg2 (H=256, r=8) S10 costs 0.52 MFLOP per step
// instead of 0.52 + 0.07 for S1 + S0, for cfg4 (r=16) 1.05 instead of 2.23 — and, more important once the matrix
// pipe is no longer the bound, one whole LDS round trip, one barrier and one operand-splitting pass per timestep
// disappear.  W10 (M10 x K10 fp32-equivalent values) lives in VGPRs of four waves, one 16-feature tile each.
//
// Tile layout of S10: MFMA rows = output features, permuted so that row 4q+j of tile t is feature
// m = MPG*j + 4t + q (MPG = M10/4 features per gate); MFMA columns = i2.  Then the four accumulator registers of lane
// (c, q) are the four gate pre-activations (i, f, g, o) of ONE hidden unit, hid = (4t+q)*I2 + c: the gate math runs on
// the accumulators, c stays in that lane's register, h goes straight back to the image S2 reads.
//
// Per timestep:
//     phase A  all 8 waves   S2: two tiles (16 features x 16 chain rows) per wave, results split into three bf16 planes
//     barrier
//     phase B  waves 0-3     S10 on split MFMAs (ttrnn_split.h), gates, h_t -> LDS;  wave 7 streams h_{t-1} to `out`
//     barrier
// (Overlapping the second half of phase A with the first half of phase B on the partner waves of each SIMD was
// measured 5 % SLOWER: the MFMA stream of the gate wave takes half of the SIMD's issue slots from the splitting
// VALU work, and the extra barrier costs more than the overlap returns.)
// Operand precision of the two LSTM kernels (k_lstm_fwd_f10 here, k_lstm_fwd_f10_nb): every fp32 operand is carried as
// TWO fp16 pieces under a power-of-two scale fixed per launch (k_f10h_scale; ttrnn_split.h, ttrnn_f10_dev.h) and a
// product is x0w0 + x0w1 + x1w0 with fp32 accumulation — against three bf16 pieces and six terms this halves S10's
// matrix-pipe time (768 -> 384 cycles per SIMD and step), the splitting VALU work (7 -> 4 instructions per pair) and the
// LDS image (3 -> 2 planes): cfg2 0.79 -> 0.56 ms, error against float64 unchanged (= the fp32-MFMA mode's).
// S2 has a contraction length of only J2 = 8, so ALL FOUR of its terms are PACKED along the 32-wide k of one fp16 MFMA:
//     core groups [w0|w1|w0|w1] x activation groups [x0|x0|x1|x1]  =  x0w0 + x0w1 + x1w0 + x1w1
// (16 matrix-pipe cycles per tile instead of two fp32 MFMAs = 64).  k_ttlinear_fwd_f10 and k_gru_fwd_f10 keep bf16 pieces
// (three / one): the S2 packing of the former is  [w0|w1|w0|w1] x [x0|x0|x1|x1]  +  [w2|w0|0|0] x [x0|x2|-|-].
// Replaces, for one layer: tensorized_rnn/lstm.py:23-32,123-133 with the hidden chain of t3nsor/ops.py:78-93.
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

// The fused core, contracted and split ONCE per launch by a small kernel of its own and stored in fragment order:
//   wfrag[((t*NM + u)*3 + plane)*64 + lane] = the 8 bf16 (16 bytes) lane (r, q) feeds the MFMA for feature tile t,
//   k-block u:  MFMA row r <-> feature m = MPG*(r&3) + 4t + (r>>2),  k = 32u + 8q + e in F10::kperm order.
// The recurrent workgroups then fetch their resident fragments with coalesced 16-byte loads (1 KB per instruction).
// (Computing W10 inside every workgroup's prologue instead cost ~1 000 scattered 4-byte loads per lane: ~50 us of the
// 860 us cfg2 launch, and half of a 160-step cfg4 launch.)
template <class S>
__global__ void __launch_bounds__(64) k_f10_prep(const float* __restrict__ packed, xbf8* __restrict__ wfrag) {
  using F = F10<S>;
  const int lane = threadIdx.x, u = blockIdx.x % F::NM, t = blockIdx.x / F::NM;
  const int r = lane & 15, q = lane >> 4;
  const int m = F::MPG * (r & 3) + 4 * t + (r >> 2);     // gate r&3, feature-within-gate 4t + (r>>2)
  const int i0 = m / F::I1, i1 = m % F::I1;
  const float* W0 = packed + woff_of<S>(0);               // [J0*R1][I0]
  const float* W1 = packed + woff_of<S>(1);               // [J1*R2][I1*R1]
  xbf8 f0, f1, f2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int slot = 4 * u + q;                           // k = 8*slot + e in F10::kperm order
    const int r2 = (slot / F::HR) * 4 + (e & 3), row2 = 2 * (slot % F::HR) + (e >> 2);
    const int j1 = row2 % F::J1, j0 = row2 / F::J1;
    const float* w1p = W1 + (j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1;
    float v = 0.f;
    for (int r1 = 0; r1 < F::R1; ++r1) v = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0], w1p[r1], v);
    __bf16 p0, p1, p2;
    split3(v, p0, p1, p2);
    f0[e] = p0; f1[e] = p1; f2[e] = p2;
  }
  xbf8* dst = wfrag + (size_t)((t * F::NM + u) * 3) * 64 + lane;
  dst[0] = f0; dst[64] = f1; dst[128] = f2;
}

template <class S>
constexpr size_t f10_wfrag_bytes() { return (size_t)F10<S>::MT * F10<S>::NM * 3 * 64 * sizeof(xbf8); }

// ---- two-piece fp16 operands (LSTM forward kernels): scale header (k_f10h_scale: ttrnn_f10_dev.h) + fragments ---------
// The fused core in fragment order (see k_f10_prep), rows pre-multiplied by -log2(e) (gates i, f, o) / 2 log2(e) (gate g)
// and by the header's 2^(ep[m] - ev[r2]), as two fp16 pieces:  wfrag[((t*NM + u)*2 + plane)*64 + lane].
template <class S>
__global__ void __launch_bounds__(64) k_f10h_prep(const float* __restrict__ packed, const float* __restrict__ hdr,
                                                  xh8* __restrict__ wfrag) {
  using F = F10<S>;
  const int lane = threadIdx.x, u = blockIdx.x % F::NM, t = blockIdx.x / F::NM;
  const int r = lane & 15, q = lane >> 4;
  const int m = F::MPG * (r & 3) + 4 * t + (r >> 2);     // gate r&3, feature-within-gate 4t + (r>>2)
  const int i0 = m / F::I1, i1 = m % F::I1;
  const float* W0 = packed + woff_of<S>(0);               // [J0*R1][I0]
  const float* W1 = packed + woff_of<S>(1);               // [J1*R2][I1*R1]
  const float gf = (r & 3) == 2 ? 2.8853900817779268f : -1.4426950408889634f;
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
    split2h_scaled(v, gf * f10h_w_scale<S>(hdr, m, r2), p0, p1);      // (pinned rounding: ttrnn_split.h)
    f0[e] = p0; f1[e] = p1;
  }
  xh8* dst = wfrag + (size_t)((t * F::NM + u) * 2) * 64 + lane;
  dst[0] = f0; dst[64] = f1;
}

// KS = 1: waves 0..MT-1 run the whole contraction of their tile.  KS = 2 (long contractions: the resident fragments
// of a whole tile row would not fit the register file): waves t and t+4 — the two waves of one SIMD — take one half of
// the k-blocks each, the second hands its partial accumulators to the first through LDS.
template <class S, int KS, bool DIAG, bool H0>
__global__ void __launch_bounds__(FAST_NT) k_lstm_fwd_f10(int B, int T, GinSrc gs, const float* __restrict__ h0,
                                                          const float* __restrict__ c0,
                                                          const float* __restrict__ packed_hid,
                                                          const float* __restrict__ hdr,
                                                          const xh8* __restrict__ wfrag,
                                                          const float* __restrict__ bias_hid, float* __restrict__ out,
                                                          float* __restrict__ hT, float* __restrict__ cT,
                                                          float* __restrict__ reserve) {
  static_assert(f10_ok<S>(), "shape not supported by the fused-core kernel");
  using F = F10<S>;
  constexpr int H = F::H;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* hbuf = reinterpret_cast<float*>(smem);                                   // fp32 h, two parities (output store)
  _Float16* hpl = reinterpret_cast<_Float16*>(smem + 2 * sizeof(float) * H);      // fp16 pieces of 2^sH h: [parity][2][H]
  _Float16* img = hpl + 2 * 2 * H;                                                // two fp16 planes [I2][K10]
  f32x4* xbuf = reinterpret_cast<f32x4*>(img + 2 * F::PLANE);                      // KS == 2: partial accumulators

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;
  static_assert(KS == 1 || (KS == 2 && F::MT == 4 && F::NM % 2 == 0), "k-split layout");
  constexpr int NU = F::NM / KS;                         // k-blocks per MFMA wave
  const bool gate_wave = wave < F::MT;
  const bool mma_wave = KS == 2 || gate_wave;
  const int tile = KS == 2 ? (wave & 3) : wave;          // S10 feature tile of this wave
  const int u0 = KS == 2 ? (wave >> 2) * NU : 0;         // its first k-block
  const F10hScales fsc = f10h_scales<S>(hdr, tile & (F::MT - 1), lane);     // diagonal power-of-two scales (ttrnn_f10_dev.h)
  const float hsc = F10H_HSC;
  const f32x4 psc = fsc.pre, usc = fsc.un;

  // S2 fragments of the m-tiles {wave + 8x}
  xh8 s1[F::XA];
#pragma unroll
  for (int x = 0; x < F::XA; ++x) f10h_load_w2<S>(s1[x], packed_hid, wave + FAST_NW * x, lane, hdr);
  xh8 w10[2][NU];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) w10[p][u][e] = (_Float16)0.f;
  if (mma_wave) {
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int p = 0; p < 2; ++p) w10[p][u] = wfrag[(size_t)((tile * F::NM + u0 + u) * 2 + p) * 64 + lane];
  }

  // the hidden unit of this lane in phase B (waves 0 .. MT-1): hid = (4*wave + q)*I2 + c, gates in acc[0..3] = i,f,g,o.
  // gin is gate-interleaved [B][T][H][4] with slots i,g,f,o.
  const float* __restrict__ gin = gs.gin;
  const float* __restrict__ xs = reinterpret_cast<const float*>(gs.x);
  const bool in1 = gs.in1 != 0;
  const bool ok = gate_wave && c < F::I2;
  const int hd = ok ? (4 * wave + q) * F::I2 + c : 0;
  float hst = (H0 && ok) ? h0[b * H + hd] : 0.f;       // H0: the caller passed an initial state (f10h_h0_expo)
  float cst = (ok && c0) ? c0[b * H + hd] : 0.f;
  float h0sc = 1.0f, h0un = 1.0f;
  if constexpr (H0) {
    const int e0 = f10h_h0_expo<FAST_NW>(hst, reinterpret_cast<float*>(img), wave, lane);
    h0sc = ldexpf(1.f, -e0); h0un = ldexpf(1.f, e0);
  }
  f32x4 bh = f32x4{0.f, 0.f, 0.f, 0.f}, gi = bh, vv = bh, bb = bh;       // slot order i,g,f,o
  // The fused core's rows are pre-multiplied (k_f10h_prep) so that an accumulator, times the header's 2^-S, IS the
  // argument of v_exp_f32: sigmoid(x) = 1 / (1 + 2^(-log2e x)), tanh(x) = 1 - 2 / (1 + 2^(2 log2e x)).  The input
  // projection + biases get the same factors (and 2^S) and enter the MFMA chain as its initial accumulator value, so
  // nothing but the merge of the two chains stands between the last MFMA and the transcendental unit.
  const f32x4 gsc = f32x4{-1.4426950408889634f, 2.8853900817779268f, -1.4426950408889634f, -1.4426950408889634f} *
                    f32x4{psc[0], psc[2], psc[1], psc[3]};      // slots i,g,f,o <- accumulator rows i,f,g,o
  XChunk<float> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
  if (ok) {
    if (bias_hid) bh = f32x4{bias_hid[hd], bias_hid[2 * H + hd], bias_hid[H + hd], bias_hid[3 * H + hd]};
    if (T > 0) {
      if (in1) {
        bb = *reinterpret_cast<const f32x4*>(gin + (H + hd) * 4);
        vv = (*reinterpret_cast<const f32x4*>(gin + hd * 4) - bb) * gsc;
        bb = (bb + bh) * gsc;
      } else {
        gi = *reinterpret_cast<const f32x4*>(gin + ((b * T) * H + hd) * 4);
      }
    }
    _Float16 p0, p1;                                       // parity 0 = h_{-1}
    split2h(hst * (hsc * h0sc), p0, p1);
    hpl[hd] = p0; hpl[H + hd] = p1;
    hbuf[hd] = hst;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) here, so that no weight-register wait lands inside the loop
  lds_barrier();
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();

  const int row10 = c < F::I2 ? c : F::I2 - 1;
  f32x4 us_t = usc * h0un;                  // step 0 runs on 2^-e0 h_0 (f10h_h0_expo); reset to usc / 1 at the end of it
  float ps_t = h0sc;
  for (int t = 0; t < T; ++t) {
    const _Float16* hp = hpl + (t & 1) * 2 * H;           // pieces of h_{t-1}
    _Float16* hn = hpl + ((t + 1) & 1) * 2 * H;           // pieces of h_t
    // ---- phase A: S2, all waves ---------------------------------------------------------------------------
    // all MFMAs first, then the splitting: the VALU work of one tile runs in the shadow of the others' MFMA latency
    // instead of behind an s_nop after every pair
    {
      f32x4 t2[F::XA][2];
#pragma unroll
      for (int x = 0; x < F::XA; ++x) {
        t2[x][0] = f10h_s2_mma<S>(s1[x], hp, 0, lane);
        t2[x][1] = f10h_s2_mma<S>(s1[x], hp, 1, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < F::XA; ++x) {
        f10h_s2_store<S>(t2[x][0], img, wave + FAST_NW * x, 0, lane);
        f10h_s2_store<S>(t2[x][1], img, wave + FAST_NW * x, 1, lane);
      }
    }
    TT_STAMP(0)
    lds_barrier();
    TT_STAMP(1)
    const size_t bt = b * T + t;
    // ---- phase B: the fused S1*S0 stage, then gates + state (lstm.py:26-32) -----------------------------------
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (mma_wave) {
      f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
      if (gate_wave) {
        // (W_in x_t + b_in + b_hid) * scale, slots i,g,f,o -> accumulator rows i,f,g,o
        f32x4 pre = in1 ? bb + xq.at(t) * vv : (gi + bh) * gsc;
        if constexpr (H0) pre = pre * ps_t;
        acc_hi = f32x4{pre[0], pre[2], pre[1], pre[3]};
      }
      f10h_s10_part<S, NU>(w10, img, row10, q, u0, acc_lo, acc_hi);
      const f32x4 un = H0 ? us_t : usc;
      acc = acc_hi * un + acc_lo * un;                  // 2^-S per row and column (2^(e0-S) at step 0 of a given h_0), exact
      if constexpr (DIAG) {
        asm volatile("" : "+v"(acc));
      }
    }
    TT_STAMP(2)
    if constexpr (KS == 2) {
      if (!gate_wave) xbuf[tile * 64 + lane] = acc;
      lds_barrier();
      if (gate_wave) acc += xbuf[tile * 64 + lane];
    }
    if (gate_wave) {
      // acc = -log2e * pre-activation (i, f, o) / 2 log2e * pre-activation (g)
      const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0]));                // lstm.py:26
      const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[1]));                // lstm.py:27
      const float gg = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[2]));  // lstm.py:28
      const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[3]));                // lstm.py:29
      const float cy = fg * cst + ig * gg;                    // lstm.py:31
      const float hy = og * ftanh(cy);                        // lstm.py:32
      if (ok) {
        cst = cy;
        hst = hy;
        _Float16 p0, p1;
        split2h(hy * hsc, p0, p1);
        hn[hd] = p0; hn[H + hd] = p1;
        hbuf[((t + 1) & 1) * H + hd] = hy;
        if (reserve) {
          *reinterpret_cast<f32x4*>(reserve + res_gate(bt, H, hd)) = f32x4{ig, gg, fg, og};
          reserve[res_cell((size_t)B * T, bt, H, hd)] = cy;
        }
        if (!in1 && t + 1 < T) gi = *reinterpret_cast<const f32x4*>(gin + ((bt + 1) * H + hd) * 4);
      }
      if (in1) xq.advance(xs, b * T, T, t, lane);
      TT_STAMP(3)
    } else if (wave == FAST_NW - 1 && t > 0 && out) {      // out == NULL: final state only
      // outputs[:, t-1, :] = h_{t-1} (lstm.py:133): an idle wave streams the complete vector out, 16 bytes per lane
      const float* hprev = hbuf + (t & 1) * H;
#pragma unroll
      for (int h4 = lane; h4 < H / 4; h4 += 64)
        *reinterpret_cast<f32x4*>(out + (bt - 1) * H + 4 * h4) = *reinterpret_cast<const f32x4*>(hprev + 4 * h4);
    }
    us_t = usc; ps_t = 1.0f;
    lds_barrier();
    TT_STAMP(4)
  }
  if (T > 0 && wave == FAST_NW - 1 && out) {
    const float* hlast = hbuf + (T & 1) * H;
#pragma unroll
    for (int h4 = lane; h4 < H / 4; h4 += 64)
      *reinterpret_cast<f32x4*>(out + (b * T + T - 1) * H + 4 * h4) = *reinterpret_cast<const f32x4*>(hlast + 4 * h4);
  }
  if (ok) {
    if (hT) hT[b * H + hd] = hst;
    if (cT) cT[b * H + hd] = cst;
  }
  if constexpr (DIAG) {
    if (lane == 0 && reserve && b < 8) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(reserve) + (b * FAST_NW + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = seg[i];
    }
  }
}

// ---- batched input projection (K-in) of the layers above the first one, on the same fused core ----------------------
// gin[n][hid][slot] = W_in x[n] + b for n = B*T rows of the previous layer's output (in = H): the matrix has the
// hidden-to-hidden shape, so the two-stage form S2 -> S10 applies unchanged; rows are independent, so a workgroup
// simply walks over rows n = blockIdx.x, blockIdx.x + gridDim.x, ...  Per row: x[n] (1 KB) is split into three bf16
// planes by wave 0 one row ahead (during the MFMA phase of the previous row), phase A = S2 tiles on all waves,
// phase B = S10 k-halves on all waves (KS = 2 layout), the gate waves add the partner's partial sums and the bias and
// store one 16-byte gate quadruple (slots i,g,f,o) per lane.  Two barriers per row.
template <class S>
__global__ void __launch_bounds__(FAST_NT) k_ttlinear_fwd_f10(long n_rows, const float* __restrict__ x,
                                                              const float* __restrict__ packed,
                                                              const xbf8* __restrict__ wfrag,
                                                              const float* __restrict__ bias, float* __restrict__ y) {
  static_assert(f10_in_ok<S>(), "shape not supported by the fused-core input projection");
  using F = F10<S>;
  constexpr int IN = F::H, HID = F::HID, XPL = F::XPL;
  constexpr int NU = F::NM / 2;
  constexpr bool DENSE = F::J2 == 8 && XPL == IN;        // x row == the S2 operand, no padding (hidden-shaped matrices)
  constexpr int XE = XPL / 64;                            // padded elements per lane of wave 0 (generic path)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* xpl = reinterpret_cast<__bf16*>(smem);                            // bf16 planes of x: [parity][3][XPL]
  __bf16* img = xpl + 2 * 3 * XPL;                                          // three bf16 planes [I2][K10]
  f32x4* xbuf = reinterpret_cast<f32x4*>(img + 3 * F::PLANE);                // partial accumulators

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const bool gate_wave = wave < F::MT;
  const int tile = wave & 3, u0 = (wave >> 2) * NU;

  xbf8 s1[F::XA], s2[F::XA];
#pragma unroll
  for (int xx = 0; xx < F::XA; ++xx) f10_load_w2<S>(s1[xx], s2[xx], packed, wave + FAST_NW * xx, lane);
  xbf8 w10[3][NU];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int p = 0; p < 3; ++p) w10[p][u] = wfrag[(size_t)((tile * F::NM + u0 + u) * 3 + p) * 64 + lane];

  const bool ok = gate_wave && c < F::I2;
  const int hd = ok ? (4 * wave + q) * F::I2 + c : 0;
  f32x4 bh = f32x4{0.f, 0.f, 0.f, 0.f};                                     // slot order i,g,f,o
  if (ok && bias) bh = f32x4{bias[hd], bias[2 * HID + hd], bias[HID + hd], bias[3 * HID + hd]};

  // wave 0 carries the x row one row ahead in registers: DENSE four floats per lane; otherwise the XE padded
  // positions p = lane + 64e of the [rows][8] operand (position (row2, j2) <- x[row2*J2 + j2], zero outside)
  const long G = gridDim.x;
  long n = blockIdx.x;
  f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](long r) {
    if constexpr (DENSE) {
      xv = *reinterpret_cast<const f32x4*>(x + r * IN + 4 * lane);
    } else {
#pragma unroll
      for (int e = 0; e < XE; ++e) {
        const int p = lane + 64 * e, row2 = p >> 3, j2 = p & 7;
        const bool in = row2 < F::ROWS2 && j2 < F::J2;
        const float v = x[r * IN + (in ? row2 * F::J2 + j2 : 0)];
        xv[e] = in ? v : 0.f;
      }
    }
  };
  auto put = [&](__bf16* dst) {
    if constexpr (DENSE) {
      store_split4(dst, XPL, 4 * lane, xv);
    } else {
#pragma unroll
      for (int e = 0; e < XE; ++e) {
        __bf16 p0, p1, p2;
        split3(xv[e], p0, p1, p2);
        dst[lane + 64 * e] = p0; dst[XPL + lane + 64 * e] = p1; dst[2 * XPL + lane + 64 * e] = p2;
      }
    }
  };
  static_assert(DENSE || XE <= 4, "x row too long for the register prefetch");
  if (wave == 0) {
    if (n < n_rows) fetch(n);
    put(xpl);
    if (n + G < n_rows) fetch(n + G);
  }
  lds_barrier();
  const int row10 = c < F::I2 ? c : F::I2 - 1;
  int par = 0;
  for (; n < n_rows; n += G, par ^= 1) {
    const __bf16* xp = xpl + par * 3 * XPL;
    // ---- phase A: S2 ------------------------------------------------------------------------------------------
    {
      f32x4 t2[F::XA][F::RT2];      // all MFMAs first, then the splitting (see k_lstm_fwd_f10)
#pragma unroll
      for (int xx = 0; xx < F::XA; ++xx)
#pragma unroll
        for (int rt = 0; rt < F::RT2; ++rt) t2[xx][rt] = f10_s2_mma<S>(s1[xx], s2[xx], xp, rt, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int xx = 0; xx < F::XA; ++xx)
#pragma unroll
        for (int rt = 0; rt < F::RT2; ++rt) f10_s2_store<S>(t2[xx][rt], img, wave + FAST_NW * xx, rt, lane);
    }
    lds_barrier();
    // ---- phase B: S10 k-halves; wave 0 first puts the next row's planes in place -----------------------------------
    if (wave == 0 && n + G < n_rows) {
      put(xpl + (par ^ 1) * 3 * XPL);
      if (n + 2 * G < n_rows) fetch(n + 2 * G);
    }
    f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
    f10_s10_part<S, NU>(w10, img, row10, q, u0, acc_lo, acc_hi);
    f32x4 acc = acc_hi + acc_lo;
    if (!gate_wave) xbuf[tile * 64 + lane] = acc;
    lds_barrier();
    if (ok) {
      acc += xbuf[tile * 64 + lane];
      // accumulator registers are i,f,g,o (reference gate order); the interleaved row wants slots i,g,f,o
      *reinterpret_cast<f32x4*>(y + ((size_t)n * HID + hd) * 4) =
          f32x4{acc[0] + bh[0], acc[2] + bh[1], acc[1] + bh[2], acc[3] + bh[3]};
    }
  }
}

template <class S>
constexpr size_t f10_lin_lds_bytes() {
  return 2 * 3 * 2 * (size_t)F10<S>::XPL + 2 * 3 * (size_t)F10<S>::PLANE + 4 * 64 * sizeof(f32x4);
}

template <class S>
static int launch_lin_f10(long n_rows, const float* packed, const void* bias, const void* x, void* y, void* ws,
                          hipStream_t stream) {
  if (!ws) return TTRNN_ERR_WORKSPACE;
  xbf8* wfrag = reinterpret_cast<xbf8*>(ws);
  hipLaunchKernelGGL((k_f10_prep<S>), dim3(F10<S>::MT * F10<S>::NM), dim3(64), 0, stream, packed, wfrag);
  constexpr size_t lds = f10_lin_lds_bytes<S>();
  static_assert(lds <= 64 * 1024, "raise the dynamic LDS limit for this shape");
  const int cus = device_cu_count();
  const long grid = n_rows < (long)cus ? n_rows : (long)cus;       // one resident workgroup per CU walks the rows
  hipLaunchKernelGGL((k_ttlinear_fwd_f10<S>), dim3((unsigned)grid), dim3(FAST_NT), lds, stream, n_rows,
                     (const float*)x, packed, wfrag, (const float*)bias, (float*)y);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// gate-interleaved LSTM input projection (ilv_mode 2) of fp32 rows through a hidden-shaped TT-matrix
bool f10_ttlinear_fwd_available(const TtShape& s, int dtype, int ilv_h, int ilv_mode) {
  if (opt(OPT_NO_F10) || dtype != TTRNN_F32 || ilv_h != 256 || ilv_mode != 2) return false;
  return shape_matches<ShpH256R8L>(s) || shape_matches<ShpH256R16L>(s) || shape_matches<ShpI40R16L>(s);
}

size_t f10_ttlinear_workspace_bytes(const TtShape& s, int dtype, int ilv_h, int ilv_mode) {
  if (dtype != TTRNN_F32 || ilv_h != 256 || ilv_mode != 2) return 0;
  if (shape_matches<ShpH256R8L>(s)) return f10_wfrag_bytes<ShpH256R8L>();
  if (shape_matches<ShpH256R16L>(s)) return f10_wfrag_bytes<ShpH256R16L>();
  if (shape_matches<ShpI40R16L>(s)) return f10_wfrag_bytes<ShpI40R16L>();
  return 0;
}

int launch_ttlinear_fwd_f10(const TtShape& s, int64_t n_rows, const float* packed, const void* bias, const void* x,
                            void* y, void* ws, hipStream_t stream) {
  if (n_rows <= 0) return TTRNN_OK;
  if (shape_matches<ShpH256R8L>(s)) return launch_lin_f10<ShpH256R8L>((long)n_rows, packed, bias, x, y, ws, stream);
  if (shape_matches<ShpH256R16L>(s)) return launch_lin_f10<ShpH256R16L>((long)n_rows, packed, bias, x, y, ws, stream);
  if (shape_matches<ShpI40R16L>(s)) return launch_lin_f10<ShpI40R16L>((long)n_rows, packed, bias, x, y, ws, stream);
  return TTRNN_ERR_UNSUPPORTED;
}

// ---- bf16-storage GRU on the same fused core ---------------------------------------------------------------------
// cfg3 (TT-GRU, bf16 storage, fp32 state / gates / accumulation): the chain runs directly on the bf16 MFMA (no
// splitting: storage precision is bf16), with W10 rounded to bf16 once.  A GRU's r, z, n pre-activations of one
// hidden unit do NOT land in one lane (o = m*I2 + i2 with I2 = 12 does not align with the gate boundaries), so the
// fused stage leaves them in an fp32 LDS vector and a third phase does the gate math, one hidden unit per thread:
//     A  S2: (16 features x 16 chain rows) tiles over the 8 waves, K = J2 = 8 zero-padded to one bf16 MFMA
//     B  S10 (waves 0-3): 8 MFMAs per wave -> gate pre-activations into gbuf
//     C  gates (gru.py:38-44), h_t -> bf16 image for S2, `out`
// three barriers per step instead of the four phases of the stage-wise kernel (ttrnn_fast_bf16.hip).
template <class S>
constexpr bool f10g_ok() {
  using F = F10<S>;
  return S::D == 3 && S::R[0] == 1 && S::R[3] == 1 && shape_ok_recurrent<S>() && F::K % 32 == 0 && F::M == 64 &&
         F::I2 <= 16 && out_size_of<S>() == 3 * F::H && F::J2 == 8 && F::ROWS2 == 32 && F::M2 % 16 == 0 &&
         F::MT2 <= 2 * FAST_NW && F::R2 % 4 == 0 && F::H <= FAST_NT;
}

template <class S>
__global__ void __launch_bounds__(64) k_f10g_prep(const float* __restrict__ packed, xbf8* __restrict__ wfrag) {
  using F = F10<S>;
  const int lane = threadIdx.x, u = blockIdx.x % F::NM, t = blockIdx.x / F::NM;
  const int r = lane & 15, q = lane >> 4;
  const int m = 16 * t + r;                                // natural feature order
  const int i0 = m / F::I1, i1 = m % F::I1;
  const float* W0 = packed + woff_of<S>(0);
  const float* W1 = packed + woff_of<S>(1);
  xbf8 f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int slot = 4 * u + q;
    const int r2 = (slot / F::HR) * 4 + (e & 3), row2 = 2 * (slot % F::HR) + (e >> 2);
    const int j1 = row2 % F::J1, j0 = row2 / F::J1;
    const float* w1p = W1 + (j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1;
    float v = 0.f;
    for (int r1 = 0; r1 < F::R1; ++r1) v = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0], w1p[r1], v);
    f[e] = (__bf16)v;
  }
  wfrag[(size_t)(t * F::NM + u) * 64 + lane] = f;
}

template <class S>
__global__ void __launch_bounds__(FAST_NT) k_gru_fwd_f10(int B, int T, GinSrc gs, const bf16_t* __restrict__ h0,
                                                         const float* __restrict__ packed_hid,
                                                         const xbf8* __restrict__ wfrag,
                                                         const bf16_t* __restrict__ bias_hid, bf16_t* __restrict__ out,
                                                         bf16_t* __restrict__ hT, float* __restrict__ reserve) {
  static_assert(f10g_ok<S>(), "shape not supported by the fused-core GRU kernel");
  using F = F10<S>;
  constexpr int H = F::H;
  constexpr int NT2 = F::MT2 * 2;                          // S2 tiles (m-tile, chain-row tile)
  constexpr int XT = (NT2 + FAST_NW - 1) / FAST_NW;        // per wave

  __shared__ __attribute__((aligned(16))) __bf16 hq[H];                  // h_{t-1}, bf16, [ROWS2][J2]
  __shared__ __attribute__((aligned(16))) __bf16 img[F::PLANE];          // S10 operand [I2][K10]
  __shared__ __attribute__((aligned(16))) float gbuf[3 * H];             // gate pre-activations of the hidden chain

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  // S2 fragments: tile id = wave + 8x -> (mt = id % MT2, rt = id / MT2); K = 8 real k values live in k-group 0
  xbf8 a2[XT];
#pragma unroll
  for (int x = 0; x < XT; ++x) {
    const int id = wave + FAST_NW * x, mt = id % F::MT2;
    const float* W2 = packed_hid + woff_of<S>(2);          // [J2][M2]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = W2[e * F::M2 + 16 * mt + c];          // lane (r = c, q): feature 16mt + r
      a2[x][e] = (__bf16)((q == 0 && id < NT2) ? v : 0.f);
    }
  }
  xbf8 w10[F::NM];
#pragma unroll
  for (int u = 0; u < F::NM; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) w10[u][e] = (__bf16)0.f;
  if (wave < 4) {
#pragma unroll
    for (int u = 0; u < F::NM; ++u) w10[u] = wfrag[(size_t)(wave * F::NM + u) * 64 + lane];
  }

  // gate phase: thread tid < H owns hidden unit tid.  gin: fp32 [B][T][H][4], slots r,z,n,-
  const float* __restrict__ gin = gs.gin;
  const bf16_t* __restrict__ xs = reinterpret_cast<const bf16_t*>(gs.x);
  const bool in1 = gs.in1 != 0;
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gin);
  const bool own = tid < H;
  const int hid = own ? tid : 0;
  float hst = (own && h0) ? ld(h0, b * H + hid) : 0.f;
  float bh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) bh[g] = (own && bias_hid) ? ld(bias_hid, g * H + hid) : 0.f;
  f32x4 gi = f32x4{0.f, 0.f, 0.f, 0.f}, vv = gi, bb = gi;
  XChunk<bf16_t> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
  if (own && T > 0) {
    if (in1) {
      bb = gin4[H + hid];
      vv = gin4[hid] - bb;
    } else {
      gi = gin4[(b * T) * H + hid];
    }
  }
  if (own) hq[hid] = (__bf16)hst;
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): keep weight-register waits out of the time loop
  lds_barrier();

  const int row10 = c < F::I2 ? c : F::I2 - 1;
  for (int t = 0; t < T; ++t) {
    // ---- A: S2 ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int x = 0; x < XT; ++x) {
      const int id = wave + FAST_NW * x;
      if (id < NT2) {
        const int mt = id % F::MT2, rt = id / F::MT2;
        const int row = 16 * rt + c;
        const xbf8 bfrag = *reinterpret_cast<const xbf8*>(hq + row * 8);     // every k-group reads the same 16 bytes
        const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[x], bfrag, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const int m0 = 16 * mt + 4 * q;
        const int i = m0 / F::R2, a0 = m0 % F::R2;
        xbf4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)acc[j];
        *reinterpret_cast<xbf4*>(img + x_off<F::K>(i, F::kperm(row, a0))) = o;
      }
    }
    lds_barrier();
    // ---- B: fused S1*S0 stage, waves 0-3 -------------------------------------------------------------------------
    if (wave < 4) {
      xbf8 af[F::NM];
#pragma unroll
      for (int u = 0; u < F::NM; ++u) af[u] = *reinterpret_cast<const xbf8*>(img + x_off<F::K>(row10, 32 * u + 8 * q));
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int u = 0; u < F::NM; u += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10[u], af[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10[u + 1], af[u + 1], acc1, 0, 0, 0);
      }
      const f32x4 acc = acc0 + acc1;
      if (c < F::I2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) gbuf[(16 * wave + 4 * q + j) * F::I2 + c] = acc[j];   // o = m*I2 + i2
      }
    }
    lds_barrier();
    // ---- C: gates + state (gru.py:38-44) ---------------------------------------------------------------------------
    const size_t bt = b * T + t;
    if (own) {
      if (in1) gi = bb + xq.at(t) * vv;
      const float hn = gbuf[2 * H + hid] + bh[2];
      const float rg = fsigmoid(gi[0] + gbuf[hid] + bh[0]);              // gru.py:38-39
      const float zg = fsigmoid(gi[1] + gbuf[H + hid] + bh[1]);          // gru.py:40-41
      const float ng = ftanh(gi[2] + rg * hn);                           // gru.py:42-43
      float hy = (1.0f - zg) * ng + zg * hst;                            // gru.py:44
      if (reserve) *reinterpret_cast<f32x4*>(reserve + (bt * H + hid) * 4) = f32x4{rg, zg, ng, hn};
      const bf16_t hb = f32_to_bf16(hy);            // rounded once: stored, fed back, kept as state
      if (out) out[bt * H + hid] = hb;              // (A/B kernel: a uniform branch; out == nullptr: final state only)
      hy = bf16_to_f32(hb);
      hst = hy;
      hq[hid] = (__bf16)hy;
      if (!in1 && t + 1 < T) gi = gin4[(bt + 1) * H + hid];
    }
    if (in1) xq.advance(xs, b * T, T, t, lane);
    lds_barrier();
  }
  if (own && hT) st(hT, b * H + hid, hst);
}

// Four-wave variant with S2 INSIDE the gate waves (the default; option `dev` bit 5 selects the eight-wave kernel above for
// A/B — the two agree to one bf16 ulp of the stored C2 / h: the MFMA sums the same eight products with k in other slots):
// wave w owns hidden units
// 64w .. 64w+63 = chain rows 8w .. 8w+7, so the S2 operand of those rows — eight consecutive units per row — lives in eight
// lanes of the same wave: paired by one DPP shift and gathered with ONE ds_bpermute per lane (the LDS crossbar, no memory, no
// barrier; S2's eight k values are spread two per k-group so that a lane needs one pair), multiplied by G2 for all
// six m-tiles (16-column tiles of which 8 columns are real: the matrix pipe idles 90 % of the step anyway) and written to the
// S10 image.  One LDS hand-off and one barrier per step fewer than k_gru_fwd_f10: [S10 | barrier | gates + S2 | barrier].
// OUT = false: the caller consumes only the final state (mnist_classifier.py:52-55 classifies the last step): no out store
// IN1: input_size == 1 as a template parameter (round 4; see k_lstm_fwd_f10q)
template <class S, bool OUT, bool IN1>
__global__ void __launch_bounds__(256, 2) k_gru_fwd_f10v(int B, int T, GinSrc gs, const bf16_t* __restrict__ h0,
                                                      const float* __restrict__ packed_hid,
                                                      const xbf8* __restrict__ wfrag,
                                                      const bf16_t* __restrict__ bias_hid, bf16_t* __restrict__ out,
                                                      bf16_t* __restrict__ hT, float* __restrict__ reserve) {
  static_assert(f10g_ok<S>() && F10<S>::H == 256 && F10<S>::J2 == 8, "shape not supported by the fused-core GRU kernel");
  using F = F10<S>;
  constexpr int H = F::H;
  __shared__ __attribute__((aligned(16))) __bf16 img[F::PLANE];          // S10 operand [I2][K10]
  __shared__ __attribute__((aligned(16))) float gbuf[3 * H];             // gate pre-activations of the hidden chain

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  // S2 fragments of ALL m-tiles: lane (r = c, q): feature 16mt + r.  The 8 real k values (j2) are spread over the four
  // k-groups, two each (slots 8q, 8q + 1 = j2 2q, 2q + 1; the other six slots of a group are zero): a lane of the B operand then
  // needs ONE pair of units of its row — one gather instead of four
  xbf8 a2[F::MT2];
  {
    const float* W2 = packed_hid + woff_of<S>(2);          // [J2][M2]
#pragma unroll
    for (int mt = 0; mt < F::MT2; ++mt)
#pragma unroll
      for (int e = 0; e < 8; ++e) a2[mt][e] = (__bf16)(e < 2 ? W2[(2 * q + e) * F::M2 + 16 * mt + c] : 0.f);
  }
  xbf8 w10[F::NM];
#pragma unroll
  for (int u = 0; u < F::NM; ++u) w10[u] = wfrag[(size_t)(wave * F::NM + u) * 64 + lane];

  const float* __restrict__ gin = gs.gin;
  const bf16_t* __restrict__ xs = reinterpret_cast<const bf16_t*>(gs.x);
  constexpr bool in1 = IN1;
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gin);
  const int hid = tid;
  float hst = h0 ? ld(h0, b * H + hid) : 0.f;
  float bh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) bh[g] = bias_hid ? ld(bias_hid, g * H + hid) : 0.f;
  f32x4 gi = f32x4{0.f, 0.f, 0.f, 0.f}, vv = gi, bb = gi;
  XChunk<bf16_t> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
  if (T > 0) {
    if (in1) {
      bb = gin4[H + hid];
      vv = gin4[hid] - bb;
    } else {
      gi = gin4[(b * T) * H + hid];
    }
  }
  // S2 of this wave's eight chain rows from the state in the lanes (hv: the bf16 bit pattern of this lane's unit)
  const int gsrc = 4 * (8 * (lane & 7) + 2 * q);                          // byte address of lane 8 c' + 2 q for ds_bpermute
  const bool bvalid = c < 8;
  auto s2_from_lanes = [&](unsigned hv) {
    // every lane first pairs its unit with its right neighbour's (DPP row_shl:1 — units 2e, 2e + 1 of a row are lanes 8c + 2e,
    // 8c + 2e + 1: the same 16-lane row); lane (c, q) of the B operand then fetches the pair (2q, 2q + 1) of chain row c
    const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hv, 0x101, 0xF, 0xF, false);      // lane i <- lane i + 1
    const unsigned pair = (hv & 0xFFFFu) | (nb << 16);
    const unsigned g = (unsigned)__builtin_amdgcn_ds_bpermute(gsrc, (int)pair);
    const xbf8 bfrag = __builtin_bit_cast(xbf8, u32x4{bvalid ? g : 0u, 0u, 0u, 0u});
    // all six products first, ONE guarded block of stores behind them (a guard per tile put a branch — and the wait for its
    // MFMA — between every two of them)
    f32x4 acc[F::MT2];
#pragma unroll
    for (int mt = 0; mt < F::MT2; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[mt], bfrag, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    if (c < 8) {                         // column c = chain row 8 wave + c; registers j: features 16mt + 4q + j
#pragma unroll
      for (int mt = 0; mt < F::MT2; ++mt) {
        const int m0 = 16 * mt + 4 * q;
        const int i = m0 / F::R2, a0 = m0 % F::R2;
        xbf4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)acc[mt][j];
        *reinterpret_cast<xbf4*>(img + x_off<F::K>(i, F::kperm(8 * wave + c, a0))) = o;
      }
    }
  };
  s2_from_lanes((unsigned)f32_to_bf16(hst).v);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): keep weight-register waits out of the time loop
  lds_barrier();

  const int row10 = c < F::I2 ? c : F::I2 - 1;
  for (int t = 0; t < T; ++t) {
    // ---- B: fused S1*S0 stage, one tile per wave --------------------------------------------------------------------
    {
      xbf8 af[F::NM];
#pragma unroll
      for (int u = 0; u < F::NM; ++u) af[u] = *reinterpret_cast<const xbf8*>(img + x_off<F::K>(row10, 32 * u + 8 * q));
      __builtin_amdgcn_sched_barrier(0);      // all eight reads in flight before the first MFMA (the scheduler issued them in pairs)
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int u = 0; u < F::NM; u += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10[u], af[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10[u + 1], af[u + 1], acc1, 0, 0, 0);
      }
      const f32x4 acc = acc0 + acc1;
      if (c < F::I2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) gbuf[(16 * wave + 4 * q + j) * F::I2 + c] = acc[j];   // o = m*I2 + i2
      }
    }
    lds_barrier();
    // ---- C + A: gates + state (gru.py:38-44), then S2 of the new state from the lanes -------------------------------
    const size_t bt = b * T + t;
    {
      if (in1) gi = bb + xq.at(t) * vv;
      const float hn = gbuf[2 * H + hid] + bh[2];
      const float rg = fsigmoid(gi[0] + gbuf[hid] + bh[0]);              // gru.py:38-39
      const float zg = fsigmoid(gi[1] + gbuf[H + hid] + bh[1]);          // gru.py:40-41
      const float ng = ftanh(gi[2] + rg * hn);                           // gru.py:42-43
      float hy = (1.0f - zg) * ng + zg * hst;                            // gru.py:44
      if (reserve) *reinterpret_cast<f32x4*>(reserve + (bt * H + hid) * 4) = f32x4{rg, zg, ng, hn};
      const bf16_t hb = f32_to_bf16(hy);            // rounded once: stored, fed back, kept as state
      if constexpr (OUT) out[bt * H + hid] = hb;
      hst = bf16_to_f32(hb);
      s2_from_lanes((unsigned)hb.v);
      // (behind the S2 of this step: a conditional global load makes the compiler wait for every memory operation in flight —
      // this step's stores included — at the point where the branches join)
      if (!in1 && t + 1 < T) gi = gin4[(bt + 1) * H + hid];
    }
    if (in1) xq.advance(xs, b * T, T, t, lane);
    lds_barrier();
  }
  if (hT) st(hT, b * H + hid, hst);
}

template <class S>
static int launch_f10g(const RnnShape& rs, GinSrc gin, const void* h0, const float* packed_hid, const void* bias_hid,
                       void* out, void* hT, float* reserve, void* ws, hipStream_t stream, int phase) {
  if (!ws) return TTRNN_ERR_WORKSPACE;
  xbf8* wfrag = reinterpret_cast<xbf8*>(ws);
  if (phase != TTRNN_PHASE_RUN)
    hipLaunchKernelGGL((k_f10g_prep<S>), dim3(4 * F10<S>::NM), dim3(64), 0, stream, packed_hid, wfrag);
  if (phase == TTRNN_PHASE_PREPARE) return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  if (!(opt(OPT_DEV) & 32)) {    // default: four waves, S2 inside the gate waves (dev bit 5: the eight-wave kernel, A/B)
    auto kern = gin.in1 ? (out ? k_gru_fwd_f10v<S, true, true> : k_gru_fwd_f10v<S, false, true>)
                        : (out ? k_gru_fwd_f10v<S, true, false> : k_gru_fwd_f10v<S, false, false>);
    hipLaunchKernelGGL(kern, dim3(rs.B), dim3(256), 0, stream, rs.B, rs.T, gin, (const bf16_t*)h0, packed_hid,
                       wfrag, rs.has_bias_hid ? (const bf16_t*)bias_hid : (const bf16_t*)nullptr, (bf16_t*)out, (bf16_t*)hT,
                       reserve);
  } else
  hipLaunchKernelGGL((k_gru_fwd_f10<S>), dim3(rs.B), dim3(FAST_NT), 0, stream, rs.B, rs.T, gin, (const bf16_t*)h0,
                     packed_hid, wfrag, rs.has_bias_hid ? (const bf16_t*)bias_hid : (const bf16_t*)nullptr,
                     (bf16_t*)out, (bf16_t*)hT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// ---- dispatch ------------------------------------------------------------------------------------------
template <class S, int KS>
static int launch_f10(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                      const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws,
                      hipStream_t stream, int phase) {
  if (!ws) return TTRNN_ERR_WORKSPACE;
  static_assert(F10H_HDR_BYTES + (size_t)F10<S>::MT * F10<S>::NM * 2 * 64 * sizeof(xh8) <= f10_wfrag_bytes<S>(), "workspace");
  float* hdr = reinterpret_cast<float*>(ws);
  xh8* wfrag = reinterpret_cast<xh8*>(reinterpret_cast<unsigned char*>(ws) + F10H_HDR_BYTES);
  static_assert((F10H_EP + F10<S>::M) * sizeof(int) <= F10H_HDR_BYTES, "header");
  // (h_0 is scaled per sample inside the recurrent kernels: f10h_h0_expo)
  // the scale header and the fragments depend on the weights only: TTRNN_PHASE_RUN finds them in ws (ttrnn_rnn_forward_phase)
  if (phase != TTRNN_PHASE_RUN) {
    hipLaunchKernelGGL((k_f10h_scale<S>), dim3(F10<S>::M), dim3(256), 0, stream, packed_hid, reinterpret_cast<int*>(ws));
    hipLaunchKernelGGL((k_f10h_prep<S>), dim3(F10<S>::MT * F10<S>::NM), dim3(64), 0, stream, packed_hid, hdr, wfrag);
  }
  if (phase == TTRNN_PHASE_PREPARE) return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  constexpr size_t lds = f10h_lds_bytes<S, KS>();
  static_assert(lds <= 64 * 1024, "raise the dynamic LDS limit for this shape");
  const float* bh = rs.has_bias_hid ? (const float*)bias_hid : (const float*)nullptr;
  const bool dg = opt(OPT_DIAG) && reserve && opt(OPT_F10_NB1);      // stamped eight-wave build; the four-wave kernel has its own
  // Four-wave workgroups (ttrnn_fast_f10q.hip; bit-identical results): always for r = 8 — 1 ... 3 % faster than the
  // eight-wave kernel even with one workgroup per CU (fewer waves per barrier, no LDS copy of h for the output store) — and
  // for r = 16 once two workgroups can share a CU (B > #CUs / 2: 5 % at B = 192, 10 % at B = 512; below that the eight-wave
  // k-split kernel is 1 % ahead).  OPT_F10_NB1: eight waves, one sample per workgroup, whatever B (A/B switch);
  // OPT_F10_NB2: two samples per eight-wave workgroup for B > #CUs (round 1's variant, A/B switch).
  const int cus = device_cu_count();
  if (!dg && !opt(OPT_F10_NB1)) {
    if (rs.B > cus && opt(OPT_F10_NB2))
      return launch_rnn_fwd_f10_nb2(rs, gin, h0, c0, packed_hid, ws, bh, out, hT, cT, reserve, stream);
    // one barrier per step, S2 inside the gate waves (ttrnn_fast_f10s.hip, round 5): measured slower, dev bit 512 selects it (A/B)
    if (KS == 1 && f10s_available(rs, h0 != nullptr))
      return launch_rnn_fwd_f10_s(rs, gin, h0, c0, packed_hid, ws, bh, out, hT, cT, reserve, stream);
    if (KS == 1 || 2 * rs.B > cus)
      return launch_rnn_fwd_f10_q(rs, gin, h0, c0, packed_hid, ws, bh, out, hT, cT, reserve, stream);
  }
  auto kern = dg ? (h0 ? k_lstm_fwd_f10<S, KS, true, true> : k_lstm_fwd_f10<S, KS, true, false>)
                 : (h0 ? k_lstm_fwd_f10<S, KS, false, true> : k_lstm_fwd_f10<S, KS, false, false>);
  hipLaunchKernelGGL(kern, dim3(rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T, gin, (const float*)h0,
                     (const float*)c0, packed_hid, hdr, wfrag, bh, (float*)out, (float*)hT, (float*)cT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// bytes of workspace launch_rnn_fwd_f10 needs for the fused core's fragments (0 when the shape has no f10 kernel)
size_t f10_workspace_bytes(const RnnShape& rs, int dtype) {
  if (dtype == TTRNN_BF16 && rs.cell == TTRNN_GRU && shape_matches<ShpH256R8G>(rs.hid_s))
    return f10gq_workspace_bytes();      // >= the eight-wave kernel's 4 * NM fragments
  if (dtype == TTRNN_F32 && rs.cell == TTRNN_GRU) return f10gh_workspace_bytes(rs);      // two fp16 pieces (ttrnn_fast_f10gh.hip)
  if (dtype != TTRNN_F32 || rs.cell != TTRNN_LSTM) return 0;
  if (shape_matches<ShpH256R8L>(rs.hid_s)) return f10_wfrag_bytes<ShpH256R8L>();
  if (shape_matches<ShpH256R16L>(rs.hid_s)) return f10_wfrag_bytes<ShpH256R16L>();
  return f2_workspace_bytes(rs, dtype);      // two-core hidden matrices (ttrnn_fast_f2.hip)
}

bool f10_rnn_fwd_available(const RnnShape& rs, int dtype) {
  if (opt(OPT_NO_F10) || rs.B < 1 || rs.T < 1) return false;
  if (dtype == TTRNN_BF16 && rs.cell == TTRNN_GRU) return shape_matches<ShpH256R8G>(rs.hid_s);
  if (dtype == TTRNN_F32 && rs.cell == TTRNN_GRU) return f10gh_available(rs, dtype) && f10gh_own_plan(rs);      // (split math mode only; r = 16 runs behind the tier's K-in: ttrnn_g2.hip)
  if (dtype != TTRNN_F32 || rs.cell != TTRNN_LSTM) return false;
  return shape_matches<ShpH256R8L>(rs.hid_s) || shape_matches<ShpH256R16L>(rs.hid_s) || f2_rnn_fwd_available(rs, dtype);
}

int launch_rnn_fwd_f10(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                       const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws,
                       hipStream_t stream, int phase) {
  if (rs.cell == TTRNN_GRU) {
    // dev bit 2 (A/B): four waves with the gates in the lanes that hold the sums (ttrnn_fast_f10gq.hip) — built for VERDICT r2
    // item 4 and measured SLOWER than the eight-wave kernel below (0.678 against 0.570 ms on cfg3), so it is not the default
    if ((opt(OPT_DEV) & 4) && f10gq_available(rs, TTRNN_BF16))
      return launch_gru_fwd_f10gq(rs, gin, h0, packed_hid, bias_hid, out, hT, reserve, ws, stream, phase);
    if (shape_matches<ShpH256R8G>(rs.hid_s))
      return launch_f10g<ShpH256R8G>(rs, gin, h0, packed_hid, bias_hid, out, hT, reserve, ws, stream, phase);
    return TTRNN_ERR_UNSUPPORTED;
  }
  if (f2_rnn_fwd_available(rs, TTRNN_F32))      // d = 2: both stages on fp16 pieces, one barrier per step (ttrnn_fast_f2.hip)
    return launch_rnn_fwd_f2(rs, gin, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, ws, stream, phase);
  // r = 8: one wave per tile row (measured 7 % faster than the k-split layout); r = 16: k-split (register budget)
  if (shape_matches<ShpH256R8L>(rs.hid_s))
    return launch_f10<ShpH256R8L, 1>(rs, gin, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, ws, stream, phase);
  if (shape_matches<ShpH256R16L>(rs.hid_s))
    return launch_f10<ShpH256R16L, 2>(rs, gin, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, ws, stream, phase);
  return TTRNN_ERR_UNSUPPORTED;
}

// ---- H = 512, r = 8 (the reference's default benchmark shape, experiments/digit_classification/benchmarking.py:75-83) ----------
// Round 4: the fused-core forward of a shape that otherwise lives on the runtime-shape tier.  K10 = J0 J1 R2 = 512, M = I0 I1 = 128:
// eight S10 tiles = eight waves (k_lstm_fwd_f10q<S, 1, ..., QW = 8>), each with its tile's whole contraction resident (sixteen
// k-blocks x two fp16 pieces = 128 VGPRs), one workgroup per CU.  The caller (ttrnn_g2.hip: fwd_t) supplies gin with BOTH biases
// folded in (k_g2_bias) and this workspace; the reverse-time kernel stays the runtime-shape tier's (same reserve format).
// (round 5: the same route for H = 384, r = 8 — out modes (8, 12, 16), six S10 tiles = six waves: benchmarking.py --hidden_size 384)
size_t f10_h512_workspace_bytes() {
  return f10_wfrag_bytes<ShpH512R8L>() > f10_wfrag_bytes<ShpH384R8L>() ? f10_wfrag_bytes<ShpH512R8L>() : f10_wfrag_bytes<ShpH384R8L>();
}
bool f10_h512_fwd_available(const RnnShape& rs, int dtype) {
  return !opt(OPT_NO_F10) && !(opt(OPT_DEV) & 8192) && rs.B >= 1 && rs.T >= 1 && dtype == TTRNN_F32 && rs.cell == TTRNN_LSTM &&
         rs.hid_blocks <= 1 && rs.in != 1 && opt(OPT_FP32_MATH) == TTRNN_MATH_SPLIT &&
         (shape_matches<ShpH512R8L>(rs.hid_s) || shape_matches<ShpH384R8L>(rs.hid_s));
}
template <class S>
static int launch_h512_t(const RnnShape& rs, const float* gin, const void* h0, const void* c0, const float* packed_hid, void* out,
                         void* hT, void* cT, float* reserve, void* ws, hipStream_t stream) {
  float* hdr = reinterpret_cast<float*>(ws);
  xh8* wfrag = reinterpret_cast<xh8*>(reinterpret_cast<unsigned char*>(ws) + F10H_HDR_BYTES);
  static_assert((F10H_EP + F10<S>::M) * sizeof(int) <= F10H_HDR_BYTES, "header");
  hipLaunchKernelGGL((k_f10h_scale<S>), dim3(F10<S>::M), dim3(256), 0, stream, packed_hid, reinterpret_cast<int*>(ws));
  hipLaunchKernelGGL((k_f10h_prep<S>), dim3(F10<S>::MT * F10<S>::NM), dim3(64), 0, stream, packed_hid, hdr, wfrag);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  GinSrc src{gin, nullptr, 0};
  if (f10s_available(rs, h0 != nullptr)) return launch_rnn_fwd_f10_s(rs, src, h0, c0, packed_hid, ws, nullptr, out, hT, cT, reserve, stream);
  return launch_rnn_fwd_f10_q(rs, src, h0, c0, packed_hid, ws, nullptr, out, hT, cT, reserve, stream);
}
int launch_rnn_fwd_f10_h512(const RnnShape& rs, const float* gin, const void* h0, const void* c0, const float* packed_hid,
                            void* out, void* hT, void* cT, float* reserve, void* ws, hipStream_t stream) {
  if (!ws) return TTRNN_ERR_WORKSPACE;
  if (shape_matches<ShpH384R8L>(rs.hid_s))
    return launch_h512_t<ShpH384R8L>(rs, gin, h0, c0, packed_hid, out, hT, cT, reserve, ws, stream);
  return launch_h512_t<ShpH512R8L>(rs, gin, h0, c0, packed_hid, out, hT, cT, reserve, ws, stream);
}

}  // namespace ttrnn
