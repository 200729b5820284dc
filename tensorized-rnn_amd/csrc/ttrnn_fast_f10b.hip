// ttrnn_fast_f10b.hip — reverse-time TT-LSTM kernel on the fused core (gfx950, fp32 storage, split-bf16 MFMAs).
//
// BPTT of one layer: for t = T-1 .. 0, from the saved gates (i,g,f,o,c) of step t and dh_t = d_out[t] + W_hid^T dg_{t+1}:
//     dg_t  (the four gate pre-activation gradients, lstm.py:26-32 differentiated)      -> d_gates_in/hid[b][t][4H]
//     dh_{t-1} = W_hid^T dg_t
// W_hid^T dg is the transposed TT chain.  With cores 1 and 0 contracted once per launch (ttrnn_f10.h) it is two stages:
//     T01: dC2[(row2,r2)][i2] = sum_m  W10[(row2,r2)][m] * dg[m][i2]              (256 features x 16 columns, K = 64)
//     T2 : dh[row2][j2]       = sum_(i2,r2) G2[j2; i2,r2] * dC2[(row2,r2)][i2]    (8 features x 32 columns,  K = 128)
// both on v_mfma_f32_16x16x32_bf16 with three-way split operands (ttrnn_split.h), like the forward kernel
// (ttrnn_fast_f10.hip).  The stage-wise kernel (ttrnn_fast_bwd.hip: three fp32-MFMA stages, four barriers) stays as the
// TTRNN_MATH_EXACT path.
//
// Per timestep (one 8-wave workgroup per sample):
//     G    waves 0-3   one hidden unit per thread: gate gradients -> fp32 row (for the HBM store) and, split into three
//                      bf16 planes, the [i2][m] operand of T01; the k order inside a row is chosen so that the four
//                      gates of a thread are 4 consecutive k (one 8-byte store per plane)
//     barrier
//     T01  all waves   2 feature tiles per wave (24 MFMAs), result split into the [row2][(i2,r2)] operand of T2; the
//                      fp32 gate-gradient row streams out to HBM in 16-byte pieces meanwhile
//     barrier
//     T2   all waves   wave = (column tile, k-block): 6 MFMAs each, partial dh sums (one slice per k-block) -> LDS
//     barrier          (the next G phase adds the four slices)
//
// Replaces, for one layer: torch autograd through tensorized_rnn/lstm.py:23-32,123-133 and t3nsor/ops.py:78-93.
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
struct F10B {
  using F = F10<S>;
  static constexpr int H = F::H;
  static constexpr int K1 = F::M;                 // T01 contraction: m = (i0,i1)                 (64)
  static constexpr int NM1 = K1 / 32;             // k-blocks of T01                               (2)
  static constexpr int FT = F::K / 16;            // T01 feature tiles (features = (row2,r2))      (16)
  static constexpr int XF = FT / FAST_NW;         // per wave                                      (2)
  static constexpr int K2 = F::I2 * F::R2;        // T2 contraction: (i2,r2)                       (128)
  static constexpr int NM2 = K2 / 32;             // k-blocks of T2 = partial-sum slices           (4)
  static constexpr int CT2 = F::ROWS2 / 16;       // T2 column tiles (columns = row2)              (2)
  static constexpr int XT2 = NM2 * CT2 / FAST_NW; // T2 (column tile, k-block) pairs per wave     (1; r = 16: 2)
  static constexpr int PL1 = F::I2 * K1;          // bf16 elements per plane of the T01 operand [I2][K1]
  static constexpr int PL2 = F::ROWS2 * K2;       // bf16 elements per plane of the T2 operand [ROWS2][K2]
  // k order of the T01 operand: thread hid holds gates g = 0..3 of m = MPG*g + hid/I2 -> 4 consecutive k
  __device__ static constexpr int k1_of_m(int m) { return (m % F::MPG) * 4 + m / F::MPG; }
  __device__ static constexpr int m_of_k1(int k) { return (k & 3) * F::MPG + (k >> 2); }
  // k order of the T2 operand: two i2 per 16-byte slot, slots of one r2-quad contiguous (conflict-free stores from
  // the T01 accumulators, same idea as F10::kperm)
  static constexpr int HI = F::I2 / 2;
  __device__ static constexpr int k2_of(int i2, int r2) { return ((r2 >> 2) * HI + (i2 >> 1)) * 8 + (i2 & 1) * 4 + (r2 & 3); }
};

template <class S>
constexpr bool f10b_ok() {
  using F = F10<S>;
  using B = F10B<S>;
  return f10_ok<S>() && F::I2 == 16 && B::K1 % 32 == 0 && B::FT % FAST_NW == 0 && B::K2 % 32 == 0 &&
         (B::NM2 * B::CT2) % FAST_NW == 0 && B::CT2 == 2 && F::J2 == 8 && F::H == 256;
}

template <class S>
constexpr size_t f10b_wfrag_elems() {      // xbf8 fragments: T01 [FT][NM1][3][64] then T2 [NM2][3][64]
  return (size_t)(F10B<S>::FT * F10B<S>::NM1 + F10B<S>::NM2) * 3 * 64;
}

// fragment order: T01: wf[((ft*NM1 + u)*3 + p)*64 + lane], lane (r, q): feature 16ft + r = row2*R2 + r2, k = 32u + 8q + e
//                 T2 : wf[T01 part + (u*3 + p)*64 + lane],  lane (r, q): feature j2 = r (< J2, else 0)
template <class S>
__global__ void __launch_bounds__(64) k_f10b_prep(const float* __restrict__ packed, xbf8* __restrict__ wfrag,
                                                  float* __restrict__ zero = nullptr, int zero_n = 0) {
  using F = F10<S>;
  using B = F10B<S>;
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  // (a caller's accumulator — the weight-gradient kernel's dW10 — is cleared here instead of by a launch of its own)
  if (zero)
    for (int e = blockIdx.x * 64 + lane; e < zero_n; e += gridDim.x * 64) zero[e] = 0.f;
  xbf8 f0, f1, f2;
  if (blockIdx.x < B::FT * B::NM1) {
    const int u = blockIdx.x % B::NM1, ft = blockIdx.x / B::NM1;
    const int f = 16 * ft + r;
    const int row2 = f / F::R2, r2 = f % F::R2;
    const int j1 = row2 % F::J1, j0 = row2 / F::J1;
    const float* W0 = packed + woff_of<S>(0);               // [J0*R1][I0]
    const float* W1 = packed + woff_of<S>(1);               // [J1*R2][I1*R1]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // LSTM shapes: gate-interleaved k order (m_of_k1); other cells: natural order (ttrnn_fast_f10w.hip, NATK)
      const int m = (f10_ok<S>() || f10_in_ok<S>()) ? B::m_of_k1(32 * u + 8 * q + e) : 32 * u + 8 * q + e;
      const int i0 = m / F::I1, i1 = m % F::I1;
      const float* w1p = W1 + (j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1;
      float v = 0.f;
      for (int r1 = 0; r1 < F::R1; ++r1) v = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0], w1p[r1], v);
      __bf16 p0, p1, p2;
      split3(v, p0, p1, p2);
      f0[e] = p0; f1[e] = p1; f2[e] = p2;
    }
    xbf8* dst = wfrag + (size_t)(blockIdx.x * 3) * 64 + lane;
    dst[0] = f0; dst[64] = f1; dst[128] = f2;
  } else {
    const int u = blockIdx.x - B::FT * B::NM1;
    const float* W2 = packed + woff_of<S>(2);               // [J2][M2 = I2*R2]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int slot = 4 * u + q;
      const int i2 = 2 * (slot % B::HI) + (e >> 2), r2 = (slot / B::HI) * 4 + (e & 3);
      const float v = (r < F::J2 && i2 < F::I2) ? W2[r * F::M2 + i2 * F::R2 + r2] : 0.f;
      __bf16 p0, p1, p2;
      split3(v, p0, p1, p2);
      f0[e] = p0; f1[e] = p1; f2[e] = p2;
    }
    xbf8* dst = wfrag + (size_t)(B::FT * B::NM1 * 3 + u * 3) * 64 + lane;
    dst[0] = f0; dst[64] = f1; dst[128] = f2;
  }
}

template <class S>
constexpr size_t f10b_lds_bytes() {
  using B = F10B<S>;
  return sizeof(float) * 4 * B::H                     // dgf: fp32 gate-gradient row
         + sizeof(float) * B::NM2 * B::H              // dhs: partial dh slices
         + 2 * 3 * (size_t)(B::PL1 + B::PL2);         // the two split operands
}

template <class S, bool DIAG>
__global__ void __launch_bounds__(FAST_NT) k_lstm_bwd_f10(int Bn, int T, const float* __restrict__ c0,
                                                          const xbf8* __restrict__ wfrag,
                                                          const float* __restrict__ reserve,
                                                          const float* __restrict__ d_out,
                                                          const float* __restrict__ d_hT,
                                                          const float* __restrict__ d_cT, float* __restrict__ dg_in,
                                                          float* __restrict__ dg_hid, float* __restrict__ d_h0,
                                                          float* __restrict__ d_c0,
                                                          unsigned long long* __restrict__ diag, BwdStats bs) {
  static_assert(f10b_ok<S>(), "shape not supported by the fused-core reverse-time kernel");
  using F = F10<S>;
  using B = F10B<S>;
  constexpr int H = F::H, GH = 4 * H;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* dgf = reinterpret_cast<float*>(smem);                               // [4H] in HBM row order
  float* dhs = dgf + GH;                                                     // [NM2][H]
  __bf16* img1 = reinterpret_cast<__bf16*>(dhs + B::NM2 * H);                 // 3 planes [I2][K1]
  __bf16* img2 = img1 + 3 * B::PL1;                                          // 3 planes [ROWS2][K2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  // resident fragments: T01 for the feature tiles {wave + 8x}; T2 for the pairs id = wave + 8x: column tile id & 1,
  // k-block id >> 1
  xbf8 w01[B::XF][B::NM1][3], w2t[B::XT2][3];
#pragma unroll
  for (int x = 0; x < B::XF; ++x)
#pragma unroll
    for (int u = 0; u < B::NM1; ++u)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        w01[x][u][p] = wfrag[(size_t)(((wave + FAST_NW * x) * B::NM1 + u) * 3 + p) * 64 + lane];
  const int ct = wave & 1;
#pragma unroll
  for (int x = 0; x < B::XT2; ++x)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      w2t[x][p] = wfrag[(size_t)(B::FT * B::NM1 * 3 + ((wave + FAST_NW * x) >> 1) * 3 + p) * 64 + lane];

  // gate phase: thread tid < H owns hidden unit tid.  record(t) = (i,g,f,o),(c,-,-,-); step t also needs c_{t-1} =
  // record(t-1).c.  Three register sets rotate their ROLES (current / next / in flight) from step to step — the time
  // loop is unrolled by three so that no register copy is needed: iteration t issues the loads of record(t-2) and
  // d_out(t-2) into the set that held record(t+1) and nobody touches them before iteration t-1.  (Consuming a value, or
  // merely copying it to another register, in the phase that issued its load put one HBM round trip per timestep on
  // the critical path: 1 800 of 4 000 cycles.  So did a conditional load of c0 inside the loop.)
  const bool own = tid < H;
  const int hid = own ? tid : 0;
  float dcs = (own && d_cT) ? d_cT[b * H + hid] : 0.f;
  const float c0v = (own && c0) ? c0[b * H + hid] : 0.f;
  const float* dptr = d_out ? d_out : reserve;
  const float dscale = d_out ? 1.0f : 0.0f;
  f32x4 ra0 = f32x4{0.f, 0.f, 0.f, 0.f}, ra1 = ra0, ra2 = ra0;      // (i,g,f,o) of the three sets
  float rb0 = 0.f, rb1 = 0.f, rb2 = 0.f;                               // c of the three sets (a dwordx4 load with dead
                                                                       // lanes gets its registers reused -> WAW wait)
  float do0 = 0.f, do1 = 0.f, do2 = 0.f;
  // by-products for the weight-gradient step (BwdStats, ttrnn_launch.h): column maxima of the gate gradients and, for
  // input_size == 1, this sample's sums of x_t dg_t and dg_t — 12 VALU operations per step in registers the G phase owns.
  // x_t rides in the rotating sets (a null x reads the reserve and is scaled by zero, as d_out)
  const float* xptr = bs.x ? reinterpret_cast<const float*>(bs.x) : reserve;
  const float xscale = bs.x ? 1.0f : 0.0f;
  float xq0 = 0.f, xq1 = 0.f, xq2 = 0.f;
  f32x4 cmx = f32x4{0.f, 0.f, 0.f, 0.f}, sxd = cmx, sdg = cmx;
  if (own) {
    dhs[hid] = d_hT ? d_hT[b * H + hid] : 0.f;
#pragma unroll
    for (int sl = 1; sl < B::NM2; ++sl) dhs[sl * H + hid] = 0.f;
    if (T > 0) {                         // set 0 = record(T-1), set 1 = record(T-2)
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

  // one timestep; (ra, rb, dout_c) = record / d_out of step t, nb = record(t-1) (its c), (fa, fb, dout_f): in-flight set
  auto step = [&](const int t, const f32x4& ra, const float& rb, const float& dout_c, const float& nb, f32x4& fa,
                  float& fb, float& dout_f, const float& x_c, float& x_f) {
    const size_t bt = b * T + t;
    // ---- G: gate gradients (lstm.py:26-32 differentiated) ---------------------------------------------------------
    if (own) {
      {                                  // record(t-2), d_out(t-2): not touched before the next iteration.  Always
        // three loads, no branch (index clamped, a null d_out reads the reserve and is scaled by zero): a conditional
        // VMEM operation makes the compiler's vmcnt bookkeeping fall back to vmcnt(0)
        const size_t b2 = t > 1 ? bt - 2 : b * T;
        fa = *reinterpret_cast<const f32x4*>(reserve + res_gate(b2, H, hid));
        fb = reserve[res_cell((size_t)Bn * T, b2, H, hid)];
        dout_f = dptr[b2 * H + hid];                 // scaled where it is consumed
        x_f = xptr[b2];
      }
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
      {
        const f32x4 pv = f32x4{p0, p1, p2, p3};
        const float xv = x_c * xscale;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          cmx[g] = fmaxf(cmx[g], fabsf(pv[g]));
          sxd[g] = fmaf(xv, pv[g], sxd[g]);
          sdg[g] += pv[g];
        }
      }
      dgf[hid] = p0; dgf[H + hid] = p1; dgf[2 * H + hid] = p2; dgf[3 * H + hid] = p3;
      // o = gate*H + hid = m*I2 + i2  ->  m = MPG*gate + hid/I2, i2 = hid%I2: the 4 gates are k = 4*(hid/I2) .. +3
      store_split4(img1, B::PL1, x_off<B::K1>(hid % F::I2, 4 * (hid / F::I2)), f32x4{p0, p1, p2, p3});
    }
    TT_STAMP(0)
    lds_barrier();
    TT_STAMP(1)
    // ---- T01: dC2 = W10 dg, two feature tiles per wave; meanwhile the fp32 row goes out to HBM -------------------
    {
      if (tid < GH / 4) {
        const f32x4 v = reinterpret_cast<const f32x4*>(dgf)[tid];
        reinterpret_cast<f32x4*>(dg_in + bt * GH)[tid] = v;
        if (dg_hid && dg_hid != dg_in) reinterpret_cast<f32x4*>(dg_hid + bt * GH)[tid] = v;
      }
      xbf8 bf[B::NM1][3];
#pragma unroll
      for (int u = 0; u < B::NM1; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bf[u][p] = *reinterpret_cast<const xbf8*>(img1 + p * B::PL1 + x_off<B::K1>(c, 32 * u + 8 * q));
#pragma unroll
      for (int x = 0; x < B::XF; ++x) {
        f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
#pragma unroll
        for (int u = 0; u < B::NM1; ++u) {
#pragma unroll
          for (int s = 0; s < 5; ++s)
            acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w01[x][u][SPLIT_TW[s]], bf[u][SPLIT_TX[s]], acc_lo, 0, 0, 0);
          acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w01[x][u][0], bf[u][0], acc_hi, 0, 0, 0);
        }
        const f32x4 acc = acc_hi + acc_lo;
        // lane (c = i2, q), registers j: features 16ft + 4q + j = (row2, r2 = 4*(q&1) + j)
        const int f0 = 16 * (wave + FAST_NW * x) + 4 * q;
        const int row2 = f0 / F::R2, r20 = f0 % F::R2;
        store_split4(img2, B::PL2, x_off<B::K2>(row2, B::k2_of(c, r20)), acc);
      }
    }
    TT_STAMP(2)
    lds_barrier();
    TT_STAMP(3)
    // ---- T2: dh_{t-1}[row2][j2], pair = (column tile ct, k-block ub): one partial-sum slice per k-block -------------
#pragma unroll
    for (int x = 0; x < B::XT2; ++x) {
      const int ub = (wave + FAST_NW * x) >> 1;
      const int row = 16 * ct + c;
      xbf8 bf[3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
        bf[p] = *reinterpret_cast<const xbf8*>(img2 + p * B::PL2 + x_off<B::K2>(row, 32 * ub + 8 * q));
      f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
#pragma unroll
      for (int s = 0; s < 5; ++s)
        acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2t[x][SPLIT_TW[s]], bf[SPLIT_TX[s]], acc_lo, 0, 0, 0);
      acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2t[x][0], bf[0], acc_hi, 0, 0, 0);
      const f32x4 acc = acc_hi + acc_lo;
      // lane (c = row2 in the column tile, q), registers j: j2 = 4q + j (valid for q < 2): hidden = row2*J2 + j2
      if (q < 2) *reinterpret_cast<f32x4*>(dhs + ub * H + row * F::J2 + 4 * q) = acc;
    }
    TT_STAMP(4)
    lds_barrier();
    TT_STAMP(5)
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

// ---- GRU ---------------------------------------------------------------------------------------------------------
// Same two transposed stages for the TT-GRU (fp32 or bf16 storage; reserve, gate gradients and all arithmetic fp32):
// the gate gradients of gru.py:38-44 are dr, dz, dn*r for the hidden chain (dn for the input chain), and the flat
// gate index o = gate*H + hid = m*I2 + i2 does not align with the MFMA tiles (I2 = 12), so the T01 operand is filled
// element by element in the natural k order (the fragments of k_f10b_prep follow, see f10_ok there).  The direct
// path dh_{t-1} += dh_t * z stays in the thread's register.
template <class S>
constexpr bool f10b_gru_ok() {
  using F = F10<S>;
  using B = F10B<S>;
  return S::D == 3 && S::R[0] == 1 && S::R[3] == 1 && shape_ok_recurrent<S>() && F::H == 256 &&
         out_size_of<S>() == 3 * F::H && F::I2 % 2 == 0 && F::I2 <= 16 && F::M == 64 && B::K1 % 32 == 0 &&
         B::FT % FAST_NW == 0 && B::K2 % 32 == 0 && B::CT2 == 2 && F::J2 == 8 && S::R[2] % 4 == 0;
}

template <class S>
constexpr size_t f10b_gru_lds_bytes() {
  using B = F10B<S>;
  return sizeof(float) * 3 * B::H + sizeof(float) * B::NM2 * B::H + 2 * 3 * (size_t)(B::PL1 + B::PL2);
}

template <class S, typename TS>
__global__ void __launch_bounds__(FAST_NT) k_gru_bwd_f10(int Bn, int T, const TS* __restrict__ out,
                                                         const TS* __restrict__ h0, const xbf8* __restrict__ wfrag,
                                                         const float* __restrict__ reserve,
                                                         const TS* __restrict__ d_out, const TS* __restrict__ d_hT,
                                                         float* __restrict__ dg_in, float* __restrict__ dg_hid,
                                                         TS* __restrict__ d_h0, BwdStats bs) {
  static_assert(f10b_gru_ok<S>(), "shape not supported by the fused-core GRU reverse-time kernel");
  using F = F10<S>;
  using B = F10B<S>;
  constexpr int H = F::H, GH = 3 * H;
  constexpr int NP = B::NM2 * B::CT2, XT = (NP + FAST_NW - 1) / FAST_NW;      // T2 (column tile, k-block) pairs

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* dgf = reinterpret_cast<float*>(smem);                               // [3H]: dr, dz, dn*r (hidden chain)
  float* dhs = dgf + GH;                                                     // [NM2][H]
  __bf16* img1 = reinterpret_cast<__bf16*>(dhs + B::NM2 * H);                 // 3 planes [I2][K1], natural k
  __bf16* img2 = img1 + 3 * B::PL1;                                          // 3 planes [ROWS2][K2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  xbf8 w01[B::XF][B::NM1][3], w2t[XT][3];
#pragma unroll
  for (int x = 0; x < B::XF; ++x)
#pragma unroll
    for (int u = 0; u < B::NM1; ++u)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        w01[x][u][p] = wfrag[(size_t)(((wave + FAST_NW * x) * B::NM1 + u) * 3 + p) * 64 + lane];
#pragma unroll
  for (int x = 0; x < XT; ++x) {
    const int id = wave + FAST_NW * x, ub = (id < NP ? id : 0) >> 1;
#pragma unroll
    for (int p = 0; p < 3; ++p) w2t[x][p] = wfrag[(size_t)(B::FT * B::NM1 * 3 + ub * 3 + p) * 64 + lane];
  }

  // three rotating sets of (record(t) = (r,z,n,hn), d_out(t), h_{t-1}) — see k_lstm_bwd_f10
  const bool own = tid < H;
  const int hid = own ? tid : 0;
  float dhd = 0.f;
  const TS* dptr = d_out ? d_out : out;
  const float dscale = d_out ? 1.0f : 0.0f;
  f32x4 ra0 = f32x4{0.f, 0.f, 0.f, 0.f}, ra1 = ra0, ra2 = ra0;
  TS do0 = TS{}, do1 = TS{}, do2 = TS{}, hp0 = TS{}, hp1 = TS{}, hp2 = TS{};
  // by-products (BwdStats; see k_lstm_bwd_f10): column maxima of both gradient rows — they differ in the n gate — and the
  // sample's sums of x_t dg_in_t and dg_in_t; a null x reads `out` and is scaled by zero
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
      const float pv[3] = {dr_pre, dz_pre, dn_pre * rg};
      {
        const float pin[3] = {dr_pre, dz_pre, dn_pre};
        const float xv = to_f32(xq) * xscale;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          cmi[g] = fmaxf(cmi[g], fabsf(pin[g]));
          sxd[g] = fmaf(xv, pin[g], sxd[g]);
          sdg[g] += pin[g];
        }
        cmh = fmaxf(cmh, fabsf(pv[2]));
      }
      dg_in[bt * GH + 2 * H + hid] = dn_pre;                // the only block where d_gates_in != d_gates_hid
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        dgf[g * H + hid] = pv[g];
        const int o = g * H + hid, m = o / F::I2, i2 = o % F::I2;     // flat gate index -> (m, i2), natural k = m
        __bf16 p0, p1, p2;
        split3(pv[g], p0, p1, p2);
        const int off = x_off<B::K1>(i2, m);
        img1[off] = p0; img1[B::PL1 + off] = p1; img1[2 * B::PL1 + off] = p2;
      }
    }
    lds_barrier();
    // ---- T01: dC2 = W10 dg; the fp32 rows go out to HBM meanwhile ----------------------------------------------------
    {
      if (tid < GH / 4) {
        const f32x4 v = reinterpret_cast<const f32x4*>(dgf)[tid];
        reinterpret_cast<f32x4*>(dg_hid + bt * GH)[tid] = v;
        if (tid < 2 * H / 4) reinterpret_cast<f32x4*>(dg_in + bt * GH)[tid] = v;
      }
      const int rowc = c < F::I2 ? c : F::I2 - 1;
      xbf8 bf[B::NM1][3];
#pragma unroll
      for (int u = 0; u < B::NM1; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bf[u][p] = *reinterpret_cast<const xbf8*>(img1 + p * B::PL1 + x_off<B::K1>(rowc, 32 * u + 8 * q));
#pragma unroll
      for (int x = 0; x < B::XF; ++x) {
        f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
#pragma unroll
        for (int u = 0; u < B::NM1; ++u) {
#pragma unroll
          for (int s = 0; s < 5; ++s)
            acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w01[x][u][SPLIT_TW[s]], bf[u][SPLIT_TX[s]], acc_lo, 0, 0, 0);
          acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w01[x][u][0], bf[u][0], acc_hi, 0, 0, 0);
        }
        const f32x4 acc = acc_hi + acc_lo;
        const int f0 = 16 * (wave + FAST_NW * x) + 4 * q;
        const int row2 = f0 / F::R2, r20 = f0 % F::R2;
        if (c < F::I2) store_split4(img2, B::PL2, x_off<B::K2>(row2, B::k2_of(c, r20)), acc);
      }
    }
    lds_barrier();
    // ---- T2: dh_{t-1}[row2][j2]: pair id = (column tile id & 1, k-block id >> 1), one slice per k-block ---------------
#pragma unroll
    for (int x = 0; x < XT; ++x) {
      const int id = wave + FAST_NW * x;
      if (id < NP) {
        const int ub = id >> 1, row = 16 * (id & 1) + c;
        xbf8 bf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bf[p] = *reinterpret_cast<const xbf8*>(img2 + p * B::PL2 + x_off<B::K2>(row, 32 * ub + 8 * q));
        f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
#pragma unroll
        for (int s = 0; s < 5; ++s)
          acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2t[x][SPLIT_TW[s]], bf[SPLIT_TX[s]], acc_lo, 0, 0, 0);
        acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2t[x][0], bf[0], acc_hi, 0, 0, 0);
        const f32x4 acc = acc_hi + acc_lo;
        if (q < 2) *reinterpret_cast<f32x4*>(dhs + ub * H + row * F::J2 + 4 * q) = acc;
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

template <class S, typename TS>
static int launch_gru_bwd_f10(const RnnShape& rs, const void* out, const void* h0, const float* packed_hid,
                              const float* reserve, const void* d_out, const void* d_hT, float* dg_in, float* dg_hid,
                              void* d_h0, void* ws, hipStream_t stream, const BwdStats& bs) {
  if (!ws) return TTRNN_ERR_WORKSPACE;
  using B = F10B<S>;
  xbf8* wfrag = reinterpret_cast<xbf8*>(ws);
  hipLaunchKernelGGL((k_f10b_prep<S>), dim3(B::FT * B::NM1 + B::NM2), dim3(64), 0, stream, packed_hid, wfrag);
  constexpr size_t lds = f10b_gru_lds_bytes<S>();
  static_assert(lds <= 64 * 1024, "raise the dynamic LDS limit for this shape");
  hipLaunchKernelGGL((k_gru_bwd_f10<S, TS>), dim3(rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T, (const TS*)out,
                     (const TS*)h0, wfrag, reserve, (const TS*)d_out, (const TS*)d_hT, dg_in, dg_hid, (TS*)d_h0, bs);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// ---- dispatch ------------------------------------------------------------------------------------------
template <class S>
static int launch_bwd_f10(const RnnShape& rs, const void* c0, const float* packed_hid, const float* reserve,
                          const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid,
                          void* d_h0, void* d_c0, void* ws, hipStream_t stream, const BwdStats& bs, void* ws_half) {
  if (!ws) return TTRNN_ERR_WORKSPACE;
  using B = F10B<S>;
  // split mode: the two-piece fp16 kernel (ttrnn_fast_f10bh.hip); option gemm_pieces = 3 keeps this file's kernel
  if (ws_half)
    return launch_lstm_bwd_f10h(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, dg_hid, d_h0, d_c0, ws_half,
                                reinterpret_cast<unsigned long long*>((char*)ws + f10b_wfrag_elems<S>() * sizeof(xbf8)),
                                stream, bs);
  xbf8* wfrag = reinterpret_cast<xbf8*>(ws);
  hipLaunchKernelGGL((k_f10b_prep<S>), dim3(B::FT * B::NM1 + B::NM2), dim3(64), 0, stream, packed_hid, wfrag);
  constexpr size_t lds = f10b_lds_bytes<S>();
  static_assert(lds <= 160 * 1024, "LDS image set too large");
  if (lds > 64 * 1024) {
    {
      if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_lstm_bwd_f10<S, false>), lds) != TTRNN_OK ||
          ensure_dynamic_lds(reinterpret_cast<const void*>(k_lstm_bwd_f10<S, true>), lds) != TTRNN_OK)
        return TTRNN_ERR_LAUNCH;
    }
  }
  // TTRNN_DIAG=1: per-phase s_memtime stamps of the first 8 workgroups land in the 4 KB behind the fragments
  const bool dg = opt(OPT_DIAG) != 0;
  auto kern = dg ? k_lstm_bwd_f10<S, true> : k_lstm_bwd_f10<S, false>;
  unsigned long long* diag =
      dg ? reinterpret_cast<unsigned long long*>((char*)ws + f10b_wfrag_elems<S>() * sizeof(xbf8)) : nullptr;
  hipLaunchKernelGGL(kern, dim3(rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T, (const float*)c0, wfrag, reserve,
                     (const float*)d_out, (const float*)d_hT, (const float*)d_cT, dg_in, dg_hid, (float*)d_h0,
                     (float*)d_c0, diag, bs);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// the fragment set alone (shared with the weight-gradient kernel, ttrnn_fast_f10w.hip)
size_t f10b_fragment_bytes(const TtShape& s) {
  if (shape_matches<ShpH256R8L>(s)) return f10b_wfrag_elems<ShpH256R8L>() * sizeof(xbf8);
  if (shape_matches<ShpH256R16L>(s)) return f10b_wfrag_elems<ShpH256R16L>() * sizeof(xbf8);
  if (shape_matches<ShpH256R8G>(s)) return f10b_wfrag_elems<ShpH256R8G>() * sizeof(xbf8);
  if (shape_matches<ShpI40R16L>(s)) return f10b_wfrag_elems<ShpI40R16L>() * sizeof(xbf8);
  return 0;
}

template <class S>
static int launch_prep_b(const float* packed, void* wfrag, hipStream_t stream, float* zero, int zero_n) {
  using B = F10B<S>;
  hipLaunchKernelGGL((k_f10b_prep<S>), dim3(B::FT * B::NM1 + B::NM2), dim3(64), 0, stream, packed,
                     reinterpret_cast<xbf8*>(wfrag), zero, zero_n);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

int launch_f10b_prep(const TtShape& s, const float* packed, void* wfrag, hipStream_t stream, float* zero, int zero_n) {
  if (shape_matches<ShpH256R8L>(s)) return launch_prep_b<ShpH256R8L>(packed, wfrag, stream, zero, zero_n);
  if (shape_matches<ShpH256R16L>(s)) return launch_prep_b<ShpH256R16L>(packed, wfrag, stream, zero, zero_n);
  if (shape_matches<ShpH256R8G>(s)) return launch_prep_b<ShpH256R8G>(packed, wfrag, stream, zero, zero_n);
  if (shape_matches<ShpI40R16L>(s)) return launch_prep_b<ShpI40R16L>(packed, wfrag, stream, zero, zero_n);
  return TTRNN_ERR_UNSUPPORTED;
}

bool f10_rnn_bwd_available(const RnnShape& rs, int dtype) {
  if (opt(OPT_NO_F10) || rs.B < 1 || rs.T < 1) return false;
  if (rs.cell == TTRNN_GRU) {
    if (shape_matches<ShpH256R8G>(rs.hid_s)) return dtype == TTRNN_F32 || dtype == TTRNN_BF16;
    // rank 16 (round 6): only the two-fp16-piece kernel has an instantiation (twelve T2 pairs: two per wave on waves 0-3)
    return shape_matches<ShpH256R16G>(rs.hid_s) && f10bh_available(rs, dtype);
  }
  if (dtype != TTRNN_F32 || rs.cell != TTRNN_LSTM) return false;
  return shape_matches<ShpH256R8L>(rs.hid_s) || shape_matches<ShpH256R16L>(rs.hid_s);
}

// fragments + 4 KB for diagnostic stamps (+ the per-sample sums of the by-products, input_size == 1)
static size_t f10b_ws_head_bytes(const RnnShape& rs, int dtype) {
  if (rs.cell == TTRNN_GRU && shape_matches<ShpH256R8G>(rs.hid_s))
    return f10b_wfrag_elems<ShpH256R8G>() * sizeof(xbf8) + 4096;
  if (rs.cell == TTRNN_GRU && shape_matches<ShpH256R16G>(rs.hid_s)) return 4096;      // (no three-piece kernel: stamps only)
  if (dtype != TTRNN_F32 || rs.cell != TTRNN_LSTM) return 0;
  if (shape_matches<ShpH256R8L>(rs.hid_s)) return f10b_wfrag_elems<ShpH256R8L>() * sizeof(xbf8) + 4096;
  if (shape_matches<ShpH256R16L>(rs.hid_s)) return f10b_wfrag_elems<ShpH256R16L>() * sizeof(xbf8) + 4096;
  return 0;
}
size_t bwd_stats_part_bytes(const RnnShape& rs) {
  return rs.in == 1 ? ((size_t)rs.B * 2 * rs.G * rs.H * sizeof(float) + 255) & ~(size_t)255 : 0;
}
size_t f10_rnn_bwd_workspace_bytes(const RnnShape& rs, int dtype) {      // [fragments, stamps | sums | two-piece kernel's]
  const size_t head = f10b_ws_head_bytes(rs, dtype);
  if (!head) return 0;
  const size_t half = f10bh_workspace_bytes(rs);
  return ((head + 255) & ~(size_t)255) + bwd_stats_part_bytes(rs) + half;
}

// per-sample sums -> stats rows 2, 3 (fixed order: sixteen interleaved sample groups per column — 16 columns x 16 groups per
// workgroup — then the sixteen partial sums pairwise); LSTM: d_gates_hid IS d_gates_in, row 1 = row 0
__global__ void __launch_bounds__(256) k_bwd_stats_finish(int lstm, int Bn, int GH, const float* __restrict__ part,
                                                          float* __restrict__ stats) {
  __shared__ float red[2][16][16];
  const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  if (part) {
    float a = 0.f, d = 0.f;
    if (c < GH)
      for (int b = grp; b < Bn; b += 16) {
        a += part[((size_t)b * 2) * GH + c];
        d += part[((size_t)b * 2 + 1) * GH + c];
      }
    red[0][grp][cl] = a;
    red[1][grp][cl] = d;
    __syncthreads();
    if (grp < 2 && c < GH) {
      float t[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) t[g] = red[grp][g][cl];
#pragma unroll
      for (int w = 8; w > 0; w >>= 1)
#pragma unroll
        for (int g = 0; g < w; ++g) t[g] = t[2 * g] + t[2 * g + 1];
      stats[(2 + grp) * GH + c] = t[0];
    }
  }
  if (lstm && grp == 0 && c < GH) stats[GH + c] = stats[c];
}

int launch_bwd_stats_finish(int cell, int Bn, int GH, const float* part, float* stats, hipStream_t stream) {
  hipLaunchKernelGGL(k_bwd_stats_finish, dim3((GH + 15) / 16), dim3(256), 0, stream, cell == TTRNN_LSTM ? 1 : 0, Bn, GH, part,
                     stats);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

int launch_rnn_bwd_f10(const RnnShape& rs, int dtype, const void* out, const void* h0, const void* c0,
                       const float* packed_hid, const float* reserve, const void* d_out, const void* d_hT,
                       const void* d_cT, float* dg_in, float* dg_hid, void* d_h0, void* d_c0, void* ws,
                       hipStream_t stream, const void* x_in1, float* stats) {
  // by-products (stats != NULL): the maxima land in stats rows 0 / 1 by atomicMax, the sums per sample behind the fragments
  BwdStats bs;
  const int GH = rs.G * rs.H;
  void* ws_half = nullptr;
  if (f10bh_available(rs, dtype))
    ws_half = (char*)ws + ((f10b_ws_head_bytes(rs, dtype) + 255) & ~(size_t)255) + bwd_stats_part_bytes(rs);
  if (stats) {
    // (the two-piece kernels' prep launch clears the maxima itself)
    if (!ws_half && hipMemsetAsync(stats, 0, (size_t)2 * GH * sizeof(float), stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
    bs.colmax = reinterpret_cast<unsigned*>(stats);
    if (ws_half && rs.cell == TTRNN_LSTM) bs.rowmax = stats + (size_t)TTRNN_BWD_STATS_ROWS * GH;      // TTRNN_BWD_STATS_ROWMAX
    if (x_in1 && rs.in == 1) {
      bs.x = x_in1;
      bs.part = reinterpret_cast<float*>((char*)ws + ((f10b_ws_head_bytes(rs, dtype) + 255) & ~(size_t)255));
    }
  }
  int st = TTRNN_ERR_UNSUPPORTED;
  if (rs.cell == TTRNN_GRU) {
    if (!shape_matches<ShpH256R8G>(rs.hid_s) && !(ws_half && shape_matches<ShpH256R16G>(rs.hid_s))) return TTRNN_ERR_UNSUPPORTED;
    if (ws_half)      // two fp16 pieces (ttrnn_fast_f10bh.hip); option gemm_pieces = 3 keeps this file's kernel
      st = launch_gru_bwd_f10h(rs, dtype, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws_half, stream, bs);
    else if (dtype == TTRNN_F32)
      st = launch_gru_bwd_f10<ShpH256R8G, float>(rs, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws,
                                                 stream, bs);
    else
      st = launch_gru_bwd_f10<ShpH256R8G, bf16_t>(rs, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws,
                                                  stream, bs);
  } else if (shape_matches<ShpH256R8L>(rs.hid_s)) {
    st = launch_bwd_f10<ShpH256R8L>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, dg_hid, d_h0, d_c0, ws, stream,
                                    bs, ws_half);
  } else if (shape_matches<ShpH256R16L>(rs.hid_s)) {
    st = launch_bwd_f10<ShpH256R16L>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, dg_hid, d_h0, d_c0, ws, stream,
                                     bs, ws_half);
  }
  if (st == TTRNN_OK && stats) st = launch_bwd_stats_finish(rs.cell, rs.B, GH, bs.part, stats, stream);
  return st;
}

}  // namespace ttrnn
