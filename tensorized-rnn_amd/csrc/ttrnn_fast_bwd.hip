// ttrnn_fast_bwd.hip — shape-specialised reverse-time (BPTT) kernel on fp32 MFMA (gfx950).
//
// Sequential part of the backward pass of one recurrent layer (ttrnn_rnn_backward): per timestep, from
// t = T-1 down to 0,
//   gate phase : dL/dh_t, dL/dc_t and the saved activations -> gradients w.r.t. the gate pre-activations
//                (written to HBM for the batched weight-gradient pass and to LDS for the chain),
//   chain      : dh_{t-1} = W_hid^T d_gates through the TRANSPOSED core chain, stage k = 0 .. d-1:
//                dA_k[row][kk] = sum_m dC_k[i][row][a] * W_k[kk][i*R_k + a]   (m = i*R_k + a)
// on the same MFMA tile machinery as the forward kernel: output features (kk) on the MFMA rows with the core
// fragments resident in VGPRs for all timesteps, chain rows on the MFMA columns.  The flat result of stage k
// IS the C-layout input of stage k+1 (the mirror image of ops.py:89-90), so stages ping-pong through LDS.
// Reference: torch autograd through tensorized_rnn/lstm.py:23-32,123-133 / gru.py:33-44,124-134.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"

namespace ttrnn {

template <class S, int k>
struct Tb {
  using F = St<S, k>;
  static constexpr int K = F::K, M = F::M, R = F::R, ROWS = F::ROWS;
  static constexpr int KT = (K + 15) / 16, RT = (ROWS + 15) / 16;
  static constexpr int WV = (R % 4 == 0) ? 4 : ((R % 2 == 0) ? 2 : 1);   // consecutive m that share i
  static constexpr int NU = M / (4 * WV);
  static constexpr int NSTEP = M / 4;
  static constexpr bool SPLIT = (KT < FAST_NW) && (FAST_NW % KT == 0);
  static constexpr int G = SPLIT ? FAST_NW / KT : 1;
  static constexpr int XK = SPLIT ? 1 : (KT + FAST_NW - 1) / FAST_NW;
  static constexpr int YR = SPLIT ? (RT + G - 1) / G : RT;
  static constexpr int NWREG = XK * NSTEP;
};

template <class S>
constexpr bool bwd_shape_ok() {
  for (int k = 0; k < S::D; ++k) {
    const int R = S::R[k], M = S::I[k] * R, K = S::J[k] * S::R[k + 1];
    const int wv = (R % 4 == 0) ? 4 : ((R % 2 == 0) ? 2 : 1);
    if (M % (4 * wv) != 0) return false;
    if (k < S::D - 1 && K % 4 != 0) return false;      // only the last transposed stage (dx) may have ragged K
  }
  return true;
}

template <class S, int k>
constexpr int nwreg_b() { return Tb<S, k>::NWREG; }

template <class S>
constexpr int maxc_of() {     // largest C-layout image (floats)
  int best = 4;
  for (int k = 0; k < S::D; ++k) {
    int rows = 1;
    for (int m = k + 1; m < S::D; ++m) rows *= S::I[m];
    for (int m = 0; m < k; ++m) rows *= S::J[m];
    const int e = rows * S::I[k] * S::R[k];
    if (e > best) best = e;
  }
  return best;
}

// C-layout offset of (row, m) for stage k: m = i*R + a  ->  i*ROWS*R + row*R + a
template <class S, int k>
__device__ __forceinline__ int b_off(int row, int m) {
  using T = Tb<S, k>;
  return (m / T::R) * (T::ROWS * T::R) + row * T::R + (m % T::R);
}

template <class S, int k, int NW_>
__device__ __forceinline__ void load_wfrag_b(float (&w)[NW_], const float* packed, int wave, int lane) {
  using T = Tb<S, k>;
  static_assert(NW_ == T::NWREG, "weight fragment array size");
  const int r = lane & 15, q = lane >> 4;
  const float* W = packed + woff_of<S>(k);
#pragma unroll
  for (int x = 0; x < T::XK; ++x) {
    const int kt = T::SPLIT ? (wave % T::KT) : (wave + FAST_NW * x);
    const int kk = 16 * kt + r;
#pragma unroll
    for (int u = 0; u < T::NU; ++u)
#pragma unroll
      for (int e = 0; e < T::WV; ++e) {
        const int m = (4 * u + q) * T::WV + e;
        const bool okw = kt < T::KT && kk < T::K;
        const float wv = W[okw ? kk * T::M + m : 0];      // branch-free: in-bounds load, then mask
        w[x * T::NSTEP + u * T::WV + e] = okw ? wv : 0.f;
      }
  }
}

// Cin: C-layout image of stage k (gradient w.r.t. its output); Aout: flat [ROWS][K] gradient w.r.t. its input
template <class S, int k, int NW_>
__device__ __forceinline__ void run_bstage(const float (&w)[NW_], const float* Cin, float* Aout, int wave, int lane) {
  using T = Tb<S, k>;
  const int c = lane & 15, q = lane >> 4;
  float bf[T::YR][T::NSTEP];
#pragma unroll
  for (int y = 0; y < T::YR; ++y) {
    const int rt = T::SPLIT ? (wave / T::KT + T::G * y) : y;
    int row = 16 * rt + c;
    row = row < T::ROWS ? row : T::ROWS - 1;
#pragma unroll
    for (int u = 0; u < T::NU; ++u) {
      const float* p = Cin + b_off<S, k>(row, (4 * u + q) * T::WV);
      if constexpr (T::WV == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        bf[y][4 * u + 0] = v[0]; bf[y][4 * u + 1] = v[1]; bf[y][4 * u + 2] = v[2]; bf[y][4 * u + 3] = v[3];
      } else if constexpr (T::WV == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(p);
        bf[y][2 * u + 0] = v[0]; bf[y][2 * u + 1] = v[1];
      } else {
        bf[y][u] = *p;
      }
    }
  }
  f32x4 acc[T::XK][T::YR];
#pragma unroll
  for (int x = 0; x < T::XK; ++x)
#pragma unroll
    for (int y = 0; y < T::YR; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < T::NSTEP; ++s)
#pragma unroll
    for (int x = 0; x < T::XK; ++x)
#pragma unroll
      for (int y = 0; y < T::YR; ++y)
        acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[x * T::NSTEP + s], bf[y][s], acc[x][y], 0, 0, 0);
#pragma unroll
  for (int x = 0; x < T::XK; ++x) {
    const int kt = T::SPLIT ? (wave % T::KT) : (wave + FAST_NW * x);
#pragma unroll
    for (int y = 0; y < T::YR; ++y) {
      const int rt = T::SPLIT ? (wave / T::KT + T::G * y) : y;
      const int row = 16 * rt + c;
      const int kk0 = 16 * kt + 4 * q;
      if (kt < T::KT && rt < T::RT && row < T::ROWS && kk0 < T::K) {
        if constexpr (T::K % 4 == 0) {
          *reinterpret_cast<f32x4*>(Aout + row * T::K + kk0) = acc[x][y];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (kk0 + j < T::K) Aout[row * T::K + kk0 + j] = acc[x][y][j];
        }
      }
    }
  }
}

// ---- last transposed stage (k = D-1): few output tiles, long contraction -> split the contraction over waves ----
// Stage D-1 produces dh[ROWS][K = J_{D-1}]: KT*RT tiles only (cfg2: 2) but M/4 k-steps (cfg2: 32).  Run as is it
// is a 32-deep dependent MFMA chain on two waves while six idle.  Instead wave w takes tile (w % tiles) and
// contraction slice (w / tiles); the CS partial results go to dhpart[slice][H] and the gate phase adds them up.
template <class S>
constexpr int last_tiles() { return Tb<S, S::D - 1>::KT * Tb<S, S::D - 1>::RT; }
template <class S>
constexpr int last_split() {
  constexpr int tiles = last_tiles<S>();
  if (tiles >= FAST_NW || FAST_NW % tiles != 0) return 1;
  int cs = FAST_NW / tiles;
  while (cs > 1 && Tb<S, S::D - 1>::NU % cs != 0) cs >>= 1;
  return cs;
}
template <class S>
constexpr int nwreg_last() { return Tb<S, S::D - 1>::NSTEP / last_split<S>(); }

template <class S, int NW_>
__device__ __forceinline__ void load_wfrag_b_last(float (&w)[NW_], const float* packed, int wave, int lane) {
  constexpr int k = S::D - 1;
  using T = Tb<S, k>;
  constexpr int tiles = last_tiles<S>(), CS = last_split<S>(), UPS = T::NU / CS;
  static_assert(NW_ == UPS * T::WV, "weight fragment array size");
  const int r = lane & 15, q = lane >> 4;
  const float* W = packed + woff_of<S>(k);
  const int tl = wave % tiles, sl = wave / tiles;
  const int kk = 16 * (tl % T::KT) + r;
#pragma unroll
  for (int u = 0; u < UPS; ++u)
#pragma unroll
    for (int e = 0; e < T::WV; ++e) {
      const int m = (4 * (sl * UPS + u) + q) * T::WV + e;
      w[u * T::WV + e] = (sl < CS && kk < T::K) ? W[kk * T::M + m] : 0.f;
    }
}

template <class S, int NW_>
__device__ __forceinline__ void run_bstage_last(const float (&w)[NW_], const float* Cin, float* dhpart, int wave,
                                                int lane) {
  constexpr int k = S::D - 1;
  using T = Tb<S, k>;
  constexpr int tiles = last_tiles<S>(), CS = last_split<S>(), UPS = T::NU / CS;
  constexpr int H = T::ROWS * T::K;
  const int c = lane & 15, q = lane >> 4;
  const int tl = wave % tiles, sl = wave / tiles;
  const int kt = tl % T::KT, rt = tl / T::KT;
  if (sl >= CS) return;
  int row = 16 * rt + c;
  const bool rok = row < T::ROWS;
  row = rok ? row : T::ROWS - 1;
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < UPS; ++u) {
    const float* p = Cin + b_off<S, k>(row, (4 * (sl * UPS + u) + q) * T::WV);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (T::WV == 4) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(p);
      v[0] = t4[0]; v[1] = t4[1]; v[2] = t4[2]; v[3] = t4[3];
    } else if constexpr (T::WV == 2) {
      const f32x2 t2 = *reinterpret_cast<const f32x2*>(p);
      v[0] = t2[0]; v[1] = t2[1];
    } else {
      v[0] = *p;
    }
#pragma unroll
    for (int e = 0; e < T::WV; ++e) {
      // two interleaved accumulators hide the 40-cycle dependent-MFMA latency
      if ((u * T::WV + e) & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u * T::WV + e], v[e], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u * T::WV + e], v[e], acc0, 0, 0, 0);
    }
  }
  const f32x4 acc = acc0 + acc1;
  const int kk0 = 16 * kt + 4 * q;
  if (rok && kk0 < T::K) {
    float* dst = dhpart + sl * H + row * T::K + kk0;
    if constexpr (T::K % 4 == 0) {
      *reinterpret_cast<f32x4*>(dst) = acc;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (kk0 + j < T::K) dst[j] = acc[j];
    }
  }
}

// dg_in / dg_hid: fp32 [B][T][G*H] (plain layout expected by ttrnn_ttlinear_backward's dy)
template <class S, int CELL, typename TS>
__global__ void __launch_bounds__(FAST_NT) k_rnn_bwd_fast(int B, int T, const TS* __restrict__ out,
                                                          const TS* __restrict__ h0, const TS* __restrict__ c0,
                                                          const float* __restrict__ packed_hid,
                                                          const float* __restrict__ reserve,
                                                          const TS* __restrict__ d_out, const TS* __restrict__ d_hT,
                                                          const TS* __restrict__ d_cT, float* __restrict__ dg_in,
                                                          float* __restrict__ dg_hid, TS* __restrict__ d_h0,
                                                          TS* __restrict__ d_c0) {
  constexpr int D = S::D;
  constexpr int H = in_size_of<S>();
  constexpr int G = CELL == TTRNN_LSTM ? 4 : 3;
  constexpr int GH = G * H;
  constexpr int HPT = (H + FAST_NT - 1) / FAST_NT;
  constexpr int MAXC = maxc_of<S>();

  constexpr int CS = last_split<S>();
  __shared__ __attribute__((aligned(16))) float dhbuf[CS * H];
  __shared__ __attribute__((aligned(16))) float bufA[MAXC];
  __shared__ __attribute__((aligned(16))) float bufB[MAXC];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t b = blockIdx.x;

  // stages 0 .. D-2 own whole tiles; stage D-1 is contraction-split (wl)
  float w0[nwreg_b<S, 0>()];
  float w1[nwreg_b<S, (D > 2 ? 1 : 0)>()];
  float w2[nwreg_b<S, (D > 3 ? 2 : 0)>()];
  float wl[nwreg_last<S>()];
  if constexpr (D > 1) load_wfrag_b<S, 0>(w0, packed_hid, wave, lane);
  if constexpr (D > 2) load_wfrag_b<S, 1>(w1, packed_hid, wave, lane);
  if constexpr (D > 3) load_wfrag_b<S, 2>(w2, packed_hid, wave, lane);
  load_wfrag_b_last<S>(wl, packed_hid, wave, lane);

  // per-thread state for its hidden units: dc (LSTM), the direct dh path (GRU), and three sets of saved-step data whose
  // ROLES rotate (the time loop is unrolled by three, no register copies): set(t) = record(t) [LSTM (i,g,f,o) + c,
  // GRU (r,z,n,hn)], d_out(t) and (GRU) h_{t-1}.  Iteration t consumes set(t) and the c of set(t-1), and ISSUES the loads
  // of set(t-2) into the registers that held set(t+1); nobody touches them before iteration t-1.  All loads are
  // unconditional (clamped indices, null pointers redirected and scaled away at the point of use) and exactly as wide
  // as what is used; raw storage bits are converted where they are consumed — see ttrnn_fast_f10b.hip for why.
  struct RecSet {
    f32x4 a[HPT];
    float c[HPT];
    TS dout[HPT], hprev[HPT];
  };
  RecSet s0, s1, s2;
  float dcs[HPT], dhd[HPT], c0v[HPT];
  const TS* dptr = d_out ? d_out : out;
  const float dscale = d_out ? 1.0f : 0.0f;
  auto issue = [&](RecSet& f, int t) {      // loads of set(t); t may be negative (clamped, never consumed)
    const size_t bt = b * T + (t > 0 ? t : 0);
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
      const int hid = tid + u * FAST_NT;
      if (hid < H) {
        f.a[u] = *reinterpret_cast<const f32x4*>(reserve + res_gate(bt, H, hid));
        if constexpr (CELL == TTRNN_LSTM) f.c[u] = reserve[res_cell((size_t)B * T, bt, H, hid)];
        f.dout[u] = dptr[bt * H + hid];
        if constexpr (CELL == TTRNN_GRU) {
          const TS* hp = t >= 1 ? out + (bt - 1) * H : (h0 ? h0 + b * H : out + bt * H);
          f.hprev[u] = hp[hid];
        }
      }
    }
  };
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int hid = tid + u * FAST_NT;
    const bool ok = hid < H;
    dcs[u] = (ok && CELL == TTRNN_LSTM && d_cT) ? ld(d_cT, b * H + hid) : 0.f;
    c0v[u] = (ok && CELL == TTRNN_LSTM && c0) ? ld(c0, b * H + hid) : 0.f;
    dhd[u] = 0.f;
    if (ok) {
      dhbuf[hid] = d_hT ? ld(d_hT, b * H + hid) : 0.f;
#pragma unroll
      for (int sl = 1; sl < CS; ++sl) dhbuf[sl * H + hid] = 0.f;
    }
    s0.a[u] = s1.a[u] = s2.a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    s0.c[u] = s1.c[u] = s2.c[u] = 0.f;
    s0.dout[u] = s1.dout[u] = s2.dout[u] = TS{};
    s0.hprev[u] = s1.hprev[u] = s2.hprev[u] = TS{};
  }
  if (T > 0) {
    issue(s0, T - 1);
    issue(s1, T - 2);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see k_lstm_fwd_fused
  lds_barrier();

  auto step = [&](const int t, const RecSet& cur, const RecSet& nxt, RecSet& fut) {
    const size_t bt = b * T + t;
    // ---- gate phase ------------------------------------------------------------------------------------
    issue(fut, t - 2);
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
      const int hid = tid + u * FAST_NT;
      if (hid < H) {
        const f32x4 qa = cur.a[u];
        float dht = dhd[u] + to_f32(cur.dout[u]) * dscale;
#pragma unroll
        for (int sl = 0; sl < CS; ++sl) dht += dhbuf[sl * H + hid];
        if constexpr (CELL == TTRNN_LSTM) {
          const float ig = qa[0], gg = qa[1], fg = qa[2], og = qa[3], cy = cur.c[u];
          const float cprev = t > 0 ? nxt.c[u] : c0v[u];
          const float tc = ftanh(cy);
          const float dct = dcs[u] + dht * og * (1.0f - tc * tc);
          const float p0 = dct * gg * ig * (1.0f - ig);
          const float p1 = dct * cprev * fg * (1.0f - fg);
          const float p2 = dct * ig * (1.0f - gg * gg);
          const float p3 = dht * tc * og * (1.0f - og);
          dcs[u] = dct * fg;
          bufA[hid] = p0; bufA[H + hid] = p1; bufA[2 * H + hid] = p2; bufA[3 * H + hid] = p3;
        } else {
          const float rg = qa[0], zg = qa[1], ng = qa[2], hn = qa[3];
          const float hprev = (t > 0 || h0) ? to_f32(cur.hprev[u]) : 0.f;
          const float dn_pre = dht * (1.0f - zg) * (1.0f - ng * ng);
          const float dz_pre = dht * (hprev - ng) * zg * (1.0f - zg);
          const float dr_pre = dn_pre * hn * rg * (1.0f - rg);
          dhd[u] = dht * zg;
          bufA[hid] = dr_pre; bufA[H + hid] = dz_pre; bufA[2 * H + hid] = dn_pre * rg;
          dg_in[bt * GH + 2 * H + hid] = dn_pre;      // the only block where d_gates_in != d_gates_hid
        }
      }
    }
    lds_barrier();
    // The gate-gradient row now sits complete in bufA in exactly the [G*H] order of the HBM row: stream it out
    // as whole 16-byte pieces (coalesced) instead of four scattered dword stores per thread.  bufA is not
    // rewritten before the second chain stage, two barriers away.
    {
      const f32x4* src4 = reinterpret_cast<const f32x4*>(bufA);
      for (int e = tid; e < GH / 4; e += FAST_NT) {
        const f32x4 v = src4[e];
        if constexpr (CELL == TTRNN_LSTM) {
          reinterpret_cast<f32x4*>(dg_in + bt * GH)[e] = v;
          if (dg_hid && dg_hid != dg_in) reinterpret_cast<f32x4*>(dg_hid + bt * GH)[e] = v;
        } else {
          reinterpret_cast<f32x4*>(dg_hid + bt * GH)[e] = v;
          if (e < 2 * H / 4) reinterpret_cast<f32x4*>(dg_in + bt * GH)[e] = v;
        }
      }
    }
    // ---- transposed chain: stage 0 .. D-1, last stage writes dh_{t-1} (flat hidden index) -------------------
    if constexpr (D == 1) {
      run_bstage_last<S>(wl, bufA, dhbuf, wave, lane);
    } else if constexpr (D == 2) {
      run_bstage<S, 0>(w0, bufA, bufB, wave, lane);
      lds_barrier();
      run_bstage_last<S>(wl, bufB, dhbuf, wave, lane);
    } else if constexpr (D == 3) {
      run_bstage<S, 0>(w0, bufA, bufB, wave, lane);
      lds_barrier();
      run_bstage<S, 1>(w1, bufB, bufA, wave, lane);
      lds_barrier();
      run_bstage_last<S>(wl, bufA, dhbuf, wave, lane);
    } else {
      run_bstage<S, 0>(w0, bufA, bufB, wave, lane);
      lds_barrier();
      run_bstage<S, 1>(w1, bufB, bufA, wave, lane);
      lds_barrier();
      run_bstage<S, 2>(w2, bufA, bufB, wave, lane);
      lds_barrier();
      run_bstage_last<S>(wl, bufB, dhbuf, wave, lane);
    }
    lds_barrier();
  };
  for (int t = T - 1; t >= 0; t -= 3) {
    step(t, s0, s1, s2);
    if (t >= 1) step(t - 1, s1, s2, s0);
    if (t >= 2) step(t - 2, s2, s0, s1);
  }
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int hid = tid + u * FAST_NT;
    if (hid < H) {
      if (d_h0) {
        float v = dhd[u];
#pragma unroll
        for (int sl = 0; sl < CS; ++sl) v += dhbuf[sl * H + hid];
        st(d_h0, b * H + hid, v);
      }
      if (CELL == TTRNN_LSTM && d_c0) st(d_c0, b * H + hid, dcs[u]);
    }
  }
}

// =================================================================================================================
// Batched TTLinear backward on MFMA (ttrnn_ttlinear_backward for the shapes in the table): over tiles of NB rows
//   1. recompute the forward chain, keeping every stage input A_k as an LDS image (stacked samples),
//   2. dC_0 = dy;  for k = 0 .. d-1:
//        dW_k += A_k^T dC_k   (MFMA with the chain rows on the 4 contraction slots; each wave owns whole 16x16
//                              tiles of dW_k and keeps them in accumulator registers across ALL row tiles),
//        dC_{k+1} = dA_k = dC_k W_k^T   (the transposed stage of the reverse-time kernel, stacked),
//   3. dx = dA_{d-1};  d_bias = column sums of dy.
// One atomic flush of the accumulators per workgroup at the end.
// =================================================================================================================

template <class S, int k, int NB>
constexpr int csz() { return St<S, k>::ROWS * St<S, k>::M; }      // C-layout image floats per sample

// stacked transposed stage: Cin [NB][C-layout] -> Aout [NB][ROWS][K] flat
template <class S, int k, int NB, int NW_>
__device__ __forceinline__ void lin_bstage(const float (&w)[NW_], const float* Cin, float* Aout, int wave, int lane) {
  using T = Tb<S, k>;
  static_assert(NW_ == T::NWREG, "weight fragment array size");
  constexpr int TOT = NB * T::ROWS, RT_ALL = (TOT + 15) / 16;
  constexpr int CSZ = csz<S, k, NB>();
  constexpr int RSTEP = T::SPLIT ? T::G : 1;
  const int c = lane & 15, q = lane >> 4;
  const int rt0 = T::SPLIT ? (wave / T::KT) : 0;
  constexpr int UC = chunk_of(T::NU);                // fragment reads per chunk (bounds live registers)
  for (int rtb = rt0; rtb < RT_ALL; rtb += 2 * RSTEP) {
    int Rr[2], smp[2], row[2];
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      int R = 16 * (rtb + y * RSTEP) + c;
      R = R < TOT ? R : TOT - 1;
      Rr[y] = R; smp[y] = R / T::ROWS; row[y] = R - smp[y] * T::ROWS;
    }
    f32x4 acc[T::XK][2];
#pragma unroll
    for (int x = 0; x < T::XK; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int u0 = 0; u0 < T::NU; u0 += UC) {
      float bf[2][UC * T::WV];
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int u = 0; u < UC; ++u) {
          const float* p = Cin + smp[y] * CSZ + b_off<S, k>(row[y], (4 * (u0 + u) + q) * T::WV);
          if constexpr (T::WV == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p);
            bf[y][4 * u + 0] = v[0]; bf[y][4 * u + 1] = v[1]; bf[y][4 * u + 2] = v[2]; bf[y][4 * u + 3] = v[3];
          } else if constexpr (T::WV == 2) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(p);
            bf[y][2 * u + 0] = v[0]; bf[y][2 * u + 1] = v[1];
          } else {
            bf[y][u] = *p;
          }
        }
#pragma unroll
      for (int x = 0; x < T::XK; ++x)
#pragma unroll
        for (int s2 = 0; s2 < UC * T::WV; ++s2) {
          const float wv = w[x * T::NSTEP + u0 * T::WV + s2];
          acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, bf[0][s2], acc[x][0], 0, 0, 0);
          acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, bf[1][s2], acc[x][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int x = 0; x < T::XK; ++x) {
      const int kt = T::SPLIT ? (wave % T::KT) : (wave + FAST_NW * x);
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int rt = rtb + y * RSTEP;
        const int R = 16 * rt + c;
        const int kk0 = 16 * kt + 4 * q;
        if (kt < T::KT && rt < RT_ALL && R < TOT && kk0 < T::K) {
          const f32x4 a = acc[x][y];
          if constexpr (T::K % 4 == 0) {
            *reinterpret_cast<f32x4*>(Aout + R * T::K + kk0) = a;      // R*K = smp*ROWS*K + row*K
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (kk0 + j < T::K) Aout[R * T::K + kk0 + j] = a[j];
          }
        }
      }
    }
  }
}

// number of dW_k tiles a wave owns
template <class S, int k>
constexpr int wg_tpw() {
  return (((St<S, k>::K + 15) / 16) * ((St<S, k>::M + 15) / 16) + FAST_NW - 1) / FAST_NW;
}

// dW_k tiles += A_k^T dC_k over the NB stacked samples.  Aimg: a_off<KP> image [NB*ROWS][KP]; Cimg: [NB][C-layout]
template <class S, int k, int NB, int TP_>
__device__ __forceinline__ void wgrad_stage(f32x4 (&acc)[TP_], const float* Aimg, const float* Cimg, int wave,
                                            int lane) {
  using F = St<S, k>;
  constexpr int KT = (F::K + 15) / 16, MT = (F::M + 15) / 16, NTILE = KT * MT;
  static_assert(TP_ == wg_tpw<S, k>(), "accumulator tile count");
  constexpr int TOT = NB * F::ROWS;
  constexpr int CSZ = csz<S, k, NB>();
  static_assert(TOT % 4 == 0, "chain rows must come in quads");
  const int r = lane & 15, q = lane >> 4;
  // tile x of this wave: tix = wave*TP_ + x -> (kt, mt).  When a wave's tiles share one kt (TP_ divides MT) the
  // A-side fragment is read once per row quad and reused for all of them.
  constexpr bool SAME_KT = (MT % TP_ == 0);
  int aoff[TP_], boff[TP_];
  bool aok[TP_], bok[TP_];
#pragma unroll
  for (int x = 0; x < TP_; ++x) {
    const int tix = wave * TP_ + x;
    const int kt = tix / MT, mt = tix - kt * MT;
    const int kk = 16 * kt + r, m = 16 * mt + r;
    aok[x] = tix < NTILE && kk < F::K;
    bok[x] = tix < NTILE && m < F::M;
    aoff[x] = aok[x] ? kk : 0;
    boff[x] = bok[x] ? m : 0;
  }
  for (int R0 = 0; R0 < TOT; R0 += 8) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int R = R0 + 4 * h + q;
      const bool rok = R < TOT;
      R = rok ? R : TOT - 1;
      const int smp = R / F::ROWS, row = R - smp * F::ROWS;
      float a0 = 0.f;
      if constexpr (SAME_KT) {
        a0 = Aimg[a_off<F::KP>(R, aoff[0])];
        a0 = (aok[0] && rok) ? a0 : 0.f;
      }
#pragma unroll
      for (int x = 0; x < TP_; ++x) {
        float a;
        if constexpr (SAME_KT) {
          a = a0;
        } else {
          a = Aimg[a_off<F::KP>(R, aoff[x])];
          a = (aok[x] && rok) ? a : 0.f;
        }
        float b = Cimg[smp * CSZ + b_off<S, k>(row, boff[x])];
        b = bok[x] ? b : 0.f;
        acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[x], 0, 0, 0);
      }
    }
  }
}

template <class S, int k, int TP_>
__device__ __forceinline__ void wgrad_flush(const f32x4 (&acc)[TP_], float* d_packed, int wave, int lane) {
  using F = St<S, k>;
  constexpr int KT = (F::K + 15) / 16, MT = (F::M + 15) / 16, NTILE = KT * MT;
  const int c = lane & 15, q = lane >> 4;
  float* dW = d_packed + woff_of<S>(k);
#pragma unroll
  for (int x = 0; x < TP_; ++x) {
    const int tix = wave * TP_ + x;
    const int kt = tix / MT, mt = tix - kt * MT;
    const int m = 16 * mt + c;
    if (tix < NTILE && m < F::M) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = 16 * kt + 4 * q + j;
        if (kk < F::K) atomicAdd(dW + kk * F::M + m, acc[x][j]);
      }
    }
  }
}

template <class S, int k, int NB>
constexpr int aimg_elems() {   // floats of the stage-k input image (stacked), k in 0..D-1
  return (k >= 0 && k < S::D) ? NB * St<S, (k >= 0 && k < S::D) ? k : 0>::ROWS * St<S, (k >= 0 && k < S::D) ? k : 0>::KP : 4;
}
template <class S, int NB>
constexpr int cmax_elems() {
  int best = 4;
  for (int k = 0; k < S::D; ++k) {
    int rows = 1;
    for (int m = k + 1; m < S::D; ++m) rows *= S::I[m];
    for (int m = 0; m < k; ++m) rows *= S::J[m];
    const int e = NB * rows * S::I[k] * S::R[k];
    if (e > best) best = e;
  }
  // the last data-gradient result ([NB][IN]) also lands in a ping-pong buffer
  int in = NB;
  for (int k = 0; k < S::D; ++k) in *= S::J[k];
  return best > in ? best : in;
}

// Register budget: keeping the forward AND transposed fragments of every stage resident costs (cfg4, r = 16)
// ~180 VGPRs before any working set -> scratch spills inside the MFMA loops.  Above the budget the fragments of a
// stage are (re)loaded from L2 right before the stage runs, once per tile of NB rows, and die with it.
template <class S>
constexpr int resident_regs() {
  int n = 0;
  if (S::D > 1) n += nwreg<S, (S::D > 1 ? 1 : 0)>();
  if (S::D > 2) n += nwreg<S, (S::D > 2 ? 2 : 0)>();
  if (S::D > 3) n += nwreg<S, (S::D > 3 ? 3 : 0)>();
  n += nwreg_b<S, 0>();
  if (S::D > 1) n += nwreg_b<S, (S::D > 1 ? 1 : 0)>();
  if (S::D > 2) n += nwreg_b<S, (S::D > 2 ? 2 : 0)>();
  if (S::D > 3) n += nwreg_b<S, (S::D > 3 ? 3 : 0)>();
  return n;
}
template <class S>
constexpr bool resident_ok() { return resident_regs<S>() <= 160; }
template <class S>
constexpr bool big_shape() { return resident_regs<S>() > 96; }   // register-hungry: keep address math in the tile loop

template <class S, int k, int NB>
__device__ __forceinline__ void fwd_stage_reload(const float* packed, const float* in, float* out, int wave, int lane) {
  // opaque zero offset: without it LICM hoists these loop-invariant loads out of the tile loop, which makes
  // the fragments resident again (-> spilled to scratch and re-read before every MFMA)
  int z = 0;
  asm volatile("" : "+v"(z));        // per-lane opaque zero: keeps the per-lane address arithmetic in the loop too
  float w[nwreg<S, k>()];
  load_wfrag<S, k>(w, packed, wave, lane + z);
  lin_stage<S, k, NB, 0>(w, in, out, wave, lane, 0);
}
template <class S, int k, int NB>
__device__ __forceinline__ void bwd_stage_reload(const float* packed, const float* in, float* out, int wave, int lane) {
  int z = 0;
  asm volatile("" : "+v"(z));
  float w[nwreg_b<S, k>()];
  load_wfrag_b<S, k>(w, packed, wave, lane + z);
  lin_bstage<S, k, NB>(w, in, out, wave, lane);
}

// NEED_DX = false drops the last transposed stage and its resident core fragments (registers!)
template <class S, int NB, typename TI, typename TDY, bool NEED_DX>
__global__ void __launch_bounds__(FAST_NT) k_ttlinear_bwd_fast(int64_t n_rows, const float* __restrict__ packed,
                                                               const TI* __restrict__ x, const TDY* __restrict__ dy,
                                                               TI* __restrict__ dx, float* __restrict__ d_packed,
                                                               float* __restrict__ d_bias) {
  constexpr int D = S::D;
  constexpr int IN = in_size_of<S>(), OUT = out_size_of<S>();
  using SL = St<S, D - 1>;
  constexpr int CMAX = cmax_elems<S, NB>();
  constexpr int OPT = (OUT + FAST_NT - 1) / FAST_NT;

  __shared__ __attribute__((aligned(16))) float a0[aimg_elems<S, 0, NB>()];
  __shared__ __attribute__((aligned(16))) float a1[aimg_elems<S, 1, NB>()];
  __shared__ __attribute__((aligned(16))) float a2[aimg_elems<S, 2, NB>()];
  __shared__ __attribute__((aligned(16))) float a3[aimg_elems<S, 3, NB>()];
  __shared__ __attribute__((aligned(16))) float cA[CMAX];
  __shared__ __attribute__((aligned(16))) float cB[CMAX];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* aimg[4] = {a0, a1, a2, a3};
  float* img_in = aimg[D - 1];       // image that receives x

  // forward fragments (stages D-1 .. 1) and transposed fragments (stages 0 .. D-1)
  float wf1[nwreg<S, (D > 1 ? 1 : 0)>()];
  float wf2[nwreg<S, (D > 2 ? 2 : 0)>()];
  float wf3[nwreg<S, (D > 3 ? 3 : 0)>()];
  constexpr bool RES = resident_ok<S>();
  if constexpr (RES && D > 1) load_wfrag<S, 1>(wf1, packed, wave, lane);
  if constexpr (RES && D > 2) load_wfrag<S, 2>(wf2, packed, wave, lane);
  if constexpr (RES && D > 3) load_wfrag<S, 3>(wf3, packed, wave, lane);
  constexpr int LASTB = NEED_DX ? D - 1 : D - 2;     // last transposed stage that has to run
  float wb0[nwreg_b<S, 0>()];
  float wb1[nwreg_b<S, (LASTB >= 1 ? 1 : 0)>()];
  float wb2[nwreg_b<S, (LASTB >= 2 ? 2 : 0)>()];
  float wb3[nwreg_b<S, (LASTB >= 3 ? 3 : 0)>()];
  if constexpr (RES && LASTB >= 0) load_wfrag_b<S, 0>(wb0, packed, wave, lane);
  if constexpr (RES && LASTB >= 1) load_wfrag_b<S, 1>(wb1, packed, wave, lane);
  if constexpr (RES && LASTB >= 2) load_wfrag_b<S, 2>(wb2, packed, wave, lane);
  if constexpr (RES && LASTB >= 3) load_wfrag_b<S, 3>(wb3, packed, wave, lane);

  f32x4 g0[wg_tpw<S, 0>()];
  f32x4 g1[wg_tpw<S, (D > 1 ? 1 : 0)>()];
  f32x4 g2[wg_tpw<S, (D > 2 ? 2 : 0)>()];
  f32x4 g3[wg_tpw<S, (D > 3 ? 3 : 0)>()];
#pragma unroll
  for (int i = 0; i < wg_tpw<S, 0>(); ++i) g0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < wg_tpw<S, (D > 1 ? 1 : 0)>(); ++i) g1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < wg_tpw<S, (D > 2 ? 2 : 0)>(); ++i) g2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < wg_tpw<S, (D > 3 ? 3 : 0)>(); ++i) g3[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbias[OPT];
#pragma unroll
  for (int u = 0; u < OPT; ++u) dbias[u] = 0.f;

  // zero once: K padding of the first image
  for (int e = tid; e < NB * SL::ROWS * SL::KP; e += FAST_NT) img_in[e] = 0.f;
  __syncthreads();

  const bool want_w = d_packed != nullptr;
  const int64_t ntiles = (n_rows + NB - 1) / NB;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t n0 = tile * NB;
    // opaque per-iteration copies: stop LICM from hoisting the (hundreds of) unrolled LDS address computations of
    // all stages out of the tile loop, where they would all be live at once and spill to scratch
    int lane_i = lane, wave_i = wave;
    if constexpr (big_shape<S>()) {      // small shapes have registers to spare: let the compiler hoist
      asm volatile("" : "+v"(lane_i));
      asm volatile("" : "+s"(wave_i));
    }
    // ---- loads: x -> first forward image; dy -> dC_0 (C-layout of stage 0 == flat o) ---------------------------
    if (want_w) {
      for (int e = tid; e < NB * IN; e += FAST_NT) {
        const int smp = e / IN, j = e - smp * IN;
        const float v = (n0 + smp < n_rows) ? ld(x, (n0 + smp) * IN + j) : 0.f;
        img_in[a_off<SL::KP>(smp * SL::ROWS + j / SL::K, j % SL::K)] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < OPT; ++u) {
      const int o = tid + u * FAST_NT;
      if (o < OUT) {
#pragma unroll
        for (int smp = 0; smp < NB; ++smp) {
          const float v = (n0 + smp < n_rows) ? ld(dy, (n0 + smp) * OUT + o) : 0.f;
          cA[smp * OUT + o] = v;
          dbias[u] += v;
        }
      }
    }
    __syncthreads();
    // ---- forward recompute: A_{D-1} -> ... -> A_0 ---------------------------------------------------------------
    if (want_w) {
      if constexpr (D == 2) {
        if constexpr (RES) lin_stage<S, 1, NB, 0>(wf1, a1, a0, wave_i, lane_i, 0);
        else fwd_stage_reload<S, 1, NB>(packed, a1, a0, wave_i, lane_i);
        __syncthreads();
      } else if constexpr (D == 3) {
        if constexpr (RES) lin_stage<S, 2, NB, 0>(wf2, a2, a1, wave_i, lane_i, 0);
        else fwd_stage_reload<S, 2, NB>(packed, a2, a1, wave_i, lane_i);
        __syncthreads();
        if constexpr (RES) lin_stage<S, 1, NB, 0>(wf1, a1, a0, wave_i, lane_i, 0);
        else fwd_stage_reload<S, 1, NB>(packed, a1, a0, wave_i, lane_i);
        __syncthreads();
      } else if constexpr (D == 4) {
        if constexpr (RES) lin_stage<S, 3, NB, 0>(wf3, a3, a2, wave_i, lane_i, 0);
        else fwd_stage_reload<S, 3, NB>(packed, a3, a2, wave_i, lane_i);
        __syncthreads();
        if constexpr (RES) lin_stage<S, 2, NB, 0>(wf2, a2, a1, wave_i, lane_i, 0);
        else fwd_stage_reload<S, 2, NB>(packed, a2, a1, wave_i, lane_i);
        __syncthreads();
        if constexpr (RES) lin_stage<S, 1, NB, 0>(wf1, a1, a0, wave_i, lane_i, 0);
        else fwd_stage_reload<S, 1, NB>(packed, a1, a0, wave_i, lane_i);
        __syncthreads();
      }
    }
    // ---- backward: per stage weight gradient (reads A_k, dC_k) and data gradient (dC_k -> dC_{k+1}) --------------
    // the two read the same dC_k and write disjoint places: no barrier between them
    if (want_w) wgrad_stage<S, 0, NB>(g0, a0, cA, wave_i, lane_i);
    if constexpr (LASTB >= 0) {
      if constexpr (RES) lin_bstage<S, 0, NB>(wb0, cA, cB, wave_i, lane_i);
      else bwd_stage_reload<S, 0, NB>(packed, cA, cB, wave_i, lane_i);
    }
    __syncthreads();
    float* last = cB;
    if constexpr (D > 1) {
      if (want_w) wgrad_stage<S, 1, NB>(g1, a1, cB, wave_i, lane_i);
      if constexpr (LASTB >= 1) {
        if constexpr (RES) lin_bstage<S, 1, NB>(wb1, cB, cA, wave_i, lane_i);
        else bwd_stage_reload<S, 1, NB>(packed, cB, cA, wave_i, lane_i);
      }
      __syncthreads();
      last = cA;
    }
    if constexpr (D > 2) {
      if (want_w) wgrad_stage<S, 2, NB>(g2, a2, cA, wave_i, lane_i);
      if constexpr (LASTB >= 2) {
        if constexpr (RES) lin_bstage<S, 2, NB>(wb2, cA, cB, wave_i, lane_i);
        else bwd_stage_reload<S, 2, NB>(packed, cA, cB, wave_i, lane_i);
      }
      __syncthreads();
      last = cB;
    }
    if constexpr (D > 3) {
      if (want_w) wgrad_stage<S, 3, NB>(g3, a3, cB, wave_i, lane_i);
      if constexpr (LASTB >= 3) {
        if constexpr (RES) lin_bstage<S, 3, NB>(wb3, cB, cA, wave_i, lane_i);
        else bwd_stage_reload<S, 3, NB>(packed, cB, cA, wave_i, lane_i);
      }
      __syncthreads();
      last = cA;
    }
    if constexpr (NEED_DX) {
      for (int e = tid; e < NB * IN; e += FAST_NT) {
        const int smp = e / IN;
        if (n0 + smp < n_rows) st(dx, n0 * IN + e, last[e]);
      }
      __syncthreads();
    }
  }
  // ---- flush ---------------------------------------------------------------------------------------------------------
  if (want_w) {
    wgrad_flush<S, 0>(g0, d_packed, wave, lane);
    if constexpr (D > 1) wgrad_flush<S, 1>(g1, d_packed, wave, lane);
    if constexpr (D > 2) wgrad_flush<S, 2>(g2, d_packed, wave, lane);
    if constexpr (D > 3) wgrad_flush<S, 3>(g3, d_packed, wave, lane);
  }
  if (d_bias) {
#pragma unroll
    for (int u = 0; u < OPT; ++u) {
      const int o = tid + u * FAST_NT;
      if (o < OUT && dbias[u] != 0.f) atomicAdd(d_bias + o, dbias[u]);
    }
  }
}

template <class S, int NB, typename TI, typename TDY>
static int launch_lin_bwd_tt(int64_t n_rows, const float* packed, const void* x, const void* dy, void* dx,
                             float* d_packed, float* d_bias, hipStream_t stream) {
  static_assert(bwd_shape_ok<S>() && shape_ok<S>(), "shape not supported by the MFMA backward path");
  const int64_t ntiles = (n_rows + NB - 1) / NB;
  const int grid = (int)(ntiles < 1 ? 1 : (ntiles > 256 ? 256 : ntiles));
  if (dx)
    hipLaunchKernelGGL((k_ttlinear_bwd_fast<S, NB, TI, TDY, true>), dim3(grid), dim3(FAST_NT), 0, stream, n_rows,
                       packed, (const TI*)x, (const TDY*)dy, (TI*)dx, d_packed, d_bias);
  else
    hipLaunchKernelGGL((k_ttlinear_bwd_fast<S, NB, TI, TDY, false>), dim3(grid), dim3(FAST_NT), 0, stream, n_rows,
                       packed, (const TI*)x, (const TDY*)dy, (TI*)dx, d_packed, d_bias);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

template <class S, int NB>
static int launch_lin_bwd(int dtype, int dy_dtype, int64_t n_rows, const float* packed, const void* x, const void* dy,
                          void* dx, float* d_packed, float* d_bias, hipStream_t stream) {
  if (dtype == TTRNN_F32 && dy_dtype == TTRNN_F32)
    return launch_lin_bwd_tt<S, NB, float, float>(n_rows, packed, x, dy, dx, d_packed, d_bias, stream);
  if (dtype == TTRNN_BF16 && dy_dtype == TTRNN_F32)
    return launch_lin_bwd_tt<S, NB, bf16_t, float>(n_rows, packed, x, dy, dx, d_packed, d_bias, stream);
  if (dtype == TTRNN_BF16 && dy_dtype == TTRNN_BF16)
    return launch_lin_bwd_tt<S, NB, bf16_t, bf16_t>(n_rows, packed, x, dy, dx, d_packed, d_bias, stream);
  return TTRNN_ERR_UNSUPPORTED;
}

// rows per tile: everything (all A_k images + two gradient images) must fit 160 KB of LDS
#define TT_LINB_SHAPES(X) \
  X(ShpI1R8L, 8)          \
  X(ShpI1R8G, 8)          \
  X(ShpI40R16L, 2)        \
  X(ShpI1R4L, 16)         \
  X(ShpH256R8L, 2)        \
  X(ShpH256R8G, 2)        \
  X(ShpH256R16L, 1)       \
  X(ShpH256R16G, 1)       \
  X(ShpH128R4L, 8)        \
  X(ShpHd256R16, 2)

bool fast_ttlinear_bwd_available(const TtShape& s, int dtype, int dy_dtype) {
  if (dtype == TTRNN_F32 && dy_dtype != TTRNN_F32) return false;
  if (dtype != TTRNN_F32 && dtype != TTRNN_BF16) return false;
#define TT_X(SHAPE, NBV) \
  if (shape_matches<SHAPE>(s)) return true;
  TT_LINB_SHAPES(TT_X)
#undef TT_X
  return false;
}

int launch_ttlinear_bwd_fast(const TtShape& s, int dtype, int dy_dtype, int64_t n_rows, const float* packed,
                             const void* x, const void* dy, void* dx, float* d_packed, float* d_bias,
                             hipStream_t stream) {
  if (n_rows == 0) return TTRNN_OK;
#define TT_X(SHAPE, NBV)        \
  if (shape_matches<SHAPE>(s)) \
    return launch_lin_bwd<SHAPE, NBV>(dtype, dy_dtype, n_rows, packed, x, dy, dx, d_packed, d_bias, stream);
  TT_LINB_SHAPES(TT_X)
#undef TT_X
  return TTRNN_ERR_UNSUPPORTED;
}

// ---- dispatch --------------------------------------------------------------------------------------------
template <class S, int CELL, typename TS>
static int launch_bwd_t(const RnnShape& rs, const void* out, const void* h0, const void* c0, const float* packed_hid,
                        const float* reserve, const void* d_out, const void* d_hT, const void* d_cT, float* dg_in,
                        float* dg_hid, void* d_h0, void* d_c0, hipStream_t stream) {
  static_assert(bwd_shape_ok<S>(), "shape not supported by the MFMA backward path");
  hipLaunchKernelGGL((k_rnn_bwd_fast<S, CELL, TS>), dim3(rs.B), dim3(FAST_NT), 0, stream, rs.B, rs.T, (const TS*)out,
                     (const TS*)h0, (const TS*)c0, packed_hid, reserve, (const TS*)d_out, (const TS*)d_hT,
                     (const TS*)d_cT, dg_in, dg_hid, (TS*)d_h0, (TS*)d_c0);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

bool fast_rnn_bwd_available(const RnnShape& rs, int dtype) {
  if ((dtype != TTRNN_F32 && dtype != TTRNN_BF16) || rs.B < 1 || rs.T < 1) return false;
  if (rs.cell == TTRNN_LSTM)
    return shape_matches<ShpH256R8L>(rs.hid_s) || shape_matches<ShpH256R16L>(rs.hid_s) ||
           shape_matches<ShpH128R4L>(rs.hid_s);
  // TT-GRU r = 16 has no fused-core reverse kernel: this family would run its stage-wise kernel (1.50 ms at benchmarking.py
  // --hidden_size 256 --gru --ttrank 16), the runtime tier's reverse kernel — same reserve — takes 0.95: the tier where it is on offer
  // (split mode / bf16; dev bit 26: the stage-wise kernel, A/B)
  // (round 6: the two-fp16-piece fused-core kernel takes rank 16 — k_gru_bwd_f10h<ShpH256R16G>; option dev2 bit 7 = the tier again)
  if (shape_matches<ShpH256R16G>(rs.hid_s))
    return (f10bh_available(rs, dtype) && !opt(OPT_NO_F10)) || (opt(OPT_DEV) & (1 << 26)) || opt(OPT_FORCE_GENERIC) ||
           !g2_rnn_bwd_available(rs, dtype);
  return shape_matches<ShpH256R8G>(rs.hid_s);
}

int launch_rnn_bwd_fast(const RnnShape& rs, int dtype, const void* out, const void* h0, const void* c0,
                        const float* packed_hid, const float* reserve, const void* d_out, const void* d_hT,
                        const void* d_cT, float* dg_in, float* dg_hid, void* d_h0, void* d_c0, hipStream_t stream) {
#define TT_TRY(SHAPE, CELL)                                                                                       \
  if (rs.cell == CELL && shape_matches<SHAPE>(rs.hid_s))                                                          \
    return dtype == TTRNN_F32                                                                                     \
               ? launch_bwd_t<SHAPE, CELL, float>(rs, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, \
                                                  dg_hid, d_h0, d_c0, stream)                                     \
               : launch_bwd_t<SHAPE, CELL, bf16_t>(rs, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT,       \
                                                   dg_in, dg_hid, d_h0, d_c0, stream)
  TT_TRY(ShpH256R8L, TTRNN_LSTM);
  TT_TRY(ShpH256R16L, TTRNN_LSTM);
  TT_TRY(ShpH128R4L, TTRNN_LSTM);
  TT_TRY(ShpH256R8G, TTRNN_GRU);
  TT_TRY(ShpH256R16G, TTRNN_GRU);
#undef TT_TRY
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
