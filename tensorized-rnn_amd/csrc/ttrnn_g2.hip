// ttrnn_g2.hip — runtime-shape two-stage MFMA kernels (gfx950): the recurrent forward and the reverse-time kernel of
// BPTT for ANY TT-LSTM / TT-GRU layer whose hidden matrix has d >= 2 cores (ttrnn_g2.h: plan and algebra).  This is the
// tier between the shape-specialised kernels (register-resident fragments: cfg1..cfg5) and the any-shape VALU kernels
// (ttrnn_generic.hip): every (hidden_size, ncores, ttrank, new_core) the reference's experiment flags can produce
// (pmnist_test.py:47-56, params_model.py) runs its time loop on the matrix cores.
//
// Arithmetic: fp32-class whatever the storage type: every product of both stages, in both directions, is the three-term product of
// two fp16 pieces per operand on v_mfma_f32_16x16x32_f16 with fp32 accumulation (ttrnn_split.h), each operand under exact
// power-of-two scales — the merged cores per row / per rank slice (prep kernels, once per launch), h under 2^9 (|h| < 1), the
// reverse kernel's gate gradients under the scale of the step's exact maximum and its intermediate dC1 under the bound
// L1(row) x max|dg_t| (the scheme of ttrnn_fast_f10bh.hip).  (Rounds 2 - 3: three bf16 pieces in the big stage, the fp32 MFMA in
// the small one.)
//
// Data movement: one workgroup of 4 or 8 waves per sample for all T steps, h / c in registers, all intermediates in LDS.
// A wave's fragments of the merged head core stay in registers where they fit (forward: 8 or 16 slots, reverse: 12) — the
// eight-wave plan is preferred where only it makes them fit — and are otherwise STREAMED from L2 every step, written by the prep
// kernel in exactly the order the wave consumes them (coalesced 16-byte loads; the reverse stream holds a unit's live k-blocks
// only), G2_PF k-blocks in flight through rolling register slots; the loads of step t+1 are issued while step t still
// multiplies.  Every thread owns hidden units tid + u * 64 nw in the gate phases; the per-step records / gate inputs are requested
// a step ahead, outside any lane-divergent region (DESIGN.md lesson 53).  Several samples share a CU when B > #CUs.
//
// Replaces, for one layer: tensorized_rnn/lstm.py:23-32,123-133 / gru.py:33-44,124-134 with the hidden chain of
// t3nsor/ops.py:78-93, and torch autograd through them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_g2.h"

namespace ttrnn {

namespace {

// ---- prep 1: contract the cores on either side of the split point ------------------------------------------------------
// packed: W_k[(j*R_{k+1} + b)*M_k + (i*R_k + a)] = G_k[a, i, j, b]  (include/ttrnn.h)
// status (may be null): the hid_blocks promise of the descriptor is CHECKED here, where every head entry passes through a
// register anyway — a non-zero entry outside its gate's rank block (which the kernels will skip) is counted in
// TTRNN_STAT_BLOCK_VIOLATIONS (the promise used to be taken on trust)
// One thread per ENTRY (row, column, rank) of a merged core: the chain through the cores before the last one runs whole (it is short:
// every rank of the previous boundary is needed), the last contraction only for the thread's own rank.  (One thread per (row, column)
// with all R ranks of every stage took 168 us on the joint matrix of a naive per-gate set — 64 workgroups, 1 152 dependent
// multiply-adds each through two rank-32 cores, its vectors in scratch.)
__global__ void __launch_bounds__(256) k_g2_merge(TtShape s, G2Mat m, const float* __restrict__ packed,
                                                  float* __restrict__ Gh, float* __restrict__ Gt, unsigned* status) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nh = (long)m.Ih * m.Jh * m.R, nt = (long)m.It * m.Jt * m.R;
  float v[G2_MAX_R], w[G2_MAX_R];
  if (t < nh) {
    const long e = t / m.R;
    const int own = (int)(t - e * m.R);
    int ih = (int)(e / m.Jh), jh = (int)(e % m.Jh);
    int ii[TTRNN_MAX_D], jj[TTRNN_MAX_D];
    for (int k = m.s - 1; k >= 0; --k) { ii[k] = ih % s.I[k]; ih /= s.I[k]; jj[k] = jh % s.J[k]; jh /= s.J[k]; }
    const float* W0 = packed + s.woff[0];
    float out;
    if (m.s == 1) {
      out = W0[(size_t)(jj[0] * s.R[1] + own) * s.M[0] + ii[0]];
    } else {
      for (int b = 0; b < s.R[1]; ++b) v[b] = W0[(size_t)(jj[0] * s.R[1] + b) * s.M[0] + ii[0]];
      for (int k = 1; k < m.s - 1; ++k) {
        const float* Wk = packed + s.woff[k];
        for (int b = 0; b < s.R[k + 1]; ++b) {
          float acc = 0.f;
          for (int a = 0; a < s.R[k]; ++a) acc = fmaf(v[a], Wk[(size_t)(jj[k] * s.R[k + 1] + b) * s.M[k] + ii[k] * s.R[k] + a], acc);
          w[b] = acc;
        }
        for (int b = 0; b < s.R[k + 1]; ++b) v[b] = w[b];
      }
      const int k = m.s - 1;
      const float* Wk = packed + s.woff[k];
      float acc = 0.f;
      for (int a = 0; a < s.R[k]; ++a) acc = fmaf(v[a], Wk[(size_t)(jj[k] * s.R[k + 1] + own) * s.M[k] + ii[k] * s.R[k] + a], acc);
      out = acc;
    }
    Gh[t] = out;
    if (m.ng > 1 && status) {
      const int g = (int)(e / m.Jh) / m.IhG;
      if (own / m.Rb != g && out != 0.f) atomicAdd(status + TTRNN_STAT_BLOCK_VIOLATIONS, 1u);
    }
  } else if (t < nh + nt) {
    const long f = (t - nh) / m.R;
    const int own = (int)(t - nh - f * m.R);
    int it = (int)(f / m.Jt), jt = (int)(f % m.Jt);
    int ii[TTRNN_MAX_D], jj[TTRNN_MAX_D];
    for (int k = s.d - 1; k >= m.s; --k) { ii[k] = it % s.I[k]; it /= s.I[k]; jj[k] = jt % s.J[k]; jt /= s.J[k]; }
    const int kl = s.d - 1;
    const float* Wl = packed + s.woff[kl];
    float out;
    if (kl == m.s) {
      out = Wl[(size_t)jj[kl] * s.M[kl] + ii[kl] * s.R[kl] + own];       // R_d = 1
    } else {
      for (int a = 0; a < s.R[kl]; ++a) v[a] = Wl[(size_t)jj[kl] * s.M[kl] + ii[kl] * s.R[kl] + a];
      for (int k = kl - 1; k > m.s; --k) {
        const float* Wk = packed + s.woff[k];
        for (int a = 0; a < s.R[k]; ++a) {
          float acc = 0.f;
          for (int b = 0; b < s.R[k + 1]; ++b) acc = fmaf(Wk[(size_t)(jj[k] * s.R[k + 1] + b) * s.M[k] + ii[k] * s.R[k] + a], v[b], acc);
          w[a] = acc;
        }
        for (int a = 0; a < s.R[k]; ++a) v[a] = w[a];
      }
      const int k = m.s;
      const float* Wk = packed + s.woff[k];
      float acc = 0.f;
      for (int b = 0; b < s.R[k + 1]; ++b) acc = fmaf(Wk[(size_t)(jj[k] * s.R[k + 1] + b) * s.M[k] + ii[k] * s.R[k] + own], v[b], acc);
      out = acc;
    }
    Gt[t - nh] = out;
  }
}

// The dense matrix of a TT-matrix from its merged cores, gate-interleaved like the chain kernel's output on identity rows
// (ttrnn_mfma.h: ytile_index): W[j][hid * 4 + slot(g)] = sum_a Gh[i_h, j_h, a] Gt[i_t, j_t, a],  j = j_h J_t + j_t, o = g H + hid = i_h I_t + i_t.
// K-in of the tier (input_size != 1) builds its dense matrix by running the any-shape chain kernel on the `in` identity rows: 83 us at the
// reference's benchmark defaults (in = 256, 4H = 2048), 165 for a naive set's joint matrix; merge + this: ~10 (behind `dev` bit 24: see fwd_t).
__global__ void __launch_bounds__(256) k_g2_dense(G2Mat m, const float* __restrict__ Gh, const float* __restrict__ Gt,
                                                  float* __restrict__ W, int ldw, int H, int ilv) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)m.in * m.out) return;
  const int j = (int)(t / m.out), o = (int)(t - (long)j * m.out);
  const int ih = o / m.It, it = o - ih * m.It, jh = j / m.Jt, jt = j - jh * m.Jt;
  const float* a = Gh + ((size_t)ih * m.Jh + jh) * m.R;
  const float* b = Gt + ((size_t)it * m.Jt + jt) * m.R;
  float acc0 = 0.f, acc1 = 0.f;
  int r = 0;
  for (; r + 1 < m.R; r += 2) { acc0 = fmaf(a[r], b[r], acc0); acc1 = fmaf(a[r + 1], b[r + 1], acc1); }
  if (r < m.R) acc0 = fmaf(a[r], b[r], acc0);
  const int g = o / H, hid = o - g * H;
  const int slot = ilv == 2 ? (g == 1 ? 2 : (g == 2 ? 1 : g)) : g;
  W[(size_t)j * ldw + hid * 4 + slot] = acc0 + acc1;
}

// ---- forward operand scales (two-piece fp16 flavour of ttrnn_split.h) ---------------------------------------------------------
// Both stages run on two-piece fp16 operands under TWO-SIDED DIAGONAL power-of-two scales (as ttrnn_f10_dev.h), so that one
// large entry of a core moves only the scale of its own row / rank slice and every other entry keeps its 22 bits:
//   tail fragments   Gt'[(i_t, a)][j_t] = Gt 2^(13 + eu[i_t] + ev[a])     each i_t row block and each rank slice a: max < 2^13
//   h image          2^9 h   (|h_t| <= 1 for t >= 1, LSTM and GRU alike; a caller's h_0 enters times a per-sample power of two
//                             that the sums are multiplied back by)
//   stage-1 sums     < J_t 2^22, times the fixed r1 = 2^(15 - 22 - ceil(log2 J_t)) < 2^15 before they are split
//   head fragments   Gh'[i_h][(j_h, a)] = Gh 2^(ep[i_h] - ev[a])           each row: max < 2^14
//   stage-2 sums     y'[i_h][i_t] = 2^(13 + 9 + r + ep[i_h] + eu[i_t]) y: undone (exactly) where the sums are stored
// eu, ev, ep: int32 exponents computed once per launch by k_g2_diag_a / _b from the merged cores, hdr = [eu: I_t | ev: 64 | ep: I_h]
// (followed by the [I_t][64] partial maxima the two kernels hand over).
__device__ __forceinline__ int g2_expo(float x) {          // x < 2^e; zero / non-finite: neutral
  if (!(x > 0.f)) return 0;
  if (!(x < 3e38f)) return 40;
  int e;
  frexpf(x, &e);
  return e < -40 ? -40 : (e > 40 ? 40 : e);
}
constexpr float G2_HSC = 512.0f;
__device__ __forceinline__ int g2_r1_expo(int Jt) {
  int ej = 0;
  while ((1 << ej) < Jt) ++ej;                             // J_t <= 2^ej
  return 15 - 22 - ej;
}
// Two launches over the merged cores (L2-resident), every load independent of the others:
//   k_g2_diag_a  one workgroup per i_t:  eu[i_t] = -expo(max_{a, j_t} |Gt|), and part[i_t][a] = max_{j_t} 2^eu |Gt|
//   k_g2_diag_b  one workgroup per i_h:  ev[a] = -expo(max_{i_t} part[i_t][a])  (every workgroup reduces it for itself; the
//                first one stores it),   ep[i_h] = 14 - expo(max_{j_h, a} 2^-ev |Gh|)
// (one 1024-thread workgroup walking both cores with LDS atomics was a chain of dependent L2 latencies: > 100 us.)
__global__ void __launch_bounds__(256) k_g2_diag_a(G2Mat m, const float* __restrict__ Gt, int* __restrict__ hdr,
                                                   float* __restrict__ part) {
  __shared__ float red[4];
  __shared__ unsigned mv[64];
  __shared__ int eus;
  const int tid = threadIdx.x, it = blockIdx.x;
  const float* row = Gt + (size_t)it * m.Jt * m.R;
  const int n = m.Jt * m.R;
  if (tid < 64) mv[tid] = 0u;
  float mx = 0.f;
  for (int i = tid; i < n; i += 256) mx = fmaxf(mx, fabsf(row[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    eus = -g2_expo(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    hdr[it] = eus;
  }
  __syncthreads();
  const float sc = ldexpf(1.f, eus);
  for (int i = tid; i < n; i += 256) atomicMax(&mv[i % m.R], __float_as_uint(fabsf(row[i]) * sc));   // |x| orders like its bits
  __syncthreads();
  if (tid < 64) part[(size_t)it * 64 + tid] = __uint_as_float(mv[tid]);
}
__global__ void __launch_bounds__(256) k_g2_diag_b(G2Mat m, const float* __restrict__ Gh, int* __restrict__ hdr,
                                                   const float* __restrict__ part) {
  __shared__ float red[4];
  __shared__ float pv[4][64];
  __shared__ int ev[64];
  const int tid = threadIdx.x, ih = blockIdx.x;
  {
    const int a = tid & 63, g = tid >> 6;
    float mx = 0.f;
    for (int it = g; it < m.It; it += 4) mx = fmaxf(mx, part[(size_t)it * 64 + a]);
    pv[g][a] = mx;
  }
  __syncthreads();
  if (tid < 64) {
    const int e = tid < m.R ? -g2_expo(fmaxf(fmaxf(pv[0][tid], pv[1][tid]), fmaxf(pv[2][tid], pv[3][tid]))) : 0;
    ev[tid] = e;
    if (ih == 0) hdr[m.It + tid] = e;
  }
  __syncthreads();
  const float* row = Gh + (size_t)ih * m.Jh * m.R;
  const int n = m.Jh * m.R;
  float mx = 0.f;
  for (int i = tid; i < n; i += 256) mx = fmaxf(mx, fabsf(row[i]) * ldexpf(1.f, -ev[i % m.R]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) hdr[m.It + 64 + ih] = 14 - g2_expo(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// ---- prep 2: merged cores -> MFMA fragments in consumption order ---------------------------------------------------------
// head stream: block (wave w, unit slot ui, local k-block kbl) at ((w*UW + ui)*KBP + kbl) (forward) — the reverse stream is compact:
// (w*bSW + ui*bNKBt + kbl), its units' live blocks back to back (bSW = the longest wave's stream padded to the slot ring)
//   forward  (REV = false): MFMA row r <-> i_h = 16 mt + r,            k = 32 kb + 8 q + e <-> (j_h, a) = divmod(k, Rp)
//   reverse  (REV = true):  MFMA row r <-> (j_h, a) = divmod(16 mt + r, Rp),  k <-> i_h
// forward (REV = false): TWO fp16 planes of the scaled Gh (k_g2_diag_a / _b), block stride 2 * 64 lanes
// reverse: TWO fp16 planes as well, every ROW (j_h, a) of head^T under its own power-of-two scale (row maximum -> [2^13, 2^14): the rows
// are T2's OUTPUT rows, the scale is undone on the accumulators: hun[row] = its inverse) — the reasoning of ttrnn_fast_f10bh.hip
template <bool REV>
__global__ void __launch_bounds__(64) k_g2_head_frag(G2Mat m, const float* __restrict__ Gh, xbf8* __restrict__ fs,
                                                     const int* __restrict__ hdr, float* __restrict__ hun = nullptr) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const int UW = REV ? m.bUW : m.UW, KBP = REV ? m.bKBP : m.KBP, U = REV ? m.bU : m.U;
  const int KSPLIT = REV ? 1 : m.KSPLIT, KPER = REV ? m.bNKBt : m.KPER, NKBt = REV ? m.bNKBt : m.NKBt;
  const int blk = blockIdx.x;                    // forward: w*(UW*KBP + wrap) + ui*KBP + kbl;  reverse (compact): w*bSW + ui*bNKBt + kbl
  // forward, wrap > 0 (k_g2_fwd_p): the wave's live blocks [nu_w * KBP] are followed by a copy of the first `wrap` of them
  const int FRS = UW * KBP + (REV ? 0 : m.wrap);
  const int w = REV ? blk / m.bSW : blk / FRS;
  int fr = REV ? 0 : blk - w * FRS;
  if (!REV && m.wrap > 0) {
    const int live = (w < U ? (U - w + m.nw - 1) / m.nw : 0) * KBP;
    if (fr >= live && live > 0) fr = (fr - live) % live;
  }
  const int kbl = REV ? (blk % m.bSW) % m.bNKBt : fr % KBP;
  const int ui = REV ? (blk % m.bSW) / m.bNKBt : fr / KBP;
  const int u = w + ui * m.nw;
  xh8 g0, g1;
#pragma unroll
  for (int e = 0; e < 8; ++e) { g0[e] = (_Float16)0.f; g1[e] = (_Float16)0.f; }
  float rsc = 1.f;                               // (REV) this lane's row scale
  if (u < U && kbl < KPER) {
    const int tile = u / KSPLIT, part = u % KSPLIT;
    const int mt = (!REV && m.cin) ? tile : tile / m.N2T;      // (cin: a forward unit is a ROW tile, its column tiles inside)
    if constexpr (REV) {
      // row maximum over i_h (zero outside the row's own gate block): the four k-groups of the lane's row share the walk
      const int rw = 16 * mt + r;
      int jh, a;
      if (m.ng > 1) { const int g = rw / m.Kg, rem = rw - g * m.Kg; jh = rem / m.Rb; a = g * m.Rb + rem % m.Rb; }
      else { jh = rw / m.Rp; a = rw % m.Rp; }
      float mx = 0.f;
      if (jh < m.Jh && a < m.R)
        for (int ih = q; ih < m.Ih; ih += 4) mx = fmaxf(mx, fabsf(Gh[((size_t)ih * m.Jh + jh) * m.R + a]));
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const int ex = g2_expo(mx);
      rsc = ldexpf(1.f, 14 - ex);
      if (kbl == 0 && (tile % m.N2T) == 0 && hun) {
        // ... and the row's L1 norm (behind the inverse scales): max_row L1 x max |dg_t| bounds T2's results, the scale of T1's operand
        float l1 = 0.f;
        if (jh < m.Jh && a < m.R)
          for (int ih = q; ih < m.Ih; ih += 4) l1 += fabsf(Gh[((size_t)ih * m.Jh + jh) * m.R + a]);
        l1 += __shfl_xor(l1, 16);
        l1 += __shfl_xor(l1, 32);
        if (q == 0) { hun[rw] = ldexpf(1.f, ex - 14); hun[m.bM2T * 16 + rw] = l1; }
      }
    }
    // block-diagonal heads: a tile walks its own gate's k-blocks only (gate of a forward tile: its output rows; of a reverse
    // tile: its (g, j_h, a') rows)
    const int gate = m.ng > 1 ? (16 * mt) / (REV ? m.Kg : m.IhG) : 0;
    const int kloc = part * KPER + kbl;
    if (kloc < NKBt) {
      const int kb = gate * NKBt + kloc;
      const int row = 16 * mt + r;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * kb + 8 * q + e;
        int ih, jh, a;
        // (j_h, a) <-> position: ng = 1: j_h * Rp + a;  ng > 1: g * Kg + j_h * Rb + a'  with a = g * Rb + a'
        auto unpos = [&](int pos, int& jh_, int& a_) {
          if (m.ng > 1) { const int g = pos / m.Kg, rem = pos - g * m.Kg; jh_ = rem / m.Rb; a_ = g * m.Rb + rem % m.Rb; }
          else { jh_ = pos / m.Rp; a_ = pos % m.Rp; }
        };
        if (REV) { unpos(row, jh, a); ih = k; }
        else { ih = row; unpos(k, jh, a); }
        float v = 0.f;
        if (ih < m.Ih && jh < m.Jh && a < m.R) v = Gh[((size_t)ih * m.Jh + jh) * m.R + a];
        if constexpr (REV) {
          _Float16 p0, p1;
          split2h(v * rsc, p0, p1);
          g0[e] = p0; g1[e] = p1;
        } else {
          _Float16 p0, p1;                        // 2^(ep[i_h] - ev[a]) Gh (k_g2_diag_a / _b)
          if (ih < m.Ih && jh < m.Jh && a < m.R) v *= ldexpf(1.f, hdr[m.It + 64 + ih] - hdr[m.It + a]);
          split2h(v, p0, p1);
          g0[e] = p0; g1[e] = p1;
        }
      }
    }
  }
  xh8* dst = reinterpret_cast<xh8*>(fs) + (size_t)blk * 2 * 64 + lane;
  dst[0] = g0; dst[64] = g1;
}

// tail fragments
//   forward (fp16 pieces of 2^a Gt, MFMA 16x16x32 layout): lane (m = lane & 15, q = lane >> 4), row (i_t, a) = divmod(16 mt + m, Rp)
//     general: k = 32 kb + 8 q + e <-> j_t, two planes            at ((mt*KB1 + kb)*2 + plane)*64 + lane   (xh8 units)
//     pack8 (J_t <= 8): k-group q holds piece (q & 1) of j_t = e   at mt*64 + lane
//   reverse (two fp16 pieces of the row-scaled tail^T): A[m][k] = Gt[(i_t, a) = divmod(32 kb + 8 q + e, Rp)][j_t = 16 mt + m]
//     at ((mt*bKB1 + kb)*2 + plane)*64 + lane   (xh8 units)
template <bool REV>
__global__ void __launch_bounds__(64) k_g2_tail_frag(G2Mat m, const float* __restrict__ Gt, float* __restrict__ ft,
                                                     const int* __restrict__ hdr, float* __restrict__ tun = nullptr) {
  const int lane = threadIdx.x, mm = lane & 15, kq = lane >> 4;
  if constexpr (REV) {
    // two fp16 pieces of tail^T, every row j_t under its own power-of-two scale (row maximum -> [2^13, 2^14); T1's output rows: the
    // scale is undone on the accumulators, tun[j_t] = its inverse)
    const int kb = blockIdx.x % m.bKB1, mt = blockIdx.x / m.bKB1;
    const int jt = 16 * mt + mm, K = m.It * m.Rp;
    float mx = 0.f;
    if (jt < m.Jt)
      for (int k = kq; k < K; k += 4) {
        const int it = k / m.Rp, a = k - it * m.Rp;
        if (a < m.R) mx = fmaxf(mx, fabsf(Gt[((size_t)it * m.Jt + jt) * m.R + a]));
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const int ex = g2_expo(mx);
    const float tsc = ldexpf(1.f, 14 - ex);
    if (kb == 0 && kq == 0 && tun) tun[jt] = ldexpf(1.f, ex - 14);
    xh8 f0, f1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * kb + 8 * kq + e, it = k / m.Rp, a = k - it * m.Rp;
      float v = 0.f;
      if (it < m.It && a < m.R && jt < m.Jt) v = Gt[((size_t)it * m.Jt + jt) * m.R + a] * tsc;
      _Float16 p0, p1;
      split2h(v, p0, p1);
      f0[e] = p0; f1[e] = p1;
    }
    xh8* dst = reinterpret_cast<xh8*>(ft);
    dst[(size_t)(blockIdx.x * 2) * 64 + lane] = f0;
    dst[(size_t)(blockIdx.x * 2 + 1) * 64 + lane] = f1;
  } else {
    const int kb = blockIdx.x % m.KB1, mt = blockIdx.x / m.KB1;
    const int row = 16 * mt + mm, it = row / m.Rp, a = row % m.Rp;
    const float tsc = (it < m.It && a < m.R) ? ldexpf(1.f, 13 + hdr[it] + hdr[m.It + a]) : 0.f;      // k_g2_diag
    xh8 f0, f1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int jt = m.pack8 ? e : 32 * kb + 8 * kq + e;
      float v = 0.f;
      if (it < m.It && a < m.R && jt < m.Jt) v = Gt[((size_t)it * m.Jt + jt) * m.R + a] * tsc;
      _Float16 p0, p1;
      split2h(v, p0, p1);
      f0[e] = m.pack8 ? ((kq & 1) ? p1 : p0) : p0;
      f1[e] = p1;
    }
    xh8* dst = reinterpret_cast<xh8*>(ft);
    if (m.pack8) dst[(size_t)mt * 64 + lane] = f0;
    else { dst[(size_t)(blockIdx.x * 2) * 64 + lane] = f0; dst[(size_t)(blockIdx.x * 2 + 1) * 64 + lane] = f1; }
  }
}

// biases of both TTLinears -> one gate-interleaved fp32 row [H][4] that the input projection adds to gin:
//   LSTM slots i,g,f,o: b_in + b_hid;   GRU slots r,z: b_in + b_hid, n: b_in only, slot 3: b_hid of the n gate (it sits
//   inside the r * (...) product, gru.py:42-43, so it travels to the gate phase as the fourth gin slot)
template <typename TS>
__global__ void __launch_bounds__(256) k_g2_bias(int cell, int H, const TS* __restrict__ bin, const TS* __restrict__ bhid,
                                                 float* __restrict__ bilv) {
  const int hid = blockIdx.x * blockDim.x + threadIdx.x;
  if (hid >= H) return;
  auto b = [&](const TS* p, int g) { return p ? ld(p, (size_t)g * H + hid) : 0.f; };
  f32x4 v;
  if (cell == TTRNN_LSTM) v = f32x4{b(bin, 0) + b(bhid, 0), b(bin, 2) + b(bhid, 2), b(bin, 1) + b(bhid, 1), b(bin, 3) + b(bhid, 3)};
  else v = f32x4{b(bin, 0) + b(bhid, 0), b(bin, 1) + b(bhid, 1), b(bin, 2), b(bhid, 2)};
  reinterpret_cast<f32x4*>(bilv)[hid] = v;
}

// x[n][in] -> xpad[n][inp] (zero-filled columns): the dense GEMM stages 8 contraction values per load
template <typename TS>
__global__ void __launch_bounds__(256) k_g2_pad_rows(long n_rows, int in, int inp, const TS* __restrict__ x, TS* __restrict__ xp) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_rows * inp) return;
  const long n = e / inp;
  const int j = (int)(e - n * inp);
  st(xp, (size_t)e, j < in ? ld(x, (size_t)n * in + j) : 0.f);
}

__device__ __forceinline__ float round_to(float v, const float*) { return v; }
__device__ __forceinline__ float round_to(float v, const bf16_t*) { return bf16_to_f32(f32_to_bf16(v)); }

// one wave's walk over its share of the head stream: G2_PF blocks in rolling register slots
struct HeadStream {
  const xbf8* base;      // this wave's blocks, + lane
  int total;             // blocks per timestep
  int seq;               // next block to CONSUME (0 .. total-1)
};


// the three-term product of two-piece fp16 operands on two chains (lo: w1x0 + w0x1, hi: w0x0)
__device__ __forceinline__ void split_block_h(const xh8 (&w)[2], const xh8 (&x)[2], f32x4& acc_lo, f32x4& acc_hi) {
  acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], x[0], acc_lo, 0, 0, 0);
  acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], x[0], acc_hi, 0, 0, 0);
  acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], x[1], acc_lo, 0, 0, 0);
}

// ---- forward ----------------------------------------------------------------------------------------------------------------
// P8: stage 1 term-packed into one MFMA per tile (J_t <= 8); otherwise three MFMAs per 32-k block with the next block's
// (or next tile pair's) operands requested before the current MFMAs — two code paths in ONE kernel spilled 347 VGPRs
// NSL: register slots of the head fragments.  G2_PF for the rolling stream and for residents of up to eight k-blocks; sixteen
// (round 4, RES only, eight-wave workgroups, UPT <= 2: 128 fragment VGPRs of the wave's 256) make the head of the reference's
// default benchmark shape (H = 512, r = 8: 256 KB of two-piece fragments = sixteen k-blocks per wave) RESIDENT — streamed it
// was 384 KB per sample-step through a 64 B/clk L2 -> CU path: 8.7 us per step at B = 512 for 1.3 us of matrix-pipe time
// IN1: input_size == 1 as a template parameter (round 4, lesson 44: the runtime flag cost the fused-core kernels 9 - 10 %)
// CIN (round 5, RES only): the unit of a wave is a ROW tile with its N2T <= 4 column tiles inside (G2Mat::cin)
template <int CELL, typename TS, int UPT, bool RES, bool DIAG, bool P8, int NSL = G2_PF, bool IN1 = false, bool CIN = false>
__global__ void __launch_bounds__(G2_NT_MAX) k_g2_fwd(G2Plan P, GinSrc gs, const float* __restrict__ bilv, const TS* __restrict__ h0,
                                                  const TS* __restrict__ c0, const xh8* __restrict__ fs2,
                                                  const float* __restrict__ ft1, const int* __restrict__ hdr,
                                                  TS* __restrict__ out, TS* __restrict__ hT,
                                                  TS* __restrict__ cT, float* __restrict__ reserve) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const G2Mat& m = P.hid;
  _Float16* hb = reinterpret_cast<_Float16*>(smem);                 // stage 1's operand: two fp16 planes of 2^9 h, [2][16*N1T][JS]
  const int HPL = 16 * m.N1T * m.JS;
  _Float16* img = reinterpret_cast<_Float16*>(smem + P.f_hb);     // stage 2's operand: two fp16 planes of 2^a C1
  float* ybuf = reinterpret_cast<float*>(smem + P.f_hb + P.f_img);
  int* s1off = reinterpret_cast<int*>(smem + P.f_hb + P.f_img + P.f_ybuf);
  float* unf = reinterpret_cast<float*>(smem + P.f_hb + P.f_img + P.f_ybuf + P.f_tab);     // [I_h]: 2^-(ep + 13 + 9 + r), then
  float* ung = unf + m.Ih;                                                                   // [I_t]: 2^-eu  (k_g2_diag_a / _b)
  xh8* lt1 = reinterpret_cast<xh8*>(smem + P.f_hb + P.f_img + P.f_ybuf + P.f_tab + P.f_sc); // tail fragments (P.f_t1 > 0)
  const xh8* ft1h = reinterpret_cast<const xh8*>(ft1);
  const int plane = (m.It < 16 * m.N2T ? m.It : 16 * m.N2T) * m.K2S;      // the image holds its I_t real rows

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int NW = m.nw, NT = NW * 64;
  const size_t b = blockIdx.x;
  const int H = P.H, T = P.T, GH = P.G * P.H, upt = P.upt;
  constexpr bool LSTM = CELL == TTRNN_LSTM;
  const float r1sc = ldexpf(1.f, g2_r1_expo(m.Jt));                 // stage-1 sums -> below 2^15 before they are split

  // ---- one-time set-up: zero the padded images, stage-1 store offsets, state --------------------------------------------------
  for (int e = tid; e < (P.f_hb + P.f_img) / 4; e += NT) reinterpret_cast<unsigned*>(smem)[e] = 0u;
  // stage-1 store offsets: lane (c, q) of tile (mt1, nt1) holds rows m1 = 16 mt1 + 4 q .. + 3 = four consecutive ranks a of ONE
  // i_t (Rp is a multiple of 4), column j_h = 16 nt1 + c:  offset = [it * K2S + pos(a)] (table, per tile and q) + j_h * RS
  for (int e = tid; e < m.T1 * 4; e += NT) {
    const int t1 = e >> 2, qq = e & 3;
    const int mt1 = t1 / m.N1T;
    const int m1 = 16 * mt1 + 4 * qq;
    const int it = m1 / m.Rp, a = m1 - it * m.Rp;
    s1off[e] = it < m.It ? it * m.K2S + (m.ng > 1 ? (a / m.Rb) * m.Kg + a % m.Rb : a) : -1;     // gate-major for block-diagonal heads
  }
  // inverse scales of the stage-2 sums, per output row i_h and column i_t (exact powers of two)
  for (int e = tid; e < m.Ih; e += NT) unf[e] = ldexpf(1.f, -(hdr[m.It + 64 + e] + 13 + 9 + g2_r1_expo(m.Jt)));
  for (int e = tid; e < m.It; e += NT) ung[e] = ldexpf(1.f, -hdr[e]);
  const int s1_rs = m.ng > 1 ? m.Rb : m.Rp;
  auto s1o = [&](int t1, int nt1) {
    const int base = s1off[t1 * 4 + q], jh = 16 * nt1 + c;
    return (base >= 0 && jh < m.Jh) ? base + jh * s1_rs : -1;
  };
  // the tail fragments are read by every wave every step: resident in LDS when they fit (else L1 / L2)
  const bool t1_lds = P.f_t1 > 0;
  if (t1_lds)
    for (int e = tid; e < (int)(m.ft1_bytes / 16); e += NT) lt1[e] = ft1h[e];
  float hst[UPT], cst[UPT];
  int hoff[UPT];
  f32x4 gi[UPT], bb[UPT];      // input_size == 1: gi holds the unit row's projection, bb the bias row
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gs.gin);
  const f32x4* bil4 = reinterpret_cast<const f32x4*>(bilv);
  const TS* xs = reinterpret_cast<const TS*>(gs.x);
  constexpr bool in1 = IN1;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    hst[u] = 0.f; cst[u] = 0.f; hoff[u] = 0;
    gi[u] = f32x4{0.f, 0.f, 0.f, 0.f}; bb[u] = gi[u];
    if (u < upt) {
      const int hid = tid + u * NT;
      if (hid < H) {
        hoff[u] = (hid / m.Jt) * m.JS + hid % m.Jt;
        hst[u] = h0 ? ld(h0, b * H + hid) : 0.f;
        cst[u] = (LSTM && c0) ? ld(c0, b * H + hid) : 0.f;
        if (in1) { gi[u] = gin4[hid]; bb[u] = bil4[hid]; }
        else if (T > 0) gi[u] = gin4[(b * T) * H + hid];
      }
    }
  }
  // A caller's h_0 may lie outside (-1, 1), the range the tail scale assumes: the image holds 2^-e0 h (e0 >= 0, per sample,
  // exact) and the stage-2 sums are multiplied back by 2^e0 — at step 0 for an LSTM (h_t = o tanh(c) is inside (-1, 1) from
  // then on), at EVERY step for a GRU (h_t = (1 - z) n + z h_{t-1} only stays below max(1, |h_0|)).
  float h0un = 1.0f, h0sc = 1.0f;
  {
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < UPT; ++u) mx = fmaxf(mx, fabsf(hst[u]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) ybuf[wave] = mx;                        // ybuf: free until the first stage 2
    __syncthreads();
    mx = 0.f;
    for (int w = 0; w < NW; ++w) mx = fmaxf(mx, ybuf[w]);
    int e0 = g2_expo(mx);
    if (e0 < 0) e0 = 0;
    h0sc = ldexpf(1.f, -e0);
    h0un = ldexpf(1.f, e0);
#pragma unroll
    for (int u = 0; u < UPT; ++u)
      if (u < upt && tid + u * NT < H) {
        _Float16 p0, p1;
        split2h(hst[u] * (h0sc * G2_HSC), p0, p1);
        hb[hoff[u]] = p0; hb[HPL + hoff[u]] = p1;
      }
  }
  // head stream of this wave
  const int nu_w = wave < m.U ? (m.U - wave + NW - 1) / NW : 0;
  const int total = nu_w * m.KBP;
  // RES (chosen by the host for the whole launch: UW * KBP <= G2_PF): every wave's share of the head core lives in its
  // register slots for all T steps; otherwise the slots roll over a stream of fragments
  const xh8* sp = fs2 + (size_t)wave * m.UW * m.KBP * 2 * 64 + lane;
  static_assert(NSL == G2_PF || (RES && !DIAG), "more than G2_PF slots: resident fragments only");
  xh8 wbuf[NSL][2];
#pragma unroll
  for (int j = 0; j < NSL; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int e = 0; e < 8; ++e) wbuf[j][p][e] = (_Float16)0.f;
      if (total > 0) wbuf[j][p] = sp[(size_t)j * 2 * 64 + p * 64];
    }
  // ---- per-wave constants of the time loop, computed ONCE (the tier is instruction-issue bound: DESIGN.md 4c) -----------------
  // first stage-1 tile pair of this wave (small and medium shapes have no other)
  const bool s1_has = wave < m.T1;
  const int s1_ta = s1_has ? wave : 0, s1_tb = (s1_has && wave + NW < m.T1) ? wave + NW : s1_ta;
  const int s1_mta = s1_ta / m.N1T, s1_nta = s1_ta - s1_mta * m.N1T;
  const int s1_mtb = s1_tb / m.N1T, s1_ntb = s1_tb - s1_mtb * m.N1T;
  // fragment index (xh8 units) / operand row of a stage-1 tile; pack8: one fragment, operand k-group q reads plane q >> 1
  const int s1_fmul = P8 ? 64 : m.KB1 * 2 * 64;
  const int s1_boff = P8 ? (q >> 1) * HPL : 8 * q;
  const int s1_fa = s1_mta * s1_fmul + lane, s1_fb = s1_mtb * s1_fmul + lane;
  const _Float16* s1_bpa = hb + (16 * s1_nta + c) * m.JS + s1_boff;
  const _Float16* s1_bpb = hb + (16 * s1_ntb + c) * m.JS + s1_boff;
  const int s1_oa = s1_has ? s1o(s1_ta, s1_nta) : -1;
  const int s1_ob = (s1_has && s1_tb != s1_ta) ? s1o(s1_tb, s1_ntb) : -1;
  // the single stage-2 unit of this wave when its head fragments are register-resident
  const int r_tile = wave / m.KSPLIT, r_part = wave - r_tile * m.KSPLIT;
  const int r_mt = r_tile / m.N2T, r_nt = r_tile - r_mt * m.N2T;
  const int r_kloc = r_part * m.KPER;                                   // first k-block inside the tile's own range
  int r_nlive = m.NKBt - r_kloc < m.KPER ? m.NKBt - r_kloc : m.KPER;
  r_nlive = (nu_w > 0 && r_nlive > 0) ? r_nlive : 0;
  const int r_kb0 = (m.ng > 1 ? (16 * r_mt) / m.IhG : 0) * m.NKBt + r_kloc;
  const _Float16* r_brow = img + (16 * r_nt + c < m.It ? 16 * r_nt + c : m.It - 1) * m.K2S + 8 * q + 32 * r_kb0;      // (rows past I_t: dropped)
  const int r_ybase = r_part * GH + (16 * r_mt + 4 * q) * m.It + 16 * r_nt + c;
  int r_ymask = 0;
  f32x4 r_un = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                   // unf / ung are complete
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool live = 16 * r_nt + c < m.It && 16 * r_mt + 4 * q + j < m.Ih;
    r_ymask |= live ? (1 << j) : 0;
    if (live) r_un[j] = unf[16 * r_mt + 4 * q + j] * ung[16 * r_nt + c];
  }
  // CIN: the same constants per column tile of the wave's row tile r_tile
  constexpr int NCI = CIN ? 4 : 1;
  const _Float16* c_brow[NCI];
  int c_ybase[NCI], c_ymask[NCI];
  f32x4 c_un[NCI];
  if constexpr (CIN) {
#pragma unroll
    for (int ct = 0; ct < NCI; ++ct) {
      const int itc = 16 * ct + c;
      c_brow[ct] = img + (itc < m.It ? itc : m.It - 1) * m.K2S + 8 * q + 32 * r_kloc;
      c_ybase[ct] = r_part * GH + (16 * r_tile + 4 * q) * m.It + itc;
      c_ymask[ct] = 0;
      c_un[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool live = ct < m.N2T && itc < m.It && 16 * r_tile + 4 * q + j < m.Ih && nu_w > 0;
        c_ymask[ct] |= live ? (1 << j) : 0;
        if (live) c_un[ct][j] = unf[16 * r_tile + 4 * q + j] * ung[itc];
      }
    }
  }
  XChunk<TS> xq;                    // input_size == 1: 64 timesteps of x per register, refilled a chunk ahead
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
  __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0) here: no wait for the set-up loads inside the time loop
  lds_barrier();
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();

  for (int t = 0; t < T; ++t) {
    const size_t bt = b * T + t;
    // gate inputs of this step: requested now, used after both stages
    // ---- stage 1 (two fp16 pieces): C1 = Gt h, split into the two fp16 planes of stage 2's operand ---------------------------
    // (tail fragments: LDS-resident or from L1 / L2 — two explicit loops: ONE pointer that may be either makes every read a
    // FLAT load, and a flat load can only be waited for with vmcnt(0): it then waits for the `out` stores of the last step)
    auto pair = [&](auto frag, int fa, int fb, const _Float16* bpa, const _Float16* bpb, int offa, int offb) {
      f32x4 acca = f32x4{0.f, 0.f, 0.f, 0.f}, accb = acca;   // two tiles: independent MFMA / split chains
      if constexpr (P8) {                                     // J_t <= 8: x0w0 + x0w1 + x1w0 + x1w1 in one MFMA per tile
        const xh8 wa = frag(fa), wb = frag(fb);
        const xh8 xa = *reinterpret_cast<const xh8*>(bpa), xb = *reinterpret_cast<const xh8*>(bpb);
        acca = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xa, acca, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, xb, accb, 0, 0, 0);
      } else {
        // one tile after the other (each a dependent chain of 3 KB1 MFMAs); the fragments of the NEXT 32-k block are
        // requested before the current block's MFMAs (they may come from L2).  (A flat walk over all of a wave's tiles that
        // also prefetches across tile boundaries was slower wherever the fragments sit in LDS — d = 2, H = 256: 1.15 -> 1.50 ms
        // — and did not help the one shape that streams them, cfg5's, either.)
        auto one = [&](int f, const _Float16* bp, f32x4& acc) {
          auto blk = [&](const xh8& w0, const xh8& w1, int kb) {
            const xh8 x0 = *reinterpret_cast<const xh8*>(bp + 32 * kb), x1 = *reinterpret_cast<const xh8*>(bp + HPL + 32 * kb);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x0, acc, 0, 0, 0);
          };
          xh8 a0 = frag(f), a1 = frag(f + 64);               // two named register sets (a runtime-indexed array would
          for (int kb = 0; kb < m.KB1; kb += 2) {               // live in scratch)
            const int k1 = kb + 1 < m.KB1 ? kb + 1 : kb;
            const xh8 b0 = frag(f + (k1 * 2) * 64), b1 = frag(f + (k1 * 2 + 1) * 64);
            blk(a0, a1, kb);
            if (kb + 1 < m.KB1) {
              const int k2 = kb + 2 < m.KB1 ? kb + 2 : kb + 1;
              a0 = frag(f + (k2 * 2) * 64); a1 = frag(f + (k2 * 2 + 1) * 64);
              blk(b0, b1, kb + 1);
            }
          }
        };
        one(fa, bpa, acca);
        if (fb != fa) one(fb, bpb, accb);        // wave-uniform (the second tile repeats the first when there is none): an
                                                 // MFMA under a lane-divergent branch would read masked-off operand loads
      }
      if (offa >= 0) store_split4_h(img, plane, offa, acca * r1sc);     // 2^r: below 2^15
      if (offb >= 0) store_split4_h(img, plane, offb, accb * r1sc);
    };
    auto stage1 = [&](auto frag) {
      if (s1_has) pair(frag, s1_fa, s1_fb, s1_bpa, s1_bpb, s1_oa, s1_ob);
      for (int t1 = wave + 2 * NW; t1 < m.T1; t1 += 2 * NW) {  // larger shapes: further pairs, decoded on the fly
        const int t1b = t1 + NW < m.T1 ? t1 + NW : t1;          // (the second tile repeats the first when there is none)
        const int mta = t1 / m.N1T, nta = t1 - mta * m.N1T;
        const int mtb = t1b / m.N1T, ntb = t1b - mtb * m.N1T;
        pair(frag, mta * s1_fmul + lane, mtb * s1_fmul + lane, hb + (16 * nta + c) * m.JS + s1_boff,
             hb + (16 * ntb + c) * m.JS + s1_boff, s1o(t1, nta), t1b != t1 ? s1o(t1b, ntb) : -1);
      }
    };
    if (t1_lds) stage1([&](int i) { return lt1[i]; });
    else stage1([&](int i) { return ft1h[i]; });
    TT_STAMP(0)
    lds_barrier();
    TT_STAMP(1)
    // ---- stage 2 (two fp16 pieces per operand) ---------------------------------------------------------------------------------
    // Two instantiations of the same body: RESIDENT (the wave's whole share of the head core sits in its register slots:
    // NO vector-memory instruction in the loop) and streaming (every slot is refilled unconditionally right after its use,
    // padding blocks included).  A CONDITIONAL refill makes hipcc guard every block with s_waitcnt vmcnt(0), which also
    // waits for the `out` store of the previous step: measured 3 300 instead of ~700 cycles for four blocks.
    auto stage2 = [&]() {
      int seq = 0;
      for (int ui = 0; ui < nu_w; ++ui) {
        const int u = wave + ui * NW;
        const int tile = u / m.KSPLIT, part = u - tile * m.KSPLIT;
        const int mt = tile / m.N2T, nt = tile - mt * m.N2T;
        const int kloc0 = part * m.KPER;                                              // inside the tile's own k range
        const int kbase = (m.ng > 1 ? (16 * mt) / m.IhG : 0) * m.NKBt;                // block-diagonal heads: the gate's range
        const _Float16* brow = img + (16 * nt + c < m.It ? 16 * nt + c : m.It - 1) * m.K2S + 8 * q + 32 * kbase;
        f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
        // operand fragments of the NEXT block are requested before the current block's MFMAs are issued
        xh8 bf[2][2];
        {
          const int kbc = kloc0 < m.NKBt ? kloc0 : m.NKBt - 1;
#pragma unroll
          for (int p = 0; p < 2; ++p) bf[0][p] = *reinterpret_cast<const xh8*>(brow + p * plane + 32 * kbc);
        }
        for (int kbl = 0; kbl < m.KBP; kbl += G2_PF) {
#pragma unroll
          for (int j = 0; j < G2_PF; ++j) {
            const int kb = kloc0 + kbl + j;
            if (kbl + j + 1 < m.KPER && kb + 1 < m.NKBt) {     // the next block is live (padding blocks are never read)
#pragma unroll
              for (int p = 0; p < 2; ++p) bf[(j + 1) & 1][p] = *reinterpret_cast<const xh8*>(brow + p * plane + 32 * (kb + 1));
            }
            if (kbl + j < m.KPER && kb < m.NKBt) split_block_h(wbuf[j], bf[j & 1], acc_lo, acc_hi);
            {
              int nxt = seq + G2_PF;                     // the block G2_PF ahead (wraps into step t+1); total >= G2_PF
              nxt -= nxt >= total ? total : 0;
#pragma unroll
              for (int p = 0; p < 2; ++p) wbuf[j][p] = sp[(size_t)nxt * 2 * 64 + p * 64];
              ++seq;
            }
          }
        }
        const f32x4 acc = acc_hi + acc_lo;
        const int itc = 16 * nt + c;
        if (itc < m.It) {
          const float ug = ung[itc];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ih = 16 * mt + 4 * q + j;
            if (ih < m.Ih) ybuf[part * GH + ih * m.It + itc] = acc[j] * (unf[ih] * ug);
          }
        }
      }
    };
    if constexpr (RES && CIN) {
      // one ROW tile per wave, its column tiles inside: every resident block is multiplied against the N2T column tiles' operands
      static_assert(NSL == G2_PF, "columns inside: the eight-slot resident form");
      f32x4 lo[NCI], hi[NCI];
#pragma unroll
      for (int ct = 0; ct < NCI; ++ct) { lo[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; hi[ct] = lo[ct]; }
#pragma unroll
      for (int j = 0; j < NSL; ++j) {
        if (j < r_nlive) {
#pragma unroll
          for (int ct = 0; ct < NCI; ++ct)
            if (ct < m.N2T) {
              xh8 bfc[2];
#pragma unroll
              for (int p = 0; p < 2; ++p) bfc[p] = *reinterpret_cast<const xh8*>(c_brow[ct] + p * plane + 32 * j);
              split_block_h(wbuf[j], bfc, lo[ct], hi[ct]);
            }
        }
      }
#pragma unroll
      for (int ct = 0; ct < NCI; ++ct) {
        const f32x4 acc = (hi[ct] + lo[ct]) * c_un[ct];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c_ymask[ct] & (1 << j)) ybuf[c_ybase[ct] + j * m.It] = acc[j];
      }
    } else if constexpr (RES) {
      // one unit, at most G2_PF blocks, every constant hoisted: fragment reads at immediate offsets, MFMAs, four stores
      if (r_nlive == NSL) {
        // every slot live (the common case of the shapes that are resident at all): no per-block conditions — sixteen pairs of
        // scalar compare + branch per step cut the unrolled body into basic blocks the scheduler cannot move the operand reads
        // across (the same effect as the runtime input_size == 1 flag in the fused-core kernels)
        f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
        xh8 bf[3][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          bf[0][p] = *reinterpret_cast<const xh8*>(r_brow + p * plane);
          bf[1][p] = *reinterpret_cast<const xh8*>(r_brow + p * plane + 32);
        }
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
          if (j + 2 < NSL) {
#pragma unroll
            for (int p = 0; p < 2; ++p) bf[(j + 2) % 3][p] = *reinterpret_cast<const xh8*>(r_brow + p * plane + 32 * (j + 2));
          }
          split_block_h(wbuf[j], bf[j % 3], acc_lo, acc_hi);
        }
        const f32x4 acc = (acc_hi + acc_lo) * r_un;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (r_ymask & (1 << j)) ybuf[r_ybase + j * m.It] = acc[j];
      } else if (r_nlive > 0) {
        f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
        xh8 bf[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) bf[0][p] = *reinterpret_cast<const xh8*>(r_brow + p * plane);
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
          if (j < r_nlive) {
            if (j + 1 < r_nlive) {
#pragma unroll
              for (int p = 0; p < 2; ++p) bf[(j + 1) & 1][p] = *reinterpret_cast<const xh8*>(r_brow + p * plane + 32 * (j + 1));
            }
            split_block_h(wbuf[j], bf[j & 1], acc_lo, acc_hi);
          }
        }
        const f32x4 acc = (acc_hi + acc_lo) * r_un;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (r_ymask & (1 << j)) ybuf[r_ybase + j * m.It] = acc[j];
      }
    } else {
      stage2();
    }
    TT_STAMP(2)
    lds_barrier();
    TT_STAMP(3)
    // ---- gates + state (lstm.py:26-32 / gru.py:38-44) --------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      if (u < upt) {
        const int hid = tid + u * NT;
        if (hid < H) {
          float y[4] = {0.f, 0.f, 0.f, 0.f};
          for (int pt = 0; pt < m.KSPLIT; pt += 2) {           // two partial sums per trip: all reads issued before the adds
            const int p1 = pt + 1 < m.KSPLIT ? pt + 1 : pt;
            const float k1 = pt + 1 < m.KSPLIT ? 1.0f : 0.0f;
            float ya[4], yb[4];
#pragma unroll
            for (int g = 0; g < (LSTM ? 4 : 3); ++g) { ya[g] = ybuf[pt * GH + g * H + hid]; yb[g] = ybuf[p1 * GH + g * H + hid]; }
#pragma unroll
            for (int g = 0; g < (LSTM ? 4 : 3); ++g) y[g] += ya[g] + k1 * yb[g];
          }
          const float un_t = (t == 0 || !LSTM) ? h0un : 1.0f;  // (the diagonal scales were undone where the sums were stored)
#pragma unroll
          for (int g = 0; g < (LSTM ? 4 : 3); ++g) y[g] *= un_t;         // exact: a power of two
          f32x4 g4 = gi[u];
          if (in1) g4 = bb[u] + xq.at(t) * gi[u];
          float hy;
          if (LSTM) {                                    // gin slots i,g,f,o
            const float ig = fsigmoid(y[0] + g4[0]);
            const float fg = fsigmoid(y[1] + g4[2]);
            const float gg = ftanh(y[2] + g4[1]);
            const float og = fsigmoid(y[3] + g4[3]);
            const float cy = fg * cst[u] + ig * gg;
            hy = og * ftanh(cy);
            cst[u] = cy;
            if (reserve) {
              *reinterpret_cast<f32x4*>(reserve + res_gate(bt, H, hid)) = f32x4{ig, gg, fg, og};
              reserve[res_cell((size_t)P.B * T, bt, H, hid)] = cy;
            }
          } else {                                       // gin slots r,z,n, b_hid of n
            const float hn = y[2] + g4[3];                 // slot 3: b_hid of the n gate (k_g2_bias)
            const float rg = fsigmoid(y[0] + g4[0]);
            const float zg = fsigmoid(y[1] + g4[1]);
            const float ng = ftanh(g4[2] + rg * hn);
            hy = (1.0f - zg) * ng + zg * hst[u];
            if (reserve) *reinterpret_cast<f32x4*>(reserve + (bt * H + hid) * 4) = f32x4{rg, zg, ng, hn};
          }
          hy = round_to(hy, out);                        // the stored value is what the next step and the next layer see
          if (out) st(out, bt * H + hid, hy);            // out == NULL: final state only (ttrnn_rnn_out_optional)
          hst[u] = hy;
          {
            _Float16 p0, p1;
            split2h(hy * (LSTM ? G2_HSC : G2_HSC * h0sc), p0, p1);
            hb[hoff[u]] = p0; hb[HPL + hoff[u]] = p1;
          }
        }
      }
    }
    // gate inputs of the NEXT step: requested after the last unit's use of the registers and after the step's stores, used a whole step
    // later.  No lane-divergent region and no other unit's math behind the request (lesson 53: inside `if (hid < H)`, with two
    // units per thread, the second unit's first use of its own gin waited — vmcnt(0) — for the first unit's request just issued)
    if (!in1) {
      const size_t bn = (t + 1 < T ? bt + 1 : bt) * H;
#pragma unroll
      for (int u = 0; u < UPT; ++u) {
        const int hid0 = tid + u * NT;
        gi[u] = gin4[bn + ((u < upt && hid0 < H) ? hid0 : 0)];
      }
    }
    if (in1) xq.advance(xs, b * T, T, t, lane);
    TT_STAMP(4)
    lds_barrier();
    TT_STAMP(5)
  }
  if constexpr (DIAG) {
    if (lane == 0 && reserve && b < 4) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(reserve) + (b * G2_NW_MAX + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = seg[i];
    }
  }
#pragma unroll
  for (int u = 0; u < UPT; ++u)
    if (u < upt) {
      const int hid = tid + u * NT;
      if (hid < H) {
        if (hT) st(hT, b * H + hid, hst[u]);
        if (LSTM && cT) st(cT, b * H + hid, cst[u]);
      }
    }
}

// ---- forward, TWO samples per workgroup (round 5) --------------------------------------------------------------------------------
// Shapes whose head does not fit the register slots stream it from L2 every step — per WORKGROUP.  With I_t <= 8 a sample fills
// half of stage 2's sixteen columns: this kernel gives the other half to a second sample (NCT = 1; I_t <= 16: a second column tile,
// NCT = 2 — two accumulator pairs per streamed block), so every streamed block feeds two
// (`--naive_tt` at H = 512: 512 KB per step and workgroup; B = 512 ran as two co-resident four-wave workgroups per CU pulling
// 1 MB per step through one L2 port).  Eight waves, the plan of the eight-wave kernel (G2Plan::pair), J_t <= 8 (stage 1 term-packed
// into one MFMA per tile).  Layout changes against k_g2_fwd: the h image holds its eight live k-slots per row only ([2][NS][16 N1T][8]);
// the stage-2 operand image has rows (sample, i_t); stage 1 walks (m tile, sample, column tile) — a wave's contiguous share, its
// tail fragments in registers, four column tiles at a time; ybuf is [NS][KSPLIT][G H].  An odd batch's last workgroup computes
// its second sample on a copy of the first and stores nothing for it.
constexpr int G2P_NS = G2_PAIR_NS;
constexpr int G2P_JS = G2_PAIR_JS;
constexpr int G2P_MAXF = G2_PAIR_MAXF;

template <int CELL, typename TS, int UPT, bool IN1, bool DIAG, int NCT = 1>
__global__ void __launch_bounds__(G2_NT_MAX) k_g2_fwd_p(G2Plan P, GinSrc gs, const float* __restrict__ bilv, const TS* __restrict__ h0,
                                                    const TS* __restrict__ c0, const xh8* __restrict__ fs2,
                                                    const float* __restrict__ ft1, const int* __restrict__ hdr,
                                                    TS* __restrict__ out, TS* __restrict__ hT,
                                                    TS* __restrict__ cT, float* __restrict__ reserve) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NS = G2P_NS, JS = G2P_JS, NW = G2_NW_MAX, NT = NW * 64;
  constexpr bool LSTM = CELL == TTRNN_LSTM;
  constexpr int NG = LSTM ? 4 : 3;
  const G2Mat& m = P.hid;
  _Float16* hb = reinterpret_cast<_Float16*>(smem);                 // stage 1's operand: two fp16 planes of 2^9 h, [2][NS][16*N1T][JS]
  const int HSS = 16 * m.N1T * JS, HPL = NS * HSS;
  _Float16* img = reinterpret_cast<_Float16*>(smem + P.f_hb);     // stage 2's operand: two fp16 planes of 2^a C1, rows (sample, i_t)
  float* ybuf = reinterpret_cast<float*>(smem + P.f_hb + P.f_img);
  int* s1off = reinterpret_cast<int*>(smem + P.f_hb + P.f_img + P.f_ybuf);                   // [M1T][4]
  float* unf = reinterpret_cast<float*>(smem + P.f_hb + P.f_img + P.f_ybuf + P.f_tab);     // [I_h]: 2^-(ep + 13 + 9 + r), then
  float* ung = unf + m.Ih;                                                                   // [I_t]: 2^-eu  (k_g2_diag_a / _b)
  const xh8* ft1h = reinterpret_cast<const xh8*>(ft1);
  const int rows2 = NS * m.It;
  const int plane = rows2 * m.K2S;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int H = P.H, T = P.T, GH = P.G * P.H, upt = P.upt;
  const float r1sc = ldexpf(1.f, g2_r1_expo(m.Jt));                 // stage-1 sums -> below 2^15 before they are split
  size_t bsm[NS];
  bool live[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const size_t bi = (size_t)blockIdx.x * NS + s;
    live[s] = bi < (size_t)P.B;
    bsm[s] = live[s] ? bi : (size_t)blockIdx.x * NS;
  }

  // ---- one-time set-up ----------------------------------------------------------------------------------------------------------
  for (int e = tid; e < (P.f_hb + P.f_img) / 4; e += NT) reinterpret_cast<unsigned*>(smem)[e] = 0u;
  // stage-1 store offsets: lane (c, q) of m tile mt1 holds rows m1 = 16 mt1 + 4 q .. + 3 = four consecutive ranks a of ONE i_t;
  // the column part (j_h = 16 nt1 + c, the sample's rows) is added per column tile
  for (int e = tid; e < m.M1T * 4; e += NT) {
    const int m1 = 16 * (e >> 2) + 4 * (e & 3);
    const int it = m1 / m.Rp, a = m1 - it * m.Rp;
    s1off[e] = it < m.It ? it * m.K2S + (m.ng > 1 ? (a / m.Rb) * m.Kg + a % m.Rb : a) : -1;     // gate-major for block-diagonal heads
  }
  for (int e = tid; e < m.Ih; e += NT) unf[e] = ldexpf(1.f, -(hdr[m.It + 64 + e] + 13 + 9 + g2_r1_expo(m.Jt)));
  for (int e = tid; e < m.It; e += NT) ung[e] = ldexpf(1.f, -hdr[e]);
  const int s1_rs = m.ng > 1 ? m.Rb : m.Rp;
  float hst[NS][UPT], cst[NS][UPT];
  int hoff[UPT];
  f32x4 gi[NS][UPT], bb[UPT];      // input_size == 1: gi[0] holds the unit row's projection, bb the bias row
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gs.gin);
  const f32x4* bil4 = reinterpret_cast<const f32x4*>(bilv);
  const TS* xs = reinterpret_cast<const TS*>(gs.x);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    hoff[u] = 0; bb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int hid = tid + u * NT;
    const bool on = u < upt && hid < H;
    if (on) hoff[u] = (hid / m.Jt) * JS + hid % m.Jt;
    if (IN1 && on) bb[u] = bil4[hid];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      hst[s][u] = 0.f; cst[s][u] = 0.f; gi[s][u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (on) {
        hst[s][u] = h0 ? ld(h0, bsm[s] * H + hid) : 0.f;
        cst[s][u] = (LSTM && c0) ? ld(c0, bsm[s] * H + hid) : 0.f;
        if (IN1) gi[s][u] = gin4[hid];
        else if (T > 0) gi[s][u] = gin4[(bsm[s] * T) * H + hid];
      }
    }
  }
  // a caller's h_0 outside (-1, 1): per sample, as in k_g2_fwd
  float h0un[NS], h0sc[NS];
  {
    float mx[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      mx[s] = 0.f;
#pragma unroll
      for (int u = 0; u < UPT; ++u) mx[s] = fmaxf(mx[s], fabsf(hst[s][u]));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx[s] = fmaxf(mx[s], __shfl_xor(mx[s], o));
      if (lane == 0) ybuf[s * NW + wave] = mx[s];                        // ybuf: free until the first stage 2
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float v = 0.f;
      for (int w = 0; w < NW; ++w) v = fmaxf(v, ybuf[s * NW + w]);
      int e0 = g2_expo(v);
      if (e0 < 0) e0 = 0;
      h0sc[s] = ldexpf(1.f, -e0);
      h0un[s] = ldexpf(1.f, e0);
#pragma unroll
      for (int u = 0; u < UPT; ++u)
        if (u < upt && tid + u * NT < H) {
          _Float16 p0, p1;
          split2h(hst[s][u] * (h0sc[s] * G2_HSC), p0, p1);
          hb[s * HSS + hoff[u]] = p0; hb[HPL + s * HSS + hoff[u]] = p1;
        }
    }
  }
  // head stream of this wave: G2_PF rolling register slots; its region ends in a copy of its first G2_PF blocks (G2Mat::wrap)
  const int nu_w = wave < m.U ? (m.U - wave + NW - 1) / NW : 0;
  const xh8* sp = fs2 + (size_t)wave * (m.UW * m.KBP + m.wrap) * 2 * 64;      // wave-uniform: scalar base + lane offset addressing
  xh8 wbuf[G2_PF][2];
#pragma unroll
  for (int j = 0; j < G2_PF; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int e = 0; e < 8; ++e) wbuf[j][p][e] = (_Float16)0.f;
      if (nu_w > 0) wbuf[j][p] = sp[j * 2 * 64 + p * 64 + lane];
    }
  // stage 1: this wave's contiguous share of the (m tile, sample, column tile) walk — at most G2P_MAXF m tiles (g2_plan_pair),
  // their tail fragments and store offsets in registers for all T steps
  const int NC = NS * m.N1T;
  const int tpw = (NC * m.M1T + NW - 1) / NW;
  const int tt0 = wave * tpw, tt1 = tt0 + tpw < NC * m.M1T ? tt0 + tpw : NC * m.M1T;
  const int mtA = tt0 < tt1 ? tt0 / NC : 1, mtB = tt0 < tt1 ? (tt1 - 1) / NC : 0;      // (an empty share: mtA > mtB)
  const int cnA = tt0 - mtA * NC, cnB = tt1 - mtB * NC;                                 // first column of m tile mtA, end column of mtB
  const int sA = cnA / m.N1T, ntA = cnA - sA * m.N1T;
  xh8 tf[G2P_MAXF];
  int so[G2P_MAXF];
  const bool s1_full = m.It * m.Rp == 16 * m.M1T && m.Jh == 16 * m.N1T;
#pragma unroll
  for (int f = 0; f < G2P_MAXF; ++f) {
    const int mt = mtA + f < m.M1T ? mtA + f : m.M1T - 1;
    tf[f] = ft1h[mt * 64 + lane];
    so[f] = s1off[mt * 4 + q];
  }
  // stage 2: the lane's column = (sample, i_t)
  bool col_on[NCT];
  int crow[NCT], col_y[NCT];
  float col_ug[NCT];
  int col_it[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int r = 16 * ct + c;
    col_on[ct] = r < rows2;
    const int cs_ = col_on[ct] ? r / m.It : 0;
    col_it[ct] = col_on[ct] ? r - cs_ * m.It : 0;
    crow[ct] = col_on[ct] ? r : rows2 - 1;
    col_y[ct] = cs_ * m.KSPLIT * GH + col_it[ct];
  }
  __syncthreads();                                   // unf / ung are complete
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) col_ug[ct] = ung[col_it[ct]];
  XChunk<TS> xq[NS];                 // input_size == 1: 64 timesteps of x per register, refilled a chunk ahead
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    xq[s].cur = 0.f; xq[s].nxt = 0.f;
    if (IN1) xq[s].init(xs, bsm[s] * T, T, lane);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0) here: no wait for the set-up loads inside the time loop
  lds_barrier();
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();

  for (int t = 0; t < T; ++t) {
    // ---- stage 1: C1 = Gt h of both samples, split into the two fp16 planes of stage 2's operand ----------------------------------
    // (no vector-memory instruction: a load here would wait, in issue order, for the head blocks requested behind stage 2)
#pragma unroll
    for (int f = 0; f < G2P_MAXF; ++f) {
      const int mt = mtA + f;
      if (mt <= mtB) {
        int cn = f == 0 ? cnA : 0;
        int cs = f == 0 ? sA : 0, cnt = f == 0 ? ntA : 0;
        const int cend = mt == mtB ? cnB : NC;
        const _Float16* bp = hb + (q >> 1) * HPL + c * JS;
        auto coff_of = [&]() { const int jh = 16 * cnt + c; return jh < m.Jh ? jh * s1_rs + cs * m.It * m.K2S : -1; };
        auto next_col = [&]() { ++cn; if (++cnt == m.N1T) { cnt = 0; ++cs; } };
        // four tiles at a time: operand reads, MFMAs and splitting passes of independent chains (one tile after the other was a
        // dependent LDS read -> MFMA -> fifteen VALU -> LDS write per tile: 6 500 cycles per step for sixteen tiles per wave)
        while (cn + 4 <= cend) {
          xh8 xv[4];
          int co[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            xv[i] = *reinterpret_cast<const xh8*>(bp + 16 * cn * JS);
            co[i] = coff_of();
            next_col();
          }
          f32x4 av[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) av[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tf[f], xv[i], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          if (s1_full) {                                 // every lane of every tile live (wave-uniform): no exec masking
#pragma unroll
            for (int i = 0; i < 4; ++i) store_split4_h(img, plane, so[f] + co[i], av[i] * r1sc);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (so[f] >= 0 && co[i] >= 0) store_split4_h(img, plane, so[f] + co[i], av[i] * r1sc);
          }
        }
        while (cn < cend) {
          const xh8 xa = *reinterpret_cast<const xh8*>(bp + 16 * cn * JS);
          const int ca = coff_of();
          next_col();
          f32x4 acca = f32x4{0.f, 0.f, 0.f, 0.f};
          acca = __builtin_amdgcn_mfma_f32_16x16x32_f16(tf[f], xa, acca, 0, 0, 0);
          if (so[f] >= 0 && ca >= 0) store_split4_h(img, plane, so[f] + ca, acca * r1sc);
        }
      }
    }
    TT_STAMP(0)
    lds_barrier();
    TT_STAMP(1)
    // ---- stage 2: the streamed head against sixteen columns = two samples' i_t ---------------------------------------------------
    // Every slot is refilled right after its use with the block G2_PF positions ahead, unconditionally (k_g2_fwd) — by a LINEAR
    // walk: the wave's stream ends in a copy of its first G2_PF blocks (G2Mat::wrap), so the refill needs no wrap-around arithmetic
    // (a 64-bit select per block before); the operand ring holds the image rows of the next blocks.
    {
      const xh8* sq = sp + (size_t)G2_PF * 2 * 64;
      for (int ui = 0; ui < nu_w; ++ui) {
        const int u = wave + ui * NW;
        const int tile = u / m.KSPLIT, part = u - tile * m.KSPLIT;
        const int kloc0 = part * m.KPER;                                              // inside the tile's own k range
        const int kbase = (m.ng > 1 ? (16 * tile) / m.IhG : 0) * m.NKBt;              // block-diagonal heads: the gate's range
        int nlive = m.NKBt - kloc0 < m.KPER ? m.NKBt - kloc0 : m.KPER;                // live blocks (the rest of KBP is padding)
        nlive = nlive > 0 ? nlive : 1;                                                // (no part is empty: g2_split)
        const _Float16* brow[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) brow[ct] = img + crow[ct] * m.K2S + 8 * q + 32 * (kbase + kloc0);
        f32x4 acc_lo[NCT], acc_hi[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { acc_lo[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_hi[ct] = acc_lo[ct]; }
        constexpr int RD = NCT == 1 ? 4 : 2;             // operand ring (its depth divides the group of eight)
        xh8 bf[RD][NCT][2];
#pragma unroll
        for (int i = 0; i < RD; ++i) {
          const int kc = i < nlive ? i : nlive - 1;
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int p = 0; p < 2; ++p) bf[i][ct][p] = *reinterpret_cast<const xh8*>(brow[ct] + p * plane + 32 * kc);
        }
        // ONE path, no per-block condition: the padding blocks of the stream are zero fragments (k_g2_head_frag) and multiply
        // the unit's last live image block (finite values: exact zeros are added)
        for (int kbl = 0; kbl < m.KBP; kbl += G2_PF) {
#pragma unroll
          for (int j = 0; j < G2_PF; ++j) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) split_block_h(wbuf[j], bf[j % RD][ct], acc_lo[ct], acc_hi[ct]);
            const int k3 = kbl + j + RD < nlive ? kbl + j + RD : nlive - 1;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
              for (int p = 0; p < 2; ++p) bf[j % RD][ct][p] = *reinterpret_cast<const xh8*>(brow[ct] + p * plane + 32 * k3);
#pragma unroll
            for (int p = 0; p < 2; ++p) wbuf[j][p] = sq[j * 2 * 64 + p * 64 + lane];
          }
          sq += (size_t)G2_PF * 2 * 64;
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const f32x4 acc = acc_hi[ct] + acc_lo[ct];
          if (col_on[ct]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ih = 16 * tile + 4 * q + j;
              if (ih < m.Ih) ybuf[col_y[ct] + part * GH + ih * m.It] = acc[j] * (unf[ih] * col_ug[ct]);
            }
          }
        }
      }
    }
    TT_STAMP(2)
    lds_barrier();
    TT_STAMP(3)
    // ---- gates + state (lstm.py:26-32 / gru.py:38-44), both samples --------------------------------------------------------------
    // The gate inputs were requested a step ago by loads the compiler does not see (below): what is still in flight in front of
    // the gate phase is the head stream's prefetch for the next step — the LAST 2 G2_PF vector-memory operations issued — and
    // vector-memory operations complete in issue order, so "all but the youngest 2 G2_PF" is exactly "the gate inputs have arrived".
    // (hipcc's own wait for the same registers was vmcnt(1): the gate phase sat out the whole prefetch burst, ~2 000 cycles of a
    // 19 000-cycle step.)  A wave without a stream (nu_w == 0) has nothing younger than its gate inputs in flight: vmcnt(0).
    if constexpr (!IN1) {
      if (nu_w > 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int u = 0; u < UPT; ++u) asm volatile("s_waitcnt vmcnt(16)" : "+v"(gi[s][u])::"memory");
      } else {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int u = 0; u < UPT; ++u) asm volatile("s_waitcnt vmcnt(0)" : "+v"(gi[s][u])::"memory");
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const size_t bt = bsm[s] * T + t;
#pragma unroll
      for (int u = 0; u < UPT; ++u) {
        const int hid = tid + u * NT;
        if (u < upt && hid < H) {
          float y[4] = {0.f, 0.f, 0.f, 0.f};
          for (int pt = 0; pt < m.KSPLIT; ++pt) {
#pragma unroll
            for (int g = 0; g < NG; ++g) y[g] += ybuf[(s * m.KSPLIT + pt) * GH + g * H + hid];
          }
          const float un_t = (t == 0 || !LSTM) ? h0un[s] : 1.0f;
#pragma unroll
          for (int g = 0; g < NG; ++g) y[g] *= un_t;         // exact: a power of two
          f32x4 g4 = gi[s][u];
          if (IN1) g4 = bb[u] + xq[s].at(t) * gi[0][u];
          float hy;
          if (LSTM) {                                    // gin slots i,g,f,o
            const float ig = fsigmoid(y[0] + g4[0]);
            const float fg = fsigmoid(y[1] + g4[2]);
            const float gg = ftanh(y[2] + g4[1]);
            const float og = fsigmoid(y[3] + g4[3]);
            const float cy = fg * cst[s][u] + ig * gg;
            hy = og * ftanh(cy);
            cst[s][u] = cy;
            if (reserve && live[s]) {
              *reinterpret_cast<f32x4*>(reserve + res_gate(bt, H, hid)) = f32x4{ig, gg, fg, og};
              reserve[res_cell((size_t)P.B * T, bt, H, hid)] = cy;
            }
          } else {                                       // gin slots r,z,n, b_hid of n
            const float hn = y[2] + g4[3];
            const float rg = fsigmoid(y[0] + g4[0]);
            const float zg = fsigmoid(y[1] + g4[1]);
            const float ng = ftanh(g4[2] + rg * hn);
            hy = (1.0f - zg) * ng + zg * hst[s][u];
            if (reserve && live[s]) *reinterpret_cast<f32x4*>(reserve + (bt * H + hid) * 4) = f32x4{rg, zg, ng, hn};
          }
          hy = round_to(hy, out);                        // the stored value is what the next step and the next layer see
          if (out && live[s]) st(out, bt * H + hid, hy);
          hst[s][u] = hy;
          {
            _Float16 p0, p1;
            split2h(hy * (LSTM ? G2_HSC : G2_HSC * h0sc[s]), p0, p1);
            hb[s * HSS + hoff[u]] = p0; hb[HPL + s * HSS + hoff[u]] = p1;
          }
        }
      }
    }
    // gate inputs of the NEXT step: requested outside any lane-divergent region (lesson 53), used a whole step later
    if (!IN1) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const size_t bn = (bsm[s] * T + (t + 1 < T ? t + 1 : t)) * H;
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
          const int hid0 = tid + u * NT;
          // (opaque to hipcc's s_waitcnt insertion: the wait is the explicit one at the head of the gate phase)
          const f32x4* gp = gin4 + bn + ((u < upt && hid0 < H) ? hid0 : 0);
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gi[s][u]) : "v"(gp) : "memory");
        }
      }
    }
    if (IN1) {
#pragma unroll
      for (int s = 0; s < NS; ++s) xq[s].advance(xs, bsm[s] * T, T, t, lane);
    }
    TT_STAMP(4)
    lds_barrier();
    TT_STAMP(5)
  }
  if constexpr (DIAG) {
    if (lane == 0 && reserve && blockIdx.x < 4) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(reserve) + (blockIdx.x * G2_NW_MAX + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = seg[i];
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int hid = tid + u * NT;
      if (u < upt && hid < H && live[s]) {
        if (hT) st(hT, bsm[s] * H + hid, hst[s][u]);
        if (LSTM && cT) st(cT, bsm[s] * H + hid, cst[s][u]);
      }
    }
}

// maximum over the 64 lanes, in every lane: rotations inside the 16-lane rows (DPP), then one lane of each row — six ds_bpermute
// round trips (__shfl_xor) were 600 cycles of the gate phase
template <int N>
__device__ __forceinline__ float g2_row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float g2_wave_max(float v) {
  v = fmaxf(v, g2_row_ror<1>(v));
  v = fmaxf(v, g2_row_ror<2>(v));
  v = fmaxf(v, g2_row_ror<4>(v));
  v = fmaxf(v, g2_row_ror<8>(v));
  const int i = __float_as_int(v);
  const float a = __int_as_float(__builtin_amdgcn_readlane(i, 0)), b = __int_as_float(__builtin_amdgcn_readlane(i, 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(i, 32)), d = __int_as_float(__builtin_amdgcn_readlane(i, 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
// the step's scale 2^(14 - e) for values < 2^e, from the bits of the maximum (biased exponent clamped to [27, 227]: the scale and its
// inverse `un` stay normal floats; a zero maximum scales by 2^114 — zeros stay zeros)
__device__ __forceinline__ float g2_step_scale(float mx, float& un) {
  int eb = (int)(__float_as_uint(mx) >> 23);
  eb = eb < 27 ? 27 : (eb > 227 ? 227 : eb);
  un = __uint_as_float((unsigned)(eb - 13) << 23);
  return __uint_as_float((unsigned)(267 - eb) << 23);
}

// ---- reverse time -----------------------------------------------------------------------------------------------------------
// per step (t = T-1 .. 0): gate gradients (one hidden unit per thread and slot) -> dg rows (HBM, for the weight gradients), the
// waves' maxima cross a barrier, the values are split into the two fp16 planes of dy under the step's scale;  T2 (resident or
// streamed head^T, three-term product) -> two fp16 planes of dC1 under the bound's scale;  T1 (tail^T, k-blocks split over the
// waves when there are few tiles) -> partial dh vectors summed by the next gate phase.
template <int CELL, typename TS, int UPT, bool RES>
__global__ void __launch_bounds__(G2_NT_MAX) k_g2_bwd(G2Plan P, const TS* __restrict__ out, const TS* __restrict__ h0,
                                                  const TS* __restrict__ c0, const float* __restrict__ reserve,
                                                  const TS* __restrict__ d_out, const TS* __restrict__ d_hT,
                                                  const TS* __restrict__ d_cT, const xh8* __restrict__ bs2,
                                                  const float* __restrict__ bt1, const float* __restrict__ hun,
                                                  float* __restrict__ dg_in,
                                                  float* __restrict__ dg_hid, TS* __restrict__ d_h0, TS* __restrict__ d_c0,
                                                  float* __restrict__ dstate, unsigned* __restrict__ colmax,
                                                  unsigned long long* __restrict__ diag) {
  // (diag: -DTTRNN_ABLATIONS builds only — per-phase s_memtime stamps of the first eight workgroups, tools/diag_stamps_g2bwd.py)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const G2Mat& m = P.hid;
  _Float16* dyimg = reinterpret_cast<_Float16*>(smem);      // two fp16 planes of 2^s dy_t (s: the step's scale), [2][I_t rows][IhS]
  __shared__ float smax[G2_NW_MAX];                        // the waves' maxima of |dg_t| (every wave is a gate wave: unit tid + u NT)
  // by-product (colmax != NULL, P.b_cmx > 0): max_n |dg[n][c]| per column, kept per (gate, unit) in LDS by the one thread that
  // computes that column's gradient every step (no synchronisation needed), handed over with one atomicMax per column at the end:
  // rows 0 / 1 of ttrnn_rnn_backward_ex's stats (TTRNN_BWD_STATS_COLMAX) — the dense weight gradient then runs on two fp16
  // pieces at every size, without its passes over dg
  float* cmx = reinterpret_cast<float*>(smem + P.b_lds);
  _Float16* dc1 = reinterpret_cast<_Float16*>(smem + P.b_dy);      // two fp16 planes of 2^s2 dC1 (s2: from the step's bound), [2][J_h rows][K1S]
  float* dhb = reinterpret_cast<float*>(smem + P.b_dy + P.b_dc1);
  int* dyoff = reinterpret_cast<int*>(smem + P.b_dy + P.b_dc1 + P.b_dh);
  int* t2off = dyoff + P.G * P.H;
  float* tunl = reinterpret_cast<float*>(t2off + m.bM2T * 4);                              // inverse row scales of tail^T [16 bM1T]
  float* hunl = tunl + m.bM1T * 16;                                                        // inverse row scales of head^T [b_hun]
  float* lt1 = reinterpret_cast<float*>(smem + P.b_dy + P.b_dc1 + P.b_dh + P.b_tab);       // tail^T fragments (P.b_t1 > 0)
  const int plane = (m.It < 16 * m.N2T ? m.It : 16 * m.N2T) * m.IhS;      // the dy image holds its I_t real rows
  const int planeC = (m.Jh < 16 * m.N1T ? m.Jh : 16 * m.N1T) * m.K1S;     // ... the dC1 image its J_h real rows

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int NW = m.nw, NT = NW * 64;
  const size_t b = blockIdx.x;
  const int H = P.H, T = P.T, GH = P.G * P.H, upt = P.b_upt;
  constexpr bool LSTM = CELL == TTRNN_LSTM;
  constexpr int NG = LSTM ? 4 : 3;

  for (int e = tid; e < (P.b_dy + P.b_dc1 + P.b_dh) / 4; e += NT) reinterpret_cast<unsigned*>(smem)[e] = 0u;
  if (colmax)
    for (int e = tid; e < P.b_cmx / 4; e += NT) cmx[e] = 0.f;
  for (int o = tid; o < GH; o += NT) {
    const int ih = o / m.It, it = o - ih * m.It;
    dyoff[o] = it * m.IhS + ih;
  }
  for (int e = tid; e < m.bM2T * 4; e += NT) {
    const int m2 = 16 * (e >> 2) + 4 * (e & 3);
    int jh, a;
    if (m.ng > 1) { const int g = m2 / m.Kg, rem = m2 - g * m.Kg; jh = rem / m.Rb; a = g * m.Rb + rem % m.Rb; }
    else { jh = m2 / m.Rp; a = m2 - jh * m.Rp; }
    t2off[e] = (jh < m.Jh && a < m.Rp) ? jh * m.K1S + a : -1;
  }
  const bool hl = P.b_hun > 0;
  for (int e = tid; e < P.b_hun; e += NT) hunl[e] = hun[e];
  for (int e = tid; e < m.bM1T * 16; e += NT) tunl[e] = hun[m.bM2T * 32 + e];
  if (tid < G2_NW_MAX) smax[tid] = 0.f;
  const bool t1_lds = P.b_t1 > 0;
  if (t1_lds)
    for (int e = tid; e < m.bM1T * m.bKB1 * 2 * 64 * 4; e += NT) lt1[e] = bt1[e];      // (xh8 fragments, copied as floats)
  // bound of T2's results: |dC1[row][.]| <= L1(row of head^T) max|dg_t|.  The largest row norm, once per launch
  float maxl1;
  {
    float l = 0.f;
    for (int e = tid; e < m.bM2T * 16; e += NT) l = fmaxf(l, hun[m.bM2T * 16 + e]);
    l = g2_wave_max(l);
    __syncthreads();
    if (lane == 0) smax[wave] = l;
    __syncthreads();
    maxl1 = fmaxf(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])), fmaxf(fmaxf(smax[4], smax[5]), fmaxf(smax[6], smax[7])));
  }
  float dhd[UPT], dcs[UPT];
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    dhd[u] = 0.f; dcs[u] = 0.f;
    if (u < upt) {
      const int hid = tid + u * NT;
      if (hid < H) {
        dhd[u] = d_hT ? ld(d_hT, b * H + hid) : 0.f;      // carried dh that does not come through the chain
        dcs[u] = (LSTM && d_cT) ? ld(d_cT, b * H + hid) : 0.f;
      }
    }
  }
  const int nu_w = wave < m.bU ? (m.bU - wave + NW - 1) / NW : 0;
  const int total = (nu_w * m.bNKBt + G2_PF - 1) / G2_PF * G2_PF;      // this wave's stream: its live blocks, padded to the slot ring
  const xh8* sp = bs2 + (size_t)wave * m.bSW * 2 * 64 + lane;
  // RES (host: one column tile, bNKBt <= 4 and bUW * bNKBt <= G2_PF): slot s holds live block (unit s / bNKBt, k-block s % bNKBt)
  // of this wave for the whole launch; otherwise the slots roll over the wave's stream
  // (block-diagonal heads, ng > 1: a unit's live k-blocks are its gate's bNKBt, not all bNKB)
  const int r_nlive = RES ? nu_w * m.bNKBt : 0;
  constexpr int NSL = RES ? G2_BSL : G2_PF;      // resident: twelve slots; the rolling stream: a ring of G2_PF
  xh8 wbuf[NSL][2];
#pragma unroll
  for (int j = 0; j < NSL; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int e = 0; e < 8; ++e) wbuf[j][p][e] = (_Float16)0.f;
      if constexpr (RES) {
        if (j < r_nlive) wbuf[j][p] = sp[(size_t)j * 2 * 64 + p * 64];
      } else {
        if (total > 0) wbuf[j][p] = sp[(size_t)j * 2 * 64 + p * 64];
      }
    }
  // the step's record (gates, cell / previous state, d_out) is requested one step ahead — right behind the gate phase of step t + 1, under
  // its T2 / T1 — instead of at the head of the step that needs it (a dependent HBM / L2 round trip on every step's critical path)
  // (four units per thread with resident head fragments: the 28 registers of a second record set spill — there the loads stay put)
  constexpr bool PREF = UPT <= 2 || !RES;
  f32x4 nq[UPT];
  float na[UPT], np[UPT], nd[UPT];
#ifdef TTRNN_ABLATIONS
  const int abl = P.abl;      // 8: no d_gates stores, 16: no record loads, 32: no column maxima, 64: no dh partial-sum reads
#else
  constexpr int abl = 0;
#endif
  // No lane-divergent branch around the loads: a load inside a divergent region is merged with the register's other value at the join,
  // and the compiler waits for it THERE — vmcnt(0) right behind the request, 1 900 of 4 800 cycles of the GRU gate phase at H = 512.
  // Lanes (and unit slots) without a unit read unit 0's record; a null d_out reads the reserve and is scaled by zero.
  const float dsc = d_out ? 1.f : 0.f;
  // (b T + t) H, carried from step to step: the 64-bit products of the row index were nineteen scalar multiplies per step
  size_t rH = T > 0 ? (b * T + (T - 1)) * (size_t)H : 0;
  const size_t cellb = (size_t)P.B * T * H * 4;             // the cells follow the gate records in the reserve (res_cell)
  auto load_unit = [&](const int tt, const size_t rq, const int u) {      // rq = (b T + tt) H
    const int hid0 = tid + u * NT;
    const int hid = (u < upt && hid0 < H) ? hid0 : 0;
    if (abl & 16) { nq[u] = f32x4{0.f, 0.f, 0.f, 0.f}; na[u] = 0.f; np[u] = 0.f; nd[u] = 0.f; return; }
    if (LSTM) {
      nq[u] = *reinterpret_cast<const f32x4*>(reserve + (rq + hid) * 4);          // res_gate
      na[u] = reserve[cellb + rq + hid];                                           // res_cell
      if (tt > 0) np[u] = reserve[cellb + rq - H + hid];      // (wave-uniform branches)
      else np[u] = c0 ? ld(c0, b * H + hid) : 0.f;
    } else {
      nq[u] = *reinterpret_cast<const f32x4*>(reserve + (rq + hid) * 4);
      na[u] = 0.f;
      if (tt > 0) np[u] = ld(out, rq - H + hid);
      else np[u] = h0 ? ld(h0, b * H + hid) : 0.f;
    }
    nd[u] = d_out ? ld(d_out, rq + hid) * dsc : 0.f;
  };
  auto load_rec = [&](const int tt, const size_t rq) {
#pragma unroll
    for (int u = 0; u < UPT; ++u) load_unit(tt, rq, u);
  };
  if (PREF && T > 0) load_rec(T - 1, rH);
  __syncthreads();
#ifdef TTRNN_ABLATIONS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = stamp();
#define G2B_STAMP(i) { const unsigned long long now_ = stamp(); seg[i] += now_ - last_; last_ = now_; }
#else
#define G2B_STAMP(i)
#endif

  for (int t = T - 1; t >= 0; --t) {
    // ---- gate gradients ------------------------------------------------------------------------------------------------------------
    // vmcnt(0), said once and unconditionally: the record requested a step ago has long arrived, but the compiler cannot know (its
    // consumers sit in a lane-divergent region) and would otherwise wait before the NEXT request overwrites the registers — behind
    // this step's d_gates stores, i.e. for their completion (ISA of the GRU instantiation: s_waitcnt vmcnt(0) between stores and loads)
    if constexpr (PREF) __builtin_amdgcn_s_waitcnt(0x0F70);
    G2B_STAMP(6)      // (diagnostic builds: the wait for the record, if any)
    float pk[UPT][4];                                         // the hidden-side gate gradients of this thread's units
    // the gate phase's LDS reads (partial dh sums, running column maxima, image offsets) go out together at the head of a unit: read one
    // by one where they are used they were sixteen dependent round trips of the phase's 3 500 cycles
    constexpr bool OFFR = UPT == 1 || (!RES && UPT == 2);       // the image offsets stay in registers until the split (where there is room)
    int dof[OFFR ? UPT : 1][4];
    float tmx = 0.f;
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
#pragma unroll
      for (int g = 0; g < 4; ++g) pk[u][g] = 0.f;
      if constexpr (!PREF) load_unit(t, rH, u);
      if (u < upt) {
        const int hid = tid + u * NT;
        if (hid < H) {
          float dht = dhd[u];
          float cmv[NG + 1];
#pragma unroll
          for (int g = 0; g <= NG; ++g) cmv[g] = 0.f;
          const bool cm_on = colmax && !(abl & 32);
          if (cm_on) {
#pragma unroll
            for (int g = 0; g < NG; ++g) cmv[g] = cmx[g * H + hid];
            if (!LSTM) cmv[NG] = cmx[NG * H + hid];
          }
          if constexpr (OFFR) {
#pragma unroll
            for (int g = 0; g < NG; ++g) dof[u][g] = dyoff[g * H + hid];
          }
          for (int pt = 0; pt < ((abl & 64) ? 0 : m.bK1SPLIT); pt += 4) {        // up to four partial sums per trip, reads issued together
            float pv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[e] = dhb[(pt + e < m.bK1SPLIT ? pt + e : pt) * H + hid];
#pragma unroll
            for (int e = 0; e < 4; ++e) dht += pt + e < m.bK1SPLIT ? pv[e] : 0.f;
          }
          dht += nd[u];
          float p[4] = {0.f, 0.f, 0.f, 0.f}, ph2 = 0.f;
          if (LSTM) {
            const f32x4 gq = nq[u];
            const float ig = gq[0], gg = gq[1], fg = gq[2], og = gq[3], cy = na[u];
            const float cprev = np[u];
            const float tc = ftanh(cy);
            const float dct = dcs[u] + dht * og * (1.0f - tc * tc);
            if (dstate) { dstate[(rH + hid) * 2] = dht; dstate[(rH + hid) * 2 + 1] = dct; }
            p[0] = dct * gg * ig * (1.0f - ig);
            p[1] = dct * cprev * fg * (1.0f - fg);
            p[2] = dct * ig * (1.0f - gg * gg);
            p[3] = dht * tc * og * (1.0f - og);
            dcs[u] = dct * fg;
            dhd[u] = 0.f;
          } else {
            const f32x4 gq = nq[u];
            const float rg = gq[0], zg = gq[1], ng = gq[2], hn = gq[3];
            const float hprev = np[u];
            if (dstate) { dstate[(rH + hid) * 2] = dht; dstate[(rH + hid) * 2 + 1] = 0.f; }
            const float dn_pre = dht * (1.0f - zg) * (1.0f - ng * ng);
            p[1] = dht * (hprev - ng) * zg * (1.0f - zg);
            p[0] = dn_pre * hn * rg * (1.0f - rg);
            p[2] = dn_pre;                                  // w.r.t. the input part of n
            ph2 = dn_pre * rg;                              // w.r.t. the hidden part of n (inside the r * (...) product)
            dhd[u] = dht * zg;
          }
          if (cm_on) {
#pragma unroll
            for (int g = 0; g < NG; ++g) cmx[g * H + hid] = fmaxf(cmv[g], fabsf(p[g]));
            if (!LSTM) cmx[NG * H + hid] = fmaxf(cmv[NG], fabsf(ph2));
          }
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            if (!(abl & 8)) dg_in[rH * NG + g * H + hid] = p[g];
            const float ph = (!LSTM && g == 2) ? ph2 : p[g];
            if (!LSTM && !(abl & 8)) dg_hid[rH * NG + g * H + hid] = ph;
            pk[u][g] = ph;
            tmx = fmaxf(tmx, fabsf(ph));
          }
        }
      }
    }
    if (PREF && t > 0) load_rec(t - 1, rH - H);
    // the step's operand scale from the EXACT maximum of |dg_t| (the gate waves' maxima cross a barrier of their own), then every
    // owner splits its values into the two fp16 planes: 2^s dy_t with the maximum in [2^13, 2^14) — ttrnn_fast_f10bh.hip's scheme
    tmx = g2_wave_max(tmx);
    if (lane == 0) smax[wave] = tmx;
    G2B_STAMP(7)      // (diagnostic builds: gate math + stores + record request, before the maxima cross)
    lds_barrier();
    float ust2;          // T2's accumulators -> T1's operand: undo the step's scale, apply 2^s2 (from the bound maxl1 max|dg_t|: cannot overflow)
    float u2;            // inverse of 2^s2, applied where T1's accumulators are stored
    {
      const float mxg = fmaxf(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])), fmaxf(fmaxf(smax[4], smax[5]), fmaxf(smax[6], smax[7])));
      float ustep;
      const float sg = g2_step_scale(mxg, ustep);
      ust2 = ustep * g2_step_scale(mxg * maxl1, u2);
#pragma unroll
      for (int u = 0; u < UPT; ++u)
        if (u < upt) {
          const int hid = tid + u * NT;
          if (hid < H) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              _Float16 s0, s1;
              split2h(pk[u][g] * sg, s0, s1);
              int off;
              if constexpr (OFFR) off = dof[u][g];
              else off = dyoff[g * H + hid];
              dyimg[off] = s0; dyimg[plane + off] = s1;
            }
          }
        }
    }
    G2B_STAMP(0)
    lds_barrier();
    G2B_STAMP(1)
    // ---- T2: dC1 = head^T dy (two fp16 pieces per operand; resident / streaming instantiations as in the forward kernel) ---------------
    // pass: which half of dC1's i_t range (bNP == 2; else 0 and every tile is computed)
    auto stageT2 = [&](const int pass) {
      // the wave's stream holds ONLY the live k-blocks of its units, back to back (unit ui, k-block kb at position ui bNKBt + kb; the
      // tail of the last group of G2_PF is padding): with every unit padded to G2_PF blocks the shapes whose reverse contraction is
      // short (K = I_h: 2 - 3 blocks) pulled 2.7 - 4 x their head through L2 every step — 28 TB/s at B = 512, the whole of T2's time.
      // (unit, k-block) of a position are carried along (wave-uniform scalars); slot j of the register ring is static.
      const int nth = m.N2T / m.bNP, it0 = pass * (m.It / m.bNP);
      const int nlive = nu_w * m.bNKBt;
      int ui = 0, kb = 0;
      const _Float16* brow = dyimg;           // current unit: its dy rows, its store offset, its column, does this pass compute it
      int off = -1, itc = 0;
      bool act = false;
      auto unit_setup = [&](const int u, const _Float16*& br, int& of, int& ic, bool& ac, int& rw) {
        const int tile = wave + u * NW;
        const int mt = tile / m.N2T, nt = tile - mt * m.N2T;
        ac = !(m.bNP > 1 && nt / nth != pass);                                           // the other half's tile: only the stream rolls on
        const int kbase = (m.ng > 1 ? (16 * mt) / m.Kg : 0) * m.bNKBt;                 // block-diagonal heads: the gate's i_h range
        br = dyimg + (16 * nt + c < m.It ? 16 * nt + c : m.It - 1) * m.IhS + 8 * q + 32 * kbase;
        of = t2off[mt * 4 + q];
        ic = 16 * nt + c;
        rw = 16 * mt + 4 * q;
      };
      int row0 = 0;                           // first of the lane's four T2 rows (their inverse scales: hunl, or fetched a unit ahead)
      f32x4 cun = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
      xh8 bf[2][2];
      if (nlive > 0) {
        unit_setup(0, brow, off, itc, act, row0);
        if (!hl) cun = *reinterpret_cast<const f32x4*>(hun + row0);
#pragma unroll
        for (int p = 0; p < 2; ++p) bf[0][p] = *reinterpret_cast<const xh8*>(brow + p * plane);
      }
      for (int ch = 0; ch < total; ch += G2_PF) {
#pragma unroll
        for (int j = 0; j < G2_PF; ++j) {
          const int pos = ch + j;
          if (pos < nlive) {
            // the next position's dy fragments are requested before this block's MFMAs
            const bool last = kb + 1 == m.bNKBt;
            const _Float16* nbrow = brow;
            int noff = off, nitc = itc, nrow0 = row0;
            bool nact = act;
            f32x4 nun = cun;
            if (pos + 1 < nlive) {
              if (last) {
                unit_setup(ui + 1, nbrow, noff, nitc, nact, nrow0);
                if (!hl) nun = *reinterpret_cast<const f32x4*>(hun + nrow0);
              }
              const int nkb = last ? 0 : kb + 1;
#pragma unroll
              for (int p = 0; p < 2; ++p) bf[(j + 1) & 1][p] = *reinterpret_cast<const xh8*>(nbrow + p * plane + 32 * nkb);
            }
            if (act) {
              split_block_h(wbuf[j], bf[j & 1], acc_lo, acc_hi);
              if (last) {
                const f32x4 un = (hl ? *reinterpret_cast<const f32x4*>(hunl + row0) : cun) * ust2;
                if (off >= 0 && itc < m.It) store_split4_h(dc1, planeC, off + (itc - it0) * m.Rp, (acc_hi + acc_lo) * un);
              }
            }
            if (last) {
              acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}; acc_hi = acc_lo;
              kb = 0; ++ui;
              brow = nbrow; off = noff; itc = nitc; act = nact; row0 = nrow0; cun = nun;
            } else {
              ++kb;
            }
          }
          int nxt = pos + G2_PF;
          nxt -= nxt >= total ? total : 0;
#pragma unroll
          for (int p = 0; p < 2; ++p) wbuf[j][p] = sp[(size_t)nxt * 2 * 64 + p * 64];
        }
      }
    };
    // resident: the dy fragments are the SAME for every row tile (one column tile): read once per step, then every live
    // slot is one split block; a unit's accumulators are flushed after its last k-block.  NKB is a compile-time constant
    // inside each instantiation of the body (slot -> k-block must index registers statically).
    auto stageT2res = [&](auto nkb_tag) {
      constexpr int NKB = decltype(nkb_tag)::value;             // live k-blocks per unit (= bNKBt)
      const _Float16* brow = dyimg + (c < m.It ? c : m.It - 1) * m.IhS + 8 * q;
      xh8 bfr[NKB][2];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int p = 0; p < 2; ++p) bfr[kb][p] = *reinterpret_cast<const xh8*>(brow + p * plane + 32 * kb);
      f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
      // several column tiles (round 5; before, such shapes never were resident: every d = 2 / d = 4 shape ran its T2 on the streamed
      // general loop): unit ui of this wave = tile wave + ui NW = (row tile, column tile), the division by N2T <= 4 as a multiply-shift
      const bool per_unit = m.ng > 1 || m.N2T > 1;
      const int n2rcp = (65536 + m.N2T - 1) / m.N2T;
      int u_mt = wave, u_itc = c;
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) {
        const int ui = sl / NKB, kb = sl % NKB;
        if (sl < r_nlive) {
          if (kb == 0) {
            acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}; acc_hi = acc_lo;
            if (per_unit) {
              // block-diagonal heads: this unit (row tile of gate g) multiplies its own gate's i_h range of dy; several column
              // tiles: its own column tile's rows of dy
              const int tile = wave + ui * NW;
              u_mt = (tile * n2rcp) >> 16;
              const int nt = tile - u_mt * m.N2T;
              u_itc = 16 * nt + c;
              const int kbase = m.ng > 1 ? ((16 * u_mt) / m.Kg) * m.bNKBt : 0;
              const _Float16* bu = dyimg + (u_itc < m.It ? u_itc : m.It - 1) * m.IhS + 8 * q + 32 * kbase;
#pragma unroll
              for (int k2 = 0; k2 < NKB; ++k2)
#pragma unroll
                for (int p = 0; p < 2; ++p) bfr[k2][p] = *reinterpret_cast<const xh8*>(bu + p * plane + 32 * k2);
            } else {
              u_mt = wave + ui * NW;
            }
          }
          split_block_h(wbuf[sl], bfr[kb], acc_lo, acc_hi);
          if (kb == NKB - 1) {
            const int off = t2off[u_mt * 4 + q];
            const f32x4 un = *reinterpret_cast<const f32x4*>(hunl + 16 * u_mt + 4 * q) * ust2;
            if (off >= 0 && u_itc < m.It) store_split4_h(dc1, planeC, off + u_itc * m.Rp, (acc_hi + acc_lo) * un);
          }
        }
      }
    };
    // streamed, the common geometry (one column tile, one pass, 1 / 2 / 4 live k-blocks per unit, the row scales in LDS): a group of
    // eight slots = 8 / NKB whole units, every per-position decision of stageT2 gone — that loop is 370 instructions per streamed
    // block (unit bookkeeping with integer divisions, thirteen exec-masked regions), issued by a wave that has its SIMD to itself
    // or shares it with one other: 47 000 of a step's 57 000 cycles at --naive_tt.  Slots past the wave's last unit hold zero
    // fragments (k_g2_head_frag) and are multiplied, their results dropped.
    auto stageT2fast = [&](auto nkb_tag) {
      constexpr int NKB = decltype(nkb_tag)::value;
      constexpr int UPG = G2_PF / NKB;
      const int ngroups = total / G2_PF;
      const int tpg = m.ng > 1 ? m.Kg / 16 : 1;            // row tiles per gate (block-diagonal heads)
      int mt = wave, gate = m.ng > 1 ? wave / tpg : 0, rem = m.ng > 1 ? wave - gate * tpg : 0;
      const _Float16* brow0 = dyimg + (c < m.It ? c : m.It - 1) * m.IhS + 8 * q;
      for (int g = 0; g < ngroups; ++g) {
        const xh8* nx = sp + (size_t)((g + 1 < ngroups ? g + 1 : 0) * G2_PF) * 2 * 64;      // the next group's blocks (the ring wraps here only)
#pragma unroll
        for (int u = 0; u < UPG; ++u) {
          const bool on = g * UPG + u < nu_w;
          const int mtc = on ? mt : wave;
          const _Float16* br = brow0 + 32 * NKB * (on ? gate : 0);
          xh8 bfr[NKB][2];
#pragma unroll
          for (int k = 0; k < NKB; ++k)
#pragma unroll
            for (int p = 0; p < 2; ++p) bfr[k][p] = *reinterpret_cast<const xh8*>(br + p * plane + 32 * k);
          const int off = t2off[mtc * 4 + q];
          const f32x4 hu = *reinterpret_cast<const f32x4*>(hunl + 16 * mtc + 4 * q);
          f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
#pragma unroll
          for (int k = 0; k < NKB; ++k) {
            split_block_h(wbuf[u * NKB + k], bfr[k], acc_lo, acc_hi);
#pragma unroll
            for (int p = 0; p < 2; ++p) wbuf[u * NKB + k][p] = nx[(size_t)(u * NKB + k) * 2 * 64 + p * 64];
          }
          if (on && off >= 0 && c < m.It) store_split4_h(dc1, planeC, off + c * m.Rp, (acc_hi + acc_lo) * (hu * ust2));
          mt += NW;
          if (m.ng > 1) { rem += NW; while (rem >= tpg) { rem -= tpg; ++gate; } }
        }
      }
    };
    if constexpr (RES) {
      if (m.bNKBt == 1) stageT2res(std::integral_constant<int, 1>{});
      else if (m.bNKBt == 2) stageT2res(std::integral_constant<int, 2>{});
      else if (m.bNKBt == 3) stageT2res(std::integral_constant<int, 3>{});
      else stageT2res(std::integral_constant<int, 4>{});
    } else {
      // (one unit per thread only: with two or four the extra operand registers spill — 15 to 54 VGPRs)
      bool done = false;
      if constexpr (UPT == 1) {
        if (m.N2T == 1 && m.bNP == 1 && hl && P.b_fast) {
          done = true;
          if (m.bNKBt == 1) stageT2fast(std::integral_constant<int, 1>{});
          else if (m.bNKBt == 2) stageT2fast(std::integral_constant<int, 2>{});
          else if (m.bNKBt == 4) stageT2fast(std::integral_constant<int, 4>{});
          else done = false;
        }
      }
      if (!done) stageT2(0);
    }
    G2B_STAMP(2)
    lds_barrier();
    G2B_STAMP(3)
    // ---- T1: dh = tail^T dC1 (two fp16 pieces per operand) ------------------------------------------------------------------------------
    auto stageT1 = [&](auto frag, auto depth, const int pass) {      // frag(i): fragment i of tail^T, from LDS or from L2 (two instantiations)
      const int kbh = m.bKB1 / m.bNP, klo = pass * kbh, khi = klo + kbh;          // this pass's k-blocks (= its i_t range)
      for (int u1 = wave; u1 < m.bU1; u1 += NW) {
        const int tile = u1 / m.bK1SPLIT, part = u1 - tile * m.bK1SPLIT;
        const int mt = tile / m.N1T, nt = tile - mt * m.N1T;
        int k0 = part * m.bKB1P;
        int k1 = k0 + m.bKB1P < m.bKB1 ? k0 + m.bKB1P : m.bKB1;
        k0 = k0 > klo ? k0 : klo;
        k1 = k1 < khi ? k1 : khi;
        const int jrow = 16 * nt + c;                                  // rows past J_h: any real row (their results are dropped)
        const _Float16* bp = dc1 + (jrow < m.Jh ? jrow : m.Jh - 1) * m.K1S + 8 * q - 32 * klo;
        const int fb = mt * m.bKB1 * 2 * 64 + lane;
        f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
        // DEPTH k-blocks' operands in flight: two where the fragments sit in LDS, four where they come from L2 (H = 768, d = 4: 96 KB of
        // tail^T, sixteen blocks per tile at two in flight = eight L2 round trips per step, 11 000 of the step's 32 000 cycles)
        constexpr int DEPTH = decltype(depth)::value;
        for (int kb = k0; kb < k1; kb += DEPTH) {
          xh8 wf[DEPTH][2], xf[DEPTH][2];
#pragma unroll
          for (int d = 0; d < DEPTH; ++d) {
            const int kc = kb + d < k1 ? kb + d : k1 - 1;
            wf[d][0] = frag(fb + kc * 2 * 64); wf[d][1] = frag(fb + kc * 2 * 64 + 64);
            xf[d][0] = *reinterpret_cast<const xh8*>(bp + 32 * kc); xf[d][1] = *reinterpret_cast<const xh8*>(bp + planeC + 32 * kc);
          }
#pragma unroll
          for (int d = 0; d < DEPTH; ++d)
            if (kb + d < k1) split_block_h(wf[d], xf[d], acc_lo, acc_hi);
        }
        const f32x4 acc = (acc_hi + acc_lo) * (*reinterpret_cast<const f32x4*>(tunl + 16 * mt + 4 * q) * u2);
        const int jh = 16 * nt + c;
        if (jh < m.Jh) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int jt = 16 * mt + 4 * q + j;
            if (jt < m.Jt) dhb[part * H + jh * m.Jt + jt] = pass == 0 ? acc[j] : dhb[part * H + jh * m.Jt + jt] + acc[j];
          }
        }
      }
    };
    if (t1_lds) stageT1([&](int i) { return reinterpret_cast<const xh8*>(lt1)[i]; }, std::integral_constant<int, 2>{}, 0);
    else stageT1([&](int i) { return reinterpret_cast<const xh8*>(bt1)[i]; }, std::integral_constant<int, RES ? 2 : 4>{}, 0);
    G2B_STAMP(4)
    lds_barrier();
    G2B_STAMP(5)
    if constexpr (!RES) {
      if (m.bNP > 1) {                                      // second half of dC1's i_t range (host: the image did not fit whole)
        stageT2(1);
        lds_barrier();
        if (t1_lds) stageT1([&](int i) { return reinterpret_cast<const xh8*>(lt1)[i]; }, std::integral_constant<int, 2>{}, 1);
        else stageT1([&](int i) { return reinterpret_cast<const xh8*>(bt1)[i]; }, std::integral_constant<int, RES ? 2 : 4>{}, 1);
        lds_barrier();
      }
    }
    rH -= H;
  }
#ifdef TTRNN_ABLATIONS
  if (diag && lane == 0 && b < 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) diag[(b * G2_NW_MAX + wave) * 8 + i] = seg[i];
  }
#endif
#undef G2B_STAMP
#pragma unroll
  for (int u = 0; u < UPT; ++u)
    if (u < upt) {
      const int hid = tid + u * NT;
      if (hid < H) {
        float dht = dhd[u];
        for (int pt = 0; pt < m.bK1SPLIT; ++pt) dht += dhb[pt * H + hid];
        if (d_h0) st(d_h0, b * H + hid, dht);
        if (LSTM && d_c0) st(d_c0, b * H + hid, dcs[u]);
        if (colmax) {
          // row 0: d_gates_in; row 1: d_gates_hid (LSTM: the same tensor; GRU: the n gate's hidden-side gradient differs)
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const float v = cmx[g * H + hid];
            atomicMax(colmax + g * H + hid, __float_as_uint(v));
            atomicMax(colmax + GH + g * H + hid, __float_as_uint((!LSTM && g == 2) ? cmx[NG * H + hid] : v));
          }
        }
      }
    }
}

int check() { return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH; }

// merged cores + fragments of one TT-matrix into ws: [Gh | Gt | head stream | tail fragments]
// hdr_out: forward only (per-block maxima of the merged cores for the fp16 scales)
int prep(const TtShape& s, const G2Mat& m, bool rev, const float* packed, void* ws, const xbf8** fs, const float** ft,
         hipStream_t stream, const int** hdr_out = nullptr, const float** hun_out = nullptr) {
  char* p = (char*)ws;
  float* Gh = (float*)p; p += g2_al((size_t)m.head_elems * 4);
  float* Gt = (float*)p; p += g2_al((size_t)m.tail_elems * 4);
  xbf8* hs = (xbf8*)p; p += g2_al((size_t)(rev ? m.bs2_bytes : m.fs2_bytes));
  float* tf = (float*)p; p += g2_al((size_t)(rev ? m.bt1_bytes : m.ft1_bytes));
  int* hdr = (int*)p;                         // forward only (g2_fwd_ws_bytes): exponents of the diagonal scales
  float* hun = (float*)p;                     // reverse only (g2_bwd_ws_bytes): head^T rows' inverse scales [16 bM2T], their L1 norms
                                              // [16 bM2T], tail^T rows' inverse scales [16 bM1T]
  const int nblk = (int)g2_merge_blocks(m);
  hipLaunchKernelGGL(k_g2_merge, dim3((unsigned)nblk), dim3(256), 0, stream, s, m, packed, Gh, Gt,
                     m.ng > 1 ? device_status_ptr() : (unsigned*)nullptr);
  if (rev) {
    hipLaunchKernelGGL(k_g2_head_frag<true>, dim3(m.nw * m.bSW), dim3(64), 0, stream, m, Gh, hs, (const int*)nullptr, hun);
    if (hun_out) *hun_out = hun;
    hipLaunchKernelGGL(k_g2_tail_frag<true>, dim3(m.bM1T * m.bKB1), dim3(64), 0, stream, m, Gt, tf, (const int*)nullptr,
                       hun + m.bM2T * 32);
  } else {
    float* dpart = (float*)(hdr + g2_diag_ints(m));
    hipLaunchKernelGGL(k_g2_diag_a, dim3(m.It), dim3(256), 0, stream, m, (const float*)Gt, hdr, dpart);
    hipLaunchKernelGGL(k_g2_diag_b, dim3(m.Ih), dim3(256), 0, stream, m, (const float*)Gh, hdr, (const float*)dpart);
    hipLaunchKernelGGL(k_g2_head_frag<false>, dim3(m.nw * (m.UW * m.KBP + m.wrap)), dim3(64), 0, stream, m, Gh, hs, (const int*)hdr);
    hipLaunchKernelGGL(k_g2_tail_frag<false>, dim3(m.M1T * m.KB1), dim3(64), 0, stream, m, Gt, tf, (const int*)hdr);
    if (hdr_out) *hdr_out = hdr;
  }
  *fs = hs;
  *ft = tf;
  return check();
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------------------
// K-in of the G2 route: input_size == 1 -> the unit row through the any-shape chain kernel (the recurrent kernel scales
// it by x_t); otherwise ONE dense split-bf16 GEMM (ttrnn_fast_gemm.hip) whose matrix is the chain kernel applied to the
// `in` identity rows — the input contraction padded to a multiple of 8, the gate-interleaved fp32 bias row added in the
// GEMM's epilogue.
static int in_pad(int in) { return (in + 7) & ~7; }

// The plan of ONE direction: eight-wave workgroups where a sample has a CU to itself (B <= #CUs), four-wave ones otherwise —
// and four-wave ones also where the eight-wave plan of that direction does not fit LDS but the four-wave plan does (the partial
// dh vectors of T1's k split are per wave group: naive per-gate sets of H = 512, r = 16 miss the limit by 256 bytes with eight
// waves; their BPTT ran on the VALU kernels, 320 ... 590 ms per pMNIST step).  The forward and the reverse-time kernel are
// separate launches: each takes its own plan.
// forward: the head fragments of every wave fit SIXTEEN register slots (and not eight): eight-wave plan, one stage-2 unit per wave
static bool g2_fwd_res16(const G2Plan& p) {
  return p.hid.ok && p.okf && p.hid.nw == 8 && p.upt <= 2 && p.hid.UW == 1 && p.hid.KBP > G2_PF && p.hid.KBP <= 16 &&
         !opt(OPT_DIAG) && !(opt(OPT_DEV) & 4096);       // (dev bit 12: A/B switch, the streamed kernel as before)
}

// reverse: every wave's head^T fragments stay in registers (one column tile, at most four live k-blocks per unit, G2_BSL slots)
static bool g2_bwd_res(const G2Plan& p) {
  // (several column tiles: resident since round 5 — dev bit 29: as before, one column tile only)
  return p.hid.ok && p.okb && p.b_hun > 0 && (p.hid.N2T == 1 || (p.hid.N2T <= 4 && p.hid.bNP == 1 && !(opt(OPT_DEV) & (1 << 29)))) &&
         p.hid.bNKBt <= 4 && p.hid.bUW * p.hid.bNKBt <= G2_BSL &&
         !(opt(OPT_DEV) & 2048 && p.hid.ng > 1);
}

static bool g2_fwd_resident(const G2Plan& p);
static void plan_for_single(G2Plan* p, const RnnShape& rs, bool backward) {
  const bool wide = rs.B <= device_cu_count();
  // forward: column tiles inside a stage-2 unit where that makes a streamed head resident (G2Mat::cin; dev bit 27: never) — the plan of
  // the batch's own wave count first, the eight-wave plan where only it gets there
  if (!backward && !(opt(OPT_DEV) & (1 << 27)) && !opt(OPT_DIAG)) {
    G2Plan a, b8;
    g2_plan(&a, rs, wide, true);
    if (a.hid.ok && a.okf && a.hid.cin) { *p = a; return; }
    if (!wide) {
      g2_plan(&b8, rs, true, true);
      G2Plan a0;
      g2_plan(&a0, rs, false, false);
      if (b8.hid.ok && b8.okf && b8.hid.cin && !(a0.hid.ok && a0.okf && g2_fwd_resident(a0))) { *p = b8; return; }
    }
  }
  if (!wide && backward && !(opt(OPT_DEV) & 4096)) {
    // the same trade in the reverse kernel: eight-wave workgroups with the head^T fragments resident (two rounds) instead of two
    // co-resident four-wave workgroups that stream them from L2 every step
    G2Plan q4, q8;
    g2_plan(&q4, rs, false);
    g2_plan(&q8, rs, true);
    if (g2_bwd_res(q8) && !g2_bwd_res(q4)) { *p = q8; return; }
  }
  if (!wide && !backward) {
    // more samples than CUs: four-wave workgroups share a CU and hide each other's latencies — unless the eight-wave plan keeps
    // the head core in registers where the four-wave plan streams it from L2 every step (two rounds of one-sample-per-CU
    // workgroups with no weight traffic beat two co-resident workgroups pulling 2 x 384 KB per step through the same L2 port)
    G2Plan q;
    g2_plan(&q, rs, true);
    if (g2_fwd_res16(q)) { *p = q; return; }
  }
  g2_plan(p, rs, wide);
  if (!wide && backward && p->hid.ok && p->okb && !(opt(OPT_DEV) & (1 << 21))) {
    // four-wave workgroups are meant to run two per CU; where the reverse kernel's LDS leaves room for ONE, its four waves sit alone
    // on their SIMDs with nothing to hide the T2 loop's bookkeeping behind (--naive_tt: 47 000 of a step's 57 000 cycles, 370
    // instructions per streamed block): the eight-wave plan instead, half the blocks per wave (dev bit 21: as before)
    const int per_cu = G2_LDS_LIMIT / (p->b_lds + p->b_cmx + G2_LDS_STATIC);
    if (per_cu < 2) {
      G2Plan q8;
      g2_plan(&q8, rs, true);
      if (q8.hid.ok && q8.okb) { *p = q8; return; }
    }
  }
  if (!wide || !p->hid.ok || (backward ? p->okb : p->okf)) return;
  G2Plan q;
  g2_plan(&q, rs, false);
  if (q.hid.ok && (backward ? q.okb : q.okf)) *p = q;
}

// forward, more samples than CUs, a head that no plan keeps in registers: TWO samples per workgroup share the stream (k_g2_fwd_p)
// — where that kernel's images fit LDS (dev bit 19: A/B switch, one sample per workgroup as before)
static void plan_for(G2Plan* p, const RnnShape& rs, bool backward) {
  plan_for_single(p, rs, backward);
  p->b_fast = (opt(OPT_DEV) & (1 << 22)) ? 0 : 1;      // (dev bit 22: the reverse kernel's streamed T2 on its general loop everywhere)
  const bool any_b = (opt(OPT_DEV) & (1 << 20)) && rs.B >= 2;      // (dev bit 20: the tests' switch — pairs whatever the batch)
  if (backward || (rs.B <= device_cu_count() && !any_b) || (opt(OPT_DEV) & (1 << 19))) return;
  if (p->hid.ok && p->okf && (p->hid.UW * p->hid.KBP <= G2_PF || g2_fwd_res16(*p))) return;      // resident
  G2Plan q;
  g2_plan_pair(&q, rs);
  if (q.okf) *p = q;
}

static bool g2_fwd_resident(const G2Plan& p) { return p.hid.UW * p.hid.KBP <= G2_PF || g2_fwd_res16(p); }

static bool g2_available(const RnnShape& rs, int dtype, bool backward) {
  if (opt(OPT_NO_G2) || rs.B < 1 || rs.T < 1) return false;
  if (dtype != TTRNN_F32 && dtype != TTRNN_BF16) return false;
  // every product of this tier is a product of 16-bit pieces: fp32 descriptors under TTRNN_MATH_EXACT take the any-shape
  // fp32 kernels instead (force_g2 is the A/B override the tests use)
  if (dtype == TTRNN_F32 && opt(OPT_FP32_MATH) == TTRNN_MATH_EXACT && !opt(OPT_FORCE_G2)) return false;
  G2Plan p;
  plan_for(&p, rs, backward);
  if (!(backward ? p.okb : p.okf)) return false;
  if (rs.in == 1) return true;
  return gemm_split_ok(in_pad(rs.in), 4 * rs.H);
}
bool g2_rnn_available(const RnnShape& rs, int dtype) { return g2_available(rs, dtype, false); }
bool g2_rnn_bwd_available(const RnnShape& rs, int dtype) { return g2_available(rs, dtype, true); }
bool g2_rnn_fwd_paired(const RnnShape& rs) {
  G2Plan p;
  plan_for(&p, rs, false);
  return p.okf && p.pair;
}

struct G2FwdWs {
  size_t gin, bilv, rec, ident, wdense, planes, xpad, lin, total;
};

static G2FwdWs g2_fwd_layout(const RnnShape& rs) {
  G2FwdWs w{};
  G2Plan p;
  plan_for(&p, rs, false);
  const bool in1 = rs.in == 1;
  const int inp = in_pad(rs.in);
  const int64_t rows = in1 ? 1 : (int64_t)rs.B * rs.T;
  w.gin = g2_al((size_t)rows * 4 * rs.H * sizeof(float));
  w.bilv = g2_al((size_t)4 * rs.H * sizeof(float));
  w.rec = g2_fwd_ws_bytes(p.hid);
  // (H = 512, r = 8 in split mode: the same region is the fused-core forward kernel's fragment workspace — sized for it whatever
  // this tier's own plan needs, so that the route report and the launch cannot disagree: ADVICE r4)
  if (f10_h512_fwd_available(rs, TTRNN_F32) && w.rec < g2_al(f10_h512_workspace_bytes())) w.rec = g2_al(f10_h512_workspace_bytes());
  if (rs.cell == TTRNN_GRU && w.rec < g2_al(f10gh_workspace_bytes(rs))) w.rec = g2_al(f10gh_workspace_bytes(rs));      // (0 for other shapes)
  if (rs.cell == TTRNN_GRU && w.rec < g2_al(f10g5_workspace_bytes(rs))) w.rec = g2_al(f10g5_workspace_bytes(rs));
  if (w.rec < g2_al(f10n_workspace_bytes(rs))) w.rec = g2_al(f10n_workspace_bytes(rs));      // (naive per-gate sets, H = 256)
  if (!in1) {
    w.ident = gemm_split_identity_bytes(rs.in);
    w.wdense = gemm_split_dense_bytes(inp, 4 * rs.H);
    w.planes = gemm_split_plane_bytes(inp, 4 * rs.H) + gemm_half_scratch_bytes((int64_t)rs.B * rs.T, inp, 4 * rs.H);
    w.xpad = inp != rs.in ? g2_al((size_t)rs.B * rs.T * inp * 4) : 0;
  }
  w.lin = g2_al(plan_ttlinear_fwd(rs.in_s, in1 ? 1 : rs.in).ws_bytes);
  if (!in1) {
    // (the same region holds the merged input cores where K-in's dense matrix is built from them: k_g2_dense)
    G2Mat mi;
    g2_plan_mat(&mi, rs.in_s, 4);
    const size_t mb = mi.ok ? g2_al((size_t)mi.head_elems * 4) + g2_al((size_t)mi.tail_elems * 4) : 0;
    if (w.lin < mb) w.lin = mb;
  }
  w.total = w.gin + w.bilv + w.rec + w.ident + w.wdense + w.planes + w.xpad + w.lin;
  return w;
}

size_t g2_rnn_fwd_workspace(const RnnShape& rs) { return g2_fwd_layout(rs).total; }

size_t g2_rnn_bwd_workspace(const RnnShape& rs) {
  G2Plan p;
  plan_for(&p, rs, true);
  return g2_bwd_ws_bytes(p.hid);
}

template <typename TS>
static int fwd_t(const RnnShape& rs, const G2Plan& P, int dtype, const void* x, const void* h0, const void* c0,
                 const float* packed_in, const void* bias_in, const float* packed_hid, const void* bias_hid, void* out,
                 void* hT, void* cT, float* reserve, void* workspace, hipStream_t stream) {
  const G2FwdWs L = g2_fwd_layout(rs);
  char* p = (char*)workspace;
  float* gin = (float*)p; p += L.gin;
  float* bilv = (float*)p; p += L.bilv;
  void* rec = p; p += L.rec;
  void* ident = p; p += L.ident;
  float* wdense = (float*)p; p += L.wdense;
  void* planes = p; p += L.planes;
  TS* xpad = (TS*)p; p += L.xpad;
  void* linws = p;
  const bool in1 = rs.in == 1;
  const int H = rs.H, ilv = rs.cell == TTRNN_LSTM ? 2 : 1;
  hipLaunchKernelGGL(k_g2_bias<TS>, dim3((H + 255) / 256), dim3(256), 0, stream, rs.cell, H,
                     rs.has_bias_in ? (const TS*)bias_in : (const TS*)nullptr,
                     rs.has_bias_hid ? (const TS*)bias_hid : (const TS*)nullptr, bilv);
  int st = check();
  if (st != TTRNN_OK) return st;
  if (in1) {
    const void* unit = unit_rows_ptr(TTRNN_F32);
    if (!unit) return TTRNN_ERR_LAUNCH;
    if (hipMemsetAsync(gin, 0, (size_t)4 * H * sizeof(float), stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
    const LinPlan lp = plan_ttlinear_fwd(rs.in_s, 1);
    st = launch_ttlinear_fwd(rs.in_s, lp, TTRNN_F32, 1, packed_in, nullptr, unit, gin, linws, stream, H, ilv);
  } else {
    const int inp = in_pad(rs.in);
    const int64_t rows = (int64_t)rs.B * rs.T;
    if (hipMemsetAsync(wdense, 0, (size_t)inp * 4 * H * sizeof(float), stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
    G2Mat mi;
    g2_plan_mat(&mi, rs.in_s, 4);
    if (mi.ok && (opt(OPT_DEV) & (1 << 24))) {
      // dev bit 24: the dense matrix from the merged cores (k_g2_merge + k_g2_dense, ~10 us) instead of the chain kernel on the identity rows
      // (83 us at the benchmark defaults) — an A/B that showed no difference in the harness' eval time (1.28 / 1.26 ms, within the noise) and
      // moved one outlier case of test_split_math_outlier_up across its bound (another order of the sums): not the default
      float* Ghi = (float*)linws;
      float* Gti = (float*)((char*)linws + g2_al((size_t)mi.head_elems * 4));
      hipLaunchKernelGGL(k_g2_merge, dim3((unsigned)g2_merge_blocks(mi)), dim3(256), 0, stream, rs.in_s, mi, packed_in, Ghi, Gti,
                         (unsigned*)nullptr);
      const long n = (long)mi.in * mi.out;
      hipLaunchKernelGGL(k_g2_dense, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mi, (const float*)Ghi, (const float*)Gti,
                         wdense, 4 * H, H, ilv);
      st = check();
    } else {
      st = launch_fill_identity(TTRNN_F32, rs.in, ident, stream);
      const LinPlan lp = plan_ttlinear_fwd(rs.in_s, rs.in);
      if (st == TTRNN_OK)
        st = launch_ttlinear_fwd(rs.in_s, lp, TTRNN_F32, rs.in, packed_in, nullptr, ident, wdense, linws, stream, H, ilv);
    }
    void* gscr = (char*)planes + gemm_split_plane_bytes(inp, 4 * H);      // two-piece fp16 GEMM: scales
    const bool ghalf = gemm_use_half(rows, inp, 4 * H);
    if (st == TTRNN_OK)
      st = ghalf ? launch_gemm_half_prep(wdense, inp, 4 * H, planes, gscr, stream)
                 : launch_gemm_split_prep(wdense, inp, 4 * H, planes, stream);
    const void* xg = x;
    if (st == TTRNN_OK && inp != rs.in) {
      const long n = (long)rows * inp;
      hipLaunchKernelGGL(k_g2_pad_rows<TS>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (long)rows, rs.in, inp,
                         (const TS*)x, xpad);
      st = check();
      xg = xpad;
    }
    if (st == TTRNN_OK)
      st = ghalf ? launch_gemm_half(dtype, rows, inp, 4 * H, xg, planes, gscr, nullptr, 0, gin, stream, bilv)
                 : launch_gemm_split(dtype, rows, inp, 4 * H, xg, planes, nullptr, 0, gin, stream, bilv);
  }
  if (st != TTRNN_OK) return st;
  // the reference's default benchmark shape (H = 512, r = 8) in split mode: the fused-core forward kernel on the gin just built
  // (both biases are folded into it), this tier's `rec` region as its fragment workspace (ttrnn_fast_f10.hip)
  // the fp32 TT-GRU shape with a fused-core forward kernel (H = 256, r = 8; input_size != 1 arrives here): that kernel on this gin
  if (dtype == TTRNN_F32 && !opt(OPT_FORCE_G2) && f10gh_available(rs, dtype)) {      // (either input size: GinSrc says which)
    if (L.rec < f10gh_workspace_bytes(rs)) return TTRNN_ERR_WORKSPACE;      // (never: g2_fwd_layout sizes it)
    GinSrc srcg{gin, x, in1 ? 1 : 0};
    return launch_gru_fwd_f10gh_g2(rs, srcg, bilv, h0, packed_hid, out, hT, reserve, rec, stream);
  }
  if (!in1 && dtype == TTRNN_F32 && !opt(OPT_FORCE_G2) && f10g5_available(rs, dtype)) {      // H = 512, r = 8: gates on the accumulators
    if (L.rec < f10g5_workspace_bytes(rs)) return TTRNN_ERR_WORKSPACE;
    return launch_gru_fwd_f10g5(rs, gin, h0, packed_hid, out, hT, reserve, rec, stream);
  }
  // the naive per-gate TT-LSTM / TT-GRU of H = 256, r = 8: one gate per wave on the fused-core scheme (ttrnn_fast_f10n.hip), either K-in
  if (dtype == TTRNN_F32 && !opt(OPT_FORCE_G2) && f10n_available(rs, dtype)) {
    if (L.rec < f10n_workspace_bytes(rs)) return TTRNN_ERR_WORKSPACE;
    GinSrc srcn{gin, x, in1 ? 1 : 0};
    return launch_lstm_fwd_f10n(rs, srcn, bilv, h0, c0, packed_hid, out, hT, cT, reserve, rec, stream);
  }
  if (f10_h512_fwd_available(rs, dtype)) {
    if (L.rec < f10_h512_workspace_bytes()) return TTRNN_ERR_WORKSPACE;      // (never: g2_fwd_layout sizes it)
    return launch_rnn_fwd_f10_h512(rs, gin, h0, c0, packed_hid, out, hT, cT, reserve, rec, stream);
  }
  const xbf8* fs2;
  const float* ft1;
  const int* hdr = nullptr;
  st = prep(rs.hid_s, P.hid, false, packed_hid, rec, &fs2, &ft1, stream, &hdr);
  if (st != TTRNN_OK) return st;
  GinSrc src{gin, x, in1 ? 1 : 0};
  if (P.pair) {
    const unsigned grid = (unsigned)((rs.B + G2_PAIR_NS - 1) / G2_PAIR_NS);
#define TT_G2_PAIR(CELLV, UPTV, IN1V, DG)                                                                                   \
  do {                                                                                                                     \
    auto kern = P.pair == 2 ? k_g2_fwd_p<CELLV, TS, UPTV, IN1V, false, 2> : k_g2_fwd_p<CELLV, TS, UPTV, IN1V, DG, 1>;      \
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), P.f_lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;           \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G2_NT_MAX), P.f_lds, stream, P, src, bilv, (const TS*)h0, (const TS*)c0,     \
                       reinterpret_cast<const xh8*>(fs2), ft1, hdr, (TS*)out, (TS*)hT, (TS*)cT, reserve);                  \
  } while (0)
    const bool dg = opt(OPT_DIAG) && reserve && P.upt == 1 && !in1 && rs.cell == TTRNN_LSTM && std::is_same<TS, float>::value && P.pair == 1;
    if (dg) {
      if constexpr (std::is_same<TS, float>::value) TT_G2_PAIR(TTRNN_LSTM, 1, false, true);
    } else if (rs.cell == TTRNN_LSTM) {
      if (P.upt == 1) { if (in1) TT_G2_PAIR(TTRNN_LSTM, 1, true, false); else TT_G2_PAIR(TTRNN_LSTM, 1, false, false); }
      else { if (in1) TT_G2_PAIR(TTRNN_LSTM, 2, true, false); else TT_G2_PAIR(TTRNN_LSTM, 2, false, false); }
    } else {
      if (P.upt == 1) { if (in1) TT_G2_PAIR(TTRNN_GRU, 1, true, false); else TT_G2_PAIR(TTRNN_GRU, 1, false, false); }
      else { if (in1) TT_G2_PAIR(TTRNN_GRU, 2, true, false); else TT_G2_PAIR(TTRNN_GRU, 2, false, false); }
    }
#undef TT_G2_PAIR
    return check();
  }
  const bool res = P.hid.UW * P.hid.KBP <= G2_PF;      // head fragments register-resident for every wave
  const bool res16 = g2_fwd_res16(P);                  // ... in sixteen slots (eight-wave workgroups, UPT <= 2)
#define TT_G2_FWD(CELLV, UPTV, SLOT)                                                                                      \
  do {                                                                                                                   \
    const bool p8 = P.hid.pack8 != 0;                                                                                   \
    if (res16 && UPTV <= 2) {                                                                                           \
      auto kern16 = in1 ? (p8 ? k_g2_fwd<CELLV, TS, (UPTV <= 2 ? UPTV : 1), true, false, true, 16, true>                 \
                              : k_g2_fwd<CELLV, TS, (UPTV <= 2 ? UPTV : 1), true, false, false, 16, true>)               \
                        : (p8 ? k_g2_fwd<CELLV, TS, (UPTV <= 2 ? UPTV : 1), true, false, true, 16>                       \
                              : k_g2_fwd<CELLV, TS, (UPTV <= 2 ? UPTV : 1), true, false, false, 16>);                    \
      if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern16), P.f_lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;     \
      hipLaunchKernelGGL(kern16, dim3(rs.B), dim3(P.hid.nw * 64), P.f_lds, stream, P, src, bilv, (const TS*)h0,          \
                         (const TS*)c0, reinterpret_cast<const xh8*>(fs2), ft1, hdr, (TS*)out, (TS*)hT, (TS*)cT, reserve); \
      break;                                                                                                             \
    }                                                                                                                    \
    if (P.hid.cin) {                                                                                                     \
      auto kernc = in1 ? (p8 ? k_g2_fwd<CELLV, TS, UPTV, true, false, true, G2_PF, true, true>                           \
                             : k_g2_fwd<CELLV, TS, UPTV, true, false, false, G2_PF, true, true>)                         \
                       : (p8 ? k_g2_fwd<CELLV, TS, UPTV, true, false, true, G2_PF, false, true>                          \
                             : k_g2_fwd<CELLV, TS, UPTV, true, false, false, G2_PF, false, true>);                       \
      if (ensure_dynamic_lds(reinterpret_cast<const void*>(kernc), P.f_lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;      \
      hipLaunchKernelGGL(kernc, dim3(rs.B), dim3(P.hid.nw * 64), P.f_lds, stream, P, src, bilv, (const TS*)h0,           \
                         (const TS*)c0, reinterpret_cast<const xh8*>(fs2), ft1, hdr, (TS*)out, (TS*)hT, (TS*)cT, reserve); \
      break;                                                                                                             \
    }                                                                                                                    \
    const bool dg = opt(OPT_DIAG) && reserve && UPTV == 1 && res && !in1;                                               \
    auto kern = dg ? (p8 ? k_g2_fwd<CELLV, TS, UPTV, true, true, true> : k_g2_fwd<CELLV, TS, UPTV, true, true, false>)   \
              : in1 ? (res ? (p8 ? k_g2_fwd<CELLV, TS, UPTV, true, false, true, G2_PF, true>                             \
                                 : k_g2_fwd<CELLV, TS, UPTV, true, false, false, G2_PF, true>)                           \
                           : (p8 ? k_g2_fwd<CELLV, TS, UPTV, false, false, true, G2_PF, true>                            \
                                 : k_g2_fwd<CELLV, TS, UPTV, false, false, false, G2_PF, true>))                         \
              : (res ? (p8 ? k_g2_fwd<CELLV, TS, UPTV, true, false, true> : k_g2_fwd<CELLV, TS, UPTV, true, false, false>) \
                     : (p8 ? k_g2_fwd<CELLV, TS, UPTV, false, false, true> : k_g2_fwd<CELLV, TS, UPTV, false, false, false>)); \
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), P.f_lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;        \
    hipLaunchKernelGGL(kern, dim3(rs.B), dim3(P.hid.nw * 64), P.f_lds, stream, P, src, bilv, (const TS*)h0, (const TS*)c0,       \
                       reinterpret_cast<const xh8*>(fs2), ft1, hdr, (TS*)out, (TS*)hT,                                   \
                       (TS*)cT, reserve);                                                                                 \
  } while (0)
  if (rs.cell == TTRNN_LSTM) {
    if (P.upt == 1) TT_G2_FWD(TTRNN_LSTM, 1, 0);
    else if (P.upt == 2) TT_G2_FWD(TTRNN_LSTM, 2, 1);
    else TT_G2_FWD(TTRNN_LSTM, 4, 2);
  } else {
    if (P.upt == 1) TT_G2_FWD(TTRNN_GRU, 1, 0);
    else if (P.upt == 2) TT_G2_FWD(TTRNN_GRU, 2, 1);
    else TT_G2_FWD(TTRNN_GRU, 4, 2);
  }
#undef TT_G2_FWD
  return check();
}

int launch_rnn_fwd_g2(const RnnShape& rs, int dtype, const void* x, const void* h0, const void* c0, const float* packed_in,
                      const void* bias_in, const float* packed_hid, const void* bias_hid, void* out, void* hT, void* cT,
                      float* reserve, void* workspace, hipStream_t stream) {
  G2Plan P;
  plan_for(&P, rs, false);
  if (!P.okf) return TTRNN_ERR_UNSUPPORTED;
  return dtype == TTRNN_F32
             ? fwd_t<float>(rs, P, dtype, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT, reserve, workspace, stream)
             : fwd_t<bf16_t>(rs, P, dtype, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT, reserve, workspace, stream);
}

template <typename TS>
static int bwd_t(const RnnShape& rs, const G2Plan& P, const void* out, const void* h0, const void* c0, const float* packed_hid,
                 const float* reserve, const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid,
                 void* d_h0, void* d_c0, void* ws, hipStream_t stream, float* dstate, float* stats) {
  const xbf8* bs2;
  const float *bt1, *hun = nullptr;
  int st = prep(rs.hid_s, P.hid, true, packed_hid, ws, &bs2, &bt1, stream, nullptr, &hun);
  if (st != TTRNN_OK) return st;
  unsigned* colmax = (stats && P.b_cmx > 0) ? reinterpret_cast<unsigned*>(stats) : nullptr;
  if (colmax && hipMemsetAsync(colmax, 0, (size_t)2 * rs.G * rs.H * sizeof(float), stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
  unsigned long long* diag = reinterpret_cast<unsigned long long*>((char*)ws + g2_bwd_ws_bytes(P.hid) - 4096);
  const size_t lds_b = (size_t)P.b_lds + (colmax ? P.b_cmx : 0);
  // head^T fragments register-resident: one column tile, <= 4 live k-blocks per unit (ng > 1: a unit's gate range only — the naive
  // per-gate sets of cfg2's size have ONE live block per unit, eight units per wave: streamed, they re-read eight padded blocks
  // per unit from L2 every step, 10.6 ms of a 14 ms training step)
  const bool res = g2_bwd_res(P);
#ifdef TTRNN_ABLATIONS
  G2Plan Pa = P;
  Pa.abl = opt(OPT_DEV) & (8 | 16 | 32 | 64);
  if (getenv("TTRNN_G2_PLAN"))
    fprintf(stderr, "g2 bwd plan: nw %d s %d It %d Jt %d Ih %d Jh %d R %d ng %d | bM2T %d N2T %d bNKBt %d bUW %d bSW %d | bM1T %d N1T %d bKB1 %d bK1SPLIT %d | "
            "lds dy %d dc1 %d dh %d tab %d (hun %d) t1 %d cmx %d = %d | upt %d res %d\n", P.hid.nw, P.hid.s, P.hid.It, P.hid.Jt, P.hid.Ih,
            P.hid.Jh, P.hid.R, P.hid.ng, P.hid.bM2T, P.hid.N2T, P.hid.bNKBt, P.hid.bUW, P.hid.bSW, P.hid.bM1T, P.hid.N1T, P.hid.bKB1,
            P.hid.bK1SPLIT, P.b_dy, P.b_dc1, P.b_dh, P.b_tab, P.b_hun, P.b_t1, P.b_cmx, P.b_lds, P.b_upt, (int)res);
#define P Pa
#endif
#define TT_G2_BWD(CELLV, UPTV, SLOT)                                                                                      \
  do {                                                                                                                   \
    auto kern = res ? k_g2_bwd<CELLV, TS, UPTV, true> : k_g2_bwd<CELLV, TS, UPTV, false>;                                \
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_b) != TTRNN_OK) return TTRNN_ERR_LAUNCH;          \
    hipLaunchKernelGGL(kern, dim3(rs.B), dim3(P.hid.nw * 64), lds_b, stream, P, (const TS*)out, (const TS*)h0, (const TS*)c0,     \
                       reserve, (const TS*)d_out, (const TS*)d_hT, (const TS*)d_cT, reinterpret_cast<const xh8*>(bs2), bt1,    \
                       hun, dg_in, dg_hid, (TS*)d_h0,                                                                        \
                       (TS*)d_c0, dstate, colmax, diag);                                                                  \
  } while (0)
  if (rs.cell == TTRNN_LSTM) {
    if (P.b_upt == 1) TT_G2_BWD(TTRNN_LSTM, 1, 0);
    else if (P.b_upt == 2) TT_G2_BWD(TTRNN_LSTM, 2, 1);
    else TT_G2_BWD(TTRNN_LSTM, 4, 2);
  } else {
    if (P.b_upt == 1) TT_G2_BWD(TTRNN_GRU, 1, 0);
    else if (P.b_upt == 2) TT_G2_BWD(TTRNN_GRU, 2, 1);
    else TT_G2_BWD(TTRNN_GRU, 4, 2);
  }
#undef TT_G2_BWD
#ifdef TTRNN_ABLATIONS
#undef P
#endif
  return check();
}

int launch_rnn_bwd_g2(const RnnShape& rs, int dtype, const void* out, const void* h0, const void* c0, const float* packed_hid,
                      const float* reserve, const void* d_out, const void* d_hT, const void* d_cT, float* dg_in,
                      float* dg_hid, void* d_h0, void* d_c0, void* ws, hipStream_t stream, float* dstate, float* stats) {
  G2Plan P;
  plan_for(&P, rs, true);
  if (!P.okb) return TTRNN_ERR_UNSUPPORTED;
  if (stats && P.b_cmx == 0) return TTRNN_ERR_UNSUPPORTED;
  return dtype == TTRNN_F32
             ? bwd_t<float>(rs, P, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, dg_hid, d_h0, d_c0, ws, stream, dstate, stats)
             : bwd_t<bf16_t>(rs, P, out, h0, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, dg_hid, d_h0, d_c0, ws, stream, dstate, stats);
}

// does the reverse-time kernel of this shape deliver the column maxima (room for them in LDS)?
bool g2_rnn_bwd_colmax(const RnnShape& rs) {
  G2Plan P;
  plan_for(&P, rs, true);
  return P.okb && P.b_cmx > 0;
}

}  // namespace ttrnn
