// ttrnn_core.h — shapes, index maps and the any-shape ("generic") TT-chain / recurrent-cell bodies.
//
// Everything here is written against an executor concept `Ex`:
//     ex.par([&](int tid, int nthr) { ... });   // run the body for every thread, then barrier
// On the GPU `DevExec` maps it to threadIdx.x + __syncthreads(); tests/hostemu re-uses the very
// same bodies with a serial host executor to check the index arithmetic without a GPU.  The host
// executor is test infrastructure only; the product library (libttrnn.so) contains DevExec only.
//
// Stage algebra (restates t3nsor/ops.py:78-93 of the reference; see include/ttrnn.h):
//   stage k (k = d-1 .. 0), per sample:
//     A_k [rows_k][K_k]        rows_k = prod_{m>k} I_m * prod_{m<k} J_m,  K_k = J_k * R_{k+1}
//     C_k [I_k][rows_k][R_k]   C_k[i][row][a] = sum_kk A_k[row][kk] * W_k[kk][i*R_k + a]
//   and the flat C_k buffer IS A_{k-1} (ops.py:89-90 `.contiguous().view`), C_0 is y[out].
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include "ttrnn.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TT_HD __host__ __device__ __forceinline__
#else
#define TT_HD inline
#endif

namespace ttrnn {

// ------------------------------------------------------------------------------------------------
// storage types
// ------------------------------------------------------------------------------------------------
struct bf16_t { uint16_t v; };

TT_HD float bf16_to_f32(bf16_t b) {
  union { uint32_t u; float f; } c; c.u = ((uint32_t)b.v) << 16; return c.f;
}
TT_HD bf16_t f32_to_bf16(float f) {   // round-to-nearest-even, NaN stays NaN
  union { uint32_t u; float f; } c; c.f = f;
  bf16_t r;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) { r.v = (uint16_t)((c.u >> 16) | 0x0040u); return r; }
  c.u += 0x7fffu + ((c.u >> 16) & 1u);
  r.v = (uint16_t)(c.u >> 16);
  return r;
}
TT_HD float to_f32(float v) { return v; }
TT_HD float to_f32(bf16_t v) { return bf16_to_f32(v); }
TT_HD float ld(const float* p, size_t i) { return p[i]; }
TT_HD float ld(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }
TT_HD void st(float* p, size_t i, float v) { p[i] = v; }
TT_HD void st(bf16_t* p, size_t i, float v) { p[i] = f32_to_bf16(v); }
// v after one round trip through the storage type of p (p itself is not touched: it may be NULL)
TT_HD float round_as(const float*, float v) { return v; }
TT_HD float round_as(const bf16_t*, float v) { return bf16_to_f32(f32_to_bf16(v)); }

// ------------------------------------------------------------------------------------------------
// shapes
// ------------------------------------------------------------------------------------------------
struct TtShape {
  int d, in_size, out_size;
  int J[TTRNN_MAX_D], I[TTRNN_MAX_D], R[TTRNN_MAX_D + 1];
  int K[TTRNN_MAX_D], M[TTRNN_MAX_D], rows[TTRNN_MAX_D];
  int woff[TTRNN_MAX_D];    // float offset of W_k  [K_k][M_k] in the packed buffer
  int wtoff[TTRNN_MAX_D];   // float offset of Wt_k [M_k][K_k]
  int wtotal;               // sum_k K_k*M_k  (packed buffer holds 2*wtotal floats)
  int maxbuf;               // floats per sample a chain pass needs in each ping-pong buffer
};

// host-side validation + derivation; returns a ttrnn_status
inline int tt_shape_init(TtShape* s, const ttrnn_ttm* w) {
  if (!s || !w) return TTRNN_ERR_NULL;
  if (w->d < 1 || w->d > TTRNN_MAX_D) return TTRNN_ERR_BAD_DESC;
  s->d = w->d;
  if (w->ranks[0] != 1 || w->ranks[w->d] != 1) return TTRNN_ERR_BAD_DESC;
  int64_t in = 1, out = 1;
  for (int k = 0; k < w->d; ++k) {
    if (w->in_modes[k] < 1 || w->out_modes[k] < 1 || w->ranks[k] < 1) return TTRNN_ERR_BAD_DESC;
    s->J[k] = w->in_modes[k]; s->I[k] = w->out_modes[k]; s->R[k] = w->ranks[k];
    in *= s->J[k]; out *= s->I[k];
    if (in > (1 << 24) || out > (1 << 24)) return TTRNN_ERR_BAD_DESC;
  }
  s->R[w->d] = 1;
  for (int k = w->d; k < TTRNN_MAX_D; ++k) { s->J[k] = s->I[k] = 1; s->R[k + 1] = 1; }
  s->in_size = (int)in; s->out_size = (int)out;
  int64_t off = 0, maxbuf = in > out ? in : out;
  for (int k = 0; k < TTRNN_MAX_D; ++k) { s->K[k] = s->M[k] = s->rows[k] = 0; s->woff[k] = s->wtoff[k] = 0; }
  for (int k = 0; k < w->d; ++k) {
    int64_t rows = 1;
    for (int m = k + 1; m < w->d; ++m) rows *= s->I[m];
    for (int m = 0; m < k; ++m) rows *= s->J[m];
    s->K[k] = s->J[k] * s->R[k + 1];
    s->M[k] = s->I[k] * s->R[k];
    if (rows * s->K[k] > (1 << 26) || rows * s->M[k] > (1 << 26)) return TTRNN_ERR_BAD_DESC;
    s->rows[k] = (int)rows;
    s->woff[k] = (int)off;
    off += (int64_t)s->K[k] * s->M[k];
    if (off > (1 << 28)) return TTRNN_ERR_BAD_DESC;
    if (rows * s->K[k] > maxbuf) maxbuf = rows * s->K[k];
    if (rows * s->M[k] > maxbuf) maxbuf = rows * s->M[k];
  }
  s->wtotal = (int)off;
  for (int k = 0; k < w->d; ++k) s->wtoff[k] = s->wtotal + s->woff[k];
  s->maxbuf = (int)maxbuf;
  return TTRNN_OK;
}

struct RnnShape {
  int cell, B, T, in, H, G;      // G = 4 (LSTM) / 3 (GRU)
  int has_bias_in, has_bias_hid;
  int hid_blocks;                // ttrnn_rnn_desc::hid_blocks (1 = general)
  TtShape in_s, hid_s;
  int bs;                        // per-sample stride (floats) of the ping-pong buffers
};

inline int rnn_shape_init(RnnShape* r, const ttrnn_rnn_desc* d) {
  if (!r || !d) return TTRNN_ERR_NULL;
  if (d->cell != TTRNN_LSTM && d->cell != TTRNN_GRU) return TTRNN_ERR_BAD_DESC;
  if (d->dtype != TTRNN_F32 && d->dtype != TTRNN_BF16) return TTRNN_ERR_BAD_DESC;
  if (d->batch < 0 || d->seq_len < 0 || d->input_size < 1 || d->hidden_size < 1) return TTRNN_ERR_BAD_DESC;
  int st = tt_shape_init(&r->in_s, &d->in_w);
  if (st != TTRNN_OK) return st;
  st = tt_shape_init(&r->hid_s, &d->hid_w);
  if (st != TTRNN_OK) return st;
  r->cell = d->cell; r->B = d->batch; r->T = d->seq_len; r->in = d->input_size; r->H = d->hidden_size;
  r->G = d->cell == TTRNN_LSTM ? 4 : 3;
  r->has_bias_in = d->has_bias_in; r->has_bias_hid = d->has_bias_hid;
  if (d->hid_blocks < 0 || d->hid_blocks > 8) return TTRNN_ERR_BAD_DESC;
  r->hid_blocks = d->hid_blocks > 1 ? d->hid_blocks : 1;
  if (r->in_s.in_size != r->in || r->hid_s.in_size != r->H) return TTRNN_ERR_BAD_DESC;
  if (r->in_s.out_size != r->G * r->H || r->hid_s.out_size != r->G * r->H) return TTRNN_ERR_BAD_DESC;
  int mb = r->in_s.maxbuf > r->hid_s.maxbuf ? r->in_s.maxbuf : r->hid_s.maxbuf;
  r->bs = (mb + 3) & ~3;
  return TTRNN_OK;
}

// ------------------------------------------------------------------------------------------------
// generic stage bodies (one thread's share)
// ------------------------------------------------------------------------------------------------
template <typename T> TT_HD T tmin(T a, T b) { return a < b ? a : b; }

// Training reserve (include/ttrnn.h).  LSTM: the activated gates [B*T][H][4] (i,g,f,o: one 16-byte record per hidden unit)
// followed by the cell states [B*T][H] — five floats per unit and step, not a padded record of eight.  GRU: [B*T][H][4]
// (r,z,n,hid_n).  `rows` = B*T.
TT_HD size_t res_gate(size_t bt, int H, int j) { return (bt * (size_t)H + j) * 4; }
TT_HD size_t res_cell(size_t rows, size_t bt, int H, int j) { return rows * (size_t)H * 4 + bt * (size_t)H + j; }

// C[s][i][row][a] = sum_kk A[s][row][kk] * W[kk][i*R+a]      (thread <- (s, 4-row tile, m))
TT_HD void stage_fwd(int tid, int nthr, const float* A, int sA, float* C, int sC, const float* W,
                     int nb, int rows, int K, int I, int R) {
  constexpr int RT = 4;
  const int M = I * R;
  const int rtiles = (rows + RT - 1) / RT;
  const int total = nb * rtiles * M;
  for (int idx = tid; idx < total; idx += nthr) {
    const int m = idx % M;
    const int q = idx / M;
    const int rt = q % rtiles;
    const int s = q / rtiles;
    const int i = m / R, a = m - i * R;
    const int row0 = rt * RT;
    const int nr = tmin(RT, rows - row0);
    const float* Ab = A + (size_t)s * sA + (size_t)row0 * K;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    if (nr == RT) {
      // the core lives in global memory (L2): eight loads are issued before the first one is consumed — one load per
      // iteration left every FMA group waiting a full L2 round trip (same sums in the same order)
      int kk = 0;
      for (; kk + 8 <= K; kk += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = W[(size_t)(kk + u) * M + m];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          acc0 = fmaf(Ab[kk + u], w[u], acc0);
          acc1 = fmaf(Ab[K + kk + u], w[u], acc1);
          acc2 = fmaf(Ab[2 * K + kk + u], w[u], acc2);
          acc3 = fmaf(Ab[3 * K + kk + u], w[u], acc3);
        }
      }
      for (; kk < K; ++kk) {
        const float w = W[(size_t)kk * M + m];
        acc0 = fmaf(Ab[kk], w, acc0);
        acc1 = fmaf(Ab[K + kk], w, acc1);
        acc2 = fmaf(Ab[2 * K + kk], w, acc2);
        acc3 = fmaf(Ab[3 * K + kk], w, acc3);
      }
    } else {
      for (int kk = 0; kk < K; ++kk) {
        const float w = W[(size_t)kk * M + m];
        acc0 = fmaf(Ab[kk], w, acc0);
        if (nr > 1) acc1 = fmaf(Ab[K + kk], w, acc1);
        if (nr > 2) acc2 = fmaf(Ab[2 * K + kk], w, acc2);
      }
    }
    float* Cb = C + (size_t)s * sC + (size_t)i * rows * R + (size_t)row0 * R + a;
    Cb[0] = acc0;
    if (nr > 1) Cb[R] = acc1;
    if (nr > 2) Cb[2 * R] = acc2;
    if (nr > 3) Cb[3 * R] = acc3;
  }
}

// dA[s][row][kk] = sum_m dC[s][i][row][a] * Wt[m][kk]          (thread <- (s, 4-row tile, kk))
TT_HD void stage_bwd_data(int tid, int nthr, const float* dC, int sC, float* dA, int sA,
                          const float* Wt, int nb, int rows, int K, int I, int R) {
  constexpr int RT = 4;
  const int rtiles = (rows + RT - 1) / RT;
  const int total = nb * rtiles * K;
  for (int idx = tid; idx < total; idx += nthr) {
    const int kk = idx % K;
    const int q = idx / K;
    const int rt = q % rtiles;
    const int s = q / rtiles;
    const int row0 = rt * RT;
    const int nr = tmin(RT, rows - row0);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    const float* Cs = dC + (size_t)s * sC + (size_t)row0 * R;
    if (nr == RT && R % 8 == 0) {
      // full row tile: no per-row conditions inside the product loop (they kept the compiler from pairing the loads of the
      // four rows with the FMAs: a classifier head's data-gradient chain ran four times as long as its forward chain)
      for (int i = 0; i < I; ++i) {
        const float* Ci = Cs + (size_t)i * rows * R;
        for (int a = 0; a < R; a += 8) {
          float w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) w[u] = Wt[(size_t)(i * R + a + u) * K + kk];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            acc0 = fmaf(Ci[a + u], w[u], acc0);
            acc1 = fmaf(Ci[R + a + u], w[u], acc1);
            acc2 = fmaf(Ci[2 * R + a + u], w[u], acc2);
            acc3 = fmaf(Ci[3 * R + a + u], w[u], acc3);
          }
        }
      }
    } else
    for (int i = 0; i < I; ++i) {
      const float* Ci = Cs + (size_t)i * rows * R;
      int a = 0;
      for (; a + 8 <= R; a += 8) {          // eight transposed-core loads in flight (see stage_fwd)
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = Wt[(size_t)(i * R + a + u) * K + kk];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          acc0 = fmaf(Ci[a + u], w[u], acc0);
          if (nr > 1) acc1 = fmaf(Ci[R + a + u], w[u], acc1);
          if (nr > 2) acc2 = fmaf(Ci[2 * R + a + u], w[u], acc2);
          if (nr > 3) acc3 = fmaf(Ci[3 * R + a + u], w[u], acc3);
        }
      }
      for (; a < R; ++a) {
        const float w = Wt[(size_t)(i * R + a) * K + kk];
        acc0 = fmaf(Ci[a], w, acc0);
        if (nr > 1) acc1 = fmaf(Ci[R + a], w, acc1);
        if (nr > 2) acc2 = fmaf(Ci[2 * R + a], w, acc2);
        if (nr > 3) acc3 = fmaf(Ci[3 * R + a], w, acc3);
      }
    }
    float* Ab = dA + (size_t)s * sA + (size_t)row0 * K + kk;
    Ab[0] = acc0;
    if (nr > 1) Ab[K] = acc1;
    if (nr > 2) Ab[2 * K] = acc2;
    if (nr > 3) Ab[3 * K] = acc3;
  }
}

// dWacc[kk][m] += sum_{s,row} A[s][row][kk] * dC[s][i][row][a]   (thread <- (4-kk tile, m))
// `add(ptr, v)` performs the accumulation (plain += into LDS, or an atomic into global).
template <class Add>
TT_HD void stage_bwd_weight(int tid, int nthr, const float* A, int sA, const float* dC, int sC,
                            float* dWacc, int nb, int rows, int K, int I, int R, Add add) {
  constexpr int KT = 4;
  const int M = I * R;
  const int ktiles = (K + KT - 1) / KT;
  const int total = ktiles * M;
  for (int idx = tid; idx < total; idx += nthr) {
    const int m = idx % M;
    const int kt = idx / M;
    const int i = m / R, a = m - i * R;
    const int k0 = kt * KT;
    const int nk = tmin(KT, K - k0);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int s = 0; s < nb; ++s) {
      const float* As = A + (size_t)s * sA + k0;
      const float* Cs = dC + (size_t)s * sC + (size_t)i * rows * R + a;
      int row = 0;
      if (nk == KT) {
        // eight rows' operands are requested before the first one is consumed: with the stage inputs in the global workspace
        // (samples too large for LDS) one row per iteration left every FMA group waiting a full L2 round trip — same sums in
        // the same order
        constexpr int RU = 8;
        for (; row + RU <= rows; row += RU) {
          float av[RU][KT], dv[RU];
#pragma unroll
          for (int u = 0; u < RU; ++u) {
            const float* Ar = As + (size_t)(row + u) * K;
#pragma unroll
            for (int j = 0; j < KT; ++j) av[u][j] = Ar[j];
            dv[u] = Cs[(size_t)(row + u) * R];
          }
#pragma unroll
          for (int u = 0; u < RU; ++u) {
            acc0 = fmaf(av[u][0], dv[u], acc0);
            acc1 = fmaf(av[u][1], dv[u], acc1);
            acc2 = fmaf(av[u][2], dv[u], acc2);
            acc3 = fmaf(av[u][3], dv[u], acc3);
          }
        }
      }
      for (; row < rows; ++row) {
        const float dc = Cs[(size_t)row * R];
        const float* Ar = As + (size_t)row * K;
        acc0 = fmaf(Ar[0], dc, acc0);
        if (nk > 1) acc1 = fmaf(Ar[1], dc, acc1);
        if (nk > 2) acc2 = fmaf(Ar[2], dc, acc2);
        if (nk > 3) acc3 = fmaf(Ar[3], dc, acc3);
      }
    }
    float* Wp = dWacc + (size_t)k0 * M + m;
    add(Wp, acc0);
    if (nk > 1) add(Wp + M, acc1);
    if (nk > 2) add(Wp + 2 * M, acc2);
    if (nk > 3) add(Wp + 3 * M, acc3);
  }
}

// ------------------------------------------------------------------------------------------------
// chain passes over nb samples; `cur` holds [nb][in] on entry (stride bs); returns the buffer that
// holds [nb][out] on exit.  W points at the packed cores' forward half (W_k at W + woff[k]).
// ------------------------------------------------------------------------------------------------
template <class Ex>
TT_HD float* chain_fwd(Ex& ex, const TtShape& s, const float* W, float* cur, float* nxt, int nb, int bs) {
  for (int k = s.d - 1; k >= 0; --k) {
    const float* Wk = W + s.woff[k];
    const int rows = s.rows[k], K = s.K[k], I = s.I[k], R = s.R[k];
    float* a = cur; float* c = nxt;
    ex.par([&](int tid, int nthr) { stage_fwd(tid, nthr, a, bs, c, bs, Wk, nb, rows, K, I, R); });
    cur = c; nxt = a;
  }
  return cur;
}

// `cur` holds d y [nb][out]; returns the buffer holding d x [nb][in].  Wt points at the TRANSPOSED
// half of the packed cores (packed + wtotal), i.e. Wt_k = Wt + woff[k].
template <class Ex>
TT_HD float* chain_bwd_data(Ex& ex, const TtShape& s, const float* Wt, float* cur, float* nxt, int nb, int bs) {
  for (int k = 0; k < s.d; ++k) {
    const float* Wtk = Wt + s.woff[k];
    const int rows = s.rows[k], K = s.K[k], I = s.I[k], R = s.R[k];
    float* c = cur; float* a = nxt;
    ex.par([&](int tid, int nthr) { stage_bwd_data(tid, nthr, c, bs, a, bs, Wtk, nb, rows, K, I, R); });
    cur = a; nxt = c;
  }
  return cur;
}

TT_HD float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// TTLinear forward over a tile of nb rows:  y = chain(x) + bias     (layers.py:121-127)
// ------------------------------------------------------------------------------------------------
// ilv_h > 0 selects the gate-interleaved output used by the hoisted input projection of the recurrent
// kernels: y[n][o], o = g*ilv_h + hid, is stored at ((n*ilv_h + hid)*4 + slot(g)); ilv_mode 1: slot = g,
// ilv_mode 2 (LSTM): slots ordered i,g,f,o so that each half-wave of the fused kernel loads one 8-byte pair.
TT_HD size_t ilv_index(int64_t n, int o, int out, int ilv_h, int ilv_mode) {
  if (ilv_h <= 0) return (size_t)n * out + o;
  const int g = o / ilv_h, hid = o - g * ilv_h;
  const int slot = (ilv_mode == 2) ? (g == 1 ? 2 : (g == 2 ? 1 : g)) : g;
  return ((size_t)n * ilv_h + hid) * 4 + slot;
}

// epi != 0 (TTRNN_EPI_LOG_SOFTMAX = 1 / TTRNN_EPI_RELU_L2NORM = 2, include/ttrnn.h; plain layout only): the row-wise epilogue of
// the callers' heads (mnist_classifier.py:55-57, speaker_encoder.py:86-89) applied to the tile's rows while they are still in the
// chain's buffer — one thread per row walks its `out` values: offered for NARROW heads only (out <= 32, the 10-class classifiers;
// ttrnn_head_forward keeps the wave-per-row k_head_epilogue launch for wider ones such as the 256-wide speaker head) — and aux[n] keeps the row's
// log-sum-exp / L2 norm for the backward pass: the head's forward is ONE launch (round 5; a separate k_head_epilogue before)
template <class Ex, typename T>
TT_HD void ttlinear_fwd_tile(Ex& ex, const TtShape& s, const float* W, const T* bias, const T* x, T* y,
                             int64_t n0, int nb, float* bufA, float* bufB, int bs, int ilv_h = 0, int ilv_mode = 0,
                             int epi = 0, float* aux = nullptr) {
  const int in = s.in_size, out = s.out_size;
  ex.par([&](int tid, int nthr) {
    for (int e = tid; e < nb * in; e += nthr) {
      const int sidx = e / in, j = e - sidx * in;
      bufA[(size_t)sidx * bs + j] = ld(x, (size_t)(n0 + sidx) * in + j);
    }
  });
  float* r = chain_fwd(ex, s, W, bufA, bufB, nb, bs);
  if (epi != 0) {
    ex.par([&](int tid, int nthr) {
      for (int sidx = tid; sidx < nb; sidx += nthr) {
        float* row = r + (size_t)sidx * bs;
        // the linear output as the storage type holds it (bf16: rounded once, as the separate k_head_epilogue launch reads it
        // back from y) — the fused and the two-launch path compute the same thing
        for (int o = 0; o < out; ++o) row[o] = round_as(y, bias ? row[o] + ld(bias, o) : row[o]);
        float a;
        if (epi == 1) {
          float m = row[0];
          for (int o = 1; o < out; ++o) m = row[o] > m ? row[o] : m;
          float sum = 0.f;
          for (int o = 0; o < out; ++o) sum += expf(row[o] - m);
          a = m + logf(sum);
          for (int o = 0; o < out; ++o) row[o] -= a;
        } else {
          float sum = 0.f;
          for (int o = 0; o < out; ++o) { const float u = row[o] > 0.f ? row[o] : 0.f; sum += u * u; }
          a = sqrtf(sum);
          for (int o = 0; o < out; ++o) row[o] = (row[o] > 0.f ? row[o] : 0.f) / a;      // 0 / 0 = NaN, as torch.norm + div
        }
        if (aux) aux[n0 + sidx] = a;
      }
    });
    bias = nullptr;      // added above
  }
  ex.par([&](int tid, int nthr) {
    for (int e = tid; e < nb * out; e += nthr) {
      const int sidx = e / out, o = e - sidx * out;
      float v = r[(size_t)sidx * bs + o];
      if (bias) v += ld(bias, o);
      st(y, ilv_index(n0 + sidx, o, out, ilv_h, ilv_mode), v);
    }
  });
}

// ------------------------------------------------------------------------------------------------
// TTLinear backward over a tile of nb rows.
//   stash: [nb][stash_stride] floats holding A_k for every stage (recomputed forward)
//   dWacc: wtotal floats accumulator (LDS or global), dbacc: out floats accumulator or NULL
// ------------------------------------------------------------------------------------------------
TT_HD int stash_floats(const TtShape& s) {
  int t = 0;
  for (int k = 0; k < s.d; ++k) t += s.rows[k] * s.K[k];
  return (t + 3) & ~3;
}

template <class Ex, typename T, typename TDY, class Add>
TT_HD void ttlinear_bwd_tile(Ex& ex, const TtShape& s, const float* W, const float* Wt, const T* x, const TDY* dy, T* dx,
                             float* dWacc, float* dbacc, int64_t n0, int nb,
                             float* stash, int ss, float* bufA, float* bufB, int bs, Add add, bool stash_far = false) {
  const int in = s.in_size, out = s.out_size;
  // recompute the forward chain, keeping every stage input A_k in the stash
  int soff[TTRNN_MAX_D];
  {
    int o = 0;
    for (int k = s.d - 1; k >= 0; --k) { soff[k] = o; o += s.rows[k] * s.K[k]; }
  }
  if (dWacc) {
    float* a0 = stash + soff[s.d - 1];
    ex.par([&](int tid, int nthr) {
      for (int e = tid; e < nb * in; e += nthr) {
        const int sidx = e / in, j = e - sidx * in;
        a0[(size_t)sidx * ss + j] = ld(x, (size_t)(n0 + sidx) * in + j);
      }
    });
    if (!stash_far) {
      for (int k = s.d - 1; k >= 1; --k) {
        const float* Wk = W + s.woff[k];
        const int rows = s.rows[k], K = s.K[k], I = s.I[k], R = s.R[k];
        const float* a = stash + soff[k];
        float* c = stash + soff[k - 1];
        ex.par([&](int tid, int nthr) { stage_fwd(tid, nthr, a, ss, c, ss, Wk, nb, rows, K, I, R); });
      }
    } else {
      // the stash lives in the global workspace, bufA / bufB on chip (LinPlan::buf_mixed): run the chain through the two
      // on-chip buffers — every output is a K-long dot product over its stage input — and copy each stage input out
      float* cur = bufA; float* nxt = bufB;
      ex.par([&](int tid, int nthr) {
        for (int e = tid; e < nb * in; e += nthr) {
          const int sidx = e / in, j = e - sidx * in;
          cur[(size_t)sidx * bs + j] = a0[(size_t)sidx * ss + j];
        }
      });
      for (int k = s.d - 1; k >= 1; --k) {
        const float* Wk = W + s.woff[k];
        const int rows = s.rows[k], K = s.K[k], I = s.I[k], R = s.R[k];
        ex.par([&](int tid, int nthr) { stage_fwd(tid, nthr, cur, bs, nxt, bs, Wk, nb, rows, K, I, R); });
        float* c = stash + soff[k - 1];
        const int cnt = s.rows[k - 1] * s.K[k - 1];
        ex.par([&](int tid, int nthr) {
          for (int e = tid; e < nb * cnt; e += nthr) {
            const int sidx = e / cnt, j = e - sidx * cnt;
            c[(size_t)sidx * ss + j] = nxt[(size_t)sidx * bs + j];
          }
        });
        float* t = cur; cur = nxt; nxt = t;
      }
    }
  }
  // d y -> bufA ; bias grad
  ex.par([&](int tid, int nthr) {
    for (int e = tid; e < nb * out; e += nthr) {
      const int sidx = e / out, o = e - sidx * out;
      bufA[(size_t)sidx * bs + o] = ld(dy, (size_t)(n0 + sidx) * out + o);
    }
  });
  if (dbacc) {
    ex.par([&](int tid, int nthr) {
      for (int o = tid; o < out; o += nthr) {
        float acc = 0.f;
        for (int sidx = 0; sidx < nb; ++sidx) acc += bufA[(size_t)sidx * bs + o];
        add(dbacc + o, acc);
      }
    });
  }
  float* cur = bufA; float* nxt = bufB;
  for (int k = 0; k < s.d; ++k) {
    const int rows = s.rows[k], K = s.K[k], I = s.I[k], R = s.R[k];
    const float* Wtk = Wt + s.woff[k];
    float* c = cur; float* a = nxt;
    const bool need_data = (k + 1 < s.d) || dx;
    if (dWacc) {
      const float* Ak = stash + soff[k];
      float* dWk = dWacc + s.woff[k];
      // the weight-grad and data-grad parts of one stage read the same dC; no barrier needed between
      ex.par([&](int tid, int nthr) {
        stage_bwd_weight(tid, nthr, Ak, ss, c, bs, dWk, nb, rows, K, I, R, add);
        if (need_data) stage_bwd_data(tid, nthr, c, bs, a, bs, Wtk, nb, rows, K, I, R);
      });
    } else if (need_data) {
      ex.par([&](int tid, int nthr) { stage_bwd_data(tid, nthr, c, bs, a, bs, Wtk, nb, rows, K, I, R); });
    }
    cur = a; nxt = c;
  }
  if (dx) {
    float* r = cur;
    ex.par([&](int tid, int nthr) {
      for (int e = tid; e < nb * in; e += nthr) {
        const int sidx = e / in, j = e - sidx * in;
        st(dx, (size_t)(n0 + sidx) * in + j, r[(size_t)sidx * bs + j]);
      }
    });
  }
}

// ------------------------------------------------------------------------------------------------
// recurrent layer forward for samples [b0, b0+nb): the reference's time loop for one layer
// (lstm.py:123-133 / gru.py:124-134) with the cell arithmetic of lstm.py:24-32 / gru.py:33-44.
//   hbuf, cbuf: [nb][H]   gin: [nb][G*H]   bufA/bufB: [nb][bs]
// ------------------------------------------------------------------------------------------------
template <class Ex, typename T>
TT_HD void rnn_fwd_body(Ex& ex, const RnnShape& rs, int b0, int nb,
                        const T* x, const T* h0, const T* c0,
                        const float* Win, const T* bin, const float* Whid, const T* bhid,
                        T* out, T* hT, T* cT, float* reserve,
                        float* bufA, float* bufB, float* hbuf, float* cbuf, float* gin) {
  const int H = rs.H, G = rs.G, GH = G * H, in = rs.in, Tn = rs.T, bs = rs.bs;
  const bool lstm = rs.cell == TTRNN_LSTM;
  const size_t RROWS = (size_t)rs.B * rs.T;   // reserve layout: res_gate / res_cell
  ex.par([&](int tid, int nthr) {
    for (int e = tid; e < nb * H; e += nthr) {
      const int s = e / H, j = e - s * H;
      hbuf[e] = h0 ? ld(h0, (size_t)(b0 + s) * H + j) : 0.f;
      if (lstm) cbuf[e] = c0 ? ld(c0, (size_t)(b0 + s) * H + j) : 0.f;
    }
  });
  for (int t = 0; t < Tn; ++t) {
    // input TTLinear: gin = W_in x_t + b_in
    ex.par([&](int tid, int nthr) {
      for (int e = tid; e < nb * in; e += nthr) {
        const int s = e / in, j = e - s * in;
        bufA[(size_t)s * bs + j] = ld(x, ((size_t)(b0 + s) * Tn + t) * in + j);
      }
    });
    float* r = chain_fwd(ex, rs.in_s, Win, bufA, bufB, nb, bs);
    ex.par([&](int tid, int nthr) {
      for (int e = tid; e < nb * GH; e += nthr) {
        const int s = e / GH, o = e - s * GH;
        float v = r[(size_t)s * bs + o];
        if (bin) v += ld(bin, o);
        gin[e] = v;
      }
    });
    // hidden TTLinear on h_{t-1}
    ex.par([&](int tid, int nthr) {
      for (int e = tid; e < nb * H; e += nthr) {
        const int s = e / H, j = e - s * H;
        bufA[(size_t)s * bs + j] = hbuf[e];
      }
    });
    r = chain_fwd(ex, rs.hid_s, Whid, bufA, bufB, nb, bs);
    // gates + state update
    ex.par([&](int tid, int nthr) {
      for (int e = tid; e < nb * H; e += nthr) {
        const int s = e / H, j = e - s * H;
        const float* gi = gin + (size_t)s * GH;
        const float* gh = r + (size_t)s * bs;
        const size_t bt = (size_t)(b0 + s) * Tn + t;
        float hy;
        if (lstm) {
          float p0 = gi[j] + gh[j], p1 = gi[H + j] + gh[H + j];
          float p2 = gi[2 * H + j] + gh[2 * H + j], p3 = gi[3 * H + j] + gh[3 * H + j];
          if (bhid) { p0 += ld(bhid, j); p1 += ld(bhid, H + j); p2 += ld(bhid, 2 * H + j); p3 += ld(bhid, 3 * H + j); }
          const float ig = sigmoidf_(p0), fg = sigmoidf_(p1), gg = tanhf(p2), og = sigmoidf_(p3);
          const float cy = fg * cbuf[e] + ig * gg;
          hy = og * tanhf(cy);
          cbuf[e] = cy;
          if (reserve) {
            float* rv = reserve + res_gate(bt, H, j);
            rv[0] = ig; rv[1] = gg; rv[2] = fg; rv[3] = og;
            reserve[res_cell(RROWS, bt, H, j)] = cy;
          }
        } else {
          float hr = gh[j], hz = gh[H + j], hn = gh[2 * H + j];
          if (bhid) { hr += ld(bhid, j); hz += ld(bhid, H + j); hn += ld(bhid, 2 * H + j); }
          const float rg = sigmoidf_(gi[j] + hr);
          const float zg = sigmoidf_(gi[H + j] + hz);
          const float ng = tanhf(gi[2 * H + j] + rg * hn);
          hy = (1.0f - zg) * ng + zg * hbuf[e];
          if (reserve) {
            float* rv = reserve + res_gate(bt, H, j);
            rv[0] = rg; rv[1] = zg; rv[2] = ng; rv[3] = hn;
          }
        }
        // the stored output is what the next step (and the next layer) sees: round once
        // (out == NULL: the caller consumes only the final state — include/ttrnn.h, ttrnn_rnn_out_optional)
        if (out) st(out + bt * H, j, hy);
        hbuf[e] = round_as(out, hy);
      }
    });
  }
  ex.par([&](int tid, int nthr) {
    for (int e = tid; e < nb * H; e += nthr) {
      const int s = e / H, j = e - s * H;
      if (hT) st(hT, (size_t)(b0 + s) * H + j, hbuf[e]);
      if (lstm && cT) st(cT, (size_t)(b0 + s) * H + j, cbuf[e]);
    }
  });
}

// ------------------------------------------------------------------------------------------------
// reverse-time part of BPTT for samples [b0, b0+nb) (see include/ttrnn.h, ttrnn_rnn_backward)
//   dh, dc, dhd: [nb][H] fp32 state     bufA/bufB: [nb][bs]
// ------------------------------------------------------------------------------------------------
template <class Ex, typename T>
TT_HD void rnn_bwd_body(Ex& ex, const RnnShape& rs, int b0, int nb,
                        const T* out, const T* h0, const T* c0, const float* Wt_hid, const float* reserve,
                        const T* d_out, const T* d_hT, const T* d_cT,
                        float* dg_in, float* dg_hid, T* d_h0, T* d_c0,
                        float* bufA, float* bufB, float* dh, float* dc, float* dhd, float* dstate = nullptr) {
  const int H = rs.H, G = rs.G, GH = G * H, Tn = rs.T, bs = rs.bs;
  const bool lstm = rs.cell == TTRNN_LSTM;
  const size_t RROWS = (size_t)rs.B * rs.T;
  ex.par([&](int tid, int nthr) {
    for (int e = tid; e < nb * H; e += nthr) {
      const int s = e / H, j = e - s * H;
      dh[e] = d_hT ? ld(d_hT, (size_t)(b0 + s) * H + j) : 0.f;
      dc[e] = (lstm && d_cT) ? ld(d_cT, (size_t)(b0 + s) * H + j) : 0.f;
    }
  });
  for (int t = Tn - 1; t >= 0; --t) {
    ex.par([&](int tid, int nthr) {
      for (int e = tid; e < nb * H; e += nthr) {
        const int s = e / H, j = e - s * H;
        const size_t b = (size_t)(b0 + s);
        const size_t bt = b * Tn + t;
        const float* rv = reserve + res_gate(bt, H, j);
        float dht = dh[e];
        if (d_out) dht += ld(d_out, bt * H + j);
        float* ga = bufA + (size_t)s * bs;
        if (lstm) {
          const float ig = rv[0], gg = rv[1], fg = rv[2], og = rv[3], cy = reserve[res_cell(RROWS, bt, H, j)];
          const float cprev = t > 0 ? reserve[res_cell(RROWS, bt - 1, H, j)] : (c0 ? ld(c0, b * H + j) : 0.f);
          const float tc = tanhf(cy);
          const float dct = dc[e] + dht * og * (1.0f - tc * tc);
          if (dstate) { dstate[(bt * H + j) * 2] = dht; dstate[(bt * H + j) * 2 + 1] = dct; }
          const float p0 = dct * gg * ig * (1.0f - ig);
          const float p1 = dct * cprev * fg * (1.0f - fg);
          const float p2 = dct * ig * (1.0f - gg * gg);
          const float p3 = dht * tc * og * (1.0f - og);
          dc[e] = dct * fg;
          ga[j] = p0; ga[H + j] = p1; ga[2 * H + j] = p2; ga[3 * H + j] = p3;
          float* gp = dg_in + bt * GH;
          gp[j] = p0; gp[H + j] = p1; gp[2 * H + j] = p2; gp[3 * H + j] = p3;
          if (dg_hid && dg_hid != dg_in) {
            float* gq = dg_hid + bt * GH;
            gq[j] = p0; gq[H + j] = p1; gq[2 * H + j] = p2; gq[3 * H + j] = p3;
          }
        } else {
          const float rg = rv[0], zg = rv[1], ng = rv[2], hn = rv[3];
          const float hprev = t > 0 ? ld(out, (bt - 1) * H + j) : (h0 ? ld(h0, b * H + j) : 0.f);
          if (dstate) { dstate[(bt * H + j) * 2] = dht; dstate[(bt * H + j) * 2 + 1] = 0.f; }
          const float dn_pre = dht * (1.0f - zg) * (1.0f - ng * ng);
          const float dz_pre = dht * (hprev - ng) * zg * (1.0f - zg);
          const float dr_pre = dn_pre * hn * rg * (1.0f - rg);
          dhd[e] = dht * zg;
          ga[j] = dr_pre; ga[H + j] = dz_pre; ga[2 * H + j] = dn_pre * rg;
          float* gp = dg_in + bt * GH;
          gp[j] = dr_pre; gp[H + j] = dz_pre; gp[2 * H + j] = dn_pre;
          float* gq = dg_hid + bt * GH;
          gq[j] = dr_pre; gq[H + j] = dz_pre; gq[2 * H + j] = dn_pre * rg;
        }
      }
    });
    float* r = chain_bwd_data(ex, rs.hid_s, Wt_hid, bufA, bufB, nb, bs);
    ex.par([&](int tid, int nthr) {
      for (int e = tid; e < nb * H; e += nthr) {
        const int s = e / H, j = e - s * H;
        float v = r[(size_t)s * bs + j];
        if (!lstm) v += dhd[e];
        dh[e] = v;
      }
    });
  }
  ex.par([&](int tid, int nthr) {
    for (int e = tid; e < nb * H; e += nthr) {
      const int s = e / H, j = e - s * H;
      if (d_h0) st(d_h0, (size_t)(b0 + s) * H + j, dh[e]);
      if (lstm && d_c0) st(d_c0, (size_t)(b0 + s) * H + j, dc[e]);
    }
  });
}

// ------------------------------------------------------------------------------------------------
// core packing (one thread's share): strided G_k[a,i,j,b] -> W_k[(j,b)][(i,a)] and Wt_k
// ------------------------------------------------------------------------------------------------
template <typename T>
TT_HD void pack_core_elems(int64_t tid, int64_t nthr, const TtShape& s, int k, const T* core,
                           const int64_t* st4, float* packed) {
  const int R0 = s.R[k], R1 = s.R[k + 1];
  const int K = s.K[k], M = s.M[k];
  const int64_t n = (int64_t)K * M;
  for (int64_t e = tid; e < n; e += nthr) {
    const int m = (int)(e % M), kk = (int)(e / M);
    const int i = m / R0, a = m - i * R0;
    const int j = kk / R1, b = kk - j * R1;
    const float v = ld(core, (size_t)(a * st4[0] + i * st4[1] + j * st4[2] + b * st4[3]));
    packed[s.woff[k] + e] = v;
    packed[s.wtoff[k] + (size_t)m * K + kk] = v;
  }
}

template <typename T>
TT_HD void unpack_core_grad_elems(int64_t tid, int64_t nthr, const TtShape& s, int k, const float* packed_grad,
                                  T* grad, const int64_t* st4) {
  const int R0 = s.R[k], R1 = s.R[k + 1];
  const int K = s.K[k], M = s.M[k];
  const int64_t n = (int64_t)K * M;
  for (int64_t e = tid; e < n; e += nthr) {
    const int m = (int)(e % M), kk = (int)(e / M);
    const int i = m / R0, a = m - i * R0;
    const int j = kk / R1, b = kk - j * R1;
    st(grad, (size_t)(a * st4[0] + i * st4[1] + j * st4[2] + b * st4[3]), packed_grad[s.woff[k] + e]);
  }
}

}  // namespace ttrnn
