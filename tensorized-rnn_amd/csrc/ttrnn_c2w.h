// ttrnn_c2w.h — plan of the CHAIN weight-gradient kernel (ttrnn_fast_c2w.hip): the weight gradients of a recurrent layer's two
// TT-matrices taken through the tensor-train chain itself instead of through the dense in x out matrix, for the shapes whose
// chain is the cheaper contraction — low ranks on large modes, i.e. the reference's own published speaker-verification
// encoder (experiments/speaker_verification/encoder/params_model.py:2-4,14-16: H = 768, n_cores = 2, rank = 2: 0.98 M
// multiply-adds per row through the chain against 4.7 M for the dense 768 x 3072 gradient that carries 6 400 parameters).
//
// Any TT-matrix with d >= 2 cores is taken as a TWO-core matrix, split at core s (as ttrnn_g2.h does for the recurrence):
//     tail  Gt[i_t][j_t][a] = cores s .. d-1      head  Gh[i_h][j_h][a] = cores 0 .. s-1       (a: the rank R_s)
//     y[i_h, i_t] = sum_{j_h, a} Gh[i_h][j_h][a] * C1[i_t][j_h][a],    C1[i_t][j_h][a] = sum_{j_t} Gt[i_t][j_t][a] * x[j_h][j_t]
// (t3nsor/ops.py:78-93 evaluates exactly this, core by core).  Per row n = (b, t) the adjoint is four small GEMMs
//     A   C1 [p = (a, i_t)][(n, j_h)]   = Gt [p][j_t]    x  [(n, j_h)][j_t]               K = J_t
//     B   dC1[q = (a, j_h)][(n, i_t)]   = Gh^T[q][i_h]   dy [n][i_h][i_t]                 K = I_h
//     C   dGh[i_h][q]                  += dy [n][i_h][i_t] C1 [n][q][i_t]                 K = (n, i_t)   <- summed over the rows
//     D   dGt[p][j_t]                  += dC1[n][p][j_h]   x  [(n, j_h)][j_t]             K = (n, j_h)   <- summed over the rows
// all four on the 16-bit matrix cores with two fp16 pieces per operand (three terms, fp32 accumulation: DESIGN.md 4a), the gate
// gradients dy read from HBM ONCE for both matrices of an LSTM layer and the bias gradients (their column sums) on the way.
// Host + device POD, no HIP types.
#pragma once
#include "ttrnn_core.h"

namespace ttrnn {

constexpr int C2_NW = 8;                 // waves per workgroup (one workgroup per CU: the LDS images of a block of rows)
constexpr int C2_NT = C2_NW * 64;
constexpr int C2_MAX_R = 16;             // rank at the split point
constexpr int C2_LDS_LIMIT = 160 * 1024;

struct C2Mat {
  int ok;
  int d, s;                              // cores, split point
  int Jh, Jt, Ih, It, R;
  int JhP, JtP;                          // J_h, J_t rounded up to 8 (the 8-element fragment chunks never straddle a row)
  int P, Q;                              // P = R*I_t (index p = a*I_t + i_t), Q = R*JhP (q = a*JhP + j_h)
  int PT, QT, IhT, JtT;                  // 16-wide tiles of P, Q, I_h, JtP
  int KA, KB;                            // 32-wide k-blocks of phase A (over JtP) and phase B (over I_h)
  int NA, NB, KC, KD;                    // per block of nb rows: column tiles of A (nb*JhP) and B (nb*I_t), k-blocks of C (nb*I_t), D (nb*JhP)
  int WA;                                // m-tiles of phase A per wave (tile pt = wave + 8*wa: its Gt fragments live in registers)
  int nC, nD, NU, UW;                    // accumulator units: C tiles (IhT x QT) then D tiles (PT x JtT); UW = units per wave (max)
  int XS, CS1, CS2;                      // row strides (halves) of the x image, the C1 image [row][q][i_t], the dC1 image [row][p][j_h]
  int QR, PR;                            // rows per sample of the C1 / dC1 image (QT*16, PT*16)
  int xs_rows;                           // rows of the x image (>= nb*JhP: tile / k-block overreach lands on zero rows)
  int in, out;
  int EX;                                // x quads per thread and block
  long head_elems, tail_elems;           // merged cores, fp32: Gh[Ih][Jh][R], Gt[It][Jt][R]
  // LDS byte offsets of this matrix's own regions
  int l_xs, l_ghf, l_tab;
  // workspace byte offsets (from the call's workspace base)
  long w_gh, w_gt, w_dgh, w_dgt, w_gtf, w_ghf, w_hdr, w_part;
};

struct C2Plan {
  int ok;
  int nmat;                              // matrices sharing one pass over dy (2: the LSTM's input and hidden matrix)
  int nb;                                // rows per block
  int big;                               // kernel variant: 0 = <NACC 4, WA 1, KA 1, EQ 4>, 1 = <NACC 8, WA 2, KA 2, EQ 8>
  int OUT, Ih, It, DS, dy_rows, EQ;      // the dy image [row][i_h][i_t] (shared: both matrices split their output modes alike)
  int l_dy, l_c1, l_dc1, lds;            // LDS byte offsets of the shared regions, total
  int c1_plane, dc1_plane;               // plane sizes (halves) of the shared C1 / dC1 regions
  int grid;
  long w_bpart, w_cmax, ws_bytes;
  C2Mat m[2];
};

inline int c2_ceil(int a, int b) { return (a + b - 1) / b; }

// chain multiply-adds x 2 per row of the four GEMMs, split at sp (no tile padding)
inline double c2_chain_flops(const TtShape& s, int sp) {
  double It = 1, Jt = 1, Ih = 1, Jh = 1;
  for (int k = sp; k < s.d; ++k) { It *= s.I[k]; Jt *= s.J[k]; }
  for (int k = 0; k < sp; ++k) { Ih *= s.I[k]; Jh *= s.J[k]; }
  const double R = s.R[sp];
  return 2.0 * (2.0 * R * It * Jh * Jt + 2.0 * R * Jh * It * Ih);
}
// the split point with the cheapest chain (0: d < 2)
inline int c2_best_split(const TtShape& s) {
  int best = 0;
  double bc = 0;
  for (int sp = 1; sp < s.d; ++sp) {
    const double c = c2_chain_flops(s, sp);
    if (best == 0 || c < bc) { best = sp; bc = c; }
  }
  return best;
}

// shape-only part of a matrix plan; false = this kernel does not take the matrix
inline bool c2_plan_mat(C2Mat* m, const TtShape& s, int sp, int nb, int big) {
  *m = C2Mat{};
  if (s.d < 2 || sp < 1 || sp >= s.d) return false;
  m->d = s.d; m->s = sp;
  m->It = m->Jt = m->Ih = m->Jh = 1;
  for (int k = sp; k < s.d; ++k) { m->It *= s.I[k]; m->Jt *= s.J[k]; }
  for (int k = 0; k < sp; ++k) { m->Ih *= s.I[k]; m->Jh *= s.J[k]; }
  m->R = s.R[sp];
  m->in = s.in_size; m->out = s.out_size;
  if (m->R > C2_MAX_R) return false;
  for (int k = 0; k <= s.d; ++k) if (s.R[k] > 64) return false;
  if (m->It % 16 != 0 || m->Jt % 4 != 0) return false;           // column tiles of B inside one row; x quads inside one j_h
  if ((nb * m->It) % 32 != 0) return false;
  m->JhP = (m->Jh + 7) & ~7; m->JtP = (m->Jt + 7) & ~7;
  m->P = m->R * m->It; m->Q = m->R * m->JhP;
  m->PT = c2_ceil(m->P, 16); m->QT = c2_ceil(m->Q, 16); m->IhT = c2_ceil(m->Ih, 16); m->JtT = c2_ceil(m->JtP, 16);
  m->KA = c2_ceil(m->JtP, 32); m->KB = c2_ceil(m->Ih, 32);
  m->NA = c2_ceil(nb * m->JhP, 16); m->NB = nb * m->It / 16; m->KC = nb * m->It / 32; m->KD = c2_ceil(nb * m->JhP, 32);
  m->WA = c2_ceil(m->PT, C2_NW);
  m->nC = m->IhT * m->QT; m->nD = m->PT * m->JtT; m->NU = m->nC + m->nD; m->UW = c2_ceil(m->NU, C2_NW);
  if (m->WA > (big ? 2 : 1) || m->KA > (big ? 2 : 1) || m->UW > (big ? 8 : 4)) return false;
  if (m->KB > 8 || m->KC > 16 || m->KD > 16 || m->NA > 64 || m->QT > 64) return false;
  m->XS = 32 * m->KA + 8; m->CS1 = m->It + 8; m->CS2 = m->JhP + 8;
  m->QR = m->QT * 16; m->PR = m->PT * 16;
  m->xs_rows = m->NA * 16 > m->KD * 32 ? m->NA * 16 : m->KD * 32;
  m->EX = c2_ceil(nb * m->in / 4, C2_NT);
  if (m->EX > 2 || m->in % 4 != 0) return false;
  m->head_elems = (long)m->Ih * m->Jh * m->R;
  m->tail_elems = (long)m->It * m->Jt * m->R;
  m->ok = 1;
  return true;
}

}  // namespace ttrnn
