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
  int WA, NAG;                           // phase A: wave w holds the Gt fragments of m-tiles pt = w % PT (+ PT*wa ... WA of them when PT > #waves)
                                         // and walks the column tiles na = w / PT, + NAG, ... (NAG = #waves / PT groups of waves share an m-tile)
  int nC, nD, NU, UW;                    // accumulator units: C tiles (IhT x QT) then D tiles (PT x JtT); UW = slots per wave
  int XS, CS1, CS2;                      // row strides (halves) of the x image, the C1 image [row][q][i_t], the dC1 image [row][p][j_h]
  int QR, PR;                            // rows per sample of the C1 / dC1 image (QT*16, PT*16)
  int xs_rows;                           // rows of the x image (>= nb*JhP: tile / k-block overreach lands on zero rows)
  int in, out;
  int EX;                                // x quads (xelem: x elements) per thread and block
  int xelem;                             // J_t not a multiple of 4 (the four-core input matrix, J_t = 10): x is staged element by element
  long head_elems, tail_elems;           // merged cores, fp32: Gh[Ih][Jh][R], Gt[It][Jt][R]
  // LDS byte offsets of this matrix's own regions
  int l_xs, l_ghf, l_tab, l_c1, l_dc1;   // (the C1 / dC1 images are per matrix: both matrices run their phases between the same two barriers)
  int c1_plane, dc1_plane;               // plane sizes (halves) of the C1 / dC1 images
  // workspace byte offsets (from the call's workspace base)
  long w_gh, w_gt, w_dgh, w_dgt, w_gtf, w_ghf, w_hdr, w_part;
};

struct C2Plan {
  int ok;
  int nmat;                              // matrices sharing one pass over dy (2: the LSTM's input and hidden matrix)
  int nb;                                // rows per block
  int big;                               // kernel variant: 0 = <NACC 4, WA 1, KA 1, EQ 4, XQ 1>, 1 = <NACC 8, WA 2, KA 2, EQ 8, XQ 2>
  int OUT, Ih, It, DS, dy_rows, EQ;      // the dy image [row][i_h][i_t] (shared: both matrices split their output modes alike)
  int l_dy, lds;                         // LDS byte offset of the shared dy image, total
  int grid;
  long w_bpart, w_cmax, ws_bytes;
  C2Mat m[2];
};

constexpr int c2_ceil(int a, int b) { return (a + b - 1) / b; }

// chain multiply-adds x 2 per row of the four GEMMs, split at sp (no tile padding)
inline double c2_chain_flops(const TtShape& s, int sp) {
  double It = 1, Jt = 1, Ih = 1, Jh = 1;
  for (int k = sp; k < s.d; ++k) { It *= s.I[k]; Jt *= s.J[k]; }
  for (int k = 0; k < sp; ++k) { Ih *= s.I[k]; Jh *= s.J[k]; }
  const double R = s.R[sp];
  return 2.0 * (2.0 * R * It * Jh * Jt + 2.0 * R * Jh * It * Ih);
}

// The two-core view (d cores split at s: J_h x J_t -> I_h x I_t through rank R) of a matrix plan, everything derived from it.
// constexpr: the shape-specialised instantiations of the kernel (the reference's speaker encoder) evaluate the SAME function at
// compile time, so their tile counts, strides and LDS offsets are literals; 0 = this kernel does not take the matrix.
constexpr bool c2_mat_from(C2Mat* m, int d, int sp, int Jh, int Jt, int Ih, int It, int R, int in, int out, int nb, int big) {
  *m = C2Mat{};
  m->d = d; m->s = sp;
  m->Jh = Jh; m->Jt = Jt; m->Ih = Ih; m->It = It; m->R = R;
  m->in = in; m->out = out;
  if (R > C2_MAX_R) return false;
  if (It % 16 != 0) return false;                                // column tiles of B inside one row
  m->xelem = Jt % 4 != 0 ? 1 : 0;                                // x quads must lie inside one j_h row: else element by element
  if ((nb * It) % 32 != 0) return false;
  m->JhP = (Jh + 7) & ~7; m->JtP = (Jt + 7) & ~7;
  m->P = R * It; m->Q = R * m->JhP;
  m->PT = c2_ceil(m->P, 16); m->QT = c2_ceil(m->Q, 16); m->IhT = c2_ceil(Ih, 16); m->JtT = c2_ceil(m->JtP, 16);
  m->KA = c2_ceil(m->JtP, 32); m->KB = c2_ceil(Ih, 32);
  m->NA = c2_ceil(nb * m->JhP, 16); m->NB = nb * It / 16; m->KC = nb * It / 32; m->KD = c2_ceil(nb * m->JhP, 32);
  m->WA = c2_ceil(m->PT, C2_NW);
  m->NAG = m->PT < C2_NW ? C2_NW / m->PT : 1;
  m->nC = m->IhT * m->QT; m->nD = m->PT * m->JtT; m->NU = m->nC + m->nD;
  m->UW = c2_ceil(m->nC, C2_NW) + c2_ceil(m->nD, C2_NW);      // accumulator slots of a wave: C tiles, then D tiles
  if (m->WA > (big ? 2 : 1) || m->KA > (big ? 2 : 1) || m->UW > (big ? 8 : 4)) return false;
  if (m->KB > 8 || m->KC > 16 || m->KD > 16 || m->NA > 64 || m->QT > 64) return false;
  // row strides (halves): a multiple of 8 (16-byte fragment reads); 16 lanes of a ds_read_b128 group at stride S/2 dwords hit
  // 64 different banks when S/2 is 4 (mod 8) — JhP = 8, 24, 40 as they are, the others 8 more
  m->XS = 32 * m->KA + 8; m->CS1 = It + 8; m->CS2 = (m->JhP % 16 == 8) ? m->JhP : m->JhP + 8;
  m->QR = m->QT * 16; m->PR = m->PT * 16;
  m->xs_rows = m->NA * 16 > m->KD * 32 ? m->NA * 16 : m->KD * 32;
  m->EX = m->xelem ? c2_ceil(nb * in, C2_NT) : c2_ceil(nb * in / 4, C2_NT);
  if (m->EX > (big ? 2 : 1) || (!m->xelem && in % 4 != 0)) return false;
  m->head_elems = (long)Ih * Jh * R;
  m->tail_elems = (long)It * Jt * R;
  m->ok = 1;
  return true;
}

// shape-only part of a matrix plan; false = this kernel does not take the matrix
inline bool c2_plan_mat(C2Mat* m, const TtShape& s, int sp, int nb, int big) {
  *m = C2Mat{};
  if (s.d < 2 || sp < 1 || sp >= s.d) return false;
  int It = 1, Jt = 1, Ih = 1, Jh = 1;
  for (int k = sp; k < s.d; ++k) { It *= s.I[k]; Jt *= s.J[k]; }
  for (int k = 0; k < sp; ++k) { Ih *= s.I[k]; Jh *= s.J[k]; }
  for (int k = 0; k <= s.d; ++k) if (s.R[k] > 64) return false;
  return c2_mat_from(m, s.d, sp, Jh, Jt, Ih, It, s.R[sp], s.in_size, s.out_size, nb, big);
}

// the call-level part: the shared dy image, the LDS carve-up (the matrices' plans are filled in); false = does not fit
constexpr bool c2_layout(C2Plan* p) {
  const int nb = p->nb, nmat = p->nmat, big = p->big;
  if (nmat == 2 && (p->m[0].Ih != p->m[1].Ih || p->m[0].It != p->m[1].It || p->m[0].out != p->m[1].out)) return false;
  p->OUT = p->m[0].out; p->Ih = p->m[0].Ih; p->It = p->m[0].It;
  if (p->OUT % 4 != 0) return false;
  p->DS = p->It + 8;
  p->dy_rows = nb * p->Ih + 32;
  p->EQ = c2_ceil(nb * p->OUT / 4, C2_NT);
  if (p->EQ > (big ? 8 : 4)) return false;
  long off = 0;
  p->l_dy = (int)off;
  long dyb = (long)2 * p->dy_rows * p->DS * 2;
  if (dyb < (long)nb * p->OUT * 4) dyb = (long)nb * p->OUT * 4;      // (the bias partials pass through this region at the end)
  off += (dyb + 15) & ~15L;
  for (int i = 0; i < nmat; ++i) {
    C2Mat& m = p->m[i];
    m.c1_plane = nb * m.QR * m.CS1; m.dc1_plane = nb * m.PR * m.CS2;
    m.l_c1 = (int)off; off += ((long)2 * m.c1_plane * 2 + 15) & ~15L;
    m.l_dc1 = (int)off; off += ((long)2 * m.dc1_plane * 2 + 15) & ~15L;
    m.l_xs = (int)off; off += ((long)2 * m.xs_rows * m.XS * 2 + 15) & ~15L;
    m.l_ghf = (int)off; off += (long)m.QT * m.KB * 2 * 1024;
    m.l_tab = (int)off;
    off += ((long)(m.NA * 16 + m.QT * 4 + 2 * m.NB) * 4 + 15) & ~15L;
  }
  if (off > (long)C2_LDS_LIMIT) return false;
  p->lds = (int)off;
  return true;
}

// ---- compile-time plans of the shape-specialised instantiations --------------------------------------------------------------
// SPEC 1: the reference's speaker encoder (experiments/speaker_verification/encoder/params_model.py:2-4,14-16 — 40 mel channels,
//         H = 768, n_cores = 2, rank = 2; tt_shape: (5, 8) x (48, 64) and (24, 32) x (48, 64)), both matrices of the LSTM layer;
// SPEC 2: its hidden matrix alone (a TT-GRU's gate gradients differ between the matrices; a layer whose input needs dx).
// SPEC 3 / 4: the same with rank 4 (the reference's result tables go up to rank 8) — plans of the kernel's large variant.
// SPEC 5 ... 8: the FOUR-core models of those tables at this size (tt_shape (4, 4, 6, 8) x (6, 8, 8, 8), input (2, 2, 2, 5)), split two
//         by two: (16, 48) x (48, 64), input (4, 10) x (48, 64); 5 / 6 rank 2 pair / hidden, 7 / 8 rank 4.
template <int SPEC>
constexpr C2Plan c2_const_plan() {
  C2Plan p{};
  if (SPEC >= 5) {
    constexpr int R4 = SPEC >= 7 ? 4 : 2;
    p.nb = SPEC == 7 ? 1 : 2; p.big = 1;               // (the planner's own choice: the largest row block whose images fit)
    if (SPEC == 5 || SPEC == 7) {
      p.nmat = 2;
      c2_mat_from(&p.m[0], 4, 2, 4, 10, 48, 64, R4, 40, 3072, p.nb, p.big);
      c2_mat_from(&p.m[1], 4, 2, 16, 48, 48, 64, R4, 768, 3072, p.nb, p.big);
    } else {
      p.nmat = 1;
      c2_mat_from(&p.m[0], 4, 2, 16, 48, 48, 64, R4, 768, 3072, p.nb, p.big);
    }
    p.ok = c2_layout(&p) ? 1 : 0;
    return p;
  }
  constexpr int R = SPEC >= 3 ? 4 : 2;
  p.nb = SPEC >= 3 ? 1 : 2; p.big = SPEC >= 3 ? 1 : 0;      // (rank 4: the images of two rows do not fit the LDS)
  if (SPEC == 1 || SPEC == 3) {
    p.nmat = 2;
    c2_mat_from(&p.m[0], 2, 1, 5, 8, 48, 64, R, 40, 3072, p.nb, p.big);
    c2_mat_from(&p.m[1], 2, 1, 24, 32, 48, 64, R, 768, 3072, p.nb, p.big);
  } else {
    p.nmat = 1;
    c2_mat_from(&p.m[0], 2, 1, 24, 32, 48, 64, R, 768, 3072, p.nb, p.big);
  }
  p.ok = c2_layout(&p) ? 1 : 0;
  return p;
}

// do the fields the KERNEL reads agree? (the workspace offsets and the grid are the launcher's)
inline bool c2_same_kernel_plan(const C2Plan& a, const C2Plan& b) {
  if (a.nmat != b.nmat || a.nb != b.nb || a.big != b.big || a.OUT != b.OUT || a.Ih != b.Ih || a.It != b.It || a.DS != b.DS ||
      a.dy_rows != b.dy_rows || a.l_dy != b.l_dy || a.lds != b.lds)
    return false;
  for (int i = 0; i < a.nmat; ++i) {
    const C2Mat &x = a.m[i], &y = b.m[i];
    if (x.d != y.d || x.s != y.s || x.Jh != y.Jh || x.Jt != y.Jt || x.Ih != y.Ih || x.It != y.It || x.R != y.R || x.in != y.in || x.xelem != y.xelem ||
        x.out != y.out || x.l_xs != y.l_xs || x.l_ghf != y.l_ghf || x.l_tab != y.l_tab || x.l_c1 != y.l_c1 || x.l_dc1 != y.l_dc1)
      return false;      // (everything else of a C2Mat follows from these through c2_mat_from)
  }
  return true;
}

}  // namespace ttrnn
