// ttrnn_g2.h — plan of the runtime-shape two-stage MFMA kernels (ttrnn_g2.hip): ANY TT-matrix with d >= 2 cores is
// evaluated as a two-core matrix, its cores contracted ONCE per launch into
//     tail  Gt[(i_t, a)][j_t]     = cores s .. d-1   (I_t = prod I_k, J_t = prod J_k over k >= s;  a = rank index R_s)
//     head  Gh[i_h][(j_h, a)]     = cores 0 .. s-1   (I_h, J_h over k < s)
// so that the per-timestep chain of t3nsor/ops.py:78-93 becomes two GEMM stages whatever d, the modes and the ranks are:
//     stage 1  C1[i_t][(j_h, a)] = sum_{j_t}     Gt[(i_t, a)][j_t] * h[j_h][j_t]        M = I_t*R, N = J_h, K = J_t   (small K)
//     stage 2  y[i_h*I_t + i_t]  = sum_{(j_h,a)} Gh[i_h][(j_h, a)] * C1[i_t][(j_h, a)]   M = I_h,   N = I_t, K = J_h*R  (the FLOPs)
// and the reverse-time chain (BPTT) the transposed pair
//     T2  dC1[i_t][(j_h, a)] = sum_{i_h}     Gh[i_h][(j_h, a)] * dy[i_h*I_t + i_t]      M = J_h*R, N = I_t, K = I_h
//     T1  dh[j_h][j_t]       = sum_{(i_t,a)} Gt[(i_t, a)][j_t] * dC1[i_t][(j_h, a)]      M = J_t,   N = J_h, K = I_t*R.
// All shape-dependent index arithmetic lives in the prep kernels (which write the merged cores as MFMA fragments in the
// order the waves consume them) and in per-thread constants computed before the time loop; the time loops see padded tile
// counts only.  Host + device POD, no HIP types.
#pragma once
#include "ttrnn_core.h"

namespace ttrnn {

constexpr int G2_NW_MAX = 8;        // waves per workgroup: 8 when every sample can own a CU (B <= #CUs: two waves per SIMD hide
constexpr int G2_NT_MAX = G2_NW_MAX * 64;   // each other's LDS / MFMA latency on the per-step critical path), else 4 (several samples share a CU)
constexpr int G2_PF = 8;            // head fragments in flight per wave (k-blocks of 32): ~770 matrix-pipe cycles of cover
constexpr int G2_HUN_LDS = 4096;    // reverse kernel: head^T row scales kept in LDS up to this many rows (G2_BSL units x 8 waves x 16)
constexpr int G2_BSL = 12;          // reverse kernel: register slots of RESIDENT head^T fragments (two fp16 pieces: 96 VGPRs)
constexpr int G2_UPT = 4;           // hidden units per thread in the gate phase: H <= 1024
constexpr int G2_MAX_R = 64;        // rank at the split point
constexpr int G2_LDS_LIMIT = 160 * 1024;
constexpr int G2_LDS_STATIC = 256;  // static __shared__ of the kernels (per-wave maxima), counted in the occupancy decisions

struct G2Mat {
  int ok;
  int nw;                           // waves per workgroup the plan (tile -> wave assignment, fragment streams) is made for
  int d, s;                         // cores, split point
  int It, Jt, Ih, Jh, R, Rp;        // Rp = R rounded up to 4 (four accumulator registers = four consecutive a)
  // block-diagonal heads (ttrnn_rnn_desc::hid_blocks = ng > 1): rank index a = (g, a'), a' < Rb; the head couples output
  // rows of gate g (IhG = Ih / ng rows each) only to ranks of block g.  The contraction index of stage 2 / row index of T2 is
  // then gate-major, k2 = g*Kg + j_h*Rb + a' (Kg = J_h*Rb, a multiple of 32), and a tile only visits its own gate's
  // k-blocks: NKBt = Kg / 32 of the NKB = ng * NKBt (forward), bNKBt = IhG / 32 of bNKB (reverse).  ng = 1: Kg = J_h*Rp.
  int ng, Rb, Kg, IhG, NKBt, bNKBt;
  int in, out;
  // forward stage 1 (fp32 MFMA 16x16x4): tiles (m1 tile, n1 tile), k steps of 4
  int M1T, N1T, KS1, T1;
  // forward stage 2 (bf16 MFMA 16x16x32, three-way split): tiles (m2, n2), k blocks of 32; a tile's k range is split
  // over KSPLIT waves when there are fewer tiles than waves; unit u = tile*KSPLIT + part, wave w takes u = w, w+NW, ..
  int M2T, N2T, NKB, T2, KSPLIT, KPER, KBP, U, UW;  // KPER = k-blocks per part, KBP = KPER padded to G2_PF, UW = units per wave (max)
  int JtS;                          // (unused since the forward stage 1 left the fp32 MFMA)
  // forward stage 1 on two-piece fp16 operands: KB1 = 32-wide k-blocks over j_t; pack8 (J_t <= 8): the four piece products
  // packed along the 32-wide k of ONE MFMA, k-groups [w0|w1|w0|w1] x [x0|x0|x1|x1]; JS = row stride (elements) of the two fp16
  // planes of the h image [2][16*N1T][JS]
  int KB1, pack8, JS;
  int K2S;                          // bf16 stage-2 operand planes [16*N2T][K2S]
  int cin;                          // forward stage 2 with the COLUMN TILES INSIDE a unit (N2T > 1: one fragment, N2T accumulator pairs; resident plans only)
  int wrap;                         // forward head stream: blocks of cyclic copy behind every wave's live blocks (k_g2_fwd_p: G2_PF; else 0)
  // reverse T2 (f16 MFMA, two pieces per operand): M = Jh*Rp, N = It, K = Ih
  int bM2T, bNKB, bT2, bKBP, bU, bUW, bSW;         // no k split: T1 reads the complete dC1; bSW = blocks of a wave's (compact) stream
  int IhS;                          // fp16 dy planes [16*N2T][IhS]
  // reverse T1 (f16 MFMA, two pieces per operand): M = Jt (one or more m tiles), N = Jh, K = It*Rp in blocks of 32, split over
  // bK1SPLIT parts of bKB1P blocks
  int bM1T, bKB1, bT1, bK1SPLIT, bKB1P, bU1;
  int K1S;                          // row stride (halves) of the two fp16 planes of the dC1 image [2][J_h rows][K1S]
  int bNP;                          // reverse-time kernel: passes over the i_t range of dC1 (1; 2 = the image holds HALF of the columns at a time)
  // element counts of the per-launch buffers (workspace)
  long head_elems, tail_elems;      // merged cores, fp32: Gh[Ih][Jh][R], Gt[It][Jt][R]
  long fs2_bytes, ft1_bytes;        // forward: head fragment stream (fp16 x 2 planes, scaled), tail fragments (fp32, scaled)
  long bs2_bytes, bt1_bytes;        // reverse: the same for T2 / T1
};

inline int g2_ceil(int a, int b) { return (a + b - 1) / b; }

// how the k range of `tiles` tiles with `nkb` blocks each is spread over the waves
inline void g2_split(int nw, int tiles, int nkb, int* ksplit, int* kper, int* kbp, int* units, int* uw) {
  int ks = 1;
  if (tiles < nw) {
    ks = nw / tiles;
    if (ks > nkb) ks = nkb;
    if (ks < 1) ks = 1;
  }
  int per = g2_ceil(nkb, ks);
  ks = g2_ceil(nkb, per);                       // no empty part
  *ksplit = ks;
  *kper = per;
  *kbp = g2_ceil(per, G2_PF) * G2_PF;
  *units = tiles * ks;
  *uw = g2_ceil(*units, nw);
}

inline void g2_plan_mat(G2Mat* m, const TtShape& s, int nw, int blocks = 1, bool cin_ok = false) {
  *m = G2Mat{};
  m->nw = nw;
  if (s.d < 2) return;
  long best = -1;
  int bs = 1;
  for (int sp = 1; sp < s.d; ++sp) {
    long It = 1, Jh = 1;
    for (int k = sp; k < s.d; ++k) It *= s.I[k];
    for (int k = 0; k < sp; ++k) Jh *= s.J[k];
    const long rp = (s.R[sp] + 3) & ~3;
    // MFMA work with the tile padding: stage 1 priced as on the fp32 MFMA (what the reverse kernel's T1 still runs on; the
    // forward's fp16 stage 1 is cheaper, the split point is shared) + stage 2
    const long c1 = (long)g2_ceil((int)(It * rp), 16) * g2_ceil((int)Jh, 16) * g2_ceil(s.in_size / (int)Jh, 4) * 32;
    const long c2 = (long)g2_ceil(s.out_size / (int)It, 16) * g2_ceil((int)It, 16) * g2_ceil((int)(Jh * rp), 32) * 96;
    long cost = c1 + c2;
    // a block-diagonal head (checked below) only visits 1/blocks of the stage-2 k range
    if (blocks > 1 && sp >= 2 && s.I[0] == blocks && s.J[0] == 1 && s.R[sp] % (4 * blocks) == 0 &&
        (s.out_size / It / blocks) % 32 == 0 && (Jh * (s.R[sp] / blocks)) % 32 == 0) cost = c1 + c2 / blocks;
    if (best < 0 || cost < best) { best = cost; bs = sp; }
  }
  m->d = s.d; m->s = bs;
  m->It = m->Jt = m->Ih = m->Jh = 1;
  for (int k = bs; k < s.d; ++k) { m->It *= s.I[k]; m->Jt *= s.J[k]; }
  for (int k = 0; k < bs; ++k) { m->Ih *= s.I[k]; m->Jh *= s.J[k]; }
  m->R = s.R[bs]; m->Rp = (m->R + 3) & ~3;
  m->in = s.in_size; m->out = s.out_size;
  if (m->R > G2_MAX_R) return;
  for (int k = 0; k <= s.d; ++k) if (s.R[k] > G2_MAX_R) return;
  // block-diagonal head: usable when the selector core lies in the head (s >= 2) and every block boundary falls on a tile /
  // k-block boundary; otherwise the joint matrix is treated as dense (correct, 1/ng of the stage-2 work is non-zero)
  m->ng = 1; m->Rb = m->Rp; m->Kg = m->Jh * m->Rp; m->IhG = m->Ih;
  if (blocks > 1 && bs >= 2 && s.J[0] == 1 && s.I[0] == blocks && m->R % blocks == 0) {
    const int rb = m->R / blocks, ihg = m->Ih / blocks;
    if (rb % 4 == 0 && m->Ih % blocks == 0 && ihg % 32 == 0 && (m->Jh * rb) % 32 == 0) {
      m->ng = blocks; m->Rb = rb; m->Kg = m->Jh * rb; m->IhG = ihg;
    }
  }
  m->M1T = g2_ceil(m->It * m->Rp, 16); m->N1T = g2_ceil(m->Jh, 16); m->KS1 = g2_ceil(m->Jt, 4);
  m->T1 = m->M1T * m->N1T;
  m->M2T = g2_ceil(m->Ih, 16); m->N2T = g2_ceil(m->It, 16);
  m->NKBt = m->ng > 1 ? m->Kg / 32 : g2_ceil(m->Jh * m->Rp, 32);
  m->NKB = m->ng * m->NKBt;
  m->T2 = m->M2T * m->N2T;
  g2_split(nw, m->T2, m->NKBt, &m->KSPLIT, &m->KPER, &m->KBP, &m->U, &m->UW);
  // Several column tiles (I_t > 16): a unit per (row tile, column tile) streams / holds the SAME head blocks once per column tile
  // (H = 768, d = 2, r = 2 — the speaker encoder's own shape — 192 KB per step for 18 KB of head).  With the column tiles inside
  // the unit a block is one fragment and N2T accumulator pairs; taken where it makes the head RESIDENT in the eight slots.
  m->cin = 0;
  if (cin_ok && m->N2T > 1 && m->N2T <= 4 && m->ng == 1) {
    int ks, kper, kbp, u, uw;
    g2_split(nw, m->M2T, m->NKBt, &ks, &kper, &kbp, &u, &uw);
    if (uw * kbp <= G2_PF && m->UW * m->KBP > G2_PF) {
      m->cin = 1; m->KSPLIT = ks; m->KPER = kper; m->KBP = kbp; m->U = u; m->UW = uw;
    }
  }
  m->JtS = 4 * m->KS1 + 1;
  m->pack8 = m->Jt <= 8;
  m->KB1 = g2_ceil(m->Jt, 32);
  m->JS = 32 * m->KB1 + 16;
  // row stride = 32 (mod 64) bytes-of-slots: ds_read_b128 serves the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
  // (MI355X guide, LDS table); with rows 16 bf16 past a multiple of 32 every group covers all 64 banks once (enumerated)
  m->K2S = 32 * m->NKB + 16;
  m->bM2T = g2_ceil(m->ng > 1 ? m->ng * m->Kg : m->Jh * m->Rp, 16); m->bNKB = g2_ceil(m->Ih, 32);
  m->bNKBt = m->ng > 1 ? m->IhG / 32 : m->bNKB;
  m->bT2 = m->bM2T * m->N2T;
  m->bKBP = g2_ceil(m->bNKBt, G2_PF) * G2_PF;
  m->bU = m->bT2;
  m->bUW = g2_ceil(m->bU, nw);
  m->bSW = g2_ceil(m->bUW * m->bNKBt, G2_PF) * G2_PF;
  m->IhS = 32 * m->bNKB + 16;
  m->bM1T = g2_ceil(m->Jt, 16); m->bKB1 = g2_ceil(m->It * m->Rp, 32);
  m->bT1 = m->bM1T * m->N1T;
  {
    int ks = 1;
    if (m->bT1 < nw) { ks = nw / m->bT1; if (ks > m->bKB1) ks = m->bKB1; if (ks < 1) ks = 1; }
    m->bKB1P = g2_ceil(m->bKB1, ks);
    m->bK1SPLIT = g2_ceil(m->bKB1, m->bKB1P);
    m->bU1 = m->bT1 * m->bK1SPLIT;
  }
  m->K1S = 32 * m->bKB1 + 16;
  m->bNP = 1;
  m->head_elems = (long)m->Ih * m->Jh * m->R;
  m->tail_elems = (long)m->It * m->Jt * m->R;
  m->fs2_bytes = (long)nw * m->UW * m->KBP * 2 * 64 * 16;
  m->ft1_bytes = (long)m->M1T * m->KB1 * (m->pack8 ? 1 : 2) * 64 * 16;     // fp16 fragments (xh8 per lane), 1 or 2 planes
  m->bs2_bytes = (long)nw * m->bSW * 2 * 64 * 16;      // two fp16 pieces per block
  m->bt1_bytes = (long)m->bM1T * m->bKB1 * 2 * 64 * 16;      // two fp16 pieces per (m tile, k-block)
  m->ok = 1;
}

struct G2Plan {
  int ok, okf, okb;                 // both kernels / the forward / the reverse-time kernel fit
  int cell, G, H, B, T;
  G2Mat hid;
  int upt;                          // hidden units per thread: every thread owns units tid + u * 64 nw
  int b_upt;                        // (= upt)
  // LDS carve-up (bytes) of the forward and the reverse-time kernel
  int f_hb, f_img, f_ybuf, f_tab, f_sc, f_t1, f_lds; // f_tab: stage-1 store offsets [T1][4] ints; f_sc: the inverse output scales
                                                     // [I_h | I_t] floats; f_t1: tail fragments (0: from L2)
  int b_dy, b_dc1, b_dh, b_tab, b_t1, b_lds;   // b_tab: dy plane offsets [G*H] + T2 store offsets [bM2T*4] ints
  int abl;                                     // -DTTRNN_ABLATIONS builds: option `dev` (result-destroying switches of the stamps tool); else 0
  int b_hun;                                   // rows of the inverse-row-scale table kept in LDS (0: read from the workspace)
  int b_fast;                                  // reverse kernel: streamed T2 on the group-of-units loop where the geometry allows (set by the launcher)
  int pair;                                    // forward only: k_g2_fwd_p's plan — two samples per workgroup, LDS carve-up for both (g2_plan_pair);
                                               // the value = column tiles of stage 2 the pair fills (1: I_t <= 8, 2: I_t <= 16)
  int b_cmx;                                   // running column maxima of the gate gradients [G*H (+ H: GRU's hidden-side n)] floats, behind
                                               // everything else; 0 = no room (the by-product is then not offered for this shape)
};

inline size_t g2_al(size_t v) { return (v + 255) & ~(size_t)255; }

// wide: 8 waves per workgroup (the caller passes B <= #CUs)
inline void g2_plan(G2Plan* p, const RnnShape& rs, bool wide, bool cin_ok = false) {
  *p = G2Plan{};
  p->cell = rs.cell; p->G = rs.G; p->H = rs.H; p->B = rs.B; p->T = rs.T;
  const int nw = wide ? G2_NW_MAX : 4;
  g2_plan_mat(&p->hid, rs.hid_s, nw, rs.hid_blocks, cin_ok);
  if (!p->hid.ok) return;
  // the gate phases run on every thread: unit tid + u * 64 nw (1, 2 or 4 hidden units per thread)
  if (rs.H > G2_UPT * 256) return;
  p->upt = g2_ceil(rs.H, 64 * nw);
  if (p->upt == 3) p->upt = 4;                 // kernels are instantiated for 1, 2, 4 units per thread
  p->b_upt = p->upt;
  const G2Mat& m = p->hid;
  p->f_hb = (int)g2_al((size_t)2 * 16 * m.N1T * m.JS * 2);       // two fp16 planes of the h image
  // forward: two fp16 planes (ttrnn_split.h, flavour b) of the I_t REAL rows (stage 2 clamps its row index; the naive per-gate
  // sets of H = 512, r = 16, d = 3 have I_t = 8: 132 KB instead of 263 — the difference between this tier and the VALU kernels)
  p->f_img = (int)g2_al((size_t)2 * (m.It < 16 * m.N2T ? m.It : 16 * m.N2T) * m.K2S * 2);
  p->f_ybuf = (int)g2_al((size_t)m.KSPLIT * rs.G * rs.H * 4);
  p->f_tab = (int)g2_al((size_t)m.T1 * 4 * 4);      // stage-1 store offsets: one per (tile, lane quarter)
  p->f_sc = (int)g2_al((size_t)(m.Ih + m.It) * 4);
  p->f_lds = p->f_hb + p->f_img + p->f_ybuf + p->f_tab + p->f_sc;
  p->f_t1 = (m.ft1_bytes <= 32 * 1024 && p->f_lds + (int)g2_al((size_t)m.ft1_bytes) <= G2_LDS_LIMIT) ? (int)g2_al((size_t)m.ft1_bytes) : 0;
  p->f_lds += p->f_t1;
  // (the I_t real rows, as the forward's operand image: T2 clamps its row index)
  p->b_dy = (int)g2_al((size_t)2 * (m.It < 16 * m.N2T ? m.It : 16 * m.N2T) * m.IhS * 2);      // two fp16 planes
  // dC1 rows: the J_h real ones, not the 16 N1T of T1's row tiles (T1 clamps its row index; H = 768, d = 2, r = 16: 24 rows of
  // 1 028 floats = 99 KB instead of 131 — the difference between this kernel and the VALU fallback for that shape)
  p->b_dc1 = (int)g2_al(((size_t)2 * (m.Jh < 16 * m.N1T ? m.Jh : 16 * m.N1T) * m.K1S) * 2);
  p->b_dh = (int)g2_al((size_t)m.bK1SPLIT * rs.H * 4);
  // (+ the inverse row scales of head^T, one float per row of T2, where they are few — every shape whose fragments can be resident;
  // the 4 096 rows of a rank-64 naive set stay in L2 and are fetched a unit ahead)
  p->b_hun = m.bM2T * 16 <= G2_HUN_LDS ? m.bM2T * 16 : 0;
  // (+ the inverse row scales of tail^T: sixteen floats per row tile of T1)
  p->b_tab = (int)g2_al(((size_t)rs.G * rs.H + (size_t)m.bM2T * 4 + (size_t)m.bM1T * 16 + (size_t)p->b_hun) * 4);
  p->b_lds = p->b_dy + p->b_dc1 + p->b_dh + p->b_tab;
  if (p->b_hun > 0 && !(m.N2T <= 4 && m.bNKBt <= 4 && m.bUW * m.bNKBt <= G2_BSL)) {
    // streamed fragments (the resident kernel needs the table): not where the table costs a co-resident workgroup (H = 768, d = 4:
    // 6 KB more took the four-wave workgroups from two per CU to one, 10.3 -> 10.9 ms per training step)
    const int tab0 = (int)g2_al(((size_t)rs.G * rs.H + (size_t)m.bM2T * 4 + (size_t)m.bM1T * 16) * 4);
    const int lds0 = p->b_lds - p->b_tab + tab0;
    if (G2_LDS_LIMIT / (lds0 + G2_LDS_STATIC) > G2_LDS_LIMIT / (p->b_lds + G2_LDS_STATIC)) { p->b_hun = 0; p->b_tab = tab0; p->b_lds = lds0; }
  }
  p->b_t1 = (m.bt1_bytes <= 32 * 1024 && p->b_lds + (int)g2_al((size_t)m.bt1_bytes) <= G2_LDS_LIMIT) ? (int)g2_al((size_t)m.bt1_bytes) : 0;
  p->b_lds += p->b_t1;
  // the forward and the reverse-time kernel have different LDS footprints (cfg5's shape: 155 KB / 183 KB): each route is
  // offered on its own, the reserve format is the same for every route
  p->okf = p->f_lds <= G2_LDS_LIMIT;
  p->okb = p->b_lds <= G2_LDS_LIMIT;
  if (!p->okb && p->hid.It % 32 == 0 && p->hid.N2T % 2 == 0) {
    // the dC1 image does not fit (H = 1024, d = 2, r = 16: 131 KB of 187): T2 and T1 run in TWO passes over halves of its i_t
    // range — T2 computes the column tiles of one half (the head stream rolls on through the others), T1 multiplies that half
    // of its k range and adds to the partial dh.  Twice the head stream, the same products; before: BPTT on the VALU kernels
    G2Mat& mm = p->hid;
    mm.bNP = 2;
    mm.K1S = 32 * (mm.bKB1 / 2) + 16;               // (I_t % 32 == 0: each half is a whole number of k-blocks)
    p->b_dc1 = (int)g2_al(((size_t)2 * (mm.Jh < 16 * mm.N1T ? mm.Jh : 16 * mm.N1T) * mm.K1S) * 2);
    p->b_lds = p->b_dy + p->b_dc1 + p->b_dh + p->b_tab;
    p->b_t1 = (mm.bt1_bytes <= 32 * 1024 && p->b_lds + (int)g2_al((size_t)mm.bt1_bytes) <= G2_LDS_LIMIT) ? (int)g2_al((size_t)mm.bt1_bytes) : 0;
    p->b_lds += p->b_t1;
    p->okb = p->b_lds <= G2_LDS_LIMIT;
    if (!p->okb) { mm.bNP = 1; mm.K1S = 32 * mm.bKB1 + 16; }
  }
  // by-product of the reverse-time kernel (TTRNN_BWD_STATS_COLMAX): only where it costs no route
  p->b_cmx = (int)g2_al((size_t)(rs.G + (rs.cell == TTRNN_GRU ? 1 : 0)) * rs.H * 4);
  if (!p->okb || p->b_lds + p->b_cmx > G2_LDS_LIMIT) p->b_cmx = 0;
  if (p->b_cmx > 0) {
    // ... and no occupancy: four-wave workgroups (B > #CUs) run two per CU where their LDS allows it — 12 KB more took
    // H = 768, d = 4 from two to one and its training step from 9.9 to 11.9 ms
    // (+ the kernel's static shared words: 2 x 81 920 dynamic bytes fill the CU exactly, and the 32 bytes of the waves' maxima
    // made it one workgroup per CU — H = 768, d = 4 again)
    const int w0 = G2_LDS_LIMIT / (p->b_lds + G2_LDS_STATIC), w1 = G2_LDS_LIMIT / (p->b_lds + p->b_cmx + G2_LDS_STATIC);
    if (w1 < (w0 < 2 ? w0 : 2)) p->b_cmx = 0;
  }
  p->ok = p->okf && p->okb;
}

// The forward plan of k_g2_fwd_p (ttrnn_g2.hip): the eight-wave plan with the LDS images of TWO samples — the h image in its
// eight live k-slots per row, the stage-2 operand image with rows (sample, i_t), two output vectors.  For streamed heads only
// (the caller checks that), I_t <= 16 (the pair in one or two column tiles of stage 2: each streamed block feeds both), J_t <= 8
// (term-packed stage 1), H <= 1024.
constexpr int G2_PAIR_NS = 2;
constexpr int G2_PAIR_JS = 8;
constexpr int G2_PAIR_MAXF = 4;      // m tiles of stage 1 a wave's share may touch (their tail fragments stay in registers)
inline void g2_plan_pair(G2Plan* p, const RnnShape& rs) {
  g2_plan(p, rs, true);
  p->okf = 0; p->ok = 0; p->okb = 0;
  if (!p->hid.ok) return;
  const G2Mat& m = p->hid;
  if (m.N2T != 1 || !m.pack8 || rs.H > 2 * G2_NT_MAX) return;      // (N2T == 1: I_t <= 16, the pair in one or two column tiles)
  {
    const int nc = G2_PAIR_NS * m.N1T, tpw = g2_ceil(nc * m.M1T, G2_NW_MAX);
    if ((tpw + nc - 2) / nc + 1 > G2_PAIR_MAXF) return;      // m tiles a contiguous run of tpw (m tile, column) pairs can touch
  }
  p->pair = g2_ceil(G2_PAIR_NS * m.It, 16);      // column tiles of stage 2 (1 or 2)
  p->hid.wrap = G2_PF;
  p->hid.fs2_bytes = (long)G2_NW_MAX * (m.UW * m.KBP + G2_PF) * 2 * 64 * 16;
  p->upt = g2_ceil(rs.H, G2_NT_MAX);
  p->b_upt = p->upt;
  p->f_hb = (int)g2_al((size_t)2 * G2_PAIR_NS * 16 * m.N1T * G2_PAIR_JS * 2);
  p->f_img = (int)g2_al((size_t)2 * G2_PAIR_NS * m.It * m.K2S * 2);
  p->f_ybuf = (int)g2_al((size_t)G2_PAIR_NS * m.KSPLIT * rs.G * rs.H * 4);
  p->f_tab = (int)g2_al((size_t)m.M1T * 4 * 4);
  p->f_sc = (int)g2_al((size_t)(m.Ih + m.It) * 4);
  p->f_lds = p->f_hb + p->f_img + p->f_ybuf + p->f_tab + p->f_sc;
  p->f_t1 = 0;                                 // (the tail fragments a wave needs stay in its registers)
  p->okf = p->f_lds + G2_LDS_STATIC <= G2_LDS_LIMIT;
}

// workspace of the recurrent forward / reverse kernels: merged cores (fp32) + fragment streams
// forward: + the int32 exponents of the diagonal scales (k_g2_diag_a / _b): [I_t | 64 | I_h], then [I_t][64] partial maxima
inline long g2_merge_blocks(const G2Mat& m) { return (((long)m.Ih * m.Jh + (long)m.It * m.Jt) * m.R + 255) / 256; }
inline size_t g2_diag_ints(const G2Mat& m) { return (size_t)m.It + 64 + (size_t)m.Ih; }
inline size_t g2_fwd_ws_bytes(const G2Mat& m) {
  return g2_al((size_t)m.head_elems * 4) + g2_al((size_t)m.tail_elems * 4) + g2_al((size_t)m.fs2_bytes) + g2_al((size_t)m.ft1_bytes) +
         g2_al(g2_diag_ints(m) * sizeof(int) + (size_t)m.It * 64 * sizeof(float));
}
inline size_t g2_bwd_ws_bytes(const G2Mat& m) {
  // (+ per row of head^T: inverse scale and L1 norm; per row of tail^T: inverse scale; + 4 KB: the diagnostic stamps of
  // -DTTRNN_ABLATIONS builds)
  return g2_al((size_t)m.head_elems * 4) + g2_al((size_t)m.tail_elems * 4) + g2_al((size_t)m.bs2_bytes) + g2_al((size_t)m.bt1_bytes) +
         g2_al(((size_t)m.bM2T * 32 + (size_t)m.bM1T * 16) * 4) + 4096;
}

}  // namespace ttrnn
