// ttrnn_c2r_dev.h — the chain weight-gradient kernel with REGISTER hand-offs between its phases (included by ttrnn_fast_c2w.hip,
// inside its anonymous namespace, after the helpers it shares with k_c2w: merged cores, header, fragments, pull-back, bias sums).
//
// k_c2w keeps the row-local results C1 and dC1 of phases A and B (ttrnn_c2w.h) in LDS images between two workgroup barriers; the
// stamps of round 6 (profiles/r6/stamps_c2w.txt) show what that costs at the reference's speaker-encoder size
// (experiments/speaker_verification/encoder/params_model.py:2-4,14-16): 8 400 cycles per block of two rows for 2 300 cycles of
// matrix-core work — three barriers, four LDS round trips, and 330 KB of LDS traffic per row against 128 bytes per clock.
//
// Here the accumulator tile of a row-local phase IS the next phase's MFMA operand, in the lane that computed it:
//   chain 1  A -> C   item (a, jh, kc):  tiles C1[(a, i_t = 16 (2 kc + h) + 4 g + j)][j_h = 16 jh + c], h = 0, 1 — eight values per lane
//            that are, split into two fp16 pieces, the B operand B[k][n = j_h] of phase C under the k order
//            k = 8 g + j -> i_t = 32 kc + 4 g + j (j < 4), 32 kc + 16 + 4 g + j - 4 (j >= 4); dy's A operand is read in that order
//            (two 8-byte reads); accumulators dGh[i_h tile][(j_h, a)] partial over kc;
//   chain 2  B -> D   item (a, nt, kd):  tiles dC1[j_h = 16 (2 kd + h) + 4 g + j][i_t = 16 nt + c] (A = Gh^T rows j_h of rank a) — the A
//            operand A[row = i_t][k] of phase D under k = 8 g + j -> j_h = 32 kd + 4 g + j (j < 4), 32 kd + 16 + 4 g + j - 4; x's B
//            operand comes through the transposing LDS read with its two row groups 16 apart; accumulators dGt[(a, i_t tile)][j_t].
// No C1 / dC1 image, no barrier between phases: the only LDS images are the staged rows (dy, x: two fp16 pieces each, double
// buffered: ONE barrier per block of rows), and every wave keeps the weight fragments of its items in registers for the whole
// launch.  Items that differ only in the rank index a share their operand reads: a wave takes all (or AC) ranks of a group.
// Shapes: compile-time (the reference's published encoder and the variants of its result tables); everything else stays on k_c2w.

constexpr int c2r_max(int a, int b) { return a > b ? a : b; }
// ranks a wave takes of a group: all of them, unless the groups x ranks then fill fewer than four waves
constexpr int c2r_ac(int G, int R) { return G * R <= 4 ? 1 : (G * R / 2 <= 4 && R % 2 == 0 ? 2 : R); }

template <int JH, int JT, int IH, int IT, int R_, int IN_>
struct C2RM {
  static constexpr int Jh = JH, Jt = JT, Ih = IH, It = IT, R = R_, in = IN_;
  static constexpr int JHT = (JH + 15) / 16, JHK = (JHT + 1) / 2, JTK = (JT + 31) / 32, JTT = (JT + 15) / 16;
  static constexpr int ITT = IT / 16, KC = (ITT + 1) / 2, IHT = (IH + 15) / 16, IHK = (IH + 31) / 32;
  static constexpr int NJH = JHT >= 2 ? 2 : 1;                 // j_h tiles of a chain-2 item
  static constexpr int G1 = JHT * KC, G2 = ITT * JHK;          // groups: (jh, kc), (nt, kd)
  static constexpr int AC1 = c2r_ac(G1, R_), AC2 = c2r_ac(G2, R_);
  static constexpr int U1 = G1 * (R_ / AC1), U2 = G2 * (R_ / AC2);      // units (<= 4: a half of the workgroup's waves)
  static constexpr int XW = 32 * JTK > 16 * JTT ? 32 * JTK : 16 * JTT;
  static constexpr int XS = XW + 8;                            // (XS / 2 = 4 mod 8 dwords: 16-byte row reads of 16 rows hit 64 banks)
  static constexpr int XR = 32 * JHK;                          // image rows of one staged row (zero past J_h)
  static constexpr int xelem = JT % 4 != 0 ? 1 : 0;
  static constexpr int XI = xelem ? IN_ : IN_ / 4;             // staged items of a row
  static constexpr int cost1 = AC1 * (2 * JTK * 3 + IHT * 3), cost2 = AC2 * (NJH * IHK * 3 + JTT * 3);
  static constexpr int NF1 = AC1 * 2 * JTK * 2, NF2 = AC2 * NJH * IHK * 2;      // fragment registers (xh8) of a unit
  static constexpr int NA1 = AC1 * IHT, NA2 = AC2 * JTT;                        // accumulators (f32x4) of a unit
  static_assert(IT % 16 == 0 && U1 <= 4 && U2 <= 4 && (IN_ % 4 == 0 || xelem) && JH * JT == IN_, "shape");
};

// which waves take which (matrix, chain), and where its registers sit (a base class: the plan below uses these in initialisers)
template <int NMAT_, class M0_, class M1_>
struct C2RW {
  // wave halves: the LAST matrix (hidden) puts chain 1 on waves 0-3 and chain 2 on 4-7; the input matrix goes the way that
  // balances the halves' MFMA counts
  static constexpr int H1 = NMAT_ == 2 ? M1_::cost1 : M0_::cost1, H2 = NMAT_ == 2 ? M1_::cost2 : M0_::cost2;
  static constexpr bool in_swap = NMAT_ == 2 && c2r_max(H1 + M0_::cost2, H2 + M0_::cost1) <= c2r_max(H1 + M0_::cost1, H2 + M0_::cost2);
  static constexpr int w0(int mi, int ch) {
    const bool hid = mi == NMAT_ - 1;
    return hid ? (ch == 0 ? 0 : 4) : (in_swap ? (ch == 0 ? 4 : 0) : (ch == 0 ? 0 : 4));
  }
  static constexpr int nf(int mi, int ch) { return mi == 0 ? (ch == 0 ? M0_::NF1 : M0_::NF2) : (ch == 0 ? M1_::NF1 : M1_::NF2); }
  static constexpr int na(int mi, int ch) { return mi == 0 ? (ch == 0 ? M0_::NA1 : M0_::NA2) : (ch == 0 ? M1_::NA1 : M1_::NA2); }
  static constexpr int off(int mi, int ch, bool frags) {      // pairs are ordered (0,0) (0,1) (1,0) (1,1) within a half
    int o = 0;
    for (int m = 0; m < NMAT_; ++m)
      for (int c = 0; c < 2; ++c) {
        if (m == mi && c == ch) return o;
        if (w0(m, c) == w0(mi, ch)) o += frags ? nf(m, c) : na(m, c);
      }
    return o;
  }
  static constexpr int foff(int mi, int ch) { return off(mi, ch, true); }
  static constexpr int aoff(int mi, int ch) { return off(mi, ch, false); }
  static constexpr int half_sum(int base, bool frags) {
    int o = 0;
    for (int m = 0; m < NMAT_; ++m)
      for (int c = 0; c < 2; ++c)
        if (w0(m, c) == base) o += frags ? nf(m, c) : na(m, c);
    return o;
  }
};

// NMAT matrices (call order: input first) sharing one pass over dy; NBR rows per block.
template <int NMAT_, int NBR_, class M0_, class M1_>
struct C2RS : C2RW<NMAT_, M0_, M1_> {
  using W = C2RW<NMAT_, M0_, M1_>;
  static constexpr int NMAT = NMAT_, NBR = NBR_;
  using M0 = M0_; using M1 = M1_;
  static constexpr int Ih = M0::Ih, It = M0::It, OUT = Ih * It, R = M0::R;
  static constexpr int NW = 8, NT = 512;
  static constexpr int NWF = c2r_max(W::half_sum(0, true), W::half_sum(4, true));
  static constexpr int NACC = c2r_max(W::half_sum(0, false), W::half_sum(4, false));
  // LDS: per buffer [dy: 2 planes][x of matrix 0: 2 planes][x of matrix 1: 2 planes]
  static constexpr int DS = It + 8;
  // rows past the block's: the k-block overreach of chain 2's last row (zero weights meet them: they only have to be finite —
  // zero-filled once), and one more where chain 1's second i_t tile runs past an odd number of tiles (I_t = 48)
  static constexpr int DYROWS = NBR * Ih + (M0::IHK * 32 - Ih) + (M0::ITT % 2 ? 1 : 0);
  static constexpr int DYP = DYROWS * DS;                                  // plane (halves)
  static constexpr int XP0 = NBR * M0::XR * M0::XS, XP1 = NMAT == 2 ? NBR * M1::XR * M1::XS : 0;
  static constexpr int L_DY = 0, L_X0 = 2 * DYP * 2, L_X1 = L_X0 + 2 * XP0 * 2;
  static constexpr int L_DUMP = ((L_X1 + 2 * XP1 * 2) + 15) & ~15;      // 16 bytes nobody reads: where threads without a staged item store
  static constexpr int BUFB = L_DUMP + 16;
  static constexpr int L_BIAS = 2 * BUFB;                              // column sums of the staged dy quads (fp32, one slot per thread and quad)
  static constexpr int LDS = L_BIAS + NBR * OUT * 4;
  static constexpr int EQ = (NBR * OUT / 4 + NT - 1) / NT;
  static constexpr int XQ0 = (NBR * M0::XI + NT - 1) / NT, XQ1 = NMAT == 2 ? (NBR * M1::XI + NT - 1) / NT : 1;
  static constexpr int XQ = c2r_max(XQ0, XQ1);
  // output tiles of the reduction: per matrix C tiles (a, jh, mt) then D tiles (a, nt, jt)
  static constexpr int NOUT0 = M0::R * M0::JHT * M0::IHT + M0::R * M0::ITT * M0::JTT;
  static constexpr int NOUT = NOUT0 + (NMAT == 2 ? M1::R * M1::JHT * M1::IHT + M1::R * M1::ITT * M1::JTT : 0);
  static_assert(LDS <= 160 * 1024 && OUT % 4 == 0 && (NMAT == 1 || (M1::Ih == Ih && M1::It == It && M1::R == R)), "plan");
};

struct C2RMatArgs {
  const float* x; const float* first; int T;
  const int* hdr; const _Float16* gtf; const _Float16* ghf;
};
struct C2RArgs {
  C2RMatArgs a[2];
  const float* dy;
  float* part;               // [grid][8 waves][NACC][64] f32x4
  float* bpart;              // [grid][OUT] or NULL
  long n_rows;
  unsigned long long* diag;  // -DTTRNN_ABLATIONS builds only: [2 waves][8] cycle sums per segment of the block loop (workgroup 0, waves 0 and 5)
  int abl;                   // -DTTRNN_ABLATIONS builds only (tools/c2w_bench.py): option dev2 >> 16 — 1: no chain 1, 2: no chain 2, 4: no staging
                             // stores, 8: no global loads, 16: no barrier (result-destroying; 0 in libttrnn.so)
};

__device__ __forceinline__ void c2r_split8(const f32x4 lo, const f32x4 hi, int e, xh8& p0, xh8& p1) {
  unsigned a[4], b[4];
  split_pair_h(ldexpf(lo[0], e), ldexpf(lo[1], e), a[0], b[0]);
  split_pair_h(ldexpf(lo[2], e), ldexpf(lo[3], e), a[1], b[1]);
  split_pair_h(ldexpf(hi[0], e), ldexpf(hi[1], e), a[2], b[2]);
  split_pair_h(ldexpf(hi[2], e), ldexpf(hi[3], e), a[3], b[3]);
  p0 = __builtin_bit_cast(xh8, u32x4{a[0], a[1], a[2], a[3]});
  p1 = __builtin_bit_cast(xh8, u32x4{b[0], b[1], b[2], b[3]});
}
// two groups of four consecutive halves -> one k-packed operand
__device__ __forceinline__ xh8 c2r_ld44(const _Float16* lo, const _Float16* hi) {
  const u32x2 a = *reinterpret_cast<const u32x2*>(lo), b = *reinterpret_cast<const u32x2*>(hi);
  return __builtin_bit_cast(xh8, u32x4{a[0], a[1], b[0], b[1]});
}
// transposed: rows (k) p[0..3 rows], q[0..3 rows] of 16 columns -> lane c holds column c's eight values
__device__ __forceinline__ xh8 c2r_tr44(const _Float16* p, const _Float16* q) {
  const c2_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c2_lds_s16x4*)(p));
  const c2_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c2_lds_s16x4*)(q));
  const c2_s16x8 v = c2_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(xh8, v);
}

// fragments of a (matrix, chain) unit -> registers.  chain 1: [ai][h][kb][plane] Gt tile (a, it = 2 kc + h); chain 2: [ai][h][kb][plane]
// Gh^T tile (a, jh = 2 kd + h).  Tiles past the shape are zeros.
template <class M, int CH>
__device__ __forceinline__ void c2r_load_frags(xh8* wf, const C2RMatArgs& a, int u, int lane) {
  constexpr int AC = CH == 0 ? M::AC1 : M::AC2;
  constexpr int RA = M::R / AC;
  const int grp = u / RA, a0 = (u % RA) * AC;
  if (CH == 0) {
    const int kc = grp % M::KC;
#pragma unroll
    for (int ai = 0; ai < AC; ++ai)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int kb = 0; kb < M::JTK; ++kb)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int it = 2 * kc + h;
            xh8 v = {};
            if (it < M::ITT) v = *reinterpret_cast<const xh8*>(a.gtf + ((size_t)((((a0 + ai) * M::ITT + it) * M::JTK + kb) * 2 + p) * 64 + lane) * 8);
            asm volatile("" : "+v"(v));
            wf[((ai * 2 + h) * M::JTK + kb) * 2 + p] = v;
          }
  } else {
    const int kd = grp % M::JHK;
#pragma unroll
    for (int ai = 0; ai < AC; ++ai)
#pragma unroll
      for (int h = 0; h < M::NJH; ++h)
#pragma unroll
        for (int kb = 0; kb < M::IHK; ++kb)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int jh = 2 * kd + h;
            xh8 v = {};
            if (jh < M::JHT) v = *reinterpret_cast<const xh8*>(a.ghf + ((size_t)((((a0 + ai) * M::JHT + jh) * M::IHK + kb) * 2 + p) * 64 + lane) * 8);
            asm volatile("" : "+v"(v));
            wf[((ai * M::NJH + h) * M::IHK + kb) * 2 + p] = v;
          }
  }
}

// chain 1 of one staged row: unit u = (group (jh, kc), ranks a0 .. a0 + AC - 1)
template <class M, int DS>
__device__ __forceinline__ void c2r_chain1(const xh8* wf, f32x4* acc, const _Float16* X0, const _Float16* X1, const _Float16* D0,
                                           const _Float16* D1, int u, int eA, int c, int g) {
  constexpr int AC = M::AC1, RA = M::R / AC;
  const int grp = u / RA;
  const int jh = grp / M::KC, kc = grp % M::KC;
  // B operand of phase A: x[j_h = 16 jh + c][j_t = 32 kb + 8 g ..]
  xh8 xb[M::JTK][2];
  const int xo = (16 * jh + c) * M::XS + 8 * g;
#pragma unroll
  for (int kb = 0; kb < M::JTK; ++kb) { xb[kb][0] = c2_ld8(X0 + xo + 32 * kb); xb[kb][1] = c2_ld8(X1 + xo + 32 * kb); }
  // A operand of phase C: dy[i_h = 16 mt + c][i_t = 32 kc + 4 g .. | 32 kc + 16 + 4 g ..]
  xh8 da[M::IHT][2];
  const int dyo = c * DS + 32 * kc + 4 * g;
#pragma unroll
  for (int mt = 0; mt < M::IHT; ++mt) {
    da[mt][0] = c2r_ld44(D0 + dyo + 16 * mt * DS, D0 + dyo + 16 * mt * DS + 16);
    da[mt][1] = c2r_ld44(D1 + dyo + 16 * mt * DS, D1 + dyo + 16 * mt * DS + 16);
  }
  // the row-local tiles of ALL ranks first, then their splits, then the gradient MFMAs: a rank's A -> split -> C is one dependent
  // chain, the ranks' chains overlap
  f32x4 r0[AC], r1[AC];
#pragma unroll
  for (int ai = 0; ai < AC; ++ai) {
    r0[ai] = f32x4{0.f, 0.f, 0.f, 0.f}; r1[ai] = r0[ai];
#pragma unroll
    for (int kb = 0; kb < M::JTK; ++kb) {
      r0[ai] = c2_mma3(wf[((ai * 2 + 0) * M::JTK + kb) * 2], wf[((ai * 2 + 0) * M::JTK + kb) * 2 + 1], xb[kb][0], xb[kb][1], r0[ai]);
      r1[ai] = c2_mma3(wf[((ai * 2 + 1) * M::JTK + kb) * 2], wf[((ai * 2 + 1) * M::JTK + kb) * 2 + 1], xb[kb][0], xb[kb][1], r1[ai]);
    }
  }
#pragma unroll
  for (int ai = 0; ai < AC; ++ai) {
    xh8 b0, b1;
    c2r_split8(r0[ai], r1[ai], eA, b0, b1);
#pragma unroll
    for (int mt = 0; mt < M::IHT; ++mt) acc[ai * M::IHT + mt] = c2_mma3(da[mt][0], da[mt][1], b0, b1, acc[ai * M::IHT + mt]);
  }
}

// chain 2 of one staged row: unit u = (group (nt, kd), ranks a0 .. a0 + AC - 1)
template <class M, int DS>
__device__ __forceinline__ void c2r_chain2(const xh8* wf, f32x4* acc, const _Float16* X0, const _Float16* X1, const _Float16* D0,
                                           const _Float16* D1, int u, int eB, int g, int qq, int pp) {
  constexpr int AC = M::AC2, RA = M::R / AC;
  const int grp = u / RA;
  const int nt = grp / M::JHK, kd = grp % M::JHK;
  // B operand of phase B: dy[k = i_h = 32 kb + 8 g + ..][n = i_t = 16 nt + c] (transposing read)
  xh8 db[M::IHK][2];
  const int dbase = (8 * g + qq) * DS + 16 * nt + 4 * pp;
#pragma unroll
  for (int kb = 0; kb < M::IHK; ++kb) {
    db[kb][0] = c2_tr8(D0 + dbase + 32 * kb * DS, DS);
    db[kb][1] = c2_tr8(D1 + dbase + 32 * kb * DS, DS);
  }
  // B operand of phase D: x[k = j_h = 32 kd + 4 g + .. | + 16][n = j_t = 16 jt + c]
  xh8 xt[M::JTT][2];
  const int xbase = (32 * kd + 4 * g + qq) * M::XS + 4 * pp;
#pragma unroll
  for (int jt = 0; jt < M::JTT; ++jt) {
    xt[jt][0] = c2r_tr44(X0 + xbase + 16 * jt, X0 + xbase + 16 * jt + 16 * M::XS);
    xt[jt][1] = c2r_tr44(X1 + xbase + 16 * jt, X1 + xbase + 16 * jt + 16 * M::XS);
  }
  f32x4 r0[AC], r1[AC];
#pragma unroll
  for (int ai = 0; ai < AC; ++ai) {
    r0[ai] = f32x4{0.f, 0.f, 0.f, 0.f}; r1[ai] = r0[ai];
#pragma unroll
    for (int kb = 0; kb < M::IHK; ++kb) {
      r0[ai] = c2_mma3(wf[((ai * M::NJH + 0) * M::IHK + kb) * 2], wf[((ai * M::NJH + 0) * M::IHK + kb) * 2 + 1], db[kb][0], db[kb][1], r0[ai]);
      if (M::NJH == 2)
        r1[ai] = c2_mma3(wf[((ai * M::NJH + M::NJH - 1) * M::IHK + kb) * 2], wf[((ai * M::NJH + M::NJH - 1) * M::IHK + kb) * 2 + 1], db[kb][0], db[kb][1], r1[ai]);
    }
  }
#pragma unroll
  for (int ai = 0; ai < AC; ++ai) {
    xh8 a0, a1;
    c2r_split8(r0[ai], r1[ai], eB, a0, a1);
#pragma unroll
    for (int jt = 0; jt < M::JTT; ++jt) acc[ai * M::JTT + jt] = c2_mma3(a0, a1, xt[jt][0], xt[jt][1], acc[ai * M::JTT + jt]);
  }
}

template <class S>
__global__ void __launch_bounds__(512) k_c2r(C2RArgs g) {
  using M0 = typename S::M0;
  using M1 = typename S::M1;
  constexpr int NT = S::NT, NBR = S::NBR, OUT = S::OUT, DS = S::DS, EQ = S::EQ, XQ = S::XQ, NMAT = S::NMAT;
  extern __shared__ __attribute__((aligned(16))) unsigned char c2_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, gq = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const long nblk = (g.n_rows + NBR - 1) / NBR;

  for (int i = tid; i < S::LDS / 16; i += NT) reinterpret_cast<f32x4*>(c2_smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- this wave's units: weight fragments into registers, accumulators ----
  xh8 wf[S::NWF];
  f32x4 acc[S::NACC];
#pragma unroll
  for (int i = 0; i < S::NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < S::NWF; ++i) wf[i] = xh8{};
  const int half = wave >> 2, uw = wave & 3;
  int eA[2] = {0, 0}, eB[2] = {0, 0};
  float sxf[2] = {1.f, 1.f};
  float sdf = 1.f;
  {
    const int* h = g.a[0].hdr;
    sxf[0] = ldexpf(1.f, 14 - h[0]); sdf = ldexpf(1.f, 14 - h[1]);
    eA[0] = h[2] + h[0] - h[4] - 14; eB[0] = h[3] + h[1] - h[5] - 14;
    if (S::w0(0, 0) == 4 * half && uw < M0::U1) c2r_load_frags<M0, 0>(wf + S::foff(0, 0), g.a[0], uw, lane);
    if (S::w0(0, 1) == 4 * half && uw < M0::U2) c2r_load_frags<M0, 1>(wf + S::foff(0, 1), g.a[0], uw, lane);
  }
  if (NMAT == 2) {
    const int* h = g.a[1].hdr;
    sxf[1] = ldexpf(1.f, 14 - h[0]);
    eA[1] = h[2] + h[0] - h[4] - 14; eB[1] = h[3] + h[1] - h[5] - 14;
    if (S::w0(1, 0) == 4 * half && uw < M1::U1) c2r_load_frags<M1, 0>(wf + S::foff(1, 0), g.a[1], uw, lane);
    if (S::w0(1, 1) == 4 * half && uw < M1::U2) c2r_load_frags<M1, 1>(wf + S::foff(1, 1), g.a[1], uw, lane);
  }
  __syncthreads();

  // ---- staging plan (k_c2w's: every load unconditional and clamped, what must not count zeroed by a keep factor at the store) ----
  constexpr int oq = OUT / 4;
  int dyo[EQ], dyg[EQ], dyr[EQ];
  f32x4 sd[EQ];
  float dk[EQ];
#pragma unroll
  for (int e = 0; e < EQ; ++e) {
    const int id = tid + NT * e;
    const bool on = id < NBR * oq;
    const int row = on ? id / oq : 0, o = on ? 4 * (id % oq) : 0;
    dyg[e] = on ? row * OUT + o : -1;
    dyr[e] = row;
    dyo[e] = on ? (row * S::Ih + o / S::It) * DS + (o % S::It) : -1;
    sd[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    dk[e] = 0.f;
  }
  int xo[NMAT][XQ], xcol[NMAT][XQ], xr[NMAT][XQ];
  unsigned xb[NMAT][XQ], xt[NMAT][XQ];
  f32x4 sx[NMAT][XQ];
  float xk[NMAT][XQ];
  const long per = (nblk + gridDim.x - 1) / gridDim.x;
  const long blk0 = (long)blockIdx.x * per, blk1 = blk0 + per < nblk ? blk0 + per : nblk;
  unsigned stq[NMAT], str[NMAT], bmaxs[NMAT];
#pragma unroll
  for (int mi = 0; mi < NMAT; ++mi) {
    const int XI = mi == 0 ? M0::XI : M1::XI, xel = mi == 0 ? M0::xelem : M1::xelem, Jt = mi == 0 ? M0::Jt : M1::Jt;
    const int XR = mi == 0 ? M0::XR : M1::XR, XS = mi == 0 ? M0::XS : M1::XS;
    const int iw = xel ? 1 : 4;
    const unsigned T = g.a[mi].T > 0 ? (unsigned)g.a[mi].T : 1u;
#pragma unroll
    for (int e = 0; e < XQ; ++e) {
      const int id = tid + NT * e;
      const bool on = id < NBR * XI;
      const int row = on ? id / XI : 0, col = on ? iw * (id % XI) : 0;
      xr[mi][e] = on ? row : -1;
      xcol[mi][e] = col;
      xo[mi][e] = on ? (row * XR + col / Jt) * XS + (col % Jt) : -1;
      sx[mi][e] = f32x4{0.f, 0.f, 0.f, 0.f};
      xk[mi][e] = 0.f;
      const unsigned n = (unsigned)blk0 * (unsigned)NBR + (unsigned)row;
      xb[mi][e] = n / T;
      xt[mi][e] = n - xb[mi][e] * T;
    }
    stq[mi] = (unsigned)NBR / T;
    str[mi] = (unsigned)NBR - stq[mi] * T;
    // index of the last sample, ONCE: inside load_block the 64-bit division (a branchy ~150-instruction expansion the compiler
    // does not hoist) ran per matrix and block — 600 cycles each in the stamps, the bulk of "load issue"
    unsigned bm = (unsigned)(g.n_rows / (long)T) - 1u;
    asm volatile("" : "+s"(bm));
    bmaxs[mi] = bm;
  }
  const bool want_bias = g.bpart != nullptr;

  // part p of NP (p < 0: everything): dy quad e goes with part e % NP, the rows of x with the last part
  auto load_block = [&](long blk, bool live, int part, int np) {
    const long n0 = blk * NBR;
    const long last = live ? g.n_rows - 1 : -1;
    const long lastv = g.n_rows - 1;
#pragma unroll
    for (int e = 0; e < EQ; ++e) {
      if (part >= 0 && e % np != part) continue;
      const int off = dyg[e] >= 0 ? dyg[e] : 0, row = dyg[e] >= 0 ? dyr[e] : 0;
      const bool in = n0 + row <= last;
      sd[e] = *reinterpret_cast<const f32x4*>(g.dy + (size_t)(in ? n0 : lastv) * OUT + (in ? off : off - row * OUT));
      dk[e] = (in && dyg[e] >= 0) ? 1.f : 0.f;
    }
    if (part >= 0 && part != np - 1) return;
#pragma unroll
    for (int mi = 0; mi < NMAT; ++mi) {
      const C2RMatArgs& a = g.a[mi];
      const int in_sz = mi == 0 ? M0::in : M1::in, xel = mi == 0 ? M0::xelem : M1::xelem;
#pragma unroll
      for (int e = 0; e < XQ; ++e) {
        const int row = xr[mi][e] >= 0 ? xr[mi][e] : 0;
        const long n = n0 + row;
        const bool in = n <= last && xr[mi][e] >= 0;
        const long nn = n <= lastv ? n : lastv;
        // (branch-free: a branch between a load and its store is a basic-block boundary the compiler sinks the load across — the
        // hidden matrix's rows were fetched right in front of their split, their whole latency exposed)
        const bool seq = a.T > 0;
        const unsigned Tm = seq ? (unsigned)a.T : 1u;
        const bool head = seq && xt[mi][e] == 0;
        // row n - 1 of the outputs; a head row takes its sample's initial state, or (none given) any valid row and is zeroed
        const unsigned bmax = bmaxs[mi];
        const unsigned bb = xb[mi][e] < bmax ? xb[mi][e] : bmax;
        const long idx = (seq && !head && nn != 0) ? nn - 1 : nn;
        const bool use_first = head && a.first != nullptr;
        const float* src = use_first ? a.first + (size_t)bb * in_sz : a.x + (size_t)idx * in_sz;
        const bool zero = !in || (head && a.first == nullptr);
        const unsigned t2 = xt[mi][e] + str[mi];
        const bool wrap = t2 >= Tm;
        xt[mi][e] = wrap ? t2 - Tm : t2;
        xb[mi][e] += stq[mi] + (wrap ? 1u : 0u);
        if (xel) sx[mi][e] = f32x4{src[xcol[mi][e]], 0.f, 0.f, 0.f};
        else sx[mi][e] = *reinterpret_cast<const f32x4*>(src + xcol[mi][e]);
        xk[mi][e] = zero ? 0.f : 1.f;
      }
    }
  };
  float* bsum = reinterpret_cast<float*>(c2_smem + S::L_BIAS);
  auto store_block = [&](unsigned char* buf) {
    _Float16* dYs = reinterpret_cast<_Float16*>(buf + S::L_DY);
    _Float16* dump = reinterpret_cast<_Float16*>(buf + S::L_DUMP);
#pragma unroll
    for (int e = 0; e < EQ; ++e) {
      const f32x4 v = sd[e] * dk[e];
      c2_store4(dyo[e] >= 0 ? dYs + dyo[e] : dump, dyo[e] >= 0 ? dYs + S::DYP + dyo[e] : dump + 4, v * sdf);
      // the bias gradients' partial sums live in LDS (a slot belongs to ONE thread: a plain read - add - write, a fixed order) —
      // twelve registers less through the chains, where the LSTM pair spilled
      if (dyg[e] >= 0) *reinterpret_cast<f32x4*>(bsum + dyg[e]) += v;
    }
#pragma unroll
    for (int mi = 0; mi < NMAT; ++mi) {
      _Float16* X0 = reinterpret_cast<_Float16*>(buf + (mi == 0 ? S::L_X0 : S::L_X1));
      const int xpl = mi == 0 ? S::XP0 : S::XP1;
      const int xel = mi == 0 ? M0::xelem : M1::xelem;
#pragma unroll
      for (int e = 0; e < XQ; ++e) {
        _Float16* q0 = xo[mi][e] >= 0 ? X0 + xo[mi][e] : dump;
        _Float16* q1 = xo[mi][e] >= 0 ? X0 + xpl + xo[mi][e] : dump + 4;
        if (xel) {
          _Float16 p0, p1;
          split2h(sx[mi][e][0] * (sxf[mi] * xk[mi][e]), p0, p1);
          *q0 = p0;
          *q1 = p1;
        } else {
          c2_store4(q0, q1, sx[mi][e] * (sxf[mi] * xk[mi][e]));
        }
      }
    }
  };

#ifdef TTRNN_ABLATIONS
  const int abl = g.abl;
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = stamp();
#define C2R_STAMP(i) { const unsigned long long now_ = stamp(); seg[i] += now_ - last_; last_ = now_; }
#else
  constexpr int abl = 0;
#define C2R_STAMP(i)
#endif
  if (blk0 < blk1) {
    load_block(blk0, true, -1, 1);
    store_block(c2_smem);
  }
  lds_barrier();
  // The two halves of the workgroup (waves 0-3 / 4-7: a SIMD runs one wave of each) run the SAME steps of a block — the chains of
  // its rows, the next block's loads and, once those have arrived, their split + stores into the other buffer — in a DIFFERENT
  // order: half 0 issues its loads first and stores before its last chain, half 1 issues after its first chain and stores last
  // (all eight waves issuing a block's loads in one burst overflow the CU's memory queue: 700 - 900 cycles of issue per block in
  // the stamps).  One compile-time body per half: the chains it runs are literals, and no branch sits between a load and its store
  // (the wait counts the compiler inserts stay exact).  Loads are issued and consumed INSIDE an iteration: carried across the back
  // edge (two blocks ahead) they cost sixteen more live registers through the chains — the LSTM pair then spills.
  auto body = [&](auto HC) {
    constexpr int H = decltype(HC)::value;
    constexpr int CH0 = S::w0(0, 0) == 4 * H ? 0 : 1;                      // this half's chain of matrix 0 / matrix 1
    constexpr int CH1 = NMAT == 2 ? (S::w0(1, 0) == 4 * H ? 0 : 1) : 0;
    constexpr int NSTEP = NBR * NMAT;
    constexpr int LOAD_AT = H == 0 ? 0 : 1;                                // before step ...
    constexpr int STORE_AT = H == 0 ? (NSTEP > 1 ? NSTEP - 1 : NSTEP) : NSTEP;
    constexpr int UN0 = CH0 == 0 ? M0::U1 : M0::U2, UN1 = CH1 == 0 ? M1::U1 : M1::U2;
    const bool on0 = UN0 == 4 || uw < UN0;                                 // (four units: no branch around the chain)
    const bool on1 = NMAT == 2 && (UN1 == 4 || uw < UN1);
    int par = 0;
    for (long blk = blk0; blk < blk1; ++blk) {
      C2R_STAMP(7)
      const bool more = blk + 1 < blk1;
      const unsigned char* buf = c2_smem + par * S::BUFB;
      unsigned char* nbuf = c2_smem + (par ^ 1) * S::BUFB;                 // (the other buffer: everybody left it at the last barrier)
      const _Float16* D0 = reinterpret_cast<const _Float16*>(buf + S::L_DY);
      const _Float16* D1 = D0 + S::DYP;
#pragma unroll
      for (int st = 0; st <= NSTEP; ++st) {
        if (st == LOAD_AT && !(abl & 8)) {
          __builtin_amdgcn_sched_barrier(0);
          load_block(more ? blk + 1 : blk, more, -1, 1);
          __builtin_amdgcn_sched_barrier(0);      // (the loads stay HERE: under register pressure the scheduler moves them down to their use)
          C2R_STAMP(0)
        }
        if (st == STORE_AT && !(abl & 4)) {
          __builtin_amdgcn_sched_barrier(0);      // (... and their split + stores stay here: hoisted, they waited for the loads right behind their issue)
          store_block(nbuf);
          C2R_STAMP(5)
        }
        if (st == NSTEP) break;
        const int r = st / NMAT, mi = st % NMAT;
        const _Float16* d0 = D0 + r * S::Ih * DS;
        const _Float16* d1 = D1 + r * S::Ih * DS;
        if (mi == 0) {
          const _Float16* X0 = reinterpret_cast<const _Float16*>(buf + S::L_X0) + r * M0::XR * M0::XS;
          const _Float16* X1 = X0 + S::XP0;
          if (CH0 == 0) { if (on0 && !(abl & 1)) c2r_chain1<M0, DS>(wf + S::foff(0, 0), acc + S::aoff(0, 0), X0, X1, d0, d1, uw, eA[0], c, gq); }
          else { if (on0 && !(abl & 2)) c2r_chain2<M0, DS>(wf + S::foff(0, 1), acc + S::aoff(0, 1), X0, X1, d0, d1, uw, eB[0], gq, qq, pp); }
        } else {
          const _Float16* X0 = reinterpret_cast<const _Float16*>(buf + S::L_X1) + r * M1::XR * M1::XS;
          const _Float16* X1 = X0 + S::XP1;
          if (CH1 == 0) { if (on1 && !(abl & 1)) c2r_chain1<M1, DS>(wf + S::foff(1, 0), acc + S::aoff(1, 0), X0, X1, d0, d1, uw, eA[1], c, gq); }
          else { if (on1 && !(abl & 2)) c2r_chain2<M1, DS>(wf + S::foff(1, 1), acc + S::aoff(1, 1), X0, X1, d0, d1, uw, eB[1], gq, qq, pp); }
        }
        C2R_STAMP(1 + (st < 4 ? st : 3))
      }
      par ^= 1;
      if (!(abl & 16)) lds_barrier();
      C2R_STAMP(6)
    }
  };
  if (half == 0) body(std::integral_constant<int, 0>{});
  else body(std::integral_constant<int, 1>{});
#ifdef TTRNN_ABLATIONS
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 5) && g.diag)
    for (int i = 0; i < 8; ++i) g.diag[(wave ? 8 : 0) + i] = seg[i];
#endif

  // ---- partial sums out: [workgroup][wave][slot][lane] f32x4 ----
#pragma unroll
  for (int i = 0; i < S::NACC; ++i)
    reinterpret_cast<f32x4*>(g.part)[(((size_t)blockIdx.x * 8 + wave) * S::NACC + i) * 64 + lane] = acc[i];
  if (want_bias) {
    __syncthreads();
    const float* bl = bsum;
    for (int o = tid; o < OUT; o += NT) {
      float s = bl[o];
      for (int r = 1; r < NBR; ++r) s += bl[r * OUT + o];
      g.bpart[(size_t)blockIdx.x * OUT + o] = s;
    }
  }
}

// ---- fixed-order reduction: one workgroup per output tile --------------------------------------------------------------------
struct C2RRed {
  TtShape s[2];
  const float* part; int grid;
  const int* hdr[2];
  float* d_packed[2]; float* dGh[2]; float* dGt[2];
};

template <class S, class M, int MI>
__device__ void c2r_reduce_tile(const C2RRed& a, int o, f32x4 (*red)[64]) {
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = lane & 15, gq = lane >> 4;
  constexpr int NC = M::R * M::JHT * M::IHT;
  const bool isC = o < NC;
  // decode the tile and its sources (wave, slot)
  int ra, t0, t1, nsrc;
  int sw[2], ss[2];
  if (isC) {
    ra = o / (M::JHT * M::IHT); t0 = (o / M::IHT) % M::JHT; t1 = o % M::IHT;      // (a, jh, mt)
    nsrc = M::KC;
    for (int k = 0; k < M::KC; ++k) {
      const int grpi = t0 * M::KC + k, u = grpi * (M::R / M::AC1) + ra / M::AC1;
      sw[k] = S::w0(MI, 0) + u; ss[k] = S::aoff(MI, 0) + (ra % M::AC1) * M::IHT + t1;
    }
  } else {
    const int v = o - NC;
    ra = v / (M::ITT * M::JTT); t0 = (v / M::JTT) % M::ITT; t1 = v % M::JTT;      // (a, nt, jt)
    nsrc = M::JHK;
    for (int k = 0; k < M::JHK; ++k) {
      const int grpi = t0 * M::JHK + k, u = grpi * (M::R / M::AC2) + ra / M::AC2;
      sw[k] = S::w0(MI, 1) + u; ss[k] = S::aoff(MI, 1) + (ra % M::AC2) * M::JTT + t1;
    }
  }
  f32x4 tot = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < nsrc; ++k) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int w = grp; w < a.grid; w += 16) v += reinterpret_cast<const f32x4*>(a.part)[(((size_t)w * 8 + sw[k]) * S::NACC + ss[k]) * 64 + lane];
    __syncthreads();
    red[grp][lane] = v;
    __syncthreads();
    if (grp == 0) {
      v = red[0][lane];
#pragma unroll
      for (int q = 1; q < 16; ++q) v += red[q][lane];
      tot += v;
    }
  }
  if (grp != 0) return;
  const TtShape& s = a.s[MI];
  const int* hdr = a.hdr[MI];
  const int ex = hdr[0], ed = hdr[1], ec1 = hdr[4], edc = hdr[5];
  if (isC) {
    const int jh = 16 * t0 + c;
    if (jh >= M::Jh) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ih = 16 * t1 + 4 * gq + j;
      if (ih >= M::Ih) continue;
      const float val = ldexpf(ldexpf(tot[j], ed - 14), ec1 - 14);
      if (s.d == 2) a.d_packed[MI][s.woff[0] + (size_t)(jh * M::R + ra) * s.M[0] + ih] += val;
      else a.dGh[MI][((size_t)ih * M::Jh + jh) * M::R + ra] = val;
    }
  } else {
    const int jt = 16 * t1 + c;
    if (jt >= M::Jt) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int it = 16 * t0 + 4 * gq + j;
      const float val = ldexpf(ldexpf(tot[j], edc - 14), ex - 14);
      if (s.d == 2) a.d_packed[MI][s.woff[1] + (size_t)jt * s.M[1] + it * M::R + ra] += val;
      else a.dGt[MI][((size_t)it * M::Jt + jt) * M::R + ra] = val;
    }
  }
}

template <class S>
__global__ void __launch_bounds__(1024) k_c2r_reduce(C2RRed a) {
  __shared__ f32x4 red[16][64];
  const int o = blockIdx.x;
  if (o < S::NOUT0) c2r_reduce_tile<S, typename S::M0, 0>(a, o, red);
  else if (S::NMAT == 2) c2r_reduce_tile<S, typename S::M1, 1>(a, o - S::NOUT0, red);
}
