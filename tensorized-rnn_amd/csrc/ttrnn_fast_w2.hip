// ttrnn_fast_w2.hip — TT-LSTM forward for the reference's own published encoder shape: hidden 768, TWO cores, rank 2, 40 mel
// channels in (experiments/speaker_verification/encoder/params_model.py:2-4,14-16; tt_shape: input (5, 8) x (48, 64), hidden
// (24, 32) x (48, 64)).  Replaces tensorized_rnn/lstm.py:23-32,101-135 + t3nsor/ops.py:78-93 for that layer — INCLUDING the
// input projection: no hoisted K-in GEMM and no [B][T][4H] buffer of pre-activations (1 GB at the encoder's batch, written and
// read back once per forward by every other route).
//
// A sample is a workgroup of FOUR waves; wave w owns the output-mode slice i1 in [16 w, 16 w + 16) through BOTH chain stages
// (the structure of ttrnn_fast_f2.hip, whose H = 128 kernel it generalises), all operands on two fp16 pieces, fp32 accumulation:
//   stage 1   C1_a[j0][i1] = sum_j1 h[j0][j1] Gt[i1][j1][a]: the state image (LDS, [j0][j1]) is the A operand, the wave's slice of
//             core 1 the B operand (registers); one MFMA tile per (rank index a, sixteen j0) — 12 MFMAs; the same for the five
//             input chain rows x[j0'][j1'] against the input matrix's core 1 — 6 MFMAs;
//   hand-off  NONE: an accumulator tile has its column i1 on the lane and four consecutive j0 in the registers — exactly what
//             stage 2's B operand wants when its contraction index runs (j0 block, a, j0 in block): the lane converts its own
//             eight values (2 ranks x 4 rows) to two fp16 pieces and feeds them back.  The input matrix's chain rows ride in the
//             k-slots that the hidden matrix's 24 = 16 + 8 rows leave empty in the second k-block: the input projection costs
//             no stage-2 MFMA at all;
//   stage 2   y[(r, gate)][i1] = sum_k Gh[(r, gate)][k] C1[k][i1], three row tiles of core 0 (both matrices' head cores side by
//             side along k, rows permuted so that a lane's four accumulator registers are i, f, g, o of ONE hidden unit) — 18 MFMAs;
//   gates     on the accumulators (lstm.py:26-32); c in registers; h_t as two fp16 pieces into the other parity of the state
//             image; ONE barrier per step.
// The same kernels serve the FOUR-core models of the reference's result tables at this size (tt_shape: (4, 4, 6, 8) x (6, 8, 8, 8)): cores
// 0 - 1 and 2 - 3 are contracted pairwise once per launch (k_w2_prep), which gives the same (48, 64) output modes over (16, 48) input
// modes — two k-blocks in stage 1, one block of sixteen chain rows, the input chain in k-blocks of its own (configuration B below).
// Scales: powers of two per launch for the weights (k_w2_prep), 2^13 for h (|h| < 1; a caller's h_0 per sample), per STEP for
// x_t (its own maximum, taken by the wave that stages it), per step for the hand-off (the larger of the two chains' bounds).
#include <hip/hip_runtime.h>
#include "ttrnn.h"
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"

namespace ttrnn {
namespace {

// Configurations: the two-core view (J0 x J1 -> 48 x 64 through R rank slots; a smaller real rank fills its slots with zeros) of
//   A  d = 2: hidden (24, 32), input (5, 8)          B  d = 4: hidden (4*4, 6*8) = (16, 48), input (2*2, 2*5) = (4, 10)
// GATES = 3: the TT-GRU of the same layer (speaker_encoder.py:33-47 `use_gru`; gru.py:33-44): output modes (48, 48), three waves, and
// FOUR rows per unit in stage 2 all the same — r, z, the hidden part of n, the input part of n (the n gate multiplies W_hn h + b_hn by
// r before the input part is added: the two chains must not meet in one accumulator, so row 2 carries zeros in the input chain's k-slots
// and row 3 zeros in the hidden chain's).
template <int J0_, int J1_, int J0I_, int J1I_, int R_, int GATES_ = 4>
struct W2T {
  static constexpr int GATES = GATES_;
  // output modes: LSTM (48, 64); TT-GRU, two cores (48, 48); TT-GRU, four cores (6, 6, 8, 8) contracted pairwise: (36, 64)
  static constexpr int J0 = J0_, J1 = J1_, I1 = (GATES_ == 3 && J0_ == 24) ? 48 : 64, I0 = GATES_ * 768 / I1, R = R_, H = J0_ * J1_;
  static constexpr int J0I = J0I_, J1I = J1I_, INP = J0I_ * J1I_;
  static_assert(H == 768 && INP == 40 && J1 % 16 == 0 && J1I <= 32 && J0I <= 8 && (R == 2 || R == 4) && I0 * I1 == GATES_ * H, "encoder shapes");
  static constexpr int NWV = I1 / 16;          // 4 waves, GRU: 3 (blockDim = 64 NWV)
  static constexpr int NR = I0 / GATES_;       // 12 / 16 values of r (unit u = r * I1 + i1)
  static constexpr int MT2 = NR / 4;           // stage-2 row tiles: r in [4 t, 4 t + 4) x 4 rows per unit (LSTM: its gates)
  static constexpr int MT0 = (J0 + 15) / 16;   // blocks of sixteen chain rows j0 (stage-1 row tiles)
  static constexpr int KB1 = (J1 + 31) / 32;   // k-blocks of stage 1
  static constexpr int M1T = J1 / 16;          // reverse T1: row tiles over j1
  static constexpr int AP = R / 2;             // pairs of rank indices: a k-block of stage 2 = (16 chain rows) x (one pair)
  // the input chain rides in the k-slots the last hidden row block leaves empty (A: rows 8 ... 15 of the second block), or gets
  // k-blocks of its own (B: sixteen hidden rows fill their block)
  static constexpr bool RIDE = (J0 % 16) != 0 && (J0 % 16) <= 8;
  static constexpr int XR0 = RIDE ? 8 : 0;     // first row of the input chain rows in the input image
  static constexpr int KB2H = MT0 * AP, KB2 = KB2H + (RIDE ? 0 : AP);      // k-blocks of stage 2 (hidden, all)
  static constexpr bool GHL = MT2 * KB2 * 2 > 12;      // stage-2 A fragments in LDS beyond twelve register slots (48 VGPRs)
  static constexpr int HS = 32 * KB1 + 8;      // row stride (halves) of the state image; 40 for the input image
  static constexpr int HROWS = 16 * MT0, XS = 40;
  // LDS (bytes): state image [2 parity][2 pieces][HROWS][HS], input image [2][2][16][XS], step exponents of x, scratch, (fragments)
  static constexpr int L_H = 0, L_X = 2 * 2 * HROWS * HS * 2, L_E = L_X + 2 * 2 * 16 * XS * 2, L_RED = L_E + 16, L_GH = L_RED + 64;
  static constexpr int GH_BYTES = MT2 * KB2 * 2 * 1024;
  static constexpr int LDS = L_GH + (GHL ? GH_BYTES : 0);
  // workspace: header (ints) | fragments xh8 [tile][piece][64 lanes]: Gt tiles (wave, a, kb), Gt_in tiles (wave, a), Gh tiles (tl, kb)
  static constexpr int HDR_BYTES = 256, T_GTI = NWV * R * KB1, T_GH = T_GTI + NWV * R, NTILES = T_GH + MT2 * KB2;
  static constexpr size_t WS_BYTES = HDR_BYTES + (size_t)NTILES * 2 * 64 * 16;
  // reverse-time kernel
  static constexpr int NT2 = R * MT0;                                       // T2' column tiles: (rank index a, block of sixteen j0)
  static constexpr bool GHLB = NT2 * 2 * 2 > 16;                            // T2' B fragments in LDS beyond sixteen slots
  static constexpr size_t BGH_BYTES = (size_t)NT2 * 2 * 2 * 64 * 16;        // T2' B operand: [n-tile][kb][piece][lane] xh8
  static constexpr size_t BGT_BYTES = (size_t)NWV * R * M1T * 2 * 64 * 8;   // T1 A operand: [wave][a][j1 tile][piece][lane] xh4
  static constexpr size_t BWS_BYTES = HDR_BYTES + BGH_BYTES + BGT_BYTES;
  static constexpr int PART = HROWS * J1;                                   // floats of a wave's partial dh ([j0][j1] = unit order)
  static constexpr int BL_GH = 2 * NWV * PART * 4 + 64;                     // [parity][wave][PART] + scratch, then (if GHLB) the fragments
  static constexpr int BLDS = BL_GH + (GHLB ? (int)BGH_BYTES : 0);
};
typedef W2T<24, 32, 5, 8, 2> W2A2;
typedef W2T<24, 32, 5, 8, 4> W2A4;
typedef W2T<16, 48, 4, 10, 2> W2B2;
typedef W2T<16, 48, 4, 10, 4> W2B4;
typedef W2T<24, 32, 5, 8, 2, 3> W2GA2;      // TT-GRU, two cores, rank slots 2 / 4
typedef W2T<24, 32, 5, 8, 4, 3> W2GA4;
typedef W2T<16, 48, 4, 10, 2, 3> W2GB2;     // TT-GRU, four cores contracted pairwise: (16, 48) x (36, 64)
typedef W2T<16, 48, 4, 10, 4, 3> W2GB4;
enum { W2_EGT = 0, W2_EC1 = 1, W2_EGTI = 2, W2_ECI = 3, W2_EGH = 4 };

__device__ __forceinline__ int w2_expo(float x) {           // x < 2^e; zero / non-finite: neutral; clamped
  if (!(x > 0.f) || !(x <= 3.4028235e38f)) return 0;
  int e;
  frexpf(x, &e);
  return e < -100 ? -100 : e;          // (the scales 2^(13 - e), 2^(14 - e) stay normal fp32 numbers over the whole range)
}
// packed core k: W_k[(j*R_{k+1} + b)*M_k + i*R_k + a]   (a: left rank, b: right rank)
__device__ __forceinline__ float w2_core(const TtShape& s, const float* pk, int k, int a, int i, int j, int b) {
  return pk[s.woff[k] + (size_t)(j * s.R[k + 1] + b) * s.M[k] + i * s.R[k] + a];
}
// tail Gt[i1][j1][a] (d = 2: core 1; d = 4: cores 2, 3 contracted), head Gh[i0][j0][a] (core 0; cores 0, 1); a < s.R[d / 2]
__device__ float w2_gt(const TtShape& s, const float* pk, int i1, int j1, int a) {
  if (s.d == 2) return w2_core(s, pk, 1, a, i1, j1, 0);
  const int ia = i1 / s.I[3], ib = i1 % s.I[3], ja = j1 / s.J[3], jb = j1 % s.J[3];
  float v = 0.f;
  for (int m = 0; m < s.R[3]; ++m) v = fmaf(w2_core(s, pk, 2, a, ia, ja, m), w2_core(s, pk, 3, m, ib, jb, 0), v);
  return v;
}
__device__ float w2_gh(const TtShape& s, const float* pk, int i0, int j0, int a) {
  if (s.d == 2) return w2_core(s, pk, 0, 0, i0, j0, a);
  const int ia = i0 / s.I[1], ib = i0 % s.I[1], ja = j0 / s.J[1], jb = j0 % s.J[1];
  float v = 0.f;
  for (int m = 0; m < s.R[1]; ++m) v = fmaf(w2_core(s, pk, 0, 0, ia, ja, m), w2_core(s, pk, 1, m, ib, jb, a), v);
  return v;
}

__device__ float w2_block_max(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// maxima of the (merged) cores of one matrix and the L1 bounds of the stage results; every workgroup takes what it needs itself
template <class S>
__device__ void w2_maxima(const TtShape& sh, const TtShape& si, const float* pk_hid, const float* pk_in, float* red, float& mgt,
                          float& mgti, float& mgh, float& l1t, float& l1ti, float& l1h, bool bounds) {
  const int tid = threadIdx.x;
  const int rh = sh.R[sh.d / 2], ri = si.R[si.d / 2];
  mgt = mgti = mgh = l1t = l1ti = l1h = 0.f;
  for (int e = tid; e < S::I1 * S::J1 * rh; e += 256) {
    const int a = e % rh, j = (e / rh) % S::J1, i = e / (rh * S::J1);
    mgt = fmaxf(mgt, fabsf(w2_gt(sh, pk_hid, i, j, a)));
  }
  for (int e = tid; e < S::I1 * S::J1I * ri; e += 256) {
    const int a = e % ri, j = (e / ri) % S::J1I, i = e / (ri * S::J1I);
    mgti = fmaxf(mgti, fabsf(w2_gt(si, pk_in, i, j, a)));
  }
  for (int e = tid; e < S::I0 * S::J0 * rh; e += 256) {
    const int a = e % rh, j = (e / rh) % S::J0, i = e / (rh * S::J0);
    mgh = fmaxf(mgh, fabsf(w2_gh(sh, pk_hid, i, j, a)));
  }
  for (int e = tid; e < S::I0 * S::J0I * ri; e += 256) {
    const int a = e % ri, j = (e / ri) % S::J0I, i = e / (ri * S::J0I);
    mgh = fmaxf(mgh, fabsf(w2_gh(si, pk_in, i, j, a)));
  }
  if (bounds) {
    if (tid < S::I1 * S::R) {            // rows (i1, a) of Gt: L1 norms over j1 (bounds of the stage-1 results)
      const int i1 = tid / S::R, a = tid % S::R;
      float s0 = 0.f, s1 = 0.f;
      if (a < rh) for (int j = 0; j < S::J1; ++j) s0 += fabsf(w2_gt(sh, pk_hid, i1, j, a));
      if (a < ri) for (int j = 0; j < S::J1I; ++j) s1 += fabsf(w2_gt(si, pk_in, i1, j, a));
      l1t = s0; l1ti = s1;
    }
    l1t = w2_block_max(l1t, red); l1ti = w2_block_max(l1ti, red);
  }
  mgt = w2_block_max(mgt, red); mgti = w2_block_max(mgti, red); mgh = w2_block_max(mgh, red);
}

// ---- prep: one workgroup per fragment tile (each takes the maxima it needs itself); workgroup 0 also writes the header ----------
template <class S>
__global__ void __launch_bounds__(256) k_w2_prep(TtShape sh, TtShape si, const float* __restrict__ pk_hid, const float* __restrict__ pk_in,
                                                 int* __restrict__ hdr, _Float16* __restrict__ frag) {
  constexpr int R = S::R;
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int rh = sh.R[sh.d / 2], ri = si.R[si.d / 2];                  // the real ranks at the split (<= R)
  float mgt, mgti, mgh, l1t, l1ti, l1h;
  w2_maxima<S>(sh, si, pk_hid, pk_in, red, mgt, mgti, mgh, l1t, l1ti, l1h, blockIdx.x == 0);
  const int egt = w2_expo(mgt), egti = w2_expo(mgti), egh = w2_expo(mgh);
  if (blockIdx.x == 0 && tid == 0) {
    hdr[W2_EGT] = egt; hdr[W2_EC1] = w2_expo(l1t); hdr[W2_EGTI] = egti; hdr[W2_ECI] = w2_expo(l1ti); hdr[W2_EGH] = egh;
  }
  const int tile = blockIdx.x;
  if (tile >= S::NTILES) return;
  _Float16* dst = frag + (size_t)tile * 1024;
  for (int e = tid; e < 512; e += 256) {
    const int j = e & 7, lane = e >> 3, n = lane & 15, g = lane >> 4;
    float v = 0.f;
    if (tile < S::T_GTI) {              // stage-1 B operand of the hidden matrix: (wave w, rank index a, k-block kb): B[k = j1][n = i1 - 16 w]
      const int kb = tile % S::KB1, a = (tile / S::KB1) % R, w = tile / (S::KB1 * R), j1 = 32 * kb + 8 * g + j;
      if (a < rh && j1 < S::J1) v = w2_gt(sh, pk_hid, 16 * w + n, j1, a) * ldexpf(1.f, 14 - egt);
    } else if (tile < S::T_GH) {        // the same of the input matrix: k = j1' < J1I, zero beyond
      const int w = (tile - S::T_GTI) / R, a = (tile - S::T_GTI) % R, j1 = 8 * g + j;
      if (j1 < S::J1I && a < ri) v = w2_gt(si, pk_in, 16 * w + n, j1, a) * ldexpf(1.f, 14 - egti);
    } else {                            // stage-2 A operand: row tile tl, k-block kb: A[row = (rr, gate)][k = (g, a, jj)]
      const int tl = (tile - S::T_GH) / S::KB2, kb = (tile - S::T_GH) % S::KB2;
      // row (rr, slot) of the tile: LSTM slot = gate; GRU slots r, z, n (hidden chain only), n (input chain only)
      const int rr = n >> 2, slot = n & 3, gate = S::GATES == 4 ? slot : (slot < 3 ? slot : 2), i0 = gate * S::NR + 4 * tl + rr;
      const bool hid_row = S::GATES == 4 || slot != 3, in_row = S::GATES == 4 || slot != 2;
      const int jj = j & 3;
      if (kb < S::KB2H) {               // (chain-row block mt, rank pair ap) of the hidden matrix — and, riding, of the input matrix
        const int mt = kb / S::AP, a = 2 * (kb % S::AP) + (j >> 2);
        if (S::RIDE && mt == S::MT0 - 1 && g >= 2) {
          const int j0 = 4 * (g - 2) + jj;
          if (j0 < S::J0I && a < ri && in_row) v = w2_gh(si, pk_in, i0, j0, a);
        } else {
          const int j0 = 16 * mt + 4 * g + jj;
          if (j0 < S::J0 && a < rh && hid_row) v = w2_gh(sh, pk_hid, i0, j0, a);
        }
      } else {                          // the input matrix's own k-blocks (one per rank pair): chain rows 4 g + jj
        const int a = 2 * (kb - S::KB2H) + (j >> 2), j0 = 4 * g + jj;
        if (j0 < S::J0I && a < ri && in_row) v = w2_gh(si, pk_in, i0, j0, a);
      }
      v *= ldexpf(1.f, 14 - egh);
    }
    _Float16 p0, p1;
    split2h(v, p0, p1);
    dst[lane * 8 + j] = p0;
    dst[512 + lane * 8 + j] = p1;
  }
}

struct W2Args {
  const float* x; const float* h0; const float* c0;
  const float* bias_in; const float* bias_hid;
  const int* hdr; const _Float16* frag;
  float* out; float* hT; float* cT; float* reserve;
  int B, T;
};

__device__ __forceinline__ xh8 w2_ld8(const _Float16* p) { return *reinterpret_cast<const xh8*>(p); }
__device__ __forceinline__ f32x4 w2_mma3(const xh8 a0, const xh8 a1, const xh8 b0, const xh8 b1, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc, 0, 0, 0);
  return acc;
}
// eight fp32 values -> the two fp16 pieces of a k-packed operand
__device__ __forceinline__ void w2_split8(const float (&v)[8], xh8& p0, xh8& p1) {
  unsigned a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split_pair_h(v[2 * i], v[2 * i + 1], a[i], b[i]);
  p0 = __builtin_bit_cast(xh8, u32x4{a[0], a[1], a[2], a[3]});
  p1 = __builtin_bit_cast(xh8, u32x4{b[0], b[1], b[2], b[3]});
}

template <class S>
__global__ void __launch_bounds__(256) k_lstm_fwd_w2(W2Args g) {
  constexpr int R = S::R;
  __shared__ __attribute__((aligned(16))) unsigned char smem[S::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int b = blockIdx.x, T = g.T;
  _Float16* himg = reinterpret_cast<_Float16*>(smem + S::L_H);       // [par][piece][HROWS][HS]
  _Float16* ximg = reinterpret_cast<_Float16*>(smem + S::L_X);       // [par][piece][16][XS]
  int* xexp = reinterpret_cast<int*>(smem + S::L_E);
  float* red = reinterpret_cast<float*>(smem + S::L_RED);
  constexpr int HP = S::HROWS * S::HS, XP = 16 * S::XS;               // plane sizes (halves)
  for (int i = tid; i < S::L_E / 16; i += 256) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragments: stage 1 pinned in registers; stage 2 in registers or in LDS in fragment order ----
  xh8 gt[R][S::KB1][2], gti[R][2], gh[S::GHL ? 1 : S::MT2][S::GHL ? 1 : S::KB2][2];
  const xh8* fr = reinterpret_cast<const xh8*>(g.frag);
#pragma unroll
  for (int a = 0; a < R; ++a)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int kb = 0; kb < S::KB1; ++kb) {
        gt[a][kb][p] = fr[((size_t)((R * wave + a) * S::KB1 + kb) * 2 + p) * 64 + lane];
        asm volatile("" : "+v"(gt[a][kb][p]));
      }
      gti[a][p] = fr[((size_t)(S::T_GTI + R * wave + a) * 2 + p) * 64 + lane];
      asm volatile("" : "+v"(gti[a][p]));
    }
  if constexpr (S::GHL) {
    for (int i = tid; i < S::GH_BYTES / 16; i += 256)
      reinterpret_cast<f32x4*>(smem + S::L_GH)[i] = reinterpret_cast<const f32x4*>(fr + (size_t)S::T_GH * 2 * 64)[i];
  } else {
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl)
#pragma unroll
      for (int kb = 0; kb < S::KB2; ++kb)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          gh[tl][kb][p] = fr[((size_t)(S::T_GH + S::KB2 * tl + kb) * 2 + p) * 64 + lane];
          asm volatile("" : "+v"(gh[tl][kb][p]));
        }
  }
  const xh8* ghl = reinterpret_cast<const xh8*>(smem + S::L_GH);
  const int egt = g.hdr[W2_EGT], ec1 = g.hdr[W2_EC1], egti = g.hdr[W2_EGTI], eci = g.hdr[W2_ECI], egh = g.hdr[W2_EGH];

  // ---- the lane's three hidden units (one per stage-2 row tile), their biases and state ----
  int unit[S::MT2];
  f32x4 bz[S::MT2];
  float cst[S::MT2], hval[S::MT2];
#pragma unroll
  for (int tl = 0; tl < S::MT2; ++tl) {
    unit[tl] = (4 * tl + q) * 64 + 16 * wave + n;
#pragma unroll
    for (int gate = 0; gate < 4; ++gate) {
      float v = 0.f;
      if (g.bias_in) v += g.bias_in[gate * S::H + unit[tl]];
      if (g.bias_hid) v += g.bias_hid[gate * S::H + unit[tl]];
      // the gates' exponent factors ride on the bias and on the step's un-scale: sigmoid(z) = 1 / (1 + 2^(-log2e z)),
      // tanh(z) = 1 - 2 / (1 + 2^(2 log2e z)) — the accumulator x factor + bias x factor IS the v_exp_f32 argument
      bz[tl][gate] = v * (gate == 2 ? 2.8853900817779268f : -1.4426950408889634f);
    }
    cst[tl] = g.c0 ? g.c0[(size_t)b * S::H + unit[tl]] : 0.f;
    hval[tl] = g.h0 ? g.h0[(size_t)b * S::H + unit[tl]] : 0.f;
  }
  // a caller's h_0 may exceed 1: its exponent per sample (0 when it does not)
  int e0 = 0;
  if (g.h0) {
    float m = fmaxf(fmaxf(fabsf(hval[0]), fabsf(hval[1])), fabsf(hval[2]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    e0 = w2_expo(m);
    e0 = e0 < 0 ? 0 : e0;
  } else {
    __syncthreads();
  }
  // unit u = j0 * J1 + j1 of the state image
  int hoff[S::MT2];
#pragma unroll
  for (int tl = 0; tl < S::MT2; ++tl) hoff[tl] = (unit[tl] / S::J1) * S::HS + (unit[tl] % S::J1);
  auto put_h = [&](int par, float scale) {
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl) {
      _Float16 p0, p1;
      split2h(hval[tl] * scale, p0, p1);
      himg[(par * 2 + 0) * HP + hoff[tl]] = p0;
      himg[(par * 2 + 1) * HP + hoff[tl]] = p1;
    }
  };
  // x_t: wave 0, lanes < INP — row XR0 + j0' of the input image, column j1'
  const int xo = (S::XR0 + lane / S::J1I) * S::XS + (lane % S::J1I);
  const float* xrow = g.x + (size_t)b * T * S::INP;
  auto put_x = [&](int par, float xv) {
    float m = lane < S::INP ? fabsf(xv) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const int ex = w2_expo(m);
    if (lane < S::INP) {
      _Float16 p0, p1;
      split2h(xv * ldexpf(1.f, 13 - ex), p0, p1);
      ximg[(par * 2 + 0) * XP + xo] = p0;
      ximg[(par * 2 + 1) * XP + xo] = p1;
    }
    if (lane == 0) xexp[par] = ex;
  };
  put_h(0, ldexpf(1.f, 13 - e0));
  float xnext = 0.f;
  if (wave == 0) {
    put_x(0, (lane < S::INP && T > 0) ? xrow[lane] : 0.f);
    if (lane < S::INP && T > 1) xnext = xrow[S::INP + lane];
  }
  __syncthreads();

  const size_t rrows = (size_t)g.B * T;
  for (int t = 0; t < T; ++t) {
    const int par = t & 1;
    const _Float16* hp = himg + par * 2 * HP;
    const _Float16* xp = ximg + par * 2 * XP;
    // A operands of stage 1: rows 16 mt + n of the state image, row n of the input image, eight k from 8 q per k-block
    xh8 ah[S::MT0][S::KB1][2], ax[2];
#pragma unroll
    for (int mt = 0; mt < S::MT0; ++mt)
#pragma unroll
      for (int kb = 0; kb < S::KB1; ++kb)
#pragma unroll
        for (int p = 0; p < 2; ++p) ah[mt][kb][p] = w2_ld8(hp + p * HP + (16 * mt + n) * S::HS + 32 * kb + 8 * q);
#pragma unroll
    for (int p = 0; p < 2; ++p) ax[p] = w2_ld8(xp + p * XP + n * S::XS + 8 * q);
    const int ex = xexp[par];
    // ---- stage 1 ----
    f32x4 d[R][S::MT0], di[R];
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < R; ++a) {
#pragma unroll
      for (int mt = 0; mt < S::MT0; ++mt) {
        f32x4 acc = z4;
#pragma unroll
        for (int kb = 0; kb < S::KB1; ++kb) acc = w2_mma3(ah[mt][kb][0], ah[mt][kb][1], gt[a][kb][0], gt[a][kb][1], acc);
        d[a][mt] = acc;
      }
      di[a] = w2_mma3(ax[0], ax[1], gti[a][0], gti[a][1], z4);
    }
    // ---- hand-off: the lane's own (a, jj) values are its k elements of stage 2; per-step exponents ----
    const int eh0 = t == 0 ? e0 : 0;
    const int ech = ec1 + eh0, ecx = eci + ex;
    const int ec = ech > ecx ? ech : ecx;
    const int sh_ = egt - ec - 13 + eh0, si_ = egti + ex - ec - 13;
    xh8 bop[S::KB2][2];
#pragma unroll
    for (int kb = 0; kb < S::KB2; ++kb) {
      float v[8];
#pragma unroll
      for (int ai = 0; ai < 2; ++ai)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          if (kb < S::KB2H) {
            const int mt = kb / S::AP, a = 2 * (kb % S::AP) + ai;
            const float hv = ldexpf(d[a][mt][jj], sh_);
            if (S::RIDE && mt == S::MT0 - 1) v[4 * ai + jj] = q < 2 ? hv : ldexpf(di[a][jj], si_);
            else v[4 * ai + jj] = hv;
          } else {
            v[4 * ai + jj] = ldexpf(di[2 * (kb - S::KB2H) + ai][jj], si_);
          }
        }
      w2_split8(v, bop[kb][0], bop[kb][1]);
    }
    // ---- stage 2 + gates ----
    const float zs = ldexpf(1.f, egh + ec - 28);
    const f32x4 zsv = f32x4{-1.4426950408889634f * zs, -1.4426950408889634f * zs, 2.8853900817779268f * zs, -1.4426950408889634f * zs};
    const size_t bt = (size_t)b * T + t;
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl) {
      f32x4 acc = z4;
#pragma unroll
      for (int kb = 0; kb < S::KB2; ++kb) {
        xh8 a0, a1;
        if constexpr (S::GHL) {
          a0 = ghl[((size_t)(S::KB2 * tl + kb) * 2 + 0) * 64 + lane];
          a1 = ghl[((size_t)(S::KB2 * tl + kb) * 2 + 1) * 64 + lane];
        } else {
          a0 = gh[tl][kb][0]; a1 = gh[tl][kb][1];
        }
        acc = w2_mma3(a0, a1, bop[kb][0], bop[kb][1], acc);
      }
      const f32x4 z = acc * zsv + bz[tl];
      const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[0])), fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[1]));
      const float gg = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[2]));
      const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[3]));
      const float cy = fg * cst[tl] + ig * gg;
      const float hy = og * ftanh(cy);
      cst[tl] = cy;
      hval[tl] = hy;
      if (g.out) g.out[bt * S::H + unit[tl]] = hy;
      if (g.reserve) {
        *reinterpret_cast<f32x4*>(g.reserve + res_gate(bt, S::H, unit[tl])) = f32x4{ig, gg, fg, og};
        g.reserve[res_cell(rrows, bt, S::H, unit[tl])] = cy;
      }
    }
    put_h(par ^ 1, 8192.f);
    if (wave == 0) {
      put_x(par ^ 1, xnext);
      if (lane < S::INP && t + 2 < T) xnext = xrow[(size_t)(t + 2) * S::INP + lane];
    }
    lds_barrier();
  }
#pragma unroll
  for (int tl = 0; tl < S::MT2; ++tl) {
    if (g.hT) g.hT[(size_t)b * S::H + unit[tl]] = hval[tl];
    if (g.cT) g.cT[(size_t)b * S::H + unit[tl]] = cst[tl];
  }
}

// ---- TT-GRU forward (gru.py:33-44) on the same structure: three waves per sample, four accumulator rows per unit ---------------------
// lane (n = i1 in the wave's slice, q), row tile tl: unit u = (4 tl + q) * 48 + 16 wave + n; accumulators: r | z | W_hn h | W_in x of
// the n gate.  A caller's h_0 outside (-1, 1) does not decay in one step as an LSTM's does (h' = (1 - z) n + z h): the state image
// is written under the exponent of the LAST step's exact maximum (a bound on this step's: |h'| <= max(1, |h|)), which every wave
// leaves in LDS next to the image.
template <class S>
__global__ void __launch_bounds__(64 * S::NWV) k_gru_fwd_w2(W2Args g) {
  constexpr int R = S::R, NTH = 64 * S::NWV;
  static_assert(S::GATES == 3, "TT-GRU configuration");
  __shared__ __attribute__((aligned(16))) unsigned char smem[S::LDS];
  __shared__ float hmax[2][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int b = blockIdx.x, T = g.T;
  _Float16* himg = reinterpret_cast<_Float16*>(smem + S::L_H);       // [par][piece][HROWS][HS]
  _Float16* ximg = reinterpret_cast<_Float16*>(smem + S::L_X);       // [par][piece][16][XS]
  int* xexp = reinterpret_cast<int*>(smem + S::L_E);
  float* red = reinterpret_cast<float*>(smem + S::L_RED);
  constexpr int HP = S::HROWS * S::HS, XP = 16 * S::XS;
  for (int i = tid; i < S::L_E / 16; i += NTH) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  xh8 gt[R][S::KB1][2], gti[R][2];
  const xh8* fr = reinterpret_cast<const xh8*>(g.frag);
#pragma unroll
  for (int a = 0; a < R; ++a)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int kb = 0; kb < S::KB1; ++kb) {
        gt[a][kb][p] = fr[((size_t)((R * wave + a) * S::KB1 + kb) * 2 + p) * 64 + lane];
        asm volatile("" : "+v"(gt[a][kb][p]));
      }
      gti[a][p] = fr[((size_t)(S::T_GTI + R * wave + a) * 2 + p) * 64 + lane];
      asm volatile("" : "+v"(gti[a][p]));
    }
  xh8 gh[S::GHL ? 1 : S::MT2][S::GHL ? 1 : S::KB2][2];
  if constexpr (S::GHL) {
    for (int i = tid; i < S::GH_BYTES / 16; i += NTH)
      reinterpret_cast<f32x4*>(smem + S::L_GH)[i] = reinterpret_cast<const f32x4*>(fr + (size_t)S::T_GH * 2 * 64)[i];
  } else {
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl)
#pragma unroll
      for (int kb = 0; kb < S::KB2; ++kb)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          gh[tl][kb][p] = fr[((size_t)(S::T_GH + S::KB2 * tl + kb) * 2 + p) * 64 + lane];
          asm volatile("" : "+v"(gh[tl][kb][p]));
        }
  }
  const xh8* ghl = reinterpret_cast<const xh8*>(smem + S::L_GH);
  const int egt = g.hdr[W2_EGT], ec1 = g.hdr[W2_EC1], egti = g.hdr[W2_EGTI], eci = g.hdr[W2_ECI], egh = g.hdr[W2_EGH];

  int unit[S::MT2];
  f32x4 bz[S::MT2];
  float hval[S::MT2];
#pragma unroll
  for (int tl = 0; tl < S::MT2; ++tl) {
    unit[tl] = (4 * tl + q) * S::I1 + 16 * wave + n;
    float br = 0.f, bzz = 0.f, bhn = 0.f, bin = 0.f;
    if (g.bias_in) { br += g.bias_in[unit[tl]]; bzz += g.bias_in[S::H + unit[tl]]; bin = g.bias_in[2 * S::H + unit[tl]]; }
    if (g.bias_hid) { br += g.bias_hid[unit[tl]]; bzz += g.bias_hid[S::H + unit[tl]]; bhn = g.bias_hid[2 * S::H + unit[tl]]; }
    bz[tl] = f32x4{br * -1.4426950408889634f, bzz * -1.4426950408889634f, bhn, bin};
    hval[tl] = g.h0 ? g.h0[(size_t)b * S::H + unit[tl]] : 0.f;
  }
  // the exponent the state image is written under: 0 while |h| < 1
  const bool track = g.h0 != nullptr;
  int e_img = 0;
  {
    float m = 0.f;
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl) m = fmaxf(m, fabsf(hval[tl]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();
    if (lane == 0) { red[wave] = m; hmax[0][wave] = m; }
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < S::NWV; ++w) m = fmaxf(m, red[w]);
    e_img = track ? w2_expo(m) : 0;
    e_img = e_img < 0 ? 0 : e_img;
  }
  int hoff[S::MT2];
#pragma unroll
  for (int tl = 0; tl < S::MT2; ++tl) hoff[tl] = (unit[tl] / S::J1) * S::HS + (unit[tl] % S::J1);
  auto put_h = [&](int par, float scale) {
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl) {
      _Float16 p0, p1;
      split2h(hval[tl] * scale, p0, p1);
      himg[(par * 2 + 0) * HP + hoff[tl]] = p0;
      himg[(par * 2 + 1) * HP + hoff[tl]] = p1;
    }
  };
  const int xo = (S::XR0 + lane / S::J1I) * S::XS + (lane % S::J1I);
  const float* xrow = g.x + (size_t)b * T * S::INP;
  auto put_x = [&](int par, float xv) {
    float m = lane < S::INP ? fabsf(xv) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const int ex = w2_expo(m);
    if (lane < S::INP) {
      _Float16 p0, p1;
      split2h(xv * ldexpf(1.f, 13 - ex), p0, p1);
      ximg[(par * 2 + 0) * XP + xo] = p0;
      ximg[(par * 2 + 1) * XP + xo] = p1;
    }
    if (lane == 0) xexp[par] = ex;
  };
  put_h(0, ldexpf(1.f, 13 - e_img));
  float xnext = 0.f;
  if (wave == 0) {
    put_x(0, (lane < S::INP && T > 0) ? xrow[lane] : 0.f);
    if (lane < S::INP && T > 1) xnext = xrow[S::INP + lane];
  }
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const int par = t & 1;
    const _Float16* hp = himg + par * 2 * HP;
    const _Float16* xp = ximg + par * 2 * XP;
    xh8 ah[S::MT0][S::KB1][2], ax[2];
#pragma unroll
    for (int mt = 0; mt < S::MT0; ++mt)
#pragma unroll
      for (int kb = 0; kb < S::KB1; ++kb)
#pragma unroll
        for (int p = 0; p < 2; ++p) ah[mt][kb][p] = w2_ld8(hp + p * HP + (16 * mt + n) * S::HS + 32 * kb + 8 * q);
#pragma unroll
    for (int p = 0; p < 2; ++p) ax[p] = w2_ld8(xp + p * XP + n * S::XS + 8 * q);
    const int ex = xexp[par];
    // the exponent of the image being read, and the one the next image is written under (the exact maximum of |h_{t-1}|)
    const int eh0 = e_img;
    int e_wr = 0;
    if (track) {
      float m = hmax[par][0];
#pragma unroll
      for (int w = 1; w < S::NWV; ++w) m = fmaxf(m, hmax[par][w]);
      e_wr = w2_expo(m);
      e_wr = e_wr < 0 ? 0 : e_wr;
    }
    // ---- stage 1 ----
    f32x4 d[R][S::MT0], di[R];
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < R; ++a) {
#pragma unroll
      for (int mt = 0; mt < S::MT0; ++mt) {
        f32x4 acc = z4;
#pragma unroll
        for (int kb = 0; kb < S::KB1; ++kb) acc = w2_mma3(ah[mt][kb][0], ah[mt][kb][1], gt[a][kb][0], gt[a][kb][1], acc);
        d[a][mt] = acc;
      }
      di[a] = w2_mma3(ax[0], ax[1], gti[a][0], gti[a][1], z4);
    }
    // ---- hand-off (k_lstm_fwd_w2's) ----
    const int ech = ec1 + eh0, ecx = eci + ex;
    const int ec = ech > ecx ? ech : ecx;
    const int sh_ = egt - ec - 13 + eh0, si_ = egti + ex - ec - 13;
    xh8 bop[S::KB2][2];
#pragma unroll
    for (int kb = 0; kb < S::KB2; ++kb) {
      float v[8];
#pragma unroll
      for (int ai = 0; ai < 2; ++ai)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          if (kb < S::KB2H) {
            const int mt = kb / S::AP, a = 2 * (kb % S::AP) + ai;
            const float hv = ldexpf(d[a][mt][jj], sh_);
            if (S::RIDE && mt == S::MT0 - 1) v[4 * ai + jj] = q < 2 ? hv : ldexpf(di[a][jj], si_);
            else v[4 * ai + jj] = hv;
          } else {
            v[4 * ai + jj] = ldexpf(di[2 * (kb - S::KB2H) + ai][jj], si_);
          }
        }
      w2_split8(v, bop[kb][0], bop[kb][1]);
    }
    // ---- stage 2 + gates (gru.py:38-44) ----
    const float zs = ldexpf(1.f, egh + ec - 28);
    const f32x4 zsv = f32x4{-1.4426950408889634f * zs, -1.4426950408889634f * zs, zs, zs};
    const size_t bt = (size_t)b * T + t;
    float wm = 0.f;
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl) {
      f32x4 acc = z4;
#pragma unroll
      for (int kb = 0; kb < S::KB2; ++kb) {
        xh8 a0, a1;
        if constexpr (S::GHL) {
          a0 = ghl[((size_t)(S::KB2 * tl + kb) * 2 + 0) * 64 + lane];
          a1 = ghl[((size_t)(S::KB2 * tl + kb) * 2 + 1) * 64 + lane];
        } else {
          a0 = gh[tl][kb][0]; a1 = gh[tl][kb][1];
        }
        acc = w2_mma3(a0, a1, bop[kb][0], bop[kb][1], acc);
      }
      const f32x4 z = acc * zsv + bz[tl];
      const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[0])), zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[1]));
      const float hn = z[2];                                     // W_hn h + b_hn
      const float ng = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * (z[3] + rg * hn)));
      const float hy = (1.0f - zg) * ng + zg * hval[tl];
      hval[tl] = hy;
      wm = fmaxf(wm, fabsf(hy));
      if (g.out) g.out[bt * S::H + unit[tl]] = hy;
      if (g.reserve) *reinterpret_cast<f32x4*>(g.reserve + res_gate(bt, S::H, unit[tl])) = f32x4{rg, zg, ng, hn};
    }
    put_h(par ^ 1, ldexpf(1.f, 13 - e_wr));
    if (track) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
      if (lane == 0) hmax[par ^ 1][wave] = wm;
    }
    e_img = e_wr;
    if (wave == 0) {
      put_x(par ^ 1, xnext);
      if (lane < S::INP && t + 2 < T) xnext = xrow[(size_t)(t + 2) * S::INP + lane];
    }
    lds_barrier();
  }
#pragma unroll
  for (int tl = 0; tl < S::MT2; ++tl)
    if (g.hT) g.hT[(size_t)b * S::H + unit[tl]] = hval[tl];
}

// ==================================================================================================================================
// Reverse-time kernel (BPTT of lstm.py:23-32 through the hidden chain) with the same ownership: wave w owns the slice i1 in
// [16 w, 16 w + 16) of the gate gradients — which are the gradients of ITS OWN units (unit u = r * 64 + i1), so the first transposed
// stage is wave-local:
//   gates     dh_t = d_out_t + (sum of the four waves' partial dh of step t + 1), dc_t in a register; the forward record gives
//             i, g, f, o, c_t, c_{t-1}: the four pre-activation gradients of the lane's three units; written to d_gates;
//   T2'       dC1^T[i1][(a, j0)] = sum_i0 dz[i0][i1] Gh[i0][j0][a]: the lane's own twelve values are the A operand (k runs over
//             (row tile, gate) exactly as the lane holds them), scaled under the WAVE's maximum (no barrier) — 24 MFMAs;
//   hand-off  none: a result tile has (a, j0) on the lane and four consecutive i1 in the registers — the B operand of a
//             16x16x16 MFMA that sums over i1;
//   T1        dh^T[j1][j0] (partial: this wave's sixteen i1) = sum_{a, i1} Gt[i1][j1][a] dC1[(a, j0)][i1] — 24 16x16x16 MFMAs;
//   exchange  the partial dh of the wave as fp32 into LDS (parity buffers), ONE barrier per step; the next step's lanes add the
//             four partials of their own units.
// By-products: the column maxima of the gate gradients (ttrnn_rnn_backward_ex: stats rows 0 / 1) — the chain weight-gradient
// kernel's bound on dy without a pass over the gigabyte of them.
enum { W2B_EGH = 0, W2B_EGT = 1, W2B_EL1 = 2 };
typedef _Float16 w2_xh4 __attribute__((ext_vector_type(4)));

template <class S>
__global__ void __launch_bounds__(256) k_w2b_prep(TtShape sh, const float* __restrict__ pk_hid, int* __restrict__ hdr,
                                                  _Float16* __restrict__ ghf, _Float16* __restrict__ gtf) {
  constexpr int R = S::R;
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int rh = sh.R[sh.d / 2];
  // maxima of the hidden matrix's (merged) cores; workgroup 0: the L1 bound of dC1 (rows (j0, a) of Gh^T over i0)
  float mgt = 0.f, mh = 0.f, l1h = 0.f;
  for (int e = tid; e < S::I1 * S::J1 * rh; e += 256) {
    const int a = e % rh, j = (e / rh) % S::J1, i = e / (rh * S::J1);
    mgt = fmaxf(mgt, fabsf(w2_gt(sh, pk_hid, i, j, a)));
  }
  for (int e = tid; e < S::I0 * S::J0 * rh; e += 256) {
    const int a = e % rh, j = (e / rh) % S::J0, i = e / (rh * S::J0);
    mh = fmaxf(mh, fabsf(w2_gh(sh, pk_hid, i, j, a)));
  }
  if (blockIdx.x == 0 && tid < S::J0 * rh) {
    const int j0 = tid / rh, a = tid % rh;
    float s0 = 0.f;
    for (int i = 0; i < S::I0; ++i) s0 += fabsf(w2_gh(sh, pk_hid, i, j0, a));
    l1h = s0;
  }
  mgt = w2_block_max(mgt, red); mh = w2_block_max(mh, red);
  if (blockIdx.x == 0) l1h = w2_block_max(l1h, red);
  const int egt = w2_expo(mgt), egh = w2_expo(mh);
  if (blockIdx.x == 0 && tid == 0) { hdr[W2B_EGH] = egh; hdr[W2B_EGT] = egt; hdr[W2B_EL1] = w2_expo(l1h); }
  const int tile = blockIdx.x;
  if (tile < S::NT2 * 2) {   // T2' B operand, tile (nt = (a, jt), kb): B[k][col n]: column (a, j0 = 16 jt + n); k = (kb, g, e)
    const int nt = tile >> 1, kb = tile & 1;
    _Float16* dst = ghf + (size_t)tile * 1024;
    for (int e = tid; e < 512; e += 256) {
      const int j = e & 7, lane = e >> 3, n = lane & 15, g = lane >> 4;
      const int a = nt / S::MT0, j0 = 16 * (nt % S::MT0) + n;
      // k element j of k-group g: kb 0: row tile tl = j >> 2 (0, 1), gate = j & 3; kb 1: tl = 2, gate = j (j < 4), zero beyond
      // (TT-GRU: four row tiles x four slots fill both k-blocks; slot 3 — the input part of n — is no row of the hidden chain: zero)
      // (three row tiles: the LSTM's packing — k-block 1 = row tile 2 in its first four slots; four row tiles: both k-blocks full)
      const int tl = S::MT2 == 4 ? 2 * kb + (j >> 2) : (kb == 0 ? (j >> 2) : 2);
      const int gate = S::MT2 == 4 ? (j & 3) : (kb == 0 ? (j & 3) : j);
      const bool live = (S::MT2 == 4 || kb == 0 || j < 4) && (S::GATES == 4 || gate < 3);
      float v = 0.f;
      if (a < rh && j0 < S::J0 && live) v = w2_gh(sh, pk_hid, gate * S::NR + 4 * tl + g, j0, a) * ldexpf(1.f, 14 - egh);
      _Float16 p0, p1;
      split2h(v, p0, p1);
      dst[lane * 8 + j] = p0;
      dst[512 + lane * 8 + j] = p1;
    }
  } else if (tile < S::NT2 * 2 + S::NWV * R * S::M1T) {   // T1 A operand (16x16x16), tile (w, a, mt): A[row j1 = 16 mt + n][k = i1 - 16 w = 4 g + j]
    const int tt = tile - S::NT2 * 2, mt = tt % S::M1T, a = (tt / S::M1T) % R, w = tt / (S::M1T * R);
    _Float16* dst = gtf + (size_t)tt * 512;
    for (int e = tid; e < 256; e += 256) {
      const int j = e & 3, lane = e >> 2, n = lane & 15, g = lane >> 4;
      const float v = a < rh ? w2_gt(sh, pk_hid, 16 * w + 4 * g + j, 16 * mt + n, a) * ldexpf(1.f, 14 - egt) : 0.f;
      _Float16 p0, p1;
      split2h(v, p0, p1);
      dst[lane * 4 + j] = p0;
      dst[256 + lane * 4 + j] = p1;
    }
  }
}

struct W2BArgs {
  const float* c0; const float* reserve;
  const float* out; const float* h0;      // TT-GRU: the layer's outputs (h_{t-1} of the gate gradients) and the initial state
  const float* d_out; const float* d_hT; const float* d_cT;
  const int* hdr; const _Float16* ghf; const _Float16* gtf;
  float* dg; float* dg_hid;               // d_gates_in; TT-GRU: d_gates_hid (its n block carries r)
  float* d_h0; float* d_c0;
  unsigned* colmax;          // stats rows 0 and 1 ([2][4H] bit patterns, zeroed by the launcher) or NULL
  int B, T;
};

__device__ __forceinline__ f32x4 w2_mma3_16(const w2_xh4 a0, const w2_xh4 a1, const w2_xh4 b0, const w2_xh4 b1, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, acc, 0, 0, 0);
  return acc;
}

template <class S>
__global__ void __launch_bounds__(256) k_lstm_bwd_w2(W2BArgs g) {
  constexpr int R = S::R;
  extern __shared__ __attribute__((aligned(16))) unsigned char w2b_smem[];      // S::BLDS bytes (above the static limit for the rank-4 instantiations)
  unsigned char* smem = w2b_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int b = blockIdx.x, T = g.T;
  float* part = reinterpret_cast<float*>(smem);                 // [parity][wave][PART]
  for (int i = tid; i < 2 * S::NWV * S::PART / 4; i += 256) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragments ----
  xh8 gh[S::GHLB ? 1 : S::NT2][2][2];
  w2_xh4 gta[R][S::M1T][2];
  {
    const xh8* f8 = reinterpret_cast<const xh8*>(g.ghf);
    if constexpr (S::GHLB) {
      for (int i = tid; i < (int)(S::BGH_BYTES / 16); i += 256)
        reinterpret_cast<f32x4*>(smem + S::BL_GH)[i] = reinterpret_cast<const f32x4*>(f8)[i];
    } else {
#pragma unroll
      for (int nt = 0; nt < S::NT2; ++nt)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            gh[nt][kb][p] = f8[((size_t)(2 * nt + kb) * 2 + p) * 64 + lane];
            asm volatile("" : "+v"(gh[nt][kb][p]));
          }
    }
    const w2_xh4* f4 = reinterpret_cast<const w2_xh4*>(g.gtf);
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
      for (int mt = 0; mt < S::M1T; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          gta[a][mt][p] = f4[((size_t)((wave * R + a) * S::M1T + mt) * 2 + p) * 64 + lane];
          asm volatile("" : "+v"(gta[a][mt][p]));
        }
  }
  const xh8* ghl = reinterpret_cast<const xh8*>(smem + S::BL_GH);
  const int egh = g.hdr[W2B_EGH], egt = g.hdr[W2B_EGT], el1 = g.hdr[W2B_EL1];
  const int shd = egh - el1 - 13;                   // T2' accumulator -> dC1 pieces (the wave's step exponent cancels)

  int unit[S::MT2];
#pragma unroll
  for (int tl = 0; tl < S::MT2; ++tl) unit[tl] = (4 * tl + q) * 64 + 16 * wave + n;
  const size_t rrows = (size_t)g.B * T;
  const size_t bt0 = (size_t)b * T;
  // ---- records: gates two steps deep, cell states three; d_out two ----
  f32x4 G0[S::MT2], G1[S::MT2];
  float C0[S::MT2], C1[S::MT2], C2[S::MT2], D0[S::MT2], D1[S::MT2];
  auto ld_gates = [&](int t, int tl) -> f32x4 {
    return t >= 0 ? *reinterpret_cast<const f32x4*>(g.reserve + res_gate(bt0 + t, S::H, unit[tl])) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto ld_cell = [&](int t, int tl) -> float {
    if (t >= 0) return g.reserve[res_cell(rrows, bt0 + t, S::H, unit[tl])];
    return (t == -1 && g.c0) ? g.c0[(size_t)b * S::H + unit[tl]] : 0.f;
  };
  auto ld_dout = [&](int t, int tl) -> float {
    return (t >= 0 && g.d_out) ? g.d_out[(bt0 + t) * S::H + unit[tl]] : 0.f;
  };
  float dc[S::MT2], cmx[S::MT2][4];
#pragma unroll
  for (int tl = 0; tl < S::MT2; ++tl) {
    G0[tl] = ld_gates(T - 1, tl); G1[tl] = ld_gates(T - 2, tl);
    C0[tl] = ld_cell(T - 1, tl); C1[tl] = ld_cell(T - 2, tl); C2[tl] = ld_cell(T - 3, tl);
    D0[tl] = ld_dout(T - 1, tl); D1[tl] = ld_dout(T - 2, tl);
    dc[tl] = g.d_cT ? g.d_cT[(size_t)b * S::H + unit[tl]] : 0.f;
    if (g.d_hT) D0[tl] += g.d_hT[(size_t)b * S::H + unit[tl]];
#pragma unroll
    for (int k = 0; k < 4; ++k) cmx[tl][k] = 0.f;
  }
  __syncthreads();

  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = T - 1; t >= 0; --t) {
    const int par = t & 1;
    // the four waves' partial dh of step t + 1 (zero in the first iteration), this lane's units
    const float* pr = part + par * S::NWV * S::PART;
    float dz[S::MT2][4];
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl) {
      const float dh = D0[tl] + ((pr[unit[tl]] + pr[S::PART + unit[tl]]) + (pr[2 * S::PART + unit[tl]] + pr[3 * S::PART + unit[tl]]));
      const float ig = G0[tl][0], gg = G0[tl][1], fg = G0[tl][2], og = G0[tl][3];
      const float tc = ftanh(C0[tl]);
      const float dov = dh * tc;
      const float dct = dc[tl] + dh * og * (1.f - tc * tc);
      dz[tl][0] = dct * gg * ig * (1.f - ig);              // i
      dz[tl][1] = dct * C1[tl] * fg * (1.f - fg);          // f   (C1 = c_{t-1})
      dz[tl][2] = dct * ig * (1.f - gg * gg);              // g
      dz[tl][3] = dov * og * (1.f - og);                   // o
      dc[tl] = dct * fg;
      float* dgp = g.dg + (bt0 + t) * (4 * S::H) + unit[tl];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dgp[k * S::H] = dz[tl][k];
        cmx[tl][k] = fmaxf(cmx[tl][k], fabsf(dz[tl][k]));
      }
      // shift the record pipeline and issue the loads of step t - 2 (gates) / t - 3 (cell)
      G0[tl] = G1[tl]; C0[tl] = C1[tl]; C1[tl] = C2[tl]; D0[tl] = D1[tl];
      G1[tl] = ld_gates(t - 2, tl);
      C2[tl] = ld_cell(t - 3, tl);
      D1[tl] = ld_dout(t - 2, tl);
    }
    // ---- the wave's maximum -> exponent; A operand of T2': k-block 0 = row tiles 0, 1 x gates, k-block 1 = row tile 2 x gates ----
    float m = 0.f;
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl)
#pragma unroll
      for (int k = 0; k < 4; ++k) m = fmaxf(m, fabsf(dz[tl][k]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const int ed = w2_expo(m);
    const float sd = ldexpf(1.f, 13 - ed);
    float v0[8], v1[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v0[k] = dz[0][k] * sd; v0[4 + k] = dz[1][k] * sd; v1[k] = dz[2][k] * sd; v1[4 + k] = 0.f; }
    xh8 a0[2], a1[2];
    w2_split8(v0, a0[0], a0[1]);
    w2_split8(v1, a1[0], a1[1]);
    // ---- T2' + T1 ----
    f32x4 zt[S::M1T][S::MT0];           // [j1 tile mt][j0 tile jt]
#pragma unroll
    for (int mt = 0; mt < S::M1T; ++mt)
#pragma unroll
      for (int jt = 0; jt < S::MT0; ++jt) zt[mt][jt] = z4;
#pragma unroll
    for (int nt = 0; nt < S::NT2; ++nt) {
      xh8 b00, b01, b10, b11;
      if constexpr (S::GHLB) {
        b00 = ghl[((size_t)(2 * nt + 0) * 2 + 0) * 64 + lane]; b01 = ghl[((size_t)(2 * nt + 0) * 2 + 1) * 64 + lane];
        b10 = ghl[((size_t)(2 * nt + 1) * 2 + 0) * 64 + lane]; b11 = ghl[((size_t)(2 * nt + 1) * 2 + 1) * 64 + lane];
      } else {
        b00 = gh[nt][0][0]; b01 = gh[nt][0][1]; b10 = gh[nt][1][0]; b11 = gh[nt][1][1];
      }
      f32x4 d = w2_mma3(a0[0], a0[1], b00, b01, z4);
      d = w2_mma3(a1[0], a1[1], b10, b11, d);
      // lane (column (a, j0), q) holds i1 = 4 q + jj: the B operand of the 16x16x16 product over i1
      unsigned p0a, p1a, p0b, p1b;
      split_pair_h(ldexpf(d[0], shd), ldexpf(d[1], shd), p0a, p1a);
      split_pair_h(ldexpf(d[2], shd), ldexpf(d[3], shd), p0b, p1b);
      const w2_xh4 b0 = __builtin_bit_cast(w2_xh4, u32x2{p0a, p0b}), b1 = __builtin_bit_cast(w2_xh4, u32x2{p1a, p1b});
      const int a = nt / S::MT0, jt = nt % S::MT0;
#pragma unroll
      for (int mt = 0; mt < S::M1T; ++mt) zt[mt][jt] = w2_mma3_16(gta[a][mt][0], gta[a][mt][1], b0, b1, zt[mt][jt]);
    }
    // ---- the wave's partial dh_{t-1}: unit u = j0 * J1 + j1, four consecutive j1 per lane and tile ----
    const float us = ldexpf(1.f, egt + el1 + ed - 28);
    float* pw = part + ((par ^ 1) * S::NWV + wave) * S::PART;
#pragma unroll
    for (int mt = 0; mt < S::M1T; ++mt)
#pragma unroll
      for (int jt = 0; jt < S::MT0; ++jt) {
        const int j0 = 16 * jt + n;
        if (j0 < S::J0) *reinterpret_cast<f32x4*>(pw + j0 * S::J1 + 16 * mt + 4 * q) = zt[mt][jt] * us;
      }
    lds_barrier();
  }
  // d_h0 / d_c0: the recurrent gradients that step 0 left (the parity written last is (0 & 1) ^ 1 = 1; T == 0: nothing ran)
  {
    const float* pr = part + (T > 0 ? 1 : 0) * S::NWV * S::PART;
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl) {
      if (g.d_h0) {
        float v = T > 0 ? ((pr[unit[tl]] + pr[S::PART + unit[tl]]) + (pr[2 * S::PART + unit[tl]] + pr[3 * S::PART + unit[tl]])) : 0.f;
        if (T == 0 && g.d_hT) v = g.d_hT[(size_t)b * S::H + unit[tl]];
        g.d_h0[(size_t)b * S::H + unit[tl]] = v;
      }
      if (g.d_c0) g.d_c0[(size_t)b * S::H + unit[tl]] = dc[tl];
      if (g.colmax) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          atomicMax(g.colmax + k * S::H + unit[tl], __float_as_uint(cmx[tl][k]));
          atomicMax(g.colmax + 4 * S::H + k * S::H + unit[tl], __float_as_uint(cmx[tl][k]));
        }
      }
    }
  }
}

// ---- TT-GRU reverse-time kernel (BPTT of gru.py:33-44 through the hidden chain), the twin of k_lstm_bwd_w2 -----------------------------
// Three waves per sample, wave w owns i1 in [16 w, 16 w + 16); a lane holds FOUR units (one per row tile) and, per unit, the hidden
// chain's gate gradients (dr, dz, dn r) in slots 0 .. 2 of the forward's four rows — slot 3 (the input part of n) is no row of the
// hidden matrix.  T2' / T1 / exchange as the LSTM kernel; the direct path dh_{t-1} += dh_t z_t stays in the lane's registers.
// d_gates_in gets (dr, dz, dn), d_gates_hid (dr, dz, dn r); by-products: both rows of column maxima.
template <class S>
__global__ void __launch_bounds__(64 * S::NWV) k_gru_bwd_w2(W2BArgs g) {
  constexpr int R = S::R, NTH = 64 * S::NWV;
  static_assert(S::GATES == 3 && (S::MT2 == 4 || S::MT2 == 3), "TT-GRU configuration");
  extern __shared__ __attribute__((aligned(16))) unsigned char w2b_smem[];
  unsigned char* smem = w2b_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int b = blockIdx.x, T = g.T;
  float* part = reinterpret_cast<float*>(smem);                 // [parity][wave][PART]
  for (int i = tid; i < 2 * S::NWV * S::PART / 4; i += NTH) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  xh8 gh[S::GHLB ? 1 : S::NT2][2][2];
  w2_xh4 gta[R][S::M1T][2];
  {
    const xh8* f8 = reinterpret_cast<const xh8*>(g.ghf);
    if constexpr (S::GHLB) {
      for (int i = tid; i < (int)(S::BGH_BYTES / 16); i += NTH)
        reinterpret_cast<f32x4*>(smem + S::BL_GH)[i] = reinterpret_cast<const f32x4*>(f8)[i];
    } else {
#pragma unroll
      for (int nt = 0; nt < S::NT2; ++nt)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            gh[nt][kb][p] = f8[((size_t)(2 * nt + kb) * 2 + p) * 64 + lane];
            asm volatile("" : "+v"(gh[nt][kb][p]));
          }
    }
    const w2_xh4* f4 = reinterpret_cast<const w2_xh4*>(g.gtf);
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
      for (int mt = 0; mt < S::M1T; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          gta[a][mt][p] = f4[((size_t)((wave * R + a) * S::M1T + mt) * 2 + p) * 64 + lane];
          asm volatile("" : "+v"(gta[a][mt][p]));
        }
  }
  const xh8* ghl = reinterpret_cast<const xh8*>(smem + S::BL_GH);
  const int egh = g.hdr[W2B_EGH], egt = g.hdr[W2B_EGT], el1 = g.hdr[W2B_EL1];
  const int shd = egh - el1 - 13;

  int unit[S::MT2];
#pragma unroll
  for (int tl = 0; tl < S::MT2; ++tl) unit[tl] = (4 * tl + q) * S::I1 + 16 * wave + n;
  const size_t bt0 = (size_t)b * T;
  constexpr int GH = 3 * S::H;
  // ---- records two steps deep: gates (r, z, n, W_hn h + b_hn), the previous state, d_out ----
  f32x4 G0[S::MT2], G1[S::MT2];
  float P0[S::MT2], P1[S::MT2], D0[S::MT2], D1[S::MT2];
  auto ld_gates = [&](int t, int tl) -> f32x4 {
    return t >= 0 ? *reinterpret_cast<const f32x4*>(g.reserve + res_gate(bt0 + t, S::H, unit[tl])) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto ld_hprev = [&](int t, int tl) -> float {                // h_{t-1}
    if (t >= 1) return g.out[(bt0 + t - 1) * S::H + unit[tl]];
    return (t == 0 && g.h0) ? g.h0[(size_t)b * S::H + unit[tl]] : 0.f;
  };
  auto ld_dout = [&](int t, int tl) -> float {
    return (t >= 0 && g.d_out) ? g.d_out[(bt0 + t) * S::H + unit[tl]] : 0.f;
  };
  float dhd[S::MT2], cmi[S::MT2][3], cmh[S::MT2];
#pragma unroll
  for (int tl = 0; tl < S::MT2; ++tl) {
    G0[tl] = ld_gates(T - 1, tl); G1[tl] = ld_gates(T - 2, tl);
    P0[tl] = ld_hprev(T - 1, tl); P1[tl] = ld_hprev(T - 2, tl);
    D0[tl] = ld_dout(T - 1, tl); D1[tl] = ld_dout(T - 2, tl);
    dhd[tl] = g.d_hT ? g.d_hT[(size_t)b * S::H + unit[tl]] : 0.f;
    cmh[tl] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) cmi[tl][k] = 0.f;
  }
  __syncthreads();

  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = T - 1; t >= 0; --t) {
    const int par = t & 1;
    const float* pr = part + par * S::NWV * S::PART;
    float dz[S::MT2][4];
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl) {
      float psum = pr[unit[tl]];
#pragma unroll
      for (int w = 1; w < S::NWV; ++w) psum += pr[w * S::PART + unit[tl]];
      const float dht = dhd[tl] + D0[tl] + psum;
      const float rg = G0[tl][0], zg = G0[tl][1], ng = G0[tl][2], hn = G0[tl][3];
      const float dn_pre = dht * (1.f - zg) * (1.f - ng * ng);
      const float dz_pre = dht * (P0[tl] - ng) * zg * (1.f - zg);
      const float dr_pre = dn_pre * hn * rg * (1.f - rg);
      dhd[tl] = dht * zg;
      dz[tl][0] = dr_pre; dz[tl][1] = dz_pre; dz[tl][2] = dn_pre * rg; dz[tl][3] = 0.f;
      float* gi = g.dg + (bt0 + t) * GH + unit[tl];
      float* gh_ = g.dg_hid + (bt0 + t) * GH + unit[tl];
      gi[0] = dr_pre; gi[S::H] = dz_pre; gi[2 * S::H] = dn_pre;
      gh_[0] = dr_pre; gh_[S::H] = dz_pre; gh_[2 * S::H] = dz[tl][2];
      cmi[tl][0] = fmaxf(cmi[tl][0], fabsf(dr_pre)); cmi[tl][1] = fmaxf(cmi[tl][1], fabsf(dz_pre));
      cmi[tl][2] = fmaxf(cmi[tl][2], fabsf(dn_pre)); cmh[tl] = fmaxf(cmh[tl], fabsf(dz[tl][2]));
      G0[tl] = G1[tl]; P0[tl] = P1[tl]; D0[tl] = D1[tl];
      G1[tl] = ld_gates(t - 2, tl);
      P1[tl] = ld_hprev(t - 2, tl);
      D1[tl] = ld_dout(t - 2, tl);
    }
    // ---- the wave's maximum -> exponent; A operand of T2': k-block 0 = row tiles 0, 1, k-block 1 = row tiles 2, 3 (x four slots) ----
    float m = 0.f;
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl)
#pragma unroll
      for (int k = 0; k < 3; ++k) m = fmaxf(m, fabsf(dz[tl][k]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const int ed = w2_expo(m);
    const float sd = ldexpf(1.f, 13 - ed);
    float v0[8], v1[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v0[k] = dz[0][k] * sd; v0[4 + k] = dz[1][k] * sd; v1[k] = dz[2][k] * sd;
      v1[4 + k] = S::MT2 == 4 ? dz[S::MT2 - 1][k] * sd : 0.f;
    }
    xh8 a0[2], a1[2];
    w2_split8(v0, a0[0], a0[1]);
    w2_split8(v1, a1[0], a1[1]);
    // ---- T2' + T1 ----
    f32x4 zt[S::M1T][S::MT0];
#pragma unroll
    for (int mt = 0; mt < S::M1T; ++mt)
#pragma unroll
      for (int jt = 0; jt < S::MT0; ++jt) zt[mt][jt] = z4;
#pragma unroll
    for (int nt = 0; nt < S::NT2; ++nt) {
      xh8 b00, b01, b10, b11;
      if constexpr (S::GHLB) {
        b00 = ghl[((size_t)(2 * nt + 0) * 2 + 0) * 64 + lane]; b01 = ghl[((size_t)(2 * nt + 0) * 2 + 1) * 64 + lane];
        b10 = ghl[((size_t)(2 * nt + 1) * 2 + 0) * 64 + lane]; b11 = ghl[((size_t)(2 * nt + 1) * 2 + 1) * 64 + lane];
      } else {
        b00 = gh[nt][0][0]; b01 = gh[nt][0][1]; b10 = gh[nt][1][0]; b11 = gh[nt][1][1];
      }
      f32x4 d = w2_mma3(a0[0], a0[1], b00, b01, z4);
      d = w2_mma3(a1[0], a1[1], b10, b11, d);
      unsigned p0a, p1a, p0b, p1b;
      split_pair_h(ldexpf(d[0], shd), ldexpf(d[1], shd), p0a, p1a);
      split_pair_h(ldexpf(d[2], shd), ldexpf(d[3], shd), p0b, p1b);
      const w2_xh4 b0 = __builtin_bit_cast(w2_xh4, u32x2{p0a, p0b}), b1 = __builtin_bit_cast(w2_xh4, u32x2{p1a, p1b});
      const int a = nt / S::MT0, jt = nt % S::MT0;
#pragma unroll
      for (int mt = 0; mt < S::M1T; ++mt) zt[mt][jt] = w2_mma3_16(gta[a][mt][0], gta[a][mt][1], b0, b1, zt[mt][jt]);
    }
    const float us = ldexpf(1.f, egt + el1 + ed - 28);
    float* pw = part + ((par ^ 1) * S::NWV + wave) * S::PART;
#pragma unroll
    for (int mt = 0; mt < S::M1T; ++mt)
#pragma unroll
      for (int jt = 0; jt < S::MT0; ++jt) {
        const int j0 = 16 * jt + n;
        if (j0 < S::J0) *reinterpret_cast<f32x4*>(pw + j0 * S::J1 + 16 * mt + 4 * q) = zt[mt][jt] * us;
      }
    lds_barrier();
  }
  {
    const float* pr = part + (T > 0 ? 1 : 0) * S::NWV * S::PART;
#pragma unroll
    for (int tl = 0; tl < S::MT2; ++tl) {
      if (g.d_h0) {
        float v = dhd[tl];
        if (T > 0) {
#pragma unroll
          for (int w = 0; w < S::NWV; ++w) v += pr[w * S::PART + unit[tl]];
        }
        g.d_h0[(size_t)b * S::H + unit[tl]] = v;
      }
      if (g.colmax) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          atomicMax(g.colmax + k * S::H + unit[tl], __float_as_uint(cmi[tl][k]));
          atomicMax(g.colmax + GH + k * S::H + unit[tl], __float_as_uint(k < 2 ? cmi[tl][k] : cmh[tl]));
        }
      }
    }
  }
}

// ---- d = 4: the cores contracted pairwise ONCE per launch into the packed layout of a two-core matrix (I0 I1, I2 I3) x (J0 J1,
// J2 J3) of rank R_2, which the prep kernels then read like the d = 2 layers' packed cores (one load per entry: with the
// contraction inside w2_gt / w2_gh every fragment tile's workgroup re-derived every merged entry for its maxima — k_w2_prep 67 us,
// k_w2b_prep 59 us against 11 / 10 us of the two-core layers).  One thread per merged entry; mats = 1: the hidden matrix only.
struct W2Merge { TtShape s4[2]; TtShape s2[2]; const float* pk4[2]; float* pk2[2]; int n[2]; };
__global__ void __launch_bounds__(256) k_w2_merge(W2Merge m, int mats) {
  long t = (long)blockIdx.x * 256 + threadIdx.x;
  for (int q = 0; q < mats; ++q) {
    if (t >= m.n[q]) { t -= m.n[q]; continue; }
    const TtShape& s = m.s4[q];
    const TtShape& z = m.s2[q];
    const int r = z.R[1];
    const int nh = z.K[0] * z.M[0];                        // head entries: [(j0 r + b)][i0]
    float v = 0.f;
    if (t < nh) {
      const int i0 = (int)(t % z.M[0]), jb = (int)(t / z.M[0]), b = jb % r, j0 = jb / r;
      const int ia = i0 / s.I[1], ib = i0 % s.I[1], ja = j0 / s.J[1], jbb = j0 % s.J[1];
      for (int k = 0; k < s.R[1]; ++k) v = fmaf(w2_core(s, m.pk4[q], 0, 0, ia, ja, k), w2_core(s, m.pk4[q], 1, k, ib, jbb, b), v);
    } else {
      const long u = t - nh;                                // tail entries: [j1][i1 r + a]
      const int ia_ = (int)(u % z.M[1]), j1 = (int)(u / z.M[1]), a = ia_ % r, i1 = ia_ / r;
      const int ia = i1 / s.I[3], ib = i1 % s.I[3], ja = j1 / s.J[3], jb = j1 % s.J[3];
      for (int k = 0; k < s.R[3]; ++k) v = fmaf(w2_core(s, m.pk4[q], 2, a, ia, ja, k), w2_core(s, m.pk4[q], 3, k, ib, jb, 0), v);
    }
    m.pk2[q][z.woff[0] + t] = v;
    return;
  }
}
// the two-core shape of a pairwise-contracted four-core matrix
static bool w2_shape2(const TtShape& s, TtShape* z) {
  ttrnn_ttm w{};
  w.d = 2;
  w.in_modes[0] = s.J[0] * s.J[1]; w.in_modes[1] = s.J[2] * s.J[3];
  w.out_modes[0] = s.I[0] * s.I[1]; w.out_modes[1] = s.I[2] * s.I[3];
  w.ranks[0] = 1; w.ranks[1] = s.R[2]; w.ranks[2] = 1;
  return tt_shape_init(z, &w) == TTRNN_OK && z->woff[0] == 0 && z->woff[1] == z->K[0] * z->M[0];
}
constexpr size_t W2_MERGE_BYTES = (size_t)((48 * 16 + 64 * 48 + 48 * 4 + 64 * 10) * 4 + 64) * 4;      // both matrices at rank 4, fp32

// which configuration serves a layer: 0 = none; 1, 2 = A (d = 2) with 2 / 4 rank slots; 3, 4 = B (d = 4: cores contracted pairwise)
int w2_config(const TtShape& hid, const TtShape& in) {
  int cfg = 0;
  if (hid.d == 2 && in.d == 2 && hid.J[0] == 24 && hid.J[1] == 32 && hid.I[0] == 48 && hid.I[1] == 64 && in.J[0] == 5 && in.J[1] == 8 &&
      in.I[0] == 48 && in.I[1] == 64)
    cfg = 1;
  else if (hid.d == 4 && in.d == 4 && hid.J[0] * hid.J[1] == 16 && hid.J[2] * hid.J[3] == 48 && hid.I[0] * hid.I[1] == 48 &&
           hid.I[2] * hid.I[3] == 64 && in.J[0] * in.J[1] == 4 && in.J[2] * in.J[3] == 10 && in.I[0] * in.I[1] == 48 && in.I[2] * in.I[3] == 64 &&
           in.I[1] == hid.I[1] && in.I[3] == hid.I[3])
    cfg = 3;
  else
    return 0;
  for (int k = 0; k <= hid.d; ++k) if (hid.R[k] > 16 || in.R[k] > 16) return 0;      // (the merges loop over the inner ranks)
  const int r = hid.R[hid.d / 2] > in.R[in.d / 2] ? hid.R[hid.d / 2] : in.R[in.d / 2];
  return r <= 2 ? cfg : (r <= 4 ? cfg + 1 : 0);
}

// the TT-GRU of the encoder's layer: 0 = none; 1, 2 = two cores with 2 / 4 rank slots
// ... 3, 4 = four cores contracted pairwise (tt_shape (4, 4, 6, 8) x (6, 6, 8, 8), input (2, 2, 2, 5)): (16, 48) x (36, 64)
int w2_config_gru(const TtShape& hid, const TtShape& in) {
  int cfg = 0;
  if (hid.d == 2 && in.d == 2 && hid.J[0] == 24 && hid.J[1] == 32 && hid.I[0] == 48 && hid.I[1] == 48 && in.J[0] == 5 && in.J[1] == 8 &&
      in.I[0] == 48 && in.I[1] == 48)
    cfg = 1;
  else if (hid.d == 4 && in.d == 4 && hid.J[0] * hid.J[1] == 16 && hid.J[2] * hid.J[3] == 48 && hid.I[0] * hid.I[1] == 36 &&
           hid.I[2] * hid.I[3] == 64 && in.J[0] * in.J[1] == 4 && in.J[2] * in.J[3] == 10 && in.I[0] * in.I[1] == 36 && in.I[2] * in.I[3] == 64 &&
           in.I[1] == hid.I[1] && in.I[3] == hid.I[3])
    cfg = 3;
  else
    return 0;
  for (int k = 0; k <= hid.d; ++k) if (hid.R[k] > 16 || in.R[k] > 16) return 0;
  const int r = hid.R[hid.d / 2] > in.R[in.d / 2] ? hid.R[hid.d / 2] : in.R[in.d / 2];
  return r <= 2 ? cfg : (r <= 4 ? cfg + 1 : 0);
}

template <class S>
static int launch_fwd_w2g_t(const RnnShape& rs, const void* x, const void* h0, const float* packed_in, const void* bias_in,
                            const float* packed_hid, const void* bias_hid, void* out, void* hT, float* reserve, void* ws,
                            hipStream_t stream, int phase) {
  int* hdr = (int*)ws;
  _Float16* frag = (_Float16*)((char*)ws + S::HDR_BYTES);
  if (phase != TTRNN_PHASE_RUN) {
    if (rs.hid_s.d == 4) {
      W2Merge m{};
      m.s4[0] = rs.hid_s; m.s4[1] = rs.in_s;
      if (!w2_shape2(rs.hid_s, &m.s2[0]) || !w2_shape2(rs.in_s, &m.s2[1])) return TTRNN_ERR_UNSUPPORTED;
      m.pk4[0] = packed_hid; m.pk4[1] = packed_in;
      m.n[0] = m.s2[0].wtotal; m.n[1] = m.s2[1].wtotal;
      m.pk2[0] = (float*)((char*)ws + S::WS_BYTES); m.pk2[1] = m.pk2[0] + ((m.n[0] + 15) & ~15);
      if ((size_t)(((m.n[0] + 15) & ~15) + m.n[1]) * 4 > W2_MERGE_BYTES) return TTRNN_ERR_UNSUPPORTED;
      hipLaunchKernelGGL(k_w2_merge, dim3((m.n[0] + m.n[1] + 255) / 256), dim3(256), 0, stream, m, 2);
      hipLaunchKernelGGL(k_w2_prep<S>, dim3(S::NTILES), dim3(256), 0, stream, m.s2[0], m.s2[1], (const float*)m.pk2[0], (const float*)m.pk2[1], hdr, frag);
    } else {
      hipLaunchKernelGGL(k_w2_prep<S>, dim3(S::NTILES), dim3(256), 0, stream, rs.hid_s, rs.in_s, packed_hid, packed_in, hdr, frag);
    }
  }
  if (phase == TTRNN_PHASE_PREPARE) return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  W2Args a{};
  a.x = (const float*)x; a.h0 = (const float*)h0; a.c0 = nullptr;
  a.bias_in = rs.has_bias_in ? (const float*)bias_in : nullptr;
  a.bias_hid = rs.has_bias_hid ? (const float*)bias_hid : nullptr;
  a.hdr = hdr; a.frag = frag;
  a.out = (float*)out; a.hT = (float*)hT; a.cT = nullptr; a.reserve = reserve;
  a.B = rs.B; a.T = rs.T;
  hipLaunchKernelGGL(k_gru_fwd_w2<S>, dim3(rs.B), dim3(64 * S::NWV), 0, stream, a);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

template <class S>
static int launch_fwd_w2_t(const RnnShape& rs, const void* x, const void* h0, const void* c0, const float* packed_in, const void* bias_in,
                           const float* packed_hid, const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws,
                           hipStream_t stream, int phase) {
  int* hdr = (int*)ws;
  _Float16* frag = (_Float16*)((char*)ws + S::HDR_BYTES);
  if (phase != TTRNN_PHASE_RUN) {
    if (rs.hid_s.d == 4) {
      W2Merge m{};
      m.s4[0] = rs.hid_s; m.s4[1] = rs.in_s;
      if (!w2_shape2(rs.hid_s, &m.s2[0]) || !w2_shape2(rs.in_s, &m.s2[1])) return TTRNN_ERR_UNSUPPORTED;
      m.pk4[0] = packed_hid; m.pk4[1] = packed_in;
      m.n[0] = m.s2[0].wtotal; m.n[1] = m.s2[1].wtotal;
      m.pk2[0] = (float*)((char*)ws + S::WS_BYTES); m.pk2[1] = m.pk2[0] + ((m.n[0] + 15) & ~15);
      if ((size_t)(((m.n[0] + 15) & ~15) + m.n[1]) * 4 > W2_MERGE_BYTES) return TTRNN_ERR_UNSUPPORTED;
      hipLaunchKernelGGL(k_w2_merge, dim3((m.n[0] + m.n[1] + 255) / 256), dim3(256), 0, stream, m, 2);
      hipLaunchKernelGGL(k_w2_prep<S>, dim3(S::NTILES), dim3(256), 0, stream, m.s2[0], m.s2[1], (const float*)m.pk2[0], (const float*)m.pk2[1], hdr, frag);
    } else {
      hipLaunchKernelGGL(k_w2_prep<S>, dim3(S::NTILES), dim3(256), 0, stream, rs.hid_s, rs.in_s, packed_hid, packed_in, hdr, frag);
    }
  }
  if (phase == TTRNN_PHASE_PREPARE) return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  W2Args a{};
  a.x = (const float*)x; a.h0 = (const float*)h0; a.c0 = (const float*)c0;
  a.bias_in = rs.has_bias_in ? (const float*)bias_in : nullptr;
  a.bias_hid = rs.has_bias_hid ? (const float*)bias_hid : nullptr;
  a.hdr = hdr; a.frag = frag;
  a.out = (float*)out; a.hT = (float*)hT; a.cT = (float*)cT; a.reserve = reserve;
  a.B = rs.B; a.T = rs.T;
  hipLaunchKernelGGL(k_lstm_fwd_w2<S>, dim3(rs.B), dim3(64 * S::NWV), 0, stream, a);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

template <class S>
static int launch_bwd_w2_t(const RnnShape& rs, const void* c0, const float* packed_hid, const float* reserve,
                           const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, void* d_h0, void* d_c0, void* ws,
                           hipStream_t stream, float* stats) {
  int* hdr = (int*)ws;
  _Float16* ghf = (_Float16*)((char*)ws + S::HDR_BYTES);
  _Float16* gtf = (_Float16*)((char*)ws + S::HDR_BYTES + S::BGH_BYTES);
  if (stats && hipMemsetAsync(stats, 0, (size_t)2 * 4 * rs.H * sizeof(float), stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
  if (rs.hid_s.d == 4) {
    W2Merge m{};
    m.s4[0] = rs.hid_s;
    if (!w2_shape2(rs.hid_s, &m.s2[0])) return TTRNN_ERR_UNSUPPORTED;
    m.pk4[0] = packed_hid; m.n[0] = m.s2[0].wtotal;
    m.pk2[0] = (float*)((char*)ws + S::BWS_BYTES);
    if ((size_t)m.n[0] * 4 > W2_MERGE_BYTES) return TTRNN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_w2_merge, dim3((m.n[0] + 255) / 256), dim3(256), 0, stream, m, 1);
    hipLaunchKernelGGL(k_w2b_prep<S>, dim3(S::NT2 * 2 + S::NWV * S::R * S::M1T), dim3(256), 0, stream, m.s2[0], (const float*)m.pk2[0], hdr, ghf, gtf);
  } else {
    hipLaunchKernelGGL(k_w2b_prep<S>, dim3(S::NT2 * 2 + S::NWV * S::R * S::M1T), dim3(256), 0, stream, rs.hid_s, packed_hid, hdr, ghf, gtf);
  }
  W2BArgs a{};
  a.c0 = (const float*)c0; a.reserve = reserve;
  a.d_out = (const float*)d_out; a.d_hT = (const float*)d_hT; a.d_cT = (const float*)d_cT;
  a.hdr = hdr; a.ghf = ghf; a.gtf = gtf;
  a.dg = dg_in; a.d_h0 = (float*)d_h0; a.d_c0 = (float*)d_c0;
  a.colmax = (unsigned*)stats;
  a.B = rs.B; a.T = rs.T;
  if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_lstm_bwd_w2<S>), S::BLDS) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
  hipLaunchKernelGGL(k_lstm_bwd_w2<S>, dim3(rs.B), dim3(256), S::BLDS, stream, a);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

template <class S>
static int launch_bwd_w2g_t(const RnnShape& rs, const void* out, const void* h0, const float* packed_hid, const float* reserve,
                            const void* d_out, const void* d_hT, float* dg_in, float* dg_hid, void* d_h0, void* ws, hipStream_t stream,
                            float* stats) {
  int* hdr = (int*)ws;
  _Float16* ghf = (_Float16*)((char*)ws + S::HDR_BYTES);
  _Float16* gtf = (_Float16*)((char*)ws + S::HDR_BYTES + S::BGH_BYTES);
  if (stats && hipMemsetAsync(stats, 0, (size_t)2 * 3 * rs.H * sizeof(float), stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
  if (rs.hid_s.d == 4) {
    W2Merge m{};
    m.s4[0] = rs.hid_s;
    if (!w2_shape2(rs.hid_s, &m.s2[0])) return TTRNN_ERR_UNSUPPORTED;
    m.pk4[0] = packed_hid; m.n[0] = m.s2[0].wtotal;
    m.pk2[0] = (float*)((char*)ws + S::BWS_BYTES);
    if ((size_t)m.n[0] * 4 > W2_MERGE_BYTES) return TTRNN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_w2_merge, dim3((m.n[0] + 255) / 256), dim3(256), 0, stream, m, 1);
    hipLaunchKernelGGL(k_w2b_prep<S>, dim3(S::NT2 * 2 + S::NWV * S::R * S::M1T), dim3(256), 0, stream, m.s2[0], (const float*)m.pk2[0], hdr, ghf, gtf);
  } else {
    hipLaunchKernelGGL(k_w2b_prep<S>, dim3(S::NT2 * 2 + S::NWV * S::R * S::M1T), dim3(256), 0, stream, rs.hid_s, packed_hid, hdr, ghf, gtf);
  }
  W2BArgs a{};
  a.reserve = reserve; a.out = (const float*)out; a.h0 = (const float*)h0;
  a.d_out = (const float*)d_out; a.d_hT = (const float*)d_hT;
  a.hdr = hdr; a.ghf = ghf; a.gtf = gtf;
  a.dg = dg_in; a.dg_hid = dg_hid; a.d_h0 = (float*)d_h0;
  a.colmax = (unsigned*)stats;
  a.B = rs.B; a.T = rs.T;
  if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_gru_bwd_w2<S>), S::BLDS) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
  hipLaunchKernelGGL(k_gru_bwd_w2<S>, dim3(rs.B), dim3(64 * S::NWV), S::BLDS, stream, a);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

}  // namespace

// the forward route of this file: fp32 storage, split fp32 math, a plain (not block-structured) TT-LSTM of the encoder's size
bool w2_rnn_fwd_available(const RnnShape& rs, int dtype) {
  if (dtype != TTRNN_F32 || rs.hid_blocks > 1 || rs.H != 768 || rs.in != 40 || opt(OPT_FP32_MATH) != TTRNN_MATH_SPLIT ||
      opt(OPT_FORCE_GENERIC) || opt(OPT_FORCE_G2) || (opt(OPT_DEV2) & 16))
    return false;
  if (rs.cell == TTRNN_GRU) return w2_config_gru(rs.hid_s, rs.in_s) > 0;
  return rs.cell == TTRNN_LSTM && w2_config(rs.hid_s, rs.in_s) > 0;
}
static constexpr size_t w2_max4(size_t a, size_t b, size_t c, size_t d) { return (a > b ? a : b) > (c > d ? c : d) ? (a > b ? a : b) : (c > d ? c : d); }
size_t w2_rnn_fwd_workspace_bytes() {      // (the query has no shape; the four-core configurations keep their contracted cores behind the fragments)
  const size_t a = w2_max4(W2A2::WS_BYTES, W2A4::WS_BYTES, W2B2::WS_BYTES + W2_MERGE_BYTES, W2B4::WS_BYTES + W2_MERGE_BYTES);
  const size_t g = w2_max4(W2GA2::WS_BYTES, W2GA4::WS_BYTES, W2GB2::WS_BYTES + W2_MERGE_BYTES, W2GB4::WS_BYTES + W2_MERGE_BYTES);
  return a > g ? a : g;
}

int launch_rnn_fwd_w2(const RnnShape& rs, const void* x, const void* h0, const void* c0, const float* packed_in, const void* bias_in,
                      const float* packed_hid, const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws,
                      hipStream_t stream, int phase) {
  if (rs.cell == TTRNN_GRU) {
    switch (w2_config_gru(rs.hid_s, rs.in_s)) {
      case 1: return launch_fwd_w2g_t<W2GA2>(rs, x, h0, packed_in, bias_in, packed_hid, bias_hid, out, hT, reserve, ws, stream, phase);
      case 2: return launch_fwd_w2g_t<W2GA4>(rs, x, h0, packed_in, bias_in, packed_hid, bias_hid, out, hT, reserve, ws, stream, phase);
      case 3: return launch_fwd_w2g_t<W2GB2>(rs, x, h0, packed_in, bias_in, packed_hid, bias_hid, out, hT, reserve, ws, stream, phase);
      case 4: return launch_fwd_w2g_t<W2GB4>(rs, x, h0, packed_in, bias_in, packed_hid, bias_hid, out, hT, reserve, ws, stream, phase);
    }
    return TTRNN_ERR_UNSUPPORTED;
  }
  switch (w2_config(rs.hid_s, rs.in_s)) {
    case 1: return launch_fwd_w2_t<W2A2>(rs, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT, reserve, ws, stream, phase);
    case 2: return launch_fwd_w2_t<W2A4>(rs, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT, reserve, ws, stream, phase);
    case 3: return launch_fwd_w2_t<W2B2>(rs, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT, reserve, ws, stream, phase);
    case 4: return launch_fwd_w2_t<W2B4>(rs, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT, reserve, ws, stream, phase);
  }
  return TTRNN_ERR_UNSUPPORTED;
}

// reverse-time kernel of the same shapes (the forward's reserve format is everybody's: ttrnn_core.h res_gate / res_cell)
bool w2_rnn_bwd_available(const RnnShape& rs, int dtype) {
  if (dtype != TTRNN_F32 || rs.hid_blocks > 1 || rs.H != 768 || rs.in != 40 || opt(OPT_FP32_MATH) != TTRNN_MATH_SPLIT ||
      opt(OPT_FORCE_GENERIC) || opt(OPT_FORCE_G2) || (opt(OPT_DEV2) & 32))
    return false;
  if (rs.cell == TTRNN_GRU) return w2_config_gru(rs.hid_s, rs.in_s) > 0;
  return rs.cell == TTRNN_LSTM && w2_config(rs.hid_s, rs.in_s) > 0;
}
size_t w2_rnn_bwd_workspace_bytes() {
  const size_t a = w2_max4(W2A2::BWS_BYTES, W2A4::BWS_BYTES, W2B2::BWS_BYTES + W2_MERGE_BYTES, W2B4::BWS_BYTES + W2_MERGE_BYTES);
  const size_t g = w2_max4(W2GA2::BWS_BYTES, W2GA4::BWS_BYTES, W2GB2::BWS_BYTES + W2_MERGE_BYTES, W2GB4::BWS_BYTES + W2_MERGE_BYTES);
  return a > g ? a : g;
}

int launch_rnn_bwd_w2(const RnnShape& rs, const void* out, const void* h0, const void* c0, const float* packed_hid, const float* reserve,
                      const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid, void* d_h0, void* d_c0, void* ws,
                      hipStream_t stream, float* stats) {
  if (rs.cell == TTRNN_GRU) {
    switch (w2_config_gru(rs.hid_s, rs.in_s)) {
      case 1: return launch_bwd_w2g_t<W2GA2>(rs, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws, stream, stats);
      case 2: return launch_bwd_w2g_t<W2GA4>(rs, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws, stream, stats);
      case 3: return launch_bwd_w2g_t<W2GB2>(rs, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws, stream, stats);
      case 4: return launch_bwd_w2g_t<W2GB4>(rs, out, h0, packed_hid, reserve, d_out, d_hT, dg_in, dg_hid, d_h0, ws, stream, stats);
    }
    return TTRNN_ERR_UNSUPPORTED;
  }
  switch (w2_config(rs.hid_s, rs.in_s)) {
    case 1: return launch_bwd_w2_t<W2A2>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, d_h0, d_c0, ws, stream, stats);
    case 2: return launch_bwd_w2_t<W2A4>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, d_h0, d_c0, ws, stream, stats);
    case 3: return launch_bwd_w2_t<W2B2>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, d_h0, d_c0, ws, stream, stats);
    case 4: return launch_bwd_w2_t<W2B4>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, d_h0, d_c0, ws, stream, stats);
  }
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
