// ttrnn_fast_f2.hip — TT-LSTM forward for TWO-core hidden matrices (pMNIST's default `--ncores 2`, BASELINE configs[0]:
// H = 128, ranks 4; gfx950): both chain stages on two-piece fp16 operands, gates on the accumulators, ONE barrier per step.
//
// Replaces: tensorized_rnn/lstm.py:23-32,123-133 + t3nsor/ops.py:78-93 for d = 2
//   (ops.py:81-90 runs k = 1 then k = 0:  C1[j0][(i1,a)] = sum_j1 G1[a,i1,j1] h[j0,j1];   y[i0][i1] = sum_(j0,a) G0[i0,j0,a] C1[j0][(i1,a)]).
//
// With d = 2 the hidden chain IS two stages, so there is nothing to fuse at prep time — what round 3's stage-wise fp32-MFMA
// kernel (k_lstm_fwd_fused: 3 barriers per step, 48 fp32 MFMAs of 32 cycles, 1 700 cycles per step, the kernel furthest below
// its roofline in VERDICT r3) leaves on the table is the hand-off structure.  Here a sample is a workgroup of TWO waves, and
// wave w owns the output-mode slice i1 in [16 w, 16 w + 16) through BOTH stages:
//   stage 1   four m-tiles (rows (i1, a), 16 each) x the J0 = 8 chain rows as MFMA columns, K = J1 = 16: the three split terms
//             are two chained v_mfma_f32_16x16x32_f16 per tile — A = [w0 | w1] against B = [x0 | x0], then A = [w0 | 0] against
//             [x1 | -] — both B operands ONE ds_read_b128 each per step, shared by all tiles;
//   hand-off  the accumulators (four consecutive rank indices a of one (i1, j0)) are split into two fp16 pieces and stored as
//             the B operand of stage 0, [i1][k = (j0, a)]: rows the SAME wave reads back — no barrier, one lgkmcnt wait;
//   stage 0   ONE tile per wave: A = G0 with its rows permuted (MFMA row 4 q + j <-> i0 = 4 j + q, so that a lane's four
//             accumulator registers are the gates i, f, g, o of ONE hidden unit), K = J0 R1 = 32 = one k-block, three terms;
//   gates     on the accumulators (initial value = input projection + biases, pre-scaled: the sum IS the v_exp_f32 argument),
//             c in a register, h_t written as two fp16 pieces of 2^6 h into the stage-1 operand image of the next step;
//   barrier   the only one: wave w's h units are the other wave's stage-1 operand too.
// Scales: two-sided diagonal powers of two as in the fused-core kernels (ttrnn_f10_dev.h) — per i1 and per rank index a for
// G1, per output row i0 for G0 — computed once per launch by k_f2_prep together with the MFMA fragments; a caller's h_0 per
// sample (f10h_h0_expo).  Only TTRNN_MATH_SPLIT takes this kernel; TTRNN_MATH_EXACT keeps the fp32-MFMA stage-wise kernel.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_f10_dev.h"

namespace ttrnn {

template <class S>
struct F2 {
  static constexpr int J0 = S::J[0], J1 = S::J[1], I0 = S::I[0], I1 = S::I[1], R1 = S::R[1];
  static constexpr int H = J0 * J1;
  static constexpr int NWV = I1 / 16;             // waves per sample: one stage-0 column tile (16 values of i1) each
  static constexpr int MT1 = 16 * R1 / 16;        // stage-1 m-tiles per wave: rows (i1, a) of its sixteen i1
  static constexpr int K0 = J0 * R1;              // stage-0 contraction
  // header (int32): exponents u[I1] | v[R1 (padded to 16)] | p[I0]
  static constexpr int HU = 0, HV = I1, HP = I1 + 16, HGUARD = I1 + 16 + I0, HDR_INTS = I1 + 16 + I0 + 1;
  static constexpr int HDR_BYTES = 512;
  // fragments behind the header: stage 1 [NWV * MT1 tiles][2 (MFMA 1 / 2)][64 lanes] xh8, stage 0 [2 pieces][64] xh8
  static constexpr size_t FRAG1 = (size_t)NWV * MT1 * 2 * 64, FRAG0 = 2 * 64;
  static constexpr size_t WS_BYTES = HDR_BYTES + (FRAG1 + FRAG0) * sizeof(xh8);
  // LDS: h pieces [parity][piece][H] fp16 + per wave the stage-0 operand [piece][16 rows][K0] fp16
  static constexpr size_t LDS_BYTES = 2 * 2 * H * 2 + (size_t)NWV * 2 * 16 * K0 * 2;
};
template <class S>
constexpr bool f2_ok() {
  using F = F2<S>;
  return S::D == 2 && S::R[0] == 1 && S::R[2] == 1 && F::J1 == 16 && F::R1 == 4 && F::J0 == 8 && F::I0 == 16 &&
         F::I1 % 16 == 0 && F::NWV >= 1 && F::NWV <= 4 && F::I0 * F::I1 == 4 * F::H && F::K0 == 32 &&
         F::HDR_INTS * sizeof(int) <= F::HDR_BYTES;
}

// byte offset of (row, 16-byte slot, byte) in a wave's stage-0 operand plane [16 rows][64 bytes]: slots XOR-swizzled by row / 4,
// so that the ds_read_b128 of the sixteen rows (same slot) and the ds_write_b64 of eight chain rows (same row) are conflict-free
__device__ __forceinline__ int f2_img_off(int row, int slot, int byte) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4) + byte; }

// ---- prep: exponents + fragments, one workgroup (the two cores are 2 560 floats) ------------------------------------------------
template <class S>
__global__ void __launch_bounds__(256) k_f2_prep(const float* __restrict__ packed, int* __restrict__ hdr_i, xh8* __restrict__ frag) {
  using F = F2<S>;
  constexpr int M1 = F::I1 * F::R1;                       // rows of W1 = (i1, a)
  __shared__ float g1[F::J1 * M1];                        // W1[j1][(i1, a)]
  __shared__ float g0[F::K0 * F::I0];                     // W0[(j0, a)][i0]
  __shared__ int eu[F::I1], ev[16], ep[F::I0];
  __shared__ int trip;
  const int tid = threadIdx.x;
  if (tid == 0) trip = 0;
  const float* W0 = packed + woff_of<S>(0);
  const float* W1 = packed + woff_of<S>(1);
  for (int i = tid; i < F::J1 * M1; i += 256) g1[i] = W1[i];
  for (int i = tid; i < F::K0 * F::I0; i += 256) g0[i] = W0[i];
  __syncthreads();
  if (tid < F::I1) {                                      // u[i1]: max over (a, j1) < 2^u
    float mx = 0.f;
    for (int j1 = 0; j1 < F::J1; ++j1)
      for (int a = 0; a < F::R1; ++a) mx = fmaxf(mx, fabsf(g1[j1 * M1 + tid * F::R1 + a]));
    eu[tid] = f10h_expo(mx);
  }
  __syncthreads();
  {                                                       // v[a]: max over (i1, j1) of 2^-u[i1] |G1| < 2^v — one thread per (i1, a)
    __shared__ unsigned vmax[16];                         // row, the rows of an `a` merged by atomicMax on the bit patterns (>= 0)
    if (tid < 16) vmax[tid] = 0u;
    __syncthreads();
    if (tid < M1) {
      float mx = 0.f;
      for (int j1 = 0; j1 < F::J1; ++j1) mx = fmaxf(mx, fabsf(g1[j1 * M1 + tid]));
      atomicMax(&vmax[tid % F::R1], __float_as_uint(mx * ldexpf(1.f, -eu[tid / F::R1])));
    }
    __syncthreads();
    if (tid < F::R1) ev[tid] = f10h_expo(__uint_as_float(vmax[tid]));
  }
  __syncthreads();
  if (tid < F::I0) {                                      // p[i0]: 2^p max over (j0, a) of 2^v[a] |G0| in [2^11, 2^12)
    float mx = 0.f;
    for (int k = 0; k < F::K0; ++k) mx = fmaxf(mx, fabsf(g0[k * F::I0 + tid]) * ldexpf(1.f, ev[k % F::R1]));
    ep[tid] = 12 - f10h_expo(mx);
  }
  __syncthreads();
  if (tid < F::I1) hdr_i[F::HU + tid] = eu[tid];
  if (tid < 16) hdr_i[F::HV + tid] = tid < F::R1 ? ev[tid] : 0;
  if (tid < F::I0) hdr_i[F::HP + tid] = ep[tid];
  // Guard.  u[i1] + v[a] is the most general scale the two stages can share (i1 is a free index of both, a the contraction index
  // of stage 0), so ONE large entry G1[a*, i1*, .] drags the rows (i1*, a != a*) of its slice down with it: from 2^-5 of the
  // slice's ceiling on, their second fp16 pieces are subnormal and those rows keep fewer than 20 of their 22 bits (a core entry
  // x 1e5: 12).  Such weights — and rank indices whose v differ by more than 8 binades, the same effect on the rows of G0 — leave
  // for the fp32-MFMA stage-wise kernel queued behind this launch (launch_f2; counted in TTRNN_STAT_GUARD_TRIPS).
  if (tid < M1) {
    float mx = 0.f;
    for (int j1 = 0; j1 < F::J1; ++j1) mx = fmaxf(mx, fabsf(g1[j1 * M1 + tid]));
    mx *= ldexpf(1.f, 5 - eu[tid / F::R1] - ev[tid % F::R1]);
    if (mx > 0.f && mx < 0.03125f) atomicOr(&trip, 1);
  }
  if (tid == 0) {
    int lo = ev[0], hi = ev[0];
    for (int a = 1; a < F::R1; ++a) { lo = ev[a] < lo ? ev[a] : lo; hi = ev[a] > hi ? ev[a] : hi; }
    if (hi - lo > 8) atomicOr(&trip, 1);
  }
  __syncthreads();
  if (tid == 0) hdr_i[F::HGUARD] = trip;
  // stage-1 fragments: tile mt (rows m = 16 mt + r = (i1, a) = (m / R1, m % R1)), lane (r, q): k = 8 q + e;
  //   MFMA 1: k < 16 -> piece 0 of j1 = k, k >= 16 -> piece 1 of j1 = k - 16;  MFMA 2: k < 16 -> piece 0, else 0
  for (int idx = tid; idx < F::NWV * F::MT1 * 64; idx += 256) {
    const int mt = idx >> 6, lane = idx & 63, r = lane & 15, q = lane >> 4;
    const int m = 16 * mt + r, i1 = m / F::R1, a = m % F::R1;
    const float sc = ldexpf(1.f, 5 - eu[i1] - ev[a]);
    xh8 f1, f2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j1 = (8 * q + e) & 15;
      _Float16 p0, p1;
      split2h(g1[j1 * M1 + m] * sc, p0, p1);
      f1[e] = q < 2 ? p0 : p1;
      f2[e] = q < 2 ? p0 : (_Float16)0.f;
    }
    frag[(size_t)(mt * 2 + 0) * 64 + lane] = f1;
    frag[(size_t)(mt * 2 + 1) * 64 + lane] = f2;
  }
  // stage-0 fragments: lane (r, q): MFMA row r <-> i0 = 4 (r & 3) + (r >> 2) (gate r & 3); k = 8 q + e = (j0 = 2 q + e / 4, a = e % 4);
  // the gate factor (-log2 e for i, f, o; 2 log2 e for g) folded in: an accumulator, unscaled, IS the exp2 argument
  if (tid < 64) {
    const int r = tid & 15, q = tid >> 4;
    const int i0 = 4 * (r & 3) + (r >> 2);
    const float gf = (r & 3) == 2 ? 2.8853900817779268f : -1.4426950408889634f;
    xh8 a0, a1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * q + e;
      _Float16 p0, p1;
      split2h(g0[k * F::I0 + i0] * ldexpf(1.f, ev[k % F::R1] + ep[i0]) * gf, p0, p1);
      a0[e] = p0; a1[e] = p1;
    }
    frag[F::FRAG1 + tid] = a0;
    frag[F::FRAG1 + 64 + tid] = a1;
  }
}

// ---- the recurrent kernel ---------------------------------------------------------------------------------------------------------
// H0: the caller passed an initial state (may lie outside (-1, 1): per-sample exponent);  OUT = false: only the final state is wanted
// IN1: input_size == 1 (the projection of the two unit rows, scaled by x_t) — a template parameter: a runtime flag put four uniform
// branches and both code paths into every step
template <class S, bool H0, bool OUT, bool IN1>
__global__ void __launch_bounds__(F2<S>::NWV * 64, 4) k_lstm_fwd_f2(int B, int T, GinSrc gs, const float* __restrict__ h0,
                                                                 const float* __restrict__ c0, const int* __restrict__ hdr,
                                                                 const xh8* __restrict__ frag, const float* __restrict__ bias_hid,
                                                                 float* __restrict__ out, float* __restrict__ hT,
                                                                 float* __restrict__ cT, float* __restrict__ reserve) {
  static_assert(f2_ok<S>(), "shape not supported by the two-core kernel");
  using F = F2<S>;
  constexpr int H = F::H, NWV = F::NWV, MT1 = F::MT1;
  if (hdr[F::HGUARD] != 0) return;                         // k_f2_prep's guard: the stage-wise kernel queued behind runs instead
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f2[];
  _Float16* hpl = reinterpret_cast<_Float16*>(smem_f2);                    // [parity][piece][H]: pieces of 2^6 h
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* img = smem_f2 + 2 * 2 * H * 2 + (size_t)wave * 2 * 16 * F::K0 * 2;     // this wave's stage-0 operand: [piece][16][K0]
  constexpr int IPL = 16 * F::K0 * 2;                                      // bytes per piece plane
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  // resident fragments: 4 tiles x 2 + 2 = 40 VGPRs
  xh8 s1a[MT1], s1b[MT1];
#pragma unroll
  for (int x = 0; x < MT1; ++x) {
    s1a[x] = frag[(size_t)((wave * MT1 + x) * 2 + 0) * 64 + lane];
    s1b[x] = frag[(size_t)((wave * MT1 + x) * 2 + 1) * 64 + lane];
  }
  const xh8 g0a = frag[F::FRAG1 + lane], g0b = frag[F::FRAG1 + 64 + lane];
  // this lane's hidden unit: outputs o = i0 * I1 + i1 with i0 = 4 j + q (gate j), i1 = 16 wave + c  ->  unit (i0 & 3) * I1 + i1
  const int i1 = 16 * wave + c;
  const int hd = q * F::I1 + i1;
  f32x4 psc, usc;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int s = hdr[F::HP + 4 * j + q] + 11 - hdr[F::HU + i1];
    psc[j] = ldexpf(1.f, s);
    usc[j] = ldexpf(1.f, -s);
  }
  const float hsc = F10H_HSC;
  const float* __restrict__ gin = gs.gin;
  const float* __restrict__ xs = reinterpret_cast<const float*>(gs.x);
  constexpr bool in1 = IN1;
  float hst = H0 ? h0[b * H + hd] : 0.f;
  float cst = c0 ? c0[b * H + hd] : 0.f;
  float h0sc = 1.0f, h0un = 1.0f;
  if constexpr (H0) {
    const int e0 = f10h_h0_expo<NWV>(hst, reinterpret_cast<float*>(smem_f2 + 2 * 2 * H * 2), wave, lane);   // (scratch: wave 0's image)
    h0sc = ldexpf(1.f, -e0); h0un = ldexpf(1.f, e0);
  }
  f32x4 bh = f32x4{0.f, 0.f, 0.f, 0.f}, gi = bh, vv = bh, bb = bh;           // slot order i,g,f,o (gin / reserve layout)
  const f32x4 gsc = f32x4{-1.4426950408889634f, 2.8853900817779268f, -1.4426950408889634f, -1.4426950408889634f} *
                    f32x4{psc[0], psc[2], psc[1], psc[3]};
  XChunk<float> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
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
  {
    _Float16 p0, p1;                                       // parity 0 = h_{-1}
    split2h(hst * (hsc * h0sc), p0, p1);
    hpl[hd] = p0; hpl[H + hd] = p1;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): no fragment-register wait inside the loop
  lds_barrier();

  f32x4 us_t = usc * h0un;                                 // step 0 runs on 2^-e0 h_0; reset at the end of it
  float ps_t = h0sc;
  // stage-1 operand address of this lane: chain row j0 = c & 7 (columns 8 .. 15 of the tiles repeat 0 .. 7 and are not used),
  // j1 = 8 (q & 1) .. + 7 — the same eight values for k-groups q and q + 2 (the term packing)
  const int hoff = (c & 7) * F::J1 + 8 * (q & 1);
  for (int t = 0; t < T; ++t) {
    const _Float16* hp = hpl + (t & 1) * 2 * H;
    _Float16* hn = hpl + ((t + 1) & 1) * 2 * H;
    // ---- stage 1 --------------------------------------------------------------------------------------------------------------
    const xh8 x0 = *reinterpret_cast<const xh8*>(hp + hoff);
    const xh8 x1 = *reinterpret_cast<const xh8*>(hp + H + hoff);
    __builtin_amdgcn_sched_barrier(0);                     // both reads in flight before the first MFMA waits for one of them
    // tiles in pairs (small term first, then the leading ones on the same accumulator): a pair's results are split while the
    // next pair multiplies.  Lanes c >= 8 hold copies of lanes c - 8 (their operand column is chain row c & 7 again) and store
    // the same values to the same addresses: no branch around the split, so that it can be scheduled between the MFMAs
    f32x4 t1[MT1];
#pragma unroll
    for (int xp = 0; xp < MT1; xp += 2) {
#pragma unroll
      for (int x = xp; x < xp + 2; ++x) t1[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(s1b[x], x1, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int x = xp; x < xp + 2; ++x) t1[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(s1a[x], x0, t1[x], 0, 0, 0);
    }
    // hand-off inside the wave: lane (j0 = c & 7, q) of tile x holds C1'[j0][i1 = 16 wave + 4 x + q][a = 0 .. 3] -> row 4 x + q, k = 4 j0
#pragma unroll
    for (int x = 0; x < MT1; ++x) {
      unsigned a0, b0, a1, b1;
      split_pair_h(t1[x][0], t1[x][1], a0, b0);
      split_pair_h(t1[x][2], t1[x][3], a1, b1);
      const int off = f2_img_off(4 * x + q, (c & 7) >> 1, 8 * (c & 1));
      *reinterpret_cast<u32x2*>(img + off) = u32x2{a0, a1};
      *reinterpret_cast<u32x2*>(img + IPL + off) = u32x2{b0, b1};
    }
    const size_t bt = b * T + t;
    // (W_in x_t + b_in + b_hid) * scale, slots i,g,f,o -> accumulator rows i,f,g,o
    f32x4 pre = in1 ? bb + xq.at(t) * vv : (gi + bh) * gsc;
    if constexpr (H0) pre = pre * ps_t;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's own stores; nobody else reads these rows
    // ---- stage 0 --------------------------------------------------------------------------------------------------------------
    const int roff = f2_img_off(c, q, 0);
    const xh8 y0 = *reinterpret_cast<const xh8*>(img + roff);
    const xh8 y1 = *reinterpret_cast<const xh8*>(img + IPL + roff);
    f32x4 acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(g0b, y0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    f32x4 acc_hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(g0a, y0, f32x4{pre[0], pre[2], pre[1], pre[3]}, 0, 0, 0);
    acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(g0a, y1, acc_lo, 0, 0, 0);
    const f32x4 un = H0 ? us_t : usc;
    const f32x4 acc = acc_hi * un + acc_lo * un;           // exact powers of two
    // ---- gates + state (lstm.py:26-32) ------------------------------------------------------------------------------------------
    const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0]));
    const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[1]));
    const float gg = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[2]));
    const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[3]));
    const float cy = fg * cst + ig * gg;
    const float hy = og * ftanh(cy);
    cst = cy;
    hst = hy;
    {
      _Float16 p0, p1;
      split2h(hy * hsc, p0, p1);
      hn[hd] = p0; hn[H + hd] = p1;
    }
    if constexpr (OUT) out[bt * H + hd] = hy;              // outputs[:, t, :] (lstm.py:133)
    if (reserve) {
      *reinterpret_cast<f32x4*>(reserve + res_gate(bt, H, hd)) = f32x4{ig, gg, fg, og};
      reserve[res_cell((size_t)B * T, bt, H, hd)] = cy;
    }
    if (!in1 && t + 1 < T) gi = *reinterpret_cast<const f32x4*>(gin + ((bt + 1) * H + hd) * 4);
    if (in1) xq.advance(xs, b * T, T, t, lane);
    us_t = usc; ps_t = 1.0f;
    if constexpr (NWV > 1) lds_barrier();                  // h_t complete: the other wave's units are this wave's operand too
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (hT) hT[b * H + hd] = hst;
  if (cT) cT[b * H + hd] = cst;
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
template <class S>
static int launch_f2(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid, const void* bias_hid,
                     void* out, void* hT, void* cT, float* reserve, void* ws, hipStream_t stream, int phase) {
  using F = F2<S>;
  if (!ws) return TTRNN_ERR_WORKSPACE;
  int* hdr = reinterpret_cast<int*>(ws);
  xh8* frag = reinterpret_cast<xh8*>(reinterpret_cast<unsigned char*>(ws) + F::HDR_BYTES);
  // header and fragments depend on the weights only: TTRNN_PHASE_RUN finds them in ws (ttrnn_rnn_forward_phase)
  if (phase != TTRNN_PHASE_RUN) hipLaunchKernelGGL((k_f2_prep<S>), dim3(1), dim3(256), 0, stream, packed_hid, hdr, frag);
  if (phase == TTRNN_PHASE_PREPARE) return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  const float* bh = rs.has_bias_hid ? (const float*)bias_hid : (const float*)nullptr;
  auto kern = gin.in1 ? (out ? (h0 ? k_lstm_fwd_f2<S, true, true, true> : k_lstm_fwd_f2<S, false, true, true>)
                             : (h0 ? k_lstm_fwd_f2<S, true, false, true> : k_lstm_fwd_f2<S, false, false, true>))
                      : (out ? (h0 ? k_lstm_fwd_f2<S, true, true, false> : k_lstm_fwd_f2<S, false, true, false>)
                             : (h0 ? k_lstm_fwd_f2<S, true, false, false> : k_lstm_fwd_f2<S, false, false, false>));
  hipLaunchKernelGGL(kern, dim3(rs.B), dim3(F::NWV * 64), F::LDS_BYTES, stream, rs.B, rs.T, gin, (const float*)h0,
                     (const float*)c0, (const int*)hdr, (const xh8*)frag, bh, (float*)out, (float*)hT, (float*)cT, reserve);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  // ... followed by the fp32-MFMA stage-wise kernel as its fallback: exactly one of the two runs (decided on the device by
  // k_f2_prep's guard word; the one that steps aside returns at once)
  GinSrc fb = gin;
  fb.run_if = hdr + F::HGUARD;
  fb.status = device_status_ptr();
  return launch_rnn_fwd_fast(rs, TTRNN_F32, fb, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, stream);
}

bool f2_rnn_fwd_available(const RnnShape& rs, int dtype) {
  if (opt(OPT_NO_F10) || rs.B < 1 || rs.T < 1 || dtype != TTRNN_F32 || rs.cell != TTRNN_LSTM || rs.hid_blocks > 1) return false;
  return shape_matches<ShpH128R4L>(rs.hid_s);
}
size_t f2_workspace_bytes(const RnnShape& rs, int dtype) {
  if (dtype != TTRNN_F32 || rs.cell != TTRNN_LSTM) return 0;
  return shape_matches<ShpH128R4L>(rs.hid_s) ? F2<ShpH128R4L>::WS_BYTES : 0;
}
int launch_rnn_fwd_f2(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                      const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws, hipStream_t stream, int phase) {
  if (shape_matches<ShpH128R4L>(rs.hid_s))
    return launch_f2<ShpH128R4L>(rs, gin, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, ws, stream, phase);
  return TTRNN_ERR_UNSUPPORTED;
}


// =====================================================================================================================================
// Reverse-time kernel for the same two-core shapes (round 4): BPTT through tensorized_rnn/lstm.py:23-32,123-133 + the transposed
// chain of t3nsor/ops.py:81-90 for d = 2,
//     T0   dC1[(j0, a)][i1] = sum_i0      G0[i0, j0, a] dy[i0][i1]          (K = I0 = 16: the three split terms as two chained MFMAs)
//     T1   dh[j1][j0]       = sum_(i1, a) G1[a, i1, j1] dC1[(j0, a)][i1]    (K = I1 R1 = 128, split over the two waves' i1 ranges)
// on two fp16 pieces — the stage-wise fp32-MFMA kernel it replaces (k_rnn_bwd_fast, 2 400 cycles per step) was 54 % of cfg1's
// training step.  Same ownership as the forward kernel: wave w holds the units (q, i1 = 16 w + c), so T0 — which contracts over
// the gate rows i0 of ONE column i1 — is wave-local: the gate gradients go to the operand image under the WAVE's own scale
// (max |dg_t| over its 64 units: four DPP rotations + readlanes, no barrier) and come back as the B operand through LDS inside the
// wave; T0's results, rescaled by the bound max_m L1(G0 row m) max|dg| (cannot overflow), feed T1 the same way; T1 sums over the
// wave's own 16 values of i1 and leaves a partial dh per wave in LDS: ONE barrier per step.  Weights: one power-of-two scale per
// output ROW of each stage (undone on the accumulators; no guard needed: a row's entries share its scale by construction).
// =====================================================================================================================================
template <class S>
struct F2B {
  using F = F2<S>;
  static constexpr int M0 = F::J0 * F::R1;                 // T0 rows (j0, a): 32
  static constexpr int MT0 = M0 / 16;                      // 2 tiles
  static constexpr int K1 = F::I1 * F::R1, NKB1 = K1 / 32; // T1 contraction 128 = 4 k-blocks, two per wave
  static constexpr int UN0 = 0, UN1 = M0, ML1 = M0 + 16, HDR_FLOATS = 64;
  static constexpr size_t FRAG0 = (size_t)MT0 * 2 * 64, FRAG1 = (size_t)NKB1 * 2 * 64;
  static constexpr size_t WS_BYTES = HDR_FLOATS * sizeof(float) + (FRAG0 + FRAG1) * sizeof(xh8);
  // LDS: dh partials [parity][wave][H] fp32 | per wave: dy image [2 pieces][16][16] fp16, dC1 image [2 pieces][J0][64] fp16
  static constexpr int DYI = 2 * 16 * 16 * 2, DCI = 2 * F::J0 * 64 * 2;
  static constexpr size_t LDS_BYTES = 2 * F::NWV * F::H * sizeof(float) + (size_t)F::NWV * (DYI + DCI);
};
template <class S>
constexpr bool f2b_ok() { return f2_ok<S>() && F2<S>::NWV == 2 && F2B<S>::M0 == 32 && F2B<S>::NKB1 == 4 && F2<S>::I0 == 16; }

template <class S>
__global__ void __launch_bounds__(256) k_f2b_prep(const float* __restrict__ packed, float* __restrict__ hdr, xh8* __restrict__ frag) {
  using F = F2<S>;
  using B = F2B<S>;
  constexpr int M1 = F::I1 * F::R1;
  __shared__ float g1[F::J1 * M1];                        // W1[j1][(i1, a)]
  __shared__ float g0[F::K0 * F::I0];                     // W0[(j0, a)][i0]
  __shared__ float s0[B::M0], s1[F::J1], l1[B::M0];
  const int tid = threadIdx.x;
  const float* W0 = packed + woff_of<S>(0);
  const float* W1 = packed + woff_of<S>(1);
  for (int i = tid; i < F::J1 * M1; i += 256) g1[i] = W1[i];
  for (int i = tid; i < F::K0 * F::I0; i += 256) g0[i] = W0[i];
  __syncthreads();
  if (tid < B::M0) {                                      // rows of T0: one scale per (j0, a), row maximum -> [2^13, 2^14)
    float mx = 0.f, l = 0.f;
    for (int i0 = 0; i0 < F::I0; ++i0) { const float v = fabsf(g0[tid * F::I0 + i0]); mx = fmaxf(mx, v); l += v; }
    s0[tid] = ldexpf(1.f, 14 - f10h_expo(mx));
    l1[tid] = l;
  } else if (tid >= 64 && tid < 64 + F::J1) {             // rows of T1: one scale per j1
    const int j1 = tid - 64;
    float mx = 0.f;
    for (int k = 0; k < M1; ++k) mx = fmaxf(mx, fabsf(g1[j1 * M1 + k]));
    s1[j1] = ldexpf(1.f, 14 - f10h_expo(mx));
  }
  __syncthreads();
  if (tid < B::M0) hdr[B::UN0 + tid] = 1.0f / s0[tid];
  if (tid < F::J1) hdr[B::UN1 + tid] = 1.0f / s1[tid];
  if (tid == 0) {
    float m = 0.f;
    for (int i = 0; i < B::M0; ++i) m = fmaxf(m, l1[i]);
    hdr[B::ML1] = m;                                       // |dC1[m][.]| <= L1(row m) max|dy|
  }
  // T0 fragments: tile mt, lane (r, q): row m = 16 mt + r; k = 8 q + e; operand order k' = 4 q_u + g <-> i0 = 4 g + q_u (the gate
  // thread (c, q_u) writes its four gate gradients as four consecutive k').  MFMA 1: k < 16 piece 0 of k' = k, else piece 1 of
  // k' = k - 16;  MFMA 2: k < 16 piece 0, else 0
  for (int idx = tid; idx < B::MT0 * 64; idx += 256) {
    const int mt = idx >> 6, lane = idx & 63, r = lane & 15, q = lane >> 4;
    const int m = 16 * mt + r;
    xh8 f1, f2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kp = (8 * q + e) & 15, i0 = 4 * (kp & 3) + (kp >> 2);
      _Float16 p0, p1;
      split2h(g0[m * F::I0 + i0] * s0[m], p0, p1);
      f1[e] = q < 2 ? p0 : p1;
      f2[e] = q < 2 ? p0 : (_Float16)0.f;
    }
    frag[(size_t)(mt * 2 + 0) * 64 + lane] = f1;
    frag[(size_t)(mt * 2 + 1) * 64 + lane] = f2;
  }
  // T1 fragments: k-block u, lane (r = j1, q): k = 32 u + 8 q + e = i1 * R1 + a (natural order)
  for (int idx = tid; idx < B::NKB1 * 64; idx += 256) {
    const int u = idx >> 6, lane = idx & 63, r = lane & 15, q = lane >> 4;
    xh8 a0, a1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 p0, p1;
      split2h(g1[r * M1 + 32 * u + 8 * q + e] * s1[r], p0, p1);
      a0[e] = p0; a1[e] = p1;
    }
    frag[B::FRAG0 + (size_t)(u * 2 + 0) * 64 + lane] = a0;
    frag[B::FRAG0 + (size_t)(u * 2 + 1) * 64 + lane] = a1;
  }
}

// x < 2^e from the exponent bits (as ttrnn_fast_f10bh.hip: step_scale): returns 2^(14 - e), un = 2^(e - 14); zero -> zeros stay zeros
__device__ __forceinline__ float f2b_step_scale(float mx, float& un) {
  int eb = (int)(__float_as_uint(mx) >> 23);
  eb = eb < 27 ? 27 : (eb > 227 ? 227 : eb);
  un = __uint_as_float((unsigned)(eb - 13) << 23);
  return __uint_as_float((unsigned)(267 - eb) << 23);
}
template <int N>
__device__ __forceinline__ float f2b_row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float f2b_wave_max(float v) {
  v = fmaxf(v, f2b_row_ror<1>(v));
  v = fmaxf(v, f2b_row_ror<2>(v));
  v = fmaxf(v, f2b_row_ror<4>(v));
  v = fmaxf(v, f2b_row_ror<8>(v));
  const int i = __float_as_int(v);
  const float a = __int_as_float(__builtin_amdgcn_readlane(i, 0)), b = __int_as_float(__builtin_amdgcn_readlane(i, 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(i, 32)), d = __int_as_float(__builtin_amdgcn_readlane(i, 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

template <class S>
__global__ void __launch_bounds__(F2<S>::NWV * 64, 4) k_lstm_bwd_f2(int Bn, int T, const float* __restrict__ c0,
                                                                    const float* __restrict__ hdr, const xh8* __restrict__ frag,
                                                                    const float* __restrict__ reserve,
                                                                    const float* __restrict__ d_out, const float* __restrict__ d_hT,
                                                                    const float* __restrict__ d_cT, float* __restrict__ dg_in,
                                                                    float* __restrict__ dg_hid, float* __restrict__ d_h0,
                                                                    float* __restrict__ d_c0) {
  static_assert(f2b_ok<S>(), "shape not supported by the two-core reverse-time kernel");
  using F = F2<S>;
  using B = F2B<S>;
  constexpr int H = F::H, GH = 4 * H;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b2[];
  float* dhs = reinterpret_cast<float*>(smem_b2);                                       // [parity][wave][H]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* dyi = smem_b2 + 2 * F::NWV * H * sizeof(float) + (size_t)wave * (B::DYI + B::DCI);   // [piece][16 rows i1][16 k'] fp16
  unsigned char* dci = dyi + B::DYI;                                                     // [piece][J0 rows][64 k] fp16
  constexpr int DYP = 16 * 16 * 2, DCP = F::J0 * 64 * 2;                                 // bytes per piece plane
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;
  const int hid = q * F::I1 + 16 * wave + c;

  xh8 t0a[B::MT0], t0b[B::MT0], t1w[2][2];
  f32x4 un0[B::MT0];
#pragma unroll
  for (int mt = 0; mt < B::MT0; ++mt) {
    t0a[mt] = frag[(size_t)(mt * 2 + 0) * 64 + lane];
    t0b[mt] = frag[(size_t)(mt * 2 + 1) * 64 + lane];
    un0[mt] = *reinterpret_cast<const f32x4*>(hdr + B::UN0 + 16 * mt + 4 * q);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int p = 0; p < 2; ++p) t1w[u][p] = frag[B::FRAG0 + (size_t)((2 * wave + u) * 2 + p) * 64 + lane];
  const f32x4 un1 = *reinterpret_cast<const f32x4*>(hdr + B::UN1 + 4 * q);
  const float maxl1 = hdr[B::ML1];

  float dcs = d_cT ? d_cT[b * H + hid] : 0.f;
  const float c0v = c0 ? c0[b * H + hid] : 0.f;
  const float* dptr = d_out ? d_out : reserve;            // a null d_out reads the reserve and is scaled by zero: no branch in the loop
  const float dscale = d_out ? 1.0f : 0.0f;
  f32x4 ra0 = f32x4{0.f, 0.f, 0.f, 0.f}, ra1 = ra0, ra2 = ra0;
  float rb0 = 0.f, rb1 = 0.f, rb2 = 0.f, do0 = 0.f, do1 = 0.f, do2 = 0.f;
  dhs[0 * H + hid] = d_hT ? d_hT[b * H + hid] : 0.f;      // parity 0: [wave 0 | wave 1] partials = (d_hT, 0)
  dhs[1 * H + hid] = 0.f;
  if (T > 0) {
    const size_t bt = b * T + (T - 1);
    const float* rv = reserve + res_gate(bt, H, hid);
    const float* rc = reserve + res_cell((size_t)Bn * T, bt, H, hid);
    ra0 = *reinterpret_cast<const f32x4*>(rv);
    rb0 = rc[0];
    do0 = dptr[bt * H + hid];
    if (T > 1) {
      ra1 = *reinterpret_cast<const f32x4*>(rv - H * 4);
      rb1 = rc[-H];
      do1 = dptr[(bt - 1) * H + hid];
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): no fragment-register wait inside the loop
  lds_barrier();

  const bool dup = dg_hid && dg_hid != dg_in;
  int par = 0;
  // three rotating register sets for the saved records (loads requested two steps ahead, never consumed or overwritten in the
  // phase that issued them: DESIGN_HISTORY.md lesson 9)
  auto step = [&](const int t, const f32x4& ra, const float& rb, const float& dout_c, const float& nb, f32x4& fa, float& fb,
                  float& dout_f) {
    const size_t bt = b * T + t;
    const float* dh_in = dhs + par * F::NWV * H;
    float* dh_out = dhs + (par ^ 1) * F::NWV * H + wave * H;
    // ---- gate gradients (lstm.py:26-32 differentiated; reserve slots i, g, f, o) -------------------------------------------------
    const float dht = dout_c * dscale + dh_in[hid] + dh_in[H + hid];
    const float ig = ra[0], gg = ra[1], fg = ra[2], og = ra[3], cy = rb;
    const float cprev = t > 0 ? nb : c0v;
    const float tc = ftanh(cy);
    const float dct = dcs + dht * og * (1.0f - tc * tc);
    const float p0 = dct * gg * ig * (1.0f - ig);                 // d pre-activation of i
    const float p1 = dct * cprev * fg * (1.0f - fg);              //                     f
    const float p2 = dct * ig * (1.0f - gg * gg);                 //                     g
    const float p3 = dht * tc * og * (1.0f - og);                 //                     o
    dcs = dct * fg;
    const float mx = f2b_wave_max(fmaxf(fmaxf(fabsf(p0), fabsf(p1)), fmaxf(fabsf(p2), fabsf(p3))));
    float ug, u2;
    const float sg = f2b_step_scale(mx, ug);
    const float s2 = f2b_step_scale(mx * maxl1, u2);              // |dC1| <= maxl1 * max|dg|: no overflow, whatever the signs
    // T0's operand: column i1 = c of this wave, k' = 4 q + gate: one 8-byte store per piece, read back by this wave only
    {
      unsigned a0, b0, a1, b1;
      split_pair_h(p0 * sg, p1 * sg, a0, b0);
      split_pair_h(p2 * sg, p3 * sg, a1, b1);
      *reinterpret_cast<u32x2*>(dyi + c * 32 + 8 * q) = u32x2{a0, a1};
      *reinterpret_cast<u32x2*>(dyi + DYP + c * 32 + 8 * q) = u32x2{b0, b1};
    }
    // d_gates rows for the weight gradients (gate-major, as the reference's pre-activation order i, f, g, o)
    float* gr = dg_in + bt * GH + hid;
    gr[0] = p0; gr[H] = p1; gr[2 * H] = p2; gr[3 * H] = p3;
    if (dup) { float* g2 = dg_hid + bt * GH + hid; g2[0] = p0; g2[H] = p1; g2[2 * H] = p2; g2[3 * H] = p3; }
    // record(t-2), d_out(t-2): always three loads, index clamped
    {
      const size_t b2 = t > 1 ? bt - 2 : b * T;
      fa = *reinterpret_cast<const f32x4*>(reserve + res_gate(b2, H, hid));
      fb = reserve[res_cell((size_t)Bn * T, b2, H, hid)];
      dout_f = dptr[b2 * H + hid];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- T0 -------------------------------------------------------------------------------------------------------------------
    const xh8 y0 = *reinterpret_cast<const xh8*>(dyi + c * 32 + 16 * (q & 1));
    const xh8 y1 = *reinterpret_cast<const xh8*>(dyi + DYP + c * 32 + 16 * (q & 1));
    __builtin_amdgcn_sched_barrier(0);
    f32x4 a0t[B::MT0];
#pragma unroll
    for (int mt = 0; mt < B::MT0; ++mt) a0t[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t0b[mt], y1, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < B::MT0; ++mt) a0t[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t0a[mt], y0, a0t[mt], 0, 0, 0);
    // lane (c = i1, q) of tile mt: dC1[(j0 = 4 mt + q, a = 0 .. 3)][i1] -> T1's operand row j0, k = 4 c + a (16-byte slots swizzled by row)
    const float t01f = ug * s2;
#pragma unroll
    for (int mt = 0; mt < B::MT0; ++mt) {
      const f32x4 v = a0t[mt] * (un0[mt] * t01f);
      unsigned a0, b0, a1, b1;
      split_pair_h(v[0], v[1], a0, b0);
      split_pair_h(v[2], v[3], a1, b1);
      const int row = 4 * mt + q;
      const int off = row * 128 + (((c >> 1) ^ (row & 7)) << 4) + 8 * (c & 1);
      *reinterpret_cast<u32x2*>(dci + off) = u32x2{a0, a1};
      *reinterpret_cast<u32x2*>(dci + DCP + off) = u32x2{b0, b1};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- T1: this wave's 64 of the 128 k values; columns j0 = c & 7 (columns 8 .. 15 repeat them) ----------------------------------
    const int row1 = c & 7;
    f32x4 alo = f32x4{0.f, 0.f, 0.f, 0.f}, ahi = alo;
    xh8 z0[2], z1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {                                   // all four operand reads in flight before the first MFMA
      const int off = row1 * 128 + (((4 * u + q) ^ row1) << 4);
      z0[u] = *reinterpret_cast<const xh8*>(dci + off);
      z1[u] = *reinterpret_cast<const xh8*>(dci + DCP + off);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(t1w[u][1], z0[u], alo, 0, 0, 0);
      ahi = __builtin_amdgcn_mfma_f32_16x16x32_f16(t1w[u][0], z0[u], ahi, 0, 0, 0);
      alo = __builtin_amdgcn_mfma_f32_16x16x32_f16(t1w[u][0], z1[u], alo, 0, 0, 0);
    }
    // lane (c = j0, q): dh[j0 * J1 + 4 q + j], j = 0 .. 3 (this wave's share of the sum over i1)
    *reinterpret_cast<f32x4*>(dh_out + row1 * F::J1 + 4 * q) = (ahi + alo) * (un1 * u2);
    par ^= 1;
    lds_barrier();
  };
  for (int t = T - 1; t >= 0; t -= 3) {
    step(t, ra0, rb0, do0, rb1, ra2, rb2, do2);
    if (t >= 1) step(t - 1, ra1, rb1, do1, rb2, ra0, rb0, do0);
    if (t >= 2) step(t - 2, ra2, rb2, do2, rb0, ra1, rb1, do1);
  }
  const float* dh_fin = dhs + par * F::NWV * H;
  if (d_h0) d_h0[b * H + hid] = dh_fin[hid] + dh_fin[H + hid];
  if (d_c0) d_c0[b * H + hid] = dcs;
}

bool f2_rnn_bwd_available(const RnnShape& rs, int dtype) {
  if (opt(OPT_NO_F10) || rs.B < 1 || rs.T < 1 || dtype != TTRNN_F32 || rs.cell != TTRNN_LSTM || rs.hid_blocks > 1) return false;
  return shape_matches<ShpH128R4L>(rs.hid_s);
}
size_t f2_rnn_bwd_workspace_bytes(const RnnShape& rs, int dtype) {
  return f2_rnn_bwd_available(rs, dtype) ? F2B<ShpH128R4L>::WS_BYTES : 0;
}
int launch_rnn_bwd_f2(const RnnShape& rs, const void* c0, const float* packed_hid, const float* reserve, const void* d_out,
                      const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid, void* d_h0, void* d_c0, void* ws,
                      hipStream_t stream) {
  using S = ShpH128R4L;
  using B = F2B<S>;
  if (!shape_matches<S>(rs.hid_s)) return TTRNN_ERR_UNSUPPORTED;
  if (!ws) return TTRNN_ERR_WORKSPACE;
  float* hdr = reinterpret_cast<float*>(ws);
  xh8* frag = reinterpret_cast<xh8*>(reinterpret_cast<unsigned char*>(ws) + B::HDR_FLOATS * sizeof(float));
  hipLaunchKernelGGL((k_f2b_prep<S>), dim3(1), dim3(256), 0, stream, packed_hid, hdr, frag);
  hipLaunchKernelGGL((k_lstm_bwd_f2<S>), dim3(rs.B), dim3(F2<S>::NWV * 64), B::LDS_BYTES, stream, rs.B, rs.T, (const float*)c0,
                     (const float*)hdr, (const xh8*)frag, reserve, (const float*)d_out, (const float*)d_hT, (const float*)d_cT, dg_in,
                     dg_hid, (float*)d_h0, (float*)d_c0);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

}  // namespace ttrnn
