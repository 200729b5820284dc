// ttrnn_fast_bigh.hip — cfg5-class K-rec (H = 1024, d = 4, r = 32; merged two-core matrix, two workgroups per sample)
// on two-piece fp16 operands (ttrnn_split.h, flavour (b)): both stages of the chain on v_mfma_f32_16x16x32_f16, three
// terms per product, instead of the fp32 MFMA of ttrnn_fast_big.hip:k_lstm_fwd_big2 (16.4 k of its 26 k cycles per step
// were matrix pipe).  Same pair structure and the same tagged-word exchange of h; what changes:
//   * stage 1 (16 rows x K = 64 x 1024 features per workgroup): the h image is two fp16 planes of 2^9 h; the core
//     fragments (two pieces of 2^a W_1, 256 KB per workgroup) are RESIDENT: three quarters in registers (96 VGPRs per
//     wave), the last quarter in LDS (64 KB) — the first version streamed them from L2 through two register slots (8.3 ms
//     per forward against 7.6: with all 256 CUs streaming, L2 delivered ~ 26 B/clk per CU);
//   * its sums are rescaled by a fixed power of two (< 2^15) and split into the two fp16 planes of the stage-0 image;
//   * stage 0 (32 local rows x K = 512 x 64 features): the fragments of 2^b W_0 are resident in registers too (64 VGPRs:
//     wave = (feature tile p, k half), both row tiles), feature rows permuted so that a lane's four accumulator
//     registers are the four gates of ONE hidden unit: the gates run in the lanes that hold the sums, after the two k
//     halves have been exchanged through LDS (wave kh keeps row tile kh, hands over row tile 1 - kh);
//   * the k order of the stage-0 operand is permuted ((a / 4, j0, a % 4) instead of (j0, a)) so that a wave's stage-1
//     store is one contiguous 512-byte run per plane.
// Scales as in the runtime-shape tier (ttrnn_g2.hip:g2_scales), all powers of two from the maxima of the merged cores;
// a caller's h_0 outside (-1, 1) is scaled per sample (ttrnn_f10_dev.h:f10h_h0_expo).
// Replaces (together with the batched input projection) the reference's per-step chain + gates for this shape:
// tensorized_rnn/lstm.py:26-32 and 77-92 (cell and sequence loop), t3nsor/ops.py:80-92 via tt_linear.py:159-160.
#include <hip/hip_runtime.h>
#include "ttrnn.h"
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_f10_dev.h"
#include "ttrnn_big.h"

namespace ttrnn {

namespace {

using S2 = ShpH1024R32L_M2;
using T1 = St<S2, 1>;
using T0 = St<S2, 0>;
constexpr int BH_H = 1024, BH_I1 = 64, BH_HR = 32, BH_K1 = 64, BH_K0 = 512, BH_R1 = 32;
constexpr int BH_PL_H = 16 * BH_K1;            // halfs per plane of the h image  [16 rows j0][64 j1]
constexpr int BH_PL_I = BH_HR * BH_K0;         // halfs per plane of the stage-0 image [32 local rows i1][512 k']
static_assert(T1::K == BH_K1 && T1::M == BH_I1 * BH_R1 && T1::ROWS == 16 && T0::K == BH_K0 && T0::M == 64 &&
                  T0::ROWS == BH_I1 && in_size_of<S2>() == BH_H && out_size_of<S2>() == 4 * BH_H && FAST_NT == 512,
              "merged two-core shape of cfg5");

// element (kk, m) of stage k inside the fragment-ordered fp32 buffer k_merge_cores_last writes (ttrnn_big.h:frag_decode)
template <int k>
__device__ __forceinline__ int frag_index(int kk, int m) {
  using T = St<S2, k>;
  return (((m >> 4) * T::NU + (kk >> 4)) * 64 + ((kk >> 2) & 3) * 16 + (m & 15)) * 4 + (kk & 3);
}

// k order of the stage-0 operand: (j0, a) -> k' = (a / 4) * 64 + j0 * 4 + a % 4 (k_bigh_prep inverts it for W_0's fragments, the
// stage-1 store of the kernel produces it)

// Diagonal power-of-two scales (as ttrnn_f10_dev.h / ttrnn_g2.hip:k_g2_diag), int32 exponents in the scratch header:
//   tail   W_1'[j_t][(i_t, a)] = W_1 2^(13 + eu[i_t] + ev[a])      each i_t block and each rank slice a: max < 2^13
//   h      2^9 h;  stage-1 sums < 64 * 2^22 = 2^28, times 2^-13 -> < 2^15 before the split
//   head   W_0'[(j_0, a)][i_h] = W_0 2^(ep[i_h] - ev[a])           each output row i_h: max < 2^14
//   sums   2^(13 + 9 - 13 + ep[i_h] + eu[i_t]) x the pre-activation of (i_h, i_t)
// so one large core entry moves only the scale of its own row / slice (round 2: one scale per merged core).
constexpr int BH_EU = 0, BH_EV = 64, BH_EP = 96, BH_PART = 160, BH_HDR_BYTES = 16384;      // BH_PART: [I_t][R_1] partial maxima (floats)
constexpr float BH_HSC = 512.0f, BH_R1SC = 1.0f / 8192.0f;

// two launches over the merged cores (fragment-ordered fp32, L2-resident), as ttrnn_g2.hip:k_g2_diag_a / _b:
//   k_bigh_diag_a  one workgroup per i_t: eu[i_t], part[i_t][a] = max_{j_t} 2^eu |W_1|
//   k_bigh_diag_b  one workgroup per i_h: ev (reduced by every workgroup, stored by the first), ep[i_h]
__global__ void __launch_bounds__(256) k_bigh_diag_a(const float* __restrict__ packed2, int* __restrict__ hdr) {
  __shared__ float red[4];
  __shared__ unsigned mv[BH_R1];
  __shared__ int eus;
  const int tid = threadIdx.x, it = blockIdx.x;
  const float* W1 = packed2 + woff_of<S2>(1);
  float* part = reinterpret_cast<float*>(hdr) + BH_PART;
  if (tid < BH_R1) mv[tid] = 0u;
  float v[8];                                    // 64 (j_t) x 32 (a) entries of this i_t: 8 per thread, a = tid % 32
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = fabsf(W1[frag_index<1>((tid >> 5) + 8 * i, it * BH_R1 + (tid & 31))]);
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) mx = fmaxf(mx, v[i]);
  float ma = mx;                                 // this thread's maximum for its a
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    eus = -f10h_expo(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    hdr[BH_EU + it] = eus;
  }
  __syncthreads();
  atomicMax(&mv[tid & 31], __float_as_uint(ma * ldexpf(1.f, eus)));
  __syncthreads();
  if (tid < BH_R1) part[it * BH_R1 + tid] = __uint_as_float(mv[tid]);
}
__global__ void __launch_bounds__(256) k_bigh_diag_b(const float* __restrict__ packed2, int* __restrict__ hdr) {
  __shared__ float red[4];
  __shared__ float pv[8][BH_R1];
  __shared__ int ev[BH_R1];
  const int tid = threadIdx.x, ih = blockIdx.x;
  const float* W0 = packed2 + woff_of<S2>(0);
  const float* part = reinterpret_cast<const float*>(hdr) + BH_PART;
  {
    const int a = tid & 31, g = tid >> 5;
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < BH_I1 / 8; ++i) mx = fmaxf(mx, part[(g + 8 * i) * BH_R1 + a]);
    pv[g][a] = mx;
  }
  __syncthreads();
  if (tid < BH_R1) {
    float mx = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) mx = fmaxf(mx, pv[g][tid]);
    ev[tid] = -f10h_expo(mx);
    if (ih == 0) hdr[BH_EV + tid] = ev[tid];
  }
  __syncthreads();
  float mx = 0.f;                                // row i_h of W_0: 512 entries kk = j0 * R1 + a, two per thread
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int kk = tid + 256 * i;
    mx = fmaxf(mx, fabsf(W0[frag_index<0>(kk, ih)]) * ldexpf(1.f, -ev[kk % BH_R1]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) hdr[BH_EP + ih] = 14 - f10h_expo(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// fragment streams (16-byte entries, one per lane):
//   f1[half][wave][x][kb][piece][lane]   stage 1: m-tile mtl = 8 wave + x of the half, k-block kb (32 values of j1)
//   f0[wave][kbl][piece][lane]           stage 0: wave = (p, kh): feature tile p, k-block 8 kh + kbl (32 values of k')
// stage-0 rows: row 4 q' + j of tile p <-> feature m0 = 16 j + 4 p + q' (gate j of unit-column mq = 4 p + q')
constexpr int BH_F1 = 2 * 8 * 8 * 2 * 2 * 64, BH_F0 = 8 * 8 * 2 * 64;
__global__ void __launch_bounds__(256) k_bigh_prep(const float* __restrict__ packed2, const int* __restrict__ hdr,
                                                   xh8* __restrict__ f1, xh8* __restrict__ f0) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int lane = e & 63, r = lane & 15, q = lane >> 4;
  xh8 p0, p1;
  if (e < BH_F1 / 2) {
    const int kb = (e >> 6) & 1, x = (e >> 7) & 7, wave = (e >> 10) & 7, half = e >> 13;
    const int m = (half * 64 + 8 * wave + x) * 16 + r;
    const float* W = packed2 + woff_of<S2>(1);
    const float tsc = ldexpf(1.f, 13 + hdr[BH_EU + m / BH_R1] + hdr[BH_EV + m % BH_R1]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      _Float16 a, b;
      split2h(W[frag_index<1>(32 * kb + 8 * q + i, m)] * tsc, a, b);
      p0[i] = a; p1[i] = b;
    }
    const size_t o = ((size_t)(e >> 6) * 2) * 64 + lane;
    f1[o] = p0;
    f1[o + 64] = p1;
  } else if (e < BH_F1 / 2 + BH_F0 / 2) {
    const int g = e - BH_F1 / 2;
    const int kbl = (g >> 6) & 7, wave = g >> 9;
    const int p = wave & 3, kh = wave >> 2;
    const int m0 = 16 * (r & 3) + 4 * p + (r >> 2);
    const float* W = packed2 + woff_of<S2>(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kp = 32 * (8 * kh + kbl) + 8 * q + i;                 // permuted k' -> (j0, a)
      const int j0 = (kp & 63) >> 2, a = (kp >> 6) * 4 + (kp & 3);
      _Float16 u, v;
      split2h(W[frag_index<0>(j0 * BH_R1 + a, m0)] * ldexpf(1.f, hdr[BH_EP + m0] - hdr[BH_EV + a]), u, v);
      p0[i] = u; p1[i] = v;
    }
    const size_t o = ((size_t)(g >> 6) * 2) * 64 + lane;
    f0[o] = p0;
    f0[o + 64] = p1;
  }
}

// OUT = false: the caller consumes only the final state (mnist_classifier.py:52-55): the [B][T][H] store is skipped
template <typename TS, bool OUT = true>
__global__ void __launch_bounds__(FAST_NT) k_lstm_fwd_big2h(int B, int T, const float* __restrict__ gin,
                                                            const TS* __restrict__ h0, const TS* __restrict__ c0,
                                                            const xh8* __restrict__ f1, const xh8* __restrict__ f0,
                                                            const int* __restrict__ hdr,
                                                            const TS* __restrict__ bias_in,
                                                            const TS* __restrict__ bias_hid, TS* __restrict__ out,
                                                            TS* __restrict__ hT, TS* __restrict__ cT,
                                                            float* __restrict__ reserve,
                                                            unsigned long long* __restrict__ hx,
                                                            unsigned* __restrict__ status) {
  constexpr int H = BH_H;
  __shared__ __attribute__((aligned(16))) _Float16 hpl[2 * BH_PL_H];      // h_{t-1}: two planes of 2^9 h
  __shared__ __attribute__((aligned(16))) f32x4 xp[4][2][64];             // stage-0 sums of the OTHER k half's row tile
  __shared__ float scr[8];
  extern __shared__ __attribute__((aligned(16))) float big_lds[];         // stage-0 image: two planes [32][512]
  _Float16* img = reinterpret_cast<_Float16*>(big_lds);
  xh8* w1l = reinterpret_cast<xh8*>(big_lds) + 2 * BH_PL_I / 8;           // chunk 3 of every wave's stage-1 fragments (64 KB)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int p = wave & 3, kh = wave >> 2;
  const size_t b = blockIdx.x >> 1;
  const int half = blockIdx.x & 1;
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gin);

  // gate phase: lane (c, q) of wave (p, kh) owns unit-column mq = 4 p + q, local row 16 kh + c
  const int mq = 4 * p + q, rl = 16 * kh + c;
  // inverse scales of the stage-0 sums of this lane's unit: gate j is output row i_h = 16 j + mq, column i_t = 32 half + rl
  f32x4 un4;
  {
    const int eut = hdr[BH_EU + half * BH_HR + rl];
#pragma unroll
    for (int j = 0; j < 4; ++j) un4[j] = ldexpf(1.f, -(13 + 9 - 13 + hdr[BH_EP + 16 * j + mq] + eut));
  }
  const int hid = mq * BH_I1 + half * BH_HR + rl;
  const int hidp = mq * BH_I1 + (1 - half) * BH_HR + rl;                  // the partner's unit at the same position
  const int ho = x_off<BH_K1>(mq, half * BH_HR + rl), hop = x_off<BH_K1>(mq, (1 - half) * BH_HR + rl);
  float hst = h0 ? ld(h0, b * H + hid) : 0.f;
  float cst = c0 ? ld(c0, b * H + hid) : 0.f;
  const float hpv = h0 ? ld(h0, b * H + hidp) : 0.f;
  float bh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
    bh[g] = (bias_hid ? ld(bias_hid, g * H + hid) : 0.f) + (bias_in ? ld(bias_in, g * H + hid) : 0.f);
  // h_0 outside (-1, 1): per sample, the image holds 2^-e0 h_0 and the first step's sums are multiplied back
  const int e0 = h0 ? f10h_h0_expo<FAST_NW>(fmaxf(fabsf(hst), fabsf(hpv)), scr, wave, lane) : 0;
  {
    const float s0 = ldexpf(BH_HSC, -e0);
    _Float16 u, v;
    split2h(hst * s0, u, v);
    hpl[ho] = u; hpl[BH_PL_H + ho] = v;
    split2h(hpv * s0, u, v);
    hpl[hop] = u; hpl[BH_PL_H + hop] = v;
  }
  float e0f = ldexpf(1.f, e0);                        // step 0 runs on 2^-e0 h_0: its sums are multiplied back
  f32x4 gi = T > 0 ? gin4[(b * T) * H + hid] : f32x4{0.f, 0.f, 0.f, 0.f};     // slots i,g,f,o; prefetched a step ahead
  bool dead = false;

  // resident stage-0 fragments
  xh8 w0f[8][2];
#pragma unroll
  for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) w0f[u][pc] = f0[((size_t)(wave * 8 + u) * 2 + pc) * 64 + lane];

  // stage-1 fragments: chunk j = m-tiles x = 2 j, 2 j + 1 of this wave; entry ((x * 2 + kb) * 2 + piece) * 64 + lane.
  // Chunks 0..2 are resident in registers (96 VGPRs), chunk 3 in LDS: nothing is streamed during the time loop.
  const xh8* f1w = f1 + (size_t)(half * 8 + wave) * (8 * 2 * 2 * 64);
  xh8 w1r[3][2][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int i = 0; i < 4; ++i) w1r[j][y][i] = f1w[(size_t)((2 * j + y) * 4 + i) * 64 + lane];
  xh8 hb[2][2];
  // two m-tiles (one local row of the stage-0 image: a = 0..15 and 16..31), their sums split into the image
  int cz = c, qz = q;                      // the lane's (c, q) behind the per-step opaque id: addresses are recomputed, not hoisted
  auto s1 = [&](const xh8 (&s)[2][4], int j) {
    f32x4 lo[2], hi[2];
#pragma unroll
    for (int y = 0; y < 2; ++y) { lo[y] = f32x4{0.f, 0.f, 0.f, 0.f}; hi[y] = lo[y]; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        lo[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(s[y][2 * kb + 1], hb[kb][0], lo[y], 0, 0, 0);
        lo[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(s[y][2 * kb], hb[kb][1], lo[y], 0, 0, 0);
        hi[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(s[y][2 * kb], hb[kb][0], hi[y], 0, 0, 0);
      }
    // x_off<512>(row = 4 wave + j, k' = (4 y + q) * 64 + 4 c): the XOR touches the low four slot bits only
    const int row = 4 * wave + j;
    const int o0 = ((row * 64 + (q >> 1) * 16 + ((((qz & 1) << 3) + (cz >> 1)) ^ (row & 15))) << 3) + ((cz & 1) << 2);
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const f32x4 v = (hi[y] + lo[y]) * BH_R1SC;
      store_split4_h(img, BH_PL_I, o0 + y * 256, v);
    }
  };
#pragma unroll
  for (int e = 0; e < 8; ++e) w1l[(wave * 8 + e) * 64 + lane] = f1w[(size_t)(3 * 8 + e) * 64 + lane];
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    int z = 0;
    asm volatile("" : "+v"(z));            // per-step opaque lane id (cz, qz below)
    cz = (lane + z) & 15;
    qz = (lane + z) >> 4;
    // ---- stage 1 ----------------------------------------------------------------------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc)
        hb[kb][pc] = *reinterpret_cast<const xh8*>(hpl + pc * BH_PL_H + x_off<BH_K1>(cz, 32 * kb + 8 * qz));
    s1(w1r[0], 0);
    s1(w1r[1], 1);
    {
      xh8 tl[2][4];
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int i = 0; i < 4; ++i) tl[y][i] = w1l[(wave * 8 + y * 4 + i) * 64 + lane];
      s1(tl, 3);
    }
    s1(w1r[2], 2);
    lds_barrier();      // LDS hand-off only: global loads / stores stay in flight (a __syncthreads waits for them)
    // ---- stage 0: both row tiles against this wave's k half; reads run PD operands ahead of the MFMAs ----------------
    f32x4 acc[2];
    {
      constexpr int PD = 4, NI = 16;       // iteration i: row tile i / 8, k-block 8 kh + i % 8
      xh8 af[NI][2];
      // x_off<512>(16 rt + c, 32 (8 kh + u) + 8 q) = c * 512 + kh * 256 + bx[u & 3] + rt * 8192 + (u >> 2) * 128 halfs with
      // bx[v] = ((4 v + q) ^ c) * 8: four addresses, the rest are immediates
      int bx[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) bx[v] = cz * 512 + kh * 256 + (((4 * v + qz) ^ cz) << 3);
      auto rd = [&](int i) {
        const int off = bx[i & 3] + (i >> 3) * 8192 + ((i & 7) >> 2) * 128;
        af[i][0] = *reinterpret_cast<const xh8*>(img + off);
        af[i][1] = *reinterpret_cast<const xh8*>(img + BH_PL_I + off);
      };
      f32x4 lo[2], hi[2];
#pragma unroll
      for (int y = 0; y < 2; ++y) { lo[y] = f32x4{0.f, 0.f, 0.f, 0.f}; hi[y] = lo[y]; }
#pragma unroll
      for (int i = 0; i < PD; ++i) rd(i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (i + PD < NI) rd(i + PD);
        __builtin_amdgcn_sched_barrier(0);
        const int y = i >> 3, u = i & 7;
        lo[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0f[u][1], af[i][0], lo[y], 0, 0, 0);
        lo[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0f[u][0], af[i][1], lo[y], 0, 0, 0);
        hi[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0f[u][0], af[i][0], hi[y], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[0] = hi[0] + lo[0];
      acc[1] = hi[1] + lo[1];
    }
    xp[p][kh][lane] = kh == 0 ? acc[1] : acc[0];
    lds_barrier();      // LDS hand-off only: global loads / stores stay in flight (a __syncthreads waits for them)
    // ---- gates (lstm.py:26-32): register j = gate j (i, f, g, o) of unit hid ---------------------------------------------
    const size_t bt = b * T + t;
    f32x4 rsv;                                     // the gates and the unrounded h_t: stored behind the exchange
    float hraw;
    {
      const f32x4 tot = (kh == 0 ? acc[0] : acc[1]) + xp[p][1 - kh][lane];
      const f32x4 usc = un4 * e0f;
      const float ig = bsigmoid(fmaf(tot[0], usc[0], gi[0] + bh[0]));
      const float fg = bsigmoid(fmaf(tot[1], usc[1], gi[2] + bh[1]));
      const float gg = btanh(fmaf(tot[2], usc[2], gi[1] + bh[2]));
      const float og = bsigmoid(fmaf(tot[3], usc[3], gi[3] + bh[3]));
      e0f = 1.0f;
      const float cy = fg * cst + ig * gg;
      float hy = og * btanh(cy);
      cst = cy;
      rsv = f32x4{ig, gg, fg, og};
      hraw = hy;
      hy = round_like(out, hy);                    // what the next step sees: rounded once to the storage type
      hst = hy;
      // swap halves of h_t with the partner workgroup (ttrnn_fast_big.hip:k_lstm_fwd_big2: self-validating 64-bit words
      // (value, step tag), relaxed agent-scope atomics, double-buffered by step parity).  The word goes out FIRST and nothing
      // else is put into the memory pipe before the partner's word is back: the poll's return is counted in order with every
      // earlier store and load of the wave — with the output / reserve stores and the next step's gin load (an HBM round trip)
      // in front of it, every step waited for them before it could see a word that was already there
      __hip_atomic_store(hx + (b * 2 + (t & 1)) * H + hid,
                         ((unsigned long long)(unsigned)(t + 1) << 32) | (unsigned long long)__float_as_uint(hy),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      _Float16 u, v;
      split2h(hy * BH_HSC, u, v);
      hpl[ho] = u; hpl[BH_PL_H + ho] = v;
    }
    {
      const unsigned long long* src = hx + (b * 2 + (t & 1)) * H + hidp;
      unsigned long long w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // bounded wait; a time-out poisons the state with NaN (see k_lstm_fwd_big2)
      long spin = 0;
      while (!dead && (unsigned)(w >> 32) != (unsigned)(t + 1)) {
        __builtin_amdgcn_s_sleep(1);
        w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spin > (1L << 21)) { dead = true; if (status) atomicAdd(status + TTRNN_STAT_PAIR_TIMEOUTS, 1u); }
      }
      const float hp = dead ? __uint_as_float(0x7FC00000u) : __uint_as_float((unsigned)w);
      _Float16 u, v;
      split2h(hp * BH_HSC, u, v);
      hpl[hop] = u; hpl[BH_PL_H + hop] = v;
    }
    // behind the exchange: this step's stores and the next step's gate inputs (consumed a whole step from now)
    if (reserve) {
      float* rv = reserve + res_gate(bt, H, hid);
      rv[0] = rsv[0]; rv[1] = rsv[1]; rv[2] = rsv[2]; rv[3] = rsv[3];
      reserve[res_cell((size_t)B * T, bt, H, hid)] = cst;
    }
    if constexpr (OUT) st(out, bt * H + hid, hraw);
    if (t + 1 < T) gi = gin4[(bt + 1) * H + hid];
    lds_barrier();      // LDS hand-off only: global loads / stores stay in flight (a __syncthreads waits for them)
  }
  if (dead) hst = cst = __uint_as_float(0x7FC00000u);       // T == time-out step: nothing downstream has seen the NaN yet
  if (hT) st(hT, b * H + hid, hst);
  if (cT) st(cT, b * H + hid, cst);
}

}  // namespace

static constexpr size_t BH_LDS_PAIR = 2 * BH_PL_I * sizeof(_Float16) + 8 * 8 * 64 * sizeof(xh8);
bool bigh_pair_resident(int dtype, int B) {
  const void* fn = dtype == TTRNN_F32 ? reinterpret_cast<const void*>(k_lstm_fwd_big2h<float>)
                                      : reinterpret_cast<const void*>(k_lstm_fwd_big2h<bf16_t>);
  return ensure_dynamic_lds(fn, BH_LDS_PAIR) == TTRNN_OK && resident_at_once(fn, FAST_NT, BH_LDS_PAIR, 2L * B);
}
size_t bigh_workspace_bytes() { return (size_t)(BH_F1 + BH_F0) * sizeof(xh8) + BH_HDR_BYTES; }

template <typename TS>
static int launch_bigh_t(const RnnShape& rs, const float* gin, const void* h0, const void* c0, const float* m2_hid,
                         const void* bias_in, const void* bias_hid, void* out, void* hT, void* cT, float* reserve,
                         unsigned long long* hxb, void* scratch, hipStream_t stream) {
  int* hdr = (int*)scratch;
  xh8* f1 = (xh8*)((char*)scratch + BH_HDR_BYTES);
  xh8* f0 = f1 + BH_F1;
  static_assert((BH_PART + BH_I1 * BH_R1) * sizeof(int) <= BH_HDR_BYTES && T1::K == 64 && T0::K == 512 && T0::M == 64, "header");
  hipLaunchKernelGGL(k_bigh_diag_a, dim3(BH_I1), dim3(256), 0, stream, m2_hid, hdr);
  hipLaunchKernelGGL(k_bigh_diag_b, dim3(64), dim3(256), 0, stream, m2_hid, hdr);
  hipLaunchKernelGGL(k_bigh_prep, dim3((BH_F1 / 2 + BH_F0 / 2 + 255) / 256), dim3(256), 0, stream, m2_hid, (const int*)hdr, f1, f0);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  // stage-0 image (64 KB) + the LDS-resident quarter of the stage-1 fragments (64 KB): one workgroup per CU
  constexpr size_t lds_pair = BH_LDS_PAIR;
  auto kern = out ? k_lstm_fwd_big2h<TS, true> : k_lstm_fwd_big2h<TS, false>;
  if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_pair) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
  const TS* bin = rs.has_bias_in ? (const TS*)bias_in : (const TS*)nullptr;
  const TS* bhid = rs.has_bias_hid ? (const TS*)bias_hid : (const TS*)nullptr;
  // (OPT_PAIR_FAULT, tests only: the last workgroup is not launched — its partner must time out, poison its sample with NaN
  // and count the event)
  hipLaunchKernelGGL(kern, dim3(2 * rs.B - (opt(OPT_PAIR_FAULT) ? 1 : 0)), dim3(FAST_NT), lds_pair, stream, rs.B, rs.T, gin,
                     (const TS*)h0, (const TS*)c0, f1, f0, (const int*)hdr, bin, bhid, (TS*)out, (TS*)hT, (TS*)cT, reserve, hxb,
                     device_status_ptr());
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

int launch_lstm_fwd_big2h(const RnnShape& rs, int dtype, const float* gin, const void* h0, const void* c0,
                          const float* m2_hid, const void* bias_in, const void* bias_hid, void* out, void* hT, void* cT,
                          float* reserve, unsigned long long* hxb, void* scratch, hipStream_t stream) {
  return dtype == TTRNN_F32
             ? launch_bigh_t<float>(rs, gin, h0, c0, m2_hid, bias_in, bias_hid, out, hT, cT, reserve, hxb, scratch, stream)
             : launch_bigh_t<bf16_t>(rs, gin, h0, c0, m2_hid, bias_in, bias_hid, out, hT, cT, reserve, hxb, scratch,
                                     stream);
}

}  // namespace ttrnn
