// ttrnn_fast_bigbh.hip — cfg5-class reverse-time kernel (pair of workgroups per sample) on two-piece fp16 operands.
// Same chain, same pair layout and the same exchange as ttrnn_fast_bigb.hip:k_lstm_bwd_big<ST, 2> (T0: dimg = A^T dg over
// the workgroup's 32 rows i23, K = 64; T1: dh = Bm dimg over its half of K = 2048; the partner's share of dh swapped once
// per step), but both stages run on v_mfma_f32_16x16x32_f16 with three terms per product (ttrnn_split.h, flavour (b))
// instead of the fp32 MFMA (16.4 k matrix-pipe cycles per step and workgroup -> 3.1 k), and NO core fragment is streamed:
//   * A^T (64 x 512, this wave's four m-tiles: 64 VGPRs) and three quarters of this wave's slice of Bm (96 VGPRs) are
//     resident in registers, the last quarter of every wave's slice (64 KB) in LDS;
//   * gate gradients have no a-priori bound: every step the workgroup takes the maximum of its 32 x 64 gate gradients
//     (wave shuffles + eight LDS words, read behind the barrier that hands the gradients to T0 anyway) and multiplies them
//     by 2^(14 - e), max < 2^e, while they are split into the MFMA operand — one scale for the whole operand, so it
//     factors out of every sum; T0's sums (< 2^33) are rescaled by 2^-18 before they are split into the fp16 image of T1;
//     the weights carry per-launch scales from their maxima; all powers of two, undone exactly on the fp32 sums;
//   * the k order of T1's operand is (r / 4, i23, r % 4) so that a wave's T0 store is contiguous.
// Replaces: torch autograd through lstm.py:123-133 / t3nsor/ops.py:81-90 for this shape (see ttrnn_fast_bigb.hip).
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

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using ST = ShpH1024R32L_T;
using T1 = St<ST, 1>;      // T0 of the chain: rows i23 (64), K = i01 (64), M = (j01, r) (512)
using T0 = St<ST, 0>;      // T1 of the chain: rows j01 (16), K = (i23, r) (2048), M = j23 (64)
constexpr int BB_H = 1024, BB_I23 = 64, BB_RL = 32, BB_K0L = 1024, BB_R = 32;
constexpr int BB_PL = 16 * BB_K0L;             // halfs per plane of T1's image [16 rows j01][1024 k']
constexpr int BB_PARTS = 32;
// scratch header: [2][BB_PARTS] partial maxima, then per ROW of the two operand matrices (512 rows of A^T, 64 of Bm) the sum
// of the pieces' absolute representation errors and the sum of the absolute scaled entries (big_guard_tripped)
constexpr int BB_ROWS = 512 + 64, BB_HDR_BYTES = 8192, BB_GUARD_OFF = 2 * BB_PARTS;
static_assert((BB_GUARD_OFF + 2 * BB_ROWS) * sizeof(float) <= BB_HDR_BYTES, "header");
constexpr int BB_FA = 8 * 4 * 2 * 2 * 64;      // fa[wave][x][kb][piece][lane]
constexpr int BB_FB = 2 * 8 * 16 * 2 * 64;     // fb[half][wave][kbl][piece][lane]
constexpr int BB_RES = 12;                     // k-blocks of a wave's Bm slice resident in registers (the other 4: LDS)
static_assert(T1::K == 64 && T1::M == 512 && T1::ROWS == BB_I23 && T0::K == 2048 && T0::M == 64 && T0::ROWS == 16 &&
                  out_size_of<ST>() == BB_H && in_size_of<ST>() == 4 * BB_H && T1::R == BB_R && FAST_NT == 512,
              "transposed merged shape of cfg5");

template <int k>
__device__ __forceinline__ int fragT_index(int kk, int m) {      // ttrnn_big.h:frag_decode inverted
  using T = St<ST, k>;
  return (((m >> 4) * T::NU + (kk >> 4)) * 64 + ((kk >> 2) & 3) * 16 + (m & 15)) * 4 + (kk & 3);
}

struct BbScales { float a, b; int ea, eb; };
// parts: [2][BB_PARTS] partial maxima (|Bm| = stage 0 of ST, then |A^T| = stage 1): A^T 2^a < 2^13, Bm 2^b < 2^14
__device__ __forceinline__ BbScales bb_scales(const float* __restrict__ parts, int lane) {
  float mb = parts[lane & (BB_PARTS - 1)], ma = parts[BB_PARTS + (lane & (BB_PARTS - 1))];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) { mb = fmaxf(mb, __shfl_xor(mb, o)); ma = fmaxf(ma, __shfl_xor(ma, o)); }
  BbScales s;
  s.ea = 13 - f10h_expo(ma);
  s.eb = 14 - f10h_expo(mb);
  s.a = ldexpf(1.f, s.ea);
  s.b = ldexpf(1.f, s.eb);
  return s;
}

__global__ void __launch_bounds__(256) k_bigbh_absmax(const float* __restrict__ fragT, float* __restrict__ parts) {
  __shared__ float red[4];
  {                                              // zero the guard's row sums (k_bigbh_prep accumulates into them)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * BB_ROWS) parts[BB_GUARD_OFF + i] = 0.f;
  }
  constexpr int N0 = T0::K * T0::M, N1 = T1::K * T1::M;
  const int which = blockIdx.x / BB_PARTS, part = blockIdx.x % BB_PARTS;
  const float* a = fragT + woff_of<ST>(which);
  const int n = which == 0 ? N0 : N1;
  float m = 0.f;
  for (int i = part * 256 + threadIdx.x; i < n; i += BB_PARTS * 256) m = fmaxf(m, fabsf(a[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// fa: m-tile mt = wave + 8 x of A^T, row r <-> m = (j01, r') = 16 mt + r, k = i01 = 32 kb + 8 q + i
// fb: wave = (mt = wave & 3, kh = wave >> 2): row r <-> j23 = 16 mt + r, k' = 32 (16 kh + kbl) + 8 q + i of the workgroup's
//     half, k' = (r' / 4) * 128 + i23l * 4 + r' % 4  <->  kk = (half * 32 + i23l) * 32 + r'
__global__ void __launch_bounds__(256) k_bigbh_prep(const float* __restrict__ fragT, float* __restrict__ parts,
                                                    xh8* __restrict__ fa, xh8* __restrict__ fb) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int lane = e & 63, r = lane & 15, q = lane >> 4;
  const BbScales sc = bb_scales(parts, threadIdx.x & 63);
  float* gerr = parts + BB_GUARD_OFF;            // [BB_ROWS] error sums, then [BB_ROWS] magnitude sums
  float esum = 0.f, asum = 0.f;
  xh8 p0, p1;
  if (e < BB_FA / 2) {
    const int kb = (e >> 6) & 1, x = (e >> 7) & 3, wave = e >> 9;
    const int m = 16 * (wave + 8 * x) + r;
    const float* W = fragT + woff_of<ST>(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      _Float16 u, v;
      const float w = W[fragT_index<1>(32 * kb + 8 * q + i, m)] * sc.a;
      split2h(w, u, v);
      p0[i] = u; p1[i] = v;
      esum += fabsf((w - (float)u) - (float)v);
      asum += fabsf(w);
    }
    atomicAdd(gerr + m, esum);
    atomicAdd(gerr + BB_ROWS + m, asum);
    const size_t o = (size_t)(e >> 6) * 2 * 64 + lane;
    fa[o] = p0;
    fa[o + 64] = p1;
  } else if (e < BB_FA / 2 + BB_FB / 2) {
    const int g = e - BB_FA / 2;
    const int kbl = (g >> 6) & 15, wave = (g >> 10) & 7, half = g >> 13;
    const int mt = wave & 3, kh = wave >> 2;
    const float* W = fragT + woff_of<ST>(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kp = 32 * (16 * kh + kbl) + 8 * q + i;
      const int i23l = (kp & 127) >> 2, rr = (kp >> 7) * 4 + (kp & 3);
      _Float16 u, v;
      const float w = W[fragT_index<0>((half * BB_RL + i23l) * BB_R + rr, 16 * mt + r)] * sc.b;
      split2h(w, u, v);
      p0[i] = u; p1[i] = v;
      esum += fabsf((w - (float)u) - (float)v);
      asum += fabsf(w);
    }
    atomicAdd(gerr + 512 + 16 * mt + r, esum);
    atomicAdd(gerr + BB_ROWS + 512 + 16 * mt + r, asum);
    const size_t o = (size_t)(g >> 6) * 2 * 64 + lane;
    fb[o] = p0;
    fb[o + 64] = p1;
  }
}

template <typename TS>
__global__ void __launch_bounds__(FAST_NT) k_lstm_bwd_big2h(int B, int T, const TS* __restrict__ c0,
                                                            const xh8* __restrict__ fa, const xh8* __restrict__ fb,
                                                            const float* __restrict__ parts,
                                                            const float* __restrict__ reserve,
                                                            const TS* __restrict__ d_out, const TS* __restrict__ d_hT,
                                                            const TS* __restrict__ d_cT, float* __restrict__ dg_in,
                                                            TS* __restrict__ d_h0, TS* __restrict__ d_c0,
                                                            unsigned long long* __restrict__ hx,
                                                            unsigned* __restrict__ status,
                                                            unsigned* __restrict__ colmax) {
  constexpr int H = BB_H, GH = 4 * BB_H, I23 = BB_I23, RL = BB_RL;
  __shared__ __attribute__((aligned(16))) float dyimg[RL * T1::K];      // gate gradients [i23 local][i01], fp32
  __shared__ __attribute__((aligned(16))) float dhp[2 * T0::M * 16];    // partial dh [k half][j23][j01]
  __shared__ float smax[2][FAST_NW];                                    // per-wave maxima of |dg|, by step parity
  // running column maxima of this thread's four gate gradients (by-product for the weight-gradient step: BwdStats,
  // ttrnn_launch.h) — in LDS: the kernel has no four registers to spare (256 VGPRs, 8 spilled)
  __shared__ __attribute__((aligned(16))) float cmx_s[FAST_NT * 4];
  extern __shared__ __attribute__((aligned(16))) float big_lds[];
  _Float16* img = reinterpret_cast<_Float16*>(big_lds);                 // T1's image: two planes [16][1024]
  xh8* wl = reinterpret_cast<xh8*>(big_lds) + 2 * BB_PL / 8;            // the non-resident quarter of Bm's fragments

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int mt1 = wave & 3, kh = wave >> 2;
  const size_t b = blockIdx.x >> 1;
  const int half = blockIdx.x & 1;
  const BbScales sc = bb_scales(parts, lane);
  // One scale per operand matrix (taken from its maximum): if a row's fp16 pieces lost more than fp32-class accuracy — a few
  // large entries pushed the rest of the matrix into fp16's subnormal range — this launch is left to the fp32-MFMA pair
  // kernel queued right behind it (ttrnn_fast_bigb.hip), which takes the opposite decision from the same sums
  if (big_guard_tripped(parts + BB_GUARD_OFF, BB_ROWS, tid, FAST_NT)) {
    if (tid == 0 && blockIdx.x == 0 && status) atomicAdd(status + TTRNN_STAT_GUARD_TRIPS, 1u);
    return;
  }

  // this thread's unit: hid = mq*64 + i23, i23 = half*RL + rl  (gate g of it is row i01 = 16g + mq of dg)
  const int rl = tid & 31, mq = tid >> 5;
  const int hid = mq * I23 + half * RL + rl;
  float dhrec = d_hT ? ld(d_hT, b * H + hid) : 0.f;
  float dcs = d_cT ? ld(d_cT, b * H + hid) : 0.f;
  const float c0v = c0 ? ld(c0, b * H + hid) : 0.f;
  f32x4 ra, na;
  float rc, nc, dcur, dnxt;
  {
    const size_t bt = b * T + (T - 1);
    const size_t bt1 = T > 1 ? bt - 1 : bt;
    ra = *reinterpret_cast<const f32x4*>(reserve + res_gate(bt, H, hid));
    rc = reserve[res_cell((size_t)B * T, bt, H, hid)];
    na = *reinterpret_cast<const f32x4*>(reserve + res_gate(bt1, H, hid));
    nc = reserve[res_cell((size_t)B * T, bt1, H, hid)];
    dcur = d_out ? ld(d_out, bt * H + hid) : 0.f;
    dnxt = d_out ? ld(d_out, bt1 * H + hid) : 0.f;
  }
  bool dead = false;
  *reinterpret_cast<f32x4*>(cmx_s + 4 * tid) = f32x4{0.f, 0.f, 0.f, 0.f};

  // resident fragments: A^T (four m-tiles), Bm k-blocks 0..BB_RES-1 of this wave's slice; the rest of the slice -> LDS
  xh8 wa[4][2][2], wb[BB_RES][2];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) wa[x][kb][pc] = fa[(size_t)(((wave * 4 + x) * 2 + kb) * 2 + pc) * 64 + lane];
  const xh8* fbw = fb + (size_t)(half * 8 + wave) * (16 * 2 * 64);
#pragma unroll
  for (int u = 0; u < BB_RES; ++u)
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) wb[u][pc] = fbw[(size_t)(u * 2 + pc) * 64 + lane];
#pragma unroll
  for (int e = 0; e < (16 - BB_RES) * 2; ++e) wl[(wave * (16 - BB_RES) * 2 + e) * 64 + lane] = fbw[(size_t)(BB_RES * 2 + e) * 64 + lane];
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    const size_t bt = b * T + t;
    const int par = t & 1;
    int z = 0;
    asm volatile("" : "+v"(z));            // per-step opaque lane id: LDS addresses are recomputed, not hoisted
    const int cz = (lane + z) & 15, qz = (lane + z) >> 4;
    // ---- G: gate gradients (lstm.py:26-32 differentiated) -----------------------------------------------------------
    f32x4 fa4;
    float fc, fd;
    {
      const float dht = dcur + dhrec;
      const float ig = ra[0], gg = ra[1], fg = ra[2], og = ra[3], cy = rc;
      const float cprev = t > 0 ? nc : c0v;
      const float tc = btanh(cy);
      const float dct = dcs + dht * og * (1.0f - tc * tc);
      const float p0 = dct * gg * ig * (1.0f - ig);             // d pre-activation of i
      const float p1 = dct * cprev * fg * (1.0f - fg);          //                     f
      const float p2 = dct * ig * (1.0f - gg * gg);             //                     g
      const float p3 = dht * tc * og * (1.0f - og);             //                     o
      dcs = dct * fg;
      dyimg[a_off<T1::K>(rl, 0 * 16 + mq)] = p0;
      dyimg[a_off<T1::K>(rl, 1 * 16 + mq)] = p1;
      dyimg[a_off<T1::K>(rl, 2 * 16 + mq)] = p2;
      dyimg[a_off<T1::K>(rl, 3 * 16 + mq)] = p3;
      float mx = fmaxf(fmaxf(fabsf(p0), fabsf(p1)), fmaxf(fabsf(p2), fabsf(p3)));
      if (colmax) {
        f32x4 cm = *reinterpret_cast<const f32x4*>(cmx_s + 4 * tid);
        cm = f32x4{fmaxf(cm[0], fabsf(p0)), fmaxf(cm[1], fabsf(p1)), fmaxf(cm[2], fabsf(p2)), fmaxf(cm[3], fabsf(p3))};
        *reinterpret_cast<f32x4*>(cmx_s + 4 * tid) = cm;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      if (lane == 0) smax[par][wave] = mx;
      float* dg = dg_in + bt * GH + hid;
      dg[0] = p0; dg[H] = p1; dg[2 * H] = p2; dg[3 * H] = p3;
      // record / d_out of step t-2, consumed two iterations from now
      const size_t b2 = t > 1 ? bt - 2 : b * T;
      fa4 = *reinterpret_cast<const f32x4*>(reserve + res_gate(b2, H, hid));
      fc = reserve[res_cell((size_t)B * T, b2, H, hid)];
      fd = d_out ? ld(d_out, b2 * H + hid) : 0.f;
    }
    lds_barrier();      // LDS hand-off only: global loads / stores stay in flight (a __syncthreads waits for them)
    // ---- this step's scale: max |dg| of the workgroup < 2^e  ->  2^(14 - e) -------------------------------------------------
    float mxw = smax[par][lane & (FAST_NW - 1)];
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) mxw = fmaxf(mxw, __shfl_xor(mxw, o));
    // (from the exponent bits: frexpf / ldexpf of the device library cost the fused-core kernel 300 cycles per stage, lesson 36;
    // the maximum's biased exponent clamped to f10h_expo's range, a zero maximum scales zeros)
    int ebm = (int)(__float_as_uint(mxw) >> 23);
    ebm = ebm < 87 ? 87 : (ebm > 167 ? 167 : ebm);
    const int eg = 140 - ebm;                              // 14 - e with mxw < 2^e, e = ebm - 126
    const float sg = __uint_as_float((unsigned)(127 + eg) << 23);
    const float un = __uint_as_float((unsigned)(127 - (sc.ea + eg - 18 + sc.eb)) << 23);
    // ---- T0: dimg[(j01, r)][i23 local] = A^T dg, rescaled and split into T1's image ------------------------------------------
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      xh8 g0[2], g1[2];                       // the two pieces of 2^eg dg, rows 16 rt + c, k-blocks 0 / 1
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const f32x4 va = *reinterpret_cast<const f32x4*>(dyimg + a_off<T1::K>(16 * rt + cz, 32 * kb + 8 * qz)) * sg;
        const f32x4 vb = *reinterpret_cast<const f32x4*>(dyimg + a_off<T1::K>(16 * rt + cz, 32 * kb + 8 * qz + 4)) * sg;
        unsigned a0, b0, a1, b1, a2, b2, a3, b3;
        split_pair_h(va[0], va[1], a0, b0);
        split_pair_h(va[2], va[3], a1, b1);
        split_pair_h(vb[0], vb[1], a2, b2);
        split_pair_h(vb[2], vb[3], a3, b3);
        g0[kb] = __builtin_bit_cast(xh8, u32x4{a0, a1, a2, a3});
        g1[kb] = __builtin_bit_cast(xh8, u32x4{b0, b1, b2, b3});
      }
#pragma unroll
      for (int xp = 0; xp < 4; xp += 2) {
        f32x4 lo[2], hi[2];
#pragma unroll
        for (int y = 0; y < 2; ++y) { lo[y] = f32x4{0.f, 0.f, 0.f, 0.f}; hi[y] = lo[y]; }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int y = 0; y < 2; ++y) {
            lo[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[xp + y][kb][1], g0[kb], lo[y], 0, 0, 0);
            lo[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[xp + y][kb][0], g1[kb], lo[y], 0, 0, 0);
            hi[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[xp + y][kb][0], g0[kb], hi[y], 0, 0, 0);
          }
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          // m0 = 16 mt + 4 q: j01 = mt / 2, r0 = 16 (mt & 1) + 4 q;  k' = (r0 / 4) * 128 + (16 rt + c) * 4
          const int mt = wave + 8 * (xp + y);
          const int row = mt >> 1;
          const int kq = 4 * (mt & 1) + q;                                   // r0 / 4
          // x_off<1024>(row, k'): slot = kq * 16 + (16 rt + c) / 2, the XOR touches its low four bits only
          const int off = ((row * 128 + kq * 16 + (((8 * rt + (cz >> 1))) ^ (row & 15))) << 3) + ((cz & 1) << 2);
          const f32x4 v = (hi[y] + lo[y]) * 3.814697265625e-06f;             // 2^-18
          store_split4_h(img, BB_PL, off, v);
        }
      }
    }
    lds_barrier();      // LDS hand-off only: global loads / stores stay in flight (a __syncthreads waits for them)
    // ---- T1 over this workgroup's K slice: wave = (m-tile, k half), reads run PD operands ahead of the MFMAs -----------------
    {
      constexpr int PD = 4, NI = 16;
      xh8 af[NI][2];
      // x_off<1024>(c, 32 (16 kh + u) + 8 q) = c * 1024 + kh * 512 + (u >> 2) * 128 + bx[u & 3], bx[v] = ((4 v + q) ^ c) * 8
      int bx[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) bx[v] = cz * 1024 + kh * 512 + (((4 * v + qz) ^ cz) << 3);
      auto rd = [&](int i) {
        const int off = bx[i & 3] + (i >> 2) * 128;
        af[i][0] = *reinterpret_cast<const xh8*>(img + off);
        af[i][1] = *reinterpret_cast<const xh8*>(img + BB_PL + off);
      };
      f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
      for (int i = 0; i < PD; ++i) rd(i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (i + PD < NI) rd(i + PD);
        xh8 w0, w1;
        if (i < BB_RES) { w0 = wb[i < BB_RES ? i : 0][0]; w1 = wb[i < BB_RES ? i : 0][1]; }
        else {
          w0 = wl[(wave * (16 - BB_RES) * 2 + (i - BB_RES) * 2) * 64 + lane];
          w1 = wl[(wave * (16 - BB_RES) * 2 + (i - BB_RES) * 2 + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, af[i][0], lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, af[i][1], lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, af[i][0], hi, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const f32x4 acc = (hi + lo) * un;
#pragma unroll
      for (int j = 0; j < 4; ++j) dhp[(kh * T0::M + 16 * mt1 + 4 * q + j) * 16 + c] = acc[j];
    }
    lds_barrier();      // LDS hand-off only: global loads / stores stay in flight (a __syncthreads waits for them)
    // ---- dh_{t-1} of the own units + the partner's share (ttrnn_fast_bigb.hip: tagged words, relaxed agent-scope atomics) ----
    {
      const int n = T - 1 - t;                              // sequence number of this step
      const int j23o = half * RL + rl, j23p = (1 - half) * RL + rl;
      const float own = dhp[j23o * 16 + mq] + dhp[(T0::M + j23o) * 16 + mq];
      const float snd = dhp[j23p * 16 + mq] + dhp[(T0::M + j23p) * 16 + mq];
      __hip_atomic_store(hx + (b * 2 + (n & 1)) * H + mq * I23 + j23p,
                         ((unsigned long long)(unsigned)(n + 1) << 32) | (unsigned long long)__float_as_uint(snd),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long* src = hx + (b * 2 + (n & 1)) * H + hid;
      unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long spin = 0;                                        // bounded; a time-out poisons dh with NaN
      while (!dead && (unsigned)(v >> 32) != (unsigned)(n + 1)) {
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spin > (1L << 21)) { dead = true; if (status) atomicAdd(status + TTRNN_STAT_PAIR_TIMEOUTS, 1u); }
      }
      dhrec = dead ? __uint_as_float(0x7FC00000u) : own + __uint_as_float((unsigned)v);
    }
    ra = na; rc = nc; dcur = dnxt;
    na = fa4; nc = fc; dnxt = fd;
  }
  if (d_h0) st(d_h0, b * H + hid, dhrec);
  if (d_c0) st(d_c0, b * H + hid, dcs);
  if (colmax) {
#pragma unroll
    for (int g = 0; g < 4; ++g) atomicMax(colmax + g * H + hid, __float_as_uint(cmx_s[4 * tid + g]));
  }
}

}  // namespace

static constexpr size_t BB_LDS_PAIR = 2 * BB_PL * sizeof(_Float16) + (size_t)8 * (16 - BB_RES) * 2 * 64 * sizeof(xh8);
bool bigbh_pair_resident(int dtype, int B) {
  const void* fn = dtype == TTRNN_F32 ? reinterpret_cast<const void*>(k_lstm_bwd_big2h<float>)
                                      : reinterpret_cast<const void*>(k_lstm_bwd_big2h<bf16_t>);
  return ensure_dynamic_lds(fn, BB_LDS_PAIR) == TTRNN_OK && resident_at_once(fn, FAST_NT, BB_LDS_PAIR, 2L * B);
}
size_t bigbh_workspace_bytes() { return (size_t)(BB_FA + BB_FB) * sizeof(xh8) + BB_HDR_BYTES; }
const float* bigbh_guard_rows(const void* scratch, int* n_rows) {
  *n_rows = BB_ROWS;
  return (const float*)scratch + BB_GUARD_OFF;
}

template <typename TS>
static int launch_bigbh_t(const RnnShape& rs, const void* c0, const float* fragT, const float* reserve, const void* d_out,
                          const void* d_hT, const void* d_cT, float* dg_in, void* d_h0, void* d_c0,
                          unsigned long long* hxb, void* scratch, hipStream_t stream, unsigned* colmax) {
  float* parts = (float*)scratch;
  xh8* fa = (xh8*)((char*)scratch + BB_HDR_BYTES);
  xh8* fb = fa + BB_FA;
  hipLaunchKernelGGL(k_bigbh_absmax, dim3(2 * BB_PARTS), dim3(256), 0, stream, fragT, parts);
  hipLaunchKernelGGL(k_bigbh_prep, dim3((BB_FA / 2 + BB_FB / 2 + 255) / 256), dim3(256), 0, stream, fragT, parts, fa, fb);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  // T1's image (64 KB) + the LDS-resident quarter of Bm's fragments (64 KB): one workgroup per CU
  constexpr size_t lds = BB_LDS_PAIR;
  if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_lstm_bwd_big2h<TS>), lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
  hipLaunchKernelGGL((k_lstm_bwd_big2h<TS>), dim3(2 * rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T, (const TS*)c0, fa, fb,
                     parts, reserve, (const TS*)d_out, (const TS*)d_hT, (const TS*)d_cT, dg_in, (TS*)d_h0, (TS*)d_c0, hxb,
                     device_status_ptr(), colmax);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

int launch_lstm_bwd_big2h(const RnnShape& rs, int dtype, const void* c0, const float* fragT, const float* reserve,
                          const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, void* d_h0, void* d_c0,
                          unsigned long long* hxb, void* scratch, hipStream_t stream, unsigned* colmax) {
  return dtype == TTRNN_F32 ? launch_bigbh_t<float>(rs, c0, fragT, reserve, d_out, d_hT, d_cT, dg_in, d_h0, d_c0, hxb,
                                                    scratch, stream, colmax)
                            : launch_bigbh_t<bf16_t>(rs, c0, fragT, reserve, d_out, d_hT, d_cT, dg_in, d_h0, d_c0, hxb,
                                                     scratch, stream, colmax);
}

}  // namespace ttrnn
