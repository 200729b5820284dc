// ttrnn_fast_bigb.hip — BPTT for the big TT shape (BASELINE cfg5: H = 1024, d = 4, r = 32): reverse-time kernel and
// batched weight gradients on the fp32 MFMA, both through the MERGED two-core matrix of ttrnn_big.h
//     W[(j01, j23)][(i01, i23)] = sum_r A[(j01, r)][i01] * Bm[j23][(i23, r)]         (A = cores 0·1, Bm = cores 2·3)
// (reference: torch autograd through lstm.py:123-133 and t3nsor/ops.py:81-90).
//
//   * k_lstm_bwd_big: one (B > #CUs/2) or two workgroups per sample, persistent over the T steps.  The transposed
//     chain dh = W dg is again two stages of 4.2 MFLOP when the MODES ARE REVERSED ("last core first" must contract
//     i01, the short side):  T0: dimg[i23][(j01,r)] = sum_i01 A[(j01,r)][i01] dg[i01][i23]   (64 rows, K = 64),
//     T1: dh[j01][j23] = sum_(i23,r) dimg[i23][(j01,r)] Bm[j23][(i23,r)]   (16 rows, K = 2048).  A pair splits over
//     i23: each workgroup owns the hidden units (and gate rows) of its 32 i23 values, runs T0 on those rows and T1
//     over its half of K, and swaps the partial dh of the other half's units once per step (relaxed agent-scope
//     atomics, as the forward pair kernel).
//   * k_ttlinear_wgrad_big: 4 workgroups per row chunk, one per slice of 16 i23 values; the gradients of the merged
//     cores (dA 512 x 64; a 64 x 512 slice of dBm) stay in MFMA accumulators for the whole launch and are flushed with
//     atomics at the end; k_bigw_finish applies the product rule back to the four TT cores.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn.h"
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_big.h"

namespace ttrnn {


// fragment-ordered (see frag_decode) cores of the transposed matrix, built from the three-core buffer of k_merge_cores01:
//   stage 1:  W^T_1[kk = i01][m = (j01, r)]  = A[(j01, r)][i01]
//   stage 0:  W^T_0[kk = (i23, a)][m = j23]  = Bm[j23][(i23, a)] = sum_r2 W'_1[(j1,r2)][(i1,a)] W'_2[j2][(i2,r2)]
template <class S3, class ST>
__global__ void __launch_bounds__(256) k_bigb_prep(const float* __restrict__ packed3, float* __restrict__ fragT) {
  using T1 = St<ST, 1>;
  using T0 = St<ST, 0>;
  constexpr int N1 = T1::K * T1::M, N0 = T0::K * T0::M;
  constexpr int J2 = S3::J[2], I1 = S3::I[1], I2 = S3::I[2], R1 = S3::R[1], R2 = S3::R[2], M0 = S3::I[0];
  static_assert(woff_of<ST>(1) == N0 && T1::K == M0 && T1::M == S3::J[0] * R1 && T0::K == I1 * I2 * R1 &&
                    T0::M == S3::J[1] * J2, "transposed shape");
  const float* W0 = packed3 + woff_of<S3>(0);
  const float* W1 = packed3 + woff_of<S3>(1);
  const float* W2 = packed3 + woff_of<S3>(2);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < N0) {
    int kk, m;
    frag_decode<ST, 0>(e, kk, m);
    const int i23 = kk / R1, a = kk % R1, i1 = i23 / I2, i2 = i23 % I2, j1 = m / J2, j2 = m % J2;
    float v = 0.f;
    for (int r2 = 0; r2 < R2; ++r2)
      v = fmaf(W1[(j1 * R2 + r2) * (I1 * R1) + i1 * R1 + a], W2[j2 * (I2 * R2) + i2 * R2 + r2], v);
    fragT[e] = v;
  } else if (e < N0 + N1) {
    int kk, m;
    frag_decode<ST, 1>(e - N0, kk, m);
    fragT[e] = W0[m * M0 + kk];
  }
}

// T0 of the transposed chain on the RL = 64 / NH local rows (i23) of the gate-gradient image dyimg[i23 local][i01]:
//   img0[j01][(i23 local, r)] = sum_i01 A[(j01,r)][i01] dy[i01][i23];  m-tiles {wave + 8x}, two row tiles at a time
template <class ST>
__device__ __forceinline__ void bigT_load1(f32x4 (&wf)[4][4], const float* __restrict__ fragT, int wave, int lane) {
  using T1 = St<ST, 1>;
  const f32x4* F1 = reinterpret_cast<const f32x4*>(fragT + woff_of<ST>(1));
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int u = 0; u < 4; ++u) wf[x][u] = F1[(size_t)((wave + FAST_NW * x) * T1::NU + u) * 64 + lane];
}

template <class ST, int NH>
__device__ __forceinline__ void bigT_stage1w(const f32x4 (&wf)[4][4], const float* dyimg, float* img0, int wave,
                                             int lane) {
  using T1 = St<ST, 1>;
  using T0 = St<ST, 0>;
  constexpr int RTL = T1::ROWS / NH / 16, K0L = T0::K / NH;
  const int c = lane & 15, q = lane >> 4;
#pragma unroll 1
  for (int rtb = 0; rtb < RTL; rtb += 2) {
    f32x4 af[2][4];
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        af[y][u] = *reinterpret_cast<const f32x4*>(dyimg + a_off<T1::K>(16 * (rtb + y) + c, (4 * u + q) * 4));
    f32x4 acc[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = acc[x][0]; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[x][u][e], af[0][u][e], acc[x][0], 0, 0, 0);
          acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[x][u][e], af[1][u][e], acc[x][1], 0, 0, 0);
        }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int m0 = 16 * (wave + FAST_NW * x) + 4 * q;       // (j01, r0 .. r0+3)
      const int j01 = m0 / T1::R, r0 = m0 % T1::R;
#pragma unroll
      for (int y = 0; y < 2; ++y)
        *reinterpret_cast<f32x4*>(img0 + a_off<K0L>(j01, (16 * (rtb + y) + c) * T1::R + r0)) = acc[x][y];
    }
  }
}

template <class ST, int NH>
__device__ __forceinline__ void bigT_stage1(const float* __restrict__ fragT, const float* dyimg, float* img0, int wave,
                                            int lane) {
  int z = 0;
  asm volatile("" : "+v"(z));          // keep the fragment loads inside this phase
  f32x4 wf[4][4];
  bigT_load1<ST>(wf, fragT, wave, lane + z);
  bigT_stage1w<ST, NH>(wf, dyimg, img0, wave, lane);
}

// T1 over this workgroup's K slice: dhp[k half][j23][j01] = partial sum_(i23,r) img0[j01][(i23,r)] Bm[j23][(i23,r)];
// wave = (m-tile wave % 4, k half wave / 4)
template <class ST, int NH>
__device__ __forceinline__ void bigT_stage0(const float* __restrict__ fragT, const float* img0, float* dhp, int half,
                                            int wave, int lane) {
  using T0 = St<ST, 0>;
  constexpr int K0L = T0::K / NH, NUL = K0L / 16, NUW = NUL / 2;
  const int c = lane & 15, q = lane >> 4;
  int z = 0;
  asm volatile("" : "+v"(z));
  const int mt = wave & 3, kh = wave >> 2;
  const f32x4* F0 = reinterpret_cast<const f32x4*>(fragT + woff_of<ST>(0));
  const int ug0 = half * NUL + kh * NUW, ul0 = kh * NUW;
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  constexpr int UC = 8;
#pragma unroll 1
  for (int u0 = 0; u0 < NUW; u0 += UC) {
    f32x4 wf[UC], af[UC];
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      wf[u] = F0[(size_t)(mt * T0::NU + ug0 + u0 + u) * 64 + lane + z];
      af[u] = *reinterpret_cast<const f32x4*>(img0 + a_off<K0L>(c, (4 * (ul0 + u0 + u) + q) * 4));
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][0], af[u][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][1], af[u][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][2], af[u][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][3], af[u][3], acc1, 0, 0, 0);
    }
  }
  const f32x4 acc = acc0 + acc1;
#pragma unroll
  for (int j = 0; j < 4; ++j) dhp[(kh * T0::M + 16 * mt + 4 * q + j) * 16 + c] = acc[j];
}

// The same stage for the time loop: chunk 0 of this wave's fragment groups stays in registers for the whole launch
// (r0), the others stream through two slots of eight groups, each requested two chunks before it is multiplied.
template <class ST, int NH>
__device__ __forceinline__ void bigT_load0(f32x4 (&w)[8], const float* __restrict__ fragT, int half, int wave, int lane,
                                           int chunk) {
  using T0 = St<ST, 0>;
  constexpr int NUL = T0::K / NH / 16, NUW = NUL / 2;
  const int mt = wave & 3, kh = wave >> 2;
  const f32x4* F0 = reinterpret_cast<const f32x4*>(fragT + woff_of<ST>(0));
#pragma unroll
  for (int u = 0; u < 8; ++u) w[u] = F0[(size_t)(mt * T0::NU + half * NUL + kh * NUW + 8 * chunk + u) * 64 + lane];
}

template <class ST, int NH>
__device__ __forceinline__ void bigT_stage0p(const f32x4 (&r0)[8], const float* __restrict__ fragT, const float* img0,
                                             float* dhp, int half, int wave, int lane) {
  using T0 = St<ST, 0>;
  constexpr int K0L = T0::K / NH, NUW = K0L / 32, NCH = NUW / 8;
  static_assert(NCH >= 3, "three or more chunks per wave");
  const int c = lane & 15, q = lane >> 4;
  const int mt = wave & 3, kh = wave >> 2;
  int z = 0;
  asm volatile("" : "+v"(z));
  f32x4 sa[8], sb[8];
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  auto mmc = [&](const f32x4 (&w)[8], int chunk) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(img0 + a_off<K0L>(c, (4 * (kh * NUW + 8 * chunk + u) + q) * 4));
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][0], af[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][1], af[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][2], af[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][3], af[3], acc1, 0, 0, 0);
    }
  };
  bigT_load0<ST, NH>(sa, fragT, half, wave, lane + z, 1);
  bigT_load0<ST, NH>(sb, fragT, half, wave, lane + z, 2);
  mmc(r0, 0);
#pragma unroll
  for (int i = 1; i < NCH; ++i) {
    if (i & 1) {
      mmc(sa, i);
      if (i + 2 < NCH) bigT_load0<ST, NH>(sa, fragT, half, wave, lane + z, i + 2);
    } else {
      mmc(sb, i);
      if (i + 2 < NCH) bigT_load0<ST, NH>(sb, fragT, half, wave, lane + z, i + 2);
    }
  }
  const f32x4 acc = acc0 + acc1;
#pragma unroll
  for (int j = 0; j < 4; ++j) dhp[(kh * T0::M + 16 * mt + 4 * q + j) * 16 + c] = acc[j];
}

// ---- reverse-time kernel ------------------------------------------------------------------------------------------------
template <class ST, int NH>
constexpr size_t bigb_lds_bytes() { return (size_t)St<ST, 0>::ROWS * (St<ST, 0>::K / NH) * sizeof(float); }

template <class ST, int NH, typename TS>
__global__ void __launch_bounds__(FAST_NT) k_lstm_bwd_big(int B, int T, const TS* __restrict__ c0,
                                                          const float* __restrict__ fragT,
                                                          const float* __restrict__ reserve,
                                                          const TS* __restrict__ d_out, const TS* __restrict__ d_hT,
                                                          const TS* __restrict__ d_cT, float* __restrict__ dg_in,
                                                          TS* __restrict__ d_h0, TS* __restrict__ d_c0,
                                                          unsigned long long* __restrict__ hx,
                                                          const float* __restrict__ guard, int guard_rows,
                                                          unsigned* __restrict__ status,
                                                          unsigned* __restrict__ colmax) {
  using T1 = St<ST, 1>;
  using T0 = St<ST, 0>;
  constexpr int H = out_size_of<ST>(), GH = in_size_of<ST>();
  // queued behind the two-piece fp16 kernel as its fallback (guard != NULL): runs only when that kernel stepped aside
  if (guard && !big_guard_tripped(guard, guard_rows, threadIdx.x, FAST_NT)) return;
  constexpr int I23 = T1::ROWS;               // 64
  constexpr int RL = I23 / NH;                // i23 values (rows of T0) of this workgroup
  constexpr int NUT = 2 / NH;                 // hidden units per thread
  constexpr int K0L = T0::K / NH;             // this workgroup's slice of T1's contraction
  constexpr int RTL = RL / 16, NUW = K0L / 32;      // row tiles of T0; 16-byte fragment groups of T1 per wave
  static_assert((NH == 1 || NH == 2) && GH == 4 * H && T1::K == 64 && T1::NU == 4 && T1::MT == 4 * FAST_NW &&
                    T0::M == 64 && T0::ROWS == 16 && I23 == 64 && H == 2 * FAST_NT && T1::R == 32 && RTL % 2 == 0 &&
                    NUW % 8 == 0,
                "pair layout of the transposed chain");

  __shared__ __attribute__((aligned(16))) float dyimg[RL * T1::K];      // gate gradients [i23 local][i01]
  __shared__ __attribute__((aligned(16))) float dhp[2 * T0::M * 16];    // partial dh [k half][j23][j01]
  extern __shared__ __attribute__((aligned(16))) float big_lds[];       // image of T1: [j01][K0L]
  float* img0 = big_lds;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t b = NH == 2 ? blockIdx.x >> 1 : blockIdx.x;
  const int half = NH == 2 ? (blockIdx.x & 1) : 0;

  // unit u of this thread: hid = mq*64 + i23, i23 = half*RL + 32u + rl  (gate g of it is row i01 = 16g + mq of dg)
  const int rl = tid & 31, mq = tid >> 5;
  int hid[NUT];
  float dhrec[NUT], dcs[NUT], c0v[NUT];
  f32x4 ra[NUT], na[NUT];
  float rc[NUT], nc[NUT], dcur[NUT], dnxt[NUT];
#pragma unroll
  for (int u = 0; u < NUT; ++u) {
    hid[u] = mq * I23 + half * RL + 32 * u + rl;
    dhrec[u] = d_hT ? ld(d_hT, b * H + hid[u]) : 0.f;
    dcs[u] = d_cT ? ld(d_cT, b * H + hid[u]) : 0.f;
    c0v[u] = c0 ? ld(c0, b * H + hid[u]) : 0.f;
    const size_t bt = b * T + (T - 1);
    const size_t bt1 = T > 1 ? bt - 1 : bt;
    ra[u] = *reinterpret_cast<const f32x4*>(reserve + res_gate(bt, H, hid[u]));
    rc[u] = reserve[res_cell((size_t)B * T, bt, H, hid[u])];
    na[u] = *reinterpret_cast<const f32x4*>(reserve + res_gate(bt1, H, hid[u]));
    nc[u] = reserve[res_cell((size_t)B * T, bt1, H, hid[u])];
    dcur[u] = d_out ? ld(d_out, bt * H + hid[u]) : 0.f;
    dnxt[u] = d_out ? ld(d_out, bt1 * H + hid[u]) : 0.f;
  }
  bool dead = false;
  f32x4 cmx[NUT];                             // column maxima of the units' gate gradients (BwdStats, ttrnn_launch.h)
#pragma unroll
  for (int u = 0; u < NUT; ++u) cmx[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  // resident for the whole launch: T0's fragments (A^T, 64 registers) and the first chunk of T1's
  f32x4 wf1[4][4], r0[8];
  bigT_load1<ST>(wf1, fragT, wave, lane);
  bigT_load0<ST, NH>(r0, fragT, half, wave, lane, 0);
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    const size_t bt = b * T + t;
    // ---- G: gate gradients (lstm.py:26-32 differentiated) -----------------------------------------------------------
    f32x4 fa[NUT];
    float fc[NUT], fd[NUT];
#pragma unroll
    for (int u = 0; u < NUT; ++u) {
      const float dht = dcur[u] + dhrec[u];
      const float ig = ra[u][0], gg = ra[u][1], fg = ra[u][2], og = ra[u][3], cy = rc[u];
      const float cprev = t > 0 ? nc[u] : c0v[u];
      const float tc = btanh(cy);
      const float dct = dcs[u] + dht * og * (1.0f - tc * tc);
      const float p0 = dct * gg * ig * (1.0f - ig);             // d pre-activation of i
      const float p1 = dct * cprev * fg * (1.0f - fg);          //                     f
      const float p2 = dct * ig * (1.0f - gg * gg);             //                     g
      const float p3 = dht * tc * og * (1.0f - og);             //                     o
      dcs[u] = dct * fg;
      cmx[u][0] = fmaxf(cmx[u][0], fabsf(p0)); cmx[u][1] = fmaxf(cmx[u][1], fabsf(p1));
      cmx[u][2] = fmaxf(cmx[u][2], fabsf(p2)); cmx[u][3] = fmaxf(cmx[u][3], fabsf(p3));
      const int row = 32 * u + rl;
      dyimg[a_off<T1::K>(row, 0 * 16 + mq)] = p0;
      dyimg[a_off<T1::K>(row, 1 * 16 + mq)] = p1;
      dyimg[a_off<T1::K>(row, 2 * 16 + mq)] = p2;
      dyimg[a_off<T1::K>(row, 3 * 16 + mq)] = p3;
      float* dg = dg_in + bt * GH + hid[u];
      dg[0] = p0; dg[H] = p1; dg[2 * H] = p2; dg[3 * H] = p3;
      // record / d_out of step t-2, consumed two iterations from now
      const size_t b2 = t > 1 ? bt - 2 : b * T;
      fa[u] = *reinterpret_cast<const f32x4*>(reserve + res_gate(b2, H, hid[u]));
      fc[u] = reserve[res_cell((size_t)B * T, b2, H, hid[u])];
      fd[u] = d_out ? ld(d_out, b2 * H + hid[u]) : 0.f;
    }
    __syncthreads();
    bigT_stage1w<ST, NH>(wf1, dyimg, img0, wave, lane);
    __syncthreads();
    bigT_stage0p<ST, NH>(r0, fragT, img0, dhp, half, wave, lane);
    __syncthreads();
    // ---- dh_{t-1} of the own units (+ the partner's share) ----------------------------------------------------------------
    if constexpr (NH == 1) {
#pragma unroll
      for (int u = 0; u < NUT; ++u) {
        const int j23 = 32 * u + rl;
        dhrec[u] = dhp[j23 * 16 + mq] + dhp[(T0::M + j23) * 16 + mq];
      }
    } else {
      const int n = T - 1 - t;                              // sequence number of this step
      const int j23o = half * RL + rl, j23p = (1 - half) * RL + rl;
      const float own = dhp[j23o * 16 + mq] + dhp[(T0::M + j23o) * 16 + mq];
      const float snd = dhp[j23p * 16 + mq] + dhp[(T0::M + j23p) * 16 + mq];
      // self-validating (value, step tag) words, relaxed agent-scope atomics, double-buffered by step parity: see the
      // forward pair kernel (ttrnn_fast_big.hip)
      __hip_atomic_store(hx + (b * 2 + (n & 1)) * H + mq * I23 + j23p,
                         ((unsigned long long)(unsigned)(n + 1) << 32) | (unsigned long long)__float_as_uint(snd),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long* src = hx + (b * 2 + (n & 1)) * H + hid[0];
      unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // bounded: a partner that is not resident must not hang the GPU; a time-out poisons dh with NaN (it then reaches every
      // gate gradient of the sample and d_h0 / d_c0), see the forward pair kernel
      long spin = 0;
      while (!dead && (unsigned)(v >> 32) != (unsigned)(n + 1)) {
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spin > (1L << 21)) { dead = true; if (status) atomicAdd(status + TTRNN_STAT_PAIR_TIMEOUTS, 1u); }
      }
      dhrec[0] = dead ? __uint_as_float(0x7FC00000u) : own + __uint_as_float((unsigned)v);
    }
#pragma unroll
    for (int u = 0; u < NUT; ++u) {
      ra[u] = na[u]; rc[u] = nc[u]; dcur[u] = dnxt[u];
      na[u] = fa[u]; nc[u] = fc[u]; dnxt[u] = fd[u];
    }
  }
#pragma unroll
  for (int u = 0; u < NUT; ++u) {
    if (d_h0) st(d_h0, b * H + hid[u], dhrec[u]);
    if (d_c0) st(d_c0, b * H + hid[u], dcs[u]);
    if (colmax) {
#pragma unroll
      for (int g = 0; g < 4; ++g) atomicMax(colmax + g * H + hid[u], __float_as_uint(cmx[u][g]));
    }
  }
}

// ---- batched input gradient: dx[n] = W dy[n] through the same two transposed stages (one row per iteration) -------------
template <class ST, typename TS>
__global__ void __launch_bounds__(FAST_NT) k_ttlinear_dx_big(int64_t n_rows, const float* __restrict__ fragT,
                                                             const float* __restrict__ dy, TS* __restrict__ dx) {
  using T1 = St<ST, 1>;
  using T0 = St<ST, 0>;
  constexpr int IN = out_size_of<ST>(), OUT = in_size_of<ST>(), I23 = T1::ROWS;
  __shared__ __attribute__((aligned(16))) float dyimg[I23 * T1::K];
  __shared__ __attribute__((aligned(16))) float dhp[2 * T0::M * 16];
  extern __shared__ __attribute__((aligned(16))) float big_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int64_t n = blockIdx.x; n < n_rows; n += gridDim.x) {
    for (int e = tid; e < OUT; e += FAST_NT) dyimg[a_off<T1::K>(e % I23, e / I23)] = dy[(size_t)n * OUT + e];   // o = i01*64 + i23
    __syncthreads();
    bigT_stage1<ST, 1>(fragT, dyimg, big_lds, wave, lane);
    __syncthreads();
    bigT_stage0<ST, 1>(fragT, big_lds, dhp, 0, wave, lane);
    __syncthreads();
    for (int j = tid; j < IN; j += FAST_NT) {                 // j = j01*64 + j23
      const int j01 = j / T0::M, j23 = j % T0::M;
      st(dx, (size_t)n * IN + j, dhp[j23 * 16 + j01] + dhp[(T0::M + j23) * 16 + j01]);
    }
    // the next row's stores to dyimg / image / dhp each sit behind a barrier that follows the last reads of this row
  }
}

// ---- batched weight gradients ---------------------------------------------------------------------------------------------
// LDS (floats): staging, double-buffered: x image (B operand of F), x rows (A operand of dBm), dy image (B operand of T),
// dy rows (B operand of dA); then img[i23l][k] (stride 528) and dimg[i23l][j01*48 + r] (stride 772): the strides put
// the four k-step rows of an MFMA operand on four different 16-bank groups
struct BigW {
  static constexpr int XI = 16 * 64, XB = 16 * 80, STG = 2 * XI + 2 * XB;
  static constexpr int IMS = 528, DMS = 772, DJS = 48;
  static constexpr int IMG = 2 * STG, DIMG = IMG + 16 * IMS, TOTAL = DIMG + 16 * DMS;
};

template <class S2, class ST, typename TS>
__global__ void __launch_bounds__(FAST_NT) k_ttlinear_wgrad_big(int64_t n_rows, int rows_per_wg,
                                                                const float* __restrict__ frag2,
                                                                const float* __restrict__ fragT,
                                                                const TS* __restrict__ x, const float* __restrict__ dy,
                                                                float* __restrict__ dA, float* __restrict__ dB,
                                                                float* __restrict__ d_bias) {
  using F1 = St<S2, 1>;      // forward stage 1: [j01 rows] K = j23 (64), M = (i23, r) (2048)
  using T1 = St<ST, 1>;      // transposed stage 1: [i23 rows] K = i01 (64), M = (j01, r) (512)
  constexpr int IN = in_size_of<S2>(), OUT = out_size_of<S2>();
  constexpr int R = S2::R[1];
  static_assert(IN == 1024 && OUT == 4096 && R == 32 && F1::K == 64 && F1::NU == 4 && T1::K == 64 && T1::NU == 4 &&
                    F1::MT == 128 && T1::MT == 32 && F1::ROWS == 16,
                "slice layout");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* img = lds + BigW::IMG;
  float* dimg = lds + BigW::DIMG;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int s = blockIdx.x & 3;                              // slice: i23 in [16s, 16s + 16)
  const int64_t n0 = (int64_t)(blockIdx.x >> 2) * rows_per_wg;
  const int64_t n1 = n0 + rows_per_wg < n_rows ? n0 + rows_per_wg : n_rows;

  f32x4 accA[4][4], accB[4][4];      // dA tiles (m-tile wave+8x, n-tile nt);  dBm tiles (m-tile mt, n-tile wave+8x)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { accA[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accB[i][j] = accA[i][j]; }
  float dbias[2] = {0.f, 0.f};

  // staging: thread -> x elements tid, tid + 512 (j01 = e / 64, j23 = e % 64) and dy slice elements (i01 = e / 16, i23l = e % 16)
  float sx[2], sd[2];
  auto stage_load = [&](int64_t n) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + e * FAST_NT;
      sx[e] = ld(x, (size_t)n * IN + idx);
      sd[e] = dy[(size_t)n * OUT + (idx >> 4) * 64 + 16 * s + (idx & 15)];
    }
  };
  auto stage_store = [&](int buf) {
    float* ximg = lds + buf * BigW::STG;
    float* xB = ximg + BigW::XI;
    float* dyimg = xB + BigW::XB;
    float* dyB = dyimg + BigW::XI;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + e * FAST_NT;
      ximg[a_off<64>(idx >> 6, idx & 63)] = sx[e];
      xB[(idx >> 6) * 80 + (idx & 63)] = sx[e];
      dyimg[a_off<64>(idx & 15, idx >> 4)] = sd[e];
      dyB[(idx & 15) * 80 + (idx >> 4)] = sd[e];
      dbias[e] += sd[e];
    }
  };
  if (n0 < n1) stage_load(n0);

  const f32x4* FW = reinterpret_cast<const f32x4*>(frag2 + woff_of<S2>(1));
  const f32x4* FT = reinterpret_cast<const f32x4*>(fragT + woff_of<ST>(1));

  for (int64_t n = n0; n < n1; ++n) {
    const int buf = (int)((n - n0) & 1);
    stage_store(buf);
    if (n + 1 < n1) stage_load(n + 1);
    __syncthreads();
    const float* ximg = lds + buf * BigW::STG;
    const float* xB = ximg + BigW::XI;
    const float* dyimg = xB + BigW::XB;
    const float* dyB = dyimg + BigW::XI;
    // ---- F: img[i23l][(j01, r)] = forward stage 1 restricted to the slice (local m-tiles wave + 8x) -------------------
    {
      int z = 0;
      asm volatile("" : "+v"(z));
      f32x4 wf[4][4], af[4];
#pragma unroll
      for (int xx = 0; xx < 4; ++xx)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          wf[xx][u] = FW[(size_t)((32 * s + wave + FAST_NW * xx) * F1::NU + u) * 64 + lane + z];
#pragma unroll
      for (int u = 0; u < 4; ++u) af[u] = *reinterpret_cast<const f32x4*>(ximg + a_off<64>(c, (4 * u + q) * 4));
#pragma unroll
      for (int xx = 0; xx < 4; ++xx) {
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xx][u][0], af[u][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xx][u][1], af[u][1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xx][u][2], af[u][2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xx][u][3], af[u][3], acc1, 0, 0, 0);
        }
        const int mtl = wave + FAST_NW * xx;                 // local m = (i23l, a): i23l = mtl / 2, a0 = 16 (mtl % 2) + 4q
        *reinterpret_cast<f32x4*>(img + (mtl >> 1) * BigW::IMS + c * R + 16 * (mtl & 1) + 4 * q) = acc0 + acc1;
      }
    }
    // ---- T: dimg[i23l][(j01, r)] = A dy on the slice's rows (m-tiles wave + 8x) ---------------------------------------
    {
      int z = 0;
      asm volatile("" : "+v"(z));
      f32x4 wf[4][4], af[4];
#pragma unroll
      for (int xx = 0; xx < 4; ++xx)
#pragma unroll
        for (int u = 0; u < 4; ++u) wf[xx][u] = FT[(size_t)((wave + FAST_NW * xx) * T1::NU + u) * 64 + lane + z];
#pragma unroll
      for (int u = 0; u < 4; ++u) af[u] = *reinterpret_cast<const f32x4*>(dyimg + a_off<64>(c, (4 * u + q) * 4));
#pragma unroll
      for (int xx = 0; xx < 4; ++xx) {
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xx][u][0], af[u][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xx][u][1], af[u][1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xx][u][2], af[u][2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xx][u][3], af[u][3], acc1, 0, 0, 0);
        }
        const int mt = wave + FAST_NW * xx;                  // m = (j01, r): j01 = mt / 2, r0 = 16 (mt % 2) + 4q
        *reinterpret_cast<f32x4*>(dimg + c * BigW::DMS + (mt >> 1) * BigW::DJS + 16 * (mt & 1) + 4 * q) = acc0 + acc1;
      }
    }
    __syncthreads();
    // ---- dA[k][i01] += sum_i23l img[i23l][k] dy[i01][i23l];  dBm[j23][(i23l,r)] += sum_j01 x[j01][j23] dimg[i23l][(j01,r)]
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
      float a[4], bv[4];
#pragma unroll
      for (int xx = 0; xx < 4; ++xx) a[xx] = img[(4 * sp + q) * BigW::IMS + 16 * (wave + FAST_NW * xx) + c];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bv[nt] = dyB[(4 * sp + q) * 80 + 16 * nt + c];
#pragma unroll
      for (int xx = 0; xx < 4; ++xx)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          accA[xx][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[xx], bv[nt], accA[xx][nt], 0, 0, 0);
    }
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
      float a[4], bv[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = xB[(4 * sp + q) * 80 + 16 * mt + c];
#pragma unroll
      for (int xx = 0; xx < 4; ++xx) {
        const int nt = wave + FAST_NW * xx;                  // n = (i23l, r): i23l = nt / 2, r = 16 (nt % 2) + c
        bv[xx] = dimg[(nt >> 1) * BigW::DMS + (4 * sp + q) * BigW::DJS + 16 * (nt & 1) + c];
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int xx = 0; xx < 4; ++xx)
          accB[mt][xx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bv[xx], accB[mt][xx], 0, 0, 0);
    }
    // the next iteration's staging goes to the other buffer; its barrier orders these reads before the next F / T stores
  }

  // ---- flush --------------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int xx = 0; xx < 4; ++xx)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        atomicAdd(dA + (size_t)(16 * (wave + FAST_NW * xx) + 4 * q + j) * 64 + 16 * nt + c, accA[xx][nt][j]);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int xx = 0; xx < 4; ++xx) {
      const int nt = wave + FAST_NW * xx;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        atomicAdd(dB + (size_t)(16 * mt + 4 * q + j) * F1::M + (16 * s + (nt >> 1)) * R + 16 * (nt & 1) + c,
                  accB[mt][xx][j]);
    }
  if (d_bias) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + e * FAST_NT;
      atomicAdd(d_bias + (idx >> 4) * 64 + 16 * s + (idx & 15), dbias[e]);
    }
  }
}

// ---- weight gradients through the DENSE gradient -------------------------------------------------------------------------
// dW[j][o] = sum_n x[n][j] dy[n][o] is one plain GEMM (8.4 MFLOP per row — half of what recomputing both merged-chain
// images per row costs, and no weight streaming), and the TT cores' gradients are linear in it:
//   dA[(j01,r)][i01] = sum_(j23,i23) dW[(j01,j23)][(i01,i23)] Bm[j23][(i23,r)],   dBm[j23][(i23,r)] = sum_(j01,i01) dW[..] A[(j01,r)][i01]
// (two projections of 134 M MACs, once per launch), then the product rule of k_bigw_finish.
// The GEMM is launch_dense_wgrad (ttrnn_fast_gemm.hip): 8 x 32 tiles of 128 x 128 = 256 workgroups, every one over ALL rows.
// natural-order merged cores for the projections:  BmN[j23][(i23, a)]  and  AT[(j01, i01)][r] = A[(j01, r)][i01]
template <class S3>
__global__ void __launch_bounds__(256) k_bigw_natural(const float* __restrict__ packed3, float* __restrict__ BmN,
                                                      float* __restrict__ AT) {
  constexpr int J2 = S3::J[2], I1 = S3::I[1], I2 = S3::I[2], R1 = S3::R[1], R2 = S3::R[2], M0 = S3::I[0];
  constexpr int NB = S3::J[1] * J2 * I1 * I2 * R1, NA = S3::J[0] * R1 * M0;
  const float* W0 = packed3 + woff_of<S3>(0);
  const float* W1 = packed3 + woff_of<S3>(1);
  const float* W2 = packed3 + woff_of<S3>(2);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < NB) {
    const int a = e % R1, i23 = (e / R1) % (I1 * I2), j23 = e / (R1 * I1 * I2);
    const int i1 = i23 / I2, i2 = i23 % I2, j1 = j23 / J2, j2 = j23 % J2;
    float v = 0.f;
    for (int r2 = 0; r2 < R2; ++r2)
      v = fmaf(W1[(j1 * R2 + r2) * (I1 * R1) + i1 * R1 + a], W2[j2 * (I2 * R2) + i2 * R2 + r2], v);
    BmN[e] = v;
  } else if (e < NB + NA) {
    const int f = e - NB;
    const int r = f % R1, i01 = (f / R1) % M0, j01 = f / (R1 * M0);
    AT[f] = W0[(j01 * R1 + r) * M0 + i01];
  }
}

// dA[(j01, r)][i01]: one workgroup per (j01, i01), the 64 x 64 block dW[(j01, :)][(i01, :)] staged in LDS
__global__ void __launch_bounds__(256) k_bigw_proj_a(const float* __restrict__ dW, const float* __restrict__ BmN,
                                                     float* __restrict__ dA) {
  __shared__ float sub[64 * 64];
  __shared__ float part[8 * 32];
  const int tid = threadIdx.x, j01 = blockIdx.x >> 6, i01 = blockIdx.x & 63;
  for (int e = tid; e < 4096; e += 256) sub[e] = dW[(size_t)(j01 * 64 + (e >> 6)) * 4096 + i01 * 64 + (e & 63)];
  __syncthreads();
  const int r = tid & 31, p = tid >> 5;
  float v = 0.f;
  for (int j23 = 8 * p; j23 < 8 * p + 8; ++j23)
#pragma unroll 8
    for (int i23 = 0; i23 < 64; ++i23) v = fmaf(sub[j23 * 64 + i23], BmN[(size_t)j23 * 2048 + i23 * 32 + r], v);
  part[p * 32 + r] = v;
  __syncthreads();
  if (tid < 32) {
    float t = 0.f;
    for (int pp = 0; pp < 8; ++pp) t += part[pp * 32 + tid];
    dA[(j01 * 32 + tid) * 64 + i01] = t;
  }
}

// dBm[j23][(i23, r)]: one workgroup per (j23, i23), the 16 x 64 gathered block dW[(:, j23)][(:, i23)] staged in LDS
__global__ void __launch_bounds__(256) k_bigw_proj_b(const float* __restrict__ dW, const float* __restrict__ AT,
                                                     float* __restrict__ dB) {
  __shared__ float sub[16 * 64];
  __shared__ float part[8 * 32];
  const int tid = threadIdx.x, j23 = blockIdx.x >> 6, i23 = blockIdx.x & 63;
  for (int e = tid; e < 1024; e += 256) sub[e] = dW[(size_t)((e >> 6) * 64 + j23) * 4096 + (e & 63) * 64 + i23];
  __syncthreads();
  const int r = tid & 31, p = tid >> 5;
  float v = 0.f;
  for (int j01 = 2 * p; j01 < 2 * p + 2; ++j01)
#pragma unroll 8
    for (int i01 = 0; i01 < 64; ++i01) v = fmaf(sub[j01 * 64 + i01], AT[(size_t)(j01 * 64 + i01) * 32 + r], v);
  part[p * 32 + r] = v;
  __syncthreads();
  if (tid < 32) {
    float t = 0.f;
    for (int pp = 0; pp < 8; ++pp) t += part[pp * 32 + tid];
    dB[(size_t)j23 * 2048 + i23 * 32 + tid] = t;
  }
}

// product rule from the merged-core gradients back to the four TT cores (packed layout of S4), accumulated into d_packed:
//   A[(j0,j1,r2)][(i0,i1)] = sum_r1 W0[(j0,r1)][i0] W1[(j1,r2)][(i1,r1)];  Bm[(j2,j3)][(i2,i3,a)] = sum_r3 W2[(j2,r3)][(i2,a)] W3[j3][(i3,r3)]
template <class S4>
__global__ void __launch_bounds__(256) k_bigw_finish(const float* __restrict__ packed4, const float* __restrict__ dA,
                                                     const float* __restrict__ dB, float* __restrict__ d_packed) {
  constexpr int J0 = S4::J[0], J1 = S4::J[1], J2 = S4::J[2], J3 = S4::J[3];
  constexpr int I0 = S4::I[0], I1 = S4::I[1], I2 = S4::I[2], I3 = S4::I[3];
  constexpr int R1 = S4::R[1], R2 = S4::R[2], R3 = S4::R[3];
  constexpr int N0 = J0 * R1 * I0, N1 = J1 * R2 * I1 * R1, N2 = J2 * R3 * I2 * R2, N3 = J3 * I3 * R3;
  constexpr int MA = I0 * I1, MB = I2 * I3 * R2;
  const float* W0 = packed4 + woff_of<S4>(0);
  const float* W1 = packed4 + woff_of<S4>(1);
  const float* W2 = packed4 + woff_of<S4>(2);
  const float* W3 = packed4 + woff_of<S4>(3);
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  float v = 0.f;
  if (e < N0) {
    const int i0 = e % I0, r1 = (e / I0) % R1, j0 = e / (I0 * R1);
    for (int j1 = 0; j1 < J1; ++j1)
      for (int r2 = 0; r2 < R2; ++r2)
        for (int i1 = 0; i1 < I1; ++i1)
          v = fmaf(dA[((j0 * J1 + j1) * R2 + r2) * MA + i0 * I1 + i1], W1[(j1 * R2 + r2) * (I1 * R1) + i1 * R1 + r1], v);
    d_packed[woff_of<S4>(0) + e] += v;
    return;
  }
  e -= N0;
  if (e < N1) {
    const int r1 = e % R1, i1 = (e / R1) % I1, r2 = (e / (R1 * I1)) % R2, j1 = e / (R1 * I1 * R2);
    for (int j0 = 0; j0 < J0; ++j0)
      for (int i0 = 0; i0 < I0; ++i0)
        v = fmaf(dA[((j0 * J1 + j1) * R2 + r2) * MA + i0 * I1 + i1], W0[(j0 * R1 + r1) * I0 + i0], v);
    d_packed[woff_of<S4>(1) + e] += v;
    return;
  }
  e -= N1;
  if (e < N2) {
    const int a = e % R2, i2 = (e / R2) % I2, r3 = (e / (R2 * I2)) % R3, j2 = e / (R2 * I2 * R3);
    for (int j3 = 0; j3 < J3; ++j3)
      for (int i3 = 0; i3 < I3; ++i3)
        v = fmaf(dB[(j2 * J3 + j3) * MB + (i2 * I3 + i3) * R2 + a], W3[j3 * (I3 * R3) + i3 * R3 + r3], v);
    d_packed[woff_of<S4>(2) + e] += v;
    return;
  }
  e -= N2;
  if (e < N3) {
    const int r3 = e % R3, i3 = (e / R3) % I3, j3 = e / (R3 * I3);
    for (int j2 = 0; j2 < J2; ++j2)
      for (int i2 = 0; i2 < I2; ++i2)
        for (int a = 0; a < R2; ++a)
          v = fmaf(dB[(j2 * J3 + j3) * MB + (i2 * I3 + i3) * R2 + a], W2[(j2 * R3 + r3) * (I2 * R2) + i2 * R2 + a], v);
    d_packed[woff_of<S4>(3) + e] += v;
  }
}

// ---- dispatch -----------------------------------------------------------------------------------------------------------
namespace {
using S4 = ShpH1024R32L;
using S3 = ShpH1024R32L_M;
using S2 = ShpH1024R32L_M2;
using ST = ShpH1024R32L_T;

constexpr size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr size_t B3 = al256((size_t)merged_elems<S3>() * sizeof(float));
constexpr size_t B2 = al256((size_t)merged2_elems<S2>() * sizeof(float));
constexpr size_t BT = al256((size_t)(St<ST, 0>::K * St<ST, 0>::M + St<ST, 1>::K * St<ST, 1>::M) * sizeof(float));
constexpr size_t BDA = (size_t)St<S2, 0>::K * St<S2, 0>::M * sizeof(float);      // 512 x 64
constexpr size_t BDB = (size_t)St<S2, 1>::K * St<S2, 1>::M * sizeof(float);      // 64 x 2048
constexpr size_t BDW = (size_t)1024 * 4096 * sizeof(float);                      // dense gradient

int device_cus() {
  const int cus = device_cu_count();
  return cus;
}

bool no_bigb() { return opt(OPT_NO_BIGB) != 0; }      // A/B switch: any-shape backward for the big shape
}  // namespace

bool big_rnn_bwd_available(const RnnShape& rs, int dtype) {
  if ((dtype != TTRNN_F32 && dtype != TTRNN_BF16) || rs.B < 1 || rs.T < 1 || rs.cell != TTRNN_LSTM || no_bigb())
    return false;
  return shape_matches<S4>(rs.hid_s);
}

size_t big_rnn_bwd_workspace(const RnnShape& rs) {
  // + tagged exchange words of the pair kernel + the fp16 fragments of its split-mode version
  return B3 + BT + al256((size_t)rs.B * 2 * rs.H * sizeof(unsigned long long)) + bigbh_workspace_bytes();
}

template <typename TS>
static int launch_bigb_t(const RnnShape& rs, const void* c0, const float* packed_hid, const float* reserve,
                         const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, void* d_h0, void* d_c0,
                         void* ws, hipStream_t stream, unsigned* colmax) {
  float* m3 = (float*)ws;
  float* mT = (float*)((char*)ws + B3);
  unsigned long long* hxb = (unsigned long long*)((char*)ws + B3 + BT);
  hipLaunchKernelGGL((k_merge_cores01<S4, S3>), dim3((merged_elems<S3>() + 255) / 256), dim3(256), 0, stream,
                     packed_hid, m3);
  hipLaunchKernelGGL((k_bigb_prep<S3, ST>), dim3((int)(BT / sizeof(float) + 255) / 256), dim3(256), 0, stream, m3, mT);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  // two workgroups per sample: CU count and the runtime's occupancy for both kernels that may run (the fp16 one and its
  // fp32-MFMA fallback); OPT_BIG_NO_PAIR: A/B switch, one workgroup per sample
  const bool halfk = opt(OPT_FP32_MATH) == TTRNN_MATH_SPLIT && !opt(OPT_BIG_FP32_MFMA);
  constexpr size_t lds_p32 = bigb_lds_bytes<ST, 2>() > 100 * 1024 ? bigb_lds_bytes<ST, 2>() : 100 * 1024;
  const bool pair = 2 * rs.B <= device_cus() && !opt(OPT_BIG_NO_PAIR) &&
                    (!halfk || bigbh_pair_resident(sizeof(TS) == 4 ? TTRNN_F32 : TTRNN_BF16, rs.B)) &&
                    ensure_dynamic_lds(reinterpret_cast<const void*>(k_lstm_bwd_big<ST, 2, TS>), lds_p32) == TTRNN_OK &&
                    resident_at_once(reinterpret_cast<const void*>(k_lstm_bwd_big<ST, 2, TS>), FAST_NT, lds_p32, 2L * rs.B);
  if (pair) {
    if (hipMemsetAsync(hxb, 0, (size_t)rs.B * 2 * rs.H * sizeof(unsigned long long), stream) != hipSuccess)
      return TTRNN_ERR_LAUNCH;
    // split mode: both stages on two-piece fp16 operands (ttrnn_fast_bigbh.hip); OPT_BIG_FP32_MFMA: A/B switch
    const float* guard = nullptr;
    int guard_rows = 0;
    if (halfk) {
      void* scr = (char*)hxb + al256((size_t)rs.B * 2 * rs.H * sizeof(unsigned long long));
      const int sh = launch_lstm_bwd_big2h(rs, sizeof(TS) == 4 ? TTRNN_F32 : TTRNN_BF16, c0, mT, reserve, d_out, d_hT, d_cT,
                                           dg_in, d_h0, d_c0, hxb, scr, stream, colmax);
      if (sh != TTRNN_OK) return sh;
      // ... followed by the fp32-MFMA pair kernel as its fallback: exactly one of the two runs (big_guard_tripped, decided on
      // the device from the representation error of the fp16 pieces; the other returns at once, a few microseconds)
      guard = bigbh_guard_rows(scr, &guard_rows);
    }
    // 64 KB image; 100 KB requested so that a second workgroup cannot share the CU (see the forward pair kernel)
    constexpr size_t lds = bigb_lds_bytes<ST, 2>() > 100 * 1024 ? bigb_lds_bytes<ST, 2>() : 100 * 1024;
    {
      if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_lstm_bwd_big<ST, 2, TS>), lds) != TTRNN_OK)
        return TTRNN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((k_lstm_bwd_big<ST, 2, TS>), dim3(2 * rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T,
                       (const TS*)c0, mT, reserve, (const TS*)d_out, (const TS*)d_hT, (const TS*)d_cT, dg_in,
                       (TS*)d_h0, (TS*)d_c0, hxb, guard, guard_rows, device_status_ptr(), colmax);
  } else {
    constexpr size_t lds = bigb_lds_bytes<ST, 1>();
    {
      if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_lstm_bwd_big<ST, 1, TS>), lds) != TTRNN_OK)
        return TTRNN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((k_lstm_bwd_big<ST, 1, TS>), dim3(rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T, (const TS*)c0,
                       mT, reserve, (const TS*)d_out, (const TS*)d_hT, (const TS*)d_cT, dg_in, (TS*)d_h0, (TS*)d_c0,
                       hxb, (const float*)nullptr, 0, device_status_ptr(), colmax);
  }
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

int launch_rnn_bwd_big(const RnnShape& rs, int dtype, const void* c0, const float* packed_hid, const float* reserve,
                       const void* d_out, const void* d_hT, const void* d_cT, float* dg_in, float* dg_hid, void* d_h0,
                       void* d_c0, void* ws, hipStream_t stream, float* stats) {
  // by-products (ttrnn_rnn_backward_ex): the column maxima of the gate gradients, stats rows 0 and 1 (LSTM: the same)
  unsigned* colmax = reinterpret_cast<unsigned*>(stats);
  const int GH = 4 * rs.H;
  if (stats && hipMemsetAsync(stats, 0, (size_t)2 * GH * sizeof(float), stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
  int st = dtype == TTRNN_F32 ? launch_bigb_t<float>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, d_h0, d_c0, ws,
                                                     stream, colmax)
                              : launch_bigb_t<bf16_t>(rs, c0, packed_hid, reserve, d_out, d_hT, d_cT, dg_in, d_h0, d_c0, ws,
                                                      stream, colmax);
  if (st == TTRNN_OK && stats) st = launch_bwd_stats_finish(TTRNN_LSTM, rs.B, GH, nullptr, stats, stream);
  if (st != TTRNN_OK) return st;
  if (dg_hid && dg_hid != dg_in &&
      hipMemcpyAsync(dg_hid, dg_in, (size_t)rs.B * rs.T * 4 * rs.H * sizeof(float), hipMemcpyDeviceToDevice, stream) !=
          hipSuccess)
    return TTRNN_ERR_LAUNCH;
  return TTRNN_OK;
}

bool big_ttlinear_bwd_available(const TtShape& s, int dtype, int dy_dtype) {
  if ((dtype != TTRNN_F32 && dtype != TTRNN_BF16) || dy_dtype != TTRNN_F32 || no_bigb()) return false;
  return shape_matches<S4>(s);
}

size_t big_ttlinear_bwd_workspace_bytes(const TtShape& s) {
  return shape_matches<S4>(s) ? B3 + B2 + BT + BDA + BDB + BDW + dense_wgrad_scratch_bytes(1024, 4096) : 0;
}

template <typename TS>
static int launch_bigw_t(int64_t n_rows, const float* packed, const void* x, const void* dy, void* dx, float* d_packed,
                         float* d_bias, void* ws, hipStream_t stream, const unsigned* x_colmax, const unsigned* dy_colmax,
                         int shift_T, const void* shift_first) {
  const bool split_math = ttrnn_get_fp32_math() == TTRNN_MATH_SPLIT;
  float* m3 = (float*)ws;
  float* m2 = (float*)((char*)ws + B3);
  float* mT = (float*)((char*)ws + B3 + B2);
  float* dA = (float*)((char*)ws + B3 + B2 + BT);
  float* dB = (float*)((char*)dA + BDA);
  hipLaunchKernelGGL((k_merge_cores01<S4, S3>), dim3((merged_elems<S3>() + 255) / 256), dim3(256), 0, stream, packed, m3);
  hipLaunchKernelGGL((k_bigb_prep<S3, ST>), dim3((int)(BT / sizeof(float) + 255) / 256), dim3(256), 0, stream, m3, mT);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  const int cus = device_cus();
  if (dx) {
    constexpr size_t lds = bigb_lds_bytes<ST, 1>();
    {
      if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_ttlinear_dx_big<ST, TS>), lds) != TTRNN_OK)
        return TTRNN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((k_ttlinear_dx_big<ST, TS>), dim3((int)(n_rows < cus ? n_rows : cus)), dim3(FAST_NT), lds, stream,
                       n_rows, mT, (const float*)dy, (TS*)dx);
    if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  }
  if (!d_packed) return TTRNN_OK;
  if (!opt(OPT_BIGW_SLICES)) {      // A/B switch: per-row merged-chain kernel instead of the dense GEMM
    float* BmN = m2;                                        // the natural-order cores reuse the m2 / dA / dB regions
    float* AT = m2 + St<S2, 1>::K * St<S2, 1>::M;
    float* dWf = (float*)((char*)dB + BDB);
    hipLaunchKernelGGL((k_bigw_natural<S3>), dim3((merged2_elems<S2>() + 255) / 256), dim3(256), 0, stream, m3, BmN, AT);
    const int sd = launch_dense_wgrad(sizeof(TS) == 4 ? TTRNN_F32 : TTRNN_BF16, n_rows, 1024, 4096, x, (const float*)dy, dWf,
                                      d_bias, stream, split_math, (float*)((char*)dWf + BDW), x_colmax, dy_colmax, shift_T,
                                      shift_first);
    if (sd != TTRNN_OK) return sd;
    hipLaunchKernelGGL(k_bigw_proj_a, dim3(16 * 64), dim3(256), 0, stream, dWf, BmN, dA);
    hipLaunchKernelGGL(k_bigw_proj_b, dim3(64 * 64), dim3(256), 0, stream, dWf, AT, dB);
    if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  } else {
  hipLaunchKernelGGL((k_merge_cores_last<S3, S2>), dim3((merged2_elems<S2>() + 255) / 256), dim3(256), 0, stream, m3, m2);
  if (hipMemsetAsync(dA, 0, BDA + BDB, stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
  constexpr size_t lds = (size_t)BigW::TOTAL * sizeof(float);
  {
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_ttlinear_wgrad_big<S2, ST, TS>), lds) != TTRNN_OK)
      return TTRNN_ERR_LAUNCH;
  }
  int chunks = cus / 4;
  if (chunks < 1) chunks = 1;
  if (n_rows < chunks) chunks = (int)n_rows;
  const int rows_per_wg = (int)((n_rows + chunks - 1) / chunks);
  chunks = (int)((n_rows + rows_per_wg - 1) / rows_per_wg);
  hipLaunchKernelGGL((k_ttlinear_wgrad_big<S2, ST, TS>), dim3(4 * chunks), dim3(FAST_NT), lds, stream, n_rows,
                     rows_per_wg, m2, mT, (const TS*)x, (const float*)dy, dA, dB, d_bias);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  }
  constexpr int NW = St<S4, 0>::K * St<S4, 0>::M + St<S4, 1>::K * St<S4, 1>::M + St<S4, 2>::K * St<S4, 2>::M +
                     St<S4, 3>::K * St<S4, 3>::M;
  hipLaunchKernelGGL((k_bigw_finish<S4>), dim3((NW + 255) / 256), dim3(256), 0, stream, packed, dA, dB, d_packed);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// dx and / or (d_packed [+ d_bias]); a bias gradient alone is not offered
int launch_ttlinear_bwd_big(const TtShape& s, int dtype, int64_t n_rows, const float* packed, const void* x,
                            const void* dy, void* dx, float* d_packed, float* d_bias, void* ws, hipStream_t stream,
                            const unsigned* x_colmax, const unsigned* dy_colmax, int shift_T, const void* shift_first) {
  (void)s;
  if (shift_T > 0 && opt(OPT_BIGW_SLICES)) return TTRNN_ERR_UNSUPPORTED;
  return dtype == TTRNN_F32 ? launch_bigw_t<float>(n_rows, packed, x, dy, dx, d_packed, d_bias, ws, stream, x_colmax,
                                                   dy_colmax, shift_T, shift_first)
                            : launch_bigw_t<bf16_t>(n_rows, packed, x, dy, dx, d_packed, d_bias, ws, stream, x_colmax,
                                                    dy_colmax, shift_T, shift_first);
}

}  // namespace ttrnn
