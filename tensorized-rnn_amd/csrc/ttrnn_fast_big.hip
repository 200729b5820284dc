// ttrnn_fast_big.hip — MFMA kernels for TT shapes that do not fit on chip (BASELINE cfg5: H = 1024, d = 4, r = 32).
//
// One sample's chain intermediates are 256 KB (> 160 KB LDS) and one TT-matrix is 405 KB of cores (> the register
// file), so the "everything resident" layout of ttrnn_fast.hip cannot hold.  These variants keep the same MFMA stage
// code (ttrnn_mfma.h:lin_stage — runtime loop over row tiles, two tiles in flight, chunked fragment reads) and the
// same persistent one-workgroup-per-sample structure, but
//   * the stage images ping-pong through a per-workgroup slab of the global workspace (2 x 256 KB, L2 / Infinity
//     Cache resident: 128 workgroups x 512 KB = 64 MB), a plain workgroup barrier orders the hand-off;
//   * core fragments are re-streamed from L2 at the start of every stage (405 KB per step per workgroup) instead
//     of living in VGPRs; the opaque per-stage lane id keeps hipcc from hoisting (and spilling) them;
//   * h, c and the gate vector stay on chip (LDS / registers) exactly as in the small-shape kernels.
// At 70 MFLOP per sample-step this regime is MFMA-bound (57 us of MFMA per step and CU), not latency-bound, so the
// L2 round trips are affordable.  Replaces the same reference code as ttrnn_fast.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn.h"
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_big.h"

namespace ttrnn {


// fragments of the m-tiles {wave + 8(x0 + x) : x < XC} of stage k (chunk of a wave's tile list)
template <class S, int k, int XC, int NW_>
__device__ __forceinline__ void load_wfrag_x(float (&w)[NW_], const float* packed, int wave, int lane, int x0) {
  using T = St<S, k>;
  static_assert(!T::SPLIT && NW_ == XC * T::NSTEP, "chunked stages own whole m-tiles");
  const int r = lane & 15, q = lane >> 4;
  const float* W = packed + woff_of<S>(k);
#pragma unroll
  for (int x = 0; x < XC; ++x) {
    const int mt = wave + FAST_NW * (x0 + x);
    const int m = 16 * mt + r;
#pragma unroll
    for (int u = 0; u < T::NU; ++u)
#pragma unroll
      for (int e = 0; e < T::WV; ++e) {
        const int kk = (4 * u + q) * T::WV + e;
        const bool okw = mt < T::MT && m < T::M && kk < T::K;
        const float wv = W[okw ? kk * T::M + m : 0];
        w[x * T::NSTEP + u * T::WV + e] = okw ? wv : 0.f;
      }
  }
}

// lin_stage (ttrnn_mfma.h) restricted to that chunk of m-tiles, one sample (NB = 1)
// out_row0: first row of the next stage's image that lives at Cout (a workgroup that owns only a slice of the rows)
template <class S, int k, int G, int XC, int NW_>
__device__ __forceinline__ void lin_stage_x(const float (&w)[NW_], const float* Ain, float* Cout, int wave, int lane,
                                            int ilv_mode, int x0, int out_row0 = 0) {
  using T = St<S, k>;
  static_assert(NW_ == XC * T::NSTEP, "fragment array size");
  constexpr int TOT = T::ROWS;
  constexpr int RT_ALL = (TOT + 15) / 16;
  constexpr int OUT = out_size_of<S>();
  const int c = lane & 15, q = lane >> 4;
  constexpr int UC = chunk_of(T::NU);
  for (int rtb = 0; rtb < RT_ALL; rtb += 2) {
    int Rr[2];
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int R = 16 * (rtb + y) + c;
      Rr[y] = R < TOT ? R : TOT - 1;
    }
    f32x4 acc[XC][2];
#pragma unroll
    for (int x = 0; x < XC; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int u0 = 0; u0 < T::NU; u0 += UC) {
      float af[2][UC * T::WV];
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int u = 0; u < UC; ++u) {
          const float* p = Ain + a_off<T::KP>(Rr[y], (4 * (u0 + u) + q) * T::WV);
          if constexpr (T::WV == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p);
            af[y][4 * u + 0] = v[0]; af[y][4 * u + 1] = v[1]; af[y][4 * u + 2] = v[2]; af[y][4 * u + 3] = v[3];
          } else if constexpr (T::WV == 2) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(p);
            af[y][2 * u + 0] = v[0]; af[y][2 * u + 1] = v[1];
          } else {
            af[y][u] = *p;
          }
        }
#pragma unroll
      for (int x = 0; x < XC; ++x)
#pragma unroll
        for (int s2 = 0; s2 < UC * T::WV; ++s2) {
          const float wv = w[x * T::NSTEP + u0 * T::WV + s2];
          acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, af[0][s2], acc[x][0], 0, 0, 0);
          if (RT_ALL > 1) acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, af[1][s2], acc[x][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int x = 0; x < XC; ++x) {
      const int mt = wave + FAST_NW * (x0 + x);
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const f32x4 a = acc[x][y];
        const int rt = rtb + y;
        const int R = 16 * rt + c;
        const int m0 = 16 * mt + 4 * q;
        if (mt < T::MT && rt < RT_ALL && R < TOT && m0 < T::M) {
          if constexpr (k > 0) {
            using N = St<S, k - 1>;
            const int i = m0 / T::R, a0 = m0 % T::R;
            const int f = i * (T::ROWS * T::R) + R * T::R + a0;
            *reinterpret_cast<f32x4*>(Cout + a_off<N::KP>(f / N::K - out_row0, f % N::K)) = a;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (m0 + j < T::M) Cout[ytile_index<G, OUT>(0, (m0 + j) * T::ROWS + R, ilv_mode)] = a[j];
          }
        }
      }
    }
  }
}

// fragment-ordered buffers (two-core matrices, see k_merge_cores_last): 16-byte loads
template <class S>
constexpr bool big_frag_order() { return S::D == 2; }

template <class S, int k, int NW_>
__device__ __forceinline__ void load_wfrag_f(float (&w)[NW_], const float* packed, int wave, int lane) {
  using T = St<S, k>;
  static_assert(NW_ == T::NWREG && T::WV == 4, "fragment array size");
  const f32x4* F = reinterpret_cast<const f32x4*>(packed + woff_of<S>(k));
#pragma unroll
  for (int x = 0; x < T::XM; ++x) {
    const int mt = T::SPLIT ? (wave % T::MT) : (wave + FAST_NW * x);
#pragma unroll
    for (int u = 0; u < T::NU; ++u) {
      const f32x4 v = F[(size_t)((mt < T::MT ? mt : 0) * T::NU + u) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[x * T::NSTEP + 4 * u + e] = mt < T::MT ? v[e] : 0.f;
    }
  }
}

template <class S, int k, int XC, int NW_>
__device__ __forceinline__ void load_wfrag_xf(float (&w)[NW_], const float* packed, int wave, int lane, int x0) {
  using T = St<S, k>;
  static_assert(!T::SPLIT && NW_ == XC * T::NSTEP && T::WV == 4, "chunked stages own whole m-tiles");
  const f32x4* F = reinterpret_cast<const f32x4*>(packed + woff_of<S>(k));
#pragma unroll
  for (int x = 0; x < XC; ++x) {
    const int mt = wave + FAST_NW * (x0 + x);
#pragma unroll
    for (int u = 0; u < T::NU; ++u) {
      const f32x4 v = F[(size_t)((mt < T::MT ? mt : 0) * T::NU + u) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[x * T::NSTEP + 4 * u + e] = mt < T::MT ? v[e] : 0.f;
    }
  }
}

// One chain stage.  Stages whose fragment set (XM m-tiles x NSTEP k-steps per wave) would not fit the register file
// go through it in chunks of XC m-tiles.
template <class S, int k, int G>
__device__ __forceinline__ void big_stage(const float* packed, const float* in, float* out, int wave, int lane,
                                          int ilv_mode) {
  using T = St<S, k>;
  if constexpr (T::NWREG <= 128) {
    int z = 0;
    asm volatile("" : "+v"(z));          // keep the fragment loads and their address arithmetic inside this stage
    float w[nwreg<S, k>()];
    if constexpr (big_frag_order<S>()) load_wfrag_f<S, k>(w, packed, wave, lane + z);
    else load_wfrag<S, k>(w, packed, wave, lane + z);
    lin_stage<S, k, 1, G>(w, in, out, wave, lane + z, ilv_mode);
  } else {
    constexpr int XC = 64 / T::NSTEP > 0 ? 64 / T::NSTEP : 1;       // <= 64 fragment registers per chunk
    static_assert(T::XM % XC == 0, "chunking of the m-tile list");
    for (int x0 = 0; x0 < T::XM; x0 += XC) {
      int z = 0;
      asm volatile("" : "+v"(z));        // per-chunk opaque lane id: no hoisting of the next chunk's loads
      float w[XC * T::NSTEP];
      if constexpr (big_frag_order<S>()) load_wfrag_xf<S, k, XC>(w, packed, wave, lane + z, x0);
      else load_wfrag_x<S, k, XC>(w, packed, wave, lane + z, x0);
      lin_stage_x<S, k, G, XC>(w, in, out, wave, lane + z, ilv_mode, x0);
    }
  }
}

template <class S>
constexpr int big_mid() {      // floats of the largest stage image
  int best = in_size_of<S>();
  for (int k = 1; k < S::D; ++k) {
    const int e = rows_of<S>(k) * S::I[k] * S::R[k];
    if (e > best) best = e;
  }
  return best;
}

// a two-core matrix whose single intermediate image (+ input image, h and gate vectors) fits the 160 KB of LDS
template <class S>
constexpr bool big_lds_images() {
  return S::D == 2 && (size_t)(big_mid<S>() + in_size_of<S>() + in_size_of<S>() + out_size_of<S>()) * sizeof(float) <=
                          156 * 1024;
}


// chain y = TT(packed) v for ONE sample whose input image already sits in img0 (a_off<KP> layout);
// the result (flat o, or gate-interleaved when G > 0) is written to `res`.
template <class S, int G>
__device__ __forceinline__ void big_chain(const float* packed, float* img0, float* imgA, float* imgB, float* res,
                                          int wave, int lane, int ilv_mode) {
  constexpr int D = S::D;
  if constexpr (D == 1) {
    big_stage<S, 0, G>(packed, img0, res, wave, lane, ilv_mode);
  } else if constexpr (D == 2) {
    big_stage<S, 1, G>(packed, img0, imgA, wave, lane, ilv_mode);
    __syncthreads();
    big_stage<S, 0, G>(packed, imgA, res, wave, lane, ilv_mode);
  } else if constexpr (D == 3) {
    big_stage<S, 2, G>(packed, img0, imgA, wave, lane, ilv_mode);
    __syncthreads();
    big_stage<S, 1, G>(packed, imgA, imgB, wave, lane, ilv_mode);
    __syncthreads();
    big_stage<S, 0, G>(packed, imgB, res, wave, lane, ilv_mode);
  } else {
    big_stage<S, 3, G>(packed, img0, imgA, wave, lane, ilv_mode);
    __syncthreads();
    big_stage<S, 2, G>(packed, imgA, imgB, wave, lane, ilv_mode);
    __syncthreads();
    big_stage<S, 1, G>(packed, imgB, imgA, wave, lane, ilv_mode);
    __syncthreads();
    big_stage<S, 0, G>(packed, imgA, res, wave, lane, ilv_mode);
  }
}

// ---- batched input projection: gin[n][H][4] = TT(packed_in) x[n] (bias is added by the recurrent kernel) ----------
template <class S, int G, typename TS>
__global__ void __launch_bounds__(FAST_NT) k_ttlinear_fwd_big(int64_t n_rows, const float* __restrict__ packed,
                                                              const TS* __restrict__ x, float* __restrict__ y,
                                                              float* __restrict__ ws, int ilv_mode) {
  constexpr int IN = in_size_of<S>(), OUT = out_size_of<S>();
  constexpr int MID = big_mid<S>();
  using SL = St<S, S::D - 1>;
  static_assert(SL::K % 4 == 0, "big-shape kernels need an unpadded first image");
  constexpr int YT = G == 0 ? OUT : (OUT / (G > 0 ? G : 1)) * 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // two-core matrices need ONE intermediate image (plus the input image): both stay in LDS when they fit; deeper
  // chains ping-pong through the L2-resident slab
  extern __shared__ __attribute__((aligned(16))) float big_lds[];
  constexpr bool LDSIMG = big_lds_images<S>();
  float* img0 = LDSIMG ? big_lds : ws + (size_t)blockIdx.x * 3 * MID;
  float* imgA = LDSIMG ? big_lds + IN : img0 + MID;
  float* imgB = imgA + MID;
  for (int64_t n = blockIdx.x; n < n_rows; n += gridDim.x) {
    for (int e = tid; e < IN; e += FAST_NT) img0[a_off<SL::KP>(e / SL::K, e % SL::K)] = ld(x, (size_t)n * IN + e);
    __syncthreads();
    big_chain<S, G>(packed, img0, imgA, imgB, y + (size_t)n * YT, wave, lane, ilv_mode);
    __syncthreads();
  }
}

// ---- persistent recurrent kernel --------------------------------------------------------------------------------
template <class S, int CELL, typename TS>
__global__ void __launch_bounds__(FAST_NT) k_rnn_fwd_big(int B, int T, const float* __restrict__ gin,
                                                         const TS* __restrict__ h0, const TS* __restrict__ c0,
                                                         const float* __restrict__ packed_hid,
                                                         const TS* __restrict__ bias_in, const TS* __restrict__ bias_hid,
                                                         TS* __restrict__ out, TS* __restrict__ hT, TS* __restrict__ cT,
                                                         float* __restrict__ reserve, float* __restrict__ ws) {
  constexpr int H = in_size_of<S>();
  constexpr int G = CELL == TTRNN_LSTM ? 4 : 3;
  constexpr int GH = G * H;
  static_assert(out_size_of<S>() == GH, "TT output size must be n_gates * hidden");
  constexpr int HPT = (H + FAST_NT - 1) / FAST_NT;
  constexpr int MID = big_mid<S>();
  using SL = St<S, S::D - 1>;
  static_assert(SL::K % 4 == 0, "big-shape kernels need an unpadded first image");

  __shared__ __attribute__((aligned(16))) float hbuf[H];
  __shared__ __attribute__((aligned(16))) float gbuf[GH];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t b = blockIdx.x;
  extern __shared__ __attribute__((aligned(16))) float big_lds[];
  constexpr bool LDSIMG = big_lds_images<S>();              // see k_ttlinear_fwd_big
  float* imgA = LDSIMG ? big_lds : ws + (size_t)blockIdx.x * 2 * MID;
  float* imgB = imgA + MID;
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gin);

  float cst[HPT], hst[HPT], bh[HPT][G];
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int hid = tid + u * FAST_NT;
    const bool ok = hid < H;
    hst[u] = (ok && h0) ? ld(h0, b * H + hid) : 0.f;
    cst[u] = (ok && c0 && CELL == TTRNN_LSTM) ? ld(c0, b * H + hid) : 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g)
      bh[u][g] = (ok && bias_hid ? ld(bias_hid, g * H + hid) : 0.f) + (ok && bias_in ? ld(bias_in, g * H + hid) : 0.f);
    if (ok) hbuf[a_off<SL::KP>(hid / SL::K, hid % SL::K)] = hst[u];
  }
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    big_chain<S, 0>(packed_hid, hbuf, imgA, imgB, gbuf, wave, lane, 0);
    __syncthreads();
    const size_t bt = b * T + t;
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
      const int hid = tid + u * FAST_NT;
      if (hid < H) {
        const f32x4 gi = gin4[bt * H + hid];          // slots i,g,f,o (LSTM) / r,z,n,- (GRU)
        float hy;
        if constexpr (CELL == TTRNN_LSTM) {
          const float ig = bsigmoid(gi[0] + gbuf[hid] + bh[u][0]);                 // lstm.py:26
          const float fg = bsigmoid(gi[2] + gbuf[H + hid] + bh[u][1]);             // lstm.py:27
          const float gg = btanh(gi[1] + gbuf[2 * H + hid] + bh[u][2]);            // lstm.py:28
          const float og = bsigmoid(gi[3] + gbuf[3 * H + hid] + bh[u][3]);         // lstm.py:29
          const float cy = fg * cst[u] + ig * gg;                                  // lstm.py:31
          hy = og * btanh(cy);                                                     // lstm.py:32
          cst[u] = cy;
          if (reserve) {
            float* rv = reserve + res_gate(bt, H, hid);
            rv[0] = ig; rv[1] = gg; rv[2] = fg; rv[3] = og;
            reserve[res_cell((size_t)B * T, bt, H, hid)] = cy;
          }
        } else {
          // bias_in must not be scaled by r: split the fused bias again for the n gate (gru.py:42-43)
          const float bn_in = bias_in ? ld(bias_in, 2 * H + hid) : 0.f;
          const float hn = gbuf[2 * H + hid] + bh[u][2] - bn_in;
          const float rg = bsigmoid(gi[0] + gbuf[hid] + bh[u][0]);
          const float zg = bsigmoid(gi[1] + gbuf[H + hid] + bh[u][1]);
          const float ng = btanh(gi[2] + bn_in + rg * hn);
          hy = (1.0f - zg) * ng + zg * hst[u];
          if (reserve) *reinterpret_cast<f32x4*>(reserve + (bt * H + hid) * 4) = f32x4{rg, zg, ng, hn};
        }
        if (out) st(out, bt * H + hid, hy);          // out == NULL: final state only (ttrnn_rnn_out_optional)
        hy = round_as(out, hy);
        hst[u] = hy;
        hbuf[a_off<SL::KP>(hid / SL::K, hid % SL::K)] = hy;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int hid = tid + u * FAST_NT;
    if (hid < H) {
      if (hT) st(hT, b * H + hid, hst[u]);
      if (CELL == TTRNN_LSTM && cT) st(cT, b * H + hid, cst[u]);
    }
  }
}


// ---- two workgroups per sample (two-core matrices, LSTM) ------------------------------------------------------------
// With B = 128 samples the one-workgroup-per-sample kernel leaves half of the 256 CUs idle.  Both stages of the
// two-core chain split cleanly over the rows of the SECOND stage: stage 1's output features are (i1, r1) and become,
// viewed flat, the rows i1 of stage 0, whose rows are independent and end up as the low part of the hidden index
// (hid = (m0 % 16) * I1 + i1).  Workgroup `half` of a pair therefore computes the m-tiles of stage 1 with
// i1 in [half*I1/2, (half+1)*I1/2) (half of W_1 streamed), stage 0 on those rows, and the gates of those H/2 hidden
// units — no exchange inside a step.  Once per step the halves of h_t are swapped through double-buffered global rows of
// self-validating 64-bit words (value, step tag) written and polled with relaxed agent-scope atomics (both workgroups
// are resident: 2B <= #CUs, one workgroup per CU).
template <class S, typename TS>
__global__ void __launch_bounds__(FAST_NT) k_lstm_fwd_big2(int B, int T, const float* __restrict__ gin,
                                                           const TS* __restrict__ h0, const TS* __restrict__ c0,
                                                           const float* __restrict__ packed,
                                                           const TS* __restrict__ bias_in,
                                                           const TS* __restrict__ bias_hid, TS* __restrict__ out,
                                                           TS* __restrict__ hT, TS* __restrict__ cT,
                                                           float* __restrict__ reserve,
                                                           unsigned long long* __restrict__ hx,
                                                           unsigned* __restrict__ status) {
  static_assert(S::D == 2 && big_frag_order<S>(), "two-core matrices in fragment order");
  using T1 = St<S, 1>;
  using T0 = St<S, 0>;
  constexpr int H = in_size_of<S>(), I1 = S::I[1], HR = T0::ROWS / 2;      // HR: stage-0 rows per workgroup
  static_assert(out_size_of<S>() == 4 * H && T0::ROWS == I1 && HR % 16 == 0 && H / 2 == FAST_NT && T0::M == 64 &&
                    T0::MT * (HR / 16) == FAST_NW && !T1::SPLIT && T1::XM % 2 == 0 && T1::K % 4 == 0,
                "pair layout");
  constexpr int XC = 64 / T1::NSTEP > 0 ? 64 / T1::NSTEP : 1;
  static_assert((T1::XM / 2) % XC == 0, "chunking of the half m-tile list");

  __shared__ __attribute__((aligned(16))) float hbuf[H];                 // h_{t-1}, image of stage 1
  __shared__ __attribute__((aligned(16))) float gb[T0::M * HR];          // pre-activations [m0][local row]
  extern __shared__ __attribute__((aligned(16))) float big_lds[];        // image of stage 0: [HR rows][K0]
  float* img = big_lds;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x >> 1;
  const int half = blockIdx.x & 1;
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gin);

  // gate phase: thread tid owns hid = (tid / HR) * I1 + half*HR + tid % HR  (m0 % 16 = tid / HR, local row tid % HR)
  const int rl = tid % HR, mq = tid / HR;
  const int hid = mq * I1 + half * HR + rl;
  const int hidp = mq * I1 + (1 - half) * HR + rl;                       // the partner's unit at the same position
  float hst = h0 ? ld(h0, b * H + hid) : 0.f;
  float cst = c0 ? ld(c0, b * H + hid) : 0.f;
  float bh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
    bh[g] = (bias_hid ? ld(bias_hid, g * H + hid) : 0.f) + (bias_in ? ld(bias_in, g * H + hid) : 0.f);
  hbuf[a_off<T1::KP>(hid / T1::K, hid % T1::K)] = hst;
  hbuf[a_off<T1::KP>(hidp / T1::K, hidp % T1::K)] = h0 ? ld(h0, b * H + hidp) : 0.f;
  f32x4 gi = T > 0 ? gin4[(b * T) * H + hid] : f32x4{0.f, 0.f, 0.f, 0.f};     // slots i,g,f,o; prefetched a step ahead
  bool dead = false;
  __syncthreads();

  // Core fragments travel AHEAD of their use (streaming 384 KB per step from L2 at ~70 GB/s per CU takes about as long as
  // the MFMAs; loaded right before use, as in the one-workgroup kernel, the two did not overlap).  Stage 1 goes through
  // its 8 m-tiles in four chunks of two with two register slots (the next chunk is requested before the current one is
  // multiplied; chunk 0 of step t+1 during stage 0 of step t); stage 0's 32 fragment groups go through three slots of
  // eight, the first three requested during stage 1.
  constexpr int XC2 = 2;
  static_assert(T1::XM / 2 == 4 * XC2 && T0::NU == 32, "four chunks of stage 1, four quarters of stage 0");
  const f32x4* F0 = reinterpret_cast<const f32x4*>(packed + woff_of<S>(0));
  const int mt0 = wave % T0::MT, rtl = wave / T0::MT;
  const int xb = half * (T1::XM / 2);
  const int row0 = 16 * rtl + c;
  float s1a[XC2 * T1::NSTEP], s1b[XC2 * T1::NSTEP];
  f32x4 q0[8], q1[8], q2[8];
  f32x4 acc0, acc1;
  auto loadq = [&](f32x4 (&qq)[8], int i, int zz) {
#pragma unroll
    for (int u = 0; u < 8; ++u) qq[u] = F0[(size_t)(mt0 * T0::NU + 8 * i + u) * 64 + lane + zz];
  };
  auto mm = [&](const f32x4 (&qq)[8], int i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(img + a_off<T0::KP>(row0, (4 * (8 * i + u) + q) * 4));
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qq[u][0], af[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qq[u][1], af[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qq[u][2], af[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qq[u][3], af[3], acc1, 0, 0, 0);
    }
  };
  load_wfrag_xf<S, 1, XC2>(s1a, packed, wave, lane, xb);

  for (int t = 0; t < T; ++t) {
    int z = 0;
    asm volatile("" : "+v"(z));            // per-step opaque lane id: the fragment loads stay where they are written
    // ---- stage 1: this workgroup's half of the m-tiles, result = its HR rows of the stage-0 image -------------------
    load_wfrag_xf<S, 1, XC2>(s1b, packed, wave, lane + z, xb + 2);
    lin_stage_x<S, 1, 0, XC2>(s1a, hbuf, img, wave, lane, 0, xb, half * HR);
    load_wfrag_xf<S, 1, XC2>(s1a, packed, wave, lane + z, xb + 4);
    loadq(q0, 0, z);
    lin_stage_x<S, 1, 0, XC2>(s1b, hbuf, img, wave, lane, 0, xb + 2, half * HR);
    load_wfrag_xf<S, 1, XC2>(s1b, packed, wave, lane + z, xb + 6);
    loadq(q1, 1, z);
    lin_stage_x<S, 1, 0, XC2>(s1a, hbuf, img, wave, lane, 0, xb + 4, half * HR);
    loadq(q2, 2, z);
    lin_stage_x<S, 1, 0, XC2>(s1b, hbuf, img, wave, lane, 0, xb + 6, half * HR);
    __syncthreads();
    // ---- stage 0 on the local rows: wave = (m-tile wave % MT, local row tile wave / MT) ---------------------------
    {
      acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1 = acc0;
      load_wfrag_xf<S, 1, XC2>(s1a, packed, wave, lane + z, xb);         // chunk 0 of step t+1
      mm(q0, 0);
      loadq(q0, 3, z);
      mm(q1, 1);
      mm(q2, 2);
      mm(q0, 3);
      const f32x4 acc = acc0 + acc1;
#pragma unroll
      for (int j = 0; j < 4; ++j) gb[(16 * mt0 + 4 * q + j) * HR + row0] = acc[j];      // [m0][local row]
    }
    __syncthreads();
    // ---- gates (lstm.py:26-32): o = m0*I1 + i1, gate = m0 / 16 ---------------------------------------------------------
    const size_t bt = b * T + t;
    {
      const float ig = bsigmoid(gi[0] + gb[(0 * 16 + mq) * HR + rl] + bh[0]);
      const float fg = bsigmoid(gi[2] + gb[(1 * 16 + mq) * HR + rl] + bh[1]);
      const float gg = btanh(gi[1] + gb[(2 * 16 + mq) * HR + rl] + bh[2]);
      const float og = bsigmoid(gi[3] + gb[(3 * 16 + mq) * HR + rl] + bh[3]);
      const float cy = fg * cst + ig * gg;
      float hy = og * btanh(cy);
      cst = cy;
      if (reserve) {
        float* rv = reserve + res_gate(bt, H, hid);
        rv[0] = ig; rv[1] = gg; rv[2] = fg; rv[3] = og;
        reserve[res_cell((size_t)B * T, bt, H, hid)] = cy;
      }
      if (out) st(out, bt * H + hid, hy);          // out == NULL: final state only (ttrnn_rnn_out_optional)
      hy = round_like(out, hy);                    // what the next step sees: rounded once to the storage type
      hst = hy;
      hbuf[a_off<T1::KP>(hid / T1::K, hid % T1::K)] = hy;
      // ---- swap halves of h_t with the partner workgroup: every unit travels as one 64-bit word (value, step tag) -------
      // Relaxed agent-scope atomics only: they are performed at the memory side, coherent across XCDs, WITHOUT the L2
      // write-back / invalidate a release / acquire pair costs (that flush evicted the streamed cores every step).  The
      // tag makes the word self-validating, so no flag, no counter and no barrier stand between the store and the
      // partner's load: one memory round trip per step instead of three.  Double-buffered by step parity: a slot is
      // rewritten two steps later, which needs the partner's h of the step in between — written after it read this one.
      __hip_atomic_store(hx + (b * 2 + (t & 1)) * H + hid,
                         ((unsigned long long)(unsigned)(t + 1) << 32) | (unsigned long long)__float_as_uint(hy),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t + 1 < T) gi = gin4[(bt + 1) * H + hid];
    }
    {
      const unsigned long long* src = hx + (b * 2 + (t & 1)) * H + hidp;
      unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // bounded: a partner that is not resident (CUs held by another stream / process) must not hang the GPU — after
      // one time-out (~0.1 s) stop waiting for the rest of the launch.  The launch cannot report that through its status
      // (no synchronisation inside the API), so the time-out POISONS the state instead: the missing half of h becomes
      // NaN, which reaches every gate of this sample at the next step and from there `out`, hT and cT — a timed-out
      // launch can never look like a result — and it is COUNTED on the device (TTRNN_STAT_PAIR_TIMEOUTS, read by
      // ttrnn_device_status).
      long spin = 0;
      while (!dead && (unsigned)(v >> 32) != (unsigned)(t + 1)) {
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spin > (1L << 21)) { dead = true; if (status) atomicAdd(status + TTRNN_STAT_PAIR_TIMEOUTS, 1u); }
      }
      hbuf[a_off<T1::KP>(hidp / T1::K, hidp % T1::K)] = dead ? __uint_as_float(0x7FC00000u) : __uint_as_float((unsigned)v);
    }
    __syncthreads();
  }
  if (dead) hst = cst = __uint_as_float(0x7FC00000u);       // T == time-out step: nothing downstream has seen the NaN yet
  if (hT) st(hT, b * H + hid, hst);
  if (cT) st(cT, b * H + hid, cst);
}

// ---- dispatch -------------------------------------------------------------------------------------------------------
static constexpr int BIG_LIN_GRID = 512;     // workgroups of the batched projection (2 per CU)

bool big_rnn_fwd_available(const RnnShape& rs, int dtype) {
  if ((dtype != TTRNN_F32 && dtype != TTRNN_BF16) || rs.B < 1 || rs.T < 1 || rs.cell != TTRNN_LSTM) return false;
  return shape_matches<ShpH1024R32L>(rs.hid_s) && shape_matches<ShpH1024R32L>(rs.in_s);
}

// how many pairs of cores are contracted per launch: 2 (default: first two and last two -> a 2-core matrix),
// 1 (first two only), 0 (the four-core chain as is).  TTRNN_BIG_MERGE is an A/B switch.
static int big_merge_level() { return opt(OPT_BIG_MERGE); }

static size_t big_gemm_bytes(const RnnShape& rs) {
  return gemm_split_identity_bytes(rs.in) + gemm_split_dense_bytes(rs.in, 4 * rs.H) + gemm_split_plane_bytes(rs.in, 4 * rs.H) +
         gemm_half_scratch_bytes((int64_t)rs.B * rs.T, rs.in, 4 * rs.H);
}

size_t big_rnn_fwd_workspace(const RnnShape& rs) {
  constexpr size_t MID = big_mid<ShpH1024R32L>();      // >= big_mid of the merged shape
  const size_t gin = (size_t)rs.B * rs.T * rs.H * 4 * sizeof(float);
  const size_t lin = (size_t)BIG_LIN_GRID * 3 * MID * sizeof(float);
  const size_t rec = (size_t)rs.B * 2 * MID * sizeof(float);
  // per TT-matrix: the 3-core and the 2-core merged packed buffers
  const size_t m3 = ((size_t)merged_elems<ShpH1024R32L_M>() * sizeof(float) + 255) & ~(size_t)255;
  const size_t m2 = ((size_t)merged2_elems<ShpH1024R32L_M2>() * sizeof(float) + 255) & ~(size_t)255;
  const size_t pair = (size_t)rs.B * 2 * rs.H * sizeof(unsigned long long);      // tagged h exchange words of the pair kernel
  // + h exchange rows and counters of the pair kernel + identity rows, dense W_in and its bf16 planes of the GEMM K-in
  return gin + (lin > rec ? lin : rec) + 2 * (m3 + m2) + pair + big_gemm_bytes(rs) + bigh_workspace_bytes();
}

template <typename TS>
static int launch_big_t(const RnnShape& rs, const void* x, const void* h0, const void* c0, const float* packed_in,
                        const void* bias_in, const float* packed_hid, const void* bias_hid, void* out, void* hT,
                        void* cT, float* reserve, void* workspace, hipStream_t stream) {
  using S4 = ShpH1024R32L;
  using S3 = ShpH1024R32L_M;
  static_assert(big_mid<S3>() <= big_mid<S4>(), "slab size");
  float* gin = (float*)workspace;
  float* slab = gin + (size_t)rs.B * rs.T * rs.H * 4;
  const int64_t n_rows = (int64_t)rs.B * rs.T;
  const int grid = (int)(n_rows < BIG_LIN_GRID ? n_rows : BIG_LIN_GRID);
  const TS* bin = rs.has_bias_in ? (const TS*)bias_in : (const TS*)nullptr;
  const TS* bhid = rs.has_bias_hid ? (const TS*)bias_hid : (const TS*)nullptr;
  const int level = big_merge_level();
  if (level > 0) {
    // the merged cores live behind the slab region (the tail of the workspace): [m3_in | m3_hid | m2_in | m2_hid]
    using S2 = ShpH1024R32L_M2;
    static_assert(big_mid<S2>() <= big_mid<S4>(), "slab size");
    const size_t b3 = ((size_t)merged_elems<S3>() * sizeof(float) + 255) & ~(size_t)255;
    const size_t b2 = ((size_t)merged2_elems<S2>() * sizeof(float) + 255) & ~(size_t)255;
    const size_t pair_bytes = (size_t)rs.B * 2 * rs.H * sizeof(unsigned long long);
    char* tail = (char*)workspace + big_rnn_fwd_workspace(rs) - 2 * (b3 + b2) - pair_bytes - big_gemm_bytes(rs) -
                 bigh_workspace_bytes();
    float* m3_in = (float*)tail;
    float* m3_hid = (float*)(tail + b3);
    float* m2_in = (float*)(tail + 2 * b3);
    float* m2_hid = (float*)(tail + 2 * b3 + b2);
    const int g3 = (merged_elems<S3>() + 255) / 256;
    hipLaunchKernelGGL((k_merge_cores01<S4, S3>), dim3(g3), dim3(256), 0, stream, packed_in, m3_in);
    hipLaunchKernelGGL((k_merge_cores01<S4, S3>), dim3(g3), dim3(256), 0, stream, packed_hid, m3_hid);
    if (level == 1) {
      hipLaunchKernelGGL((k_ttlinear_fwd_big<S3, 4, TS>), dim3(grid), dim3(FAST_NT), 0, stream, n_rows, m3_in,
                         (const TS*)x, gin, slab, 2);
      if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
      hipLaunchKernelGGL((k_rnn_fwd_big<S3, TTRNN_LSTM, TS>), dim3(rs.B), dim3(FAST_NT), 0, stream, rs.B, rs.T, gin,
                         (const TS*)h0, (const TS*)c0, m3_hid, bin, bhid, (TS*)out, (TS*)hT, (TS*)cT, reserve, slab);
      return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
    }
    const int g2 = (merged2_elems<S2>() + 255) / 256;
    hipLaunchKernelGGL((k_merge_cores_last<S3, S2>), dim3(g2), dim3(256), 0, stream, m3_in, m2_in);
    hipLaunchKernelGGL((k_merge_cores_last<S3, S2>), dim3(g2), dim3(256), 0, stream, m3_hid, m2_hid);
    // the two-core chain keeps its images in LDS (dynamic, > 64 KB: raise the limit once per kernel)
    static_assert(big_lds_images<S2>(), "two-core images must fit LDS");
    constexpr size_t lds_lin = (size_t)(big_mid<S2>() + in_size_of<S2>()) * sizeof(float);
    constexpr size_t lds_rec = (size_t)big_mid<S2>() * sizeof(float);
    {
      if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_ttlinear_fwd_big<S2, 4, TS>), lds_lin) != TTRNN_OK ||
          ensure_dynamic_lds(reinterpret_cast<const void*>(k_rnn_fwd_big<S2, TTRNN_LSTM, TS>), lds_rec) != TTRNN_OK)
        return TTRNN_ERR_LAUNCH;
    }
    const int cus = device_cu_count();
    const int grid2 = (int)(n_rows < cus ? n_rows : cus);          // one workgroup per CU (LDS)
    // OPT_BIG_NO_GEMM: A/B switch, K-in through the merged chain, row by row
    if (opt(OPT_FP32_MATH) == TTRNN_MATH_SPLIT && !opt(OPT_BIG_NO_GEMM) && gemm_split_ok(rs.in, 4 * rs.H)) {
      // K-in as one dense split-bf16 GEMM: W_in (gate-interleaved columns) = the chain kernel on the unit rows
      char* gt = tail + 2 * (b3 + b2) + pair_bytes;
      void* ident = gt;
      float* wdense = (float*)(gt + gemm_split_identity_bytes(rs.in));
      void* planes = (char*)wdense + gemm_split_dense_bytes(rs.in, 4 * rs.H);
      int st = launch_fill_identity(sizeof(TS) == 4 ? TTRNN_F32 : TTRNN_BF16, rs.in, ident, stream);
      if (st != TTRNN_OK) return st;
      hipLaunchKernelGGL((k_ttlinear_fwd_big<S2, 4, TS>), dim3(rs.in < cus ? rs.in : cus), dim3(FAST_NT), lds_lin, stream,
                         (int64_t)rs.in, m2_in, (const TS*)ident, wdense, slab, 2);
      if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
      // two-piece fp16 operands (three MFMA terms) where the scale passes pay off, else three bf16 pieces (gemm_use_half)
      void* gscr = (char*)planes + gemm_split_plane_bytes(rs.in, 4 * rs.H);
      const int gdt = sizeof(TS) == 4 ? TTRNN_F32 : TTRNN_BF16;
      if (!gemm_use_half(n_rows, rs.in, 4 * rs.H)) {
        st = launch_gemm_split_prep(wdense, rs.in, 4 * rs.H, planes, stream);
        if (st != TTRNN_OK) return st;
        st = launch_gemm_split(gdt, n_rows, rs.in, 4 * rs.H, x, planes, nullptr, rs.H, gin, stream);
      } else {
        st = launch_gemm_half_prep(wdense, rs.in, 4 * rs.H, planes, gscr, stream);
        if (st != TTRNN_OK) return st;
        st = launch_gemm_half(gdt, n_rows, rs.in, 4 * rs.H, x, planes, gscr, nullptr, rs.H, gin, stream);
      }
      if (st != TTRNN_OK) return st;
    } else {
      hipLaunchKernelGGL((k_ttlinear_fwd_big<S2, 4, TS>), dim3(grid2), dim3(FAST_NT), lds_lin, stream, n_rows, m2_in,
                         (const TS*)x, gin, slab, 2);
      if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
    }
    // two workgroups per sample need all 2B workgroups resident at once: CU count AND the occupancy the runtime reports for
    // the kernel that would run (one 512-thread workgroup with > 100 KB of LDS per CU); OPT_BIG_NO_PAIR: A/B switch
    const bool halfk = opt(OPT_FP32_MATH) == TTRNN_MATH_SPLIT && !opt(OPT_BIG_FP32_MFMA);
    constexpr size_t lds_img2 = (size_t)(St<S2, 0>::ROWS / 2) * St<S2, 0>::KP * sizeof(float);
    constexpr size_t lds_pair2 = lds_img2 > 100 * 1024 ? lds_img2 : 100 * 1024;
    const bool pair_fits =
        2 * rs.B <= cus && !opt(OPT_BIG_NO_PAIR) &&
        (halfk ? bigh_pair_resident(sizeof(TS) == 4 ? TTRNN_F32 : TTRNN_BF16, rs.B)
               : (ensure_dynamic_lds(reinterpret_cast<const void*>(k_lstm_fwd_big2<S2, TS>), lds_pair2) == TTRNN_OK &&
                  resident_at_once(reinterpret_cast<const void*>(k_lstm_fwd_big2<S2, TS>), FAST_NT, lds_pair2, 2L * rs.B)));
    if (pair_fits) {
      // two workgroups per sample: tagged h exchange words [B][2][H] behind the merged cores (tag 0 = never written)
      unsigned long long* hxb = (unsigned long long*)(tail + 2 * (b3 + b2));
      if (hipMemsetAsync(hxb, 0, pair_bytes, stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
      // split mode: both stages on two-piece fp16 operands (ttrnn_fast_bigh.hip); OPT_BIG_FP32_MFMA: A/B switch
      if (halfk)
        return launch_lstm_fwd_big2h(rs, sizeof(TS) == 4 ? TTRNN_F32 : TTRNN_BF16, gin, h0, c0, m2_hid, bin, bhid, out, hT,
                                     cT, reserve, hxb, tail + 2 * (b3 + b2) + pair_bytes + big_gemm_bytes(rs), stream);
      // the image needs 64 KB; asking for 100 KB keeps a second workgroup off the CU (76 KB each would fit twice, and
      // the dispatcher then packs the pairs onto half of the CUs: measured no faster than one workgroup per sample)
      constexpr size_t lds_img = (size_t)(St<S2, 0>::ROWS / 2) * St<S2, 0>::KP * sizeof(float);
      constexpr size_t lds_pair = lds_img > 100 * 1024 ? lds_img : 100 * 1024;
      {
        if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_lstm_fwd_big2<S2, TS>), lds_pair) != TTRNN_OK)
          return TTRNN_ERR_LAUNCH;
      }
      // (OPT_PAIR_FAULT, tests only: the last workgroup is not launched — its partner must time out, poison its sample
      // with NaN and count the event)
      hipLaunchKernelGGL((k_lstm_fwd_big2<S2, TS>), dim3(2 * rs.B - (opt(OPT_PAIR_FAULT) ? 1 : 0)), dim3(FAST_NT), lds_pair,
                         stream, rs.B, rs.T, gin, (const TS*)h0, (const TS*)c0, m2_hid, bin, bhid, (TS*)out, (TS*)hT,
                         (TS*)cT, reserve, hxb, device_status_ptr());
      return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((k_rnn_fwd_big<S2, TTRNN_LSTM, TS>), dim3(rs.B), dim3(FAST_NT), lds_rec, stream, rs.B, rs.T,
                       gin, (const TS*)h0, (const TS*)c0, m2_hid, bin, bhid, (TS*)out, (TS*)hT, (TS*)cT, reserve,
                       slab);
    return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((k_ttlinear_fwd_big<S4, 4, TS>), dim3(grid), dim3(FAST_NT), 0, stream, n_rows, packed_in,
                     (const TS*)x, gin, slab, 2);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  hipLaunchKernelGGL((k_rnn_fwd_big<S4, TTRNN_LSTM, TS>), dim3(rs.B), dim3(FAST_NT), 0, stream, rs.B, rs.T, gin,
                     (const TS*)h0, (const TS*)c0, packed_hid, bin, bhid, (TS*)out, (TS*)hT, (TS*)cT, reserve, slab);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

int launch_rnn_fwd_big(const RnnShape& rs, int dtype, const void* x, const void* h0, const void* c0,
                       const float* packed_in, const void* bias_in, const float* packed_hid, const void* bias_hid,
                       void* out, void* hT, void* cT, float* reserve, void* workspace, hipStream_t stream) {
  return dtype == TTRNN_F32 ? launch_big_t<float>(rs, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT, cT,
                                                  reserve, workspace, stream)
                            : launch_big_t<bf16_t>(rs, x, h0, c0, packed_in, bias_in, packed_hid, bias_hid, out, hT,
                                                   cT, reserve, workspace, stream);
}

}  // namespace ttrnn
