// ttrnn_big.h — shapes and per-launch core-merging kernels shared by the big-shape translation units
// (ttrnn_fast_big.hip: forward; ttrnn_fast_bigb.hip: reverse-time kernel and weight gradients).  Device-only (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include "ttrnn_core.h"
#include "ttrnn_mfma.h"

namespace ttrnn {

using ShpH1024R32L = Shp<4, 4, 4, 8, 8, 8, 8, 8, 8, 32, 32, 32>;   // cfg5: TT-LSTM H = in = 1024, d = 4, r = 32

// Contracting cores 0 and 1 once per launch turns a d-core TT-matrix into a (d-1)-core one with first modes
// (J0*J1, I0*I1) and the same input / output index order:  G01[(i0,i1), (j0,j1), r2] = sum_r1 G0[i0,j0,r1] G1[r1,i1,j1,r2].
// For cfg5 the chain then costs 21.5 instead of 35.1 MFLOP per sample-step (the two stages with contraction lengths 128
// over 256 / 512 rows become one with 512 over 64 rows) and runs on the very same stage code.
using ShpH1024R32L_M = Shp<3, 16, 8, 8, 1, 64, 8, 8, 1, 32, 32, 1>;

// packed4: [W_0 | W_1 | W_2 | W_3 ...] of S4;  packed3: [W'_0 = merged | W'_1 = W_2 | W'_2 = W_3] of S3
template <class S4, class S3>
__global__ void __launch_bounds__(256) k_merge_cores01(const float* __restrict__ packed4, float* __restrict__ packed3) {
  constexpr int J1 = S4::J[1], I0 = S4::I[0], I1 = S4::I[1], R1 = S4::R[1], R2 = S4::R[2];
  constexpr int M0 = S3::I[0], K0 = S3::J[0] * S3::R[1];
  constexpr int N0 = K0 * M0;
  constexpr int N1 = S3::J[1] * S3::R[2] * S3::I[1] * S3::R[1], N2 = S3::J[2] * S3::R[3] * S3::I[2] * S3::R[2];
  static_assert(S3::J[0] == S4::J[0] * J1 && M0 == I0 * I1 && S3::R[1] == R2 && S3::D == 3 && S4::D == 4, "merged shape");
  const float* W0 = packed4 + woff_of<S4>(0);
  const float* W1 = packed4 + woff_of<S4>(1);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < N0) {
    const int kk = e / M0, m = e % M0;
    const int jj = kk / R2, r2 = kk % R2, j0 = jj / J1, j1 = jj % J1, i0 = m / I1, i1 = m % I1;
    float v = 0.f;
    for (int r1 = 0; r1 < R1; ++r1)
      v = fmaf(W0[(j0 * R1 + r1) * I0 + i0], W1[(j1 * R2 + r2) * (I1 * R1) + i1 * R1 + r1], v);
    packed3[woff_of<S3>(0) + e] = v;
  } else if (e < N0 + N1) {
    packed3[woff_of<S3>(1) + e - N0] = packed4[woff_of<S4>(2) + e - N0];
  } else if (e < N0 + N1 + N2) {
    packed3[woff_of<S3>(2) + e - N0 - N1] = packed4[woff_of<S4>(3) + e - N0 - N1];
  }
}

// ... and the same for the LAST two cores: G23[r2, (i2,i3), (j2,j3)] = sum_r3 G2[r2,i2,j2,r3] G3[r3,i3,j3,0]; the chain is
// then two stages of 4.2 MFLOP each (16 x 64 x 2048 and 64 x 512 x 64), the cost of the dense 4096 x 1024 product
using ShpH1024R32L_M2 = Shp<2, 16, 64, 1, 1, 64, 64, 1, 1, 32, 1, 1>;

// packed3: [W'_0 | W'_1 | W'_2] of S3;  packed2: [W''_0 = W'_0 | W''_1 = merged W'_1,W'_2] of S2
// The two-core buffer is stored in MFMA-FRAGMENT order (what load_wfrag would gather with one 4-byte load per k-step
// becomes one coalesced 16-byte load per four k-steps):  for stage k, m-tile mt, fragment group u, lane (r, q):
//     frag_k[((mt*NU_k + u)*64 + lane)*4 + e] = W_k[kk = (4u + q)*4 + e][m = 16 mt + r]
template <class S, int k>
__device__ __forceinline__ void frag_decode(int idx, int& kk, int& m) {
  using T = St<S, k>;
  static_assert(T::WV == 4 && T::K % 16 == 0 && T::M % 16 == 0, "fragment order needs 16-byte fragment reads");
  const int e = idx & 3, lane = (idx >> 2) & 63, g = idx >> 8;
  const int u = g % T::NU, mt = g / T::NU;
  kk = (4 * u + (lane >> 4)) * 4 + e;
  m = 16 * mt + (lane & 15);
}

template <class S3, class S2>
__global__ void __launch_bounds__(256) k_merge_cores_last(const float* __restrict__ packed3,
                                                          float* __restrict__ packed2) {
  constexpr int J2 = S3::J[2], I1 = S3::I[1], I2 = S3::I[2], R1 = S3::R[1], R2 = S3::R[2];
  constexpr int M0 = S3::I[0], N0 = S3::J[0] * R1 * M0;
  constexpr int K1 = S2::J[1], M1 = S2::I[1] * S2::R[1];       // W''_1 [J1*J2][I1*I2*R1]
  static_assert(S2::J[1] == S3::J[1] * J2 && S2::I[1] == I1 * I2 && S2::R[1] == R1 && S2::J[0] == S3::J[0] &&
                    S2::I[0] == S3::I[0] && S3::R[3] == 1, "merged shape");
  const float* W0 = packed3 + woff_of<S3>(0);
  const float* W1 = packed3 + woff_of<S3>(1);                  // [J1*R2][I1*R1]
  const float* W2 = packed3 + woff_of<S3>(2);                  // [J2*1][I2*R2]
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < N0) {
    int kk, m;
    frag_decode<S2, 0>(e, kk, m);
    packed2[woff_of<S2>(0) + e] = W0[kk * M0 + m];
  } else if (e < N0 + K1 * M1) {
    int kk, mm;                                                // kk = j1*J2 + j2;  mm = (i1*I2 + i2)*R1 + a
    frag_decode<S2, 1>(e - N0, kk, mm);
    const int j1 = kk / J2, j2 = kk % J2, a = mm % R1, ii = mm / R1, i1 = ii / I2, i2 = ii % I2;
    float v = 0.f;
    for (int r2 = 0; r2 < R2; ++r2)
      v = fmaf(W1[(j1 * R2 + r2) * (I1 * R1) + i1 * R1 + a], W2[j2 * (I2 * R2) + i2 * R2 + r2], v);
    packed2[woff_of<S2>(1) + e - N0] = v;
  }
}

// transposed merged matrix (4096 -> 1024) with reversed mode order: inputs (i23, i01), outputs (j23, j01), rank 32
// (ttrnn_fast_bigb.hip:k_bigb_prep builds its fragment-ordered cores)
using ShpH1024R32L_T = Shp<2, 64, 64, 1, 1, 64, 16, 1, 1, 32, 1, 1>;

template <class S2>
constexpr int merged2_elems() { return S2::J[0] * S2::R[1] * S2::I[0] + S2::J[1] * S2::I[1] * S2::R[1]; }

template <class S3>
constexpr int merged_elems() {
  return S3::J[0] * S3::R[1] * S3::I[0] + S3::J[1] * S3::R[2] * S3::I[1] * S3::R[1] +
         S3::J[2] * S3::R[3] * S3::I[2] * S3::R[2];
}

// what the next step sees of a stored h: rounded once to the storage type
__device__ __forceinline__ float round_like(const float*, float v) { return v; }
__device__ __forceinline__ float round_like(const bf16_t*, float v) { return bf16_to_f32(f32_to_bf16(v)); }

// Guard of the kernels that carry a whole operand matrix under ONE power-of-two scale (the cfg5-class reverse-time kernel):
// rows[0..n) = per matrix row the summed absolute representation error of the two fp16 pieces, rows[n..2n) = the summed
// absolute scaled entries (k_bigbh_prep).  Normal pieces carry a relative error <= 2^-22 per entry; a row whose error sum
// exceeds 2^-20 of its magnitude sum has its bulk in fp16's subnormal range (a few entries 2^16 and more above the rest of
// the matrix), and the launch must not run on these pieces.  Every thread of the workgroup calls this (barrier inside).
__device__ __forceinline__ bool big_guard_tripped(const float* __restrict__ rows, int n, int tid, int nthreads) {
  int bad = 0;
  for (int i = tid; i < n; i += nthreads) bad |= rows[i] > 9.5367431640625e-07f * rows[n + i] ? 1 : 0;
  return __syncthreads_or(bad) != 0;
}

__device__ __forceinline__ float bsigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float btanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

}  // namespace ttrnn
