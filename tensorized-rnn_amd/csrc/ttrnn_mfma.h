// ttrnn_mfma.h — compile-time TT shapes and the MFMA chain-stage machinery shared by the
// shape-specialised kernels (ttrnn_fast.hip: persistent recurrent kernels; ttrnn_fast_lin.hip: batched
// TTLinear).  Device-only (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include "ttrnn_core.h"

namespace ttrnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

static constexpr int FAST_NW = 8;            // waves per workgroup
static constexpr int FAST_NT = FAST_NW * 64;

// ---- compile-time shapes ---------------------------------------------------------------------------
template <int D_, int J0, int J1, int J2, int J3, int I0, int I1, int I2, int I3, int R1, int R2, int R3>
struct Shp {
  static constexpr int D = D_;
  static constexpr int J[4] = {J0, J1, J2, J3};
  static constexpr int I[4] = {I0, I1, I2, I3};
  static constexpr int R[5] = {1, D_ > 1 ? R1 : 1, D_ > 2 ? R2 : 1, D_ > 3 ? R3 : 1, 1};
};

constexpr bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
constexpr int chunk_of(int nu) { return nu % 4 == 0 ? 4 : (nu % 3 == 0 ? 3 : (nu % 2 == 0 ? 2 : 1)); }

template <class S>
constexpr int rows_of(int k) {   // chain rows of stage k: prod_{m>k} I_m * prod_{m<k} J_m
  int r = 1;
  for (int m = k + 1; m < S::D; ++m) r *= S::I[m];
  for (int m = 0; m < k; ++m) r *= S::J[m];
  return r;
}

template <class S, int k>
struct St {
  static constexpr int R = S::R[k], I = S::I[k], J = S::J[k], R1 = S::R[k + 1];
  static constexpr int K = J * R1, M = I * R;
  static constexpr int KP = (K + 3) & ~3;      // image row length: only the first stage (K = J_{d-1}) can need padding
  static constexpr int ROWS = rows_of<S>(k);
  static constexpr int MT = (M + 15) / 16, RT = (ROWS + 15) / 16;
  static constexpr int WV = (KP % 16 == 0) ? 4 : ((KP % 8 == 0) ? 2 : 1);  // floats per fragment read
  static constexpr int NU = KP / (4 * WV);                                 // fragment reads per row tile
  static constexpr int NSTEP = KP / 4;                                     // MFMA k-steps
  // tile ownership: whole m-tiles per wave, or (few m-tiles) the row tiles of one m-tile split over waves
  static constexpr bool SPLIT = (MT < FAST_NW) && (FAST_NW % MT == 0);
  static constexpr int G = SPLIT ? FAST_NW / MT : 1;
  static constexpr int XM = SPLIT ? 1 : (MT + FAST_NW - 1) / FAST_NW;
  static constexpr int YR = SPLIT ? (RT + G - 1) / G : RT;
  static constexpr int NWREG = XM * NSTEP;
  static constexpr int IN_ELEMS = ROWS * K, OUT_ELEMS = ROWS * M;
};

template <class S, int k>
constexpr int nwreg() { return St<S, k>::NWREG; }

template <class S>
constexpr int in_size_of() { int n = 1; for (int k = 0; k < S::D; ++k) n *= S::J[k]; return n; }
template <class S>
constexpr int out_size_of() { int n = 1; for (int k = 0; k < S::D; ++k) n *= S::I[k]; return n; }
template <class S>
constexpr int woff_of(int k) {   // float offset of W_k in the canonical packed buffer
  int off = 0;
  for (int m = 0; m < k; ++m) off += S::J[m] * S::R[m + 1] * S::I[m] * S::R[m];
  return off;
}
template <class S>
constexpr int maxmid_of() {      // largest intermediate (floats) that has to live in a ping-pong buffer
  int best = 4;
  for (int k = 1; k < S::D; ++k) {
    int rows = 1;
    for (int m = k + 1; m < S::D; ++m) rows *= S::I[m];
    for (int m = 0; m < k; ++m) rows *= S::J[m];
    int e = rows * S::I[k] * S::R[k];
    if (e > best) best = e;
  }
  return best;
}
template <class S>
constexpr bool shape_ok() {      // what the MFMA path needs: inner ranks multiple of 4 (b128 tile stores)
  for (int k = 1; k < S::D; ++k)
    if (S::R[k] % 4 != 0) return false;
  return true;
}
template <class S>
constexpr bool shape_ok_recurrent() {   // recurrent kernels also write h element-wise into an unpadded image
  return shape_ok<S>() && (S::J[S::D - 1] % 4 == 0);
}

// float offset of element (row, kk) in the LDS image of a [rows][K] stage input.  For K a multiple of
// 32 with K/4 a power of two the 16-byte slot index is XOR-swizzled with the row so that the
// ds_read_b128 fragment reads (lane (c,q) reads slot 4u+q of row 16*rt+c) hit 16 distinct slots per
// 16-lane group.
template <int K>
__device__ __forceinline__ int a_off(int row, int kk) {
  if constexpr (K % 32 == 0 && is_pow2(K / 4)) {
    constexpr int ns = K / 4;
    const int slot = kk >> 2;
    const int g = (ns >= 16) ? (row & 15) : ((row >> 1) & 7);
    return ((row * ns + (slot ^ g)) << 2) + (kk & 3);
  } else {
    return row * K + kk;
  }
}

__device__ __forceinline__ void lds_barrier() {
  // LDS-only hand-off between the waves of the workgroup: drain this wave's LDS ops, then barrier.
  // Deliberately no vmcnt wait: the gate-input prefetch and the output stores stay in flight.
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Diagnostic builds only (TTRNN_DIAG=1): per-phase cycle stamps, MI355X guide section 7 "In-kernel stamps".
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define TT_STAMP(idx)                                   \
  if constexpr (DIAG) {                                 \
    const unsigned long long now_ = stamp();            \
    seg[idx] += now_ - last_;                           \
    last_ = now_;                                       \
  }

// ---- one chain stage ---------------------------------------------------------------------------------
template <class S, int k, int NW_>
__device__ __forceinline__ void load_wfrag(float (&w)[NW_], const float* packed,
                                           int wave, int lane) {
  using T = St<S, k>;
  static_assert(NW_ == T::NWREG, "weight fragment array size");
  const int r = lane & 15, q = lane >> 4;
  const float* W = packed + woff_of<S>(k);
#pragma unroll
  for (int x = 0; x < T::XM; ++x) {
    const int mt = T::SPLIT ? (wave % T::MT) : (wave + FAST_NW * x);
    const int m = 16 * mt + r;
#pragma unroll
    for (int u = 0; u < T::NU; ++u)
#pragma unroll
      for (int e = 0; e < T::WV; ++e) {
        const int kk = (4 * u + q) * T::WV + e;
        // branch-free: always load an in-bounds element, then mask (a conditional load costs a branch each)
        const bool okw = mt < T::MT && m < T::M && kk < T::K;
        const float wv = W[okw ? kk * T::M + m : 0];
        w[x * T::NSTEP + u * T::WV + e] = okw ? wv : 0.f;
      }
  }
}

// Fragment reads + MFMAs of stage k for the tiles this wave owns.  Ain: LDS image [ROWS][K] (a_off<K>).
// D[p = 4q + j][c]: p <-> output feature of m-tile mt, c <-> chain row 16*rt + c.
template <class S, int k, int NW_, int XM_, int YR_>
__device__ __forceinline__ void stage_mma(const float (&w)[NW_], const float* Ain, f32x4 (&acc)[XM_][YR_], int wave,
                                          int lane) {
  using T = St<S, k>;
  static_assert(NW_ == T::NWREG && XM_ == T::XM && YR_ == T::YR, "stage tile bookkeeping");
  const int c = lane & 15, q = lane >> 4;
  float af[T::YR][T::NSTEP];
#pragma unroll
  for (int y = 0; y < T::YR; ++y) {
    const int rt = T::SPLIT ? (wave / T::MT + T::G * y) : y;
    int row = 16 * rt + c;
    row = row < T::ROWS ? row : T::ROWS - 1;
#pragma unroll
    for (int u = 0; u < T::NU; ++u) {
      const float* p = Ain + a_off<T::KP>(row, (4 * u + q) * T::WV);
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
  }
#pragma unroll
  for (int x = 0; x < T::XM; ++x)
#pragma unroll
    for (int y = 0; y < T::YR; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < T::NSTEP; ++s)
#pragma unroll
    for (int x = 0; x < T::XM; ++x)
#pragma unroll
      for (int y = 0; y < T::YR; ++y)
        acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[x * T::NSTEP + s], af[y][s], acc[x][y], 0, 0, 0);
}

// Cout: LDS image of the next stage input (k > 0) or the flat gate pre-activation vector (k == 0).
template <class S, int k, int NW_>
__device__ __forceinline__ void run_stage(const float (&w)[NW_], const float* Ain, float* Cout, int wave, int lane) {
  using T = St<S, k>;
  const int c = lane & 15, q = lane >> 4;
  f32x4 acc[T::XM][T::YR];
  stage_mma<S, k>(w, Ain, acc, wave, lane);
#pragma unroll
  for (int x = 0; x < T::XM; ++x) {
    const int mt = T::SPLIT ? (wave % T::MT) : (wave + FAST_NW * x);
#pragma unroll
    for (int y = 0; y < T::YR; ++y) {
      const int rt = T::SPLIT ? (wave / T::MT + T::G * y) : y;
      const int row = 16 * rt + c;
      const int m0 = 16 * mt + 4 * q;
      if (mt < T::MT && rt < T::RT && row < T::ROWS && m0 < T::M) {
        if constexpr (k > 0) {
          // C_k flat index f = i*ROWS*R + row*R + a  ==  A_{k-1} flat index (ops.py:89-90)
          using N = St<S, k - 1>;
          const int i = m0 / T::R, a0 = m0 % T::R;
          const int f = i * (T::ROWS * T::R) + row * T::R + a0;
          float* p = Cout + a_off<N::KP>(f / N::K, f % N::K);
          *reinterpret_cast<f32x4*>(p) = acc[x][y];
        } else {
          // R_0 = 1: m is i_0 and the result is y[o], o = i_0*ROWS + row
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (m0 + j < T::M) Cout[(m0 + j) * T::ROWS + row] = acc[x][y][j];
        }
      }
    }
  }
}

// Gate non-linearities on the hardware transcendental unit (v_exp_f32 / v_rcp_f32, 1 ulp each):
// absolute error ~1e-7, far inside the 1e-5 parity tolerance, and ~10 instructions instead of the
// ~80-instruction branchy libm expansions (the gate phase is on the per-step critical path).
__device__ __forceinline__ float fsigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float ftanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// input_size == 1 (GinSrc::in1): the scalar inputs x[b][t] of 64 consecutive timesteps sit in ONE VGPR of every wave
// (lane l <-> step t0 + l) and are refilled a whole chunk ahead.  The time loop reads x_t with v_readlane and never
// waits on a load it has just issued (a per-step `global_load ; s_waitcnt vmcnt(0)` exposes one L2/HBM round trip
// per timestep on the critical path).
template <typename TS>
struct XChunk {
  float cur, nxt;
  __device__ __forceinline__ void init(const TS* xs, size_t base, int T, int lane) {
    cur = lane < T ? ld(xs, base + lane) : 0.f;
    nxt = 64 + lane < T ? ld(xs, base + 64 + lane) : 0.f;
  }
  __device__ __forceinline__ float at(int t) const {      // t must lie in the current chunk
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur), t & 63));
  }
  __device__ __forceinline__ void advance(const TS* xs, size_t base, int T, int t, int lane) {   // after step t's use
    if ((t & 63) == 63) {
      cur = nxt;
      const int i = t + 65 + lane;
      nxt = i < T ? ld(xs, base + i) : 0.f;
    }
  }
};

// ---- batched (NB stacked samples) forward stage, shared by ttrnn_fast_lin.hip and ttrnn_fast_bwd.hip ------------
// index of output feature o of stacked sample smp inside the LDS output tile
template <int G, int OUT>
__device__ __forceinline__ int ytile_index(int smp, int o, int ilv_mode) {
  if constexpr (G == 0) {
    return smp * OUT + o;
  } else {
    constexpr int H = OUT / G;
    const int g = o / H, hid = o - g * H;
    const int slot = (ilv_mode == 2) ? (g == 1 ? 2 : (g == 2 ? 1 : g)) : g;
    return (smp * H + hid) * 4 + slot;
  }
}

// one chain stage over NB stacked samples; runtime loop over row tiles, two tiles in flight per iteration
template <class S, int k, int NB, int G, int NW_>
__device__ __forceinline__ void lin_stage(const float (&w)[NW_], const float* Ain, float* Cout, int wave, int lane,
                                          int ilv_mode) {
  using T = St<S, k>;
  static_assert(NW_ == T::NWREG, "weight fragment array size");
  constexpr int TOT = NB * T::ROWS;                 // stacked chain rows
  constexpr int RT_ALL = (TOT + 15) / 16;
  constexpr int OUT = out_size_of<S>();
  constexpr int RSTEP = T::SPLIT ? T::G : 1;
  const int c = lane & 15, q = lane >> 4;
  const int rt0 = T::SPLIT ? (wave / T::MT) : 0;
  constexpr int UC = chunk_of(T::NU);                // fragment reads per chunk (bounds live registers)
  for (int rtb = rt0; rtb < RT_ALL; rtb += 2 * RSTEP) {
    int Rr[2];
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int R = 16 * (rtb + y * RSTEP) + c;
      Rr[y] = R < TOT ? R : TOT - 1;
    }
    f32x4 acc[T::XM][2];
#pragma unroll
    for (int x = 0; x < T::XM; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
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
      for (int x = 0; x < T::XM; ++x)
#pragma unroll
        for (int s2 = 0; s2 < UC * T::WV; ++s2) {
          const float wv = w[x * T::NSTEP + u0 * T::WV + s2];
          acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, af[0][s2], acc[x][0], 0, 0, 0);
          acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, af[1][s2], acc[x][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int x = 0; x < T::XM; ++x) {
      const int mt = T::SPLIT ? (wave % T::MT) : (wave + FAST_NW * x);
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const f32x4 a = acc[x][y];
        const int rt = rtb + y * RSTEP;
        const int R = 16 * rt + c;
        const int m0 = 16 * mt + 4 * q;
        if (mt < T::MT && rt < RT_ALL && R < TOT && m0 < T::M) {
          const int smp = R / T::ROWS, row = R - smp * T::ROWS;
          if constexpr (k > 0) {
            // C_k flat index within the sample == A_{k-1} flat index (ops.py:89-90)
            using N = St<S, k - 1>;
            const int i = m0 / T::R, a0 = m0 % T::R;
            const int f = i * (T::ROWS * T::R) + row * T::R + a0;
            float* p = Cout + a_off<N::KP>(smp * N::ROWS + f / N::K, f % N::K);
            *reinterpret_cast<f32x4*>(p) = a;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (m0 + j < T::M) Cout[ytile_index<G, OUT>(smp, (m0 + j) * T::ROWS + row, ilv_mode)] = a[j];
          }
        }
      }
    }
  }
}

// ---- shape table ----------------------------------------------------------------------------------------
template <class S>
static bool shape_matches(const TtShape& s) {
  if (s.d != S::D) return false;
  for (int k = 0; k < S::D; ++k)
    if (s.J[k] != S::J[k] || s.I[k] != S::I[k] || s.R[k] != S::R[k]) return false;
  return true;
}

// hidden-to-hidden / layer-to-layer TT matrices
using ShpH256R8L = Shp<3, 4, 8, 8, 1, 8, 8, 16, 1, 8, 8, 1>;     // cfg2: TT-LSTM H=256 d=3 r=8
using ShpH256R8G = Shp<3, 4, 8, 8, 1, 8, 8, 12, 1, 8, 8, 1>;     // cfg3: TT-GRU  H=256 d=3 r=8
using ShpH256R16L = Shp<3, 4, 8, 8, 1, 8, 8, 16, 1, 16, 16, 1>;  // cfg4: TT-LSTM H=256 d=3 r=16
using ShpH256R16G = Shp<3, 4, 8, 8, 1, 8, 8, 12, 1, 16, 16, 1>;  //       TT-GRU  H=256 d=3 r=16
using ShpH128R4L = Shp<2, 8, 16, 1, 1, 16, 32, 1, 1, 4, 1, 1>;   // cfg1: TT-LSTM H=128 d=2 r=4
using ShpH256N = Shp<3, 4, 8, 8, 1, 4, 8, 8, 1, 8, 8, 1>;        // ONE gate of a naive per-gate TT-LSTM / TT-GRU, H=256 d=3 r=8 (tt_linearset.py:5-38)
using ShpH384R8L = Shp<3, 6, 8, 8, 1, 8, 12, 16, 1, 8, 8, 1>;    // benchmarking.py --hidden_size 384: TT-LSTM H=384 d=3 r=8 (forward only)
using ShpH512R8L = Shp<3, 8, 8, 8, 1, 8, 16, 16, 1, 8, 8, 1>;    // benchmarking.py defaults: TT-LSTM H=512 d=3 r=8 (forward only)
// input-to-hidden TT matrices of the first layer
using ShpI1R8L = Shp<3, 1, 1, 1, 1, 8, 8, 16, 1, 8, 8, 1>;       // cfg2: in=1  -> 4*256
using ShpI1R8G = Shp<3, 1, 1, 1, 1, 8, 8, 12, 1, 8, 8, 1>;       // cfg3: in=1  -> 3*256
using ShpI40R16L = Shp<3, 2, 4, 5, 1, 8, 8, 16, 1, 16, 16, 1>;   // cfg4: in=40 -> 4*256
using ShpI1R4L = Shp<2, 1, 1, 1, 1, 16, 32, 1, 1, 4, 1, 1>;      // cfg1: in=1  -> 4*128
// TTLinear heads of the callers (speaker_encoder.py:47-48: 256 -> 256, d = 3, r = 16)
using ShpHd256R16 = Shp<3, 4, 8, 8, 1, 4, 8, 8, 1, 16, 16, 1>;
// (the classifier of benchmarking.py on the cfg5 model, 1024 -> 256 with d = 4, r = 32, does NOT fit this path: its chain images
// need 266 KB of LDS per row; it stays on the any-shape kernels)

}  // namespace ttrnn
