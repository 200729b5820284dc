// ttrnn_fast.hip — shape-specialised persistent recurrent kernel (gfx950, fp32 MFMA).
//
// One workgroup (8 waves) owns one sample for the whole sequence:
//   * the hidden TT chain runs stage by stage on v_mfma_f32_16x16x4_f32 (exact fp32, same rate as
//     the fp32 VALU peak but one operand VGPR per 1024 MACs);  the MFMA "A" operand is the CORE
//     (output features on the 16 MFMA rows), the "B" operand the activations (chain rows on the 16
//     MFMA columns) — so each lane's 4 accumulator registers are 4 consecutive rank indices and a
//     stage result is stored with one ds_write_b128 per tile;
//   * every core fragment a wave needs is loaded ONCE into VGPRs (cfg2: 26 registers per lane) and
//     stays there for all timesteps: no weight traffic at all inside the time loop;
//   * chain intermediates ping-pong through LDS in an XOR-swizzled [row][K] image whose b128
//     fragment reads are bank-conflict-free (slot ^= f(row), see a_off);
//   * the cell state c lives in a register of the thread that owns the hidden unit, h goes straight
//     back into the LDS image the first chain stage reads;
//   * the non-recurrent input projection W_in x_t + b_in is hoisted out of the recurrence (one
//     batched TTLinear launch over all B*T rows on all CUs); the time loop prefetches its row for
//     step t+1 into registers while step t computes, and the per-step barriers are raw s_barrier +
//     lgkmcnt(0) so that prefetch (and the output stores) stay in flight across them.
//
// Replaces, for one layer: tensorized_rnn/lstm.py:23-32,123-133 and gru.py:33-44,124-134 with the
// hidden-weights chain of t3nsor/ops.py:78-93 (reference file:line).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"

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
  static constexpr int ROWS = rows_of<S>(k);
  static constexpr int MT = (M + 15) / 16, RT = (ROWS + 15) / 16;
  static constexpr int WV = (K % 16 == 0) ? 4 : ((K % 8 == 0) ? 2 : 1);   // floats per fragment read
  static constexpr int NU = K / (4 * WV);                                 // fragment reads per row tile
  static constexpr int NSTEP = K / 4;                                     // MFMA k-steps
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
constexpr bool shape_ok() {      // what the MFMA path needs: ranks multiple of 4, K multiple of 4
  for (int k = 0; k < S::D; ++k) {
    if (k > 0 && S::R[k] % 4 != 0) return false;
    if ((S::J[k] * S::R[k + 1]) % 4 != 0) return false;
  }
  return true;
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
        w[x * T::NSTEP + u * T::WV + e] = (mt < T::MT && m < T::M) ? W[kk * T::M + m] : 0.f;
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
      const float* p = Ain + a_off<T::K>(row, (4 * u + q) * T::WV);
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
          float* p = Cout + a_off<N::K>(f / N::K, f % N::K);
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

// ---- the persistent kernel ---------------------------------------------------------------------------
// gin: fp32 [B][T][G*H] = W_in x_t + b_in (hoisted).  One workgroup per sample.
template <class S, int CELL>
__global__ void __launch_bounds__(FAST_NT) k_rnn_fwd_fast(int B, int T, const float* __restrict__ gin,
                                                          const float* __restrict__ h0, const float* __restrict__ c0,
                                                          const float* __restrict__ packed_hid,
                                                          const float* __restrict__ bias_hid, float* __restrict__ out,
                                                          float* __restrict__ hT, float* __restrict__ cT,
                                                          float* __restrict__ reserve) {
  constexpr int D = S::D;
  constexpr int H = in_size_of<S>();
  constexpr int G = CELL == TTRNN_LSTM ? 4 : 3;
  constexpr int GH = G * H;
  static_assert(out_size_of<S>() == GH, "TT output size must be n_gates * hidden");
  constexpr int HPT = (H + FAST_NT - 1) / FAST_NT;       // hidden units per thread
  constexpr int MID = maxmid_of<S>();
  using SL = St<S, D - 1>;                               // first stage executed (reads h)

  __shared__ __attribute__((aligned(16))) float hbuf[H];
  __shared__ __attribute__((aligned(16))) float bufA[MID];
  __shared__ __attribute__((aligned(16))) float bufB[D > 2 ? MID : 4];
  __shared__ __attribute__((aligned(16))) float gbuf[GH];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t b = blockIdx.x;

  // resident core fragments
  float w0[nwreg<S, 0>()];
  float w1[nwreg<S, (D > 1 ? 1 : 0)>()];
  float w2[nwreg<S, (D > 2 ? 2 : 0)>()];
  float w3[nwreg<S, (D > 3 ? 3 : 0)>()];
  load_wfrag<S, 0>(w0, packed_hid, wave, lane);
  if constexpr (D > 1) load_wfrag<S, 1>(w1, packed_hid, wave, lane);
  if constexpr (D > 2) load_wfrag<S, 2>(w2, packed_hid, wave, lane);
  if constexpr (D > 3) load_wfrag<S, 3>(w3, packed_hid, wave, lane);

  // per-thread recurrent state and constants for its hidden units
  // gin is gate-interleaved: [B][T][H][4], slot order i,g,f,o (LSTM) / r,z,n,- (GRU): one 16-byte load
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gin);
  float cst[HPT], hst[HPT], bh[HPT][G];
  f32x4 gi[HPT];
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int hid = tid + u * FAST_NT;
    const bool ok = hid < H;
    hst[u] = (ok && h0) ? h0[b * H + hid] : 0.f;
    cst[u] = (ok && c0 && CELL == TTRNN_LSTM) ? c0[b * H + hid] : 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) bh[u][g] = (ok && bias_hid) ? bias_hid[g * H + hid] : 0.f;
    gi[u] = (ok && T > 0) ? gin4[(b * T) * H + hid] : f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok) hbuf[a_off<SL::K>(hid / SL::K, hid % SL::K)] = hst[u];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see k_lstm_fwd_fused
  lds_barrier();

  for (int t = 0; t < T; ++t) {
    // ---- hidden chain: stage D-1 .. 0 --------------------------------------------------------------
    if constexpr (D == 1) {
      run_stage<S, 0>(w0, hbuf, gbuf, wave, lane);
    } else if constexpr (D == 2) {
      run_stage<S, 1>(w1, hbuf, bufA, wave, lane);
      lds_barrier();
      run_stage<S, 0>(w0, bufA, gbuf, wave, lane);
    } else if constexpr (D == 3) {
      run_stage<S, 2>(w2, hbuf, bufA, wave, lane);
      lds_barrier();
      run_stage<S, 1>(w1, bufA, bufB, wave, lane);
      lds_barrier();
      run_stage<S, 0>(w0, bufB, gbuf, wave, lane);
    } else {
      run_stage<S, 3>(w3, hbuf, bufA, wave, lane);
      lds_barrier();
      run_stage<S, 2>(w2, bufA, bufB, wave, lane);
      lds_barrier();
      run_stage<S, 1>(w1, bufB, bufA, wave, lane);
      lds_barrier();
      run_stage<S, 0>(w0, bufA, gbuf, wave, lane);
    }
    lds_barrier();
    // ---- gates + state update (lstm.py:26-32 / gru.py:38-44) ----------------------------------------
    const size_t bt = b * T + t;
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
      const int hid = tid + u * FAST_NT;
      if (hid < H) {
        float hy;
        if constexpr (CELL == TTRNN_LSTM) {
          const float ig = fsigmoid(gi[u][0] + gbuf[hid] + bh[u][0]);
          const float fg = fsigmoid(gi[u][2] + gbuf[H + hid] + bh[u][1]);
          const float gg = ftanh(gi[u][1] + gbuf[2 * H + hid] + bh[u][2]);
          const float og = fsigmoid(gi[u][3] + gbuf[3 * H + hid] + bh[u][3]);
          const float cy = fg * cst[u] + ig * gg;
          hy = og * ftanh(cy);
          cst[u] = cy;
          if (reserve) {
            float* rv = reserve + (bt * H + hid) * 5;
            rv[0] = ig; rv[1] = gg; rv[2] = fg; rv[3] = og; rv[4] = cy;
          }
        } else {
          const float hn = gbuf[2 * H + hid] + bh[u][2];
          const float rg = fsigmoid(gi[u][0] + gbuf[hid] + bh[u][0]);
          const float zg = fsigmoid(gi[u][1] + gbuf[H + hid] + bh[u][1]);
          const float ng = ftanh(gi[u][2] + rg * hn);
          hy = (1.0f - zg) * ng + zg * hst[u];
          if (reserve) *reinterpret_cast<f32x4*>(reserve + (bt * H + hid) * 4) = f32x4{rg, zg, ng, hn};
        }
        hst[u] = hy;
        out[bt * H + hid] = hy;
        hbuf[a_off<SL::K>(hid / SL::K, hid % SL::K)] = hy;
        // prefetch the hoisted input projection of the next step; consumed one iteration later
        if (t + 1 < T) gi[u] = gin4[(bt + 1) * H + hid];
      }
    }
    lds_barrier();
  }
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int hid = tid + u * FAST_NT;
    if (hid < H) {
      if (hT) hT[b * H + hid] = hst[u];
      if (CELL == TTRNN_LSTM && cT) cT[b * H + hid] = cst[u];
    }
  }
}

// ---- LSTM: last chain stage fused with the gate arithmetic --------------------------------------------
// o = gate*H + hid and o = i_0*ROWS_0 + row give i_0 = gate*P + par, hid = par*ROWS_0 + row (P = I_0/4).
// Loading the stage-0 core with its rows permuted — MFMA row p = 4*qq + j of m-tile mt <-> i_0 = j*P + (4*mt+qq) —
// puts the four gate pre-activations of ONE hidden unit into the four accumulator registers of ONE lane
// (lane (c, q): hid = (4*mt+q)*ROWS_0 + 16*rt + c).  The gate math then runs straight on the accumulators:
// no gate buffer in LDS and one barrier fewer per timestep; c and the hoisted gate inputs live in that lane.
template <class S>
constexpr bool lstm_fusable() { return S::D >= 2 && S::R[0] == 1 && S::I[0] % 4 == 0; }

// PAIR variant (P <= 2, one m-tile): MFMA row p = 4*qq + j holds, for pair = qq >> 1 and par = qq & 1, gate
// (pair ? f : i) in j = 0 and (pair ? o : g) in j = 1; rows j = 2, 3 are zero.  Lanes 0-31 then own (i, g) and
// lanes 32-63 own (f, o) of the same hidden unit: every lane evaluates two non-linearities instead of five and
// one v_permlane32_swap hands i*g to the lane that keeps c.
template <class S>
constexpr bool lstm_pair_mode() { return S::I[0] / 4 <= 2; }

template <class S, int NW_>
__device__ __forceinline__ void load_wfrag0_lstm(float (&w)[NW_], const float* packed, int wave, int lane) {
  using T = St<S, 0>;
  static_assert(NW_ == T::NWREG, "weight fragment array size");
  constexpr int P = S::I[0] / 4;
  const int r = lane & 15, q = lane >> 4;
  const int j = r & 3, qq = r >> 2;
#pragma unroll
  for (int x = 0; x < T::XM; ++x) {
    const int mt = T::SPLIT ? (wave % T::MT) : (wave + FAST_NW * x);
    int par = 4 * mt + qq;
    int i0 = j * P + par;
    if constexpr (lstm_pair_mode<S>()) {
      const int pair = qq >> 1;
      par = (j < 2) ? (qq & 1) : P;                                      // rows j = 2,3 -> zero
      const int gate = (j == 0) ? (pair ? 1 : 0) : (pair ? 3 : 2);
      i0 = gate * P + (qq & 1);
    }
#pragma unroll
    for (int u = 0; u < T::NU; ++u)
#pragma unroll
      for (int e = 0; e < T::WV; ++e) {
        const int kk = (4 * u + q) * T::WV + e;
        w[x * T::NSTEP + u * T::WV + e] = (mt < T::MT && par < P) ? packed[kk * T::M + i0] : 0.f;   // M = I_0
      }
  }
}

template <class S, bool DIAG>
__global__ void __launch_bounds__(FAST_NT) k_lstm_fwd_fused(int B, int T, const float* __restrict__ gin,
                                                            const float* __restrict__ h0, const float* __restrict__ c0,
                                                            const float* __restrict__ packed_hid,
                                                            const float* __restrict__ bias_hid, float* __restrict__ out,
                                                            float* __restrict__ hT, float* __restrict__ cT,
                                                            float* __restrict__ reserve) {
  constexpr int D = S::D;
  constexpr int H = in_size_of<S>();
  constexpr int GH = 4 * H;
  static_assert(out_size_of<S>() == GH, "TT output size must be 4 * hidden");
  constexpr int MID = maxmid_of<S>();
  using SL = St<S, D - 1>;
  using T0 = St<S, 0>;
  constexpr int P = S::I[0] / 4;
  static_assert(P * T0::ROWS == H, "hidden index decomposition");

  __shared__ __attribute__((aligned(16))) float hbuf[H];
  __shared__ __attribute__((aligned(16))) float bufA[MID];
  __shared__ __attribute__((aligned(16))) float bufB[D > 2 ? MID : 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  float w0[nwreg<S, 0>()];
  float w1[nwreg<S, (D > 1 ? 1 : 0)>()];
  float w2[nwreg<S, (D > 2 ? 2 : 0)>()];
  float w3[nwreg<S, (D > 3 ? 3 : 0)>()];
  load_wfrag0_lstm<S>(w0, packed_hid, wave, lane);
  if constexpr (D > 1) load_wfrag<S, 1>(w1, packed_hid, wave, lane);
  if constexpr (D > 2) load_wfrag<S, 2>(w2, packed_hid, wave, lane);
  if constexpr (D > 3) load_wfrag<S, 3>(w3, packed_hid, wave, lane);

  // hidden units owned by this lane (one per stage-0 tile of the wave).  gin is gate-interleaved
  // [B][T][H][4] with slot order i,g,f,o.  PAIR: lanes 0-31 take slots (i,g), lanes 32-63 (f,o) and own c.
  constexpr bool PAIR = lstm_pair_mode<S>();
  constexpr int NG = PAIR ? 2 : 4;                  // gates evaluated per lane
  const int pair = PAIR ? (q >> 1) : 1;             // 1 = this lane owns c / h of its hidden unit
  const float sgn = (PAIR && pair == 0) ? 2.0f : 1.0f;   // second gate: tanh (= 2*sigmoid(2x) - 1) or sigmoid
  int hid[T0::XM][T0::YR];
  bool ok[T0::XM][T0::YR];
  float cst[T0::XM][T0::YR], hst[T0::XM][T0::YR], bh[T0::XM][T0::YR][NG], gi[T0::XM][T0::YR][NG];
#pragma unroll
  for (int x = 0; x < T0::XM; ++x) {
    const int mt = T0::SPLIT ? (wave % T0::MT) : (wave + FAST_NW * x);
#pragma unroll
    for (int y = 0; y < T0::YR; ++y) {
      const int rt = T0::SPLIT ? (wave / T0::MT + T0::G * y) : y;
      const int row = 16 * rt + c, par = PAIR ? (q & 1) : (4 * mt + q);
      ok[x][y] = mt < T0::MT && rt < T0::RT && row < T0::ROWS && par < P;
      hid[x][y] = ok[x][y] ? par * T0::ROWS + row : 0;
      const int hd = hid[x][y];
      hst[x][y] = (ok[x][y] && h0) ? h0[b * H + hd] : 0.f;
      cst[x][y] = (ok[x][y] && c0) ? c0[b * H + hd] : 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        // slot s of the interleaved layout <-> reference gate index: i,g,f,o -> 0,2,1,3
        const int slot = PAIR ? 2 * pair + g : g;
        const int gate = slot == 1 ? 2 : (slot == 2 ? 1 : slot);
        bh[x][y][g] = (ok[x][y] && bias_hid) ? bias_hid[gate * H + hd] : 0.f;
        gi[x][y][g] = (ok[x][y] && T > 0) ? gin[((b * T) * H + hd) * 4 + slot] : 0.f;
      }
      if (ok[x][y] && pair) hbuf[a_off<SL::K>(hd / SL::K, hd % SL::K)] = hst[x][y];
    }
  }
  // All pre-loop global loads (core fragments, biases, first gate inputs) must be provably complete here:
  // otherwise hipcc guards the first use of a weight register inside the loop with s_waitcnt vmcnt(0), which in
  // steady state drains the gate-input prefetch issued a few hundred cycles earlier (one exposed HBM latency
  // per timestep).  0x0F70 = vmcnt(0) only.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  lds_barrier();
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();

  for (int t = 0; t < T; ++t) {
    const float* last;   // input image of stage 0
    if constexpr (D == 2) {
      run_stage<S, 1>(w1, hbuf, bufA, wave, lane);
      TT_STAMP(2)
      last = bufA;
    } else if constexpr (D == 3) {
      run_stage<S, 2>(w2, hbuf, bufA, wave, lane);
      TT_STAMP(0)
      lds_barrier();
      TT_STAMP(1)
      run_stage<S, 1>(w1, bufA, bufB, wave, lane);
      TT_STAMP(2)
      last = bufB;
    } else {
      run_stage<S, 3>(w3, hbuf, bufA, wave, lane);
      lds_barrier();
      run_stage<S, 2>(w2, bufA, bufB, wave, lane);
      lds_barrier();
      run_stage<S, 1>(w1, bufB, bufA, wave, lane);
      last = bufA;
    }
    lds_barrier();
    TT_STAMP(3)
    f32x4 acc[T0::XM][T0::YR];
    stage_mma<S, 0>(w0, last, acc, wave, lane);
    if constexpr (DIAG) {
      asm volatile("" : "+v"(acc[0][0]));
    }
    TT_STAMP(4)
    const size_t bt = b * T + t;
#pragma unroll
    for (int x = 0; x < T0::XM; ++x)
#pragma unroll
      for (int y = 0; y < T0::YR; ++y) {
        const int hd = hid[x][y];
        if constexpr (PAIR) {
          // lanes 0-31: (i, g); lanes 32-63: (f, o).  u = sigmoid(first), v = tanh|sigmoid(second)
          const float u = fsigmoid(acc[x][y][0] + gi[x][y][0] + bh[x][y][0]);                 // lstm.py:26-27
          const float a1 = acc[x][y][1] + gi[x][y][1] + bh[x][y][1];
          const float v = sgn * fsigmoid(sgn * a1) + (1.0f - sgn);                           // lstm.py:28-29
          const float prod = u * v;                                                          // i*g on lanes 0-31
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(prod), __float_as_uint(prod), false, false);
          const float ig_g = __uint_as_float(sw[0]);                                         // lanes 32-63 <- lanes 0-31
          const float cy = u * cst[x][y] + ig_g;                                             // lstm.py:31
          const float hy = v * ftanh(cy);                                                    // lstm.py:32
          if (reserve && ok[x][y]) {
            float* rv = reserve + (bt * H + hd) * 5 + 2 * pair;                              // i,g | f,o,c
            rv[0] = u; rv[1] = v;
            if (pair) rv[2] = cy;
          }
          if (ok[x][y] && pair) {
            cst[x][y] = cy;
            hst[x][y] = hy;
            out[bt * H + hd] = hy;                                                           // lstm.py:133
            hbuf[a_off<SL::K>(hd / SL::K, hd % SL::K)] = hy;
          }
          if (ok[x][y] && t + 1 < T) {
            const f32x2 nx = *reinterpret_cast<const f32x2*>(gin + ((bt + 1) * H + hd) * 4 + 2 * pair);
            gi[x][y][0] = nx[0]; gi[x][y][1] = nx[1];
          }
        } else if (ok[x][y]) {
          const float ig = fsigmoid(acc[x][y][0] + gi[x][y][0] + bh[x][y][0]);     // lstm.py:26
          const float gg = ftanh(acc[x][y][2] + gi[x][y][1] + bh[x][y][1]);        // lstm.py:28
          const float fg = fsigmoid(acc[x][y][1] + gi[x][y][2] + bh[x][y][2]);     // lstm.py:27
          const float og = fsigmoid(acc[x][y][3] + gi[x][y][3] + bh[x][y][3]);     // lstm.py:29
          const float cy = fg * cst[x][y] + ig * gg;                               // lstm.py:31
          const float hy = og * ftanh(cy);                                         // lstm.py:32
          cst[x][y] = cy;
          hst[x][y] = hy;
          out[bt * H + hd] = hy;                                                   // lstm.py:133
          hbuf[a_off<SL::K>(hd / SL::K, hd % SL::K)] = hy;
          if (reserve) {
            float* rv = reserve + (bt * H + hd) * 5;
            rv[0] = ig; rv[1] = gg; rv[2] = fg; rv[3] = og; rv[4] = cy;
          }
          if (t + 1 < T) {
            const f32x4 nx = *reinterpret_cast<const f32x4*>(gin + ((bt + 1) * H + hd) * 4);
            gi[x][y][0] = nx[0]; gi[x][y][1] = nx[1]; gi[x][y][2] = nx[2]; gi[x][y][3] = nx[3];
          }
        }
      }
    TT_STAMP(5)
    lds_barrier();
    TT_STAMP(6)
  }
#pragma unroll
  for (int x = 0; x < T0::XM; ++x)
#pragma unroll
    for (int y = 0; y < T0::YR; ++y)
      if (ok[x][y] && pair) {
        if (hT) hT[b * H + hid[x][y]] = hst[x][y];
        if (cT) cT[b * H + hid[x][y]] = cst[x][y];
      }
  if constexpr (DIAG) {
    // stamps leave through a buffer nothing else reads: the caller passes a scratch `reserve`
    if (lane == 0 && reserve && b < 8) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(reserve) + (b * FAST_NW + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = seg[i];
    }
  }
}

// ---- dispatch ------------------------------------------------------------------------------------------
template <class S>
static bool shape_matches(const TtShape& s) {
  if (s.d != S::D) return false;
  for (int k = 0; k < S::D; ++k)
    if (s.J[k] != S::J[k] || s.I[k] != S::I[k] || s.R[k] != S::R[k]) return false;
  return true;
}

template <class S, int CELL>
static int launch_one(const RnnShape& rs, const float* gin, const void* h0, const void* c0, const float* packed_hid,
                      const void* bias_hid, void* out, void* hT, void* cT, float* reserve, hipStream_t stream) {
  static_assert(shape_ok<S>(), "shape not supported by the MFMA path");
  if constexpr (CELL == TTRNN_LSTM && lstm_fusable<S>()) {
    const char* diag = getenv("TTRNN_DIAG");
    if (diag && diag[0] == '1' && reserve)   // diagnostic build: phase stamps overwrite the reserve buffer
      hipLaunchKernelGGL((k_lstm_fwd_fused<S, true>), dim3(rs.B), dim3(FAST_NT), 0, stream, rs.B, rs.T, gin,
                         (const float*)h0, (const float*)c0, packed_hid,
                         rs.has_bias_hid ? (const float*)bias_hid : (const float*)nullptr, (float*)out, (float*)hT,
                         (float*)cT, reserve);
    else
      hipLaunchKernelGGL((k_lstm_fwd_fused<S, false>), dim3(rs.B), dim3(FAST_NT), 0, stream, rs.B, rs.T, gin,
                         (const float*)h0, (const float*)c0, packed_hid,
                         rs.has_bias_hid ? (const float*)bias_hid : (const float*)nullptr, (float*)out, (float*)hT,
                         (float*)cT, reserve);
    return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((k_rnn_fwd_fast<S, CELL>), dim3(rs.B), dim3(FAST_NT), 0, stream, rs.B, rs.T, gin,
                     (const float*)h0, (const float*)c0, packed_hid,
                     rs.has_bias_hid ? (const float*)bias_hid : (const float*)nullptr, (float*)out, (float*)hT,
                     (float*)cT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// hidden-weight shapes with a specialised kernel
using ShpH256R8L = Shp<3, 4, 8, 8, 1, 8, 8, 16, 1, 8, 8, 1>;     // cfg2: TT-LSTM H=256 d=3 r=8
using ShpH256R8G = Shp<3, 4, 8, 8, 1, 8, 8, 12, 1, 8, 8, 1>;     // cfg3: TT-GRU  H=256 d=3 r=8
using ShpH256R16L = Shp<3, 4, 8, 8, 1, 8, 8, 16, 1, 16, 16, 1>;  // cfg4: TT-LSTM H=256 d=3 r=16
using ShpH256R16G = Shp<3, 4, 8, 8, 1, 8, 8, 12, 1, 16, 16, 1>;  //       TT-GRU  H=256 d=3 r=16
using ShpH128R4L = Shp<2, 8, 16, 1, 1, 16, 32, 1, 1, 4, 1, 1>;   // cfg1: TT-LSTM H=128 d=2 r=4

bool fast_rnn_fwd_available(const RnnShape& rs, int dtype) {
  if (dtype != TTRNN_F32 || rs.B < 1 || rs.T < 1) return false;
  if (rs.cell == TTRNN_LSTM)
    return shape_matches<ShpH256R8L>(rs.hid_s) || shape_matches<ShpH256R16L>(rs.hid_s) ||
           shape_matches<ShpH128R4L>(rs.hid_s);
  return shape_matches<ShpH256R8G>(rs.hid_s) || shape_matches<ShpH256R16G>(rs.hid_s);
}

int launch_rnn_fwd_fast(const RnnShape& rs, const float* gin, const void* h0, const void* c0,
                        const float* packed_hid, const void* bias_hid, void* out, void* hT, void* cT, float* reserve,
                        hipStream_t stream) {
#define TT_TRY(SHAPE, CELL)                                                                         \
  if (rs.cell == CELL && shape_matches<SHAPE>(rs.hid_s))                                            \
    return launch_one<SHAPE, CELL>(rs, gin, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, stream)
  TT_TRY(ShpH256R8L, TTRNN_LSTM);
  TT_TRY(ShpH256R16L, TTRNN_LSTM);
  TT_TRY(ShpH128R4L, TTRNN_LSTM);
  TT_TRY(ShpH256R8G, TTRNN_GRU);
  TT_TRY(ShpH256R16G, TTRNN_GRU);
#undef TT_TRY
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
