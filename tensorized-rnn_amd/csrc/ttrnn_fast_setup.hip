// ttrnn_fast_setup.hip — ONE launch for everything an input_size == 1 fused-core forward does before its recurrent kernel (gfx950).
//
// Until round 5 the headline call (cfg2: TT-LSTM in = 1, H = 256, d = 3, r = 8) paid four helper launches in front of K-rec —
// ttrnn_pack_cores2 (4.3 us), the chain kernel on the two unit input rows (9.9 us), k_f10h_scale (4.8 us), k_f10h_prep (4.5 us) —
// 41 us of a 518 us call with their boundaries: each is latency-bound work for a handful of workgroups.  None of them needs
// another's output through anything but the packed cores, and every workgroup can pack the 5 376 + 704 parameters it needs itself:
//     workgroups 0 .. M-1   row m of the fused core W10: hidden cores (strided Parameters -> packed layout in LDS) -> eu / ev (every
//                           workgroup, as k_f10h_scale) -> the row's 256 entries -> ep[m] -> the row's part of the MFMA fragments
//     workgroup  M          packs BOTH TT-matrices into the global packed buffers (the recurrent kernel, the backward pass and
//                           later calls read them), then runs the chain kernel's body (ttrnn_lin_dev.h) on the two unit rows
//                           x = [1, 0] from its LDS copy of the packed input cores
// Same expressions, same fmaf chains, same MFMA tiles as the four kernels it replaces: outputs are BIT-IDENTICAL to theirs
// (tests/test_gpu_parity.py::test_fused_setup_launch_is_bit_identical).  The C entry point is ttrnn_rnn_forward_cores.
// Replaces, per call: t3nsor/ops.py:47-51 (transpose / parameter views) + the weight-only part of t3nsor/ops.py:78-93 for the two
// TTLinears of tensorized_rnn/tt_lstm.py:16-40 / gru.py:148-172.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_f10.h"
#include "ttrnn_f10_dev.h"
#include "ttrnn_lin_dev.h"

namespace ttrnn {
namespace {

struct SetupArgs {
  const void* core_h[3];
  const void* core_i[3];
  int64_t st_h[12], st_i[12];
  float* packed_hid;
  float* packed_in;
  const float* bias_in;
  const float* unit;          // device constant {1, 0}
  float* gin;                 // [2][H][4]
  int* hdr;                   // F10H_HDR_BYTES
  xh8* wfrag;
};

template <class S>
constexpr int wtotal_of() { return woff_of<S>(S::D); }

// core k of a TT-matrix, strided Parameter -> packed W_k [K_k][M_k] (+ Wt_k [M_k][K_k] when wt != nullptr): pack_core_elems
template <class S, int k>
__device__ __forceinline__ void setup_pack_core(const float* core, const int64_t* st4, float* w, float* wt, int tid, int nthr) {
  constexpr int R0 = S::R[k], R1 = S::R[k + 1];
  constexpr int K = S::J[k] * R1, M = S::I[k] * R0;
  for (int e = tid; e < K * M; e += nthr) {
    const int m = e % M, kk = e / M;
    const int i = m / R0, a = m - i * R0;
    const int j = kk / R1, b = kk - j * R1;
    const float v = core[(size_t)(a * st4[0] + i * st4[1] + j * st4[2] + b * st4[3])];
    w[woff_of<S>(k) + e] = v;
    if (wt) wt[woff_of<S>(k) + (size_t)m * K + kk] = v;
  }
}

// GRU = false: LSTM fragments (rows permuted so that a lane holds i, f, g, o of one unit; gate factors folded in: k_f10h_prep)
// GRU = true:  fp32 GRU fragments (natural row order, no gate factor: k_f10gh_prep)
template <class S, class SI, bool GRU>
__global__ void __launch_bounds__(FAST_NT) k_f10_setup(SetupArgs a) {
  using F = F10<S>;
  static_assert(S::D == 3 && SI::D == 3 && F::K == 256 && F::I2 <= 16 && F::R2 <= 16, "fused set-up: d = 3, K10 = 256");
  constexpr int WT = wtotal_of<S>(), WTI = wtotal_of<SI>();
  __shared__ __attribute__((aligned(16))) float lw[WT];          // hidden cores, packed W layout
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  if (blockIdx.x == F::M) {
    // ---- the packing workgroup + the unit-row input projection ------------------------------------------------------------
    __shared__ __attribute__((aligned(16))) float li[WTI];       // input cores, packed W layout
    setup_pack_core<S, 0>((const float*)a.core_h[0], a.st_h + 0, a.packed_hid, a.packed_hid + WT, tid, FAST_NT);
    setup_pack_core<S, 1>((const float*)a.core_h[1], a.st_h + 4, a.packed_hid, a.packed_hid + WT, tid, FAST_NT);
    setup_pack_core<S, 2>((const float*)a.core_h[2], a.st_h + 8, a.packed_hid, a.packed_hid + WT, tid, FAST_NT);
    setup_pack_core<SI, 0>((const float*)a.core_i[0], a.st_i + 0, li, nullptr, tid, FAST_NT);
    setup_pack_core<SI, 1>((const float*)a.core_i[1], a.st_i + 4, li, nullptr, tid, FAST_NT);
    setup_pack_core<SI, 2>((const float*)a.core_i[2], a.st_i + 8, li, nullptr, tid, FAST_NT);
    setup_pack_core<SI, 0>((const float*)a.core_i[0], a.st_i + 0, a.packed_in, a.packed_in + WTI, tid, FAST_NT);
    setup_pack_core<SI, 1>((const float*)a.core_i[1], a.st_i + 4, a.packed_in, a.packed_in + WTI, tid, FAST_NT);
    setup_pack_core<SI, 2>((const float*)a.core_i[2], a.st_i + 8, a.packed_in, a.packed_in + WTI, tid, FAST_NT);
    __syncthreads();
    // (two rows per tile: a row's result does not depend on the rows stacked with it, and the sixteen-row instantiation of the
    // batched launch would bring 140 KB of LDS images into this kernel)
    ttlinear_fwd_fast_body<SI, 2, GRU ? 3 : 4, float, float>(2, li, a.bias_in, a.unit, a.gin, GRU ? 1 : 2, 0, 1);
    return;
  }

  // ---- row m of the fused core ---------------------------------------------------------------------------------------------
  setup_pack_core<S, 0>((const float*)a.core_h[0], a.st_h + 0, lw, nullptr, tid, FAST_NT);
  setup_pack_core<S, 1>((const float*)a.core_h[1], a.st_h + 4, lw, nullptr, tid, FAST_NT);
  setup_pack_core<S, 2>((const float*)a.core_h[2], a.st_h + 8, lw, nullptr, tid, FAST_NT);
  __shared__ unsigned mx[32];
  __shared__ int eu[16], ev[16];
  __shared__ float red[FAST_NW];
  __shared__ int ep_s;
  __shared__ __attribute__((aligned(16))) _Float16 fr[2][F::K];         // the row's pieces in fragment order
  if (tid < 32) mx[tid] = 0u;
  __syncthreads();
  const float* W2 = lw + woff_of<S>(2);                   // [J2][M2], m2 = i2 R2 + r2
  for (int i = tid; i < F::J2 * F::M2; i += FAST_NT)     // non-negative floats order like their bit patterns
    atomicMax(&mx[(i % F::M2) / F::R2], __float_as_uint(fabsf(W2[i])));
  __syncthreads();
  if (tid < 16) eu[tid] = tid < F::I2 ? -f10h_expo(__uint_as_float(mx[tid])) : 0;
  __syncthreads();
  for (int i = tid; i < F::J2 * F::M2; i += FAST_NT) {
    const int m2 = i % F::M2;
    atomicMax(&mx[16 + m2 % F::R2], __float_as_uint(fabsf(W2[i]) * ldexpf(1.f, eu[m2 / F::R2])));
  }
  __syncthreads();
  if (tid < 16) ev[tid] = tid < F::R2 ? -f10h_expo(__uint_as_float(mx[16 + tid])) : 0;
  __syncthreads();
  const int m = blockIdx.x;
  const int i0 = m / F::I1, i1 = m % F::I1;
  const float* W0 = lw + woff_of<S>(0);                   // [J0*R1][I0]
  const float* W1 = lw + woff_of<S>(1);                   // [J1*R2][I1*R1]
  // the row's entry for fragment element f = 32 u + 8 q + e (k in F10::kperm order): the fmaf chain of k_f10h_scale / k_f10h_prep
  float v = 0.f;
  int r2 = 0;
  if (tid < F::K) {
    const int slot = tid >> 3, e = tid & 7;              // slot = 4u + q
    r2 = (slot / F::HR) * 4 + (e & 3);
    const int row2 = 2 * (slot % F::HR) + (e >> 2);
    const int j1 = row2 % F::J1, j0 = row2 / F::J1;
    const float* w1p = W1 + (j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1;
    for (int r1 = 0; r1 < F::R1; ++r1) v = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0], w1p[r1], v);
  }
  float best = tid < F::K ? fabsf(v) * ldexpf(1.f, -ev[r2]) : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
  if (lane == 0) red[wave] = best;
  __syncthreads();
  if (tid == 0) {
    float b2 = red[0];
#pragma unroll
    for (int w = 1; w < FAST_NW; ++w) b2 = fmaxf(b2, red[w]);
    ep_s = 12 - f10h_expo(b2);
    a.hdr[F10H_EP + m] = ep_s;
  }
  if (blockIdx.x == 0 && tid < 16) {
    a.hdr[F10H_EU + tid] = eu[tid];
    a.hdr[F10H_EV + tid] = ev[tid];
  }
  __syncthreads();
  // fragment position of row m: tile t, MFMA row r
  int t, r;
  float gf = 1.0f;
  if constexpr (GRU) {
    t = m / 16; r = m % 16;
  } else {
    const int g = m / F::MPG, w = m % F::MPG;             // m = MPG (r & 3) + 4t + (r >> 2)
    t = w / 4; r = g + 4 * (w % 4);
    gf = g == 2 ? 2.8853900817779268f : -1.4426950408889634f;
  }
  if (tid < F::K) {
    _Float16 p0, p1;
    const float sc = ldexpf(1.f, ep_s - ev[r2]);          // f10h_w_scale
    if constexpr (GRU) split2h(v * sc, p0, p1);
    else split2h_scaled(v, gf * sc, p0, p1);
    fr[0][tid] = p0; fr[1][tid] = p1;
  }
  __syncthreads();
  if (tid < 2 * F::NM * 4) {                              // (plane, u, q): one 16-byte fragment slot each
    const int q = tid & 3, u = (tid >> 2) % F::NM, p = tid / (4 * F::NM);
    const xh8 val = *reinterpret_cast<const xh8*>(&fr[p][32 * u + 8 * q]);
    a.wfrag[(size_t)((t * F::NM + u) * 2 + p) * 64 + r + 16 * q] = val;
  }
}

template <class S, class SI, bool GRU>
int launch_setup_t(const SetupArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL((k_f10_setup<S, SI, GRU>), dim3(F10<S>::M + 1), dim3(FAST_NT), 0, stream, a);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

}  // namespace

// which (cell, shapes) have the fused set-up launch: fp32 storage, split math, input_size == 1, contiguous-or-strided fp32 cores
bool f10_setup_available(const RnnShape& rs, int dtype) {
  if (opt(OPT_DEV) & 1024 * 64) return false;
  if (dtype != TTRNN_F32 || opt(OPT_FP32_MATH) != TTRNN_MATH_SPLIT || opt(OPT_NO_F10) || opt(OPT_NO_IN1) || rs.in != 1) return false;
  if (rs.cell == TTRNN_LSTM) return shape_matches<ShpH256R8L>(rs.hid_s) && shape_matches<ShpI1R8L>(rs.in_s);
  return f10gh_available(rs, dtype) && shape_matches<ShpI1R8G>(rs.in_s);
}

// packed_in / packed_hid: written (2 * wtotal floats each); gin: the two unit rows [2][H][4]; ws: scale header + fragments
int launch_f10_setup(const RnnShape& rs, const void* const* cores_in, const int64_t* strides_in, const void* bias_in,
                     const void* const* cores_hid, const int64_t* strides_hid, float* packed_in, float* packed_hid, float* gin,
                     void* ws, hipStream_t stream) {
  if (!ws) return TTRNN_ERR_WORKSPACE;
  SetupArgs a;
  for (int k = 0; k < 3; ++k) {
    a.core_h[k] = cores_hid[k];
    a.core_i[k] = cores_in[k];
  }
  for (int q = 0; q < 12; ++q) {
    a.st_h[q] = strides_hid[q];
    a.st_i[q] = strides_in[q];
  }
  a.packed_hid = packed_hid;
  a.packed_in = packed_in;
  a.bias_in = rs.has_bias_in ? (const float*)bias_in : nullptr;
  a.unit = (const float*)unit_rows_ptr(TTRNN_F32);
  if (!a.unit) return TTRNN_ERR_LAUNCH;
  a.gin = gin;
  a.hdr = reinterpret_cast<int*>(ws);
  a.wfrag = reinterpret_cast<xh8*>(reinterpret_cast<unsigned char*>(ws) + F10H_HDR_BYTES);
  if (rs.cell == TTRNN_LSTM) return launch_setup_t<ShpH256R8L, ShpI1R8L, false>(a, stream);
  return launch_setup_t<ShpH256R8G, ShpI1R8G, true>(a, stream);
}

}  // namespace ttrnn
