// ttrnn_fast_f10gq.hip — the bf16-storage TT-GRU recurrent kernel (cfg3) as FOUR-wave workgroups with the gate math in the
// lanes that hold the fused stage's sums (gfx950).
//
// ttrnn_fast_f10.hip:k_gru_fwd_f10 leaves the hidden chain's pre-activations in an fp32 LDS vector and runs the gates as a
// third phase, one hidden unit per thread — three barriers per step, 1 785 cycles for less arithmetic than the LSTM kernel's
// 1 640 — because a GRU's r, z, n of one hidden unit do not share a lane of the fused stage's MFMA tiles: output o =
// m * I2 + i2 with I2 = 12 (m = i0 * I1 + i1 the row of the fused core), gate = o / 256, and 256 is no multiple of 12.
// Unit u = 12 m + c has
//     r at row m,                          column c
//     z at row m + 21 + (c >= 8),          column (c + 4) % 12          (256 = 21 * 12 + 4)
//     n at row m + 42 + (c >= 4),          column (c + 8) % 12          (512 = 42 * 12 + 8)
// so the columns differ by a ROTATION and the rows by a carry that depends on the column.  For TWO consecutive unit rows m, m + 1
// the z values lie in the three rows m + 21 .. m + 23 and the n values in m + 42 .. m + 44, so one lane group (q) of one wave
// takes a PAIR of unit rows through three 16 x 16 tiles over the SAME B operand (columns 12 .. 15 repeat columns 0 .. 3)
//     T1  rows m, m + 1                  -> r of units 12 m + c and 12 (m + 1) + c, in place
//     T2  rows m + 21, m + 22, m + 23    -> z, four columns to the right: one DPP row rotation per value (the repeated columns
//                                           make the rotation by 4 of 16 lanes a rotation of the 12 columns)
//     T3  rows m + 42, m + 43, m + 44    -> n, eight columns to the right: two DPP moves and a select per value
// after which lane (c, q) holds r, z, n of its two units in registers (row m + 21 / m + 22 resp. m + 22 / m + 23 by c >= 8 for
// z, likewise by c >= 4 for n): 24 MFMAs per wave and step instead of 8 (a pipe that idled 90 %) on the same 8 operand reads,
// 11 row pairs over the 16 (wave, q) slots of four waves, NO gate-vector round trip through LDS and two barriers per step
// instead of three.  (Feeding the three tiles three ROTATED reads of the image instead cost 3x the LDS reads: 0.78 ms against
// the eight-wave kernel's 0.57.)  Same bf16 operands and fp32 accumulation as k_gru_fwd_f10 (which sums the even and the odd
// k-blocks in two chains: the two kernels agree to fp32 rounding, not bit for bit).
// MEASURED (cfg3, B = 256, T = 784): 0.678 ms against the eight-wave kernel's 0.570 — with one wave per SIMD nothing overlaps the
// serial chain S2 -> barrier -> 24 MFMAs -> rotations -> two units' gates of each wave, and the two barriers saved do not pay
// for it.  Kept as an A/B variant (option dev, bit 2), not dispatched by default.
// Replaces tensorized_rnn/gru.py:33-44,124-134 with the hidden chain of t3nsor/ops.py:78-93 for the cfg3 shape.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_f10.h"

namespace ttrnn {

namespace {
constexpr int GQ_W = 4;                                    // waves per workgroup

// lane i of every 16-lane row receives the value of lane (i - N) mod 16 of its row (DPP row_ror:N)
template <int N>
__device__ __forceinline__ float dpp_row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}

template <class S>
constexpr bool f10gq_ok() {
  using F = F10<S>;
  return S::D == 3 && S::R[0] == 1 && S::R[3] == 1 && shape_ok_recurrent<S>() && F::K % 32 == 0 && F::M == 64 && F::I2 == 12 &&
         out_size_of<S>() == 3 * F::H && F::H == 256 && F::J2 == 8 && F::ROWS2 == 32 && F::M2 % 16 == 0 && F::R2 % 4 == 0 &&
         (2 * F::MT2) % GQ_W == 0;
}

// W10 row of accumulator register j of lane group q of wave w in tile tau (see the header); rows past the matrix and the
// unused registers repeat a valid row (their sums are never selected)
__host__ __device__ constexpr int gq_row(int w, int q, int tau, int j) {
  const int m = 2 * (w + GQ_W * q);
  int r = tau == 0 ? m + (j & 1) : (tau == 1 ? m + 21 + (j < 3 ? j : 0) : m + 42 + (j < 3 ? j : 0));
  return r > 63 ? 63 : r;
}

// fragments of the three tiles: wfrag[((w * 3 + tau) * NM + u) * 64 + lane], lane (rho = 4 q + j, k group kg), k = 32 u + 8 kg + e
// in F10::kperm order (as ttrnn_fast_f10.hip:k_f10g_prep)
template <class S>
__global__ void __launch_bounds__(64) k_f10gq_prep(const float* __restrict__ packed, xbf8* __restrict__ wfrag) {
  using F = F10<S>;
  const int lane = threadIdx.x;
  const int u = blockIdx.x % F::NM, tau = (blockIdx.x / F::NM) % 3, w = blockIdx.x / (3 * F::NM);
  const int rho = lane & 15, kg = lane >> 4;
  const int m = gq_row(w, rho >> 2, tau, rho & 3);
  const int i0 = m / F::I1, i1 = m % F::I1;
  const float* W0 = packed + woff_of<S>(0);
  const float* W1 = packed + woff_of<S>(1);
  xbf8 f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int slot = 4 * u + kg;
    const int r2 = (slot / F::HR) * 4 + (e & 3), row2 = 2 * (slot % F::HR) + (e >> 2);
    const int j1 = row2 % F::J1, j0 = row2 / F::J1;
    const float* w1p = W1 + (j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1;
    float v = 0.f;
    for (int r1 = 0; r1 < F::R1; ++r1) v = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0], w1p[r1], v);
    f[e] = (__bf16)v;
  }
  wfrag[(size_t)blockIdx.x * 64 + lane] = f;
}

template <class S>
__global__ void __launch_bounds__(GQ_W * 64) k_gru_fwd_f10gq(int B, int T, GinSrc gs, const bf16_t* __restrict__ h0,
                                                             const float* __restrict__ packed_hid,
                                                             const xbf8* __restrict__ wfrag,
                                                             const bf16_t* __restrict__ bias_hid, bf16_t* __restrict__ out,
                                                             bf16_t* __restrict__ hT, float* __restrict__ reserve) {
  static_assert(f10gq_ok<S>(), "shape not supported by the four-wave fused-core GRU kernel");
  using F = F10<S>;
  constexpr int H = F::H;
  constexpr int XT = 2 * F::MT2 / GQ_W;                    // S2 tiles (m-tile, chain-row tile) per wave

  __shared__ __attribute__((aligned(16))) __bf16 hq[H];                  // h_{t-1}, bf16, [ROWS2][J2]
  __shared__ __attribute__((aligned(16))) __bf16 img[F::PLANE];          // S10 operand [I2][K10]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  // S2 fragments: tile id = wave + 4 x -> (mt = id % MT2, rt = id / MT2); the 8 real k values live in k-group 0
  xbf8 a2[XT];
#pragma unroll
  for (int x = 0; x < XT; ++x) {
    const int id = wave + GQ_W * x, mt = id % F::MT2;
    const float* W2 = packed_hid + woff_of<S>(2);          // [J2][M2]
#pragma unroll
    for (int e = 0; e < 8; ++e) a2[x][e] = (__bf16)(q == 0 ? W2[e * F::M2 + 16 * mt + c] : 0.f);
  }
  xbf8 w10[3][F::NM];
#pragma unroll
  for (int tau = 0; tau < 3; ++tau)
#pragma unroll
    for (int u = 0; u < F::NM; ++u) w10[tau][u] = wfrag[(size_t)((wave * 3 + tau) * F::NM + u) * 64 + lane];

  // this lane's two hidden units: 12 m + c and 12 (m + 1) + c, m = 2 (wave + 4 q)
  const int mrow = 2 * (wave + GQ_W * q);
  const bool own[2] = {c < F::I2 && 12 * mrow + c < H, c < F::I2 && 12 * (mrow + 1) + c < H};
  const int hid[2] = {own[0] ? 12 * mrow + c : 0, own[1] ? 12 * (mrow + 1) + c : 0};
  const float* __restrict__ gin = gs.gin;
  const bf16_t* __restrict__ xs = reinterpret_cast<const bf16_t*>(gs.x);
  const bool in1 = gs.in1 != 0;
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gin);
  float hst[2], bh[2][3];
  f32x4 gi[2], vv[2], bb[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    hst[s] = (own[s] && h0) ? ld(h0, b * H + hid[s]) : 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[s][g] = (own[s] && bias_hid) ? ld(bias_hid, g * H + hid[s]) : 0.f;
    gi[s] = f32x4{0.f, 0.f, 0.f, 0.f}; vv[s] = gi[s]; bb[s] = gi[s];
    if (own[s] && T > 0) {
      if (in1) {
        bb[s] = gin4[H + hid[s]];
        vv[s] = gin4[hid[s]] - bb[s];
      } else {
        gi[s] = gin4[(b * T) * H + hid[s]];
      }
    }
    if (own[s]) hq[hid[s]] = (__bf16)hst[s];
  }
  XChunk<bf16_t> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
  // image row this lane feeds the B operand of all three tiles: column c, columns 12 .. 15 repeat columns 0 .. 3
  const int brow = c < F::I2 ? c : c - F::I2;
  const bool zc = c >= 8, nc = c >= 4;                     // the carries of the z / n rows
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): keep weight-register waits out of the time loop
  lds_barrier();

  for (int t = 0; t < T; ++t) {
    // ---- A: S2 ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int x = 0; x < XT; ++x) {
      const int id = wave + GQ_W * x;
      const int mt = id % F::MT2, rt = id / F::MT2;
      const int row = 16 * rt + c;
      const xbf8 bfrag = *reinterpret_cast<const xbf8*>(hq + row * 8);       // every k-group reads the same 16 bytes
      const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[x], bfrag, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const int m0 = 16 * mt + 4 * q;
      const int i = m0 / F::R2, a0 = m0 % F::R2;
      xbf4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (__bf16)acc[j];
      *reinterpret_cast<xbf4*>(img + x_off<F::K>(i, F::kperm(row, a0))) = o;
    }
    lds_barrier();
    // ---- B: the fused stage as three tiles, then gates + state in the same lanes (gru.py:38-44) ---------------------
    f32x4 t1 = f32x4{0.f, 0.f, 0.f, 0.f}, t2 = t1, t3 = t1;
    {
      constexpr int PD = 4;
      xbf8 af[F::NM];
#pragma unroll
      for (int u = 0; u < PD; ++u) af[u] = *reinterpret_cast<const xbf8*>(img + x_off<F::K>(brow, 32 * u + 8 * q));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < F::NM; ++u) {
        if (u + PD < F::NM) af[u + PD] = *reinterpret_cast<const xbf8*>(img + x_off<F::K>(brow, 32 * (u + PD) + 8 * q));
        __builtin_amdgcn_sched_barrier(0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10[0][u], af[u], t1, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10[1][u], af[u], t2, 0, 0, 0);
        t3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10[2][u], af[u], t3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // z sits four columns to the right, n eight (columns 12 .. 15 = 0 .. 3): lane c <- lane c + 4 is a rotation of the 16-lane
    // row by 12 to the right; lane c <- lane c + 8 (c < 8) resp. lane c - 4 (c >= 8) for n
    float zv[3], nv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      zv[j] = dpp_row_ror<12>(t2[j]);
      const float n8 = dpp_row_ror<8>(t3[j]), n4 = dpp_row_ror<4>(t3[j]);
      nv[j] = zc ? n4 : n8;
    }
    const size_t bt = b * T + t;
    const float xt = in1 ? xq.at(t) : 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float yr = t1[s];
      const float yz = zc ? zv[s + 1] : zv[s];
      const float yn = nc ? nv[s + 1] : nv[s];
      if (in1) gi[s] = bb[s] + xt * vv[s];
      const float hn = yn + bh[s][2];
      const float rg = fsigmoid(gi[s][0] + yr + bh[s][0]);              // gru.py:38-39
      const float zg = fsigmoid(gi[s][1] + yz + bh[s][1]);              // gru.py:40-41
      const float ng = ftanh(gi[s][2] + rg * hn);                       // gru.py:42-43
      float hy = (1.0f - zg) * ng + zg * hst[s];                        // gru.py:44
      if (own[s]) {
        if (reserve) *reinterpret_cast<f32x4*>(reserve + (bt * H + hid[s]) * 4) = f32x4{rg, zg, ng, hn};
        const bf16_t hb = f32_to_bf16(hy);            // rounded once: stored, fed back, kept as state
        if (out) out[bt * H + hid[s]] = hb;
        hy = bf16_to_f32(hb);
        hst[s] = hy;
        hq[hid[s]] = (__bf16)hy;
        if (!in1 && t + 1 < T) gi[s] = gin4[(bt + 1) * H + hid[s]];
      }
    }
    if (in1) xq.advance(xs, b * T, T, t, lane);
    lds_barrier();
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
    if (own[s] && hT) st(hT, b * H + hid[s], hst[s]);
}
}  // namespace

size_t f10gq_workspace_bytes() { return (size_t)GQ_W * 3 * F10<ShpH256R8G>::NM * 64 * sizeof(xbf8); }

bool f10gq_available(const RnnShape& rs, int dtype) {
  return dtype == TTRNN_BF16 && rs.cell == TTRNN_GRU && shape_matches<ShpH256R8G>(rs.hid_s);
}

int launch_gru_fwd_f10gq(const RnnShape& rs, GinSrc gin, const void* h0, const float* packed_hid, const void* bias_hid,
                         void* out, void* hT, float* reserve, void* ws, hipStream_t stream, int phase) {
  using S = ShpH256R8G;
  if (!ws) return TTRNN_ERR_WORKSPACE;
  xbf8* wfrag = reinterpret_cast<xbf8*>(ws);
  if (phase != TTRNN_PHASE_RUN)
    hipLaunchKernelGGL((k_f10gq_prep<S>), dim3(GQ_W * 3 * F10<S>::NM), dim3(64), 0, stream, packed_hid, wfrag);
  if (phase == TTRNN_PHASE_PREPARE) return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  hipLaunchKernelGGL((k_gru_fwd_f10gq<S>), dim3(rs.B), dim3(GQ_W * 64), 0, stream, rs.B, rs.T, gin, (const bf16_t*)h0,
                     packed_hid, wfrag, rs.has_bias_hid ? (const bf16_t*)bias_hid : (const bf16_t*)nullptr, (bf16_t*)out,
                     (bf16_t*)hT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

}  // namespace ttrnn
