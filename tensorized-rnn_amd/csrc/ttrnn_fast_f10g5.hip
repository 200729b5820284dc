// ttrnn_fast_f10g5.hip — fused-core TT-GRU forward kernel for fp32 storage, H = 512, r = 8: GATES ON THE ACCUMULATORS (gfx950).
//
// The reference's benchmark defaults with --gru (experiments/digit_classification/benchmarking.py:75-83: in = 256, H = 512, d = 3,
// rank 8, batch 512, 160 steps) give the hidden matrix the modes J = (8, 8, 8), I = (8, 12, 16).  Unlike H = 256 (I2 = 12: the
// r / z / n of a unit fall into different columns, SURVEY 7.2 — k_gru_fwd_f10vh hands the gate vector through LDS) this shape has
// I2 = 16 and H / 16 = 32 rows of the fused core per gate: flat gate index o = g H + hid = m I2 + i2 with m = 32 g + hid / 16,
// i2 = hid % 16 — the three pre-activations of a unit share a COLUMN and sit 32 rows apart.  So the TT-LSTM kernel's layout
// (ttrnn_fast_f10q.hip, eight waves at H = 512) carries over with three gates: MFMA row 4q + j of tile t is fused-core row
// m = 32 j + 4 t + q for j < 3 (row 3 of every quad is a zero row), and lane (c, q) of wave t holds r, z and the hidden part of n
// of ONE unit, hid = 64 t + 16 q + c = 64 t + lane, in accumulator registers 0, 1, 2: the gate math (gru.py:38-44) runs on the
// accumulators, h goes straight back to the image S2 reads — two barriers per step, no gate vector in LDS.
//     phase A  S2: one m-tile x four chain-row tiles per wave (four terms packed into one MFMA per tile), split -> S10 image
//     phase B  S10: sixteen k-blocks x three terms, gates, h -> LDS
// Same two-piece fp16 arithmetic and diagonal power-of-two scales as the LSTM kernels (ttrnn_f10_dev.h; k_f10h_scale); the rows
// of r and z carry -log2(e) (sigmoid = rcp(1 + exp2(.))), the n row carries no factor (its hidden part is multiplied by r before
// the tanh: gru.py:42-43).  gin: the runtime-shape tier's K-in (one dense GEMM over the B T rows, slots r, z with both biases,
// n with the input bias, slot 3 = the hidden bias of n: ttrnn_g2.hip:k_g2_bias).  A caller's h_0 outside (-1, 1): as
// k_gru_fwd_f10vh, the state's scale per sample and per step while it stays large.
// Replaces, for one layer: tensorized_rnn/gru.py:33-44,124-134 with the hidden chain of t3nsor/ops.py:78-93.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_f10.h"
#include "ttrnn_f10_dev.h"

namespace ttrnn {

using ShpH512R8G = Shp<3, 8, 8, 8, 1, 8, 12, 16, 1, 8, 8, 1>;      // benchmarking.py defaults with --gru: TT-GRU H = 512 d = 3 r = 8

template <class S>
struct F10G3 {
  using F = F10<S>;
  static constexpr int H = F::H;
  static constexpr int MPG = F::M / 3;                     // fused-core rows per gate
  static constexpr int NT = MPG / 4;                       // S10 tiles = waves (each: four rows of every gate + four zero rows)
};

template <class S>
constexpr bool f10g3_ok() {
  using F = F10<S>;
  using G = F10G3<S>;
  return S::D == 3 && S::R[0] == 1 && S::R[3] == 1 && shape_ok_recurrent<S>() && out_size_of<S>() == 3 * F::H && F::I2 == 16 &&
         F::M % 3 == 0 && G::MPG * F::I2 == F::H && G::MPG % 4 == 0 && G::NT == 8 && F::K % 64 == 0 && F::J2 == 8 &&
         F::ROWS2 == 64 && F::MT2 % G::NT == 0 && F::R2 % 4 == 0 && F::R2 <= 16;
}

// fragments: tile t, k-block u, plane p: lane (r, q) = MFMA row r <-> gate r & 3 (3: zero row), m = MPG (r & 3) + 4 t + (r >> 2)
template <class S>
__global__ void __launch_bounds__(64) k_f10g3_prep(const float* __restrict__ packed, const float* __restrict__ hdr,
                                                   xh8* __restrict__ wfrag) {
  using F = F10<S>;
  using G = F10G3<S>;
  const int lane = threadIdx.x, u = blockIdx.x % F::NM, t = blockIdx.x / F::NM;
  const int r = lane & 15, q = lane >> 4;
  const int g = r & 3;
  const int m = G::MPG * (g < 3 ? g : 0) + 4 * t + (r >> 2);
  const int i0 = m / F::I1, i1 = m % F::I1;
  const float* W0 = packed + woff_of<S>(0);               // [J0*R1][I0]
  const float* W1 = packed + woff_of<S>(1);               // [J1*R2][I1*R1]
  const float gf = g == 2 ? 1.0f : -1.4426950408889634f;
  xh8 f0, f1;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int slot = 4 * u + q;                           // k = 8*slot + e in F10::kperm order
    const int r2 = (slot / F::HR) * 4 + (e & 3), row2 = 2 * (slot % F::HR) + (e >> 2);
    const int j1 = row2 % F::J1, j0 = row2 / F::J1;
    const float* w1p = W1 + (j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1;
    float v = 0.f;
    for (int r1 = 0; r1 < F::R1; ++r1) v = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0], w1p[r1], v);
    _Float16 p0 = (_Float16)0.f, p1 = p0;
    if (g < 3) split2h_scaled(v, gf * f10h_w_scale<S>(hdr, m, r2), p0, p1);
    f0[e] = p0; f1[e] = p1;
  }
  xh8* dst = wfrag + (size_t)((t * F::NM + u) * 2) * 64 + lane;
  dst[0] = f0; dst[64] = f1;
}

template <class S>
constexpr size_t f10g3_ws_bytes() { return F10H_HDR_BYTES + (size_t)F10G3<S>::NT * F10<S>::NM * 2 * 64 * sizeof(xh8); }
template <class S>
constexpr size_t f10g3_lds_bytes() { return 2 * 2 * 2 * (size_t)F10<S>::H + 2 * 2 * (size_t)F10<S>::PLANE; }

template <class S, bool H0, bool OUT>
__global__ void __launch_bounds__(F10G3<S>::NT * 64, 1) k_gru_fwd_f10g5(int B, int T, const float* __restrict__ gin,
                                                                        const float* __restrict__ h0,
                                                                        const float* __restrict__ packed_hid,
                                                                        const float* __restrict__ hdr,
                                                                        const xh8* __restrict__ wfrag, float* __restrict__ out,
                                                                        float* __restrict__ hT, float* __restrict__ reserve) {
  static_assert(f10g3_ok<S>(), "shape not supported by the in-lane-gates fused-core GRU kernel");
  using F = F10<S>;
  using G = F10G3<S>;
  constexpr int QW = G::NT;
  constexpr int H = F::H;
  constexpr int XQ = F::MT2 / QW;                        // S2 m-tiles per wave
  constexpr int RT2 = F::RT2;                            // chain-row tiles of S2
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_g5[];
  __shared__ float hmax[QW];
  _Float16* hpl = reinterpret_cast<_Float16*>(smem_g5);  // fp16 pieces of the scaled h: [parity][2][H]
  _Float16* img = hpl + 2 * 2 * H;                       // two fp16 planes [I2][K10]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  xh8 s1[XQ];
#pragma unroll
  for (int x = 0; x < XQ; ++x) f10h_load_w2<S>(s1[x], packed_hid, wave + QW * x, lane, hdr);
  xh8 w10[2][F::NM];
#pragma unroll
  for (int u = 0; u < F::NM; ++u)
#pragma unroll
    for (int p = 0; p < 2; ++p) w10[p][u] = wfrag[(size_t)((wave * F::NM + u) * 2 + p) * 64 + lane];
  // scales of accumulator register j (gate j) of lane (c, q): row m = MPG j + 4 wave + q, column i2 = c
  f32x4 psc, usc;
  {
    const int* e = reinterpret_cast<const int*>(hdr);
    const int eu = e[F10H_EU + c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = e[F10H_EP + G::MPG * (j < 3 ? j : 0) + 4 * wave + q] + eu + 12;
      psc[j] = ldexpf(1.f, s);
      usc[j] = ldexpf(1.f, -s);
    }
  }
  const int hd = 64 * wave + lane;                        // = (4 wave + q) I2 + c
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gin);
  float hst = H0 ? h0[b * H + hd] : 0.f;
  int e_cur = 0;
  if constexpr (H0) e_cur = __builtin_amdgcn_readfirstlane(f10h_h0_expo<QW>(hst, reinterpret_cast<float*>(img), wave, lane));
  float hsc = ldexpf(F10H_HSC, -e_cur);
  float esc = ldexpf(1.f, e_cur);                         // the accumulators of a step run on 2^-e h: multiplied back
  f32x4 gi = T > 0 ? gin4[(b * T) * H + hd] : f32x4{0.f, 0.f, 0.f, 0.f};
  // slots r, z <- (gin + biases) (-log2 e) scale;  the hidden part of n starts from its bias (slot 3), unit factor
  const f32x4 gsc = f32x4{-1.4426950408889634f * psc[0], -1.4426950408889634f * psc[1], psc[2], 0.f};
  {
    _Float16 p0, p1;                                       // parity 0 = h_{-1}
    split2h(hst * hsc, p0, p1);
    hpl[hd] = p0; hpl[H + hd] = p1;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) here, so that no weight-register wait lands inside the loop
  lds_barrier();

  for (int t = 0; t < T; ++t) {
    const _Float16* hp = hpl + (t & 1) * 2 * H;           // pieces of h_{t-1}
    _Float16* hn_ = hpl + ((t + 1) & 1) * 2 * H;          // pieces of h_t
    // ---- phase A: S2 (MFMAs first, then the splitting) -------------------------------------------------------------
    constexpr int TPW = XQ * RT2;
    static_assert(TPW % 4 == 0, "tiles in groups of four");
#pragma unroll
    for (int g0 = 0; g0 < TPW; g0 += 4) {
      f32x4 t2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) t2[i] = f10h_s2_mma<S>(s1[(g0 + i) / RT2], hp, (g0 + i) % RT2, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) f10h_s2_store<S>(t2[i], img, wave + QW * ((g0 + i) / RT2), (g0 + i) % RT2, lane);
    }
    lds_barrier();
    const size_t bt = b * T + t;
    // ---- phase B: the fused S1*S0 stage, then gates + state (gru.py:38-44) ---------------------------------------
    f32x4 acc;
    {
      // the scaled domain of this step's accumulators carries 2^-e of the state's scale (H0): the initial values alike
      f32x4 pre = f32x4{gi[0], gi[1], gi[3], 0.f} * gsc;
      if constexpr (H0) pre = pre * ldexpf(1.f, -e_cur);
      f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = pre;
      f10h_s10_part<S, F::NM>(w10, img, c, q, 0, acc_lo, acc_hi);
      const f32x4 un = H0 ? usc * esc : usc;
      acc = acc_hi * un + acc_lo * un;
    }
    const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0]));               // gru.py:38-39
    const float zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[1]));               // gru.py:40-41
    const float hn = acc[2];                                                                      // W_hn h + b_hn
    const float ng = ftanh(gi[2] + rg * hn);                                                      // gru.py:42-43
    const float hy = (1.0f - zg) * ng + zg * hst;                                                 // gru.py:44
    hst = hy;
    if constexpr (H0) {
      if (e_cur > 0) {                   // workgroup-uniform: the state was outside (-1, 1) — re-derive its exponent
        float mx = fabsf(hy);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) hmax[wave] = mx;
        lds_barrier();
        mx = hmax[0];
#pragma unroll
        for (int w = 1; w < QW; ++w) mx = fmaxf(mx, hmax[w]);
        const int e = f10h_expo(mx);
        e_cur = __builtin_amdgcn_readfirstlane(e < 0 ? 0 : e);
        hsc = ldexpf(F10H_HSC, -e_cur);
        esc = ldexpf(1.f, e_cur);
      }
    }
    {
      _Float16 p0, p1;
      split2h(hy * hsc, p0, p1);
      hn_[hd] = p0; hn_[H + hd] = p1;
    }
    if constexpr (OUT) out[bt * H + hd] = hy;               // outputs[:, t, :] (gru.py:134)
    if (reserve) *reinterpret_cast<f32x4*>(reserve + (bt * H + hd) * 4) = f32x4{rg, zg, ng, hn};
    if (t + 1 < T) gi = gin4[(bt + 1) * H + hd];
    lds_barrier();
  }
  if (hT) hT[b * H + hd] = hst;
}

bool f10g5_available(const RnnShape& rs, int dtype) {
  return !opt(OPT_NO_F10) && !(opt(OPT_DEV) & 256) && rs.B >= 1 && rs.T >= 1 && dtype == TTRNN_F32 && rs.cell == TTRNN_GRU &&
         rs.hid_blocks <= 1 && rs.in != 1 && opt(OPT_FP32_MATH) == TTRNN_MATH_SPLIT && shape_matches<ShpH512R8G>(rs.hid_s);
}
size_t f10g5_workspace_bytes(const RnnShape& rs) {
  return shape_matches<ShpH512R8G>(rs.hid_s) ? f10g3_ws_bytes<ShpH512R8G>() : 0;
}
// gin: the runtime-shape tier's (ttrnn_g2.hip: fwd_t); ws: its rec region (f10g5_workspace_bytes)
int launch_gru_fwd_f10g5(const RnnShape& rs, const float* gin, const void* h0, const float* packed_hid, void* out, void* hT,
                         float* reserve, void* ws, hipStream_t stream) {
  using S = ShpH512R8G;
  if (!ws) return TTRNN_ERR_WORKSPACE;
  if (!shape_matches<S>(rs.hid_s)) return TTRNN_ERR_UNSUPPORTED;
  static_assert((F10H_EP + F10<S>::M) * sizeof(int) <= F10H_HDR_BYTES, "header");
  float* hdr = reinterpret_cast<float*>(ws);
  xh8* wfrag = reinterpret_cast<xh8*>(reinterpret_cast<unsigned char*>(ws) + F10H_HDR_BYTES);
  hipLaunchKernelGGL((k_f10h_scale<S>), dim3(F10<S>::M), dim3(256), 0, stream, packed_hid, reinterpret_cast<int*>(ws));
  hipLaunchKernelGGL((k_f10g3_prep<S>), dim3(F10G3<S>::NT * F10<S>::NM), dim3(64), 0, stream, packed_hid, hdr, wfrag);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  constexpr size_t lds = f10g3_lds_bytes<S>();
  auto kern = out ? (h0 ? k_gru_fwd_f10g5<S, true, true> : k_gru_fwd_f10g5<S, false, true>)
                  : (h0 ? k_gru_fwd_f10g5<S, true, false> : k_gru_fwd_f10g5<S, false, false>);
  if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(rs.B), dim3(F10G3<S>::NT * 64), lds, stream, rs.B, rs.T, gin, (const float*)h0, packed_hid, hdr, wfrag,
                     (float*)out, (float*)hT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

}  // namespace ttrnn
