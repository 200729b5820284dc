// ttrnn_fast_f10gh.hip — the fused-core TT-GRU forward kernel for fp32 storage on two fp16 pieces (gfx950).
//
// The reference's GRU exists in fp32 only (tensorized_rnn/gru.py:25-50, time loop :104-136); until round 5 this library's
// fused-core GRU kernel (k_gru_fwd_f10v, ttrnn_fast_f10.hip) served bf16 storage and fp32 GRUs ran on the runtime-shape tier.
// This kernel is that four-wave kernel's structure — S2 of the new state INSIDE the gate waves (the state of a chain row lives in
// eight lanes of one wave: gathered through the LDS crossbar, no memory, no barrier), [S10 | barrier | gates + S2 | barrier] per
// step — on the arithmetic of the fp32 LSTM kernels (ttrnn_f10_dev.h): every fp32 operand as TWO fp16 pieces under exact
// power-of-two DIAGONAL scales (k_f10h_scale: eu per output-mode index, ev per rank index of core 2, ep per row of the fused
// core), products x0w0 + x0w1 + x1w0 (S10) / all four terms (S2: packed along the k of ONE MFMA), fp32 accumulation:
//     S2   on tile pairs (ttrnn_f10_dev.h: f10p_*): four ds_bpermute of the lanes' packed piece pairs, six MFMAs, THREE splitting
//          passes with every lane live, results into the two fp16 planes of the S10 operand
//     S10  one 16-feature tile per wave, 8 k-blocks x 3 terms = 24 MFMAs, un-scaled per (row, column) into the fp32 gate vector
//     gates (gru.py:38-44) one hidden unit per thread.
// A GRU's r, z, n of one unit do not land in one lane (o = m I2 + i2 with I2 = 12 crosses the gate boundaries: SURVEY 7.2), so
// the fused core's rows keep their natural order and carry no gate factor.
// h_0: |h_t| <= max(1, |h_{t-1}|) only (h_t = (1-z) n + z h_{t-1}), so a caller's h_0 outside (-1, 1) fixes the scale of h for as
// long as the state stays large: the per-sample exponent e_t (max |h_t| < 2^e_t) is recomputed every step WHILE it is positive
// (one extra barrier in those steps only; a workgroup-uniform branch) and stays 0 afterwards.
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

template <class S>
constexpr bool f10gh_ok() {
  using F = F10<S>;
  return S::D == 3 && S::R[0] == 1 && S::R[3] == 1 && shape_ok_recurrent<S>() && F::K % 64 == 0 && F::M == 64 &&
         F::I2 <= 16 && out_size_of<S>() == 3 * F::H && F::J2 == 8 && F::ROWS2 == 32 && F::M2 % 16 == 0 && F::R2 % 4 == 0 &&
         F::R2 <= 16 && F::H == 256;
}

// The fused core in fragment order, NATURAL feature order (MFMA row r of tile t = feature 16t + r), rows under the header's
// 2^(ep[m] - ev[r2]), as two fp16 pieces:  wfrag[((t*NM + u)*2 + plane)*64 + lane]   (the LSTM twin: k_f10h_prep)
template <class S>
__global__ void __launch_bounds__(64) k_f10gh_prep(const float* __restrict__ packed, const float* __restrict__ hdr,
                                                   xh8* __restrict__ wfrag) {
  using F = F10<S>;
  const int lane = threadIdx.x, u = blockIdx.x % F::NM, t = blockIdx.x / F::NM;
  const int r = lane & 15, q = lane >> 4;
  const int m = 16 * t + r;
  const int i0 = m / F::I1, i1 = m % F::I1;
  const float* W0 = packed + woff_of<S>(0);               // [J0*R1][I0]
  const float* W1 = packed + woff_of<S>(1);               // [J1*R2][I1*R1]
  xh8 f0, f1;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int slot = 4 * u + q;                           // k = 8*slot + e in F10::kperm order
    const int r2 = (slot / F::HR) * 4 + (e & 3), row2 = 2 * (slot % F::HR) + (e >> 2);
    const int j1 = row2 % F::J1, j0 = row2 / F::J1;
    const float* w1p = W1 + (j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1;
    float v = 0.f;
    for (int r1 = 0; r1 < F::R1; ++r1) v = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0], w1p[r1], v);
    _Float16 p0, p1;
    split2h(v * f10h_w_scale<S>(hdr, m, r2), p0, p1);
    f0[e] = p0; f1[e] = p1;
  }
  xh8* dst = wfrag + (size_t)((t * F::NM + u) * 2) * 64 + lane;
  dst[0] = f0; dst[64] = f1;
}

template <class S>
constexpr size_t f10gh_ws_bytes() { return F10H_HDR_BYTES + (size_t)4 * F10<S>::NM * 2 * 64 * sizeof(xh8); }

// H0: the caller passed an initial state; OUT = false: only the final state is consumed; IN1: input_size == 1 (lesson 44: template
// parameters, not runtime flags, inside a persistent time loop); DIAG: s_memtime stamps (tools/diag_stamps.py);
// G2GIN: gin comes from the runtime-shape tier's K-in (ttrnn_g2.hip: one dense GEMM over the B T rows) in ITS convention — slots
// r, z carry both biases, slot n the input bias, slot 3 the hidden bias of the n gate (k_g2_bias) — and bias_hid is not read
template <class S, bool H0, bool OUT, bool IN1, bool DIAG = false, bool G2GIN = false>
__global__ void __launch_bounds__(256, 2) k_gru_fwd_f10vh(int B, int T, GinSrc gs, const float* __restrict__ h0,
                                                          const float* __restrict__ packed_hid,
                                                          const float* __restrict__ hdr, const xh8* __restrict__ wfrag,
                                                          const float* __restrict__ bias_hid, float* __restrict__ out,
                                                          float* __restrict__ hT, float* __restrict__ reserve) {
  static_assert(f10gh_ok<S>(), "shape not supported by the two-piece fused-core GRU kernel");
  using F = F10<S>;
  constexpr int H = F::H;
  __shared__ __attribute__((aligned(16))) _Float16 img[2 * F::PLANE];    // S10 operand, two fp16 planes [I2][K10]
  __shared__ __attribute__((aligned(16))) float gbuf[3 * H + 128];       // SCALED gate sums of the hidden chain (+ a dump for the padding columns)
  __shared__ float hmax[4];                                              // H0: the waves' maxima of |h_t|

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  // S2 fragments of ALL m-tiles, as tile pairs with the four terms packed along k (f10p_load_w2): in registers, or (r = 16: twelve
  // m-tiles, 48 VGPRs beside 128 of fused-core fragments) in LDS — every wave needs the same set, wave 0 writes it
  constexpr bool A2LDS = F10P<S>::NP > 3;
  __shared__ __attribute__((aligned(16))) xh8 afr[A2LDS ? 2 * F10P<S>::NP * 64 : 1];
  xh8 a1[A2LDS ? 1 : F10P<S>::NP], a2[A2LDS ? 1 : F10P<S>::NP];
  if constexpr (A2LDS) {
    if (wave == 0) {
      xh8 t1[F10P<S>::NP], t2[F10P<S>::NP];
      f10p_load_w2<S>(t1, t2, packed_hid, lane, hdr);
#pragma unroll
      for (int p = 0; p < F10P<S>::NP; ++p) {
        afr[(2 * p) * 64 + lane] = t1[p];
        afr[(2 * p + 1) * 64 + lane] = t2[p];
      }
    }
    __syncthreads();
  } else {
    f10p_load_w2<S>(a1, a2, packed_hid, lane, hdr);
  }
  F10pLane<S> ln;
  ln.init(wave, lane);
  xh8 w10[2][F::NM];
#pragma unroll
  for (int u = 0; u < F::NM; ++u)
#pragma unroll
    for (int p = 0; p < 2; ++p) w10[p][u] = wfrag[(size_t)((wave * F::NM + u) * 2 + p) * 64 + lane];
  // The accumulators of S10 carry 2^(ep[m] + eu[i2] + 12); they go to the gate vector as they are (one add per register) and the
  // GATE thread multiplies its three sums back (a per-thread constant folded into an fma): o = g H + hid = m I2 + i2
  f32x4 usc;                // gate g of this thread's unit (slot 3 unused)
  {
    const int* e = reinterpret_cast<const int*>(hdr);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int o = g * H + tid;
      usc[g] = ldexpf(1.f, -(e[F10H_EP + o / F::I2] + e[F10H_EU + o % F::I2] + 12));
    }
    usc[3] = 0.f;
  }
  // where lane (c, q) puts accumulator register j: row m = 16 wave + 4q + j, column i2 = c (columns >= I2: the dump)
  const int gdst = c < F::I2 ? (16 * wave + 4 * q) * F::I2 + c : 3 * H + lane;

  const float* __restrict__ gin = gs.gin;
  const float* __restrict__ xs = reinterpret_cast<const float*>(gs.x);
  constexpr bool in1 = IN1;
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gin);
  const int hid = tid;
  float hst = H0 ? h0[b * H + hid] : 0.f;
  float bh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) bh[g] = (!G2GIN && bias_hid) ? bias_hid[g * H + hid] : 0.f;
  f32x4 gi = f32x4{0.f, 0.f, 0.f, 0.f}, vv = gi, bb = gi;
  XChunk<float> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
  if (T > 0) {
    if (in1) {
      if constexpr (G2GIN) {                  // the tier's convention: the unit row's projection + its bias row (handed over as `bias_hid`)
        bb = reinterpret_cast<const f32x4*>(bias_hid)[hid];
        vv = gin4[hid];
      } else {
        bb = gin4[H + hid];
        vv = gin4[hid] - bb;
      }
    } else {
      gi = gin4[(b * T) * H + hid];
    }
  }
  // the exponent of the state's scale (H0 only; 0 = the constant 2^6 of |h| < 1)
  int e_cur = 0;
  if constexpr (H0) e_cur = __builtin_amdgcn_readfirstlane(f10h_h0_expo<4>(hst, gbuf, wave, lane));
  float hsc = ldexpf(F10H_HSC, -e_cur);
  f32x4 un_t = usc * ldexpf(1.f, e_cur);

  // S2 of this wave's eight chain rows from the state in the lanes
  auto s2_from_lanes = [&](float hscaled) {
    if constexpr (A2LDS) f10p_s2_lds<S>(afr, lane, ln, f10p_pack(hscaled), img);
    else f10p_s2<S>(a1, a2, ln, f10p_pack(hscaled), img);
  };
  s2_from_lanes(hst * hsc);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): keep weight-register waits out of the time loop
  lds_barrier();

  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();
  const int row10 = c < F::I2 ? c : F::I2 - 1;
  for (int t = 0; t < T; ++t) {
    // ---- B: fused S1*S0 stage, one tile per wave --------------------------------------------------------------------
    {
      f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
      f10h_s10_part<S, F::NM>(w10, img, row10, q, 0, acc_lo, acc_hi);
      const f32x4 acc = acc_hi + acc_lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) gbuf[gdst + j * F::I2] = acc[j];   // o = m*I2 + i2
    }
    TT_STAMP(0)
    lds_barrier();
    TT_STAMP(1)
    // ---- C + A: gates + state (gru.py:38-44), then S2 of the new state from the lanes -------------------------------
    const size_t bt = b * T + t;
    {
      if (in1) gi = bb + xq.at(t) * vv;
      const f32x4 un = H0 ? un_t : usc;
      const float hn = fmaf(gbuf[2 * H + hid], un[2], G2GIN ? gi[3] : bh[2]);
      const float rg = fsigmoid(gi[0] + fmaf(gbuf[hid], un[0], bh[0]));              // gru.py:38-39
      const float zg = fsigmoid(gi[1] + fmaf(gbuf[H + hid], un[1], bh[1]));          // gru.py:40-41
      const float ng = ftanh(gi[2] + rg * hn);                           // gru.py:42-43
      const float hy = (1.0f - zg) * ng + zg * hst;                      // gru.py:44
      if (reserve) *reinterpret_cast<f32x4*>(reserve + (bt * H + hid) * 4) = f32x4{rg, zg, ng, hn};
      if constexpr (OUT) out[bt * H + hid] = hy;                         // outputs[:, t, :] (gru.py:134)
      hst = hy;
      if constexpr (H0) {
        if (e_cur > 0) {                 // workgroup-uniform: the state was outside (-1, 1) — re-derive its exponent
          float mx = fabsf(hy);
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
          if (lane == 0) hmax[wave] = mx;
          lds_barrier();
          mx = fmaxf(fmaxf(hmax[0], hmax[1]), fmaxf(hmax[2], hmax[3]));
          const int e = f10h_expo(mx);
          e_cur = __builtin_amdgcn_readfirstlane(e < 0 ? 0 : e);
          hsc = ldexpf(F10H_HSC, -e_cur);
          un_t = usc * ldexpf(1.f, e_cur);
        }
      }
      TT_STAMP(2)
      s2_from_lanes(hy * hsc);
      // (behind the S2 of this step: a conditional global load makes the compiler wait for every memory operation in flight —
      // this step's stores included — at the point where the branches join)
      if (!in1 && t + 1 < T) gi = gin4[(bt + 1) * H + hid];
    }
    if (in1) xq.advance(xs, b * T, T, t, lane);
    TT_STAMP(3)
    lds_barrier();
    TT_STAMP(4)
  }
  if (hT) hT[b * H + hid] = hst;
  if constexpr (DIAG) {
    if (lane == 0 && reserve && b < 8) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(reserve) + (b * 8 + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = seg[i];
    }
  }
}

template <class S>
static int launch_gh(const RnnShape& rs, GinSrc gin, const void* h0, const float* packed_hid, const void* bias_hid, void* out,
                     void* hT, float* reserve, void* ws, hipStream_t stream, int phase) {
  if (!ws) return TTRNN_ERR_WORKSPACE;
  static_assert((F10H_EP + F10<S>::M) * sizeof(int) <= F10H_HDR_BYTES, "header");
  float* hdr = reinterpret_cast<float*>(ws);
  xh8* wfrag = reinterpret_cast<xh8*>(reinterpret_cast<unsigned char*>(ws) + F10H_HDR_BYTES);
  // the scale header and the fragments depend on the weights only: TTRNN_PHASE_RUN finds them in ws (ttrnn_rnn_forward_phase)
  if (phase != TTRNN_PHASE_RUN) {
    hipLaunchKernelGGL((k_f10h_scale<S>), dim3(F10<S>::M), dim3(256), 0, stream, packed_hid, reinterpret_cast<int*>(ws));
    hipLaunchKernelGGL((k_f10gh_prep<S>), dim3(4 * F10<S>::NM), dim3(64), 0, stream, packed_hid, hdr, wfrag);
  }
  if (phase == TTRNN_PHASE_PREPARE) return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  const float* bh = rs.has_bias_hid ? (const float*)bias_hid : (const float*)nullptr;
  auto kern = gin.in1 ? (out ? (h0 ? k_gru_fwd_f10vh<S, true, true, true> : k_gru_fwd_f10vh<S, false, true, true>)
                             : (h0 ? k_gru_fwd_f10vh<S, true, false, true> : k_gru_fwd_f10vh<S, false, false, true>))
                      : (out ? (h0 ? k_gru_fwd_f10vh<S, true, true, false> : k_gru_fwd_f10vh<S, false, true, false>)
                             : (h0 ? k_gru_fwd_f10vh<S, true, false, false> : k_gru_fwd_f10vh<S, false, false, false>));
  if (opt(OPT_DIAG) && reserve && out && !h0)      // stamped build (diagnostics)
    kern = gin.in1 ? k_gru_fwd_f10vh<S, false, true, true, true> : k_gru_fwd_f10vh<S, false, true, false, true>;
  hipLaunchKernelGGL(kern, dim3(rs.B), dim3(256), 0, stream, rs.B, rs.T, gin, (const float*)h0, packed_hid, hdr, wfrag, bh,
                     (float*)out, (float*)hT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// The same recurrent kernel behind the runtime-shape tier's K-in (input_size != 1: the tier builds gin as ONE dense two-piece GEMM
// over the B T rows; the chain kernel this file's own route would use there is 2 - 3 x slower at 81 920 rows: 1.07 against 0.72 ms for
// benchmarking.py --gru --hidden_size 256).  ws: f10gh_workspace_bytes (the tier's `rec` region)
template <class S>
static int launch_gh_g2(const RnnShape& rs, GinSrc src, const float* bilv, const void* h0, const float* packed_hid, void* out, void* hT,
                        float* reserve, void* ws, hipStream_t stream) {
  if (!ws) return TTRNN_ERR_WORKSPACE;
  float* hdr = reinterpret_cast<float*>(ws);
  xh8* wfrag = reinterpret_cast<xh8*>(reinterpret_cast<unsigned char*>(ws) + F10H_HDR_BYTES);
  hipLaunchKernelGGL((k_f10h_scale<S>), dim3(F10<S>::M), dim3(256), 0, stream, packed_hid, reinterpret_cast<int*>(ws));
  hipLaunchKernelGGL((k_f10gh_prep<S>), dim3(4 * F10<S>::NM), dim3(64), 0, stream, packed_hid, hdr, wfrag);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  // input_size == 1: the tier hands over the unit row's projection (gin) and its bias row (bilv, through the kernel's bias_hid slot)
  auto kern = src.in1 ? (out ? (h0 ? k_gru_fwd_f10vh<S, true, true, true, false, true> : k_gru_fwd_f10vh<S, false, true, true, false, true>)
                             : (h0 ? k_gru_fwd_f10vh<S, true, false, true, false, true> : k_gru_fwd_f10vh<S, false, false, true, false, true>))
                      : (out ? (h0 ? k_gru_fwd_f10vh<S, true, true, false, false, true> : k_gru_fwd_f10vh<S, false, true, false, false, true>)
                             : (h0 ? k_gru_fwd_f10vh<S, true, false, false, false, true> : k_gru_fwd_f10vh<S, false, false, false, false, true>));
  hipLaunchKernelGGL(kern, dim3(rs.B), dim3(256), 0, stream, rs.B, rs.T, src, (const float*)h0, packed_hid, hdr, wfrag,
                     src.in1 ? bilv : (const float*)nullptr, (float*)out, (float*)hT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

int launch_gru_fwd_f10gh_g2(const RnnShape& rs, GinSrc src, const float* bilv, const void* h0, const float* packed_hid, void* out,
                            void* hT, float* reserve, void* ws, hipStream_t stream) {
  if (shape_matches<ShpH256R8G>(rs.hid_s)) return launch_gh_g2<ShpH256R8G>(rs, src, bilv, h0, packed_hid, out, hT, reserve, ws, stream);
  if (shape_matches<ShpH256R16G>(rs.hid_s)) return launch_gh_g2<ShpH256R16G>(rs, src, bilv, h0, packed_hid, out, hT, reserve, ws, stream);
  return TTRNN_ERR_UNSUPPORTED;
}

// does this file's own plan (unit-row K-in + fused set-up launch, input_size == 1) exist for the shape?  (r = 8 only)
bool f10gh_own_plan(const RnnShape& rs) { return shape_matches<ShpH256R8G>(rs.hid_s); }

// fp32-storage TT-GRU, split math mode (dev bit 256: keep the runtime-shape tier's kernel, A/B)
// (r = 16: behind the tier's K-in only, either input size; this file's own plan has no unit-row chain kernel for that shape)
bool f10gh_available(const RnnShape& rs, int dtype) {
  return !opt(OPT_NO_F10) && !(opt(OPT_DEV) & 256) && rs.B >= 1 && rs.T >= 1 && dtype == TTRNN_F32 && rs.cell == TTRNN_GRU &&
         rs.hid_blocks <= 1 && opt(OPT_FP32_MATH) == TTRNN_MATH_SPLIT &&
         (shape_matches<ShpH256R8G>(rs.hid_s) || shape_matches<ShpH256R16G>(rs.hid_s));
}
size_t f10gh_workspace_bytes(const RnnShape& rs) {
  if (shape_matches<ShpH256R8G>(rs.hid_s)) return f10gh_ws_bytes<ShpH256R8G>();
  if (shape_matches<ShpH256R16G>(rs.hid_s)) return f10gh_ws_bytes<ShpH256R16G>();
  return 0;
}
int launch_gru_fwd_f10gh(const RnnShape& rs, GinSrc gin, const void* h0, const float* packed_hid, const void* bias_hid, void* out,
                         void* hT, float* reserve, void* ws, hipStream_t stream, int phase) {
  if (shape_matches<ShpH256R8G>(rs.hid_s))
    return launch_gh<ShpH256R8G>(rs, gin, h0, packed_hid, bias_hid, out, hT, reserve, ws, stream, phase);
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
