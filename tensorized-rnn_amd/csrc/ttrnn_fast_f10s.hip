// ttrnn_fast_f10s.hip — the fused-core TT-LSTM forward kernel with ONE barrier per timestep (gfx950, round 5).
//
// k_lstm_fwd_f10q (ttrnn_fast_f10q.hip) hands h_t from the gate lanes to S2 through LDS: [S2 + split | barrier | S10 + gates +
// h -> LDS | barrier].  But wave w of that kernel owns the gates of hidden units 64w .. 64w+63 — exactly chain rows 8w .. 8w+7 of
// S2 — so S2 of the NEW state needs no other wave's data.  Here every wave runs S2 of its own eight chain rows right behind its
// gate math: the state goes from the lanes into the MFMA's B operand through the LDS crossbar (four ds_bpermute of the packed fp16
// piece pairs), two m-tiles share one MFMA (ttrnn_f10_dev.h: f10p_*, every result lane live), and the split results land in the
// OTHER parity of a double-buffered S10 image.  Per step:
//     S10 (image[t & 1]) -> gates + state -> S2 of h_t -> image[(t + 1) & 1] -> barrier
// one barrier, no LDS copy of h, and the S2 MFMAs (2 x MT2 / 2 per wave instead of MT2 / 4) sit behind the gate chain of the same
// wave.  Same operand scales, same S10 arithmetic as k_lstm_fwd_f10q; S2 sums the same four terms per product in TWO chained
// MFMAs ((x0 + x1) w0, then (x0 + x1) w1) instead of one — results agree with the two-barrier kernel to an ulp of an fp32
// accumulator, not bit for bit (tests/test_gpu_parity.py::test_single_barrier_forward_kernel_*).
// Replaces the same reference code as ttrnn_fast_f10.hip: tensorized_rnn/lstm.py:23-32,123-133 + t3nsor/ops.py:78-93.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <type_traits>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_f10.h"
#include "ttrnn_f10_dev.h"

namespace ttrnn {

template <class S>
constexpr int f10s_waves() { return F10<S>::MT; }

template <class S>
constexpr size_t f10s_lds_bytes() {
  // two parities x two fp16 planes of the S10 operand
  return 2 * 2 * 2 * (size_t)F10<S>::PLANE;
}

template <class S>
constexpr bool f10s_ok() {
  using F = F10<S>;
  // wave w <-> units 64w .. 64w+63 <-> chain rows 8w .. 8w+7: sixteen columns i2, four feature rows per k-group
  return f10_ok<S>() && F::I2 == 16 && F::ROWS2 == 8 * F::MT && F::MT2 % 2 == 0;
}

// KH, H0, OUT, IN1, DIAG: as k_lstm_fwd_f10q
template <class S, int KH, bool H0, bool OUT, bool IN1, bool DIAG = false>
__global__ void __launch_bounds__(f10s_waves<S>() * 64, f10s_waves<S>() == 4 ? 2 : 1) k_lstm_fwd_f10s(int B, int T, GinSrc gs, const float* __restrict__ h0,
                                                               const float* __restrict__ c0,
                                                               const float* __restrict__ packed_hid,
                                                               const float* __restrict__ hdr,
                                                               const xh8* __restrict__ wfrag,
                                                               const float* __restrict__ bias_hid,
                                                               float* __restrict__ out, float* __restrict__ hT,
                                                               float* __restrict__ cT, float* __restrict__ reserve) {
  static_assert(f10s_ok<S>(), "shape not supported by the single-barrier fused-core kernel");
  using F = F10<S>;
  constexpr int QW = f10s_waves<S>();
  static_assert((QW == 4 || QW == 8) && F::NM % KH == 0, "one S10 tile per wave");
  constexpr int H = F::H;
  constexpr int NH = F::NM / KH;                         // k-blocks per half

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  _Float16* img = reinterpret_cast<_Float16*>(smem_s);   // [parity][2 planes][I2][K10]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const F10hScales fsc = f10h_scales<S>(hdr, wave, lane);      // diagonal power-of-two scales (ttrnn_f10_dev.h)
  const float hsc = F10H_HSC;
  const f32x4 psc = fsc.pre, usc = fsc.un;
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  xh8 a1[F10P<S>::NP], a2[F10P<S>::NP];
  f10p_load_w2<S>(a1, a2, packed_hid, lane, hdr);
  F10pLane<S> ln;
  ln.init(wave, lane);
  xh8 w10[KH][2][NH];
#pragma unroll
  for (int kh = 0; kh < KH; ++kh)
#pragma unroll
    for (int u = 0; u < NH; ++u)
#pragma unroll
      for (int p = 0; p < 2; ++p) w10[kh][p][u] = wfrag[(size_t)((wave * F::NM + kh * NH + u) * 2 + p) * 64 + lane];

  // the hidden unit of this lane: hid = (4*wave + q)*I2 + c = 64 wave + lane, gates in acc[0..3] = i,f,g,o; gin slots i,g,f,o
  const float* __restrict__ gin = gs.gin;
  const float* __restrict__ xs = reinterpret_cast<const float*>(gs.x);
  constexpr bool in1 = IN1;
  const int hd = 64 * wave + lane;
  float hst = H0 ? h0[b * H + hd] : 0.f;
  float cst = c0 ? c0[b * H + hd] : 0.f;
  float h0sc = 1.0f, h0un = 1.0f;
  if constexpr (H0) {
    const int e0 = f10h_h0_expo<QW>(hst, reinterpret_cast<float*>(img), wave, lane);
    h0sc = ldexpf(1.f, -e0); h0un = ldexpf(1.f, e0);
  }
  f32x4 bh = f32x4{0.f, 0.f, 0.f, 0.f}, gi = bh, vv = bh, bb = bh;       // slot order i,g,f,o
  const f32x4 gsc = f32x4{-1.4426950408889634f, 2.8853900817779268f, -1.4426950408889634f, -1.4426950408889634f} *
                    f32x4{psc[0], psc[2], psc[1], psc[3]};      // slots i,g,f,o <- accumulator rows i,f,g,o
  XChunk<float> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
  if (bias_hid) bh = f32x4{bias_hid[hd], bias_hid[2 * H + hd], bias_hid[H + hd], bias_hid[3 * H + hd]};
  if (T > 0) {
    if (in1) {
      bb = *reinterpret_cast<const f32x4*>(gin + (H + hd) * 4);
      vv = (*reinterpret_cast<const f32x4*>(gin + hd * 4) - bb) * gsc;
      bb = (bb + bh) * gsc;
    } else {
      gi = *reinterpret_cast<const f32x4*>(gin + ((b * T) * H + hd) * 4);
    }
  }
  f10p_s2<S>(a1, a2, ln, f10p_pack(hst * (hsc * h0sc)), img);            // S2 of h_{-1} -> parity 0
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) here, so that no weight-register wait lands inside the loop
  lds_barrier();

  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();
  f32x4 us_t = usc * h0un;                  // step 0 runs on 2^-e0 h_0 (f10h_h0_expo); reset to usc / 1 at the end of it
  float ps_t = h0sc;
  // one timestep; PAR = t & 1 as a compile-time constant (the loop below is unrolled by two): with a run-time parity every one
  // of S10's sixteen fragment reads and S2's stores pays an address addition per step — measured +140 cycles on S10 alone
  auto step = [&](const int t, auto parc) {
    constexpr int PAR = decltype(parc)::value;
    const _Float16* ic = img + PAR * 2 * F::PLANE;               // S2 of h_{t-1}
    _Float16* in_ = img + (1 - PAR) * 2 * F::PLANE;              // S2 of h_t
    const size_t bt = b * T + t;
    // ---- the fused S1*S0 stage, then gates + state (lstm.py:26-32) -----------------------------------------------
    f32x4 acc;
    {
      // (W_in x_t + b_in + b_hid) * scale, slots i,g,f,o -> accumulator rows i,f,g,o
      f32x4 pre = in1 ? bb + xq.at(t) * vv : (gi + bh) * gsc;
      if constexpr (H0) pre = pre * ps_t;
      f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = f32x4{pre[0], pre[2], pre[1], pre[3]};
      f10h_s10_part<S, NH>(w10[0], ic, c, q, 0, acc_lo, acc_hi);
      const f32x4 un = H0 ? us_t : usc;
      acc = acc_hi * un + acc_lo * un;                    // 2^-S per row and column (2^(e0-S) at step 0 of a given h_0), exact
      if constexpr (KH == 2) {
        f32x4 bl = f32x4{0.f, 0.f, 0.f, 0.f}, bhh = bl;
        f10h_s10_part<S, NH>(w10[1], ic, c, q, NH, bl, bhh);
        acc += bhh * un + bl * un;
      }
      if constexpr (DIAG) asm volatile("" : "+v"(acc));
    }
    TT_STAMP(0)
    const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0]));                // lstm.py:26
    const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[1]));                // lstm.py:27
    const float gg = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[2]));  // lstm.py:28
    const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[3]));                // lstm.py:29
    const float cy = fg * cst + ig * gg;                    // lstm.py:31
    const float hy = og * ftanh(cy);                        // lstm.py:32
    cst = cy;
    hst = hy;
    TT_STAMP(1)
    // ---- S2 of the new state, inside the wave ------------------------------------------------------------------------
    f10p_s2<S>(a1, a2, ln, f10p_pack(hy * hsc), in_);
    TT_STAMP(2)
    if constexpr (OUT) out[bt * H + hd] = hy;               // outputs[:, t, :] (lstm.py:133): 256 contiguous bytes per wave
    if (reserve) {
      *reinterpret_cast<f32x4*>(reserve + res_gate(bt, H, hd)) = f32x4{ig, gg, fg, og};
      reserve[res_cell((size_t)B * T, bt, H, hd)] = cy;
    }
    if (!in1 && t + 1 < T) gi = *reinterpret_cast<const f32x4*>(gin + ((bt + 1) * H + hd) * 4);
    if (in1) xq.advance(xs, b * T, T, t, lane);
    us_t = usc; ps_t = 1.0f;
    TT_STAMP(3)
    lds_barrier();
    TT_STAMP(4)
  };
  int t = 0;
  for (; t + 1 < T; t += 2) {
    step(t, std::integral_constant<int, 0>{});
    step(t + 1, std::integral_constant<int, 1>{});
  }
  if (t < T) step(t, std::integral_constant<int, 0>{});
  if (hT) hT[b * H + hd] = hst;
  if (cT) cT[b * H + hd] = cst;
  if constexpr (DIAG) {
    if (lane == 0 && reserve && b < 8) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(reserve) + (b * 8 + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = seg[i];
    }
  }
}

template <class S, int KH>
static int launch_s(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid, const void* ws,
                    const float* bh, void* out, void* hT, void* cT, float* reserve, hipStream_t stream) {
  const float* hdr = reinterpret_cast<const float*>(ws);
  const xh8* wfrag = reinterpret_cast<const xh8*>(reinterpret_cast<const unsigned char*>(ws) + F10H_HDR_BYTES);
  constexpr size_t lds = f10s_lds_bytes<S>();
  constexpr int QW = f10s_waves<S>();
  static_assert((QW == 4 ? 2 : 1) * lds <= 160 * 1024, "workgroups per CU");
  constexpr bool HV = true;
  auto kern = gin.in1 ? (out ? (h0 ? k_lstm_fwd_f10s<S, KH, HV, true, true> : k_lstm_fwd_f10s<S, KH, false, true, true>)
                             : (h0 ? k_lstm_fwd_f10s<S, KH, HV, false, true> : k_lstm_fwd_f10s<S, KH, false, false, true>))
                      : (out ? (h0 ? k_lstm_fwd_f10s<S, KH, HV, true, false> : k_lstm_fwd_f10s<S, KH, false, true, false>)
                             : (h0 ? k_lstm_fwd_f10s<S, KH, HV, false, false> : k_lstm_fwd_f10s<S, KH, false, false, false>));

  if (opt(OPT_DIAG) && reserve && out && !h0)      // stamped build (diagnostics)
    kern = gin.in1 ? k_lstm_fwd_f10s<S, KH, false, true, true, true> : k_lstm_fwd_f10s<S, KH, false, true, false, true>;
  if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(rs.B), dim3(QW * 64), lds, stream, rs.B, rs.T, gin, (const float*)h0,
                     (const float*)c0, packed_hid, hdr, wfrag, bh, (float*)out, (float*)hT, (float*)cT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// MEASURED SLOWER than the two-barrier kernel and therefore NOT the default (option dev bit 512 selects it, A/B): cfg2 0.495 against
// 0.478 ms (profiles/r5/stamps_cfg2_f10s.txt against stamps_cfg2_f10q.txt: the barrier it removes costs 88 + 111 -> 98 cycles, but
// the gate wave's own chain grows — S10 673 -> 719, gates + S2 + stores 987 -> 1 076: eight chained S2 MFMAs, four crossbar gathers
// and the splitting passes now sit behind the gate math of the SAME wave instead of being spread over the step's other half); the
// eight-wave H = 512 instantiation spilled and lost 8 % (1.22 against 1.13 ms) and is not built.  DESIGN.md lesson 56.
bool f10s_available(const RnnShape& rs, bool with_h0) {
  (void)with_h0;
  if (!(opt(OPT_DEV) & 512)) return false;
  return shape_matches<ShpH256R8L>(rs.hid_s);
}

// ws: scale header + the fragments k_f10h_scale / k_f10h_prep built for this launch (ttrnn_fast_f10.hip)
int launch_rnn_fwd_f10_s(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                         const void* ws, const float* bias_hid, void* out, void* hT, void* cT, float* reserve,
                         hipStream_t stream) {
  if (shape_matches<ShpH256R8L>(rs.hid_s))
    return launch_s<ShpH256R8L, 1>(rs, gin, h0, c0, packed_hid, ws, bias_hid, out, hT, cT, reserve, stream);
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
