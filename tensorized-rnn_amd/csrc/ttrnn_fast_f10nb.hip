// ttrnn_fast_f10nb.hip — the fused-core TT-LSTM forward kernel with TWO samples per workgroup (gfx950); a translation unit
// of its own: next to it in ttrnn_fast_f10.hip the one-sample kernel compiled to different (1.4 % slower) code.
// Replaces the same reference code as ttrnn_fast_f10.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"
#include "ttrnn_f10.h"
#include "ttrnn_f10_dev.h"

namespace ttrnn {

// k_lstm_fwd_f10_nb: the workgroup carries NB = 2 samples through every phase (own LDS images and states, the SAME resident core
// fragments).  The kernel needs more than half of the register file, so one workgroup owns a CU; with more samples than
// CUs the workgroups of one CU ran one after the other, each paying the per-step barriers, LDS round trips and
// transcendental chains alone — two samples per step share them.
template <class S, int KS, int NB>
__global__ void __launch_bounds__(FAST_NT) k_lstm_fwd_f10_nb(int B, int T, GinSrc gs, const float* __restrict__ h0,
                                                          const float* __restrict__ c0,
                                                          const float* __restrict__ packed_hid,
                                                          const float* __restrict__ hdr,
                                                          const xh8* __restrict__ wfrag,
                                                          const float* __restrict__ bias_hid, float* __restrict__ out,
                                                          float* __restrict__ hT, float* __restrict__ cT,
                                                          float* __restrict__ reserve) {
  static_assert(f10_ok<S>(), "shape not supported by the fused-core kernel");
  static_assert(NB == 2, "two samples per workgroup (one: k_lstm_fwd_f10, whose LDS offsets are compile-time constants)");
  constexpr bool DIAG = false;
  using F = F10<S>;
  constexpr int H = F::H;
  constexpr size_t SMP = f10h_lds_bytes<S, KS>();                           // LDS bytes of one sample

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_nb[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  static_assert(KS == 1 || (KS == 2 && F::MT == 4 && F::NM % 2 == 0), "k-split layout");
  constexpr int NU = F::NM / KS;                         // k-blocks per MFMA wave
  const bool gate_wave = wave < F::MT;
  const bool mma_wave = KS == 2 || gate_wave;
  const int tile = KS == 2 ? (wave & 3) : wave;          // S10 feature tile of this wave
  const int u0 = KS == 2 ? (wave >> 2) * NU : 0;         // its first k-block
  const F10hScales fsc = f10h_scales<S>(hdr, tile & (F::MT - 1), lane);     // diagonal power-of-two scales (ttrnn_f10_dev.h)
  const float hsc = F10H_HSC;
  const f32x4 psc = fsc.pre, usc = fsc.un;

  // S2 fragments of the m-tiles {wave + 8x}
  xh8 s1[F::XA];
#pragma unroll
  for (int x = 0; x < F::XA; ++x) f10h_load_w2<S>(s1[x], packed_hid, wave + FAST_NW * x, lane, hdr);
  xh8 w10[2][NU];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) w10[p][u][e] = (_Float16)0.f;
  if (mma_wave) {
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int p = 0; p < 2; ++p) w10[p][u] = wfrag[(size_t)((tile * F::NM + u0 + u) * 2 + p) * 64 + lane];
  }

  // the hidden unit of this lane in phase B (waves 0 .. MT-1): hid = (4*wave + q)*I2 + c, gates in acc[0..3] = i,f,g,o.
  // gin is gate-interleaved [B][T][H][4] with slots i,g,f,o.
  const float* __restrict__ gin = gs.gin;
  const float* __restrict__ xs = reinterpret_cast<const float*>(gs.x);
  const bool in1 = gs.in1 != 0;
  const bool ok = gate_wave && c < F::I2;
  const int hd = ok ? (4 * wave + q) * F::I2 + c : 0;
  f32x4 bh = f32x4{0.f, 0.f, 0.f, 0.f}, vv = bh, bb = bh;       // slot order i,g,f,o
  // pre-scaled accumulators, exactly as in k_lstm_fwd_f10 (same arithmetic per sample: the two kernels agree bit for bit)
  const f32x4 gsc = f32x4{-1.4426950408889634f, 2.8853900817779268f, -1.4426950408889634f, -1.4426950408889634f} *
                    f32x4{psc[0], psc[2], psc[1], psc[3]};      // slots i,g,f,o <- accumulator rows i,f,g,o
  if (ok) {
    if (bias_hid) bh = f32x4{bias_hid[hd], bias_hid[2 * H + hd], bias_hid[H + hd], bias_hid[3 * H + hd]};
    if (T > 0 && in1) {
      bb = *reinterpret_cast<const f32x4*>(gin + (H + hd) * 4);
      vv = (*reinterpret_cast<const f32x4*>(gin + hd * 4) - bb) * gsc;
      bb = (bb + bh) * gsc;
    }
  }
  // per-sample state; a workgroup whose second sample lies past the batch carries a copy of the last one (never stored)
  size_t bs[NB];
  bool live[NB];
  float hst[NB], cst[NB], h0sc[NB], h0un[NB];       // h0sc / h0un: per-sample 2^-e0 / 2^e0 (f10h_h0_expo)
  f32x4 gi[NB];
  XChunk<float> xq[NB];
  float* hbuf[NB];
  _Float16 *hpl[NB], *img[NB];
  f32x4* xbuf[NB];
#pragma unroll
  for (int sm = 0; sm < NB; ++sm) {
    const size_t bb0 = (size_t)blockIdx.x * NB + sm;
    live[sm] = bb0 < (size_t)B;
    bs[sm] = live[sm] ? bb0 : (size_t)B - 1;
    unsigned char* base = smem_nb + sm * SMP;
    hbuf[sm] = reinterpret_cast<float*>(base);                                 // fp32 h, two parities (output store)
    hpl[sm] = reinterpret_cast<_Float16*>(base + 2 * sizeof(float) * H);       // fp16 pieces of 2^sH h: [parity][2][H]
    img[sm] = hpl[sm] + 2 * 2 * H;                                             // two fp16 planes [I2][K10]
    xbuf[sm] = reinterpret_cast<f32x4*>(img[sm] + 2 * F::PLANE);                // KS == 2: partial accumulators
    hst[sm] = (ok && h0) ? h0[bs[sm] * H + hd] : 0.f;
    cst[sm] = (ok && c0) ? c0[bs[sm] * H + hd] : 0.f;
    {
      const int e0 = h0 ? f10h_h0_expo<FAST_NW>(hst[sm], reinterpret_cast<float*>(img[sm]), wave, lane) : 0;   // uniform branch
      h0sc[sm] = ldexpf(1.f, -e0); h0un[sm] = ldexpf(1.f, e0);
    }
    gi[sm] = f32x4{0.f, 0.f, 0.f, 0.f};
    xq[sm].cur = 0.f; xq[sm].nxt = 0.f;
    if (in1) xq[sm].init(xs, bs[sm] * T, T, lane);
    if (ok) {
      if (T > 0 && !in1) gi[sm] = *reinterpret_cast<const f32x4*>(gin + ((bs[sm] * T) * H + hd) * 4);
      _Float16 p0, p1;                                       // parity 0 = h_{-1}
      split2h(hst[sm] * (hsc * h0sc[sm]), p0, p1);
      hpl[sm][hd] = p0; hpl[sm][H + hd] = p1;
      hbuf[sm][hd] = hst[sm];
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) here, so that no weight-register wait lands inside the loop
  lds_barrier();
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();

  const int row10 = c < F::I2 ? c : F::I2 - 1;
  for (int t = 0; t < T; ++t) {
    // ---- phase A: S2, all waves ---------------------------------------------------------------------------
    // all MFMAs first, then the splitting: the VALU work of one tile runs in the shadow of the others' MFMA latency
    // instead of behind an s_nop after every pair
#pragma unroll
    for (int sm = 0; sm < NB; ++sm) {
      const _Float16* hp = hpl[sm] + (t & 1) * 2 * H;     // pieces of h_{t-1}
      f32x4 t2[F::XA][2];
#pragma unroll
      for (int x = 0; x < F::XA; ++x) {
        t2[x][0] = f10h_s2_mma<S>(s1[x], hp, 0, lane);
        t2[x][1] = f10h_s2_mma<S>(s1[x], hp, 1, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < F::XA; ++x) {
        f10h_s2_store<S>(t2[x][0], img[sm], wave + FAST_NW * x, 0, lane);
        f10h_s2_store<S>(t2[x][1], img[sm], wave + FAST_NW * x, 1, lane);
      }
    }
    TT_STAMP(0)
    lds_barrier();
    TT_STAMP(1)
    // ---- phase B: the fused S1*S0 stage, then gates + state (lstm.py:26-32) -----------------------------------
    f32x4 acc[NB];
#pragma unroll
    for (int sm = 0; sm < NB; ++sm) {
      acc[sm] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (mma_wave) {
        f32x4 acc_lo = f32x4{0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
        if (gate_wave) {
          const f32x4 pre = (in1 ? bb + xq[sm].at(t) * vv : (gi[sm] + bh) * gsc) * (t == 0 ? h0sc[sm] : 1.0f);   // step 0: 2^-e0 h_0
          acc_hi = f32x4{pre[0], pre[2], pre[1], pre[3]};
        }
        f10h_s10_part<S, NU>(w10, img[sm], row10, q, u0, acc_lo, acc_hi);
        const f32x4 us_t = t == 0 ? usc * h0un[sm] : usc;
        acc[sm] = acc_hi * us_t + acc_lo * us_t;            // 2^-S (2^(e0-S) at step 0), exact
        if constexpr (DIAG) {
          asm volatile("" : "+v"(acc[sm]));
        }
      }
    }
    TT_STAMP(2)
    if constexpr (KS == 2) {
      if (!gate_wave) {
#pragma unroll
        for (int sm = 0; sm < NB; ++sm) xbuf[sm][tile * 64 + lane] = acc[sm];
      }
      lds_barrier();
      if (gate_wave) {
#pragma unroll
        for (int sm = 0; sm < NB; ++sm) acc[sm] += xbuf[sm][tile * 64 + lane];
      }
    }
    if (gate_wave) {
#pragma unroll
      for (int sm = 0; sm < NB; ++sm) {
        const size_t bt = bs[sm] * T + t;
        _Float16* hn = hpl[sm] + ((t + 1) & 1) * 2 * H;     // pieces of h_t
        const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[sm][0]));                // lstm.py:26
        const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[sm][1]));                // lstm.py:27
        const float gg = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[sm][2]));  // lstm.py:28
        const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[sm][3]));                // lstm.py:29
        const float cy = fg * cst[sm] + ig * gg;                        // lstm.py:31
        const float hy = og * ftanh(cy);                                // lstm.py:32
        if (ok) {
          cst[sm] = cy;
          hst[sm] = hy;
          _Float16 p0, p1;
          split2h(hy * hsc, p0, p1);
          hn[hd] = p0; hn[H + hd] = p1;
          hbuf[sm][((t + 1) & 1) * H + hd] = hy;
          if (reserve && live[sm]) {
            *reinterpret_cast<f32x4*>(reserve + res_gate(bt, H, hd)) = f32x4{ig, gg, fg, og};
            reserve[res_cell((size_t)B * T, bt, H, hd)] = cy;
          }
          if (!in1 && t + 1 < T) gi[sm] = *reinterpret_cast<const f32x4*>(gin + ((bt + 1) * H + hd) * 4);
        }
        if (in1) xq[sm].advance(xs, bs[sm] * T, T, t, lane);
      }
      TT_STAMP(3)
    } else if (wave == FAST_NW - 1 && t > 0 && out) {      // out == NULL: final state only
      // outputs[:, t-1, :] = h_{t-1} (lstm.py:133): an idle wave streams the complete vector out, 16 bytes per lane
#pragma unroll
      for (int sm = 0; sm < NB; ++sm) {
        if (!live[sm]) continue;
        const float* hprev = hbuf[sm] + (t & 1) * H;
#pragma unroll
        for (int h4 = lane; h4 < H / 4; h4 += 64)
          *reinterpret_cast<f32x4*>(out + (bs[sm] * T + t - 1) * H + 4 * h4) =
              *reinterpret_cast<const f32x4*>(hprev + 4 * h4);
      }
    }
    lds_barrier();
    TT_STAMP(4)
  }
#pragma unroll
  for (int sm = 0; sm < NB; ++sm) {
    if (!live[sm]) continue;
    if (T > 0 && wave == FAST_NW - 1 && out) {
      const float* hlast = hbuf[sm] + (T & 1) * H;
#pragma unroll
      for (int h4 = lane; h4 < H / 4; h4 += 64)
        *reinterpret_cast<f32x4*>(out + (bs[sm] * T + T - 1) * H + 4 * h4) = *reinterpret_cast<const f32x4*>(hlast + 4 * h4);
    }
    if (ok) {
      if (hT) hT[bs[sm] * H + hd] = hst[sm];
      if (cT) cT[bs[sm] * H + hd] = cst[sm];
    }
  }
  if constexpr (DIAG) {
    const size_t b = blockIdx.x;
    if (lane == 0 && reserve && b < 8) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(reserve) + (b * FAST_NW + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = seg[i];
    }
  }
}

template <class S, int KS>
static int launch_nb2(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                      const void* ws, const float* bh, void* out, void* hT, void* cT, float* reserve,
                      hipStream_t stream) {
  const float* hdr = reinterpret_cast<const float*>(ws);
  const xh8* wfrag = reinterpret_cast<const xh8*>(reinterpret_cast<const unsigned char*>(ws) + F10H_HDR_BYTES);
  constexpr size_t lds = 2 * f10h_lds_bytes<S, KS>();
  static_assert(lds <= 160 * 1024, "two samples must fit the LDS");
  {
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_lstm_fwd_f10_nb<S, KS, 2>), lds) != TTRNN_OK)
      return TTRNN_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((k_lstm_fwd_f10_nb<S, KS, 2>), dim3((rs.B + 1) / 2), dim3(FAST_NT), lds, stream, rs.B, rs.T, gin,
                     (const float*)h0, (const float*)c0, packed_hid, hdr, wfrag, bh, (float*)out, (float*)hT,
                     (float*)cT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// wfrag: scale header + the fragments k_f10h_scale / k_f10h_prep built for this launch (ttrnn_fast_f10.hip)
int launch_rnn_fwd_f10_nb2(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                           const void* wfrag, const float* bias_hid, void* out, void* hT, void* cT, float* reserve,
                           hipStream_t stream) {
  if (shape_matches<ShpH256R8L>(rs.hid_s))
    return launch_nb2<ShpH256R8L, 1>(rs, gin, h0, c0, packed_hid, wfrag, bias_hid, out, hT, cT, reserve, stream);
  if (shape_matches<ShpH256R16L>(rs.hid_s))
    return launch_nb2<ShpH256R16L, 2>(rs, gin, h0, c0, packed_hid, wfrag, bias_hid, out, hT, cT, reserve, stream);
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
