// ttrnn_fast_f10x.hip — the fused-core TT-LSTM recurrent kernel on the fp32 MFMA (TTRNN_MATH_EXACT, gfx950).
//
// Same contraction order as ttrnn_fast_f10.hip — cores 1 and 0 contracted once per launch, per step S2 then one fused
// stage S10 with the gates on its accumulators — but every product is an fp32 x fp32 product on
// v_mfma_f32_16x16x4_f32: no operand splitting, fp32 LDS images.  Compared with the stage-wise exact kernel
// (ttrnn_fast.hip: S2, S1, S0 + gates, three barriers) the fused stage needs 256 instead of 256 + 64 MFMAs per step
// and one LDS round trip and one barrier less; the step stays bound by the fp32 matrix pipe (cfg2: 2 048 + 256
// cycles per SIMD).
//
// Per timestep:  phase A  all 8 waves   S2 (4 fp32 MFMAs per wave) -> fp32 image [I2][K10] (XOR-swizzled, a_off)
//                barrier
//                phase B  waves 0-3     S10: 16-byte fragment reads three k-groups ahead, K10/4 MFMAs per wave, gates,
//                                       h_t -> LDS; wave 7 streams h_{t-1} to `out`
//                barrier
// k order of the fused stage: MFMA step (u, j) contracts image position 16u + 4q + j on lane group q, so that one
// ds_read_b128 per lane feeds four consecutive MFMAs; image position p holds (row2, r2) = ((p/4) % ROWS2,
// 4*(p / (4*ROWS2)) + p%4) — the 16-byte slots of one r2-quad are contiguous over row2, so the 8 lanes of a
// ds_write_b128 group (8 consecutive chain rows) fill 128 contiguous bytes (the natural order (row2, r2) left every
// other slot out: 2-way bank conflicts on every S2 store).  k_f10x_prep lays the core fragments out the same way.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_f10.h"

namespace ttrnn {

// wfrag[(t*K/4 + 4u + j)*64 + lane] = W10[k = 16u + 4q + j][m(r)],  m(r) = MPG*(r&3) + 4t + (r>>2)  (lane = (r, q))
template <class S>
__global__ void __launch_bounds__(64) k_f10x_prep(const float* __restrict__ packed, float* __restrict__ wfrag) {
  using F = F10<S>;
  const int lane = threadIdx.x, st = blockIdx.x % (F::K / 4), t = blockIdx.x / (F::K / 4);
  const int r = lane & 15, q = lane >> 4;
  const int u = st >> 2, j = st & 3;
  const int p = 16 * u + 4 * q + j;                        // image position (see the header)
  const int row2 = (p >> 2) % F::ROWS2, r2 = 4 * (p / (4 * F::ROWS2)) + (p & 3);
  const int j1 = row2 % F::J1, j0 = row2 / F::J1;
  const int m = F::MPG * (r & 3) + 4 * t + (r >> 2);
  const int i0 = m / F::I1, i1 = m % F::I1;
  const float* W0 = packed + woff_of<S>(0);
  const float* w1p = packed + woff_of<S>(1) + (j1 * F::R2 + r2) * (F::I1 * F::R1) + i1 * F::R1;
  float v = 0.f;
  for (int r1 = 0; r1 < F::R1; ++r1) v = fmaf(W0[(j0 * F::R1 + r1) * F::I0 + i0], w1p[r1], v);
  wfrag[(size_t)blockIdx.x * 64 + lane] = v;
}

template <class S>
constexpr size_t f10x_lds_bytes() { return sizeof(float) * (2 * F10<S>::H + F10<S>::I2 * F10<S>::K); }

// IN1: input_size == 1 as a template parameter (round 4; see k_lstm_fwd_f10q)
template <class S, bool IN1>
__global__ void __launch_bounds__(FAST_NT) k_lstm_fwd_f10x(int B, int T, GinSrc gs, const float* __restrict__ h0,
                                                           const float* __restrict__ c0,
                                                           const float* __restrict__ packed_hid,
                                                           const float* __restrict__ wfrag,
                                                           const float* __restrict__ bias_hid, float* __restrict__ out,
                                                           float* __restrict__ hT, float* __restrict__ cT,
                                                           float* __restrict__ reserve) {
  static_assert(f10_ok<S>() && !St<S, 2>::SPLIT && St<S, 2>::MT % FAST_NW == 0, "shape not supported");
  using F = F10<S>;
  using T2 = St<S, 2>;
  constexpr int H = F::H, K = F::K, NST = K / 4, NG = K / 16;      // MFMA k-steps, 16-byte fragment groups

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* hbuf = reinterpret_cast<float*>(smem);            // h, two parities, [ROWS2][J2] = flat hidden index
  float* img = hbuf + 2 * H;                               // S10 operand, fp32 [I2][K] (a_off swizzle)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;
  const bool gate_wave = wave < F::MT;

  float w2[nwreg<S, 2>()];
  load_wfrag<S, 2>(w2, packed_hid, wave, lane);
  float w10[NST];
#pragma unroll
  for (int s = 0; s < NST; ++s) w10[s] = gate_wave ? wfrag[(size_t)(wave * NST + s) * 64 + lane] : 0.f;

  const float* __restrict__ gin = gs.gin;
  const float* __restrict__ xs = reinterpret_cast<const float*>(gs.x);
  constexpr bool in1 = IN1;
  const bool ok = gate_wave && c < F::I2;
  const int hd = ok ? (4 * wave + q) * F::I2 + c : 0;
  float hst = (ok && h0) ? h0[b * H + hd] : 0.f;
  float cst = (ok && c0) ? c0[b * H + hd] : 0.f;
  f32x4 bh = f32x4{0.f, 0.f, 0.f, 0.f}, gi = bh, vv = bh, bb = bh;       // slot order i,g,f,o
  XChunk<float> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
  if (ok) {
    if (bias_hid) bh = f32x4{bias_hid[hd], bias_hid[2 * H + hd], bias_hid[H + hd], bias_hid[3 * H + hd]};
    if (T > 0) {
      if (in1) {
        bb = *reinterpret_cast<const f32x4*>(gin + (H + hd) * 4);
        vv = *reinterpret_cast<const f32x4*>(gin + hd * 4) - bb;
      } else {
        gi = *reinterpret_cast<const f32x4*>(gin + ((b * T) * H + hd) * 4);
      }
    }
    hbuf[hd] = hst;                                        // parity 0 = h_{-1}
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) here, so that no weight-register wait lands inside the loop
  lds_barrier();

  const int row10 = c < F::I2 ? c : F::I2 - 1;
  for (int t = 0; t < T; ++t) {
    const float* hp = hbuf + (t & 1) * H;
    // ---- phase A: S2 on the fp32 MFMA, all waves -------------------------------------------------------------
    {
      f32x4 acc[T2::XM][T2::YR];
      stage_mma<S, 2>(w2, hp, acc, wave, lane);
#pragma unroll
      for (int xm = 0; xm < T2::XM; ++xm)
#pragma unroll
        for (int y = 0; y < T2::YR; ++y) {
          const int row2 = 16 * y + c, m0 = 16 * (wave + FAST_NW * xm) + 4 * q;     // feature m0 = (i2, r2 .. r2+3)
          const int r20 = m0 % F::R2;                        // 4 consecutive r2 = one slot: quad r20/4 of chain row row2
          *reinterpret_cast<f32x4*>(img + a_off<K>(m0 / F::R2, ((r20 >> 2) * F::ROWS2 + row2) * 4)) = acc[xm][y];
        }
    }
    lds_barrier();
    const size_t bt = b * T + t;
    if (gate_wave) {
      // ---- phase B: fused S1*S0 stage on the fp32 MFMA, then gates + state (lstm.py:26-32) ----------------------
      constexpr int PD = 3;
      f32x4 af[NG];
#pragma unroll
      for (int u = 0; u < PD; ++u) af[u] = *reinterpret_cast<const f32x4*>(img + a_off<K>(row10, 16 * u + 4 * q));
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int u = 0; u < NG; ++u) {
        if (u + PD < NG) af[u + PD] = *reinterpret_cast<const f32x4*>(img + a_off<K>(row10, 16 * (u + PD) + 4 * q));
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w10[4 * u + 0], af[u][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w10[4 * u + 1], af[u][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w10[4 * u + 2], af[u][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w10[4 * u + 3], af[u][3], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const f32x4 acc = acc0 + acc1;
      if (in1) gi = bb + xq.at(t) * vv;      // W_in x_t + b_in from the two unit rows (GinSrc)
      const float ig = fsigmoid(acc[0] + gi[0] + bh[0]);      // lstm.py:26
      const float fg = fsigmoid(acc[1] + gi[2] + bh[2]);      // lstm.py:27
      const float gg = ftanh(acc[2] + gi[1] + bh[1]);         // lstm.py:28
      const float og = fsigmoid(acc[3] + gi[3] + bh[3]);      // lstm.py:29
      const float cy = fg * cst + ig * gg;                    // lstm.py:31
      const float hy = og * ftanh(cy);                        // lstm.py:32
      if (ok) {
        cst = cy;
        hst = hy;
        hbuf[((t + 1) & 1) * H + hd] = hy;
        if (reserve) {
          *reinterpret_cast<f32x4*>(reserve + res_gate(bt, H, hd)) = f32x4{ig, gg, fg, og};
          reserve[res_cell((size_t)B * T, bt, H, hd)] = cy;
        }
        if (!in1 && t + 1 < T) gi = *reinterpret_cast<const f32x4*>(gin + ((bt + 1) * H + hd) * 4);
      }
      if (in1) xq.advance(xs, b * T, T, t, lane);
    } else if (wave == FAST_NW - 1 && t > 0 && out) {      // out == NULL: final state only
      // outputs[:, t-1, :] = h_{t-1} (lstm.py:133)
#pragma unroll
      for (int h4 = lane; h4 < H / 4; h4 += 64)
        *reinterpret_cast<f32x4*>(out + (bt - 1) * H + 4 * h4) = *reinterpret_cast<const f32x4*>(hp + 4 * h4);
    }
    lds_barrier();
  }
  if (T > 0 && wave == FAST_NW - 1 && out) {
    const float* hlast = hbuf + (T & 1) * H;
#pragma unroll
    for (int h4 = lane; h4 < H / 4; h4 += 64)
      *reinterpret_cast<f32x4*>(out + (b * T + T - 1) * H + 4 * h4) = *reinterpret_cast<const f32x4*>(hlast + 4 * h4);
  }
  if (ok) {
    if (hT) hT[b * H + hd] = hst;
    if (cT) cT[b * H + hd] = cst;
  }
}

// ---- dispatch ------------------------------------------------------------------------------------------
template <class S>
static int launch_f10x(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                       const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws,
                       hipStream_t stream) {
  if (!ws) return TTRNN_ERR_WORKSPACE;
  using F = F10<S>;
  float* wfrag = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL((k_f10x_prep<S>), dim3(F::MT * (F::K / 4)), dim3(64), 0, stream, packed_hid, wfrag);
  constexpr size_t lds = f10x_lds_bytes<S>();
  static_assert(lds <= 64 * 1024, "raise the dynamic LDS limit for this shape");
  const float* bh = rs.has_bias_hid ? (const float*)bias_hid : (const float*)nullptr;
  auto kern = gin.in1 ? k_lstm_fwd_f10x<S, true> : k_lstm_fwd_f10x<S, false>;
  hipLaunchKernelGGL(kern, dim3(rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T, gin, (const float*)h0,
                     (const float*)c0, packed_hid, wfrag, bh, (float*)out, (float*)hT, (float*)cT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// fragments: MT * K/4 * 64 floats; never larger than the split-mode fragment set the workspace query reserves
bool f10x_rnn_fwd_available(const RnnShape& rs, int dtype) {
  if (opt(OPT_NO_F10) || dtype != TTRNN_F32 || rs.cell != TTRNN_LSTM || rs.B < 1 || rs.T < 1) return false;
  return shape_matches<ShpH256R8L>(rs.hid_s);
}

int launch_rnn_fwd_f10x(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                        const void* bias_hid, void* out, void* hT, void* cT, float* reserve, void* ws,
                        hipStream_t stream) {
  if (shape_matches<ShpH256R8L>(rs.hid_s))
    return launch_f10x<ShpH256R8L>(rs, gin, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, ws, stream);
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
