// ttrnn_fast_bf16.hip — persistent recurrent kernel for bf16 storage on the bf16 MFMA (gfx950).
//
// Same structure as ttrnn_fast.hip (one 8-wave workgroup per sample, chain stages through LDS, core fragments
// resident in VGPRs, hoisted input projection prefetched across raw barriers) with the arithmetic of the
// "bf16 storage / fp32 accumulate and state" configuration (BASELINE cfg3):
//   * chain stages run on v_mfma_f32_16x16x32_bf16 (8 k-elements per lane and instruction, fp32 accumulators):
//     16x the per-instruction work of the fp32 MFMA, so the chain's MFMA time all but disappears and the step is
//     bounded by its four LDS/barrier phases;
//   * LDS images between stages hold bf16 (half the LDS traffic), in 16-byte slots of 8 elements with the same
//     XOR swizzle; accumulators are converted once per tile (v_cvt_pk_bf16_f32) and stored with one ds_write_b64;
//   * gate pre-activations, gate math, c and h stay fp32; h is rounded to bf16 exactly once, when it is stored
//     to `out` and to the image the next step reads (the reference has no bf16 path; tolerance 2e-2 vs fp32).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"

namespace ttrnn {

typedef __bf16 hbf8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf4 __attribute__((ext_vector_type(4)));

template <class S, int k>
struct Sh {
  using F = St<S, k>;
  static constexpr int K = F::K, M = F::M, R = F::R, ROWS = F::ROWS;
  static constexpr int KI = (K + 7) & ~7;         // image row length in elements (16-byte slots of 8)
  static constexpr int NSL = KI / 8;              // slots per row
  static constexpr int NM = (K + 31) / 32;        // MFMA instructions per tile (32 k-elements each)
  static constexpr int MT = F::MT, RT = F::RT;
  static constexpr bool SPLIT = F::SPLIT;
  static constexpr int G = F::G, XM = F::XM, YR = F::YR;
  static constexpr int NWREG = XM * NM;           // resident fragments (4 VGPRs each)
};

template <class S, int k>
constexpr int nfrag() { return Sh<S, k>::NWREG; }

// element offset of (row, kk) in a bf16 image with rows of KI elements
template <int KI>
__device__ __forceinline__ int h_off(int row, int kk) {
  constexpr int ns = KI / 8;
  const int slot = kk >> 3;
  if constexpr (ns >= 8 && is_pow2(ns)) {
    const int g = (ns >= 16) ? (row & 15) : ((row >> 1) & 7);
    return ((row * ns + (slot ^ g)) << 3) + (kk & 7);
  } else {
    return row * KI + kk;
  }
}

template <class S, int k, int NF_>
__device__ __forceinline__ void load_hfrag(hbf8 (&w)[NF_], const float* packed, int wave, int lane) {
  using T = Sh<S, k>;
  static_assert(NF_ == T::NWREG, "fragment array size");
  const int r = lane & 15, q = lane >> 4;
  const float* W = packed + woff_of<S>(k);
#pragma unroll
  for (int x = 0; x < T::XM; ++x) {
    const int mt = T::SPLIT ? (wave % T::MT) : (wave + FAST_NW * x);
    const int m = 16 * mt + r;
#pragma unroll
    for (int u = 0; u < T::NM; ++u) {
      hbf8 f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kk = 32 * u + 8 * q + e;
        const bool okw = mt < T::MT && m < T::M && kk < T::K;
        const float v = W[okw ? kk * T::M + m : 0];
        f[e] = (__bf16)(okw ? v : 0.f);
      }
      w[x * T::NM + u] = f;
    }
  }
}

// Ain: bf16 image [ROWS][KI];  Cout: bf16 image of the next stage (k > 0) or the fp32 gate vector (k == 0)
template <class S, int k, int NF_>
__device__ __forceinline__ void run_hstage(const hbf8 (&w)[NF_], const __bf16* Ain, void* Cout, int wave, int lane) {
  using T = Sh<S, k>;
  const int c = lane & 15, q = lane >> 4;
  hbf8 af[T::YR][T::NM];
#pragma unroll
  for (int y = 0; y < T::YR; ++y) {
    const int rt = T::SPLIT ? (wave / T::MT + T::G * y) : y;
    int row = 16 * rt + c;
    row = row < T::ROWS ? row : T::ROWS - 1;
#pragma unroll
    for (int u = 0; u < T::NM; ++u) {
      const int slot = 4 * u + q;
      hbf8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
      if (slot < T::NSL) v = *reinterpret_cast<const hbf8*>(Ain + h_off<T::KI>(row, 8 * slot));
      af[y][u] = v;
    }
  }
  f32x4 acc[T::XM][T::YR];
#pragma unroll
  for (int x = 0; x < T::XM; ++x)
#pragma unroll
    for (int y = 0; y < T::YR; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < T::NM; ++u)
#pragma unroll
    for (int x = 0; x < T::XM; ++x)
#pragma unroll
      for (int y = 0; y < T::YR; ++y)
        acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[x * T::NM + u], af[y][u], acc[x][y], 0, 0, 0);
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
          using N = Sh<S, k - 1>;
          const int i = m0 / T::R, a0 = m0 % T::R;
          const int f = i * (T::ROWS * T::R) + row * T::R + a0;
          hbf4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (__bf16)acc[x][y][j];
          *reinterpret_cast<hbf4*>(reinterpret_cast<__bf16*>(Cout) + h_off<N::KI>(f / N::K, f % N::K)) = o;
        } else {
          float* gb = reinterpret_cast<float*>(Cout);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (m0 + j < T::M) gb[(m0 + j) * T::ROWS + row] = acc[x][y][j];
        }
      }
    }
  }
}

template <class S>
constexpr int maxmid_h() {      // largest intermediate image in bf16 elements
  int best = 8;
  for (int k = 1; k < S::D; ++k) {
    int rows = 1;
    for (int m = k + 1; m < S::D; ++m) rows *= S::I[m];
    for (int m = 0; m < k; ++m) rows *= S::J[m];
    const int e = rows * S::I[k] * S::R[k];
    if (e > best) best = e;
  }
  return best;
}

__device__ __forceinline__ float hsigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float htanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// gin: fp32 [B][T][H][4] gate-interleaved (LSTM i,g,f,o / GRU r,z,n,-).  One workgroup per sample.
template <class S, int CELL>
__global__ void __launch_bounds__(FAST_NT) k_rnn_fwd_bf16(int B, int T, GinSrc gs,
                                                          const bf16_t* __restrict__ h0, const bf16_t* __restrict__ c0,
                                                          const float* __restrict__ packed_hid,
                                                          const bf16_t* __restrict__ bias_hid, bf16_t* __restrict__ out,
                                                          bf16_t* __restrict__ hT, bf16_t* __restrict__ cT,
                                                          float* __restrict__ reserve) {
  constexpr int D = S::D;
  constexpr int H = in_size_of<S>();
  constexpr int G = CELL == TTRNN_LSTM ? 4 : 3;
  constexpr int GH = G * H;
  static_assert(out_size_of<S>() == GH, "TT output size must be n_gates * hidden");
  constexpr int HPT = (H + FAST_NT - 1) / FAST_NT;
  constexpr int MID = maxmid_h<S>();
  using SL = Sh<S, D - 1>;
  constexpr int HIMG = SL::ROWS * SL::KI;

  __shared__ __attribute__((aligned(16))) __bf16 hbuf[HIMG];
  __shared__ __attribute__((aligned(16))) __bf16 bufA[MID];
  __shared__ __attribute__((aligned(16))) __bf16 bufB[D > 2 ? MID : 8];
  __shared__ __attribute__((aligned(16))) float gbuf[GH];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t b = blockIdx.x;

  hbf8 w0[nfrag<S, 0>()];
  hbf8 w1[nfrag<S, (D > 1 ? 1 : 0)>()];
  hbf8 w2[nfrag<S, (D > 2 ? 2 : 0)>()];
  hbf8 w3[nfrag<S, (D > 3 ? 3 : 0)>()];
  load_hfrag<S, 0>(w0, packed_hid, wave, lane);
  if constexpr (D > 1) load_hfrag<S, 1>(w1, packed_hid, wave, lane);
  if constexpr (D > 2) load_hfrag<S, 2>(w2, packed_hid, wave, lane);
  if constexpr (D > 3) load_hfrag<S, 3>(w3, packed_hid, wave, lane);

  // zero the image padding once (rows of K < 8 elements)
  for (int e = tid; e < HIMG; e += FAST_NT) hbuf[e] = (__bf16)0.f;
  __syncthreads();

  const float* __restrict__ gin = gs.gin;
  const bf16_t* __restrict__ xs = reinterpret_cast<const bf16_t*>(gs.x);
  const bool in1 = gs.in1 != 0;
  const f32x4* gin4 = reinterpret_cast<const f32x4*>(gin);
  float cst[HPT], hst[HPT], bh[HPT][G];
  f32x4 gi[HPT], vv[HPT], bb[HPT];
  XChunk<bf16_t> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int hid = tid + u * FAST_NT;
    const bool ok = hid < H;
    hst[u] = (ok && h0) ? ld(h0, b * H + hid) : 0.f;
    cst[u] = (ok && c0 && CELL == TTRNN_LSTM) ? ld(c0, b * H + hid) : 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) bh[u][g] = (ok && bias_hid) ? ld(bias_hid, g * H + hid) : 0.f;
    vv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    bb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    gi[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok && T > 0) {
      if (in1) {
        bb[u] = gin4[H + hid];
        vv[u] = gin4[hid] - bb[u];
      } else {
        gi[u] = gin4[(b * T) * H + hid];
      }
    }
    if (ok) hbuf[h_off<SL::KI>(hid / SL::K, hid % SL::K)] = (__bf16)hst[u];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): keep weight-register waits out of the time loop
  lds_barrier();

  for (int t = 0; t < T; ++t) {
    if constexpr (D == 1) {
      run_hstage<S, 0>(w0, hbuf, gbuf, wave, lane);
    } else if constexpr (D == 2) {
      run_hstage<S, 1>(w1, hbuf, bufA, wave, lane);
      lds_barrier();
      run_hstage<S, 0>(w0, bufA, gbuf, wave, lane);
    } else if constexpr (D == 3) {
      run_hstage<S, 2>(w2, hbuf, bufA, wave, lane);
      lds_barrier();
      run_hstage<S, 1>(w1, bufA, bufB, wave, lane);
      lds_barrier();
      run_hstage<S, 0>(w0, bufB, gbuf, wave, lane);
    } else {
      run_hstage<S, 3>(w3, hbuf, bufA, wave, lane);
      lds_barrier();
      run_hstage<S, 2>(w2, bufA, bufB, wave, lane);
      lds_barrier();
      run_hstage<S, 1>(w1, bufB, bufA, wave, lane);
      lds_barrier();
      run_hstage<S, 0>(w0, bufA, gbuf, wave, lane);
    }
    lds_barrier();
    const size_t bt = b * T + t;
    const float xt = in1 ? xq.at(t) : 0.f;
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
      const int hid = tid + u * FAST_NT;
      if (hid < H) {
        float hy;
        if (in1) gi[u] = bb[u] + xt * vv[u];      // W_in x_t + b_in from the two unit rows (GinSrc)
        if constexpr (CELL == TTRNN_LSTM) {
          const float ig = hsigmoid(gi[u][0] + gbuf[hid] + bh[u][0]);              // lstm.py:26
          const float fg = hsigmoid(gi[u][2] + gbuf[H + hid] + bh[u][1]);          // lstm.py:27
          const float gg = htanh(gi[u][1] + gbuf[2 * H + hid] + bh[u][2]);         // lstm.py:28
          const float og = hsigmoid(gi[u][3] + gbuf[3 * H + hid] + bh[u][3]);      // lstm.py:29
          const float cy = fg * cst[u] + ig * gg;                                  // lstm.py:31
          hy = og * htanh(cy);                                                     // lstm.py:32
          cst[u] = cy;
          if (reserve) {
            float* rv = reserve + res_gate(bt, H, hid);
            rv[0] = ig; rv[1] = gg; rv[2] = fg; rv[3] = og;
            reserve[res_cell((size_t)B * T, bt, H, hid)] = cy;
          }
        } else {
          const float hn = gbuf[2 * H + hid] + bh[u][2];
          const float rg = hsigmoid(gi[u][0] + gbuf[hid] + bh[u][0]);              // gru.py:38-39
          const float zg = hsigmoid(gi[u][1] + gbuf[H + hid] + bh[u][1]);          // gru.py:40-41
          const float ng = htanh(gi[u][2] + rg * hn);                              // gru.py:42-43
          hy = (1.0f - zg) * ng + zg * hst[u];                                     // gru.py:44
          if (reserve) *reinterpret_cast<f32x4*>(reserve + (bt * H + hid) * 4) = f32x4{rg, zg, ng, hn};
        }
        const bf16_t hb = f32_to_bf16(hy);            // rounded once: stored, fed back, kept as state
        if (out) out[bt * H + hid] = hb;              // out == NULL: final state only (ttrnn_rnn_out_optional)
        hy = bf16_to_f32(hb);
        hst[u] = hy;
        hbuf[h_off<SL::KI>(hid / SL::K, hid % SL::K)] = (__bf16)hy;
        if (!in1 && t + 1 < T) gi[u] = gin4[(bt + 1) * H + hid];
      }
    }
    if (in1) xq.advance(xs, b * T, T, t, lane);
    lds_barrier();
  }
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int hid = tid + u * FAST_NT;
    if (hid < H) {
      if (hT) st(hT, b * H + hid, hst[u]);
      if (CELL == TTRNN_LSTM && cT) st(cT, b * H + hid, cst[u]);
    }
  }
}

template <class S, int CELL>
static int launch_bf16(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                       const void* bias_hid, void* out, void* hT, void* cT, float* reserve, hipStream_t stream) {
  static_assert(shape_ok_recurrent<S>(), "shape not supported by the MFMA path");
  hipLaunchKernelGGL((k_rnn_fwd_bf16<S, CELL>), dim3(rs.B), dim3(FAST_NT), 0, stream, rs.B, rs.T, gin,
                     (const bf16_t*)h0, (const bf16_t*)c0, packed_hid,
                     rs.has_bias_hid ? (const bf16_t*)bias_hid : (const bf16_t*)nullptr, (bf16_t*)out, (bf16_t*)hT,
                     (bf16_t*)cT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

bool fast_rnn_fwd_bf16_available(const RnnShape& rs, int dtype) {
  if (dtype != TTRNN_BF16 || rs.B < 1 || rs.T < 1) return false;
  const bool keep_fp32 = opt(OPT_BF16_FP32_MFMA) != 0;      // A/B switch: keep bf16 storage on the fp32 MFMA kernels
  if (keep_fp32) return false;
  if (rs.cell == TTRNN_LSTM) return shape_matches<ShpH256R8L>(rs.hid_s) || shape_matches<ShpH256R16L>(rs.hid_s);
  return shape_matches<ShpH256R8G>(rs.hid_s) || shape_matches<ShpH256R16G>(rs.hid_s);
}

int launch_rnn_fwd_bf16(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0,
                        const float* packed_hid, const void* bias_hid, void* out, void* hT, void* cT, float* reserve,
                        hipStream_t stream) {
#define TT_TRY(SHAPE, CELL)                               \
  if (rs.cell == CELL && shape_matches<SHAPE>(rs.hid_s)) \
    return launch_bf16<SHAPE, CELL>(rs, gin, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, stream)
  TT_TRY(ShpH256R8L, TTRNN_LSTM);
  TT_TRY(ShpH256R16L, TTRNN_LSTM);
  TT_TRY(ShpH256R8G, TTRNN_GRU);
  TT_TRY(ShpH256R16G, TTRNN_GRU);
#undef TT_TRY
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
