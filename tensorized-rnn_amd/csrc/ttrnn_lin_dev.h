// ttrnn_lin_dev.h — device body of the shape-specialised batched TTLinear forward on fp32 MFMA (ttrnn_fast_lin.hip), shared with the
// fused set-up kernel (ttrnn_fast_setup.hip).  Device-only (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include "ttrnn_core.h"
#include "ttrnn_mfma.h"

namespace ttrnn {

__device__ __forceinline__ void store4(float* y, size_t idx, f32x4 v) { *reinterpret_cast<f32x4*>(y + idx) = v; }
__device__ __forceinline__ void store4(bf16_t* y, size_t idx, f32x4 v) {
  typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
  u16x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = f32_to_bf16(v[j]).v;
  *reinterpret_cast<u16x4*>(y + idx) = r;
}

template <class S, int k, int NB>
constexpr int mid_elems() {      // floats of the stage-k output image (input of stage k-1), k in 1..D-1
  return (k >= 1 && k < S::D) ? NB * St<S, (k >= 1 && k < S::D) ? k : 0>::ROWS * St<S, (k >= 1 && k < S::D) ? k : 0>::M : 4;
}

// G = 0: plain y[n][out];  G = 3/4: gate-interleaved y[n][H][4]
// (the body of k_ttlinear_fwd_fast, ttrnn_fast_lin.hip; also run by ONE workgroup of the fused set-up kernel of ttrnn_fast_setup.hip on
// the two unit rows of an input_size == 1 layer, with `packed` pointing at that workgroup's LDS copy of the packed cores:
// bid / nblk = this workgroup's index / the number of workgroups walking the row tiles; FAST_NT threads)
template <class S, int NB, int G, typename TI, typename TO>
__device__ __forceinline__ void ttlinear_fwd_fast_body(int64_t n_rows, const float* __restrict__ packed,
                                                       const TI* __restrict__ bias, const TI* __restrict__ x,
                                                       TO* __restrict__ y, int ilv_mode, int bid, int nblk) {
  constexpr int D = S::D;
  constexpr int IN = in_size_of<S>(), OUT = out_size_of<S>();
  using SL = St<S, D - 1>;
  constexpr int YT = G == 0 ? OUT : (OUT / (G > 0 ? G : 1)) * 4;     // floats per sample in the output tile
  constexpr int IMG0 = NB * SL::ROWS * SL::KP;

  __shared__ __attribute__((aligned(16))) float img0[IMG0 > 4 ? IMG0 : 4];
  __shared__ __attribute__((aligned(16))) float mid1[mid_elems<S, 1, NB>()];
  __shared__ __attribute__((aligned(16))) float mid2[mid_elems<S, 2, NB>()];
  __shared__ __attribute__((aligned(16))) float mid3[mid_elems<S, 3, NB>()];
  __shared__ __attribute__((aligned(16))) float ytile[NB * YT];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  float w0[nwreg<S, 0>()];
  float w1[nwreg<S, (D > 1 ? 1 : 0)>()];
  float w2[nwreg<S, (D > 2 ? 2 : 0)>()];
  float w3[nwreg<S, (D > 3 ? 3 : 0)>()];
  load_wfrag<S, 0>(w0, packed, wave, lane);
  if constexpr (D > 1) load_wfrag<S, 1>(w1, packed, wave, lane);
  if constexpr (D > 2) load_wfrag<S, 2>(w2, packed, wave, lane);
  if constexpr (D > 3) load_wfrag<S, 3>(w3, packed, wave, lane);

  // zero once: the K padding of the first image is never overwritten; a 3-gate tile keeps slot 3 = 0
  for (int e = tid; e < IMG0; e += FAST_NT) img0[e] = 0.f;
  if constexpr (G == 3)
    for (int e = tid; e < NB * YT; e += FAST_NT) ytile[e] = 0.f;
  __syncthreads();

  const int64_t ntiles = (n_rows + NB - 1) / NB;
  for (int64_t tile = bid; tile < ntiles; tile += nblk) {
    const int64_t n0 = tile * NB;
    // x rows -> first image: sample smp, feature j -> chain row j / K, column j % K
    for (int e = tid; e < NB * IN; e += FAST_NT) {
      const int smp = e / IN, j = e - smp * IN;
      const float v = (n0 + smp < n_rows) ? ld(x, (n0 + smp) * IN + j) : 0.f;
      img0[a_off<SL::KP>(smp * SL::ROWS + j / SL::K, j % SL::K)] = v;
    }
    __syncthreads();
    if constexpr (D == 1) {
      lin_stage<S, 0, NB, G>(w0, img0, ytile, wave, lane, ilv_mode);
    } else if constexpr (D == 2) {
      lin_stage<S, 1, NB, G>(w1, img0, mid1, wave, lane, ilv_mode);
      __syncthreads();
      lin_stage<S, 0, NB, G>(w0, mid1, ytile, wave, lane, ilv_mode);
    } else if constexpr (D == 3) {
      lin_stage<S, 2, NB, G>(w2, img0, mid2, wave, lane, ilv_mode);
      __syncthreads();
      lin_stage<S, 1, NB, G>(w1, mid2, mid1, wave, lane, ilv_mode);
      __syncthreads();
      lin_stage<S, 0, NB, G>(w0, mid1, ytile, wave, lane, ilv_mode);
    } else {
      lin_stage<S, 3, NB, G>(w3, img0, mid3, wave, lane, ilv_mode);
      __syncthreads();
      lin_stage<S, 2, NB, G>(w2, mid3, mid2, wave, lane, ilv_mode);
      __syncthreads();
      lin_stage<S, 1, NB, G>(w1, mid2, mid1, wave, lane, ilv_mode);
      __syncthreads();
      lin_stage<S, 0, NB, G>(w0, mid1, ytile, wave, lane, ilv_mode);
    }
    __syncthreads();
    // coalesced copy-out (+ bias)
    if constexpr (G > 0) {
      constexpr int H = OUT / G;
      const f32x4* yt4 = reinterpret_cast<const f32x4*>(ytile);
      for (int e = tid; e < NB * H; e += FAST_NT) {
        const int smp = e / H, hid = e - smp * H;
        if (n0 + smp < n_rows) {
          f32x4 v = yt4[e];
          if (bias) {
#pragma unroll
            for (int sl = 0; sl < G; ++sl) {
              const int g = (ilv_mode == 2) ? (sl == 1 ? 2 : (sl == 2 ? 1 : sl)) : sl;
              v[sl] += ld(bias, g * H + hid);
            }
          }
          store4(y, ((n0 + smp) * H + hid) * 4, v);
        }
      }
    } else if constexpr (OUT % 4 == 0) {
      const f32x4* yt4 = reinterpret_cast<const f32x4*>(ytile);
      constexpr int O4 = OUT / 4;
      for (int e = tid; e < NB * O4; e += FAST_NT) {
        const int smp = e / O4, o4 = e - smp * O4;
        if (n0 + smp < n_rows) {
          f32x4 v = yt4[e];
          if (bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += ld(bias, 4 * o4 + j);
          }
          store4(y, ((n0 + smp) * O4 + o4) * 4, v);
        }
      }
    } else {
      for (int e = tid; e < NB * OUT; e += FAST_NT) {
        const int smp = e / OUT, o = e - smp * OUT;
        if (n0 + smp < n_rows) st(y, (n0 + smp) * OUT + o, ytile[e] + (bias ? ld(bias, o) : 0.f));
      }
    }
    __syncthreads();
  }
}

}  // namespace ttrnn
