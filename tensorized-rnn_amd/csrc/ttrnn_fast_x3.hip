// ttrnn_fast_x3.hip — fp32 TT-LSTM recurrent kernel with the chain on SPLIT-bf16 MFMAs (gfx950).
//
// The fp32 MFMA (v_mfma_f32_16x16x4_f32, 256 FLOP/clk/CU) bounds ttrnn_fast.hip's time loop: cfg2 spends 2816 of
// its ~4400 cycles per timestep inside that pipe.  The bf16 MFMA (v_mfma_f32_16x16x32_bf16) is 16x faster per
// FLOP, and an fp32 product can be rebuilt from bf16 pieces without giving up fp32 accuracy:
//     x = x0 + x1 + x2,  w = w0 + w1 + w2      (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); |x_i| <= 2^-9 |x_{i-1}|)
//     x*w = x0w0 + (x0w1 + x1w0) + (x0w2 + x1w1 + x2w0) + O(2^-26 |xw|)
// Each bf16*bf16 product is exact in fp32 and the MFMA accumulates in fp32, so the six-term sum carries a
// relative error below fp32's own rounding unit (2^-24) per product — the same class of result as the fp32 MFMA
// (summation order differs, as it does between any two fp32 GEMM implementations), at 6*16 = 96 instead of 256
// MFMA cycles per 16x16x32 block: 2.67x less matrix-pipe time.  tests/test_gpu_parity.py measures both modes
// against the fp64 oracle.
//
// Where the pieces come from:
//   * cores are split ONCE per launch into three resident bf16 fragment sets (VGPRs);
//   * activations are split by the PRODUCER of each chain intermediate (accumulators -> three bf16 planes in
//     LDS, v_cvt_pk_bf16_f32), so every element is split once, not once per consumer;
//   * stages whose contraction length is not a multiple of 32 (the first stage, K = J_{d-1}) stay on the fp32 MFMA.
// Everything else — one workgroup per sample, persistent over T, gate math on the stage-0 accumulators (PAIR lanes),
// hoisted input projection prefetched across raw barriers — is ttrnn_fast.hip's k_lstm_fwd_fused.
//
// Replaces, for one layer: tensorized_rnn/lstm.py:23-32,123-133 with the hidden chain of t3nsor/ops.py:78-93.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_mfma.h"

namespace ttrnn {

typedef __bf16 xbf8 __attribute__((ext_vector_type(8)));
typedef __bf16 xbf4 __attribute__((ext_vector_type(4)));

// wave grid of a split stage: GM groups along the output-feature tiles, GR = 8 / GM along the chain-row tiles.
// Few m-groups = every wave reads few row tiles (LDS reads are the cost that replaces the MFMA time); the bound
// is the resident core fragments, 12 VGPRs per (m-tile, 32-wide k block).
constexpr int x3_pick_gm(int mt, int nm, int budget) {
  for (int g = 1; g < FAST_NW; g *= 2)
    if (((mt + g - 1) / g) * nm * 12 <= budget) return g;
  return FAST_NW;
}

template <class S, int k>
struct Sx {
  using F = St<S, k>;
  static constexpr int K = F::K, M = F::M, R = F::R, ROWS = F::ROWS;
  static constexpr int NSL = K / 8;               // 16-byte slots per image row
  static constexpr int NM = K / 32;               // 32-wide k blocks
  static constexpr int MT = F::MT, RT = F::RT;
  static constexpr int GM = x3_pick_gm(MT, NM, 64), GR = FAST_NW / GM;
  static constexpr int XM = (MT + GM - 1) / GM, YR = (RT + GR - 1) / GR;
  static constexpr int NF = XM * NM;              // resident fragments per plane (4 VGPRs each)
  static constexpr int PLANE = ROWS * K;          // bf16 elements per plane of the input image
};

template <class S, int k>
constexpr bool x3_stage() { return St<S, k>::K % 32 == 0; }

// element offset of (row, kk) inside one bf16 plane [ROWS][K]; 16-byte slots XOR-swizzled so that the ds_read_b128
// fragment reads (lane (c, q) -> slot 4u+q of row 16rt+c) are conflict-free for the b128 lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X guide, LDS table).
template <int K>
__device__ __forceinline__ int x_off(int row, int kk) {
  constexpr int ns = K / 8;
  const int slot = kk >> 3;
  int g = 0;
  if constexpr (ns == 4) g = (-(row >> 2)) & 3;
  else if constexpr (ns == 8) g = (row >> 1) & 7;
  else if constexpr (ns >= 16 && is_pow2(ns)) g = row & 15;
  return ((row * ns + (slot ^ g)) << 3) + (kk & 7);
}

__device__ __forceinline__ void split3(float v, __bf16& p0, __bf16& p1, __bf16& p2) {
  p0 = (__bf16)v;
  float r = v - (float)p0;      // exact
  p1 = (__bf16)r;
  r -= (float)p1;               // exact
  p2 = (__bf16)r;
}

// two elements at a time: v_cvt_pk_bf16_f32 rounds both (RNE), a shift / a mask turn the packed pair back into
// floats, the subtractions are exact.  9 VALU instructions per pair and three packed dwords out.
typedef __bf16 xbf2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const xbf2 p = __builtin_convertvector(f32x2{a, b}, xbf2);
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pk_bf16(a, b);
  float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
  p1 = pk_bf16(ra, rb);
  ra -= __uint_as_float(p1 << 16);
  rb -= __uint_as_float(p1 & 0xffff0000u);
  p2 = pk_bf16(ra, rb);
}

// four consecutive elements -> three 8-byte stores
__device__ __forceinline__ void store_split4(__bf16* img, int plane_elems, int off, f32x4 v) {
  unsigned a0, b0, c0, a1, b1, c1;
  split_pair(v[0], v[1], a0, b0, c0);
  split_pair(v[2], v[3], a1, b1, c1);
  *reinterpret_cast<u32x2*>(img + off) = u32x2{a0, a1};
  *reinterpret_cast<u32x2*>(img + plane_elems + off) = u32x2{b0, b1};
  *reinterpret_cast<u32x2*>(img + 2 * plane_elems + off) = u32x2{c0, c1};
}

// resident split fragments of core k.  PERM0: stage-0 rows permuted for the fused LSTM gate phase (see
// ttrnn_fast.hip: load_wfrag0_lstm) — PAIR mode only.
template <class S, int k, bool PERM0, int NF_>
__device__ __forceinline__ void load_xfrag(xbf8 (&w)[3][NF_], const float* packed, int wave, int lane) {
  using T = Sx<S, k>;
  static_assert(NF_ == T::NF, "fragment array size");
  const int r = lane & 15, q = lane >> 4;
  const float* W = packed + woff_of<S>(k);
  const int gm = wave % T::GM;
#pragma unroll
  for (int x = 0; x < T::XM; ++x) {
    const int mt = gm + T::GM * x;
    int m = 16 * mt + r;
    bool okm = mt < T::MT && m < T::M;
    if constexpr (PERM0) {
      constexpr int P = S::I[0] / 4;
      const int j = r & 3, qq = r >> 2, pair = qq >> 1;
      const int gate = (j == 0) ? (pair ? 1 : 0) : (pair ? 3 : 2);
      m = gate * P + (qq & 1);
      okm = mt < T::MT && j < 2 && (qq & 1) < P;
    }
#pragma unroll
    for (int u = 0; u < T::NM; ++u) {
      xbf8 f0, f1, f2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kk = 32 * u + 8 * q + e;
        const float v = W[okm ? kk * T::M + m : 0];
        __bf16 p0, p1, p2;
        split3(okm ? v : 0.f, p0, p1, p2);
        f0[e] = p0; f1[e] = p1; f2[e] = p2;
      }
      w[0][x * T::NM + u] = f0;
      w[1][x * T::NM + u] = f1;
      w[2][x * T::NM + u] = f2;
    }
  }
}

// MFMAs of split stage k for the tiles this wave owns.  Ain: three bf16 planes [ROWS][K] (x_off), PLANE apart.
template <class S, int k, int NF_, int XM_, int YR_>
__device__ __forceinline__ void x3_mma(const xbf8 (&w)[3][NF_], const __bf16* Ain, f32x4 (&acc)[XM_][YR_], int wave,
                                       int lane) {
  using T = Sx<S, k>;
  static_assert(NF_ == T::NF && XM_ == T::XM && YR_ == T::YR, "stage tile bookkeeping");
  const int c = lane & 15, q = lane >> 4;
  const int gr = wave / T::GM;
  xbf8 af[3][T::YR][T::NM];
#pragma unroll
  for (int y = 0; y < T::YR; ++y) {
    const int rt = gr + T::GR * y;
    int row = 16 * rt + c;
    row = row < T::ROWS ? row : T::ROWS - 1;
#pragma unroll
    for (int u = 0; u < T::NM; ++u) {
      const int off = x_off<T::K>(row, 32 * u + 8 * q);
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p][y][u] = *reinterpret_cast<const xbf8*>(Ain + p * T::PLANE + off);
    }
  }
#pragma unroll
  for (int x = 0; x < T::XM; ++x)
#pragma unroll
    for (int y = 0; y < T::YR; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  // smallest terms first: (w2,x0) (w0,x2) (w1,x1) | (w1,x0) (w0,x1) | (w0,x0)
  constexpr int TW[6] = {2, 0, 1, 1, 0, 0};
  constexpr int TX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int u = 0; u < T::NM; ++u)
#pragma unroll
      for (int x = 0; x < T::XM; ++x)
#pragma unroll
        for (int y = 0; y < T::YR; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[TW[t]][x * T::NM + u], af[TX[t]][y][u], acc[x][y], 0, 0, 0);
}

// split stage k > 0 whose consumer (stage k-1) is a split stage too: result goes out as three bf16 planes
template <class S, int k, int NF_>
__device__ __forceinline__ void run_xstage(const xbf8 (&w)[3][NF_], const __bf16* Ain, __bf16* Cout, int wave, int lane) {
  using T = Sx<S, k>;
  using N = Sx<S, k - 1>;
  static_assert(k > 0 && x3_stage<S, k - 1>(), "consumer must be a split stage");
  const int c = lane & 15, q = lane >> 4;
  const int gm = wave % T::GM, gr = wave / T::GM;
  f32x4 acc[T::XM][T::YR];
  x3_mma<S, k>(w, Ain, acc, wave, lane);
#pragma unroll
  for (int x = 0; x < T::XM; ++x) {
    const int mt = gm + T::GM * x;
#pragma unroll
    for (int y = 0; y < T::YR; ++y) {
      const int rt = gr + T::GR * y;
      const int row = 16 * rt + c;
      const int m0 = 16 * mt + 4 * q;
      if (mt < T::MT && rt < T::RT && row < T::ROWS && m0 < T::M) {
        const int i = m0 / T::R, a0 = m0 % T::R;
        const int f = i * (T::ROWS * T::R) + row * T::R + a0;     // C_k flat == A_{k-1} flat (ops.py:89-90)
        store_split4(Cout, N::PLANE, x_off<N::K>(f / N::K, f % N::K), acc[x][y]);
      }
    }
  }
}

// fp32-MFMA stage k (reads an fp32 image, ttrnn_mfma.h) feeding a split stage: same MFMAs, split epilogue
template <class S, int k, int NW_>
__device__ __forceinline__ void run_stage_to_x3(const float (&w)[NW_], const float* Ain, __bf16* Cout, int wave, int lane) {
  using T = St<S, k>;
  using N = Sx<S, k - 1>;
  static_assert(k > 0 && x3_stage<S, k - 1>(), "consumer must be a split stage");
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
        const int i = m0 / T::R, a0 = m0 % T::R;
        const int f = i * (T::ROWS * T::R) + row * T::R + a0;
        store_split4(Cout, N::PLANE, x_off<N::K>(f / N::K, f % N::K), acc[x][y]);
      }
    }
  }
}

// shapes this file handles: d = 3, first executed stage on the fp32 MFMA (K = J_2 < 32), stages 1 and 0 split,
// PAIR gate layout (I_0 / 4 <= 2, one stage-0 m-tile)
template <class S>
constexpr bool x3_lstm_ok() {
  return S::D == 3 && S::R[0] == 1 && S::I[0] % 4 == 0 && S::I[0] / 4 <= 2 && !x3_stage<S, 2>() && x3_stage<S, 1>() &&
         x3_stage<S, 0>() && shape_ok_recurrent<S>() && Sx<S, 0>::MT == 1 && Sx<S, 0>::YR == 1;
}

template <class S>
constexpr size_t x3_lds_bytes() {
  return sizeof(float) * in_size_of<S>() + 2 * 3 * (size_t)(Sx<S, 1>::PLANE + Sx<S, 0>::PLANE);
}

template <class S, bool DIAG>
__global__ void __launch_bounds__(FAST_NT) k_lstm_fwd_x3(int B, int T, GinSrc gs, const float* __restrict__ h0,
                                                         const float* __restrict__ c0,
                                                         const float* __restrict__ packed_hid,
                                                         const float* __restrict__ bias_hid, float* __restrict__ out,
                                                         float* __restrict__ hT, float* __restrict__ cT,
                                                         float* __restrict__ reserve) {
  static_assert(x3_lstm_ok<S>(), "shape not supported by the split-bf16 kernel");
  constexpr int H = in_size_of<S>();
  static_assert(out_size_of<S>() == 4 * H, "TT output size must be 4 * hidden");
  using SL = St<S, 2>;
  using X1 = Sx<S, 1>;
  using X0 = Sx<S, 0>;
  constexpr int P = S::I[0] / 4;
  static_assert(P * X0::ROWS == H, "hidden index decomposition");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* hbuf = reinterpret_cast<float*>(smem);                       // fp32 image [ROWS_2][J_2] of h
  __bf16* img1 = reinterpret_cast<__bf16*>(smem + sizeof(float) * H); // 3 planes, input of stage 1
  __bf16* img0 = img1 + 3 * X1::PLANE;                                // 3 planes, input of stage 0

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const size_t b = blockIdx.x;

  float w2[nwreg<S, 2>()];
  xbf8 w1[3][X1::NF];
  xbf8 w0[3][X0::NF];
  load_wfrag<S, 2>(w2, packed_hid, wave, lane);
  load_xfrag<S, 1, false>(w1, packed_hid, wave, lane);
  load_xfrag<S, 0, true>(w0, packed_hid, wave, lane);

  // the hidden unit of this lane (stage 0: one m-tile, row tile = wave): lanes 0-31 evaluate (i, g), lanes 32-63
  // (f, o) and own c / h.  gin is gate-interleaved [B][T][H][4], slots i,g,f,o.
  const int pair = q >> 1;
  const float sgn = pair == 0 ? 2.0f : 1.0f;            // second gate: tanh (= 2*sigmoid(2x) - 1) or sigmoid
  const float* __restrict__ gin = gs.gin;
  const float* __restrict__ xs = reinterpret_cast<const float*>(gs.x);
  const bool in1 = gs.in1 != 0;
  const int row0 = 16 * (wave / X0::GM) + c, par = q & 1;
  const bool ok = row0 < X0::ROWS && par < P;
  const int hd = ok ? par * X0::ROWS + row0 : 0;
  float hst = (ok && h0) ? h0[b * H + hd] : 0.f;
  float cst = (ok && c0) ? c0[b * H + hd] : 0.f;
  float bh[2], gi[2], vv[2], bb[2];
  XChunk<float> xq;
  xq.cur = 0.f; xq.nxt = 0.f;
  if (in1) xq.init(xs, b * T, T, lane);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int slot = 2 * pair + g;
    const int gate = slot == 1 ? 2 : (slot == 2 ? 1 : slot);    // i,g,f,o -> reference gate index 0,2,1,3
    bh[g] = (ok && bias_hid) ? bias_hid[gate * H + hd] : 0.f;
    vv[g] = 0.f; bb[g] = 0.f; gi[g] = 0.f;
    if (ok && T > 0) {
      if (in1) {
        bb[g] = gin[(H + hd) * 4 + slot];
        vv[g] = gin[hd * 4 + slot] - bb[g];
      } else {
        gi[g] = gin[((b * T) * H + hd) * 4 + slot];
      }
    }
  }
  if (ok && pair) hbuf[a_off<SL::K>(hd / SL::K, hd % SL::K)] = hst;
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) here, so that no weight-register wait lands inside the loop
  lds_barrier();
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
  if constexpr (DIAG) last_ = stamp();

  for (int t = 0; t < T; ++t) {
    run_stage_to_x3<S, 2>(w2, hbuf, img1, wave, lane);
    TT_STAMP(0)
    lds_barrier();
    TT_STAMP(1)
    run_xstage<S, 1>(w1, img1, img0, wave, lane);
    TT_STAMP(2)
    lds_barrier();
    TT_STAMP(3)
    f32x4 acc[1][1];
    x3_mma<S, 0>(w0, img0, acc, wave, lane);
    if constexpr (DIAG) {
      asm volatile("" : "+v"(acc[0][0]));
    }
    TT_STAMP(4)
    const size_t bt = b * T + t;
    {
      if (in1) {      // W_in x_t + b_in = b + x_t * (chain(1) - chain(0)), x_t from the resident chunk
        const float xt = xq.at(t);
        gi[0] = bb[0] + xt * vv[0];
        gi[1] = bb[1] + xt * vv[1];
      }
      const float u = fsigmoid(acc[0][0][0] + gi[0] + bh[0]);                                 // lstm.py:26-27
      const float a1 = acc[0][0][1] + gi[1] + bh[1];
      const float v = sgn * fsigmoid(sgn * a1) + (1.0f - sgn);                                // lstm.py:28-29
      const float prod = u * v;                                                               // i*g on lanes 0-31
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(prod), __float_as_uint(prod), false, false);
      const float ig_g = __uint_as_float(sw[0]);                                              // lanes 32-63 <- lanes 0-31
      const float cy = u * cst + ig_g;                                                        // lstm.py:31
      const float hy = v * ftanh(cy);                                                         // lstm.py:32
      if (reserve && ok) {
        float* rv = reserve + (bt * H + hd) * 8 + 2 * pair;                                   // i,g | f,o,c
        rv[0] = u; rv[1] = v;
        if (pair) rv[2] = cy;
      }
      if (ok && pair) {
        cst = cy;
        hst = hy;
        hbuf[a_off<SL::K>(hd / SL::K, hd % SL::K)] = hy;
      }
      if (in1) {
        xq.advance(xs, b * T, T, t, lane);
      } else if (ok && t + 1 < T) {      // prefetch the hoisted projection of the next step, consumed one step later
        const f32x2 nx = *reinterpret_cast<const f32x2*>(gin + ((bt + 1) * H + hd) * 4 + 2 * pair);
        gi[0] = nx[0]; gi[1] = nx[1];
      }
    }
    TT_STAMP(5)
    lds_barrier();
    TT_STAMP(6)
    // outputs[:, t, :] = h_t (lstm.py:133): one coalesced store per timestep by the last wave
    if (wave == FAST_NW - 1) {
#pragma unroll
      for (int h4 = lane; h4 < H / 4; h4 += 64) {
        const int hd0 = 4 * h4;
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hbuf + a_off<SL::K>(hd0 / SL::K, hd0 % SL::K));
        *reinterpret_cast<f32x4*>(out + bt * H + hd0) = hv;
      }
    }
  }
  if (ok && pair) {
    if (hT) hT[b * H + hd] = hst;
    if (cT) cT[b * H + hd] = cst;
  }
  if constexpr (DIAG) {
    if (lane == 0 && reserve && b < 8) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(reserve) + (b * FAST_NW + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = seg[i];
    }
  }
}

// ---- dispatch ------------------------------------------------------------------------------------------
template <class S>
static int launch_x3(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                     const void* bias_hid, void* out, void* hT, void* cT, float* reserve, hipStream_t stream) {
  constexpr size_t lds = x3_lds_bytes<S>();
  const float* bh = rs.has_bias_hid ? (const float*)bias_hid : (const float*)nullptr;
  const char* diag = getenv("TTRNN_DIAG");
  const bool dg = diag && diag[0] == '1' && reserve;
  auto kern = dg ? k_lstm_fwd_x3<S, true> : k_lstm_fwd_x3<S, false>;
  if (lds > 64 * 1024) {
    static bool raised[2] = {false, false};
    if (!raised[dg]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return TTRNN_ERR_LAUNCH;
      raised[dg] = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(rs.B), dim3(FAST_NT), lds, stream, rs.B, rs.T, gin, (const float*)h0,
                     (const float*)c0, packed_hid, bh, (float*)out, (float*)hT, (float*)cT, reserve);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

bool x3_rnn_fwd_available(const RnnShape& rs, int dtype) {
  if (dtype != TTRNN_F32 || rs.cell != TTRNN_LSTM || rs.B < 1 || rs.T < 1) return false;
  return shape_matches<ShpH256R8L>(rs.hid_s);
}

int launch_rnn_fwd_x3(const RnnShape& rs, GinSrc gin, const void* h0, const void* c0, const float* packed_hid,
                      const void* bias_hid, void* out, void* hT, void* cT, float* reserve, hipStream_t stream) {
  if (shape_matches<ShpH256R8L>(rs.hid_s))
    return launch_x3<ShpH256R8L>(rs, gin, h0, c0, packed_hid, bias_hid, out, hT, cT, reserve, stream);
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
