// ttrnn_fast_gemm3.hip — the batched input projection (K-in) as a GEMM on PRE-SPLIT two-piece fp16 operands (gfx950).
//
// ttrnn_fast_gemm.hip:k_gemm_split<., HALF> splits the rows of x into fp16 pieces while it stages them: every one of the
// M / 256 feature-tile workgroups that reads a row repeats the split (cfg5: 16 times), the split's VALU work and its
// ds_write_b128s sit between the MFMA clusters of the same waves, and there is one barrier per 32-k chunk — 43 % of the
// f16 matrix peak at cfg5's size (round 2).  Here
//   * one pass over x (k_split_rows: one wave per row — row maximum, power-of-two scale, the two fp16 pieces) writes the
//     operand ONCE, already in the GEMM's tile order [piece][k chunk][row][32], 4 bytes per element like the fp32 source;
//     it replaces the row-scale pass, which read all of x anyway;
//   * the GEMM itself moves no operand through registers: both operands arrive by LDS-DMA (global_load_lds_dwordx4,
//     16 KB contiguous per piece tile, 8 instructions per thread and stage), the XOR swizzle of ttrnn_split.h:x_off applied
//     to the per-lane SOURCE address (the DMA's LDS destination is lane-linear), fragment reads unchanged;
//   * 256 features x 256 rows per workgroup, wave tile 128 x 64 (24 ds_read_b128 feed 96 MFMAs per stage), two 64 KB
//     stage buffers, the DMA of stage s+1 in flight while stage s multiplies, one raw barrier per stage;
//   * terms x0w0 + x0w1 + x1w0 per product, fp32 accumulate, per-row scale of x and per-column scale of W undone in the
//     epilogue (ttrnn_fast_gemm.hip's conventions: gate-interleaved columns, one 16-byte store per lane).
// Replaces t3nsor/layers.py:121-127 -> ops.py:54-93 for the input_weights of a whole sequence (lstm.py:25).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"

// Result-destroying ablations (no output stores) exist only in the harness build of the library (`make ablation`,
// -DTTRNN_ABLATIONS -> tools/bin/libttrnn_abl.so); the shipped library compiles them out.
#ifdef TTRNN_ABLATIONS
#define G3_NOSTORE(dev) ((dev) & 2)
#else
#define G3_NOSTORE(dev) 0
#endif

namespace ttrnn {

namespace {
constexpr int G3_TF = 256, G3_TR = 256, G3_BK = 32;
constexpr int G3_PL = 256 * G3_BK;                       // fp16 elements per piece tile (16 KB)
constexpr int G3_STAGE = 4 * G3_PL;                      // W piece 0, W piece 1, x piece 0, x piece 1
constexpr size_t G3_LDS = (size_t)2 * G3_STAGE * sizeof(_Float16);      // 128 KB
constexpr size_t G3_LDS_P = G3_LDS + 2 * 768 * sizeof(float);            // + the persistent kernel's epilogue constants
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int g3_expo(float x) {        // x < 2^e; zero / non-finite: neutral (as ttrnn_fast_gemm.hip:g_expo)
  if (!(x > 0.f) || !(x <= 3.4028235e38f)) return 0;
  int e;
  frexpf(x, &e);
  return e < -100 ? -100 : e;
}
__device__ __forceinline__ float g3_unscale2(float a, float u, float v) { return a * fminf(u, v) * fmaxf(u, v); }
__device__ __forceinline__ void g3_ld8(const float* p, size_t i, f32x4& a, f32x4& b) {
  a = *reinterpret_cast<const f32x4*>(p + i);
  b = *reinterpret_cast<const f32x4*>(p + i + 4);
}
__device__ __forceinline__ void g3_ld8(const bf16_t* p, size_t i, f32x4& a, f32x4& b) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(p + i);
  a = f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xFFFF0000u), __uint_as_float(v[1] << 16),
            __uint_as_float(v[1] & 0xFFFF0000u)};
  b = f32x4{__uint_as_float(v[2] << 16), __uint_as_float(v[2] & 0xFFFF0000u), __uint_as_float(v[3] << 16),
            __uint_as_float(v[3] & 0xFFFF0000u)};
}

// x[n][K] (storage type) -> rs[n] = 2^(14 - e) with max_k |x[n][k]| < 2^e, and the two fp16 pieces of rs[n] x[n][:] as
// planes[piece][k chunk][n][32] (row count of a chunk = n_pad; k zero-filled up to 32 * KCn).  One wave per row: the row
// is read for its maximum and read again (L1 / L2) for the split.
template <typename TS>
__global__ void __launch_bounds__(256) k_split_rows(const TS* __restrict__ x, int64_t n_rows, int64_t n_pad, int K, int KCn,
                                                    float* __restrict__ rs, _Float16* __restrict__ planes) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  const int Kp = 32 * KCn;
  const size_t plane = (size_t)KCn * n_pad * 32;
  for (int64_t n = w; n < n_rows; n += nw) {
    const TS* row = x + (size_t)n * K;
    float m = 0.f;
    for (int k = 8 * lane; k < K; k += 512) {
      f32x4 a, b;
      g3_ld8(row, (size_t)k, a, b);
      m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))),
                         fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3])))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float sc = ldexpf(1.f, 14 - g3_expo(m));
    if (lane == 0) rs[n] = sc;
    for (int k = 8 * lane; k < Kp; k += 512) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
      if (k < K) g3_ld8(row, (size_t)k, a, b);              // K % 8 == 0: a group of 8 is in or out as a whole
      a = a * sc; b = b * sc;
      unsigned p0[4], p1[4];
      split_pair_h(a[0], a[1], p0[0], p1[0]);
      split_pair_h(a[2], a[3], p0[1], p1[1]);
      split_pair_h(b[0], b[1], p0[2], p1[2]);
      split_pair_h(b[2], b[3], p0[3], p1[3]);
      const size_t off = ((size_t)(k >> 5) * n_pad + n) * 32 + (k & 31);
      *reinterpret_cast<u32x4*>(planes + off) = u32x4{p0[0], p0[1], p0[2], p0[3]};
      *reinterpret_cast<u32x4*>(planes + plane + off) = u32x4{p1[0], p1[1], p1[2], p1[3]};
    }
  }
}

__device__ __forceinline__ void g3_dma16(const _Float16* src, _Float16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// y[n][m'] (fp32, row stride M) = sum_k x[n][k] W[k][m'] (+ bias): wpl = [2][KCn][M][32] (k_gemm_prep_h), xpl = [2][KCn][n_pad][32]
// (k_split_rows), scales = [column scales of W (g_rs_off(M) floats) | row scales of x]
template <typename TS, int VAR>
__global__ void __launch_bounds__(FAST_NT) k_gemm3h(int64_t n_rows, int64_t n_pad, int KCn, int M,
                                                    const _Float16* __restrict__ wpl, const _Float16* __restrict__ xpl,
                                                    const float* __restrict__ wsc, const float* __restrict__ rowsc,
                                                    const TS* __restrict__ bias, int Hb, const float* __restrict__ bias_ilv,
                                                    float* __restrict__ y, int dev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  _Float16* lds = reinterpret_cast<_Float16*>(smem3);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  // ---- tile of this workgroup: the 32 workgroups an XCD runs side by side form an 8 (feature tiles) x 4 (row tiles) block -------
  const int MT = M / G3_TF;
  const int64_t RT = n_pad / G3_TR;
  int mt_tile;
  int64_t rt_tile;
  if (MT % 8 == 0) {
    const int xcd = blockIdx.x & 7;
    const int64_t i = blockIdx.x >> 3;
    const int64_t g = (i >> 5) * 8 + xcd;
    const int within = (int)(i & 31);
    const int smc = MT / 8;
    mt_tile = (int)(g % smc) * 8 + (within & 7);
    rt_tile = (g / smc) * 4 + (within >> 3);
  } else {
    // few feature tiles (cfg4: 4): all feature tiles of a row tile on ONE XCD, so that its L2 serves the row tile's pieces of x
    // to all of them (consecutive block ids go to different XCDs: every feature tile fetched the rows from HBM again, 815 MB per
    // cfg4 layer against 420 MB of operands + output)
    const int xcd = blockIdx.x & 7;
    const int64_t i = blockIdx.x >> 3;
    rt_tile = (i / MT) * 8 + xcd;
    mt_tile = (int)(i % MT);
  }
  if (rt_tile >= RT) return;
  const int m0 = mt_tile * G3_TF;
  const int64_t n0 = rt_tile * G3_TR;
  const int wm = wave & 1, wr = wave >> 1;               // wave tile: features [128 wm, +128) x rows [64 wr, +64)

  // ---- LDS-DMA sources: LDS slot s = tid + 512 j (16 bytes) of a piece tile holds row s >> 2, k slot (s & 3) ^ g(row) — the
  // swizzle of x_off<32> moved to the source address; both j share the thread's k slot (g has period 4 in row >> 2) -----------
  const int srow = tid >> 2;
  const int sks = (tid & 3) ^ ((-(srow >> 2)) & 3);
  const size_t wplane = (size_t)KCn * M * 32, xplane = (size_t)KCn * n_pad * 32;
  const _Float16* wsrc = wpl + (size_t)(m0 + srow) * 32 + 8 * sks;
  const _Float16* xsrc = xpl + (size_t)(n0 + srow) * 32 + 8 * sks;
  auto stage_dma = [&](int buf, int kc) {
    _Float16* dst = lds + buf * G3_STAGE + wave * 512;    // this wave's 1 KB of the first half of a piece tile
    const _Float16* ws = wsrc + (size_t)kc * M * 32;
    const _Float16* xs = xsrc + (size_t)kc * n_pad * 32;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        g3_dma16(ws + p * wplane + (size_t)j * 128 * 32, dst + p * G3_PL + j * 4096);
        g3_dma16(xs + p * xplane + (size_t)j * 128 * 32, dst + (2 + p) * G3_PL + j * 4096);
      }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ri = 0; ri < 4; ++ri) acc[mi][ri] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment offsets (elements) inside a piece tile: x_off<32>(row, 8 q)
  int aoff[8], boff[4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) aoff[mi] = x_off<G3_BK>(wm * 128 + 16 * mi + c, 8 * q);
#pragma unroll
  for (int ri = 0; ri < 4; ++ri) boff[ri] = x_off<G3_BK>(wr * 64 + 16 * ri + c, 8 * q);

  if constexpr (VAR == 2) {
    // ---- ping-pong schedule: the two waves of a SIMD (waves w and w + 4) run ONE SLOT apart, so that in every slot one of
    // them multiplies (24 MFMAs = one quadrant of its tile) while the other reads its next fragments from LDS and issues the
    // DMAs; every slot ends in a workgroup barrier.  Per stage and wave:
    //   R1 A[0..3], B[0..1] | M1 A0 x B0 | R2 B[2..3], A[4..5] + the DMAs of the next stage | M2 A0 x B1 | R3 A[6..7] | M3 A1 x B1
    //   | R4 wait for the next stage's DMAs | M4 A1 x B0
    // The next stage's buffer was last read a stage ago; its DMAs have 5 slots to land; their completion (vmcnt(0) in R4)
    // is followed by two barriers before the first read of the new buffer (R1 of the earlier group).
    const int grp = wave >> 2;
    auto slot_end = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mma_block = [&](const xh8 (&af)[4][2], const xh8 (&bq)[2][2], int mi0, int ri0) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) {
          f32x4& a = acc[mi0 + mi][ri0 + ri];
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][1], bq[ri][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][0], bq[ri][1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][0], bq[ri][0], a, 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    };
    stage_dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();            // the later group starts one slot behind
    for (int kc = 0; kc < KCn; ++kc) {
      const int buf = kc & 1;
      const _Float16* Ws = lds + buf * G3_STAGE;
      const _Float16* Xs = Ws + 2 * G3_PL;
      xh8 a0[4][2], a1[4][2], b0[2][2], b1[2][2];
      // R1
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) b0[i][p] = *reinterpret_cast<const xh8*>(Xs + p * G3_PL + boff[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) a0[i][p] = *reinterpret_cast<const xh8*>(Ws + p * G3_PL + aoff[i]);
      slot_end();
      mma_block(a0, b0, 0, 0);                              // M1
      slot_end();
      // R2
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) b1[i][p] = *reinterpret_cast<const xh8*>(Xs + p * G3_PL + boff[2 + i]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) a1[i][p] = *reinterpret_cast<const xh8*>(Ws + p * G3_PL + aoff[4 + i]);
      if (kc + 1 < KCn) stage_dma(buf ^ 1, kc + 1);
      slot_end();
      mma_block(a0, b1, 0, 2);                              // M2
      slot_end();
      // R3
#pragma unroll
      for (int i = 2; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) a1[i][p] = *reinterpret_cast<const xh8*>(Ws + p * G3_PL + aoff[4 + i]);
      slot_end();
      mma_block(a1, b1, 4, 2);                              // M3
      slot_end();
      // R4
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      slot_end();
      mma_block(a1, b0, 4, 0);                              // M4
      slot_end();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();            // the earlier group leaves one slot early: same barrier count
  } else {
  stage_dma(0, 0);
  for (int kc = 0; kc < KCn; ++kc) {
    const int buf = kc & 1;
    // stage kc has landed (this wave's DMAs: vmcnt; everybody's: the barrier), and nobody reads the other buffer any more
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kc + 1 < KCn) stage_dma(buf ^ 1, kc + 1);
    const _Float16* Ws = lds + buf * G3_STAGE;
    const _Float16* Xs = Ws + 2 * G3_PL;
    xh8 bf[4][2];
#pragma unroll
    for (int ri = 0; ri < 4; ++ri)
#pragma unroll
      for (int p = 0; p < 2; ++p) bf[ri][p] = *reinterpret_cast<const xh8*>(Xs + p * G3_PL + boff[ri]);
    {
      // every fragment of the stage is requested before the first MFMA: the second half's reads land behind the first half's
      // 48 MFMAs (the compiler waits with a counted lgkmcnt)
      xh8 af[8][2];
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int p = 0; p < 2; ++p) af[mi][p] = *reinterpret_cast<const xh8*>(Ws + p * G3_PL + aoff[mi]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          f32x4& a = acc[mi][ri];
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][1], bf[ri][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][0], bf[ri][1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][0], bf[ri][0], a, 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    }
  }

  }
  // ---- epilogue: lane (c, q) of tile (mi, ri) holds features m0 + 128 wm + 16 mi + 4 q .. + 3 of row n0 + 64 wr + 16 ri + c ----
  float unr[4];
#pragma unroll
  for (int ri = 0; ri < 4; ++ri) {
    const int64_t n = n0 + wr * 64 + 16 * ri + c;
    unr[ri] = 1.0f / rowsc[n < n_rows ? n : n_rows - 1];
  }
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int mf = m0 + wm * 128 + 16 * mi + 4 * q;
    f32x4 bh = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) {
      const int hd = mf >> 2;
      bh = f32x4{ld(bias, hd), ld(bias, 2 * Hb + hd), ld(bias, Hb + hd), ld(bias, 3 * Hb + hd)};     // slots i,g,f,o
    } else if (bias_ilv) {
      bh = *reinterpret_cast<const f32x4*>(bias_ilv + mf);
    }
    const f32x4 ws4 = *reinterpret_cast<const f32x4*>(wsc + mf);
    const f32x4 unf = f32x4{1.0f / ws4[0], 1.0f / ws4[1], 1.0f / ws4[2], 1.0f / ws4[3]};
#pragma unroll
    for (int ri = 0; ri < 4; ++ri) {
      const int64_t n = n0 + wr * 64 + 16 * ri + c;
      // (TTRNN_ABLATIONS builds only, dev bit 1: nothing stores — everything is computed and dropped)
      if (n < n_rows && (!G3_NOSTORE(dev) || acc[mi][ri][0] == 12345.678f))
        *reinterpret_cast<f32x4*>(y + (size_t)n * M + mf) =
            f32x4{g3_unscale2(acc[mi][ri][0], unf[0], unr[ri]), g3_unscale2(acc[mi][ri][1], unf[1], unr[ri]),
                  g3_unscale2(acc[mi][ri][2], unf[2], unr[ri]), g3_unscale2(acc[mi][ri][3], unf[3], unr[ri])} + bh;
    }
  }
}

// ---- the same GEMM as a PERSISTENT kernel (default; option `dev` bit 6 selects k_gemm3h<., 0> for A/B) --------------------------------
// One workgroup per CU walks over the tiles vb = blockIdx.x, + gridDim.x, ... of k_gemm3h's launch grid (same tile <-> XCD
// mapping: the stride is a multiple of 8).  What the non-persistent kernel pays per tile — the wait for its 256 KB of output
// stores before the workgroup may retire, the launch of the next workgroup, the latency of its first stage's DMAs with nothing
// else running on the CU — is hidden here: the LAST stage of a tile issues the DMAs of the NEXT tile's first stage (the
// other stage buffer is free by then), the epilogue's 32 stores per lane go out behind them, and the next tile's first stage
// waits with `s_waitcnt vmcnt(32)`: vector-memory operations complete in issue order (MI355X guide, s_waitcnt), so at most
// the 32 younger stores are still in flight when the wait returns — the DMAs have landed, the stores drain under the next
// tile's first MFMAs.  (A tile with fewer than 32 stores per lane — the last row tile, the harness's no-store experiment — is
// followed by a plain vmcnt(0).)
template <typename TS, bool MT8>
__global__ void __launch_bounds__(FAST_NT) k_gemm3p(int64_t n_rows, int64_t n_pad, int KCn, int M, int nblk,
                                                    const _Float16* __restrict__ wpl, const _Float16* __restrict__ xpl,
                                                    const float* __restrict__ wsc, const float* __restrict__ rowsc,
                                                    const TS* __restrict__ bias, int Hb, const float* __restrict__ bias_ilv,
                                                    float* __restrict__ y, int dev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  _Float16* lds = reinterpret_cast<_Float16*>(smem3);
  // behind the two stage buffers: the epilogue's per-tile constants, double-buffered by tile parity — [256] inverse column
  // scales, [256] bias values (feature order), [256] inverse row scales.  They are fetched with ONE global load per thread
  // during the previous tile's last stage, so that the epilogue itself issues nothing but stores to the vector-memory queue
  // (a load between the stores would make the wave wait for every store before it)
  float* par = reinterpret_cast<float*>(smem3 + G3_LDS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int MT = M / G3_TF;
  const int RT = (int)(n_pad / G3_TR);
  // tile of virtual block vb: k_gemm3h's mapping (MT8: 8 x 4 tile blocks per XCD; few feature tiles: a row tile's feature tiles
  // on one XCD).  32-bit scalars throughout: with 64-bit tile indices the scalar registers ran out and the DMA addresses moved
  // into vector registers, which then spilled
  auto tile_of = [&](int vb, int& mt_tile, int& rt_tile) -> bool {
    const int xcd = vb & 7, i = vb >> 3;
    if constexpr (MT8) {
      const int g = (i >> 5) * 8 + xcd, within = i & 31, smc = MT >> 3;
      mt_tile = (g % smc) * 8 + (within & 7);
      rt_tile = (g / smc) * 4 + (within >> 3);
    } else {
      rt_tile = (i / MT) * 8 + xcd;
      mt_tile = i % MT;
    }
    return rt_tile < RT;
  };
  const int stride = (int)gridDim.x;
  int vb = (int)blockIdx.x;
  int mt_tile = 0, rt_tile = 0;
  while (vb < nblk && !tile_of(vb, mt_tile, rt_tile)) vb += stride;
  if (vb >= nblk) return;
  const int wm = wave & 1, wr = wave >> 1;               // wave tile: features [128 wm, +128) x rows [64 wr, +64)
  const int srow = tid >> 2;
  const unsigned lane_src = (unsigned)(srow * 32 + 8 * ((tid & 3) ^ ((-(srow >> 2)) & 3)));      // the thread's 16 bytes inside a [256][32] piece tile
  const unsigned lane_src1 = lane_src + 128u * 32u;              // rows 128 .. 255 of the piece tile
  const size_t wplane = (size_t)KCn * M * 32, xplane = (size_t)KCn * n_pad * 32;
  // (uniform tile offsets + one 32-bit lane offset: the DMAs take their base from scalar registers)
  auto stage_dma = [&](int mt, int rt, int buf, int kc) {
    _Float16* dst = lds + buf * G3_STAGE + wave * 512;
    const _Float16* ws = wpl + ((size_t)kc * M + (size_t)mt * G3_TF) * 32;
    const _Float16* xs = xpl + ((size_t)kc * n_pad + (size_t)rt * G3_TR) * 32;
    // (opaque to the optimiser: otherwise the eight per-DMA bases are hoisted out of the stage loop — sixteen scalar registers
    // per tile flavour; the scalar file overflowed into vector registers and those spilled)
    asm volatile("" : "+s"(ws), "+s"(xs));
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        g3_dma16((ws + p * wplane) + (j ? lane_src1 : lane_src), dst + p * G3_PL + j * 4096);
        g3_dma16((xs + p * xplane) + (j ? lane_src1 : lane_src), dst + (2 + p) * G3_PL + j * 4096);
      }
  };
  // the tile's epilogue constants: three loads per thread under uniform control flow only (feature f's column scale and bias,
  // row f's scale; both halves of the workgroup load the same 256 of each — divergent branches around the loads made the
  // compiler wait for each load where the branches join, i.e. inside the stage that issued it)
  auto par_load = [&](int mt, int rt, float& v0, float& v1, float& v2) {
    const unsigned f = (unsigned)tid & 255u;
    const float* wsc_t = wsc + (size_t)mt * G3_TF;
    const float* rs_t = rowsc + (size_t)rt * G3_TR;
    const int64_t left = n_rows - 1 - (int64_t)rt * G3_TR;                  // >= 0 for every tile that exists
    const unsigned last = left > 255 ? 255u : (unsigned)left;
    const unsigned mf = (unsigned)mt * G3_TF + f, hd = mf >> 2, sl = mf & 3u;       // gate-interleaved columns: slots i,g,f,o of unit hd
    const unsigned bi = (sl == 0 ? 0u : sl == 1 ? 2u * Hb : sl == 2 ? (unsigned)Hb : 3u * Hb) + hd;
    v0 = wsc_t[f];
    v1 = rs_t[f < last ? f : last];
    v2 = 0.f;
    if (bias) v2 = ld(bias, bi);
    else if (bias_ilv) v2 = bias_ilv[mf];
  };
  // (the reciprocals are taken HERE, a stage after the loads were issued)
  auto par_store = [&](int slot, float v0, float v1, float v2) {
    float* ps = par + slot * 768;
    if (tid < 256) {
      ps[tid] = 1.0f / v0;                                         // [0, 256): inverse column scales
      ps[512 + tid] = 1.0f / v1;                                   // [512, 768): inverse row scales
    } else {
      ps[tid] = v2;                                                // [256, 512): bias
    }
  };
  // fragment offsets: x_off<32>(row + 16 i, 8 q) = x_off<32>(row, 8 q) + 512 i (the swizzle has period 4 in row >> 2), so one
  // register per operand and immediate offsets in the reads (the twelve separate offsets of k_gemm3h cost this kernel spills —
  // and a spill reload is a vector-memory operation whose wait would also wait for the previous tile's stores)
  const int aoff0 = x_off<G3_BK>(wm * 128 + c, 8 * q), boff0 = x_off<G3_BK>(wr * 64 + c, 8 * q);

  // (Tried on top, no gain: starting the workgroups of a CU quadruple a quarter of a tile apart, so that the chip does not
  // alternate between 256 CUs multiplying and 256 CUs storing — 2.89 against 2.82 ms at cfg5's size; non-temporal stores — 2.87.)
  {
    float v0, v1, v2;
    par_load(mt_tile, rt_tile, v0, v1, v2);
    par_store(0, v0, v1, v2);                                        // visible behind the first stage's barrier
  }
  stage_dma(mt_tile, rt_tile, 0, 0);
  bool counted = false;                 // the previous tile of this workgroup left exactly 32 stores per lane behind its DMAs
  int slot = 0;
  for (;;) {
    const int m0 = mt_tile * G3_TF;
    const int64_t n0 = (int64_t)rt_tile * G3_TR;
    // the next tile of this workgroup (its first stage and its constants are requested during this tile's last stage)
    int nvb = vb + stride;
    int nmt = 0, nrt = 0;
    bool has_next = false;
    while (nvb < nblk && !(has_next = tile_of(nvb, nmt, nrt))) nvb += stride;

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ri = 0; ri < 4; ++ri) acc[mi][ri] = f32x4{0.f, 0.f, 0.f, 0.f};
    float nv0 = 1.f, nv1 = 1.f, nv2 = 0.f;
    for (int kc = 0; kc < KCn; ++kc) {
      const int buf = kc & 1;
      // stage kc has landed (this wave's DMAs: vmcnt; everybody's: the barrier), and nobody reads the other buffer any more
      if (kc == 0 && counted) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kc + 1 < KCn) {
        stage_dma(mt_tile, rt_tile, buf ^ 1, kc + 1);
        if (kc + 2 == KCn && has_next) par_load(nmt, nrt, nv0, nv1, nv2);      // the next tile's constants: back by the last stage
      } else if (has_next) {
        stage_dma(nmt, nrt, buf ^ 1, 0);                 // KCn is even: buf ^ 1 == 0, where every tile starts; only stores follow
      }
      const _Float16* Ws = lds + buf * G3_STAGE;
      const _Float16* Xs = Ws + 2 * G3_PL;
      xh8 bf[4][2];
#pragma unroll
      for (int ri = 0; ri < 4; ++ri)
#pragma unroll
        for (int p = 0; p < 2; ++p) bf[ri][p] = *reinterpret_cast<const xh8*>(Xs + p * G3_PL + boff0 + 512 * ri);
      xh8 af[8][2];
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int p = 0; p < 2; ++p) af[mi][p] = *reinterpret_cast<const xh8*>(Ws + p * G3_PL + aoff0 + 512 * mi);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          f32x4& a = acc[mi][ri];
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][1], bf[ri][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][0], bf[ri][1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][0], bf[ri][0], a, 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    }
    if (has_next) par_store(slot ^ 1, nv0, nv1, nv2);          // read two barriers from now (the next tile's epilogue)
    // ---- epilogue: lane (c, q) of tile (mi, ri) holds features m0 + 128 wm + 16 mi + 4 q .. + 3 of row n0 + 64 wr + 16 ri + c ----
    const float* ps = par + slot * 768;
    float unr[4];
#pragma unroll
    for (int ri = 0; ri < 4; ++ri) unr[ri] = ps[512 + wr * 64 + 16 * ri + c];
    const bool full = n0 + G3_TR <= n_rows && !G3_NOSTORE(dev);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int fl = wm * 128 + 16 * mi + 4 * q;
      const int mf = m0 + fl;
      const f32x4 unf = *reinterpret_cast<const f32x4*>(ps + fl);
      const f32x4 bh = *reinterpret_cast<const f32x4*>(ps + 256 + fl);
#pragma unroll
      for (int ri = 0; ri < 4; ++ri) {
        const int64_t n = n0 + wr * 64 + 16 * ri + c;
        if (n < n_rows && (!G3_NOSTORE(dev) || acc[mi][ri][0] == 12345.678f)) {
          const f32x4 v = f32x4{g3_unscale2(acc[mi][ri][0], unf[0], unr[ri]), g3_unscale2(acc[mi][ri][1], unf[1], unr[ri]),
                                g3_unscale2(acc[mi][ri][2], unf[2], unr[ri]), g3_unscale2(acc[mi][ri][3], unf[3], unr[ri])} + bh;
          *reinterpret_cast<f32x4*>(y + (size_t)n * M + mf) = v;
        }
      }
    }
    if (!has_next) break;
    counted = full;
    vb = nvb; mt_tile = nmt; rt_tile = nrt;
    slot ^= 1;
  }
}

size_t al256g3(size_t v) { return (v + 255) & ~(size_t)255; }
int g3_chunks(int K) { return ((K + 2 * G3_BK - 1) / (2 * G3_BK)) * 2; }          // = ttrnn_fast_gemm.hip:gemm_chunks
int64_t g3_npad(int64_t n_rows) { return (n_rows + G3_TR - 1) / G3_TR * G3_TR; }
}  // namespace

// the pre-split GEMM takes over where 256-feature tiles fill the chip (the same threshold as k_gemm_split<., HALF, 2>)
bool gemm3_ok(int64_t n_rows, int K, int M) {
  return !opt(OPT_NO_GEMM3) && K >= 8 && K % 8 == 0 && M % G3_TF == 0 && (M / G3_TF) * ((n_rows + G3_TR - 1) / G3_TR) >= 1024;
}
// workspace of the pieces of x: two fp16 planes over the padded rows
size_t gemm3_xplane_bytes(int64_t n_rows, int K) {
  return al256g3((size_t)2 * g3_chunks(K) * G3_BK * (size_t)g3_npad(n_rows) * sizeof(_Float16));
}

// planes / scratch: as launch_gemm_half (launch_gemm_half_prep has filled the W planes and the column scales); xplanes:
// gemm3_xplane_bytes
int launch_gemm3h(int dtype, int64_t n_rows, int K, int M, const void* x, const void* planes, void* scratch, void* xplanes,
                  const void* bias, int Hb, float* y, hipStream_t stream, const float* bias_ilv) {
  if (n_rows <= 0) return TTRNN_OK;
  const int KCn = g3_chunks(K);
  const int64_t n_pad = g3_npad(n_rows);
  float* wsc = (float*)scratch;
  float* rs = wsc + (((size_t)M * sizeof(float) + 255) & ~(size_t)255) / sizeof(float);      // = g_rs_off(M)
  const int grid_s = (int)((n_rows + 3) / 4 < 4096 ? (n_rows + 3) / 4 : 4096);
  if (dtype == TTRNN_F32)
    hipLaunchKernelGGL(k_split_rows<float>, dim3(grid_s), dim3(256), 0, stream, (const float*)x, n_rows, n_pad, K, KCn, rs,
                       (_Float16*)xplanes);
  else
    hipLaunchKernelGGL(k_split_rows<bf16_t>, dim3(grid_s), dim3(256), 0, stream, (const bf16_t*)x, n_rows, n_pad, K, KCn, rs,
                       (_Float16*)xplanes);
  const int MT = M / G3_TF;
  const int64_t RT = n_pad / G3_TR;
  int64_t grid;
  if (MT % 8 == 0) {
    const int64_t supers = (int64_t)(MT / 8) * ((RT + 3) / 4);
    grid = ((supers + 7) / 8) * 8 * 32;
  } else {
    grid = 8 * ((RT + 7) / 8) * (int64_t)MT;
  }
  // dev bit 0 (harness A/B): the ping-pong schedule.  Both schedules land within 3 % of each other (tools/gemm3_bench:
  // 2.25 / 2.31 ms without the output stores at cfg5's size = the 56-59 % of the f16 peak the chip sustains on random operands),
  // the lockstep one is ahead once the 2.1 GB of output stores are in
  const bool var1 = (opt(OPT_DEV) & 1) == 0;
#define G3_LAUNCH(TSV, VARV)                                                                                               \
  do {                                                                                                                    \
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_gemm3h<TSV, VARV>), G3_LDS) != TTRNN_OK) return TTRNN_ERR_LAUNCH; \
    hipLaunchKernelGGL((k_gemm3h<TSV, VARV>), dim3((unsigned)grid), dim3(FAST_NT), G3_LDS, stream, n_rows, n_pad, KCn, M,  \
                       (const _Float16*)planes, (const _Float16*)xplanes, (const float*)wsc, (const float*)rs,            \
                       (const TSV*)bias, Hb, bias_ilv, y, opt(OPT_DEV));                                                  \
  } while (0)
  // default: the persistent kernel (k_gemm3p), one workgroup per CU; dev bit 6: one workgroup per tile (A/B)
  if (var1 && !(opt(OPT_DEV) & 64) && grid < (int64_t)1 << 30) {
    const int64_t pgrid = grid < (int64_t)device_cu_count() ? grid : (int64_t)device_cu_count() / 8 * 8;
#define G3_LAUNCH_P(TSV, MT8V)                                                                                                  \
  do {                                                                                                                    \
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_gemm3p<TSV, MT8V>), G3_LDS_P) != TTRNN_OK) return TTRNN_ERR_LAUNCH;   \
    hipLaunchKernelGGL((k_gemm3p<TSV, MT8V>), dim3((unsigned)pgrid), dim3(FAST_NT), G3_LDS_P, stream, n_rows, n_pad, KCn, M, (int)grid, \
                       (const _Float16*)planes, (const _Float16*)xplanes, (const float*)wsc, (const float*)rs,            \
                       (const TSV*)bias, Hb, bias_ilv, y, opt(OPT_DEV));                                                  \
  } while (0)
    if (dtype == TTRNN_F32) { if (MT % 8 == 0) G3_LAUNCH_P(float, true); else G3_LAUNCH_P(float, false); }
    else { if (MT % 8 == 0) G3_LAUNCH_P(bf16_t, true); else G3_LAUNCH_P(bf16_t, false); }
#undef G3_LAUNCH_P
    return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  }
  if (dtype == TTRNN_F32) { if (var1) G3_LAUNCH(float, 0); else G3_LAUNCH(float, 2); }
  else { if (var1) G3_LAUNCH(bf16_t, 0); else G3_LAUNCH(bf16_t, 2); }
#undef G3_LAUNCH
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

}  // namespace ttrnn
