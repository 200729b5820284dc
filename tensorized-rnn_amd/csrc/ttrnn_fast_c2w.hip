// ttrnn_fast_c2w.hip — weight gradients of a recurrent layer's TT-matrices THROUGH THE CHAIN (plan and index conventions:
// ttrnn_c2w.h).  Replaces, for the low-rank shapes of the reference's speaker-verification encoder, the autograd of the two
// TTLinear calls of a cell over all B*T rows (tensorized_rnn/lstm.py:23-26, gru.py:33-36 -> t3nsor/layers.py:121-127 ->
// t3nsor/ops.py:78-93): dense in x out gradients cost 4.8 x the chain's FLOPs at H = 768, d = 2, r = 2 and were 38 % of that
// shape's training step.
//
// One persistent workgroup per CU walks blocks of nb rows.  Per block: x and dy are split into two fp16 pieces under the launch's
// power-of-two scales and staged ONCE (dy for both matrices); phases A, B (row-local GEMMs: C1, dC1 into LDS images, rescaled by
// their bounds and split again) and C, D (the gradient GEMMs, contraction over rows x a mode, accumulators in registers for the
// whole launch).  k-strided operands come out of the row-major images through gfx950's transposing LDS read
// (ds_read_b64_tr_b16), so every image exists once.  Partial sums leave as per-workgroup slabs in MFMA fragment order and are
// added in a fixed order by k_c2_reduce: gradients are bitwise repeatable.
#include "ttrnn.h"
#include "ttrnn_c2w.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
#include "ttrnn_split.h"
#include <type_traits>

namespace ttrnn {
namespace {

typedef short c2_s16x4 __attribute__((ext_vector_type(4)));
typedef short c2_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) c2_s16x4 c2_lds_s16x4;

__device__ __forceinline__ int c2_expo(float x) {           // x < 2^e; zero / non-finite: neutral; clamped (scales stay normal)
  if (!(x > 0.f) || !(x <= 3.4028235e38f)) return 0;
  int e;
  frexpf(x, &e);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

// ---- max |a| over a buffer (bit pattern, atomicMax): only where the caller has no bound to hand over ----------------------------
__global__ void __launch_bounds__(256) k_c2_absmax(const float* __restrict__ a, size_t n4, unsigned* __restrict__ out) {
  __shared__ float red[256];
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a)[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(out, __float_as_uint(red[0]));
}

// ---- merged cores -------------------------------------------------------------------------------------------------------------
// packed core k: W_k[(j*R_{k+1} + b)*M_k + i*R_k + a], M_k = I_k*R_k  (a: left rank, b: right rank)
__device__ __forceinline__ float c2_core(const TtShape& s, const float* packed, int k, int a, int i, int j, int b) {
  return packed[s.woff[k] + (size_t)(j * s.R[k + 1] + b) * s.M[k] + i * s.R[k] + a];
}

// Gh[ih][jh][a] (cores 0 .. s-1), Gt[it][jt][a] (cores s .. d-1): a side has one or two cores (c2_plan).  One thread per entry.
__global__ void __launch_bounds__(256) k_c2_merge(TtShape s, C2Mat m, const float* __restrict__ packed, float* __restrict__ Gh,
                                                  float* __restrict__ Gt) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nh = m.head_elems, nt = m.tail_elems;
  if (t < nh) {
    const int a = (int)(t % m.R);
    const long e = t / m.R;
    const int jh = (int)(e % m.Jh), ih = (int)(e / m.Jh);
    float v;
    if (m.s == 1) {
      v = c2_core(s, packed, 0, 0, ih, jh, a);
    } else {      // two cores: sum over the rank between them
      const int i1 = ih % s.I[1], i0 = ih / s.I[1], j1 = jh % s.J[1], j0 = jh / s.J[1];
      v = 0.f;
      for (int r = 0; r < s.R[1]; ++r) v = fmaf(c2_core(s, packed, 0, 0, i0, j0, r), c2_core(s, packed, 1, r, i1, j1, a), v);
    }
    Gh[t] = v;
  } else if (t < nh + nt) {
    const long u = t - nh;
    const int a = (int)(u % m.R);
    const long e = u / m.R;
    const int jt = (int)(e % m.Jt), it = (int)(e / m.Jt);
    float v;
    if (s.d - m.s == 1) {
      v = c2_core(s, packed, m.s, a, it, jt, 0);
    } else {
      const int k0 = m.s, k1 = m.s + 1;
      const int i1 = it % s.I[k1], i0 = it / s.I[k1], j1 = jt % s.J[k1], j0 = jt / s.J[k1];
      v = 0.f;
      for (int r = 0; r < s.R[k1]; ++r) v = fmaf(c2_core(s, packed, k0, a, i0, j0, r), c2_core(s, packed, k1, r, i1, j1, 0), v);
    }
    Gt[u] = v;
  }
}

// ---- scale header + weight fragments ------------------------------------------------------------------------------------------
// hdr (ints): [0] ex  max|x| < 2^ex        [1] ed  max|dy| < 2^ed       [2] egt max|Gt| < 2^egt     [3] egh max|Gh| < 2^egh
//             [4] ec1 |C1| <= L1(Gt rows) max|x| < 2^ec1                [5] edc |dC1| <= L1(Gh^T rows) max|dy| < 2^edc
// GtF: A operand of phase A, tile (pt, kb): lane (c, g) holds Gt'[p = 16 pt + c][j_t = 32 kb + 8 g + 0..7], two planes
// GhF: A operand of phase B, tile (qt, kb): lane (c, g) holds Gh'[i_h = 32 kb + 8 g + 0..7][q = 16 qt + c]
// One workgroup per matrix.
struct C2Prep {
  C2Mat m;
  const float* Gh; const float* Gt;
  const unsigned* x_cmax; int x_n;       // bounds of x: n entries (bit patterns), the maximum is taken
  const unsigned* dy_cmax; int dy_n;
  int* hdr; _Float16* gtf; _Float16* ghf;
};
struct C2PrepArgs { C2Prep p[2]; };

__device__ float c2_block_max(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// Workgroup 0 of a matrix writes the header; workgroup 1 + t builds fragment tile t (tail tiles first), taking the maximum it
// needs over the merged core itself (a single workgroup doing everything was a chain of sixteen dependent load -> split -> store
// rounds: 28 us).  blockIdx.y = matrix.
__global__ void __launch_bounds__(256) k_c2_prep(C2PrepArgs args) {
  __shared__ float red[256];
  const C2Prep& a = args.p[blockIdx.y];
  const C2Mat& m = a.m;
  const int tid = threadIdx.x;
  const int ntt = m.PT * m.KA, nth = m.QT * m.KB;
  if ((int)blockIdx.x > ntt + nth) return;
  if (blockIdx.x == 0) {
    float mx = 0.f, md = 0.f, mgt = 0.f, mgh = 0.f, l1t = 0.f, l1h = 0.f;
    for (int i = tid; i < a.x_n; i += 256) mx = fmaxf(mx, __uint_as_float(a.x_cmax[i]));
    for (int i = tid; i < a.dy_n; i += 256) md = fmaxf(md, __uint_as_float(a.dy_cmax[i]));
    for (long i = tid; i < m.tail_elems; i += 256) mgt = fmaxf(mgt, fabsf(a.Gt[i]));
    for (long i = tid; i < m.head_elems; i += 256) mgh = fmaxf(mgh, fabsf(a.Gh[i]));
    for (int p = tid; p < m.P; p += 256) {            // row p = (a, i_t) of Gt: sum over j_t
      const int ra = p / m.It, it = p % m.It;
      float s = 0.f;
      for (int jt = 0; jt < m.Jt; ++jt) s += fabsf(a.Gt[((long)it * m.Jt + jt) * m.R + ra]);
      l1t = fmaxf(l1t, s);
    }
    for (int q = tid; q < m.Jh * m.R; q += 256) {      // row (j_h, a) of Gh^T: sum over i_h
      const int jh = q / m.R, ra = q % m.R;
      float s = 0.f;
      for (int ih = 0; ih < m.Ih; ++ih) s += fabsf(a.Gh[((long)ih * m.Jh + jh) * m.R + ra]);
      l1h = fmaxf(l1h, s);
    }
    mx = c2_block_max(mx, red); md = c2_block_max(md, red);
    mgt = c2_block_max(mgt, red); mgh = c2_block_max(mgh, red);
    l1t = c2_block_max(l1t, red); l1h = c2_block_max(l1h, red);
    if (tid == 0) {
      a.hdr[0] = c2_expo(mx); a.hdr[1] = c2_expo(md); a.hdr[2] = c2_expo(mgt); a.hdr[3] = c2_expo(mgh);
      a.hdr[4] = c2_expo(l1t * mx); a.hdr[5] = c2_expo(l1h * md);
    }
    return;
  }
  const int tile = (int)blockIdx.x - 1;
  const bool tail = tile < ntt;
  float mg = 0.f;
  if (tail) for (long i = tid; i < m.tail_elems; i += 256) mg = fmaxf(mg, fabsf(a.Gt[i]));
  else for (long i = tid; i < m.head_elems; i += 256) mg = fmaxf(mg, fabsf(a.Gh[i]));
  mg = c2_block_max(mg, red);
  const float sg = ldexpf(1.f, 14 - c2_expo(mg));
  for (int e = tid; e < 512; e += 256) {
    const int j = e & 7, lane = e >> 3;
    float v = 0.f;
    _Float16* dst;
    if (tail) {
      const int kb = tile % m.KA, pt = tile / m.KA;
      const int p = 16 * pt + (lane & 15), jt = 32 * kb + 8 * (lane >> 4) + j;
      if (p < m.P && jt < m.Jt) v = a.Gt[((long)(p % m.It) * m.Jt + jt) * m.R + p / m.It] * sg;
      dst = a.gtf + (size_t)tile * 1024;                 // [(tile*2 + plane)][lane][8]
    } else {
      const int th = tile - ntt;
      const int kb = th % m.KB, qt = th / m.KB;
      const int q = 16 * qt + (lane & 15), ih = 32 * kb + 8 * (lane >> 4) + j;
      const int ra = q / m.JhP, jh = q % m.JhP;
      if (q < m.Q && jh < m.Jh && ih < m.Ih) v = a.Gh[((long)ih * m.Jh + jh) * m.R + ra] * sg;
      dst = a.ghf + (size_t)th * 1024;
    }
    _Float16 p0, p1;
    split2h(v, p0, p1);
    dst[lane * 8 + j] = p0;
    dst[512 + lane * 8 + j] = p1;
  }
}

// ---- the chain weight-gradient kernel --------------------------------------------------------------------------------------------
struct C2MatArgs {
  const float* x;            // rows of the operand: [n_rows][in] (T == 0) or the layer's outputs read one step back (T > 0)
  const float* first;        // T > 0: [B][in] rows for t = 0 (NULL: zeros)
  int T;
  const int* hdr;
  const _Float16* gtf;
  const _Float16* ghf;
  float* part;               // [grid][NU][64] f32x4
};
struct C2Args {
  C2Plan pl;
  C2MatArgs a[2];
  const float* dy;           // [n_rows][OUT]
  float* bpart;              // [grid][OUT] or NULL
  long n_rows;
  unsigned long long* diag;  // -DTTRNN_ABLATIONS builds only: [2 waves][8] cycle sums per segment of the block loop (workgroup 0, waves 0 and 5)
  int abl;                   // -DTTRNN_ABLATIONS builds only (tools/c2w_bench.py): option dev2 >> 16 — 1: no phases A / B, 2: no phases C / D,
                             // 4: no staging stores, 8: no global loads (result-destroying; 0 in libttrnn.so)
};

__device__ __forceinline__ xh8 c2_ld8(const _Float16* p) { return *reinterpret_cast<const xh8*>(p); }

// B operand B[k = k0 + 8 g + j][n = n0 + c] out of a row-major [k][n] fp16 image with row stride S (halves): `p` points at
// element (k0 + 8 g + qq, n0 + 4 pp) of the lane (qq = (lane & 15) >> 2, pp = lane & 3)
__device__ __forceinline__ xh8 c2_tr8(const _Float16* p, int S) {
  const c2_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c2_lds_s16x4*)(p));
  const c2_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c2_lds_s16x4*)(p + 4 * S));
  const c2_s16x8 v = c2_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(xh8, v);
}

__device__ __forceinline__ f32x4 c2_mma3(const xh8 a0, const xh8 a1, const xh8 b0, const xh8 b1, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc, 0, 0, 0);
  return acc;
}

// four consecutive halves of both planes <- four fp32 values (8-byte stores)
__device__ __forceinline__ void c2_store4(_Float16* p0, _Float16* p1, f32x4 v) {
  unsigned a0, b0, a1, b1;
  split_pair_h(v[0], v[1], a0, b0);
  split_pair_h(v[2], v[3], a1, b1);
  *reinterpret_cast<u32x2*>(p0) = u32x2{a0, a1};
  *reinterpret_cast<u32x2*>(p1) = u32x2{b0, b1};
}

// Kernel variants: NW waves per workgroup; NACC accumulator units per wave and matrix; WA x KAM register slots of tail
// fragments; EQ / XQ staged quads of dy / x per thread and block.
// SPEC != 0: the plan is a compile-time constant (ttrnn_c2w.h: c2_const_plan — the speaker encoder's shapes): every tile count,
// stride and LDS offset below is a literal, the loops unroll and no plan field sits in an SGPR (the run-time plan's ~150 fields
// spilled 90 SGPRs into VGPR lanes: a v_readlane in front of every use).  SPEC == 0: any plan, at run time.
template <int SPEC, int NMAT, int NW, int NACC, int WA, int KAM, int EQ, int XQ>
__global__ void __launch_bounds__(NW * 64) k_c2w(C2Args g) {
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char c2_smem[];
  constexpr C2Plan kpl = c2_const_plan<SPEC ? SPEC : 2>();
  static_assert(kpl.ok == 1 && (SPEC == 0 || kpl.nmat == NMAT), "compile-time plan");
  const C2Plan& pl = SPEC ? kpl : g.pl;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < NW);
  const int c = lane & 15, gq = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int nb = pl.nb, OUT = pl.OUT, DS = pl.DS;
  const long nblk = (g.n_rows + nb - 1) / nb;

  // ---- LDS: zero everything once (padding rows / columns are never written afterwards: they stay finite zeros) ----
  for (int i = tid; i < pl.lds / 16; i += NT) reinterpret_cast<f32x4*>(c2_smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  _Float16* dYs = reinterpret_cast<_Float16*>(c2_smem + pl.l_dy);
  const int dpl = pl.dy_rows * DS;                              // plane size (halves)

  // ---- tables (store offsets: looked up AFTER the MFMAs, off the critical loads) + per-thread constants -----------------
  xh8 gta[NMAT][WA][KAM][2];
  int moffA[NMAT][WA];
  int uA[NMAT][NACC], uB[NMAT][NACC];
  f32x4 acc[NMAT][NACC];
  float sxf[NMAT];
  int eA[NMAT], eB[NMAT];
  float sdf = 1.f;
#pragma unroll
  for (int mi = 0; mi < NMAT; ++mi) {
    const C2Mat& m = pl.m[mi];
    int* tab = reinterpret_cast<int*>(c2_smem + m.l_tab);
    int* a_ = tab;                   // [NA*16]  C1 image offset of column n of phase A (-1: past the block's rows)
    int* bm = a_ + m.NA * 16;        // [QT*4]   dC1 image offset of the four rows m0 .. m0+3 of phase B (-1: padding)
    int* bd = bm + m.QT * 4;         // [NB]     dy image offset of the column tile
    int* bs = bd + m.NB;             // [NB]     dC1 image offset of the column tile
    for (int n = tid; n < m.NA * 16; n += NT) a_[n] = n < nb * m.JhP ? ((n / m.JhP) * m.QR + (n % m.JhP)) * m.CS1 : -1;
    for (int e = tid; e < m.QT * 4; e += NT) {
      const int m0 = 16 * (e >> 2) + 4 * (e & 3);
      bm[e] = m0 < m.Q ? (m0 / m.JhP) * m.It * m.CS2 + (m0 % m.JhP) : -1;
    }
    for (int e = tid; e < m.NB; e += NT) {
      const int n0 = 16 * e, rs = n0 / m.It, it0 = n0 % m.It;
      bd[e] = rs * pl.Ih * DS + it0;
      bs[e] = (rs * m.PR + it0) * m.CS2;
    }
    // head fragments -> LDS (fragment order: linear copy)
    {
      _Float16* dst = reinterpret_cast<_Float16*>(c2_smem + m.l_ghf);
      const int n16 = m.QT * m.KB * 2 * 64;      // 16-byte units
      for (int i = tid; i < n16; i += NT)
        reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(g.a[mi].ghf)[i];
    }
    // tail fragments -> registers (tile pt = wave % PT + PT wa; the wave groups w / PT share the column tiles)
#pragma unroll
    for (int wa = 0; wa < WA; ++wa) {
      const int pt = m.PT < NW ? (wave < m.PT * m.NAG ? wave % m.PT : m.PT) : wave + NW * wa;
      const int m0 = 16 * pt + 4 * gq;
      moffA[mi][wa] = (pt < m.PT && m0 < m.P) ? (m0 / m.It) * m.JhP * m.CS1 + (m0 % m.It) : -1;
#pragma unroll
      for (int kb = 0; kb < KAM; ++kb)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          xh8 v = {};
          if (pt < m.PT && kb < m.KA) v = *reinterpret_cast<const xh8*>(g.a[mi].gtf + ((size_t)((pt * m.KA + kb) * 2 + p) * 64 + lane) * 8);
          asm volatile("" : "+v"(v));      // pinned: under register pressure the compiler re-loaded these loop invariants from
          gta[mi][wa][kb][p] = v;          // global memory INSIDE the block loop (and every use then waited for all loads in flight)
        }
    }
    // accumulator slots: i < SC = ceil(nC / NW): C tile u = wave + NW i (tiles (mt, qt)); then D tile v = wave + NW (i - SC) (tiles
    // (pt, jt)) — a slot's KIND is the same in every wave (a literal under a compile-time plan), only its validity is per wave
    const int SC = c2_ceil(m.nC, NW);
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < SC) {
        const int u = wave + NW * i;
        uA[mi][i] = u < m.nC ? (16 * (u / m.QT) + c) * DS + 8 * gq : -1;                  // dy image: row i_h = 16 mt + c, chunk gq of a k-block
        uB[mi][i] = (16 * (u % m.QT) + c) * m.CS1 + 8 * gq;                              // C1 image: row q = 16 qt + c
      } else {
        const int v = wave + NW * (i - SC);
        uA[mi][i] = v < m.nD ? (16 * (v / m.JtT) + c) * m.CS2 : -1;                       // dC1 image: row p = 16 pt + c
        uB[mi][i] = (8 * gq + qq) * m.XS + 16 * (v % m.JtT) + 4 * pp;                    // transposed read of the x image: (row 8 gq + qq, columns 4 pp ..)
      }
    }
    const int* h = g.a[mi].hdr;
    sxf[mi] = ldexpf(1.f, 14 - h[0]);
    sdf = ldexpf(1.f, 14 - h[1]);                               // (the same dy, the same bound for both matrices)
    eA[mi] = h[2] + h[0] - h[4] - 14;
    eB[mi] = h[3] + h[1] - h[5] - 14;
  }
  __syncthreads();

  // ---- staging plan: dy quads id = tid + NT e < nb*OUT/4; x quads id < nb*in/4 ---------------------------------------
  const int oq = OUT / 4;
  int dyo[EQ], dyg[EQ], dyr[EQ];         // LDS offset / offset from the block's first row (-1: no such quad) / row of the block
  f32x4 sd[EQ], dbs[EQ];
#pragma unroll
  for (int e = 0; e < EQ; ++e) {
    const int id = tid + NT * e;
    const bool on = id < nb * oq;
    const int row = on ? id / oq : 0, o = on ? 4 * (id % oq) : 0;
    dyg[e] = on ? row * OUT + o : -1;
    dyr[e] = row;
    dyo[e] = (row * pl.Ih + o / pl.It) * DS + (o % pl.It);
    dbs[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    sd[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // x quads: row of the block, column; (sample, step) of the row carried from block to block (no division in the loop)
  int xo[NMAT][XQ], xcol[NMAT][XQ], xr[NMAT][XQ];
  unsigned xb[NMAT][XQ], xt[NMAT][XQ];
  f32x4 sx[NMAT][XQ];
  // a workgroup walks a CONTIGUOUS range of blocks: consecutive blocks share a page (24 KB of dy per block at the speaker
  // encoder's size: a stride of gridDim.x blocks = 6 MB put every load of every block on a page of its own — a TLB miss per issue
  // point, ~2 000 cycles each in the stamps)
  const long per = (nblk + gridDim.x - 1) / gridDim.x;
  const long blk0 = (long)blockIdx.x * per, blk1 = blk0 + per < nblk ? blk0 + per : nblk;
  const unsigned step = (unsigned)nb;
#pragma unroll
  for (int mi = 0; mi < NMAT; ++mi) {
    const C2Mat& m = pl.m[mi];
    const int iq = m.xelem ? m.in : m.in / 4;          // staged items of a row: quads, or single elements (xelem)
    const int iw = m.xelem ? 1 : 4;
#pragma unroll
    for (int e = 0; e < XQ; ++e) {
      const int id = tid + NT * e;
      const bool on = id < nb * iq;
      const int row = on ? id / iq : 0, col = on ? iw * (id % iq) : 0;
      xr[mi][e] = on ? row : -1;
      xcol[mi][e] = col;
      xo[mi][e] = (row * m.JhP + col / m.Jt) * m.XS + (col % m.Jt);
      sx[mi][e] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned n = (unsigned)blk0 * (unsigned)nb + (unsigned)row;
      const unsigned T = g.a[mi].T > 0 ? (unsigned)g.a[mi].T : 1u;
      xb[mi][e] = n / T;
      xt[mi][e] = n - xb[mi][e] * T;
    }
  }
  unsigned stq[NMAT], str[NMAT], bmaxs[NMAT];          // step = stq T + str; index of the last sample
#pragma unroll
  for (int mi = 0; mi < NMAT; ++mi) {
    const unsigned T = g.a[mi].T > 0 ? (unsigned)g.a[mi].T : 1u;
    stq[mi] = step / T;
    str[mi] = step - stq[mi] * T;
    // (computed ONCE: inside load_block this 64-bit division — a branchy ~150-instruction expansion the compiler does not hoist —
    // ran per matrix and block, 600 cycles each in the stamps of k_c2r)
    unsigned bm = (unsigned)(g.n_rows / (long)T) - 1u;
    asm volatile("" : "+s"(bm));
    bmaxs[mi] = bm;
  }
  const bool want_bias = g.bpart != nullptr;

  // loads of block `blk` (the rows' (sample, step) are in xb / xt); advances them to the block after
  // The loads of a block are ISSUED at `NPT` points spread over the previous block's phases (point = the quads e with
  // e % NPT == point; x rides with point 1): issued in one burst by every CU at once they fill the memory pipeline's queues and
  // the issuing waves stall for the burst's whole transfer time (stamps: 5 000 of 17 000 cycles per block).
  constexpr int NPT = 4;
  // Every load is UNCONDITIONAL (clamped to a valid address; what must not count is zeroed by the `keep` factor at the store):
  // behind an exec-masked branch or a zero-initialised select the compiler's wait-count pass put s_waitcnt vmcnt(0) in front of
  // every load of the loop, so the five loads of a block went out one memory round trip after the other — 5 000 of a block's
  // 17 000 cycles in the stamps.
  float dk[EQ], xk[NMAT][XQ];
#pragma unroll
  for (int e = 0; e < EQ; ++e) dk[e] = 0.f;
#pragma unroll
  for (int mi = 0; mi < NMAT; ++mi)
#pragma unroll
    for (int e = 0; e < XQ; ++e) xk[mi][e] = 0.f;
  // (also no `if (more)` around the loop's loads and stores: two ifs on one condition are two unrelated paths to the wait-count
  // pass — "issued but never consumed" put vmcnt(0) in front of every address computation; the iteration of a range's last block
  // re-reads that block with keep = 0 instead)
  auto load_block = [&](long blk, int point, bool live = true) {
    const long n0 = blk * nb;
    const long last = live ? g.n_rows - 1 : -1;      // (not live: every row counts as past the end)
    const long lastv = g.n_rows - 1;
#pragma unroll
    for (int e = 0; e < EQ; ++e)
      if (point < 0 || e % NPT == point) {
        const int off = dyg[e] >= 0 ? dyg[e] : 0, row = dyg[e] >= 0 ? dyr[e] : 0;
        const bool in = n0 + row <= last;
        sd[e] = *reinterpret_cast<const f32x4*>(g.dy + (size_t)(in ? n0 : lastv) * OUT + (in ? off : off - row * OUT));
        dk[e] = (in && dyg[e] >= 0) ? 1.f : 0.f;
      }
    if (point >= 0 && point != 1) return;
#pragma unroll
    for (int mi = 0; mi < NMAT; ++mi) {
      const C2Mat& m = pl.m[mi];
      const C2MatArgs& a = g.a[mi];
#pragma unroll
      for (int e = 0; e < XQ; ++e) {
        const int row = xr[mi][e] >= 0 ? xr[mi][e] : 0;
        const long n = n0 + row;
        const bool in = n <= last && xr[mi][e] >= 0;
        const long nn = n <= lastv ? n : lastv;
        const float* src;
        bool zero = !in;
        if (a.T > 0) {
          const bool head = xt[mi][e] == 0;
          // row n - 1 of the outputs; a head row takes its sample's initial state, or (none given) any valid row and is zeroed
          // (the sample index is clamped: the re-read of a range's last block carries (sample, step) one block further)
          const unsigned bmax = bmaxs[mi];
          const unsigned bb = xb[mi][e] < bmax ? xb[mi][e] : bmax;
          // (and the row never goes below 0: in that re-read the carried step no longer belongs to row n)
          src = (head && a.first) ? a.first + (size_t)bb * m.in : a.x + (size_t)((head || nn == 0) ? nn : nn - 1) * m.in;
          zero = zero || (head && !a.first);
          unsigned t2 = xt[mi][e] + str[mi];
          const bool wrap = t2 >= (unsigned)a.T;
          xt[mi][e] = wrap ? t2 - (unsigned)a.T : t2;
          xb[mi][e] += stq[mi] + (wrap ? 1u : 0u);
        } else {
          src = a.x + (size_t)nn * m.in;
        }
        if (m.xelem) sx[mi][e] = f32x4{src[xcol[mi][e]], 0.f, 0.f, 0.f};
        else sx[mi][e] = *reinterpret_cast<const f32x4*>(src + xcol[mi][e]);
        xk[mi][e] = zero ? 0.f : 1.f;
      }
    }
  };
  auto store_block = [&]() {
#pragma unroll
    for (int e = 0; e < EQ; ++e) {
      const f32x4 v = sd[e] * dk[e];
      if (dyg[e] >= 0) c2_store4(dYs + dyo[e], dYs + dpl + dyo[e], v * sdf);
      dbs[e] += v;
    }
#pragma unroll
    for (int mi = 0; mi < NMAT; ++mi) {
      _Float16* X0 = reinterpret_cast<_Float16*>(c2_smem + pl.m[mi].l_xs);
      const int xpl = pl.m[mi].xs_rows * pl.m[mi].XS;
#pragma unroll
      for (int e = 0; e < XQ; ++e)
        if (xr[mi][e] >= 0) {
          if (pl.m[mi].xelem) {
            _Float16 p0, p1;
            split2h(sx[mi][e][0] * (sxf[mi] * xk[mi][e]), p0, p1);
            X0[xo[mi][e]] = p0;
            X0[xpl + xo[mi][e]] = p1;
          } else {
            c2_store4(X0 + xo[mi][e], X0 + xpl + xo[mi][e], sx[mi][e] * (sxf[mi] * xk[mi][e]));
          }
        }
    }
  };

#ifdef TTRNN_ABLATIONS
  const int abl = g.abl;
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = stamp();
#define C2_STAMP(i) { const unsigned long long now_ = stamp(); seg[i] += now_ - last_; last_ = now_; }
#else
  constexpr int abl = 0;
#define C2_STAMP(i)
#endif
  // The loop body ISSUES the next block's loads at its phase points and CONSUMES them (split + LDS stores) at its own end, once
  // the images are free: no load is pending across the back edge, so the wait counts the compiler inserts are exact (with the
  // consuming stores at the top of the next iteration it put s_waitcnt vmcnt(0) in front of register uses all over the body).
  if (blk0 < blk1) {
    load_block(blk0, -1);
    store_block();
  }
  lds_barrier();
  for (long blk = blk0; blk < blk1; ++blk) {
    C2_STAMP(7)
    const bool more = blk + 1 < blk1;
    const long nxt = more ? blk + 1 : blk;
    if (!(abl & 8)) load_block(nxt, 0, more);
    C2_STAMP(1)
    // both matrices run their row-local phases A, B between the same two barriers, then their gradient phases C, D: half the
    // barriers of matrix-after-matrix, and the small input matrix fills the waves the hidden one leaves idle
#pragma unroll
    for (int mi = 0; mi < NMAT; ++mi) {
      const C2Mat& m = pl.m[mi];
      const int* tab = reinterpret_cast<const int*>(c2_smem + m.l_tab);
      const int* tA = tab; const int* tBm = tA + m.NA * 16; const int* tBd = tBm + m.QT * 4; const int* tBs = tBd + m.NB;
      const _Float16* X0 = reinterpret_cast<const _Float16*>(c2_smem + m.l_xs);
      const _Float16* X1 = X0 + m.xs_rows * m.XS;
      const _Float16* GhF = reinterpret_cast<const _Float16*>(c2_smem + m.l_ghf);
      _Float16* C1s = reinterpret_cast<_Float16*>(c2_smem + m.l_c1);
      _Float16* dC1s = reinterpret_cast<_Float16*>(c2_smem + m.l_dc1);
      const int c1p = m.c1_plane, dcp = m.dc1_plane;
      if (!(abl & 1)) {
      // ---- phase A: C1[p][(row, j_h)] = Gt[p][j_t] x[(row, j_h)][j_t]  ->  C1 image [row][q][i_t]; two column tiles at a time ----
#pragma unroll
      for (int wa = 0; wa < WA; ++wa) {
        const int pt = m.PT < NW ? (wave < m.PT * m.NAG ? wave % m.PT : m.PT) : wave + NW * wa;
        if (pt < m.PT) {
          const int na0 = m.PT < NW ? wave / m.PT : 0;
          const int nja = c2_ceil(m.NA, 2 * m.NAG);             // (a literal under a compile-time plan: the loop unrolls)
          for (int ja = 0; ja < nja; ++ja) {
            const int na = na0 + 2 * m.NAG * ja;
            if (na >= m.NA) break;
            const int nb2 = na + m.NAG;
            const bool has2 = nb2 < m.NA;
            const int nc = has2 ? nb2 : na;
            f32x4 r0 = f32x4{0.f, 0.f, 0.f, 0.f}, r1 = r0;
            const int x0 = (16 * na + c) * m.XS + 8 * gq, x1 = (16 * nc + c) * m.XS + 8 * gq;
            xh8 b[2][KAM][2];
#pragma unroll
            for (int kb = 0; kb < KAM; ++kb) {
              const int ko = kb < m.KA ? 32 * kb : 0;
              b[0][kb][0] = c2_ld8(X0 + x0 + ko); b[0][kb][1] = c2_ld8(X1 + x0 + ko);
              b[1][kb][0] = c2_ld8(X0 + x1 + ko); b[1][kb][1] = c2_ld8(X1 + x1 + ko);
            }
#pragma unroll
            for (int kb = 0; kb < KAM; ++kb)
              if (kb < m.KA) {
                r0 = c2_mma3(gta[mi][wa][kb][0], gta[mi][wa][kb][1], b[0][kb][0], b[0][kb][1], r0);
                r1 = c2_mma3(gta[mi][wa][kb][0], gta[mi][wa][kb][1], b[1][kb][0], b[1][kb][1], r1);
              }
            // (no padding columns / rows under the compile-time plans: the guards fold away and the offsets are arithmetic)
            const bool afull = SPEC != 0 && m.NA * 16 == nb * m.JhP && m.P == m.PT * 16 && m.PT % NW == 0;
            const int n0a = 16 * na + c, n1a = 16 * nc + c;
            const int off0 = afull ? ((n0a / m.JhP) * m.QR + (n0a % m.JhP)) * m.CS1 : tA[n0a];
            const int off1 = !has2 ? -1 : afull ? ((n1a / m.JhP) * m.QR + (n1a % m.JhP)) * m.CS1 : tA[n1a];
            if (afull || moffA[mi][wa] >= 0) {
              if (afull || off0 >= 0) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ldexpf(r0[j], eA[mi]);
                c2_store4(C1s + off0 + moffA[mi][wa], C1s + c1p + off0 + moffA[mi][wa], v);
              }
              if (off1 >= 0) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ldexpf(r1[j], eA[mi]);
                c2_store4(C1s + off1 + moffA[mi][wa], C1s + c1p + off1 + moffA[mi][wa], v);
              }
            }
          }
        }
      }
      if (mi == 0 && !(abl & 8)) load_block(nxt, 1, more);
      C2_STAMP(2)
      // ---- phase B: dC1[q][(row, i_t)] = Gh^T[q][i_h] dy[row][i_h][i_t]  ->  dC1 image [row][p][j_h]; two k-blocks in flight ----
      {
        const int nub = c2_ceil(m.QT * m.NB, NW);       // units u = qt NB + nt = wave, wave + NW, ... (a literal trip count under a compile-time plan)
        for (int ib = 0; ib < nub; ++ib) {
          const int u = wave + NW * ib;
          if (u >= m.QT * m.NB) break;
          const int qt = u / m.NB, nt = u - qt * m.NB;
          f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
          const bool bfull = SPEC != 0 && m.Q == m.QT * 16;
          const int n0b = 16 * nt, m0b = 16 * qt + 4 * gq;
          const int dbase = (bfull ? (n0b / m.It) * pl.Ih * DS + n0b % m.It : tBd[nt]) + (8 * gq + qq) * DS + 4 * pp;
          const int mo = bfull ? (m0b / m.JhP) * m.It * m.CS2 + m0b % m.JhP : tBm[qt * 4 + gq];
          const int so = (bfull ? ((n0b / m.It) * m.PR + n0b % m.It) * m.CS2 : tBs[nt]) + c * m.CS2;
          for (int kb = 0; kb < m.KB; kb += 2) {
            const bool has2 = kb + 1 < m.KB;
            const int k1 = has2 ? kb + 1 : kb;
            const _Float16* af0 = GhF + ((size_t)((qt * m.KB + kb) * 2) * 64 + lane) * 8;
            const _Float16* af1 = GhF + ((size_t)((qt * m.KB + k1) * 2) * 64 + lane) * 8;
            const xh8 a00 = c2_ld8(af0), a01 = c2_ld8(af0 + 512), a10 = c2_ld8(af1), a11 = c2_ld8(af1 + 512);
            const xh8 b00 = c2_tr8(dYs + dbase + 32 * kb * DS, DS), b01 = c2_tr8(dYs + dpl + dbase + 32 * kb * DS, DS);
            const xh8 b10 = c2_tr8(dYs + dbase + 32 * k1 * DS, DS), b11 = c2_tr8(dYs + dpl + dbase + 32 * k1 * DS, DS);
            r = c2_mma3(a00, a01, b00, b01, r);
            if (has2) r = c2_mma3(a10, a11, b10, b11, r);
          }
          if (bfull || mo >= 0) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ldexpf(r[j], eB[mi]);
            c2_store4(dC1s + so + mo, dC1s + dcp + so + mo, v);
          }
        }
      }
      }
      C2_STAMP(3)
    }
    if (!(abl & 8)) load_block(nxt, 2, more);
    lds_barrier();
    C2_STAMP(4)
#pragma unroll
    for (int mi = 0; mi < NMAT; ++mi) {
      const C2Mat& m = pl.m[mi];
      const _Float16* X0 = reinterpret_cast<const _Float16*>(c2_smem + m.l_xs);
      const _Float16* X1 = X0 + m.xs_rows * m.XS;
      const _Float16* C1s = reinterpret_cast<const _Float16*>(c2_smem + m.l_c1);
      const _Float16* dC1s = reinterpret_cast<const _Float16*>(c2_smem + m.l_dc1);
      const int c1p = m.c1_plane, dcp = m.dc1_plane;
      if (!(abl & 2)) {
      // ---- phases C, D: the accumulator units of this wave; two k-blocks in flight ----
      const int SC = c2_ceil(m.nC, NW);
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (i < SC && (m.nC >= NW * (i + 1) || uA[mi][i] >= 0)) {
          // k = (row, i_t): chunk k0 = 32 kb lies in row k0 / I_t (I_t a multiple of 16: a k-block's four chunks share the row
          // only when I_t is a multiple of 32 — the row is taken per chunk below)
          for (int kb = 0; kb < m.KC; kb += 2) {
            const bool has2 = kb + 1 < m.KC;
            const int k1 = has2 ? kb + 1 : kb;
            int ra0 = 0, ra1 = 0;                 // rows of the two chunks 32 kb + 8 gq (compare chains: nb <= 4)
            const int kk0 = 32 * kb + 8 * gq, kk1 = 32 * k1 + 8 * gq;
#pragma unroll
            for (int r = 1; r < 4; ++r) { ra0 += (r < nb && kk0 >= r * m.It) ? 1 : 0; ra1 += (r < nb && kk1 >= r * m.It) ? 1 : 0; }
            const int oa0 = ra0 * (pl.Ih * DS - m.It) + 32 * kb, oa1 = ra1 * (pl.Ih * DS - m.It) + 32 * k1;
            const int ob0 = ra0 * (m.QR * m.CS1 - m.It) + 32 * kb, ob1 = ra1 * (m.QR * m.CS1 - m.It) + 32 * k1;
            const xh8 a00 = c2_ld8(dYs + uA[mi][i] + oa0), a01 = c2_ld8(dYs + dpl + uA[mi][i] + oa0);
            const xh8 b00 = c2_ld8(C1s + uB[mi][i] + ob0), b01 = c2_ld8(C1s + c1p + uB[mi][i] + ob0);
            const xh8 a10 = c2_ld8(dYs + uA[mi][i] + oa1), a11 = c2_ld8(dYs + dpl + uA[mi][i] + oa1);
            const xh8 b10 = c2_ld8(C1s + uB[mi][i] + ob1), b11 = c2_ld8(C1s + c1p + uB[mi][i] + ob1);
            acc[mi][i] = c2_mma3(a00, a01, b00, b01, acc[mi][i]);
            if (has2) acc[mi][i] = c2_mma3(a10, a11, b10, b11, acc[mi][i]);
          }
        } else if (i >= SC && (m.nD >= NW * (i - SC + 1) || uA[mi][i] >= 0)) {
          for (int kb = 0; kb < m.KD; kb += 2) {
            const bool has2 = kb + 1 < m.KD;
            const int k1 = has2 ? kb + 1 : kb;
            const int kk0 = 32 * kb + 8 * gq, kk1 = 32 * k1 + 8 * gq;
            int ra0 = 0, ra1 = 0;
#pragma unroll
            for (int r = 1; r < 4; ++r) { ra0 += (r < nb && kk0 >= r * m.JhP) ? 1 : 0; ra1 += (r < nb && kk1 >= r * m.JhP) ? 1 : 0; }
            // past the block's rows (k >= nb JhP) the x image is zero: any finite dC1 address will do (row 0)
            const int oa0 = kk0 < nb * m.JhP ? ra0 * (m.PR * m.CS2 - m.JhP) + kk0 : 0;
            const int oa1 = kk1 < nb * m.JhP ? ra1 * (m.PR * m.CS2 - m.JhP) + kk1 : 0;
            const xh8 a00 = c2_ld8(dC1s + uA[mi][i] + oa0), a01 = c2_ld8(dC1s + dcp + uA[mi][i] + oa0);
            const xh8 a10 = c2_ld8(dC1s + uA[mi][i] + oa1), a11 = c2_ld8(dC1s + dcp + uA[mi][i] + oa1);
            const int xb0 = uB[mi][i] + 32 * kb * m.XS, xb1 = uB[mi][i] + 32 * k1 * m.XS;
            const xh8 b00 = c2_tr8(X0 + xb0, m.XS), b01 = c2_tr8(X1 + xb0, m.XS);
            const xh8 b10 = c2_tr8(X0 + xb1, m.XS), b11 = c2_tr8(X1 + xb1, m.XS);
            acc[mi][i] = c2_mma3(a00, a01, b00, b01, acc[mi][i]);
            if (has2) acc[mi][i] = c2_mma3(a10, a11, b10, b11, acc[mi][i]);
          }
        }
      }
      }
      C2_STAMP(5)
    }
    if (!(abl & 8)) load_block(nxt, 3, more);
    lds_barrier();
    C2_STAMP(6)
    if (!(abl & 4)) store_block();      // (everybody is past the last reads of the dy / x images)
    C2_STAMP(0)
    lds_barrier();
  }
#ifdef TTRNN_ABLATIONS
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 5) && g.diag)
    for (int i = 0; i < 8; ++i) g.diag[(wave ? 8 : 0) + i] = seg[i];
#endif

  // ---- partial sums out: [workgroup][unit][lane] f32x4 (fragment order; k_c2_reduce knows the maps) ----
#pragma unroll
  for (int mi = 0; mi < NMAT; ++mi) {
    const C2Mat& m = pl.m[mi];
    const int SC = c2_ceil(m.nC, NW);
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int u = i < SC ? wave + NW * i : m.nC + wave + NW * (i - SC);      // (unit numbering of k_c2_reduce: C tiles, then D tiles)
      if (uA[mi][i] >= 0 && i < SC + c2_ceil(m.nD, NW))
        reinterpret_cast<f32x4*>(g.a[mi].part)[((size_t)blockIdx.x * m.NU + u) * 64 + lane] = acc[mi][i];
    }
  }
  if (want_bias) {          // column sums of dy: the block rows of a column meet in LDS, added in row order
    float* bl = reinterpret_cast<float*>(c2_smem + pl.l_dy);
#pragma unroll
    for (int e = 0; e < EQ; ++e)
      if (dyg[e] >= 0) *reinterpret_cast<f32x4*>(bl + dyg[e]) = dbs[e];
    __syncthreads();
    for (int o = tid; o < OUT; o += NT) {
      float s = bl[o];
      for (int r = 1; r < nb; ++r) s += bl[r * OUT + o];
      g.bpart[(size_t)blockIdx.x * OUT + o] = s;
    }
  }
}

// ---- fixed-order reduction of the slabs; un-scale; scatter --------------------------------------------------------------------
// One workgroup per (matrix, unit): thread (lane, grp) sums the workgroups grp, grp + 16, ... in order, the sixteen groups are added
// in order through LDS.  d == 2: straight into the packed core gradients (accumulated: the caller zero-fills); d > 2: into the
// merged-core gradients dGh / dGt, pulled back onto the cores by k_c2_pull.
struct C2Red {
  C2Mat m;
  TtShape s;
  const float* part; int grid;
  const int* hdr;
  float* d_packed; float* dGh; float* dGt;
};
struct C2RedArgs { C2Red r[2]; int nu0; };

__global__ void __launch_bounds__(1024) k_c2_reduce(C2RedArgs args) {
  __shared__ f32x4 red[16][64];
  const int mi = (int)blockIdx.x >= args.nu0 ? 1 : 0;
  const C2Red& a = args.r[mi];
  const C2Mat& m = a.m;
  const int u = (int)blockIdx.x - (mi ? args.nu0 : 0);
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int w = grp; w < a.grid; w += 16) v += reinterpret_cast<const f32x4*>(a.part)[((size_t)w * m.NU + u) * 64 + lane];
  red[grp][lane] = v;
  __syncthreads();
  if (grp != 0) return;
  v = red[0][lane];
#pragma unroll
  for (int k = 1; k < 16; ++k) v += red[k][lane];
  const int c = lane & 15, gq = lane >> 4;
  const int ex = a.hdr[0], ed = a.hdr[1], ec1 = a.hdr[4], edc = a.hdr[5];
  if (u < m.nC) {
    const int mt = u / m.QT, qt = u % m.QT;
    const int q = 16 * qt + c, ra = q / m.JhP, jh = q % m.JhP;
    if (q >= m.Q || jh >= m.Jh) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ih = 16 * mt + 4 * gq + j;
      if (ih >= m.Ih) continue;
      const float val = ldexpf(ldexpf(v[j], ed - 14), ec1 - 14);
      if (m.d == 2) a.d_packed[a.s.woff[0] + (size_t)(jh * m.R + ra) * a.s.M[0] + ih] += val;
      else a.dGh[((size_t)ih * m.Jh + jh) * m.R + ra] = val;
    }
  } else {
    const int v2 = u - m.nC, pt = v2 / m.JtT, jtt = v2 % m.JtT;
    const int jt = 16 * jtt + c;
    if (jt >= m.Jt) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = 16 * pt + 4 * gq + j;
      if (p >= m.P) continue;
      const int ra = p / m.It, it = p % m.It;
      const float val = ldexpf(ldexpf(v[j], edc - 14), ex - 14);
      if (m.d == 2) a.d_packed[a.s.woff[1] + (size_t)jt * a.s.M[1] + it * m.R + ra] += val;
      else a.dGt[((size_t)it * m.Jt + jt) * m.R + ra] = val;
    }
  }
}

// d > 2: a side of two cores (ka, ka + 1) with merged gradient dM[(ia, ib)][(ja, jb)][e] (e = the side's outer rank: the right
// rank of the head, the left rank of the tail).  One thread per core entry, sums in a fixed order.
__global__ void __launch_bounds__(256) k_c2_pull(TtShape s, C2Mat m, const float* __restrict__ packed, const float* __restrict__ dGh,
                                                 const float* __restrict__ dGt, float* __restrict__ d_packed) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long base = 0;
  for (int k = 0; k < s.d; ++k) {
    const long n = (long)s.K[k] * s.M[k];
    if (t >= base + n) { base += n; continue; }
    const long e = t - base;
    // packed index e = (j*R_{k+1} + b)*M_k + i*R_k + a
    const int a = (int)(e % s.R[k]), i = (int)((e / s.R[k]) % s.I[k]);
    const int b = (int)((e / s.M[k]) % s.R[k + 1]), j = (int)(e / ((long)s.M[k] * s.R[k + 1]));
    const bool head = k < m.s;
    const int k0 = head ? 0 : m.s, nside = head ? m.s : s.d - m.s;
    const float* dM = head ? dGh : dGt;
    const int Jm = head ? m.Jh : m.Jt;
    float sum = 0.f;
    if (nside == 1) {
      // the side IS this core: head e-rank = b (a = 0), tail e-rank = a (b = 0)
      sum = dM[((size_t)i * Jm + j) * m.R + (head ? b : a)];
    } else if (k == k0) {
      // first core of the side, G_a[a, i, j, b]: partner G_b[b, ib, jb, e']; head: a = 0, outer rank e' = right rank of G_b;
      // tail: outer rank = a (left rank of this core), partner's right rank is 1
      const int kb_ = k + 1;
      for (int ib = 0; ib < s.I[kb_]; ++ib)
        for (int jb = 0; jb < s.J[kb_]; ++jb) {
          const size_t row = ((size_t)(i * s.I[kb_] + ib) * Jm + (j * s.J[kb_] + jb)) * m.R;
          if (head) {
            for (int r = 0; r < m.R; ++r) sum = fmaf(dM[row + r], c2_core(s, packed, kb_, b, ib, jb, r), sum);
          } else {
            sum = fmaf(dM[row + a], c2_core(s, packed, kb_, b, ib, jb, 0), sum);
          }
        }
    } else {
      // second core of the side, G_b[a, i, j, b]: partner G_a[a0, ia, ja, a]; head: a0 = 0, outer rank = b; tail: outer = a0, b = 0
      const int ka_ = k - 1;
      for (int ia = 0; ia < s.I[ka_]; ++ia)
        for (int ja = 0; ja < s.J[ka_]; ++ja) {
          const size_t row = ((size_t)(ia * s.I[k] + i) * Jm + (ja * s.J[k] + j)) * m.R;
          if (head) {
            sum = fmaf(dM[row + b], c2_core(s, packed, ka_, 0, ia, ja, a), sum);
          } else {
            for (int r = 0; r < m.R; ++r) sum = fmaf(dM[row + r], c2_core(s, packed, ka_, r, ia, ja, a), sum);
          }
        }
    }
    d_packed[s.woff[k] + e] += sum;
    return;
  }
}

// d_bias (+)= sum over the workgroups' partial column sums, in a fixed order (sixteen groups of workgroups, then the groups); up to
// two destinations (the LSTM's two bias vectors).  64 columns per workgroup.
__global__ void __launch_bounds__(1024) k_c2_bias(const float* __restrict__ bpart, int grid, int OUT, float* __restrict__ d0,
                                                  float* __restrict__ d1) {
  __shared__ float red[16][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + col;
  float s = 0.f;
  if (o < OUT)
    for (int w = grp; w < grid; w += 16) s += bpart[(size_t)w * OUT + o];
  red[grp][col] = s;
  __syncthreads();
  if (grp != 0 || o >= OUT) return;
  s = red[0][col];
#pragma unroll
  for (int k = 1; k < 16; ++k) s += red[k][col];
  if (d0) d0[o] += s;
  if (d1) d1[o] += s;
}

size_t c2_al(size_t v) { return (v + 255) & ~(size_t)255; }

// the plan of a call for ONE choice of split points; false = not taken
bool c2_plan_at(C2Plan* pl, const TtShape* const* shapes, int nmat, const int* sp, int cus) {
  for (int big = 0; big < 2; ++big)
    for (int nb = 4; nb >= 1; nb >>= 1) {
      C2Plan p{};
      p.nmat = nmat; p.nb = nb; p.big = big;
      bool ok = true;
      for (int i = 0; i < nmat && ok; ++i) ok = c2_plan_mat(&p.m[i], *shapes[i], sp[i], nb, big);
      if (!ok) continue;
      if (!c2_layout(&p)) continue;
      p.grid = cus;
      // workspace
      size_t w = 0;
      p.w_cmax = (long)w; w += 256;                                   // [0]: max|x| of matrix 0, [1]: of matrix 1, [2]: max|dy| (computed here when no bound came)
      p.w_bpart = (long)w; w += c2_al((size_t)cus * p.OUT * 4);
      for (int i = 0; i < nmat; ++i) {
        C2Mat& m = p.m[i];
        m.w_gh = (long)w; w += c2_al((size_t)m.head_elems * 4);
        m.w_gt = (long)w; w += c2_al((size_t)m.tail_elems * 4);
        m.w_dgh = (long)w; w += c2_al((size_t)m.head_elems * 4);
        m.w_dgt = (long)w; w += c2_al((size_t)m.tail_elems * 4);
        m.w_gtf = (long)w; w += c2_al((size_t)m.PT * m.KA * 2 * 1024);
        m.w_ghf = (long)w; w += c2_al((size_t)m.QT * m.KB * 2 * 1024);
        m.w_hdr = (long)w; w += 256;
        m.w_part = (long)w; w += c2_al((size_t)cus * m.NU * 1024);
      }
      p.ws_bytes = (long)w;
      p.ok = 1;
      *pl = p;
      return true;
    }
  return false;
}

// The plan of a call: the matrices (1 or 2) share one pass over dy, so they split their output modes at the SAME core (the dy
// image [row][i_h][i_t] is one); of the splits whose sides have one or two cores, the cheapest in chain FLOPs that the kernel takes.
bool c2_plan(C2Plan* pl, const TtShape* const* shapes, int nmat, int cus) {
  *pl = C2Plan{};
  if (nmat < 1 || nmat > 2) return false;
  const int d = shapes[0]->d;
  if (d < 2 || d > 4 || (nmat == 2 && shapes[1]->d != d)) return false;
  bool found = false;
  double bc = 0;
  for (int k = 1; k < d; ++k) {
    if (k > 2 || d - k > 2) continue;
    double cst = 0;
    for (int i = 0; i < nmat; ++i) cst += c2_chain_flops(*shapes[i], k);
    if (found && cst >= bc) continue;
    const int sp[2] = {k, k};
    C2Plan p;
    if (!c2_plan_at(&p, shapes, nmat, sp, cus)) continue;
    *pl = p; bc = cst; found = true;
  }
  return found;
}

template <int NMAT>
int c2_launch_main(const C2Args& a, hipStream_t stream) {
  const C2Plan& pl = a.pl;
  const void* fn;
  void (*kf)(C2Args);
  constexpr int SP = NMAT == 2 ? 1 : 2;
  constexpr C2Plan spk = c2_const_plan<SP>();
  constexpr int SP4 = NMAT == 2 ? 3 : 4, SD2 = NMAT == 2 ? 5 : 6, SD4 = NMAT == 2 ? 7 : 8;
  constexpr C2Plan spk4 = c2_const_plan<SP4>(), sd2 = c2_const_plan<SD2>(), sd4 = c2_const_plan<SD4>();
  const bool ct = !(opt(OPT_DEV2) & 4);
  if (pl.big && ct && c2_same_kernel_plan(pl, spk4)) kf = k_c2w<SP4, NMAT, C2_NW, 8, 2, 2, 8, 2>;
  else if (pl.big && ct && c2_same_kernel_plan(pl, sd2)) kf = k_c2w<SD2, NMAT, C2_NW, 8, 2, 2, 8, 2>;
  else if (pl.big && ct && c2_same_kernel_plan(pl, sd4)) kf = k_c2w<SD4, NMAT, C2_NW, 8, 2, 2, 8, 2>;
  else if (pl.big) kf = k_c2w<0, NMAT, C2_NW, 8, 2, 2, 8, 2>;
  else if (c2_same_kernel_plan(pl, spk) && !(opt(OPT_DEV2) & 4)) kf = k_c2w<SP, NMAT, C2_NW, 4, 1, 1, 4, 1>;      // (dev2 bit 2: the run-time-plan kernel, A/B)
  else kf = k_c2w<0, NMAT, C2_NW, 4, 1, 1, 4, 1>;
  fn = reinterpret_cast<const void*>(kf);
  if (ensure_dynamic_lds(fn, pl.lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
  hipLaunchKernelGGL(kf, dim3(pl.grid), dim3(C2_NT), pl.lds, stream, a);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// ---- the register hand-off kernel (ttrnn_c2r_dev.h): compile-time shapes -----------------------------------------------------
#include "ttrnn_c2r_dev.h"

// the encoder layer of the reference (params_model.py:2-4,14-16: 40 -> 768, two cores) and the four-core models of its result
// tables split two by two; LSTM (I_t = 64) at ranks 2 and 4, the TT-GRU's hidden matrix (I_t = 48) at rank 2
template <int R> using C2R_In2 = C2RM<5, 8, 48, 64, R, 40>;
template <int R> using C2R_Hid2 = C2RM<24, 32, 48, 64, R, 768>;
template <int R> using C2R_In4 = C2RM<4, 10, 48, 64, R, 40>;
template <int R> using C2R_Hid4 = C2RM<16, 48, 48, 64, R, 768>;
template <int R> using C2R_GruHid2 = C2RM<24, 32, 48, 48, R, 768>;
template <int R> using C2R_GruHid4 = C2RM<16, 48, 36, 64, R, 768>;      // the four-core TT-GRU's hidden matrix: (6, 6, 8, 8) contracted pairwise
constexpr int C2R_NBR = 2;

struct C2RDims { int Jh, Jt, Ih, It, R, in; };
template <class M> constexpr C2RDims c2r_dims() { return C2RDims{M::Jh, M::Jt, M::Ih, M::It, M::R, M::in}; }

bool c2r_dims_of(const TtShape& s, C2RDims* o) {
  if (s.d != 2 && s.d != 4) return false;
  const int sp = s.d / 2;
  int It = 1, Jt = 1, Ih = 1, Jh = 1;
  for (int k = sp; k < s.d; ++k) { It *= s.I[k]; Jt *= s.J[k]; }
  for (int k = 0; k < sp; ++k) { Ih *= s.I[k]; Jh *= s.J[k]; }
  *o = C2RDims{Jh, Jt, Ih, It, s.R[sp], s.in_size};
  return true;
}
bool c2r_same(const C2RDims& a, const C2RDims& b) {
  return a.Jh == b.Jh && a.Jt == b.Jt && a.Ih == b.Ih && a.It == b.It && a.R == b.R && a.in == b.in;
}
template <class S>
bool c2r_takes(const TtShape* const* shapes, int nmat) {
  if (nmat != S::NMAT) return false;
  C2RDims d0, d1;
  if (!c2r_dims_of(*shapes[0], &d0) || !c2r_same(d0, c2r_dims<typename S::M0>())) return false;
  if (nmat == 2 && (!c2r_dims_of(*shapes[1], &d1) || !c2r_same(d1, c2r_dims<typename S::M1>()) || shapes[1]->d != shapes[0]->d)) return false;
  return true;
}

// the call's matrices as k_c2_merge / k_c2_prep / k_c2_pull see them: q = a * (16 JHT) + j_h (tiles of Gh^T do not straddle ranks)
template <class M>
bool c2r_mat(C2Mat* m, const TtShape& s) {
  if (!c2_mat_from(m, s.d, s.d / 2, M::Jh, M::Jt, M::Ih, M::It, M::R, s.in_size, s.out_size, 1, 1)) return false;
  m->JhP = 16 * M::JHT; m->Q = M::R * m->JhP; m->QT = m->Q / 16;
  return m->KA == M::JTK && m->KB == M::IHK && m->PT == M::R * M::ITT;
}

struct C2RWs { long cmax, bpart, part, gh[2], gt[2], dgh[2], dgt[2], gtf[2], ghf[2], hdr[2], bytes; };
template <class S>
C2RWs c2r_ws(const C2Mat* m, int cus) {
  C2RWs w{};
  size_t o = 0;
  w.cmax = (long)o; o += 256;
  w.bpart = (long)o; o += c2_al((size_t)cus * S::OUT * 4);
  w.part = (long)o; o += c2_al((size_t)cus * 8 * S::NACC * 1024);
  for (int i = 0; i < S::NMAT; ++i) {
    w.gh[i] = (long)o; o += c2_al((size_t)m[i].head_elems * 4);
    w.gt[i] = (long)o; o += c2_al((size_t)m[i].tail_elems * 4);
    w.dgh[i] = (long)o; o += c2_al((size_t)m[i].head_elems * 4);
    w.dgt[i] = (long)o; o += c2_al((size_t)m[i].tail_elems * 4);
    w.gtf[i] = (long)o; o += c2_al((size_t)m[i].PT * m[i].KA * 2 * 1024);
    w.ghf[i] = (long)o; o += c2_al((size_t)m[i].QT * m[i].KB * 2 * 1024);
    w.hdr[i] = (long)o; o += 256;
  }
  w.bytes = (long)o;
  return w;
}

template <class S>
bool c2r_mats(C2Mat* m, const TtShape* const* shapes) {
  if (!c2r_mat<typename S::M0>(&m[0], *shapes[0])) return false;
  if (S::NMAT == 2 && !c2r_mat<typename S::M1>(&m[1], *shapes[1])) return false;
  return true;
}

template <class S>
size_t c2r_workspace(const TtShape* const* shapes) {
  C2Mat m[2];
  if (!c2r_mats<S>(m, shapes)) return 0;
  return (size_t)c2r_ws<S>(m, device_cu_count()).bytes;
}

template <class S>
int c2r_launch(const TtShape* const* shapes, int64_t n_rows, const float* const* packed, const float* const* x,
               const float* const* first, const int* T, const float* dy, const unsigned* const* x_cmax, const int* x_cn,
               const unsigned* dy_cmax, float* const* d_packed, float* d_bias0, float* d_bias1, void* workspace,
               size_t workspace_bytes, hipStream_t stream) {
  constexpr int nmat = S::NMAT;
  C2Mat m[2];
  if (!c2r_mats<S>(m, shapes)) return TTRNN_ERR_UNSUPPORTED;
  const int cus = device_cu_count();
  const C2RWs w = c2r_ws<S>(m, cus);
  if (!workspace || workspace_bytes < (size_t)w.bytes) return TTRNN_ERR_WORKSPACE;
  if (n_rows <= 0) return TTRNN_OK;
  if (n_rows >= ((int64_t)1 << 31)) return TTRNN_ERR_UNSUPPORTED;
  const long nblk = (n_rows + S::NBR - 1) / S::NBR;
  const int grid = nblk < cus ? (int)nblk : cus;
  char* ws = (char*)workspace;
  unsigned* cm = (unsigned*)(ws + w.cmax);
  bool need_zero = !dy_cmax;
  for (int i = 0; i < nmat; ++i) need_zero = need_zero || !x_cmax[i];
  if (need_zero && hipMemsetAsync(cm, 0, 256, stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
  if (!dy_cmax) {
    const size_t n4 = (size_t)n_rows * S::OUT / 4;
    hipLaunchKernelGGL(k_c2_absmax, dim3((unsigned)(n4 / 1024 + 1 < 2048 ? n4 / 1024 + 1 : 2048)), dim3(256), 0, stream, dy, n4, cm + 2);
  }
  C2PrepArgs pa{};
  C2RArgs ka{};
  C2RRed ra{};
  ka.dy = dy; ka.n_rows = (long)n_rows;
#ifdef TTRNN_ABLATIONS
  ka.abl = opt(OPT_DEV2) >> 16;
  ka.diag = (unsigned long long*)(ws + w.cmax + 64);
#endif
  ka.part = (float*)(ws + w.part);
  ka.bpart = (d_bias0 || d_bias1) ? (float*)(ws + w.bpart) : nullptr;
  ra.part = ka.part; ra.grid = grid;
  int ntile = 0;
  for (int i = 0; i < nmat; ++i) {
    const TtShape& s = *shapes[i];
    float* Gh = (float*)(ws + w.gh[i]);
    float* Gt = (float*)(ws + w.gt[i]);
    if (!x_cmax[i]) {
      const size_t n4 = (size_t)n_rows * m[i].in / 4;
      hipLaunchKernelGGL(k_c2_absmax, dim3((unsigned)(n4 / 1024 + 1 < 1024 ? n4 / 1024 + 1 : 1024)), dim3(256), 0, stream, x[i], n4, cm + i);
      if (T[i] > 0 && first[i]) {
        const size_t f4 = (size_t)(n_rows / T[i]) * m[i].in / 4;
        hipLaunchKernelGGL(k_c2_absmax, dim3((unsigned)(f4 / 1024 + 1)), dim3(256), 0, stream, first[i], f4, cm + i);
      }
    }
    const long nm = m[i].head_elems + m[i].tail_elems;
    hipLaunchKernelGGL(k_c2_merge, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, stream, s, m[i], packed[i], Gh, Gt);
    C2Prep& p = pa.p[i];
    p.m = m[i]; p.Gh = Gh; p.Gt = Gt;
    p.x_cmax = x_cmax[i] ? x_cmax[i] : cm + i; p.x_n = x_cmax[i] ? x_cn[i] : 1;
    p.dy_cmax = dy_cmax ? dy_cmax : cm + 2; p.dy_n = dy_cmax ? S::OUT : 1;
    p.hdr = (int*)(ws + w.hdr[i]); p.gtf = (_Float16*)(ws + w.gtf[i]); p.ghf = (_Float16*)(ws + w.ghf[i]);
    C2RMatArgs& a = ka.a[i];
    a.x = x[i]; a.first = first[i]; a.T = T[i]; a.hdr = p.hdr; a.gtf = p.gtf; a.ghf = p.ghf;
    ra.s[i] = s; ra.hdr[i] = p.hdr; ra.d_packed[i] = d_packed[i];
    ra.dGh[i] = (float*)(ws + w.dgh[i]); ra.dGt[i] = (float*)(ws + w.dgt[i]);
    const int t = 1 + m[i].PT * m[i].KA + m[i].QT * m[i].KB;
    if (t > ntile) ntile = t;
  }
  hipLaunchKernelGGL(k_c2_prep, dim3(ntile, nmat), dim3(256), 0, stream, pa);
  void (*kf)(C2RArgs) = k_c2r<S>;
  if (ensure_dynamic_lds(reinterpret_cast<const void*>(kf), S::LDS) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
  hipLaunchKernelGGL(kf, dim3(grid), dim3(S::NT), S::LDS, stream, ka);
  hipLaunchKernelGGL(k_c2r_reduce<S>, dim3(S::NOUT), dim3(1024), 0, stream, ra);
  for (int i = 0; i < nmat; ++i)
    if (m[i].d > 2) {
      const TtShape& s = *shapes[i];
      hipLaunchKernelGGL(k_c2_pull, dim3((unsigned)((s.wtotal + 255) / 256)), dim3(256), 0, stream, s, m[i], packed[i],
                         (const float*)(ws + w.dgh[i]), (const float*)(ws + w.dgt[i]), d_packed[i]);
    }
  if (ka.bpart)
    hipLaunchKernelGGL(k_c2_bias, dim3((S::OUT + 63) / 64), dim3(1024), 0, stream, (const float*)ka.bpart, grid, S::OUT, d_bias0, d_bias1);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

// the instantiations: c2r_each(f) calls f.template operator()<S>() for each until one returns true.  Rank 4 only where the kernel
// keeps its fragments in registers without spilling (the four-core hidden matrix alone: 234 VGPRs); the other rank-4 shapes —
// 64 to 128 fragment registers per wave on top of the staging — stay on k_c2w's compile-time plans.
template <class F>
bool c2r_each(F&& f) {
  return f.template operator()<C2RS<2, C2R_NBR, C2R_In2<2>, C2R_Hid2<2>>>() || f.template operator()<C2RS<1, C2R_NBR, C2R_Hid2<2>, C2R_Hid2<2>>>() ||
         f.template operator()<C2RS<2, C2R_NBR, C2R_In4<2>, C2R_Hid4<2>>>() || f.template operator()<C2RS<1, C2R_NBR, C2R_Hid4<2>, C2R_Hid4<2>>>() ||
         f.template operator()<C2RS<1, C2R_NBR, C2R_Hid4<4>, C2R_Hid4<4>>>() ||
         f.template operator()<C2RS<1, C2R_NBR, C2R_GruHid2<2>, C2R_GruHid2<2>>>() ||
         f.template operator()<C2RS<1, C2R_NBR, C2R_GruHid4<2>, C2R_GruHid4<2>>>() || f.template operator()<C2RS<1, C2R_NBR, C2R_GruHid4<4>, C2R_GruHid4<4>>>();
}

struct C2RWsVisit {
  const TtShape* const* shapes; int nmat; size_t* out;
  template <class S> bool operator()() const {
    if (!c2r_takes<S>(shapes, nmat)) return false;
    *out = c2r_workspace<S>(shapes);
    return *out > 0;
  }
};
struct C2RLaunchVisit {
  const TtShape* const* shapes; int nmat; int64_t n_rows; const float* const* packed; const float* const* x;
  const float* const* first; const int* T; const float* dy; const unsigned* const* x_cmax; const int* x_cn;
  const unsigned* dy_cmax; float* const* d_packed; float* d_bias0; float* d_bias1; void* workspace; size_t workspace_bytes;
  hipStream_t stream; int* status;
  template <class S> bool operator()() const {
    if (!c2r_takes<S>(shapes, nmat) || c2r_workspace<S>(shapes) == 0) return false;
    *status = c2r_launch<S>(shapes, n_rows, packed, x, first, T, dy, x_cmax, x_cn, dy_cmax, d_packed, d_bias0, d_bias1, workspace,
                            workspace_bytes, stream);
    return true;
  }
};
// bytes of the hand-off kernel's workspace; 0 = it does not take the call (option dev2 bit 6: never — the A/B against k_c2w)
size_t c2r_workspace_bytes(const TtShape* const* shapes, int nmat) {
  if (opt(OPT_DEV2) & 64) return 0;
  size_t b = 0;
  c2r_each(C2RWsVisit{shapes, nmat, &b});
  return b;
}

}  // namespace

// ---- what ttrnn_api.hip sees ------------------------------------------------------------------------------------------------------
// The chain is taken where the dense gradient costs more than `c` times its FLOPs (c = 1.5), for the matrix that
// decides — the hidden one; an input matrix that rides along shares the pass over dy whatever its own ratio.
bool c2w_prefers_chain(const TtShape& s) {
  if (s.d < 2 || s.d > 4) return false;
  double bc = -1;
  for (int k = 1; k < s.d; ++k) {
    if (k > 2 || s.d - k > 2) continue;
    const double cst = c2_chain_flops(s, k);
    if (bc < 0 || cst < bc) bc = cst;
  }
  return bc > 0 && 2.0 * s.in_size * s.out_size > 1.5 * bc;
}

// small_only: only plans of the small kernel variant count.  That is what the router offers: the large variant (P > 128 rows of
// Gt: rank 4 at H = 768, the d = 4 shapes) spills its run-time plan and LOSES to the dense gradient (H = 768, d = 2, r = 4 at
// the speaker encoder's size: 4.9 ms against 1.8) — it stays reachable for tests through option dev2 bit 3.
size_t c2w_workspace_bytes(const TtShape* const* shapes, int nmat, bool small_only) {
  const size_t rb = c2r_workspace_bytes(shapes, nmat);
  if (rb) return rb;
  C2Plan pl;
  if (!c2_plan(&pl, shapes, nmat, device_cu_count())) return 0;
  if (small_only && pl.big) {
    // ... unless the plan has a compile-time instantiation (rank 4 at the encoder's size: no spills, and ahead of the dense gradient)
    constexpr C2Plan s3 = c2_const_plan<3>(), s4 = c2_const_plan<4>(), s5 = c2_const_plan<5>(), s6 = c2_const_plan<6>(),
                     s7 = c2_const_plan<7>(), s8 = c2_const_plan<8>();
    const bool two = pl.nmat == 2;
    if (!c2_same_kernel_plan(pl, two ? s3 : s4) && !c2_same_kernel_plan(pl, two ? s5 : s6) && !c2_same_kernel_plan(pl, two ? s7 : s8)) return 0;
  }
  return (size_t)pl.ws_bytes;
}

// x[i] / first[i] / T[i]: operand rows of matrix i (T > 0: the layer's outputs, read one step back; first = h_0 or NULL);
// x_cmax[i] (n = x_cn[i] bit patterns) / dy_cmax (n = OUT) optional bounds; d_packed[i] accumulated into; d_bias0 / d_bias1
// (either may be NULL) receive the column sums of dy.
int launch_c2w(const TtShape* const* shapes, int nmat, int64_t n_rows, const float* const* packed, const float* const* x,
               const float* const* first, const int* T, const float* dy, const unsigned* const* x_cmax, const int* x_cn,
               const unsigned* dy_cmax, float* const* d_packed, float* d_bias0, float* d_bias1, void* workspace,
               size_t workspace_bytes, hipStream_t stream) {
  if (c2r_workspace_bytes(shapes, nmat)) {
    int st = TTRNN_ERR_UNSUPPORTED;
    if (c2r_each(C2RLaunchVisit{shapes, nmat, n_rows, packed, x, first, T, dy, x_cmax, x_cn, dy_cmax, d_packed, d_bias0, d_bias1,
                                workspace, workspace_bytes, stream, &st}))
      return st;
  }
  C2Plan pl;
  const int cus = device_cu_count();
  if (!c2_plan(&pl, shapes, nmat, cus)) return TTRNN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < (size_t)pl.ws_bytes) return TTRNN_ERR_WORKSPACE;
  if (n_rows <= 0) return TTRNN_OK;
  if (n_rows >= ((int64_t)1 << 31)) return TTRNN_ERR_UNSUPPORTED;      // (32-bit (sample, step) arithmetic of the staged rows)
  const long nblk = (n_rows + pl.nb - 1) / pl.nb;
  if (nblk < pl.grid) pl.grid = (int)nblk;
  char* ws = (char*)workspace;
  unsigned* cm = (unsigned*)(ws + pl.w_cmax);
  bool need_zero = !dy_cmax;
  for (int i = 0; i < nmat; ++i) need_zero = need_zero || !x_cmax[i];
  if (need_zero && hipMemsetAsync(cm, 0, 256, stream) != hipSuccess) return TTRNN_ERR_LAUNCH;
  if (!dy_cmax) {
    const size_t n4 = (size_t)n_rows * pl.OUT / 4;
    hipLaunchKernelGGL(k_c2_absmax, dim3((unsigned)(n4 / 1024 + 1 < 2048 ? n4 / 1024 + 1 : 2048)), dim3(256), 0, stream, dy, n4, cm + 2);
  }
  C2PrepArgs pa{};
  C2Args ka{};
  C2RedArgs ra{};
  ka.pl = pl; ka.dy = dy; ka.n_rows = (long)n_rows;
#ifdef TTRNN_ABLATIONS
  ka.abl = opt(OPT_DEV2) >> 16;
  ka.diag = (unsigned long long*)(ws + pl.w_cmax + 64);
#endif
  ka.bpart = (d_bias0 || d_bias1) ? (float*)(ws + pl.w_bpart) : nullptr;
  for (int i = 0; i < nmat; ++i) {
    const C2Mat& m = pl.m[i];
    const TtShape& s = *shapes[i];
    float* Gh = (float*)(ws + m.w_gh);
    float* Gt = (float*)(ws + m.w_gt);
    if (!x_cmax[i]) {
      // no bound from the caller: the maximum over the operand rows (for T > 0 this reads the layer's outputs and `first`)
      const size_t n4 = (size_t)n_rows * m.in / 4;
      hipLaunchKernelGGL(k_c2_absmax, dim3((unsigned)(n4 / 1024 + 1 < 1024 ? n4 / 1024 + 1 : 1024)), dim3(256), 0, stream, x[i], n4, cm + i);
      if (T[i] > 0 && first[i]) {
        const size_t f4 = (size_t)(n_rows / T[i]) * m.in / 4;
        hipLaunchKernelGGL(k_c2_absmax, dim3((unsigned)(f4 / 1024 + 1)), dim3(256), 0, stream, first[i], f4, cm + i);
      }
    }
    const long nm = m.head_elems + m.tail_elems;
    hipLaunchKernelGGL(k_c2_merge, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, stream, s, m, packed[i], Gh, Gt);
    C2Prep& p = pa.p[i];
    p.m = m; p.Gh = Gh; p.Gt = Gt;
    p.x_cmax = x_cmax[i] ? x_cmax[i] : cm + i; p.x_n = x_cmax[i] ? x_cn[i] : 1;
    p.dy_cmax = dy_cmax ? dy_cmax : cm + 2; p.dy_n = dy_cmax ? pl.OUT : 1;
    p.hdr = (int*)(ws + m.w_hdr); p.gtf = (_Float16*)(ws + m.w_gtf); p.ghf = (_Float16*)(ws + m.w_ghf);
    C2MatArgs& a = ka.a[i];
    a.x = x[i]; a.first = first[i]; a.T = T[i]; a.hdr = p.hdr; a.gtf = p.gtf; a.ghf = p.ghf; a.part = (float*)(ws + m.w_part);
    C2Red& r = ra.r[i];
    r.m = m; r.s = s; r.part = a.part; r.grid = pl.grid; r.hdr = p.hdr; r.d_packed = d_packed[i];
    r.dGh = (float*)(ws + m.w_dgh); r.dGt = (float*)(ws + m.w_dgt);
  }
  int ntile = 0;
  for (int i = 0; i < nmat; ++i) {
    const int t = 1 + pl.m[i].PT * pl.m[i].KA + pl.m[i].QT * pl.m[i].KB;
    if (t > ntile) ntile = t;
  }
  hipLaunchKernelGGL(k_c2_prep, dim3(ntile, nmat), dim3(256), 0, stream, pa);
  int st = nmat == 2 ? c2_launch_main<2>(ka, stream) : c2_launch_main<1>(ka, stream);
  if (st != TTRNN_OK) return st;
  ra.nu0 = pl.m[0].NU;
  const int nu = pl.m[0].NU + (nmat == 2 ? pl.m[1].NU : 0);
  hipLaunchKernelGGL(k_c2_reduce, dim3(nu), dim3(1024), 0, stream, ra);
  for (int i = 0; i < nmat; ++i) {
    const C2Mat& m = pl.m[i];
    if (m.d > 2) {
      const TtShape& s = *shapes[i];
      hipLaunchKernelGGL(k_c2_pull, dim3((unsigned)((s.wtotal + 255) / 256)), dim3(256), 0, stream, s, m, packed[i],
                         (const float*)(ws + m.w_dgh), (const float*)(ws + m.w_dgt), d_packed[i]);
    }
  }
  if (ka.bpart)
    hipLaunchKernelGGL(k_c2_bias, dim3((pl.OUT + 63) / 64), dim3(1024), 0, stream, (const float*)ka.bpart, pl.grid, pl.OUT, d_bias0, d_bias1);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

}  // namespace ttrnn
