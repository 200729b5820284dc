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

__global__ void __launch_bounds__(256) k_c2_prep(C2PrepArgs args) {
  __shared__ float red[256];
  const C2Prep& a = args.p[blockIdx.x];
  const C2Mat& m = a.m;
  const int tid = threadIdx.x;
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
  const int ex = c2_expo(mx), ed = c2_expo(md), egt = c2_expo(mgt), egh = c2_expo(mgh);
  const int ec1 = c2_expo(l1t * mx), edc = c2_expo(l1h * md);
  if (tid == 0) {
    a.hdr[0] = ex; a.hdr[1] = ed; a.hdr[2] = egt; a.hdr[3] = egh; a.hdr[4] = ec1; a.hdr[5] = edc;
  }
  const float sgt = ldexpf(1.f, 14 - egt), sgh = ldexpf(1.f, 14 - egh);
  const long ngt = (long)m.PT * m.KA * 64 * 8;
  const long pl_t = ngt, pl_h = (long)m.QT * m.KB * 64 * 8;
  for (long e = tid; e < ngt; e += 256) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    const long tile = e >> 9;
    const int kb = (int)(tile % m.KA), pt = (int)(tile / m.KA);
    const int p = 16 * pt + (lane & 15), jt = 32 * kb + 8 * (lane >> 4) + j;
    float v = 0.f;
    if (p < m.P && jt < m.Jt) v = a.Gt[((long)(p % m.It) * m.Jt + jt) * m.R + p / m.It] * sgt;
    _Float16 p0, p1;
    split2h(v, p0, p1);
    // [(tile*2 + plane)][lane][8]
    a.gtf[(tile * 2 + 0) * 512 + lane * 8 + j] = p0;
    a.gtf[(tile * 2 + 1) * 512 + lane * 8 + j] = p1;
  }
  (void)pl_t;
  for (long e = tid; e < pl_h; e += 256) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    const long tile = e >> 9;
    const int kb = (int)(tile % m.KB), qt = (int)(tile / m.KB);
    const int q = 16 * qt + (lane & 15), ih = 32 * kb + 8 * (lane >> 4) + j;
    const int ra = q / m.JhP, jh = q % m.JhP;
    float v = 0.f;
    if (q < m.Q && jh < m.Jh && ih < m.Ih) v = a.Gh[((long)ih * m.Jh + jh) * m.R + ra] * sgh;
    _Float16 p0, p1;
    split2h(v, p0, p1);
    a.ghf[(tile * 2 + 0) * 512 + lane * 8 + j] = p0;
    a.ghf[(tile * 2 + 1) * 512 + lane * 8 + j] = p1;
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

template <int NMAT, int NACC, int WA, int KAM, int EQ>
__global__ void __launch_bounds__(C2_NT) k_c2w(C2Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c2_smem[];
  const C2Plan& pl = g.pl;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, gq = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int nb = pl.nb, OUT = pl.OUT, DS = pl.DS;
  const long nblk = (g.n_rows + nb - 1) / nb;

  // ---- LDS: zero everything once (padding rows / columns are never written afterwards: they stay finite zeros) ----
  for (int i = tid; i < pl.lds / 16; i += C2_NT) reinterpret_cast<f32x4*>(c2_smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  _Float16* dYs = reinterpret_cast<_Float16*>(c2_smem + pl.l_dy);
  const int dpl = pl.dy_rows * DS;                              // plane size (halves)
  _Float16* C1s = reinterpret_cast<_Float16*>(c2_smem + pl.l_c1);
  _Float16* dC1s = reinterpret_cast<_Float16*>(c2_smem + pl.l_dc1);

  // ---- tables + per-thread constants -------------------------------------------------------------------------------
  const int* tA[NMAT]; const int* tBm[NMAT]; const int* tBd[NMAT]; const int* tBs[NMAT];
  const int* tCa[NMAT]; const int* tCb[NMAT]; const int* tDa[NMAT];
  _Float16* Xs[NMAT]; const _Float16* GhF[NMAT];
  int xpl[NMAT];
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
    int* a_ = tab;                   // [NA*16]
    int* bm = a_ + m.NA * 16;        // [QT*4]
    int* bd = bm + m.QT * 4;         // [NB]   dy image offset of the column tile
    int* bs = bd + m.NB;             // [NB]   dC1 image offset of the column tile
    int* ca = bs + m.NB;             // [KC*4]
    int* cb = ca + m.KC * 4;         // [KC*4]
    int* da = cb + m.KC * 4;         // [KD*4]
    for (int n = tid; n < m.NA * 16; n += C2_NT) a_[n] = n < nb * m.JhP ? ((n / m.JhP) * m.QR + (n % m.JhP)) * m.CS1 : -1;
    for (int e = tid; e < m.QT * 4; e += C2_NT) {
      const int m0 = 16 * (e >> 2) + 4 * (e & 3);
      bm[e] = m0 < m.Q ? (m0 / m.JhP) * m.It * m.CS2 + (m0 % m.JhP) : -1;
    }
    for (int e = tid; e < m.NB; e += C2_NT) {
      const int n0 = 16 * e, rs = n0 / m.It, it0 = n0 % m.It;
      bd[e] = rs * pl.Ih * DS + it0;
      bs[e] = (rs * m.PR + it0) * m.CS2;
    }
    for (int e = tid; e < m.KC * 4; e += C2_NT) {
      const int k0 = 32 * (e >> 2) + 8 * (e & 3), rs = k0 / m.It, i0 = k0 % m.It;
      ca[e] = rs * pl.Ih * DS + i0;
      cb[e] = rs * m.QR * m.CS1 + i0;
    }
    for (int e = tid; e < m.KD * 4; e += C2_NT) {
      const int k0 = 32 * (e >> 2) + 8 * (e & 3);
      da[e] = k0 < nb * m.JhP ? (k0 / m.JhP) * m.PR * m.CS2 + (k0 % m.JhP) : 0;
    }
    tA[mi] = a_; tBm[mi] = bm; tBd[mi] = bd; tBs[mi] = bs; tCa[mi] = ca; tCb[mi] = cb; tDa[mi] = da;
    Xs[mi] = reinterpret_cast<_Float16*>(c2_smem + m.l_xs);
    xpl[mi] = m.xs_rows * m.XS;
    // head fragments -> LDS (fragment order: linear copy)
    {
      _Float16* dst = reinterpret_cast<_Float16*>(c2_smem + m.l_ghf);
      const int n16 = m.QT * m.KB * 2 * 64;      // 16-byte units
      for (int i = tid; i < n16; i += C2_NT)
        reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(g.a[mi].ghf)[i];
      GhF[mi] = dst;
    }
    // tail fragments -> registers (tile pt = wave + 8 wa)
#pragma unroll
    for (int wa = 0; wa < WA; ++wa) {
      const int pt = wave + C2_NW * wa;
      const int m0 = 16 * pt + 4 * gq;
      moffA[mi][wa] = (pt < m.PT && m0 < m.P) ? (m0 / m.It) * m.JhP * m.CS1 + (m0 % m.It) : -1;
#pragma unroll
      for (int kb = 0; kb < KAM; ++kb)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          xh8 v = {};
          if (pt < m.PT && kb < m.KA) v = *reinterpret_cast<const xh8*>(g.a[mi].gtf + ((size_t)((pt * m.KA + kb) * 2 + p) * 64 + lane) * 8);
          gta[mi][wa][kb][p] = v;
        }
    }
    // accumulator units u = wave + 8 i: C tiles (mt, qt) first, then D tiles (pt, jt)
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int u = wave + C2_NW * i;
      acc[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u < m.nC) {
        uA[mi][i] = (16 * (u / m.QT) + c) * DS;                 // dy image row of i_h = 16 mt + c
        uB[mi][i] = (16 * (u % m.QT) + c) * m.CS1;              // C1 image row of q = 16 qt + c
      } else {
        const int v = u - m.nC;
        uA[mi][i] = (16 * (v / m.JtT) + c) * m.CS2;             // dC1 image row of p = 16 pt + c
        uB[mi][i] = qq * m.XS + 16 * (v % m.JtT) + 4 * pp;      // transposed read of the x image: (row qq, columns 4 pp ..)
      }
    }
    const int* h = g.a[mi].hdr;
    sxf[mi] = ldexpf(1.f, 14 - h[0]);
    sdf = ldexpf(1.f, 14 - h[1]);                               // (the same dy, the same bound for both matrices)
    eA[mi] = h[2] + h[0] - h[4] - 14;
    eB[mi] = h[3] + h[1] - h[5] - 14;
  }

  // ---- staging plan: dy quads id = tid + 512 e < nb*OUT/4; x quads id < nb*in/4 ---------------------------------------
  const int oq = OUT / 4;
  int dyo[EQ], dyg[EQ], dyr[EQ];
  f32x4 sd[EQ], dbs[EQ];
#pragma unroll
  for (int e = 0; e < EQ; ++e) {
    const int id = tid + C2_NT * e;
    const bool on = id < nb * oq;
    const int row = on ? id / oq : 0, o = on ? 4 * (id % oq) : 0;
    dyr[e] = on ? row : -1;
    dyg[e] = row * OUT + o;
    dyo[e] = (row * pl.Ih + o / pl.It) * DS + (o % pl.It);
    dbs[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    sd[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int xo[NMAT][2], xcol[NMAT][2], xr[NMAT][2];
  f32x4 sx[NMAT][2];
#pragma unroll
  for (int mi = 0; mi < NMAT; ++mi) {
    const C2Mat& m = pl.m[mi];
    const int iq = m.in / 4;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int id = tid + C2_NT * e;
      const bool on = e < m.EX && id < nb * iq;
      const int row = on ? id / iq : 0, col = on ? 4 * (id % iq) : 0;
      xr[mi][e] = on ? row : -1;
      xcol[mi][e] = col;
      xo[mi][e] = (row * m.JhP + col / m.Jt) * m.XS + (col % m.Jt);
      sx[mi][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const bool want_bias = g.bpart != nullptr;

  auto load_block = [&](long blk) {
    const long n0 = blk * nb;
#pragma unroll
    for (int e = 0; e < EQ; ++e)
      if (dyr[e] >= 0) {
        const long n = n0 + dyr[e];
        sd[e] = n < g.n_rows ? *reinterpret_cast<const f32x4*>(g.dy + (size_t)n0 * OUT + dyg[e]) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int mi = 0; mi < NMAT; ++mi) {
      const C2Mat& m = pl.m[mi];
      const C2MatArgs& a = g.a[mi];
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (xr[mi][e] >= 0) {
          const long n = n0 + xr[mi][e];
          f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
          if (n < g.n_rows) {
            if (a.T > 0) {
              const long b = n / a.T;
              const bool head = n == b * a.T;
              if (!head) v = *reinterpret_cast<const f32x4*>(a.x + (size_t)(n - 1) * m.in + xcol[mi][e]);
              else if (a.first) v = *reinterpret_cast<const f32x4*>(a.first + (size_t)b * m.in + xcol[mi][e]);
            } else {
              v = *reinterpret_cast<const f32x4*>(a.x + (size_t)n * m.in + xcol[mi][e]);
            }
          }
          sx[mi][e] = v;
        }
    }
  };
  auto store_block = [&]() {
#pragma unroll
    for (int e = 0; e < EQ; ++e)
      if (dyr[e] >= 0) {
        c2_store4(dYs + dyo[e], dYs + dpl + dyo[e], sd[e] * sdf);
        dbs[e] += sd[e];
      }
#pragma unroll
    for (int mi = 0; mi < NMAT; ++mi)
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (xr[mi][e] >= 0) c2_store4(Xs[mi] + xo[mi][e], Xs[mi] + xpl[mi] + xo[mi][e], sx[mi][e] * sxf[mi]);
  };

  if ((long)blockIdx.x < nblk) load_block(blockIdx.x);
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    store_block();
    if (blk + gridDim.x < nblk) load_block(blk + gridDim.x);
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < NMAT; ++mi) {
      const C2Mat& m = pl.m[mi];
      const _Float16* X0 = Xs[mi];
      const _Float16* X1 = Xs[mi] + xpl[mi];
      const int c1p = pl.c1_plane, dcp = pl.dc1_plane;
      // ---- phase A: C1[p][(row, j_h)] = Gt[p][j_t] x[(row, j_h)][j_t]  ->  C1 image [row][q][i_t] ----
#pragma unroll
      for (int wa = 0; wa < WA; ++wa) {
        const int pt = wave + C2_NW * wa;
        if (pt < m.PT) {
          for (int na = 0; na < m.NA; ++na) {
            f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
            const int xrow = (16 * na + c) * m.XS + 8 * gq;
#pragma unroll
            for (int kb = 0; kb < KAM; ++kb)
              if (kb < m.KA) {
                const xh8 b0 = c2_ld8(X0 + xrow + 32 * kb), b1 = c2_ld8(X1 + xrow + 32 * kb);
                r = c2_mma3(gta[mi][wa][kb][0], gta[mi][wa][kb][1], b0, b1, r);
              }
            const int off = tA[mi][16 * na + c];
            if (off >= 0 && moffA[mi][wa] >= 0) {
              f32x4 v;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = ldexpf(r[j], eA[mi]);
              c2_store4(C1s + off + moffA[mi][wa], C1s + c1p + off + moffA[mi][wa], v);
            }
          }
        }
      }
      // ---- phase B: dC1[q][(row, i_t)] = Gh^T[q][i_h] dy[row][i_h][i_t]  ->  dC1 image [row][p][j_h] ----
      for (int u = wave; u < m.QT * m.NB; u += C2_NW) {
        const int qt = u / m.NB, nt = u % m.NB;
        f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
        const int dbase = tBd[mi][nt] + (8 * gq + qq) * DS + 4 * pp;
        for (int kb = 0; kb < m.KB; ++kb) {
          const _Float16* af = GhF[mi] + ((size_t)((qt * m.KB + kb) * 2) * 64 + lane) * 8;
          const xh8 a0 = c2_ld8(af), a1 = c2_ld8(af + 512);
          const xh8 b0 = c2_tr8(dYs + dbase + 32 * kb * DS, DS), b1 = c2_tr8(dYs + dpl + dbase + 32 * kb * DS, DS);
          r = c2_mma3(a0, a1, b0, b1, r);
        }
        const int mo = tBm[mi][qt * 4 + gq];
        if (mo >= 0) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ldexpf(r[j], eB[mi]);
          const int off = tBs[mi][nt] + c * m.CS2 + mo;
          c2_store4(dC1s + off, dC1s + dcp + off, v);
        }
      }
      __syncthreads();
      // ---- phases C, D: the accumulator units of this wave ----
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        const int u = wave + C2_NW * i;
        if (u < m.nC) {
          for (int kb = 0; kb < m.KC; ++kb) {
            const int ka = tCa[mi][kb * 4 + gq], kbo = tCb[mi][kb * 4 + gq];
            const xh8 a0 = c2_ld8(dYs + uA[mi][i] + ka), a1 = c2_ld8(dYs + dpl + uA[mi][i] + ka);
            const xh8 b0 = c2_ld8(C1s + uB[mi][i] + kbo), b1 = c2_ld8(C1s + c1p + uB[mi][i] + kbo);
            acc[mi][i] = c2_mma3(a0, a1, b0, b1, acc[mi][i]);
          }
        } else if (u < m.NU) {
          for (int kb = 0; kb < m.KD; ++kb) {
            const int ka = tDa[mi][kb * 4 + gq];
            const xh8 a0 = c2_ld8(dC1s + uA[mi][i] + ka), a1 = c2_ld8(dC1s + dcp + uA[mi][i] + ka);
            const int xb = uB[mi][i] + (32 * kb + 8 * gq) * m.XS;
            const xh8 b0 = c2_tr8(X0 + xb, m.XS), b1 = c2_tr8(X1 + xb, m.XS);
            acc[mi][i] = c2_mma3(a0, a1, b0, b1, acc[mi][i]);
          }
        }
      }
      __syncthreads();
    }
  }

  // ---- partial sums out: [workgroup][unit][lane] f32x4 (fragment order; k_c2_reduce knows the maps) ----
#pragma unroll
  for (int mi = 0; mi < NMAT; ++mi) {
    const C2Mat& m = pl.m[mi];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int u = wave + C2_NW * i;
      if (u < m.NU) reinterpret_cast<f32x4*>(g.a[mi].part)[((size_t)blockIdx.x * m.NU + u) * 64 + lane] = acc[mi][i];
    }
  }
  if (want_bias) {          // column sums of dy: the block rows of a column meet in LDS, added in row order
    float* bl = reinterpret_cast<float*>(c2_smem + pl.l_dy);
#pragma unroll
    for (int e = 0; e < EQ; ++e)
      if (dyr[e] >= 0) *reinterpret_cast<f32x4*>(bl + dyg[e]) = dbs[e];
    __syncthreads();
    for (int o = tid; o < OUT; o += C2_NT) {
      float s = bl[o];
      for (int r = 1; r < nb; ++r) s += bl[r * OUT + o];
      g.bpart[(size_t)blockIdx.x * OUT + o] = s;
    }
  }
}

// ---- fixed-order reduction of the slabs; un-scale; scatter --------------------------------------------------------------------
// One workgroup per (matrix, unit): thread (lane, grp) sums the workgroups grp, grp + 4, ... in order, the four groups are added
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

__global__ void __launch_bounds__(256) k_c2_reduce(C2RedArgs args) {
  __shared__ f32x4 red[4][64];
  const int mi = (int)blockIdx.x >= args.nu0 ? 1 : 0;
  const C2Red& a = args.r[mi];
  const C2Mat& m = a.m;
  const int u = (int)blockIdx.x - (mi ? args.nu0 : 0);
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int w = grp; w < a.grid; w += 4) v += reinterpret_cast<const f32x4*>(a.part)[((size_t)w * m.NU + u) * 64 + lane];
  red[grp][lane] = v;
  __syncthreads();
  if (grp != 0) return;
  v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
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

// d_bias (+)= sum over the workgroups' partial column sums, in order; up to two destinations (the LSTM's two bias vectors)
__global__ void __launch_bounds__(256) k_c2_bias(const float* __restrict__ bpart, int grid, int OUT, float* __restrict__ d0,
                                                 float* __restrict__ d1) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= OUT) return;
  float s = 0.f;
  for (int w = 0; w < grid; ++w) s += bpart[(size_t)w * OUT + o];
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
      if (nmat == 2 && (p.m[0].Ih != p.m[1].Ih || p.m[0].It != p.m[1].It || p.m[0].out != p.m[1].out)) return false;
      p.OUT = p.m[0].out; p.Ih = p.m[0].Ih; p.It = p.m[0].It;
      if (p.OUT % 4 != 0) return false;
      p.DS = p.It + 8;
      p.dy_rows = nb * p.Ih + 32;
      p.EQ = c2_ceil(nb * p.OUT / 4, C2_NT);
      if (p.EQ > (big ? 8 : 4)) continue;
      // LDS carve-up
      size_t off = 0;
      p.l_dy = (int)off;
      size_t dyb = (size_t)2 * p.dy_rows * p.DS * 2;
      if (dyb < (size_t)nb * p.OUT * 4) dyb = (size_t)nb * p.OUT * 4;      // (the bias partials pass through this region at the end)
      off += (dyb + 15) & ~(size_t)15;
      int c1h = 0, dch = 0;
      for (int i = 0; i < nmat; ++i) {
        const C2Mat& m = p.m[i];
        if (nb * m.QR * m.CS1 > c1h) c1h = nb * m.QR * m.CS1;
        if (nb * m.PR * m.CS2 > dch) dch = nb * m.PR * m.CS2;
      }
      p.c1_plane = c1h; p.dc1_plane = dch;
      p.l_c1 = (int)off; off += ((size_t)2 * c1h * 2 + 15) & ~(size_t)15;
      p.l_dc1 = (int)off; off += ((size_t)2 * dch * 2 + 15) & ~(size_t)15;
      for (int i = 0; i < nmat; ++i) {
        C2Mat& m = p.m[i];
        m.l_xs = (int)off; off += ((size_t)2 * m.xs_rows * m.XS * 2 + 15) & ~(size_t)15;
        m.l_ghf = (int)off; off += (size_t)m.QT * m.KB * 2 * 1024;
        m.l_tab = (int)off;
        off += ((size_t)(m.NA * 16 + m.QT * 4 + 2 * m.NB + 2 * m.KC * 4 + m.KD * 4) * 4 + 15) & ~(size_t)15;
      }
      if (off > (size_t)C2_LDS_LIMIT) continue;
      p.lds = (int)off;
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
  if (pl.big) {
    auto fn = k_c2w<NMAT, 8, 2, 2, 8>;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(fn), pl.lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
    hipLaunchKernelGGL(fn, dim3(pl.grid), dim3(C2_NT), pl.lds, stream, a);
  } else {
    auto fn = k_c2w<NMAT, 4, 1, 1, 4>;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(fn), pl.lds) != TTRNN_OK) return TTRNN_ERR_LAUNCH;
    hipLaunchKernelGGL(fn, dim3(pl.grid), dim3(C2_NT), pl.lds, stream, a);
  }
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
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

size_t c2w_workspace_bytes(const TtShape* const* shapes, int nmat) {
  C2Plan pl;
  if (!c2_plan(&pl, shapes, nmat, device_cu_count())) return 0;
  return (size_t)pl.ws_bytes;
}

// x[i] / first[i] / T[i]: operand rows of matrix i (T > 0: the layer's outputs, read one step back; first = h_0 or NULL);
// x_cmax[i] (n = x_cn[i] bit patterns) / dy_cmax (n = OUT) optional bounds; d_packed[i] accumulated into; d_bias0 / d_bias1
// (either may be NULL) receive the column sums of dy.
int launch_c2w(const TtShape* const* shapes, int nmat, int64_t n_rows, const float* const* packed, const float* const* x,
               const float* const* first, const int* T, const float* dy, const unsigned* const* x_cmax, const int* x_cn,
               const unsigned* dy_cmax, float* const* d_packed, float* d_bias0, float* d_bias1, void* workspace,
               size_t workspace_bytes, hipStream_t stream) {
  C2Plan pl;
  const int cus = device_cu_count();
  if (!c2_plan(&pl, shapes, nmat, cus)) return TTRNN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < (size_t)pl.ws_bytes) return TTRNN_ERR_WORKSPACE;
  if (n_rows <= 0) return TTRNN_OK;
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
  hipLaunchKernelGGL(k_c2_prep, dim3(nmat), dim3(256), 0, stream, pa);
  int st = nmat == 2 ? c2_launch_main<2>(ka, stream) : c2_launch_main<1>(ka, stream);
  if (st != TTRNN_OK) return st;
  ra.nu0 = pl.m[0].NU;
  const int nu = pl.m[0].NU + (nmat == 2 ? pl.m[1].NU : 0);
  hipLaunchKernelGGL(k_c2_reduce, dim3(nu), dim3(256), 0, stream, ra);
  for (int i = 0; i < nmat; ++i) {
    const C2Mat& m = pl.m[i];
    if (m.d > 2) {
      const TtShape& s = *shapes[i];
      hipLaunchKernelGGL(k_c2_pull, dim3((unsigned)((s.wtotal + 255) / 256)), dim3(256), 0, stream, s, m, packed[i],
                         (const float*)(ws + m.w_dgh), (const float*)(ws + m.w_dgt), d_packed[i]);
    }
  }
  if (ka.bpart)
    hipLaunchKernelGGL(k_c2_bias, dim3((pl.OUT + 255) / 256), dim3(256), 0, stream, (const float*)ka.bpart, pl.grid, pl.OUT, d_bias0, d_bias1);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

}  // namespace ttrnn
