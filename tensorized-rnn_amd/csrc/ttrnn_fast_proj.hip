// ttrnn_fast_proj.hip — the adjoint of "three TT cores -> dense matrix" as three small launches (gfx950).
//
// The dense-gradient backward (ttrnn_fast_gemm.hip) leaves dW[j][o] = x^T dy of a TTLinear as ONE dense matrix; the gradients
// of the cores are its image under the adjoint of (G0, G1, G2) -> W,
//     W[(j0,j1,j2)][(i0,i1,i2)] = sum_{r1,r2} G0[i0,j0,r1] G1[r1,i1,j1,r2] G2[r2,i2,j2]          (t3nsor/ops.py:54-93 on unit rows)
// Rounds 1-3 computed that image by running the fused-core weight-gradient kernel on the `in` unit rows with dW's rows as dy
// (k_ttlinear_wgrad_f10 + k_f10w_finish: 60 + 24 us for r = 16, 33 + 10 for r = 8 — one row per CU, a kernel built to
// stream 10^5 rows).  It is 10 M multiply-adds:
//     P[a][m][r2]   = sum_r1 G0[i0,j0,r1] G1[r1,i1,j1,r2]                 a = (j0,j1), m = (i0,i1)
//     dP[a][m][r2]  = sum_{j2,i2} dW[(a,j2)][(m,i2)] G2[r2,i2,j2]
//     dG2[r2,i2,j2] = sum_{a,m}   dW[(a,j2)][(m,i2)] P[a][m][r2]
//     dG0[i0,j0,r1] = sum_{j1,i1,r2} dP[a][m][r2] G1[r1,i1,j1,r2]         dG1[r1,i1,j1,r2] = sum_{j0,i0} dP[a][m][r2] G0[i0,j0,r1]
// all in fp32 FMAs (the kernel it replaces multiplied dC2 = W10 dy on three bf16 pieces), every sum in a fixed order (partials
// + a reduction, no atomics: repeatable bit for bit), accumulated INTO d_packed like every weight-gradient kernel of the library.
// Packed core layout (include/ttrnn.h): W_k[(j R_{k+1} + b) M_k + (i R_k + a)] = G_k[a, i, j, b], M_k = I_k R_k.
#include <hip/hip_runtime.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"

namespace ttrnn {

namespace {

constexpr int PJ_NCH = 32;        // chunks of the (a, m) sum of dG2

struct Proj3 {
  int J0, J1, J2, I0, I1, I2, R1, R2;
  int M0, M1, M2;
  long w0, w1, w2;
  int A, Mm;                      // A = J0 J1, Mm = I0 I1
  long out;                       // row stride of dW (= I0 I1 I2)
};

__device__ __forceinline__ float g0(const Proj3& p, const float* W, int i0, int j0, int r1) {
  return W[p.w0 + (long)(j0 * p.R1 + r1) * p.M0 + i0];
}
__device__ __forceinline__ float g1(const Proj3& p, const float* W, int r1, int i1, int j1, int r2) {
  return W[p.w1 + (long)(j1 * p.R2 + r2) * p.M1 + i1 * p.R1 + r1];
}

// one thread per (a, m, r2): P and dP, [A][Mm][R2] each.  (Four accumulators: the 128-term sum is a chain of dependent FMAs
// behind L2 loads otherwise — 30 us for cfg4's matrix.)
__global__ void __launch_bounds__(256) k_proj3_p(Proj3 p, const float* __restrict__ packed, const float* __restrict__ dW,
                                                 float* __restrict__ P, float* __restrict__ dP) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)p.A * p.Mm * p.R2;
  if (t >= n) return;
  const int r2 = (int)(t % p.R2);
  const long e = t / p.R2;
  const int m = (int)(e % p.Mm), a = (int)(e / p.Mm);
  const int i1 = m % p.I1, i0 = m / p.I1, j1 = a % p.J1, j0 = a / p.J1;
  float pv = 0.f;
  for (int r1 = 0; r1 < p.R1; ++r1) pv = fmaf(g0(p, packed, i0, j0, r1), g1(p, packed, r1, i1, j1, r2), pv);
  float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
  const int i2q = p.I2 & ~3;
  for (int j2 = 0; j2 < p.J2; ++j2) {
    const float* row = dW + ((long)a * p.J2 + j2) * p.out + (long)m * p.I2;
    const float* gc = packed + p.w2 + (long)j2 * p.M2 + r2;                    // G2[r2, i2, j2] = W2[j2 M2 + i2 R2 + r2]
    for (int i2 = 0; i2 < i2q; i2 += 4) {
      d0 = fmaf(row[i2], gc[(long)i2 * p.R2], d0);
      d1 = fmaf(row[i2 + 1], gc[(long)(i2 + 1) * p.R2], d1);
      d2 = fmaf(row[i2 + 2], gc[(long)(i2 + 2) * p.R2], d2);
      d3 = fmaf(row[i2 + 3], gc[(long)(i2 + 3) * p.R2], d3);
    }
    for (int i2 = i2q; i2 < p.I2; ++i2) d0 = fmaf(row[i2], gc[(long)i2 * p.R2], d0);
  }
  P[t] = pv;
  dP[t] = (d0 + d1) + (d2 + d3);
}

// workgroup (j2, chunk): partial dG2[j2][i2][r2] over the chunk's (a, m) pairs -> part[chunk][j2][i2 R2 + r2]
__global__ void __launch_bounds__(256) k_proj3_g2(Proj3 p, const float* __restrict__ dW, const float* __restrict__ P,
                                                  float* __restrict__ part) {
  const int j2 = blockIdx.x, ch = blockIdx.y;
  const long E = (long)p.A * p.Mm;
  const long e0 = E * ch / PJ_NCH, e1 = E * (ch + 1) / PJ_NCH;
  const int nout = p.I2 * p.R2;
  const long rstep = (long)p.J2 * p.out;                                        // dW rows a -> a + 1 (same j2)
  for (int o = threadIdx.x; o < nout; o += blockDim.x) {
    const int r2 = o % p.R2, i2 = o / p.R2;
    float acc0 = 0.f, acc1 = 0.f;
    // (a, m) walked without divisions: dW[(a J2 + j2) out + m I2 + i2], P[e R2 + r2]
    long a = e0 / p.Mm;
    int m = (int)(e0 % p.Mm);
    const float* dwp = dW + (a * p.J2 + j2) * p.out + i2;
    const float* pp = P + e0 * p.R2 + r2;
    long e = e0;
    for (; e + 1 < e1; e += 2) {
      const float w0 = dwp[(long)m * p.I2];
      int m1 = m + 1;
      const float* dwp1 = dwp;
      if (m1 == p.Mm) { m1 = 0; dwp1 += rstep; }
      const float w1 = dwp1[(long)m1 * p.I2];
      acc0 = fmaf(w0, pp[0], acc0);
      acc1 = fmaf(w1, pp[p.R2], acc1);
      pp += 2 * p.R2;
      m = m1 + 1;
      dwp = dwp1;
      if (m == p.Mm) { m = 0; dwp += rstep; }
    }
    if (e < e1) acc0 = fmaf(dwp[(long)m * p.I2], pp[0], acc0);
    part[((long)ch * p.J2 + j2) * nout + o] = acc0 + acc1;
  }
}

// d_packed += the cores' gradients.  Blocks [0, nb0): G0 — ONE WAVE per entry (its sum has J1 I1 R2 = 1 024 terms: a thread of
// its own took 130 us), lanes stride over the terms, fixed-order butterfly at the end; the other blocks: one thread per entry of
// G1 (J0 I0 terms) and G2 (the PJ_NCH partials)
__global__ void __launch_bounds__(256) k_proj3_fin(Proj3 p, int nb0, const float* __restrict__ packed, const float* __restrict__ dP,
                                                   const float* __restrict__ part, float* __restrict__ d_packed) {
  const long n0 = (long)p.I0 * p.J0 * p.R1, n1 = (long)p.R1 * p.I1 * p.J1 * p.R2, n2 = (long)p.R2 * p.I2 * p.J2;
  if ((int)blockIdx.x < nb0) {
    const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);                   // packed index (j0 R1 + r1) M0 + i0
    const int lane = threadIdx.x & 63;
    if (t >= n0) return;
    const int i0 = (int)(t % p.M0);
    const int r1 = (int)((t / p.M0) % p.R1), j0 = (int)(t / p.M0 / p.R1);
    const int nq = p.J1 * p.I1 * p.R2;
    float acc = 0.f;
    for (int q = lane; q < nq; q += 64) {
      const int r2 = q % p.R2, i1 = (q / p.R2) % p.I1, j1 = q / p.R2 / p.I1;
      acc = fmaf(dP[(((long)(j0 * p.J1 + j1)) * p.Mm + i0 * p.I1 + i1) * p.R2 + r2], g1(p, packed, r1, i1, j1, r2), acc);
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) acc += __shfl_xor(acc, sh);
    if (lane == 0) d_packed[p.w0 + t] += acc;
    return;
  }
  const long u0 = ((long)blockIdx.x - nb0) * blockDim.x + threadIdx.x;
  if (u0 < n1) {
    // packed index (j1 R2 + r2) M1 + i1 R1 + r1
    const long u = u0;
    const int r1 = (int)(u % p.R1), i1 = (int)((u / p.R1) % p.I1);
    const int r2 = (int)((u / p.M1) % p.R2), j1 = (int)(u / p.M1 / p.R2);
    float acc0 = 0.f, acc1 = 0.f;
    for (int j0 = 0; j0 < p.J0; ++j0) {
      const float* dp = dP + (((long)(j0 * p.J1 + j1)) * p.Mm + i1) * p.R2 + r2;
      int i0 = 0;
      for (; i0 + 1 < p.I0; i0 += 2) {
        acc0 = fmaf(dp[(long)i0 * p.I1 * p.R2], g0(p, packed, i0, j0, r1), acc0);
        acc1 = fmaf(dp[(long)(i0 + 1) * p.I1 * p.R2], g0(p, packed, i0 + 1, j0, r1), acc1);
      }
      if (i0 < p.I0) acc0 = fmaf(dp[(long)i0 * p.I1 * p.R2], g0(p, packed, i0, j0, r1), acc0);
    }
    d_packed[p.w1 + u] += acc0 + acc1;
  } else if (u0 < n1 + n2) {
    // packed index j2 M2 + i2 R2 + r2 = j2 nout + o
    const long u = u0 - n1;
    const int nout = p.I2 * p.R2;
    float acc = 0.f;
#pragma unroll 8
    for (int ch = 0; ch < PJ_NCH; ++ch) acc += part[(long)ch * p.J2 * nout + u];
    d_packed[p.w2 + u] += acc;
  }
}

// ---- d = 2 (round 4: cfg1's shape reaches the dense-gradient route): W[(j0,j1)][(i0,i1)] = sum_a G0[i0,j0,a] G1[a,i1,j1] ----
//     dG0[i0,j0,a] = sum_{i1,j1} dW[(j0,j1)][(i0,i1)] G1[a,i1,j1]        dG1[a,i1,j1] = sum_{i0,j0} dW[(j0,j1)][(i0,i1)] G0[i0,j0,a]
// ONE wave per core entry, lanes over the terms, butterfly at the end: fixed order, no atomics, accumulated INTO d_packed.
struct Proj2 {
  int J0, J1, I0, I1, R1, M0, M1;
  long w0, w1, out;
};
__global__ void __launch_bounds__(256) k_proj2(Proj2 p, const float* __restrict__ packed, const float* __restrict__ dW,
                                               float* __restrict__ d_packed) {
  const int lane = threadIdx.x & 63;
  const long e = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long n0 = (long)p.I0 * p.J0 * p.R1, n1 = (long)p.R1 * p.I1 * p.J1;
  if (e >= n0 + n1) return;
  float acc = 0.f;
  long dst;
  if (e < n0) {                                            // dG0: e = (j0 R1 + a) M0 + i0 (the packed order of W_0)
    const int i0 = (int)(e % p.M0), ja = (int)(e / p.M0), a = ja % p.R1, j0 = ja / p.R1;
    for (int t = lane; t < p.I1 * p.J1; t += 64) {
      const int i1 = t % p.I1, j1 = t / p.I1;
      acc = fmaf(dW[(long)(j0 * p.J1 + j1) * p.out + i0 * p.I1 + i1], packed[p.w1 + (long)j1 * p.M1 + i1 * p.R1 + a], acc);
    }
    dst = p.w0 + e;
  } else {                                                 // dG1: e - n0 = j1 M1 + i1 R1 + a (the packed order of W_1)
    const long f = e - n0;
    const int ia = (int)(f % p.M1), j1 = (int)(f / p.M1), a = ia % p.R1, i1 = ia / p.R1;
    for (int t = lane; t < p.I0 * p.J0; t += 64) {
      const int i0 = t % p.I0, j0 = t / p.I0;
      acc = fmaf(dW[(long)(j0 * p.J1 + j1) * p.out + i0 * p.I1 + i1], packed[p.w0 + (long)(j0 * p.R1 + a) * p.M0 + i0], acc);
    }
    dst = p.w1 + f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) d_packed[dst] += acc;
}
static bool proj2_shape(const TtShape& s, Proj2* p) {
  if (s.d != 2 || s.R[0] != 1 || s.R[2] != 1) return false;
  p->J0 = s.J[0]; p->J1 = s.J[1]; p->I0 = s.I[0]; p->I1 = s.I[1]; p->R1 = s.R[1];
  p->M0 = s.M[0]; p->M1 = s.M[1]; p->w0 = s.woff[0]; p->w1 = s.woff[1]; p->out = s.out_size;
  return true;
}

bool proj3_shape(const TtShape& s, Proj3* p) {
  if (s.d != 3 || s.R[0] != 1 || s.R[3] != 1) return false;
  p->J0 = s.J[0]; p->J1 = s.J[1]; p->J2 = s.J[2];
  p->I0 = s.I[0]; p->I1 = s.I[1]; p->I2 = s.I[2];
  p->R1 = s.R[1]; p->R2 = s.R[2];
  p->M0 = s.M[0]; p->M1 = s.M[1]; p->M2 = s.M[2];
  p->w0 = s.woff[0]; p->w1 = s.woff[1]; p->w2 = s.woff[2];
  p->A = s.J[0] * s.J[1]; p->Mm = s.I[0] * s.I[1];
  p->out = s.out_size;
  return true;
}


// ---- d = 4 (round 5) ------------------------------------------------------------------------------------------------------------
// W = G0 G1 (G2 G3): the last two cores contracted over r3 are ONE last core of modes (I2 I3, J2 J3),
//     Q[r2, (i2,i3), (j2,j3)] = sum_r3 G2[r2,i2,j2,r3] G3[r3,i3,j3],
// so the three launches above run on (G0, G1, Q) — on a scratch copy [W0 | W1 | WQ] of the packed cores and a zeroed scratch
// gradient — and a last launch adds dW0, dW1 to d_packed and pulls dQ back:
//     dG2[r2,i2,j2,r3] = sum_{i3,j3} dQ[r2,(i2,i3),(j2,j3)] G3[r3,i3,j3]          dG3[r3,i3,j3] = sum_{r2,i2,j2} dQ[..] G2[r2,i2,j2,r3]
// Every d = 4 matrix of the dense-gradient route came through the any-shape chain kernel on the `in` identity rows before (the joint
// matrix of a naive per-gate set of H = 512: 1.7 ms of its 9.6 ms training step; H = 768, d = 4: 1.0 of 8.2).
struct Proj4 {
  int I2, I3, J2, J3, R2, R3;
  int M2, M3;
  long w2, w3;                    // W2, W3 in the packed buffer
  long n01;                       // floats of W0 | W1 (= woff[2])
  long nq;                        // floats of WQ = J2 J3 I2 I3 R2
};

// scratch packed [W0 | W1 | WQ], scratch gradient zeroed
__global__ void __launch_bounds__(256) k_proj4_prep(Proj4 p, const float* __restrict__ packed, float* __restrict__ sp,
                                                    float* __restrict__ ds) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.n01 + p.nq) return;
  ds[t] = 0.f;
  if (t < p.n01) { sp[t] = packed[t]; return; }
  const long u = t - p.n01;                                  // WQ index j' MQ + i' R2 + r2
  const int MQ = p.I2 * p.I3 * p.R2;
  const int jq = (int)(u / MQ), rem = (int)(u % MQ);
  const int iq = rem / p.R2, r2 = rem % p.R2;
  const int i2 = iq / p.I3, i3 = iq % p.I3, j2 = jq / p.J3, j3 = jq % p.J3;
  float acc = 0.f;
  for (int r3 = 0; r3 < p.R3; ++r3)
    acc = fmaf(packed[p.w2 + (long)(j2 * p.R3 + r3) * p.M2 + i2 * p.R2 + r2], packed[p.w3 + (long)j3 * p.M3 + i3 * p.R3 + r3], acc);
  sp[t] = acc;
}

// d_packed[W0 | W1] += scratch; dG2: one thread per entry (I3 J3 terms); dG3: one wave per entry (R2 I2 J2 terms, butterfly)
__global__ void __launch_bounds__(256) k_proj4_fin(Proj4 p, int nb3, const float* __restrict__ packed, const float* __restrict__ ds,
                                                   float* __restrict__ d_packed) {
  const float* dq = ds + p.n01;                              // dQ[j' MQ + i' R2 + r2]
  const int MQ = p.I2 * p.I3 * p.R2;
  if ((int)blockIdx.x < nb3) {
    const long v = (long)blockIdx.x * 4 + (threadIdx.x >> 6);     // packed index j3 M3 + i3 R3 + r3
    const int lane = threadIdx.x & 63;
    const long n3 = (long)p.J3 * p.M3;
    if (v >= n3) return;
    const int j3 = (int)(v / p.M3), rem = (int)(v % p.M3), i3 = rem / p.R3, r3 = rem % p.R3;
    const int nt = p.R2 * p.I2 * p.J2;
    float acc = 0.f;
    for (int e = lane; e < nt; e += 64) {
      const int r2 = e % p.R2, i2 = (e / p.R2) % p.I2, j2 = e / p.R2 / p.I2;
      acc = fmaf(dq[(long)(j2 * p.J3 + j3) * MQ + (i2 * p.I3 + i3) * p.R2 + r2],
                 packed[p.w2 + (long)(j2 * p.R3 + r3) * p.M2 + i2 * p.R2 + r2], acc);
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) acc += __shfl_xor(acc, sh);
    if (lane == 0) d_packed[p.w3 + v] += acc;
    return;
  }
  const long t = ((long)blockIdx.x - nb3) * blockDim.x + threadIdx.x;
  const long n2 = (long)p.J2 * p.R3 * p.M2;
  if (t < p.n01) {
    d_packed[t] += ds[t];
  } else if (t < p.n01 + n2) {
    const long u = t - p.n01;                                // packed index (j2 R3 + r3) M2 + i2 R2 + r2
    const int rem = (int)(u % p.M2), i2 = rem / p.R2, r2 = rem % p.R2;
    const int jr = (int)(u / p.M2), r3 = jr % p.R3, j2 = jr / p.R3;
    float acc0 = 0.f, acc1 = 0.f;
    for (int j3 = 0; j3 < p.J3; ++j3) {
      const float* dr = dq + (long)(j2 * p.J3 + j3) * MQ + (long)(i2 * p.I3) * p.R2 + r2;
      const float* g3 = packed + p.w3 + (long)j3 * p.M3 + r3;
      int i3 = 0;
      for (; i3 + 1 < p.I3; i3 += 2) {
        acc0 = fmaf(dr[(long)i3 * p.R2], g3[(long)i3 * p.R3], acc0);
        acc1 = fmaf(dr[(long)(i3 + 1) * p.R2], g3[(long)(i3 + 1) * p.R3], acc1);
      }
      if (i3 < p.I3) acc0 = fmaf(dr[(long)i3 * p.R2], g3[(long)i3 * p.R3], acc0);
    }
    d_packed[p.w2 + u] += acc0 + acc1;
  }
}

// (G0, G1, Q) as a three-core shape over the scratch buffers, and the d = 4 bookkeeping
bool proj4_shape(const TtShape& s, Proj3* p, Proj4* q) {
  if (s.d != 4 || s.R[0] != 1 || s.R[4] != 1) return false;
  q->I2 = s.I[2]; q->I3 = s.I[3]; q->J2 = s.J[2]; q->J3 = s.J[3]; q->R2 = s.R[2]; q->R3 = s.R[3];
  q->M2 = s.M[2]; q->M3 = s.M[3]; q->w2 = s.woff[2]; q->w3 = s.woff[3];
  q->n01 = s.woff[2];
  q->nq = (long)s.J[2] * s.J[3] * s.I[2] * s.I[3] * s.R[2];
  p->J0 = s.J[0]; p->J1 = s.J[1]; p->J2 = s.J[2] * s.J[3];
  p->I0 = s.I[0]; p->I1 = s.I[1]; p->I2 = s.I[2] * s.I[3];
  p->R1 = s.R[1]; p->R2 = s.R[2];
  p->M0 = s.M[0]; p->M1 = s.M[1]; p->M2 = p->I2 * p->R2;
  p->w0 = s.woff[0]; p->w1 = s.woff[1]; p->w2 = s.woff[2];
  p->A = s.J[0] * s.J[1]; p->Mm = s.I[0] * s.I[1];
  p->out = s.out_size;
  return true;
}

}  // namespace

// workspace: P | dP ([A][Mm][R2] floats each) | the PJ_NCH partial dG2
size_t proj3_workspace_bytes(const TtShape& s) {
  Proj3 p;
  if (opt(OPT_DEV) & 1024) return 0;             // A/B: the fused-core weight-gradient kernel on the unit rows, as before
  Proj2 p2;
  if (proj2_shape(s, &p2)) return 256;           // d = 2: one launch, no scratch (a non-zero answer = "offered")
  Proj4 p4;
  size_t extra = 0;                               // d = 4: + the scratch packed cores and the scratch gradient [W0 | W1 | WQ] each
  if (proj4_shape(s, &p, &p4)) {
    if (opt(OPT_DEV) & (1 << 23)) return 0;       // A/B: the any-shape chain kernel on the identity rows, as before
    extra = 2 * (size_t)(p4.n01 + p4.nq);
  } else if (!proj3_shape(s, &p)) {
    return 0;
  }
  const size_t pe = (size_t)p.A * p.Mm * p.R2, g2 = (size_t)p.J2 * p.I2 * p.R2;
  return ((2 * pe + (size_t)PJ_NCH * g2 + extra) * sizeof(float) + 255) & ~(size_t)255;
}

// d_packed += adjoint of (cores -> dense)(dW), dW = fp32 [in_size][out_size]
int launch_proj3(const TtShape& s, const float* packed, const float* dW, float* d_packed, void* ws, hipStream_t stream) {
  Proj2 p2;
  if (proj2_shape(s, &p2)) {
    const long n = (long)p2.I0 * p2.J0 * p2.R1 + (long)p2.R1 * p2.I1 * p2.J1;
    hipLaunchKernelGGL(k_proj2, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, p2, packed, dW, d_packed);
    return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
  }
  Proj3 p;
  Proj4 p4;
  const bool four = proj4_shape(s, &p, &p4);
  if ((!four && !proj3_shape(s, &p)) || !ws) return TTRNN_ERR_UNSUPPORTED;
  const size_t pe = (size_t)p.A * p.Mm * p.R2;
  float* P = (float*)ws;
  float* dP = P + pe;
  float* part = dP + pe;
  float* d_real = d_packed;
  const float* packed_real = packed;
  if (four) {
    // the three launches run on the scratch copy [W0 | W1 | WQ] and accumulate into the zeroed scratch gradient
    float* sp = part + (size_t)PJ_NCH * p.J2 * p.I2 * p.R2;
    float* ds = sp + (p4.n01 + p4.nq);
    const long n = p4.n01 + p4.nq;
    hipLaunchKernelGGL(k_proj4_prep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p4, packed, sp, ds);
    packed = sp;
    d_packed = ds;
  }
  hipLaunchKernelGGL(k_proj3_p, dim3((unsigned)((pe + 255) / 256)), dim3(256), 0, stream, p, packed, dW, P, dP);
  hipLaunchKernelGGL(k_proj3_g2, dim3(p.J2, PJ_NCH), dim3(256), 0, stream, p, dW, (const float*)P, part);
  const long n0 = (long)p.I0 * p.J0 * p.R1, n12 = (long)p.R1 * p.I1 * p.J1 * p.R2 + (long)p.R2 * p.I2 * p.J2;
  const int nb0 = (int)((n0 + 3) / 4);
  hipLaunchKernelGGL(k_proj3_fin, dim3((unsigned)(nb0 + (n12 + 255) / 256)), dim3(256), 0, stream, p, nb0, packed,
                     (const float*)dP, (const float*)part, d_packed);
  if (four) {
    const long n3 = (long)p4.J3 * p4.M3, n2 = (long)p4.J2 * p4.R3 * p4.M2;
    const int nb3 = (int)((n3 + 3) / 4);
    hipLaunchKernelGGL(k_proj4_fin, dim3((unsigned)(nb3 + (p4.n01 + n2 + 255) / 256)), dim3(256), 0, stream, p4, nb3, packed_real,
                       (const float*)d_packed, d_real);
  }
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

}  // namespace ttrnn
