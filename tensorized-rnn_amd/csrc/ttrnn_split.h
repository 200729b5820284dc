// ttrnn_split.h — three-way bf16 splitting of fp32 operands for the bf16 MFMA (TTRNN_MATH_SPLIT), shared by the
// split-precision kernels.  Device-only (gfx950).
//
//     x = x0 + x1 + x2   with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)   (|x_i| <= 2^-9 |x_{i-1}|, RNE)
//     x*w = x0w0 + (x0w1 + x1w0) + (x0w2 + x1w1 + x2w0) + O(2^-26 |xw|)
// Every bf16*bf16 product is exact in fp32 and v_mfma_f32_16x16x32_bf16 accumulates in fp32: the six-term sum has a
// per-product relative error below fp32's rounding unit (2^-24) at 6*16 = 96 instead of 8*32 = 256 matrix-pipe cycles
// per 16x16x32 block of the fp32 MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include "ttrnn_mfma.h"

namespace ttrnn {

typedef __bf16 xbf8 __attribute__((ext_vector_type(8)));
typedef __bf16 xbf4 __attribute__((ext_vector_type(4)));

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
  else if constexpr (ns >= 16 && ns % 16 == 0) g = row & 15;      // (XOR inside aligned blocks of 16 slots: K = 384 has 48)
  return ((row * ns + (slot ^ g)) << 3) + (kk & 7);
}

__device__ __forceinline__ void split3(float v, __bf16& p0, __bf16& p1, __bf16& p2) {
  p0 = (__bf16)v;
  float r = v - (float)p0;      // exact
  p1 = (__bf16)r;
  r -= (float)p1;               // exact
  p2 = (__bf16)r;
}

// two elements at a time: v_cvt_pk_bf16_f32 rounds both (RNE); three packed dwords out.
typedef __bf16 xbf2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const xbf2 p = __builtin_convertvector(f32x2{a, b}, xbf2);
  return __builtin_bit_cast(unsigned, p);
}
// residual of a packed pair: v - (its bf16 piece) in ONE instruction, v_dot2c_f32_bf16 with the packed constants
// (lo -1, hi 0) / (lo 0, hi -1): pieces.lo * -1 + pieces.hi * 0 + v.  Bit-identical to shift / mask + subtract (checked
// on 2^20 pairs over 60 binades incl. zeros and denormals) at 7 instead of 11 VALU instructions per pair.  The constants
// are built from bits behind an opaque asm: hipcc folds a bf16x2 literal {-1, 0} into the INLINE constant -1.0, whose
// bits are (lo 0, hi -1) — the other piece.
__device__ __forceinline__ float bf16_residual(unsigned pieces, unsigned sel, float v) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(xbf2, pieces), __builtin_bit_cast(xbf2, sel), v, false);
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  unsigned sel_lo = 0x0000BF80u, sel_hi = 0xBF800000u;
  asm volatile("" : "+s"(sel_lo), "+s"(sel_hi));
  p0 = pk_bf16(a, b);
  float ra = bf16_residual(p0, sel_lo, a), rb = bf16_residual(p0, sel_hi, b);
  p1 = pk_bf16(ra, rb);
  ra = bf16_residual(p1, sel_lo, ra);
  rb = bf16_residual(p1, sel_hi, rb);
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

// ---- two-piece fp16 split (the fused-core LSTM forward kernels) ------------------------------------------------------
// v = p0 + p1 + e with p0 = fp16(v), p1 = fp16(v - p0), |e| <= 2^-23 |v| (22+ significand bits; fp32 carries 24) as long
// as p1 is a normal fp16 number; below that |e| <= 2^-25 in the SCALED domain the caller chooses (powers of two, exact).
// A product x*w is taken as x0 w0 + x0 w1 + x1 w0 on fp16 MFMAs with fp32 accumulation (error-compensated product of
// Ootomo & Yokota, "Recovering single precision accuracy from Tensor Cores ...", 2022: x1 w1 <= 2^-22 |x w| is dropped).
// Measured against a float64 evaluation the recurrence has the same error as a genuine fp32 evaluation (tools/
// split_precision_sim.py; tests/test_gpu_parity.py::test_split_math_error_vs_fp64_is_fp32_class, ::test_split_math_operand_ranges).
typedef _Float16 xh8 __attribute__((ext_vector_type(8)));
typedef _Float16 xh2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_f16(float a, float b) {          // v_cvt_pk_f16_f32 (round to nearest even)
  const xh2 p = __builtin_convertvector(f32x2{a, b}, xh2);
  return __builtin_bit_cast(unsigned, p);
}
// v - (one fp16 half of `pieces`), exact, ONE instruction: v_fma_mix_f32 reads the half as an fp16 source operand
__device__ __forceinline__ float f16_residual_lo(unsigned pieces, float v) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pieces), "v"(v));
  return r;
}
__device__ __forceinline__ float f16_residual_hi(unsigned pieces, float v) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pieces), "v"(v));
  return r;
}
__device__ __forceinline__ void split_pair_h(float a, float b, unsigned& p0, unsigned& p1) {
  p0 = pk_f16(a, b);
  p1 = pk_f16(f16_residual_lo(p0, a), f16_residual_hi(p0, b));
}
__device__ __forceinline__ void split2h(float v, _Float16& p0, _Float16& p1) {
  p0 = (_Float16)v;
  p1 = (_Float16)(v - (float)p0);
}
// The pieces of a product v s whose factor s is NOT a power of two (the fused core's rows carry the gate factors -log2 e / 2 log2 e):
// p0 = fp16(fl32(v s)), p1 = fp16(v s - p0) with the residual taken from the UNROUNDED product (one fma).  Under -ffp-contract=fast
// hipcc is free to form that fma or not wherever split2h(v * s) is inlined — k_f10h_prep did, another instantiation of the same
// expression (the fused set-up kernel) did not, and two of 16 384 fragment entries differed by an ulp.  Pinned here: the
// multiplication and the fma are opaque to the optimiser (an explicit fmaf was not enough: behind it the conversion to fp16 folded
// into v_fma_mixlo_f16 in one kernel and not in the other — one rounding against two, one entry of 16 384), so every kernel that
// builds these fragments agrees bit for bit, and with what k_f10h_prep has produced since round 3 (tools/ab_vs_r4.py).
__device__ __forceinline__ void split2h_scaled(float v, float s, _Float16& p0, _Float16& p1) {
  float t, r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(s));
  p0 = (_Float16)t;
  const float p0f = (float)p0;
  asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(r) : "v"(v), "v"(s), "v"(p0f));      // (opaque: an fptrunc(fma) folds into ONE v_fma_mixlo_f16,
  p1 = (_Float16)r;                                                            // i.e. one rounding instead of two, where hipcc sees it)
}
// four consecutive elements -> two 8-byte stores (planes 0 and 1)
__device__ __forceinline__ void store_split4_h(_Float16* img, int plane_elems, int off, f32x4 v) {
  unsigned a0, b0, a1, b1;
  split_pair_h(v[0], v[1], a0, b0);
  split_pair_h(v[2], v[3], a1, b1);
  *reinterpret_cast<u32x2*>(img + off) = u32x2{a0, a1};
  *reinterpret_cast<u32x2*>(img + plane_elems + off) = u32x2{b0, b1};
}

// MFMA term order of the six-term product, smallest terms first: (w2,x0) (w0,x2) (w1,x1) | (w1,x0) (w0,x1) | (w0,x0)
static constexpr int SPLIT_TW[6] = {2, 0, 1, 1, 0, 0};
static constexpr int SPLIT_TX[6] = {0, 2, 1, 0, 1, 0};

}  // namespace ttrnn
