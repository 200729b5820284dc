// ttrnn_f10.h — compile-time layout of the fused-core kernels (ttrnn_fast_f10.hip forward, ttrnn_fast_f10b.hip
// reverse-time): cores 1 and 0 of a d = 3 TT-matrix contracted into W10[(j0,j1,r2)][(i0,i1)].  Device-only (gfx950).
#pragma once
#include "ttrnn_mfma.h"
#include "ttrnn_split.h"

namespace ttrnn {

template <class S>
struct F10 {
  static constexpr int H = in_size_of<S>();       // input features (= hidden size for hidden-to-hidden matrices)
  static constexpr int HID = out_size_of<S>() / 4; // hidden units of the LSTM this matrix feeds (4 gates)
  static constexpr int J0 = S::J[0], J1 = S::J[1], I0 = S::I[0], I1 = S::I[1], I2 = S::I[2];
  static constexpr int R1 = S::R[1], R2 = S::R[2];
  static constexpr int K = J0 * J1 * R2;          // contraction length of the fused stage
  static constexpr int M = I0 * I1;               // output features of the fused stage (x I2 columns)
  static constexpr int MPG = M / 4;               // features per gate
  static constexpr int MT = M / 16;               // 16-feature tiles = active waves of phase B
  static constexpr int NM = K / 32;               // 32-wide k blocks
  static constexpr int PLANE = I2 * K;            // bf16 elements per plane of the [I2][K] image
  static constexpr int J2 = S::J[2];
  static constexpr int ROWS2 = J0 * J1;           // chain rows of S2
  static constexpr int M2 = I2 * R2;              // output features of S2
  static constexpr int MT2 = M2 / 16;             // S2 m-tiles
  static constexpr int XA = MT2 / 8;              // S2 m-tiles per wave (each over all chain-row tiles)
  static constexpr int RT2 = (ROWS2 + 15) / 16;   // chain-row tiles of S2
  static constexpr int XPL = RT2 * 16 * 8;        // bf16 elements per plane of the S2 operand [RT2*16 rows][8]: j2 and
                                                  // the rows zero-padded (hidden shapes: exactly the H values of h)
  // Order of the contraction index inside an image row.  The natural order (row2, r2) makes the 16 lanes of a
  // ds_write_b64 group (same q, 16 chain rows) hit only the first or only the second half of 16 slots: a 2-way bank
  // conflict on every store of phase A.  With two chain rows per 16-byte slot and the slots of one r2-quad contiguous —
  //     slot = (r2 >> 2) * ROWS2/2 + (row2 >> 1),   kk = 8*slot + (row2 & 1) * 4 + (r2 & 3)
  // — those 16 lanes fill 8 consecutive slots (128 contiguous bytes; the XOR swizzle of x_off keeps an aligned block
  // of 8 slots together).  The fused core's fragments are built in the same order (k_f10_prep).
  static constexpr int HR = ROWS2 / 2;            // slots per r2-quad
  __device__ static constexpr int kperm(int row2, int r2) {
    return ((r2 >> 2) * HR + (row2 >> 1)) * 8 + (row2 & 1) * 4 + (r2 & 3);
  }
};

// input-to-hidden matrices of an LSTM layer (any in_size whose last mode is <= 8): the batched kernels only
template <class S>
constexpr bool f10_in_ok() {
  using F = F10<S>;
  return S::D == 3 && S::R[0] == 1 && S::R[3] == 1 && shape_ok<S>() && F::K % 32 == 0 && F::M % 16 == 0 &&
         F::I2 <= 16 && F::MPG % 4 == 0 && F::MPG * F::I2 == F::HID && F::MT == 4 && F::MPG == 4 * F::MT &&
         out_size_of<S>() == 4 * F::HID && F::HID == 256 && S::R[2] % 4 == 0 && F::J2 <= 8 && F::ROWS2 <= 32 &&
         F::ROWS2 % 4 == 0 && F::M2 % 128 == 0 && F::NM % 2 == 0 && FAST_NW == 8;
}

// hidden-to-hidden matrices (recurrent kernels): in = hidden, last mode exactly 8
template <class S>
constexpr bool f10_ok() {
  using F = F10<S>;
  return S::D == 3 && S::R[0] == 1 && S::R[3] == 1 && shape_ok_recurrent<S>() && F::K % 32 == 0 && F::M % 16 == 0 &&
         F::I2 <= 16 && F::MPG % 4 == 0 && F::MPG * F::I2 == F::H && F::MT <= FAST_NW && F::MPG == 4 * F::MT &&
         out_size_of<S>() == 4 * F::H && S::R[2] % 4 == 0 && F::J2 == 8 && (F::ROWS2 == 32 || F::ROWS2 == 48 || F::ROWS2 == 64) && F::M2 % 128 == 0 &&
         FAST_NW == 8;      // (ROWS2 = 48 / 64: H = 384 / 512, the four-/six-/eight-wave kernel k_lstm_fwd_f10q only)
}

}  // namespace ttrnn
