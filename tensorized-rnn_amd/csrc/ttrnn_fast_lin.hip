// ttrnn_fast_lin.hip — shape-specialised batched TTLinear forward on fp32 MFMA (gfx950).
//
// y[n][out] = TT(cores) x[n][in] + bias for n_rows rows; used for
//   * the hoisted input projection of the recurrent kernels (all B*T rows in one launch on all CUs, output
//     gate-interleaved per hidden unit, see ttrnn_core.h:ilv_index), and
//   * TTLinear.forward itself (t3nsor/layers.py:121-127 -> ops.py:54-93) for the shapes in the table.
// A workgroup (8 waves) takes NB rows at a time: the chain rows of all NB samples are stacked in one LDS image
// per stage ([NB*ROWS_k][K_k], same XOR-swizzled layout and MFMA tile code as the recurrent kernel, core
// fragments resident in VGPRs), the last stage lands in an LDS tile that already has the final memory layout
// and leaves with coalesced 16-byte stores (+ bias).
#include <hip/hip_runtime.h>
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_mfma.h"
#include "ttrnn_lin_dev.h"

namespace ttrnn {

// G = 0: plain y[n][out];  G = 3/4: gate-interleaved y[n][H][4]   (body: ttrnn_lin_dev.h)
template <class S, int NB, int G, typename TI, typename TO>
__global__ void __launch_bounds__(FAST_NT) k_ttlinear_fwd_fast(int64_t n_rows, const float* __restrict__ packed,
                                                               const TI* __restrict__ bias,
                                                               const TI* __restrict__ x, TO* __restrict__ y,
                                                               int ilv_mode) {
  ttlinear_fwd_fast_body<S, NB, G, TI, TO>(n_rows, packed, bias, x, y, ilv_mode, (int)blockIdx.x, (int)gridDim.x);
}

// ---- dispatch --------------------------------------------------------------------------------------------
template <class S, int NB, int G, typename TI, typename TO>
static int launch_lin_t(int64_t n_rows, const float* packed, const void* bias, const void* x, void* y, int ilv_mode,
                        hipStream_t stream) {
  static_assert(shape_ok<S>(), "shape not supported by the MFMA path");
  const int64_t ntiles = (n_rows + NB - 1) / NB;
  const int grid = (int)(ntiles < 1 ? 1 : (ntiles > 1024 ? 1024 : ntiles));
  hipLaunchKernelGGL((k_ttlinear_fwd_fast<S, NB, G, TI, TO>), dim3(grid), dim3(FAST_NT), 0, stream, n_rows, packed,
                     (const TI*)bias, (const TI*)x, (TO*)y, ilv_mode);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

template <class S, int NB, int G>
static int launch_lin(int dtype, bool y_f32, int64_t n_rows, const float* packed, const void* bias, const void* x,
                      void* y, int ilv_mode, hipStream_t stream) {
  if (dtype == TTRNN_F32) return launch_lin_t<S, NB, G, float, float>(n_rows, packed, bias, x, y, ilv_mode, stream);
  if (y_f32) return launch_lin_t<S, NB, G, bf16_t, float>(n_rows, packed, bias, x, y, ilv_mode, stream);
  return launch_lin_t<S, NB, G, bf16_t, bf16_t>(n_rows, packed, bias, x, y, ilv_mode, stream);
}

// NB per shape: as many rows as fit 160 KB of LDS with the exact per-stage images
#define TT_LIN_SHAPES(X)      \
  X(ShpI1R8L, 16)             \
  X(ShpI1R8G, 16)             \
  X(ShpI40R16L, 4)            \
  X(ShpI1R4L, 32)             \
  X(ShpH256R8L, 4)            \
  X(ShpH256R8G, 4)            \
  X(ShpH256R16L, 2)           \
  X(ShpH256R16G, 2)           \
  X(ShpH128R4L, 16)           \
  X(ShpHd256R16, 4)

bool fast_ttlinear_fwd_available(const TtShape& s, int dtype, int ilv_h) {
  if (dtype != TTRNN_F32 && dtype != TTRNN_BF16) return false;
#define TT_X(SHAPE, NBV)                                                            \
  if (shape_matches<SHAPE>(s))                                                      \
    return ilv_h == 0 || (s.out_size % ilv_h == 0 && s.out_size / ilv_h == (out_size_of<SHAPE>() % 3 == 0 ? 3 : 4));
  TT_LIN_SHAPES(TT_X)
#undef TT_X
  return false;
}

int launch_ttlinear_fwd_fast(const TtShape& s, int dtype, bool y_f32, int64_t n_rows, const float* packed,
                             const void* bias, const void* x, void* y, int ilv_h, int ilv_mode, hipStream_t stream) {
  if (n_rows == 0) return TTRNN_OK;
  const int G = ilv_h > 0 ? s.out_size / ilv_h : 0;
#define TT_X(SHAPE, NBV)                                                                                        \
  if (shape_matches<SHAPE>(s)) {                                                                                \
    if (G == 0) return launch_lin<SHAPE, NBV, 0>(dtype, y_f32, n_rows, packed, bias, x, y, ilv_mode, stream);   \
    if constexpr (out_size_of<SHAPE>() % 3 == 0) {                                                              \
      if (G == 3) return launch_lin<SHAPE, NBV, 3>(dtype, true, n_rows, packed, bias, x, y, ilv_mode, stream);  \
    } else {                                                                                                    \
      if (G == 4) return launch_lin<SHAPE, NBV, 4>(dtype, true, n_rows, packed, bias, x, y, ilv_mode, stream);  \
    }                                                                                                           \
    return TTRNN_ERR_UNSUPPORTED;                                                                               \
  }
  TT_LIN_SHAPES(TT_X)
#undef TT_X
  return TTRNN_ERR_UNSUPPORTED;
}

}  // namespace ttrnn
