// ttrnn_head.hip — the TTLinear heads of the two callers of the path with their row-wise epilogues fused into one call
// (SURVEY.md 8(f) N1):
//   TTRNN_EPI_LOG_SOFTMAX   y = log_softmax(TT(x) + b)                experiments/digit_classification/mnist_classifier.py:55-57
//   TTRNN_EPI_RELU_L2NORM   u = relu(TT(x) + b), y = u / ||u||_2      experiments/speaker_verification/encoder/speaker_encoder.py:86-89
// The chain itself is ttrnn_ttlinear_forward / _backward (whatever kernel the shape routes to); the epilogue is a row-local
// reduction over <= a few hundred outputs — one wave per row, butterfly shuffles, fp32 — applied in place to the rows the
// chain kernel just wrote (they are still in L2), and its adjoint turns dy into the pre-activation gradient the chain's
// backward consumes.  aux[n] keeps the row's log-sum-exp / L2 norm for the backward pass.
#include <hip/hip_runtime.h>
#include <math.h>
#include "ttrnn.h"
#include "ttrnn_core.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"

namespace ttrnn {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// one wave per row, four rows per workgroup
template <typename TS>
__global__ void __launch_bounds__(256) k_head_epilogue(int mode, long n_rows, int out, TS* __restrict__ y, float* __restrict__ aux) {
  const int lane = threadIdx.x & 63;
  const long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= n_rows) return;
  TS* row = y + (size_t)n * out;
  if (mode == TTRNN_EPI_LOG_SOFTMAX) {
    float m = -INFINITY;
    for (int o = lane; o < out; o += 64) m = fmaxf(m, ld(row, o));
    m = wave_max(m);
    float s = 0.f;
    for (int o = lane; o < out; o += 64) s += expf(ld(row, o) - m);
    const float lse = m + logf(wave_sum(s));
    for (int o = lane; o < out; o += 64) st(row, o, ld(row, o) - lse);
    if (aux && lane == 0) aux[n] = lse;
  } else {
    float s = 0.f;
    for (int o = lane; o < out; o += 64) { const float u = fmaxf(ld(row, o), 0.f); s = fmaf(u, u, s); }
    const float nrm = sqrtf(wave_sum(s));
    for (int o = lane; o < out; o += 64) st(row, o, fmaxf(ld(row, o), 0.f) / nrm);      // 0 / 0 = NaN, as torch.norm + div
    if (aux && lane == 0) aux[n] = nrm;
  }
}

// dz (fp32) = adjoint of the epilogue applied to dy, from the SAVED outputs y
template <typename TS>
__global__ void __launch_bounds__(256) k_head_epilogue_bwd(int mode, long n_rows, int out, const TS* __restrict__ y,
                                                           const float* __restrict__ aux, const TS* __restrict__ dy,
                                                           float* __restrict__ dz) {
  const int lane = threadIdx.x & 63;
  const long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= n_rows) return;
  const TS* yr = y + (size_t)n * out;
  const TS* dr = dy + (size_t)n * out;
  float* zr = dz + (size_t)n * out;
  if (mode == TTRNN_EPI_LOG_SOFTMAX) {
    float s = 0.f;
    for (int o = lane; o < out; o += 64) s += ld(dr, o);
    s = wave_sum(s);
    for (int o = lane; o < out; o += 64) zr[o] = ld(dr, o) - expf(ld(yr, o)) * s;
  } else {
    float s = 0.f;
    for (int o = lane; o < out; o += 64) s = fmaf(ld(yr, o), ld(dr, o), s);
    s = wave_sum(s);
    const float inv = 1.0f / aux[n];
    for (int o = lane; o < out; o += 64) {
      const float yv = ld(yr, o);
      zr[o] = yv > 0.f ? (ld(dr, o) - yv * s) * inv : 0.f;
    }
  }
}

size_t al256h(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace
}  // namespace ttrnn

using namespace ttrnn;

extern "C" {

size_t ttrnn_head_workspace(const ttrnn_ttm* w, int64_t n_rows) {
  TtShape s;
  if (tt_shape_init(&s, w) != TTRNN_OK || n_rows < 0) return 0;
  return al256h(ttrnn_ttlinear_workspace(w, n_rows)) + al256h((size_t)n_rows * s.out_size * sizeof(float));
}

int ttrnn_head_forward(const ttrnn_ttm* w, int dtype, int epilogue, int64_t n_rows, const float* packed, const void* bias,
                       const void* x, void* y, float* aux, void* workspace, size_t workspace_bytes, void* stream) {
  if (epilogue != TTRNN_EPI_NONE && epilogue != TTRNN_EPI_LOG_SOFTMAX && epilogue != TTRNN_EPI_RELU_L2NORM)
    return TTRNN_ERR_UNSUPPORTED;
  TtShape s;
  int st = tt_shape_init(&s, w);
  if (st != TTRNN_OK) return st;
  // shapes on the any-shape forward kernel (the classifier heads: 256 -> 10 and the like): chain + epilogue in ONE launch (round 5;
  // option dev bit 18: the separate epilogue launch, A/B)
  if (epilogue != TTRNN_EPI_NONE && n_rows > 0 && (dtype == TTRNN_F32 || dtype == TTRNN_BF16) && packed && x && y &&
      s.out_size <= 32 &&      // (one thread per row in there: ~4 serial passes over `out` LDS values — fine for 10, not for 256)
      !(opt(OPT_DEV) & (1 << 18)) && (opt(OPT_FORCE_GENERIC) || !fast_ttlinear_fwd_available(s, dtype, 0))) {
    const LinPlan p = plan_ttlinear_fwd(s, n_rows);
    if (p.ws_bytes > 0 && (!workspace || workspace_bytes < p.ws_bytes)) return TTRNN_ERR_WORKSPACE;
    return launch_ttlinear_fwd(s, p, dtype, n_rows, packed, bias, x, y, workspace, (hipStream_t)stream, 0, 0, epilogue, aux);
  }
  st = ttrnn_ttlinear_forward(w, dtype, n_rows, packed, bias, x, y, workspace, workspace_bytes, stream);
  if (st != TTRNN_OK || epilogue == TTRNN_EPI_NONE || n_rows == 0) return st;
  const unsigned grid = (unsigned)((n_rows + 3) / 4);
  if (dtype == TTRNN_F32)
    hipLaunchKernelGGL(k_head_epilogue<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, epilogue, (long)n_rows, s.out_size,
                       (float*)y, aux);
  else
    hipLaunchKernelGGL(k_head_epilogue<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, epilogue, (long)n_rows, s.out_size,
                       (bf16_t*)y, aux);
  return hipGetLastError() == hipSuccess ? TTRNN_OK : TTRNN_ERR_LAUNCH;
}

int ttrnn_head_backward(const ttrnn_ttm* w, int dtype, int epilogue, int64_t n_rows, const float* packed, const void* x,
                        const void* y, const float* aux, const void* dy, void* dx, float* d_packed, float* d_bias,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (epilogue == TTRNN_EPI_NONE)
    return ttrnn_ttlinear_backward(w, dtype, dtype, n_rows, packed, x, dy, dx, d_packed, d_bias, workspace, workspace_bytes, stream);
  if (epilogue != TTRNN_EPI_LOG_SOFTMAX && epilogue != TTRNN_EPI_RELU_L2NORM) return TTRNN_ERR_UNSUPPORTED;
  TtShape s;
  int st = tt_shape_init(&s, w);
  if (st != TTRNN_OK) return st;
  if (n_rows < 0) return TTRNN_ERR_BAD_DESC;
  if (n_rows == 0) return TTRNN_OK;
  if (!y || !dy || (epilogue == TTRNN_EPI_RELU_L2NORM && !aux)) return TTRNN_ERR_NULL;
  const size_t lin = al256h(ttrnn_ttlinear_workspace(w, n_rows));
  if (!workspace || workspace_bytes < ttrnn_head_workspace(w, n_rows)) return TTRNN_ERR_WORKSPACE;
  float* dz = (float*)((char*)workspace + lin);
  const unsigned grid = (unsigned)((n_rows + 3) / 4);
  if (dtype == TTRNN_F32)
    hipLaunchKernelGGL(k_head_epilogue_bwd<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, epilogue, (long)n_rows,
                       s.out_size, (const float*)y, aux, (const float*)dy, dz);
  else
    hipLaunchKernelGGL(k_head_epilogue_bwd<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, epilogue, (long)n_rows,
                       s.out_size, (const bf16_t*)y, aux, (const bf16_t*)dy, dz);
  if (hipGetLastError() != hipSuccess) return TTRNN_ERR_LAUNCH;
  return ttrnn_ttlinear_backward(w, dtype, TTRNN_F32, n_rows, packed, x, dz, dx, d_packed, d_bias, workspace, lin, stream);
}

}  // extern "C"
