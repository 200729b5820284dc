// tools/wgrad_bench.hip — development harness (not part of the library): times launch_dense_wgrad (two fp16 pieces, column
// maxima given) on synthetic operands with parts of the kernel switched off through option `dev` (bits 3..5), to see what
// a chunk's time is made of.   make -C tensorized-rnn_amd/csrc bench_wgrad && tools/bin/wgrad_bench [rows in out [bias 0|1]]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "ttrnn.h"
#include "ttrnn_launch.h"
#include "ttrnn_opts.h"
using namespace ttrnn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_fill(float* p, size_t n, unsigned seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)(i * 2654435761u) ^ seed;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  p[i] = ((h & 0xFFFFFF) / 8388608.0f - 1.0f) * scale;
}
__global__ void k_ones(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 1.0f; }
int main(int argc, char** argv) {
  long rows = argc > 1 ? atol(argv[1]) : 81920;
  int in = argc > 2 ? atoi(argv[2]) : 256, out = argc > 3 ? atoi(argv[3]) : 1024;
  const bool with_bias = argc > 4 && atoi(argv[4]) != 0;      // 4th argument 1: also produce the bias gradient (column sums of dy)
  float *x, *dy, *dW, *cm, *scr, *db = nullptr;
  if (with_bias) { CK(hipMalloc(&db, (size_t)out * 4)); CK(hipMemset(db, 0, (size_t)out * 4)); }
  CK(hipMalloc(&x, (size_t)rows * in * 4)); CK(hipMalloc(&dy, (size_t)rows * out * 4)); CK(hipMalloc(&dW, (size_t)in * out * 4));
  CK(hipMalloc(&cm, (size_t)(in + out) * 4));
  const size_t sb = dense_wgrad_scratch_bytes(in, out);
  CK(hipMalloc(&scr, sb));
  hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)rows * in + 255) / 256)), dim3(256), 0, 0, x, (size_t)rows * in, 1u, 1.0f);
  hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)rows * out + 255) / 256)), dim3(256), 0, 0, dy, (size_t)rows * out, 2u, 1.0f);
  hipLaunchKernelGGL(k_ones, dim3((in + out + 255) / 256), dim3(256), 0, 0, cm, in + out);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"full", "no MFMA", "no split/LDS store", "no loads", "no MFMA, no store", "loads only (no MFMA/store/LDS reads)"};
  const int devs[] = {0, 8, 16, 32, 24, 8 | 16 | 64};
  printf("rows %ld in %d out %d%s\n", rows, in, out, with_bias ? " + bias gradient" : "");
  for (int v = 0; v < 6; ++v) {
    opt_set("dev", devs[v]);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
      CK(hipEventRecord(e0, 0));
      int st = launch_dense_wgrad(TTRNN_F32, rows, in, out, x, dy, dW, db, 0, true, scr, (const unsigned*)cm, (const unsigned*)(cm + in));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      if (st != 0) { printf("launch failed %d\n", st); return 1; }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms < best) best = ms;
    }
    printf("%-40s %.3f ms\n", names[v], best);
  }
  opt_set("dev", 0);
  return 0;
}
