// tools/mfma_issue_bench.hip — development harness (not part of the library): how often can ONE wave issue
// v_mfma_f32_16x16x32_f16 / _32x32x16_f16, with one or four accumulator chains, with one or two waves per SIMD?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_issue_bench.hip -o tools/bin/mfma_issue_bench && tools/bin/mfma_issue_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 xh8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS, int SHAPE>
__global__ void __launch_bounds__(512) k(unsigned long long* out, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  xh8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16 big[2];
  for (int i = 0; i < 16; ++i) { big[0][i] = 0.f; big[1][i] = 0.f; }
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (SHAPE == 16) acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u % CHAINS], 0, 0, 0);
      else if constexpr (SHAPE == 1616) {      // v_mfma_f32_16x16x16_f16 (round 6: the reverse kernel's T1 of ttrnn_fast_w2.hip)
        typedef _Float16 xh4 __attribute__((ext_vector_type(4)));
        const xh4 a4 = xh4{a[0], a[1], a[2], a[3]}, b4 = xh4{b[0], b[1], b[2], b[3]};
        acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[u % CHAINS], 0, 0, 0);
      }
      else big[u % (CHAINS > 2 ? 2 : CHAINS)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big[u % (CHAINS > 2 ? 2 : CHAINS)], 0, 0, 0);
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  float s = 0.f;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][3];
  s += big[0][0] + big[1][5];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  if (s == 12345.f) sink[0] = s;
}

template <int CHAINS, int SHAPE>
void run(const char* name, int threads, unsigned long long* d, float* sink) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, SHAPE>), dim3(256), dim3(threads), 0, 0, d, sink, 100);      // warm-up
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<CHAINS, SHAPE>), dim3(256), dim3(threads), 0, 0, d, sink, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double flops = 256.0 * (threads / 64) * iters * 8.0 * 16384.0 * (SHAPE == 16 ? 1.0 : SHAPE == 1616 ? 0.5 : 2.0);
  printf("%-40s %d waves/SIMD: %.1f ticks per MFMA per wave, %.3f ms, %.0f TFLOP/s, %.2f GHz ticks\n", name, threads / 256,
         (double)h[0] / (iters * 8.0), ms, flops / ms * 1e-9, (double)h[0] / ms * 1e-6);
}

int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 4);
  for (int threads : {256, 512}) {
    run<1, 16>("16x16x32 f16, one accumulator chain", threads, d, sink);
    run<2, 16>("16x16x32 f16, two chains", threads, d, sink);
    run<4, 16>("16x16x32 f16, four chains", threads, d, sink);
    run<1, 1616>("16x16x16 f16, one accumulator chain", threads, d, sink);
    run<4, 1616>("16x16x16 f16, four chains", threads, d, sink);
    run<1, 32>("32x32x16 f16, one chain", threads, d, sink);
    run<2, 32>("32x32x16 f16, two chains", threads, d, sink);
  }
  return 0;
}
