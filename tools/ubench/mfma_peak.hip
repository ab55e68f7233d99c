// micro-benchmark: sustained fp32 MFMA rate for the instruction patterns used by the gcn kernels
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512, 2) void k16(int iters, float *out) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(512, 2) void k32(int iters, float *out) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 123.456f) out[0] = s;
}
template <typename F>
void run(const char *name, F launch, double flop_per_block_iter, int threads) {
  float *d; hipMalloc(&d, 4);
  const int iters = 2000, blocks = 256 * 4;
  launch(blocks, threads, 10, d); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); launch(blocks, threads, iters, d); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop_per_block_iter * blocks * iters / ms / 1e9);
}
int main() {
  // flops per block-iter: waves * 16 * NACC * flops_per_mfma
  run("16x16x4 acc4 512thr", [](int b, int t, int it, float *d) { hipLaunchKernelGGL(k16<4>, dim3(b), dim3(t), 0, 0, it, d); }, 8.0 * 16 * 4 * 2048, 512);
  run("16x16x4 acc8 512thr", [](int b, int t, int it, float *d) { hipLaunchKernelGGL(k16<8>, dim3(b), dim3(t), 0, 0, it, d); }, 8.0 * 16 * 8 * 2048, 512);
  run("16x16x4 acc4 256thr", [](int b, int t, int it, float *d) { hipLaunchKernelGGL(k16<4>, dim3(b), dim3(t), 0, 0, it, d); }, 4.0 * 16 * 4 * 2048, 256);
  run("32x32x2 acc2 512thr", [](int b, int t, int it, float *d) { hipLaunchKernelGGL(k32<2>, dim3(b), dim3(t), 0, 0, it, d); }, 8.0 * 16 * 2 * 4096, 512);
  run("32x32x2 acc4 256thr", [](int b, int t, int it, float *d) { hipLaunchKernelGGL(k32<4>, dim3(b), dim3(t), 0, 0, it, d); }, 4.0 * 16 * 4 * 4096, 256);
  return 0;
}
