// micro-benchmark: what fp32 MFMA rate does the chip sustain, and at what shader clock, as a function of how densely
// the matrix pipe is used?  One wave per SIMD (a 100 KB LDS allocation keeps a second workgroup off the CU) issues
// blocks of 16 v_mfma_f32_16x16x4_f32 on random operands, IDLE cycles of s_nop between blocks: density =
// 512 / (512 + IDLE).  The shader clock is s_memtime cycles of one wave / elapsed time.  ~100 ms per point so that
// the power management settles.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/mfma_density tools/ubench/mfma_density.hip && tools/ubench/mfma_density
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int IDLE16>   // idle cycles after a block, in units of 16
__global__ __launch_bounds__(256) void kd(int iters, const float *in, float *out, unsigned long long *cyc) {
  extern __shared__ float lds[];
  f4 acc[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
  float a[4][4], b[4];
  for (int m = 0; m < 4; ++m)
    for (int s = 0; s < 4; ++s) a[m][s] = in[(threadIdx.x * 16 + m * 4 + s) & 4095];
  for (int s = 0; s < 4; ++s) b[s] = in[(threadIdx.x * 7 + s + 1) & 4095];
  if (threadIdx.x == 999) lds[0] = a[0][0];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %4, %20, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %20, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %12, %20, %2\n\tv_mfma_f32_16x16x4_f32 %3, %16, %20, %3\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %5, %21, %0\n\tv_mfma_f32_16x16x4_f32 %1, %9, %21, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %13, %21, %2\n\tv_mfma_f32_16x16x4_f32 %3, %17, %21, %3\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %6, %22, %0\n\tv_mfma_f32_16x16x4_f32 %1, %10, %22, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %14, %22, %2\n\tv_mfma_f32_16x16x4_f32 %3, %18, %22, %3\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %7, %23, %0\n\tv_mfma_f32_16x16x4_f32 %1, %11, %23, %1\n\t"
        "v_mfma_f32_16x16x4_f32 %2, %15, %23, %2\n\tv_mfma_f32_16x16x4_f32 %3, %19, %23, %3"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
        : "v"(a[0][0]), "v"(a[0][1]), "v"(a[0][2]), "v"(a[0][3]), "v"(a[1][0]), "v"(a[1][1]), "v"(a[1][2]), "v"(a[1][3]),
          "v"(a[2][0]), "v"(a[2][1]), "v"(a[2][2]), "v"(a[2][3]), "v"(a[3][0]), "v"(a[3][1]), "v"(a[3][2]), "v"(a[3][3]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
#pragma unroll
    for (int k = 0; k < IDLE16; ++k) asm volatile("s_nop 15");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
  if (blockIdx.x == 3 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int IDLE16>
void run(const float *in, float *out, unsigned long long *cyc, int blocks) {
  const int iters0 = 20000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(kd<IDLE16>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kd<IDLE16>, dim3(blocks), dim3(256), 100 * 1024, 0, iters0, in, out, cyc);
  hipDeviceSynchronize();
  // ~100 ms: 512 + 16 * IDLE16 cycles per iteration at ~2.2 GHz
  const int iters = (int)(0.1 * 2.2e9 / (512 + 16 * IDLE16));
  hipEventRecord(e0);
  hipLaunchKernelGGL(kd<IDLE16>, dim3(blocks), dim3(256), 100 * 1024, 0, iters, in, out, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double flops = (double)blocks * 4 * 16 * 2048.0 * iters;
  printf("idle %4d cycles/block  density %.3f  %8.2f ms  %7.1f TFLOP/s  shader clock %.3f GHz  pipe busy %.3f\n", 16 * IDLE16,
         512.0 / (512 + 16 * IDLE16), ms, flops / ms / 1e9, c / (ms * 1e6), 512.0 * iters / c);
}
int main(int argc, char **argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  std::vector<float> h(4096);
  srand(1);
  for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  float *in, *out; unsigned long long *cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4); hipMalloc(&cyc, 8);
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  run<0>(in, out, cyc, blocks);
  run<2>(in, out, cyc, blocks);
  run<4>(in, out, cyc, blocks);
  run<8>(in, out, cyc, blocks);
  run<16>(in, out, cyc, blocks);
  run<32>(in, out, cyc, blocks);
  run<0>(in, out, cyc, blocks);
  return 0;
}
