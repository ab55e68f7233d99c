// micro-benchmark: when does vector (v_fma_f32) work hide behind v_mfma_f32_16x16x32_f16 on gfx950?
// 512-thread workgroups (2 waves per SIMD), one per CU, 256 workgroups.  A "unit" = 12 MFMAs (4 accumulators x 3) +
// NV vector instructions, the shape of the fp16 plane-pair graph-conv unit (tools/ubench/gcn3h_proto.hip).
//   mode 0: waves 0-3 MFMA units only (waves 4-7 exit)            mode 1: waves 4-7 vector work only
//   mode 2: waves 0-3 MFMA only, waves 4-7 vector only, together   (cross-wave overlap?)
//   mode 3: waves 0-3 only, each unit = block of 12 MFMAs, then NV v_fma                    (one wave per SIMD, blocks)
//   mode 4: waves 0-3 only, each unit = 12 x (1 MFMA, NV/12 v_fma)                         (one wave per SIMD, interleaved)
//   mode 5: all 8 waves, blocks                                    mode 6: all 8 waves, interleaved
//   hipcc -O3 --offload-arch=gfx950 -o mfma16_valu_overlap tools/ubench/mfma16_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define MF(c) "v_mfma_f32_16x16x32_f16 %" #c ", %7, %8, %" #c "\n\t"
#define VF3 "v_fma_f32 %4, %4, %9, %10\n\tv_fma_f32 %5, %5, %9, %10\n\tv_fma_f32 %6, %6, %9, %10\n\t"

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(int units, float *out) {
  const int wave = threadIdx.x >> 6;
  f4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f4{0, 0, 0, 0};
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(i * 0.5f); }
  float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, m = 1.0001f, c = 0.5f;
  const bool mfma_wave = wave < 4;
  if ((MODE == 0 || MODE == 3 || MODE == 4) && !mfma_wave) return;
  if (MODE == 1 && mfma_wave) return;
  for (int u = 0; u < units; ++u) {
    if (MODE == 0 || (MODE == 2 && mfma_wave)) {
      asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3)
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(x0), "+v"(x1), "+v"(x2) : "v"(a), "v"(b), "v"(m), "v"(c));
    } else if (MODE == 1 || (MODE == 2 && !mfma_wave)) {
      asm volatile(VF3 VF3 VF3 VF3 VF3 VF3 VF3 VF3 VF3 VF3 VF3 VF3
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(x0), "+v"(x1), "+v"(x2) : "v"(a), "v"(b), "v"(m), "v"(c));
    } else if (MODE == 3 || MODE == 5) {
      asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3)
                   VF3 VF3 VF3 VF3 VF3 VF3 VF3 VF3 VF3 VF3 VF3 VF3
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(x0), "+v"(x1), "+v"(x2) : "v"(a), "v"(b), "v"(m), "v"(c));
    } else {
      asm volatile(MF(0) VF3 MF(1) VF3 MF(2) VF3 MF(3) VF3 MF(0) VF3 MF(1) VF3 MF(2) VF3 MF(3) VF3 MF(0) VF3 MF(1) VF3 MF(2) VF3 MF(3) VF3
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(x0), "+v"(x1), "+v"(x2) : "v"(a), "v"(b), "v"(m), "v"(c));
    }
  }
  float s = x0 + x1 + x2;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}
#undef MF
#undef VF3

template <int MODE>
float run(int units, float *d) {
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, 10, d); (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, units, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float *d; (void)hipMalloc(&d, 64);
  const int units = 20000;
  const float t0 = run<0>(units, d), t1 = run<1>(units, d), t2 = run<2>(units, d), t3 = run<3>(units, d), t4 = run<4>(units, d),
              t5 = run<5>(units, d), t6 = run<6>(units, d);
  const double cyc = 1.0;
  printf("per unit (12 MFMA 16x16x32 f16 + 36 v_fma), us per 1000 units:\n");
  printf("  MFMA waves alone (1 per SIMD)                 %8.3f   (%.1f cycles per MFMA at 2.0 GHz)\n", t0 * 1e3 / units * 1e3 * cyc, t0 * 1e-3 / units / 12 * 2.0e9);
  printf("  vector waves alone (1 per SIMD)               %8.3f\n", t1 * 1e3 / units * 1e3);
  printf("  MFMA wave + vector wave on every SIMD         %8.3f   (sum %.3f, max %.3f)\n", t2 * 1e6 / units, (t0 + t1) * 1e6 / units, (t0 > t1 ? t0 : t1) * 1e6 / units);
  printf("  one wave per SIMD, block of 12 then 36        %8.3f\n", t3 * 1e6 / units);
  printf("  one wave per SIMD, 12 x (1 MFMA, 3 v_fma)     %8.3f\n", t4 * 1e6 / units);
  printf("  two waves per SIMD, blocks (2 units per SIMD) %8.3f\n", t5 * 1e6 / units);
  printf("  two waves per SIMD, interleaved               %8.3f\n", t6 * 1e6 / units);
  return 0;
}
