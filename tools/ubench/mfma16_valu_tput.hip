// micro-benchmark: do THROUGHPUT-bound vector instructions (independent v_fma_f32 / v_cvt_pk_f16_f32 / v_pk_fma_f32, 12 registers
// round-robin: no dependency stalls) overlap with v_mfma_f32_16x16x32_f16 on gfx950?  (mfma16_valu_overlap.hip asked the same
// with a latency-bound chain of three registers.)  Unit = 12 MFMAs (4 accumulators x 3) + 36 vector instructions.
//   hipcc -O3 --offload-arch=gfx950 -o mfma16_valu_tput tools/ubench/mfma16_valu_tput.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define MF(c) "v_mfma_f32_16x16x32_f16 %" #c ", %16, %17, %" #c "\n\t"
#define V1(r) "v_fma_f32 %" #r ", %" #r ", %18, %19\n\t"
#define VA V1(4) V1(5) V1(6)
#define VB V1(7) V1(8) V1(9)
#define VC V1(10) V1(11) V1(12)
#define VD V1(13) V1(14) V1(15)
#define OPS : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), \
              "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]) : "v"(a), "v"(b), "v"(m), "v"(c)

// MODE 0: MFMA only; 1: vector only; 2: block of 12 MFMAs then 36 vector; 3: 12 x (1 MFMA, 3 vector)
template <int MODE, int WAVES>
__global__ __launch_bounds__(512, 2) void k(int units, float *out) {
  const int wave = threadIdx.x >> 6;
  if (wave >= WAVES) return;
  f4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f4{0, 0, 0, 0};
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(i * 0.5f); }
  float x[12];
  for (int i = 0; i < 12; ++i) x[i] = threadIdx.x + i;
  float m = 1.0001f, c = 0.5f;
  for (int u = 0; u < units; ++u) {
    if (MODE == 0) asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) OPS);
    else if (MODE == 1) asm volatile(VA VB VC VD VA VB VC VD VA VB VC VD OPS);
    else if (MODE == 2) asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) VA VB VC VD VA VB VC VD VA VB VC VD OPS);
    else asm volatile(MF(0) VA MF(1) VB MF(2) VC MF(3) VD MF(0) VA MF(1) VB MF(2) VC MF(3) VD MF(0) VA MF(1) VB MF(2) VC MF(3) VD OPS);
  }
  float s = 0.f;
  for (int i = 0; i < 12; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}

template <int MODE, int WAVES>
float run(int units, float *d) {
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(512), 0, 0, 10, d); (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(512), 0, 0, units, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e6f / units;      // ns per unit
}
int main() {
  float *d; (void)hipMalloc(&d, 64);
  const int units = 20000;
  printf("ns per unit (12 MFMA 16x16x32 f16 + 36 INDEPENDENT v_fma_f32); one wave per SIMD | two waves per SIMD (each its own units)\n");
  printf("  MFMA only            %7.1f | %7.1f\n", run<0, 4>(units, d), run<0, 8>(units, d));
  printf("  vector only          %7.1f | %7.1f\n", run<1, 4>(units, d), run<1, 8>(units, d));
  printf("  block 12 + 36        %7.1f | %7.1f\n", run<2, 4>(units, d), run<2, 8>(units, d));
  printf("  12 x (1 MFMA, 3 vec) %7.1f | %7.1f\n", run<3, 4>(units, d), run<3, 8>(units, d));
  return 0;
}
