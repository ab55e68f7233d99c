// micro-benchmark: issue cost of the instruction kinds the ST-GCN kernels mix with fp32 MFMAs, one wave per SIMD
// (256-thread workgroups, one per CU): time per wave-instruction relative to v_mfma_f32_16x16x4_f32.
//   hipcc -O3 --offload-arch=gfx950 -o issue_costs issue_costs.hip && ./issue_costs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, float *out) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  f4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f4{0, 0, 0, 0};
  float x0 = a, x1 = b, x2 = a + b, x3 = a - b, x4 = 1.f, x5 = 2.f, x6 = 3.f, x7 = 4.f;
  f2 p0 = {a, b}, p1 = {b, a}, p2 = {a, a}, p3 = {b, b};
  unsigned addr = lane * 4, sa = (lane & 48) * 4;   // sa: 16 lanes share one address
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {        // 16 MFMAs, 4 independent accumulators
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    } else if (MODE == 1) { // 16 MFMAs, ONE accumulator (dependent chain)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
    } else if (MODE == 2) { // 16 MFMAs, two accumulators alternating
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[1], 0, 0, 0);
      }
    } else if (MODE == 3) { // 16 v_fma_f32, 8 independent chains
      asm volatile(REP16("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %2, %2, %1, %1\n\t") : "+v"(x0), "+v"(x1), "+v"(x2) : : );
    } else if (MODE == 4) { // 16 v_pk_fma_f32
      asm volatile(REP16("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %2, %2, %1, %1\n\t") : "+v"(p0), "+v"(p1), "+v"(p2) : : );
    } else if (MODE == 5) { // DPP add (row_shr)
      asm volatile(REP16("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t") : "+v"(x0), "+v"(x1) : : );
    } else if (MODE == 6) { // v_permlane32_swap
      asm volatile(REP16("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : );
    } else if (MODE == 7) { // ds_read_b32 conflict-free, 32 per wait
      asm volatile(REP16("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:256\n\t") "s_waitcnt lgkmcnt(0)" : "=v"(x0), "=v"(x1) : "v"(addr) : "memory");
    } else if (MODE == 8) { // ds_add_f32, 64 distinct addresses
      asm volatile(REP16("ds_add_f32 %0, %1\n\tds_add_f32 %0, %1 offset:256\n\t") "s_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(x0) : "memory");
    } else if (MODE == 9) { // ds_add_f32, 16 lanes per address (4 distinct addresses)
      asm volatile(REP16("ds_add_f32 %0, %1\n\tds_add_f32 %0, %1 offset:256\n\t") "s_waitcnt lgkmcnt(0)" : : "v"(sa), "v"(x0) : "memory");
    } else if (MODE == 10) { // v_mov_b32
      asm volatile(REP16("v_mov_b32 %0, %1\n\tv_mov_b32 %2, %3\n\t") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : );
    } else if (MODE == 11) { // 16 MFMA interleaved with 16 v_fma (same wave)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
          asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x4) : "v"(x5));
        }
    } else if (MODE == 12) { // 16 MFMA interleaved with 32 v_fma (same wave)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
          asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %2, %2, %1, %1" : "+v"(x4), "+v"(x6) : "v"(x5));
        }
    } else if (MODE == 13) { // 16 MFMA interleaved with 16 ds_read_b32
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
          asm volatile("ds_read_b32 %0, %1" : "=v"(x7) : "v"(addr) : "memory");
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (MODE == 14) { // 16 MFMA interleaved with 16 v_pk_fma
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
          asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(p1));
        }
    } else if (MODE == 15) { // 16 x v_mfma_f32_32x32x2_f32 worth the same FLOPs = 8 instrs, 2 accumulators
      typedef float f16v __attribute__((ext_vector_type(16)));
      static_assert(sizeof(f16v) == 64, "");
    }
  }
  float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p2.x + p3.y;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s + lds[lane];
}

template <int MODE>
float run(int iters, float *d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, 100, d);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, iters, d);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *d;
  hipMalloc(&d, 1024);
  const int iters = 200000;
  const char *names[] = {"16 mfma (4 acc)", "16 mfma (1 acc, dependent)", "16 mfma (2 acc)", "32 v_fma_f32", "32 v_pk_fma_f32",
                         "32 v_add dpp (+s_nop 1)", "32 v_permlane32_swap", "32 ds_read_b32", "32 ds_add_f32 distinct",
                         "32 ds_add_f32 16-way", "32 v_mov_b32", "16 mfma + 16 v_fma", "16 mfma + 32 v_fma", "16 mfma + 16 ds_read",
                         "16 mfma + 16 v_pk_fma"};
  float t[15];
  t[0] = run<0>(iters, d); t[1] = run<1>(iters, d); t[2] = run<2>(iters, d); t[3] = run<3>(iters, d); t[4] = run<4>(iters, d);
  t[5] = run<5>(iters, d); t[6] = run<6>(iters, d); t[7] = run<7>(iters, d); t[8] = run<8>(iters, d); t[9] = run<9>(iters, d);
  t[10] = run<10>(iters, d); t[11] = run<11>(iters, d); t[12] = run<12>(iters, d); t[13] = run<13>(iters, d); t[14] = run<14>(iters, d);
  const double mfma_ns = t[0] * 1e6 / iters / 16;     // per MFMA
  printf("one wave per SIMD; MFMA 16x16x4 f32 = %.2f ns each (= 32 cycles at %.2f GHz if the pipe is full)\n", mfma_ns, 32.0 / mfma_ns);
  for (int i = 0; i < 15; ++i)
    printf("%-30s %8.3f ms  %7.2f ns / iteration  = %6.2f MFMA-times\n", names[i], t[i], t[i] * 1e6 / iters, t[i] * 1e6 / iters / mfma_ns);
  return 0;
}
