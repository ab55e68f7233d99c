// micro-benchmark: do fp32 MFMAs of one wave overlap with VALU / LDS / SALU work of the OTHER wave on the same SIMD?
// 512-thread workgroups (2 waves per SIMD), one per CU.  Waves 0-3 run an MFMA-only loop; waves 4-7 run (a) nothing,
// (b) a v_fma_f32 loop, (c) a ds_read_b32 loop, (d) an s_add loop.  If the pipes are independent the MFMA waves keep
// their time in every variant.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(int iters, float *out, int other_iters) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  if (MODE == 5 || MODE == 6) {   // ONE kind of wave: 12 ds_read_b32 (MODE 6: + 12 dependent v_fma) around every 16 MFMAs
    if (wave >= 4) return;
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    unsigned addr = (threadIdx.x & 63) * 4;
    float r[12];
    for (int it = 0; it < iters * 4; ++it) {
      asm volatile("ds_read_b32 %0, %12\n\tds_read_b32 %1, %12 offset:256\n\tds_read_b32 %2, %12 offset:512\n\tds_read_b32 %3, %12 offset:768\n\t"
                   "ds_read_b32 %4, %12 offset:1024\n\tds_read_b32 %5, %12 offset:1280\n\tds_read_b32 %6, %12 offset:1536\n\tds_read_b32 %7, %12 offset:1792\n\t"
                   "ds_read_b32 %8, %12 offset:2048\n\tds_read_b32 %9, %12 offset:2304\n\tds_read_b32 %10, %12 offset:2560\n\tds_read_b32 %11, %12 offset:2816"
                   : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]) : "v"(addr));
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (MODE == 6) {
#pragma unroll
        for (int j = 0; j < 12; ++j) b = fmaf(r[j], 1e-9f, b);
      }
    }
    float s = b;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s + r[0] + r[11];
    return;
  }
  if (wave < 4) {
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
  } else {
    if (MODE == 1) {          // VALU
      float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
      for (int it = 0; it < other_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f); }
      }
      if (x0 + x1 + x2 + x3 == 123.456f) out[1] = x0;
    } else if (MODE == 2) {   // LDS reads
      float s = 0; int idx = threadIdx.x;
      for (int it = 0; it < other_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { s += lds[(idx + 64 * u) & 4095]; }
        idx += 1;
      }
      if (s == 123.456f) out[2] = s;
    } else if (MODE == 4) {   // pure LDS reads: no VALU consumer, fixed address register
      unsigned addr = (threadIdx.x & 63) * 4;
      float r0, r1, r2, r3;
      for (int it = 0; it < other_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                       : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr));
      }
      if (r0 + r1 + r2 + r3 == 123.456f) out[2] = r0;
    } else if (MODE == 3) {   // SALU
      int s = blockIdx.x;
      for (int it = 0; it < other_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) s = __builtin_amdgcn_readfirstlane(s * 3 + u);
      }
      if (s == 12345) out[3] = s;
    }
  }
}

template <int MODE>
float run(int iters, int other_iters, float *d) {
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, 10, d, 10); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, d, other_iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float *d; hipMalloc(&d, 64);
  const int iters = 4000;    // 4000 * 64 MFMAs per wave
  const double mfma_flop = 256.0 * 4 * iters * 64 * 2048;
  float t0 = run<0>(iters, 0, d);
  printf("MFMA only (4 waves/CU, one per SIMD):          %7.3f ms  %6.1f TFLOP/s\n", t0, mfma_flop / t0 / 1e9);
  // size the partner loops so that alone they take about as long as the MFMA loop
  for (int oi : {iters * 4, iters * 8}) {
    float tv = run<1>(0, oi, d), tl = run<2>(0, oi, d), ts = run<3>(0, oi / 4, d);
    float tv2 = run<1>(iters, oi, d), tl2 = run<2>(iters, oi, d), ts2 = run<3>(iters, oi / 4, d);
    printf("partner iters %d: VALU alone %7.3f ms, with MFMA %7.3f (sum %7.3f, max %7.3f)\n", oi, tv, tv2, tv + t0, tv > t0 ? tv : t0);
    printf("                   LDS  alone %7.3f ms, with MFMA %7.3f (sum %7.3f, max %7.3f)\n", tl, tl2, tl + t0, tl > t0 ? tl : t0);
    printf("                   SALU alone %7.3f ms, with MFMA %7.3f (sum %7.3f, max %7.3f)\n", ts, ts2, ts + t0, ts > t0 ? ts : t0);
  }
  {
    float tl = run<4>(0, iters * 4, d), tl2 = run<4>(iters, iters * 4, d);
    printf("pure ds_read_b32 partner (no VALU): alone %7.3f ms, with MFMA %7.3f (sum %7.3f)\n", tl, tl2, tl + t0);
    float t5 = run<5>(iters, 0, d), t6 = run<6>(iters, 0, d);
    printf("same wave: 12 ds_read around every 16 MFMAs: %7.3f ms (MFMA only %7.3f); + 12 dependent v_fma: %7.3f ms\n", t5, t0, t6);
  }
  return 0;
}
