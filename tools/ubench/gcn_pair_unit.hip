// micro-benchmark for the NEXT design of the graph-conv kernels (DESIGN.md section 5 "Round 5"; not in the product):
// one MFMA of K = 32 taking 16 channels x TWO PLANES.  Z(w) = sum_k W_k U_k(w) sums over planes, so
//     A = [W_k | W_k'] (16 rows x (2 x 16) channels),  B = [U_k(w); U_k'(w)] ((2 x 16) channels x 16 frames)
// puts two (plane, joint) units of today's schedule into one v_mfma_f32_16x16x32_f16 chain, with the 16-channel slices
// and 16-frame tiles of gcn3 unchanged.  The two halves of the wave build DIFFERENT aggregates: lanes 0-31 (k groups 0,1)
// plane k from its neighbour list, lanes 32-63 plane k' from its own -- per-lane gather offsets and coefficients.
// Measured here, per "step" = one 16-channel phase of two units (2 x 64 rows x 16 channels x 16 frames):
//   fp32 : today's form -- per unit 4 values per lane = two-entry aggregate (8 ds_read_b32 + FMAs), 16 v_mfma_f32_16x16x4_f32
//   f16x3: one pair -- 8 values per lane (16 ds_read_b32 + FMAs), split into two fp16 parts, 4 row tiles x 3 terms
//   bf16x6: the same with three bf16 parts, 4 row tiles x 6 terms
// X slice [16 channels][RS] fp32 in LDS as gcn3 has it; operands of W in registers (the kernels stream them from L2).
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/gcn_pair_unit tools/ubench/gcn_pair_unit.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

constexpr int RS = 848;          // floats per channel row of a slice (16 frames x 53 joints)
constexpr int V = 53, NW = 8, UNITS = 28;   // units per wave and phase (gcn3: ~57 per wave: 454 / 8); pairs = UNITS / 2

template <int MODE>   // 0 fp32, 1 f16x3, 2 bf16x6
__global__ __launch_bounds__(NW * 64, 1) void kern(int iters, const float *__restrict__ X, const float *__restrict__ W,
                                                   float *__restrict__ out) {
  extern __shared__ float xs[];                        // [16][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  for (int e = tid; e < 16 * RS; e += NW * 64) xs[e] = X[(size_t)blockIdx.x * 16 * RS + e];
  __syncthreads();
  float checksum = 0.f;
  if constexpr (MODE == 0) {
    // unit (k, w): lane (g, r = frame): b[s] = sum_j c_j X[4 s + g][r][v_j], s < 4; A[m][s] = W_k[16 m + r][4 s + g]
    float a[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int s = 0; s < 4; ++s) a[m][s] = W[(16 * m + r) * 16 + 4 * s + g];
    const float *xb = xs + g * RS + r * V;
    f4 acc[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        const int v0 = (u * 5 + wave) % (V - 1), v1 = v0 + 1;            // wave-uniform "neighbour list"
        float b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) b[s] = fmaf(xb[4 * s * RS + v1], 0.75f, xb[4 * s * RS + v0] * 1.25f);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[s], acc[m], 0, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) checksum += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  } else {
    // pair (k, k'; w): lane (kg = g, n = r): 8 values = channels 8 (g & 1) + i of plane g >> 1; list of the lane's half
    constexpr int NPARTS = MODE == 1 ? 2 : 3;
    h8 ah[2][4]; b8 ab[3][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = W[(16 * m + r) * 16 + 8 * (g & 1) + i] * (1.f + 0.1f * (g >> 1));
      if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const _Float16 p = (_Float16)w[i]; ah[0][m][i] = p; ah[1][m][i] = (_Float16)((w[i] - (float)p) * 2048.f); }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const __bf16 p = (__bf16)w[i]; const float r1 = w[i] - (float)p; const __bf16 q = (__bf16)r1;
          ab[0][m][i] = p; ab[1][m][i] = q; ab[2][m][i] = (__bf16)(r1 - (float)q);
        }
      }
    }
    const bool up = lane >= 32;
    const float *xb = xs + 8 * (g & 1) * RS + r * V;
    const float c0 = up ? 0.9f : 1.25f, c1 = up ? 1.1f : 0.75f;       // the half's coefficients (per lane, loop-invariant)
    f4 hi[4], lo[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) hi[m] = lo[m] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < UNITS / 2; ++u) {
        const int va = (u * 10 + wave) % (V - 1), vb = (u * 10 + 5 + wave) % (V - 1);   // plane k's / plane k''s first neighbour
        const int o = up ? vb : va;                                                    // per lane: one v_cndmask per pair
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fmaf(xb[i * RS + o + 1], c1, xb[i * RS + o] * c0);
        if constexpr (MODE == 1) {
          h8 b1, b2;
#pragma unroll
          for (int i = 0; i < 8; ++i) { const _Float16 p = (_Float16)x[i]; b1[i] = p; b2[i] = (_Float16)((x[i] - (float)p) * 2048.f); }
#pragma unroll
          for (int m = 0; m < 4; ++m) lo[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0][m], b2, lo[m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 4; ++m) lo[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1][m], b1, lo[m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 4; ++m) hi[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0][m], b1, hi[m], 0, 0, 0);
        } else {
          b8 b1, b2, b3;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const __bf16 p = (__bf16)x[i]; const float r1 = x[i] - (float)p; const __bf16 q = (__bf16)r1;
            b1[i] = p; b2[i] = q; b3[i] = (__bf16)(r1 - (float)q);
          }
#define T6(A_, B_, ACC) _Pragma("unroll") for (int m = 0; m < 4; ++m) ACC[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[A_][m], B_, ACC[m], 0, 0, 0);
          T6(0, b3, lo) T6(1, b2, lo) T6(2, b1, lo) T6(0, b2, lo) T6(1, b1, lo) T6(0, b1, hi)
#undef T6
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) checksum += hi[m][0] + hi[m][1] + lo[m][2] + lo[m][3];
    (void)NPARTS;
  }
  if (checksum == 123.456f) out[0] = checksum;
}

template <int MODE>
double run(const char *name, const float *X, const float *W, float *out, double base) {
  const int blocks = 256;
  const size_t lds = 16 * RS * sizeof(float);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(NW * 64), lds, 0, 50, X, W, out);
  hipDeviceSynchronize();
  int iters = 200; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(NW * 64), lds, 0, iters, X, W, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) iters = (int)(iters * 50.0 / ms) + 1;
  }
  const double us = ms / iters * 1e3;
  // work of one pass: UNITS units x 64 rows x 16 channels x 16 frames x 2 FLOP, per wave
  const double tf = 2.0 * UNITS * 64 * 16 * 16 * NW * blocks * (double)iters / ms / 1e9;
  printf("%-8s %7.3f us per phase of %d units per wave   %6.1f fp32-equivalent TFLOP/s   x%.2f\n", name, us, UNITS, tf,
         base > 0 ? base / us : 1.0);
  return us;
}

int main() {
  float *X, *W, *out;
  hipMalloc(&X, (size_t)256 * 16 * RS * 4); hipMalloc(&W, 64 * 16 * 4); hipMalloc(&out, 64);
  float *h = (float *)malloc((size_t)256 * 16 * RS * 4);
  srand(3);
  for (size_t i = 0; i < (size_t)256 * 16 * RS; ++i) h[i] = (float)(rand() / (double)RAND_MAX * 2 - 1);
  hipMemcpy(X, h, (size_t)256 * 16 * RS * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, h, 64 * 16 * 4, hipMemcpyHostToDevice);
  const double b = run<0>("fp32", X, W, out, 0);
  run<1>("f16x3", X, W, out, b);
  run<2>("bf16x6", X, W, out, b);
  return 0;
}
