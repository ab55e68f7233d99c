// micro-benchmark: can an error-compensated SPLIT-bf16 product replace the exact-fp32 MFMA of
// the ST-GCN kernels?  NOT wired into the product: the exact-fp32 path stays the default and the headline (dtype f32).
//
// Unit = the graph conv's dense product  Y (64 x N) = W (64 x 64) . X (64 x N)  on a tile of N = 512 columns that sits
// in LDS as fp32, channel-major like the tensor ([channel][column], stride 514: conflict-free 4-byte reads), W held
// in registers as A operands.  Variants, all with fp32 accumulators:
//   f32   : v_mfma_f32_16x16x4_f32, 16 k-steps x 4 row tiles per 16-column tile           (what gcn3 / tconv3 run)
//   bf16x1: v_mfma_f32_16x16x32_bf16, operands rounded to bf16 once                        (rate reference only)
//   bf16x3: W = W1 + W2 (+ W3), X = X1 + X2 (+ X3) in bf16;  W1X1 + W1X2 + W2X1            (3 terms)
//   bf16x6: ... + W1X3 + W2X2 + W3X1                                                       (6 terms: every product
//           whose weight is >= 2^-16 of the leading one)
// W is split once per launch on the host side of the kernel (registers); X is split WHILE IT IS READ from LDS: 8 fp32
// values per lane and K = 32 -> three packed bf16 fragments (v_cvt_pk_bf16_f32, shift back, subtract).
// Reports: time per pass over the tile, fp32-equivalent TFLOP/s (2 x 64 x 64 x N per pass, whole chip), VGPRs, and the
// max error against a float64 product next to the fp32 MFMA's own error.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/mfma_bf16_split tools/ubench/mfma_bf16_split.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int C = 64;          // channels: rows of W, reduction length
constexpr int N = 512;         // columns of the tile
constexpr int RS = N + 2;      // LDS row stride (floats): rows 8 apart land 16 banks apart
constexpr int NW = 8;          // waves per workgroup; wave w owns the 16-column tiles w, w + 8, ...
constexpr int NT = N / 16 / NW;

__device__ __forceinline__ void split3(const float (&x)[8], bf8 &p1, bf8 &p2, bf8 &p3) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 a = (__bf16)x[i];
    const float r1 = x[i] - (float)a;
    const __bf16 b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    p1[i] = a; p2[i] = b; p3[i] = (__bf16)r2;
  }
}
__device__ __forceinline__ void split2(const float (&x)[8], bf8 &p1, bf8 &p2) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 a = (__bf16)x[i];
    p1[i] = a; p2[i] = (__bf16)(x[i] - (float)a);
  }
}

__device__ __forceinline__ void splith(const float (&x)[8], h8 &p1, h8 &p2) {      // x = p1 + 2^-11 p2 (22 significand bits)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 a = (_Float16)x[i];
    p1[i] = a; p2[i] = (_Float16)((x[i] - (float)a) * 2048.f);
  }
}

// MODE 0: f32 MFMA; 1: bf16 single term; 3: 3-term split; 6: 6-term split;
// MODE 13 / 14: fp16 TWO-part split with the residual scaled by 2^11 (fp16 has 11 significand bits: two parts carry 22 of
//   fp32's 24; the scaling keeps the residual out of fp16's subnormals), terms W1 X1 | W1 X2' + W2' X1 (| W2' X2') in
//   separate accumulators combined as hi + 2^-11 lo (+ 2^-22 lo2): 3 / 4 MFMAs per product and 4 bytes per element
// AGG: the B operand is a two-entry aggregate c0 x(col) + c1 x(col') built with FMAs from two LDS gathers per value, like
//      a (plane, joint) unit of the graph conv (average neighbour-list length 2.1) -- the split then follows the combine;
// REUSE: the (split) B operand feeds REUSE sets of A operands (3 = the taps of the temporal conv, whose B operand is the
//      input itself: one split serves three times the MFMAs).  Timing-only variants: the stored result is checked for
//      AGG = 0, REUSE = 1.
template <int MODE, int AGG = 0, int REUSE = 1>
__global__ __launch_bounds__(NW * 64, 1) void kern(int iters, const float *__restrict__ W, const float *__restrict__ X,
                                                   float *__restrict__ Y, int store) {
  extern __shared__ float xs[];                        // [C][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const float *xg = X + (size_t)blockIdx.x * C * N;
  for (int e = tid; e < C * N; e += NW * 64) xs[(e / N) * RS + e % N] = xg[e];
  __syncthreads();

  // one column tile's accumulators live at a time: the tile is passed over `iters` times (a rate benchmark: in the
  // product the accumulators of a wave's columns persist over the planes of the graph conv instead)
  float *yg = Y + (size_t)blockIdx.x * C * N;
  float checksum = 0.f;
  auto finish = [&](int t, const f4 (&acc)[4]) __attribute__((always_inline)) {
    if (store) {      // D: col = lane & 15, row = 4 (lane >> 4) + q
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) yg[(16 * m + 4 * g + q) * N + 16 * (wave + NW * t) + r] = acc[m][q];
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) checksum += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    }
  };

  if constexpr (MODE == 0) {
    // A operand of step s, row tile m: W[16 m + r][4 s + g]
    float a[4][16];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int s = 0; s < 16; ++s) a[m][s] = W[(16 * m + r) * C + 4 * s + g];
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      f4 acc[4] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
      const float *xb = xs + g * RS + 16 * (wave + NW * t) + r;      // B of step s: X[4 s + g][col]
      for (int it = 0; it < iters; ++it) {
        float b[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) b[s] = xb[4 * s * RS];
        if constexpr (AGG) {
#pragma unroll
          for (int s = 0; s < 16; ++s) b[s] = fmaf(xb[4 * s * RS + 16], 0.75f, b[s] * 1.25f);
        }
        asm volatile("" ::: "memory");      // the tile is re-read every pass, like a new tile would be
#pragma unroll
        for (int u = 0; u < REUSE; ++u)
#pragma unroll
          for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[s], acc[m], 0, 0, 0);
      }
      finish(t, acc);
    }
  } else if constexpr (MODE == 13 || MODE == 14) {
    h8 a1[2][4], a2[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = W[(16 * m + r) * C + 32 * h + 8 * g + i];
        splith(w, a1[h][m], a2[h][m]);
      }
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      f4 hi[4], lo[4], l2[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) hi[m] = lo[m] = l2[m] = f4{0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float *xb = xs + (32 * h + 8 * g) * RS + 16 * (wave + NW * t) + r;
          float x[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = xb[i * RS];
          if constexpr (AGG) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fmaf(xb[i * RS + 16], 0.75f, x[i] * 1.25f);
          }
          h8 b1, b2;
          splith(x, b1, b2);
#pragma unroll
          for (int u = 0; u < REUSE; ++u) {
            if constexpr (MODE == 14) {
#pragma unroll
              for (int m = 0; m < 4; ++m) l2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[h][m], b2, l2[m], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) lo[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[h][m], b2, lo[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) lo[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[h][m], b1, lo[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) hi[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[h][m], b1, hi[m], 0, 0, 0);
          }
        }
      }
      f4 acc[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = hi[m] + lo[m] * (1.f / 2048.f) + l2[m] * (1.f / 4194304.f);
      finish(t, acc);
    }
  } else {
    // A operand of K-half h, row tile m: W[16 m + r][32 h + 8 g + i], i < 8, split into bf16 planes
    constexpr int NP = MODE == 1 ? 1 : (MODE == 3 ? 2 : 3);
    bf8 a[NP][2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = W[(16 * m + r) * C + 32 * h + 8 * g + i];
        if constexpr (NP == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) a[0][h][m][i] = (__bf16)w[i];
        } else if constexpr (NP == 2) {
          split2(w, a[0][h][m], a[1][h][m]);
        } else {
          split3(w, a[0][h][m], a[1][h][m], a[2][h][m]);
        }
      }
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      f4 acc[4] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
      for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float *xb = xs + (32 * h + 8 * g) * RS + 16 * (wave + NW * t) + r;     // X[32 h + 8 g + i][col]
          float x[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = xb[i * RS];
          if constexpr (AGG) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fmaf(xb[i * RS + 16], 0.75f, x[i] * 1.25f);
          }
          bf8 b1, b2, b3;
          if constexpr (NP == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) b1[i] = (__bf16)x[i];
          } else if constexpr (NP == 2) {
            split2(x, b1, b2);
          } else {
            split3(x, b1, b2, b3);
          }
          // term-major, row tiles inside: consecutive MFMAs go to four different accumulators (no back-to-back
          // dependence); smallest terms first, the accumulator sees them before the leading product
#define TERM(ap, bp)                                                                                                 \
  _Pragma("unroll") for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ap][h][m], bp, acc[m], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < REUSE; ++u) {
            if constexpr (NP == 3) { TERM(0, b3) TERM(1, b2) TERM(2, b1) }
            if constexpr (NP >= 2) { TERM(0, b2) TERM(1, b1) }
            TERM(0, b1)
          }
#undef TERM
        }
      }
      finish(t, acc);
    }
  }

  if (!store && checksum == 123.456f) Y[0] = checksum;
}

template <int MODE, int AGG = 0, int REUSE = 1>
void run(const char *name, const float *W, const float *X, float *Y, const std::vector<double> &ref, double ref_max,
         int blocks, double base_ms[1]) {
  const size_t lds = (size_t)C * RS * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&kern<MODE, AGG, REUSE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncAttributes attr;
  hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&kern<MODE, AGG, REUSE>));
  // accuracy: one pass, stored
  hipMemset(Y, 0, (size_t)blocks * C * N * sizeof(float));
  hipLaunchKernelGGL((kern<MODE, AGG, REUSE>), dim3(blocks), dim3(NW * 64), lds, 0, 1, W, X, Y, 1);
  hipDeviceSynchronize();
  std::vector<float> y((size_t)C * N);
  hipMemcpy(y.data(), Y, y.size() * sizeof(float), hipMemcpyDeviceToHost);
  double err = 0;
  for (size_t i = 0; i < y.size(); ++i) err = fmax(err, fabs((double)y[i] - ref[i]));
  // rate: ~50 ms
  const int iters0 = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kern<MODE, AGG, REUSE>), dim3(blocks), dim3(NW * 64), lds, 0, iters0, W, X, Y, 0);
  hipDeviceSynchronize();
  float ms = 0;
  int iters = iters0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<MODE, AGG, REUSE>), dim3(blocks), dim3(NW * 64), lds, 0, iters, W, X, Y, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) iters = (int)(iters * 50.0 / ms) + 1;
  }
  const double flop = 2.0 * C * C * N * (double)iters * blocks * REUSE;
  const double tf = flop / ms / 1e9;
  if (MODE == 0) base_ms[0] = ms / iters;
  if (AGG == 0 && REUSE == 1)
    printf("%-22s %8.3f us/pass  %7.1f fp32-equivalent TFLOP/s  x%4.2f vs f32 MFMA   VGPRs %3d  max err %.3e (%.2e of range)\n",
           name, ms / iters * 1e3, tf, base_ms[0] / (ms / iters), attr.numRegs, err, err / ref_max);
  else
    printf("%-22s %8.3f us/pass  %7.1f fp32-equivalent TFLOP/s  x%4.2f vs f32 MFMA   VGPRs %3d\n", name, ms / iters * 1e3, tf,
           base_ms[0] / (ms / iters), attr.numRegs);
}

int main() {
  const int blocks = 256;
  std::vector<float> W((size_t)C * C), X((size_t)blocks * C * N);
  srand(7);
  auto rnd = []() { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0); };
  for (auto &v : W) v = rnd() * 0.3f;                                  // He-like scale for fan-in 64
  for (auto &v : X) v = rnd() * 2.0f + 0.3f * rnd();                   // activations, no special structure
  std::vector<double> ref((size_t)C * N);
  double ref_max = 0;
  for (int o = 0; o < C; ++o)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < C; ++k) s += (double)W[o * C + k] * (double)X[(size_t)k * N + n];       // block 0
      ref[(size_t)o * N + n] = s;
      ref_max = fmax(ref_max, fabs(s));
    }
  float *dW, *dX, *dY;
  hipMalloc(&dW, W.size() * sizeof(float)); hipMalloc(&dX, X.size() * sizeof(float)); hipMalloc(&dY, X.size() * sizeof(float));
  hipMemcpy(dW, W.data(), W.size() * sizeof(float), hipMemcpyHostToDevice);
  hipMemcpy(dX, X.data(), X.size() * sizeof(float), hipMemcpyHostToDevice);
  printf("Y (64 x %d) = W (64 x 64) . X per workgroup, %d workgroups of %d waves, X tile fp32 in LDS; |Y| max %.3f\n", N, blocks,
         NW, ref_max);
  double base[1] = {0};
  run<0>("f32", dW, dX, dY, ref, ref_max, blocks, base);
  run<1>("bf16x1", dW, dX, dY, ref, ref_max, blocks, base);
  run<3>("bf16x3", dW, dX, dY, ref, ref_max, blocks, base);
  run<6>("bf16x6", dW, dX, dY, ref, ref_max, blocks, base);
  run<13>("f16x3 (2-part)", dW, dX, dY, ref, ref_max, blocks, base);
  run<14>("f16x4 (2-part)", dW, dX, dY, ref, ref_max, blocks, base);
  printf("B operand = two-entry aggregate (graph-conv unit):\n");
  run<0, 1>("f32    + aggregate", dW, dX, dY, ref, ref_max, blocks, base);
  run<3, 1>("bf16x3 + aggregate", dW, dX, dY, ref, ref_max, blocks, base);
  run<6, 1>("bf16x6 + aggregate", dW, dX, dY, ref, ref_max, blocks, base);
  run<13, 1>("f16x3  + aggregate", dW, dX, dY, ref, ref_max, blocks, base);
  printf("one B operand, three A sets (temporal-conv taps):\n");
  run<0, 0, 3>("f32    x 3 taps", dW, dX, dY, ref, ref_max, blocks, base);
  run<3, 0, 3>("bf16x3 x 3 taps", dW, dX, dY, ref, ref_max, blocks, base);
  run<6, 0, 3>("bf16x6 x 3 taps", dW, dX, dY, ref, ref_max, blocks, base);
  run<13, 0, 3>("f16x3  x 3 taps", dW, dX, dY, ref, ref_max, blocks, base);
  return 0;
}
