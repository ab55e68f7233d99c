// embed.hip -- first layer of the ST-GCN embedding MLPs: pointwise Conv1d(3 -> 64), gfx950.
//
// Reference: `pos_embed[0]` / `sk_feat[0]` of models/p2rnet/modules/stgcn.py:46-63 (SingleConv 'cbr',
// sub_modules.py), applied to (B, 3, T*J) joint offsets and (B, 3, T*knn) hip-trajectory windows.
// A K=3 GEMM is a pure streaming problem (12 B in, 256 B out per point): the library GEMM / implicit-GEMM
// kernels it dispatches to spend 5x the HBM time on it.  Forward: each lane owns four consecutive points,
// keeps their 12 inputs in registers and writes the 64 output rows with 16-byte stores.  Weight/bias
// gradient: one workgroup per (sample, channel) row reduces dout against the three input rows; per-row
// partials are summed by the caller (deterministic).
#include "p2r_common.h"

namespace {

constexpr int EM_THREADS = 256;
constexpr int EM_C = 64;

__global__ __launch_bounds__(EM_THREADS) void embed3_fwd_kernel(int L, const float *__restrict__ x,
                                                                const float *__restrict__ W,
                                                                const float *__restrict__ bias,
                                                                float *__restrict__ out) {
  __shared__ float ws[EM_C * 4];
  for (int e = threadIdx.x; e < EM_C; e += EM_THREADS) {
    ws[4 * e + 0] = W[3 * e + 0]; ws[4 * e + 1] = W[3 * e + 1]; ws[4 * e + 2] = W[3 * e + 2];
    ws[4 * e + 3] = bias ? bias[e] : 0.f;
  }
  __syncthreads();
  const int n = blockIdx.y;
  const float *xn = x + (size_t)n * 3 * L;
  float *on = out + (size_t)n * EM_C * L;
  const int l0 = (blockIdx.x * EM_THREADS + threadIdx.x) * 4;
  if (l0 >= L) return;
  const bool vec = (L % 4 == 0) && l0 + 3 < L;
  float xv[3][4];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (vec) {
      const float4 v = *reinterpret_cast<const float4 *>(xn + (size_t)d * L + l0);
      xv[d][0] = v.x; xv[d][1] = v.y; xv[d][2] = v.z; xv[d][3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[d][i] = l0 + i < L ? xn[(size_t)d * L + l0 + i] : 0.f;
    }
  }
#pragma unroll 8
  for (int c = 0; c < EM_C; ++c) {
    const float4 w = *reinterpret_cast<const float4 *>(ws + 4 * c);
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = w.w + ((w.x * xv[0][i] + w.y * xv[1][i]) + w.z * xv[2][i]);
    if (vec) {
      *reinterpret_cast<float4 *>(on + (size_t)c * L + l0) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (l0 + i < L) on[(size_t)c * L + l0 + i] = o[i];
    }
  }
}

// partial[row = n*64 + c] = (sum dout*x0, sum dout*x1, sum dout*x2, sum dout)
__global__ __launch_bounds__(EM_THREADS) void embed3_wgrad_kernel(int L, const float *__restrict__ x,
                                                                  const float *__restrict__ dout,
                                                                  float4 *__restrict__ partial) {
  const int row = blockIdx.x, n = row / EM_C;
  const float *dr = dout + (size_t)row * L;
  const float *xn = x + (size_t)n * 3 * L;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (L % 4 == 0);
  const int L4 = vec ? L >> 2 : 0;
  for (int i = threadIdx.x; i < L4; i += EM_THREADS) {
    const float4 g = reinterpret_cast<const float4 *>(dr)[i];
    const float4 a = reinterpret_cast<const float4 *>(xn)[i];
    const float4 b = reinterpret_cast<const float4 *>(xn + L)[i];
    const float4 c = reinterpret_cast<const float4 *>(xn + 2 * (size_t)L)[i];
    s[0] += (g.x * a.x + g.y * a.y) + (g.z * a.z + g.w * a.w);
    s[1] += (g.x * b.x + g.y * b.y) + (g.z * b.z + g.w * b.w);
    s[2] += (g.x * c.x + g.y * c.y) + (g.z * c.z + g.w * c.w);
    s[3] += (g.x + g.y) + (g.z + g.w);
  }
  for (int i = (L4 << 2) + threadIdx.x; i < L; i += EM_THREADS) {
    const float g = dr[i];
    s[0] += g * xn[i]; s[1] += g * xn[L + i]; s[2] += g * xn[2 * (size_t)L + i]; s[3] += g;
  }
  __shared__ float red[EM_THREADS / 64][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s[q] += __shfl_xor(s[q], off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) red[threadIdx.x >> 6][q] = s[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < EM_THREADS / 64; ++w)
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] += red[w][q];
    partial[row] = make_float4(t[0], t[1], t[2], t[3]);
  }
}

}  // namespace

// x (N,3,L), W [64][3], bias [64] or NULL -> out (N,64,L) = W . x + bias
extern "C" int p2r_embed3_forward(int N, int L, const float *x, const float *W, const float *bias, float *out,
                                  void *stream) {
  if (N < 0 || L <= 0 || N > 65535) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  dim3 grid(p2r_cdiv(L, EM_THREADS * 4), N);
  hipLaunchKernelGGL(embed3_fwd_kernel, grid, dim3(EM_THREADS), 0, p2r_stream(stream), L, x, W, bias, out);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// partial [N*64][4] = per-(sample, channel) (dW[c][0], dW[c][1], dW[c][2], dbias[c]); the caller sums over N.
extern "C" int p2r_embed3_weight_grad(int N, int L, const float *x, const float *dout, float *partial,
                                      void *stream) {
  if (N < 0 || L <= 0) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  hipLaunchKernelGGL(embed3_wgrad_kernel, dim3(N * EM_C), dim3(EM_THREADS), 0, p2r_stream(stream), L, x, dout,
                     reinterpret_cast<float4 *>(partial));
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
