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

// moments (optional) [gridDim.y * gridDim.x][10]: per workgroup (count, mean of the three INPUT coordinates, their six
// co-moments sum (x_d - mean_d)(x_e - mean_e), d <= e).  The output is an affine map of three inputs, so the batch
// statistics of all 64 output channels follow from these ten numbers (embed3_stats_kernel): the statistics pass over
// the (N, 64, L) output -- 21x the bytes -- is not needed.  Sums are taken about the workgroup's first point.
__global__ __launch_bounds__(EM_THREADS) void embed3_fwd_kernel(int L, const float *__restrict__ x,
                                                                const float *__restrict__ W,
                                                                const float *__restrict__ bias,
                                                                float *__restrict__ out, float *__restrict__ moments) {
  __shared__ float ws[EM_C * 4];
  __shared__ float red[EM_THREADS / 64][10];
  for (int e = threadIdx.x; e < EM_C; e += EM_THREADS) {
    ws[4 * e + 0] = W[3 * e + 0]; ws[4 * e + 1] = W[3 * e + 1]; ws[4 * e + 2] = W[3 * e + 2];
    ws[4 * e + 3] = bias ? bias[e] : 0.f;
  }
  __syncthreads();
  const int n = blockIdx.y;
  const float *xn = x + (size_t)n * 3 * L;
  float *on = out + (size_t)n * EM_C * L;
  const int l0 = (blockIdx.x * EM_THREADS + threadIdx.x) * 4;
  const bool live = l0 < L;
  if (!live && !moments) return;
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
  if (moments) {
    const int lb = blockIdx.x * EM_THREADS * 4;                 // first point of the workgroup: the pivot
    const float p0 = xn[lb], p1 = xn[(size_t)L + lb], p2 = xn[2 * (size_t)L + lb];
    float m[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (l0 + i < L) {
        const float a = xv[0][i] - p0, b = xv[1][i] - p1, c = xv[2][i] - p2;
        m[0] += 1.f; m[1] += a; m[2] += b; m[3] += c;
        m[4] += a * a; m[5] += a * b; m[6] += a * c; m[7] += b * b; m[8] += b * c; m[9] += c * c;
      }
#pragma unroll
    for (int q = 0; q < 10; ++q) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) m[q] += __shfl_xor(m[q], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int q = 0; q < 10; ++q) red[threadIdx.x >> 6][q] = m[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float t[10];
#pragma unroll
      for (int q = 0; q < 10; ++q) t[q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
      const float cnt = t[0], ia = t[1] / cnt, ib = t[2] / cnt, ic = t[3] / cnt;
      float *o = moments + ((size_t)n * gridDim.x + blockIdx.x) * 10;
      o[0] = cnt; o[1] = p0 + ia; o[2] = p1 + ib; o[3] = p2 + ic;
      o[4] = t[4] - t[1] * ia; o[5] = t[5] - t[1] * ib; o[6] = t[6] - t[1] * ic;
      o[7] = t[7] - t[2] * ib; o[8] = t[8] - t[2] * ic; o[9] = t[9] - t[3] * ic;
    }
    if (!live) return;
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

// part [P][10] (embed3_fwd_kernel) -> stats [64][3] = (count, mean, M2) of every output channel c of W . x + bias:
// mean_c = bias_c + W_c . mean_x, M2_c = W_c^T Cov W_c with Cov the co-moment matrix of the inputs about their global
// mean (workgroup entries merged as M2 = sum [M2_p + n_p (mean_p - mean)(mean_p - mean)^T], in fp64).
__global__ __launch_bounds__(256) void embed3_stats_kernel(int P, const float *__restrict__ part,
                                                           const float *__restrict__ W, const float *__restrict__ bias,
                                                           float *__restrict__ stats) {
  __shared__ double red[4][10];
  __shared__ double tot[10];
  const int tid = threadIdx.x;
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int p = tid; p < P; p += 256) {
    const float *e = part + (size_t)p * 10;
    const double n = (double)e[0];
    a[0] += n; a[1] += n * (double)e[1]; a[2] += n * (double)e[2]; a[3] += n * (double)e[3];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a[q] += __shfl_xor(a[q], off, 64);
    if ((tid & 63) == 0) red[tid >> 6][q] = a[q];
  }
  __syncthreads();
  if (tid == 0) {
    const double n = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    tot[0] = n;
    for (int q = 1; q < 4; ++q) tot[q] = ((red[0][q] + red[1][q]) + (red[2][q] + red[3][q])) / n;
  }
  __syncthreads();
  const double m0 = tot[1], m1 = tot[2], m2 = tot[3];
  double c[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int p = tid; p < P; p += 256) {
    const float *e = part + (size_t)p * 10;
    const double n = (double)e[0], d0 = (double)e[1] - m0, d1 = (double)e[2] - m1, d2 = (double)e[3] - m2;
    c[0] += (double)e[4] + n * d0 * d0; c[1] += (double)e[5] + n * d0 * d1; c[2] += (double)e[6] + n * d0 * d2;
    c[3] += (double)e[7] + n * d1 * d1; c[4] += (double)e[8] + n * d1 * d2; c[5] += (double)e[9] + n * d2 * d2;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 6; ++q) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c[q] += __shfl_xor(c[q], off, 64);
    if ((tid & 63) == 0) red[tid >> 6][q] = c[q];
  }
  __syncthreads();
  if (tid < 6) tot[4 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  __syncthreads();
  if (tid < EM_C) {
    const double w0 = (double)W[3 * tid], w1 = (double)W[3 * tid + 1], w2 = (double)W[3 * tid + 2];
    const double mean = (bias ? (double)bias[tid] : 0.0) + w0 * m0 + w1 * m1 + w2 * m2;
    double M2 = w0 * w0 * tot[4] + w1 * w1 * tot[7] + w2 * w2 * tot[9] +
                2.0 * (w0 * w1 * tot[5] + w0 * w2 * tot[6] + w1 * w2 * tot[8]);
    if (M2 < 0.0) M2 = 0.0;
    stats[3 * tid] = (float)tot[0];
    stats[3 * tid + 1] = (float)mean;
    stats[3 * tid + 2] = (float)M2;
  }
}

// partial[row = n*64 + c] = (sum dout*x0, sum dout*x1, sum dout*x2, sum dout)
// z / coef (both or neither): dout is then the BatchNorm-backward form coef[0][c] dout + coef[1][c] z + coef[2][c] of a
// masked gradient and the layer's saved output z (csrc/embed_bwd.hip), formed on the fly.
__global__ __launch_bounds__(EM_THREADS) void embed3_wgrad_kernel(int L, const float *__restrict__ x,
                                                                  const float *__restrict__ dout,
                                                                  float4 *__restrict__ partial,
                                                                  const float *__restrict__ z,
                                                                  const float *__restrict__ coef) {
  const int row = blockIdx.x, n = row / EM_C, ch = row - n * EM_C;
  const float *dr = dout + (size_t)row * L;
  const float *zr = z ? z + (size_t)row * L : nullptr;
  const float ca = coef ? coef[ch] : 1.f, cb = coef ? coef[EM_C + ch] : 0.f, cc = coef ? coef[2 * EM_C + ch] : 0.f;
  const float *xn = x + (size_t)n * 3 * L;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (L % 4 == 0);
  const int L4 = vec ? L >> 2 : 0;
  for (int i = threadIdx.x; i < L4; i += EM_THREADS) {
    float4 g = reinterpret_cast<const float4 *>(dr)[i];
    if (zr) {
      const float4 zz = reinterpret_cast<const float4 *>(zr)[i];
      g.x = ca * g.x + cb * zz.x + cc; g.y = ca * g.y + cb * zz.y + cc;
      g.z = ca * g.z + cb * zz.z + cc; g.w = ca * g.w + cb * zz.w + cc;
    }
    const float4 a = reinterpret_cast<const float4 *>(xn)[i];
    const float4 b = reinterpret_cast<const float4 *>(xn + L)[i];
    const float4 c = reinterpret_cast<const float4 *>(xn + 2 * (size_t)L)[i];
    s[0] += (g.x * a.x + g.y * a.y) + (g.z * a.z + g.w * a.w);
    s[1] += (g.x * b.x + g.y * b.y) + (g.z * b.z + g.w * b.w);
    s[2] += (g.x * c.x + g.y * c.y) + (g.z * c.z + g.w * c.w);
    s[3] += (g.x + g.y) + (g.z + g.w);
  }
  for (int i = (L4 << 2) + threadIdx.x; i < L; i += EM_THREADS) {
    const float g = zr ? ca * dr[i] + cb * zr[i] + cc : dr[i];
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
  hipLaunchKernelGGL(embed3_fwd_kernel, grid, dim3(EM_THREADS), 0, p2r_stream(stream), L, x, W, bias, out,
                     (float *)nullptr);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// The same forward that also returns the batch statistics of its output for the BatchNorm that follows:
// stats [1][64][3] = (count, mean, M2) per output channel (the entry format of p2r_bn_finalize, width 3), derived from the
// moments of the three inputs -- no pass over the output.  scratch: N * ceil(L / 1024) * 10 floats.
extern "C" int p2r_embed3_forward_stats(int N, int L, const float *x, const float *W, const float *bias, float *out,
                                        float *scratch, float *stats, void *stream) {
  if (N <= 0 || L <= 0 || N > 65535 || !scratch || !stats) return P2R_EINVAL;
  dim3 grid(p2r_cdiv(L, EM_THREADS * 4), N);
  hipLaunchKernelGGL(embed3_fwd_kernel, grid, dim3(EM_THREADS), 0, p2r_stream(stream), L, x, W, bias, out, scratch);
  P2R_LAUNCH_CHECK();
  hipLaunchKernelGGL(embed3_stats_kernel, dim3(1), dim3(256), 0, p2r_stream(stream), (int)(grid.x * N), scratch, W, bias,
                     stats);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// partial [N*64][4] = per-(sample, channel) (dW[c][0], dW[c][1], dW[c][2], dbias[c]); the caller sums over N.
extern "C" int p2r_embed3_weight_grad(int N, int L, const float *x, const float *dout, float *partial,
                                      void *stream) {
  if (N < 0 || L <= 0) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  hipLaunchKernelGGL(embed3_wgrad_kernel, dim3(N * EM_C), dim3(EM_THREADS), 0, p2r_stream(stream), L, x, dout,
                     reinterpret_cast<float4 *>(partial), (const float *)nullptr, (const float *)nullptr);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// the same with the output gradient in its BatchNorm-backward form: dout := coef[0][c] g + coef[1][c] z + coef[2][c]
// (g, z (N,64,L); coef [3][64]), formed while g and z are read -- the gradient of the layer's output is never stored.
extern "C" int p2r_embed3_weight_grad_lazy(int N, int L, const float *x, const float *g, const float *z, const float *coef,
                                           float *partial, void *stream) {
  if (N < 0 || L <= 0 || !z || !coef) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  hipLaunchKernelGGL(embed3_wgrad_kernel, dim3(N * EM_C), dim3(EM_THREADS), 0, p2r_stream(stream), L, x, g,
                     reinterpret_cast<float4 *>(partial), z, coef);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
