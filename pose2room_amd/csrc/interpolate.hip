// interpolate.hip -- three_nn / three_interpolate (+grad) for gfx950.
//
// Replaces three_nn_kernel, three_interpolate_kernel and
// three_interpolate_grad_kernel (reference _ext-src/src/interpolate_gpu.cu:9-154).
// P2RNet itself never calls them (PointnetFPModule is API surface only), so
// they are built for the pointnet2_ops call surface and the stress shapes.
//
// MI355X design
//  three_nn          : thread per unknown point, grid over (cloud, 256-point
//                      tiles) instead of one block per cloud; the known cloud
//                      streams through LDS as float4 tiles that every lane reads
//                      at the same address (broadcast, no bank conflict).  The
//                      scan stays in ascending k with strict <, so ties resolve
//                      exactly as in the reference.  FP32-VALU bound.
//  three_interpolate : thread per output point j walking a channel chunk;
//                      idx/weight loaded once per thread, writes coalesced along
//                      j.  HBM bound.
//  grad              : destination rows (GC x m) accumulated in LDS with
//                      ds_add_f32 and written back whole (overwrite); global
//                      atomics only when a row does not fit LDS.
#include "p2r_common.h"

#include <algorithm>

namespace {

constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 1024;  // known points per LDS tile (16 KiB)

__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(
    int n, int m, int n_tiles, const float *__restrict__ unknown,
    const float *__restrict__ known, float *__restrict__ dist2, int *__restrict__ idx) {
  __shared__ float4 s_known[NN_TILE];
  const int batch = blockIdx.x / n_tiles;
  const int j = (blockIdx.x % n_tiles) * NN_THREADS + threadIdx.x;
  const float *u = unknown + (size_t)batch * n * 3;
  const float *kn = known + (size_t)batch * m * 3;

  const bool active = j < n;
  const float ux = active ? u[j * 3 + 0] : 0.f;
  const float uy = active ? u[j * 3 + 1] : 0.f;
  const float uz = active ? u[j * 3 + 2] : 0.f;
  // reference keeps best* as double 1e40 and compares the fp32 d against it;
  // that is the same predicate as fp32 compare against +inf, and (float)1e40
  // is +inf on output.
  float best1 = INFINITY, best2 = INFINITY, best3 = INFINITY;
  int besti1 = 0, besti2 = 0, besti3 = 0;

  for (int base = 0; base < m; base += NN_TILE) {
    const int tile = min(NN_TILE, m - base);
    if (base > 0) __syncthreads();
    for (int t = threadIdx.x; t < tile; t += NN_THREADS) {
      const float *p = kn + (size_t)(base + t) * 3;
      s_known[t] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
    if (active) {
#pragma unroll 4
      for (int t = 0; t < tile; ++t) {
        const float4 p = s_known[t];
        const float d = p2r_sqdist(ux, uy, uz, p.x, p.y, p.z);
        const int k = base + t;
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
    }
  }
  if (active) {
    float *d2 = dist2 + ((size_t)batch * n + j) * 3;
    int *id = idx + ((size_t)batch * n + j) * 3;
    d2[0] = best1; d2[1] = best2; d2[2] = best3;
    id[0] = besti1; id[1] = besti2; id[2] = besti3;
  }
}

constexpr int TI_THREADS = 256;
constexpr int TI_CCHUNK = 16;

__global__ __launch_bounds__(TI_THREADS) void three_interpolate_kernel(
    int c, int m, int n, int n_tiles, int c_tiles, const float *__restrict__ points,
    const int *__restrict__ idx, const float *__restrict__ weight, float *__restrict__ out) {
  int bid = blockIdx.x;
  const int nt = bid % n_tiles; bid /= n_tiles;
  const int ct = bid % c_tiles;
  const int batch = bid / c_tiles;
  const int j = nt * TI_THREADS + threadIdx.x;
  if (j >= n) return;
  const float *p = points + (size_t)batch * c * m;
  const int *id = idx + ((size_t)batch * n + j) * 3;
  const float *w = weight + ((size_t)batch * n + j) * 3;
  float *o = out + (size_t)batch * c * n;
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int i1 = id[0], i2 = id[1], i3 = id[2];
  const int l0 = ct * TI_CCHUNK, l1 = min(l0 + TI_CCHUNK, c);
#pragma unroll 4
  for (int l = l0; l < l1; ++l) {
    const float *row = p + (size_t)l * m;
    // reference order: p1*w1 + p2*w2 + p3*w3, no contraction
    o[(size_t)l * n + j] = row[i1] * w1 + row[i2] * w2 + row[i3] * w3;
  }
}

constexpr int TG_THREADS = 512;

__global__ __launch_bounds__(TG_THREADS) void three_interpolate_grad_lds_kernel(
    int c, int n, int m, int c_tiles, int GC, const float *__restrict__ grad_out,
    const int *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ grad_points) {
  extern __shared__ float s_acc[];  // [GC][m]
  const int ct = blockIdx.x % c_tiles;
  const int batch = blockIdx.x / c_tiles;
  const int l0 = ct * GC;
  const int nl = min(GC, c - l0);
  const float *g = grad_out + ((size_t)batch * c + l0) * n;
  const int *id = idx + (size_t)batch * n * 3;
  const float *w = weight + (size_t)batch * n * 3;
  float *gp = grad_points + ((size_t)batch * c + l0) * m;

  for (int t = threadIdx.x; t < nl * m; t += TG_THREADS) s_acc[t] = 0.f;
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += TG_THREADS) {
    const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
    const int i1 = id[j * 3 + 0], i2 = id[j * 3 + 1], i3 = id[j * 3 + 2];
    for (int l = 0; l < nl; ++l) {
      const float go = g[(size_t)l * n + j];
      atomicAdd(&s_acc[l * m + i1], go * w1);
      atomicAdd(&s_acc[l * m + i2], go * w2);
      atomicAdd(&s_acc[l * m + i3], go * w3);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nl * m; t += TG_THREADS) gp[t] = s_acc[t];
}

__global__ __launch_bounds__(256) void three_interpolate_grad_atomic_kernel(
    int c, int n, int m, long long total, const float *__restrict__ grad_out,
    const int *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ grad_points) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total;
       t += (long long)gridDim.x * 256) {
    const int j = (int)(t % n);
    const long long bl = t / n;
    const int batch = (int)(bl / c);
    const int *id = idx + ((size_t)batch * n + j) * 3;
    const float *w = weight + ((size_t)batch * n + j) * 3;
    const float go = grad_out[t];
    float *gp = grad_points + bl * m;
    atomicAdd(gp + id[0], go * w[0]);
    atomicAdd(gp + id[1], go * w[1]);
    atomicAdd(gp + id[2], go * w[2]);
  }
}

}  // namespace

extern "C" int p2r_three_nn(int b, int n, int m, const float *unknown, const float *known,
                            float *dist2, int *idx, void *stream) {
  if (b < 0 || n < 0 || m < 0) return P2R_EINVAL;
  if (b == 0 || n == 0) return P2R_OK;
  const int n_tiles = p2r_cdiv(n, NN_THREADS);
  const long long blocks = (long long)b * n_tiles;
  if (blocks > 0x7fffffffLL) return P2R_EINVAL;
  hipLaunchKernelGGL(three_nn_kernel, dim3((unsigned)blocks), dim3(NN_THREADS), 0,
                     p2r_stream(stream), n, m, n_tiles, unknown, known, dist2, idx);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_three_interpolate(int b, int c, int m, int n, const float *points,
                                     const int *idx, const float *weight, float *out,
                                     void *stream) {
  if (b < 0 || c < 0 || n < 0 || m < 0) return P2R_EINVAL;
  if (b == 0 || c == 0 || n == 0) return P2R_OK;
  const int n_tiles = p2r_cdiv(n, TI_THREADS);
  const int c_tiles = p2r_cdiv(c, TI_CCHUNK);
  const long long blocks = (long long)b * c_tiles * n_tiles;
  if (blocks > 0x7fffffffLL) return P2R_EINVAL;
  hipLaunchKernelGGL(three_interpolate_kernel, dim3((unsigned)blocks), dim3(TI_THREADS), 0,
                     p2r_stream(stream), c, m, n, n_tiles, c_tiles, points, idx, weight, out);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                          const int *idx, const float *weight,
                                          float *grad_points, void *stream) {
  if (b < 0 || c < 0 || n < 0 || m < 0) return P2R_EINVAL;
  if (b == 0 || c == 0 || m == 0) return P2R_OK;
  hipStream_t st = p2r_stream(stream);
  const size_t row_bytes = (size_t)m * sizeof(float);
  int gc = 16;
  while (gc > 1 && gc * row_bytes > 64 * 1024) gc >>= 1;
  while (gc > 4 && (long long)b * p2r_cdiv(c, gc) < 512) gc >>= 1;
  if (gc * row_bytes <= 64 * 1024) {
    const int c_tiles = p2r_cdiv(c, gc);
    const long long blocks = (long long)b * c_tiles;
    if (blocks > 0x7fffffffLL) return P2R_EINVAL;
    hipLaunchKernelGGL(three_interpolate_grad_lds_kernel, dim3((unsigned)blocks), dim3(TG_THREADS),
                       gc * row_bytes, st, c, n, m, c_tiles, gc, grad_out, idx, weight, grad_points);
    P2R_LAUNCH_CHECK();
    return P2R_OK;
  }
  hipError_t e = hipMemsetAsync(grad_points, 0, (size_t)b * c * m * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  const long long total = (long long)b * c * n;
  if (total == 0) return P2R_OK;
  const int blocks = (int)std::min<long long>(p2r_cdiv(total, 256), 256 * 16);
  hipLaunchKernelGGL(three_interpolate_grad_atomic_kernel, dim3(blocks), dim3(256), 0, st, c, n, m,
                     total, grad_out, idx, weight, grad_points);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
