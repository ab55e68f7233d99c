// bn_act.hip -- BatchNorm (train / eval) fused with residual add and ReLU, gfx950.
//
// Replaces, inside st_gcn_block (reference models/p2rnet/modules/stgcn_layers.py:399-439),
// the chains  BatchNorm2d -> ReLU  (tcn.0, tcn.1)  and
// BatchNorm2d -> Dropout(0) -> (+ residual) -> ReLU  (tcn.3, tcn.4, :437-439), which run as
// 3-5 separate HBM passes each way in the reference (MIOpen BN + elementwise kernels).
//
// MI355X design: HBM bound, so the only lever is the number of passes over the
// (N, C, L = T*V) activation.  Forward = one reduction pass (per-row partial sums, 16-byte
// loads, one workgroup per (n, c) row so the grid fills the chip) + one apply pass that
// normalises, adds the residual and applies ReLU in registers.  Backward = one reduction
// pass (sum g and sum g*xhat with the ReLU mask recomputed from y) + one apply pass
// producing dx and the residual gradient together.  Partial sums are written per row and
// combined by the caller in fp64 (deterministic, no atomics).
#include "p2r_common.h"

namespace {

constexpr int BN_THREADS = 256;

// These passes stream 0.4 GB tensors once: nontemporal loads and stores (no allocation in the caches for data that
// is not reused before it is evicted anyway) measured 4.9 -> 5.6 TB/s on bn_apply (tools/dev_bn_exp.py).
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float *p) {
  const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st_stream(float *p, const float4 &v) {
  __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v *>(p));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ void block_sum2(float &a, float &b) {
  __shared__ float sa[BN_THREADS / 64], sb[BN_THREADS / 64];
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sa[w] = a; sb[w] = b; }
  __syncthreads();
  a = 0.f; b = 0.f;
#pragma unroll
  for (int i = 0; i < BN_THREADS / 64; ++i) { a += sa[i]; b += sb[i]; }
}

// partial[row] = (L, mean, M2 = sum (x - mean)^2) of the row of length L.  Sums are taken about a pivot (the row's
// first element): sum x^2 - (sum x)^2 / L in fp32 loses the variance once |mean| >> std (relative error of the
// variance ~ 1e-7 mean^2 / var), differences from a value of the row itself do not.
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(int L, const float *__restrict__ x,
                                                              float *__restrict__ partial) {
  const float *row = x + (size_t)blockIdx.x * L;
  const float c = row[0];
  float s = 0.f, q = 0.f;
  const bool vec = ((uintptr_t)row % 16 == 0);
  const int L4 = vec ? (L >> 2) : 0;
  for (int i = threadIdx.x; i < L4; i += BN_THREADS) {
    float4 v = ld_stream(row + 4 * i);
    v.x -= c; v.y -= c; v.z -= c; v.w -= c;
    s += (v.x + v.y) + (v.z + v.w);
    q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  for (int i = (L4 << 2) + threadIdx.x; i < L; i += BN_THREADS) {
    const float v = row[i] - c;
    s += v; q += v * v;
  }
  block_sum2(s, q);
  if (threadIdx.x == 0) {
    const float n = (float)L, d = s / n;
    partial[3 * (size_t)blockIdx.x] = n;
    partial[3 * (size_t)blockIdx.x + 1] = c + d;
    partial[3 * (size_t)blockIdx.x + 2] = fmaxf(q - s * d, 0.f);
  }
}

// y = relu?(x * scale[c] + shift[c] + res)
// AMAX (split16 mode): the float bits of max |y| over the tensor are merged into *amax (zeroed by the caller) -- the
// range word of the split16 graph conv that consumes y (split16.h)
__device__ __forceinline__ void bn_block_amax(unsigned m, unsigned *amax) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  if ((threadIdx.x & 63) == 0 && m != 0) atomicMax(amax, m);
}
__device__ __forceinline__ unsigned bn_abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

template <bool RELU, bool RES, bool AMAX = false>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(int C, int L, int chunks,
                                                              const float *__restrict__ x,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift,
                                                              const float *__restrict__ res,
                                                              float *__restrict__ y,
                                                              unsigned char *__restrict__ mask,
                                                              unsigned *__restrict__ amax = nullptr) {
  unsigned am = 0;
  // last rows first: the producer (temporal conv, ascending tiles) wrote them last and part of them is still on
  // chip (-3 % in the step); the consumer that follows starts with the rows this pass writes last
  const int bid = (int)(gridDim.x - 1 - blockIdx.x);
  const int rowi = bid / chunks;
  const int chunk = bid % chunks;
  const int c = rowi % C;
  const float sc = scale[c], sh = shift[c];
  const size_t base = (size_t)rowi * L;
  const int per = (((L + chunks - 1) / chunks) + 3) & ~3;   // multiple of 4: chunk starts stay 16-byte aligned
  const int lo = chunk * per, hi = min(L, lo + per);
  const bool vec = (((uintptr_t)(x + base) | (uintptr_t)(y + base) | (RES ? (uintptr_t)(res + base) : 0)) % 16 == 0) &&
                   (lo % 4 == 0);
  int i = lo + threadIdx.x * 4;
  if (vec) {
    for (; i + 3 < hi; i += BN_THREADS * 4) {
      float4 v = ld_stream(x + base + i);
      v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
      if (RES) {
        const float4 r = ld_stream(res + base + i);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      st_stream(y + base + i, v);
      if (AMAX) am = max(max(am, bn_abs_bits(v.x)), max(max(bn_abs_bits(v.y), bn_abs_bits(v.z)), bn_abs_bits(v.w)));
      if (RELU && mask)     // one byte per element (y > 0): the backward reads 1 B instead of the 4 B of y
        *reinterpret_cast<uchar4 *>(mask + base + i) = make_uchar4(v.x > 0.f, v.y > 0.f, v.z > 0.f, v.w > 0.f);
    }
    // ragged tail of this chunk: elements [hi - (hi-lo)%4, hi)
    const int tail = lo + ((hi - lo) & ~3);
    for (int j = tail + threadIdx.x; j < hi; j += BN_THREADS) {
      float v = fmaf(x[base + j], sc, sh);
      if (RES) v += res[base + j];
      if (RELU) v = fmaxf(v, 0.f);
      y[base + j] = v;
      if (AMAX) am = max(am, bn_abs_bits(v));
      if (RELU && mask) mask[base + j] = v > 0.f;
    }
  } else {
    for (int j = lo + threadIdx.x; j < hi; j += BN_THREADS) {
      float v = fmaf(x[base + j], sc, sh);
      if (RES) v += res[base + j];
      if (RELU) v = fmaxf(v, 0.f);
      y[base + j] = v;
      if (AMAX) am = max(am, bn_abs_bits(v));
      if (RELU && mask) mask[base + j] = v > 0.f;
    }
  }
  if (AMAX) bn_block_amax(am, amax);
}

// partial[row] = (sum g, sum g * xhat),  g = dy * (RELU ? y > 0 : 1),  xhat = (x - mean) * invstd
// MASK: 0 = no ReLU, 1 = ReLU mask from the saved output y (> 0), 2 = ReLU mask recomputed
// from the input (x * mscale[c] + mshift[c] > 0; the normalised activation was never stored),
// 3 = ReLU mask from the byte map written by bn_apply (passed through the `y` pointer).
template <int MASK>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(int C, int L, const float *__restrict__ dy,
                                                                   const float *__restrict__ y,
                                                                   const float *__restrict__ x,
                                                                   const float *__restrict__ mean,
                                                                   const float *__restrict__ invstd,
                                                                   const float *__restrict__ mscale,
                                                                   const float *__restrict__ mshift,
                                                                   float2 *__restrict__ partial) {
  constexpr bool RELU = MASK == 1;
  const int c = blockIdx.x % C;
  const float mu = mean[c], is = invstd[c];
  const float msc = MASK == 2 ? mscale[c] : 0.f, msh = MASK == 2 ? mshift[c] : 0.f;
  const size_t base = (size_t)blockIdx.x * L;
  float s = 0.f, q = 0.f;
  const unsigned char *m8 = reinterpret_cast<const unsigned char *>(y);
  const bool vec = (((uintptr_t)(dy + base) | (uintptr_t)(x + base) | (RELU ? (uintptr_t)(y + base) : 0)) % 16 == 0) &&
                   (MASK != 3 || (base % 4 == 0 && (uintptr_t)m8 % 4 == 0));
  const int L4 = vec ? (L >> 2) : 0;
  for (int i = threadIdx.x; i < L4; i += BN_THREADS) {
    float4 g = ld_stream(dy + base + 4 * i);
    const float4 xv = ld_stream(x + base + 4 * i);
    if (MASK == 3) {
      const uchar4 mk = reinterpret_cast<const uchar4 *>(m8 + base)[i];
      g.x = mk.x ? g.x : 0.f; g.y = mk.y ? g.y : 0.f; g.z = mk.z ? g.z : 0.f; g.w = mk.w ? g.w : 0.f;
    }
    if (RELU) {
      const float4 yv = reinterpret_cast<const float4 *>(y + base)[i];
      g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
      g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
    }
    if (MASK == 2) {
      g.x = fmaf(xv.x, msc, msh) > 0.f ? g.x : 0.f; g.y = fmaf(xv.y, msc, msh) > 0.f ? g.y : 0.f;
      g.z = fmaf(xv.z, msc, msh) > 0.f ? g.z : 0.f; g.w = fmaf(xv.w, msc, msh) > 0.f ? g.w : 0.f;
    }
    s += (g.x + g.y) + (g.z + g.w);
    q += (g.x * ((xv.x - mu) * is) + g.y * ((xv.y - mu) * is)) + (g.z * ((xv.z - mu) * is) + g.w * ((xv.w - mu) * is));
  }
  for (int i = (L4 << 2) + threadIdx.x; i < L; i += BN_THREADS) {
    float g = dy[base + i];
    if (RELU) g = y[base + i] > 0.f ? g : 0.f;
    if (MASK == 2) g = fmaf(x[base + i], msc, msh) > 0.f ? g : 0.f;
    if (MASK == 3) g = m8[base + i] ? g : 0.f;
    s += g; q += g * ((x[base + i] - mu) * is);
  }
  block_sum2(s, q);
  if (threadIdx.x == 0) partial[blockIdx.x] = make_float2(s, q);
}

// dx = k[c] * (g - m1[c] - xhat * m2[c]);  dres = g
template <int MASK, bool RES, bool AMAX = false>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(int C, int L, int chunks,
                                                                  const float *__restrict__ dy,
                                                                  const float *__restrict__ y,
                                                                  const float *__restrict__ x,
                                                                  const float *__restrict__ mean,
                                                                  const float *__restrict__ invstd,
                                                                  const float *__restrict__ kscale,
                                                                  const float *__restrict__ m1,
                                                                  const float *__restrict__ m2,
                                                                  const float *__restrict__ mscale,
                                                                  const float *__restrict__ mshift,
                                                                  float *__restrict__ dx,
                                                                  float *__restrict__ dres,
                                                                  unsigned *__restrict__ amax = nullptr) {
  constexpr bool RELU = MASK == 1;
  unsigned am = 0;                      // AMAX: max |dx| (split16.h), merged into *amax (zeroed by the caller)
  const int rowi = blockIdx.x / chunks;
  const int chunk = blockIdx.x % chunks;
  const int c = rowi % C;
  const float mu = mean[c], is = invstd[c], kk = kscale[c], a1 = m1[c], a2 = m2[c];
  const float msc = MASK == 2 ? mscale[c] : 0.f, msh = MASK == 2 ? mshift[c] : 0.f;
  const size_t base = (size_t)rowi * L;
  const int per = (((L + chunks - 1) / chunks) + 3) & ~3;   // multiple of 4: chunk starts stay 16-byte aligned
  const int lo = chunk * per, hi = min(L, lo + per);
  const unsigned char *m8 = reinterpret_cast<const unsigned char *>(y);
  auto one = [&](float g, float yv, float xv, float &dxo, float &dro) {
    if (RELU || MASK == 3) g = yv > 0.f ? g : 0.f;       // MASK 3: yv carries the mask byte
    if (MASK == 2) g = fmaf(xv, msc, msh) > 0.f ? g : 0.f;
    const float xh = (xv - mu) * is;
    dxo = kk * (g - a1 - xh * a2);
    dro = g;
  };
  const bool vec = (((uintptr_t)(dy + base) | (uintptr_t)(x + base) | (uintptr_t)(dx + base) |
                     (RELU ? (uintptr_t)(y + base) : 0) | (RES ? (uintptr_t)(dres + base) : 0)) % 16 == 0) &&
                   (MASK != 3 || (base % 4 == 0 && (uintptr_t)m8 % 4 == 0));
  int tail = lo;
  if (vec) {
    tail = lo + ((hi - lo) & ~3);
    for (int j = lo + threadIdx.x * 4; j + 3 < hi; j += BN_THREADS * 4) {
      const float4 g4 = ld_stream(dy + base + j);
      const float4 x4 = ld_stream(x + base + j);
      float4 y4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (RELU) y4 = *reinterpret_cast<const float4 *>(y + base + j);
      if (MASK == 3) {
        const uchar4 mk = *reinterpret_cast<const uchar4 *>(m8 + base + j);
        y4 = make_float4((float)mk.x, (float)mk.y, (float)mk.z, (float)mk.w);
      }
      float4 o, rr;
      one(g4.x, y4.x, x4.x, o.x, rr.x); one(g4.y, y4.y, x4.y, o.y, rr.y);
      one(g4.z, y4.z, x4.z, o.z, rr.z); one(g4.w, y4.w, x4.w, o.w, rr.w);
      st_stream(dx + base + j, o);
      if (AMAX) am = max(max(am, bn_abs_bits(o.x)), max(max(bn_abs_bits(o.y), bn_abs_bits(o.z)), bn_abs_bits(o.w)));
      if (RES) st_stream(dres + base + j, rr);
    }
  }
  for (int j = tail + threadIdx.x; j < hi; j += BN_THREADS) {
    float o, rr;
    one(dy[base + j], RELU ? y[base + j] : (MASK == 3 ? (float)m8[base + j] : 0.f), x[base + j], o, rr);
    dx[base + j] = o;
    if (AMAX) am = max(am, bn_abs_bits(o));
    if (RES) dres[base + j] = rr;
  }
  if (AMAX) bn_block_amax(am, amax);
}

}  // namespace

extern "C" int p2r_bn_stats(int rows, int L, const float *x, float *partial /* [rows][3] */, void *stream) {
  if (rows < 0 || L <= 0) return P2R_EINVAL;
  if (rows == 0) return P2R_OK;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(rows), dim3(BN_THREADS), 0, p2r_stream(stream), L, x, partial);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

static int bn_chunks(int rows, int L) {
  int chunks = 1;
  while ((long long)rows * chunks < 2048 && (L / (chunks * 2)) >= 4096) chunks *= 2;
  return chunks;
}

extern "C" int p2r_bn_apply(int N, int C, int L, const float *x, const float *scale, const float *shift,
                            const float *res, int relu, float *y, unsigned char *relu_mask, void *stream) {
  if (N < 0 || C <= 0 || L <= 0) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  const int rows = N * C, chunks = bn_chunks(rows, L);
  dim3 grid(rows * chunks), blk(BN_THREADS);
  hipStream_t st = p2r_stream(stream);
  if (relu && res) hipLaunchKernelGGL((bn_apply_kernel<true, true>), grid, blk, 0, st, C, L, chunks, x, scale, shift, res, y, relu_mask);
  else if (relu) hipLaunchKernelGGL((bn_apply_kernel<true, false>), grid, blk, 0, st, C, L, chunks, x, scale, shift, res, y, relu_mask);
  else if (res) hipLaunchKernelGGL((bn_apply_kernel<false, true>), grid, blk, 0, st, C, L, chunks, x, scale, shift, res, y, relu_mask);
  else hipLaunchKernelGGL((bn_apply_kernel<false, false>), grid, blk, 0, st, C, L, chunks, x, scale, shift, res, y, relu_mask);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_bn_bwd_reduce(int N, int C, int L, const float *dy, const float *y, const float *x,
                                 const float *mean, const float *invstd, int relu, const float *mscale,
                                 const float *mshift, float *partial, void *stream) {
  if (N < 0 || C <= 0 || L <= 0 || relu < 0 || relu > 3) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  hipStream_t st = p2r_stream(stream);
  float2 *pp = reinterpret_cast<float2 *>(partial);
  if (relu == 3) hipLaunchKernelGGL(bn_bwd_reduce_kernel<3>, dim3(N * C), dim3(BN_THREADS), 0, st, C, L, dy, y, x, mean, invstd, mscale, mshift, pp);
  else if (relu == 1) hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(N * C), dim3(BN_THREADS), 0, st, C, L, dy, y, x, mean, invstd, mscale, mshift, pp);
  else if (relu == 2) hipLaunchKernelGGL(bn_bwd_reduce_kernel<2>, dim3(N * C), dim3(BN_THREADS), 0, st, C, L, dy, y, x, mean, invstd, mscale, mshift, pp);
  else hipLaunchKernelGGL(bn_bwd_reduce_kernel<0>, dim3(N * C), dim3(BN_THREADS), 0, st, C, L, dy, y, x, mean, invstd, mscale, mshift, pp);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_bn_bwd_apply(int N, int C, int L, const float *dy, const float *y, const float *x,
                                const float *mean, const float *invstd, const float *kscale,
                                const float *m1, const float *m2, int relu, const float *mscale,
                                const float *mshift, float *dx, float *dres, void *stream) {
  if (N < 0 || C <= 0 || L <= 0 || relu < 0 || relu > 3) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  const int rows = N * C, chunks = bn_chunks(rows, L);
  dim3 grid(rows * chunks), blk(BN_THREADS);
  hipStream_t st = p2r_stream(stream);
#define P2R_BWA(M, R) hipLaunchKernelGGL((bn_bwd_apply_kernel<M, R>), grid, blk, 0, st, C, L, chunks, dy, y, x, mean, invstd, kscale, m1, m2, mscale, mshift, dx, dres)
  if (relu == 3 && dres) P2R_BWA(3, true);
  else if (relu == 3) P2R_BWA(3, false);
  else if (relu == 1 && dres) P2R_BWA(1, true);
  else if (relu == 1) P2R_BWA(1, false);
  else if (relu == 2 && dres) P2R_BWA(2, true);
  else if (relu == 2) P2R_BWA(2, false);
  else if (dres) P2R_BWA(0, true);
  else P2R_BWA(0, false);
#undef P2R_BWA
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// split16 mode: p2r_bn_bwd_apply for the chain's form (mask bytes, relu = 3) that also leaves the float bits of max |dx|
// in *amax_bits (split16.h: the range word of the split16 temporal conv's data gradient, which consumes dx).
extern "C" int p2r_bn_bwd_apply_amax(int N, int C, int L, const float *dy, const unsigned char *mask, const float *x,
                                     const float *mean, const float *invstd, const float *kscale, const float *m1,
                                     const float *m2, float *dx, float *dres, unsigned *amax_bits, void *stream) {
  if (N < 0 || C <= 0 || L <= 0 || !amax_bits) return P2R_EINVAL;
  hipStream_t st = p2r_stream(stream);
  hipError_t e = hipMemsetAsync(amax_bits, 0, sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  if (N == 0) return P2R_OK;
  const int rows = N * C, chunks = bn_chunks(rows, L);
  dim3 grid(rows * chunks), blk(BN_THREADS);
  const float *y = reinterpret_cast<const float *>(mask), *none = nullptr;
  if (dres) hipLaunchKernelGGL((bn_bwd_apply_kernel<3, true, true>), grid, blk, 0, st, C, L, chunks, dy, y, x, mean, invstd,
                               kscale, m1, m2, none, none, dx, dres, amax_bits);
  else hipLaunchKernelGGL((bn_bwd_apply_kernel<3, false, true>), grid, blk, 0, st, C, L, chunks, dy, y, x, mean, invstd,
                          kscale, m1, m2, none, none, dx, dres, amax_bits);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// split16 mode: p2r_bn_apply with ReLU (+ residual, + mask bytes) that also leaves the float bits of max |y| in *amax_bits
// (the range word of the split16 graph conv that consumes y).
extern "C" int p2r_bn_apply_amax(int N, int C, int L, const float *x, const float *scale, const float *shift,
                                 const float *res, float *y, unsigned char *mask, unsigned *amax_bits, void *stream) {
  if (N < 0 || C <= 0 || L <= 0 || !amax_bits) return P2R_EINVAL;
  hipStream_t st = p2r_stream(stream);
  hipError_t e = hipMemsetAsync(amax_bits, 0, sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  if (N == 0) return P2R_OK;
  const int rows = N * C, chunks = bn_chunks(rows, L);
  dim3 grid(rows * chunks), blk(BN_THREADS);
  if (res) hipLaunchKernelGGL((bn_apply_kernel<true, true, true>), grid, blk, 0, st, C, L, chunks, x, scale, shift, res, y, mask, amax_bits);
  else hipLaunchKernelGGL((bn_apply_kernel<true, false, true>), grid, blk, 0, st, C, L, chunks, x, scale, shift, res, y, mask, amax_bits);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// ---- statistics finalisation: kernel partials -> per-channel constants, one tiny launch -------------
// partial [P][C][width]; one workgroup per channel combines its P entries in fp64.
//   width 3: (count, mean, M2) per entry -- p2r_bn_stats rows (n, c) and the third-generation conv epilogues; merged
//            as mean = sum n_p mean_p / n, M2 = sum [M2_p + n_p (mean_p - mean)^2] (every term non-negative: no
//            cancellation whatever the mean is);
//   width 2: (sum, sum of squares) per entry -- the first- and second-generation conv epilogues (fallback shapes).
namespace {

constexpr int FIN_THREADS = 256;

__device__ __forceinline__ void fin_block_sum(double &a, double &b) {
  __shared__ double sa[FIN_THREADS / 64], sb[FIN_THREADS / 64];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sa[w] = a; sb[w] = b; }
  __syncthreads();
  a = 0.0; b = 0.0;
#pragma unroll
  for (int i = 0; i < FIN_THREADS / 64; ++i) { a += sa[i]; b += sb[i]; }
}

// out [4][C] = mean, invstd, scale = gamma*invstd, shift = beta - mean*scale; running statistics updated in
// place with `momentum` (unbiased variance, like nn.BatchNorm) unless momentum < 0.
__global__ __launch_bounds__(FIN_THREADS) void bn_finalize_kernel(int P, int C, int width, const float *__restrict__ partial,
                                                                  double M, const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, double eps,
                                                                  double momentum, float *__restrict__ running_mean,
                                                                  float *__restrict__ running_var,
                                                                  long long *__restrict__ num_batches_tracked,
                                                                  float *__restrict__ out) {
  const int c = blockIdx.x;
  if (c == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
  double s = 0.0, q = 0.0;                 // sum and M2 of the channel
  if (width == 3) {
    double n = 0.0;
    for (int p = threadIdx.x; p < P; p += FIN_THREADS) {
      const float *e = partial + ((size_t)p * C + c) * 3;
      n += (double)e[0]; s += (double)e[0] * (double)e[1];
    }
    fin_block_sum(n, s);
    const double mean = s / n;
    __syncthreads();                       // fin_block_sum's staging is reused below
    double m2 = 0.0, unused = 0.0;
    for (int p = threadIdx.x; p < P; p += FIN_THREADS) {
      const float *e = partial + ((size_t)p * C + c) * 3;
      const double d = (double)e[1] - mean;
      m2 += (double)e[2] + (double)e[0] * d * d;
    }
    fin_block_sum(m2, unused);
    M = n;                                 // the entries carry their own counts
    s = mean * n;
    q = m2;
  } else {
    for (int p = threadIdx.x; p < P; p += FIN_THREADS) {
      const float *e = partial + ((size_t)p * C + c) * 2;
      s += (double)e[0]; q += (double)e[1];
    }
    fin_block_sum(s, q);
    const double mean = s / M;
    q = q - mean * mean * M;               // M2
  }
  if (threadIdx.x == 0) {
    const double mean = s / M;
    double var = q / M;
    if (var < 0.0) var = 0.0;
    const float mean_f = (float)mean, invstd_f = (float)(1.0 / sqrt(var + eps));
    const float scale = gamma[c] * invstd_f;
    out[c] = mean_f;
    out[C + c] = invstd_f;
    out[2 * C + c] = scale;
    out[3 * C + c] = beta[c] - mean_f * scale;
    if (momentum >= 0.0) {
      const float mom = (float)momentum;
      const double unbiased = var * (M / (M - 1.0 > 1.0 ? M - 1.0 : 1.0));
      running_mean[c] = running_mean[c] * (1.f - mom) + mom * mean_f;
      running_var[c] = running_var[c] * (1.f - mom) + mom * (float)unbiased;
    }
  }
}

// out [4][C] = sum g, sum g*xhat, (sum g)/M, (sum g*xhat)/M
__global__ __launch_bounds__(FIN_THREADS) void bn_bwd_finalize_kernel(int P, int C, const float2 *__restrict__ partial,
                                                                      double M, float *__restrict__ out) {
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int p = threadIdx.x; p < P; p += FIN_THREADS) {
    const float2 v = partial[(size_t)p * C + c];
    s += (double)v.x; q += (double)v.y;
  }
  fin_block_sum(s, q);
  if (threadIdx.x == 0) {
    out[c] = (float)s;
    out[C + c] = (float)q;
    out[2 * C + c] = (float)(s / M);
    out[3 * C + c] = (float)(q / M);
  }
}

}  // namespace

extern "C" int p2r_bn_finalize(int P, int C, int width, const float *partial, double M, const float *gamma,
                               const float *beta, double eps, double momentum, float *running_mean, float *running_var,
                               long long *num_batches_tracked, float *out, void *stream) {
  if (P <= 0 || C <= 0 || M <= 0.0 || (width != 2 && width != 3)) return P2R_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(FIN_THREADS), 0, p2r_stream(stream), P, C, width, partial, M, gamma,
                     beta, eps, momentum, running_mean, running_var, num_batches_tracked, out);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_bn_bwd_finalize(int P, int C, const float *partial, double M, float *out, void *stream) {
  if (P <= 0 || C <= 0 || M <= 0.0) return P2R_EINVAL;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(FIN_THREADS), 0, p2r_stream(stream), P, C,
                     reinterpret_cast<const float2 *>(partial), M, out);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// ---- per-(channel, joint) sums over samples and frames -------------------------------
// out_partial[row][w] = sum_t x[row][t*V + w]   (row = (n, c)); the caller sums over n.
// Used for the gradient of the graph-conv bias term (a (C, V) table).
namespace {
__global__ __launch_bounds__(256) void colsum_kernel(int T, int V, const float *__restrict__ x,
                                                     float *__restrict__ out_partial) {
  __shared__ float s_acc[256];
  const float *row = x + (size_t)blockIdx.x * T * V;
  const int lanes_used = (256 / V) * V;          // whole frames per sweep
  const int tid = threadIdx.x;
  float acc = 0.f;
  if (tid < lanes_used) {
    const int stride = lanes_used;                // multiple of V: every thread keeps its joint
    for (int i = tid; i < T * V; i += stride) acc += row[i];
  }
  s_acc[tid] = acc;
  __syncthreads();
  if (tid < V) {
    float s = 0.f;
    for (int j = tid; j < lanes_used; j += V) s += s_acc[j];
    out_partial[(size_t)blockIdx.x * V + tid] = s;
  }
}
}  // namespace

extern "C" int p2r_colsum(int rows, int T, int V, const float *x, float *out_partial, void *stream) {
  if (rows < 0 || T <= 0 || V <= 0 || V > 256) return P2R_EINVAL;
  if (rows == 0) return P2R_OK;
  hipLaunchKernelGGL(colsum_kernel, dim3(rows), dim3(256), 0, p2r_stream(stream), T, V, x, out_partial);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// ---- sum of per-workgroup partials over the leading axis ---------------------------------------------------------
// in [P][M] -> out [M] (or, tr64: M = G * 64 * 64 and every 64 x 64 block comes out transposed).  The weight- and
// adjacency-gradient kernels leave one partial per persistent workgroup (P = 256; the graph-conv weight gradient is
// 46 MB of them); a generic strided reduction reads that at a fraction of the HBM rate, this one streams it: a thread
// owns four consecutive outputs (16-byte loads, eight rows in flight), rows are added in order (deterministic).
namespace {
__global__ __launch_bounds__(256) void sum_leading_kernel(int P, long long M4, const float4 *__restrict__ in,
                                                          float *__restrict__ out, int tr64) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M4) return;
  // eight rows in flight, two interleaved accumulators per component (rows p, p+2, .. and p+1, p+3, ..) combined at the
  // end: the summation order is fixed (deterministic) and the rounding error grows with P / 2 instead of P
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = make_float4(0.f, 0.f, 0.f, 0.f);
  int p = 0;
  for (; p + 8 <= P; p += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld_stream(reinterpret_cast<const float *>(in + (size_t)(p + u) * M4 + i));
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
      acc2.x += v[u + 1].x; acc2.y += v[u + 1].y; acc2.z += v[u + 1].z; acc2.w += v[u + 1].w;
    }
  }
  for (; p < P; ++p) {
    const float4 v = in[(size_t)p * M4 + i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  acc.x += acc2.x; acc.y += acc2.y; acc.z += acc2.z; acc.w += acc2.w;
  if (!tr64) {
    reinterpret_cast<float4 *>(out)[i] = acc;
  } else {
    const long long e = 4 * i;                       // (block, a, b..b+3) -> (block, b.., a)
    const long long blk = e >> 12;
    const int a = (int)((e >> 6) & 63), b = (int)(e & 63);
    float *o = out + (blk << 12) + a;
    o[(b + 0) * 64] = acc.x; o[(b + 1) * 64] = acc.y; o[(b + 2) * 64] = acc.z; o[(b + 3) * 64] = acc.w;
  }
}

// The same sum for many rows (P >= 32: the 256 per-workgroup partials of a step's weight gradients).  With one thread per
// four outputs a thread walks all P rows on its own -- 32 dependent rounds of eight loads, and 12-44 workgroups for the
// whole device at the shapes of a train step (20 us for 12-46 MB).  Here a workgroup owns 16 float4 columns and its 16
// thread rows each take the rows p = ty, ty + 16, ...: two rounds of eight loads per thread, 16x the workgroups; the 16
// partial sums meet in LDS and are added in the order ty = 0..15 (fixed: deterministic).
__global__ __launch_bounds__(256) void sum_leading_rows_kernel(int P, long long M4, const float4 *__restrict__ in,
                                                               float *__restrict__ out, int tr64) {
  __shared__ float4 red[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const long long i = (long long)blockIdx.x * 16 + tx;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < M4) {
    int p = ty;
    for (; p + 7 * 16 < P; p += 8 * 16) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld_stream(reinterpret_cast<const float *>(in + (size_t)(p + 16 * u) * M4 + i));
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; p < P; p += 16) {
      const float4 v = in[(size_t)p * M4 + i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty != 0 || i >= M4) return;
#pragma unroll
  for (int r = 1; r < 16; ++r) {
    const float4 v = red[r][tx];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (!tr64) {
    reinterpret_cast<float4 *>(out)[i] = acc;
  } else {
    const long long e = 4 * i;                       // (block, a, b..b+3) -> (block, b.., a)
    const long long blk = e >> 12;
    const int a = (int)((e >> 6) & 63), b = (int)(e & 63);
    float *o = out + (blk << 12) + a;
    o[(b + 0) * 64] = acc.x; o[(b + 1) * 64] = acc.y; o[(b + 2) * 64] = acc.z; o[(b + 3) * 64] = acc.w;
  }
}
}  // namespace

extern "C" int p2r_sum_leading(int P, long long M, const float *in, float *out, int tr64, void *stream) {
  if (P <= 0 || M <= 0 || (M % 4) != 0 || ((uintptr_t)in % 16) != 0 || ((uintptr_t)out % 16) != 0) return P2R_EINVAL;
  if (tr64 && (M % 4096) != 0) return P2R_EINVAL;
  const long long M4 = M / 4;
  if (P >= 32) {
    const long long rblocks = (M4 + 15) / 16;
    if (rblocks > 0x7fffffffLL) return P2R_EINVAL;
    hipLaunchKernelGGL(sum_leading_rows_kernel, dim3((unsigned)rblocks), dim3(256), 0, p2r_stream(stream), P, M4,
                       reinterpret_cast<const float4 *>(in), out, tr64);
    P2R_LAUNCH_CHECK();
    return P2R_OK;
  }
  const long long blocks = (M4 + 255) / 256;
  if (blocks > 0x7fffffffLL) return P2R_EINVAL;
  hipLaunchKernelGGL(sum_leading_kernel, dim3((unsigned)blocks), dim3(256), 0, p2r_stream(stream), P, M4,
                     reinterpret_cast<const float4 *>(in), out, tr64);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
