// group_gather.hip -- index gather / scatter-add kernels for gfx950.
//
// Replaces group_points_kernel / group_points_grad_kernel (reference
// _ext-src/src/group_points_gpu.cu:8-75) and gather_points_kernel /
// gather_points_grad_kernel (src/sampling_gpu.cu:8-57).  gather_points is
// group_points with nsample = 1, so both pairs share one implementation.
//
// Reference launch shape: one block per cloud, each thread walking nsample
// indices serially and writing with stride nsample (uncoalesced), and the
// backward does one global float atomicAdd per element.
//
// MI355X design
//  forward : HBM-write bound.  A thread owns 4 consecutive (j,k) slots: one
//            16-byte idx load, four L1/L2-resident source reads, one 16-byte
//            coalesced store; it keeps the four indices in registers and walks
//            a chunk of channels, so the index tensor is read once per chunk,
//            not once per channel.
//  backward: gather form, deterministic (round 5).  A workgroup stages the index
//            list of its cloud and a chunk of GC gradient rows ([GC][S]) in LDS;
//            every thread owns destination points and walks the index list IN
//            ORDER (16-byte broadcast reads), adding the staged gradient of each
//            hit to GC register accumulators: no atomics at all, the sum of a
//            destination is taken in ascending slot order -- bit for bit the
//            sequential loop of the CPU restatement -- and the rows leave with
//            coalesced stores, overwriting grad_points (no zero-fill pass).
//            The first form of this kernel scatter-added with ds_add_f32 from
//            eight waves: order of arrival, i.e. run-to-run differences in the
//            last bit of d features, which train-mode BatchNorm amplified to
//            1e-3 of the backbone's gradients.  Index lists too long for LDS
//            fall back to that form, rows too long for it to global atomics on
//            a zeroed buffer (both order-free like the reference's atomics).
#include "p2r_common.h"

#include <algorithm>

namespace {

constexpr int GP_THREADS = 256;
constexpr int GP_CCHUNK = 16;  // channels walked per block in the forward

// out[b,l,s] = points[b,l,idx[b,s]],  s in [0, S = npoints*nsample)
__global__ __launch_bounds__(GP_THREADS) void group_fwd_kernel(
    int c, int n, int S, int s_tiles, int c_tiles, int vec_ok, const float *__restrict__ points,
    const int *__restrict__ idx, float *__restrict__ out) {
  int bid = blockIdx.x;
  const int st = bid % s_tiles; bid /= s_tiles;
  const int ct = bid % c_tiles;
  const int batch = bid / c_tiles;

  const float *p = points + (size_t)batch * c * n;
  const int *id = idx + (size_t)batch * S;
  float *o = out + (size_t)batch * c * S;

  const int s0 = (st * GP_THREADS + threadIdx.x) * 4;
  if (s0 >= S) return;
  const int l0 = ct * GP_CCHUNK;
  const int l1 = min(l0 + GP_CCHUNK, c);

  if (vec_ok) {  // S % 4 == 0 and 16-byte aligned bases: whole quad in range
    const int4 ii = *reinterpret_cast<const int4 *>(id + s0);
#pragma unroll 4
    for (int l = l0; l < l1; ++l) {
      const float *row = p + (size_t)l * n;
      float4 v;
      v.x = row[ii.x]; v.y = row[ii.y]; v.z = row[ii.z]; v.w = row[ii.w];
      *reinterpret_cast<float4 *>(o + (size_t)l * S + s0) = v;
    }
  } else {
    for (int s = s0; s < min(s0 + 4, S); ++s) {
      const int ii = id[s];
      for (int l = l0; l < l1; ++l) o[(size_t)l * S + s] = p[(size_t)l * n + ii];
    }
  }
}

constexpr int GG_THREADS = 512;

// grad_points[b,l,:] = sum over s of grad_out[b,l,s] scattered at idx[b,s];
// rows of GC channels accumulated in LDS.  dynamic LDS = GC * n floats.
__global__ __launch_bounds__(GG_THREADS) void group_grad_lds_kernel(
    int c, int n, int S, int c_tiles, int GC, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points) {
  extern __shared__ float s_acc[];  // [GC][n]
  const int ct = blockIdx.x % c_tiles;
  const int batch = blockIdx.x / c_tiles;
  const int l0 = ct * GC;
  const int nl = min(GC, c - l0);

  const float *g = grad_out + ((size_t)batch * c + l0) * S;
  const int *id = idx + (size_t)batch * S;
  float *gp = grad_points + ((size_t)batch * c + l0) * n;

  for (int t = threadIdx.x; t < nl * n; t += GG_THREADS) s_acc[t] = 0.f;
  __syncthreads();
  for (int s = threadIdx.x; s < S; s += GG_THREADS) {
    const int ii = id[s];
    for (int l = 0; l < nl; ++l) atomicAdd(&s_acc[l * n + ii], g[(size_t)l * S + s]);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nl * n; t += GG_THREADS) gp[t] = s_acc[t];
}

// Deterministic gather form: LDS = idx [SP] ints + gradient rows [GC][SP] floats, SP = S rounded up to 4.
// A thread owns the destinations tid, tid + GG_THREADS, ..; a slot whose index equals the destination adds its GC
// staged values to the thread's accumulators.  Hits are rare (S / n per destination), the walk itself is one
// broadcast 16-byte LDS read and four compares per four slots.
template <int GC>
__global__ __launch_bounds__(GG_THREADS) void group_grad_scan_kernel(
    int c, int n, int S, int SP, int c_tiles, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points) {
  extern __shared__ float s_raw[];
  int *ids = reinterpret_cast<int *>(s_raw);          // [SP]
  float *gl = s_raw + SP;                             // [GC][SP]
  const int ct = blockIdx.x % c_tiles;
  const int batch = blockIdx.x / c_tiles;
  const int l0 = ct * GC;
  const int nl = min(GC, c - l0);
  const float *g = grad_out + ((size_t)batch * c + l0) * S;
  const int *id = idx + (size_t)batch * S;
  float *gp = grad_points + ((size_t)batch * c + l0) * n;

  for (int t = threadIdx.x; t < SP; t += GG_THREADS) ids[t] = t < S ? id[t] : -1;
  for (int l = 0; l < GC; ++l)
    for (int t = threadIdx.x; t < SP; t += GG_THREADS) gl[l * SP + t] = (l < nl && t < S) ? g[(size_t)l * S + t] : 0.f;
  __syncthreads();
  for (int ii = threadIdx.x; ii < n; ii += GG_THREADS) {
    float acc[GC];
#pragma unroll
    for (int l = 0; l < GC; ++l) acc[l] = 0.f;
    for (int s4 = 0; s4 < SP; s4 += 4) {
      const int4 q = *reinterpret_cast<const int4 *>(ids + s4);
      const int qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (qq[e] == ii) {
#pragma unroll
          for (int l = 0; l < GC; ++l) acc[l] += gl[l * SP + s4 + e];
        }
    }
#pragma unroll
    for (int l = 0; l < GC; ++l)
      if (l < nl) gp[(size_t)l * n + ii] = acc[l];
  }
}

// The same sums with the index list SORTED first (P2RNet's shape: S = 2048 slots, n = 512 points): the keys
// (destination << SB | slot) go through a bitonic network in LDS (66 barrier stages for 2048 keys, ~5 us), after which a
// destination's slots are a contiguous, ascending run found by two binary searches -- 64 LDS reads per thread instead of
// the 2048-slot walk (measured in the train step: 0.284 ms as a walk, 0.11 ms as ds_add_f32 scatter before that).
// NP = keys padded to a power of two (<= 4096), SB = log2(NP).
template <int GC>
__global__ __launch_bounds__(GG_THREADS) void group_grad_sort_kernel(
    int c, int n, int S, int SP, int NP, int SB, int c_tiles, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points) {
  extern __shared__ float s_raw[];
  unsigned *keys = reinterpret_cast<unsigned *>(s_raw);   // [NP]
  float *gl = s_raw + NP;                                 // [GC][SP]
  const int ct = blockIdx.x % c_tiles;
  const int batch = blockIdx.x / c_tiles;
  const int l0 = ct * GC;
  const int nl = min(GC, c - l0);
  const float *g = grad_out + ((size_t)batch * c + l0) * S;
  const int *id = idx + (size_t)batch * S;
  float *gp = grad_points + ((size_t)batch * c + l0) * n;
  const int tid = threadIdx.x;

  // an index outside [0, n) takes the padding key: it sorts behind every destination and is never summed (the scan form
  // ignores such entries too; shifting it would drop its high bits and could alias a valid destination)
  for (int t = tid; t < NP; t += GG_THREADS)
    keys[t] = (t < S && (unsigned)id[t] < (unsigned)n) ? ((unsigned)id[t] << SB) | (unsigned)t : 0xffffffffu;
  // the gradient rows are fetched while the network runs (plain loads into registers would not survive the barriers'
  // register pressure for GC = 16: staged first, the loads of the last rows overlap the first sort stages)
  for (int l = 0; l < GC; ++l)
    for (int t = tid; t < SP; t += GG_THREADS) gl[l * SP + t] = (l < nl && t < S) ? g[(size_t)l * S + t] : 0.f;
  __syncthreads();
  for (int k = 2; k <= NP; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < NP / 2; t += GG_THREADS) {
        const int i = ((t / j) * 2 * j) + (t % j), p = i + j;
        const unsigned a = keys[i], b = keys[p];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { keys[i] = b; keys[p] = a; }
      }
      __syncthreads();
    }
  const unsigned smask = (1u << SB) - 1u;
  for (int ii = tid; ii < n; ii += GG_THREADS) {
    // [lo, hi): the run of destination ii = keys in [ii << SB, (ii + 1) << SB)
    int lo = 0, hi = NP;
    const unsigned klo = (unsigned)ii << SB, khi = (unsigned)(ii + 1) << SB;
    {
      int a = 0, b = NP;
      while (a < b) { const int m = (a + b) >> 1; if (keys[m] < klo) a = m + 1; else b = m; }
      lo = a; b = NP;
      while (a < b) { const int m = (a + b) >> 1; if (keys[m] < khi) a = m + 1; else b = m; }
      hi = a;
    }
    float acc[GC];
#pragma unroll
    for (int l = 0; l < GC; ++l) acc[l] = 0.f;
    for (int p = lo; p < hi; ++p) {
      const int sl = (int)(keys[p] & smask);
#pragma unroll
      for (int l = 0; l < GC; ++l) acc[l] += gl[l * SP + sl];
    }
#pragma unroll
    for (int l = 0; l < GC; ++l)
      if (l < nl) gp[(size_t)l * n + ii] = acc[l];
  }
}

template <int GC>
int group_grad_sort_launch(int b, int c, int n, int S, int SP, int NP, int SB, const float *grad_out, const int *idx,
                           float *grad_points, hipStream_t st) {
  const int c_tiles = p2r_cdiv(c, GC);
  const long long blocks = (long long)b * c_tiles;
  if (blocks > 0x7fffffffLL) return P2R_EINVAL;
  const size_t lds = ((size_t)NP + (size_t)GC * SP) * sizeof(float);
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(group_grad_sort_kernel<GC>, lds_ok);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(group_grad_sort_kernel<GC>, dim3((unsigned)blocks), dim3(GG_THREADS), lds, st, c, n, S, SP, NP, SB,
                     c_tiles, grad_out, idx, grad_points);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

template <int GC>
int group_grad_scan_launch(int b, int c, int n, int S, int SP, const float *grad_out, const int *idx, float *grad_points,
                           hipStream_t st) {
  const int c_tiles = p2r_cdiv(c, GC);
  const long long blocks = (long long)b * c_tiles;
  if (blocks > 0x7fffffffLL) return P2R_EINVAL;
  const size_t lds = (size_t)(GC + 1) * SP * sizeof(float);
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(group_grad_scan_kernel<GC>, lds_ok);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(group_grad_scan_kernel<GC>, dim3((unsigned)blocks), dim3(GG_THREADS), lds, st, c, n, S, SP, c_tiles,
                     grad_out, idx, grad_points);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// Fallback for rows longer than LDS: global atomics onto a zeroed buffer.
__global__ __launch_bounds__(GP_THREADS) void group_grad_atomic_kernel(
    int c, int n, int S, long long total, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points) {
  for (long long t = (long long)blockIdx.x * GP_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * GP_THREADS) {
    const int s = (int)(t % S);
    const long long bl = t / S;  // batch * c + l
    const int batch = (int)(bl / c);
    const int ii = idx[(size_t)batch * S + s];
    atomicAdd(grad_points + bl * n + ii, grad_out[t]);
  }
}

int group_forward(int b, int c, int n, int S, const float *points, const int *idx, float *out,
                  hipStream_t st) {
  if (b == 0 || c == 0 || S == 0) return P2R_OK;
  const int s_tiles = p2r_cdiv(S, GP_THREADS * 4);
  const int c_tiles = p2r_cdiv(c, GP_CCHUNK);
  const long long blocks = (long long)b * c_tiles * s_tiles;
  if (blocks > 0x7fffffffLL) return P2R_EINVAL;
  const int vec_ok = (S % 4 == 0) && (((uintptr_t)idx | (uintptr_t)out) % 16 == 0);
  hipLaunchKernelGGL(group_fwd_kernel, dim3((unsigned)blocks), dim3(GP_THREADS), 0, st, c, n, S,
                     s_tiles, c_tiles, vec_ok, points, idx, out);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

int group_backward(int b, int c, int n, int S, const float *grad_out, const int *idx,
                   float *grad_points, hipStream_t st) {
  if (b == 0 || c == 0 || n == 0) return P2R_OK;
  // deterministic gather form whenever the index list and at least 4 gradient rows fit the LDS
  const int SP = (S + 3) & ~3;
  const size_t per_row = (size_t)std::max(SP, 4) * sizeof(float);
  const size_t fit = (152 * 1024) / per_row;            // rows of SP floats (one of them the index list)
  if (S > 0 && fit >= 5) {
    const int sp = std::max(SP, 4);
    // sorted form: index lists of 256 .. 4096 slots whose keys fit 32 bits
    int NP = 256, SB = 8;
    while (NP < S) { NP <<= 1; ++SB; }
    if (S >= 256 && NP <= 4096 && (long long)n <= (1LL << (31 - SB))) {
      const size_t room = 152 * 1024 - (size_t)NP * sizeof(float);
      const int gfit = (int)(room / ((size_t)sp * sizeof(float)));
      // 8-row chunks: two workgroups per CU, one sorting while the other sums (108 vs 112 us with 16-row chunks)
      if (gfit >= 8 && c > 4) return group_grad_sort_launch<8>(b, c, n, S, sp, NP, SB, grad_out, idx, grad_points, st);
      if (gfit >= 4) return group_grad_sort_launch<4>(b, c, n, S, sp, NP, SB, grad_out, idx, grad_points, st);
    }
    // scan form: every destination walks the whole index list -- O(n * S / 4) LDS reads per workgroup.  Bounded: past
    // 2^22 (destination, slot) pairs (the P2RNet shapes are n <= 1024, S <= 2048) the order-free forms below take over
    // (LDS scatter / global atomics: same sums up to the order of the additions, documented in include/p2r_hip.h)
    if ((long long)n * sp <= (1LL << 22)) {
      if (fit >= 17 && c > 8) return group_grad_scan_launch<16>(b, c, n, S, sp, grad_out, idx, grad_points, st);
      if (fit >= 9 && c > 4) return group_grad_scan_launch<8>(b, c, n, S, sp, grad_out, idx, grad_points, st);
      return group_grad_scan_launch<4>(b, c, n, S, sp, grad_out, idx, grad_points, st);
    }
  }
  const size_t row_bytes = (size_t)n * sizeof(float);
  // Channel chunk: as many rows as fit 64 KiB of LDS (2 blocks / CU), up to 16,
  // while keeping >= ~512 blocks in flight when the problem is large enough.
  int gc = 16;
  while (gc > 1 && (gc * row_bytes > 64 * 1024)) gc >>= 1;
  while (gc > 1 && (long long)b * p2r_cdiv(c, gc) < 512 && gc > 4) gc >>= 1;
  if (gc * row_bytes <= 64 * 1024) {
    const int c_tiles = p2r_cdiv(c, gc);
    const long long blocks = (long long)b * c_tiles;
    if (blocks > 0x7fffffffLL) return P2R_EINVAL;
    const size_t lds = (size_t)gc * row_bytes;
    hipLaunchKernelGGL(group_grad_lds_kernel, dim3((unsigned)blocks), dim3(GG_THREADS), lds, st, c, n,
                       S, c_tiles, gc, grad_out, idx, grad_points);
    P2R_LAUNCH_CHECK();
    return P2R_OK;
  }
  hipError_t e = hipMemsetAsync(grad_points, 0, (size_t)b * c * n * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  const long long total = (long long)b * c * S;
  if (total == 0) return P2R_OK;
  const int blocks = (int)std::min<long long>(p2r_cdiv(total, GP_THREADS), 256 * 16);
  hipLaunchKernelGGL(group_grad_atomic_kernel, dim3(blocks), dim3(GP_THREADS), 0, st, c, n, S, total,
                     grad_out, idx, grad_points);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

}  // namespace

extern "C" int p2r_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                                const int *idx, float *out, void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return P2R_EINVAL;
  if ((long long)npoints * nsample > 0x7fffffffLL) return P2R_EINVAL;
  return group_forward(b, c, n, npoints * nsample, points, idx, out, p2r_stream(stream));
}

extern "C" int p2r_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                     const float *grad_out, const int *idx, float *grad_points,
                                     void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return P2R_EINVAL;
  if ((long long)npoints * nsample > 0x7fffffffLL) return P2R_EINVAL;
  return group_backward(b, c, n, npoints * nsample, grad_out, idx, grad_points, p2r_stream(stream));
}

extern "C" int p2r_gather_points(int b, int c, int n, int npoints, const float *points,
                                 const int *idx, float *out, void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0) return P2R_EINVAL;
  return group_forward(b, c, n, npoints, points, idx, out, p2r_stream(stream));
}

extern "C" int p2r_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                      const int *idx, float *grad_points, void *stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0) return P2R_EINVAL;
  return group_backward(b, c, n, npoints, grad_out, idx, grad_points, p2r_stream(stream));
}
