// seed_ops.hip -- the seams between the ST-GCN backbone and the rest of P2RNet (gfx950): frame gather at the seed
// frames in front of `conv_joint` (reference models/p2rnet/modules/stgcn.py:142-149) and the short-row reductions of
// the embedding (stgcn.py:105-130: mean over the 20-frame window, broadcast add over the 53 joints).
//
// The reference computes conv_joint on all T frames and gathers afterwards; conv_joint is pointwise in time, so only
// the seed frames go through the 3392 -> 256 product here, and the gather doubles as the re-layout from
// (B, 64, T, J) to (B, S, 64 J) rows.  As ATen ops that is an advanced-indexing gather (255 us at bs=32, T=1024) and,
// backward, a zero fill + index sort + indexing_backward + permute copy (400 us); here one streaming launch each way:
// forward one workgroup per (sample, seed) copies 64 runs of J floats into one contiguous row; backward one workgroup
// per (sample, 8 consecutive frames) finds the seeds that picked each frame (wave ballots over the seed list: any order,
// duplicates allowed), adds their rows in seed order (deterministic) and writes, per channel, one contiguous run of
// 8 x J floats, zeros included -- the whole gradient tensor is written exactly once.
#include "p2r_common.h"

namespace {

__global__ __launch_bounds__(256) void gather_frames_kernel(int C, int T, int J, int S, const float *__restrict__ x,
                                                            const long long *__restrict__ inds, float *__restrict__ out) {
  const int b = blockIdx.x / S;
  const long long t = inds[blockIdx.x];
  const int n = C * J;
  float *o = out + (size_t)blockIdx.x * n;
  if (t < 0 || t >= T) {                          // cannot happen with the model's seed selection; keep the row defined
    for (int e = threadIdx.x; e < n; e += 256) o[e] = 0.f;
    return;
  }
  const float *xb = x + ((size_t)b * C * T + t) * J;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int c = e / J, j = e - c * J;
    o[e] = xb[(size_t)c * T * J + j];
  }
}

constexpr int GF_TT = 8;            // frames per workgroup: runs of GF_TT * J floats (1696 bytes at J = 53) per channel
constexpr int GF_MAXHIT = 16;       // seeds per frame kept in the list; more (S >> T only) falls back to a scan

__global__ __launch_bounds__(256) void gather_frames_grad_kernel(int C, int T, int J, int S,
                                                                 const float *__restrict__ dout,
                                                                 const long long *__restrict__ inds,
                                                                 float *__restrict__ dx) {
  __shared__ int hits[GF_TT][GF_MAXHIT];
  __shared__ int nhit[GF_TT];
  const int chunks = (T + GF_TT - 1) / GF_TT;
  const int b = blockIdx.x / chunks, t0 = (blockIdx.x - b * chunks) * GF_TT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // wave w: the seeds of this sample that picked frames t0 + 2 w and t0 + 2 w + 1, ascending
  for (int f = 2 * wave; f < 2 * wave + 2; ++f) {
    const long long t = t0 + f;
    int cnt = 0;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      const bool hit = s < S && inds[(size_t)b * S + s] == t;
      const unsigned long long m = __ballot(hit);
      const int slot = cnt + (int)__builtin_popcountll(m & lt);
      if (hit && slot < GF_MAXHIT) hits[f][slot] = s;
      cnt += (int)__builtin_popcountll(m);
    }
    if (lane == 0) nhit[f] = cnt;
  }
  __syncthreads();
  const int n = C * J, run = GF_TT * J;
  for (int e = tid; e < run; e += 256) {          // thread <-> (frame, joint) of the chunk, all channels
    const int f = e / J, j = e - f * J;
    if (t0 + f >= T) continue;
    const int cnt = nhit[f];
    float *xb = dx + ((size_t)b * C * T + t0) * J + e;
    for (int c = 0; c < C; ++c) {
      float acc = 0.f;
      if (cnt <= GF_MAXHIT) {
        for (int i = 0; i < cnt; ++i) acc += dout[((size_t)b * S + hits[f][i]) * n + c * J + j];
      } else {
        for (int s = 0; s < S; ++s)
          if (inds[(size_t)b * S + s] == (long long)(t0 + f)) acc += dout[((size_t)b * S + s) * n + c * J + j];
      }
      xb[(size_t)c * T * J] = acc;
    }
  }
}

// out[r] = scale * sum_v x[r * V + v] for short rows (V <= 64): a workgroup stages 256 consecutive rows through LDS
// with 16-byte-free coalesced loads (the rows are contiguous), odd row stride in LDS, one row per thread.
__global__ __launch_bounds__(256) void rowsum_short_kernel(long long rows, int V, float scale, const float *__restrict__ x,
                                                           float *__restrict__ out) {
  extern __shared__ float tile[];
  const long long r0 = (long long)blockIdx.x * 256;
  const int nr = (int)(rows - r0 < 256 ? rows - r0 : 256);
  const int n = nr * V, Vp = V | 1;
  const float *src = x + r0 * V;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int r = i / V, v = i - r * V;
    tile[r * Vp + v] = src[i];
  }
  __syncthreads();
  if ((int)threadIdx.x < nr) {
    const float *p = tile + threadIdx.x * Vp;
    float a0 = 0.f, a1 = 0.f;
    int v = 0;
    for (; v + 2 <= V; v += 2) { a0 += p[v]; a1 += p[v + 1]; }
    if (v < V) a0 += p[v];
    out[r0 + threadIdx.x] = (a0 + a1) * scale;
  }
}

}  // namespace

// x (b, c, t, j) f32, inds (b, s) int64 frame indices -> out (b, s, c * j): out[b, s, ci * j + ji] = x[b, ci, inds[b, s], ji]
// (the rows `conv_joint` multiplies: stgcn.py:142-149 with the gather moved in front of the pointwise convolution).
extern "C" int p2r_gather_frames(int b, int c, int t, int j, int s, const float *x, const long long *inds, float *out,
                                 void *stream) {
  if (b < 0 || c <= 0 || t <= 0 || j <= 0 || s < 0) return P2R_EINVAL;
  if (b == 0 || s == 0) return P2R_OK;
  hipLaunchKernelGGL(gather_frames_kernel, dim3((unsigned)(b * s)), dim3(256), 0, p2r_stream(stream), c, t, j, s, x, inds,
                     out);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// its gradient: dout (b, s, c * j) -> dx (b, c, t, j), overwritten everywhere (frames no seed picked get zeros; a frame
// picked by several seeds gets the sum of their rows, added in seed order).
extern "C" int p2r_gather_frames_grad(int b, int c, int t, int j, int s, const float *dout, const long long *inds,
                                      float *dx, void *stream) {
  if (b < 0 || c <= 0 || t <= 0 || j <= 0 || s < 0) return P2R_EINVAL;
  if (b == 0) return P2R_OK;
  hipLaunchKernelGGL(gather_frames_grad_kernel, dim3((unsigned)(b * ((t + GF_TT - 1) / GF_TT))), dim3(256), 0,
                     p2r_stream(stream), c, t, j, s, dout, inds, dx);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// out[r] = scale * sum_v x[r * v_len + v], r < rows, 1 <= v_len <= 64.
extern "C" int p2r_rowsum_short(long long rows, int v_len, float scale, const float *x, float *out, void *stream) {
  if (rows < 0 || v_len < 1 || v_len > 64) return P2R_EINVAL;
  if (rows == 0) return P2R_OK;
  const size_t lds = 256 * (size_t)(v_len | 1) * sizeof(float);
  hipLaunchKernelGGL(rowsum_short_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), lds, p2r_stream(stream), rows,
                     v_len, scale, x, out);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
