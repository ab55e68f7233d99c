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
    if (cnt <= 2) {        // the usual case (S <= T, seeds spread over the sequence): no inner loop, eight channels in flight
      const float *r0 = dout + ((size_t)b * S + (cnt > 0 ? hits[f][0] : 0)) * n + j;
      const float *r1 = dout + ((size_t)b * S + (cnt > 1 ? hits[f][1] : 0)) * n + j;
      int c = 0;
      for (; c + 8 <= C; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v[u] = cnt > 0 ? r0[(c + u) * J] : 0.f;
          if (cnt > 1) v[u] += r1[(c + u) * J];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) xb[(size_t)(c + u) * T * J] = v[u];
      }
      for (; c < C; ++c) {
        float v = cnt > 0 ? r0[c * J] : 0.f;
        if (cnt > 1) v += r1[c * J];
        xb[(size_t)c * T * J] = v;
      }
      continue;
    }
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

// ---- tail of the vote head (vote_center.py:50-58, network.py:68-71) ----------------------------------------------------
// net (B,S,3+C) = conv_input's output per seed: xyz offset + feature residual.  vote_xyz = hip + net[..., :3];
// f = seed_features + net[..., 3:]; vote_features = f / |f|_2 (no epsilon, as the reference) -- written channel-major
// (B,C,S), the layout the vote aggregation reads, so the (B,S,C) tensor the reference holds is its transposed view.
// One workgroup per 64 seeds: a wave reads whole (B,S,.) rows (one seed = 64 lanes x 4 channels 64 apart), reduces the norm
// with lane shuffles, and the normalised tile is transposed through LDS.  C = 256.
constexpr int VF_C = 256, VF_COLS = 64, VF_RS = VF_COLS + 1;

__global__ __launch_bounds__(256) void vote_finish_kernel(int S, const float *__restrict__ net,
                                                          const float *__restrict__ seed_features,
                                                          const float *__restrict__ hip, float *__restrict__ vote_xyz,
                                                          float *__restrict__ feat_ncl, float *__restrict__ inv_norm) {
  extern __shared__ float tile[];                       // [256][VF_RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t col0 = (size_t)blockIdx.x * VF_COLS;      // S % 64 == 0: the tile lies inside one sample
  const int b = (int)(col0 / S), s0 = (int)(col0 - (size_t)b * S);
  for (int i = 0; i < 16; ++i) {
    const int cl = 16 * wave + i;
    const size_t col = col0 + cl;
    const float *nr = net + col * (3 + VF_C);
    const float *sr = seed_features + col * VF_C;
    float f[4];                                           // channels lane, lane + 64, lane + 128, lane + 192
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = sr[lane + 64 * e] + nr[3 + lane + 64 * e];
    float ss = (f[0] * f[0] + f[1] * f[1]) + (f[2] * f[2] + f[3] * f[3]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = 1.f / sqrtf(ss);
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(lane + 64 * e) * VF_RS + cl] = f[e] * inv;
    if (lane < 3) vote_xyz[col * 3 + lane] = hip[col * 3 + lane] + nr[lane];
    if (lane == 3) inv_norm[col] = inv;
  }
  __syncthreads();
  float *ob = feat_ncl + (size_t)b * VF_C * S + s0;
  for (int c = wave; c < VF_C; c += 4) ob[(size_t)c * S + lane] = tile[c * VF_RS + lane];
}

// gradient: d_xyz (B,S,3), d_feat (B,C,S) w.r.t. the normalised features, feat (B,C,S), inv_norm (B,S) ->
// d_net (B,S,3+C) = (d_xyz, df), d_sf (B,S,C) = df, df = inv (dy - y (y . dy)).
__global__ __launch_bounds__(256) void vote_finish_grad_kernel(int S, const float *__restrict__ d_xyz,
                                                               const float *__restrict__ d_feat,
                                                               const float *__restrict__ feat,
                                                               const float *__restrict__ inv_norm,
                                                               float *__restrict__ d_net, float *__restrict__ d_sf) {
  extern __shared__ float tile[];                       // dy [256][VF_RS], y [256][VF_RS]
  float *dy = tile, *y = tile + VF_C * VF_RS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t col0 = (size_t)blockIdx.x * VF_COLS;
  const int b = (int)(col0 / S), s0 = (int)(col0 - (size_t)b * S);
  const float *gb = d_feat + (size_t)b * VF_C * S + s0, *yb = feat + (size_t)b * VF_C * S + s0;
  for (int c = wave; c < VF_C; c += 4) {
    dy[c * VF_RS + lane] = d_feat ? gb[(size_t)c * S + lane] : 0.f;
    y[c * VF_RS + lane] = yb[(size_t)c * S + lane];
  }
  __syncthreads();
  for (int i = 0; i < 16; ++i) {
    const int cl = 16 * wave + i;
    const size_t col = col0 + cl;
    float g[4], v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { g[e] = dy[(lane + 64 * e) * VF_RS + cl]; v[e] = y[(lane + 64 * e) * VF_RS + cl]; }
    float dot = (g[0] * v[0] + g[1] * v[1]) + (g[2] * v[2] + g[3] * v[3]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, 64);
    const float inv = inv_norm[col];
    float df[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) df[e] = inv * (g[e] - v[e] * dot);
    float *nr = d_net + col * (3 + VF_C);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      nr[3 + lane + 64 * e] = df[e];
      d_sf[col * VF_C + lane + 64 * e] = df[e];
    }
    if (lane < 3) nr[lane] = d_xyz ? d_xyz[col * 3 + lane] : 0.f;
  }
}

// ---- seed selection by arc length (stgcn.py:96-101) --------------------------------------------------------------------
// inds[b, s] = argmin_t |cum[b, t] - target[b, s]| (first minimum), exactly the fp32 expression torch evaluates
// (`torch.argmin(torch.abs(cum.unsqueeze(-1) - target.unsqueeze(1)), dim=1)`) without the (B, T, S) difference tensor:
// the cumulative arc length of a sample sits in LDS, a thread walks it for one target.
__global__ __launch_bounds__(256) void nearest_prefix_kernel(int T, int S, const float *__restrict__ cum,
                                                             const float *__restrict__ target,
                                                             long long *__restrict__ inds) {
  extern __shared__ float row[];
  const int b = blockIdx.y;
  for (int t = threadIdx.x; t < T; t += 256) row[t] = cum[(size_t)b * T + t];
  __syncthreads();
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const float tg = target[(size_t)b * S + s];
  float best = fabsf(row[0] - tg);
  int arg = 0;
  for (int t = 1; t < T; ++t) {
    const float d = fabsf(row[t] - tg);
    if (d < best) { best = d; arg = t; }          // strict: the first minimum wins, like argmin; NaN never wins
  }
  inds[(size_t)b * S + s] = arg;
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

// Tail of the vote head, see vote_finish_kernel.  net (b,s,3+256), seed_features (b,s,256), hip (b,s,3) ->
// vote_xyz (b,s,3), feat_ncl (b,256,s) = normalised vote features channel-major, inv_norm (b,s) (saved for the gradient).
// s % 64 == 0, C = 256.
extern "C" int p2r_vote_finish(int b, int s, int C, const float *net, const float *seed_features, const float *hip,
                               float *vote_xyz, float *feat_ncl, float *inv_norm, void *stream) {
  if (b < 0 || s <= 0 || s % VF_COLS != 0 || C != VF_C) return P2R_EINVAL;
  if (b == 0) return P2R_OK;
  const size_t lds = (size_t)VF_C * VF_RS * sizeof(float);
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(vote_finish_kernel, lds_ok, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(vote_finish_kernel, dim3((unsigned)((size_t)b * s / VF_COLS)), dim3(256), lds, p2r_stream(stream), s, net,
                     seed_features, hip, vote_xyz, feat_ncl, inv_norm);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// its gradient: d_xyz (b,s,3) or NULL, d_feat (b,256,s) or NULL, feat_ncl / inv_norm from the forward ->
// d_net (b,s,3+256), d_sf (b,s,256).
extern "C" int p2r_vote_finish_grad(int b, int s, int C, const float *d_xyz, const float *d_feat, const float *feat_ncl,
                                    const float *inv_norm, float *d_net, float *d_sf, void *stream) {
  if (b < 0 || s <= 0 || s % VF_COLS != 0 || C != VF_C) return P2R_EINVAL;
  if (b == 0) return P2R_OK;
  const size_t lds = (size_t)2 * VF_C * VF_RS * sizeof(float);
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(vote_finish_grad_kernel, lds_ok, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(vote_finish_grad_kernel, dim3((unsigned)((size_t)b * s / VF_COLS)), dim3(256), lds, p2r_stream(stream),
                     s, d_xyz, d_feat, feat_ncl, inv_norm, d_net, d_sf);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// inds[b, s] = first t minimising |cum[b, t] - target[b, s]| in fp32; cum (b,t), target (b,s) f32, inds (b,s) int64.
extern "C" int p2r_nearest_prefix(int b, int t, int s, const float *cum, const float *target, long long *inds,
                                  void *stream) {
  if (b < 0 || t <= 0 || s < 0 || t > 40000) return P2R_EINVAL;
  if (b == 0 || s == 0) return P2R_OK;
  static unsigned char lds_ok[P2R_MAX_DEVICES];   // t > 16384: more than the 64 KB a kernel may use by default
  hipError_t e = p2r_allow_big_lds(nearest_prefix_kernel, lds_ok, 40000 * (int)sizeof(float));
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nearest_prefix_kernel, dim3((unsigned)((s + 255) / 256), (unsigned)b), dim3(256),
                     (size_t)t * sizeof(float), p2r_stream(stream), t, s, cum, target, inds);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
