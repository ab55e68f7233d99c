// ball_query.hip -- radius neighbour query for gfx950.
//
// Replaces query_ball_point_kernel (reference _ext-src/src/ball_query_gpu.cu:9-54),
// which gives each centre to ONE thread that scans all n points serially (one
// block per cloud: 32 blocks on a 256-CU chip at bs=32).
//
// MI355X design: one wave per centre.  The cloud is staged once per block in
// LDS as float4 (coalesced HBM read, conflict-free ds_read_b128); the 64 lanes
// test 64 consecutive points per step, a 64-bit ballot plus a lane-prefix
// popcount turns the hit mask into output slots in ascending index order --
// exactly the order the reference's serial scan produces -- and the wave stops
// as soon as nsample slots are filled.  Grid = b * ceil(m / 16) workgroups, so even the P2RNet shape (b=32, m=128) fills the chip.
#include "p2r_common.h"

namespace {

constexpr int BQ_WAVES = 4;    // waves per block
constexpr int BQ_CPW = 4;      // centres per wave, state kept in registers
constexpr int BQ_CPB = BQ_WAVES * BQ_CPW;
constexpr int BQ_TILE = 2048;  // points staged in LDS per pass (32 KiB)

__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(
    int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx) {
  __shared__ float4 s_pts[BQ_TILE];
  const int batch = blockIdx.y;
  const float *pts = xyz + (size_t)batch * n * 3;
  const float *ctr = new_xyz + (size_t)batch * m * 3;
  int *out = idx + (size_t)batch * m * nsample;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j0 = blockIdx.x * BQ_CPB + wave * BQ_CPW;  // this wave's first centre
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  // (cnt, first) of each owned centre carry across LDS tiles; statically indexed
  // so they stay in (scalar) registers.
  int cnt[BQ_CPW], first[BQ_CPW];
  float cx[BQ_CPW], cy[BQ_CPW], cz[BQ_CPW];
#pragma unroll
  for (int c = 0; c < BQ_CPW; ++c) {
    const int j = j0 + c;
    cnt[c] = (j < m) ? 0 : nsample;  // out-of-range centres are "already full"
    first[c] = 0;
    const int jj = (j < m) ? j : 0;
    cx[c] = ctr[jj * 3 + 0]; cy[c] = ctr[jj * 3 + 1]; cz[c] = ctr[jj * 3 + 2];
  }

  for (int base = 0; base < n; base += BQ_TILE) {
    const int tile_n = min(BQ_TILE, n - base);
    if (base > 0) __syncthreads();  // previous tile fully consumed
    for (int t = threadIdx.x; t < tile_n; t += BQ_WAVES * 64) {
      const float *p = pts + (size_t)(base + t) * 3;
      s_pts[t] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();

#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) {
      const int j = j0 + c;
      for (int k0 = 0; k0 < tile_n && cnt[c] < nsample; k0 += 64) {
        const int k = k0 + lane;
        bool hit = false;
        if (k < tile_n) {
          const float4 p = s_pts[k];
          // reference order: (new - p)^2 summed x, y, z
          const float d2 = p2r_sqdist(cx[c], cy[c], cz[c], p.x, p.y, p.z);
          hit = d2 < radius2;
        }
        const unsigned long long mask = __ballot(hit);
        if (mask != 0ull) {
          if (cnt[c] == 0) first[c] = base + k0 + (int)__builtin_ctzll(mask);
          const int slot = cnt[c] + (int)__builtin_popcountll(mask & lt_mask);
          if (hit && slot < nsample) out[j * nsample + slot] = base + k;
          cnt[c] += (int)__builtin_popcountll(mask);
        }
      }
    }
  }

  // Tail fill: slots [cnt, nsample) repeat the first hit (the reference fills
  // every slot with the first hit, then overwrites the leading ones); a centre
  // with no hit gets the zeros the reference's zero-initialised output keeps.
#pragma unroll
  for (int c = 0; c < BQ_CPW; ++c) {
    const int j = j0 + c;
    if (j < m) {
      const int filled = min(cnt[c], nsample);
      const int fill = cnt[c] > 0 ? first[c] : 0;
      for (int s = filled + lane; s < nsample; s += 64) out[j * nsample + s] = fill;
    }
  }
}

}  // namespace

extern "C" int p2r_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                              const float *xyz, int *idx, void *stream) {
  if (b < 0 || n < 0 || m < 0 || nsample < 0) return P2R_EINVAL;
  if (b == 0 || m == 0 || nsample == 0) return P2R_OK;
  const float radius2 = radius * radius;  // ball_query_gpu.cu:22, fp32
  dim3 grid(p2r_cdiv(m, BQ_CPB), b);
  hipLaunchKernelGGL(ball_query_kernel, grid, dim3(BQ_WAVES * 64), 0, p2r_stream(stream), n, m,
                     radius2, nsample, new_xyz, xyz, idx);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
