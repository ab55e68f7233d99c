// sa_votes.hip -- fused vote aggregation (PointnetSAModuleVotes hot chain), gfx950.
//
// Replaces, for inference, the chain of PointnetSAModuleVotes.forward (reference
// pointnet2_modules.py:220-259 with mlp=[256,256,256], bn=False, use_xyz=False,
// pooling='max'):
//     ball_query -> group_points(xyz) [dead] -> group_points(features)
//       -> Conv2d1x1+ReLU -> Conv2d1x1+ReLU -> max_pool over nsample
// i.e. ~8 launches and a (B,256,128,16) tensor written and re-read three times.
//
// MI355X design: a workgroup owns 4 balls of one cloud.  Each of its 4 waves runs the
// ballot ball query for one centre, the 4 x 16 neighbour feature columns are gathered
// into LDS once (256 x 64 floats), and the two 256x256 layers run on
// v_mfma_f32_16x16x4_f32 (exact fp32): a 16-column MFMA n-tile is exactly one ball, so
// the max over nsample is a 16-lane butterfly on the accumulator tile.  The hidden
// activation only ever lives in LDS.  Weights stream from L2 as 64-channel chunks of
// A operands.  HBM: features gathered once, (B,256,128) written once.
#include "p2r_common.h"

namespace {

typedef float floatx4v __attribute__((ext_vector_type(4)));

constexpr int SA_C = 256;         // C0 = C1 = C2
constexpr int SA_S = 16;          // nsample == MFMA n-tile width
constexpr int SA_BALLS = 4;       // balls per workgroup (= waves)
constexpr int SA_COLS = SA_BALLS * SA_S;   // 64
constexpr int SA_RS = SA_COLS + 4;         // LDS row stride (floats)

// acc[m][n] += W[rows 64*wave + 16m .. +16][K chunk] . act[K chunk][cols 16n .. +16]
__device__ __forceinline__ void sa_layer(const float *__restrict__ W, const float *__restrict__ act,
                                         int wave, int g, int r, floatx4v (&acc)[4][4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = floatx4v{0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < SA_C; kc += 64) {
    float a[4][16];   // W[row 64*wave + 16m + r][kc + 16g .. +16)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float4 *wp = reinterpret_cast<const float4 *>(W + (size_t)(64 * wave + 16 * m + r) * SA_C + kc + 16 * g);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 u = wp[q];
        a[m][4 * q + 0] = u.x; a[m][4 * q + 1] = u.y; a[m][4 * q + 2] = u.z; a[m][4 * q + 3] = u.w;
      }
    }
    const float *brow = act + (kc + 16 * g) * SA_RS + r;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float b[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = brow[s * SA_RS + 16 * n];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[n], acc[m][n], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(256) void sa_votes_kernel(int n, int m, float radius2, const float *__restrict__ xyz,
                                                       const float *__restrict__ new_xyz,
                                                       const float *__restrict__ features,
                                                       const float *__restrict__ w1, const float *__restrict__ b1,
                                                       const float *__restrict__ w2, const float *__restrict__ b2,
                                                       int *__restrict__ idx, float *__restrict__ out) {
  extern __shared__ float lds[];
  float *gs = lds;                         // [256][SA_RS] gathered features
  float *hs = lds + SA_C * SA_RS;          // [256][SA_RS] hidden activation
  __shared__ int s_idx[SA_BALLS][SA_S];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int groups = (m + SA_BALLS - 1) / SA_BALLS;
  const int batch = blockIdx.x / groups;
  const int j0 = (blockIdx.x % groups) * SA_BALLS;
  const float *pts = xyz + (size_t)batch * n * 3;
  const float *feat = features + (size_t)batch * SA_C * n;

  // ---- ball query: wave w <-> centre j0 + w (same rule as ball_query.hip) ----------
  {
    const int j = j0 + wave;
    int cnt = 0, first = 0;
    if (j < m) {
      const float cx = new_xyz[((size_t)batch * m + j) * 3 + 0];
      const float cy = new_xyz[((size_t)batch * m + j) * 3 + 1];
      const float cz = new_xyz[((size_t)batch * m + j) * 3 + 2];
      const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      for (int k0 = 0; k0 < n && cnt < SA_S; k0 += 64) {
        const int k = k0 + lane;
        bool hit = false;
        if (k < n) hit = p2r_sqdist(cx, cy, cz, pts[k * 3 + 0], pts[k * 3 + 1], pts[k * 3 + 2]) < radius2;
        const unsigned long long mask = __ballot(hit);
        if (mask != 0ull) {
          if (cnt == 0) first = k0 + (int)__builtin_ctzll(mask);
          const int slot = cnt + (int)__builtin_popcountll(mask & lt_mask);
          if (hit && slot < SA_S) s_idx[wave][slot] = k;
          cnt += (int)__builtin_popcountll(mask);
        }
      }
    }
    const int filled = min(cnt, SA_S);
    if (lane >= filled && lane < SA_S) s_idx[wave][lane] = cnt > 0 ? first : 0;
  }
  __syncthreads();
  if (tid < SA_COLS) {
    const int j = j0 + (tid >> 4);
    if (j < m) idx[((size_t)batch * m + j) * SA_S + (tid & 15)] = s_idx[tid >> 4][tid & 15];
  }

  // ---- gather the 64 neighbour columns of all 256 channels into LDS ----------------
  {
    const int col = tid & 63;
    const int src = s_idx[col >> 4][col & 15];
    for (int c = tid >> 6; c < SA_C; c += 4) gs[c * SA_RS + col] = feat[(size_t)c * n + src];
  }
  __syncthreads();

  floatx4v acc[4][4];
  // ---- layer 1: hs = relu(W1 . gs + b1) --------------------------------------------------
  sa_layer(w1, gs, wave, g, r, acc);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 64 * wave + 16 * mt + 4 * g + q;
      const float bb = b1[row];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) hs[row * SA_RS + 16 * nt + r] = fmaxf(acc[mt][nt][q] + bb, 0.f);
    }
  __syncthreads();
  // ---- layer 2 + max over the 16 samples of each ball ------------------------------------
  sa_layer(w2, hs, wave, g, r, acc);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 64 * wave + 16 * mt + 4 * g + q;
      const float bb = b2[row];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        float v = fmaxf(acc[mt][nt][q] + bb, 0.f);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 16));
        const int j = j0 + nt;
        if (r == 0 && j < m) out[((size_t)batch * SA_C + row) * m + j] = v;
      }
    }
}

}  // namespace

extern "C" int p2r_sa_votes_forward(int b, int n, int m, int nsample, float radius, int C0, int C1, int C2,
                                    const float *xyz, const float *new_xyz, const float *features,
                                    const float *w1, const float *b1, const float *w2, const float *b2,
                                    int *idx, float *out, void *stream) {
  if (b < 0 || n <= 0 || m < 0) return P2R_EINVAL;
  if (nsample != SA_S || C0 != SA_C || C1 != SA_C || C2 != SA_C) return P2R_EINVAL;
  if (b == 0 || m == 0) return P2R_OK;
  const int groups = (m + SA_BALLS - 1) / SA_BALLS;
  const size_t lds = 2 * (size_t)SA_C * SA_RS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)sa_votes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(sa_votes_kernel, dim3((unsigned)(b * groups)), dim3(256), lds, p2r_stream(stream), n, m,
                     radius * radius, xyz, new_xyz, features, w1, b1, w2, b2, idx, out);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
