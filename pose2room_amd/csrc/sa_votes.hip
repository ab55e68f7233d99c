// sa_votes.hip -- fused vote aggregation (PointnetSAModuleVotes hot chain), gfx950.
//
// Replaces the chain of PointnetSAModuleVotes.forward (reference
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
//
// Training: the same kernel also saves what the backward needs -- the grouped features G, the hidden activation H
// (both (B,256,M,16)) and the arg-max sample of every (channel, ball) -- and sa_votes_backward_kernel runs the two
// transposed layers on the same tiling: dZ2 (one non-zero per (channel, ball), at the arg-max, where the output is
// positive) -> dH = W2^T dZ2 -> dZ1 = dH * (H > 0) -> dG = W1^T dZ1.  dZ2, dZ1 and dG are written once; the weight
// gradients dW2 = dZ2 H^T, dW1 = dZ1 G^T are one split-K MFMA product each (gemm_nt_kernel), and dG goes back to
// the points through group_points_grad (LDS row accumulation, deterministic).
#include "pw_mma.h"

namespace {

constexpr int SA_C = 256;         // C0 = C1 = C2
constexpr int SA_S = 16;          // nsample == MFMA n-tile width
constexpr int SA_BALLS = 4;       // balls per workgroup (= waves)
constexpr int SA_COLS = SA_BALLS * SA_S;   // 64
constexpr int SA_RS = PW_RS;              // LDS row stride (floats)

// acc[m][n] = W[rows 64*wave + 16m .. +16][0 .. 256) . act[0 .. 256)[cols 16n .. +16]: the software-pipelined product of
// pw_mma.h (weights of the next 64 reduction indices are in flight while the current 64 are multiplied; as a plain
// load-then-multiply loop the L2 latency of every chunk was exposed: forward 0.34 -> see DESIGN.md)
__device__ __forceinline__ void sa_layer(const float *__restrict__ W, const float *__restrict__ act,
                                         int wave, int g, int r, floatx4v (&acc)[4][4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = floatx4v{0.f, 0.f, 0.f, 0.f};
  int rowm[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) rowm[m] = 64 * wave + 16 * m + r;
  pw_product<false>(W, act, SA_C, SA_C, SA_C / 16, 4, rowm, g, r, acc);
}

template <bool TRAIN>
__global__ __launch_bounds__(256, 2) void sa_votes_kernel(int n, int m, float radius2, const float *__restrict__ xyz,
                                                       const float *__restrict__ new_xyz,
                                                       const float *__restrict__ features,
                                                       const float *__restrict__ w1, const float *__restrict__ b1,
                                                       const float *__restrict__ w2, const float *__restrict__ b2,
                                                       int *__restrict__ idx, float *__restrict__ out,
                                                       float *__restrict__ G, float *__restrict__ H,
                                                       unsigned char *__restrict__ amax) {
  extern __shared__ float lds[];
  // ONE [256][SA_RS] tile: the gathered features during the first product, the hidden activation afterwards (a barrier
  // separates the last read of the one from the first write of the other).  With two tiles (139 KB) a CU held one
  // workgroup and its ball query, gather and epilogues ran with the matrix pipe idle; with one (70 KB) two workgroups
  // share a CU and one's products cover the other's serial phases.
  float *gs = lds;
  float *hs = lds;
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
    const int jc = j0 + (col >> 4);
    for (int c = tid >> 6; c < SA_C; c += 4) {
      const float v = feat[(size_t)c * n + src];
      gs[c * SA_RS + col] = v;
      if (TRAIN && jc < m) G[(((size_t)batch * SA_C + c) * m + jc) * SA_S + (col & 15)] = v;
    }
  }
  __syncthreads();

  floatx4v acc[4][4];
  // ---- layer 1: hs = relu(W1 . gs + b1) --------------------------------------------------
  sa_layer(w1, gs, wave, g, r, acc);
  __syncthreads();                         // every wave has finished reading the gathered features
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 64 * wave + 16 * mt + 4 * g + q;
      const float bb = b1[row];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float h = fmaxf(acc[mt][nt][q] + bb, 0.f);
        hs[row * SA_RS + 16 * nt + r] = h;
        if (TRAIN && j0 + nt < m) H[(((size_t)batch * SA_C + row) * m + j0 + nt) * SA_S + r] = h;
      }
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
        int am = r;                       // sample that holds the maximum; ties go to the lowest sample like max_pool2d
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
          const float ov = __shfl_xor(v, off, 16);
          const int oa = __shfl_xor(am, off, 16);
          if (ov > v || (ov == v && oa < am)) { v = ov; am = oa; }
        }
        const int j = j0 + nt;
        if (r == 0 && j < m) {
          out[((size_t)batch * SA_C + row) * m + j] = v;
          if (TRAIN) amax[((size_t)batch * SA_C + row) * m + j] = (unsigned char)am;
        }
      }
    }
}

// ---- backward of the two layers on the forward's tiling (4 balls = 64 columns per workgroup) -------------------
// dout, out (B,256,M); amax u8 (B,256,M); H (B,256,M,16) saved by the training forward; w2t = W2^T, w1t = W1^T
// (row = input channel of the layer).  Writes dZ2, dZ1, dG (B,256,M,16).
__global__ __launch_bounds__(256, 2) void sa_votes_backward_kernel(int m, const float *__restrict__ dout,
                                                                const float *__restrict__ out,
                                                                const unsigned char *__restrict__ amax,
                                                                const float *__restrict__ H,
                                                                const float *__restrict__ w2t,
                                                                const float *__restrict__ w1t,
                                                                float *__restrict__ dZ2, float *__restrict__ dZ1,
                                                                float *__restrict__ dG) {
  extern __shared__ float lds[];
  float *zs = lds;                         // [256][SA_RS] dZ2 tile ...
  float *ys = lds;                         // ... and, after a barrier, the dZ1 tile in the same place (see the forward)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int groups = (m + SA_BALLS - 1) / SA_BALLS;
  const int batch = blockIdx.x / groups;
  const int j0 = (blockIdx.x % groups) * SA_BALLS;

  // dZ2[c][ball, s] = dout[c][ball] at s = argmax when the pooled output is positive (ReLU), else 0
  {
    const int col = tid & 63, ball = col >> 4, smp = col & 15;
    const int j = j0 + ball;
    for (int c = tid >> 6; c < SA_C; c += 4) {
      float v = 0.f;
      if (j < m) {
        const size_t o = ((size_t)batch * SA_C + c) * m + j;
        if ((int)amax[o] == smp && out[o] > 0.f) v = dout[o];
        dZ2[o * SA_S + smp] = v;
      }
      zs[c * SA_RS + col] = v;
    }
  }
  __syncthreads();
  floatx4v acc[4][4];
  // dH = W2^T . dZ2 ;  dZ1 = dH * (H > 0)
  sa_layer(w2t, zs, wave, g, r, acc);
  __syncthreads();                         // every wave has finished reading dZ2
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 64 * wave + 16 * mt + 4 * g + q;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        float v = 0.f;
        if (j0 + nt < m) {
          const size_t o = (((size_t)batch * SA_C + row) * m + j0 + nt) * SA_S + r;
          v = H[o] > 0.f ? acc[mt][nt][q] : 0.f;
          dZ1[o] = v;
        }
        ys[row * SA_RS + 16 * nt + r] = v;
      }
    }
  __syncthreads();
  // dG = W1^T . dZ1
  sa_layer(w1t, ys, wave, g, r, acc);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 64 * wave + 16 * mt + 4 * g + q;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        if (j0 + nt < m) dG[(((size_t)batch * SA_C + row) * m + j0 + nt) * SA_S + r] = acc[mt][nt][q];
    }
}

// ---- C[M][N] partials of  sum_b sum_l A[b][m][l] * B[b][n][l]  (weight gradients of the 1x1 convs) -----------------
// A, B (nb, 256, L) row-major.  Grid = (4 x 4 output tiles of 64 x 64) x SPLIT column ranges; a workgroup of 4 waves
// (wave w: rows 16 w .. of every m-tile? no: n-tile w) walks its columns in chunks of 64 staged in LDS;
// v_mfma_f32_16x16x4_f32 with the reduction index = 4 consecutive columns.  partial [SPLIT][256][256].
constexpr int GN_CH = 64;                 // columns per LDS chunk
constexpr int GN_RS = GN_CH + 1;          // odd row stride: conflict-free reads down a column of rows
__global__ __launch_bounds__(256) void gemm_nt_kernel(int nb, int L, int split, const float *__restrict__ A,
                                                      const float *__restrict__ B, float *__restrict__ partial) {
  __shared__ float as[64 * GN_RS], bs[64 * GN_RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int tile = blockIdx.x % 16, part = blockIdx.x / 16;
  const int m0 = (tile >> 2) * 64, n0 = (tile & 3) * 64;
  const long long total = (long long)nb * L;                 // all columns, batch-major
  const long long per = ((total + split - 1) / split + GN_CH - 1) / GN_CH * GN_CH;
  const long long lo = part * per, hi = lo + per < total ? lo + per : total;
  floatx4v acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = floatx4v{0.f, 0.f, 0.f, 0.f};
  for (long long c0 = lo; c0 < hi; c0 += GN_CH) {
    __syncthreads();
    // stage 64 rows x 64 columns of both operands: 16-byte loads when a chunk lies inside one sample and is aligned
    if (L % GN_CH == 0 && c0 + GN_CH <= hi) {
      const long long bidx = c0 / L, l = c0 - bidx * L;
      for (int e = tid; e < 64 * (GN_CH / 4); e += 256) {
        const int row = e >> 4, c4 = e & 15;
        const float4 va = *reinterpret_cast<const float4 *>(A + ((size_t)bidx * 256 + m0 + row) * L + l + 4 * c4);
        const float4 vb = *reinterpret_cast<const float4 *>(B + ((size_t)bidx * 256 + n0 + row) * L + l + 4 * c4);
        float *pa = as + row * GN_RS + 4 * c4, *pb = bs + row * GN_RS + 4 * c4;
        pa[0] = va.x; pa[1] = va.y; pa[2] = va.z; pa[3] = va.w;
        pb[0] = vb.x; pb[1] = vb.y; pb[2] = vb.z; pb[3] = vb.w;
      }
    } else {
      for (int e = tid; e < 64 * GN_CH; e += 256) {
        const int row = e >> 6, cc = e & 63;
        const long long col = c0 + cc;
        float va = 0.f, vb = 0.f;
        if (col < hi) {
          const long long bidx = col / L, l = col - bidx * L;
          va = A[((size_t)bidx * 256 + m0 + row) * L + l];
          vb = B[((size_t)bidx * 256 + n0 + row) * L + l];
        }
        as[row * GN_RS + cc] = va;
        bs[row * GN_RS + cc] = vb;
      }
    }
    __syncthreads();
    // wave w owns output columns n0 + 16 w .. +16 for all four 16-row m-tiles
#pragma unroll 4
    for (int k = 0; k < GN_CH; k += 4) {
      const float b = bs[(16 * wave + r) * GN_RS + k + g];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[(16 * mt + r) * GN_RS + k + g], b, acc[mt], 0, 0, 0);
    }
  }
  float *outp = partial + (size_t)part * 256 * 256;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) outp[(size_t)(m0 + 16 * mt + 4 * g + q) * 256 + n0 + 16 * wave + r] = acc[mt][q];
}

}  // namespace

extern "C" int p2r_sa_votes_forward(int b, int n, int m, int nsample, float radius, int C0, int C1, int C2,
                                    const float *xyz, const float *new_xyz, const float *features,
                                    const float *w1, const float *b1, const float *w2, const float *b2,
                                    int *idx, float *out, float *G, float *H, unsigned char *amax, void *stream) {
  if (b < 0 || n <= 0 || m < 0) return P2R_EINVAL;
  const bool train = G != nullptr;
  if (train && (!H || !amax)) return P2R_EINVAL;
  if (nsample != SA_S || C0 != SA_C || C1 != SA_C || C2 != SA_C) return P2R_EINVAL;
  if (b == 0 || m == 0) return P2R_OK;
  const int groups = (m + SA_BALLS - 1) / SA_BALLS;
  const size_t lds = (size_t)SA_C * SA_RS * sizeof(float);
  auto kern = train ? sa_votes_kernel<true> : sa_votes_kernel<false>;
  static unsigned char lds_ok[2][P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(kern, lds_ok[train ? 1 : 0], (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(b * groups)), dim3(256), lds, p2r_stream(stream), n, m,
                     radius * radius, xyz, new_xyz, features, w1, b1, w2, b2, idx, out, G, H, amax);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// Backward of the fused vote aggregation (autograd of pointnet2_modules.py:236-243): dout, out (b,256,m); amax, H from
// the training forward; w2t / w1t = transposed layer weights [in][out].  Writes dZ2, dZ1, dG (b,256,m,16): the
// pre-activation gradients of the two layers (inputs of p2r_gemm_nt for the weight gradients) and the gradient of
// the grouped features (input of p2r_group_points_grad).
extern "C" int p2r_sa_votes_backward(int b, int m, int nsample, int C, const float *dout, const float *out,
                                     const unsigned char *amax, const float *H, const float *w2t, const float *w1t,
                                     float *dZ2, float *dZ1, float *dG, void *stream) {
  if (b < 0 || m < 0 || nsample != SA_S || C != SA_C) return P2R_EINVAL;
  if (b == 0 || m == 0) return P2R_OK;
  const int groups = (m + SA_BALLS - 1) / SA_BALLS;
  const size_t lds = (size_t)SA_C * SA_RS * sizeof(float);
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(sa_votes_backward_kernel, lds_ok, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(sa_votes_backward_kernel, dim3((unsigned)(b * groups)), dim3(256), lds, p2r_stream(stream), m,
                     dout, out, amax, H, w2t, w1t, dZ2, dZ1, dG);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// partial [split][256][256]:  sum over the leading axis = sum_b A[b] . B[b]^T  for A, B (nb,256,L) -- the weight
// gradient of a 1x1 convolution with 256 channels each side (dW = dZ . X^T over samples and positions).
extern "C" int p2r_gemm_nt_256(int nb, int L, int split, const float *A, const float *B, float *partial, void *stream) {
  if (nb < 0 || L <= 0 || split < 1 || split > 1024) return P2R_EINVAL;
  if (nb == 0) return P2R_OK;
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(16 * split), dim3(256), 0, p2r_stream(stream), nb, L, split, A, B, partial);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
