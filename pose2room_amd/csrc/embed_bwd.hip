// embed_bwd.hip -- backward of a 64 -> 64 pointwise layer of the embedding MLPs in ONE pass (gfx950).
//
// The embedding MLPs (reference models/p2rnet/modules/stgcn.py:46-63: SingleConv 'cbr', 'cbr', 'c' on (B, 3, T*J) joint
// offsets and (B, 3, T*knn) trajectory windows) carry 64-channel activations of 0.17-0.44 GB per layer at bs=32, T=1024
// through layers of 8 kFLOP per point: every pass over them is HBM time.  Layer by layer the backward used to be
// three passes (data gradient, BatchNorm-backward apply, weight gradient: 3.1 GB per layer); here it is one:
//
//   dz      = the gradient of this layer's conv output, formed while the tile is staged: either the incoming tensor
//             itself or the BatchNorm-backward form a*g + b*z + c of (masked gradient g, saved conv output z)
//   dW     += dz . A^T,  A = relu(zp * scale + shift) = the layer's input, recomputed from the previous layer's saved
//             conv output zp (never stored)                                           [MFMA, accumulators persistent]
//   g_prev  = (W^T dz) masked by A > 0, written once; per-channel sums (sum g_prev, sum g_prev * zp_hat) for the
//             previous BatchNorm's backward leave with it                              [MFMA + epilogue]
//   db     += row sums of dz
// i.e. read g, z, zp once (3 x 4 B per element), write g_prev once: 1.8 GB per layer.  A workgroup owns 64-column tiles
// in a persistent loop, the next tile's three 16 KB pieces are in flight (registers) while the current one is
// multiplied, W^T stays in registers for the whole kernel.
#include "p2r_common.h"

namespace {

typedef float floatx4v __attribute__((ext_vector_type(4)));

constexpr int EB_C = 64, EB_COLS = 64, EB_RS = 68;

struct EbArgs {
  const float *g, *z, *coef;      // dz = coef ? coef[0][c] g + coef[1][c] z + coef[2][c] : g
  const float *zp, *fin;          // previous conv output; fin [4][64] = mean, invstd, scale, shift of its BatchNorm
  const float *W;                 // [64 out][64 in]
  float *g_prev, *sums, *dw_part, *db_part;
  long long tiles;                // N * L / 64
  int L;
};

__global__ __launch_bounds__(256, 2) void embed_bwd_kernel(EbArgs p) {
  __shared__ float dzs[EB_C * EB_RS], acts[EB_C * EB_RS], zraw[EB_C * EB_RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const bool lazy = p.coef != nullptr;

  // W^T as the A operand of the data-gradient product: step s holds W[co = 4 s + g][ci = 16 wave + r]
  float wt[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) wt[s] = p.W[(4 * s + g) * EB_C + 16 * wave + r];
  // constants of this lane's four output rows ci = 16 wave + 4 g + q
  float c_mean[4], c_is[4], c_sc[4], c_sh[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ci = 16 * wave + 4 * g + q;
    c_mean[q] = p.fin[ci]; c_is[q] = p.fin[64 + ci]; c_sc[q] = p.fin[128 + ci]; c_sh[q] = p.fin[192 + ci];
  }
  // staging: thread -> rows (tid >> 4) + 16 i, float4 column tid & 15
  const int srow = tid >> 4, sc4 = tid & 15;
  float t_a[4], t_b[4], t_c[4], t_s[4], t_t[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = srow + 16 * i;
    t_a[i] = lazy ? p.coef[row] : 1.f; t_b[i] = lazy ? p.coef[64 + row] : 0.f; t_c[i] = lazy ? p.coef[128 + row] : 0.f;
    t_s[i] = p.fin[128 + row]; t_t[i] = p.fin[192 + row];
  }

  floatx4v accw[4];                                      // dW rows co = 16 wave + 4 g + q, columns ci = 16 n + r
#pragma unroll
  for (int n = 0; n < 4; ++n) accw[n] = floatx4v{0.f, 0.f, 0.f, 0.f};
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  float4 rg[4], rz[4], rp[4];
  auto fetch = [&](long long tile) {
    const long long col0 = tile * EB_COLS;
    const long long n = col0 / p.L;
    const size_t base = (size_t)n * EB_C * p.L + (size_t)(col0 - n * p.L) + 4 * sc4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t o = base + (size_t)(srow + 16 * i) * p.L;
      rg[i] = *reinterpret_cast<const float4 *>(p.g + o);
      if (lazy) rz[i] = *reinterpret_cast<const float4 *>(p.z + o);
      rp[i] = *reinterpret_cast<const float4 *>(p.zp + o);
    }
  };

  long long tile = blockIdx.x;
  if (tile < p.tiles) fetch(tile);
  for (; tile < p.tiles; tile += gridDim.x) {
    __syncthreads();                                     // the previous tile has been consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 16 * i;
      float4 d = rg[i];
      if (lazy) {
        d.x = t_a[i] * d.x + t_b[i] * rz[i].x + t_c[i]; d.y = t_a[i] * d.y + t_b[i] * rz[i].y + t_c[i];
        d.z = t_a[i] * d.z + t_b[i] * rz[i].z + t_c[i]; d.w = t_a[i] * d.w + t_b[i] * rz[i].w + t_c[i];
      }
      const float4 zp = rp[i];
      const float4 a = make_float4(fmaxf(zp.x * t_s[i] + t_t[i], 0.f), fmaxf(zp.y * t_s[i] + t_t[i], 0.f),
                                   fmaxf(zp.z * t_s[i] + t_t[i], 0.f), fmaxf(zp.w * t_s[i] + t_t[i], 0.f));
      *reinterpret_cast<float4 *>(dzs + row * EB_RS + 4 * sc4) = d;
      *reinterpret_cast<float4 *>(acts + row * EB_RS + 4 * sc4) = a;
      *reinterpret_cast<float4 *>(zraw + row * EB_RS + 4 * sc4) = zp;
    }
    __syncthreads();
    const long long col0 = tile * EB_COLS;
    const long long nidx = col0 / p.L;
    const size_t obase = (size_t)nidx * EB_C * p.L + (size_t)(col0 - nidx * p.L);
    if (tile + gridDim.x < p.tiles) fetch(tile + gridDim.x);

    // ---- data gradient: dA[ci][col] = sum_co W[co][ci] dz[co][col] ---------------------------------------------------
    floatx4v acca[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acca[n] = floatx4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float *brow = dzs + (4 * s + g) * EB_RS + r;
#pragma unroll
      for (int n = 0; n < 4; ++n) acca[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[s], brow[16 * n], acca[n], 0, 0, 0);
    }
    // ---- weight gradient: dW[co][ci] += sum_col dz[co][col] A[ci][col] -----------------------------------------------
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a = dzs[(16 * wave + r) * EB_RS + 4 * s + g];
#pragma unroll
      for (int n = 0; n < 4; ++n)
        accw[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acts[(16 * n + r) * EB_RS + 4 * s + g], accw[n], 0, 0, 0);
    }
    if (p.db_part) {
      const float *rowp = dzs + (tid >> 2) * EB_RS + 16 * (tid & 3);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += rowp[i];
      bsum += s;
    }
    // ---- mask, sums, store: lane holds rows ci = 16 wave + 4 g + q, columns 16 n + r ---------------------------------
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ci = 16 * wave + 4 * g + q;
      float *o = p.g_prev + obase + (size_t)ci * p.L + r;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float zr = zraw[ci * EB_RS + 16 * n + r];
        const float v = (zr * c_sc[q] + c_sh[q] > 0.f) ? acca[n][q] : 0.f;
        s1[q] += v;
        s2[q] += v * ((zr - c_mean[q]) * c_is[q]);
        o[16 * n] = v;
      }
    }
  }

  // ---- per-workgroup partials ---------------------------------------------------------------------------------------
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = p2r_row16_sum(s1[q]), b = p2r_row16_sum(s2[q]);
    if (r == 0) {
      float *e = p.sums + ((size_t)blockIdx.x * EB_C + 16 * wave + 4 * g + q) * 2;
      e[0] = a; e[1] = b;
    }
  }
  float *dw = p.dw_part + (size_t)blockIdx.x * EB_C * EB_C;
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int q = 0; q < 4; ++q) dw[(16 * wave + 4 * g + q) * EB_C + 16 * n + r] = accw[n][q];
  if (p.db_part) {
    bsum += __shfl_xor(bsum, 1);
    bsum += __shfl_xor(bsum, 2);
    if ((tid & 3) == 0) p.db_part[(size_t)blockIdx.x * EB_C + (tid >> 2)] = bsum;
  }
}

}  // namespace

// One-pass backward of a pointwise 64 -> 64 layer behind a BatchNorm + ReLU (see the header of this file).
//   g, z (N,64,L), coef [3][64]: dz = coef[0] g + coef[1] z + coef[2]; coef == NULL: dz = g (z unused)
//   zp (N,64,L), fin [4][64] = (mean, invstd, scale, shift): the layer's input is relu(zp * scale + shift)
//   W [64][64] (Conv1d weight, [out][in])
//   g_prev (N,64,L): (W^T dz) where the input is positive, else 0
//   sums [n_blocks][64][2] = (sum g_prev, sum g_prev * (zp - mean) * invstd) per workgroup
//   dw_part [n_blocks][64][64] (out, in), db_part [n_blocks][64] or NULL: per-workgroup partials of dW, db
// L % 64 == 0, all tensors 16-byte aligned.
extern "C" int p2r_embed_layer_backward(int N, int L, const float *g, const float *z, const float *coef, const float *zp,
                                        const float *fin, const float *W, float *g_prev, float *sums, int n_blocks,
                                        float *dw_part, float *db_part, void *stream) {
  if (N < 0 || L <= 0 || L % EB_COLS != 0 || n_blocks < 1 || !g || !zp || !fin || !W || !g_prev || !sums || !dw_part)
    return P2R_EINVAL;
  if (coef && !z) return P2R_EINVAL;
  if ((((uintptr_t)g | (uintptr_t)zp | (uintptr_t)g_prev | (uintptr_t)(z ? z : g)) & 15) != 0) return P2R_EINVAL;
  EbArgs a;
  a.g = g; a.z = z; a.coef = coef; a.zp = zp; a.fin = fin; a.W = W;
  a.g_prev = g_prev; a.sums = sums; a.dw_part = dw_part; a.db_part = db_part;
  a.tiles = (long long)N * L / EB_COLS;
  a.L = L;
  // workgroups beyond the tile count would write nothing: every workgroup must own at least one tile for its partials
  // to be defined, so the caller's n_blocks is clamped here and the unused partial rows are zeroed
  int blocks = n_blocks;
  if ((long long)blocks > a.tiles) blocks = (int)(a.tiles > 0 ? a.tiles : 1);
  if (blocks < n_blocks) {
    hipError_t e = hipMemsetAsync(sums + (size_t)blocks * 128, 0, (size_t)(n_blocks - blocks) * 128 * sizeof(float),
                                  p2r_stream(stream));
    if (e == hipSuccess)
      e = hipMemsetAsync(dw_part + (size_t)blocks * 4096, 0, (size_t)(n_blocks - blocks) * 4096 * sizeof(float),
                         p2r_stream(stream));
    if (e == hipSuccess && db_part)
      e = hipMemsetAsync(db_part + (size_t)blocks * 64, 0, (size_t)(n_blocks - blocks) * 64 * sizeof(float),
                         p2r_stream(stream));
    if (e != hipSuccess) return (int)e;
  }
  if (N == 0) {
    return (int)hipMemsetAsync(sums, 0, (size_t)blocks * 128 * sizeof(float), p2r_stream(stream));
  }
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, p2r_stream(stream), a);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
