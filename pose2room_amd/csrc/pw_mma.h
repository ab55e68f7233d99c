// pw_mma.h -- the fp32 MFMA product shared by the point-wise layer kernels (pw_layers.hip) and the fused vote
// aggregation (sa_votes.hip): acc[m][n] += W[row tile m][k] . tile[k][16 n + column], for one wave, four 16-row tiles x
// four 16-column tiles, B operands from an LDS tile of row stride PW_RS, A operands streamed from L2 and software-pipelined
// one group of 64 k ahead.
#pragma once
#include "p2r_common.h"

namespace {

typedef float floatx4v __attribute__((ext_vector_type(4)));

constexpr int PW_RS = 68;         // LDS row stride (floats): 4 * 68 = 16 mod 32 banks

// ---- A-operand (weight) groups: 4 chunks of 16 k, 4 m-tiles --------------------------------------------------------
// chunk c covers k = 16 c + 4 g + s (s = 0..3 is the MFMA step, g the lane's k slot): row-major weights give one
// 16-byte load per (chunk, m-tile); transposed reads four 4-byte loads, each coalesced over the 16 rows of the tile.
template <bool WT>
__device__ __forceinline__ void pw_load_group(const float *__restrict__ W, int K, int rows, int c0, int nch, int nmt,
                                              const int (&rowm)[4], int g, float4 (&A)[4][4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (c0 + j < nch) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m < nmt) {
          const int kk = 16 * (c0 + j) + 4 * g;
          if (!WT) {
            A[j][m] = *reinterpret_cast<const float4 *>(W + (size_t)rowm[m] * K + kk);
          } else {
            const int k0 = min(kk + 0, K - 1), k1 = min(kk + 1, K - 1), k2 = min(kk + 2, K - 1), k3 = min(kk + 3, K - 1);
            A[j][m].x = W[(size_t)k0 * rows + rowm[m]];
            A[j][m].y = W[(size_t)k1 * rows + rowm[m]];
            A[j][m].z = W[(size_t)k2 * rows + rowm[m]];
            A[j][m].w = W[(size_t)k3 * rows + rowm[m]];
          }
        }
      }
    }
  }
}

__device__ __forceinline__ void pw_compute_group(const float *__restrict__ tile, int c0, int nch, int nmt, int g, int r,
                                                 const float4 (&A)[4][4], floatx4v (&acc)[4][4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (c0 + j < nch) {
      const float *brow = tile + (16 * (c0 + j) + 4 * g) * PW_RS + r;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float b[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = brow[s * PW_RS + 16 * n];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (m < nmt) {
            const float a = s == 0 ? A[j][m].x : s == 1 ? A[j][m].y : s == 2 ? A[j][m].z : A[j][m].w;
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[m][n], 0, 0, 0);
          }
        }
      }
    }
  }
}

template <bool WT>
__device__ __forceinline__ void pw_product(const float *__restrict__ W, const float *__restrict__ tile, int K, int rows,
                                           int nch, int nmt, const int (&rowm)[4], int g, int r,
                                           floatx4v (&acc)[4][4]) {
  float4 A0[4][4], A1[4][4];
  pw_load_group<WT>(W, K, rows, 0, nch, nmt, rowm, g, A0);
  for (int c0 = 0; c0 < nch; c0 += 8) {
    if (c0 + 4 < nch) pw_load_group<WT>(W, K, rows, c0 + 4, nch, nmt, rowm, g, A1);
    pw_compute_group(tile, c0, nch, nmt, g, r, A0, acc);
    if (c0 + 4 < nch) {
      if (c0 + 8 < nch) pw_load_group<WT>(W, K, rows, c0 + 8, nch, nmt, rowm, g, A0);
      pw_compute_group(tile, c0 + 4, nch, nmt, g, r, A1, acc);
    }
  }
}

}  // namespace
