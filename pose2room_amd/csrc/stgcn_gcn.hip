// stgcn_gcn.hip -- fused spatial graph convolution of the ST-GCN backbone, gfx950.
//
// First generation.  The forward / data-gradient kernel below (gcn_fused_kernel) now serves joint counts other than
// 53 only; for the P2RNet skeleton that op runs on stgcn_gcn2.hip.  The weight- and adjacency-gradient kernels
// (gcn_dw_kernel, gcn_dcoef_kernel) are in use for every shape.
//
// Replaces ConvTemporalGraphical.forward (reference
// models/p2rnet/modules/stgcn_layers.py:57-67):
//     y = Conv2d_1x1(x)            (N,64,T,V) -> (N, K*64, T, V)   K = 11 partitions
//     z = einsum('nkctv,kvw->nctw', y.view(N,K,64,T,V), A*importance)
// which is 98.8 % of the reference step's FLOPs and, unfused, round-trips an
// 11x-wide (N,704,T,V) tensor through HBM several times per block.
//
// MI355X design.  Per frame the op is  Z = sum_k W_k . (X . A_k) + bias-term with
// X (64 x V), W_k (64 x 64) and A_k (V x V).  Two facts shape the kernel:
//   * the K adjacency planes are 97 % zeros (971 non-zeros of 30 899 for the
//     53-joint skeleton, max_hop 5), and `A * importance` keeps that support, so the
//     graph product X . A_k is a short gather-FMA per element, not a GEMM;
//   * what is left is a dense 64 x (K*64) by (K*64) x columns product: MFMA work.
// A workgroup owns a tile of F frames of one sequence (F*V <= 384 columns), stages
// the X tile in LDS once, and every lane builds its MFMA B-operand values
// (X . A_k at its own column) on the fly from LDS with the per-column neighbour
// list of plane k -- the aggregated tensor never exists in memory -- while
// v_mfma_f32_16x16x4_f32 (exact fp32) accumulates Z over the K planes in registers.
// W_k streams from L2 straight into A-operand VGPRs.  HBM traffic is the
// algorithmic minimum: read X once, write Z once.
//
// The same kernel computes the data gradient dX = sum_k W_k^T . (dZ . A_k^T) when
// it is given the transposed weights and the row-wise neighbour lists.
//
// K order inside an MFMA step is permuted (lane-half h, step s  <->  channel
// 32*h + s) so that a lane's 32 A-operand values are 128 contiguous bytes of a W
// row and its B-operand rows advance by a constant LDS stride.
#include "p2r_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int GC_C = 64;        // channels (in == out for every P2RNet block)
constexpr int GC_NP = 384;      // padded tile width (columns) = 12 MFMA n-tiles of 32
constexpr int GC_MAXK = 16;
constexpr int GC_MAXL = 12;     // longest neighbour list supported per (plane, column)

struct GcnParams {
  int T, V, K, F;               // frames, joints, planes, frames per tile
  int tiles_per_seq;
  int Lk[GC_MAXK];              // neighbour-list length of plane k
  int Lofs[GC_MAXK];            // row offset of plane k in the nbr / coef tables
};

typedef float floatx4_t __attribute__((ext_vector_type(4)));

constexpr int GC_ROW4 = GC_NP + 1;  // float4 elements per group of four rows (X tile is row-interleaved)
constexpr int GC_THREADS = 512;     // 8 waves: two per SIMD, so one wave's LDS / L2 waits hide
                                    // under the other's MFMAs
constexpr int GC_NT16 = 3;          // 16-column n-tiles per wave (8 waves x 3 x 16 = 384 columns)

// One (plane, n-tile) unit with v_mfma_f32_16x16x4_f32: 16 k-steps; lane (g = lane>>4,
// r = lane&15) builds B[k = g][col = r] = (X . A_k)[channel 16g + s][column] from L
// gathered LDS values and feeds the four 16-row output tiles.  The X tile is stored with
// rows interleaved in groups of four (float4 per (row group, column)), so one ds_read_b128
// serves four consecutive k-steps; gathers of the next chunk are issued before the FMAs /
// MFMAs of the current one.  Long neighbour lists are processed in two halves to bound the
// number of live gather registers.
template <int L>
__device__ __forceinline__ void agg_mfma16(const float4 *__restrict__ xg4, const int (&off)[GC_MAXL],
                                           const float (&cf)[GC_MAXL], const float (&a)[4][16],
                                           floatx4_t (&acc)[4]) {
  constexpr int H = L > 6 ? 2 : 1;            // list halves
  constexpr int LC = (L + H - 1) / H;         // entries per half
  float4 xc[LC], xn[LC];
#pragma unroll
  for (int j = 0; j < LC; ++j) xc[j] = xg4[off[j]];
  float b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ch = 0; ch < 4 * H; ++ch) {        // chunk = (row quad qd, list half hf)
    const int qd = ch / H, hf = ch % H;
    if (ch + 1 < 4 * H) {
      const int nqd = (ch + 1) / H, nhf = (ch + 1) % H;
#pragma unroll
      for (int j = 0; j < LC; ++j)
        if (nhf * LC + j < L) xn[j] = xg4[nqd * GC_ROW4 + off[nhf * LC + j]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < LC; ++j)
      if (hf * LC + j < L) {
        const float c = cf[hf * LC + j];
        b[0] = fmaf(c, xc[j].x, b[0]); b[1] = fmaf(c, xc[j].y, b[1]);
        b[2] = fmaf(c, xc[j].z, b[2]); b[3] = fmaf(c, xc[j].w, b[3]);
      }
    if (hf == H - 1) {
      __builtin_amdgcn_s_setprio(1);       // the wave that has its B operands ready goes first: the SIMD's other wave
                                           // is then in its gather phase, which fits under these MFMAs
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][4 * qd + t], b[t], acc[m], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      b[0] = b[1] = b[2] = b[3] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < LC; ++j) xc[j] = xn[j];
  }
}

__global__ __launch_bounds__(GC_THREADS, 2) void gcn_fused_kernel(
    GcnParams p, int ltot, const float *__restrict__ x, const float *__restrict__ W,
    const uint8_t *__restrict__ nbr, const float *__restrict__ coef,
    const float *__restrict__ bias_cv, float *__restrict__ z, float *__restrict__ stats_partial) {
  extern __shared__ float xs[];                       // [16 row groups][GC_ROW4] float4, then the int2 table
  float4 *xs4 = reinterpret_cast<float4 *>(xs);
  int2 *tbl = reinterpret_cast<int2 *>(xs + 16 * GC_ROW4 * 4);
  float *rowstat = reinterpret_cast<float *>(tbl + ltot * p.V);   // [8 waves][64][2] per-row (sum, sum of squares) of the output tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int g = lane >> 4;
  const int r = lane & 15;

  const int seq = blockIdx.x / p.tiles_per_seq;
  const int tile = blockIdx.x % p.tiles_per_seq;
  const int t0 = tile * p.F;
  const int frames = min(p.F, p.T - t0);
  const int ncols = frames * p.V;
  const size_t row_stride = (size_t)p.T * p.V;
  const float *xgm = x + (size_t)seq * GC_C * row_stride + (size_t)t0 * p.V;
  float *zg = z + (size_t)seq * GC_C * row_stride + (size_t)t0 * p.V;

  int colv[GC_NT16], fbase[GC_NT16], wj[GC_NT16];
  bool valid[GC_NT16];
#pragma unroll
  for (int i = 0; i < GC_NT16; ++i) {
    const int col = (wave * GC_NT16 + i) * 16 + r;
    colv[i] = col;
    valid[i] = col < ncols;
    const int f = valid[i] ? col / p.V : 0;
    wj[i] = valid[i] ? col - f * p.V : 0;
    fbase[i] = f * p.V;
  }

  // accumulators start from the bias table (64 x V, L2 resident): these loads land while the tile is staged
  floatx4_t acc[GC_NT16][4];
#pragma unroll
  for (int i = 0; i < GC_NT16; ++i)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[i][m][q] = bias_cv ? bias_cv[(16 * m + 4 * g + q) * p.V + wj[i]] : 0.f;

  // ---- stage the X tile (64 rows of `ncols` contiguous floats) and the (nbr, coef) table
#pragma unroll 1
  for (int rg = wave; rg < GC_C / 4; rg += GC_THREADS / 64) {   // row group = 4 consecutive rows
    float v[4][GC_NP / 64];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float *src = xgm + (size_t)(4 * rg + h) * row_stride;
#pragma unroll
      for (int i = 0; i < GC_NP / 64; ++i) {
        const int q = 64 * i + lane;
        v[h][i] = q < ncols ? src[q] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < GC_NP / 64; ++i)
      xs4[rg * GC_ROW4 + 64 * i + lane] = make_float4(v[0][i], v[1][i], v[2][i], v[3][i]);
  }
  for (int e = tid; e < ltot * p.V; e += GC_THREADS)
    tbl[e] = make_int2((int)nbr[e], __float_as_int(coef[e]));
  __syncthreads();

  const float4 *xg = xs4 + 4 * g * GC_ROW4;   // this lane group's 16 input channels = 4 row groups

  for (int k = 0; k < p.K; ++k) {
    // A operands: W_k[row 16m + r][channels 16g .. 16g+15]
    float a[4][16];
    const int kw = k;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float4 *wp = reinterpret_cast<const float4 *>(W + ((size_t)kw * GC_C + 16 * m + r) * GC_C + 16 * g);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 u = wp[q];
        a[m][4 * q + 0] = u.x; a[m][4 * q + 1] = u.y; a[m][4 * q + 2] = u.z; a[m][4 * q + 3] = u.w;
      }
    }
    const int Lp = p.Lk[k];
    const int lofs = p.Lofs[k];
#pragma unroll
    for (int i = 0; i < GC_NT16; ++i) {
      int off[GC_MAXL];
      float cf[GC_MAXL];
      int L = 0;                         // longest list among this n-tile's 16 columns (padded slots have coef 0)
#pragma unroll
      for (int j = 0; j < GC_MAXL; ++j) {
        const bool on = j < Lp;
        const int2 e = tbl[(on ? lofs + j : lofs) * p.V + wj[i]];
        off[j] = fbase[i] + e.x;
        cf[j] = (on && valid[i]) ? __int_as_float(e.y) : 0.f;
        if (__ballot(cf[j] != 0.f) != 0ull) L = j + 1;
      }
      if (L == 0) continue;              // plane k does not reach these columns
      switch (L) {
        case 1: agg_mfma16<1>(xg, off, cf, a, acc[i]); break;
        case 2:
        case 3: agg_mfma16<3>(xg, off, cf, a, acc[i]); break;
        case 4:
        case 5: agg_mfma16<5>(xg, off, cf, a, acc[i]); break;
        case 6:
        case 7:
        case 8: agg_mfma16<8>(xg, off, cf, a, acc[i]); break;
        default: agg_mfma16<12>(xg, off, cf, a, acc[i]); break;
      }
    }
  }

  // ---- epilogue: 16x16 tile D[row = 4g + q][col = r]
  float s1[4][4], s2[4][4];            // this lane's share of the per-row output statistics
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) s1[m][q] = s2[m][q] = 0.f;
#pragma unroll
  for (int i = 0; i < GC_NT16; ++i) {
    if (!valid[i]) continue;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * m + 4 * g + q;
        const float v = acc[i][m][q];
        zg[(size_t)row * row_stride + colv[i]] = v;
        s1[m][q] += v;
        s2[m][q] = fmaf(v, v, s2[m][q]);
      }
    }
  }
  // The BatchNorm that follows (tcn.0, stgcn_layers.py:400) needs sum / sum of squares per channel of exactly
  // this output: reduce them here instead of re-reading z -- 16 lanes (DPP), then the 8 waves (LDS atomics).
  if (stats_partial) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a = p2r_row16_sum(s1[m][q]), b = p2r_row16_sum(s2[m][q]);
        if (r == 0)     // one slot per wave: plain stores, no contention
          *reinterpret_cast<float2 *>(rowstat + (wave * GC_C + 16 * m + 4 * g + q) * 2) = make_float2(a, b);
      }
    __syncthreads();
    if (tid < 2 * GC_C) {
      float t = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < GC_THREADS / 64; ++w8) t += rowstat[w8 * 2 * GC_C + tid];
      stats_partial[(size_t)blockIdx.x * 2 * GC_C + tid] = t;
    }
  }
}

}  // namespace

// x (N,64,T,V) -> z (N,64,T,V).  W [K][64][64] (row = output channel); nbr u8 /
// coef f32 tables [sum_k L_k][V] (column-wise neighbour lists of plane k, padded
// with coef 0); Lk[K] on the host; bias_cv [64][V] or NULL.  stats_partial (optional)
// [N * ceil(T / F)][64][2], F = 384 / V: per-workgroup (sum, sum of squares) of z per channel.
extern "C" int p2r_stgcn_gcn_forward(int N, int T, int V, int K, const int *Lk_host, const float *x,
                                     const float *W, const uint8_t *nbr, const float *coef,
                                     const float *bias_cv, float *z, float *stats_partial, void *stream) {
  if (N < 0 || T <= 0 || V <= 0 || V > 128 || K <= 0 || K > GC_MAXK) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  GcnParams p;
  p.T = T; p.V = V; p.K = K;
  p.F = GC_NP / V;
  if (p.F < 1) return P2R_EINVAL;
  if (p.F > T) p.F = T;
  p.tiles_per_seq = p2r_cdiv(T, p.F);
  int ofs = 0;
  for (int k = 0; k < K; ++k) {
    if (Lk_host[k] < 1 || Lk_host[k] > GC_MAXL) return P2R_EINVAL;
    p.Lk[k] = Lk_host[k];
    p.Lofs[k] = ofs;
    ofs += Lk_host[k];
  }
  const long long blocks = (long long)N * p.tiles_per_seq;
  if (blocks > 0x7fffffffLL) return P2R_EINVAL;
  const size_t lds = (size_t)16 * GC_ROW4 * sizeof(float4) + (size_t)ofs * V * sizeof(int2) + (GC_THREADS / 64) * 2 * GC_C * sizeof(float);
  if (lds > 160 * 1024) return P2R_EINVAL;
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  {
    hipError_t e = p2r_allow_big_lds(gcn_fused_kernel, lds_ok);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(gcn_fused_kernel, dim3((unsigned)blocks), dim3(GC_THREADS), lds,
                     p2r_stream(stream), p, ofs, x, W, nbr, coef, bias_cv, z, stats_partial);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// =============================================================================
// Weight gradient:  dW_k[c][ci] = sum over (n, t, w) of dZ[c, t, w] * (X . A_k)[ci, t, w]
// (reference: autograd through stgcn_layers.py:62-65).
//
// GEMM view per plane k: M = 64 (c), N = 64 (ci), reduction over every column of
// the batch.  A persistent grid of workgroups walks the (sequence, 4-frame) tiles;
// dZ and X tiles are staged in LDS (row length odd: conflict-free column reads);
// reduction steps are ordered (joint w major, frame minor) so the 4 columns of one
// v_mfma_f32_16x16x4_f32 step are the 4 frames of ONE joint: the neighbour list of
// (k, w) is then wave-uniform (scalar loads), every (ci, column) aggregate is built
// exactly once, and it feeds the 4 output-row tiles.  A wave keeps the accumulator
// tiles of its planes (12 / DW_SETS planes x 64 rows x 16 ci columns) in registers across all
// its tiles and writes one partial per workgroup; partials are summed deterministically
// by the caller.
// =============================================================================
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int DW_F = 4;
constexpr int DW_MAXV = 64;
#define DW_SETS 4   // 16 waves = four per SIMD at 128 VGPRs: measured 1.45 ms vs 1.58 (12 waves) and 1.68 (8 waves)
constexpr int DW_PL = 12 / DW_SETS;   // planes per wave set (the sets cover up to 12 planes)
constexpr int DW_THREADS = 64 * 4 * DW_SETS;
constexpr int DW_NW = DW_THREADS / 64;
constexpr int DW_CS = (GC_C * DW_MAXV + DW_THREADS - 1) / DW_THREADS;   // (channel, joint) column sums owned per thread

__host__ __device__ constexpr int dw_row_len(int V) {   // row stride == 2 (mod 32): conflict-free column reads
  int r = DW_F * V;
  while (r % 32 != 2) ++r;
  return r;
}

struct DwSets {                   // host-balanced split of the planes over the DW_SETS groups of four waves
  int plane[DW_SETS][DW_PL];      // plane id or -1
};

// b[f] += sum_j coef * X[ci row][frame f, joint nbr_j]  for one plane (accumulates into b).
// L = compile-time list capacity (a few sizes only, to bound code size), Lr = real length.
template <int L>
__device__ __forceinline__ void dw_plane(const int2 *__restrict__ trow, int tstride,
                                         const float *__restrict__ xrow, int V, bool live, int Lr,
                                         float (&b)[DW_F]) {
  int2 e[L];
#pragma unroll
  for (int j = 0; j < L; ++j) {
    e[j] = trow[(j < Lr ? j : 0) * tstride];
    if (j >= Lr) e[j].y = 0;                     // coefficient 0 for the padding slots
  }
  float xv[L][DW_F];
#pragma unroll
  for (int j = 0; j < L; ++j)
#pragma unroll
    for (int f = 0; f < DW_F; ++f) xv[j][f] = xrow[f * V + e[j].x];
#pragma unroll
  for (int j = 0; j < L; ++j) {
    const float cf = live ? __int_as_float(e[j].y) : 0.f;
#pragma unroll
    for (int f = 0; f < DW_F; ++f) b[f] = fmaf(cf, xv[j][f], b[f]);
  }
}

template <int VS>
__global__ __launch_bounds__(DW_THREADS, DW_SETS == 2 ? 2 : 1) void gcn_dw_kernel(GcnParams p, DwSets sets, int n_seq,
                                                               int row_len, int ltot,
                                                               const float *__restrict__ x,
                                                               const float *__restrict__ dz,
                                                               const uint8_t *__restrict__ nbr,
                                                               const float *__restrict__ coef,
                                                               float *__restrict__ dw_partial,
                                                               float *__restrict__ colsum_partial,
                                                               int colsum_of_x) {
  extern __shared__ float lds[];
  const int V = VS > 0 ? VS : p.V;                    // compile-time joint count: immediate LDS offsets
  const int RL = VS > 0 ? dw_row_len(VS) : row_len;
  float *dzs = lds;                                   // [64][RL]
  float *xs = lds + GC_C * RL;                   // [64][RL]
  int2 *tbl = reinterpret_cast<int2 *>(lds + 2 * GC_C * RL);   // [ltot][V] (nbr, coef)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int g = lane >> 4;                            // joint within the group of 4 / MFMA k index
  const int r = lane & 15;
  const int nt = wave & 3;                            // 16 ci columns owned by this wave
  const int half = wave >> 2;                         // which plane set

  for (int e = tid; e < ltot * V; e += DW_THREADS)
    tbl[e] = make_int2((int)nbr[e], __float_as_int(coef[e]));

  const int tiles_per_seq = (p.T + DW_F - 1) / DW_F;
  const int total_tiles = n_seq * tiles_per_seq;
  const size_t row_stride = (size_t)p.T * V;
  const int n_groups = (V + 3) / 4;

  // longest real list (non-zero coefficient) among the four joints of each (plane, joint group): the
  // gathers of one MFMA step only need that many slots, and a group a plane does not reach is skipped
  int *glen = reinterpret_cast<int *>(tbl + ltot * V);             // [K][DW_MAXV / 4]
  float *cs = reinterpret_cast<float *>(glen + p.K * (DW_MAXV / 4)); // [64][V] running column sums of dZ
  for (int e = tid; e < GC_C * V; e += DW_THREADS) cs[e] = 0.f;
  __syncthreads();
  for (int e = tid; e < p.K * n_groups; e += DW_THREADS) {
    const int k = e / n_groups, wg = e - k * n_groups;
    int m = 0;
    for (int j = 0; j < p.Lk[k]; ++j)
      for (int w = 4 * wg; w < min(4 * wg + 4, V); ++w)
        if (__int_as_float(tbl[(p.Lofs[k] + j) * V + w].y) != 0.f) m = j + 1;
    glen[k * (DW_MAXV / 4) + wg] = m;
  }

  floatx4 acc[DW_PL][4];
  int pl_k[DW_PL], pl_row[DW_PL];   // this wave's planes, hoisted out of the loops (SGPRs)
#pragma unroll
  for (int kk = 0; kk < DW_PL; ++kk) {
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[kk][m] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int k = sets.plane[half][kk];
    pl_k[kk] = k;
    pl_row[kk] = k >= 0 ? p.Lofs[k] * V : 0;
  }

  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int seq = tile / tiles_per_seq;
    const int t0 = (tile % tiles_per_seq) * DW_F;
    const int ncols = min(DW_F, p.T - t0) * V;
    const float *xg = x + (size_t)seq * GC_C * row_stride + (size_t)t0 * V;
    const float *dg = dz + (size_t)seq * GC_C * row_stride + (size_t)t0 * V;
    // stage both tiles: all loads of two rows are issued before the first LDS write so the HBM latency is paid
    // once per row pair, not once per 64-column chunk; the first row pair is requested BEFORE the barrier that
    // waits for the other waves to finish the previous tile, so its latency overlaps that wait
    {
      const int c = wave;
      float vx[2][4], vd[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float *sx = xg + (size_t)(c + DW_NW * h) * row_stride;
        const float *sd = dg + (size_t)(c + DW_NW * h) * row_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = lane + 64 * i;
          const bool in = q < ncols && (c + DW_NW * h) < GC_C;
          vx[h][i] = in ? sx[q] : 0.f;
          vd[h][i] = in ? sd[q] : 0.f;
        }
      }
      __syncthreads();  // previous tile fully consumed (and the table written, first time)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = lane + 64 * i;
          if (q < RL && (c + DW_NW * h) < GC_C) {
            xs[(c + DW_NW * h) * RL + q] = vx[h][i];
            dzs[(c + DW_NW * h) * RL + q] = vd[h][i];
          }
        }
    }
#pragma unroll 1
    for (int c = wave + 2 * (DW_THREADS / 64); c < GC_C; c += 2 * (DW_THREADS / 64)) {
      float vx[2][4], vd[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float *sx = xg + (size_t)(c + DW_NW * h) * row_stride;
        const float *sd = dg + (size_t)(c + DW_NW * h) * row_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = lane + 64 * i;
          const bool in = q < ncols && (c + DW_NW * h) < GC_C;
          vx[h][i] = in ? sx[q] : 0.f;
          vd[h][i] = in ? sd[q] : 0.f;
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = lane + 64 * i;
          if (q < RL && (c + DW_NW * h) < GC_C) {
            xs[(c + DW_NW * h) * RL + q] = vx[h][i];
            dzs[(c + DW_NW * h) * RL + q] = vd[h][i];
          }
        }
    }
    __syncthreads();

    if (colsum_partial) {   // sum of dZ over the tile's frames per (channel, joint): the bias-table gradient
#pragma unroll 1
      for (int n = 0; n < DW_CS; ++n) {
        const int idx = tid + DW_THREADS * n;
        if (idx < GC_C * V) {
          const int c = idx / V, w = idx - c * V;
          const float *dp = (colsum_of_x ? xs : dzs) + c * RL + w;
          float sum = 0.f;
#pragma unroll
          for (int f = 0; f < DW_F; ++f) sum += dp[f * V];      // frames past the sequence end are zero-filled
          cs[idx] += sum;                                         // element owned by this thread: no atomics
        }
      }
    }

    const float *xrow = xs + (16 * nt + r) * RL;   // this lane's ci row
    const float *drow = dzs + r * RL;               // + 16*m rows
    for (int wg = 0; wg < n_groups; ++wg) {
      const int w = 4 * wg + g;
      const int *gl_w = glen + wg;
      const bool live = w < V;
      const int wc = live ? w : 0;
      float a[4][DW_F];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int f = 0; f < DW_F; ++f) a[m][f] = live ? drow[16 * m * RL + f * V + wc] : 0.f;
#pragma unroll
      for (int kk = 0; kk < DW_PL; ++kk) {
        if (pl_k[kk] < 0) continue;                      // uniform
        float b[DW_F];
        const int2 *trow = tbl + pl_row[kk] + wc;
        const int Lr = __builtin_amdgcn_readfirstlane(gl_w[pl_k[kk] * (DW_MAXV / 4)]);
        if (Lr == 0) continue;                           // all-zero aggregate: nothing to accumulate
#pragma unroll
        for (int f = 0; f < DW_F; ++f) b[f] = 0.f;
        // lists longer than six go in two passes: bounds the live gather registers (no scratch)
        // tighter register budget: passes of at most three list entries
        for (int j0 = 0; j0 < Lr; j0 += 3) {
          const int rem = Lr - j0;
          if (rem <= 1) dw_plane<1>(trow + j0 * V, V, xrow, V, live, rem, b);
          else dw_plane<3>(trow + j0 * V, V, xrow, V, live, rem < 3 ? rem : 3, b);
        }
#pragma unroll
        for (int f = 0; f < DW_F; ++f)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[kk][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][f], b[f], acc[kk][m], 0, 0, 0);

      }
    }
  }

  if (colsum_partial) {
#pragma unroll
    for (int n = 0; n < DW_CS; ++n) {
      const int idx = tid + DW_THREADS * n;
      if (idx < GC_C * V) colsum_partial[(size_t)blockIdx.x * GC_C * V + idx] = cs[idx];
    }
  }
  // partial[block][k][c][ci]: D[row = 4*g + q][col = r] -> c = 16*m + row, ci = 16*nt + r
  float *out = dw_partial + (size_t)blockIdx.x * p.K * GC_C * GC_C;
#pragma unroll
  for (int kk = 0; kk < DW_PL; ++kk) {
    const int k = sets.plane[half][kk];
    if (k < 0) continue;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        out[((size_t)k * GC_C + 16 * m + 4 * g + q) * GC_C + 16 * nt + r] = acc[kk][m][q];
  }
}

}  // namespace

// =============================================================================
// Adjacency gradient at the non-zero entries:
//   dcoef[k][j][w] = sum over (n, t, ci) of X[ci, t, v_j(k,w)] * (W_k^T dZ)[ci, t, w]
// (the gradient reaching `A * importance`, stgcn.py:134, needed for edge_importance).
// Dense part H_k = W_k^T . dZ on MFMA exactly like the forward kernel, but per plane
// (no accumulation over k); the dZ B-operands are loaded once per tile into VGPRs
// and reused by all planes, the X tile sits in LDS for the gathers, and the per-lane
// partial products are reduced with LDS float atomics into a [sum L_k][V] table that
// each persistent workgroup writes out once.
// =============================================================================
namespace {

constexpr int DC_THREADS = 512;
// X tile layout for this kernel: rows interleaved in groups of four, xs4[row >> 2][col] = float4 of
// rows 4q..4q+3 -- the four accumulator rows a lane holds per 16x16 tile are consecutive, so one
// ds_read_b128 feeds four FMAs (4x fewer LDS instructions than a row-major tile).
constexpr int DC_ROW4 = GC_NP + 1;  // float4 elements per row group (odd: spreads the row groups over banks)

// Per-lane partial of dcoef for one (plane, n-tile): sum over this lane's 16 rows of
// H[row][col] * X[row][(frame, neighbour_j)], for each of the L neighbours; reduced over
// the four lane groups with LDS float atomics.
// v summed over the four 16-lane rows of the wave (every lane gets the total):
// v_permlane32_swap pairs rows {0,1} with {2,3}, v_permlane16_swap pairs odd with even rows.
__device__ __forceinline__ float rows4_sum(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int L>
__device__ __forceinline__ void dc_reduce(const floatx4_t (&h)[4], const float4 *__restrict__ xs4, int g,
                                          const int2 *__restrict__ trow, int V, int fbase,
                                          float *__restrict__ dcs_row, int Lr) {
  int nb[L];
#pragma unroll
  for (int j = 0; j < L; ++j) nb[j] = trow[(j < Lr ? j : 0) * V].x;   // L = capacity, Lr = real length
  float part[L];
#pragma unroll
  for (int j = 0; j < L; ++j) part[j] = 0.f;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float4 *xr = xs4 + (4 * m + g) * DC_ROW4 + fbase;     // rows 16m + 4g .. +3
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const float4 xv = xr[nb[j]];
      part[j] = fmaf(h[m][0], xv.x, part[j]);
      part[j] = fmaf(h[m][1], xv.y, part[j]);
      part[j] = fmaf(h[m][2], xv.z, part[j]);
      part[j] = fmaf(h[m][3], xv.w, part[j]);
    }
  }
  // sum over the four lane groups in the VALU (gfx950 lane swaps), then one LDS atomic per column
#pragma unroll
  for (int j = 0; j < L; ++j) part[j] = rows4_sum(part[j]);
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < L; ++j)
      if (j < Lr) atomicAdd(dcs_row + j * V, part[j]);
  }
}

__global__ __launch_bounds__(DC_THREADS, 2) void gcn_dcoef_kernel(GcnParams p, int n_seq, int ltot,
                                                                  const float *__restrict__ x,
                                                                  const float *__restrict__ dz,
                                                                  const float *__restrict__ Wt,
                                                                  const uint8_t *__restrict__ nbr,
                                                                  const float *__restrict__ coef,
                                                                  float *__restrict__ dcoef_partial) {
  extern __shared__ float lds[];
  float4 *xs4 = reinterpret_cast<float4 *>(lds);                     // [16 row groups][DC_ROW4] float4
  int2 *tbl = reinterpret_cast<int2 *>(lds + 16 * DC_ROW4 * 4);      // [ltot][V] (nbr, unused)
  float *dcs = reinterpret_cast<float *>(tbl + ltot * p.V);          // [ltot][V]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int g = lane >> 4;
  const int r = lane & 15;

  for (int e = tid; e < ltot * p.V; e += DC_THREADS) {
    // .y = 1 marks a real list slot: non-zero coefficient when the coefficient table is given, otherwise slot 0 and
    // every later slot with a non-zero joint (lists are sorted and padded with joint 0)
    const int nb = (int)nbr[e];
    tbl[e] = make_int2(nb, coef ? (coef[e] != 0.f ? 1 : 0) : (nb != 0 ? 1 : 2));
    dcs[e] = 0.f;
  }
  __syncthreads();
  // A tile's columns keep their joint across the persistent loop (column = frame * V + joint, tiles start at
  // whole frames), so the longest real list among a wave's 16 columns is fixed per (n-tile, plane); with the
  // coefficient table a plane that reaches none of the 16 columns gets length 0 and its MFMAs are skipped.
  // Columns of a tile are taken JOINT-major (column c of the MFMA n-tiles = frame c % F of joint c / F, F = frames of
  // a full tile): the 16 columns of an n-tile then span 3-4 joints instead of 16, and 33 % of the (plane, n-tile)
  // units of the row lists are empty instead of 19 % (177 vs 213 units of 264 per tile).
  //
  // The n-tiles are dealt to the waves by cost (number of planes that reach the tile), not in order: with the
  // P2RNet graph the first 8 of the 24 tiles carry 8-11 planes each and the rest 6, so waves 0-2 would own
  // 27-33 (plane, n-tile) units against 18 for the others; dealt greedily (largest first, to the least loaded
  // wave) every wave owns 20-23.
  constexpr int DC_NTT = (DC_THREADS / 64) * GC_NT16;
  int *s_cost = reinterpret_cast<int *>(dcs + ltot * p.V), *s_tile = s_cost + DC_NTT;   // [DC_NTT] each
  for (int t = wave; t < DC_NTT; t += DC_THREADS / 64) {
    const int col = t * 16 + r;
    const bool in = col < p.F * p.V;
    const int w = in ? col / p.F : 0;
    int c = 0;
    for (int k = 0; k < p.K; ++k) {
      bool any = false;
      for (int j = 0; j < p.Lk[k]; ++j) {
        const int mark = tbl[(p.Lofs[k] + j) * p.V + w].y;
        if (in && (mark == 1 || (mark == 2 && j == 0))) any = true;
      }
      if (__ballot(any) != 0ull) ++c;
    }
    if (lane == 0) s_cost[t] = c;
  }
  __syncthreads();
  if (tid == 0) {
    int load[DC_THREADS / 64], cnt[DC_THREADS / 64];
    for (int w = 0; w < DC_THREADS / 64; ++w) load[w] = cnt[w] = 0;
    for (int n = 0; n < DC_NTT; ++n) {
      int best = -1;
      for (int t = 0; t < DC_NTT; ++t)
        if (s_cost[t] >= 0 && (best < 0 || s_cost[t] > s_cost[best])) best = t;
      int to = -1;
      for (int w = 0; w < DC_THREADS / 64; ++w)
        if (cnt[w] < GC_NT16 && (to < 0 || load[w] < load[to])) to = w;
      s_tile[to * GC_NT16 + cnt[to]++] = best;
      load[to] += s_cost[best] + 1;
      s_cost[best] = -1;
    }
  }
  __syncthreads();
  int ntile[GC_NT16];
#pragma unroll
  for (int i = 0; i < GC_NT16; ++i) ntile[i] = __builtin_amdgcn_readfirstlane(s_tile[wave * GC_NT16 + i]);

  int tlen[GC_NT16];                       // plane k's value lives in lane k of the wave
#pragma unroll
  for (int i = 0; i < GC_NT16; ++i) {
    const int col = ntile[i] * 16 + r;
    const bool in = col < p.F * p.V;
    const int w = in ? col / p.F : 0;
    int mine = 0;
    for (int k = 0; k < p.K; ++k) {
      int len = 0;
      for (int j = 0; j < p.Lk[k]; ++j) {
        const int mark = tbl[(p.Lofs[k] + j) * p.V + w].y;       // 1 real, 0 padding, 2 "joint 0 without coefficients"
        if (in && (mark == 1 || (mark == 2 && j == 0))) len = j + 1;
      }
      // wave maximum by ballots over the candidate lengths
      int m = 0;
      for (int c = 1; c <= p.Lk[k]; ++c)
        if (__ballot(len >= c) != 0ull) m = c;
      if (lane == k) mine = m;
    }
    tlen[i] = mine;
  }

  const int total_tiles = n_seq * p.tiles_per_seq;
  const size_t row_stride = (size_t)p.T * p.V;

  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int seq = tile / p.tiles_per_seq;
    const int t0 = (tile % p.tiles_per_seq) * p.F;
    const int ncols = min(p.F, p.T - t0) * p.V;
    const float *xgm = x + (size_t)seq * GC_C * row_stride + (size_t)t0 * p.V;
    const float *dg = dz + (size_t)seq * GC_C * row_stride + (size_t)t0 * p.V;
    // the B operands of this tile are requested first: their HBM latency then overlaps the wait for the other
    // waves (barrier) and the staging of the gather tile
    int fbase[GC_NT16], wj[GC_NT16];
    bool valid[GC_NT16];
    float bz[GC_NT16][16];               // dZ[c = 16g + s][col]: B operands, reused by every plane
#pragma unroll
    for (int i = 0; i < GC_NT16; ++i) {
      const int col = ntile[i] * 16 + r;
      const int jn = col / p.F, f = col - jn * p.F;          // joint-major column -> (joint, frame)
      valid[i] = jn < p.V && f * p.V < ncols;
      wj[i] = valid[i] ? jn : 0;
      fbase[i] = valid[i] ? f * p.V : 0;
#pragma unroll
      for (int s = 0; s < 16; ++s)
        bz[i][s] = valid[i] ? dg[(size_t)(16 * g + s) * row_stride + fbase[i] + wj[i]] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int rg = wave; rg < GC_C / 4; rg += DC_THREADS / 64) {       // row group = 4 consecutive rows
      float v[4][GC_NP / 64];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float *src = xgm + (size_t)(4 * rg + h) * row_stride;
#pragma unroll
        for (int i = 0; i < GC_NP / 64; ++i) {
          const int q = 64 * i + lane;
          v[h][i] = q < ncols ? src[q] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < GC_NP / 64; ++i)
        xs4[rg * DC_ROW4 + 64 * i + lane] = make_float4(v[0][i], v[1][i], v[2][i], v[3][i]);
    }

    __syncthreads();

    for (int k = 0; k < p.K; ++k) {
      float a[4][16];                     // Wt[k][ci = 16m + r][c = 16g .. 16g+15]
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float4 *wp = reinterpret_cast<const float4 *>(Wt + ((size_t)k * GC_C + 16 * m + r) * GC_C + 16 * g);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 u = wp[q];
          a[m][4 * q + 0] = u.x; a[m][4 * q + 1] = u.y; a[m][4 * q + 2] = u.z; a[m][4 * q + 3] = u.w;
        }
      }
      const int lofs = p.Lofs[k];
#pragma unroll
      for (int i = 0; i < GC_NT16; ++i) {
        const int L = __builtin_amdgcn_readlane(tlen[i], k);   // longest real list in this n-tile
        if (L == 0) continue;                                  // plane k reaches none of these columns
        floatx4_t h[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) h[m] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            h[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], bz[i][s], h[m], 0, 0, 0);
        // h[m][q] = H_k[ci = 16m + 4g + q][col_i]
        if (valid[i]) {
          const int2 *trow = tbl + lofs * p.V + wj[i];
          float *drow = dcs + lofs * p.V + wj[i];
          // exact capacities for the common lengths: a padded slot costs 4 LDS reads + 16 FMAs + a lane reduction,
          // and VALU time is MFMA time lost on gfx950
          switch (L) {
            case 1: dc_reduce<1>(h, xs4, g, trow, p.V, fbase[i], drow, L); break;
            case 2: dc_reduce<2>(h, xs4, g, trow, p.V, fbase[i], drow, L); break;
            case 3: dc_reduce<3>(h, xs4, g, trow, p.V, fbase[i], drow, L); break;
            case 4: dc_reduce<4>(h, xs4, g, trow, p.V, fbase[i], drow, L); break;
            case 5: dc_reduce<5>(h, xs4, g, trow, p.V, fbase[i], drow, L); break;
            case 6: dc_reduce<6>(h, xs4, g, trow, p.V, fbase[i], drow, L); break;
            case 7:
            case 8: dc_reduce<8>(h, xs4, g, trow, p.V, fbase[i], drow, L); break;
            default: dc_reduce<12>(h, xs4, g, trow, p.V, fbase[i], drow, L); break;
          }
        }
      }
    }
  }
  __syncthreads();
  float *out = dcoef_partial + (size_t)blockIdx.x * ltot * p.V;
  for (int q = tid; q < ltot * p.V; q += DC_THREADS) out[q] = dcs[q];
}

}  // namespace

static int gcn_fill_params(GcnParams &p, int T, int V, int K, const int *Lk_host, int F) {
  if (T <= 0 || V <= 0 || V > DW_MAXV || K <= 0 || K > GC_MAXK) return P2R_EINVAL;
  p.T = T; p.V = V; p.K = K; p.F = F < T ? F : T;
  if (p.F < 1) return P2R_EINVAL;
  p.tiles_per_seq = p2r_cdiv(T, p.F);
  int ofs = 0;
  for (int k = 0; k < K; ++k) {
    if (Lk_host[k] < 1 || Lk_host[k] > GC_MAXL) return P2R_EINVAL;
    p.Lk[k] = Lk_host[k];
    p.Lofs[k] = ofs;
    ofs += Lk_host[k];
  }
  return ofs;
}

// dW partials: x, dz (N,64,T,V) -> dw_partial [n_blocks][K][64][64] = sum over columns of
// dz[row][col] * (x . lists)[column-of-partial][col]; the caller sums over the leading axis.
// colsum_partial (optional) [n_blocks][64][V] = per-workgroup sums over samples and frames of the
// `dz` argument (or of the `x` argument when colsum_of_x != 0), also summed by the caller.  Returns the number of workgroups used through *n_blocks
// when dw_partial is NULL (size query), else launches.  K must be 11.
extern "C" int p2r_stgcn_gcn_weight_grad(int N, int T, int V, int K, const int *Lk_host, const float *x,
                                         const float *dz, const uint8_t *nbr, const float *coef,
                                         int n_blocks, float *dw_partial, float *colsum_partial, int colsum_of_x,
                                         void *stream) {
  GcnParams p;
  const int ltot = gcn_fill_params(p, T, V, K, Lk_host, DW_F);
  if (ltot < 0) return ltot;
  if (K > DW_SETS * DW_PL || N < 0 || n_blocks < 1) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  // DW_SETS plane sets with balanced gather work (longest lists first, greedy)
  DwSets sets;
  int cnt[DW_SETS] = {0}, load[DW_SETS] = {0}, order[GC_MAXK];
  for (int k = 0; k < K; ++k) order[k] = k;
  for (int i = 0; i < K; ++i)
    for (int j = i + 1; j < K; ++j)
      if (p.Lk[order[j]] > p.Lk[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
  for (int h = 0; h < DW_SETS; ++h)
    for (int i = 0; i < DW_PL; ++i) sets.plane[h][i] = -1;
  for (int i = 0; i < K; ++i) {
    int h = -1;
    for (int c = 0; c < DW_SETS; ++c)
      if (cnt[c] < DW_PL && (h < 0 || load[c] < load[h])) h = c;
    sets.plane[h][cnt[h]++] = order[i];
    load[h] += p.Lk[order[i]] + 4;   // gathers + the 16 MFMAs a plane costs per frame group
  }
  const int row_len = dw_row_len(V);
  const size_t lds = 2 * (size_t)GC_C * row_len * sizeof(float) + (size_t)ltot * V * sizeof(int2) +
                     (size_t)K * (DW_MAXV / 4) * sizeof(int) + (size_t)GC_C * V * sizeof(float);
  if (lds > 160 * 1024 || row_len > 256) return P2R_EINVAL;
  if (V == 53) {   // the P2RNet skeleton
    static unsigned char lds_ok[P2R_MAX_DEVICES];
    hipError_t e = p2r_allow_big_lds(gcn_dw_kernel<53>, lds_ok);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(gcn_dw_kernel<53>, dim3(n_blocks), dim3(DW_THREADS), lds, p2r_stream(stream), p, sets, N,
                       row_len, ltot, x, dz, nbr, coef, dw_partial, colsum_partial, colsum_of_x);
  } else {
    static unsigned char lds_ok[P2R_MAX_DEVICES];
    hipError_t e = p2r_allow_big_lds(gcn_dw_kernel<0>, lds_ok);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(gcn_dw_kernel<0>, dim3(n_blocks), dim3(DW_THREADS), lds, p2r_stream(stream), p, sets, N,
                       row_len, ltot, x, dz, nbr, coef, dw_partial, colsum_partial, colsum_of_x);
  }
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// dcoef partials: x, dz (N,64,T,V), Wt [K][64(ci)][64(c)] (transposed planes) ->
// dcoef_partial [n_blocks][sum L_k][V]; the caller sums over the leading axis.
extern "C" int p2r_stgcn_gcn_coef_grad(int N, int T, int V, int K, const int *Lk_host, const float *x,
                                       const float *dz, const float *Wt, const uint8_t *nbr, const float *coef,
                                       int n_blocks, float *dcoef_partial, void *stream) {
  GcnParams p;
  const int ltot = gcn_fill_params(p, T, V, K, Lk_host, GC_NP / (V > 0 ? V : 1));
  if (ltot < 0) return ltot;
  if (N < 0 || n_blocks < 1) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  const size_t lds = (size_t)16 * DC_ROW4 * sizeof(float4) + (size_t)ltot * V * (sizeof(int2) + sizeof(float)) +
                     2 * (DC_THREADS / 64) * GC_NT16 * sizeof(int);
  if (lds > 160 * 1024) return P2R_EINVAL;
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  {
    hipError_t e = p2r_allow_big_lds(gcn_dcoef_kernel, lds_ok);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(gcn_dcoef_kernel, dim3(n_blocks), dim3(DC_THREADS), lds, p2r_stream(stream), p, N, ltot, x,
                     dz, Wt, nbr, coef, dcoef_partial);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
