// stgcn_gcn3_dw.hip -- weight gradient of the fused graph convolution, statically scheduled (gfx950).
//
//   dW_k[c][ci] = sum over (n, t, w) of dZ[c, t, w] * (X . A_k)[ci, t, w]
//               = sum over (n, t, v) of V_k[c, t, v] * X[ci, t, v],      V_k(v) = sum_{j} a_k(v, w_j) dZ(:, :, w_j)
// (reference: autograd through models/p2rnet/modules/stgcn_layers.py:62-65; second form = aggregation on the gradient
// side through the ROW lists, whose (plane, joint) units are empty 214 times of 583).  The operator of gcn_dw_kernel in
// stgcn_gcn.hip, rebuilt with the work list resolved at build time (tools/gen_gcn_sched.py, W3_BODY_<set>):
//
//   * GEMM view per plane: M = 64 (ci), N = 64 (c), reduction over every (frame, joint) column of the batch.  One MFMA
//     k-step = the 4 frames of a tile at ONE joint v: lane (g, r) supplies A[r][g] = X[16m + r][frame g][v] and
//     B[g][r] = V_k[16n + r][frame g][v].  The row list of the unit is then wave-uniform -- its joints are ds_read
//     immediates, empty units are skipped exactly (the first generation grouped four joints per k-step and ran 17 %
//     more MFMAs than units) -- and the aggregate of a (plane, joint) is built exactly once per wave that needs it.
//   * A wave owns a SET of planes and one HALF of the columns c (two n-tiles, whose B values are built as packed pairs
//     off one two-address LDS read per entry): 8 waves = 4 sets x 2 halves, accumulators (2-4 planes x 2 n x 4 m tiles)
//     in registers for the whole kernel, written once per workgroup.
//   * Tiles of 4 frames: X and dZ tile (54 KB each, rows of 4 x 53 floats as the tensors have them) are copied by
//     LDS-DMA in 16-byte pieces with per-lane offsets computed once; persistent workgroups.  Both tiles must be whole
//     (64 rows) at once and two more do not fit the 160 KB, so the copy of the next tile starts when the last wave has
//     finished the current one (the first generation staged through VGPRs in two dependent batches instead).
//   * The bias-table gradient (sum of dZ over samples and frames per (channel, joint)) is taken from the resident tile.
#include "p2r_common.h"

#include "gcn3_sched.inc"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int W3_F = 4;                      // frames per tile = one MFMA k-step per joint
constexpr int W3_V = G3_V;
constexpr int W3_NW = 8;
constexpr int W3_RL = W3_F * W3_V;           // 212 floats per row (848 bytes: 16-byte pieces)
constexpr int W3_TILE = 64 * W3_RL;          // floats per tensor tile
constexpr int W3_NV4 = W3_TILE / 4;          // 3392 float4 = 53 pieces of 64
constexpr int W3_PIECES = (W3_NV4 + 63) / 64;
constexpr int W3_PW = (W3_PIECES + W3_NW - 1) / W3_NW;      // 7 per wave and tensor
constexpr int W3_CS = (64 * W3_V + W3_NW * 64 - 1) / (W3_NW * 64);   // (channel, joint) sums owned per thread: 7

struct W3Params {
  int T, ltot;
  int tiles_per_seq, total_tiles;
};

constexpr int w3_wave_set[W3_NW] = W3_WAVE_SET;
constexpr int w3_wave_half[W3_NW] = W3_WAVE_HALF;
constexpr int w3_set_planes[4][W3_MAXPL] = W3_SET_PLANES;

__device__ __forceinline__ unsigned w3_lds_addr(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
__device__ __forceinline__ void w3_dma16(const float *base, unsigned voff, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(w3_lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// 8 MFMAs of one (plane, joint) unit: 4 m-tiles x 2 n-tiles, one k-step, accumulating in place (see stgcn_gcn3.hip)
__device__ __forceinline__ void w3_mfma8(f32x4 (&acc)[2][4], const float (&a)[4], const f32x2 &b) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %8, %12, %0\n\tv_mfma_f32_16x16x4_f32 %1, %9, %12, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, %10, %12, %2\n\tv_mfma_f32_16x16x4_f32 %3, %11, %12, %3\n\t"
      "v_mfma_f32_16x16x4_f32 %4, %8, %13, %4\n\tv_mfma_f32_16x16x4_f32 %5, %9, %13, %5\n\t"
      "v_mfma_f32_16x16x4_f32 %6, %10, %13, %6\n\tv_mfma_f32_16x16x4_f32 %7, %11, %13, %7"
      : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
        "+v"(acc[1][2]), "+v"(acc[1][3])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b.x), "v"(b.y));
}

// gathers of one step: dv[j] = (dZ[16 n0 + r][frame g][w_j], dZ[16 (n0 + 1) + r][frame g][w_j]); coefficients as
// broadcast LDS reads at immediate offsets of an opaque base (see stgcn_gcn3.hip)
constexpr int W3_MAXNE = 12;

template <int NE>
__device__ __forceinline__ void w3_gather_n(const char *dl, const char *cl, const int (&off)[W3_MAXNE],
                                            const int (&ci)[W3_MAXNE], f32x2 (&dv)[W3_MAXNE], float (&cf)[W3_MAXNE]) {
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    dv[j].x = *reinterpret_cast<const float *>(dl + off[j]);
    dv[j].y = *reinterpret_cast<const float *>(dl + off[j] + 16 * W3_RL * 4);
    cf[j] = *reinterpret_cast<const float *>(cl + 4 * ci[j]);
  }
}
template <int NE>
__device__ __forceinline__ f32x2 w3_combine(const f32x2 (&dv)[W3_MAXNE], const float (&cf)[W3_MAXNE]) {
  f32x2 v = dv[0] * f32x2{cf[0], cf[0]};
#pragma unroll
  for (int j = 1; j < NE; ++j) v = __builtin_elementwise_fma(dv[j], f32x2{cf[j], cf[j]}, v);
  return v;
}

#define W3_UNPACK(o0, c0, o1, c1, o2, c2, o3, c3, o4, c4, o5, c5, o6, c6, o7, c7, o8, c8, o9, c9, o10, c10, o11, c11) \
  constexpr int off_[W3_MAXNE] = {o0, o1, o2, o3, o4, o5, o6, o7, o8, o9, o10, o11};                                   \
  constexpr int ci_[W3_MAXNE] = {c0, c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11};

#define W3_A(set, joint)                                                                            \
  {                                                                                                 \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_)                                                \
        aS[set][m_] = *reinterpret_cast<const float *>(xl + 4 * (joint) + m_ * 16 * W3_RL * 4);      \
  }
#define W3_FIRST(ne, ...)                                        \
  {                                                              \
    W3_UNPACK(__VA_ARGS__)                                       \
    f32x2 dv_[W3_MAXNE]; float cf_[W3_MAXNE];                    \
    w3_gather_n<ne>(dl, cl, off_, ci_, dv_, cf_);                \
    b_cur = w3_combine<ne>(dv_, cf_);                            \
  }
#define W3_STEP(aset, slot, ne, ...)                             \
  {                                                              \
    W3_UNPACK(__VA_ARGS__)                                       \
    f32x2 dv_[W3_MAXNE]; float cf_[W3_MAXNE];                    \
    w3_gather_n<ne>(dl, cl, off_, ci_, dv_, cf_);                \
    __builtin_amdgcn_sched_barrier(0);                           \
    w3_mfma8(acc[slot], aS[aset], b_cur);                        \
    __builtin_amdgcn_sched_barrier(0);                           \
    b_cur = w3_combine<ne>(dv_, cf_);                            \
  }
#define W3_LAST(aset, slot)                \
  {                                        \
    __builtin_amdgcn_sched_barrier(0);     \
    w3_mfma8(acc[slot], aS[aset], b_cur);  \
  }

template <int WAVE>
__device__ __forceinline__ void w3_wave_main(const W3Params &p, float *lds, const float *__restrict__ x,
                                             const float *__restrict__ dz, float *__restrict__ dw_partial,
                                             float *__restrict__ colsum_partial) {
  constexpr int V = W3_V, RL = W3_RL, NW = W3_NW;
  constexpr int SET = w3_wave_set[WAVE], HALF = w3_wave_half[WAVE];
  constexpr int wave = WAVE;
  float *xs = lds;                                    // [64][RL] X tile
  float *ds = lds + W3_TILE;                          // [64][RL] dZ tile
  float *coef_l = lds + 2 * W3_TILE;                  // [ltot][V]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int g = lane >> 4, r = lane & 15;
  const size_t row_stride = (size_t)p.T * V;

  // lane bases: A operand X[16 m + r][frame g][.], gathers dZ[32 HALF + 16 n + r][frame g][.]
  const char *xl = reinterpret_cast<const char *>(xs + r * RL + g * V);
  const char *dl = reinterpret_cast<const char *>(ds + (32 * HALF + r) * RL + g * V);
  unsigned cl_off = (unsigned)((coef_l - lds) * sizeof(float));
  asm volatile("" : "+v"(cl_off));                    // opaque: see stgcn_gcn3.hip
  const char *cl = reinterpret_cast<const char *>(lds) + cl_off;

  // this wave's DMA pieces of a tile (same element offsets for both tensors)
  unsigned doff[W3_PW];
#pragma unroll
  for (int i = 0; i < W3_PW; ++i) {
    const int pc = i * NW + wave, e = pc * 64 + lane;
    const int row = e / (RL / 4), c4 = e - row * (RL / 4);
    doff[i] = (pc < W3_PIECES && e < W3_NV4) ? (unsigned)(((size_t)row * row_stride + 4 * c4) * sizeof(float)) : 0xffffffffu;
  }
  auto copy_tile = [&](int tile) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * W3_F;
    const float *xg = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    const float *dg = dz + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
#pragma unroll
    for (int i = 0; i < W3_PW; ++i)
      if (doff[i] != 0xffffffffu) {
        w3_dma16(xg, doff[i], xs + (i * NW + wave) * 256);
        w3_dma16(dg, doff[i], ds + (i * NW + wave) * 256);
      }
  };

  f32x4 acc[W3_MAXPL][2][4];                          // [plane of the set][n of the half][m]
#pragma unroll
  for (int s = 0; s < W3_MAXPL; ++s)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[s][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float aS[2][4];
  f32x2 b_cur;
  // The bias-table gradient (sum of dZ over samples and frames per (channel, joint)) is taken from the resident tile:
  // every thread owns seven (channel, joint) elements; their 28 LDS reads are unconditional (a thread without a 7th
  // element re-reads element 0 and drops the sum) so that they issue back to back in front of one wait.
  float cs[W3_CS];
  int cso[W3_CS];                                     // position in the dZ tile (first frame)
#pragma unroll
  for (int i = 0; i < W3_CS; ++i) {
    cs[i] = 0.f;
    const int idx = tid + NW * 64 * i;
    cso[i] = idx < 64 * V ? (idx / V) * RL + idx % V : 0;
  }

  // Tile order (round 5).  A tile row is 848 bytes of a 3,392-byte-per-16-frames channel row: 6.6 cache lines, so
  // neighbouring tiles share a line at each end.  With tile = blockIdx + i * grid the two owners of a shared line sat
  // on different XCDs (block b runs on XCD b % 8), each XCD's L2 fetched the line for itself and the launch moved
  // 1.22x its algorithmic bytes (profiles/r4_gcn3_pmc_traffic.json).  Now the 32 workgroups of an XCD walk 32
  // CONSECUTIVE tiles per round: the shared lines are hits in that XCD's L2.
  const int per_xcd = (gridDim.x & 7) == 0 ? (int)(gridDim.x >> 3) : 0;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  auto tile_of = [&](int i) { return per_xcd ? (i * 8 + xcd) * per_xcd + slot : (int)(blockIdx.x + i * gridDim.x); };
  int it = 0;
  int tile = tile_of(0);
  // Staggered start: the copy of a tile cannot overlap its own workgroup's MFMAs (no room for a second pair of
  // tiles), and workgroups running in lock-step all copy at the same moments -- 28 MB per round at the HBM rate,
  // with every matrix pipe idle.  A quarter of the workgroups each start 0 / 3.5 / 7 / 10.5 us late, so some
  // compute while others copy (measured: 1.03 -> 0.94 ms).
  for (int d = 0; d < (int)((per_xcd ? slot : blockIdx.x) & 3); ++d) __builtin_amdgcn_s_sleep(127);
  if (tile < p.total_tiles) copy_tile(tile);
  for (; tile < p.total_tiles; tile = tile_of(++it)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // own pieces have landed
    __syncthreads();                                       // ... everybody's

    if (colsum_partial) {
      float dv4[W3_CS][4];
#pragma unroll
      for (int i = 0; i < W3_CS; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f) dv4[i][f] = ds[cso[i] + f * V];
#pragma unroll
      for (int i = 0; i < W3_CS; ++i) cs[i] += (dv4[i][0] + dv4[i][1]) + (dv4[i][2] + dv4[i][3]);
    }
    if constexpr (SET == 0) { W3_BODY_0 } else if constexpr (SET == 1) { W3_BODY_1 }
    else if constexpr (SET == 2) { W3_BODY_2 } else { W3_BODY_3 }

    __syncthreads();                                       // nobody reads the tiles any more
    const int ntile = tile_of(it + 1);
    if (ntile < p.total_tiles) copy_tile(ntile);
  }

  if (colsum_partial) {
#pragma unroll
    for (int i = 0; i < W3_CS; ++i) {
      const int idx = tid + NW * 64 * i;
      if (idx < 64 * V) colsum_partial[(size_t)blockIdx.x * 64 * V + idx] = cs[i];
    }
  }
  // partial[block][k][ci][c]: D[row = 4 g + q][col = r] -> ci = 16 m + row, c = 32 HALF + 16 n + r
  float *out = dw_partial + (size_t)blockIdx.x * G3_K * 64 * 64;
#pragma unroll
  for (int s = 0; s < W3_MAXPL; ++s) {
    const int k = w3_set_planes[SET][s];
    if (k < 0) continue;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          out[((size_t)k * 64 + 16 * m + 4 * g + q) * 64 + 32 * HALF + 16 * n + r] = acc[s][n][m][q];
  }
}

__global__ __launch_bounds__(W3_NW * 64, 2) void gcn3_dw_kernel(W3Params p, const float *__restrict__ x,
                                                                const float *__restrict__ dz,
                                                                const float *__restrict__ coef,
                                                                float *__restrict__ dw_partial,
                                                                float *__restrict__ colsum_partial) {
  extern __shared__ float lds[];
  float *coef_l = lds + 2 * W3_TILE;
  const int tid = threadIdx.x;
  for (int e = tid; e < p.ltot * W3_V; e += W3_NW * 64) coef_l[e] = coef[e];
  __syncthreads();
  switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
    case 0: w3_wave_main<0>(p, lds, x, dz, dw_partial, colsum_partial); break;
    case 1: w3_wave_main<1>(p, lds, x, dz, dw_partial, colsum_partial); break;
    case 2: w3_wave_main<2>(p, lds, x, dz, dw_partial, colsum_partial); break;
    case 3: w3_wave_main<3>(p, lds, x, dz, dw_partial, colsum_partial); break;
    case 4: w3_wave_main<4>(p, lds, x, dz, dw_partial, colsum_partial); break;
    case 5: w3_wave_main<5>(p, lds, x, dz, dw_partial, colsum_partial); break;
    case 6: w3_wave_main<6>(p, lds, x, dz, dw_partial, colsum_partial); break;
    default: w3_wave_main<7>(p, lds, x, dz, dw_partial, colsum_partial); break;
  }
}

}  // namespace

// Weight gradient, statically scheduled for the P2RNet skeleton (the caller checks p2r_stgcn_gcn3_signature(1)
// against its row tables first).
//   x  (N,64,T,53): input of the graph conv       dz (N,64,T,53): gradient of its output
//   coef [ltot][53]: row coefficient table (values of A * importance at the row-list entries, zeros at padded slots)
//   dw_partial [n_blocks][K][64][64]: per-workgroup partial of dW_k TRANSPOSED, [k][ci][c]; the caller sums over the
//     leading axis and transposes (p2r_sum_leading with tr64).
//   colsum_partial (optional) [n_blocks][64][53]: per-workgroup sums of dz over samples and frames.
// T % 4 == 0 and x, dz 16-byte aligned (P2R_EINVAL otherwise: use p2r_stgcn_gcn_weight_grad).
extern "C" int p2r_stgcn_gcn3_weight_grad(int N, int T, int V, int K, int ltot, const float *x, const float *dz,
                                          const float *coef, int n_blocks, float *dw_partial, float *colsum_partial,
                                          void *stream) {
  if (N < 0 || T <= 0 || V != W3_V || K != G3_K || ltot <= 0 || n_blocks < 1) return P2R_EINVAL;
  if (T % W3_F != 0 || T > (1 << 20) || ((uintptr_t)x % 16) != 0 || ((uintptr_t)dz % 16) != 0) return P2R_EINVAL;
  if (N == 0) return P2R_EINVAL;
  W3Params p;
  p.T = T; p.ltot = ltot;
  p.tiles_per_seq = T / W3_F;
  const long long tiles = (long long)N * p.tiles_per_seq;
  if (tiles > 0x7fffffffLL) return P2R_EINVAL;
  p.total_tiles = (int)tiles;
  const size_t lds = (size_t)2 * W3_TILE * sizeof(float) + (size_t)ltot * V * sizeof(float);
  if (lds > 160 * 1024) return P2R_EINVAL;
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(gcn3_dw_kernel, lds_ok);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(gcn3_dw_kernel, dim3(n_blocks), dim3(W3_NW * 64), lds, p2r_stream(stream), p, x, dz, coef,
                     dw_partial, colsum_partial);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
