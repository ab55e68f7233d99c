// PROTOTYPE, not part of the product: the WEIGHT gradient of the fused graph convolution on two-part fp16 MFMA products
//   dW_k[c][ci] = sum over (n, t, v) of V_k[c, t, v] * X[ci, t, v],   V_k(v) = sum_j a_k(v, w_j) dZ(:, :, w_j)   (row lists)
// (stgcn_gcn3_dw.hip's operator and skeleton: persistent workgroups, 4-frame tiles of X and dZ copied by LDS-DMA as the
// tensors have them, a wave = a set of planes x one half of the columns c, accumulators for the whole kernel).  What
// changes is the k-step: v_mfma_f32_16x16x32_f16 takes K = 32 = (frame kg of the tile) x (8 joints of a group on the
// REGISTER index i), so the row lists stay wave-uniform immediates, the X operand of lane (kg, r) is 8 consecutive floats of
// a tile row, and a unit = (plane, group of 8 joints): 56 live of 77 instead of 369 (plane, joint) units of one fp32
// k-step each.  Both operands are runtime tensors: X is split once per tile by the whole workgroup on its way into LDS
// (register-staged, fp16 operand slots) and serves every wave and plane; the aggregate V_k is built in fp32 (the coefficient table carries the power of two that lifts the gradient
// into fp16's range) and split per unit; three products per tile (x1 v2, x2 v1, x1 v1).  Schedule:
// tools/gen_gcn_dwh_sched.py -> gcn3dwh_sched.inc.  No bias-table gradient here.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "gcn3dwh_sched.inc"
#ifdef ABL_NO_MFMA      // timing ablation
#define ABL_MFMA 0
#else
#define ABL_MFMA 1
#endif

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int V = 53, F = 4, NW = 8, C = 64;
constexpr int RL = F * V;                 // 212 floats per tile row
constexpr int TILE = C * RL;
constexpr int NV4 = TILE / 4;             // 3392 float4 = 53 pieces of 64
constexpr int PIECES = (NV4 + 63) / 64;
constexpr int PW = (PIECES + NW - 1) / NW;
constexpr int set_planes[4][DW_MAXPL] = DW_SET_PLANES;
constexpr int cs_off[4] = DW_CS_OFF;
// X in LDS: already split, [row][frame][part][56 joints] halves (joints 53-55 zero): what lane (kg, r) reads for a group is
// one 16-byte slot per part.  Built once per tile by the whole workgroup (eight waves would otherwise split the same group)
constexpr int XJ = 56, XROW = F * 2 * XJ;            // 448 halves = 896 bytes per row
constexpr int XTILE_F = C * XROW / 2;                // the split tile in floats (57,344 bytes)
constexpr int XITEMS = C * F * (XJ / 4);             // (row, frame, 4-joint chunk) items: 3584 = 7 per thread
struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };

struct Params { int T, tiles_per_seq, total_tiles; float inv_scale; };
struct Split { h8 p, q; };

__device__ __forceinline__ Split split8(const float (&v)[8]) {
  Split s;
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    f2 x = {v[i], v[i + 1]};
    asm volatile("" : "+v"(x));                            // the split sees VALUES (tools/ubench/split_probe.hip)
    const h2 p = __builtin_convertvector(x, h2);
    const h2 q = __builtin_convertvector(x - __builtin_convertvector(p, f2), h2);
    s.p[i] = p.x; s.p[i + 1] = p.y; s.q[i] = q.x; s.q[i + 1] = q.y;
  }
  return s;
}
__device__ __forceinline__ unsigned lds_addr(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
__device__ __forceinline__ void dma16(const float *base, unsigned voff, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

#define DW_A(grp)                                                                                \
  {                                                                                              \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) {                                           \
      A[m_].p = *reinterpret_cast<const h8 *>(xl + m_ * (16 * XROW * 2) + 16 * (grp));           \
      A[m_].q = *reinterpret_cast<const h8 *>(xl + m_ * (16 * XROW * 2) + XJ * 2 + 16 * (grp));  \
    }                                                                                            \
  }
// the dZ row of (n-tile nt of the wave's half, frame kg): 53 registers for a pass over the wave's units
#define DW_ROW(nt)                                                                               \
  {                                                                                              \
    _Pragma("unroll") for (int w_ = 0; w_ < V; ++w_) dzr[w_] = *reinterpret_cast<const float *>(dl + (nt) * (16 * RL * 4) + 4 * w_); \
  }
#ifdef ABL_NO_G         // timing ablation: no multiply-adds
#define DW_F(i, first, w, e) { asm volatile("" : "+v"(Vg[i])); }
#else
#define DW_F(i, first, w, e)                                                                     \
  {                                                                                              \
    const float c_ = *reinterpret_cast<const float *>(cl + 4 * (e));                             \
    Vg[i] = (first) ? c_ * dzr[w] : fmaf(c_, dzr[w], Vg[i]);                                     \
  }
#endif
#define DW_Z(i) { Vg[i] = 0.f; }
// GEN_GATHER=lds: both n-tiles of a unit together, the aggregate's operands read from the dZ tile at immediate offsets
#define DW_G(i, first, off, e)                                                                   \
  {                                                                                              \
    const float c_ = *reinterpret_cast<const float *>(cl + 4 * (e));                             \
    const float d0_ = *reinterpret_cast<const float *>(dl + (off)), d1_ = *reinterpret_cast<const float *>(dl + 16 * RL * 4 + (off)); \
    Vg[i] = (first) ? c_ * d0_ : fmaf(c_, d0_, Vg[i]);                                           \
    Vh[i] = (first) ? c_ * d1_ : fmaf(c_, d1_, Vh[i]);                                           \
  }
#define DW_Z2(i) { Vg[i] = 0.f; Vh[i] = 0.f; }
#define DW_M2(slot)                                                                              \
  {                                                                                              \
    const Split b0_ = split8(Vg), b1_ = split8(Vh);                                              \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) {                                           \
      acc[slot][0][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].p, b0_.q, acc[slot][0][m_], 0, 0, 0); \
      acc[slot][0][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].q, b0_.p, acc[slot][0][m_], 0, 0, 0); \
      acc[slot][0][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].p, b0_.p, acc[slot][0][m_], 0, 0, 0); \
      acc[slot][1][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].p, b1_.q, acc[slot][1][m_], 0, 0, 0); \
      acc[slot][1][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].q, b1_.p, acc[slot][1][m_], 0, 0, 0); \
      acc[slot][1][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].p, b1_.p, acc[slot][1][m_], 0, 0, 0); \
    }                                                                                            \
  }
#define DW_M(slot, nt)                                                                           \
  {                                                                                              \
    const Split b_ = split8(Vg);                                                                 \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) {                                           \
      if (ABL_MFMA == 0) { asm volatile("" : "+v"(acc[slot][nt][m_]) : "v"(A[m_].p), "v"(A[m_].q), "v"(b_.p), "v"(b_.q)); continue; } \
      acc[slot][nt][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].p, b_.q, acc[slot][nt][m_], 0, 0, 0); \
      acc[slot][nt][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].q, b_.p, acc[slot][nt][m_], 0, 0, 0); \
      acc[slot][nt][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].p, b_.p, acc[slot][nt][m_], 0, 0, 0); \
    }                                                                                            \
  }

template <int SET>
__device__ __forceinline__ void wave_main(const Params &p, float *lds, const float *__restrict__ x, const float *__restrict__ dz,
                                          float *__restrict__ part) {
  float *xt = lds, *dt = lds + XTILE_F, *coef_l = dt + TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = lane >> 4, r = lane & 15, half = wave & 1;
  const size_t row_stride = (size_t)p.T * V;
  // A operand: X[16 m + r][frame kg][joint 8 grp + i]; gathers: dZ[16 (2 half + n) + r][frame kg][w]
  const char *xl = reinterpret_cast<const char *>(xt) + (r * F + kg) * (2 * XJ * 2);
  const char *dl = reinterpret_cast<const char *>(dt + (32 * half + r) * RL + kg * V);
  unsigned cl_off = (unsigned)((coef_l - lds + cs_off[SET]) * sizeof(float));
  asm volatile("" : "+v"(cl_off));                     // opaque base: the coefficient reads stay LDS reads (stgcn_gcn3.hip)
  const char *cl = reinterpret_cast<const char *>(lds) + cl_off;

  int doff[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int pc = i * NW + wave, e = pc * 64 + lane;
    const int row = e / (RL / 4), c4 = e - row * (RL / 4);
    doff[i] = (pc < PIECES && e < NV4) ? (int)(((size_t)row * row_stride + 4 * c4) * sizeof(float)) : -1;
  }

  // this thread's items of the X tile: global offset (floats) and LDS offset (halves) of (row, frame, chunk)
  int xg[7], xo[7], xn[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int item = i * NW * 64 + tid, row = item / (F * (XJ / 4)), rem = item % (F * (XJ / 4)), f = rem / (XJ / 4), ch = rem % (XJ / 4);
    xg[i] = (int)((size_t)row * row_stride + f * V + 4 * ch);
    xo[i] = (row * F + f) * (2 * XJ) + 4 * ch;
    xn[i] = V - 4 * ch < 4 ? V - 4 * ch : 4;          // valid joints of the chunk (the last one: 1)
  }
  f32x4 acc[DW_MAXPL][2][4];
#pragma unroll
  for (int s = 0; s < DW_MAXPL; ++s)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[s][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  Split A[4];
  float Vg[8], Vh[8], dzr[V];

  // The X tile goes through registers (it is converted on the way).  -DPREFETCH_X issues its loads for the NEXT tile
  // before this tile's units so that they land under them: 28 more live registers, 132 spilled, 0.725 instead of 0.627 ms
  // -- off.  The dZ tile has no second home in LDS and is copied between the tiles.
  F4 xv[7];
  auto tile_base = [&](int tile) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * F;
    return (size_t)seq * C * row_stride + (size_t)t0 * V;
  };
  auto fetch_x = [&](size_t base) {
#pragma unroll
    for (int i = 0; i < 7; ++i) xv[i] = *reinterpret_cast<const F4 *>(x + base + xg[i]);
  };
  auto put_x = [&]() {
    _Float16 *xs = reinterpret_cast<_Float16 *>(xt);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      f2 a = {xv[i].x, xn[i] > 1 ? xv[i].y : 0.f}, b = {xn[i] > 2 ? xv[i].z : 0.f, xn[i] > 3 ? xv[i].w : 0.f};
      const h2 pa = __builtin_convertvector(a, h2), pb = __builtin_convertvector(b, h2);
      const h2 qa = __builtin_convertvector(a - __builtin_convertvector(pa, f2), h2);
      const h2 qb = __builtin_convertvector(b - __builtin_convertvector(pb, f2), h2);
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<h4 *>(xs + xo[i]) = h4{pa.x, pa.y, pb.x, pb.y};
      *reinterpret_cast<h4 *>(xs + xo[i] + XJ) = h4{qa.x, qa.y, qb.x, qb.y};
    }
  };
#ifdef PREFETCH_X
  if ((int)blockIdx.x < p.total_tiles) fetch_x(tile_base(blockIdx.x));
#endif
  for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
    const size_t base = tile_base(tile);
#pragma unroll
    for (int i = 0; i < PW; ++i)
      if (doff[i] >= 0) dma16(dz + base, doff[i], dt + (i * NW + wave) * 256);
#ifndef PREFETCH_X
    fetch_x(base);
#endif
    put_x();
#ifdef PREFETCH_X
    if (tile + (int)gridDim.x < p.total_tiles) fetch_x(tile_base(tile + gridDim.x));
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");   // the dZ pieces, not the seven X loads behind them
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if constexpr (SET == 0) { DW_BODY_0 } else if constexpr (SET == 1) { DW_BODY_1 }
    else if constexpr (SET == 2) { DW_BODY_2 } else { DW_BODY_3 }
    __syncthreads();                                   // both tiles are free again
  }
  // partial [block][k][ci][c] (dW_k transposed, as stgcn_gcn3_dw.hip leaves it): ci = 16 m + 4 kg + q, c = 16 (2 half + n) + r
#pragma unroll
  for (int s = 0; s < DW_MAXPL; ++s) {
    const int k = set_planes[SET][s];
    if (k < 0) continue;
    float *o = part + ((size_t)blockIdx.x * DW_K + k) * C * C;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) o[(size_t)(16 * m + 4 * kg + q) * C + 16 * (2 * half + n) + r] = acc[s][n][m][q] * p.inv_scale;
  }
}

__global__ __launch_bounds__(NW * 64, 2) void gcn3dwh_kernel(Params p, const float *__restrict__ x, const float *__restrict__ dz,
                                                             const float *__restrict__ coef, float *__restrict__ part) {
  extern __shared__ float lds[];
  float *coef_l = lds + XTILE_F + TILE;
  const int tid = threadIdx.x;
  for (int e = tid; e < DW_NCS; e += NW * 64) coef_l[e] = coef[e];
  __syncthreads();
  switch (__builtin_amdgcn_readfirstlane(tid >> 7)) {
    case 0: wave_main<0>(p, lds, x, dz, part); break;
    case 1: wave_main<1>(p, lds, x, dz, part); break;
    case 2: wave_main<2>(p, lds, x, dz, part); break;
    default: wave_main<3>(p, lds, x, dz, part); break;
  }
}
}  // namespace

// x, dz (N,64,T,53) f32, 16-byte aligned (x with 4 readable floats behind its last element), T % 4 == 0; coef f32 [DW_NCS]: the coefficient STREAM -- the entries
// proto_gcn3dwh_stream lists of the flattened ROW-form table, in schedule order -- times the power of two 1 / inv_scale; part [blocks][K][64 ci][64 c] (blocks = min(N T / 4, 256); NULL: returns the count).
extern "C" int proto_gcn3dwh(int N, int T, int ltot, const float *x, const float *dz, const float *coef, float inv_scale,
                             float *part, void *stream) {
  if (N <= 0 || T <= 0 || T % F != 0 || ltot != DW_LTOT) return -1;
  Params p;
  p.T = T; p.tiles_per_seq = T / F; p.total_tiles = N * p.tiles_per_seq; p.inv_scale = inv_scale;
  const int blocks = p.total_tiles < 256 ? p.total_tiles : 256;
  if (!part) return blocks;
  const size_t lds = ((size_t)XTILE_F + TILE + (size_t)DW_NCS) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gcn3dwh_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return -2;
  hipLaunchKernelGGL(gcn3dwh_kernel, dim3(blocks), dim3(NW * 64), lds, (hipStream_t)stream, p, x, dz, coef, part);
  return hipGetLastError() == hipSuccess ? blocks : -3;
}

// the coefficient stream's entries: indices into the flattened [DW_LTOT][53] row-form coefficient table
extern "C" int proto_gcn3dwh_stream(int *out) {
  constexpr int idx[DW_NCS] = DW_CS_IDX;
  if (out) for (int i = 0; i < DW_NCS; ++i) out[i] = idx[i];
  return DW_NCS;
}
