// stgcn_gcn3dwh.hip -- WEIGHT gradient of the fused graph convolution in `split16` arithmetic (opt-in mode; split16.h),
// statically scheduled for the P2RNet skeleton, gfx950.  Same operator as p2r_stgcn_gcn3_weight_grad (the autograd of
// reference stgcn_layers.py:57-67 with respect to conv.weight, plus the bias-table gradient):
//   dW_k[c][ci] = sum over (n, t, v) of V_k[c, t, v] * X[ci, t, v],   V_k(v) = sum_j a_k(v, w_j) dZ(:, :, w_j)   (row lists)
// Skeleton of stgcn_gcn3_dw.hip (persistent workgroups, 4-frame tiles of X and dZ, a wave = a set of planes x one half of the
// columns c, accumulators for the whole kernel).  What changes is the k-step: v_mfma_f32_16x16x32_f16 takes
// K = 32 = (frame kg of the tile) x (8 joints of a group on the REGISTER index i), so the row lists stay wave-uniform
// immediates, the X operand of lane (kg, r) is 8 consecutive joints of a tile row, and a unit = (plane, group of 8 joints):
// 56 live of 77 instead of 369 (plane, joint) units of one fp32 k-step each.  Both operands are runtime tensors:
//   * X (an activation) is split ONCE per tile by the whole workgroup on its way into LDS (register-staged, fp16 operand
//     slots [row][frame][part][56 joints]) and serves every wave and plane;
//   * the aggregate V_k is built in fp32 from the dZ tile (register-staged into a bank-friendly layout), with the coefficients as a
//     contiguous stream in schedule order that carries the power of two of dZ's range word, and split per unit; dZ is a
//     gradient -- heavy-tailed -- so its residual part is kept scaled by 2^11 and meets 2^-11 x1 (split16.h), formed
//     from x1 when a group's operands are read;
//   * three products per tile (x1' v2', x2 v1, x1 v1), builtin MFMAs.
// The bias-table gradient (column sums of dZ over the frames) is taken from the same LDS tile, in fp32.
// Schedule: tools/gen_gcn_split_dw_sched.py -> gcn3dwh_sched.inc.
#include "p2r_common.h"
#include "split16.h"

#include "gcn3dwh_sched.inc"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int V = 53, F = 4, NW = 8, C = 64;
constexpr int set_planes[4][DW_MAXPL] = DW_SET_PLANES;
constexpr int cs_off[4] = DW_CS_OFF;
__device__ const int dw_cs_idx[DW_NCS] = DW_CS_IDX;      // the coefficient stream: indices into the flattened [ltot][53] table
// Both tiles are staged through registers into layouts chosen for the LDS banks (profiles/r6_split16_mfma_util.json: with
// the dZ tile as LDS-DMA leaves it -- rows of 212 floats -- and X slots of 224 bytes, 49 % of the kernel's LDS cycles were
// bank conflicts):
//  * dZ [row c][frame][joint]: row stride DZR = 321 floats (1 mod 32), frame stride DZF = 80 (16 mod 32).  A gather reads,
//    per 32-lane half of the wave, rows r = 0..15 at frames kg and kg + 1 of one joint: banks r + 16 kg (+ joint): all 32
//    different;
//  * X, already split: [frame][row] slots of XSLOT = 240 bytes = [part][56 joints] halves + 16 bytes of padding.  A
//    16-byte operand read serves 8 lanes per cycle -- rows r .. r + 7 of one frame: 60 r mod 32 words = eight different
//    4-bank groups.
constexpr int DZR = 321, DZF = 80;
constexpr int DZ_TILE = C * DZR;                     // floats (82,176 bytes)
constexpr int XJ = 56, XSLOT = 240;                  // joints per part (53 + 3 zeros), bytes per (frame, row) slot
constexpr int XTILE_F = F * C * XSLOT / 4;           // the split tile in floats (61,440 bytes)
constexpr int BITEMS = (C * V + NW * 64 - 1) / (NW * 64);      // (channel, joint) sums of the bias-table gradient per thread: 7

struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
struct Params { int T, tiles_per_seq, total_tiles; const unsigned *x_amax, *dz_amax; };
struct Split { p2r_h8 p, q; };

// a group's X operands: the two parts of rows 16 m + r, frame kg, joints 8 grp .. 8 grp + 7, and 2^-11 x1 for the product
// with the aggregate's scaled residual
#define DW_A(grp)                                                                                \
  {                                                                                              \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) {                                           \
      A[m_].p = *reinterpret_cast<const p2r_h8 *>(xl + m_ * (16 * XSLOT) + 16 * (grp));          \
      A[m_].q = *reinterpret_cast<const p2r_h8 *>(xl + m_ * (16 * XSLOT) + XJ * 2 + 16 * (grp)); \
      As[m_] = A[m_].p * (_Float16)(1.0 / P2R_RES_SCALE);                                        \
    }                                                                                            \
  }
// both n-tiles of a unit together, the aggregate's operands read from the dZ tile at immediate offsets
#define DW_G(i, first, off, e)                                                                   \
  {                                                                                              \
    const float c_ = *reinterpret_cast<const float *>(cl + 4 * (e));                             \
    const float d0_ = *reinterpret_cast<const float *>(dl + (off)), d1_ = *reinterpret_cast<const float *>(dl + 16 * DZR * 4 + (off)); \
    Vg[i] = (first) ? c_ * d0_ : fmaf(c_, d0_, Vg[i]);                                           \
    Vh[i] = (first) ? c_ * d1_ : fmaf(c_, d1_, Vh[i]);                                           \
  }
#define DW_Z2(i) { Vg[i] = 0.f; Vh[i] = 0.f; }
#define DW_M2(slot)                                                                              \
  {                                                                                              \
    const P2RSplit8 b0_ = p2r_split8(Vg), b1_ = p2r_split8(Vh);                                  \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) {                                           \
      acc[slot][0][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(As[m_], b0_.q, acc[slot][0][m_], 0, 0, 0); \
      acc[slot][0][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].q, b0_.p, acc[slot][0][m_], 0, 0, 0); \
      acc[slot][0][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].p, b0_.p, acc[slot][0][m_], 0, 0, 0); \
      acc[slot][1][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(As[m_], b1_.q, acc[slot][1][m_], 0, 0, 0); \
      acc[slot][1][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].q, b1_.p, acc[slot][1][m_], 0, 0, 0); \
      acc[slot][1][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m_].p, b1_.p, acc[slot][1][m_], 0, 0, 0); \
    }                                                                                            \
  }

template <int SET>
__device__ __forceinline__ void wave_main(const Params &p, float *lds, const float *__restrict__ x, const float *__restrict__ dz,
                                          float *__restrict__ part, float *__restrict__ bpart, float xs, float inv) {
  float *xt = lds, *dt = lds + XTILE_F, *coef_l = dt + DZ_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = lane >> 4, r = lane & 15, half = wave & 1;
  const size_t row_stride = (size_t)p.T * V;
  // A operand: X[16 m + r][frame kg][joint 8 grp + i]; gathers: dZ[16 (2 half + n) + r][frame kg][w]
  const char *xl = reinterpret_cast<const char *>(xt) + (kg * C + r) * XSLOT;
  const char *dl = reinterpret_cast<const char *>(dt + (32 * half + r) * DZR + kg * DZF);
  unsigned cl_off = (unsigned)((coef_l - lds + cs_off[SET]) * sizeof(float));
  asm volatile("" : "+v"(cl_off));                     // opaque base: the coefficient reads stay LDS reads (stgcn_gcn3.hip)
  const char *cl = reinterpret_cast<const char *>(lds) + cl_off;

  // this thread's items of the two tiles, (row, frame, 4-joint chunk): global offset (floats, the same for X and dZ) and
  // the LDS offsets (X: halves; dZ: floats)
  // Recomputed for every tile from an opaque copy of the thread index, so that the compiler cannot keep these 28 values in
  // registers through the tile's work: held for the whole kernel they pushed 148 registers into scratch INSIDE the unit
  // loop (0.64 ms per launch; 24 spilled registers and 0.51 ms this way -- the integer arithmetic is ~100 instructions
  // per tile).
  int xg[7], xo[7], zo[7];
  bool xlast[7];                                       // the chunk that holds joint 52 alone (53 = 13 x 4 + 1)
  auto offsets = [&]() {
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int item = i * NW * 64 + tid_, row = item / (F * (XJ / 4)), rem = item % (F * (XJ / 4)), f = rem / (XJ / 4), ch = rem % (XJ / 4);
      xg[i] = (int)((size_t)row * row_stride + f * V + 4 * ch);
      xo[i] = (f * C + row) * (XSLOT / 2) + 4 * ch;
      zo[i] = row * DZR + f * DZF + 4 * ch;
      xlast[i] = 4 * ch + 4 > V;
    }
  };
  f32x4 acc[DW_MAXPL][2][4];
#pragma unroll
  for (int s = 0; s < DW_MAXPL; ++s)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[s][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bacc[BITEMS];
#pragma unroll
  for (int i = 0; i < BITEMS; ++i) bacc[i] = 0.f;
  Split A[4];
  p2r_h8 As[4];
  float Vg[8], Vh[8];

  // Both tiles go through registers: X is scaled and converted on the way, dZ lands in its bank-friendly layout.  Neither
  // has a second home in LDS: both are fetched between the tiles, all fourteen loads of a thread in flight together.
  float xv[7][4], zv[7][4];
  auto tile_base = [&](int tile) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * F;
    return (size_t)seq * C * row_stride + (size_t)t0 * V;
  };
  auto fetch = [&](const float *t, size_t base, float (&v)[7][4]) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const float *src = t + base + xg[i];
      if (!xlast[i]) {       // one 16-byte load at 4-byte alignment
        const F4 u = *reinterpret_cast<const F4 *>(src);
        v[i][0] = u.x; v[i][1] = u.y; v[i][2] = u.z; v[i][3] = u.w;
      } else {               // joint 52 alone: nothing is read past the row's 53rd joint (the tensor may end there)
        v[i][0] = src[0]; v[i][1] = 0.f; v[i][2] = 0.f; v[i][3] = 0.f;
      }
    }
  };
  auto put_dz = [&]() {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      float *d = dt + zo[i];
      d[0] = zv[i][0];
      if (!xlast[i]) { d[1] = zv[i][1]; d[2] = zv[i][2]; d[3] = zv[i][3]; }
    }
  };
  auto put_x = [&]() {
    _Float16 *xsl = reinterpret_cast<_Float16 *>(xt);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      p2r_f2 a = {xv[i][0] * xs, xv[i][1] * xs}, b = {xv[i][2] * xs, xv[i][3] * xs};
      asm volatile("" : "+v"(a), "+v"(b));
      const p2r_h2 pa = __builtin_convertvector(a, p2r_h2), pb = __builtin_convertvector(b, p2r_h2);
      const p2r_h2 qa = __builtin_convertvector(a - __builtin_convertvector(pa, p2r_f2), p2r_h2);
      const p2r_h2 qb = __builtin_convertvector(b - __builtin_convertvector(pb, p2r_f2), p2r_h2);
      *reinterpret_cast<h4 *>(xsl + xo[i]) = h4{pa.x, pa.y, pb.x, pb.y};
      *reinterpret_cast<h4 *>(xsl + xo[i] + XJ) = h4{qa.x, qa.y, qb.x, qb.y};
    }
  };
  // Tile order as in stgcn_gcn3_dw.hip: a tile row is 848 bytes of a channel row -- 6.6 cache lines, neighbouring tiles
  // share a line at each end -- and workgroup b runs on XCD b % 8: the 32 workgroups of an XCD walk 32 CONSECUTIVE tiles
  // per round, so the shared lines are hits in that XCD's L2 (measured here before the change: 1.36x the algorithmic reads)
  const int per_xcd = (gridDim.x & 7) == 0 ? (int)(gridDim.x >> 3) : 0;
  const int xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  auto tile_of = [&](int i) { return per_xcd ? (i * 8 + xcd) * per_xcd + xslot : (int)(blockIdx.x + i * gridDim.x); };
  int it = 0;
  for (int tile = tile_of(0); tile < p.total_tiles; tile = tile_of(++it)) {
    const size_t base = tile_base(tile);
    offsets();
    fetch(x, base, xv);
    fetch(dz, base, zv);
    put_x();
    put_dz();
    __syncthreads();
    if (bpart) {      // bias-table gradient: column sums of the dZ tile over its frames (fp32, as the tensor has it)
#pragma unroll
      for (int i = 0; i < BITEMS; ++i) {
        const int item = i * NW * 64 + tid;
        if (item < C * V) {
          const int c = item / V, v = item - c * V;
          const float *d = dt + c * DZR + v;
          bacc[i] += (d[0] + d[DZF]) + (d[2 * DZF] + d[3 * DZF]);
        }
      }
    }
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
        for (int q = 0; q < 4; ++q) o[(size_t)(16 * m + 4 * kg + q) * C + 16 * (2 * half + n) + r] = acc[s][n][m][q] * inv;
  }
  if (bpart) {
#pragma unroll
    for (int i = 0; i < BITEMS; ++i) {
      const int item = i * NW * 64 + tid;
      if (item < C * V) bpart[(size_t)blockIdx.x * C * V + item] = bacc[i];
    }
  }
}

__global__ __launch_bounds__(NW * 64, 2) void gcn3dwh_kernel(Params p, const float *__restrict__ x, const float *__restrict__ dz,
                                                             const float *__restrict__ coef, float *__restrict__ part,
                                                             float *__restrict__ bpart) {
  extern __shared__ float lds[];
  float *coef_l = lds + XTILE_F + DZ_TILE;
  const int tid = threadIdx.x;
  // operand scales from the range words: x (activation) is scaled when it is split, dZ through the coefficient stream;
  // the accumulators hold 2^(S_x + S_dz) dW
  float xs, xinv, gs, ginv;
  p2r_split_scale(p.x_amax, xs, xinv);
  p2r_split_scale(p.dz_amax, gs, ginv);
  for (int e = tid; e < DW_NCS; e += NW * 64) coef_l[e] = coef[dw_cs_idx[e]] * gs;
  __syncthreads();
  switch (__builtin_amdgcn_readfirstlane(tid >> 7)) {
    case 0: wave_main<0>(p, lds, x, dz, part, bpart, xs, xinv * ginv); break;
    case 1: wave_main<1>(p, lds, x, dz, part, bpart, xs, xinv * ginv); break;
    case 2: wave_main<2>(p, lds, x, dz, part, bpart, xs, xinv * ginv); break;
    default: wave_main<3>(p, lds, x, dz, part, bpart, xs, xinv * ginv); break;
  }
}
}  // namespace

extern "C" unsigned long long p2r_stgcn_gcn3h_weight_grad_signature(void) { return DW_SIGNATURE; }

// x, dz (N,64,T,53) f32, 16-byte aligned, T % 4 == 0; coef [ltot][53] the row-form coefficient table (as
// p2r_stgcn_gcn3_weight_grad takes it); dw_partial [n_blocks][K][64 ci][64 c] (dW_k transposed), dbias_partial
// [n_blocks][64][53] or NULL: both summed over the leading axis by the caller; x_amax / dz_amax: range words (NULL: scale 1).
extern "C" int p2r_stgcn_gcn3h_weight_grad(int N, int T, int V_, int K, int ltot, const float *x, const float *dz,
                                           const float *coef, int n_blocks, float *dw_partial, float *dbias_partial,
                                           const unsigned *x_amax, const unsigned *dz_amax, void *stream) {
  if (N < 0 || T <= 0 || T % F != 0 || V_ != V || K != DW_K || ltot != DW_LTOT || n_blocks < 1 || n_blocks > 65535) return P2R_EINVAL;
  if (((uintptr_t)x % 16) != 0 || ((uintptr_t)dz % 16) != 0 || !x || !dz || !coef || !dw_partial) return P2R_EINVAL;
  if ((long long)C * T * V * 4 >= (1LL << 31)) return P2R_EINVAL;          // 32-bit byte offsets inside a sample
  const long long tiles = (long long)N * (T / F);
  if (tiles > 0x7fffffffLL) return P2R_EINVAL;
  Params p;
  p.T = T; p.tiles_per_seq = T / F; p.total_tiles = (int)tiles; p.x_amax = x_amax; p.dz_amax = dz_amax;
  const size_t lds = ((size_t)XTILE_F + DZ_TILE + (size_t)DW_NCS) * sizeof(float);
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(gcn3dwh_kernel, lds_ok);
  if (e != hipSuccess) return (int)e;
  // every workgroup writes its partial (zeros when it has no tile): the caller sums all n_blocks rows
  hipLaunchKernelGGL(gcn3dwh_kernel, dim3(n_blocks), dim3(NW * 64), lds, p2r_stream(stream), p, x, dz, coef, dw_partial,
                     dbias_partial);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
