// stgcn_gcn3_grad.hip -- adjacency gradient of the fused graph convolution, statically scheduled (gfx950).
//
//   dcoef[k][j][v] = sum over (n, t, c) of Y_k[c, t, v] * dZ[c, t, w_j(k, v)],     Y_k = W_k . X
// at the non-zero entries (k, v, w_j) of the adjacency (the gradient reaching `A * importance`,
// reference models/p2rnet/modules/stgcn.py:134 / stgcn_layers.py:62-65 through autograd) -- the operator of
// gcn_dcoef_kernel in stgcn_gcn.hip, rebuilt on the skeleton of stgcn_gcn3.hip:
//
//   * MFMA n-tile = 16 frames of ONE joint v (no gather on the MFMA side at all: the B operand of the product
//     Y_k(v) = W_k . X(v) is X itself).  A wave owns up to 7 joints and keeps their B operands -- all 64 input channels
//     of 16 frames -- in registers for the whole tile (16 VGPRs per joint, loaded once per tile straight from HBM/L2).
//   * The product is split along its OUTPUT rows: phase p computes rows 16p..16p+15 of Y_k(v) (16 k-steps into one
//     4-register tile) and reduces them at once against rows 16p..16p+15 of dZ at the row-list joints -- so of dZ
//     only a 16-row slice has to be resident (two 53 KB LDS buffers, filled by LDS-DMA like the X slices of the
//     forward kernel), and nothing of size 64 rows x tile ever is.  The slice is stored with its rows permuted
//     (LDS row 4q+g holds row 4g+q) so that the accumulator layout of the MFMA (lane (g, r) holds rows 4g..4g+3 of
//     frame r) reads it bank-conflict-free with immediate offsets.
//   * The (plane, joint) units of the ROW lists that are empty (214 of 583) are skipped exactly; the work list is
//     code generated at build time (tools/gen_gcn_sched.py, D3_BODY_<wave>), as in stgcn_gcn3.hip.
//   * Each product is reduced over the wave with DPP row sums + two lane swaps; one lane accumulates it into the
//     workgroup's LDS table [ltot][V] (owned entries: no contention), written out once per workgroup.
#include "p2r_common.h"

#include "gcn3_sched.inc"

// Cycle trace (profiling hook, off in the product build; see stgcn_tconv3.hip): -DP2R_CYCLE_TRACE, tools/dev_g3_trace.py
#ifdef P2R_CYCLE_TRACE
__device__ unsigned long long d3_trace[8 * 32];
extern "C" int p2r_debug_d3_trace(unsigned long long *dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(d3_trace), sizeof(d3_trace));
}
#define D3_TRACE_TILE(tile) const bool trace_on = blockIdx.x == 7 && (tile) == 7 + 3 * (int)gridDim.x
#define D3_MARK(i) do { if (trace_on && lane == 0) d3_trace[wave * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define D3_TRACE_TILE(tile)
#define D3_MARK(i)
#endif
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int D3_F = 16;
constexpr int D3_NW = 8;
constexpr int D3_SLOTS = 7;
constexpr int D3_V = G3_V;
constexpr int D3_RS = D3_F * D3_V;          // 848
constexpr int D3_BUF = 16 * D3_RS;          // floats per 16-row slice
constexpr int D3_NV4 = D3_BUF / 4;
constexpr int D3_PIECES = (D3_NV4 + 63) / 64;          // 53
constexpr int D3_PW = (D3_PIECES + D3_NW - 1) / D3_NW; // 7

struct D3Params {
  int T, ltot;
  int tiles_per_seq, total_tiles;
};

constexpr int d3_slot_joints[D3_NW][D3_SLOTS] = G3_SLOT_JOINTS_1;
constexpr int d3_plane0[D3_NW] = G3_PLANE0_1;

__device__ __forceinline__ unsigned d3_lds_addr(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
__device__ __forceinline__ void d3_dma16(const float *base, int voff, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(d3_lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// Y tile of one (plane, joint) unit: 16 k-steps (all 64 input channels) into ONE accumulator; the first MFMA starts
// from the inline constant 0.  Two assembly blocks of 8 (operand-count limit of one asm statement).
__device__ __forceinline__ void d3_mfma16(f32x4 &h, const float (&a)[16], const float (&b)[16]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %1, %9, 0\n\tv_mfma_f32_16x16x4_f32 %0, %2, %10, %0\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %3, %11, %0\n\tv_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\tv_mfma_f32_16x16x4_f32 %0, %6, %14, %0\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %7, %15, %0\n\tv_mfma_f32_16x16x4_f32 %0, %8, %16, %0"
      : "=&v"(h)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
        "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
  asm volatile(
      "v_mfma_f32_16x16x4_f32 %0, %1, %9, %0\n\tv_mfma_f32_16x16x4_f32 %0, %2, %10, %0\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %3, %11, %0\n\tv_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\tv_mfma_f32_16x16x4_f32 %0, %6, %14, %0\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %7, %15, %0"
      : "+v"(h)
      : "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]),
        "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11]), "v"(b[12]), "v"(b[13]), "v"(b[14]), "v"(b[15]));
  // the last one through the builtin: the compiler then knows that `h` comes out of the matrix pipe and provides
  // the MFMA -> VALU wait states itself (filling them with independent instructions where it can)
  h = __builtin_amdgcn_mfma_f32_16x16x4f32(a[15], b[15], h, 0, 0, 0);
}

// dv[j][q] = dZ slice row (4 g + q), frame r, joint of entry j  (LDS row 4 q + g: offset q * 4 * RS floats)
template <int NE, int O0, int O1, int O2, int O3, int O4, int O5>
__device__ __forceinline__ void d3_gather(const char *xl, float (&dv)[6][4]) {
  constexpr int off[6] = {O0, O1, O2, O3, O4, O5};
#pragma unroll
  for (int j = 0; j < NE; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) dv[j][q] = *reinterpret_cast<const float *>(xl + off[j] + q * 4 * D3_RS * 4);
}

// v summed over the 64 lanes (every lane gets the total): DPP within the rows of 16, gfx950 lane swaps across them
__device__ __forceinline__ float d3_wave_sum(float v) {
  v = p2r_row16_sum(v);
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Wave reduction of the step's per-lane products.  VALU work is matrix-pipe time on gfx950 (fp32 MFMAs issue through
// the vector datapath), so the reduction is built to need as few vector instructions per entry as possible:
//   * products as packed pairs (v_pk_mul / v_pk_fma on the register pairs the MFMA tile and the 2-address LDS reads
//     deliver) + one add;
//   * entries are reduced TWO per register: v_permlane32_swap exchanges the upper half of entry A with the lower half
//     of entry B, one add folds both -- lanes 0-31 then carry A, lanes 32-63 carry B through the same four DPP row
//     stages, and one row_bcast15 add leaves A's total in lane 31 and B's in lane 63;
//   * an unpaired entry takes row_bcast15 + row_bcast31 (total in lane 63) instead of two lane swaps;
//   * all chains of a step advance stage by stage, so independent instructions fill the DPP wait states;
//   * lanes 31 / 63 add the totals into the workgroup's LDS table (no return value: nothing waits for the atomic).
template <int CTRL>
__device__ __forceinline__ float d3_add_dpp(float v) {      // bound_ctrl: lanes without a source add 0
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// ds_add_f32 of ONE lane (exec = `mask` for the one instruction) at an immediate offset of the table base: no VALU,
// no branch.  The code around it is wave-uniform, so exec is all ones before and after.
template <int OFF>
__device__ __forceinline__ void d3_lane_add(unsigned base, float v, unsigned long long mask) {
  asm volatile("s_mov_b64 exec, %2\n\tds_add_f32 %0, %1 offset:%3\n\ts_mov_b64 exec, -1"
               : : "v"(base), "v"(v), "s"(mask), "n"(OFF) : "memory");
}
template <int NE, int C0, int C1, int C2, int C3, int C4, int C5>
__device__ __forceinline__ void d3_reduce(const f32x4 &h, const float (&dv)[6][4], unsigned dcs) {
  constexpr unsigned long long L31 = 1ull << 31, L63 = 1ull << 63;
  constexpr int NP = NE / 2, NV = NP + (NE & 1);
  const f32x2 h01 = {h[0], h[1]}, h23 = {h[2], h[3]};
  float t[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    f32x2 pr = h01 * f32x2{dv[j][0], dv[j][1]};
    pr = __builtin_elementwise_fma(h23, f32x2{dv[j][2], dv[j][3]}, pr);
    t[j] = pr.x + pr.y;
  }
  float v[NV];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[2 * q]), __float_as_uint(t[2 * q + 1]), false, false);
    v[q] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  if (NE & 1) v[NP] = t[NE - 1];
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = d3_add_dpp<0xB1>(v[q]);      // quad_perm [1,0,3,2]
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = d3_add_dpp<0x4E>(v[q]);      // quad_perm [2,3,0,1]
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = d3_add_dpp<0x141>(v[q]);     // row_half_mirror
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = d3_add_dpp<0x140>(v[q]);     // row_mirror: every lane of a row holds the row sum
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = d3_add_dpp<0x142>(v[q]);     // row_bcast15: rows 1 / 3 += rows 0 / 2
  if (NE & 1) v[NP] = d3_add_dpp<0x143>(v[NP]);                    // row_bcast31: rows 2, 3 += lane 31: total in lane 63
  if (NP > 0) { d3_lane_add<4 * C0>(dcs, v[0], L31); d3_lane_add<4 * C1>(dcs, v[0], L63); }
  if (NP > 1) { d3_lane_add<4 * C2>(dcs, v[1], L31); d3_lane_add<4 * C3>(dcs, v[1], L63); }
  if (NP > 2) { d3_lane_add<4 * C4>(dcs, v[2], L31); d3_lane_add<4 * C5>(dcs, v[2], L63); }
  if (NE == 1) d3_lane_add<4 * C0>(dcs, v[0], L63);
  if (NE == 3) d3_lane_add<4 * C2>(dcs, v[1], L63);
  if (NE == 5) d3_lane_add<4 * C4>(dcs, v[2], L63);
}

#define D3_VISIT(set, plane, next, wrap, piece)                                      \
  {                                                                                  \
    load_a(aS[(set) ^ 1], next, (wrap) ? ((ph + 1) & 3) : ph);                        \
    if ((piece) >= 0 && copy) dma_piece(piece);                                      \
  }
#define D3_STEP(set, slot, ne, o0, c0, o1, c1, o2, c2, o3, c3, o4, c4, o5, c5)   \
  {                                                                              \
    float dv_[6][4];                                                             \
    d3_gather<ne, o0, o1, o2, o3, o4, o5>(xl, dv_);                              \
    __builtin_amdgcn_sched_barrier(0);                                           \
    d3_mfma16(h, aS[set], bz[slot]);                                             \
    __builtin_amdgcn_sched_barrier(0);                                           \
    d3_reduce<ne, c0, c1, c2, c3, c4, c5>(h, dv_, dcs_off);                      \
  }
#define D3_CONT(ne, o0, c0, o1, c1, o2, c2, o3, c3, o4, c4, o5, c5)   \
  {                                                                   \
    float dv_[6][4];                                                  \
    d3_gather<ne, o0, o1, o2, o3, o4, o5>(xl, dv_);                   \
    d3_reduce<ne, c0, c1, c2, c3, c4, c5>(h, dv_, dcs_off);        \
  }
#define D3_END(parity, pieces)                                                            \
  {                                                                                       \
    if (copy) { _Pragma("unroll") for (int i_ = pieces; i_ < D3_PW; ++i_) dma_piece(i_); } \
    if (parity) {                                                                         \
      _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) aS[0][e_] = aS[1][e_];            \
    }                                                                                     \
  }

template <int WAVE>
__device__ __forceinline__ void d3_wave_main(const D3Params &p, float *lds, const float *__restrict__ x,
                                             const float *__restrict__ dz, const float *__restrict__ Wp) {
  constexpr int V = D3_V, RS = D3_RS, BUF = D3_BUF, NW = D3_NW, SLOTS = D3_SLOTS;
  constexpr int wave = WAVE;
  // [ltot][V] accumulated gradient of this workgroup, addressed off a VGPR base the compiler cannot fold (a known
  // base makes every entry address its own hoisted scalar constant: hundreds of spilled SGPRs)
  unsigned dcs_off = (unsigned)(2 * BUF * sizeof(float));
  asm volatile("" : "+v"(dcs_off));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int g = lane >> 4, r = lane & 15;
  constexpr const int (&sj)[SLOTS] = d3_slot_joints[WAVE];

  const size_t row_stride = (size_t)p.T * V;
  const char *xl0 = reinterpret_cast<const char *>(lds + g * RS + r * V);   // lane's gather base (LDS row g, frame r)

  // this wave's DMA pieces of a 16-row slice; LDS row l = 4 q + g holds slice row 4 g + q
  int doff[D3_PW];
#pragma unroll
  for (int i = 0; i < D3_PW; ++i) {
    const int pc = i * NW + wave;
    const int e = pc * 64 + lane;
    const int lrow = e / (RS / 4), c4 = e - lrow * (RS / 4);
    const int row = 4 * (lrow & 3) + (lrow >> 2);
    doff[i] = (pc < D3_PIECES && e < D3_NV4) ? (int)(((size_t)row * row_stride + 4 * c4) * sizeof(float)) : -1;
  }
  float bz[SLOTS][16];                                // X[4 kk + g][frame r][joint of the slot]
  float aS[2][16];                                    // two A-operand sets: W_k[16 p + r][4 kk + g], kk = 0..15
  f32x4 h;
  auto load_a = [&](float (&a)[16], int k, int ph) {
    // forward planes in kernel order: Wp[k][ph'][m][lane][s] = W_k[16 m + r][16 ph' + 4 s + g]; here m = ph, kk = 4 ph' + s
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const float4 u = reinterpret_cast<const float4 *>(Wp)[((size_t)(k * 4 + pp) * 4 + ph) * 64 + lane];
      a[4 * pp + 0] = u.x; a[4 * pp + 1] = u.y; a[4 * pp + 2] = u.z; a[4 * pp + 3] = u.w;
    }
  };

  int tile = blockIdx.x;
  if (tile < p.total_tiles) {       // prologue: slice 0 of the first tile, A operands of the first plane
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * D3_F;
    const float *dr = dz + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
#pragma unroll
    for (int i = 0; i < D3_PW; ++i)
      if (doff[i] >= 0) d3_dma16(dr, doff[i], lds + (i * NW + wave) * 256);
  }
  load_a(aS[0], d3_plane0[WAVE], 0);

  for (; tile < p.total_tiles; tile += gridDim.x) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * D3_F;
    const float *dg = dz + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    const int ntile = tile + gridDim.x;
    const bool has_next = ntile < p.total_tiles;
    const int nseq = has_next ? ntile / p.tiles_per_seq : 0, nt0 = has_next ? (ntile % p.tiles_per_seq) * D3_F : 0;
    const float *ndg = dz + (size_t)nseq * 64 * row_stride + (size_t)nt0 * V;
    D3_TRACE_TILE(tile);
    D3_MARK(0);

    // B operands of the tile: X[4 kk + g][frame r][joint of the slot] for all 64 input channels, register-resident for
    // the whole tile.  They come through LDS: buffer 1 is free between the last phase of one tile and the second of
    // the next, and the four 16-channel slices of X pass through it one after the other as LDS-DMA pieces (whole
    // 1 KB rows per instruction, the layout of the dZ slices: LDS row 4 g + kappa holds slice row 4 kappa + g), each
    // read back with 28 immediate-offset ds_reads per lane.  Loaded straight from global memory (a 16- or 12-byte run
    // of joints per lane and channel), a load instruction touched ~100 cache lines for 1 KB of data and the 32 of
    // them per wave kept the address path of the CU busy for 15-37 thousand cycles per tile with nothing else to
    // run (cycle trace, round 3).
    {
      float *xb = lds + BUF;
      const char *xr = reinterpret_cast<const char *>(xb + 4 * g * RS + r * V);      // kappa = 0: + kappa * RS floats
      const float *xs = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                    // nobody reads buffer 1 any more
#pragma unroll
        for (int i = 0; i < D3_PW; ++i)
          if (doff[i] >= 0) d3_dma16(xs + (size_t)sl * 16 * row_stride, doff[i], xb + (i * NW + wave) * 256);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SLOTS; ++i)
          if (sj[i] >= 0) {
#pragma unroll
            for (int kp = 0; kp < 4; ++kp)
              bz[i][4 * sl + kp] = *reinterpret_cast<const float *>(xr + (kp * RS + sj[i]) * 4);
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the phase-0 barrier follows: buffer 1 is written next
    }

#pragma unroll 1
    for (int ph = 0; ph < 4; ++ph) {
      D3_MARK(1 + 3 * ph);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // own pieces of slice `ph` (and, first phase, the B operands)
      D3_MARK(2 + 3 * ph);
      __syncthreads();
      D3_MARK(3 + 3 * ph);
      float *buf_nxt = lds + ((ph + 1) & 1) * BUF;
      const char *xl = xl0 + (ph & 1) * BUF * sizeof(float);
      const bool copy = ph + 1 < 4 || has_next;
      const float *src = (ph + 1 < 4) ? dg + (size_t)(ph + 1) * 16 * row_stride : ndg;
      auto dma_piece = [&](int i) {
        if (doff[i] >= 0) d3_dma16(src, doff[i], buf_nxt + (i * NW + wave) * 256);
      };
      if constexpr (WAVE == 0) { D3_BODY_0 } else if constexpr (WAVE == 1) { D3_BODY_1 }
      else if constexpr (WAVE == 2) { D3_BODY_2 } else if constexpr (WAVE == 3) { D3_BODY_3 }
      else if constexpr (WAVE == 4) { D3_BODY_4 } else if constexpr (WAVE == 5) { D3_BODY_5 }
      else if constexpr (WAVE == 6) { D3_BODY_6 } else { D3_BODY_7 }
    }
    D3_MARK(13);
  }
}

__global__ __launch_bounds__(D3_NW * 64, 2) void gcn3_dcoef_kernel(D3Params p, const float *__restrict__ x,
                                                                   const float *__restrict__ dz,
                                                                   const float *__restrict__ Wp,
                                                                   float *__restrict__ dcoef_partial) {
  extern __shared__ float lds[];
  float *dcs = lds + 2 * D3_BUF;
  const int tid = threadIdx.x;
  for (int e = tid; e < p.ltot * D3_V; e += D3_NW * 64) dcs[e] = 0.f;
  __syncthreads();
  switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
    case 0: d3_wave_main<0>(p, lds, x, dz, Wp); break;
    case 1: d3_wave_main<1>(p, lds, x, dz, Wp); break;
    case 2: d3_wave_main<2>(p, lds, x, dz, Wp); break;
    case 3: d3_wave_main<3>(p, lds, x, dz, Wp); break;
    case 4: d3_wave_main<4>(p, lds, x, dz, Wp); break;
    case 5: d3_wave_main<5>(p, lds, x, dz, Wp); break;
    case 6: d3_wave_main<6>(p, lds, x, dz, Wp); break;
    default: d3_wave_main<7>(p, lds, x, dz, Wp); break;
  }
  __syncthreads();
  float *out = dcoef_partial + (size_t)blockIdx.x * p.ltot * D3_V;
  for (int e = tid; e < p.ltot * D3_V; e += D3_NW * 64) out[e] = dcs[e];
}

}  // namespace

// Adjacency gradient at the row-list entries, statically scheduled for the P2RNet skeleton (the caller checks
// p2r_stgcn_gcn3_signature(1) against its row tables first).
//   x   (N,64,T,53): input of the graph conv            dz (N,64,T,53): gradient of its output
//   Wp  [K][4][4][64][4]: the forward planes in kernel order (as for p2r_stgcn_gcn3_forward, form 0)
//   dcoef_partial [n_blocks][ltot][53]: per-workgroup sums in the layout of the row coefficient table
//     (entry (lofs_k + j, v) <-> A[k][v][w_j(k, v)]); padded slots stay 0; the caller sums over the leading axis.
// T % 16 == 0 and x, dz 16-byte aligned (P2R_EINVAL otherwise: use p2r_stgcn_gcn_coef_grad).
extern "C" int p2r_stgcn_gcn3_coef_grad(int N, int T, int V, int K, int ltot, const float *x, const float *dz,
                                        const float *Wp, int n_blocks, float *dcoef_partial, void *stream) {
  if (N < 0 || T <= 0 || V != D3_V || K != G3_K || ltot <= 0 || n_blocks < 1) return P2R_EINVAL;
  if (T % D3_F != 0 || T > (1 << 20) || ((uintptr_t)dz % 16) != 0 || ((uintptr_t)x % 4) != 0) return P2R_EINVAL;
  if (N == 0) return hipMemsetAsync(dcoef_partial, 0, (size_t)n_blocks * ltot * V * sizeof(float), p2r_stream(stream));
  D3Params p;
  p.T = T; p.ltot = ltot;
  p.tiles_per_seq = T / D3_F;
  const long long tiles = (long long)N * p.tiles_per_seq;
  if (tiles > 0x7fffffffLL) return P2R_EINVAL;
  p.total_tiles = (int)tiles;
  const size_t lds = (size_t)2 * D3_BUF * sizeof(float) + (size_t)ltot * V * sizeof(float);
  if (lds > 160 * 1024) return P2R_EINVAL;
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(gcn3_dcoef_kernel, lds_ok);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(gcn3_dcoef_kernel, dim3(n_blocks), dim3(D3_NW * 64), lds, p2r_stream(stream), p, x, dz, Wp,
                     dcoef_partial);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
