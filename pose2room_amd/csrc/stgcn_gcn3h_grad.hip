// stgcn_gcn3h_grad.hip -- adjacency gradient of the fused graph convolution in `split16` arithmetic (opt-in mode;
// split16.h), statically scheduled (gfx950).  Same operator, work list and data movement as stgcn_gcn3_grad.hip:
//
//   dcoef[k][j][v] = sum over (n, t, c) of Y_k[c, t, v] * dZ[c, t, w_j(k, v)],     Y_k = W_k . X
// at the non-zero entries (k, v, w_j) of the adjacency (reference models/p2rnet/modules/stgcn.py:134 /
// stgcn_layers.py:62-65 through autograd).  What changes is the MFMA product Y_k(v) = W_k . X(v) -- 16 output rows x 16
// frames, K = the 64 input channels: SIX v_mfma_f32_16x16x32_f16 (two k-steps x three products of two-part fp16 operands,
// 96 matrix-pipe cycles) instead of SIXTEEN v_mfma_f32_16x16x4_f32 (512 cycles).  The reduction of the product against
// the gathered dZ rows stays what it was -- fp32 vector arithmetic on the fp32 dZ slice, so the heavy tail of the
// gradient never meets fp16 -- and now runs BESIDE the matrix pipe instead of in its issue slots: on gfx950 an fp32 MFMA
// issues through the vector datapath, a 16-bit one does not (tools/ubench/mfma16_valu_overlap.hip).  The exact kernel
// spent 21 % of its wave time waiting with the matrix pipe 64 % busy; here the vector reduction is the critical path.
//   * B operands (X of the wave's joints, register-resident for a tile) are converted to fp16 parts when they are read
//     from the staging slices: 16 registers per joint, as before.  K index (kg, i) of a k-step ks <-> channel
//     32 ks + 16 (i >> 2) + 4 (i & 3) + kg: the slices' LDS layout and read pattern of the exact kernel serve unchanged.
//   * A operands arrive pre-split from the host (prepare_chain) in that K order: Wd[k][ph][part][ks][lane][8].
//   * X is scaled by the power of two of its range word when it is converted, W by its own; the products are scaled back
//     when the workgroup's table is written out.
// MFMAs through the builtin (the exact kernel's two assembly blocks are gone).
#include "p2r_common.h"
#include "split16.h"

#include "gcn3_sched.inc"

// (d3_reduce indexes small arrays behind compile-time conditions that the front end does not fold before it warns)
#pragma clang diagnostic ignored "-Warray-bounds"

#define D3_TRACE_TILE(tile)
#define D3_MARK(i)
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
  const unsigned *x_amax;
  const float *winv;
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

// Y tile of one (plane, joint) unit: two k-steps (all 64 input channels) x three products into ONE accumulator
struct D3A { p2r_h8 p[2], q[2]; };            // [k-step]: parts of W_k rows 16 ph + r
struct D3B { p2r_h8 p[2], q[2]; };            // [k-step]: parts of X (channels of the k-step, frame r, the slot's joint)
__device__ __forceinline__ void d3_mfma6(f32x4 &h, const D3A &a, const D3B &b) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.p[ks], b.q[ks], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.q[ks], b.p[ks], acc, 0, 0, 0);
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.p[ks], b.p[ks], acc, 0, 0, 0);
  h = acc;
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
#ifdef D3_NO_FENCE
#define D3_FENCE
#else
#define D3_FENCE __builtin_amdgcn_sched_barrier(0);
#endif
#define D3_STEP(set, slot, ne, o0, c0, o1, c1, o2, c2, o3, c3, o4, c4, o5, c5)   \
  {                                                                              \
    float dv_[6][4];                                                             \
    d3_gather<ne, o0, o1, o2, o3, o4, o5>(xl, dv_);                              \
    D3_FENCE                                                                     \
    d3_mfma6(h, aS[set], bz[slot]);                                             \
    D3_FENCE                                                                     \
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
    if (parity) aS[0] = aS[1];                                                            \
  }

template <int WAVE>
__device__ __forceinline__ void d3_wave_main(const D3Params &p, float *lds, const float *__restrict__ x,
                                             const float *__restrict__ dz, const p2r_h8 *__restrict__ Wd, float xscale) {
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
  D3B bz[SLOTS];                                      // fp16 parts of X[channels of the k-step][frame r][joint of the slot]
  D3A aS[2];                                          // two A-operand sets: parts of W_k[16 p + r][channels of the k-step]
  f32x4 h;
  auto load_a = [&](D3A &a, int k, int ph) {
    // Wd[k][ph][part][ks][lane] (16 bytes each)
    const p2r_h8 *wd = Wd + ((size_t)(k * 4 + ph) * 2 * 2) * 64 + lane;
    a.p[0] = wd[0]; a.p[1] = wd[64]; a.q[0] = wd[128]; a.q[1] = wd[192];
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
            // slice sl = channels 16 sl + 4 kp + g: K values i = 4 (sl & 1) + kp of k-step sl >> 1 (see the top)
            float v4[4];
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) v4[kp] = *reinterpret_cast<const float *>(xr + (kp * RS + sj[i]) * 4) * xscale;
#pragma unroll
            for (int kp = 0; kp < 4; kp += 2) {
              p2r_f2 xx = {v4[kp], v4[kp + 1]};
              asm volatile("" : "+v"(xx));                  // the split sees VALUES (split16.h)
              const p2r_h2 ph_ = __builtin_convertvector(xx, p2r_h2);
              const p2r_h2 qh_ = __builtin_convertvector(xx - __builtin_convertvector(ph_, p2r_f2), p2r_h2);
              bz[i].p[sl >> 1][4 * (sl & 1) + kp] = ph_.x; bz[i].p[sl >> 1][4 * (sl & 1) + kp + 1] = ph_.y;
              bz[i].q[sl >> 1][4 * (sl & 1) + kp] = qh_.x; bz[i].q[sl >> 1][4 * (sl & 1) + kp + 1] = qh_.y;
            }
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

__global__ __launch_bounds__(D3_NW * 64, 2) void gcn3h_dcoef_kernel(D3Params p, const float *__restrict__ x,
                                                                    const float *__restrict__ dz,
                                                                    const p2r_h8 *__restrict__ Wd,
                                                                    float *__restrict__ dcoef_partial) {
  extern __shared__ float lds[];
  float *dcs = lds + 2 * D3_BUF;
  const int tid = threadIdx.x;
  for (int e = tid; e < p.ltot * D3_V; e += D3_NW * 64) dcs[e] = 0.f;
  float xs, xinv;
  p2r_split_scale(p.x_amax, xs, xinv);
  const float inv = xinv * p.winv[0];          // the products carry 2^(S_x + S_w)
  __syncthreads();
  switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
    case 0: d3_wave_main<0>(p, lds, x, dz, Wd, xs); break;
    case 1: d3_wave_main<1>(p, lds, x, dz, Wd, xs); break;
    case 2: d3_wave_main<2>(p, lds, x, dz, Wd, xs); break;
    case 3: d3_wave_main<3>(p, lds, x, dz, Wd, xs); break;
    case 4: d3_wave_main<4>(p, lds, x, dz, Wd, xs); break;
    case 5: d3_wave_main<5>(p, lds, x, dz, Wd, xs); break;
    case 6: d3_wave_main<6>(p, lds, x, dz, Wd, xs); break;
    default: d3_wave_main<7>(p, lds, x, dz, Wd, xs); break;
  }
  __syncthreads();
  float *out = dcoef_partial + (size_t)blockIdx.x * p.ltot * D3_V;
  for (int e = tid; e < p.ltot * D3_V; e += D3_NW * 64) out[e] = dcs[e] * inv;
}

}  // namespace

// Adjacency gradient at the row-list entries in split16 arithmetic: arguments and result of p2r_stgcn_gcn3_coef_grad with
//   Wd     fp16 [K][4 ph][2 parts][2 ks][64 lanes][8]: the parts of 2^S_w W_k (forward planes) in A-operand order,
//          Wd[k][ph][part][ks][16 kg + r][i] = part of 2^S_w W_k[16 ph + r][32 ks + 16 (i >> 2) + 4 (i & 3) + kg]
//   winv   device float 2^-S_w;   x_amax: range word of x (NULL: scale 1).  dz is used in fp32 (no range word).
extern "C" int p2r_stgcn_gcn3h_coef_grad(int N, int T, int V, int K, int ltot, const float *x, const float *dz,
                                         const void *Wd, const float *winv, int n_blocks, float *dcoef_partial,
                                         const unsigned *x_amax, void *stream) {
  if (N < 0 || T <= 0 || V != D3_V || K != G3_K || ltot <= 0 || n_blocks < 1 || !Wd || !winv) return P2R_EINVAL;
  if (T % D3_F != 0 || T > (1 << 19) || ((uintptr_t)dz % 16) != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)Wd % 16) != 0)
    return P2R_EINVAL;
  if (N == 0) return hipMemsetAsync(dcoef_partial, 0, (size_t)n_blocks * ltot * V * sizeof(float), p2r_stream(stream));
  D3Params p;
  p.T = T; p.ltot = ltot;
  p.tiles_per_seq = T / D3_F;
  p.x_amax = x_amax; p.winv = winv;
  const long long tiles = (long long)N * p.tiles_per_seq;
  if (tiles > 0x7fffffffLL) return P2R_EINVAL;
  p.total_tiles = (int)tiles;
  const size_t lds = (size_t)2 * D3_BUF * sizeof(float) + (size_t)ltot * V * sizeof(float);
  if (lds > 160 * 1024) return P2R_EINVAL;
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(gcn3h_dcoef_kernel, lds_ok);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(gcn3h_dcoef_kernel, dim3(n_blocks), dim3(D3_NW * 64), lds, p2r_stream(stream), p, x, dz,
                     reinterpret_cast<const p2r_h8 *>(Wd), dcoef_partial);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
