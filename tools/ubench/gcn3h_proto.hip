// PROTOTYPE, not part of the product (libp2r_hip.so does not contain it, bench.py does not run it): the FORWARD of the
// fused graph convolution  Z(w) = bias(w) + sum_k W_k . (X . A_k)(w)  on two-part fp16 MFMA products with
// K = 32 = 16 channels x TWO PLANES per v_mfma_f32_16x16x32_f16 (DESIGN.md section 5 "Round 5").  Skeleton of
// pose2room_amd/csrc/stgcn_gcn3.hip (persistent workgroups of 8 straight-line wave programs, 16-frame tiles, four
// 16-channel slices by LDS-DMA into two buffers, accumulators of up to 7 joints per wave in registers, staged whole-row
// stores); what changes is the step:
//   * a step = a (plane pair, output joint) unit.  K index k = 8 kg + i of lane (kg, r): frame r, channel kg + 4 (i & 3) of
//     the slice, plane a of the pair for i < 4, plane b for i >= 4 -- every lane walks BOTH neighbour lists with four
//     channels each: LDS offsets and coefficient indices are immediates common to the wave, nobody multiplies by zero;
//   * the aggregate is split into two fp16 parts x = x1 + x2 (x2 unscaled: its absolute error, 3e-8, is what counts
//     behind a BatchNorm + ReLU), W comes pre-split from the host as 2^S W = w1 + w2: three MFMAs per row tile
//     (w1 x2, w2 x1, w1 x1) into ONE accumulator at scale 2^S;
//   * 253 units per tile and phase instead of 454, 12 MFMAs of 16 cycles each instead of 16 of 32;
//   * software pipeline: split(u), loads of unit u + 1, the 12 MFMAs of u, multiply-adds of u + 1.
// Forward only, no statistics epilogue.  Schedule: tools/gen_gcn_pair_sched.py -> gcn3h_sched.inc.
#include <hip/hip_runtime.h>
#include <cstdint>

#ifdef H3_FORM_R      // the data gradient: row lists, W_k^T (same kernel, other schedule)
#include "gcn3h_sched_r.inc"
#else
#include "gcn3h_sched.inc"
#endif
#ifdef ABL_NO_SPLIT     // timing ablation: the aggregate's bits go to the MFMAs unsplit
#define ABL_SPLIT 0
#else
#define ABL_SPLIT 1
#endif
#ifdef ABL_NO_LOAD_A    // timing ablation: the A operands are loaded once
#define ABL_LOAD_A 0
#else
#define ABL_LOAD_A 1
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int V = 53, F = 16, CP = 16, NPH = 4, NW = 8, SLOTS = 7;
constexpr int RS = F * V;            // 848
constexpr int BUF = CP * RS;
constexpr int NV4 = BUF / 4;
constexpr int PIECES = (NV4 + 63) / 64;          // 53
constexpr int PW = (PIECES + NW - 1) / NW;       // 7
constexpr int slot_joints[NW][SLOTS] = H3_SLOT_JOINTS;
constexpr int plane0[NW] = {H3_PLANE0_0, H3_PLANE0_1, H3_PLANE0_2, H3_PLANE0_3, H3_PLANE0_4, H3_PLANE0_5, H3_PLANE0_6, H3_PLANE0_7};

struct Params { int T, tiles_per_seq, total_tiles; float scale, inv_scale; unsigned long long *prof; float *stats; const float *addend; };

__device__ __forceinline__ unsigned lds_addr(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
__device__ __forceinline__ void dma16(const float *base, int voff, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// Sum of v over the 16 lanes of a DPP row (csrc/p2r_common.h)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));  // row_mirror
  return v;
}
constexpr int ST = 3;      // floats per (wave, row) statistics entry: (sum, sum of squares) about the pivot, pivot -- at the accumulators' scale

// 12 MFMAs of a unit as ONE assembly block that accumulates in place ("+v" pins a tile to its registers for the whole
// kernel; through the builtin the compiler renames the accumulators along every chain and reconciles the eight wave
// programs through scratch: 409-733 spilled registers).  What the compiler does not do for assembly: the leading s_nop
// covers VALU write -> MFMA read; an in-flight v_mfma_f32_16x16x32_f16 still reads its four-register B operand after it
// has issued, so the NEXT unit's split must not write the same registers right behind the block (measured with one
// register set: 0.2 of range wrong in a few joints) -- the B parts alternate between two register sets by unit parity.
__device__ __forceinline__ void mfma12(f32x4 (&acc)[4], const h8 (&a1)[4], const h8 (&a2)[4], const h8 &b1, const h8 &b2) {
#ifdef ABL_NO_MFMA      // timing ablation: the operands stay live, no matrix instruction
  asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
               : "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(a2[0]), "v"(a2[1]), "v"(a2[2]), "v"(a2[3]), "v"(b1), "v"(b2));
  return;
#endif
#ifdef MFMA_BUILTIN
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[m], b2, acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[m], b1, acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[m], b1, acc[m], 0, 0, 0);
  return;
#endif
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %4, %13, %0\n\tv_mfma_f32_16x16x32_f16 %1, %5, %13, %1\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %6, %13, %2\n\tv_mfma_f32_16x16x32_f16 %3, %7, %13, %3\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %8, %12, %0\n\tv_mfma_f32_16x16x32_f16 %1, %9, %12, %1\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %10, %12, %2\n\tv_mfma_f32_16x16x32_f16 %3, %11, %12, %3\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %4, %12, %0\n\tv_mfma_f32_16x16x32_f16 %1, %5, %12, %1\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %6, %12, %2\n\tv_mfma_f32_16x16x32_f16 %3, %7, %12, %3"
#ifdef TRAIL_NOPS
      "\n\ts_nop 15\n\ts_nop 3"     // MFMA write -> a spill store reading the accumulator (the compiler does not know)
#endif
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
      : "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(a2[0]), "v"(a2[1]), "v"(a2[2]), "v"(a2[3]), "v"(b1), "v"(b2));
}

// x = x1 + x2 for a pair of values: x1 by v_cvt_pk_f16_f32, x2 = fp16(x - x1) by v_fma_mix (fp32 arithmetic on the fp16
// source, one instruction per value instead of convert back + subtract + convert)
__device__ __forceinline__ void split2(float x0, float x1, unsigned &p, unsigned &r) {
  const h2 ph = __builtin_convertvector(f2{x0, x1}, h2);
  p = __builtin_bit_cast(unsigned, ph);
#ifdef SPLIT_C
  r = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{x0, x1} - __builtin_convertvector(ph, f2), h2));
  return;
#endif
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(r) : "v"(x0), "v"(x1), "v"(p));
}

// A chunk of a unit's entry list.  Loads: xv[j][i] = X[channel kg + 4 i][frame r][joint of entry j] (rows of
// neighbouring channels sit 16 banks apart: the two 16-lane groups of a 32-lane LDS access do not collide), c[j] = the
// entry's coefficient (uniform address: a broadcast read).  Combine: entry j belongs to plane half h_j; f_j = first
// entry of that half (multiply instead of multiply-add); ZERO bit h = the half has no entry in this unit.
template <int NE, int O0, int C0, int O1, int C1, int O2, int C2, int O3, int C3>
__device__ __forceinline__ void h3_gather(const char *xl, const char *cl, float (&xv)[4][4], float (&c)[4]) {
  constexpr int off[4] = {O0, O1, O2, O3}, ci[4] = {C0, C1, C2, C3};
#pragma unroll
  for (int j = 0; j < NE; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) xv[j][i] = *reinterpret_cast<const float *>(xl + off[j] + i * (4 * RS * 4));
    c[j] = *reinterpret_cast<const float *>(cl + 4 * ci[j]);
  }
}
template <int NE, int ZERO, int H0, int F0, int H1, int F1, int H2, int F2, int H3, int F3>
__device__ __forceinline__ void h3_combine(const float (&xv)[4][4], const float (&c)[4], float (&x)[8]) {
  constexpr int h[4] = {H0, H1, H2, H3}, f[4] = {F0, F1, F2, F3};
  if (ZERO & 1) { x[0] = 0.f; x[1] = 0.f; x[2] = 0.f; x[3] = 0.f; }
  if (ZERO & 2) { x[4] = 0.f; x[5] = 0.f; x[6] = 0.f; x[7] = 0.f; }
#pragma unroll
  for (int j = 0; j < NE; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) x[4 * h[j] + i] = f[j] ? c[j] * xv[j][i] : fmaf(c[j], xv[j][i], x[4 * h[j] + i]);
}

#define H3_PIECE(piece) { if (copy) dma_piece(piece); }
#define H3_VISIT(pair) { if (ABL_LOAD_A) load_a(aS, pair, ph); }
#ifdef ABL_NO_GATHER    // timing ablation: no LDS gathers, no multiply-adds
#define H3_G(k, ne, o0, c0, o1, c1, o2, c2, o3, c3)
#define H3_C(k, ne, zero, h0, f0, h1, f1, h2, f2, h3, f3) { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) asm volatile("" : "+v"(xagg[i_])); }
#else
#define H3_G(k, ne, o0, c0, o1, c1, o2, c2, o3, c3) h3_gather<ne, o0, c0, o1, c1, o2, c2, o3, c3>(xl, cl, xv_[k], cf_[k]);
#define H3_C(k, ne, zero, h0, f0, h1, f1, h2, f2, h3, f3) h3_combine<ne, zero, h0, f0, h1, f1, h2, f2, h3, f3>(xv_[k], cf_[k], xagg);
#endif
#define H3_S(par)                                                                         \
  {                                                                                       \
    unsigned p_[4], r_[4];                                                                \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                    \
      if (ABL_SPLIT) split2(xagg[2 * q_], xagg[2 * q_ + 1], p_[q_], r_[q_]);             \
      else { p_[q_] = __builtin_bit_cast(unsigned, xagg[2 * q_]); r_[q_] = __builtin_bit_cast(unsigned, xagg[2 * q_ + 1]); } \
    }                                                                                     \
    b1_[par] = __builtin_bit_cast(h8, u4{p_[0], p_[1], p_[2], p_[3]});                     \
    b2_[par] = __builtin_bit_cast(h8, u4{r_[0], r_[1], r_[2], r_[3]});                     \
  }
#define H3_M(slot, par)                                       \
  {                                                           \
    __builtin_amdgcn_sched_barrier(0);                        \
    mfma12(acc[slot], aS[0], aS[1], b1_[par], b2_[par]);      \
    __builtin_amdgcn_sched_barrier(0);                        \
  }
// interleave mode (GEN_MODE=interleave): single instructions, placed by the generator between the MFMAs of the unit in
// front and fenced (the compiler neither clusters the MFMAs nor hoists the vector work: sched_group_barrier patterns
// were ignored here).  MFMAs through the builtin: the compiler sees every hazard.
#define H3_F __builtin_amdgcn_sched_barrier(0);
#define H3_MF(slot, par, term, m)                                                                                           \
  acc[slot][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aS[(term) == 1][m], (term) == 0 ? b2_[par] : b1_[par], acc[slot][m], 0, 0, 0);
#define H3_GC(k, j, ci) cf_[k][j] = *reinterpret_cast<const float *>(cl + 4 * (ci));
#define H3_GX(k, j, i, off) xv_[k][j][i] = *reinterpret_cast<const float *>(xl + (off) + (i) * (4 * RS * 4));
#define H3_C1(k, j, i, h, f) xagg[4 * (h) + (i)] = (f) ? cf_[k][j] * xv_[k][j][i] : fmaf(cf_[k][j], xv_[k][j][i], xagg[4 * (h) + (i)]);
#define H3_Z1(zero)                                                              \
  {                                                                              \
    if ((zero) & 1) { xagg[0] = 0.f; xagg[1] = 0.f; xagg[2] = 0.f; xagg[3] = 0.f; } \
    if ((zero) & 2) { xagg[4] = 0.f; xagg[5] = 0.f; xagg[6] = 0.f; xagg[7] = 0.f; } \
  }
#define H3_S1(par, q)                                                            \
  {                                                                              \
    unsigned p_, r_;                                                             \
    split2(xagg[2 * (q)], xagg[2 * (q) + 1], p_, r_);                            \
    const h2 ph_ = __builtin_bit_cast(h2, p_), rh_ = __builtin_bit_cast(h2, r_); \
    b1_[par][2 * (q)] = ph_.x; b1_[par][2 * (q) + 1] = ph_.y;                    \
    b2_[par][2 * (q)] = rh_.x; b2_[par][2 * (q) + 1] = rh_.y;                    \
  }
#define H3_M2(slot, par, set)                                 \
  {                                                           \
    __builtin_amdgcn_sched_barrier(0);                        \
    mfma12(acc[slot], aS2[set][0], aS2[set][1], b1_[par], b2_[par]); \
    __builtin_amdgcn_sched_barrier(0);                        \
  }
#define H3_VISIT2(set, pair, wrap) { load_a(aS2[set], pair, (wrap) ? ((ph + 1) & (NPH - 1)) : ph); }
// two A register sets by visit parity; a phase with an odd number of visits leaves the next phase's first set in [1]
#define H3_END2(pieces, parity)                                                           \
  {                                                                                       \
    if (copy) { _Pragma("unroll") for (int i_ = pieces; i_ < PW; ++i_) dma_piece(i_); }   \
    if (parity) {                                                                         \
      _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) aS2[0][q_][m_] = aS2[1][q_][m_]; \
    }                                                                                     \
  }
#define H3_END(pieces, pair0)                                                             \
  {                                                                                       \
    if (copy) { _Pragma("unroll") for (int i_ = pieces; i_ < PW; ++i_) dma_piece(i_); }   \
    load_a(aS, pair0, (ph + 1) & (NPH - 1));                                              \
  }

// -DPROFILE: s_memtime stamps of every wave's lane 0 -> prof[block][wave][4] = cycles (waiting at the phase start,
// phase bodies, between the last body and the epilogue, epilogue)
#ifdef PROFILE
#define PROF_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); \
    if (i == 1) pw_ += t_ - tl_; if (i == 2) pb_ += t_ - tl_; if (i == 3) pg_ += t_ - tl_; if (i == 4) pe_ += t_ - tl_; tl_ = t_; }
#define PROF_OUT { if (lane == 0 && p.prof) { unsigned long long *o_ = p.prof + ((size_t)blockIdx.x * NW + WAVE) * 4; \
    o_[0] = pw_; o_[1] = pb_; o_[2] = pg_; o_[3] = pe_; } }
#else
#define PROF_T(i)
#define PROF_OUT
#endif

template <int WAVE>
__device__ __forceinline__ void wave_main(const Params &p, float *lds, const float *__restrict__ x,
                                          const h8 *__restrict__ Wp, float *__restrict__ z) {
  constexpr int wave = WAVE;
  float *bias_l = lds + 2 * BUF;                       // [64][V]
  float *coef_l = bias_l + 64 * V;                     // [ltot + 1][V] (last row zeros)
  float *rowstat = coef_l + (H3_LTOT + 1) * V;         // [NW][64][ST]
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = lane >> 4, r = lane & 15;
  constexpr const int (&sj)[SLOTS] = slot_joints[WAVE];
  const size_t row_stride = (size_t)p.T * V;
  const char *xl0 = reinterpret_cast<const char *>(lds + g * RS + r * V);   // channels g + 4 i, frame r
  unsigned cl_off = (unsigned)((coef_l - lds) * sizeof(float));
  asm volatile("" : "+v"(cl_off));                     // opaque base: see stgcn_gcn3.hip
  const char *cl = reinterpret_cast<const char *>(lds) + cl_off;

  int doff[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int pc = i * NW + wave, e = pc * 64 + lane;
    const int row = e / (RS / 4), c4 = e - row * (RS / 4);
    doff[i] = (pc < PIECES && e < NV4) ? (int)(((size_t)row * row_stride + 4 * c4) * sizeof(float)) : -1;
  }

#ifdef PROFILE
  unsigned long long tl_ = __builtin_readcyclecounter(), pw_ = 0, pb_ = 0, pg_ = 0, pe_ = 0;
#endif
  f32x4 acc[SLOTS][4];
#if H3_ASETS == 2
  h8 aS2[2][2][4];                                     // [set][part][m]
  h8 (&aS)[2][4] = aS2[0];
#else
  h8 aS[2][4];                                         // [part][m]: [W_a | W_b] rows 16 m + r, this lane's 8 k values
#endif
  float xagg[8], xv_[5][4][4], cf_[5][4];
  h8 b1_[2], b2_[2];
  auto load_a = [&](h8 (&a)[2][4], int pair, int ph) {   // one register set: see tools/gen_gcn_pair_sched.py
    // Wp[pair][ph][part][m][lane] (16 bytes each)
    const h8 *wp = Wp + ((size_t)(pair * NPH + ph) * 2 * 4) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int m = 0; m < 4; ++m) a[q][m] = wp[(q * 4 + m) * 64];
  };

  int tile = blockIdx.x;
  if (tile < p.total_tiles) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * F;
    const float *xr = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
#pragma unroll
    for (int i = 0; i < PW; ++i)
      if (doff[i] >= 0) dma16(xr, doff[i], lds + (i * NW + wave) * 256);
  }
  load_a(aS, plane0[WAVE], 0);

  for (; tile < p.total_tiles; tile += gridDim.x) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * F;
    const float *xg = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    float *zg = z + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    const int ntile = tile + gridDim.x;
    const bool has_next = ntile < p.total_tiles;
    const int nseq = has_next ? ntile / p.tiles_per_seq : 0, nt0 = has_next ? (ntile % p.tiles_per_seq) * F : 0;
    const float *nxg = x + (size_t)nseq * 64 * row_stride + (size_t)nt0 * V;

    // accumulators start from the bias table, at the accumulators' scale 2^S
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const float *bl = bias_l + 4 * g * V + (sj[i] >= 0 ? sj[i] : 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][m][q] = bl[(16 * m + q) * V] * p.scale;
    }

#pragma unroll 1
    for (int ph = 0; ph < NPH; ++ph) {
      PROF_T(0)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      PROF_T(1)
      float *buf_nxt = lds + ((ph + 1) & 1) * BUF;
      const char *xl = xl0 + (ph & 1) * BUF * sizeof(float);
      const bool copy = ph + 1 < NPH || has_next;
      const float *src = (ph + 1 < NPH) ? xg + (size_t)(ph + 1) * CP * row_stride : nxg;
      auto dma_piece = [&](int i) {
        if (doff[i] >= 0) dma16(src, doff[i], buf_nxt + (i * NW + wave) * 256);
      };
      if constexpr (WAVE == 0) { H3_BODY_0 } else if constexpr (WAVE == 1) { H3_BODY_1 }
      else if constexpr (WAVE == 2) { H3_BODY_2 } else if constexpr (WAVE == 3) { H3_BODY_3 }
      else if constexpr (WAVE == 4) { H3_BODY_4 } else if constexpr (WAVE == 5) { H3_BODY_5 }
      else if constexpr (WAVE == 6) { H3_BODY_6 } else { H3_BODY_7 }
      PROF_T(2)
    }

    // statistics of the stored values for the BatchNorm that follows (stgcn_gcn3.hip's epilogue, on the SCALED accumulators:
    // sums about a pivot per (wave, row), two rows per packed instruction; the scale leaves in the final merge)
    if (p.stats) {
      float *rs = rowstat + wave * 64 * ST;
      const bool first = tile == (int)blockIdx.x;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          float *e0 = rs + ST * (16 * m + 4 * g + 2 * qp), *e1 = e0 + ST;
          f2 c;
          c.x = first ? row16_sum(acc[0][m][2 * qp]) * 0.0625f : e0[2];
          c.y = first ? row16_sum(acc[0][m][2 * qp + 1]) * 0.0625f : e1[2];
          f2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
          for (int i = 0; i < SLOTS; ++i)
            if (sj[i] >= 0) {
              const f2 v = f2{acc[i][m][2 * qp], acc[i][m][2 * qp + 1]} - c;
              s1 += v;
              s2 = __builtin_elementwise_fma(v, v, s2);
            }
          const float s1x = row16_sum(s1.x), s1y = row16_sum(s1.y);
          const float s2x = row16_sum(s2.x), s2y = row16_sum(s2.y);
          if (r == 0) {
            e0[0] += s1x; e0[1] += s2x;
            e1[0] += s1y; e1[1] += s2y;
            if (first) { e0[2] = c.x; e1[2] = c.y; }
          }
        }
    }
    PROF_T(3)
#ifndef ABL_NO_EPI      // (timing ablation: no staged stores)
    // ---- epilogue: the tile leaves through LDS as whole rows (stgcn_gcn3.hip), scaled back by 2^-S -------------------
    {
      float *stg = lds + ((NPH - 1) & 1) * BUF;
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // MFMA results -> VALU reads: not the compiler's business for assembly
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i)
          if (sj[i] >= 0) {
            float *d0 = stg + 4 * g * RS + r * V + sj[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) d0[q * RS] = acc[i][m][q] * p.inv_scale;
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float4 *zrow = reinterpret_cast<float4 *>(zg + (size_t)16 * m * row_stride);
        // the data gradient's addend (the gradient of the block's residual branch), added on the way out
        const float4 *arow = p.addend ? reinterpret_cast<const float4 *>(p.addend + (zg - z) + (size_t)16 * m * row_stride) : nullptr;
        const float4 *srow = reinterpret_cast<const float4 *>(stg);
#pragma unroll
        for (int it = 0; it < (NV4 + NW * 64 - 1) / (NW * 64); ++it) {
          const int e = it * NW * 64 + tid;
          if (e < NV4) {
            const int row = e / (RS / 4), c4 = e - row * (RS / 4);
            float4 v = srow[e];
            if (arow) {
              const float4 a4 = arow[(size_t)row * (row_stride / 4) + c4];
              v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
            }
            zrow[(size_t)row * (row_stride / 4) + c4] = v;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
#else
    { _Pragma("unroll") for (int i = 0; i < SLOTS; ++i) _Pragma("unroll") for (int m = 0; m < 4; ++m) asm volatile("" :: "v"(acc[i][m])); }
#endif
    PROF_T(4)
  }
  PROF_OUT
}

__global__ __launch_bounds__(NW * 64, 2) void gcn3h_kernel(Params p, const float *__restrict__ x,
                                                           const h8 *__restrict__ Wp, const float *__restrict__ coef,
                                                           const float *__restrict__ bias_cv, float *__restrict__ z) {
  extern __shared__ float lds[];
  float *bias_l = lds + 2 * BUF;
  float *coef_l = bias_l + 64 * V;
  const int tid = threadIdx.x;
  for (int e = tid; e < 64 * V; e += NW * 64) bias_l[e] = bias_cv ? bias_cv[e] : 0.f;
  for (int e = tid; e < (H3_LTOT + 1) * V; e += NW * 64) coef_l[e] = coef[e];
  float *rowstat = coef_l + (H3_LTOT + 1) * V;
  for (int e = tid; e < NW * 64 * ST; e += NW * 64) rowstat[e] = 0.f;
  __syncthreads();
  switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
    case 0: wave_main<0>(p, lds, x, Wp, z); break;
    case 1: wave_main<1>(p, lds, x, Wp, z); break;
    case 2: wave_main<2>(p, lds, x, Wp, z); break;
    case 3: wave_main<3>(p, lds, x, Wp, z); break;
    case 4: wave_main<4>(p, lds, x, Wp, z); break;
    case 5: wave_main<5>(p, lds, x, Wp, z); break;
    case 6: wave_main<6>(p, lds, x, Wp, z); break;
    default: wave_main<7>(p, lds, x, Wp, z); break;
  }
  if (p.stats) {                  // [64][3] = (count, mean, M2) of the workgroup's tiles: the eight waves' entries merged
    __syncthreads();
    if (tid < 64) {
      const int ntiles = (p.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
      const float per_joint = (float)(ntiles * F);
      float nw[NW], mw[NW], qw[NW];
      float msum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float *e = rowstat + (w * 64 + tid) * ST;
        int nj = 0;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) nj += slot_joints[w][i] >= 0;
        nw[w] = per_joint * (float)nj;
        const float s1 = e[0] * p.inv_scale, s2 = e[1] * p.inv_scale * p.inv_scale, c = e[2] * p.inv_scale;
        const float d = s1 / nw[w];
        mw[w] = c + d;
        qw[w] = fmaxf(s2 - s1 * d, 0.f);
        msum = fmaf(nw[w], mw[w] - mw[0], msum);
      }
      const float n = per_joint * (float)V;
      const float mean = mw[0] + msum / n;
      float m2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) m2 += qw[w] + nw[w] * (mw[w] - mean) * (mw[w] - mean);
      float *o = p.stats + (size_t)blockIdx.x * 192 + 3 * tid;
      o[0] = n; o[1] = mean; o[2] = m2;
    }
  }
}
}  // namespace

// x (N,64,T,53) f32; Wp: fp16 A operands [pair][phase][part][m][lane][8] built by tools/dev_gcn_f16.py from 2^S W (k index
// 8 kg + i of a lane = channel 16 ph + kg + 4 (i & 3), plane a of the pair for i < 4, plane b for i >= 4); coef f32
// [ltot + 1][53]: the column-form coefficient table + one row of zeros; bias_cv (64,53) or NULL; scale = 2^S.
// T % 16 == 0, x / z 16-byte aligned.
static unsigned long long *g_prof = nullptr;
static float *g_stats = nullptr;
static const float *g_addend = nullptr;
// the next launches add this tensor (same shape as z) to the result on the way out (NULL: off)
extern "C" void proto_gcn3h_addend(const void *buf) { g_addend = reinterpret_cast<const float *>(buf); }
// the next launches also write (count, mean, M2) per workgroup and channel: [min(tiles, 256)][64][3] (NULL: off)
extern "C" void proto_gcn3h_stats(void *buf) { g_stats = reinterpret_cast<float *>(buf); }
extern "C" void proto_gcn3h_profile(void *buf) { g_prof = reinterpret_cast<unsigned long long *>(buf); }   // [256][8][4] u64, -DPROFILE

extern "C" int proto_gcn3h_forward(int N, int T, int ltot1, const float *x, const void *Wp, const float *coef,
                                   const float *bias_cv, float scale, float *z, void *stream) {
  if (N <= 0 || T <= 0 || T % F != 0 || ltot1 != H3_LTOT + 1) return 1;
  Params p;
  p.T = T; p.tiles_per_seq = T / F; p.total_tiles = N * p.tiles_per_seq;
  p.scale = scale; p.inv_scale = 1.f / scale; p.prof = g_prof; p.stats = g_stats; p.addend = g_addend;
  const int blocks = p.total_tiles < 256 ? p.total_tiles : 256;
  const size_t lds = ((size_t)2 * BUF + 64 * V + (size_t)ltot1 * V + NW * 64 * ST) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gcn3h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(gcn3h_kernel, dim3(blocks), dim3(NW * 64), lds, (hipStream_t)stream, p, x,
                     reinterpret_cast<const h8 *>(Wp), coef, bias_cv, z);
  return (int)hipGetLastError();
}

extern "C" int proto_gcn3h_pairs(int *out) {          // the plane pairs of the schedule (6 x 2)
  constexpr int pr[H3_NPAIRS][2] = H3_PAIRS;
  for (int i = 0; i < H3_NPAIRS; ++i) { out[2 * i] = pr[i][0]; out[2 * i + 1] = pr[i][1]; }
  return H3_NPAIRS;
}
