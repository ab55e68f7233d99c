// stgcn_gcn3h_body.h -- the fused graph convolution of st_gcn_block in `split16` arithmetic (opt-in mode; split16.h),
// statically scheduled for the P2RNet skeleton, gfx950.  Included by stgcn_gcn3h_fwd.hip (column lists: the forward
// Z(w) = bias(w) + sum_k W_k . (X . A_k)(w), reference stgcn_layers.py:57-67) and by stgcn_gcn3h_dx.hip (H3_FORM_R: row
// lists with W_k^T = its data gradient); each translation unit includes its own generated schedule first
// (tools/gen_gcn_split_sched.py -> gcn3h_sched_{c,r}.inc).
//
// Skeleton of stgcn_gcn3.hip (persistent workgroups of 8 straight-line wave programs, 16-frame tiles, four 16-channel
// slices by LDS-DMA into two buffers, accumulators of up to 7 output joints per wave in registers, whole-row staged
// stores, the (count, mean, M2) statistics epilogue); what changes is the step:
//   * a step = a (plane PAIR, output joint) unit: one v_mfma_f32_16x16x32_f16 takes K = 16 channels x two planes.  K
//     index k = 8 kg + i of lane (kg, r): frame r, channel kg + 4 (i & 3) of the slice, plane a of the pair for i < 4,
//     plane b for i >= 4 -- every lane walks BOTH neighbour lists with four channels each (LDS offsets and coefficient
//     indices are immediates common to the wave);
//   * the fp32 aggregate of a unit is split into two fp16 parts in registers; W arrives pre-split from the host as
//     2^S_w W = w1 + w2 in the MFMA's lane order: three MFMAs per 16-row block (w1 x2, w2 x1, w1 x1) into ONE accumulator
//     at scale 2^(S_w + S_x); the operand scale 2^S_x of x (from its range word) rides in the coefficient table;
//   * 253 (forward) / 196 (data gradient) units per tile and phase instead of 454, 12 MFMAs of 16 cycles each instead of
//     16 of 32.
// MFMAs through the compiler's builtin between scheduling fences and the split through plain conversions: every hazard
// is the compiler's to track (the round-5 prototype used assembly blocks and found three ways to read stale registers).
#include "p2r_common.h"
#include "split16.h"

// H3_RES_SCALED (the data gradient: heavy-tailed operand, split16.h): the aggregate's residual part is kept scaled by
// 2^11 and meets the weight part 2^-11 w1.  A third weight plane in registers does not fit next to 112 accumulator
// registers (1,400 spilled registers, measured): 2^-11 w1 is formed from w1 right in front of its four MFMAs (four packed
// fp16 multiplies per 16-row block: exact powers of two).  The forward (activations behind a BatchNorm + ReLU: not
// heavy-tailed, 6e-7 of range against the exact kernel through six blocks) uses plain residuals.
#ifdef H3_RES_SCALED
#define H3_HAS_ADDEND true       // the data gradient adds the residual branch's gradient on the way out
#else
#define H3_HAS_ADDEND false
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned h3_u4 __attribute__((ext_vector_type(4)));

constexpr int H3_V = 53, H3_F = 16, H3_CP = 16, H3_NPH = 4, H3_NW = 8, H3_SLOTS = 7;
constexpr int H3_RS = H3_F * H3_V;            // 848
constexpr int H3_BUF = H3_CP * H3_RS;
constexpr int H3_NV4 = H3_BUF / 4;
constexpr int H3_PIECES = (H3_NV4 + 63) / 64;          // 53
constexpr int H3_PW = (H3_PIECES + H3_NW - 1) / H3_NW; // 7
constexpr int H3_ST = 3;      // floats per (wave, row) statistics entry: (sum, sum of squares) about the pivot, pivot -- at the accumulators' scale
constexpr int h3_slot_joints[H3_NW][H3_SLOTS] = H3_SLOT_JOINTS;
constexpr int h3_plane0[H3_NW] = {H3_PLANE0_0, H3_PLANE0_1, H3_PLANE0_2, H3_PLANE0_3, H3_PLANE0_4, H3_PLANE0_5, H3_PLANE0_6, H3_PLANE0_7};

struct H3Params {
  int T, tiles_per_seq, total_tiles;
  float *stats;
  const float *addend;
  const unsigned char *addend_mask;
  const unsigned *x_amax;
  const float *winv;
};

__device__ __forceinline__ unsigned h3_lds_addr(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
__device__ __forceinline__ void h3_dma16(const float *base, int voff, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(h3_lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// the 12 MFMAs of a unit: a1 / a2 = the parts (w1, w2) of [W_a | W_b] rows 16 m + r, b1 / b2 = the parts of the aggregate
// (H3_RES_SCALED: b2 = its residual scaled by 2^11, met by 2^-11 w1: split16.h)
__device__ __forceinline__ void h3_mfma12(f32x4 (&acc)[4], const p2r_h8 (&a1)[4], const p2r_h8 (&a2)[4], const p2r_h8 &b1,
                                          const p2r_h8 &b2) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#ifdef H3_RES_SCALED
    const p2r_h8 a1s = a1[m] * (_Float16)(1.0 / P2R_RES_SCALE);
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1s, b2, acc[m], 0, 0, 0);
#else
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[m], b2, acc[m], 0, 0, 0);
#endif
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[m], b1, acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[m], b1, acc[m], 0, 0, 0);
}

// x = x1 + x2 for a pair of values (split16.h: the split sees VALUES)
__device__ __forceinline__ void h3_split2(float x0, float x1, unsigned &p, unsigned &r) {
  p2r_f2 x = {x0, x1};
  asm volatile("" : "+v"(x));
  const p2r_h2 ph = __builtin_convertvector(x, p2r_h2);
  p = __builtin_bit_cast(unsigned, ph);
#ifdef H3_RES_SCALED
  r = __builtin_bit_cast(unsigned, __builtin_convertvector((x - __builtin_convertvector(ph, p2r_f2)) * P2R_RES_SCALE, p2r_h2));
#else
  r = __builtin_bit_cast(unsigned, __builtin_convertvector(x - __builtin_convertvector(ph, p2r_f2), p2r_h2));
#endif
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
    for (int i = 0; i < 4; ++i) xv[j][i] = *reinterpret_cast<const float *>(xl + off[j] + i * (4 * H3_RS * 4));
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
#define H3_VISIT(pair) { load_a(aS, pair, ph); }
#define H3_G(k, ne, o0, c0, o1, c1, o2, c2, o3, c3) h3_gather<ne, o0, c0, o1, c1, o2, c2, o3, c3>(xl, cl, xv_[k], cf_[k]);
#define H3_C(k, ne, zero, h0, f0, h1, f1, h2, f2, h3, f3) h3_combine<ne, zero, h0, f0, h1, f1, h2, f2, h3, f3>(xv_[k], cf_[k], xagg);
#define H3_S(par)                                                                         \
  {                                                                                       \
    unsigned p_[4], r_[4];                                                                \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) h3_split2(xagg[2 * q_], xagg[2 * q_ + 1], p_[q_], r_[q_]); \
    b1_[par] = __builtin_bit_cast(p2r_h8, h3_u4{p_[0], p_[1], p_[2], p_[3]});              \
    b2_[par] = __builtin_bit_cast(p2r_h8, h3_u4{r_[0], r_[1], r_[2], r_[3]});              \
  }
#define H3_M(slot, par)                                       \
  {                                                           \
    __builtin_amdgcn_sched_barrier(0);                        \
    h3_mfma12(acc[slot], aS[0], aS[1], b1_[par], b2_[par]);   \
    __builtin_amdgcn_sched_barrier(0);                        \
  }
#define H3_END(pieces, pair0)                                                             \
  {                                                                                       \
    if (copy) { _Pragma("unroll") for (int i_ = pieces; i_ < H3_PW; ++i_) dma_piece(i_); } \
    load_a(aS, pair0, (ph + 1) & (H3_NPH - 1));                                           \
  }

template <int WAVE>
__device__ __forceinline__ void h3_wave_main(const H3Params &p, float *lds, const float *__restrict__ x,
                                             const p2r_h8 *__restrict__ Wp, float *__restrict__ z, float scale,
                                             float inv_scale) {
  constexpr int V = H3_V, F = H3_F, CP = H3_CP, NPH = H3_NPH, NW = H3_NW, SLOTS = H3_SLOTS, RS = H3_RS, BUF = H3_BUF;
  constexpr int NV4 = H3_NV4, PIECES = H3_PIECES, PW = H3_PW, ST = H3_ST;
  constexpr int wave = WAVE;
  float *bias_l = lds + 2 * BUF;                       // [64][V]
  float *coef_l = bias_l + 64 * V;                     // [ltot + 1][V] (last row zeros)
  float *rowstat = coef_l + (H3_LTOT + 1) * V;         // [NW][64][ST]
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = lane >> 4, r = lane & 15;
  constexpr const int (&sj)[SLOTS] = h3_slot_joints[WAVE];
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

  f32x4 acc[SLOTS][4];
  p2r_h8 aS[2][4];                                     // [part][m]: [W_a | W_b] rows 16 m + r, this lane's 8 k values
  float xagg[8], xv_[2][4][4], cf_[2][4];
  p2r_h8 b1_[2], b2_[2];
  auto load_a = [&](p2r_h8 (&a)[2][4], int pair, int ph) {
    // Wp[pair][ph][part][m][lane] (16 bytes each; three parts, the third -- 2^-11 w1 -- is not loaded: see the top)
    const p2r_h8 *wp = Wp + ((size_t)(pair * NPH + ph) * 3 * 4) * 64 + lane;
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
      if (doff[i] >= 0) h3_dma16(xr, doff[i], lds + (i * NW + wave) * 256);
  }
  load_a(aS, h3_plane0[WAVE], 0);

  for (; tile < p.total_tiles; tile += gridDim.x) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * F;
    const float *xg = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    float *zg = z + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    const int ntile = tile + gridDim.x;
    const bool has_next = ntile < p.total_tiles;
    const int nseq = has_next ? ntile / p.tiles_per_seq : 0, nt0 = has_next ? (ntile % p.tiles_per_seq) * F : 0;
    const float *nxg = x + (size_t)nseq * 64 * row_stride + (size_t)nt0 * V;

    // accumulators start from the bias table, at the accumulators' scale
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const float *bl = bias_l + 4 * g * V + (sj[i] >= 0 ? sj[i] : 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][m][q] = bl[(16 * m + q) * V] * scale;
    }

#pragma unroll 1
    for (int ph = 0; ph < NPH; ++ph) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      float *buf_nxt = lds + ((ph + 1) & 1) * BUF;
      const char *xl = xl0 + (ph & 1) * BUF * sizeof(float);
      const bool copy = ph + 1 < NPH || has_next;
      const float *src = (ph + 1 < NPH) ? xg + (size_t)(ph + 1) * CP * row_stride : nxg;
      auto dma_piece = [&](int i) {
        if (doff[i] >= 0) h3_dma16(src, doff[i], buf_nxt + (i * NW + wave) * 256);
      };
      if constexpr (WAVE == 0) { H3_BODY_0 } else if constexpr (WAVE == 1) { H3_BODY_1 }
      else if constexpr (WAVE == 2) { H3_BODY_2 } else if constexpr (WAVE == 3) { H3_BODY_3 }
      else if constexpr (WAVE == 4) { H3_BODY_4 } else if constexpr (WAVE == 5) { H3_BODY_5 }
      else if constexpr (WAVE == 6) { H3_BODY_6 } else { H3_BODY_7 }
    }

    // statistics of the stored values for the BatchNorm that follows (stgcn_gcn3.hip's epilogue, on the SCALED
    // accumulators: sums about a pivot per (wave, row), two rows per packed instruction; the scale leaves in the merge)
    if (p.stats) {
      float *rs = rowstat + wave * 64 * ST;
      const bool first = tile == (int)blockIdx.x;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          float *e0 = rs + ST * (16 * m + 4 * g + 2 * qp), *e1 = e0 + ST;
          p2r_f2 c;
          c.x = first ? p2r_row16_sum(acc[0][m][2 * qp]) * 0.0625f : e0[2];
          c.y = first ? p2r_row16_sum(acc[0][m][2 * qp + 1]) * 0.0625f : e1[2];
          p2r_f2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
          for (int i = 0; i < SLOTS; ++i)
            if (sj[i] >= 0) {
              const p2r_f2 v = p2r_f2{acc[i][m][2 * qp], acc[i][m][2 * qp + 1]} - c;
              s1 += v;
              s2 = __builtin_elementwise_fma(v, v, s2);
            }
          const float s1x = p2r_row16_sum(s1.x), s1y = p2r_row16_sum(s1.y);
          const float s2x = p2r_row16_sum(s2.x), s2y = p2r_row16_sum(s2.y);
          if (r == 0) {
            e0[0] += s1x; e0[1] += s2x;
            e1[0] += s1y; e1[1] += s2y;
            if (first) { e0[2] = c.x; e1[2] = c.y; }
          }
        }
    }
    // ---- epilogue: the tile leaves through LDS as whole rows (stgcn_gcn3.hip), scaled back ---------------------------
    {
      float *stg = lds + ((NPH - 1) & 1) * BUF;
      // store mapping: 32 threads per channel row (16 rows x 32 = the workgroup), 7 float4 each at a stride of 32 -- every
      // address of a round is one 64-bit base plus an immediate (a thread walking the tile linearly needs a division and
      // 64-bit arithmetic per float4; with the addend's two extra streams that cost 1,000 spilled registers)
      constexpr int R4 = RS / 4;                            // float4 per row (212)
      constexpr int ITS = (R4 + 31) / 32;                   // 7 (the last one: 20 of 32 threads)
      const int srow_i = tid >> 5, scol = tid & 31;
      const size_t toff = (size_t)(zg - z) + (size_t)srow_i * row_stride + 4 * scol;     // floats; + 16 m row_stride per round
      const float4 *sbase = reinterpret_cast<const float4 *>(stg) + srow_i * R4 + scol;
      // the data gradient's addend (the gradient of the block's residual branch) is added on the way out -- where its
      // ReLU mask byte is set, when it arrives unmasked.  A round's share of it is requested at the top of the round, in
      // front of the staging writes and their barrier (behind the barrier the loads are exposed four times per tile:
      // +0.16 ms per launch, measured)
      float4 apf[ITS];
      uchar4 mpf[ITS];
      auto fetch_addend = [&](int m) {
        const size_t goff = toff + (size_t)16 * m * row_stride;
        const float4 *arow = reinterpret_cast<const float4 *>(p.addend + goff);
        const uchar4 *mrow = p.addend_mask ? reinterpret_cast<const uchar4 *>(p.addend_mask + goff) : nullptr;
#pragma unroll
        for (int it = 0; it < ITS; ++it)
          if (32 * it + 31 < R4 || 32 * it + scol < R4) {
            apf[it] = arow[32 * it];
            mpf[it] = mrow ? mrow[32 * it] : make_uchar4(1, 1, 1, 1);
          }
      };
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#ifndef H3_ADDEND_LATE
        if (H3_HAS_ADDEND && p.addend) fetch_addend(m);       // in flight under the round's staging writes and barrier
#endif
#pragma unroll
        for (int i = 0; i < SLOTS; ++i)
          if (sj[i] >= 0) {
            float *d0 = stg + 4 * g * RS + r * V + sj[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) d0[q * RS] = acc[i][m][q] * inv_scale;
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float4 *zrow = reinterpret_cast<float4 *>(z + toff + (size_t)16 * m * row_stride);
#ifdef H3_ADDEND_LATE
        if (H3_HAS_ADDEND && p.addend) fetch_addend(m);
#endif
#pragma unroll
        for (int it = 0; it < ITS; ++it)
          if (32 * it + 31 < R4 || 32 * it + scol < R4) {
            float4 v = sbase[32 * it];
            if (H3_HAS_ADDEND && p.addend) {
              const float4 a4 = apf[it];
              const uchar4 mk = mpf[it];
              v.x += mk.x ? a4.x : 0.f; v.y += mk.y ? a4.y : 0.f; v.z += mk.z ? a4.z : 0.f; v.w += mk.w ? a4.w : 0.f;
            }
            zrow[32 * it] = v;
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
}

__global__ __launch_bounds__(H3_NW * 64, 2) void H3_KERNEL(H3Params p, const float *__restrict__ x,
                                                           const p2r_h8 *__restrict__ Wp, const float *__restrict__ coef,
                                                           const float *__restrict__ bias_cv, float *__restrict__ z) {
  constexpr int V = H3_V, NW = H3_NW, ST = H3_ST, BUF = H3_BUF;
  extern __shared__ float lds[];
  float *bias_l = lds + 2 * BUF;
  float *coef_l = bias_l + 64 * V;
  const int tid = threadIdx.x;
  // scale of the accumulators: 2^S_x (operand x, from its range word; carried by the coefficients) * 2^S_w (weights)
  float xs, xinv;
  p2r_split_scale(p.x_amax, xs, xinv);
  const float winv = p.winv[0];
  const float scale = xs * (1.f / winv), inv_scale = xinv * winv;
  for (int e = tid; e < 64 * V; e += NW * 64) bias_l[e] = bias_cv ? bias_cv[e] : 0.f;
  for (int e = tid; e < (H3_LTOT + 1) * V; e += NW * 64) coef_l[e] = e < H3_LTOT * V ? coef[e] * xs : 0.f;
  float *rowstat = coef_l + (H3_LTOT + 1) * V;
  for (int e = tid; e < NW * 64 * ST; e += NW * 64) rowstat[e] = 0.f;
  __syncthreads();
  switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
    case 0: h3_wave_main<0>(p, lds, x, Wp, z, scale, inv_scale); break;
    case 1: h3_wave_main<1>(p, lds, x, Wp, z, scale, inv_scale); break;
    case 2: h3_wave_main<2>(p, lds, x, Wp, z, scale, inv_scale); break;
    case 3: h3_wave_main<3>(p, lds, x, Wp, z, scale, inv_scale); break;
    case 4: h3_wave_main<4>(p, lds, x, Wp, z, scale, inv_scale); break;
    case 5: h3_wave_main<5>(p, lds, x, Wp, z, scale, inv_scale); break;
    case 6: h3_wave_main<6>(p, lds, x, Wp, z, scale, inv_scale); break;
    default: h3_wave_main<7>(p, lds, x, Wp, z, scale, inv_scale); break;
  }
  if (p.stats) {                  // [64][3] = (count, mean, M2) of the workgroup's tiles: the eight waves' entries merged
    __syncthreads();
    if (tid < 64) {
      const int ntiles = (p.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
      const float per_joint = (float)(ntiles * H3_F);
      float nw[NW], mw[NW], qw[NW];
      float msum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float *e = rowstat + (w * 64 + tid) * ST;
        int nj = 0;
#pragma unroll
        for (int i = 0; i < H3_SLOTS; ++i) nj += h3_slot_joints[w][i] >= 0;
        nw[w] = per_joint * (float)nj;
        const float s1 = e[0] * inv_scale, s2 = e[1] * inv_scale * inv_scale, c = e[2] * inv_scale;
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

// Launch: x, z (N,64,T,53) f32, 16-byte aligned, T % 16 == 0.
int h3_launch(int N, int T, const float *x, const void *Wp, const float *winv, const float *coef, const float *bias_cv,
              const float *addend, const unsigned char *addend_mask, float *z, float *stats_partial, int *n_partials,
              const unsigned *x_amax, void *stream) {
  if (N < 0 || T <= 0 || T % H3_F != 0 || T > (1 << 19)) return P2R_EINVAL;      // (16 channel rows x T x 53 floats as a 32-bit byte offset)
  if (((uintptr_t)x % 16) != 0 || ((uintptr_t)z % 16) != 0 || ((uintptr_t)addend % 16) != 0 ||
      ((uintptr_t)addend_mask % 4) != 0 || ((uintptr_t)Wp % 16) != 0 || (addend_mask && !addend))
    return P2R_EINVAL;
  if (n_partials) *n_partials = 0;
  if (N == 0) return P2R_OK;
  const long long tiles = (long long)N * (T / H3_F);
  if (tiles > 0x7fffffffLL) return P2R_EINVAL;
  const int blocks = (int)(tiles < 256 ? tiles : 256);
  if (n_partials) *n_partials = blocks;
  if (!z) return P2R_OK;
  if (!Wp || !winv || !coef) return P2R_EINVAL;
  H3Params p;
  p.T = T; p.tiles_per_seq = T / H3_F; p.total_tiles = (int)tiles;
  p.stats = stats_partial; p.addend = addend; p.addend_mask = addend_mask; p.x_amax = x_amax; p.winv = winv;
  const size_t lds = ((size_t)2 * H3_BUF + 64 * H3_V + (size_t)(H3_LTOT + 1) * H3_V + H3_NW * 64 * H3_ST) * sizeof(float);
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(H3_KERNEL, lds_ok);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(H3_KERNEL, dim3(blocks), dim3(H3_NW * 64), lds, p2r_stream(stream), p, x,
                     reinterpret_cast<const p2r_h8 *>(Wp), coef, bias_cv, z);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

}  // namespace
