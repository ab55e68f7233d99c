// PROTOTYPE, not part of the product (libp2r_hip.so does not contain it, bench.py does not run it): the FORWARD of the
// fused graph convolution  Z(w) = bias(w) + sum_k W_k . (X . A_k)(w)  on two-part fp16 MFMA products with
// K = 32 = 16 channels x TWO PLANES per v_mfma_f32_16x16x32_f16 (DESIGN.md section 5 "Round 5").  Skeleton of
// pose2room_amd/csrc/stgcn_gcn3.hip (persistent workgroups of 8 straight-line wave programs, 16-frame tiles, four
// 16-channel slices by LDS-DMA into two buffers, accumulators of up to 7 joints per wave in registers, staged whole-row
// stores); what changes is the step:
//   * a step = a (plane pair, output joint) unit.  Lane (g, r): frame r, 8 channels 8 (g & 1) + i of the slice; lanes
//     0-31 build plane a's aggregate, lanes 32-63 plane b's, from the UNION of the two neighbour lists (same LDS offsets
//     for both halves; the coefficient of an entry is plane a's or plane b's value, zero where the plane lacks it);
//   * the aggregate is split into two fp16 parts x = x1 + x2 (x2 unscaled: its absolute error, 3e-8, is what counts
//     behind a BatchNorm + ReLU), W comes pre-split from the host as 2^S W = w1 + w2: three MFMAs per row tile
//     (w1 x2, w2 x1, w1 x1) into ONE accumulator at scale 2^S;
//   * 253 units per tile and phase instead of 454, 12 MFMAs of 16 cycles each instead of 16 of 32.
// Forward only, no statistics epilogue.  Schedule: tools/gen_gcn_pair_sched.py -> gcn3h_sched.inc.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "gcn3h_sched.inc"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int V = 53, F = 16, CP = 16, NPH = 4, NW = 8, SLOTS = 7;
constexpr int RS = F * V;            // 848
constexpr int BUF = CP * RS;
constexpr int NV4 = BUF / 4;
constexpr int PIECES = (NV4 + 63) / 64;          // 53
constexpr int PW = (PIECES + NW - 1) / NW;       // 7
constexpr int slot_joints[NW][SLOTS] = H3_SLOT_JOINTS;
constexpr int plane0[NW] = {H3_PLANE0_0, H3_PLANE0_1, H3_PLANE0_2, H3_PLANE0_3, H3_PLANE0_4, H3_PLANE0_5, H3_PLANE0_6, H3_PLANE0_7};

struct Params { int T, tiles_per_seq, total_tiles; float scale, inv_scale; };

__device__ __forceinline__ unsigned lds_addr(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
__device__ __forceinline__ void dma16(const float *base, int voff, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// 12 MFMAs of a unit.  Through the builtin, not an assembly block: the compiler's hazard recogniser has to see them -- an
// in-flight v_mfma_f32_16x16x32_f16 still reads its A / B registers after it has issued, and the vector instructions
// that build the NEXT unit's operands in the same registers right behind an opaque assembly block corrupted them
// (measured: 0.2 of range wrong in a few joints; 16 trailing s_nop left 2e-5).  The fp32 16x16x4 form of gcn3 does not
// show this (one-register operands).
__device__ __forceinline__ void mfma12(f32x4 (&acc)[4], const h8 (&a1)[4], const h8 (&a2)[4], const h8 &b1, const h8 &b2) {
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[m], b2, acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[m], b1, acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[m], b1, acc[m], 0, 0, 0);
}

// one chunk of a unit's union list: x[i] (+)= sum_j c_j X[8 (g & 1) + i][frame r][joint of entry j],
// c_j = plane a's coefficient in lanes 0-31, plane b's in lanes 32-63
template <int FIRST, int NE, int O0, int A0, int B0, int O1, int A1, int B1, int O2, int A2, int B2>
__device__ __forceinline__ void chunk(const char *xl, const char *cl, bool up, float (&x)[8]) {
  constexpr int off[3] = {O0, O1, O2}, ia[3] = {A0, A1, A2}, ib[3] = {B0, B1, B2};
  float xv[3][8], c[3];
#pragma unroll
  for (int j = 0; j < NE; ++j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[j][i] = *reinterpret_cast<const float *>(xl + off[j] + i * RS * 4);
    const float ca = *reinterpret_cast<const float *>(cl + 4 * ia[j]), cb = *reinterpret_cast<const float *>(cl + 4 * ib[j]);
    c[j] = up ? cb : ca;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v = FIRST ? c[0] * xv[0][i] : fmaf(c[0], xv[0][i], x[i]);
#pragma unroll
    for (int j = 1; j < NE; ++j) v = fmaf(c[j], xv[j][i], v);
    x[i] = v;
  }
}

#define H3_VISIT(set, pair, next, wrap, piece)                                  \
  {                                                                             \
    load_a(aS[(set) ^ 1], next, (wrap) ? ((ph + 1) & (NPH - 1)) : ph);           \
    if ((piece) >= 0 && copy) dma_piece(piece);                                 \
  }
#define H3_B(first, ne, o0, a0, b0, o1, a1, b1, o2, a2, b2) chunk<first, ne, o0, a0, b0, o1, a1, b1, o2, a2, b2>(xl, cl, up, xagg);
// The split x = x1 + x2 has to see the aggregate as an fp32 VALUE.  Without the empty asm the compiler contracts
// fp16(c * x) of a one-entry list into v_fma_mixlo_f16 (one rounding of the exact product) in one place and keeps
// fp16(fp32(c * x)) in another; where fp32(c * x) falls exactly between two fp16 numbers the two disagree, x2 is taken
// against the other neighbour and x1 + x2 is off by one fp16 ulp (measured: 4 of 7 M aggregates, 1.8e-5 of range).
#define H3_M(set, slot)                                                                   \
  {                                                                                       \
    h8 b1_, b2_;                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                    \
      float v_ = xagg[i_];                                                                \
      asm volatile("" : "+v"(v_));   /* see the note on H3_M */                           \
      const _Float16 p_ = (_Float16)v_;                                                   \
      b1_[i_] = p_; b2_[i_] = (_Float16)(v_ - (float)p_);                                 \
    }                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    mfma12(acc[slot], aS[set][0], aS[set][1], b1_, b2_);                                  \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  }
#define H3_END(parity, pieces, pair0)                                                     \
  {                                                                                       \
    if (copy) { _Pragma("unroll") for (int i_ = pieces; i_ < PW; ++i_) dma_piece(i_); }   \
    if (parity) {                                                                         \
      _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) aS[0][q_][m_] = aS[1][q_][m_]; \
    }                                                                                     \
  }

template <int WAVE>
__device__ __forceinline__ void wave_main(const Params &p, float *lds, const float *__restrict__ x,
                                          const h8 *__restrict__ Wp, float *__restrict__ z) {
  constexpr int wave = WAVE;
  float *bias_l = lds + 2 * BUF;                       // [64][V]
  float *coef_l = bias_l + 64 * V;                     // [ltot + 1][V] (last row zeros)
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = lane >> 4, r = lane & 15;
  const bool up = lane >= 32;
  constexpr const int (&sj)[SLOTS] = slot_joints[WAVE];
  const size_t row_stride = (size_t)p.T * V;
  const char *xl0 = reinterpret_cast<const char *>(lds + 8 * (g & 1) * RS + r * V);   // channels 8 (g & 1) .., frame r
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
  h8 aS[2][2][4];                                      // [set][part][m]: [W_a | W_b] rows 16 m + r, this lane's 8 k values
  float xagg[8];
  auto load_a = [&](h8 (&a)[2][4], int pair, int ph) {
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
  load_a(aS[0], plane0[WAVE], 0);

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
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
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
    }

    // ---- epilogue: the tile leaves through LDS as whole rows (stgcn_gcn3.hip), scaled back by 2^-S -------------------
    {
      float *stg = lds + ((NPH - 1) & 1) * BUF;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
        const float4 *srow = reinterpret_cast<const float4 *>(stg);
#pragma unroll
        for (int it = 0; it < (NV4 + NW * 64 - 1) / (NW * 64); ++it) {
          const int e = it * NW * 64 + tid;
          if (e < NV4) {
            const int row = e / (RS / 4), c4 = e - row * (RS / 4);
            zrow[(size_t)row * (row_stride / 4) + c4] = srow[e];
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
}

__global__ __launch_bounds__(NW * 64, 2) void gcn3h_kernel(Params p, int ltot1, const float *__restrict__ x,
                                                           const h8 *__restrict__ Wp, const float *__restrict__ coef,
                                                           const float *__restrict__ bias_cv, float *__restrict__ z) {
  extern __shared__ float lds[];
  float *bias_l = lds + 2 * BUF;
  float *coef_l = bias_l + 64 * V;
  const int tid = threadIdx.x;
  for (int e = tid; e < 64 * V; e += NW * 64) bias_l[e] = bias_cv ? bias_cv[e] : 0.f;
  for (int e = tid; e < ltot1 * V; e += NW * 64) coef_l[e] = coef[e];
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
}
}  // namespace

// x (N,64,T,53) f32; Wp: fp16 A operands [pair][phase][part][m][lane][8] built by tools/dev_gcn_f16.py from 2^S W;
// coef f32 [ltot + 1][53]: the column-form coefficient table + one row of zeros; bias_cv (64,53) or NULL; scale = 2^S.
// T % 16 == 0, x / z 16-byte aligned.
extern "C" int proto_gcn3h_forward(int N, int T, int ltot1, const float *x, const void *Wp, const float *coef,
                                   const float *bias_cv, float scale, float *z, void *stream) {
  if (N <= 0 || T <= 0 || T % F != 0 || ltot1 != H3_LTOT + 1) return 1;
  Params p;
  p.T = T; p.tiles_per_seq = T / F; p.total_tiles = N * p.tiles_per_seq;
  p.scale = scale; p.inv_scale = 1.f / scale;
  const int blocks = p.total_tiles < 256 ? p.total_tiles : 256;
  const size_t lds = ((size_t)2 * BUF + 64 * V + (size_t)ltot1 * V) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gcn3h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(gcn3h_kernel, dim3(blocks), dim3(NW * 64), lds, (hipStream_t)stream, p, ltot1, x,
                     reinterpret_cast<const h8 *>(Wp), coef, bias_cv, z);
  return (int)hipGetLastError();
}

extern "C" int proto_gcn3h_pairs(int *out) {          // the plane pairs of the schedule (6 x 2)
  constexpr int pr[H3_NPAIRS][2] = H3_PAIRS;
  for (int i = 0; i < H3_NPAIRS; ++i) { out[2 * i] = pr[i][0]; out[2 * i + 1] = pr[i][1]; }
  return H3_NPAIRS;
}
