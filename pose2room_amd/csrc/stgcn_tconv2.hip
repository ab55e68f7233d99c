// stgcn_tconv2.hip -- temporal (3,1) convolution of st_gcn_block fused with the preceding BatchNorm + ReLU, second
// generation, gfx950.
//
// Same operator as tconv_fused_kernel<3> in stgcn_tconv.hip (reference models/p2rnet/modules/stgcn_layers.py:399-411:
// BatchNorm2d -> ReLU -> Conv2d(64, 64, (3,1), padding (1,0)), and -- with flipped / transposed taps and no input
// transform -- its data gradient):
//     out[n,c,t,w] = bias[c] + sum_{p<3} sum_ci W[p][c][ci] * h[n,ci,t+p-1,w],   h = relu(x*scale+shift) or x, 0 outside [0,T)
//
// It runs on the skeleton of the second-generation graph convolution (stgcn_gcn2.hip), whose measurements it
// inherits:
//   * MFMA n-tile = 16 frames of ONE joint; the three taps are three "planes" whose B operand is the input shifted by
//     one frame: lane (g = lane >> 4, r = lane & 15) reads channel 4s+g at frame r+p-1 of the joint -- no gather, no
//     list, one address add per k-step group.  53 is odd and the row stride 848 == 16 (mod 32): conflict-free.
//   * the 64 input channels in four phases of 16; a phase's slice = 16 rows x 16 frames x 53 joints (16-byte LDS-DMA
//     pieces, the tensor's own order) plus the two halo frames t0-1 and t0+16 (4-byte pieces) in one of two LDS
//     buffers, copied under the previous phase's MFMAs;
//   * BatchNorm affine + ReLU are applied in place to each slice when it has landed in LDS (the normalised
//     activation never exists in HBM).  Positions outside the sequence hold NaN in that mode -- max(NaN, 0) = 0 is
//     exactly the zero padding of the convolution -- and 0 in the plain mode;
//   * persistent workgroups; a wave owns up to 7 consecutive joints x 64 rows x 16 frames of accumulators across the
//     phases; the tile leaves through LDS as whole rows; the epilogue also emits per-channel (sum, sum of squares).
// On gfx950 fp32 MFMAs and VALU work of the co-resident wave do not overlap, which is why the per-record work here is
// four LDS reads at immediate offsets and nothing else.
#include "p2r_common.h"

namespace {

typedef float f32x4t __attribute__((ext_vector_type(4)));

constexpr int T2_F = 16;            // frames per tile = columns of an MFMA n-tile
constexpr int T2_CP = 16;           // channels per phase
constexpr int T2_NPH = 4;
constexpr int T2_NW = 8;            // waves per workgroup
constexpr int T2_SLOTS = 7;         // joints per wave (consecutive)

struct T2Params {
  int T, tiles_per_seq, total_tiles;
  int vec;                          // rows 16-byte aligned and T*V % 4 == 0: 16-byte DMA pieces for full tiles
};

__device__ __forceinline__ unsigned t2_lds_addr(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
// LDS-DMA as inline assembly (see stgcn_gcn2.hip: keeps hipcc from waiting for the copy in front of every LDS read)
__device__ __forceinline__ void t2_dma16(const float *src, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(t2_lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
__device__ __forceinline__ void t2_dma4(const float *src, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(t2_lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

// BWD (data-gradient instance only): the statistics epilogue emits the two sums of the BatchNorm + ReLU backward
// of the layer in front instead -- per channel sum g' and sum g' * zhat with g' = out where relu'(scale*z+shift),
// zhat = (z - mean) * invstd, z = bwd_z (the saved activation), constants from bwd_fin [4][64] = (mean, invstd,
// scale, shift).  The rows leave through LDS anyway; the extra traffic is one read of z.
// TAPS = 3: the (3,1) temporal convolution; TAPS = 1: a pointwise 64 -> 64 convolution over the same layout (the
// embedding MLPs): no halo frames, one plane.
template <int VT, bool XFORM, bool BWD, int TAPS>
__global__ __launch_bounds__(T2_NW * 64, 2) void tconv2_kernel(
    T2Params p, const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ Wp, const float *__restrict__ bias, float *__restrict__ out,
    float *__restrict__ stats_partial, const float *__restrict__ bwd_z, const float *__restrict__ bwd_fin) {
  constexpr int V = VT, NW = T2_NW, SLOTS = T2_SLOTS;
  constexpr int RS = T2_F * V;                        // main row stride (floats): 848 == 16 (mod 32)
  constexpr int MAIN = T2_CP * RS;                    // floats of the 16-frame part of a slice
  constexpr int HRS = 2 * V;                          // halo row: frame t0-1, frame t0+16
  constexpr int HALO = T2_CP * HRS;
  constexpr int BUF = MAIN + HALO;                    // floats per phase buffer (61,056 bytes)
  extern __shared__ float lds[];
  float *rowstat = lds + 2 * BUF;                     // [NW][64][2]
  float *aff = rowstat + NW * 128;                    // [64][2] (scale, shift) of the input transform
  float *bias_l = aff + 128;                          // [64]
  float *bstat = bias_l + 64;                         // [64][2] (mean, invstd) of the BWD epilogue

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;

  for (int e = tid; e < NW * 128; e += NW * 64) rowstat[e] = 0.f;
  if (tid < 64) {
    aff[2 * tid] = XFORM ? scale[tid] : (BWD ? bwd_fin[128 + tid] : 1.f);
    aff[2 * tid + 1] = XFORM ? shift[tid] : (BWD ? bwd_fin[192 + tid] : 0.f);
    bias_l[tid] = bias ? bias[tid] : 0.f;
    bstat[2 * tid] = BWD ? bwd_fin[tid] : 0.f;
    bstat[2 * tid + 1] = BWD ? bwd_fin[64 + tid] : 1.f;
  }
  __syncthreads();

  // joints of this wave: consecutive runs 7,7,7,7,7,6,6,6 (waves w and w + 4 share a SIMD: 14,13,13,13 units per tap)
  const int j0 = wave < 5 ? 7 * wave : 35 + 6 * (wave - 5);
  const int nslots = V == 53 ? (wave < 5 ? 7 : 6) : 0;

  const size_t row_stride = (size_t)p.T * V;
  const float fillv = XFORM ? __int_as_float(0x7fc00000) : 0.f;   // outside the sequence: NaN -> relu gives the zero padding

  // per-lane read positions inside a buffer for (tap, k-step s): channel 4s+g, frame r+tap-1 (halo for -1 / 16)
  int rd[TAPS][4];
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp) {
    const int f = r + tp - (TAPS - 1) / 2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ch = 4 * s + g;
      rd[tp][s] = f < 0 ? MAIN + ch * HRS : (f >= T2_F ? MAIN + ch * HRS + V : ch * RS + f * V);
    }
  }

  // ---- LDS-DMA of one phase slice ------------------------------------------------------------------------------
  constexpr int NV4 = MAIN / 4;
  constexpr int PIECES16 = (NV4 + 63) / 64;           // 53
  constexpr int PW16 = (PIECES16 + NW - 1) / NW;      // 7 per wave
  constexpr int PIECES4 = (MAIN + 63) / 64;           // 212
  constexpr int PW4 = (PIECES4 + NW - 1) / NW;        // 27 per wave
  constexpr int PIECESH = (HALO + 63) / 64;           // 27
  constexpr int PWH = TAPS > 1 ? (PIECESH + NW - 1) / NW : 0;   // 4 per wave; no halo for a single tap
  constexpr int MPT = (PW16 + TAPS - 1) / TAPS;       // main / halo pieces issued per tap visit
  constexpr int HPT = TAPS > 1 ? (PWH + TAPS - 1) / TAPS : 0;
  // main part: piece_i in [0, PW16); frames >= `frames` of a ragged tile are filled, not copied
  auto dma_main = [&](int piece_i, float *buf, const float *xrow0, int frames) {
    if (p.vec && frames == T2_F) {
      const int pc = piece_i * NW + wave;
      if (piece_i < PW16 && pc < PIECES16) {
        const int e = pc * 64 + lane;
        const int row = e / (RS / 4), c4 = e - row * (RS / 4);
        if (e < NV4) t2_dma16(xrow0 + (size_t)row * row_stride + 4 * c4, buf + pc * 256);
      }
    } else {
      constexpr int PER = (PW4 + PW16 - 1) / PW16;
#pragma unroll 1
      for (int q = 0; q < PER; ++q) {
        const int pi = piece_i * PER + q;
        const int pc = pi * NW + wave;
        if (pi < PW4 && pc < PIECES4) {
          const int e = pc * 64 + lane;
          const int row = e / RS, col = e - row * RS;
          if (e < MAIN) {
            if (col < frames * V) t2_dma4(xrow0 + (size_t)row * row_stride + col, buf + pc * 64);
            else buf[e] = fillv;
          }
        }
      }
    }
  };
  // halo part: piece_i in [0, PWH); lo / hi: frame t0-1 / t0+16 exists in the sequence
  auto dma_halo = [&](int piece_i, float *buf, const float *xrow0, bool lo, bool hi) {
    const int pc = piece_i * NW + wave;
    if (piece_i < PWH && pc < PIECESH) {
      const int e = pc * 64 + lane;
      if (e < HALO) {
        const int row = e / HRS, q = e - row * HRS;
        const int h = q >= V ? 1 : 0, v = q - h * V;
        if (h ? hi : lo) t2_dma4(xrow0 + (size_t)row * row_stride + (h ? RS : -V) + v, buf + MAIN + pc * 64);
        else buf[MAIN + e] = fillv;
      }
    }
  };

  f32x4t acc[SLOTS][4];
  float a_nxt[4][4];
  auto load_a = [&](int tp, int ph) {
    const float4 *wp = reinterpret_cast<const float4 *>(Wp) + ((size_t)(tp * T2_NPH + ph) * 4) * 64 + lane;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float4 u = wp[m * 64];
      a_nxt[m][0] = u.x; a_nxt[m][1] = u.y; a_nxt[m][2] = u.z; a_nxt[m][3] = u.w;
    }
  };

  int tile = blockIdx.x;
  if (tile < p.total_tiles) {       // prologue: phase 0 of the first tile
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * T2_F;
    const int fr = min(T2_F, p.T - t0);
    const float *xr = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    for (int i = 0; i < PW16; ++i) dma_main(i, lds, xr, fr);
    for (int i = 0; i < PWH; ++i) dma_halo(i, lds, xr, t0 > 0, t0 + T2_F < p.T);
    load_a(0, 0);
  }

  for (; tile < p.total_tiles; tile += gridDim.x) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * T2_F;
    const int frames = min(T2_F, p.T - t0);
    const float *xg = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    float *og = out + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    const float *zg = BWD ? bwd_z + (size_t)seq * 64 * row_stride + (size_t)t0 * V : nullptr;
    const int ntile = tile + gridDim.x;
    const bool has_next = ntile < p.total_tiles;
    const int nseq = has_next ? ntile / p.tiles_per_seq : 0, nt0 = has_next ? (ntile % p.tiles_per_seq) * T2_F : 0;
    const int nfr = min(T2_F, p.T - nt0);
    const float *nxg = x + (size_t)nseq * 64 * row_stride + (size_t)nt0 * V;

#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][m][q] = bias_l[16 * m + 4 * g + q];

#pragma unroll 1
    for (int ph = 0; ph < T2_NPH; ++ph) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // own pieces of slice `ph` have landed
      __syncthreads();                                      // ... everybody's; and nobody reads the other buffer any more
      float *buf_nxt = lds + ((ph + 1) & 1) * BUF;
      const float *bufc = lds + (ph & 1) * BUF;
      const bool copy = ph + 1 < T2_NPH || has_next;
      const bool same = ph + 1 < T2_NPH;
      const float *src = same ? xg + (size_t)(ph + 1) * T2_CP * row_stride : nxg;
      const int sfr = same ? frames : nfr;
      const bool slo = same ? t0 > 0 : nt0 > 0, shi = same ? t0 + T2_F < p.T : nt0 + T2_F < p.T;
      if (XFORM) {
        // BatchNorm affine + ReLU once per element, in place in the slice that has just landed (VALU work is matrix-pipe
        // time on gfx950: transforming the B operand as it is read costs 8 VALU per 16 MFMAs, three times per element)
        float *cur = lds + (ph & 1) * BUF;
        float4 *cur4 = reinterpret_cast<float4 *>(cur);
#pragma unroll
        for (int it = 0; it < (NV4 + NW * 64 - 1) / (NW * 64); ++it) {
          const int e = it * NW * 64 + tid;
          if (e < NV4) {
            const int ch = T2_CP * ph + e / (RS / 4);
            const float sc = aff[2 * ch], sh = aff[2 * ch + 1];
            float4 v = cur4[e];
            v.x = fmaxf(fmaf(v.x, sc, sh), 0.f); v.y = fmaxf(fmaf(v.y, sc, sh), 0.f);
            v.z = fmaxf(fmaf(v.z, sc, sh), 0.f); v.w = fmaxf(fmaf(v.w, sc, sh), 0.f);
            cur4[e] = v;
          }
        }
        if (TAPS > 1)
          for (int e = tid; e < HALO; e += NW * 64) {
            const int ch = T2_CP * ph + e / HRS;
            cur[MAIN + e] = fmaxf(fmaf(cur[MAIN + e], aff[2 * ch], aff[2 * ch + 1]), 0.f);
          }
        __syncthreads();
      }

#pragma unroll 1
      for (int tp = 0; tp < TAPS; ++tp) {
        float a[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s) a[m][s] = a_nxt[m][s];
        {   // prefetch the A operands of the next (tap, phase); a share of the DMA pieces of the next slice
          int ntp = tp + 1, nph = ph;
          if (ntp == TAPS) { ntp = 0; nph = (ph + 1) & (T2_NPH - 1); }
          load_a(ntp, nph);
        }
        if (copy) {
          for (int i = tp * MPT; i < min(tp * MPT + MPT, PW16); ++i) dma_main(i, buf_nxt, src, sfr);
          for (int i = tp * HPT; i < min(tp * HPT + HPT, PWH); ++i) dma_halo(i, buf_nxt, src, slo, shi);
        }
        const float *b0 = bufc + rd[tp][0] + j0, *b1 = bufc + rd[tp][1] + j0, *b2 = bufc + rd[tp][2] + j0,
                    *b3 = bufc + rd[tp][3] + j0;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
          if (i < nslots) {
            const float b[4] = {b0[i], b1[i], b2[i], b3[i]};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int m = 0; m < 4; ++m)
                acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[s], acc[i][m], 0, 0, 0);
          }
        }
      }
      if (copy) {
        for (int i = TAPS * MPT; i < PW16; ++i) dma_main(i, buf_nxt, src, sfr);
        for (int i = TAPS * HPT; i < PWH; ++i) dma_halo(i, buf_nxt, src, slo, shi);
      }
    }

    // ---- epilogue: statistics of the tile, then the tile itself through LDS as whole rows ------------------------
    float *rs = rowstat + wave * 128;
    if (BWD) {
      if (!(p.vec && frames == T2_F)) {   // ragged tile: from the accumulators, z read where the lane's values go
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = 16 * m + 4 * g + q;
            const float sc = aff[2 * c], sh = aff[2 * c + 1], mu = bstat[2 * c], is = bstat[2 * c + 1];
            const float *zrow = zg + (size_t)c * row_stride + r * V + j0;
            float s1 = 0.f, s2 = 0.f;
            if (r < frames) {
#pragma unroll
              for (int i = 0; i < SLOTS; ++i)
                if (i < nslots) {
                  const float zz = zrow[i];
                  const float gm = fmaf(zz, sc, sh) > 0.f ? acc[i][m][q] : 0.f;
                  s1 += gm;
                  s2 = fmaf(gm, (zz - mu) * is, s2);
                }
            }
            s1 = p2r_row16_sum(s1);
            s2 = p2r_row16_sum(s2);
            if (r == 0) {
              rs[2 * c] += s1;
              rs[2 * c + 1] += s2;
            }
          }
      }
    } else if (stats_partial) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float s1 = 0.f, s2 = 0.f;
          if (r < frames) {
#pragma unroll
            for (int i = 0; i < SLOTS; ++i)
              if (i < nslots) {
                const float v = acc[i][m][q];
                s1 += v;
                s2 = fmaf(v, v, s2);
              }
          }
          s1 = p2r_row16_sum(s1);
          s2 = p2r_row16_sum(s2);
          if (r == 0) {
            rs[2 * (16 * m + 4 * g + q)] += s1;
            rs[2 * (16 * m + 4 * g + q) + 1] += s2;
          }
        }
    }
    if (p.vec && frames == T2_F) {
      float *stg = lds + ((T2_NPH - 1) & 1) * BUF;          // main part of the last phase's buffer: free now
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i)
          if (i < nslots) {
            float *d0 = stg + 4 * g * RS + r * V + j0 + i;
#pragma unroll
            for (int q = 0; q < 4; ++q) d0[q * RS] = acc[i][m][q];
          }
        constexpr int R4 = RS / 4;                          // float4 per row (212)
        constexpr int RIT = (R4 + 63) / 64;                 // 4
        float4 zv[BWD ? 2 : 1][BWD ? RIT : 1];
        if (BWD) {   // the saved activation of this wave's two rows: in flight across the staging barrier
          const float4 *z4 = reinterpret_cast<const float4 *>(zg + (size_t)(16 * m + 2 * wave) * row_stride);
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int it = 0; it < RIT; ++it) {
              const int c4 = it * 64 + lane;
              zv[rr][it] = c4 < R4 ? z4[(size_t)rr * (row_stride / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float4 *orow = reinterpret_cast<float4 *>(og + (size_t)16 * m * row_stride);
        const float4 *srow = reinterpret_cast<const float4 *>(stg);
        if (BWD) {   // a wave takes two whole rows: the per-channel sums stay in registers until the row is done
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr, c = 16 * m + row;
            const float sc = aff[2 * c], sh = aff[2 * c + 1], mu = bstat[2 * c], is = bstat[2 * c + 1];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int it = 0; it < RIT; ++it) {
              const int c4 = it * 64 + lane;
              if (c4 < R4) {
                const float4 v = srow[row * R4 + c4];
                orow[(size_t)row * (row_stride / 4) + c4] = v;
                const float4 zz = zv[rr][it];
                const float g0 = fmaf(zz.x, sc, sh) > 0.f ? v.x : 0.f, g1 = fmaf(zz.y, sc, sh) > 0.f ? v.y : 0.f;
                const float g2 = fmaf(zz.z, sc, sh) > 0.f ? v.z : 0.f, g3 = fmaf(zz.w, sc, sh) > 0.f ? v.w : 0.f;
                s1 += (g0 + g1) + (g2 + g3);
                s2 = fmaf(g0, (zz.x - mu) * is, s2); s2 = fmaf(g1, (zz.y - mu) * is, s2);
                s2 = fmaf(g2, (zz.z - mu) * is, s2); s2 = fmaf(g3, (zz.w - mu) * is, s2);
              }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
              s1 += __shfl_xor(s1, off, 64);
              s2 += __shfl_xor(s2, off, 64);
            }
            if (lane == 0) {
              rs[2 * c] += s1;
              rs[2 * c + 1] += s2;
            }
          }
        } else {
#pragma unroll
          for (int it = 0; it < (NV4 + NW * 64 - 1) / (NW * 64); ++it) {
            const int e = it * NW * 64 + tid;
            if (e < NV4) {
              const int row = e / (RS / 4), c4 = e - row * (RS / 4);
              orow[(size_t)row * (row_stride / 4) + c4] = srow[e];
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    } else {
      // ragged tiles / unaligned rows: the wave's consecutive joints of one (row, frame) are contiguous
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float *drow = og + (size_t)(16 * m + 4 * g + q) * row_stride + r * V + j0;
          if (r < frames) {
#pragma unroll
            for (int i = 0; i < SLOTS; ++i)
              if (i < nslots) drow[i] = acc[i][m][q];
          }
        }
    }
  }

  if (stats_partial) {
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += rowstat[w * 128 + tid];
      stats_partial[(size_t)blockIdx.x * 128 + tid] = t;
    }
  }
}

}  // namespace

// x (N,64,T,53) -> out (N,64,T,53).  Wp [3 taps][4 phases][4 m-tiles][64 lanes][4] f32:
// Wp[p][ph][m][16 g + r][s] = W[p][16 m + r][16 ph + 4 s + g] with W[p] (64 x 64, row = output channel) the weights
// of tap dt = p - 1 (the same permutation as the graph conv's planes); scale / shift [64] or both NULL (input
// transform relu(x*scale+shift)); bias [64] or NULL; stats_partial (optional) [n_partials][64][2].
// Call with out == NULL to query *n_partials.
template <bool XFORM, bool BWD, int TAPS>
static int tconv2_launch(const T2Params &p, int blocks, size_t lds, const float *x, const float *scale,
                         const float *shift, const float *Wp, const float *bias, float *out, float *stats_partial,
                         const float *bwd_z, const float *bwd_fin, void *stream) {
  auto kern = tconv2_kernel<53, XFORM, BWD, TAPS>;
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(kern, lds_ok);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(T2_NW * 64), lds, p2r_stream(stream), p, x, scale, shift, Wp, bias, out,
                     stats_partial, bwd_z, bwd_fin);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_stgcn_tconv2_forward(int N, int T, int V, int taps, const float *x, const float *scale, const float *shift,
                                        const float *Wp, const float *bias, float *out, float *stats_partial,
                                        int *n_partials, const float *bwd_z, const float *bwd_fin, void *stream) {
  if (N < 0 || T <= 0 || V != 53 || (taps != 1 && taps != 3) || (scale == nullptr) != (shift == nullptr)) return P2R_EINVAL;
  if ((bwd_z == nullptr) != (bwd_fin == nullptr) || (bwd_z && (scale || !stats_partial))) return P2R_EINVAL;
  if (n_partials) *n_partials = 0;
  if (N == 0) return P2R_OK;
  T2Params p;
  p.T = T;
  p.tiles_per_seq = p2r_cdiv(T, T2_F);
  const long long tiles = (long long)N * p.tiles_per_seq;
  if (tiles > 0x7fffffffLL) return P2R_EINVAL;
  p.total_tiles = (int)tiles;
  p.vec = (((size_t)T * V) % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0) ? 1 : 0;
  const int blocks = (int)(tiles < 256 ? tiles : 256);
  if (n_partials) *n_partials = blocks;
  if (!out) return P2R_OK;
  const size_t lds = (size_t)2 * (T2_CP * T2_F * V + T2_CP * 2 * V) * sizeof(float) + (size_t)T2_NW * 128 * sizeof(float) +
                     (size_t)(128 + 64 + 128) * sizeof(float);
  if (lds > 160 * 1024) return P2R_EINVAL;
#define P2R_T2(XF, BW) (taps == 3 ? tconv2_launch<XF, BW, 3>(p, blocks, lds, x, scale, shift, Wp, bias, out, stats_partial, bwd_z, bwd_fin, stream) \
                                   : tconv2_launch<XF, BW, 1>(p, blocks, lds, x, scale, shift, Wp, bias, out, stats_partial, bwd_z, bwd_fin, stream))
  if (bwd_z) return P2R_T2(false, true);
  if (scale) return P2R_T2(true, false);
  return P2R_T2(false, false);
#undef P2R_T2
}
