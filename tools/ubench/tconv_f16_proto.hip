// PROTOTYPE, not part of the product: tconv_bf16_proto.hip with the operands split into TWO fp16 parts, the residual
// scaled by 2^11 (x = x1 + 2^-11 x2': 22 significand bits, nothing in fp16's subnormals), three MFMAs per product
// (W1 X1 | W1 X2' + W2' X1 in two accumulators, combined as hi + 2^-11 lo), 4 bytes per element in LDS instead of 6
// (tools/ubench/mfma_bf16_split.hip: 1.3e-7 of range vs float64, the fp32 MFMA itself 2.5e-7).  fp16's RANGE is the
// caveat: |x| <= 65504 -- fine behind a BatchNorm + ReLU, not guaranteed for gradients.  Built and driven by tools/dev_tconv_bf16.py, which checks it against the product's
// exact-fp32 kernel (p2r_stgcn_tconv3_forward) and a float64 convolution and times both.
//
//     out[n,c,t,w] = bias[c] + sum_{p<3} sum_ci W[p][c][ci] * h[n,ci,t+p-1,w],   h = relu(x*scale+shift) or x
//
// How it differs from stgcn_tconv3.hip, and why:
//   * v_mfma_f32_16x16x32_bf16 reduces over K = 32 channels per instruction: channel phases of 32 (two per tile).
//   * The B operand is the input itself, so the split into three bf16 parts happens ONCE per element, in the staging
//     pass that already applies BatchNorm + ReLU, and serves 3 taps x 64 output rows: the MFMA loop has no vector
//     arithmetic at all (B operands are three 16-byte LDS reads per (tap, 16 columns)).
//   * LDS holds only bf16 operands, [plane][8-channel group][column][8 channels]: 16-byte reads of consecutive columns
//     are conflict-free and the four channel groups of a phase are 21 x 256 bytes apart.  6 bytes per element instead
//     of 4 make a 16-frame tile of 32 channels 160 KB, so the tile is 4 frames (+ one halo frame each side: 1.5x the
//     read traffic) with two phase buffers of 64.5 KB; the input is staged through REGISTERS (range-checked buffer
//     loads issued before a phase's MFMAs, transformed and written behind them), there is no fp32 image in LDS.
//   * A operands (W of three taps, two phases, split into three bf16 planes) stay in registers for the whole kernel:
//     wave w owns the 16 output rows 16 (w & 3) .. (72 VGPRs) and 7 of the 14 sixteen-column tiles (w >> 2); the two
//     waves of a SIMD (w, w + 4) share the rows and split the columns.  A B operand is read by four waves (3 x 16 bytes
//     per six MFMAs): 2 k LDS cycles per phase, far from a limit.
//   * Two accumulators per column tile: the five small terms and the leading product W1 X1 are summed separately and
//     added at the end (shorter dependent MFMA chains; the small terms are not absorbed one by one into a large sum).
//   * n-tile = 16 CONSECUTIVE (frame, joint) columns: a tap is a column offset of 53, no per-joint tiles (the graph
//     conv needs those for its wave-uniform neighbour lists; the temporal conv does not).
// Not in the prototype: the statistics / BatchNorm-backward epilogues of the product kernel, the single-tap instance.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 bf8 __attribute__((ext_vector_type(8)));   // (name kept from the bf16 prototype)


namespace {
#ifndef PROTO_F
#define PROTO_F 4
#endif
constexpr int F = PROTO_F, V = 53, C = 64, NW = 8;
constexpr int OC = F * V;            // 212 output columns per tile
constexpr int WC = (F + 2) * V;      // 318 window columns (frames t0-1 .. t0+4)
constexpr int NT = (OC + 15) / 16;   // column tiles: 14 (F = 4) / 27 (F = 8)
constexpr int CP = (NT * 16 + 2 * V + 15) / 16 * 16;   // columns per 8-channel row in LDS: 336 / 544 (x 16 B = a multiple of 256 B)
constexpr int NTW = (NT + 1) / 2;    // column tiles per wave: 7 / 14
constexpr int SC = (WC + 63) / 64;   // staged columns per thread: 5 / 9
constexpr int BUF = 2 * 4 * CP * 8;  // fp16 elements per phase buffer (43,008 bytes)

struct Params { int T, tiles_per_seq, total_tiles; };

__device__ __forceinline__ void split2(const float (&x)[8], bf8 &p1, bf8 &p2) {    // x = p1 + 2^-11 p2
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 a = (_Float16)x[i];
    p1[i] = a; p2[i] = (_Float16)((x[i] - (float)a) * 2048.f);
  }
}

template <bool XFORM>
__global__ __launch_bounds__(NW * 64, 2) void tconv3b_kernel(Params p, const float *__restrict__ x,
                                                             const float *__restrict__ scale,
                                                             const float *__restrict__ shift,
                                                             const float *__restrict__ W,      // [3][64][64] (tap, c, ci)
                                                             const float *__restrict__ bias, float *__restrict__ out) {
  extern __shared__ _Float16 lds[];                    // [2][3][4][CP][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int rs = p.T * V;                            // floats per channel row

  // ---- A operands: W[tap][16 mq + r][32 ph + 8 g + i], three bf16 planes, for the whole kernel -----------------------
  const int mq = wave & 3, nh = wave >> 2;
  const int nt0 = NTW * nh;                          // column tiles nt0 .. nt0 + NTW - 1 (those < NT)
  bf8 a[2][3][2];                                    // [phase][tap][part]
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) {
      float w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = W[((size_t)tp * C + 16 * mq + r) * C + 32 * ph + 8 * g + i];
      split2(w, a[ph][tp][0], a[ph][tp][1]);
    }

  // ---- staging: wave = (8-channel group kg, half hc of its channels); lane = column; 5 columns x 4 channels per thread --
  const int kg = wave & 3, hc = wave >> 2;
  float pre[SC][4];
  auto issue_loads = [&](int tile, int ph) __attribute__((always_inline)) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * F;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = 32 * ph + 8 * kg + 4 * hc + j;
      // the window of this channel row as one descriptor: base = frame t0 - 1 (may lie in front of the row: the offset
      // of a lane is then compared against the row's byte range below), 318 columns
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float *>(x + ((size_t)seq * C + ch) * rs), 0, 4 * rs, 0x00020000);
#pragma unroll
      for (int i = 0; i < SC; ++i) {
        // a column outside the channel row (frame -1 of the first tile, frame T of the last; a negative offset wraps to
        // a huge unsigned one) comes back as zero from the range check; columns >= 318 of the last round land in the
        // unused tail of the LDS row
        const int c = lane + 64 * i;
        const int off = (c < WC) ? 4 * ((t0 - 1) * V + c) : -4;
#ifdef ABL_NO_LOAD
        pre[i][j] = (float)off;
#else
        pre[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
#endif
      }
    }
  };
  auto write_phase = [&](int tile, int ph, _Float16 *buf) __attribute__((always_inline)) {
    const int t0 = (tile % p.tiles_per_seq) * F;
    float sc[4], sh[4];
    if (XFORM) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { sc[j] = scale[32 * ph + 8 * kg + 4 * hc + j]; sh[j] = shift[32 * ph + 8 * kg + 4 * hc + j]; }
    }
#pragma unroll
    for (int i = 0; i < SC; ++i) {
      const int c = lane + 64 * i;
      if (c >= CP) continue;
      const int gc = (t0 - 1) * V + c;
      const bool in = c < WC && gc >= 0 && gc < rs;  // outside the sequence: the zero padding of the convolution
      typedef _Float16 bf4 __attribute__((ext_vector_type(4)));
      bf4 q1, q2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = pre[i][j];
        if (XFORM) u = fmaxf(fmaf(u, sc[j], sh[j]), 0.f);
        u = in ? u : 0.f;
        const _Float16 a1 = (_Float16)u;
        q1[j] = a1; q2[j] = (_Float16)((u - (float)a1) * 2048.f);
      }
      *reinterpret_cast<bf4 *>(buf + ((size_t)(0 * 4 + kg) * CP + c) * 8 + 4 * hc) = q1;
      *reinterpret_cast<bf4 *>(buf + ((size_t)(1 * 4 + kg) * CP + c) * 8 + 4 * hc) = q2;
    }
  };
  // columns WC .. CP-1 of every row are read by the padding lanes of the last column tile: keep them finite
  constexpr int C0 = SC * 64 < CP ? SC * 64 : CP;      // columns [C0, CP) are never staged
  for (int e = tid; e < 2 * 2 * 4 * (CP - C0); e += NW * 64) {
    const int row = e / (CP - C0 > 0 ? CP - C0 : 1), c = C0 + e % (CP - C0 > 0 ? CP - C0 : 1);
    *reinterpret_cast<bf8 *>(lds + ((size_t)row * CP + c) * 8) = bf8{0, 0, 0, 0, 0, 0, 0, 0};
  }

  // tiles in XCD-local order (block b runs on XCD b % 8): the 32 workgroups of an XCD walk 32 consecutive tiles per round,
  // so the halo frames two neighbouring tiles share are hits in that XCD's L2 instead of a second HBM fetch
  const int per_xcd = (gridDim.x & 7) == 0 ? (int)(gridDim.x >> 3) : 0;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  auto tile_of = [&](int i) { return per_xcd ? (i * 8 + xcd) * per_xcd + slot : (int)(blockIdx.x + i * gridDim.x); };
  int it = 0;
  int tile = tile_of(0);
  if (tile >= p.total_tiles) return;
  issue_loads(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  write_phase(tile, 0, lds);
  // Waves 0-3 (one per SIMD) stage the next phase BEFORE their MFMAs, waves 4-7 behind them: the vector arithmetic of one
  // wave of a SIMD then runs under the bf16 MFMAs of the other (they use different pipes).  The early wave holds the
  // input of phase k + 1 in registers when phase k starts, so its first set is fetched here.
  const bool early = wave < 4;
  if (early) { issue_loads(tile, 1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __syncthreads();

  int cur = 0;
  for (; tile < p.total_tiles; tile = tile_of(++it)) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * F;
    const int ntile = tile_of(it + 1);
    f4 hi[NTW], lo[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { hi[i][q] = bias ? bias[16 * mq + 4 * g + q] : 0.f; lo[i][q] = 0.f; }
    }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      // the next phase's input on its way while this phase multiplies
      const bool more = ph == 0 || ntile < p.total_tiles;
      const int wtile = ph == 0 ? tile : ntile;        // tile of the phase staged during this one
      if (early) {
        if (more) write_phase(wtile, ph ^ 1, lds + (size_t)(cur ^ 1) * BUF);
        // ... and the phase after that on its way: (wtile, ph) if ph == 1 follows (tile, 0) ... i.e. two phases ahead
        const int t2 = ph == 0 ? ntile : tile_of(it + 1);    // phase k + 2: (ntile, 0) after (tile, 1); (ntile, 1) after (ntile, 0)
        if (t2 < p.total_tiles) issue_loads(t2, ph);
      } else if (more) {
        issue_loads(wtile, ph ^ 1);
      }
      const _Float16 *buf = lds + (size_t)cur * BUF;
      auto bfrag = [&](int plane, int col) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf8 *>(buf + ((size_t)(plane * 4 + g) * CP + col) * 8);
      };
#ifdef ABL_NO_MFMA
#define MF(acc, ap, bp) acc[0] += (float)bp[0] * (float)a[ph][tp][ap][0];
#else
#define MF(acc, ap, bp) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ph][tp][ap], bp, acc, 0, 0, 0);
#endif
#pragma unroll
      for (int i = 0; i + 1 < NTW; i += 2) {           // column tiles in pairs: two independent accumulator chains
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
          const int c0 = 16 * (nt0 + i) + r + tp * V, c1 = c0 + 16;      // window column of the output column at this tap
          const bf8 b1 = bfrag(0, c0), b2 = bfrag(1, c0);
          const bf8 d1 = bfrag(0, c1), d2 = bfrag(1, c1);
          MF(lo[i], 0, b2) MF(lo[i + 1], 0, d2) MF(hi[i], 0, b1) MF(hi[i + 1], 0, d1) MF(lo[i], 1, b1) MF(lo[i + 1], 1, d1)
        }
      }
      if constexpr (NTW & 1) {
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {               // the odd tile
          const int c0 = 16 * (nt0 + NTW - 1) + r + tp * V;
          const bf8 b1 = bfrag(0, c0), b2 = bfrag(1, c0);
          MF(lo[NTW - 1], 0, b2) MF(hi[NTW - 1], 0, b1) MF(lo[NTW - 1], 1, b1)
        }
      }
#undef MF
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (more && !early) write_phase(wtile, ph ^ 1, lds + (size_t)(cur ^ 1) * BUF);
      __syncthreads();
      cur ^= 1;
    }
    // D: row 4 g + q of the 16-row block, column r of the 16-column tile
    float *og = out + (size_t)seq * C * rs + (size_t)t0 * V;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const int col = 16 * (nt0 + i) + r;
      if (col >= OC) continue;
#pragma unroll
#ifdef ABL_NO_STORE
      for (int q = 0; q < 4; ++q) if (hi[i][q] + lo[i][q] == 1.2345f) og[0] = 1.f;
#else
      for (int q = 0; q < 4; ++q) og[(size_t)(16 * mq + 4 * g + q) * rs + col] = hi[i][q] + lo[i][q] * (1.f / 2048.f);
#endif
    }
  }
}
}  // namespace

// x (N,64,T,53) f32, W (3,64,64) = [tap][c][ci], scale / shift (64) or NULL (no input transform), bias (64) or NULL ->
// out (N,64,T,53).  T % 4 == 0.
extern "C" int proto_tconv3h_forward(int N, int T, const float *x, const float *scale, const float *shift, const float *W,
                                     const float *bias, float *out, void *stream) {
  if (N <= 0 || T <= 0 || T % F != 0) return 1;
  Params p;
  p.T = T; p.tiles_per_seq = T / F;
  p.total_tiles = N * p.tiles_per_seq;
  const int blocks = p.total_tiles < 256 ? p.total_tiles : 256;
  const size_t lds = (size_t)2 * BUF * sizeof(_Float16);
  hipError_t e;
  if (scale) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tconv3b_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(tconv3b_kernel<true>, dim3(blocks), dim3(NW * 64), lds, (hipStream_t)stream, p, x, scale, shift, W, bias, out);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tconv3b_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(tconv3b_kernel<false>, dim3(blocks), dim3(NW * 64), lds, (hipStream_t)stream, p, x, scale, shift, W, bias, out);
  }
  return (int)hipGetLastError();
}
