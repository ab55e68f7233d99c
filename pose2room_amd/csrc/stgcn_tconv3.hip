// stgcn_tconv3.hip -- temporal (3,1) / pointwise convolution of st_gcn_block fused with the preceding BatchNorm + ReLU,
// third generation (statically scheduled), gfx950.
//
// Same operator and data movement as stgcn_tconv2.hip (reference models/p2rnet/modules/stgcn_layers.py:399-411 and its
// data gradient):
//     out[n,c,t,w] = bias[c] + sum_{p<TAPS} sum_ci W[p][c][ci] * h[n,ci,t+p-(TAPS-1)/2,w],  h = relu(x*scale+shift) or x
// rebuilt the way stgcn_gcn3.hip rebuilt the graph convolution: every wave runs its own straight-line program (its
// joints are ds_read immediates, the taps are unrolled, the A-operand sets ping-pong by name, the DMA pieces have
// precomputed per-lane offsets and fixed places), the 16 MFMAs of a (tap, joint) unit are one assembly block that
// accumulates in place, and the waves of a workgroup never share code between the first instruction and the last.
// The second generation paid for the opposite on every count: the compiler renamed the accumulators along the MFMA
// chains and reconciled them through scratch when the tap loop was unrolled (tried in round 3: 100 spilled VGPRs), the
// piece offsets were recomputed with 64-bit multiplies per piece, and the pieces of the next slice, spread over the
// three tap visits of a 4.6 us phase, were still in flight at the phase's end.
//
// Full tiles of 16-byte aligned rows only (T % 16 == 0): anything else runs on stgcn_tconv2.hip.
#include "p2r_common.h"

// Cycle trace (profiling hook, off in the product build): with -DP2R_CYCLE_TRACE the waves of workgroup 7 stamp
// s_memtime at the section boundaries of their fourth tile; tools/dev_t3_trace.py reads the stamps back through
// p2r_debug_t3_trace (make -C pose2room_amd/csrc clean all FLAGS="... -DP2R_CYCLE_TRACE").  DESIGN.md quotes it.
#ifdef P2R_CYCLE_TRACE
__device__ unsigned long long t3_trace[8 * 32];
extern "C" int p2r_debug_t3_trace(unsigned long long *dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(t3_trace), sizeof(t3_trace));
}
#define T3_TRACE_TILE(tile) const bool trace_on = blockIdx.x == 7 && (tile) == 7 + 3 * (int)gridDim.x
#define T3_MARK(i) do { if (trace_on && lane == 0) t3_trace[wave * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define T3_TRACE_TILE(tile)
#define T3_MARK(i)
#endif
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int T3_F = 16, T3_CP = 16, T3_NPH = 4, T3_NW = 8, T3_SLOTS = 7, T3_V = 53;
constexpr int T3_ST = 3;   // floats per (wave, row) statistics entry: (sum, sum of squares) about the pivot, pivot
constexpr int T3_RS = T3_F * T3_V;                  // 848
// LDS image of a slice (round 5): 16 channel rows of T3_RSP floats; a row is a contiguous WINDOW of the tensor's channel
// row -- [3 unused][frame t0-1][frames t0 .. t0+15][frame t0+16][19 unused] -- so the halo frames of the three-tap
// instances arrive with the same 16-byte DMA stream as the slice itself (frame t0 sits 56 floats = 14 pieces into the
// window: 16-byte aligned on both sides), and EVERY tap reads frame r + tap - 1 at one stride of 53 floats.  Row stride
// 976 = 16 mod 32: the two channel rows of a 32-lane read group use disjoint banks (21 r and 16 + 21 r, r < 16).
// Before, the halo frames lived in a separate [16][2 x 53] area filled by 27 four-byte DMA instructions per slice; the
// one lane of a group that read there landed on a bank its group already used in taps 0 and 2: one extra LDS cycle on a
// two-cycle read = the 28-39 % SQ_LDS_BANK_CONFLICT of profiles/r4_tconv_mfma_util.json.
// Single-tap instances keep the plain [16][848] image.
constexpr int T3_RSP3 = 976;                        // window length of the three-tap instances (floats)
constexpr int T3_OFF3 = 56;                         // frame t0 inside the window
constexpr int T3_BUF = T3_CP * T3_RSP3;             // floats per phase buffer (62,464 bytes; single-tap: 54,272 used)

struct T3Params {
  int T, tiles_per_seq, total_tiles;
};

__device__ __forceinline__ unsigned t3_lds_addr(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
__device__ __forceinline__ void t3_dma16(const float *base, unsigned voff, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(t3_lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}
__device__ __forceinline__ void t3_dma4(const float *base, unsigned voff, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(t3_lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}
// A operands (the weights of one (tap, phase): 4 x 16 bytes per lane) live in registers the compiler does not
// allocate -- the kernels are limited to v0..v223 (amdgpu_num_vgpr), set 0 = v[224:239], set 1 = v[240:255], named in
// the assembly text -- and are waited for
// explicitly.  Two reasons.  Left to the compiler, its s_waitcnt in front of their first use counts only the loads it
// knows about -- vmcnt(0..4) -- while the hardware counter also holds the slice pieces issued after them by t3_dma16 /
// t3_dma4; vector-memory operations retire in order, so every wave stood at the top of a visit until most of the NEXT
// slice had arrived (all 256 workgroups at once: ~3 us of a 13 us phase).  With the count spelled out (the pieces
// issued after the operands may stay in flight) the copy runs under the MFMAs.  And a load whose destination is a
// compiler-visible register cannot be waited for by hand: the compiler considers the value present after the load
// statement and is free to copy it before the wait (it did: the set rotation was hoisted above the s_waitcnt).
#define T3_SET0 "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239"
#define T3_SET1 "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
template <int SET>
__device__ __forceinline__ void t3_load_a(const float *base, unsigned lane_off) {
  if constexpr (SET == 0)
    asm volatile("global_load_dwordx4 v[224:227], %0, %1\n\tglobal_load_dwordx4 v[228:231], %0, %1 offset:1024\n\tglobal_load_dwordx4 v[232:235], %0, %1 offset:2048\n\tglobal_load_dwordx4 v[236:239], %0, %1 offset:3072"
                 :: "v"(lane_off), "s"(base) : "memory", T3_SET0);
  else
    asm volatile("global_load_dwordx4 v[240:243], %0, %1\n\tglobal_load_dwordx4 v[244:247], %0, %1 offset:1024\n\tglobal_load_dwordx4 v[248:251], %0, %1 offset:2048\n\tglobal_load_dwordx4 v[252:255], %0, %1 offset:3072"
                 :: "v"(lane_off), "s"(base) : "memory", T3_SET1);
}
template <int LATER>   // LATER: vector-memory operations issued after the awaited ones that may still be in flight
__device__ __forceinline__ void t3_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LATER) : "memory");
}
__device__ __forceinline__ void t3_rotate_a() {   // set 0 <- set 1
  asm volatile("v_mov_b32 v224, v240\n\tv_mov_b32 v225, v241\n\tv_mov_b32 v226, v242\n\tv_mov_b32 v227, v243\n\tv_mov_b32 v228, v244\n\tv_mov_b32 v229, v245\n\tv_mov_b32 v230, v246\n\tv_mov_b32 v231, v247\n\tv_mov_b32 v232, v248\n\tv_mov_b32 v233, v249\n\tv_mov_b32 v234, v250\n\tv_mov_b32 v235, v251\n\tv_mov_b32 v236, v252\n\tv_mov_b32 v237, v253\n\tv_mov_b32 v238, v254\n\tv_mov_b32 v239, v255" ::: T3_SET0);
}
// the 16 MFMAs of one (tap, joint) unit, accumulating in place (see stgcn_gcn3.hip); A[m][s] = v[224 + 16 SET + 4 m + s]
template <int SET>
__device__ __forceinline__ void t3_mfma16(f32x4 (&acc)[4], const float (&b)[4]) {
  if constexpr (SET == 0)
    asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, v224, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, v228, %4, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, v232, %4, %2\n\tv_mfma_f32_16x16x4_f32 %3, v236, %4, %3\n\t"
      "v_mfma_f32_16x16x4_f32 %0, v225, %5, %0\n\tv_mfma_f32_16x16x4_f32 %1, v229, %5, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, v233, %5, %2\n\tv_mfma_f32_16x16x4_f32 %3, v237, %5, %3\n\t"
      "v_mfma_f32_16x16x4_f32 %0, v226, %6, %0\n\tv_mfma_f32_16x16x4_f32 %1, v230, %6, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, v234, %6, %2\n\tv_mfma_f32_16x16x4_f32 %3, v238, %6, %3\n\t"
      "v_mfma_f32_16x16x4_f32 %0, v227, %7, %0\n\tv_mfma_f32_16x16x4_f32 %1, v231, %7, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, v235, %7, %2\n\tv_mfma_f32_16x16x4_f32 %3, v239, %7, %3"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
  else
    asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, v240, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, v244, %4, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, v248, %4, %2\n\tv_mfma_f32_16x16x4_f32 %3, v252, %4, %3\n\t"
      "v_mfma_f32_16x16x4_f32 %0, v241, %5, %0\n\tv_mfma_f32_16x16x4_f32 %1, v245, %5, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, v249, %5, %2\n\tv_mfma_f32_16x16x4_f32 %3, v253, %5, %3\n\t"
      "v_mfma_f32_16x16x4_f32 %0, v242, %6, %0\n\tv_mfma_f32_16x16x4_f32 %1, v246, %6, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, v250, %6, %2\n\tv_mfma_f32_16x16x4_f32 %3, v254, %6, %3\n\t"
      "v_mfma_f32_16x16x4_f32 %0, v243, %7, %0\n\tv_mfma_f32_16x16x4_f32 %1, v247, %7, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, v251, %7, %2\n\tv_mfma_f32_16x16x4_f32 %3, v255, %7, %3"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}

// The 4 MFMAs of a quarter unit: one 16-row block MQ of one joint (see the joint assignment in t3_wave_main)
template <int SET, int MQ>
__device__ __forceinline__ void t3_mfma4(f32x4 &acc, const float (&b)[4]) {
  constexpr int A0 = 224 + 16 * SET + 4 * MQ;
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x4_f32 %0, v[%5:%5], %1, %0\n\tv_mfma_f32_16x16x4_f32 %0, v[%6:%6], %2, %0\n\t"
               "v_mfma_f32_16x16x4_f32 %0, v[%7:%7], %3, %0\n\tv_mfma_f32_16x16x4_f32 %0, v[%8:%8], %4, %0"
               : "+v"(acc) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "n"(A0), "n"(A0 + 1), "n"(A0 + 2), "n"(A0 + 3));
}

// XFORM: input transform relu(x * scale + shift) applied in place to each slice; BWD (data-gradient instance): the
// statistics epilogue emits the two sums of the BatchNorm + ReLU backward of the layer in front (see stgcn_tconv2.hip).
template <bool XFORM, bool BWD, int TAPS, int WAVE, bool ADDCT = false>
__device__ __forceinline__ void t3_wave_main(const T3Params &p, float *lds, const float *__restrict__ x,
                                             const float *__restrict__ Wp, float *__restrict__ out, bool want_stats,
                                             const float *__restrict__ bwd_z, const float *__restrict__ add_ct = nullptr) {
  constexpr int V = T3_V, NW = T3_NW, SLOTS = T3_SLOTS, RS = T3_RS, BUF = T3_BUF;
  constexpr int RSP = TAPS > 1 ? T3_RSP3 : RS;        // LDS row stride = window length
  constexpr int OFF0 = TAPS > 1 ? T3_OFF3 : 0;        // frame t0 inside a row
  constexpr int NV4 = T3_CP * RSP / 4;                // 16-byte pieces of a slice: 3904 = 61 x 64 / 3392 = 53 x 64
  constexpr int PIECES16 = NV4 / 64;
  static_assert(NV4 == PIECES16 * 64, "partial piece");
  constexpr int wave = WAVE;
  // joints of this wave: consecutive runs 7,7,7,7,6,6,6,6 = joints 0..51; joint 52 is split by its four 16-row blocks
  // over waves 4..7.  Waves w and w + 4 share a SIMD: 13.25 units per tap on each (with whole joints only, 53 joints
  // over four SIMDs are 14,13,13,13 and three SIMDs idle 7 % of every phase: cycle trace, round 3)
  constexpr int j0 = WAVE < 4 ? 7 * WAVE : 28 + 6 * (WAVE - 4);
  constexpr int nslots = WAVE < 4 ? 7 : 6;
  constexpr bool QUARTER = WAVE >= 4;                  // + row block MQ of joint 52
  constexpr int MQ = WAVE >= 4 ? WAVE - 4 : 0;
  constexpr int JQ = T3_V - 1;
  constexpr int NU = nslots + (QUARTER ? 1 : 0);       // units per tap visit (the last one a quarter)
  constexpr int PW16 = (PIECES16 + NW - 1) / NW;
  float *rowstat = lds + 2 * BUF;                     // [NW][64][T3_ST]
  float *aff = rowstat + NW * 64 * T3_ST;                  // [64][2] (scale, shift) of the input transform
  float *bias_l = aff + 128;                          // [64]
  float *bstat = bias_l + 64;                         // [64][2] (mean, invstd) of the BWD epilogue

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int g = lane >> 4, r = lane & 15;
  const size_t row_stride = (size_t)p.T * V;
  const float fillv = XFORM ? __int_as_float(0x7fc00000) : 0.f;   // outside the sequence: NaN -> relu gives the zero padding

  // lane's read position inside a buffer for (tap, k-step s): channel 4s+g, frame r+tap-(TAPS-1)/2 (-1 .. 16: the halo
  // frames are part of the row), the wave's first joint; the slot's joint is an immediate
  unsigned rd[TAPS][4];
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp) {
    const int f = r + tp - (TAPS - 1) / 2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ch = 4 * s + g;
      rd[tp][s] = (unsigned)(ch * RSP + OFF0 + f * V + j0) * 4u;
    }
  }

  // this wave's DMA pieces of a slice; offsets relative to the start of channel row 0's window (frame t0, minus OFF0
  // floats).  lmask / rmask, bit i: this lane's piece i lies in the window's part in front of frame t0 / behind frame
  // t0+15 -- read only when that side of the tile is inside the sequence
  unsigned moff[PW16];
  unsigned lmask = 0, rmask = 0;
#pragma unroll
  for (int i = 0; i < PW16; ++i) {
    const int pc = i * NW + wave, e = pc * 64 + lane;
    const int row = e / (RSP / 4), c4 = e - row * (RSP / 4);
    moff[i] = pc < PIECES16 ? (unsigned)(((size_t)row * row_stride + 4 * c4) * sizeof(float)) : 0xffffffffu;
    if (TAPS > 1 && pc < PIECES16) {
      lmask |= (unsigned)(c4 < OFF0 / 4) << i;
      rmask |= (unsigned)(c4 >= (OFF0 + RS) / 4) << i;
    }
  }
  // base = address of channel row 0's window; lo / hi: frame t0-1 / t0+16 exists in the sequence.  Inside a sequence
  // (62 of 64 tiles at T = 1024) every piece is one unconditional DMA; at its ends the pieces outside are not read (the
  // window would leave the channel row -- at the first and last row of the tensor, the allocation) but filled with the
  // padding value.  Either way a wave issues N16 vector-memory operations per slice (t3_wait_vm<N16>): the 32 pieces
  // of a row boundary never fill a whole 64-lane instruction.
  constexpr int N16 = (PIECES16 - WAVE + NW - 1) / NW;
  // t3_wait_vm<N16> counts one vector-memory operation per piece instruction: at a sequence edge the lanes outside take
  // the LDS fill instead, and the count only holds while every 64-lane piece instruction keeps at least one DMA lane
  // (the compiler's execz skip would otherwise drop the instruction and the wait would return early)
  static_assert(TAPS == 1 || ((T3_RSP3 - T3_RS) / 4 < 64 && T3_OFF3 / 4 < 64 && (T3_RSP3 - T3_OFF3 - T3_RS) / 4 < 64),
                "an all-outside run of 16-byte pieces must be shorter than one 64-lane piece instruction");
  auto copy_slice = [&](float *buf, const float *base, bool lo, bool hi) {
    if (TAPS == 1 || (lo && hi)) {
#pragma unroll
      for (int i = 0; i < N16; ++i) t3_dma16(base, moff[i], buf + (i * NW + wave) * 256);
    } else {
#pragma unroll
      for (int i = 0; i < N16; ++i) {
        const bool outside = (((lmask >> i) & 1u) && !lo) || (((rmask >> i) & 1u) && !hi);
        if (!outside) t3_dma16(base, moff[i], buf + (i * NW + wave) * 256);
        else *reinterpret_cast<float4 *>(buf + (i * NW + wave) * 256 + lane * 4) = float4{fillv, fillv, fillv, fillv};
      }
    }
  };

  f32x4 acc[SLOTS][4];
  f32x4 accq = {0.f, 0.f, 0.f, 0.f};                   // QUARTER: rows 16 MQ .. 16 MQ + 15 of joint 52
  // A-operand sets: visit `tp` of a phase multiplies with set tp & 1 and loads the next visit's operands into the
  // other one; a phase ends with the next phase's first operands in set 1 (TAPS is odd), moved to set 0 once they
  // have arrived
  static_assert(TAPS & 1, "set rotation");
  const unsigned a_lane = (unsigned)lane * 16u;
  auto a_base = [&](int tp, int ph) { return Wp + (size_t)(tp * T3_NPH + ph) * 4 * 64 * 4; };

  // BWD: a wave stores rows 2 wave + {0, 1} of every 16-row group, i.e. the same eight channels in every tile: the
  // two sums of each are kept per lane across the tiles and combined over the lanes once, at the end of the kernel
  // (per tile, the 6-step lane reduction and the LDS update cost a fifth of the tile: cycle trace, round 3)
  float bsum[BWD ? 4 : 1][2][2];
#pragma unroll
  for (int m = 0; m < (BWD ? 4 : 1); ++m)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) bsum[m][rr][0] = bsum[m][rr][1] = 0.f;

  int tile = blockIdx.x;
  if (tile < p.total_tiles) {       // prologue: phase 0 of the first tile
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * T3_F;
    copy_slice(lds, x + (size_t)seq * 64 * row_stride + (size_t)t0 * V - OFF0, t0 > 0, t0 + T3_F < p.T);
  }
  t3_load_a<1>(a_base(0, 0), a_lane);

  for (; tile < p.total_tiles; tile += gridDim.x) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * T3_F;
    const float *xg = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    float *og = out + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    const float *zg = BWD ? bwd_z + (size_t)seq * 64 * row_stride + (size_t)t0 * V : nullptr;
    const int ntile = tile + gridDim.x;
    const bool has_next = ntile < p.total_tiles;
    const int nseq = has_next ? ntile / p.tiles_per_seq : 0, nt0 = has_next ? (ntile % p.tiles_per_seq) * T3_F : 0;
    const float *nxg = x + (size_t)nseq * 64 * row_stride + (size_t)nt0 * V;
    T3_TRACE_TILE(tile);
    T3_MARK(0);

    if constexpr (ADDCT) {
      // accumulators start from bias[c] + add_ct[n, c, t]: a (sample, channel, frame) term broadcast over the joints
      // (the position embedding added to the joint embedding, stgcn.py:129-130) costs 16 four-byte loads per lane and
      // tile here instead of a pass over the (N, 64, T, 53) output
      const float *ab = add_ct + ((size_t)seq * 64 + 4 * g) * p.T + t0 + r;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = bias_l[16 * m + 4 * g + q] + ab[(size_t)(16 * m + q) * p.T];
#pragma unroll
          for (int i = 0; i < SLOTS; ++i) acc[i][m][q] = v;
          if (QUARTER && m == MQ) accq[q] = v;
        }
    } else {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[i][m][q] = bias_l[16 * m + 4 * g + q];
      if (QUARTER) {
#pragma unroll
        for (int q = 0; q < 4; ++q) accq[q] = bias_l[16 * MQ + 4 * g + q];
      }
    }

#pragma unroll 1
    for (int ph = 0; ph < T3_NPH; ++ph) {
      T3_MARK(1 + 4 * ph);
      t3_wait_vm<0>();                                      // own pieces of slice `ph` and the phase's first operands have landed
      t3_rotate_a();
      T3_MARK(2 + 4 * ph);
      __syncthreads();                                      // ... everybody's; and nobody reads the other buffer any more
      float *buf_nxt = lds + ((ph + 1) & 1) * BUF;
      const bool copy = ph + 1 < T3_NPH || has_next;
      const bool same = ph + 1 < T3_NPH;
      const float *src = (same ? xg + (size_t)(ph + 1) * T3_CP * row_stride : nxg) - OFF0;
      const bool slo = same ? t0 > 0 : nt0 > 0, shi = same ? t0 + T3_F < p.T : nt0 + T3_F < p.T;
      if (XFORM) {
        // BatchNorm affine + ReLU once per element, in place in the slice that has just landed
        // 32 threads per channel row (16 rows x 32 = the workgroup), 7 float4 each at a stride of 32: the row's
        // constants are read once and the addresses are immediates of one base (a thread walking the slice linearly
        // crossed rows: a division and two constant reads per float4)
        float *cur = lds + (ph & 1) * BUF;
        {
          const int row = tid >> 5, col = tid & 31;
          const int ch = T3_CP * ph + row;
          const float sc = aff[2 * ch], sh = aff[2 * ch + 1];
          float4 *rp = reinterpret_cast<float4 *>(cur + row * RSP) + col;
          constexpr int R4 = TAPS > 1 ? (OFF0 + RS + V + 3) / 4 : RS / 4;      // float4 per row: 240 (halo frames included) / 212
#pragma unroll
          for (int it = 0; it < (R4 + 31) / 32; ++it)
            if (32 * it + 31 < R4 || 32 * it + col < R4) {
              float4 v = rp[32 * it];
              v.x = fmaxf(fmaf(v.x, sc, sh), 0.f); v.y = fmaxf(fmaf(v.y, sc, sh), 0.f);
              v.z = fmaxf(fmaf(v.z, sc, sh), 0.f); v.w = fmaxf(fmaf(v.w, sc, sh), 0.f);
              rp[32 * it] = v;
            }
        }
        __syncthreads();
      }

      T3_MARK(3 + 4 * ph);
      const char *bufc = reinterpret_cast<const char *>(lds + (ph & 1) * BUF);
      float b_cur[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) b_cur[s] = *reinterpret_cast<const float *>(bufc + rd[0][s]);
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp) {
        // this visit's operands (requested one visit ago): the second visit leaves the main pieces of the next slice,
        // issued after them, in flight; the third follows them in the queue -- by then they have been under way for
        // two thirds of the phase
        if (tp == 1) {
          if (copy) t3_wait_vm<N16>();
          else t3_wait_vm<0>();
        } else if (tp > 1) {
          t3_wait_vm<0>();
        }
        {   // A operands of the next (tap, phase) into the other set; every piece of the next slice in the first visit
          int ntp = tp + 1, nph = ph;
          if (ntp == TAPS) { ntp = 0; nph = (ph + 1) & (T3_NPH - 1); }
          if (tp & 1) t3_load_a<0>(a_base(ntp, nph), a_lane);
          else t3_load_a<1>(a_base(ntp, nph), a_lane);
        }
        if (tp == 0 && copy) copy_slice(buf_nxt, src, slo, shi);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          // B operand of the next unit requested before this unit's MFMAs (unit nslots, if any, is the quarter: joint 52)
          const int ni = i + 1 < NU ? i + 1 : 0, ntp2 = i + 1 < NU ? tp : tp + 1;
          const int nj = ni < nslots ? ni : JQ - j0;
          float b_nxt[4];
          if (ntp2 < TAPS) {
#pragma unroll
            for (int s = 0; s < 4; ++s) b_nxt[s] = *reinterpret_cast<const float *>(bufc + rd[ntp2][s] + 4 * nj);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (i < nslots) {
            if (tp & 1) t3_mfma16<1>(acc[i < nslots ? i : 0], b_cur);
            else t3_mfma16<0>(acc[i < nslots ? i : 0], b_cur);
          } else {
            if (tp & 1) t3_mfma4<1, MQ>(accq, b_cur);
            else t3_mfma4<0, MQ>(accq, b_cur);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (ntp2 < TAPS) {
#pragma unroll
            for (int s = 0; s < 4; ++s) b_cur[s] = b_nxt[s];
          }
        }
        if (tp == 0) T3_MARK(20 + ph);
        if (tp == 1) T3_MARK(24 + ph);
      }
      T3_MARK(4 + 4 * ph);
    }

    T3_MARK(17);
    // ---- epilogue: statistics of the tile, then the tile itself through LDS as whole rows ------------------------
    // sums about a pivot per (wave, row), merged with the counts at the end of the kernel (see stgcn_gcn3.hip)
    float *rs = rowstat + wave * 64 * T3_ST;
    if (!BWD && want_stats) {
      const bool first = tile == (int)blockIdx.x;
      // two rows (q, q + 1) per instruction: the accumulator tile is four consecutive registers, so the differences,
      // sums and squares of a pair are one packed instruction each (VALU time is matrix-pipe time: half the count)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          float *e0 = rs + T3_ST * (16 * m + 4 * g + 2 * qp), *e1 = e0 + T3_ST;
          f32x2 c;
          c.x = first ? p2r_row16_sum(acc[0][m][2 * qp]) * 0.0625f : e0[2];
          c.y = first ? p2r_row16_sum(acc[0][m][2 * qp + 1]) * 0.0625f : e1[2];
          f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
          for (int i = 0; i < nslots; ++i) {
            const f32x2 v = f32x2{acc[i][m][2 * qp], acc[i][m][2 * qp + 1]} - c;
            s1 += v;
            s2 = __builtin_elementwise_fma(v, v, s2);
          }
          if (QUARTER && m == MQ) {
            const f32x2 v = f32x2{accq[2 * qp], accq[2 * qp + 1]} - c;
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
    T3_MARK(18);
    {
      float *stg = lds + ((T3_NPH - 1) & 1) * BUF;          // main part of the last phase's buffer: free now
      constexpr int R4 = RS / 4;                            // float4 per row (212)
      constexpr int RIT = (R4 + 63) / 64;                   // 4
      // BWD: the saved activation of the two rows this wave stores in a round.  A row's buffer is reloaded for the next
      // round as soon as the row has been processed -- the loads then have the rest of the round, its closing barrier
      // and the next staging to arrive, instead of being issued right in front of the barrier they are needed behind --
      // at no cost in registers.
      float4 zv[BWD ? 2 : 1][BWD ? RIT : 1];
      auto load_z = [&](int m, int rr) {
        const float4 *z4 = reinterpret_cast<const float4 *>(zg + (size_t)(16 * m + 2 * wave + rr) * row_stride);
#pragma unroll
        for (int it = 0; it < (BWD ? RIT : 1); ++it) {
          const int c4 = it * 64 + lane;
          zv[BWD ? rr : 0][it] = c4 < R4 ? z4[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      if (BWD) { load_z(0, 0); load_z(0, 1); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int i = 0; i < nslots; ++i) {
          float *d0 = stg + 4 * g * RS + r * V + j0 + i;
#pragma unroll
          for (int q = 0; q < 4; ++q) d0[q * RS] = acc[i][m][q];
        }
        if (QUARTER && m == MQ) {
          float *d0 = stg + 4 * g * RS + r * V + JQ;
#pragma unroll
          for (int q = 0; q < 4; ++q) d0[q * RS] = accq[q];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float4 *orow = reinterpret_cast<float4 *>(og + (size_t)16 * m * row_stride);
        const float4 *srow = reinterpret_cast<const float4 *>(stg);
        if (BWD) {   // a wave takes two whole rows
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr, c = 16 * m + row;
            const float sc = aff[2 * c], sh = aff[2 * c + 1], mu = bstat[2 * c];
            float s1 = bsum[m][rr][0], s2 = bsum[m][rr][1];
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
                s2 = fmaf(g0, zz.x - mu, s2); s2 = fmaf(g1, zz.y - mu, s2);      // * invstd at the end of the kernel
                s2 = fmaf(g2, zz.z - mu, s2); s2 = fmaf(g3, zz.w - mu, s2);
              }
            }
            bsum[m][rr][0] = s1; bsum[m][rr][1] = s2;
            if (m + 1 < 4) load_z(m + 1, rr);
          }
        } else {
          constexpr int SV4 = T3_CP * RS / 4;                // float4 of the 16-row staging tile
#pragma unroll
          for (int it = 0; it < (SV4 + NW * 64 - 1) / (NW * 64); ++it) {
            const int e = it * NW * 64 + tid;
            if (e < SV4) {
              const int row = e / (RS / 4), c4 = e - row * (RS / 4);
              orow[(size_t)row * (row_stride / 4) + c4] = srow[e];
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    T3_MARK(19);
  }
  if (BWD) {
    float *rs = rowstat + wave * 64 * T3_ST;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        float s1 = bsum[m][rr][0], s2 = bsum[m][rr][1];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          s1 += __shfl_xor(s1, off, 64);
          s2 += __shfl_xor(s2, off, 64);
        }
        const int c = 16 * m + 2 * wave + rr;
        if (lane == 0) {
          rs[T3_ST * c] = s1;
          rs[T3_ST * c + 1] = s2 * bstat[2 * c + 1];
        }
      }
  }
}

template <bool XFORM, bool BWD, int TAPS, bool ADDCT = false>
__global__ __launch_bounds__(T3_NW * 64, 2) __attribute__((amdgpu_num_vgpr(224))) void tconv3_kernel(
    T3Params p, const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ Wp, const float *__restrict__ bias, float *__restrict__ out,
    float *__restrict__ stats_partial, const float *__restrict__ bwd_z, const float *__restrict__ bwd_fin,
    const float *__restrict__ add_ct) {
  constexpr int NW = T3_NW;
  extern __shared__ float lds[];
  float *rowstat = lds + 2 * T3_BUF;
  float *aff = rowstat + NW * 64 * T3_ST;
  float *bias_l = aff + 128;
  float *bstat = bias_l + 64;
  const int tid = threadIdx.x;
  for (int e = tid; e < NW * 64 * T3_ST; e += NW * 64) rowstat[e] = 0.f;
  if (tid < 64) {
    aff[2 * tid] = XFORM ? scale[tid] : (BWD ? bwd_fin[128 + tid] : 1.f);
    aff[2 * tid + 1] = XFORM ? shift[tid] : (BWD ? bwd_fin[192 + tid] : 0.f);
    bias_l[tid] = bias ? bias[tid] : 0.f;
    bstat[2 * tid] = BWD ? bwd_fin[tid] : 0.f;
    bstat[2 * tid + 1] = BWD ? bwd_fin[64 + tid] : 1.f;
  }
  __syncthreads();
  const bool want_stats = stats_partial != nullptr;
  switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
    case 0: t3_wave_main<XFORM, BWD, TAPS, 0, ADDCT>(p, lds, x, Wp, out, want_stats, bwd_z, add_ct); break;
    case 1: t3_wave_main<XFORM, BWD, TAPS, 1, ADDCT>(p, lds, x, Wp, out, want_stats, bwd_z, add_ct); break;
    case 2: t3_wave_main<XFORM, BWD, TAPS, 2, ADDCT>(p, lds, x, Wp, out, want_stats, bwd_z, add_ct); break;
    case 3: t3_wave_main<XFORM, BWD, TAPS, 3, ADDCT>(p, lds, x, Wp, out, want_stats, bwd_z, add_ct); break;
    case 4: t3_wave_main<XFORM, BWD, TAPS, 4, ADDCT>(p, lds, x, Wp, out, want_stats, bwd_z, add_ct); break;
    case 5: t3_wave_main<XFORM, BWD, TAPS, 5, ADDCT>(p, lds, x, Wp, out, want_stats, bwd_z, add_ct); break;
    case 6: t3_wave_main<XFORM, BWD, TAPS, 6, ADDCT>(p, lds, x, Wp, out, want_stats, bwd_z, add_ct); break;
    default: t3_wave_main<XFORM, BWD, TAPS, 7, ADDCT>(p, lds, x, Wp, out, want_stats, bwd_z, add_ct); break;
  }
  if (stats_partial) {
    __syncthreads();
    if constexpr (BWD) {          // [64][2] plain sums
      if (tid < 128) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += rowstat[(w * 64 + (tid >> 1)) * T3_ST + (tid & 1)];
        stats_partial[(size_t)blockIdx.x * 128 + tid] = t;
      }
    } else if (tid < 64) {        // [64][3] = (count, mean, M2) of the workgroup's tiles: the eight waves' entries merged
      const int ntiles = (p.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
      const float per_joint = (float)(ntiles * T3_F);
      float nw[NW], mw[NW], qw[NW];
      float msum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float *e = rowstat + (w * 64 + tid) * T3_ST;
        // joints per (wave, row): see t3_wave_main (waves 4..7 own row block w - 4 of joint 52 on top of their six)
        nw[w] = per_joint * (float)(w < 4 ? 7 : 6 + ((tid >> 4) == w - 4 ? 1 : 0));
        const float d = e[0] / nw[w];
        mw[w] = e[2] + d;
        qw[w] = fmaxf(e[1] - e[0] * d, 0.f);
        msum = fmaf(nw[w], mw[w] - mw[0], msum);          // about the first wave's mean: small terms
      }
      const float n = per_joint * (float)T3_V;
      const float mean = mw[0] + msum / n;
      float m2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) m2 += qw[w] + nw[w] * (mw[w] - mean) * (mw[w] - mean);
      float *o = stats_partial + (size_t)blockIdx.x * 192 + 3 * tid;
      o[0] = n; o[1] = mean; o[2] = m2;
    }
  }
}

template <bool XFORM, bool BWD, int TAPS, bool ADDCT = false>
int tconv3_launch(const T3Params &p, int blocks, size_t lds, const float *x, const float *scale, const float *shift,
                  const float *Wp, const float *bias, float *out, float *stats_partial, const float *bwd_z,
                  const float *bwd_fin, void *stream, const float *add_ct = nullptr) {
  auto kern = tconv3_kernel<XFORM, BWD, TAPS, ADDCT>;
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(kern, lds_ok);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(T3_NW * 64), lds, p2r_stream(stream), p, x, scale, shift, Wp, bias, out,
                     stats_partial, bwd_z, bwd_fin, add_ct);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

}  // namespace

// Arguments and semantics of p2r_stgcn_tconv2_forward; additionally T % 16 == 0 and x, out (and bwd_z) 16-byte aligned
// (P2R_EINVAL otherwise: the caller uses p2r_stgcn_tconv2_forward).
static int tconv3_forward_impl(int N, int T, int V, int taps, const float *x, const float *scale, const float *shift,
                               const float *Wp, const float *bias, float *out, float *stats_partial,
                               int *n_partials, const float *bwd_z, const float *bwd_fin, const float *add_ct,
                               void *stream) {
  if (N < 0 || T <= 0 || V != T3_V || (taps != 1 && taps != 3) || (scale == nullptr) != (shift == nullptr)) return P2R_EINVAL;
  if ((bwd_z == nullptr) != (bwd_fin == nullptr) || (bwd_z && (scale || !stats_partial))) return P2R_EINVAL;
  if (T % T3_F != 0 || T > (1 << 20) || ((uintptr_t)x % 16) != 0 || ((uintptr_t)out % 16) != 0 || ((uintptr_t)bwd_z % 16) != 0)
    return P2R_EINVAL;
  if (n_partials) *n_partials = 0;
  if (N == 0) return P2R_OK;
  T3Params p;
  p.T = T;
  p.tiles_per_seq = T / T3_F;
  const long long tiles = (long long)N * p.tiles_per_seq;
  if (tiles > 0x7fffffffLL) return P2R_EINVAL;
  p.total_tiles = (int)tiles;
  const int blocks = (int)(tiles < 256 ? tiles : 256);
  if (n_partials) *n_partials = blocks;
  if (!out) return P2R_OK;
  const size_t lds = (size_t)2 * T3_BUF * sizeof(float) + (size_t)T3_NW * 64 * T3_ST * sizeof(float) +
                     (size_t)(128 + 64 + 128) * sizeof(float);
#define P2R_T3(XF, BW) (taps == 3 ? tconv3_launch<XF, BW, 3>(p, blocks, lds, x, scale, shift, Wp, bias, out, stats_partial, bwd_z, bwd_fin, stream) \
                                   : tconv3_launch<XF, BW, 1>(p, blocks, lds, x, scale, shift, Wp, bias, out, stats_partial, bwd_z, bwd_fin, stream))
  if (add_ct) {       // single-tap forward through a BatchNorm + ReLU with a (sample, channel, frame) addend
    if (taps != 1 || !scale || bwd_z) return P2R_EINVAL;
    return tconv3_launch<true, false, 1, true>(p, blocks, lds, x, scale, shift, Wp, bias, out, stats_partial, bwd_z, bwd_fin,
                                               stream, add_ct);
  }
  if (bwd_z) return P2R_T3(false, true);
  if (scale) return P2R_T3(true, false);
  return P2R_T3(false, false);
#undef P2R_T3
}

extern "C" int p2r_stgcn_tconv3_forward(int N, int T, int V, int taps, const float *x, const float *scale, const float *shift,
                                        const float *Wp, const float *bias, float *out, float *stats_partial,
                                        int *n_partials, const float *bwd_z, const float *bwd_fin, void *stream) {
  return tconv3_forward_impl(N, T, V, taps, x, scale, shift, Wp, bias, out, stats_partial, n_partials, bwd_z, bwd_fin, nullptr,
                             stream);
}

// The single-tap forward (pointwise 64 -> 64 convolution behind a BatchNorm + ReLU) with an addend add_ct (N,64,T) that
// is broadcast over the joints: out[n,c,t,w] = bias[c] + add_ct[n,c,t] + sum_ci W[c][ci] relu(x[n,ci,t,w] scale + shift)
// -- the last layer of the joint embedding plus the position embedding (stgcn.py:126-130) in one pass.
extern "C" int p2r_stgcn_tconv3_forward_add(int N, int T, int V, const float *x, const float *scale, const float *shift,
                                            const float *Wp, const float *bias, const float *add_ct, float *out,
                                            void *stream) {
  if (!add_ct) return P2R_EINVAL;
  return tconv3_forward_impl(N, T, V, 1, x, scale, shift, Wp, bias, out, nullptr, nullptr, nullptr, nullptr, add_ct, stream);
}
