// stgcn_gcn3.hip -- fused spatial graph convolution of the ST-GCN backbone, third generation: STATICALLY SCHEDULED.
//
// Same operator and same data movement as stgcn_gcn2.hip (ConvTemporalGraphical.forward, reference
// models/p2rnet/modules/stgcn_layers.py:57-67, and its data gradient):
//     Z[:, (t, w)] = bias_cv[:, w] + sum_k W_k . ( sum_{v in list_k(w)} a_k(v, w) X[:, (t, v)] )
// n-tile = 16 frames of one joint, four channel phases of 16 through two LDS-DMA'd slice buffers, persistent
// workgroups, accumulators of up to 7 joints per wave in registers, whole-row stores through LDS, BatchNorm
// statistics / BatchNorm-backward sums in the epilogue.
//
// What changed: the per-wave work list is no longer DATA walked at run time (records in SGPRs, a slot dispatch chain,
// scalar loads one record ahead) but CODE.  Measured on MI355X (round 3, tools/dev_gcn2_exp.py): doubling the MFMAs
// of gcn2 adds exactly their roofline time (0.77 ms for 123.7 GFLOP), i.e. the matrix pipe runs at peak whenever it
// is fed, and 0.58 ms of the 1.35 ms launch is everything else -- of which the operand gathers are 0.18 ms, stores
// 0.04, DMA 0.03 and 0.32 ms the record handling itself.  fp32 MFMAs share the SIMD's vector issue with whatever the
// co-resident wave does (tools/ubench/issue_costs.hip: a VALU op costs 4-8 cycles of matrix-pipe time, an LDS read
// 2), so the only way to win that time back is not to execute those instructions.  The adjacency PATTERN is a
// property of the skeleton; only the coefficient values change from step to step.  tools/gen_gcn_sched.py therefore
// resolves the schedule at build time (gcn3_sched.inc): per wave a straight-line sequence of steps in which the
// accumulator slot is a register name, the source joints are ds_read immediates, the coefficients are scalar loads
// at immediate offsets of the coefficient table, the A-operand sets ping-pong by name and the DMA pieces of the next
// slice have fixed places.  No stream, no fill pre-launch, no branches inside a phase.
//
// The host only launches this kernel for tables whose pattern_signature equals the generated one
// (p2r_stgcn_gcn3_signature); every other adjacency runs on gcn2 / the first-generation kernel.
#include "p2r_common.h"

#include "gcn3_sched.inc"

// Cycle trace (profiling hook, off in the product build; see stgcn_tconv3.hip): -DP2R_CYCLE_TRACE, tools/dev_g3_trace.py
#ifdef P2R_CYCLE_TRACE
__device__ unsigned long long g3_trace[8 * 32];
extern "C" int p2r_debug_g3_trace(unsigned long long *dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g3_trace), sizeof(g3_trace));
}
#define G3_TRACE_TILE(tile) const bool trace_on = blockIdx.x == 7 && (tile) == 7 + 3 * (int)gridDim.x
#define G3_MARK(i) do { if (trace_on && lane == 0) g3_trace[wave * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define G3_TRACE_TILE(tile)
#define G3_MARK(i)
#endif
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int G3_F = 16;             // frames per tile = columns of an MFMA n-tile
constexpr int G3_CP = 16;            // channels per phase
constexpr int G3_NPH = 4;            // phases (64 channels)
constexpr int G3_NW = 8;
constexpr int G3_SLOTS = 7;
constexpr int G3_RS = G3_F * G3_V;   // LDS row stride (floats): 848 == 16 (mod 32)
constexpr int G3_BUF = G3_CP * G3_RS;
constexpr int G3_NV4 = G3_BUF / 4;                     // float4 elements per slice: 3392 = 53 pieces of 64
constexpr int G3_PIECES = (G3_NV4 + 63) / 64;          // 53
constexpr int G3_PW = (G3_PIECES + G3_NW - 1) / G3_NW; // 7 per wave
static_assert(G3_V == 53 && G3_K == 11, "schedule generated for another skeleton");

struct G3Params {
  int T;
  int tiles_per_seq, total_tiles;
};

constexpr int g3_slot_joints[2][G3_NW][G3_SLOTS] = {G3_SLOT_JOINTS_0, G3_SLOT_JOINTS_1};
constexpr int g3_plane0[2][G3_NW] = {G3_PLANE0_0, G3_PLANE0_1};            // first plane of each wave's schedule
constexpr int g3_first_slot(int form, int w) {                             // first used accumulator slot of a wave
  for (int i = 0; i < G3_SLOTS; ++i)
    if (g3_slot_joints[form][w][i] >= 0) return i;
  return 0;
}
constexpr int g3_wave_joints(int form, int w) {                            // joints a wave owns
  int n = 0;
  for (int i = 0; i < G3_SLOTS; ++i) n += g3_slot_joints[form][w][i] >= 0;
  return n;
}
constexpr int G3_ST = 3;   // floats per (wave, row) statistics entry: (sum, sum of squares) about the pivot, pivot

__device__ __forceinline__ unsigned g3_lds_addr(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
// one 1 KB LDS-DMA piece: lane's 16 bytes at (uniform base + per-lane byte offset) -> LDS dst + 16 * lane.  Inline
// assembly for the reason given in stgcn_gcn2.hip (no vmcnt(0) in front of later LDS reads); M0 = LDS destination.
__device__ __forceinline__ void g3_dma16(const float *base, int voff, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(g3_lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// xv[j][s] = X[row 4 s + g][frame r, joint of entry j]: ds_read with immediate offsets off the lane's base
template <int NE, int O0, int O1, int O2, int O3, int O4, int O5>
__device__ __forceinline__ void g3_gather(const char *xl, float (&xv)[6][4]) {
  constexpr int off[6] = {O0, O1, O2, O3, O4, O5};
#pragma unroll
  for (int j = 0; j < NE; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) xv[j][s] = *reinterpret_cast<const float *>(xl + off[j] + s * 4 * G3_RS * 4);
}
// The coefficients of the step's entries: broadcast LDS reads (uniform address, immediate offset) issued with the
// gathers.  They are loop-invariant over the phases, so as scalar loads from global memory the compiler hoists every
// one of them out of the phase loop -- ~130 live SGPRs per wave, spilled to VGPR lanes and from there to scratch
// (measured: 744 SGPR + 423 VGPR spills); LDS reads stay where they are written.
template <int NE, int C0, int C1, int C2, int C3, int C4, int C5>
__device__ __forceinline__ void g3_coefs(const char *cl, float (&c)[6]) {
  constexpr int ci[6] = {C0, C1, C2, C3, C4, C5};
#pragma unroll
  for (int j = 0; j < NE; ++j) c[j] = *reinterpret_cast<const float *>(cl + 4 * ci[j]);
}
template <int NE>
__device__ __forceinline__ void g3_combine(const float (&c)[6], const float (&xv)[6][4], float (&b)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float v = c[0] * xv[0][s];
#pragma unroll
    for (int j = 1; j < NE; ++j) v = fmaf(c[j], xv[j][s], v);
    b[s] = v;
  }
}

// The 16 MFMAs of a step as ONE assembly block that accumulates in place.  Issued through the builtin, the
// compiler renames the accumulators along every chain (the home register of a tile rotates from step to step), and
// where the eight per-wave bodies meet again -- once per phase -- it reconciles the 112 accumulator registers through
// scratch (measured: ~300 spilled VGPRs, 28 16-byte reloads per wave and phase, all in vmcnt order with the DMA
// pieces).  "+v" pins each tile to its registers for the whole kernel.  The operands were written by VALU / LDS
// instructions the compiler has already waited for (lgkmcnt) -- the leading s_nop covers the VALU -> MFMA read
// wait states, which it does not insert for assembly; MFMA -> MFMA accumulation on the same registers needs none.
__device__ __forceinline__ void g3_mfma16(f32x4 (&acc)[4], const float (&a)[4][4], const float (&b)[4]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %4, %20, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %20, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, %12, %20, %2\n\tv_mfma_f32_16x16x4_f32 %3, %16, %20, %3\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %5, %21, %0\n\tv_mfma_f32_16x16x4_f32 %1, %9, %21, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, %13, %21, %2\n\tv_mfma_f32_16x16x4_f32 %3, %17, %21, %3\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %6, %22, %0\n\tv_mfma_f32_16x16x4_f32 %1, %10, %22, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, %14, %22, %2\n\tv_mfma_f32_16x16x4_f32 %3, %18, %22, %3\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %7, %23, %0\n\tv_mfma_f32_16x16x4_f32 %1, %11, %23, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %2, %15, %23, %2\n\tv_mfma_f32_16x16x4_f32 %3, %19, %23, %3"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
      : "v"(a[0][0]), "v"(a[0][1]), "v"(a[0][2]), "v"(a[0][3]), "v"(a[1][0]), "v"(a[1][1]), "v"(a[1][2]), "v"(a[1][3]),
        "v"(a[2][0]), "v"(a[2][1]), "v"(a[2][2]), "v"(a[2][3]), "v"(a[3][0]), "v"(a[3][1]), "v"(a[3][2]), "v"(a[3][3]),
        "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
#define G3_MFMA(set, slot) g3_mfma16(acc[slot], aS[set], b_cur);

#define G3_FIRST(ne, o0, c0, o1, c1, o2, c2, o3, c3, o4, c4, o5, c5)   \
  {                                                                    \
    float xv_[6][4], cf_[6];                                           \
    g3_gather<ne, o0, o1, o2, o3, o4, o5>(xl, xv_);                    \
    g3_coefs<ne, c0, c1, c2, c3, c4, c5>(cl, cf_);                     \
    g3_combine<ne>(cf_, xv_, b_cur);                                   \
  }
#define G3_VISIT(set, plane, next, wrap, piece)                                          \
  {                                                                                      \
    load_a(aS[(set) ^ 1], next, (wrap) ? ((ph + 1) & (G3_NPH - 1)) : ph);                 \
    if ((piece) >= 0 && copy) dma_piece(piece);                                          \
  }
#define G3_STEP(set, slot, ne, o0, c0, o1, c1, o2, c2, o3, c3, o4, c4, o5, c5)   \
  {                                                                              \
    float xv_[6][4], cf_[6];                                                     \
    g3_gather<ne, o0, o1, o2, o3, o4, o5>(xl, xv_);                              \
    g3_coefs<ne, c0, c1, c2, c3, c4, c5>(cl, cf_);                               \
    __builtin_amdgcn_sched_barrier(0);                                           \
    G3_MFMA(set, slot)                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    g3_combine<ne>(cf_, xv_, b_cur);                                             \
  }
#define G3_LAST(set, slot)             \
  {                                    \
    __builtin_amdgcn_sched_barrier(0); \
    G3_MFMA(set, slot)                 \
  }
#define G3_END(parity, pieces, plane0)                                                    \
  {                                                                                       \
    if (copy) { _Pragma("unroll") for (int i_ = pieces; i_ < G3_PW; ++i_) dma_piece(i_); } \
    if (parity) {                                                                         \
      _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) \
          aS[0][m_][s_] = aS[1][m_][s_];                                                  \
    }                                                                                     \
  }

// One wave's whole life inside the persistent kernel, instantiated per (schedule, wave): the eight waves of a
// workgroup run eight different straight-line programs.  They are kept apart from the first instruction to the last
// (the kernel is a switch over the wave index around this function) -- merged once per phase, as a switch inside a
// common loop, the compiler reconciles the eight register assignments of the 112 accumulator registers at every
// merge with copies and scratch traffic.  Barriers are counted by the hardware, not matched by address, so the waves
// of a workgroup may meet at different program counters.
// BWD (data-gradient launches): the statistics epilogue emits the reduction pass of the BatchNorm + residual + ReLU
// backward of the block in front, exactly as gcn2_kernel<.., true> does (see stgcn_gcn2.hip).
// MADD (BWD launches): the addend is masked on the way in -- addend * (addend_mask != 0) -- i.e. the residual-branch
// gradient g = dout * relu_mask of the block is formed here from the incoming gradient and the mask bytes instead of
// being written (444 MB at bs=32, T=1024) by the BatchNorm-backward pass and read back.
template <int FORM, bool BWD, int WAVE, bool MADD = false>
__device__ __forceinline__ void g3_wave_main(
    const G3Params &p, float *lds, const float *__restrict__ x, const float *__restrict__ Wp,
    const float *__restrict__ addend, float *__restrict__ z, bool want_stats, const float *__restrict__ bwd_u,
    const unsigned char *__restrict__ bwd_mask, const unsigned char *__restrict__ addend_mask = nullptr) {
  constexpr int V = G3_V, RS = G3_RS, BUF = G3_BUF, NW = G3_NW, SLOTS = G3_SLOTS;
  constexpr int wave = WAVE;
  float *rowstat = lds + 2 * BUF;                                         // [NW][64][G3_ST]
  float *bias_l = rowstat + NW * 64 * G3_ST;                              // [64][V] bias table (zeros without bias)
  float *bstat = bias_l + 64 * V;                                         // [64][2] (mean, invstd) of the BWD epilogue
  float *coef_l = bstat + 128;                                            // [ltot][V] coefficient table

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int g = lane >> 4, r = lane & 15;

  constexpr const int (&sj)[SLOTS] = g3_slot_joints[FORM][WAVE];   // joint of each accumulator slot, -1: unused

  const size_t row_stride = (size_t)p.T * V;
  const char *xl0 = reinterpret_cast<const char *>(lds + g * RS + r * V);   // lane's gather base (row g, frame r)
  // base of the coefficient table in a VGPR whose value the compiler cannot see: folded to a constant, every
  // coefficient address becomes its own scalar constant, hoisted out of the loops and spilled (measured: ~1000 SGPR
  // spills); off an opaque base the entries are 16-bit immediates of one ds_read address register
  unsigned cl_off = (unsigned)((coef_l - lds) * sizeof(float));
  asm volatile("" : "+v"(cl_off));
  const char *cl = reinterpret_cast<const char *>(lds) + cl_off;

  // this wave's pieces of a slice: piece i covers float4 elements (i * NW + wave) * 64 + lane of the 16 x 848 slice;
  // the lane's source byte offset inside the slice (row * row_stride + column) is the same for every slice
  int doff[G3_PW];
#pragma unroll
  for (int i = 0; i < G3_PW; ++i) {
    const int pc = i * NW + wave;
    const int e = pc * 64 + lane;
    const int row = e / (RS / 4), c4 = e - row * (RS / 4);
    doff[i] = (pc < G3_PIECES && e < G3_NV4) ? (int)(((size_t)row * row_stride + 4 * c4) * sizeof(float)) : -1;
  }

  f32x4 acc[SLOTS][4];
  float aS[2][4][4];                                  // two A-operand sets: W'[k][ph][m][lane][s]
  float b_cur[4];
  auto load_a = [&](float (&a)[4][4], int k, int ph) {
    const float4 *wp = reinterpret_cast<const float4 *>(Wp) + ((size_t)(k * G3_NPH + ph) * 4) * 64 + lane;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float4 u = wp[m * 64];
      a[m][0] = u.x; a[m][1] = u.y; a[m][2] = u.z; a[m][3] = u.w;
    }
  };

  int tile = blockIdx.x;
  // prologue: phase 0 of the first tile, all pieces at once; A operands of the wave's first plane
  if (tile < p.total_tiles) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * G3_F;
    const float *xr = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
#pragma unroll
    for (int i = 0; i < G3_PW; ++i)
      if (doff[i] >= 0) g3_dma16(xr, doff[i], lds + (i * NW + wave) * 256);
  }
  load_a(aS[0], g3_plane0[FORM][WAVE], 0);

  for (; tile < p.total_tiles; tile += gridDim.x) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * G3_F;
    const float *xg = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    float *zg = z + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    const float *ag = addend ? addend + (size_t)seq * 64 * row_stride + (size_t)t0 * V : nullptr;
    const float *ug = BWD ? bwd_u + (size_t)seq * 64 * row_stride + (size_t)t0 * V : nullptr;
    const unsigned char *mg = BWD ? bwd_mask + (size_t)seq * 64 * row_stride + (size_t)t0 * V : nullptr;
    const unsigned char *amg = MADD ? addend_mask + (size_t)seq * 64 * row_stride + (size_t)t0 * V : nullptr;
    const int ntile = tile + gridDim.x;
    const bool has_next = ntile < p.total_tiles;
    const int nseq = has_next ? ntile / p.tiles_per_seq : 0, nt0 = has_next ? (ntile % p.tiles_per_seq) * G3_F : 0;
    const float *nxg = x + (size_t)nseq * 64 * row_stride + (size_t)nt0 * V;
    G3_TRACE_TILE(tile);
    G3_MARK(0);

    // accumulators start from the bias table
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const float *bl = bias_l + 4 * g * V + (sj[i] >= 0 ? sj[i] : 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][m][q] = bl[(16 * m + q) * V];
    }

#pragma unroll 1
    for (int ph = 0; ph < G3_NPH; ++ph) {
      // slice `ph` has landed (every wave waited for its own pieces) and nobody reads the other buffer any more
      G3_MARK(1 + 3 * ph);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      G3_MARK(2 + 3 * ph);
      __syncthreads();
      G3_MARK(3 + 3 * ph);
      float *buf_nxt = lds + ((ph + 1) & 1) * BUF;
      const char *xl = xl0 + (ph & 1) * BUF * sizeof(float);
      const bool copy = ph + 1 < G3_NPH || has_next;
      const float *src = (ph + 1 < G3_NPH) ? xg + (size_t)(ph + 1) * G3_CP * row_stride : nxg;
      auto dma_piece = [&](int i) {
        if (doff[i] >= 0) g3_dma16(src, doff[i], buf_nxt + (i * NW + wave) * 256);
      };

      if constexpr (FORM == 0) {
        if constexpr (WAVE == 0) { G3_BODY_0_0 } else if constexpr (WAVE == 1) { G3_BODY_0_1 }
        else if constexpr (WAVE == 2) { G3_BODY_0_2 } else if constexpr (WAVE == 3) { G3_BODY_0_3 }
        else if constexpr (WAVE == 4) { G3_BODY_0_4 } else if constexpr (WAVE == 5) { G3_BODY_0_5 }
        else if constexpr (WAVE == 6) { G3_BODY_0_6 } else { G3_BODY_0_7 }
      } else {
        if constexpr (WAVE == 0) { G3_BODY_1_0 } else if constexpr (WAVE == 1) { G3_BODY_1_1 }
        else if constexpr (WAVE == 2) { G3_BODY_1_2 } else if constexpr (WAVE == 3) { G3_BODY_1_3 }
        else if constexpr (WAVE == 4) { G3_BODY_1_4 } else if constexpr (WAVE == 5) { G3_BODY_1_5 }
        else if constexpr (WAVE == 6) { G3_BODY_1_6 } else { G3_BODY_1_7 }
      }
    }

    G3_MARK(13);
    // ---- epilogue: D[row = 16 m + 4 g + q][frame r] of joint sj[i]; statistics of the stored values.
    // The sums are taken about a pivot per (wave, row) -- the mean of the first 16 values the wave produces for the
    // row -- and merged at the end of the kernel with the counts (see bn_act.hip: sum v^2 - (sum v)^2 / n in fp32
    // loses the variance once |mean| >> std).
    float *rs = rowstat + wave * 64 * G3_ST;
    if (!BWD && want_stats) {
      constexpr int I0 = g3_first_slot(FORM, WAVE);
      const bool first = tile == (int)blockIdx.x;
      // two rows (q, q + 1) per instruction: the accumulator tile is four consecutive registers, so the differences,
      // sums and squares of a pair are one packed instruction each (VALU time is matrix-pipe time: half the count)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          float *e0 = rs + G3_ST * (16 * m + 4 * g + 2 * qp), *e1 = e0 + G3_ST;
          f32x2 c;
          c.x = first ? p2r_row16_sum(acc[I0][m][2 * qp]) * 0.0625f : e0[2];
          c.y = first ? p2r_row16_sum(acc[I0][m][2 * qp + 1]) * 0.0625f : e1[2];
          f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
          for (int i = 0; i < SLOTS; ++i)
            if (sj[i] >= 0) {
              const f32x2 v = f32x2{acc[i][m][2 * qp], acc[i][m][2 * qp + 1]} - c;
              s1 += v;
              s2 = __builtin_elementwise_fma(v, v, s2);
            }
          const float s1x = p2r_row16_sum(s1.x), s1y = p2r_row16_sum(s1.y);
          const float s2x = p2r_row16_sum(s2.x), s2y = p2r_row16_sum(s2.y);
          if (r == 0) {             // entries owned by (wave, row): plain read-modify-write, deterministic
            e0[0] += s1x; e0[1] += s2x;
            e1[0] += s1y; e1[1] += s2y;
            if (first) { e0[2] = c.x; e1[2] = c.y; }
          }
        }
    }
    G3_MARK(14);
    {
      // The tile leaves through LDS (see stgcn_gcn2.hip): the slice buffer of the last phase is free once every wave
      // has finished it; 16 rows at a time are laid out there as the tensor has them and written as whole
      // 16-byte-per-lane rows.  LDS-only waits in front of the barriers: the global stores stay in flight.
      float *stg = lds + ((G3_NPH - 1) & 1) * BUF;
      constexpr int R4 = RS / 4;                            // float4 per row (212)
      constexpr int RIT = (R4 + 63) / 64;                   // 4
      // BWD: saved activation, mask bytes and addend of the two rows this wave stores in a round.  A row's buffers are
      // reloaded for the next round as soon as the row has been processed (see stgcn_tconv3.hip): the loads then have
      // the rest of the round, its closing barrier and the next staging to arrive.
      // ROWS: the row-per-wave store form with its operands fetched a round ahead -- the BatchNorm-backward launches, and
      // the plain launch with a masked addend (as a straight load-add-store loop its 0.55 GB of addend + mask reads
      // cost as much as the whole sums epilogue)
      constexpr bool ROWS = BWD || MADD;
      float4 uv[BWD ? 2 : 1][BWD ? RIT : 1];
      float4 ad[ROWS ? 2 : 1][ROWS ? RIT : 1];
      unsigned mk[BWD ? 2 : 1][BWD ? RIT : 1];
      unsigned mk2[MADD ? 2 : 1][MADD ? RIT : 1];
      auto load_bwd = [&](int m, int rr) {
        const size_t r0 = (size_t)(16 * m + 2 * wave + rr) * row_stride;
        const float4 *u4 = reinterpret_cast<const float4 *>(BWD ? ug + r0 : nullptr);
        const unsigned *m4 = reinterpret_cast<const unsigned *>(BWD ? mg + r0 : nullptr);
        const float4 *a4 = reinterpret_cast<const float4 *>(ag ? ag + r0 : nullptr);
        const unsigned *am4 = reinterpret_cast<const unsigned *>(MADD ? amg + r0 : nullptr);
#pragma unroll
        for (int it = 0; it < (ROWS ? RIT : 1); ++it) {
          const int c4 = it * 64 + lane;
          if (BWD) {
            uv[BWD ? rr : 0][it] = c4 < R4 ? u4[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
            mk[BWD ? rr : 0][it] = c4 < R4 ? m4[c4] : 0u;
          }
          if (a4 && c4 < R4) ad[ROWS ? rr : 0][it] = a4[c4];
          if (MADD) mk2[MADD ? rr : 0][it] = c4 < R4 ? am4[c4] : 0u;
        }
      };
      if (ROWS) { load_bwd(0, 0); load_bwd(0, 1); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i)
          if (sj[i] >= 0) {
            float *d0 = stg + 4 * g * RS + r * V + sj[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) d0[q * RS] = acc[i][m][q];
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float4 *zrow = reinterpret_cast<float4 *>(zg + (size_t)16 * m * row_stride);
        const float4 *arow = reinterpret_cast<const float4 *>(ag ? ag + (size_t)16 * m * row_stride : nullptr);
        const float4 *srow = reinterpret_cast<const float4 *>(stg);
        if (ROWS) {   // a wave takes two whole rows: the per-channel sums stay in registers until the row is done
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr, c = 16 * m + row;
            const float mu = BWD ? bstat[2 * c] : 0.f, is = BWD ? bstat[2 * c + 1] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int it = 0; it < RIT; ++it) {
              const int c4 = it * 64 + lane;
              if (c4 < R4) {
                float4 v = srow[row * R4 + c4];
                if (arow) {
                  if (MADD) {
                    const unsigned m2 = mk2[MADD ? rr : 0][it];
                    v.x += (m2 & 0xffu) ? ad[rr][it].x : 0.f; v.y += (m2 & 0xff00u) ? ad[rr][it].y : 0.f;
                    v.z += (m2 & 0xff0000u) ? ad[rr][it].z : 0.f; v.w += (m2 & 0xff000000u) ? ad[rr][it].w : 0.f;
                  } else {
                    v.x += ad[rr][it].x; v.y += ad[rr][it].y; v.z += ad[rr][it].z; v.w += ad[rr][it].w;
                  }
                }
                zrow[(size_t)row * (row_stride / 4) + c4] = v;
                if (BWD) {
                  const float4 uu = uv[BWD ? rr : 0][it];
                  const unsigned mm = mk[BWD ? rr : 0][it];
                  const float g0 = (mm & 0xffu) ? v.x : 0.f, g1 = (mm & 0xff00u) ? v.y : 0.f;
                  const float g2 = (mm & 0xff0000u) ? v.z : 0.f, g3 = (mm & 0xff000000u) ? v.w : 0.f;
                  s1 += (g0 + g1) + (g2 + g3);
                  s2 = fmaf(g0, (uu.x - mu) * is, s2); s2 = fmaf(g1, (uu.y - mu) * is, s2);
                  s2 = fmaf(g2, (uu.z - mu) * is, s2); s2 = fmaf(g3, (uu.w - mu) * is, s2);
                }
              }
            }
            if (BWD) {
#pragma unroll
              for (int off = 32; off >= 1; off >>= 1) {
                s1 += __shfl_xor(s1, off, 64);
                s2 += __shfl_xor(s2, off, 64);
              }
              if (lane == 0) {
                rs[G3_ST * c] += s1;
                rs[G3_ST * c + 1] += s2;
              }
            }
            if (m + 1 < 4) load_bwd(m + 1, rr);
          }
        } else {
#pragma unroll
          for (int it = 0; it < (G3_NV4 + NW * 64 - 1) / (NW * 64); ++it) {
            const int e = it * NW * 64 + tid;
            if (e < G3_NV4) {
              const int row = e / (RS / 4), c4 = e - row * (RS / 4);
              float4 v = srow[e];
              if (arow) {           // e.g. the gradient of the block's residual branch, added on the way out
                const float4 ad4 = arow[(size_t)row * (row_stride / 4) + c4];
                v.x += ad4.x; v.y += ad4.y; v.z += ad4.z; v.w += ad4.w;
              }
              zrow[(size_t)row * (row_stride / 4) + c4] = v;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    G3_MARK(15);
  }

}

template <int FORM, bool BWD, bool MADD = false>
__global__ __launch_bounds__(G3_NW * 64, 2) void gcn3_kernel(
    G3Params p, int ltot, const float *__restrict__ x, const float *__restrict__ Wp, const float *__restrict__ coef,
    const float *__restrict__ bias_cv, const float *__restrict__ addend, float *__restrict__ z,
    float *__restrict__ stats_partial, const float *__restrict__ bwd_u, const unsigned char *__restrict__ bwd_mask,
    const float *__restrict__ bwd_fin, const unsigned char *__restrict__ addend_mask) {
  constexpr int V = G3_V, BUF = G3_BUF, NW = G3_NW;
  extern __shared__ float lds[];
  float *rowstat = lds + 2 * BUF;
  float *bias_l = rowstat + NW * 64 * G3_ST;
  float *bstat = bias_l + 64 * V;
  float *coef_l = bstat + 128;
  const int tid = threadIdx.x;
  for (int e = tid; e < NW * 64 * G3_ST; e += NW * 64) rowstat[e] = 0.f;
  for (int e = tid; e < 64 * V; e += NW * 64) bias_l[e] = bias_cv ? bias_cv[e] : 0.f;
  for (int e = tid; e < ltot * V; e += NW * 64) coef_l[e] = coef[e];
  if (tid < 64) {
    bstat[2 * tid] = BWD ? bwd_fin[tid] : 0.f;
    bstat[2 * tid + 1] = BWD ? bwd_fin[64 + tid] : 1.f;
  }
  __syncthreads();

  const bool want_stats = stats_partial != nullptr;
  switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
    case 0: g3_wave_main<FORM, BWD, 0, MADD>(p, lds, x, Wp, addend, z, want_stats, bwd_u, bwd_mask, addend_mask); break;
    case 1: g3_wave_main<FORM, BWD, 1, MADD>(p, lds, x, Wp, addend, z, want_stats, bwd_u, bwd_mask, addend_mask); break;
    case 2: g3_wave_main<FORM, BWD, 2, MADD>(p, lds, x, Wp, addend, z, want_stats, bwd_u, bwd_mask, addend_mask); break;
    case 3: g3_wave_main<FORM, BWD, 3, MADD>(p, lds, x, Wp, addend, z, want_stats, bwd_u, bwd_mask, addend_mask); break;
    case 4: g3_wave_main<FORM, BWD, 4, MADD>(p, lds, x, Wp, addend, z, want_stats, bwd_u, bwd_mask, addend_mask); break;
    case 5: g3_wave_main<FORM, BWD, 5, MADD>(p, lds, x, Wp, addend, z, want_stats, bwd_u, bwd_mask, addend_mask); break;
    case 6: g3_wave_main<FORM, BWD, 6, MADD>(p, lds, x, Wp, addend, z, want_stats, bwd_u, bwd_mask, addend_mask); break;
    default: g3_wave_main<FORM, BWD, 7, MADD>(p, lds, x, Wp, addend, z, want_stats, bwd_u, bwd_mask, addend_mask); break;
  }

  if (stats_partial) {
    __syncthreads();
    if constexpr (BWD) {          // [64][2] plain sums
      if (tid < 128) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += rowstat[(w * 64 + (tid >> 1)) * G3_ST + (tid & 1)];
        stats_partial[(size_t)blockIdx.x * 128 + tid] = t;
      }
    } else if (tid < 64) {        // [64][3] = (count, mean, M2) of the workgroup's tiles: the eight waves' entries merged
      const int ntiles = (p.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
      const float per_joint = (float)(ntiles * G3_F);
      float nw[NW], mw[NW], qw[NW];
      float msum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float *e = rowstat + (w * 64 + tid) * G3_ST;
        nw[w] = per_joint * (float)g3_wave_joints(FORM, w);
        const float d = e[0] / nw[w];
        mw[w] = e[2] + d;
        qw[w] = fmaxf(e[1] - e[0] * d, 0.f);
        msum = fmaf(nw[w], mw[w] - mw[0], msum);          // about the first wave's mean: small terms
      }
      const float n = per_joint * (float)V;
      const float mean = mw[0] + msum / n;
      float m2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) m2 += qw[w] + nw[w] * (mw[w] - mean) * (mw[w] - mean);
      float *o = stats_partial + (size_t)blockIdx.x * 192 + 3 * tid;
      o[0] = n; o[1] = mean; o[2] = m2;
    }
  }
}

template <int FORM, bool BWD, bool MADD = false>
int gcn3_launch(const G3Params &p, int ltot, int blocks, size_t lds, const float *x, const float *Wp, const float *coef,
                const float *bias_cv, const float *addend, float *z, float *stats_partial, const float *bwd_u,
                const unsigned char *bwd_mask, const float *bwd_fin, void *stream_h,
                const unsigned char *addend_mask = nullptr) {
  auto kern = gcn3_kernel<FORM, BWD, MADD>;
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(kern, lds_ok);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(G3_NW * 64), lds, p2r_stream(stream_h), p, ltot, x, Wp, coef, bias_cv, addend,
                     z, stats_partial, bwd_u, bwd_mask, bwd_fin, addend_mask);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

}  // namespace

// Signature of the adjacency pattern the static schedule of `form` (0: column lists / forward, 1: row lists / data
// gradient) was generated for: pose2room_amd.p2rnet.gcn_tables.pattern_signature of the run-time tables must equal it.
extern "C" unsigned long long p2r_stgcn_gcn3_signature(int form) {
  return form == 0 ? G3_SIGNATURE_0 : (form == 1 ? G3_SIGNATURE_1 : 0ull);
}

// x (N,64,T,53) -> z (N,64,T,53), statically scheduled (see the head of this file).  Arguments as
// p2r_stgcn_gcn2_forward without the work stream: `form` selects the schedule, coef f32 [ltot][53] is the coefficient
// table the schedule indexes (values of A * importance at the list entries, zeros at padded slots).
// Requirements beyond gcn2's (P2R_EINVAL otherwise; the caller then uses gcn2): T % 16 == 0 and x, z, addend 16-byte
// aligned.  n_partials: number of workgroups = rows of stats_partial (call with z == NULL to query).
static int gcn3_forward_impl(int N, int T, int V, int K, int ltot, int form, const float *x, const float *Wp,
                             const float *coef, const float *bias_cv, const float *addend, float *z,
                             float *stats_partial, int *n_partials, const float *bwd_u,
                             const unsigned char *bwd_mask, const float *bwd_fin, const unsigned char *addend_mask,
                             void *stream_h) {
  if (N < 0 || T <= 0 || V != G3_V || K != G3_K || ltot <= 0 || (form != 0 && form != 1)) return P2R_EINVAL;
  if (T % G3_F != 0 || T > (1 << 20)) return P2R_EINVAL;
  const bool bwd = bwd_u != nullptr;
  if (bwd != (bwd_mask != nullptr) || bwd != (bwd_fin != nullptr) || (bwd && (!stats_partial || form != 1))) return P2R_EINVAL;
  if (bwd && (((uintptr_t)bwd_u % 16) != 0 || ((uintptr_t)bwd_mask % 4) != 0)) return P2R_EINVAL;
  if (((uintptr_t)x % 16) != 0 || ((uintptr_t)z % 16) != 0 || ((uintptr_t)addend % 16) != 0) return P2R_EINVAL;
  if (n_partials) *n_partials = 0;
  if (N == 0) return P2R_OK;
  G3Params p;
  p.T = T;
  p.tiles_per_seq = T / G3_F;
  const long long tiles = (long long)N * p.tiles_per_seq;
  if (tiles > 0x7fffffffLL) return P2R_EINVAL;
  p.total_tiles = (int)tiles;
  const int blocks = (int)(tiles < 256 ? tiles : 256);
  if (n_partials) *n_partials = blocks;
  if (!z) return P2R_OK;
  const size_t lds = (size_t)2 * G3_BUF * sizeof(float) + (size_t)G3_NW * 64 * G3_ST * sizeof(float) +
                     (size_t)64 * V * sizeof(float) + 128 * sizeof(float) + (size_t)ltot * V * sizeof(float);
  if (lds > 160 * 1024) return P2R_EINVAL;
  if (form == 0) {
    return gcn3_launch<0, false>(p, ltot, blocks, lds, x, Wp, coef, bias_cv, addend, z, stats_partial, nullptr, nullptr,
                                 nullptr, stream_h);
  }
  if (addend_mask) {     // addend masked on the way in (the ST-GCN chain), with or without the BatchNorm-backward epilogue
    if (!addend || ((uintptr_t)addend_mask % 4) != 0) return P2R_EINVAL;
    if (bwd)
      return gcn3_launch<1, true, true>(p, ltot, blocks, lds, x, Wp, coef, bias_cv, addend, z, stats_partial, bwd_u,
                                        bwd_mask, bwd_fin, stream_h, addend_mask);
    return gcn3_launch<1, false, true>(p, ltot, blocks, lds, x, Wp, coef, bias_cv, addend, z, stats_partial, nullptr, nullptr,
                                       nullptr, stream_h, addend_mask);
  }
  if (bwd) return gcn3_launch<1, true>(p, ltot, blocks, lds, x, Wp, coef, bias_cv, addend, z, stats_partial, bwd_u, bwd_mask,
                                       bwd_fin, stream_h);
  return gcn3_launch<1, false>(p, ltot, blocks, lds, x, Wp, coef, bias_cv, addend, z, stats_partial, nullptr, nullptr, nullptr,
                               stream_h);
}

extern "C" int p2r_stgcn_gcn3_forward(int N, int T, int V, int K, int ltot, int form, const float *x, const float *Wp,
                                      const float *coef, const float *bias_cv, const float *addend, float *z,
                                      float *stats_partial, int *n_partials, const float *bwd_u,
                                      const unsigned char *bwd_mask, const float *bwd_fin, void *stream_h) {
  return gcn3_forward_impl(N, T, V, K, ltot, form, x, Wp, coef, bias_cv, addend, z, stats_partial, n_partials, bwd_u, bwd_mask,
                           bwd_fin, nullptr, stream_h);
}

// The data-gradient launch (form 1; with the BatchNorm-backward epilogue when bwd_* are given, plain when they and
// stats_partial are NULL) whose addend is MASKED on the way in: z += addend where addend_mask (N,64,T,53 bytes) is non-zero.  With addend = the gradient arriving at a block's
// output and addend_mask = that block's ReLU mask this is the residual-branch gradient dout * mask, which the
// BatchNorm-backward pass then does not have to write.  addend_mask 4-byte aligned.
extern "C" int p2r_stgcn_gcn3_data_gradient_masked_addend(int N, int T, int V, int K, int ltot, const float *x,
                                                          const float *Wp, const float *coef, const float *addend,
                                                          const unsigned char *addend_mask, float *z,
                                                          float *stats_partial, const float *bwd_u,
                                                          const unsigned char *bwd_mask, const float *bwd_fin,
                                                          void *stream_h) {
  if (!addend_mask) return P2R_EINVAL;
  return gcn3_forward_impl(N, T, V, K, ltot, 1, x, Wp, coef, nullptr, addend, z, stats_partial, nullptr, bwd_u, bwd_mask,
                           bwd_fin, addend_mask, stream_h);
}
