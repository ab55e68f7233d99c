// stgcn_gcn2.hip -- fused spatial graph convolution of the ST-GCN backbone, second generation, gfx950.
//
// Same operator as stgcn_gcn.hip (ConvTemporalGraphical.forward, reference
// models/p2rnet/modules/stgcn_layers.py:57-67, and its data gradient):
//     Z[:, (t, w)] = bias_cv[:, w] + sum_k W_k . ( sum_{v in list_k(w)} a_k(v, w) X[:, (t, v)] )
// with X, Z (64 x T*V) per sequence, W_k (64 x 64), and the sparse neighbour lists of the K = 11 planes.
//
// What changed against the first generation, and why:
//   * An MFMA n-tile is now 16 FRAMES OF ONE JOINT, not 16 consecutive (frame, joint) columns.  The
//     neighbour list of a (plane, n-tile) unit is then wave-uniform: a unit whose list is empty is
//     skipped exactly (454 of 583 units remain in the forward, 369 in the data gradient; tiles of 16
//     consecutive columns skipped almost nothing), the list entries are broadcast LDS reads instead of
//     per-lane table look-ups, and the B-operand gathers are conflict-free by construction: lane
//     (g = lane >> 4, r = lane & 15) reads row (4 s + g) at column 53 r + v -- 53 is odd, so the 16
//     frames fall in 16 distinct banks, and the row stride 16 * 53 = 848 == 16 (mod 32) puts the second
//     lane group of a 32-lane LDS pass on the other 16 banks.
//   * The 64 input channels are processed in four phases of 16 (one MFMA k-step group per phase and
//     m-tile), each phase's 16 x 848 slice of the X tile living in one of two 53 KB LDS buffers.  The
//     slices are copied by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write pass) in the global
//     order [channel][frame][joint], one 1 KB piece per wave and plane iteration, so the copy of phase
//     p+1 is spread under the MFMAs of phase p and a wave never waits for a piece it has just issued.
//   * Persistent workgroups (one per CU) walk the tiles; accumulators (64 rows x 16 frames per joint,
//     up to 7 joints per wave) live in registers across the four phases; the joints are dealt to the
//     waves so that every wave has the same number of non-empty (plane, joint) units.
//   * W is pre-permuted by the caller to [plane][phase][m-tile][lane][4] so that a lane's four
//     A-operand values of a phase are one 16-byte load and a wave's load is one contiguous 1 KB.
//
// Bit-level behaviour: every output element is the same fmaf / MFMA chain for a given column
// whatever the tile shape; run-to-run deterministic (no atomics).
#include "p2r_common.h"

#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int G2_F = 16;            // frames per tile = columns of an MFMA n-tile
constexpr int G2_CP = 16;           // channels per phase
constexpr int G2_NPH = 4;           // phases (64 channels)

struct G2Params {
  int T, V, K;
  int tiles_per_seq, total_tiles;
  int ltot;                         // rows of the (nbr, coef) tables = sum of the per-plane list lengths
  int vec;                          // 1: rows are 16-byte aligned and T*V % 4 == 0 -> 16-byte DMA pieces
};

// LDS-DMA pieces as inline assembly: with the builtin, hipcc treats every later LDS read as possibly aliasing the
// copy in flight and puts `s_waitcnt vmcnt(0)` in front of it -- which here would stall every record fetch on the
// piece (and the A-operand prefetch) just issued.  The copies land in the buffer nobody reads during the current
// phase; the issuing wave waits for them (vmcnt(0)) right before the phase barrier.  M0 = LDS destination of lane 0.
__device__ __forceinline__ unsigned g2_lds_addr(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
__device__ __forceinline__ void g2_dma16(const float *src, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(g2_lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
__device__ __forceinline__ void g2_dma4(const float *src, float *lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(g2_lds_addr(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

// Work stream of a wave (global memory, written by gcn2_fill_stream_kernel from the caller's template and the current
// coefficients right before the main kernel): one 48-byte record per pass,
//   [0]      descriptor: bits 0-2 accumulator slot, bit 3 first record of a visit (switch the A operands), bit 4 at most two entries, bits 8-11
//            plane, bits 12-15 plane of the wave's next visit (15: first plane of the next phase).  A visit is one
//            pass over the slots in ascending order for one plane; a list longer than six entries continues in an
//            extra visit of the same plane (the operator is linear in the B operand)
//   [1..3]   six 16-bit byte offsets of the source joints inside an LDS row (joint * VS * 4)
//   [4..9]   six coefficients (0 for padding)          [10..11] unused
constexpr int G2_UMAX = 80;                  // records per wave (incl. two zero records read ahead by the pipeline)
constexpr int G2_REC = 12;                   // dwords per record
constexpr int G2_HDR = 16;                   // header dwords per wave: [0] records, [1] first plane, [2..8] slot joints, [9] visits
constexpr int G2_WSTRIDE = G2_UMAX * G2_REC + G2_HDR;

// template (table indices in the coefficient slots) + coefficient table -> work stream (values)
__global__ void gcn2_fill_stream_kernel(int n, const int *__restrict__ templ, const float *__restrict__ coef,
                                        int *__restrict__ work) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  int v = templ[e];
  const int q = e % G2_WSTRIDE, f = q % G2_REC;
  if (q < G2_UMAX * G2_REC && f >= 4 && f < 10) v = v >= 0 ? __float_as_int(coef[v]) : 0;
  work[e] = v;
}

// BWD (data-gradient launches): the statistics epilogue emits the reduction pass of the BatchNorm + residual + ReLU
// backward of the block in front (what p2r_bn_bwd_reduce with relu = 3 computes from the stored result): per channel
// (sum g', sum g' * uhat), g' = (op(x) + addend) where the ReLU mask byte is set, uhat = (u - mean) * invstd with
// u = bwd_u the saved input of that BatchNorm, (mean, invstd) = rows 0, 1 of bwd_fin [4][64].
template <int NW, int SLOTS, int VT, int LAYOUT, bool BWD>
__global__ __launch_bounds__(NW * 64, NW / 4) void gcn2_kernel(
    G2Params p, const float *__restrict__ x, const float *__restrict__ Wp, const int *__restrict__ stream_g,
    const float *__restrict__ bias_cv, const float *__restrict__ addend, float *__restrict__ z,
    float *__restrict__ stats_partial, const float *__restrict__ bwd_u, const unsigned char *__restrict__ bwd_mask,
    const float *__restrict__ bwd_fin) {
  constexpr int V = VT;
  constexpr int RS = G2_F * V;                        // LDS row stride (floats): 848 == 16 (mod 32)
  constexpr int BUF = G2_CP * RS;                     // floats per phase buffer
  // position of (frame f, joint v) inside a row of 16 * V floats: LAYOUT 0 = [f][v] (the (N,C,T,V) tensor as it is),
  // LAYOUT 1 = [v][f] (16-frame blocks stored joint-major)
  constexpr int FS = LAYOUT == 0 ? V : 1, VS = LAYOUT == 0 ? 1 : G2_F;
  extern __shared__ float lds[];
  float *rowstat = lds + 2 * BUF;                                         // [NW][64][2]
  float *bias_l = rowstat + NW * 128;                                     // [64][V] bias table (zeros without bias)
  float *bstat = bias_l + 64 * V;                                         // [64][2] (mean, invstd) of the BWD epilogue

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;

  for (int e = tid; e < NW * 128; e += NW * 64) rowstat[e] = 0.f;
  for (int e = tid; e < 64 * V; e += NW * 64) bias_l[e] = bias_cv ? bias_cv[e] : 0.f;
  if (tid < 64) {
    bstat[2 * tid] = BWD ? bwd_fin[tid] : 0.f;
    bstat[2 * tid + 1] = BWD ? bwd_fin[64 + tid] : 1.f;
  }
  __syncthreads();

  // The work stream is read with SCALAR loads (wave-uniform addresses into read-only global memory): descriptors,
  // offsets and coefficients live in SGPRs.  On gfx950 fp32 MFMAs and VALU instructions of the two waves of a SIMD
  // do not overlap (tools/ubench/mfma_valu_overlap.hip: their times add up; SALU, scalar loads and LDS reads do
  // overlap), so every v_readfirstlane / v_mov / v_and taken out of the record handling is MFMA time won.
  const int *ws = stream_g + wave * G2_WSTRIDE;
  const int4 *st = reinterpret_cast<const int4 *>(ws);                 // 3 int4 per record
  const int *hdr = ws + G2_UMAX * G2_REC;
  const int nvisits = __builtin_amdgcn_readfirstlane(hdr[9]);
  const int plane0 = __builtin_amdgcn_readfirstlane(hdr[1]);
  int sj[7];                                          // joint of each accumulator slot, -1: unused.  Slots 0-3 hold a
                                                      // run of consecutive joints (sj[0] + i), slots 4-6 a second run
#pragma unroll
  for (int i = 0; i < 7; ++i) sj[i] = i < SLOTS ? __builtin_amdgcn_readfirstlane(hdr[2 + i]) : -1;
  const int na = (sj[0] >= 0) + (sj[1] >= 0) + (sj[2] >= 0) + (sj[3] >= 0);
  const int nb = (sj[4] >= 0) + (sj[5] >= 0) + (sj[6] >= 0);

  const size_t row_stride = (size_t)p.T * V;
  const char *xl0 = reinterpret_cast<const char *>(lds + g * RS + r * FS);   // lane's gather base (row g, frame r)

  // ---- LDS-DMA of one phase slice (16 channel rows x frames*V floats), split into per-wave pieces ----------
  constexpr int NV4 = BUF / 4;                        // float4 elements per slice
  constexpr int PIECES16 = (NV4 + 63) / 64;           // wave instructions per slice (53 for V = 53)
  constexpr int PW16 = (PIECES16 + NW - 1) / NW;      // per wave
  constexpr int PIECES4 = (BUF + 63) / 64;
  constexpr int PW4 = (PIECES4 + NW - 1) / NW;
  auto dma_piece = [&](int piece_i, float *buf, const float *xrow0, int valid_cols) {
    // piece_i: this wave's piece_i-th piece of the slice; xrow0: global address of (channel row 0 of the slice, frame t0)
    if (p.vec && valid_cols == RS) {
      const int pc = piece_i * NW + wave;
      if (piece_i < PW16 && pc < PIECES16) {
        const int e = pc * 64 + lane;                 // float4 index in the slice
        const int row = e / (RS / 4), c4 = e - row * (RS / 4);
        if (e < NV4) g2_dma16(xrow0 + (size_t)row * row_stride + 4 * c4, buf + pc * 256);
      }
    } else {
      // tail tiles / unaligned rows: 4-byte pieces, several per call
      constexpr int PER = (PW4 + PW16 - 1) / PW16;
#pragma unroll 1
      for (int q = 0; q < PER; ++q) {
        const int pi = piece_i * PER + q;
        const int pc = pi * NW + wave;
        if (pi < PW4 && pc < PIECES4) {
          const int e = pc * 64 + lane;
          const int row = e / RS, col = e - row * RS;
          if (e < BUF && col < valid_cols) g2_dma4(xrow0 + (size_t)row * row_stride + col, buf + pc * 64);
        }
      }
    }
  };

  f32x4 acc[SLOTS][4];
  float a_nxt[4][4];                                  // A operands of the next (plane, phase): W'[k][ph][m][lane][s]
  auto load_a = [&](int k, int ph) {
    const float4 *wp = reinterpret_cast<const float4 *>(Wp) + ((size_t)(k * G2_NPH + ph) * 4) * 64 + lane;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float4 u = wp[m * 64];
      a_nxt[m][0] = u.x; a_nxt[m][1] = u.y; a_nxt[m][2] = u.z; a_nxt[m][3] = u.w;
    }
  };

  // gathers of one record: xv[j][s] = X[row 4 s + g][frame r, joint j-th source], first NE entries
  auto gather = [&](auto ne, const int4 &e0, const char *xl, float (&xv)[6][4]) {
    constexpr int NE = decltype(ne)::value;
    const int w3[3] = {e0.y, e0.z, e0.w};
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int off = (j & 1) ? (int)((unsigned)w3[j >> 1] >> 16) : (w3[j >> 1] & 0xffff);
      const float *src = reinterpret_cast<const float *>(xl + off);
#pragma unroll
      for (int s = 0; s < 4; ++s) xv[j][s] = src[s * 4 * RS];
    }
  };
  auto combine = [&](auto ne, const int4 &e1, const int4 &e2, const float (&xv)[6][4], float (&b)[4]) {
    constexpr int NE = decltype(ne)::value;
    const float c[6] = {__int_as_float(e1.x), __int_as_float(e1.y), __int_as_float(e1.z),
                        __int_as_float(e1.w), __int_as_float(e2.x), __int_as_float(e2.y)};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float v = c[0] * xv[0][s];
#pragma unroll
      for (int j = 1; j < NE; ++j) v = fmaf(c[j], xv[j][s], v);
      b[s] = v;
    }
  };
  using NE2 = std::integral_constant<int, 2>;
  using NE6 = std::integral_constant<int, 6>;

  int tile = blockIdx.x;
  // prologue: phase 0 of the first tile, all pieces at once
  if (tile < p.total_tiles) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * G2_F;
    const int vc = min(G2_F, p.T - t0) * V;
    const float *xr = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    for (int i = 0; i < PW16; ++i) dma_piece(i, lds, xr, vc);
    load_a(plane0, 0);
  }

  for (; tile < p.total_tiles; tile += gridDim.x) {
    const int seq = tile / p.tiles_per_seq, t0 = (tile % p.tiles_per_seq) * G2_F;
    const int frames = min(G2_F, p.T - t0);
    const float *xg = x + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    float *zg = z + (size_t)seq * 64 * row_stride + (size_t)t0 * V;
    const float *ag = addend ? addend + (size_t)seq * 64 * row_stride + (size_t)t0 * V : nullptr;
    const float *ug = BWD ? bwd_u + (size_t)seq * 64 * row_stride + (size_t)t0 * V : nullptr;
    const unsigned char *mg = BWD ? bwd_mask + (size_t)seq * 64 * row_stride + (size_t)t0 * V : nullptr;
    const int ntile = tile + gridDim.x;
    const bool has_next = ntile < p.total_tiles;
    const int nseq = has_next ? ntile / p.tiles_per_seq : 0, nt0 = has_next ? (ntile % p.tiles_per_seq) * G2_F : 0;
    const int nvc = min(G2_F, p.T - nt0) * V;
    const float *nxg = x + (size_t)nseq * 64 * row_stride + (size_t)nt0 * V;

    // accumulators start from the bias table (LDS copy: 16 reads per slot off one base register)
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const float *bl = bias_l + 4 * g * V + (sj[i] >= 0 ? sj[i] : 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][m][q] = bl[(16 * m + q) * V];
    }

#pragma unroll 1
    for (int ph = 0; ph < G2_NPH; ++ph) {
      // slice `ph` has landed (every wave waited for its own pieces) and nobody reads the other buffer any more
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      float *buf_nxt = lds + ((ph + 1) & 1) * BUF;
      const char *xl = xl0 + (ph & 1) * BUF * sizeof(float);
      // source of the slice that is copied during this phase
      const bool copy = ph + 1 < G2_NPH || has_next;
      const float *src = (ph + 1 < G2_NPH) ? xg + (size_t)(ph + 1) * G2_CP * row_stride : nxg;
      const int svc = (ph + 1 < G2_NPH) ? frames * V : nvc;
      int pieces = 0;

      // Software pipeline over the wave's records: while the 16 MFMAs of a record run, the B operand of the next
      // record is gathered and the record after that is fetched.  The accumulator slot is a compile-time index
      // (unrolled slot loop; a record is consumed by the body of its slot), so the tiles stay put in registers.
      float a[4][4], b_cur[4];
      int4 n0 = st[0], n1 = st[1], n2 = st[2];
      {
        float xv[6][4];
        gather(NE6{}, n0, xl, xv);
        combine(NE6{}, n1, n2, xv, b_cur);
      }
      int d = __builtin_amdgcn_readfirstlane(n0.x);     // descriptor of the record whose B operand is b_cur
      n0 = st[3]; n1 = st[4]; n2 = st[5];
      int u = 0;
#pragma unroll 1
      for (int visit = 0; visit < nvisits; ++visit) {
        // first record of a plane: take the prefetched A operands, prefetch the next plane's, issue one DMA piece --
        // both have a whole plane of MFMAs to land before anything waits for them
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s) a[m][s] = a_nxt[m][s];
        {
          const int nk = (d >> 12) & 15;
          if (nk == 15) load_a(plane0, (ph + 1) & (G2_NPH - 1));
          else load_a(nk, ph);
        }
        if (copy && pieces < PW16) dma_piece(pieces, buf_nxt, src, svc);
        ++pieces;
        bool started = false;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
          if ((d & 7) == i && (!(d & 8) || !started)) {
            started = true;
            // the next record's B operand is gathered under this record's MFMAs; short records (<= 2 entries, most
            // of them) read a third of the LDS words.  One MFMA block for both kinds keeps the accumulators in place.
            const bool short_rec = (__builtin_amdgcn_readfirstlane(n0.x) & 16) != 0;
            float xv[6][4];
            if (short_rec) gather(NE2{}, n0, xl, xv);
            else gather(NE6{}, n0, xl, xv);
            const int4 m0 = st[3 * u + 6], m1 = st[3 * u + 7], m2 = st[3 * u + 8];
            __builtin_amdgcn_sched_barrier(0);       // all LDS reads are requested before the MFMAs ...
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int m = 0; m < 4; ++m)
                acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b_cur[s], acc[i][m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);       // ... and consumed after them: no s_waitcnt inside the MFMA run
            if (short_rec) combine(NE2{}, n1, n2, xv, b_cur);
            else combine(NE6{}, n1, n2, xv, b_cur);
            d = __builtin_amdgcn_readfirstlane(n0.x);
            n0 = m0; n1 = m1; n2 = m2;
            ++u;
          }
        }
      }
      for (; copy && pieces < PW16; ++pieces) dma_piece(pieces, buf_nxt, src, svc);
    }

    // ---- epilogue: D[row = 16 m + 4 g + q][frame r] of joint sj[i]; statistics of the stored values.
    float *rs = rowstat + wave * 128;
    if (!BWD && stats_partial) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float s1 = 0.f, s2 = 0.f;
          if (r < frames) {
#pragma unroll
            for (int i = 0; i < SLOTS; ++i)
              if (sj[i] >= 0) {
                const float v = acc[i][m][q];
                s1 += v;
                s2 = fmaf(v, v, s2);
              }
          }
          s1 = p2r_row16_sum(s1);
          s2 = p2r_row16_sum(s2);
          if (r == 0) {             // slot owned by (wave, row): plain read-modify-write, deterministic
            rs[2 * (16 * m + 4 * g + q)] += s1;
            rs[2 * (16 * m + 4 * g + q) + 1] += s2;
          }
        }
    }
    if (LAYOUT == 0 && p.vec && frames == G2_F) {
      // Full tiles leave through LDS: a (row, frame) line of the (N,C,T,V) tensor holds 16 consecutive joints, which
      // belong to several waves -- stored from the accumulators it would go out as 12/16-byte fragments (measured:
      // 2.4x the algorithmic HBM write bytes, partial lines).  The slice buffer of the last phase is free once every
      // wave has finished it: 16 rows at a time are laid out there as the tensor has them and written as whole
      // 16-byte-per-lane rows.  LDS-only waits in front of the barriers: the global stores stay in flight.
      float *stg = lds + ((G2_NPH - 1) & 1) * BUF;
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
        constexpr int R4 = RS / 4;                          // float4 per row (212)
        constexpr int RIT = (R4 + 63) / 64;                 // 4
        float4 uv[BWD ? 2 : 1][BWD ? RIT : 1];
        float4 ad[BWD ? 2 : 1][BWD ? RIT : 1];
        unsigned mk[BWD ? 2 : 1][BWD ? RIT : 1];
        if (BWD) {   // saved activation, mask bytes and addend of this wave's two rows: in flight across the staging barrier
          const size_t r0 = (size_t)(16 * m + 2 * wave) * row_stride;
          const float4 *u4 = reinterpret_cast<const float4 *>(ug + r0);
          const unsigned *m4 = reinterpret_cast<const unsigned *>(mg + r0);
          const float4 *a4 = reinterpret_cast<const float4 *>(ag ? ag + r0 : nullptr);
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int it = 0; it < RIT; ++it) {
              const int c4 = it * 64 + lane;
              uv[rr][it] = c4 < R4 ? u4[(size_t)rr * (row_stride / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
              mk[rr][it] = c4 < R4 ? m4[(size_t)rr * (row_stride / 4) + c4] : 0u;
              if (a4 && c4 < R4) ad[rr][it] = a4[(size_t)rr * (row_stride / 4) + c4];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float4 *zrow = reinterpret_cast<float4 *>(zg + (size_t)16 * m * row_stride);
        const float4 *arow = reinterpret_cast<const float4 *>(ag ? ag + (size_t)16 * m * row_stride : nullptr);
        const float4 *srow = reinterpret_cast<const float4 *>(stg);
        if (BWD) {   // a wave takes two whole rows: the per-channel sums stay in registers until the row is done
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr, c = 16 * m + row;
            const float mu = bstat[2 * c], is = bstat[2 * c + 1];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int it = 0; it < RIT; ++it) {
              const int c4 = it * 64 + lane;
              if (c4 < R4) {
                float4 v = srow[row * R4 + c4];
                if (arow) { v.x += ad[rr][it].x; v.y += ad[rr][it].y; v.z += ad[rr][it].z; v.w += ad[rr][it].w; }
                zrow[(size_t)row * (row_stride / 4) + c4] = v;
                const float4 uu = uv[rr][it];
                const unsigned mm = mk[rr][it];
                const float g0 = (mm & 0xffu) ? v.x : 0.f, g1 = (mm & 0xff00u) ? v.y : 0.f;
                const float g2 = (mm & 0xff0000u) ? v.z : 0.f, g3 = (mm & 0xff000000u) ? v.w : 0.f;
                s1 += (g0 + g1) + (g2 + g3);
                s2 = fmaf(g0, (uu.x - mu) * is, s2); s2 = fmaf(g1, (uu.y - mu) * is, s2);
                s2 = fmaf(g2, (uu.z - mu) * is, s2); s2 = fmaf(g3, (uu.w - mu) * is, s2);
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
              float4 v = srow[e];
              if (arow) {           // e.g. the gradient of the block's residual branch, added on the way out
                const float4 ad = arow[(size_t)row * (row_stride / 4) + c4];
                v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
              }
              zrow[(size_t)row * (row_stride / 4) + c4] = v;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    } else {
      // ragged tiles / unaligned rows: straight from the accumulators.  A wave's slots 0-3 hold CONSECUTIVE joints
      // and so do its slots 4-6, so the values of one (row, frame) of a run are contiguous and leave as one 16- or
      // 12-byte store per lane (4-byte aligned).
      typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
      typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
      typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * m + 4 * g + q;
          float *drow = zg + (size_t)row * row_stride + r * FS;
          if (r < frames && ag) {
            const float *arow = ag + (size_t)row * row_stride + r * FS;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i)
              if (sj[i] >= 0) acc[i][m][q] += arow[sj[i] * VS];
          }
          if (BWD) {
            const float mu = bstat[2 * row], is = bstat[2 * row + 1];
            float s1 = 0.f, s2 = 0.f;
            if (r < frames) {
              const float *urow = ug + (size_t)row * row_stride + r * FS;
              const unsigned char *mrow = mg + (size_t)row * row_stride + r * FS;
#pragma unroll
              for (int i = 0; i < SLOTS; ++i)
                if (sj[i] >= 0) {
                  const float gm = mrow[sj[i] * VS] ? acc[i][m][q] : 0.f;
                  s1 += gm;
                  s2 = fmaf(gm, (urow[sj[i] * VS] - mu) * is, s2);
                }
            }
            s1 = p2r_row16_sum(s1);
            s2 = p2r_row16_sum(s2);
            if (r == 0) {
              rs[2 * row] += s1;
              rs[2 * row + 1] += s2;
            }
          }
          if (r < frames) {
            if (LAYOUT == 0 && SLOTS == 7) {
              float *da = drow + sj[0];
              if (na == 4) *reinterpret_cast<f4u *>(da) = f4u{acc[0][m][q], acc[1][m][q], acc[2][m][q], acc[3][m][q]};
              else if (na == 3) *reinterpret_cast<f3u *>(da) = f3u{acc[0][m][q], acc[1][m][q], acc[2][m][q]};
              else if (na == 2) *reinterpret_cast<f2u *>(da) = f2u{acc[0][m][q], acc[1][m][q]};
              else if (na == 1) da[0] = acc[0][m][q];
              float *db = drow + sj[4];
              if (nb == 3) *reinterpret_cast<f3u *>(db) = f3u{acc[4][m][q], acc[5][m][q], acc[6][m][q]};
              else if (nb == 2) *reinterpret_cast<f2u *>(db) = f2u{acc[4][m][q], acc[5][m][q]};
              else if (nb == 1) db[0] = acc[4][m][q];
            } else {
#pragma unroll
              for (int i = 0; i < SLOTS; ++i)
                if (sj[i] >= 0) drow[sj[i] * VS] = acc[i][m][q];
            }
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

constexpr int G2_NW = 8, G2_SLOTS = 7;
constexpr int G2_LAYOUT = 0;          // LDS position of (frame, joint) inside a row: the tensor's own [frame][joint] order

}  // namespace

// x (N,64,T,V) -> z (N,64,T,V), V = 53.
//   Wp     [K][4 phases][4 m-tiles][64 lanes][4]: Wp[k][ph][m][16 g + r][s] = W_k[16 m + r][16 ph + 4 s + g]
//          (W_k (64 x 64), row = output channel; pass W_k^T for the data gradient)
//   coef   f32 [ltot][V]: coefficient table (values of A * importance at the list entries, as for
//          p2r_stgcn_gcn_forward); `stream` refers to it by flat index
//   stream_work int32 [8][80 * 12 + 16]: scratch, receives the stream with the coefficient values filled in
//   stream int32 [8 waves][80 * 12 + 16]: the static per-wave work stream (record layout in the kernel source;
//          built once per adjacency pattern by pose2room_amd/p2rnet/gcn_tables.build_stream).  Every joint must be
//          owned by exactly one (wave, slot); the joints of a wave's slots 0-3 must be consecutive (unused slots last),
//          and likewise those of its slots 4-6.
//   bias_cv [64][V] or NULL.   addend (N,64,T,V) or NULL: added to the result (z = op(x) + addend; the statistics
//   are those of op(x)).   stats_partial (optional) [n_partials][64][2].
// n_partials: number of workgroups = rows of stats_partial (returned through *n_partials; call with z == NULL to
// query it).
extern "C" int p2r_stgcn_gcn2_forward(int N, int T, int V, int K, int ltot, const float *x, const float *Wp,
                                      const float *coef, const int *stream, int *stream_work,
                                      const float *bias_cv, const float *addend, float *z, float *stats_partial,
                                      int *n_partials, const float *bwd_u, const unsigned char *bwd_mask,
                                      const float *bwd_fin, void *stream_h) {
  if (N < 0 || T <= 0 || V != 53 || K <= 0 || K >= 15 || ltot <= 0) return P2R_EINVAL;
  const bool bwd = bwd_u != nullptr;
  if (bwd != (bwd_mask != nullptr) || bwd != (bwd_fin != nullptr) || (bwd && !stats_partial)) return P2R_EINVAL;
  if (bwd && (((uintptr_t)bwd_u % 16) != 0 || ((uintptr_t)bwd_mask % 4) != 0)) return P2R_EINVAL;
  if (n_partials) *n_partials = 0;
  if (N == 0) return P2R_OK;
  G2Params p;
  p.T = T; p.V = V; p.K = K; p.ltot = ltot;
  p.tiles_per_seq = p2r_cdiv(T, G2_F);
  const long long tiles = (long long)N * p.tiles_per_seq;
  if (tiles > 0x7fffffffLL) return P2R_EINVAL;
  p.total_tiles = (int)tiles;
  // one predicate for the 16-byte DMA pieces of x AND the 16-byte row stores / loads of the staged epilogue: z, the
  // addend (a contiguous view may start anywhere) and the saved activation of the BWD epilogue must be aligned too
  p.vec = (((size_t)T * V) % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)z % 16) == 0 &&
           ((uintptr_t)addend % 16) == 0) ? 1 : 0;
  const int blocks = (int)(tiles < 256 ? tiles : 256);
  if (n_partials) *n_partials = blocks;
  if (!z) return P2R_OK;
  if (!stream_work) return P2R_EINVAL;
  const size_t lds = (size_t)2 * G2_CP * G2_F * V * sizeof(float) + (size_t)G2_NW * 128 * sizeof(float) +
                     (size_t)64 * V * sizeof(float) + 128 * sizeof(float);
  if (lds > 160 * 1024) return P2R_EINVAL;
  const int n_stream = G2_NW * G2_WSTRIDE;
  hipLaunchKernelGGL(gcn2_fill_stream_kernel, dim3((n_stream + 255) / 256), dim3(256), 0, p2r_stream(stream_h), n_stream,
                     stream, coef, stream_work);
  P2R_LAUNCH_CHECK();
  if (bwd) {
    auto kern = gcn2_kernel<G2_NW, G2_SLOTS, 53, G2_LAYOUT, true>;
    static unsigned char lds_ok[P2R_MAX_DEVICES];
    hipError_t e = p2r_allow_big_lds(kern, lds_ok);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(G2_NW * 64), lds, p2r_stream(stream_h), p, x, Wp, stream_work, bias_cv,
                       addend, z, stats_partial, bwd_u, bwd_mask, bwd_fin);
  } else {
    auto kern = gcn2_kernel<G2_NW, G2_SLOTS, 53, G2_LAYOUT, false>;
    static unsigned char lds_ok[P2R_MAX_DEVICES];
    hipError_t e = p2r_allow_big_lds(kern, lds_ok);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(G2_NW * 64), lds, p2r_stream(stream_h), p, x, Wp, stream_work, bias_cv,
                       addend, z, stats_partial, nullptr, nullptr, nullptr);
  }
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
