// stgcn_tconv.hip -- temporal (3,1) convolution of st_gcn_block fused with the preceding
// BatchNorm + ReLU, gfx950.
//
// Replaces tcn.0-tcn.2 of the reference's st_gcn_block (models/p2rnet/modules/
// stgcn_layers.py:399-411): BatchNorm2d -> ReLU -> Conv2d(64, 64, (3,1), padding (1,0)),
// which run there as a normalisation pass, a ReLU pass and an (NCHW<->NHWC transposed)
// implicit-GEMM convolution, each a full HBM round trip of the (N,64,T,V) activation.
//
// MI355X design: u = sum_dt W_dt . shift_dt(h) is the same "sum over planes of W_k . P_k(X)"
// shape as the graph convolution (stgcn_gcn.hip) with three planes whose operator is a
// one-frame shift, so the same MFMA tiling applies: a workgroup stages F+2 frames (one
// halo frame each side, zero outside the sequence) in LDS -- applying the BatchNorm affine
// and the ReLU while staging, so the normalised activation never exists in HBM -- and
// every lane reads its B operand at column offset (dt+1)*V.  v_mfma_f32_16x16x4_f32,
// 8 waves, A operands (W_dt rows) streamed from L2.  The weight gradient walks the same
// tiles with the roles of the MFMA dimensions swapped (reduction over columns).
#include "p2r_common.h"

namespace {

typedef float floatx4c __attribute__((ext_vector_type(4)));

constexpr int TC_C = 64;
constexpr int TC_NPO = 384;               // output columns per tile (24 n-tiles of 16)
constexpr int TC_THREADS = 512;
constexpr int TC_NT = 3;                  // n-tiles per wave

// ---- forward / data-gradient --------------------------------------------------------
// out[n,c,t,w] = bias[c] + sum_{p<3} sum_ci W[p][c][ci] * h[n,ci,t+p-1,w]
// h = relu(x*scale+shift) if scale != NULL else x; zero outside [0,T).
// Persistent: each workgroup walks tiles blockIdx.x, +gridDim.x, ...; the global loads of the
// NEXT tile are issued into registers before the MFMA phase of the current one, so HBM latency
// and transfer hide under compute even though only one workgroup fits a CU (LDS).
template <int TAPS>
__global__ __launch_bounds__(TC_THREADS, 2) void tconv_fused_kernel(
    int n_seq, int T, int V, int F, int tiles_per_seq, int row_len, const float *__restrict__ x,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ W,
    const float *__restrict__ bias, float *__restrict__ out, float *__restrict__ stats_partial) {
  extern __shared__ float hs[];   // [64][row_len], frames t0-HALO .. t0+F-1+HALO; then [64][2] output statistics
  constexpr int HALO = (TAPS - 1) / 2;
  constexpr int NCH = TAPS == 1 ? TC_NPO / 64 : 8;   // 64-column chunks per tile row (tile + halo <= 64 * NCH)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const size_t row_stride = (size_t)T * V;
  const int rs = (int)row_stride;        // T * V < 2^29 (checked by the launcher)
  const int total_tiles = n_seq * tiles_per_seq;

  // this lane's share of a tile: 8 rows (wave, wave+8, ..) x NCH chunks, each ONE buffer load through a descriptor of
  // the row (columns outside the tile or the tensor come back as zero from the hardware range check: no predicate,
  // clamp or exec-masked branch in the load path, see tconv_dw_kernel)
  float pre[8][NCH];
  auto issue_loads = [&](int tile) __attribute__((always_inline)) {
    const int seq = tile / tiles_per_seq;
    const int t0 = (tile % tiles_per_seq) * F;
    const int frames = min(F, T - t0);
    const int col0 = (t0 - HALO) * V;
    const int bytes = 4 * min(rs, col0 + (frames + 2 * HALO) * V);
    const float *xg = x + (size_t)seq * TC_C * row_stride;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float *>(xg + (size_t)(wave + 8 * h) * row_stride), 0, bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        pre[h][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, 4 * lane + 4 * (col0 + 64 * i), 0, 0));
    }
  };

  // one slot per (wave, row, moment): plain read-modify-writes by the one lane that owns the row in its wave, summed
  // over the waves in a fixed order at the end -- LDS atomics across waves made the sums (and with them every later
  // BatchNorm of the step) depend on wave arrival order from run to run
  float *rowstat = hs + TC_C * row_len;
  for (int e = tid; e < (TC_THREADS / 64) * 2 * TC_C; e += TC_THREADS) rowstat[e] = 0.f;

  int tile = blockIdx.x;
  if (tile < total_tiles) issue_loads(tile);
  for (; tile < total_tiles; tile += gridDim.x) {
    const int seq = tile / tiles_per_seq;
    const int t0 = (tile % tiles_per_seq) * F;
    const int frames = min(F, T - t0);
    const int ncols = frames * V;
    float *og = out + (size_t)seq * TC_C * row_stride + (size_t)t0 * V;

    // registers -> LDS with the BatchNorm affine + ReLU applied on the way
    bool in[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int q = lane + 64 * i, gc = (t0 - HALO) * V + q;
      in[i] = q < (frames + 2 * HALO) * V && gc >= 0 && gc < rs;
    }
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      const int c = wave + 8 * h;
      const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int q = lane + 64 * i;
        float v = pre[h][i];
        if (scale) v = fmaxf(fmaf(v, sc, sh), 0.f);
        v = in[i] ? v : 0.f;                                   // outside the sequence / tile
        if (q < row_len) hs[c * row_len + q] = v;
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < total_tiles) issue_loads(tile + gridDim.x);

    int colv[TC_NT], base[TC_NT];
    bool valid[TC_NT];
#pragma unroll
    for (int i = 0; i < TC_NT; ++i) {
      const int col = (wave * TC_NT + i) * 16 + r;
      colv[i] = col;
      valid[i] = col < ncols;
      base[i] = valid[i] ? col : 0;           // input column of plane p: base + p*V
    }
    floatx4c acc[TC_NT][4];              // start from the bias: its (L2-resident) loads overlap the first W wait
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float bv = bias ? bias[16 * m + 4 * g + q] : 0.f;
#pragma unroll
        for (int i = 0; i < TC_NT; ++i) acc[i][m][q] = bv;
      }

    const float *hg = hs + g * 16 * row_len;
#pragma unroll 1
    for (int ph = 0; ph < 2 * TAPS; ++ph) {    // (tap p, half of the 16 k-steps): keeps A operands at 32 VGPRs
      const int p = ph >> 1, hf = ph & 1;
      float a[4][8];                           // W[p][row 16m + r][ci 16g + 8hf .. +8)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float4 *wp = reinterpret_cast<const float4 *>(W + ((size_t)p * TC_C + 16 * m + r) * TC_C + 16 * g + 8 * hf);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float4 u = wp[q];
          a[m][4 * q + 0] = u.x; a[m][4 * q + 1] = u.y; a[m][4 * q + 2] = u.z; a[m][4 * q + 3] = u.w;
        }
      }
#pragma unroll
      for (int i = 0; i < TC_NT; ++i) {
        const float *hb = hg + base[i] + p * V + 8 * hf * row_len;
        float b[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) b[s2] = valid[i] ? hb[s2 * row_len] : 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s2], b[s2], acc[i][m], 0, 0, 0);
      }
    }

    // Besides the store, the batch statistics of the output for the BatchNorm that follows (tcn.3 / the next
    // MLP stage) are taken from the values just computed: lanes of a row by DPP, waves and tiles by LDS atomics.
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * m + 4 * g + q;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < TC_NT; ++i) {
          if (!valid[i]) continue;
          const float v = acc[i][m][q];
          og[(size_t)row * row_stride + colv[i]] = v;
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
        if (stats_partial) {
          s1 = p2r_row16_sum(s1);
          s2 = p2r_row16_sum(s2);
          if (r == 0) {
            float *slot = rowstat + wave * 2 * TC_C + 2 * row;
            slot[0] += s1;
            slot[1] += s2;
          }
        }
      }
    __syncthreads();   // every wave is done with the LDS tile before it is overwritten
  }
  __syncthreads();   // (a workgroup without tiles reaches this point without having passed a barrier)
  if (stats_partial && tid < 2 * TC_C) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < TC_THREADS / 64; ++w) tot += rowstat[w * 2 * TC_C + tid];
    stats_partial[(size_t)blockIdx.x * 2 * TC_C + tid] = tot;
  }
}

// ---- weight gradient ------------------------------------------------------------------
// dW[p][c][ci] = sum_{n,t,w} dout[n,c,t,w] * h[n,ci,t+p-1,w]   (h as above)
// persistent grid; wave = (ci tile nt = wave&3, row group mh = wave>>2): TAPS planes x TW_MT row
// tiles of accumulators; reduction steps of 4 consecutive columns.
constexpr int TW_F = 4;            // frames per tile; the single-tap launch for 20 columns per frame (the position-embedding MLP) uses 12: see p2r_stgcn_tconv_weight_grad
#define TW_WAVES 16   // four waves per SIMD: same time as 8 for the 3-tap form, -15 % for the single-tap form
constexpr int TW_THREADS = 64 * TW_WAVES;
constexpr int TW_MT = 4 / (TW_WAVES / 4);     // 16-row c tiles per wave (waves = 4 ci tiles x TW_WAVES/4 row groups)

// VS = compile-time joint count (0: run time).  With VS fixed the reduction loop is fully unrolled and every
// LDS read carries an immediate offset: no address arithmetic on the VALU, which on gfx950 does not overlap
// with the partner waves' MFMAs.
__host__ __device__ constexpr int tw_row(int cols) {   // + room for the last (partial) 4-column step and one prefetched
  int r = cols + 7;                                    // step, then up to == 2 (mod 32): conflict-free column reads
  while (r % 32 != 2) ++r;
  return r;
}
// DZ: the launch also forms the gradient of the BatchNorm INPUT on its way -- the tile already holds z (the h tile
// is relu(z * scale + shift)), so with the gradient dh of the BatchNorm-ReLU output loaded next to it
//   dz = scale * ((z * scale + shift > 0 ? dh : 0) - m1 - (z - mean) * invstd * m2)       (bn_bwd_apply, mask form 2)
// leaves from the staging pass and the separate apply pass (read dh, read z, write dz) is gone: one more read and
// one write of the tensor in a kernel that leaves two thirds of the HBM rate unused.
// AMAX (DZ only; split16 mode): the float bits of max |dz| are merged into *amax (zeroed by the launcher) -- the range word
// of the split16 graph-conv gradient kernels that consume dz (split16.h).
template <int TAPS, int VS, int F = TW_F, bool DZ = false, bool XF = true, bool AMAX = false>
__global__ __launch_bounds__(TW_THREADS, TW_WAVES == 8 ? 2 : 1) void tconv_dw_kernel(
    int n_seq, int T, int V_, int row_d_, int row_h_, const float *__restrict__ x,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ dout,
    float *__restrict__ dw_partial, float *__restrict__ dbias_partial, const float *__restrict__ dh = nullptr,
    const float *__restrict__ fin = nullptr, const float *__restrict__ m12 = nullptr, float *__restrict__ dz = nullptr,
    unsigned *__restrict__ amax = nullptr) {
  static_assert(!AMAX || DZ, "the range word belongs to dz");
  unsigned am = 0;
  constexpr int HALO = (TAPS - 1) / 2;
  // 64-column chunks of the h tile per row (the unrolled instances stage only the chunks that hold columns: the pad
  // columns behind them are zeroed once, below, and never written again)
  constexpr int NH = VS > 0 ? ((F + 2 * HALO) * VS + 63) / 64 : (TAPS == 1 ? 4 : 6);
  // joint count and row strides are compile-time constants in the fully unrolled instances (write predicates fold)
  const int V = VS > 0 ? VS : V_;
  const int row_d = VS > 0 ? tw_row(F * VS) : row_d_, row_h = VS > 0 ? tw_row((F + 2 * HALO) * VS) : row_h_;
  static_assert(!DZ || VS > 0, "the fused BatchNorm-backward form exists for the unrolled instances");
  // chunks of the h row that hold columns of the tile's own frames (the ones dz is formed for)
  constexpr int ZI0 = DZ ? (HALO * VS) / 64 : 0, ZI1 = DZ ? ((HALO + F) * VS - 1) / 64 : -1, NZ = ZI1 - ZI0 + 1;
  extern __shared__ float lds[];
  float *ds = lds;                       // [64][row_d]   dout tile, frames t0 .. t0+F-1
  float *hs = lds + TC_C * row_d;        // [64][row_h]   h tile, frames t0-HALO .. t0+F-1+HALO

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // rows and row pointers in SGPRs
  const int g = lane >> 4, r = lane & 15;
  const int nt = wave & 3, mh = wave >> 2;          // 16 ci columns x (64 / TW_MH) c rows per wave
  const int tiles_per_seq = (T + F - 1) / F;
  const int total_tiles = n_seq * tiles_per_seq;
  const size_t row_stride = (size_t)T * V;

  // pad columns past the staged chunks (row strides are rounded up) are read by the last partial
  // 4-column step: keep them zero
  for (int e = tid; e < TC_C * (row_d + row_h); e += TW_THREADS) lds[e] = 0.f;

  float bsum[TC_C / (TW_THREADS / 64)];    // per-lane share of the row sums of dout (bias gradient), rows wave, wave+8, ..
#pragma unroll
  for (int h = 0; h < TC_C / (TW_THREADS / 64); ++h) bsum[h] = 0.f;
  floatx4c acc[TAPS][TW_MT];
#pragma unroll
  for (int p = 0; p < TAPS; ++p)
#pragma unroll
    for (int m = 0; m < TW_MT; ++m) acc[p][m] = floatx4c{0.f, 0.f, 0.f, 0.f};

  // Persistent loop with register prefetch: the global loads of the NEXT tile are issued right after the current
  // tile has been written to LDS, so HBM latency and transfer hide under the MFMA phase (one workgroup per CU).
  constexpr int NR = TC_C / (TW_THREADS / 64);     // rows per wave: wave, wave + 8, ...
  float ph_[NR][NH], pd_[NR][4], pz_[NR][NZ > 0 ? NZ : 1];
  // The loads of a tile are NH + 4 (+ NZ) pieces per row, each ONE `buffer_load_dword` through a descriptor of the row
  // built from wave-uniform values: the hardware range check returns zero for columns outside the tile or the tensor (a
  // negative column wraps to a huge unsigned offset), so the load path has no predicate, no clamp and no exec-masked
  // branch -- with one, hipcc's wait insertion put `s_waitcnt vmcnt(0)` in front of every piece.  issue_loads(tile)
  // issues all pieces (the first tile); the fully unrolled instances spread them over the reduction steps of the
  // current tile instead, so that no wave has to push 40 loads through a memory pipeline that all 256 workgroups
  // fill at the same moment before it reaches its MFMAs.
  constexpr int PER_ROW = NH + 4 + NZ, PIECES = NR * PER_ROW;
  const int rs = (int)row_stride;           // T * V < 2^29 (checked by the launcher): 32-bit byte offsets inside a row
  const int lane4 = lane * 4;
  size_t ld_off = 0;                        // element offset of the tile's sequence
  int ld_t0v = 0, ld_col0 = 0, ld_hcols = 0, ld_hbytes = 0, ld_dbytes = 0, ld_zbytes = 0;
  auto set_tile = [&](int tile, bool valid) __attribute__((always_inline)) {
    const int seq = tile / tiles_per_seq;
    const int t0 = (tile % tiles_per_seq) * F;
    const int frames = min(F, T - t0);
    ld_hcols = (frames + 2 * HALO) * V;
    ld_off = (size_t)seq * TC_C * row_stride;
    ld_t0v = t0 * V;
    ld_col0 = (t0 - HALO) * V;
    ld_hbytes = valid ? 4 * min(rs, ld_col0 + ld_hcols) : 0;       // no next tile: every load returns zero
    ld_dbytes = valid ? 4 * frames * V : 0;
    ld_zbytes = valid ? 4 * (ld_t0v + frames * V) : 0;             // dh / dz: up to the end of the tile's own frames
  };
  auto issue_piece = [&](int j) __attribute__((always_inline)) {   // j = hh * PER_ROW + i, a compile-time constant at every call site
    const int hh = j / PER_ROW, i = j % PER_ROW;
    const int c = wave + hh * (TW_THREADS / 64);
    const size_t row = ld_off + (size_t)c * row_stride;
    if (i < NH) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + row), 0, ld_hbytes, 0x00020000);
      ph_[hh][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane4 + 4 * (ld_col0 + 64 * i), 0, 0));
    } else if (i < NH + 4) {
      const __amdgpu_buffer_rsrc_t rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dout + row + ld_t0v), 0, ld_dbytes, 0x00020000);
      pd_[hh][i - NH] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane4 + 256 * (i - NH), 0, 0));
    } else {
      const int iz = i - NH - 4;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dh + row), 0, ld_zbytes, 0x00020000);
      pz_[hh][iz] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane4 + 4 * (ld_col0 + 64 * (ZI0 + iz)), 0, 0));
    }
  };
  auto issue_loads = [&](int tile) __attribute__((always_inline)) {
    set_tile(tile, true);
#pragma unroll
    for (int hh = 0; hh < NR; ++hh)
#pragma unroll
      for (int i = 0; i < PER_ROW; ++i) issue_piece(hh * PER_ROW + i);
  };

  int tile = blockIdx.x;
  if (tile < total_tiles) issue_loads(tile);
  for (; tile < total_tiles; tile += gridDim.x) {
    __syncthreads();                                   // previous tile fully consumed
    // ld_* still describe the tile that sits in the registers
    // a tile inside its sequence needs no validity mask except for the tail of the last chunk (columns the loads got
    // as zeros would otherwise become relu(shift))
    const bool interior = VS > 0 && ld_col0 >= 0 && ld_col0 + (F + 2 * HALO) * V <= rs;     // wave-uniform
    bool hin[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int q = lane + 64 * i, gc = ld_col0 + q;
      hin[i] = interior ? q < (F + 2 * HALO) * V : (q < ld_hcols && gc >= 0 && gc < rs);
    }
#pragma unroll
    for (int hh = 0; hh < NR; ++hh) {
      const int c = wave + hh * (TW_THREADS / 64);
      const float sc = XF ? scale[c] : 1.f, sh = XF ? shift[c] : 0.f;
      bsum[hh] += (pd_[hh][0] + pd_[hh][1]) + (pd_[hh][2] + pd_[hh][3]);
      if constexpr (DZ) {
        const float mu = fin[c], is = fin[64 + c], kk = sc, a1 = m12[c], a2 = m12[64 + c];
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            dz + ld_off + (size_t)c * row_stride, 0, ld_zbytes, 0x00020000);
#pragma unroll
        for (int iz = 0; iz < NZ; ++iz) {
          const int i = ZI0 + iz;
          const float zv = ph_[hh][i];
          const float gq = fmaf(zv, sc, sh) > 0.f ? pz_[hh][iz] : 0.f;
          const float xh = (zv - mu) * is;
          const float o = kk * (gq - a1 - xh * a2);
          // columns of the halo frame in front (chunk ZI0 only) are another tile's; past the tile's last frame the
          // descriptor drops the store
          if (64 * i >= HALO * VS || lane + 64 * i >= HALO * VS) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), rsrc, lane4 + 4 * (ld_col0 + 64 * i), 0, 0);
            // (the same range test the descriptor applies to the store: what it drops is not part of dz)
            if (AMAX && (unsigned)(lane4 + 4 * (ld_col0 + 64 * i)) < (unsigned)ld_zbytes)
              am = max(am, __float_as_uint(o) & 0x7fffffffu);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NH; ++i) {
        const int q = lane + 64 * i;
        float v = ph_[hh][i];
        if (XF) v = fmaxf(fmaf(v, sc, sh), 0.f);
        if (!(VS > 0 && interior && 64 * (i + 1) <= (F + 2 * HALO) * VS)) v = hin[i] ? v : 0.f;
        if (q < row_h) hs[c * row_h + q] = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = lane + 64 * i;
        if (q < row_d) ds[c * row_d + q] = pd_[hh][i];
      }
    }
    __syncthreads();
    const bool more = tile + (int)gridDim.x < total_tiles;
    if constexpr (VS > 0) {
      set_tile(more ? tile + gridDim.x : tile, more);
    } else {
      if (more) issue_loads(tile + gridDim.x);
    }

    const float *drow = ds + (16 * TW_MT * mh + r) * row_d + g;  // + 16*m rows, + 4*s columns
    const float *hrow = hs + (16 * nt + r) * row_h + g;          // + p*V, + 4*s columns
    // operands of step s+1 are read while the MFMAs of step s run (the rows are zero-padded past the last step)
    float a[TW_MT], b[TAPS];
#pragma unroll
    for (int m = 0; m < TW_MT; ++m) a[m] = drow[16 * m * row_d];
    if constexpr (VS > 0) {
      constexpr int STEPS = (F * VS + 3) / 4;
      constexpr int PPS = (PIECES + STEPS - 1) / STEPS;          // pieces of the next tile per reduction step
#pragma unroll
      for (int p = 0; p < TAPS; ++p) b[p] = hrow[p * VS];
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        float na[TW_MT], nb[TAPS];
#pragma unroll
        for (int k = 0; k < PPS; ++k)
          if (PPS * s + k < PIECES) issue_piece(PPS * s + k);
#pragma unroll
        for (int m = 0; m < TW_MT; ++m) na[m] = drow[16 * m * row_d + 4 * s + 4];
#pragma unroll
        for (int p = 0; p < TAPS; ++p) nb[p] = hrow[p * VS + 4 * s + 4];
        __builtin_amdgcn_sched_barrier(0);   // keep the reads of step s+1 ahead of the MFMAs of step s
#pragma unroll
        for (int p = 0; p < TAPS; ++p)
#pragma unroll
          for (int m = 0; m < TW_MT; ++m)
            acc[p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[p], acc[p][m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < TW_MT; ++m) a[m] = na[m];
#pragma unroll
        for (int p = 0; p < TAPS; ++p) b[p] = nb[p];
      }
    } else {
      const int steps = (F * V + 3) / 4;
#pragma unroll
      for (int p = 0; p < TAPS; ++p) b[p] = hrow[p * V];
      for (int s = 0; s < steps; ++s) {
        float na[TW_MT], nb[TAPS];
#pragma unroll
        for (int m = 0; m < TW_MT; ++m) na[m] = drow[16 * m * row_d + 4 * s + 4];
#pragma unroll
        for (int p = 0; p < TAPS; ++p) nb[p] = hrow[p * V + 4 * s + 4];
#pragma unroll
        for (int p = 0; p < TAPS; ++p)
#pragma unroll
          for (int m = 0; m < TW_MT; ++m)
            acc[p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[p], acc[p][m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < TW_MT; ++m) a[m] = na[m];
#pragma unroll
        for (int p = 0; p < TAPS; ++p) b[p] = nb[p];
      }
    }
  }
  if constexpr (AMAX) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) am = max(am, (unsigned)__shfl_xor((int)am, off, 64));
    if (lane == 0 && am != 0) atomicMax(amax, am);
  }
  if (dbias_partial) {
#pragma unroll
    for (int hh = 0; hh < TC_C / (TW_THREADS / 64); ++hh) {
      float v = bsum[hh];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) dbias_partial[(size_t)blockIdx.x * TC_C + wave + hh * (TW_THREADS / 64)] = v;
    }
  }
  // partial[block][c][ci][p] (the layout of Conv2d.weight (c, ci, taps, 1)): D[row = 4g + q][col = r] -> c = 16*TW_MT*mh + 16*m + row, ci = 16*nt + r
  float *outp = dw_partial + (size_t)blockIdx.x * TAPS * TC_C * TC_C;
#pragma unroll
  for (int p = 0; p < TAPS; ++p)
#pragma unroll
    for (int m = 0; m < TW_MT; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        outp[((size_t)(16 * TW_MT * mh + 16 * m + 4 * g + q) * TC_C + 16 * nt + r) * TAPS + p] = acc[p][m][q];
}

}  // namespace

// x (N,64,T,V); W [taps][64][64] (plane p = temporal tap dt = p - (taps-1)/2, row = output channel);
// scale/shift [64] or NULL (input transform relu(x*scale+shift)); bias [64] or NULL.  taps = 3: the
// (3,1) temporal convolution; taps = 1: a pointwise 64->64 convolution over the same layout.
template <int TAPS>
static int tconv_forward_launch(int N, int T, int V, const float *x, const float *scale, const float *shift,
                                const float *W, const float *bias, float *out, float *stats_partial,
                                int *n_partials, void *stream) {
  constexpr int HALO = (TAPS - 1) / 2;
  int F = TC_NPO / V;
  if (F < 1) return P2R_EINVAL;
  if (F > T) F = T;
  const int tiles_per_seq = p2r_cdiv(T, F);
  int row_len = (F + 2 * HALO) * V;
  if (row_len % 2 == 0) ++row_len;               // odd stride: see stgcn_gcn.hip
  const size_t lds = (size_t)TC_C * row_len * sizeof(float) + (size_t)(TC_THREADS / 64) * 2 * TC_C * sizeof(float);
  if (lds > 160 * 1024 || row_len > 512) return P2R_EINVAL;
  if ((long long)T * V >= (1LL << 29)) return P2R_EINVAL;        // the kernel addresses a row with 32-bit byte offsets
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  {
    hipError_t e = p2r_allow_big_lds(tconv_fused_kernel<TAPS>, lds_ok);
    if (e != hipSuccess) return (int)e;
  }
  const long long tiles = (long long)N * tiles_per_seq;
  if (tiles > 0x7fffffffLL) return P2R_EINVAL;
  const int blocks = (int)(tiles < 256 ? tiles : 256);      // persistent: one workgroup per CU
  if (n_partials) *n_partials = blocks;
  if (!out) return P2R_OK;                                  // size query
  hipLaunchKernelGGL(tconv_fused_kernel<TAPS>, dim3(blocks), dim3(TC_THREADS), lds, p2r_stream(stream), N, T, V,
                     F, tiles_per_seq, row_len, x, scale, shift, W, bias, out, stats_partial);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_stgcn_tconv_forward(int N, int T, int V, int taps, const float *x, const float *scale,
                                       const float *shift, const float *W, const float *bias, float *out,
                                       float *stats_partial, int *n_partials, void *stream) {
  if (N < 0 || T <= 0 || V <= 0 || V > 128 || (taps != 1 && taps != 3)) return P2R_EINVAL;
  if (n_partials) *n_partials = 0;
  if (N == 0) return P2R_OK;
  return taps == 3 ? tconv_forward_launch<3>(N, T, V, x, scale, shift, W, bias, out, stats_partial, n_partials, stream)
                   : tconv_forward_launch<1>(N, T, V, x, scale, shift, W, bias, out, stats_partial, n_partials, stream);
}

template <int TAPS, int VS, int F = TW_F, bool DZ = false, bool AMAX = false>
static int tconv_dw_launch(int N, int T, int V, const float *x, const float *scale, const float *shift,
                           const float *dout, int n_blocks, float *dw_partial, float *dbias_partial,
                           void *stream, const float *dh = nullptr, const float *fin = nullptr, const float *m12 = nullptr,
                           float *dz = nullptr, unsigned *amax = nullptr) {
  constexpr int HALO = (TAPS - 1) / 2;
  const int row_d = tw_row(F * V), row_h = tw_row((F + 2 * HALO) * V);
  const size_t lds = (size_t)TC_C * (row_d + row_h) * sizeof(float);
  if (lds > 160 * 1024 || F * V > 256 || (F + 2 * HALO) * V > (TAPS == 1 ? 256 : 384)) return P2R_EINVAL;
  if ((long long)T * V >= (1LL << 29)) return P2R_EINVAL;        // the kernel addresses a row with 32-bit byte offsets
  static unsigned char lds_ok[2][P2R_MAX_DEVICES];
  if (scale || DZ) {     // with the input transform relu(x * scale + shift)
    hipError_t e = p2r_allow_big_lds(tconv_dw_kernel<TAPS, VS, F, DZ, true, AMAX>, lds_ok[0]);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((tconv_dw_kernel<TAPS, VS, F, DZ, true, AMAX>), dim3(n_blocks), dim3(TW_THREADS), lds,
                       p2r_stream(stream), N, T, V, row_d, row_h, x, scale, shift, dout, dw_partial, dbias_partial, dh, fin,
                       m12, dz, amax);
  } else {
    hipError_t e = p2r_allow_big_lds(tconv_dw_kernel<TAPS, VS, F, false, false>, lds_ok[1]);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((tconv_dw_kernel<TAPS, VS, F, false, false>), dim3(n_blocks), dim3(TW_THREADS), lds, p2r_stream(stream),
                       N, T, V, row_d, row_h, x, scale, shift, dout, dw_partial, dbias_partial, dh, fin, m12, dz);
  }
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// dw_partial [n_blocks][64 c][64 ci][taps] and (optional) dbias_partial [n_blocks][64] = row sums of dout,
// both summed over the leading axis by the caller.
extern "C" int p2r_stgcn_tconv_weight_grad(int N, int T, int V, int taps, const float *x, const float *scale,
                                           const float *shift, const float *dout, int n_blocks,
                                           float *dw_partial, float *dbias_partial, void *stream) {
  if (N < 0 || T <= 0 || V <= 0 || V > 64 || n_blocks < 1 || (taps != 1 && taps != 3)) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  // 20 columns per frame, one tap: the 64 -> 64 layers of the position-embedding MLP (stgcn.py:46-63, knn = 20).  A
  // 4-frame tile is 20 reduction steps there -- less work than the staging and the two barriers around it -- so this
  // shape gets 12-frame tiles (60 steps, 123 KB of LDS) and its own fully unrolled instance.
  if (V == 20 && taps == 1)
    return tconv_dw_launch<1, 20, 12>(N, T, V, x, scale, shift, dout, n_blocks, dw_partial, dbias_partial, stream);
  if (V == 53)   // the P2RNet skeleton: fully unrolled reduction loop
    return taps == 3
               ? tconv_dw_launch<3, 53>(N, T, V, x, scale, shift, dout, n_blocks, dw_partial, dbias_partial, stream)
               : tconv_dw_launch<1, 53>(N, T, V, x, scale, shift, dout, n_blocks, dw_partial, dbias_partial, stream);
  return taps == 3 ? tconv_dw_launch<3, 0>(N, T, V, x, scale, shift, dout, n_blocks, dw_partial, dbias_partial, stream)
                   : tconv_dw_launch<1, 0>(N, T, V, x, scale, shift, dout, n_blocks, dw_partial, dbias_partial, stream);
}

// The weight gradient of the (3,1) temporal convolution of the 53-joint skeleton with the BatchNorm-backward apply
// pass of its INPUT riding on the tile staging (stgcn_layers.py:399-412, the `tcn` Sequential: BatchNorm2d, ReLU,
// Conv2d): besides dw_partial / dbias_partial as above,
//   dz = scale * ((x * scale + shift > 0 ? dh : 0) - m1 - (x - mean) * invstd * m2)
// with fin [4][64] = mean, invstd, scale, shift (p2r_bn_finalize) and m12 [2][64] = m1, m2 (rows 2, 3 of
// p2r_bn_bwd_finalize; zeros in evaluation mode): p2r_bn_bwd_apply's mask form 2.  P2R_EINVAL for other shapes.
extern "C" int p2r_stgcn_tconv_weight_grad_dz(int N, int T, int V, int taps, const float *x, const float *fin,
                                              const float *dout, const float *dh, const float *m12, float *dz,
                                              int n_blocks, float *dw_partial, float *dbias_partial, void *stream) {
  if (N < 0 || T <= 0 || V != 53 || taps != 3 || n_blocks < 1 || !fin || !dh || !m12 || !dz) return P2R_EINVAL;
  if (N == 0) return P2R_OK;
  return tconv_dw_launch<3, 53, TW_F, true>(N, T, V, x, fin + 128, fin + 192, dout, n_blocks, dw_partial, dbias_partial,
                                            stream, dh, fin, m12, dz);
}

// split16 mode: the same launch, additionally leaving the float bits of max |dz| in *amax_bits (split16.h).
extern "C" int p2r_stgcn_tconv_weight_grad_dz_amax(int N, int T, int V, int taps, const float *x, const float *fin,
                                                   const float *dout, const float *dh, const float *m12, float *dz,
                                                   int n_blocks, float *dw_partial, float *dbias_partial,
                                                   unsigned *amax_bits, void *stream) {
  if (N < 0 || T <= 0 || V != 53 || taps != 3 || n_blocks < 1 || !fin || !dh || !m12 || !dz || !amax_bits) return P2R_EINVAL;
  hipError_t e = hipMemsetAsync(amax_bits, 0, sizeof(unsigned), p2r_stream(stream));
  if (e != hipSuccess) return (int)e;
  if (N == 0) return P2R_OK;
  return tconv_dw_launch<3, 53, TW_F, true, true>(N, T, V, x, fin + 128, fin + 192, dout, n_blocks, dw_partial,
                                                  dbias_partial, stream, dh, fin, m12, dz, amax_bits);
}
