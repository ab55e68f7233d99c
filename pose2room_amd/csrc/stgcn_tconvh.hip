// stgcn_tconvh.hip -- the temporal (3,1) convolution of st_gcn_block in `split16` arithmetic (opt-in; the default build of
// the step runs stgcn_tconv3.hip), gfx950.
//
// Same operator as p2r_stgcn_tconv3_forward (reference models/p2rnet/modules/stgcn_layers.py:399-411 and its data
// gradient):
//     out[n,c,t,w] = bias[c] + sum_{p<3} sum_ci W[p][c][ci] * h[n,ci,t+p-1,w],  h = relu(x*scale+shift) or x
// with every fp32 product on three v_mfma_f32_16x16x32_f16 of two-part fp16 operands (split16.h).
//
// Design ("walk"; prototyped in round 5 as tools/ubench/tconv_f16w_proto.hip): a 256-thread workgroup walks along the
// frames of one sample.  Every input frame is loaded, transformed and split exactly ONCE (no halo re-reads) into a ring
// of four frames in LDS kept as fp16 operand slots [part][k-step][channel group][column][8 channels] -- 16 bytes, what
// lane (kg, r) of the B operand reads for column r.  Loads come straight from the tensor one frame ahead: lane l of
// wave w holds column l of channel groups w and w + 4 (16 coalesced 4-byte loads; lanes 53..63 idle), so the split
// result IS a slot and leaves as one 16-byte LDS write.  Wave w owns output channels 16 w .. 16 w + 15: the three taps'
// weights arrive pre-split from the host (prepare_chain) and stay in 48 registers; per output frame 4 column tiles (64
// columns, 53 real) x 2 k-steps x 3 taps x 3 products = 72 MFMAs.  The tile leaves from the registers (16 consecutive
// columns of a row per quarter wave).  Compiler-scheduled builtins only: no MFMA in inline assembly.
//
// The epilogues of the exact kernel are carried: forward = (count, mean, M2) of the stored values per workgroup and
// channel for the BatchNorm that follows (sums about a pivot; a wave owns whole channels, so nothing is merged across
// waves); data gradient = the two sums of the BatchNorm + ReLU backward of the layer in front.
#include "p2r_common.h"
#include "split16.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH_V = 53, TH_C = 64, TH_RING = 4;
constexpr int TH_FRAME = 2 * 2 * 4 * 64 * 8;          // halves per ring frame: [part][ks][kg][column][8]

template <bool XFORM, bool BWD>
__global__ __launch_bounds__(256, 2) void tconvh_kernel(
    int T, int FC, const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
    const p2r_h8 *__restrict__ Wh, const float *__restrict__ winv, const float *__restrict__ bias,
    float *__restrict__ out, float *__restrict__ stats_partial, const float *__restrict__ bwd_z,
    const float *__restrict__ bwd_fin, const unsigned *__restrict__ x_amax) {
  constexpr int V = TH_V, C = TH_C, RING = TH_RING;
  __shared__ _Float16 ring[RING * TH_FRAME];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kg = lane >> 4, r = lane & 15;
  const int chunks = T / FC;
  const int n = blockIdx.x / chunks, t0 = (blockIdx.x % chunks) * FC;
  const size_t rowlen = (size_t)T * V;
  const float *xb = x + (size_t)n * C * rowlen;
  float *ob = out + (size_t)n * C * rowlen;

  // operand scale of x (the data gradient's incoming gradient; forward activations are used as they are) and the
  // inverse of the accumulators' scale: 2^-(S_w + S_x)
  float xs, xinv;
  p2r_split_scale(x_amax, xs, xinv);
  const float inv = xinv * winv[0];

  // A operands: 2^S_w W[tap][co = 16 wave + r][ci = 32 ks + 8 kg + i], split on the host: Wh[part][tap][ks][wave][lane],
  // parts (w1, w2, w1' = 2^-11 w1: the partner of the operand's scaled residual, split16.h)
  p2r_h8 A1[3][2], A2[3][2], A1s[3][2];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      A1[p][ks] = Wh[(((0 * 3 + p) * 2 + ks) * 4 + wv) * 64 + lane];
      A2[p][ks] = Wh[(((1 * 3 + p) * 2 + ks) * 4 + wv) * 64 + lane];
      A1s[p][ks] = Wh[(((2 * 3 + p) * 2 + ks) * 4 + wv) * 64 + lane];
    }
  float bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = bias ? bias[16 * wv + 4 * kg + q] : 0.f;

  // this lane's share of an input frame: column `lane`, channel groups wave and wave + 4 (8 channels each)
  const bool col_ok = lane < V;
  const float *lcol = xb + lane;
  float sc[2][8], sh[2][8];
  if (XFORM) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = 8 * (wv + 4 * j) + e;
        sc[j][e] = scale[ch]; sh[j][e] = shift[ch];
      }
  }
  float raw[2][8];
  auto fetch = [&](int t) {
    const bool in = t >= 0 && t < T && col_ok;
    if (in) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[j][e] = lcol[(size_t)(8 * (wv + 4 * j) + e) * rowlen + (size_t)t * V];
    }
    return in;
  };
  // slot of (channel group g, column c) inside a frame: [ks = g / 4][kg = g % 4][c][8]; group wave + 4 j: ks = j, kg = wave
  _Float16 *wslot = ring + ((size_t)wv * 64 + lane) * 8;
  auto put = [&](int t, bool in) {
    _Float16 *d = wslot + (size_t)((t + RING) % RING) * TH_FRAME;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (XFORM)      // BatchNorm affine + ReLU, clamped to fp16's range (the zero padding of the sequence is 0, not relu(shift))
          v[e] = in ? __builtin_amdgcn_fmed3f(fmaf(raw[j][e], sc[j][e], sh[j][e]), 0.f, P2R_H16_MAX) : 0.f;
        else
          v[e] = in ? raw[j][e] * xs : 0.f;
      }
      const P2RSplit8 sp = p2r_split8(v);
      *reinterpret_cast<p2r_h8 *>(d + j * (4 * 64 * 8)) = sp.p;
      *reinterpret_cast<p2r_h8 *>(d + (2 * 4 * 64 * 8) + j * (4 * 64 * 8)) = sp.q;
    }
  };

  // per-lane state of the epilogues.  Forward statistics: sums about a pivot per channel (the mean of the first
  // frame's first 16 columns), over this lane's columns and all frames of the chunk.  Data gradient: sum g, sum g (z - mean).
  const bool want_stats = stats_partial != nullptr;
  float pivot[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float bsc[4], bsh[4], bmu[4];
  if (BWD) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 16 * wv + 4 * kg + q;
      bmu[q] = bwd_fin[c]; bsc[q] = bwd_fin[128 + c]; bsh[q] = bwd_fin[192 + c];
    }
  }
  const float *zb = BWD ? bwd_z + (size_t)n * C * rowlen + (size_t)(16 * wv + 4 * kg) * rowlen + r : nullptr;
  const bool last_ok = r < V - 48;                         // column tile 3 holds columns 48..63: 48..52 are real

  // prologue: frames t0 - 1 and t0 into the ring, frame t0 + 1 in flight
  bool in = fetch(t0 - 1); put(t0 - 1, in);
  in = fetch(t0); put(t0, in);
  in = fetch(t0 + 1);
  for (int s = t0; s < t0 + FC; ++s) {
    put(s + 1, in);
    in = fetch(s + 2);
    float zv[4][4];
    if (BWD) {   // the saved activation at this frame's output positions: requested in front of the MFMAs
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          zv[nt][q] = (nt < 3 || last_ok) ? zb[(size_t)q * rowlen + (size_t)s * V + 16 * nt] : 0.f;
    }
    __syncthreads();                 // frame s + 1 complete; everybody is done with frame s - 2's slot (= s + 2's)
    f32x4 hi[4], lo[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { hi[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const _Float16 *fr = ring + (size_t)((s + p - 1 + RING) % RING) * TH_FRAME;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const p2r_h8 b1 = *reinterpret_cast<const p2r_h8 *>(fr + (((0 * 2 + ks) * 4 + kg) * 64 + 16 * nt + r) * 8);
          const p2r_h8 b2 = *reinterpret_cast<const p2r_h8 *>(fr + (((1 * 2 + ks) * 4 + kg) * 64 + 16 * nt + r) * 8);
          // the two small products in their own accumulator (added to the large one once, at the end)
          lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1s[p][ks], b2, lo[nt], 0, 0, 0);
          lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[p][ks], b1, lo[nt], 0, 0, 0);
          hi[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[p][ks], b1, hi[nt], 0, 0, 0);
        }
    }
    // D[co = 16 wave + 4 kg + q][column 16 nt + r]
    float *orow = ob + (size_t)(16 * wv + 4 * kg) * rowlen + (size_t)s * V + r;
    float val[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) val[nt][q] = fmaf(hi[nt][q] + lo[nt][q], inv, bq[q]);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      if (nt < 3 || last_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) orow[(size_t)q * rowlen + 16 * nt] = val[nt][q];
      }
    if (BWD) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        if (nt < 3 || last_ok) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float g = fmaf(zv[nt][q], bsc[q], bsh[q]) > 0.f ? val[nt][q] : 0.f;
            s1[q] += g;
            s2[q] = fmaf(g, zv[nt][q] - bmu[q], s2[q]);
          }
        }
    } else if (want_stats) {
      if (s == t0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pivot[q] = p2r_row16_sum(val[0][q]) * 0.0625f;
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        if (nt < 3 || last_ok) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float d = val[nt][q] - pivot[q];
            s1[q] += d;
            s2[q] = fmaf(d, d, s2[q]);
          }
        }
    }
  }
  if (want_stats) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float a = p2r_row16_sum(s1[q]), b = p2r_row16_sum(s2[q]);
      const int c = 16 * wv + 4 * kg + q;
      if (r == 0) {
        if (BWD) {        // [64][2]: (sum g, sum g * xhat)
          float *o = stats_partial + ((size_t)blockIdx.x * C + c) * 2;
          o[0] = a; o[1] = b * bwd_fin[64 + c];
        } else {          // [64][3]: (count, mean, M2)
          const float cnt = (float)FC * (float)V, d = a / cnt;
          float *o = stats_partial + ((size_t)blockIdx.x * C + c) * 3;
          o[0] = cnt; o[1] = pivot[q] + d; o[2] = fmaxf(b - a * d, 0.f);
        }
      }
    }
  }
}

}  // namespace

// frames per workgroup: the longest of 64 / 32 / 16 that divides T
static inline int tconvh_chunk(int T) { return T % 64 == 0 ? 64 : (T % 32 == 0 ? 32 : 16); }

extern "C" int p2r_stgcn_tconvh_forward(int N, int T, int V, const float *x, const float *scale, const float *shift,
                                        const void *Wh, const float *winv, const float *bias, float *out,
                                        float *stats_partial, int *n_partials, const float *bwd_z, const float *bwd_fin,
                                        const unsigned *x_amax, void *stream) {
  if (N < 0 || T <= 0 || V != TH_V || (scale == nullptr) != (shift == nullptr)) return P2R_EINVAL;
  if ((bwd_z == nullptr) != (bwd_fin == nullptr) || (bwd_z && (scale || !stats_partial))) return P2R_EINVAL;
  if (T % 16 != 0 || T > (1 << 20)) return P2R_EINVAL;
  if (n_partials) *n_partials = 0;
  if (N == 0) return P2R_OK;
  const int FC = tconvh_chunk(T);
  const long long blocks = (long long)N * (T / FC);
  if (blocks > 0x7fffffffLL) return P2R_EINVAL;
  if (n_partials) *n_partials = (int)blocks;
  if (!out) return P2R_OK;
  if (!Wh || !winv || ((uintptr_t)Wh % 16) != 0) return P2R_EINVAL;
  const p2r_h8 *wh = reinterpret_cast<const p2r_h8 *>(Wh);
  hipStream_t st = p2r_stream(stream);
#define P2R_TH(XF, BW) hipLaunchKernelGGL((tconvh_kernel<XF, BW>), dim3((unsigned)blocks), dim3(256), 0, st, T, FC, x, scale, \
                                          shift, wh, winv, bias, out, stats_partial, bwd_z, bwd_fin, x_amax)
  if (bwd_z) P2R_TH(false, true);
  else if (scale) P2R_TH(true, false);
  else P2R_TH(false, false);
#undef P2R_TH
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
