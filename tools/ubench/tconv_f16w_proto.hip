// PROTOTYPE, not part of the product: the temporal (3,1) convolution (forward with the BatchNorm + ReLU input transform,
// or plain = the data gradient's form) on two-part fp16 MFMA products, second design ("walk"): a 256-thread workgroup
// WALKS along the frames of one sample instead of staging (F + 2)-frame tiles -- every input frame is loaded, transformed
// and split exactly once (tconv_f16_proto.hip: 4-frame tiles + 2 halo frames = 1.5x the reads), into a ring of four
// frames in LDS kept as fp16 operand slots [part][k-step][channel group][column][8 channels] (16 bytes, what lane (kg, r)
// of the B operand reads for column r; 16 KB per frame).  Loads come straight from the tensor, two frames ahead: lane l
// of wave w holds column l of channel groups w and w + 4 (16 coalesced 4-byte loads), so the split result is a slot and
// goes out as one 16-byte LDS write (the first version loaded 8 joints of one channel per lane and scattered halves with
// ds_write_b16: eight-way bank conflicts, LDS-bound at 0.29 ms).  Wave w owns output channels 16 w .. 16 w + 15: W of the three taps split once
// into 48 registers; per output frame 4 column tiles (64 columns, 53 valid) x 2 k-steps x 3 taps x 3 products.  The tile
// leaves from the registers: 16 consecutive columns of a row per quarter wave.
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };

#ifndef RING_FRAMES
#define RING_FRAMES 4
#endif
constexpr int V = 53, C = 64, RING = RING_FRAMES;     // 3: 48 KB, three workgroups per CU, but a second barrier per frame

struct Split { h8 p, q; };
__device__ __forceinline__ Split split8(const float (&v)[8]) {
  Split s;
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    f2 x = {v[i], v[i + 1]};
    asm volatile("" : "+v"(x));                            // the split sees VALUES (tools/ubench/split_probe.hip)
    const h2 p = __builtin_convertvector(x, h2);
    const h2 q = __builtin_convertvector(x - __builtin_convertvector(p, f2), h2);
    s.p[i] = p.x; s.p[i + 1] = p.y; s.q[i] = q.x; s.q[i + 1] = q.y;
  }
  return s;
}
__device__ __forceinline__ void load8(const float *p, float (&v)[8]) {
  const F4 a = *reinterpret_cast<const F4 *>(p), b = *reinterpret_cast<const F4 *>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

template <int FC>
__global__ __launch_bounds__(256, RING == 3 ? 3 : 2) void tconv_f16w_kernel(int T, const float *__restrict__ x, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, const float *__restrict__ W,
                                                            const float *__restrict__ bias, float wscale, float *__restrict__ out) {
  // ring[frame & 3][part][ks][kg][column 0..63][8 channels]: halves
  __shared__ _Float16 ring[RING][2][2][4][64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kg = lane >> 4, r = lane & 15;
  const int chunks = T / FC;
  const int n = blockIdx.x / chunks, t0 = (blockIdx.x % chunks) * FC;
  const size_t rowlen = (size_t)T * V;
  const float *xb = x + (size_t)n * C * rowlen;
  float *ob = out + (size_t)n * C * rowlen;
  const bool xform = scale != nullptr;

  // A operands: W[tap][co = 16 wave + r][ci = 32 ks + 8 kg + i], scaled by 2^S and split once
  Split A[3][2];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
      load8(W + ((size_t)p * C + 16 * wave + r) * C + 32 * ks + 8 * kg, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= wscale;
      A[p][ks] = split8(v);
    }
  float bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = bias ? bias[16 * wave + 4 * kg + q] : 0.f;
  const float inv = 1.f / wscale;

  // this lane's share of an input frame: column `lane` (joints 53..63: whatever follows in memory, never used), channel
  // groups wave and wave + 4 (8 channels each): 16 coalesced 4-byte loads per lane and frame, and what comes out of the
  // split IS an operand slot -- one 16-byte LDS write per part and group, no transposition
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const float *lcol = xb + lane;
  float sc[2][8], sh[2][8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = 8 * (wv + 4 * j) + e;
      sc[j][e] = xform ? scale[ch] : 1.f; sh[j][e] = xform ? shift[ch] : 0.f;
    }
  float rawA[2][8];                             // one frame in flight (two, in a loop unrolled twice: no faster)
  auto fetch = [&](int t, float (&raw)[2][8]) {
    const bool in = t >= 0 && t < T;
    if (in) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[j][e] = lcol[(size_t)(8 * (wv + 4 * j) + e) * rowlen + (size_t)t * V];
    }
    return in;
  };
  // slot of (group g, column c): [ks' = g / 4][kg' = g % 4][c][8]; group wave + 4 j: ks' = j, kg' = wave
  _Float16 *wslot = &ring[0][0][0][wv][lane][0];
  auto put = [&](int t, bool in, const float (&raw)[2][8]) {
    _Float16 *d = wslot + (size_t)((t + RING) % RING) * (2 * 2 * 4 * 64 * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = in ? (xform ? fmaxf(fmaf(raw[j][e], sc[j][e], sh[j][e]), 0.f) : raw[j][e]) : 0.f;
      const Split sp = split8(v);
      *reinterpret_cast<h8 *>(d + j * (4 * 64 * 8)) = sp.p;
      *reinterpret_cast<h8 *>(d + (2 * 4 * 64 * 8) + j * (4 * 64 * 8)) = sp.q;
    }
  };
  // prologue: frames t0 - 1 and t0 into the ring, frame t0 + 1 in flight
  bool inA = fetch(t0 - 1, rawA); put(t0 - 1, inA, rawA);
  inA = fetch(t0, rawA); put(t0, inA, rawA);
  inA = fetch(t0 + 1, rawA);
  auto frame = [&](int s) {
    __syncthreads();                 // frame s + 1 complete; everybody is done with frame s - 2's slot (= s + 2's)
    f32x4 hi[4], lo[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { hi[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const _Float16 *fr = &ring[(s + p - 1 + RING) % RING][0][0][0][0][0];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const h8 bp = *reinterpret_cast<const h8 *>(fr + (((0 * 2 + ks) * 4 + kg) * 64 + 16 * nt + r) * 8);
          const h8 bqv = *reinterpret_cast<const h8 *>(fr + (((1 * 2 + ks) * 4 + kg) * 64 + 16 * nt + r) * 8);
#ifdef ABL_NO_MFMA                                   // timing ablation
          asm volatile("" : "+v"(hi[nt]), "+v"(lo[nt]) : "v"(bp), "v"(bqv));
          continue;
#endif
          lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[p][ks].p, bqv, lo[nt], 0, 0, 0);
          lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[p][ks].q, bp, lo[nt], 0, 0, 0);
          hi[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[p][ks].p, bp, hi[nt], 0, 0, 0);
        }
    }
    // D[co = 16 wave + 4 kg + q][column 16 nt + r]
    float *orow = ob + (size_t)(16 * wave + 4 * kg) * rowlen + (size_t)s * V + r;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#ifdef ABL_NO_STORE                                    // timing ablation
      if (hi[nt][0] == 123.456f)
#else
      if (16 * nt + r < V)
#endif
      {
#pragma unroll
        for (int q = 0; q < 4; ++q) orow[(size_t)q * rowlen + 16 * nt] = (hi[nt][q] + lo[nt][q]) * inv + bq[q];
      }
  };
  for (int s = t0; s < t0 + FC; ++s) {
    if (RING == 3) __syncthreads();   // frame s - 2's slot is frame s + 1's: everybody has to be done reading it
    put(s + 1, inA, rawA);
    inA = fetch(s + 2, rawA);
    frame(s);
  }
}
}  // namespace

// The entry of tools/dev_tconv_bf16.py (VARIANT=f16w): x, out (N,64,T,53) f32, x with at least 16 readable floats behind its
// last element; scale / shift [64] or NULL (plain); W [3][64][64] ([tap][co][ci]) f32; bias [64] or NULL.  T % 64 == 0.
extern "C" int proto_tconv3h_forward(int N, int T, const float *x, const float *scale, const float *shift, const float *W,
                                     const float *bias, float *out, void *stream) {
  constexpr int FC = 64;
  if (N <= 0 || T <= 0 || T % FC != 0) return 1;
  hipLaunchKernelGGL(tconv_f16w_kernel<FC>, dim3(N * (T / FC)), dim3(256), 0, (hipStream_t)stream, T, x, scale, shift, W, bias,
                     256.f, out);
  return (int)hipGetLastError();
}
