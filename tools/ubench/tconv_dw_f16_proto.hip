// PROTOTYPE, not part of the product: the WEIGHT gradient of the temporal (3,1) convolution
//   dW[tap][co][ci] = sum_{n,t,j} dout[n,co,t,j] * h[n,ci,t+tap-1,j],   h = relu(scale[ci] * z + shift[ci])  (0 outside 0 <= t < T)
// on two-part fp16 MFMA products (DESIGN.md section 5 "Round 5").  Both operands are runtime tensors, the reduction runs
// over the columns: K of v_mfma_f32_16x16x32_f16 = 32 joints of ONE frame (two k-steps per frame: joints 0-31 and 32-63, the
// 11 joints past 52 zeroed in the A operand -- 53/64 of the MFMA work is useful, the kernel is HBM-bound anyway), so a tap is
// a whole-frame shift of the B operand and no tile ever straddles a frame.  Lane (kg, r) of an operand holds 8 consecutive
// floats of one channel row, loaded straight from the tensors (4-byte aligned 16-byte loads, one frame ahead), transformed,
// scaled (the gradient: x 2^S) and split into x = x1 + x2.  Wave w of a 256-thread workgroup owns the 16 output rows
// co = 16 w .. 16 w + 15: 3 taps x 4 column tiles = 12 accumulator tiles (x 2: large and small products); the split gradient frames t-1, t, t+1 stay in its
// registers (a frame of h meets three frames of the gradient); the frame of h is split ONCE per workgroup -- wave w
// builds column tile w -- and passed around as fp16 operand registers through LDS (16 KB per frame, two buffers, one
// barrier per frame); three products per tile and k-step (a1 b2, a2 b1, a1 b1).
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };

constexpr int V = 53, C = 64;

struct Split { h8 p, q; };
// x (8 floats) -> fp16 parts; the residual through the compiler's own convert / subtract (the MFMAs are builtins here:
// inline assembly next to them is not covered by the hazard recogniser, see gcn3h_proto.hip)
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
__global__ __launch_bounds__(256, 2) void tconv_dw_f16_kernel(int T, const float *__restrict__ z, const float *__restrict__ scale,
                                                              const float *__restrict__ shift, const float *__restrict__ dout,
                                                              float gscale, float *__restrict__ part) {
  // the split frame of h, once per workgroup: [buffer][part][nt][ks][lane] (16 bytes each) -- wave w builds column tile w
  __shared__ h8 bl[2][2][4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kg = lane >> 4, r = lane & 15;
  const int chunks = T / FC;
  const int n = blockIdx.x / chunks, t0 = (blockIdx.x % chunks) * FC;
  const size_t rowlen = (size_t)T * V;
  const float *zb = z + (size_t)n * C * rowlen, *db = dout + (size_t)n * C * rowlen;
  // A operand: rows co = 16 wave + r; this wave's share of the B operand: rows ci = 16 wave + r.  k-step ks: joints 32 ks + 8 kg + i
  const float *arow = db + (size_t)(16 * wave + r) * rowlen + 8 * kg;
  const float *brow = zb + (size_t)(16 * wave + r) * rowlen + 8 * kg;
  const float bsc = scale ? scale[16 * wave + r] : 1.f, bsh = shift ? shift[16 * wave + r] : 0.f;
  // joints >= 53 of k-step 1: zero in the A operand (kg 2: joints 48..55 -> i < 5; kg 3: none)
  const int nvalid1 = kg < 2 ? 8 : (kg == 2 ? 5 : 0);

  f32x4 acc[3][4];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[p][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 lo[3][4];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) lo[p][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  Split A[3][2];                                 // gradient frames s - 1, s, s + 1 (rolling), two k-steps
  float ra[2][8], rb[2][8];                      // raw frames in flight: gradient frame s + 2, h frame s + 1
  auto fetch_a = [&](int t) {
    const bool in = t >= t0 && t < t0 + FC;      // this workgroup's frames only: the neighbours' gradient is theirs
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (in) load8(arow + (size_t)t * V + 32 * ks, ra[ks]);
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) ra[ks][i] = 0.f;
      }
    }
  };
  auto split_a = [&](Split (&dst)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (ks == 0 || i < nvalid1) ? ra[ks][i] * gscale : 0.f;
      dst[ks] = split8(v);
    }
  };
  auto fetch_b = [&](int t) {
    const bool in = t >= 0 && t < T;             // zero padding of the sequence
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (in) load8(brow + (size_t)t * V + 32 * ks, rb[ks]);
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) rb[ks][i] = 0.f;
      }
    }
    return in;
  };
  auto split_b = [&](int buf, bool in) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = in ? fmaxf(fmaf(rb[ks][i], bsc, bsh), 0.f) : 0.f;
      const Split b = split8(v);
      bl[buf][0][wave][ks][lane] = b.p;
      bl[buf][1][wave][ks][lane] = b.q;
    }
  };
  // prologue: A = grad(t0 - 2) = 0, grad(t0 - 1) = 0, grad(t0); h(t0 - 1) in buffer 0; in flight: grad(t0 + 1), h(t0)
  fetch_a(t0 - 2); split_a(A[0]);
  split_a(A[1]);
  fetch_a(t0); split_a(A[2]);
  bool bin = fetch_b(t0 - 1); split_b(0, bin);
  fetch_a(t0 + 1);
  bin = fetch_b(t0);
  __syncthreads();
  // frame s of h, s = t0 - 1 .. t0 + FC: tap 0 meets the gradient of s + 1, tap 1 of s, tap 2 of s - 1
  for (int s = t0 - 1, it = 0; s <= t0 + FC; ++s, ++it) {
    const int cur = it & 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const h8 bp = bl[cur][0][nt][ks][lane], bq = bl[cur][1][nt][ks][lane];
#pragma unroll
        for (int p = 0; p < 3; ++p) {            // tap p: gradient frame s - p + 1 = A[2 - p]
          const Split &a = A[2 - p][ks];
          // the two small products in their own accumulator: 384 additions of 2^-11-sized terms into the large sum cost
          // accuracy (8.2e-7 -> 5.5e-7 of range vs float64); a fourth term a2 b2 changes nothing (measured)
          lo[p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.p, bq, lo[p][nt], 0, 0, 0);
          lo[p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.q, bp, lo[p][nt], 0, 0, 0);
          acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.p, bp, acc[p][nt], 0, 0, 0);
        }
      }
    // the next frame: h(s + 1) -> the other buffer, gradient window one frame on; then the loads of the frame after
    split_b(cur ^ 1, bin);
    A[0][0] = A[1][0]; A[0][1] = A[1][1]; A[1][0] = A[2][0]; A[1][1] = A[2][1];
    split_a(A[2]);
    fetch_a(s + 3);
    bin = fetch_b(s + 2);
    __syncthreads();
  }
  // partial dW of this workgroup: [co][ci][tap] (the Conv2d weight's layout), co = 16 wave + 4 kg + q, ci = 16 nt + r
  float *o = part + (size_t)blockIdx.x * C * C * 3;
  const float inv = 1.f / gscale;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[p][nt][q] += lo[p][nt][q];
        o[((size_t)(16 * wave + 4 * kg + q) * C + 16 * nt + r) * 3 + p] = acc[p][nt][q] * inv;
      }
}
}  // namespace

// z, dout (N,64,T,53) f32 with at least 64 readable floats behind the last element; scale / shift [64] or NULL (no
// transform); gscale = the power of two that lifts the gradient into fp16's range; part [N * T / 64][64][64][3].
// T % 64 == 0.  Returns the number of partials.
extern "C" int proto_tconv_dw_f16(int N, int T, const float *z, const float *scale, const float *shift, const float *dout,
                                  float gscale, float *part, void *stream) {
  constexpr int FC = 64;
  if (N <= 0 || T <= 0 || T % FC != 0) return -1;
  const int blocks = N * (T / FC);
  if (part)
    hipLaunchKernelGGL(tconv_dw_f16_kernel<FC>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, T, z, scale, shift, dout, gscale, part);
  return hipGetLastError() == hipSuccess ? blocks : -2;
}
