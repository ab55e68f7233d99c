// nn_distance.hip -- bidirectional nearest-neighbour (chamfer) distance, gfx950.
//
// Replaces nn_distance (reference net_utils/nn_distance.py:34-61), which
// materialises two (B,N,M,C) `repeat`ed tensors, a (B,N,M,C) difference and a
// (B,N,M) distance matrix in HBM and then runs two torch.min reductions --
// seven kernels and ~5x the algorithmic bytes -- and, in the loss, is called in
// a Python loop over the batch (models/loss.py:127-131).
//
// MI355X design: one launch, one thread per output element (B*(N+M) threads):
// the thread keeps its own point in registers and scans the other cloud, whose
// rows are shared by every lane of the same batch row (L1/L2 broadcast reads).
// Nothing but the inputs and the four (B,N)/(B,M) outputs touches HBM.  The
// per-pair sum runs over C in ascending order with one rounding per op, and the
// scan keeps the first minimal index, so dist and idx are bit-identical to the
// reference's CPU result.  Backward is a second single launch, gather-form and
// deterministic (no atomics): each thread owns one output row of grad_pc1 or
// grad_pc2 and sums its contributions in a fixed order.
#include "p2r_common.h"

namespace {

template <int MODE>
__device__ __forceinline__ float nnd_term(float diff, float delta) {
  if (MODE == P2R_NND_L1SMOOTH) {  // huber_loss, nn_distance.py:27-32
    const float abs_error = fabsf(diff);
    const float quadratic = abs_error > delta ? delta : abs_error;
    const float linear = abs_error - quadratic;
    return 0.5f * (quadratic * quadratic) + delta * linear;
  } else if (MODE == P2R_NND_L1) {
    return fabsf(diff);
  } else {
    return diff * diff;
  }
}

template <int MODE>
__device__ __forceinline__ float nnd_dterm(float diff, float delta) {
  if (MODE == P2R_NND_L1SMOOTH) {
    const float sg = (float)((diff > 0.0f) - (diff < 0.0f));
    return fabsf(diff) <= delta ? diff : delta * sg;
  } else if (MODE == P2R_NND_L1) {
    return (float)((diff > 0.0f) - (diff < 0.0f));
  } else {
    return 2.0f * diff;
  }
}

template <int MODE, int CC>  // CC = compile-time C (3) or 0 for runtime C
__device__ __forceinline__ float nnd_pair(const float *a, const float *q, int C, float delta) {
  const int c_end = CC ? CC : C;
  float s = nnd_term<MODE>(a[0] - q[0], delta);
#pragma unroll
  for (int c = 1; c < c_end; ++c) s = s + nnd_term<MODE>(a[c] - q[c], delta);
  return s;
}

constexpr int ND_THREADS = 256;

template <int MODE, int CC>
__global__ __launch_bounds__(ND_THREADS) void nn_distance_kernel(
    int B, int N, int M, int C, float delta, const float *__restrict__ pc1,
    const float *__restrict__ pc2, float *__restrict__ dist1, int64_t *__restrict__ idx1,
    float *__restrict__ dist2, int64_t *__restrict__ idx2) {
  const long long total = (long long)B * (N + M);
  for (long long t = (long long)blockIdx.x * ND_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * ND_THREADS) {
    const int b = (int)(t / (N + M));
    const int r = (int)(t % (N + M));
    const bool first_dir = r < N;  // true: row of pc1, min over pc2
    const int self = first_dir ? r : r - N;
    const int n_other = first_dir ? M : N;
    const float *mine = (first_dir ? pc1 + ((size_t)b * N + self) * C : pc2 + ((size_t)b * M + self) * C);
    const float *other = first_dir ? pc2 + (size_t)b * M * C : pc1 + (size_t)b * N * C;
    float best = INFINITY;
    int besti = 0;
    for (int o = 0; o < n_other; ++o) {
      const float *q = other + (size_t)o * C;
      // pc_diff = pc1 - pc2 whichever side this thread reduces for (nn_distance.py:49)
      const float s = first_dir ? nnd_pair<MODE, CC>(mine, q, C, delta)
                                : nnd_pair<MODE, CC>(q, mine, C, delta);
      if (s < best || o == 0) { best = s; besti = o; }
    }
    if (first_dir) {
      if (dist1) dist1[(size_t)b * N + self] = best;
      if (idx1) idx1[(size_t)b * N + self] = besti;
    } else {
      if (dist2) dist2[(size_t)b * M + self] = best;
      if (idx2) idx2[(size_t)b * M + self] = besti;
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(ND_THREADS) void nn_distance_grad_kernel(
    int B, int N, int M, int C, float delta, const float *__restrict__ pc1,
    const float *__restrict__ pc2, const int64_t *__restrict__ idx1,
    const int64_t *__restrict__ idx2, const float *__restrict__ g1, const float *__restrict__ g2,
    float *__restrict__ grad_pc1, float *__restrict__ grad_pc2) {
  // one thread per (b, row, c)
  const long long total = (long long)B * (N + M) * C;
  for (long long t = (long long)blockIdx.x * ND_THREADS + threadIdx.x; t < total;
       t += (long long)gridDim.x * ND_THREADS) {
    const int c = (int)(t % C);
    const long long br = t / C;
    const int b = (int)(br / (N + M));
    const int r = (int)(br % (N + M));
    const float *a = pc1 + (size_t)b * N * C;
    const float *q = pc2 + (size_t)b * M * C;
    const int64_t *i1 = idx1 + (size_t)b * N;
    const int64_t *i2 = idx2 + (size_t)b * M;
    const float *gg1 = g1 ? g1 + (size_t)b * N : nullptr;
    const float *gg2 = g2 ? g2 + (size_t)b * M : nullptr;
    if (r < N) {
      const int i = r;
      float acc = 0.f;
      if (gg1) {  // dist1[i] pairs (i, idx1[i])
        const int j = (int)i1[i];
        acc = acc + gg1[i] * nnd_dterm<MODE>(a[i * C + c] - q[j * C + c], delta);
      }
      if (gg2) {  // every dist2[j] whose arg-min is i, ascending j
        for (int j = 0; j < M; ++j)
          if ((int)i2[j] == i) acc = acc + gg2[j] * nnd_dterm<MODE>(a[i * C + c] - q[j * C + c], delta);
      }
      grad_pc1[((size_t)b * N + i) * C + c] = acc;
    } else {
      const int j = r - N;
      float acc = 0.f;
      if (gg1) {
        for (int i = 0; i < N; ++i)
          if ((int)i1[i] == j) acc = acc - gg1[i] * nnd_dterm<MODE>(a[i * C + c] - q[j * C + c], delta);
      }
      if (gg2) {
        const int i = (int)i2[j];
        acc = acc - gg2[j] * nnd_dterm<MODE>(a[i * C + c] - q[j * C + c], delta);
      }
      grad_pc2[((size_t)b * M + j) * C + c] = acc;
    }
  }
}

int grid_for(long long total) {
  long long blocks = (total + ND_THREADS - 1) / ND_THREADS;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

extern "C" int p2r_nn_distance(int B, int N, int M, int C, const float *pc1, const float *pc2,
                               int mode, float delta, float *dist1, int64_t *idx1, float *dist2,
                               int64_t *idx2, void *stream) {
  if (B < 0 || N < 0 || M < 0 || C <= 0 || mode < 0 || mode > 2) return P2R_EINVAL;
  // torch.min over an empty dimension is an error in the reference
  // (nn_distance.py:57-58); report it instead of producing garbage.
  if (B > 0 && (N == 0 || M == 0)) return P2R_EINVAL;
  if (B == 0) return P2R_OK;
  hipStream_t st = p2r_stream(stream);
  const int grid = grid_for((long long)B * (N + M));
#define P2R_ND_LAUNCH(MODE)                                                                        \
  do {                                                                                             \
    if (C == 3)                                                                                    \
      hipLaunchKernelGGL((nn_distance_kernel<MODE, 3>), dim3(grid), dim3(ND_THREADS), 0, st, B, N, \
                         M, C, delta, pc1, pc2, dist1, idx1, dist2, idx2);                         \
    else                                                                                           \
      hipLaunchKernelGGL((nn_distance_kernel<MODE, 0>), dim3(grid), dim3(ND_THREADS), 0, st, B, N, \
                         M, C, delta, pc1, pc2, dist1, idx1, dist2, idx2);                         \
  } while (0)
  if (mode == P2R_NND_L1SMOOTH) P2R_ND_LAUNCH(P2R_NND_L1SMOOTH);
  else if (mode == P2R_NND_L1) P2R_ND_LAUNCH(P2R_NND_L1);
  else P2R_ND_LAUNCH(P2R_NND_L2);
#undef P2R_ND_LAUNCH
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_nn_distance_grad(int B, int N, int M, int C, const float *pc1, const float *pc2,
                                    int mode, float delta, const int64_t *idx1,
                                    const int64_t *idx2, const float *g1, const float *g2,
                                    float *grad_pc1, float *grad_pc2, void *stream) {
  if (B < 0 || N < 0 || M < 0 || C <= 0 || mode < 0 || mode > 2) return P2R_EINVAL;
  if (B == 0 || (N + M) == 0) return P2R_OK;
  hipStream_t st = p2r_stream(stream);
  const int grid = grid_for((long long)B * (N + M) * C);
  if (mode == P2R_NND_L1SMOOTH)
    hipLaunchKernelGGL(nn_distance_grad_kernel<P2R_NND_L1SMOOTH>, dim3(grid), dim3(ND_THREADS), 0, st,
                       B, N, M, C, delta, pc1, pc2, idx1, idx2, g1, g2, grad_pc1, grad_pc2);
  else if (mode == P2R_NND_L1)
    hipLaunchKernelGGL(nn_distance_grad_kernel<P2R_NND_L1>, dim3(grid), dim3(ND_THREADS), 0, st, B, N,
                       M, C, delta, pc1, pc2, idx1, idx2, g1, g2, grad_pc1, grad_pc2);
  else
    hipLaunchKernelGGL(nn_distance_grad_kernel<P2R_NND_L2>, dim3(grid), dim3(ND_THREADS), 0, st, B, N,
                       M, C, delta, pc1, pc2, idx1, idx2, g1, g2, grad_pc1, grad_pc2);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
