// fps.hip -- furthest point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel (reference
// _ext-src/src/sampling_gpu.cu:69-229).  The reference runs one 512-thread block
// per cloud with a 9-level shared-memory tree and 9 __syncthreads per round.
//
// MI355X design: the cloud lives in VGPRs for the whole kernel (PPT points per
// lane: xyz + running min-distance), a round is PPT distance updates per lane,
// one 64-bit wave arg-max (value bits in the high word, tie-priority in the low
// word) and, only when a cloud spans several waves, one LDS slot exchange with
// a single barrier.  No global traffic inside the round loop except the 12-byte
// read of the newly picked point (LDS copy when the cloud fits).
//
// Tie rule (index-exact parity): the reference's winner among equal distances
// is a function of its block size bs = opt_n_threads(n): thread tid = k % bs
// keeps the lowest k on ties (strict >), and the halving tree lets the lower
// slot win at every level, so across threads the tid with the smallest
// bit-reversed value wins.  That is a total order
//     (d2 desc, bitrev_L(k % bs) asc, k / bs asc)
// which we encode as key = brev32(k % bs) | (k / bs)  (smaller = preferred) and
// reduce in any tree shape we like.
#include "p2r_common.h"

namespace {

__device__ __forceinline__ long long fps_pack(float d2, unsigned key) {
  // d2 >= 0 (or the -1 "no candidate" sentinel): signed compare of the float
  // bit pattern orders them like the float compare of the reference.
  return ((long long)__float_as_int(d2) << 32) | (long long)(unsigned)(~key);
}

__device__ __forceinline__ unsigned fps_key(int k, int L) {
  const unsigned bsm1 = (1u << L) - 1u;
  return __brev((unsigned)k & bsm1) | ((unsigned)k >> L);
}

__device__ __forceinline__ int fps_unkey(unsigned key, int L) {
  const unsigned bsm1 = (1u << L) - 1u;
  const unsigned tid = __brev(key) & bsm1;
  const unsigned q = (L == 0) ? key : (key & ((1u << (32 - L)) - 1u));
  return (int)(tid + (q << L));
}

__device__ __forceinline__ long long i64max(long long a, long long b) { return a > b ? a : b; }

// All-lanes max over the wave; every lane ends with the result.  The round loop of FPS is a chain of dependent
// arg-max reductions, so the latency of this function IS the kernel: DPP moves inside the 16-lane rows and the
// gfx950 lane swaps across rows (a few cycles each) instead of six ds_bpermute round trips through the LDS crossbar
// per 32-bit half (~1400 cycles per round before, measured 0.68 us per round at n = 512).
__device__ __forceinline__ void i64max_parts(int &hi, unsigned &lo, int phi, unsigned plo) {
  const bool take = phi > hi || (phi == hi && plo > lo);
  hi = take ? phi : hi;
  lo = take ? plo : lo;
}

template <int CTRL>
__device__ __forceinline__ void i64max_dpp(int &hi, unsigned &lo) {
  const int phi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  const unsigned plo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
  i64max_parts(hi, lo, phi, plo);
}

__device__ __forceinline__ long long wave_max_i64(long long v) {
  int hi = (int)(v >> 32);
  unsigned lo = (unsigned)(v & 0xFFFFFFFFll);
  i64max_dpp<0xB1>(hi, lo);     // quad_perm [1,0,3,2]
  i64max_dpp<0x4E>(hi, lo);     // quad_perm [2,3,0,1]
  i64max_dpp<0x141>(hi, lo);    // row_half_mirror
  i64max_dpp<0x140>(hi, lo);    // row_mirror: every lane of a 16-lane row holds the row's max
  {
    auto ph = __builtin_amdgcn_permlane16_swap((unsigned)hi, (unsigned)hi, false, false);
    auto pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    int h0 = (int)ph[0]; unsigned l0 = pl[0];
    i64max_parts(h0, l0, (int)ph[1], pl[1]);
    hi = h0; lo = l0;
  }
  {
    auto ph = __builtin_amdgcn_permlane32_swap((unsigned)hi, (unsigned)hi, false, false);
    auto pl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    int h0 = (int)ph[0]; unsigned l0 = pl[0];
    i64max_parts(h0, l0, (int)ph[1], pl[1]);
    hi = h0; lo = l0;
  }
  return ((long long)hi << 32) | (long long)lo;
}

// Register-resident FPS: W waves per cloud, PPT points per lane, n <= 64*W*PPT.
template <int W, int PPT, bool XYZ_IN_LDS>
__global__ __launch_bounds__(W * 64) void fps_reg_kernel(int n, int m, int L,
                                                         const float *__restrict__ dataset,
                                                         int *__restrict__ idxs) {
  constexpr int T = W * 64;
  extern __shared__ float s_dyn[];        // XYZ_IN_LDS: n*3 floats
  __shared__ long long s_slot[2][W > 1 ? W : 1];

  const int tid = threadIdx.x;
  const float *ds = dataset + (size_t)blockIdx.x * n * 3;
  int *out = idxs + (size_t)blockIdx.x * m;

  float px[PPT], py[PPT], pz[PPT], td[PPT];
  unsigned nkey[PPT];  // ~key, 0 marks "never a candidate"
  bool live[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int k = tid + p * T;
    const bool in = k < n;
    const float x = in ? ds[k * 3 + 0] : 0.f;
    const float y = in ? ds[k * 3 + 1] : 0.f;
    const float z = in ? ds[k * 3 + 2] : 0.f;
    px[p] = x; py[p] = y; pz[p] = z;
    td[p] = 1e10f;
    const float mag = (x * x) + (y * y) + (z * z);
    // reference: `if (mag <= 1e-3) continue;` compares in double; float(1e-3) is
    // the first float above 0.001, so the float test `mag < 1e-3f` is identical.
    live[p] = in && !(mag < 1e-3f);
    nkey[p] = ~fps_key(k, L);
    if (XYZ_IN_LDS && in) {
      s_dyn[k * 3 + 0] = x; s_dyn[k * 3 + 1] = y; s_dyn[k * 3 + 2] = z;
    }
  }
  if (tid == 0) out[0] = 0;
  if (XYZ_IN_LDS) __syncthreads();

  const long long kEmpty = ((long long)__float_as_int(-1.0f) << 32) | 0xFFFFFFFFll;  // -> k = 0
  int old = 0;
  for (int j = 1; j < m; ++j) {
    float x1, y1, z1;
    if (XYZ_IN_LDS) {
      x1 = s_dyn[old * 3 + 0]; y1 = s_dyn[old * 3 + 1]; z1 = s_dyn[old * 3 + 2];
    } else {
      x1 = ds[old * 3 + 0]; y1 = ds[old * 3 + 1]; z1 = ds[old * 3 + 2];
    }
    long long best = kEmpty;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const float d = p2r_sqdist(px[p], py[p], pz[p], x1, y1, z1);
      const float d2 = fminf(d, td[p]);
      if (live[p]) {
        td[p] = d2;
        best = i64max(best, ((long long)__float_as_int(d2) << 32) | (long long)nkey[p]);
      }
    }
    best = wave_max_i64(best);
    if (W > 1) {
      const int buf = j & 1;
      if ((tid & 63) == 0) s_slot[buf][tid >> 6] = best;
      __syncthreads();
      long long r = s_slot[buf][0];
#pragma unroll
      for (int w = 1; w < W; ++w) r = i64max(r, s_slot[buf][w]);
      best = r;
    }
    old = fps_unkey(~(unsigned)(best & 0xFFFFFFFFll), L);
    old = __builtin_amdgcn_readfirstlane(old);
    if (tid == 0) out[j] = old;
  }
}

// Streaming FPS for clouds that do not fit the register file: xyz re-read from
// L2 every round, running distances in the caller's `temp` scratch.
__global__ __launch_bounds__(1024) void fps_stream_kernel(int n, int m, int L,
                                                          const float *__restrict__ dataset,
                                                          float *__restrict__ temp,
                                                          int *__restrict__ idxs) {
  constexpr int T = 1024, W = 16;
  __shared__ long long s_slot[2][W];
  const int tid = threadIdx.x;
  const float *ds = dataset + (size_t)blockIdx.x * n * 3;
  float *tp = temp + (size_t)blockIdx.x * n;
  int *out = idxs + (size_t)blockIdx.x * m;
  for (int k = tid; k < n; k += T) tp[k] = 1e10f;  // sampling.cpp:74-76
  if (tid == 0) out[0] = 0;
  const long long kEmpty = ((long long)__float_as_int(-1.0f) << 32) | 0xFFFFFFFFll;
  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
    long long best = kEmpty;
    for (int k = tid; k < n; k += T) {
      const float x = ds[k * 3 + 0], y = ds[k * 3 + 1], z = ds[k * 3 + 2];
      const float mag = (x * x) + (y * y) + (z * z);
      if (mag < 1e-3f) continue;
      const float d = p2r_sqdist(x, y, z, x1, y1, z1);
      const float d2 = fminf(d, tp[k]);
      tp[k] = d2;
      best = i64max(best, fps_pack(d2, fps_key(k, L)));
    }
    best = wave_max_i64(best);
    const int buf = j & 1;
    if ((tid & 63) == 0) s_slot[buf][tid >> 6] = best;
    __syncthreads();
    long long r = s_slot[buf][0];
#pragma unroll
    for (int w = 1; w < W; ++w) r = i64max(r, s_slot[buf][w]);
    old = fps_unkey(~(unsigned)(r & 0xFFFFFFFFll), L);
    old = __builtin_amdgcn_readfirstlane(old);
    if (tid == 0) out[j] = old;
  }
}

// Cooperative FPS for clouds beyond one workgroup's register file: G
// workgroups per cloud (b*G <= 256, so every workgroup is resident), each
// holding a contiguous slice in VGPRs.  A round is the register-resident round
// above plus one exchange of the G slice winners through global memory:
// workgroup g publishes {distance bits | round} and {~key | round} as two
// 8-byte granules with agent-scope stores, and its first wave polls the 2*G
// granules of the cloud (one load instruction) until every tag shows the
// current round -- data-tagged, so no fences and no counters.  Slots alternate
// by round parity; the host zeroes them before the launch (round tags start at
// 1).  The sequential chain per round is one LDS barrier, one publish/poll
// (~2 us) and one LDS broadcast, instead of n/1024 dependent L2 sweeps.
template <int PPT>
__global__ __launch_bounds__(1024) void fps_coop_kernel(int b, int G, int S, int n, int m, int L,
                                                        const float *__restrict__ dataset,
                                                        unsigned long long *__restrict__ slots,
                                                        int *__restrict__ idxs) {
  constexpr int T = 1024, W = 16;
  __shared__ long long s_slot[2][W];
  __shared__ int s_old[2];
  const int tid = threadIdx.x;
  const int cloud = blockIdx.x % b, g = blockIdx.x / b;  // a cloud's workgroups share an XCD when b | 8
  const float *ds = dataset + (size_t)cloud * n * 3;
  int *out = idxs + (size_t)cloud * m;
  unsigned long long *cs = slots + (size_t)cloud * 4 * G;  // [parity][2][G]
  const int k0 = g * S, k1 = min(n, k0 + S);

  float px[PPT], py[PPT], pz[PPT], td[PPT];
  unsigned nkey[PPT];
  bool live[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int k = k0 + tid + p * T;
    const bool in = k < k1;
    const float x = in ? ds[(size_t)k * 3 + 0] : 0.f;
    const float y = in ? ds[(size_t)k * 3 + 1] : 0.f;
    const float z = in ? ds[(size_t)k * 3 + 2] : 0.f;
    px[p] = x; py[p] = y; pz[p] = z;
    td[p] = 1e10f;
    const float mag = (x * x) + (y * y) + (z * z);
    live[p] = in && !(mag < 1e-3f);
    nkey[p] = ~fps_key(k, L);
  }
  if (g == 0 && tid == 0) out[0] = 0;

  const long long kEmpty = ((long long)__float_as_int(-1.0f) << 32) | 0xFFFFFFFFll;
  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = ds[(size_t)old * 3 + 0], y1 = ds[(size_t)old * 3 + 1], z1 = ds[(size_t)old * 3 + 2];
    long long best = kEmpty;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const float d = p2r_sqdist(px[p], py[p], pz[p], x1, y1, z1);
      const float d2 = fminf(d, td[p]);
      if (live[p]) {
        td[p] = d2;
        best = i64max(best, ((long long)__float_as_int(d2) << 32) | (long long)nkey[p]);
      }
    }
    best = wave_max_i64(best);
    const int buf = j & 1;
    if ((tid & 63) == 0) s_slot[buf][tid >> 6] = best;
    __syncthreads();
    if (tid < 64) {
      long long r = s_slot[buf][0];
#pragma unroll
      for (int w = 1; w < W; ++w) r = i64max(r, s_slot[buf][w]);
      unsigned long long *row = cs + (size_t)buf * 2 * G;
      const unsigned long long tag = (unsigned)j;
      if (tid == 0) {
        __hip_atomic_store(row + g, ((unsigned long long)(unsigned)(r >> 32) << 32) | tag, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(row + G + g, ((unsigned long long)(unsigned)r << 32) | tag, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      // lanes 0..G-1 watch the distance granules, lanes 32..32+G-1 the key granules
      const int q = tid & 31;  // G <= 32
      const bool watch = q < G;
      const unsigned long long *src = row + (tid < 32 ? 0 : G) + (watch ? q : 0);
      unsigned long long v = 0;
      int spins = 0;
      bool lost = false;
      for (;;) {
        v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = !watch || (unsigned)v == (unsigned)j;
        if (__all(ok)) break;
        if (++spins > (1 << 22)) { lost = true; break; }   // never hang the device on a peer that is not resident
        __builtin_amdgcn_s_sleep(1);
      }
      if (lost) {   // loud, not silent: the remaining picks become -1 (out of range for every consumer)
        if (tid == 0) s_old[buf] = -1;
        __syncthreads();
        if (g == 0)
          for (int jj = j + tid; jj < m; jj += 64) out[jj] = -1;
        return;
      }
      const unsigned hi = (unsigned)(v >> 32);
      const unsigned other = __shfl_xor(hi, 32, 64);  // pair distance bits with key bits
      long long cand = kEmpty;
      if (watch && tid < 32) cand = ((long long)(int)hi << 32) | (long long)other;
      cand = wave_max_i64(cand);
      if (tid == 0) s_old[buf] = fps_unkey(~(unsigned)(cand & 0xFFFFFFFFll), L);
    }
    __syncthreads();
    old = s_old[buf];
    old = __builtin_amdgcn_readfirstlane(old);
    if (old < 0) return;                                   // the first wave gave up (see above)
    if (g == 0 && tid == 0) out[j] = old;
  }
}

// The workgroups of fps_coop_kernel wait for each other, so all of them must be resident at once.  A plain launch only
// ASSUMES that (b*G <= compute units); a kernel of another stream or process holding compute units breaks it.  The
// launch therefore goes through hipLaunchCooperativeKernel, which places the grid as a whole or refuses it
// (hipErrorCooperativeLaunchTooLarge: the caller then takes the one-workgroup-per-cloud kernel).  The bounded spin in
// the kernel stays as the last line of defence: a peer that never shows up ends the launch with -1 picks, which the
// Python binding turns into a RuntimeError (pointnet2_ops/_ext.py).  Returns P2R_OK, 1 = "not placed, use the
// fallback", or an error code.
template <int PPT>
int launch_coop(int b, int G, int S, int n, int m, int L, const float *dataset, float *temp, int *idxs,
                hipStream_t st) {
  const size_t slot_bytes = (size_t)b * 4 * G * sizeof(unsigned long long);
  int dev = 0, coop = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1;
  if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop) return 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_coop_kernel<PPT>, 1024, 0) != hipSuccess || per_cu < 1)
    return 1;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || b * G > cus) return 1;
  {
    hipError_t e = hipMemsetAsync(temp, 0, slot_bytes, st);
    if (e != hipSuccess) return (int)e;
  }
  unsigned long long *slots = reinterpret_cast<unsigned long long *>(temp);
  void *args[] = {&b, &G, &S, &n, &m, &L, &dataset, &slots, &idxs};
  hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(fps_coop_kernel<PPT>), dim3(b * G),
                                            dim3(1024), args, 0, st);
  if (e == hipErrorCooperativeLaunchTooLarge || e == hipErrorNotSupported) {
    (void)hipGetLastError();
    return 1;
  }
  if (e != hipSuccess) return (int)e;
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

template <int W, int PPT>
int launch_reg(int b, int n, int m, int L, const float *dataset, int *idxs, hipStream_t st) {
  const size_t xyz_bytes = (size_t)n * 3 * sizeof(float);
  if (xyz_bytes <= 64 * 1024) {
    hipLaunchKernelGGL((fps_reg_kernel<W, PPT, true>), dim3(b), dim3(W * 64), xyz_bytes, st, n, m, L,
                       dataset, idxs);
  } else {
    hipLaunchKernelGGL((fps_reg_kernel<W, PPT, false>), dim3(b), dim3(W * 64), 0, st, n, m, L,
                       dataset, idxs);
  }
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

}  // namespace

extern "C" int p2r_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                           int *idxs, void *stream) {
  if (b < 0 || n < 0 || m < 0) return P2R_EINVAL;
  if (b == 0 || m == 0) return P2R_OK;
  if (n == 0) return P2R_EINVAL;
  hipStream_t st = p2r_stream(stream);
  const int bs = p2r_ref_opt_n_threads(n);
  const int L = __builtin_ctz((unsigned)bs);
  // Smallest register tiling that covers n.  One wave holds up to 512 points;
  // larger clouds add waves (<= 16) before adding points per lane.
  if (n <= 64) return launch_reg<1, 1>(b, n, m, L, dataset, idxs, st);
  if (n <= 128) return launch_reg<1, 2>(b, n, m, L, dataset, idxs, st);
  if (n <= 256) return launch_reg<1, 4>(b, n, m, L, dataset, idxs, st);
  if (n <= 512) return launch_reg<1, 8>(b, n, m, L, dataset, idxs, st);
  if (n <= 1024) return launch_reg<2, 8>(b, n, m, L, dataset, idxs, st);
  if (n <= 2048) return launch_reg<4, 8>(b, n, m, L, dataset, idxs, st);
  if (n <= 4096) return launch_reg<8, 8>(b, n, m, L, dataset, idxs, st);
  if (n <= 8192) return launch_reg<16, 8>(b, n, m, L, dataset, idxs, st);
  if (n <= 16384) return launch_reg<16, 16>(b, n, m, L, dataset, idxs, st);
  if (temp == nullptr) return P2R_EINVAL;
  // Several workgroups per cloud when the batch leaves compute units idle
  // (the `temp` scratch, b*n floats, holds the exchange slots).
  // every workgroup of the launch must be resident at once (they wait for each other): one per compute unit
  static int n_cu[P2R_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= P2R_MAX_DEVICES) return P2R_EINVAL;
  if (n_cu[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 1;
    n_cu[dev] = v;
  }
  int G = 1;
  while (G * 2 <= 16 && b * (G * 2) <= n_cu[dev]) G *= 2;  // 16 per cloud measured best (32: slower polls)
  const int S = (n + G - 1) / G;
  if (G >= 2 && S <= 16384 && (reinterpret_cast<uintptr_t>(temp) & 7) == 0 &&
      (size_t)b * 4 * G * sizeof(unsigned long long) <= (size_t)b * n * sizeof(float)) {
    int rc;
    if (S <= 1024) rc = launch_coop<1>(b, G, S, n, m, L, dataset, temp, idxs, st);
    else if (S <= 2048) rc = launch_coop<2>(b, G, S, n, m, L, dataset, temp, idxs, st);
    else if (S <= 4096) rc = launch_coop<4>(b, G, S, n, m, L, dataset, temp, idxs, st);
    else if (S <= 8192) rc = launch_coop<8>(b, G, S, n, m, L, dataset, temp, idxs, st);
    else rc = launch_coop<16>(b, G, S, n, m, L, dataset, temp, idxs, st);
    if (rc != 1) return rc;      // placed (or a real error); 1: co-residency not guaranteed -> one workgroup per cloud
  }
  hipLaunchKernelGGL(fps_stream_kernel, dim3(b), dim3(1024), 0, st, n, m, L, dataset, temp, idxs);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
