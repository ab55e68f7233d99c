// p2r_common.h -- shared device/host helpers for libp2r_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2r_hip.h"

// Parity rule: every distance / interpolation expression that the oracle
// (oracle/p2r_oracle.c) evaluates in source order must round identically here,
// so contraction of a*b+c into fma is disabled for all parity-critical code.
// Kernels that WANT fused multiply-add (MFMA tiles, fmaf) ask for it explicitly.
#pragma clang fp contract(off)

#define P2R_WAVE 64

#define P2R_LAUNCH_CHECK()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

static inline hipStream_t p2r_stream(void *s) { return (hipStream_t)s; }

// include/cuda_utils.h:15-19 of the reference: the block size its kernels are
// launched with.  Only the FPS tie rule depends on it.
static inline int p2r_ref_opt_n_threads(int work_size) {
  if (work_size <= 0) return 1;
  const int pow_2 = (int)(__builtin_log((double)work_size) / __builtin_log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

static inline int p2r_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: a process that launches on a
// second GPU must set it there too.  `done` is one flag per device (zero-initialised static array owned by the call
// site); racing threads at worst both set the attribute.
#define P2R_MAX_DEVICES 64
template <typename K>
static inline hipError_t p2r_allow_big_lds(K kernel, unsigned char (&done)[P2R_MAX_DEVICES], int bytes = 160 * 1024) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < P2R_MAX_DEVICES && done[dev]) return hipSuccess;
  e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && dev >= 0 && dev < P2R_MAX_DEVICES) done[dev] = 1;
  return e;
}

// Squared distance exactly as the reference writes it:
// (a-b)*(a-b) + (c-d)*(c-d) + (e-f)*(e-f), left-to-right, no contraction.
__device__ __forceinline__ float p2r_sqdist(float ax, float ay, float az,
                                            float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return dx * dx + dy * dy + dz * dz;
}

// Sum of v over the 16 lanes of a DPP row (lanes 16g .. 16g+15); every lane of the row gets the total.
// Pure VALU (quad_perm / row_half_mirror / row_mirror butterflies), no LDS traffic.
__device__ __forceinline__ float p2r_row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));  // row_mirror
  return v;
}
