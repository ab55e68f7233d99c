// split16.hip -- the range word of the opt-in split16 arithmetic (split16.h): max |x| of a tensor as float bits in device
// memory.  The kernels that WRITE the operand tensors of the split16 kernels emit this word as a by-product
// (p2r_bn_bwd_apply_amax, p2r_stgcn_tconv_weight_grad_dz_amax, p2r_bn_apply_amax); this standalone pass serves every
// other producer (one read of the tensor).
#include "p2r_common.h"

namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void absmax_kernel(long long n, const float *__restrict__ x, unsigned *__restrict__ amax) {
  // |x| as unsigned: for non-negative floats the bit patterns order like the values (NaN patterns sort above infinity:
  // a NaN anywhere makes the word a NaN pattern, which p2r_split_scale treats as "no scale")
  unsigned m = 0;
  const long long n4 = ((uintptr_t)x % 16 == 0) ? n / 4 : 0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(x) + i);
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
    m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
  }
  for (long long i = 4 * n4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  __shared__ unsigned sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(amax, max(max(sm[0], sm[1]), max(sm[2], sm[3])));
}

}  // namespace

extern "C" int p2r_absmax_bits(long long n, const float *x, unsigned *amax_bits, void *stream) {
  if (n < 0 || !amax_bits) return P2R_EINVAL;
  hipStream_t st = p2r_stream(stream);
  hipError_t e = hipMemsetAsync(amax_bits, 0, sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  if (n == 0) return P2R_OK;
  const long long want = (n / 4 + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, st, n, x, amax_bits);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
