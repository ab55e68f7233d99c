// split16.h -- helpers of the opt-in `split16` arithmetic of the ST-GCN kernels (gfx950).
//
// In this mode an fp32 product a * b runs on the 16-bit matrix pipe as THREE v_mfma_f32_16x16x32_f16 products of
// two-part fp16 operands, accumulated in fp32:
//     a = a1 + a2,  a1 = fp16(a),  a2 = fp16(a - a1)        (22 of fp32's 24 significand bits)
//     a * b ~= a1 b1 + (a1 b2 + a2 b1)                      (the a2 b2 term is 2^-22 of the product: dropped)
// fp16 has fp32's precision problem turned into a RANGE problem (|x| <= 65504, normal numbers from 6e-5):
//  * every operand tensor is multiplied by a power of two that puts its largest magnitude into [2^12, 2^13) before it is
//    split (exact in fp32; the 8x headroom is for the graph conv, which splits coefficient-weighted SUMS of neighbours),
//    and the accumulators are scaled back by the inverse power when they leave.  The largest magnitude of a tensor is
//    one word in device memory (the float bits of max |x|, produced by the kernel that wrote the tensor or by
//    p2r_absmax_bits): the scale never visits the host;
//  * gradient tensors are heavy-tailed (measured on the P2RNet backward: max |x| ~ 2^20 x the typical element), so one
//    scale per tensor leaves the RESIDUAL x - x1 of a typical element in fp16's subnormals (absolute error 2^-25 at the
//    tensor's scale = 2^-17 of such an element).  The residual of a runtime operand is therefore kept scaled by 2^11,
//    b2' = fp16((b - b1) * 2^11) -- a normal number whenever b1 is -- and the product that consumes it takes a weight
//    part scaled the other way, a1' = 2^-11 a1 (a third fp16 plane from the host, exact):
//        a * b ~= a1 b1 + (a1' b2' + a2 b1)
//    which keeps every term at one scale in one accumulator and 22 significand bits down to 2^-27 of the tensor's
//    maximum.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 p2r_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 p2r_h2 __attribute__((ext_vector_type(2)));
typedef float p2r_f2 __attribute__((ext_vector_type(2)));

// amax_bits: float bits of max |x| over the tensor (NULL: the tensor is used as it is, scale 1).
// -> s = 2^S with max|x| * 2^S in [2^12, 2^13), inv = 2^-S.  Zero, subnormal, infinite or NaN maxima give S = 0.
__device__ __forceinline__ void p2r_split_scale(const unsigned *__restrict__ amax_bits, float &s, float &inv) {
  s = 1.f; inv = 1.f;
  if (amax_bits == nullptr) return;
  const int eb = (int)((*amax_bits >> 23) & 0xffu);          // max|x| in [2^(eb-127), 2^(eb-126))
  if (eb == 0 || eb == 255) return;
  int S = 139 - eb;                                          // 12 - (eb - 127)
  S = S > 126 ? 126 : S;                                     // (eb >= 2 keeps 2^-S normal on the other side)
  s = __uint_as_float((unsigned)(S + 127) << 23);
  inv = __uint_as_float((unsigned)(127 - S) << 23);
}

#define P2R_RES_SCALE 2048.f      // 2^11: scale of a runtime operand's residual part (see above)

struct P2RSplit8 { p2r_h8 p, q; };
// x (8 floats) -> x1 = fp16(x), x2' = fp16((x - x1) * 2^11).  The empty asm makes the split see VALUES: under fp
// contraction the compiler may otherwise round an expression once for x1 and differently inside the fused subtraction,
// and x1 + x2 then misses by one fp16 ulp at exact ties (tools/ubench/split_probe.hip).  Plain conversions and
// arithmetic -- no inline assembly next to the MFMA builtins, whose hazards only the compiler tracks.
__device__ __forceinline__ P2RSplit8 p2r_split8(const float (&v)[8]) {
  P2RSplit8 s;
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    p2r_f2 x = {v[i], v[i + 1]};
    asm volatile("" : "+v"(x));
    const p2r_h2 p = __builtin_convertvector(x, p2r_h2);
    const p2r_h2 q = __builtin_convertvector((x - __builtin_convertvector(p, p2r_f2)) * P2R_RES_SCALE, p2r_h2);
    s.p[i] = p.x; s.p[i + 1] = p.y; s.q[i] = q.x; s.q[i + 1] = q.y;
  }
  return s;
}

// largest finite fp16: operands are clamped to it where no scale is available (forward activations)
#define P2R_H16_MAX 65504.f
