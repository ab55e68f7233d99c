// Probe of a compiler behaviour that cost the fp16 plane-pair prototype (gcn3h_proto.hip) 1.8e-5 of range:
//   v = c * x;  p = (_Float16)v;  q = (_Float16)(v - (float)p);
// hipcc -O3 (fp-contract=fast) emits p_for_the_subtraction = v_fma_mixlo_f16(c, x, 0) -- ONE rounding of the exact
// product -- but stores p = v_cvt_pk_f16_f32(v_mul_f32(c, x)).  Where fp32(c * x) is an exact fp16 tie the two differ
// by one fp16 ulp and p + q misses v by that ulp.  Arguments: pairs of fp32 bit patterns (hex) c x; prints p, q.
//   hipcc -O3 --offload-arch=gfx950 -o split_probe tools/ubench/split_probe.hip
//   ./split_probe 407eaced 3e1535a1      ->  p 38a4 q 0c00   (q should be 8c00: v - p = -2^-12)
// An empty asm("" : "+v"(v)) between the product and the split removes it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
__global__ void k(const float *c, const float *x, unsigned short *out, int n) {
  int i = threadIdx.x; if (i >= n) return;
  float v = c[i] * x[i];
  _Float16 p = (_Float16)v; _Float16 q = (_Float16)(v - (float)p);
  out[4*i] = __builtin_bit_cast(unsigned short, p); out[4*i+1] = __builtin_bit_cast(unsigned short, q);
  {
#pragma clang fp contract(off)
  float v2 = c[i] * x[i];
  _Float16 p2 = (_Float16)v2; _Float16 q2 = (_Float16)(v2 - (float)p2);
  out[4*i+2] = __builtin_bit_cast(unsigned short, p2); out[4*i+3] = __builtin_bit_cast(unsigned short, q2);
  }
}
int main(int argc, char **argv) {
  int n = (argc - 1) / 2; float hc[64], hx[64];
  for (int i = 0; i < n; ++i) { uint32_t a = strtoul(argv[1+2*i], 0, 16), b = strtoul(argv[2+2*i], 0, 16); memcpy(&hc[i], &a, 4); memcpy(&hx[i], &b, 4); }
  float *c, *x; unsigned short *o, ho[256];
  (void)hipMalloc(&c, 256); (void)hipMalloc(&x, 256); (void)hipMalloc(&o, 512);
  (void)hipMemcpy(c, hc, 4*n, hipMemcpyHostToDevice); (void)hipMemcpy(x, hx, 4*n, hipMemcpyHostToDevice);
  k<<<1, 64>>>(c, x, o, n); (void)hipMemcpy(ho, o, 8*n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("c %.9g x %.9g prod %.9g: contracted p %04x q %04x | plain p %04x q %04x\n", hc[i], hx[i], hc[i]*hx[i], ho[4*i], ho[4*i+1], ho[4*i+2], ho[4*i+3]);
}
