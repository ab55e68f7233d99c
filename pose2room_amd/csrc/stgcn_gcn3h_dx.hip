// stgcn_gcn3h_dx.hip -- data gradient of the fused graph convolution in split16 arithmetic: the kernel of
// stgcn_gcn3h_body.h over the ROW lists with the transposed planes, dX = sum_k W_k^T (dZ . A_k^T) (+ the gradient of
// the block's identity branch, masked on the way in); schedule gcn3h_sched_r.inc (tools/gen_gcn_split_sched.py).
#include "gcn3h_sched_r.inc"
#define H3_KERNEL gcn3h_dx_kernel
#define H3_RES_SCALED 1      // the operand is a gradient: heavy-tailed (split16.h)
#include "stgcn_gcn3h_body.h"

static const int h3r_pairs[H3_NPAIRS][2] = H3_PAIRS;

unsigned long long p2r_gcn3h_signature_r(void) { return H3_SIGNATURE; }
int p2r_gcn3h_pairs_r(int *out) {
  if (!out) return P2R_EINVAL;
  for (int i = 0; i < H3_NPAIRS; ++i) { out[2 * i] = h3r_pairs[i][0]; out[2 * i + 1] = h3r_pairs[i][1]; }
  return H3_NPAIRS;
}

extern "C" int p2r_stgcn_gcn3h_data_gradient(int N, int T, int V, int K, int ltot, const float *dz, const void *Wh,
                                             const float *winv, const float *coef, const float *addend,
                                             const unsigned char *addend_mask, float *dx, const unsigned *dz_amax,
                                             void *stream) {
  if (V != H3_V || K != 11 || ltot != H3_LTOT) return P2R_EINVAL;
  return h3_launch(N, T, dz, Wh, winv, coef, nullptr, addend, addend_mask, dx, nullptr, nullptr, dz_amax, stream);
}
