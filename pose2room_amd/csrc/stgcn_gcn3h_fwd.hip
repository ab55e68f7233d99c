// stgcn_gcn3h_fwd.hip -- forward of the fused graph convolution in split16 arithmetic (column lists); the kernel is
// stgcn_gcn3h_body.h, the schedule gcn3h_sched_c.inc (tools/gen_gcn_split_sched.py).
#include "gcn3h_sched_c.inc"
#define H3_KERNEL gcn3h_fwd_kernel
#include "stgcn_gcn3h_body.h"

static const int h3c_pairs[H3_NPAIRS][2] = H3_PAIRS;

// form 0: this file's schedule (forward); form 1: the data gradient's (stgcn_gcn3h_dx.hip)
unsigned long long p2r_gcn3h_signature_r(void);
int p2r_gcn3h_pairs_r(int *out);

extern "C" unsigned long long p2r_stgcn_gcn3h_signature(int form) {
  return form == 0 ? H3_SIGNATURE : (form == 1 ? p2r_gcn3h_signature_r() : 0ULL);
}
extern "C" int p2r_stgcn_gcn3h_pairs(int form, int *out) {
  if (form == 1) return p2r_gcn3h_pairs_r(out);
  if (form != 0 || !out) return P2R_EINVAL;
  for (int i = 0; i < H3_NPAIRS; ++i) { out[2 * i] = h3c_pairs[i][0]; out[2 * i + 1] = h3c_pairs[i][1]; }
  return H3_NPAIRS;
}

extern "C" int p2r_stgcn_gcn3h_forward(int N, int T, int V, int K, int ltot, const float *x, const void *Wh,
                                       const float *winv, const float *coef, const float *bias_cv, float *z,
                                       float *stats_partial, int *n_partials, const unsigned *x_amax, void *stream) {
  if (V != H3_V || K != 11 || ltot != H3_LTOT) return P2R_EINVAL;
  return h3_launch(N, T, x, Wh, winv, coef, bias_cv, nullptr, nullptr, z, stats_partial, n_partials, x_amax, stream);
}
