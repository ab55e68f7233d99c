// det_loss.hip -- BoxNetDetectionLoss as three launches (forward, finalise, backward), gfx950.
//
// Replaces the reference's models/loss.py:35-189 -- vote loss (:90-115) with its nearest-joint vote selection,
// proposal -> ground-truth assignment and objectness labelling (:117-150), centre chamfer incl. the padded-GT quirk,
// size / heading / semantic-class terms gathered through the assignment (:42-88), the weighted total (:167) and the
// three logged ratios (:169-178) -- which in torch is ~150 micro-kernels forward and ~200 backward, each a launch and
// a pass over a few KB.  Everything here is a few hundred KB per batch, so the kernel is latency-bound by design:
// one workgroup per sample computes that sample's partial sums and the un-normalised per-element gradients, a
// one-workgroup pass combines the partials in a fixed order into the ten scalars, and the backward scales the stored
// gradient pieces by the global normalisers.
//
// Arithmetic follows the reference's expressions: squared distances as (dx*dx + dy*dy) + dz*dz without contraction,
// first minimum on ties (torch.min on the reference's CPU path), Huber with delta 1, f64 for the heading branch
// (the mixture head emits f64), `total` accumulated as the reference's left-to-right expression
// ((((10 v + 5 o) + 10 c) + 10 s) in f32, then + 10 h + sem in f64).
#include "p2r_common.h"

namespace {

constexpr int DL_T = 256;            // threads per workgroup
constexpr int DL_MAXG = 32;          // ground-truth slots supported
constexpr int DL_NPART = 12;         // f32 partial sums per sample
constexpr float DL_FAR_AWAY = 1.0e18f;

enum { P_NUM_VOTE, P_DEN_VOTE, P_NUM_OBJ, P_DEN_OBJ, P_NLABEL, P_NUM_C1, P_NUM_C2, P_DEN_BOX, P_NUM_SIZE, P_NUM_SEM,
       P_NACC, P_UNUSED };

struct DetLossShape {
  int B, S, J, T, K, G, NC;          // samples, seeds, joints, frames, proposals, GT slots, classes
  int j0;                            // origin joint
  float near_thr, far_thr, w0, w1;   // objectness labelling thresholds and class weights
};

__device__ __forceinline__ float huber(float e) {           // nn_distance.py:15-32, delta = 1
  const float a = fabsf(e), q = fminf(a, 1.f);
  return 0.5f * q * q + (a - q);
}
__device__ __forceinline__ float huber_grad(float e) {      // autograd of the above (clamp passes its gradient at |e| == 1)
  const float a = fabsf(e);
  const float s = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
  return (a <= 1.f ? a : 1.f) * s;
}
__device__ __forceinline__ double huber64(double e) {
  const double a = fabs(e), q = fmin(a, 1.0);
  return 0.5 * q * q + (a - q);
}
__device__ __forceinline__ double huber_grad64(double e) {
  const double a = fabs(e);
  const double s = e > 0.0 ? 1.0 : (e < 0.0 ? -1.0 : 0.0);
  return (a <= 1.0 ? a : 1.0) * s;
}

__global__ __launch_bounds__(DL_T) void det_loss_forward_kernel(
    DetLossShape sh, const float *__restrict__ seed_skeleton, const float *__restrict__ vote_xyz,
    const long long *__restrict__ seed_inds, const float *__restrict__ vote_label,
    const long long *__restrict__ vote_label_mask, const float *__restrict__ agg_xyz,
    const float *__restrict__ center, const float *__restrict__ size, const double *__restrict__ heading,
    const float *__restrict__ obj_scores, const float *__restrict__ sem_scores,
    const float *__restrict__ center_label, const float *__restrict__ box_mask, const float *__restrict__ gt_size,
    const float *__restrict__ gt_heading, const long long *__restrict__ gt_cls,
    float *__restrict__ partial /* [B][12] */, double *__restrict__ partial64 /* [B] */,
    float *__restrict__ g_vote /* [B][S][3] */, float *__restrict__ g_obj /* [B][K][2] */,
    float *__restrict__ g_c1 /* [B][K][3] */, float *__restrict__ g_c2 /* [B][K][3] */,
    float *__restrict__ g_size /* [B][K][3] */, double *__restrict__ g_head /* [B][K][2] */,
    float *__restrict__ g_sem /* [B][K][NC] */) {
  extern __shared__ float lds[];
  float *red = lds;                                  // [DL_T][DL_NPART]
  double *red64 = reinterpret_cast<double *>(lds + DL_T * DL_NPART);   // [DL_T]
  float *dmat = reinterpret_cast<float *>(red64 + DL_T);               // [G][K] centre distances
  float *gtc = dmat + sh.G * sh.K;                   // [G][3] centre labels, [G] mask
  int *idx2 = reinterpret_cast<int *>(gtc + 4 * sh.G);                 // [G] nearest proposal of every GT slot
  float *c2acc = reinterpret_cast<float *>(idx2 + sh.G);               // [K][3] dist2-side centre gradient

  const int b = blockIdx.x, tid = threadIdx.x;
  float acc[DL_NPART];
#pragma unroll
  for (int i = 0; i < DL_NPART; ++i) acc[i] = 0.f;
  double acc64 = 0.0;

  for (int e = tid; e < sh.G; e += DL_T) {
    gtc[4 * e + 0] = center_label[((size_t)b * sh.G + e) * 3 + 0];
    gtc[4 * e + 1] = center_label[((size_t)b * sh.G + e) * 3 + 1];
    gtc[4 * e + 2] = center_label[((size_t)b * sh.G + e) * 3 + 2];
    gtc[4 * e + 3] = box_mask[(size_t)b * sh.G + e];
  }

  // ---- vote loss (loss.py:90-115): per seed, the GT vote nearest to any joint of the seed skeleton ---------------
  for (int s = tid; s < sh.S; s += DL_T) {
    const size_t bs = (size_t)b * sh.S + s;
    const long long fr = seed_inds[bs];
    const float mask = (float)vote_label_mask[((size_t)b * sh.T + fr) * sh.J + sh.j0];
    const float *vl = vote_label + (((size_t)b * sh.T + fr) * sh.J + sh.j0) * 9;
    const float *sk = seed_skeleton + bs * sh.J * 3;
    const float ox = sk[sh.j0 * 3 + 0], oy = sk[sh.j0 * 3 + 1], oz = sk[sh.j0 * 3 + 2];
    float vx[3], vy[3], vz[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) { vx[v] = ox + vl[3 * v + 0]; vy[v] = oy + vl[3 * v + 1]; vz[v] = oz + vl[3 * v + 2]; }
    float best = 0.f;
    int bestv = 0;
    for (int j = 0; j < sh.J; ++j) {                 // dist2[j] = min_v, ind2[j] = first arg-min; then first arg-min over j
      const float px = sk[3 * j + 0], py = sk[3 * j + 1], pz = sk[3 * j + 2];
      float dj = 0.f;
      int vj = 0;
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const float d = p2r_sqdist(vx[v], vy[v], vz[v], px, py, pz);
        if (v == 0 || d < dj) { dj = d; vj = v; }
      }
      if (j == 0 || dj < best) { best = dj; bestv = vj; }
    }
    const float ex = vote_xyz[bs * 3 + 0] - (bestv == 0 ? vx[0] : bestv == 1 ? vx[1] : vx[2]);
    const float ey = vote_xyz[bs * 3 + 1] - (bestv == 0 ? vy[0] : bestv == 1 ? vy[1] : vy[2]);
    const float ez = vote_xyz[bs * 3 + 2] - (bestv == 0 ? vz[0] : bestv == 1 ? vz[1] : vz[2]);
    const float el = (huber(ex) + huber(ey) + huber(ez)) / 3.f;
    acc[P_NUM_VOTE] += el * mask;
    acc[P_DEN_VOTE] += mask;
    g_vote[bs * 3 + 0] = mask * huber_grad(ex) / 3.f;
    g_vote[bs * 3 + 1] = mask * huber_grad(ey) / 3.f;
    g_vote[bs * 3 + 2] = mask * huber_grad(ez) / 3.f;
  }
  __syncthreads();      // gtc visible

  // ---- proposals: assignment, objectness, centre (dist1 side), size, heading, class ------------------------------
  for (int k = tid; k < sh.K; k += DL_T) {
    const size_t bk = (size_t)b * sh.K + k;
    // assignment over the valid GT boxes (padded rows pushed to a far sentinel, as p2rnet/loss.py did in torch)
    const float ax = agg_xyz[bk * 3 + 0], ay = agg_xyz[bk * 3 + 1], az = agg_xyz[bk * 3 + 2];
    const float cx = center[bk * 3 + 0], cy = center[bk * 3 + 1], cz = center[bk * 3 + 2];
    float d_as = 0.f, d_c1 = 0.f;
    int as = 0, i1 = 0;
    for (int gi = 0; gi < sh.G; ++gi) {
      const float gx = gtc[4 * gi], gy = gtc[4 * gi + 1], gz = gtc[4 * gi + 2];
      const bool valid = gtc[4 * gi + 3] > 0.f;
      const float da = valid ? p2r_sqdist(ax, ay, az, gx, gy, gz)
                             : p2r_sqdist(ax, ay, az, DL_FAR_AWAY, DL_FAR_AWAY, DL_FAR_AWAY);
      if (gi == 0 || da < d_as) { d_as = da; as = gi; }
      const float dc = p2r_sqdist(cx, cy, cz, gx, gy, gz);       // ALL slots, zero padding included (loss.py:64)
      dmat[gi * sh.K + k] = dc;
      if (gi == 0 || dc < d_c1) { d_c1 = dc; i1 = gi; }
    }
    const float eu = sqrtf(d_as + 1e-6f);
    const bool near = eu < sh.near_thr, far = eu > sh.far_thr;
    const float label = near ? 1.f : 0.f, omask = (near || far) ? 1.f : 0.f;
    // objectness: class-weighted cross entropy, reduction none
    const float s0 = obj_scores[bk * 2 + 0], s1 = obj_scores[bk * 2 + 1];
    const float mx = fmaxf(s0, s1);
    const float lse = mx + logf(expf(s0 - mx) + expf(s1 - mx));
    const float w = near ? sh.w1 : sh.w0;
    const float nll = w * (lse - (near ? s1 : s0));
    acc[P_NUM_OBJ] += nll * omask;
    acc[P_DEN_OBJ] += omask;
    acc[P_NLABEL] += label;
    const int pred = s1 > s0 ? 1 : 0;                           // argmax: first maximum on ties
    acc[P_NACC] += ((pred == (near ? 1 : 0)) ? 1.f : 0.f) * omask;
    const float p0 = expf(s0 - lse), p1 = expf(s1 - lse);
    g_obj[bk * 2 + 0] = omask * w * (p0 - (near ? 0.f : 1.f));
    g_obj[bk * 2 + 1] = omask * w * (p1 - (near ? 1.f : 0.f));
    // centre, dist1 side
    acc[P_NUM_C1] += d_c1 * label;
    g_c1[bk * 3 + 0] = label * 2.f * (cx - gtc[4 * i1 + 0]);
    g_c1[bk * 3 + 1] = label * 2.f * (cy - gtc[4 * i1 + 1]);
    g_c1[bk * 3 + 2] = label * 2.f * (cz - gtc[4 * i1 + 2]);
    c2acc[3 * k + 0] = c2acc[3 * k + 1] = c2acc[3 * k + 2] = 0.f;
    // size / heading / class through the assignment
    float ls = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float e = size[bk * 3 + d] - gt_size[((size_t)b * sh.G + as) * 3 + d];
      ls += huber(e);
      g_size[bk * 3 + d] = label * huber_grad(e) / 3.f;
    }
    acc[P_NUM_SIZE] += (ls / 3.f) * label;
    double lh = 0.0;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const double e = heading[bk * 2 + d] - (double)gt_heading[((size_t)b * sh.G + as) * 2 + d];
      lh += huber64(e);
      g_head[bk * 2 + d] = (double)label * huber_grad64(e) / 2.0;
    }
    acc64 += (lh / 2.0) * (double)label;
    const int cls = (int)gt_cls[(size_t)b * sh.G + as];
    const float *sc = sem_scores + bk * sh.NC;
    float m = sc[0];
    for (int c = 1; c < sh.NC; ++c) m = fmaxf(m, sc[c]);
    float se = 0.f;
    for (int c = 0; c < sh.NC; ++c) se += expf(sc[c] - m);
    const float lse2 = m + logf(se);
    acc[P_NUM_SEM] += (lse2 - sc[cls]) * label;
    for (int c = 0; c < sh.NC; ++c) g_sem[bk * sh.NC + c] = label * (expf(sc[c] - lse2) - (c == cls ? 1.f : 0.f));
  }
  __syncthreads();      // dmat complete

  // ---- centre, dist2 side: nearest proposal of every GT slot (first minimum), masked by box_label_mask -----------
  for (int gi = tid; gi < sh.G; gi += DL_T) {
    float d2 = 0.f;
    int k2 = 0;
    for (int k = 0; k < sh.K; ++k) {
      const float d = dmat[gi * sh.K + k];
      if (k == 0 || d < d2) { d2 = d; k2 = k; }
    }
    idx2[gi] = k2;
    acc[P_NUM_C2] += d2 * gtc[4 * gi + 3];
    acc[P_DEN_BOX] += gtc[4 * gi + 3];
  }
  __syncthreads();
  if (tid == 0) {        // a proposal may be the nearest of several slots: serial, deterministic accumulation
    for (int gi = 0; gi < sh.G; ++gi) {
      const int k2 = idx2[gi];
      const size_t bk = (size_t)b * sh.K + k2;
      const float bm = gtc[4 * gi + 3];
#pragma unroll
      for (int d = 0; d < 3; ++d) c2acc[3 * k2 + d] += bm * 2.f * (center[bk * 3 + d] - gtc[4 * gi + d]);
    }
  }
  __syncthreads();
  for (int e = tid; e < sh.K * 3; e += DL_T) g_c2[(size_t)b * sh.K * 3 + e] = c2acc[e];

  // ---- per-sample partial sums (fixed order) ------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < DL_NPART; ++i) red[tid * DL_NPART + i] = acc[i];
  red64[tid] = acc64;
  __syncthreads();
  if (tid < DL_NPART) {
    float t = 0.f;
    for (int i = 0; i < DL_T; ++i) t += red[i * DL_NPART + tid];
    partial[(size_t)b * DL_NPART + tid] = t;
  }
  if (tid == DL_NPART) {
    double t = 0.0;
    for (int i = 0; i < DL_T; ++i) t += red64[i];
    partial64[b] = t;
  }
}

// out32: [0] vote [1] objectness [2] center [3] size [4] sem_cls [5] pos_ratio [6] neg_ratio [7] obj_acc
//        [8] 1/(den_vote+1e-6) [9] 1/(den_obj+1e-6) [10] 1/n_pos [11] 1/(den_box+1e-6)      out64: [0] heading [1] total
__global__ void det_loss_finalize_kernel(int B, int K, const float *__restrict__ partial,
                                         const double *__restrict__ partial64, float *__restrict__ out32,
                                         double *__restrict__ out64) {
  if (threadIdx.x != 0) return;
  float t[DL_NPART];
  for (int i = 0; i < DL_NPART; ++i) t[i] = 0.f;
  double h = 0.0;
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < DL_NPART; ++i) t[i] += partial[(size_t)b * DL_NPART + i];
    h += partial64[b];
  }
  const float n_pos = t[P_NLABEL] + 1e-6f;
  const float vote = t[P_NUM_VOTE] / (t[P_DEN_VOTE] + 1e-6f);
  const float obj = t[P_NUM_OBJ] / (t[P_DEN_OBJ] + 1e-6f);
  const float c1 = t[P_NUM_C1] / n_pos, c2 = t[P_NUM_C2] / (t[P_DEN_BOX] + 1e-6f);
  const float cen = (c1 + c2) / 2.f;
  const float siz = t[P_NUM_SIZE] / n_pos;
  const float sem = t[P_NUM_SEM] / n_pos;
  const double head = h / (double)n_pos;
  const float total_n = (float)B * (float)K;
  out32[0] = vote; out32[1] = obj; out32[2] = cen; out32[3] = siz; out32[4] = sem;
  const float pos_ratio = t[P_NLABEL] / total_n;
  out32[5] = pos_ratio;
  out32[6] = t[P_DEN_OBJ] / total_n - pos_ratio;
  out32[7] = t[P_NACC] / (t[P_DEN_OBJ] + 1e-6f);
  out32[8] = 1.f / (t[P_DEN_VOTE] + 1e-6f);
  out32[9] = 1.f / (t[P_DEN_OBJ] + 1e-6f);
  out32[10] = 1.f / n_pos;
  out32[11] = 1.f / (t[P_DEN_BOX] + 1e-6f);
  const float t32 = ((10.f * vote + 5.f * obj) + 10.f * cen) + 10.f * siz;
  out64[0] = head;
  out64[1] = ((double)t32 + 10.0 * head) + (double)sem;
}

// coef [6] f64: upstream gradient reaching each term = d total * weight + d term  (vote, objectness, center, size,
// heading, sem_cls), built by the caller.
__global__ __launch_bounds__(DL_T) void det_loss_backward_kernel(
    int B, int S, int K, int NC, const double *__restrict__ coef, const float *__restrict__ out32,
    const float *__restrict__ g_vote, const float *__restrict__ g_obj, const float *__restrict__ g_c1,
    const float *__restrict__ g_c2, const float *__restrict__ g_size, const double *__restrict__ g_head,
    const float *__restrict__ g_sem, float *__restrict__ d_vote, float *__restrict__ d_obj,
    float *__restrict__ d_center, float *__restrict__ d_size, double *__restrict__ d_head,
    float *__restrict__ d_sem) {
  const float cv = (float)coef[0] * out32[8], co = (float)coef[1] * out32[9];
  const float cc1 = (float)coef[2] * 0.5f * out32[10], cc2 = (float)coef[2] * 0.5f * out32[11];
  const float cs = (float)coef[3] * out32[10], cm = (float)coef[5] * out32[10];
  const double ch = coef[4] * (double)out32[10];
  const size_t gid = (size_t)blockIdx.x * DL_T + threadIdx.x, stride = (size_t)gridDim.x * DL_T;
  for (size_t e = gid; e < (size_t)B * S * 3; e += stride) d_vote[e] = cv * g_vote[e];
  for (size_t e = gid; e < (size_t)B * K * 2; e += stride) { d_obj[e] = co * g_obj[e]; d_head[e] = ch * g_head[e]; }
  for (size_t e = gid; e < (size_t)B * K * 3; e += stride) {
    d_center[e] = cc1 * g_c1[e] + cc2 * g_c2[e];
    d_size[e] = cs * g_size[e];
  }
  for (size_t e = gid; e < (size_t)B * K * NC; e += stride) d_sem[e] = cm * g_sem[e];
}

}  // namespace

// Forward of BoxNetDetectionLoss (models/loss.py:152-189).  Shapes: seed_skeleton (B,S,J,3) f32, vote_xyz (B,S,3),
// seed_inds (B,S) i64, vote_label (B,T,J,9), vote_label_mask (B,T,J) i64, agg_xyz / center / size (B,K,3) f32,
// heading (B,K,2) f64, obj_scores (B,K,2), sem_scores (B,K,NC), center_label (B,G,3), box_mask (B,G), gt_size (B,G,3),
// gt_heading (B,G,2) f32, gt_cls (B,G) i64.  Outputs: out32 [12] f32 and out64 [2] f64 (layout above the finalise
// kernel), scratch partial [B][12] f32 / partial64 [B] f64, and the un-normalised gradient pieces g_* consumed by
// p2r_det_loss_backward.
extern "C" int p2r_det_loss_forward(int B, int S, int J, int T, int K, int G, int NC, int j0, float near_thr,
                                    float far_thr, float w0, float w1, const float *seed_skeleton,
                                    const float *vote_xyz, const int64_t *seed_inds, const float *vote_label,
                                    const int64_t *vote_label_mask, const float *agg_xyz, const float *center,
                                    const float *size, const double *heading, const float *obj_scores,
                                    const float *sem_scores, const float *center_label, const float *box_mask,
                                    const float *gt_size, const float *gt_heading, const int64_t *gt_cls,
                                    float *partial, double *partial64, float *out32, double *out64, float *g_vote,
                                    float *g_obj, float *g_c1, float *g_c2, float *g_size, double *g_head,
                                    float *g_sem, void *stream) {
  if (B <= 0 || S <= 0 || J <= 0 || T <= 0 || K <= 0 || G <= 0 || G > DL_MAXG || NC <= 0 || j0 < 0 || j0 >= J)
    return P2R_EINVAL;
  DetLossShape sh{B, S, J, T, K, G, NC, j0, near_thr, far_thr, w0, w1};
  const size_t lds = (size_t)DL_T * DL_NPART * sizeof(float) + (size_t)DL_T * sizeof(double) +
                     (size_t)G * K * sizeof(float) + (size_t)4 * G * sizeof(float) + (size_t)G * sizeof(int) +
                     (size_t)3 * K * sizeof(float);
  if (lds > 64 * 1024) return P2R_EINVAL;
  hipLaunchKernelGGL(det_loss_forward_kernel, dim3(B), dim3(DL_T), lds, p2r_stream(stream), sh, seed_skeleton,
                     vote_xyz, reinterpret_cast<const long long *>(seed_inds), vote_label,
                     reinterpret_cast<const long long *>(vote_label_mask), agg_xyz, center, size, heading, obj_scores,
                     sem_scores, center_label, box_mask, gt_size, gt_heading,
                     reinterpret_cast<const long long *>(gt_cls), partial, partial64, g_vote, g_obj, g_c1, g_c2,
                     g_size, g_head, g_sem);
  P2R_LAUNCH_CHECK();
  hipLaunchKernelGGL(det_loss_finalize_kernel, dim3(1), dim3(64), 0, p2r_stream(stream), B, K, partial, partial64,
                     out32, out64);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

// Backward: coef [6] f64 (device) = gradient reaching (vote, objectness, center, size, heading, sem_cls); out32 and the
// g_* pieces from the forward -> gradients of vote_xyz, objectness_scores, center, size, heading (f64), sem_cls_scores.
extern "C" int p2r_det_loss_backward(int B, int S, int K, int NC, const double *coef, const float *out32,
                                     const float *g_vote, const float *g_obj, const float *g_c1, const float *g_c2,
                                     const float *g_size, const double *g_head, const float *g_sem, float *d_vote,
                                     float *d_obj, float *d_center, float *d_size, double *d_head, float *d_sem,
                                     void *stream) {
  if (B <= 0 || S <= 0 || K <= 0 || NC <= 0) return P2R_EINVAL;
  const size_t n = (size_t)B * (S * 3 > K * NC ? S * 3 : K * NC);
  int blocks = (int)((n + DL_T - 1) / DL_T);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(det_loss_backward_kernel, dim3(blocks), dim3(DL_T), 0, p2r_stream(stream), B, S, K, NC, coef,
                     out32, g_vote, g_obj, g_c1, g_c2, g_size, g_head, g_sem, d_vote, d_obj, d_center, d_size, d_head,
                     d_sem);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
