// nms3d.hip -- greedy 3D axis-aligned NMS for gfx950, fp64 like the reference.
//
// Replaces nms_3d_faster / nms_3d_faster_samecls (reference net_utils/nms.py:41-119),
// a NumPy loop run per sample on the host after a device->host copy of the
// predictions (net_utils/ap_helper.py:216-232).
//
// MI355X design: one workgroup per box set, the whole problem (<= 1024 boxes x
// 8 doubles = 64 KiB) resident in LDS, so the batch of B samples is one launch
// and nothing leaves the GPU.  Pick order is computed by rank counting (each
// thread counts the boxes that precede its own: O(K^2) compares, no sort
// network), then the greedy sweep walks the order with ONE barrier per pivot:
// every still-alive later box tests itself against the pivot in parallel.
// All arithmetic is IEEE fp64 in the reference's operation order (contraction
// off), so keep masks are bit-exact with NumPy.
#include "p2r_common.h"

namespace {

constexpr int NMS_MAXK = 1024;

__global__ __launch_bounds__(NMS_MAXK) void nms3d_kernel(
    int K, int stride, const double *__restrict__ boxes, const uint8_t *__restrict__ valid,
    double thr, int old_type, int same_cls, uint8_t *__restrict__ keep, int *__restrict__ pick,
    int *__restrict__ npick) {
  extern __shared__ double s_mem[];
  // layout: box[K][8] (x1,y1,z1,x2,y2,z2,score,cls) | area[K] | order[K] (int) | alive[K] (int)
  double *s_box = s_mem;
  double *s_area = s_box + (size_t)K * 8;
  int *s_order = reinterpret_cast<int *>(s_area + K);
  int *s_alive = s_order + K;
  __shared__ int s_nvalid;

  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const double *bx = boxes + (size_t)b * K * stride;
  const uint8_t *vd = valid ? valid + (size_t)b * K : nullptr;

  bool ok = false;
  double score = 0.0;
  if (t == 0) s_nvalid = 0;
  if (t < K) {
    ok = vd ? (vd[t] != 0) : true;
    const double *r = bx + (size_t)t * stride;
    double v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (c < stride) ? r[c] : 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s_box[t * 8 + c] = v[c];
    s_area[t] = (v[3] - v[0]) * (v[4] - v[1]) * (v[5] - v[2]);  // nms.py:49
    score = v[6];
    s_order[t] = -1;
    s_alive[t] = ok ? 1 : -1;  // -1: not taking part
  }
  __syncthreads();

  // Rank = number of participating boxes picked before this one: higher score
  // first; equal scores: higher index first (ascending stable argsort, popped
  // from the back, nms.py:51-55).
  int rank = -1;
  if (t < K && ok) {
    rank = 0;
    for (int u = 0; u < K; ++u) {
      if (s_alive[u] < 0) continue;
      const double su = s_box[u * 8 + 6];
      rank += (su > score) || (su == score && u > t);
    }
    atomicAdd(&s_nvalid, 1);
  }
  __syncthreads();
  const int nvalid = s_nvalid;
  // s_alive is re-indexed by rank from here on.
  __syncthreads();
  if (t < K) s_alive[t] = 1;
  if (rank >= 0) s_order[rank] = t;
  __syncthreads();

  double mine[8];
  double my_area = 0.0;
  if (rank >= 0) {
#pragma unroll
    for (int c = 0; c < 8; ++c) mine[c] = s_box[t * 8 + c];
    my_area = s_area[t];
  }
  for (int p = 0; p + 1 < nvalid; ++p) {
    if (s_alive[p]) {  // uniform: everyone reads the same word
      if (rank > p && s_alive[rank]) {
        const int i = s_order[p];
        const double *bi = s_box + i * 8;
        const double xx1 = fmax(bi[0], mine[0]), yy1 = fmax(bi[1], mine[1]), zz1 = fmax(bi[2], mine[2]);
        const double xx2 = fmin(bi[3], mine[3]), yy2 = fmin(bi[4], mine[4]), zz2 = fmin(bi[5], mine[5]);
        const double l = fmax(0.0, xx2 - xx1), w = fmax(0.0, yy2 - yy1), h = fmax(0.0, zz2 - zz1);
        double o;
        if (old_type) {
          o = (l * w * h) / my_area;  // nms.py:70-71
        } else {
          const double inter = l * w * h;
          o = inter / (s_area[i] + my_area - inter);  // nms.py:73-74
        }
        if (same_cls) o = o * (double)(bi[7] == mine[7]);  // nms.py:115
        if (o > thr) s_alive[rank] = 0;
      }
    }
    __syncthreads();
  }

  if (t < K) {
    const bool kept = rank >= 0 && s_alive[rank] != 0;
    keep[(size_t)b * K + t] = kept ? 1 : 0;
    if (pick) pick[(size_t)b * K + t] = -1;
  }
  __syncthreads();
  if (pick || npick) {
    if (rank >= 0 && s_alive[rank]) {
      int pos = 0;
      for (int q = 0; q < rank; ++q) pos += s_alive[q] != 0;
      if (pick) pick[(size_t)b * K + pos] = t;
    }
    if (t == 0 && npick) {
      int cnt = 0;
      for (int q = 0; q < nvalid; ++q) cnt += s_alive[q] != 0;
      npick[b] = cnt;
    }
  }
}

}  // namespace

extern "C" int p2r_nms3d(int B, int K, int stride, const double *boxes, const uint8_t *valid,
                         double overlap_threshold, int old_type, int same_cls, uint8_t *keep,
                         int *pick, int *npick, void *stream) {
  if (B < 0 || K < 0 || K > NMS_MAXK || (stride != 7 && stride != 8)) return P2R_EINVAL;
  if (same_cls && stride != 8) return P2R_EINVAL;
  if (B == 0) return P2R_OK;
  hipStream_t st = p2r_stream(stream);
  if (K == 0) {
    if (npick) {
      hipError_t e = hipMemsetAsync(npick, 0, sizeof(int) * (size_t)B, st);
      if (e != hipSuccess) return (int)e;
    }
    return P2R_OK;
  }
  const int threads = ((K + 63) / 64) * 64;
  const size_t lds = (size_t)K * 8 * sizeof(double) + (size_t)K * sizeof(double) +
                     2 * (size_t)K * sizeof(int);
  hipLaunchKernelGGL(nms3d_kernel, dim3(B), dim3(threads), lds, st, K, stride, boxes, valid,
                     overlap_threshold, old_type, same_cls, keep, pick, npick);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
