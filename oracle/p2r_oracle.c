/*
 * p2r_oracle.c -- CPU restatement of the Pose2Room hot-path kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * path in pose2room_amd/csrc.  Nothing under pose2room_amd/ may import,
 * link or call it; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do.
 *
 * Every function follows one reference kernel line by line (citations are
 * relative to /root/reference/external/pointnet2_ops_lib/pointnet2_ops/_ext-src
 * unless a full path is given) and keeps the reference's evaluation order:
 * fp32 arithmetic exactly as written in the source, no FMA contraction
 * (build with -ffp-contract=off, see oracle/Makefile), 32-bit int indices.
 *
 * Parity status: the reference ships no test or golden vector for the nine
 * _ext kernels and they cannot be compiled here (CUDA only), so for those
 * nine this restatement *is* the definition ("parity unpinned" by the
 * reference).  nn_distance and nms_3d_faster are pinned
 * against the imported reference Python through tests/golden/.
 *
 * Where the reference scatters with atomicAdd (order undefined on a GPU) the
 * oracle accumulates in ascending (l, j, k) order; tests compare those
 * outputs with a tolerance, everything else bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define P2R_TOTAL_THREADS 512 /* include/cuda_utils.h:13 */

/* include/cuda_utils.h:15-19 -- block size the reference launches with. */
int p2r_oracle_opt_n_threads(int work_size) {
  if (work_size <= 0) return 1;
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > P2R_TOTAL_THREADS) t = P2R_TOTAL_THREADS;
  if (t < 1) t = 1;
  return t;
}

/* src/sampling_gpu.cu:59-65 (__update). */
static void fps_update(float *dists, int *dists_i, int idx1, int idx2) {
  const float v1 = dists[idx1], v2 = dists[idx2];
  const int i1 = dists_i[idx1], i2 = dists_i[idx2];
  dists[idx1] = fmaxf(v1, v2);
  dists_i[idx1] = v2 > v1 ? i2 : i1;
}

/*
 * src/sampling_gpu.cu:69-173 furthest_point_sampling_kernel, emulated thread
 * by thread for the block size of sampling_gpu.cu:175-229.
 * dataset (b,n,3) f32, temp (b,n) f32 pre-filled with 1e10 by the caller
 * (sampling.cpp:74-76), idxs (b,m) i32.
 */
void p2r_oracle_furthest_point_sampling(int b, int n, int m,
                                        const float *dataset, float *temp,
                                        int *idxs) {
  if (m <= 0) return; /* sampling_gpu.cu:73 */
  const int block_size = p2r_oracle_opt_n_threads(n);
  float *dists = (float *)malloc(sizeof(float) * (size_t)block_size);
  int *dists_i = (int *)malloc(sizeof(int) * (size_t)block_size);
  for (int bi = 0; bi < b; ++bi) {
    const float *ds = dataset + (size_t)bi * n * 3;
    float *tp = temp + (size_t)bi * n;
    int *out = idxs + (size_t)bi * m;
    int old = 0;
    out[0] = old; /* :85-86 */
    for (int j = 1; j < m; ++j) {
      const float x1 = ds[old * 3 + 0];
      const float y1 = ds[old * 3 + 1];
      const float z1 = ds[old * 3 + 2];
      for (int tid = 0; tid < block_size; ++tid) {
        int besti = 0;
        float best = -1;
        for (int k = tid; k < n; k += block_size) {
          const float x2 = ds[k * 3 + 0];
          const float y2 = ds[k * 3 + 1];
          const float z2 = ds[k * 3 + 2];
          const float mag = (x2 * x2) + (y2 * y2) + (z2 * z2);
          if (mag <= 1e-3) continue; /* :101, double-precision compare */
          const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +
                          (z2 - z1) * (z2 - z1);
          const float d2 = fminf(d, tp[k]);
          tp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      /* :113-168 -- shared-memory tree, halving stride. */
      for (int h = block_size / 2; h >= 1; h /= 2)
        for (int tid = 0; tid < h; ++tid) fps_update(dists, dists_i, tid, tid + h);
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(dists);
  free(dists_i);
}

/* src/sampling_gpu.cu:8-30 gather_points_kernel. points (b,c,n), idx (b,m). */
void p2r_oracle_gather_points(int b, int c, int n, int m, const float *points,
                              const int *idx, float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[i * m + j];
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
      }
}

/* src/sampling_gpu.cu:34-57 gather_points_grad_kernel (grad_points zeroed by
 * the caller, sampling.cpp:52-54). */
void p2r_oracle_gather_points_grad(int b, int c, int n, int m,
                                   const float *grad_out, const int *idx,
                                   float *grad_points) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[i * m + j];
        grad_points[((size_t)i * c + l) * n + a] +=
            grad_out[((size_t)i * c + l) * m + j];
      }
}

/* src/ball_query_gpu.cu:9-44 query_ball_point_kernel. new_xyz (b,m,3),
 * xyz (b,n,3), idx (b,m,nsample) zeroed by the caller (ball_query.cpp:19-21). */
void p2r_oracle_ball_query(int b, int n, int m, float radius, int nsample,
                           const float *new_xyz, const float *xyz, int *idx) {
  const float radius2 = radius * radius;
  for (int bi = 0; bi < b; ++bi) {
    const float *x = xyz + (size_t)bi * n * 3;
    const float *nx = new_xyz + (size_t)bi * m * 3;
    int *id = idx + (size_t)bi * m * nsample;
    for (int j = 0; j < m; ++j) {
      const float new_x = nx[j * 3 + 0];
      const float new_y = nx[j * 3 + 1];
      const float new_z = nx[j * 3 + 2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float px = x[k * 3 + 0];
        const float py = x[k * 3 + 1];
        const float pz = x[k * 3 + 2];
        const float d2 = (new_x - px) * (new_x - px) +
                         (new_y - py) * (new_y - py) +
                         (new_z - pz) * (new_z - pz);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) id[j * nsample + l] = k;
          id[j * nsample + cnt] = k;
          ++cnt;
        }
      }
    }
  }
}

/* src/group_points_gpu.cu:8-29 group_points_kernel. points (b,c,n),
 * idx (b,npoints,nsample), out (b,c,npoints,nsample). */
void p2r_oracle_group_points(int b, int c, int n, int npoints, int nsample,
                             const float *points, const int *idx, float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *p = points + (size_t)bi * n * c;
    const int *id = idx + (size_t)bi * npoints * nsample;
    float *o = out + (size_t)bi * npoints * nsample * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) {
          const int ii = id[j * nsample + k];
          o[((size_t)l * npoints + j) * nsample + k] = p[(size_t)l * n + ii];
        }
  }
}

/* src/group_points_gpu.cu:43-65 group_points_grad_kernel (grad_points zeroed
 * by the caller, group_points.cpp:46-48). */
void p2r_oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                  const float *grad_out, const int *idx,
                                  float *grad_points) {
  for (int bi = 0; bi < b; ++bi) {
    const float *g = grad_out + (size_t)bi * npoints * nsample * c;
    const int *id = idx + (size_t)bi * npoints * nsample;
    float *gp = grad_points + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) {
          const int ii = id[j * nsample + k];
          gp[(size_t)l * n + ii] += g[((size_t)l * npoints + j) * nsample + k];
        }
  }
}

/* src/interpolate_gpu.cu:9-59 three_nn_kernel. unknown (b,n,3), known (b,m,3),
 * dist2 (b,n,3) f32, idx (b,n,3) i32. */
void p2r_oracle_three_nn(int b, int n, int m, const float *unknown,
                         const float *known, float *dist2, int *idx) {
  for (int bi = 0; bi < b; ++bi) {
    const float *u = unknown + (size_t)bi * n * 3;
    const float *kn = known + (size_t)bi * m * 3;
    float *d2 = dist2 + (size_t)bi * n * 3;
    int *id = idx + (size_t)bi * n * 3;
    for (int j = 0; j < n; ++j) {
      const float ux = u[j * 3 + 0];
      const float uy = u[j * 3 + 1];
      const float uz = u[j * 3 + 2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = kn[k * 3 + 0];
        const float y = kn[k * 3 + 1];
        const float z = kn[k * 3 + 2];
        const float d =
            (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      d2[j * 3 + 0] = (float)best1;
      d2[j * 3 + 1] = (float)best2;
      d2[j * 3 + 2] = (float)best3;
      id[j * 3 + 0] = besti1;
      id[j * 3 + 1] = besti2;
      id[j * 3 + 2] = besti3;
    }
  }
}

/* src/interpolate_gpu.cu:72-101 three_interpolate_kernel. points (b,c,m),
 * idx (b,n,3), weight (b,n,3), out (b,c,n). */
void p2r_oracle_three_interpolate(int b, int c, int m, int n,
                                  const float *points, const int *idx,
                                  const float *weight, float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *p = points + (size_t)bi * m * c;
    const int *id = idx + (size_t)bi * n * 3;
    const float *w = weight + (size_t)bi * n * 3;
    float *o = out + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
        const int i1 = id[j * 3 + 0], i2 = id[j * 3 + 1], i3 = id[j * 3 + 2];
        o[(size_t)l * n + j] = p[(size_t)l * m + i1] * w1 +
                               p[(size_t)l * m + i2] * w2 +
                               p[(size_t)l * m + i3] * w3;
      }
  }
}

/* src/interpolate_gpu.cu:116-144 three_interpolate_grad_kernel (grad_points
 * zeroed by the caller, interpolate.cpp:82-84). */
void p2r_oracle_three_interpolate_grad(int b, int c, int n, int m,
                                       const float *grad_out, const int *idx,
                                       const float *weight, float *grad_points) {
  for (int bi = 0; bi < b; ++bi) {
    const float *g = grad_out + (size_t)bi * n * c;
    const int *id = idx + (size_t)bi * n * 3;
    const float *w = weight + (size_t)bi * n * 3;
    float *gp = grad_points + (size_t)bi * m * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
        const int i1 = id[j * 3 + 0], i2 = id[j * 3 + 1], i3 = id[j * 3 + 2];
        const float go = g[(size_t)l * n + j];
        gp[(size_t)l * m + i1] += go * w1;
        gp[(size_t)l * m + i2] += go * w2;
        gp[(size_t)l * m + i3] += go * w3;
      }
  }
}

/* /root/reference/net_utils/nn_distance.py:15-32 huber_loss, per element. */
static float nnd_huber(float error, float delta) {
  const float abs_error = fabsf(error);
  const float quadratic = abs_error > delta ? delta : abs_error; /* clamp(max=delta) */
  const float linear = abs_error - quadratic;
  return 0.5f * (quadratic * quadratic) + delta * linear;
}

/*
 * /root/reference/net_utils/nn_distance.py:34-61 nn_distance.
 * pc1 (B,N,C), pc2 (B,M,C); mode 0 = squared L2 (:56), 1 = smooth-L1 (:52),
 * 2 = L1 (:54).  dist1/idx1 (B,N) = min over M (:57), dist2/idx2 (B,M) = min
 * over N (:58); first minimal index wins a tie (torch.min CPU semantics).
 * The per-pair sum runs over C in ascending order, one fp32 rounding per add.
 */
void p2r_oracle_nn_distance(int B, int N, int M, int C, const float *pc1,
                            const float *pc2, int mode, float delta,
                            float *dist1, int64_t *idx1, float *dist2,
                            int64_t *idx2) {
  for (int b = 0; b < B; ++b) {
    const float *a = pc1 + (size_t)b * N * C;
    const float *q = pc2 + (size_t)b * M * C;
    for (int i = 0; i < N; ++i) { dist1[(size_t)b * N + i] = INFINITY; idx1[(size_t)b * N + i] = 0; }
    for (int j = 0; j < M; ++j) { dist2[(size_t)b * M + j] = INFINITY; idx2[(size_t)b * M + j] = 0; }
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < M; ++j) {
        float s = 0.0f;
        for (int c = 0; c < C; ++c) {
          const float diff = a[i * C + c] - q[j * C + c];
          float t;
          if (mode == 1) t = nnd_huber(diff, delta);
          else if (mode == 2) t = fabsf(diff);
          else t = diff * diff;
          s = (c == 0) ? t : s + t;
        }
        if (s < dist1[(size_t)b * N + i] || j == 0) { dist1[(size_t)b * N + i] = s; idx1[(size_t)b * N + i] = j; }
        if (s < dist2[(size_t)b * M + j] || i == 0) { dist2[(size_t)b * M + j] = s; idx2[(size_t)b * M + j] = i; }
      }
  }
}

/*
 * Backward of nn_distance through the two torch.min gathers
 * (autograd of nn_distance.py:49-58): grad flows to the arg-min pair only.
 * g1 (B,N), g2 (B,M) -> grad_pc1 (B,N,C), grad_pc2 (B,M,C), both overwritten.
 */
void p2r_oracle_nn_distance_grad(int B, int N, int M, int C, const float *pc1,
                                 const float *pc2, int mode, float delta,
                                 const int64_t *idx1, const int64_t *idx2,
                                 const float *g1, const float *g2,
                                 float *grad_pc1, float *grad_pc2) {
  memset(grad_pc1, 0, sizeof(float) * (size_t)B * N * C);
  memset(grad_pc2, 0, sizeof(float) * (size_t)B * M * C);
  for (int b = 0; b < B; ++b) {
    const float *a = pc1 + (size_t)b * N * C;
    const float *q = pc2 + (size_t)b * M * C;
    float *ga = grad_pc1 + (size_t)b * N * C;
    float *gq = grad_pc2 + (size_t)b * M * C;
    for (int pass = 0; pass < 2; ++pass) {
      const int cnt = pass == 0 ? N : M;
      for (int t = 0; t < cnt; ++t) {
        const int i = pass == 0 ? t : (int)idx2[(size_t)b * M + t];
        const int j = pass == 0 ? (int)idx1[(size_t)b * N + t] : t;
        const float g = pass == 0 ? g1[(size_t)b * N + t] : g2[(size_t)b * M + t];
        for (int c = 0; c < C; ++c) {
          const float diff = a[i * C + c] - q[j * C + c];
          float d;
          if (mode == 1) {
            const float ad = fabsf(diff);
            const float sg = (diff > 0.0f) - (diff < 0.0f);
            d = ad <= delta ? diff : delta * sg;
          } else if (mode == 2) {
            d = (float)((diff > 0.0f) - (diff < 0.0f));
          } else {
            d = 2.0f * diff;
          }
          ga[i * C + c] += g * d;
          gq[j * C + c] -= g * d;
        }
      }
    }
  }
}

/*
 * /root/reference/net_utils/nms.py:41-77 nms_3d_faster and :79-119
 * nms_3d_faster_samecls, fp64 throughout.
 * boxes (K,stride) rows = [x1,y1,z1,x2,y2,z2,score(,cls)].  Returns the number
 * of picks; pick[] receives original box indices in pick order (descending
 * score).  np.argsort (nms.py:51) is not stable; the oracle orders equal
 * scores by ascending index (what a stable argsort yields), fixtures use
 * distinct scores.
 */
typedef struct { double s; int i; } nms_ent;
static int nms_cmp(const void *pa, const void *pb) {
  const nms_ent *a = (const nms_ent *)pa, *b = (const nms_ent *)pb;
  if (a->s < b->s) return -1;
  if (a->s > b->s) return 1;
  return (a->i > b->i) - (a->i < b->i);
}
int p2r_oracle_nms3d(int K, int stride, const double *boxes,
                     double overlap_threshold, int old_type, int same_cls,
                     int *pick) {
  if (K <= 0) return 0;
  nms_ent *ord = (nms_ent *)malloc(sizeof(nms_ent) * (size_t)K);
  double *area = (double *)malloc(sizeof(double) * (size_t)K);
  int *I = (int *)malloc(sizeof(int) * (size_t)K);
  for (int k = 0; k < K; ++k) {
    const double *r = boxes + (size_t)k * stride;
    ord[k].s = r[6];
    ord[k].i = k;
    area[k] = (r[3] - r[0]) * (r[4] - r[1]) * (r[5] - r[2]); /* :49 */
  }
  qsort(ord, (size_t)K, sizeof(nms_ent), nms_cmp);
  for (int k = 0; k < K; ++k) I[k] = ord[k].i;
  int size = K, npick = 0;
  while (size != 0) { /* :53 */
    const int last = size;
    const int i = I[last - 1];
    pick[npick++] = i;
    const double *bi = boxes + (size_t)i * stride;
    int w = 0;
    for (int t = 0; t < last - 1; ++t) {
      const int r = I[t];
      const double *br = boxes + (size_t)r * stride;
      const double xx1 = fmax(bi[0], br[0]), yy1 = fmax(bi[1], br[1]), zz1 = fmax(bi[2], br[2]);
      const double xx2 = fmin(bi[3], br[3]), yy2 = fmin(bi[4], br[4]), zz2 = fmin(bi[5], br[5]);
      const double l = fmax(0.0, xx2 - xx1), ww = fmax(0.0, yy2 - yy1), h = fmax(0.0, zz2 - zz1);
      double o;
      if (old_type) {
        o = (l * ww * h) / area[r]; /* :70-71 */
      } else {
        const double inter = l * ww * h;
        o = inter / (area[i] + area[r] - inter); /* :73-74 */
      }
      if (same_cls) o = o * (double)(bi[7] == br[7]); /* :115 */
      if (!(o > overlap_threshold)) I[w++] = r; /* :76 keeps o<=thr (and NaN) */
    }
    size = w;
  }
  free(ord); free(area); free(I);
  return npick;
}

/*
 * /root/reference/net_utils/nms.py:7-39 nms_2d_faster, fp64, restated in TWO dimensions (on purpose not through
 * p2r_oracle_nms3d: the product maps the 2-D boxes onto the 3-D kernel with a unit extent, and this is what that
 * mapping is checked against).  boxes (K,5) rows = [x1,y1,x2,y2,score].
 */
int p2r_oracle_nms2d(int K, const double *boxes, double overlap_threshold, int old_type, int *pick) {
  if (K <= 0) return 0;
  nms_ent *ord = (nms_ent *)malloc(sizeof(nms_ent) * (size_t)K);
  double *area = (double *)malloc(sizeof(double) * (size_t)K);
  int *I = (int *)malloc(sizeof(int) * (size_t)K);
  for (int k = 0; k < K; ++k) {
    const double *r = boxes + (size_t)k * 5;
    ord[k].s = r[4];
    ord[k].i = k;
    area[k] = (r[2] - r[0]) * (r[3] - r[1]); /* :13 */
  }
  qsort(ord, (size_t)K, sizeof(nms_ent), nms_cmp);
  for (int k = 0; k < K; ++k) I[k] = ord[k].i;
  int size = K, npick = 0;
  while (size != 0) { /* :17 */
    const int last = size;
    const int i = I[last - 1];
    pick[npick++] = i;
    const double *bi = boxes + (size_t)i * 5;
    int w = 0;
    for (int t = 0; t < last - 1; ++t) {
      const int r = I[t];
      const double *br = boxes + (size_t)r * 5;
      const double xx1 = fmax(bi[0], br[0]), yy1 = fmax(bi[1], br[1]);
      const double xx2 = fmin(bi[2], br[2]), yy2 = fmin(bi[3], br[3]);
      const double ww = fmax(0.0, xx2 - xx1), h = fmax(0.0, yy2 - yy1);
      double o;
      if (old_type) {
        o = (ww * h) / area[r]; /* :31 */
      } else {
        const double inter = ww * h;
        o = inter / (area[i] + area[r] - inter); /* :33-34 */
      }
      if (!(o > overlap_threshold)) I[w++] = r; /* :36 */
    }
    size = w;
  }
  free(ord); free(area); free(I);
  return npick;
}
