/*
 * p2r_hip.h -- C ABI of libp2r_hip.so, the MI355X (gfx950) implementation of
 * the Pose2Room hot path.
 *
 * Drop-in boundary.  The first nine entry points have exactly the argument
 * lists of the reference's internal `*_kernel_wrapper` launchers (the natural
 * C ABI underneath its pybind module `pointnet2_ops._ext`), plus a trailing
 * `hipStream_t` (passed as void*) and an int status instead of the reference's
 * `exit(-1)` (include/cuda_utils.h:30-39).  Citations are relative to
 * /root/reference/external/pointnet2_ops_lib/pointnet2_ops/_ext-src.
 *
 * Conventions
 *   - every pointer is a device pointer into HBM; tensors are contiguous,
 *     fp32 / int32 exactly as in the reference; index arithmetic is 32-bit
 *     per batch row like the reference's.
 *   - the library never allocates, never synchronises and never touches the
 *     default stream: it enqueues on `stream` and returns.
 *   - return value: 0 on success, a hipError_t code if a launch failed,
 *     P2R_EINVAL for invalid sizes.  No entry point calls exit().
 *   - gradient entry points OVERWRITE their output (they do not require the
 *     caller to zero it, unlike the reference's atomicAdd kernels), except
 *     where noted.
 */
#ifndef P2R_HIP_H
#define P2R_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2R_OK 0
#define P2R_EINVAL (-22)

/* Library / build identification (sanity check for loaders). */
int p2r_abi_version(void);          /* currently 3 (adds the split16 entry points); 2 = rounds 3-5; 1 = rounds 1-2 (p2r_bn_finalize without `width`) */
const char *p2r_build_arch(void);   /* "gfx950" */

/* ---- pointnet2_ops._ext: the nine reference launchers ------------------ */

/* replaces furthest_point_sampling_kernel_wrapper (src/sampling.cpp:11-13,
 * src/sampling_gpu.cu:175-229).  dataset (b,n,3) f32; temp (b,n) f32 scratch
 * (the reference requires it pre-filled with 1e10; this library fills it
 * itself when it needs it and ignores it -- it may be NULL -- when n is small
 * enough for the register-resident kernel, n <= 16384); idxs (b,m) i32.
 * Index-exact with the reference, including its block-size dependent
 * tie-break (see DESIGN.md "FPS tie rule"). */
int p2r_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                float *temp, int *idxs, void *stream);

/* replaces gather_points_kernel_wrapper (src/sampling.cpp:4-6,
 * src/sampling_gpu.cu:8-30).  points (b,c,n), idx (b,npoints) -> out (b,c,npoints). */
int p2r_gather_points(int b, int c, int n, int npoints, const float *points,
                      const int *idx, float *out, void *stream);

/* replaces gather_points_grad_kernel_wrapper (src/sampling.cpp:7-9,
 * src/sampling_gpu.cu:34-57).  grad_out (b,c,npoints), idx (b,npoints) ->
 * grad_points (b,c,n), overwritten. */
int p2r_gather_points_grad(int b, int c, int n, int npoints,
                           const float *grad_out, const int *idx,
                           float *grad_points, void *stream);

/* replaces query_ball_point_kernel_wrapper (src/ball_query.cpp:4-6,
 * src/ball_query_gpu.cu:9-54).  new_xyz (b,m,3), xyz (b,n,3) ->
 * idx (b,m,nsample) i32; every slot is written (rows with no neighbour are
 * written as zeros, the value the reference's zero-initialised output keeps). */
int p2r_ball_query(int b, int n, int m, float radius, int nsample,
                   const float *new_xyz, const float *xyz, int *idx,
                   void *stream);

/* replaces group_points_kernel_wrapper (src/group_points.cpp:4-6,
 * src/group_points_gpu.cu:8-39).  points (b,c,n), idx (b,npoints,nsample) ->
 * out (b,c,npoints,nsample). */
int p2r_group_points(int b, int c, int n, int npoints, int nsample,
                     const float *points, const int *idx, float *out,
                     void *stream);

/* replaces group_points_grad_kernel_wrapper (src/group_points.cpp:8-10,
 * src/group_points_gpu.cu:43-75).  grad_out (b,c,npoints,nsample) ->
 * grad_points (b,c,n), overwritten.  Index lists that fit the LDS with n * slots
 * <= 2^22 are summed in ascending slot order (deterministic, bit-equal to the
 * sequential loop); larger problems use an order-free form (LDS scatter or global
 * atomics: the same sums up to the order of the additions).  Slots whose index is
 * outside [0, n) contribute nothing in the ordered forms. */
int p2r_group_points_grad(int b, int c, int n, int npoints, int nsample,
                          const float *grad_out, const int *idx,
                          float *grad_points, void *stream);

/* replaces three_nn_kernel_wrapper (src/interpolate.cpp:4-5,
 * src/interpolate_gpu.cu:9-68).  unknown (b,n,3), known (b,m,3) ->
 * dist2 (b,n,3) f32 (squared), idx (b,n,3) i32. */
int p2r_three_nn(int b, int n, int m, const float *unknown, const float *known,
                 float *dist2, int *idx, void *stream);

/* replaces three_interpolate_kernel_wrapper (src/interpolate.cpp:6-8,
 * src/interpolate_gpu.cu:72-111).  points (b,c,m), idx (b,n,3),
 * weight (b,n,3) -> out (b,c,n). */
int p2r_three_interpolate(int b, int c, int m, int n, const float *points,
                          const int *idx, const float *weight, float *out,
                          void *stream);

/* replaces three_interpolate_grad_kernel_wrapper (src/interpolate.cpp:9-12,
 * src/interpolate_gpu.cu:116-154).  grad_out (b,c,n) -> grad_points (b,c,m),
 * overwritten. */
int p2r_three_interpolate_grad(int b, int c, int n, int m,
                               const float *grad_out, const int *idx,
                               const float *weight, float *grad_points,
                               void *stream);

/* ---- net_utils/nn_distance.py (chamfer), /root/reference/net_utils ----- */

#define P2R_NND_L2 0       /* nn_distance.py:56 */
#define P2R_NND_L1SMOOTH 1 /* nn_distance.py:52, huber_loss :15-32 */
#define P2R_NND_L1 2       /* nn_distance.py:54 */

/* replaces nn_distance (nn_distance.py:34-61).  pc1 (B,N,C), pc2 (B,M,C) f32
 * -> dist1 (B,N) f32, idx1 (B,N) i64, dist2 (B,M) f32, idx2 (B,M) i64; first
 * minimal index on ties.  Any of the four outputs may be NULL. */
int p2r_nn_distance(int B, int N, int M, int C, const float *pc1,
                    const float *pc2, int mode, float delta, float *dist1,
                    int64_t *idx1, float *dist2, int64_t *idx2, void *stream);

/* backward of the above through both min-gathers: g1 (B,N), g2 (B,M) (either
 * may be NULL = zero) -> grad_pc1 (B,N,C), grad_pc2 (B,M,C), overwritten,
 * deterministic (no atomics). */
int p2r_nn_distance_grad(int B, int N, int M, int C, const float *pc1,
                         const float *pc2, int mode, float delta,
                         const int64_t *idx1, const int64_t *idx2,
                         const float *g1, const float *g2, float *grad_pc1,
                         float *grad_pc2, void *stream);

/* ---- models/loss.py: BoxNetDetectionLoss -------------------------------- */

/* replaces BoxNetDetectionLoss.__call__ (loss.py:152-189: vote loss :90-115, proposal/GT
 * correspondence and objectness :117-150, centre / size / heading / class terms :42-88)
 * -- some 150 torch micro-kernels forward -- by two launches.  seed_skeleton (B,S,J,3),
 * vote_xyz (B,S,3) f32, seed_inds (B,S) i64, vote_label (B,T,J,9) f32, vote_label_mask
 * (B,T,J) i64, agg_xyz / center / size (B,K,3) f32, heading (B,K,2) f64, obj_scores (B,K,2),
 * sem_scores (B,K,NC) f32, center_label (B,G,3), box_mask (B,G), gt_size (B,G,3), gt_heading
 * (B,G,2) f32, gt_cls (B,G) i64.
 * out32 [12] f32: vote, objectness, center, size, sem_cls, pos_ratio, neg_ratio, obj_acc and the
 * four normalisers' reciprocals; out64 [2] f64: heading_loss, total.  partial [B][12] f32 and
 * partial64 [B] f64 are scratch; g_vote (B,S,3), g_obj (B,K,2), g_c1, g_c2, g_size (B,K,3),
 * g_head (B,K,2) f64, g_sem (B,K,NC) receive the un-normalised gradient pieces. */
int p2r_det_loss_forward(int B, int S, int J, int T, int K, int G, int NC, int j0, float near_thr,
                         float far_thr, float w0, float w1, const float *seed_skeleton,
                         const float *vote_xyz, const int64_t *seed_inds, const float *vote_label,
                         const int64_t *vote_label_mask, const float *agg_xyz, const float *center,
                         const float *size, const double *heading, const float *obj_scores,
                         const float *sem_scores, const float *center_label, const float *box_mask,
                         const float *gt_size, const float *gt_heading, const int64_t *gt_cls,
                         float *partial, double *partial64, float *out32, double *out64, float *g_vote,
                         float *g_obj, float *g_c1, float *g_c2, float *g_size, double *g_head,
                         float *g_sem, void *stream);

/* autograd of the above: coef [6] f64 on the device = gradient reaching (vote, objectness,
 * center, size, heading, sem_cls) -> gradients of vote_xyz, objectness_scores, center, size,
 * heading (f64) and sem_cls_scores, overwritten. */
int p2r_det_loss_backward(int B, int S, int K, int NC, const double *coef, const float *out32,
                          const float *g_vote, const float *g_obj, const float *g_c1, const float *g_c2,
                          const float *g_size, const double *g_head, const float *g_sem, float *d_vote,
                          float *d_obj, float *d_center, float *d_size, double *d_head, float *d_sem,
                          void *stream);

/* ---- net_utils/nms.py -------------------------------------------------- */

/* replaces nms_3d_faster / nms_3d_faster_samecls (nms.py:41-77, :79-119),
 * batched over B independent box sets.  boxes (B,K,stride) f64 rows
 * [x1,y1,z1,x2,y2,z2,score(,cls)], stride 7 or 8; valid (B,K) u8 or NULL
 * (all valid) selects the rows that take part (the reference's
 * nonempty_box_mask sub-selection, ap_helper.py:216-228).
 * Outputs: keep (B,K) u8 mask over the original rows; pick (B,K) i32 =
 * surviving original indices in pick order (descending score), padded with
 * -1; npick (B) i32.  pick / npick may be NULL.  K <= 1024.
 * Equal scores are ordered as a stable ascending argsort would (the
 * reference's np.argsort is unstable there). */
int p2r_nms3d(int B, int K, int stride, const double *boxes,
              const uint8_t *valid, double overlap_threshold, int old_type,
              int same_cls, uint8_t *keep, int *pick, int *npick, void *stream);

/* ---- ST-GCN spatial graph convolution (models/p2rnet/modules/stgcn_layers.py) --- */

/* replaces ConvTemporalGraphical.forward (stgcn_layers.py:57-67): Conv2d 1x1
 * (64 -> K*64) followed by einsum('nkctv,kvw->nctw') with A*importance, fused;
 * the (N,K*64,T,V) intermediate never exists.  x (N,64,T,V) f32 -> z (N,64,T,V).
 * W [K][64][64] = the conv weight (row = output channel of plane k); the
 * adjacency is passed in sparse column-list form: nbr u8 / coef f32 tables
 * [sum_k Lk][V] (j-th source joint of column w in plane k and its A*importance
 * value, zero-padded), Lk_host[K] on the HOST; bias_cv [64][V] = the conv bias
 * pushed through the graph product, or NULL.  Called with transposed planes and
 * row lists it yields the data gradient.  K <= 16, Lk <= 12, V <= 128.
 * Table convention (all graph-conv entry points): each list is sorted by joint
 * index and padded with (joint 0, coefficient 0), so a slot j >= 1 holding joint 0
 * is padding; the kernels skip the padded tail of a tile's lists.
 * stats_partial, when not NULL, receives [N * ceil(T / (384 / V))][64][2]: per
 * workgroup (sum, sum of squares) of z per channel -- the batch statistics the
 * BatchNorm after the graph conv needs (tcn.0, stgcn_layers.py:400), taken from the
 * accumulators instead of a second pass over z. */
int p2r_stgcn_gcn_forward(int N, int T, int V, int K, const int *Lk_host,
                          const float *x, const float *W, const uint8_t *nbr,
                          const float *coef, const float *bias_cv, float *z,
                          float *stats_partial, void *stream);

/* Second generation of the same operator (csrc/stgcn_gcn2.hip): MFMA n-tile = 16 frames of one joint, so a
 * (plane, joint) unit with an empty neighbour list is skipped exactly and the list is wave-uniform; channel
 * phases of 16 double-buffered in LDS by LDS-DMA; persistent workgroups; software-pipelined work stream.
 * V must be 53.
 *   Wp     [K][4][4][64][4] f32: Wp[k][ph][m][16 g + r][s] = W_k[16 m + r][16 ph + 4 s + g]
 *   coef   [ltot][V] as above;  stream int32 [8][80 * 12 + 16]: the static per-wave work stream
 *   (record layout in csrc/stgcn_gcn2.hip; built by pose2room_amd/p2rnet/gcn_tables.build_stream);
 *   stream_work: scratch of the same size (the stream with the current coefficients, read by scalar loads)
 *   addend (N,64,T,V) or NULL is added to the result on the way out (the residual-branch gradient in the
 *   data-gradient launch); stats_partial [*n_partials][64][2] (optional); z == NULL queries *n_partials.
 *   bwd_u / bwd_mask / bwd_fin (all or none; data-gradient launch, stats_partial required): the statistics
 *   epilogue then emits the reduction pass of the BatchNorm + residual + ReLU backward of the block in front
 *   (what p2r_bn_bwd_reduce with relu = 3 computes from the stored result): per channel (sum g', sum g' * uhat),
 *   g' = result where the ReLU mask byte bwd_mask (N,64,T,V) u8 is set, uhat = (u - mean) * invstd,
 *   u = bwd_u (N,64,T,V) the saved input of that BatchNorm, bwd_fin [4][64] rows 0, 1 = mean, invstd. */
int p2r_stgcn_gcn2_forward(int N, int T, int V, int K, int ltot, const float *x, const float *Wp,
                           const float *coef, const int *stream, int *stream_work,
                           const float *bias_cv, const float *addend, float *z, float *stats_partial,
                           int *n_partials, const float *bwd_u, const unsigned char *bwd_mask,
                           const float *bwd_fin, void *stream_h);

/* Third generation of the same operator (csrc/stgcn_gcn3.hip): the data movement of the second generation with the
 * per-wave work list resolved at BUILD time (tools/gen_gcn_sched.py -> csrc/gcn3_sched.inc): accumulator slots,
 * source joints and coefficient-table indices are immediates of a straight-line program per wave; no work stream,
 * no fill pre-launch.  The schedule is generated for the P2RNet skeleton (stgcn_layers.py:151-205: 53 joints, 11
 * planes); p2r_stgcn_gcn3_signature(form) returns the 64-bit signature of the neighbour-table pattern of
 * form 0 (column lists, forward) / 1 (row lists, data gradient) -- pose2room_amd.p2rnet.gcn_tables.pattern_signature
 * of the caller's tables must equal it, every other adjacency goes through p2r_stgcn_gcn2_forward.
 * Arguments as p2r_stgcn_gcn2_forward without the stream; additionally T % 16 == 0 and x, z, addend 16-byte aligned
 * (P2R_EINVAL otherwise: use the second generation).
 * stats_partial of a forward-statistics launch (no bwd_*) is [*n_partials][64][3] = (count, mean, M2) per workgroup
 * and channel -- sums about a pivot per (wave, row), merged with their counts at the end of the kernel, so the
 * variance survives |mean| >> std (p2r_bn_finalize width 3); with bwd_* it is [*n_partials][64][2] as before. */
unsigned long long p2r_stgcn_gcn3_signature(int form);
int p2r_stgcn_gcn3_forward(int N, int T, int V, int K, int ltot, int form, const float *x, const float *Wp,
                           const float *coef, const float *bias_cv, const float *addend, float *z,
                           float *stats_partial, int *n_partials, const float *bwd_u,
                           const unsigned char *bwd_mask, const float *bwd_fin, void *stream_h);

/* The data-gradient launch of p2r_stgcn_gcn3_forward (form = 1, no bias table) whose addend is MASKED on the way in:
 * z += addend where addend_mask (N,64,T,53 bytes, 4-byte aligned) is non-zero.  In the backward of a chain of
 * st_gcn blocks (stgcn_layers.py:413-439: `x = self.tcn(x) + res; return self.relu(x), A`) addend = the gradient
 * arriving at the previous block's output and addend_mask = that block's ReLU mask: the residual-branch gradient
 * dout * mask, which the BatchNorm-backward pass then does not have to write.  bwd_u / bwd_mask / bwd_fin and
 * stats_partial as in p2r_stgcn_gcn3_forward: all given = with the BatchNorm-backward sums epilogue, all NULL = plain
 * (the caller runs p2r_bn_bwd_reduce itself, e.g. on another stream under the gradient kernels that follow). */
int p2r_stgcn_gcn3_data_gradient_masked_addend(int N, int T, int V, int K, int ltot, const float *x,
                                               const float *Wp, const float *coef, const float *addend,
                                               const unsigned char *addend_mask, float *z, float *stats_partial,
                                               const float *bwd_u, const unsigned char *bwd_mask,
                                               const float *bwd_fin, void *stream_h);

/* Adjacency gradient at the row-list entries, statically scheduled like p2r_stgcn_gcn3_forward
 * (csrc/stgcn_gcn3_grad.hip; requires p2r_stgcn_gcn3_signature(1) == signature of the caller's row tables):
 *   dcoef[lofs_k + j][v] = sum over (n, t, c) of (W_k . x)[c, t, v] * dz[c, t, w_j(k, v)]
 * x, dz (N,64,T,53); Wp [K][4][4][64][4] = the forward planes in kernel order; dcoef_partial
 * [n_blocks][ltot][53] per-workgroup sums, summed over the leading axis by the caller.  T % 16 == 0, dz 16-byte
 * aligned (P2R_EINVAL otherwise: use p2r_stgcn_gcn_coef_grad).  Replaces the autograd of the einsum
 * w.r.t. `A * edge_importance` (stgcn_layers.py:62-65, stgcn.py:134). */
int p2r_stgcn_gcn3_coef_grad(int N, int T, int V, int K, int ltot, const float *x, const float *dz,
                             const float *Wp, int n_blocks, float *dcoef_partial, void *stream);

/* Weight gradient, statically scheduled like p2r_stgcn_gcn3_forward (csrc/stgcn_gcn3_dw.hip; requires
 * p2r_stgcn_gcn3_signature(1) == signature of the caller's row tables):
 *   dw_partial [n_blocks][K][64][64] = per-workgroup partial of dW_k TRANSPOSED ([k][ci][c]),
 *   dW_k[c][ci] = sum over (n, t, v) of (sum_j a_k(v, w_j) dz[c, t, w_j]) * x[ci, t, v]
 * (aggregation on the gradient side through the row lists; coef [ltot][53] = row coefficient table), summed over the
 * leading axis and transposed by the caller (p2r_sum_leading, tr64).  colsum_partial (optional) [n_blocks][64][53] =
 * per-workgroup sums of dz over samples and frames (gradient of the bias table).  T % 4 == 0, x and dz 16-byte
 * aligned (P2R_EINVAL otherwise: use p2r_stgcn_gcn_weight_grad).  Replaces the autograd of the 1x1 conv weight
 * (stgcn_layers.py:57-67). */
int p2r_stgcn_gcn3_weight_grad(int N, int T, int V, int K, int ltot, const float *x, const float *dz,
                               const float *coef, int n_blocks, float *dw_partial, float *colsum_partial,
                               void *stream);

/* weight gradient of the above (autograd of stgcn_layers.py:62-65): with G_k = x aggregated
 * through the lists of plane k, dw_partial [n_blocks][K][64][64] holds per-workgroup sums over
 * all columns of dz[a][col] * G_k[b][col] at [k][a][b], to be summed over the leading axis by the
 * caller (deterministic).  K <= 12, V <= 64.  Called as (x, dz, column lists) this is dW_k[c][ci];
 * called as (dz, x, row lists) it is dW_k transposed ([k][ci][c]) -- the same numbers with the
 * aggregation on the gradient side, where the row lists leave more (plane, joint-group) units
 * empty to skip.  colsum_partial, when not NULL, receives [n_blocks][64][V] partial sums over
 * samples and frames of the `dz` argument (colsum_of_x = 0) or of the `x` argument
 * (colsum_of_x != 0): the gradient of the bias table bias_cv, read from the same tiles. */
int p2r_stgcn_gcn_weight_grad(int N, int T, int V, int K, const int *Lk_host,
                              const float *x, const float *dz,
                              const uint8_t *nbr, const float *coef,
                              int n_blocks, float *dw_partial, float *colsum_partial,
                              int colsum_of_x, void *stream);

/* gradient w.r.t. the non-zero adjacency entries (reaches edge_importance,
 * stgcn.py:134): H_k = Wt_k^T . dz on MFMA, reduced against x gathered through the lists:
 * dcoef_partial [n_blocks][sum_k Lk][V] (entry (k, j, w) pairs column w of dz with the j-th
 * listed column of x), summed over the leading axis by the caller.  Wt [K][64][64] = the
 * planes with rows = the reduce index of dz's channels.  Two equivalent ways to call it:
 * (x, dz, transposed planes, column lists) or (dz, x, forward planes, row lists) -- the
 * second one lets whole (plane, 16-column) units be skipped, the row lists being emptier.
 * coef ([sum_k Lk][V] floats or NULL) only tells real slots (non-zero) from padding (zero): pass a
 * 0/1 table of the adjacency pattern rather than the current coefficients, since a real entry whose
 * coefficient is momentarily zero still has a gradient; with NULL every slot 0 and every later slot
 * holding a non-zero joint counts as real and no unit is skipped. */
int p2r_stgcn_gcn_coef_grad(int N, int T, int V, int K, const int *Lk_host,
                            const float *x, const float *dz, const float *Wt,
                            const uint8_t *nbr, const float *coef, int n_blocks,
                            float *dcoef_partial, void *stream);

/* ---- BatchNorm + residual + ReLU of st_gcn_block (stgcn_layers.py:399-439) -------- */

/* per-row partial statistics for the batch statistics: x viewed as [rows = N*C][L] ->
 * partial [rows][3] = (L, mean of the row, M2 = sum (x - mean)^2), taken about a pivot (the row's first
 * element) so that the variance survives |mean| >> std in fp32; p2r_bn_finalize (width 3) merges the rows
 * of a channel. */
int p2r_bn_stats(int rows, int L, const float *x, float *partial, void *stream);

/* y = relu?(x * scale[c] + shift[c] + res?)  over x (N,C,L); res may be NULL.
 * Replaces BatchNorm2d (normalise with given statistics) -> [+ residual] -> ReLU.
 * relu_mask, when not NULL (and relu != 0), receives one byte per element (y > 0): the
 * backward kernels then read 1 byte instead of the 4 bytes of y (relu mode 3 below). */
int p2r_bn_apply(int N, int C, int L, const float *x, const float *scale,
                 const float *shift, const float *res, int relu, float *y,
                 unsigned char *relu_mask, void *stream);

/* backward reductions: g = dy * mask, mask by `relu`: 0 none, 1 (y > 0), 2 recomputed
 * (x * mscale[c] + mshift[c] > 0), 3 = the byte map of p2r_bn_apply passed through the `y`
 * pointer; partial [N*C][2] = (sum g, sum g * xhat),
 * xhat = (x - mean[c]) * invstd[c]. */
int p2r_bn_bwd_reduce(int N, int C, int L, const float *dy, const float *y,
                      const float *x, const float *mean, const float *invstd,
                      int relu, const float *mscale, const float *mshift,
                      float *partial, void *stream);

/* dx = kscale[c] * (g - m1[c] - xhat * m2[c]); dres = g (dres may be NULL). */
int p2r_bn_bwd_apply(int N, int C, int L, const float *dy, const float *y,
                     const float *x, const float *mean, const float *invstd,
                     const float *kscale, const float *m1, const float *m2,
                     int relu, const float *mscale, const float *mshift,
                     float *dx, float *dres, void *stream);

/* statistics finalisation (one small launch instead of a dozen elementwise ones): partial [P][C][width];
 * width 3 = (count, mean, M2) entries from p2r_bn_stats (P = N) or from the gcn3 / tconv3 epilogues'
 * stats_partial, merged in fp64 as mean = sum n_p mean_p / n, M2 = sum [M2_p + n_p (mean_p - mean)^2];
 * width 2 = (sum, sum of squares) entries from the first- and second-generation conv epilogues;
 * M = elements per channel (width 3: the counts of the entries are used).  out [4][C] = mean, invstd = 1/sqrt(var + eps), scale = gamma*invstd,
 * shift = beta - mean*scale (biased variance, fp64 combination).  momentum >= 0 also updates
 * running_mean / running_var in place as nn.BatchNorm does (unbiased variance); momentum < 0 leaves
 * them untouched (they may be NULL then).  num_batches_tracked (device int64 scalar, may be NULL) is
 * incremented by one, as nn.BatchNorm's forward does in training mode. */
int p2r_bn_finalize(int P, int C, int width, const float *partial, double M, const float *gamma,
                    const float *beta, double eps, double momentum, float *running_mean,
                    float *running_var, long long *num_batches_tracked, float *out, void *stream);

/* backward counterpart: partial [P][C][2] = (sum g, sum g*xhat) rows -> out [4][C] =
 * sum g (= dbeta), sum g*xhat (= dgamma), (sum g)/M, (sum g*xhat)/M. */
int p2r_bn_bwd_finalize(int P, int C, const float *partial, double M, float *out, void *stream);

/* ---- temporal (3,1) convolution of st_gcn_block, BatchNorm+ReLU fused on the input ---- */

/* replaces tcn.0-tcn.2 (stgcn_layers.py:399-411): out[n,c,t,w] = bias[c] + sum_p sum_ci
 * W[p][c][ci] * h[n,ci,t+p-(taps-1)/2,w], h = relu(x*scale+shift) when scale != NULL else x,
 * zero outside [0,T).  x, out (N,64,T,V); W [taps][64][64]; scale/shift/bias [64] or NULL.
 * taps = 3 is the temporal convolution; taps = 1 the pointwise 64->64 Conv1d of the embedding
 * MLPs (stgcn.py:46-63, sub_modules.py SingleConv 'cbr') with the preceding BatchNorm+ReLU
 * fused the same way.  With the taps reversed and transposed it yields the data gradient.
 * stats_partial, when not NULL, receives [*n_partials][64][2] per-workgroup (sum, sum of
 * squares) of out per channel (batch statistics for the BatchNorm that follows);
 * *n_partials (when not NULL) returns the number of workgroups, also when out == NULL
 * (size query, nothing is launched). */
int p2r_stgcn_tconv_forward(int N, int T, int V, int taps, const float *x, const float *scale,
                            const float *shift, const float *W, const float *bias,
                            float *out, float *stats_partial, int *n_partials, void *stream);

/* weight gradient: dw_partial [n_blocks][64 c][64 ci][taps] (the layout of the Conv2d weight; summed by the caller) of
 * sum_{n,t,w} dout[n,c,t,w] * h[n,ci,t+p-(taps-1)/2,w] with h as above; dbias_partial,
 * when not NULL, [n_blocks][64] = per-workgroup sums of dout per channel (bias gradient). */
int p2r_stgcn_tconv_weight_grad(int N, int T, int V, int taps, const float *x, const float *scale,
                                const float *shift, const float *dout, int n_blocks,
                                float *dw_partial, float *dbias_partial, void *stream);

/* The same launch with the BatchNorm-backward apply pass of the convolution's INPUT BatchNorm riding on its tile
 * staging (V = 53, taps = 3 only; P2R_EINVAL otherwise).  Replaces p2r_bn_bwd_apply(relu = 2) + the launch above in
 * the backward of `tcn` = BatchNorm2d, ReLU, Conv2d((3,1)) (stgcn_layers.py:399-412):
 *   dz[n,c,t,w] = scale[c] * ((x * scale[c] + shift[c] > 0 ? dh : 0) - m1[c] - (x - mean[c]) * invstd[c] * m2[c])
 * dh = gradient w.r.t. the BatchNorm-ReLU output (the data gradient of the convolution); fin [4][64] = mean, invstd,
 * scale, shift (p2r_bn_finalize); m12 [2][64] = m1, m2 (rows 2, 3 of p2r_bn_bwd_finalize; zeros in evaluation mode).
 * dz must not alias x or dh. */
int p2r_stgcn_tconv_weight_grad_dz(int N, int T, int V, int taps, const float *x, const float *fin,
                                   const float *dout, const float *dh, const float *m12, float *dz, int n_blocks,
                                   float *dw_partial, float *dbias_partial, void *stream);

/* Second generation of p2r_stgcn_tconv_forward for V = 53, taps = 3 (temporal conv) or 1 (the pointwise
 * 64 -> 64 convolutions of the embedding MLPs; Wp [1][4][4][64][4]) (csrc/stgcn_tconv2.hip): MFMA n-tile = 16
 * frames of one joint, channel phases double-buffered in LDS by LDS-DMA, persistent workgroups, input transform
 * applied on the B operand.  Wp [3][4][4][64][4]: Wp[p][ph][m][16 g + r][s] = W[p][16 m + r][16 ph + 4 s + g].
 * scale / shift both NULL = no input transform (the data-gradient launch, with flipped / transposed taps).
 * stats_partial [*n_partials][64][2] (optional); out == NULL queries *n_partials.
 * bwd_z / bwd_fin (both or neither; data-gradient launch only, stats_partial required): the statistics epilogue
 * then emits the reduction pass of the BatchNorm + ReLU backward of the layer in front (what p2r_bn_bwd_reduce
 * with relu = 2 computes from `out`): per channel (sum g', sum g' * zhat), g' = out where scale*z + shift > 0,
 * zhat = (z - mean) * invstd, z = bwd_z (N,64,T,53), bwd_fin [4][64] = (mean, invstd, scale, shift). */
int p2r_stgcn_tconv2_forward(int N, int T, int V, int taps, const float *x, const float *scale, const float *shift,
                             const float *Wp, const float *bias, float *out, float *stats_partial,
                             int *n_partials, const float *bwd_z, const float *bwd_fin, void *stream);

/* Third generation of the same operator (csrc/stgcn_tconv3.hip): per-wave straight-line programs, in-place MFMA
 * blocks, precomputed DMA piece offsets (the recipe of p2r_stgcn_gcn3_forward).  Same arguments and results, except
 * that the forward statistics are [*n_partials][64][3] = (count, mean, M2) entries as p2r_stgcn_gcn3_forward writes
 * them; requires T % 16 == 0 and x, out, bwd_z 16-byte aligned (P2R_EINVAL otherwise: use p2r_stgcn_tconv2_forward). */
int p2r_stgcn_tconv3_forward(int N, int T, int V, int taps, const float *x, const float *scale, const float *shift,
                             const float *Wp, const float *bias, float *out, float *stats_partial,
                             int *n_partials, const float *bwd_z, const float *bwd_fin, void *stream);

/* single-tap p2r_stgcn_tconv3_forward (pointwise 64 -> 64 convolution behind a BatchNorm + ReLU) plus an addend
 * add_ct (N,64,T) broadcast over the joints: out[n,c,t,w] = bias[c] + add_ct[n,c,t] + sum_ci W[c][ci] relu(x scale + shift)
 * -- the last layer of the joint embedding and the position embedding's broadcast add (stgcn.py:126-130) in one pass.
 * Same size / alignment conditions as p2r_stgcn_tconv3_forward. */
int p2r_stgcn_tconv3_forward_add(int N, int T, int V, const float *x, const float *scale, const float *shift,
                                 const float *Wp, const float *bias, const float *add_ct, float *out, void *stream);

/* ---- first layer of the embedding MLPs: pointwise Conv1d(3 -> 64) ------------------------ */

/* pos_embed[0] / sk_feat[0] (stgcn.py:46-63): x (N,3,L), W [64][3], bias [64] or NULL ->
 * out (N,64,L) = W . x + bias. */
int p2r_embed3_forward(int N, int L, const float *x, const float *W, const float *bias,
                       float *out, void *stream);

/* the same forward + the batch statistics of its output: stats [1][64][3] = (count, mean, M2) per output channel in the
 * entry format of p2r_bn_finalize (width 3), derived from the first and second moments of the three INPUT rows (the
 * output is an affine map of them), i.e. without a pass over the (N,64,L) output.  scratch: N * ceil(L/1024) * 10 floats. */
int p2r_embed3_forward_stats(int N, int L, const float *x, const float *W, const float *bias,
                             float *out, float *scratch, float *stats, void *stream);

/* its weight / bias gradient: partial [N*64][4] = per (sample, channel) row
 * (sum dout*x0, sum dout*x1, sum dout*x2, sum dout); the caller sums over samples. */
int p2r_embed3_weight_grad(int N, int L, const float *x, const float *dout, float *partial,
                           void *stream);

/* p2r_embed3_weight_grad with the output gradient in its BatchNorm-backward form coef[0][c] g + coef[1][c] z + coef[2][c]
 * (g = masked gradient, z = the layer's saved output, both (N,64,L); coef [3][64]), formed while the tensors are read. */
int p2r_embed3_weight_grad_lazy(int N, int L, const float *x, const float *g, const float *z, const float *coef,
                                float *partial, void *stream);

/* One-pass backward of a pointwise 64 -> 64 layer of the embedding MLPs (stgcn.py:46-63) behind a BatchNorm + ReLU
 * (csrc/embed_bwd.hip): dz = coef ? coef[0] g + coef[1] z + coef[2] : g is the gradient of the layer's conv output
 * (g, z (N,64,L), coef [3][64]); the layer's input is relu(zp * fin[2] + fin[3]) (zp (N,64,L) = the previous layer's conv
 * output, fin [4][64] = mean, invstd, scale, shift of its BatchNorm); W [64][64] = the Conv1d weight [out][in].
 * Outputs: g_prev (N,64,L) = (W^T dz) where the input is positive else 0; sums [n_blocks][64][2] = per-workgroup
 * (sum g_prev, sum g_prev * (zp - mean) * invstd); dw_part [n_blocks][64][64] and db_part [n_blocks][64] (optional):
 * per-workgroup partials of dW [out][in] and of the bias gradient.  L % 64 == 0, tensors 16-byte aligned. */
int p2r_embed_layer_backward(int N, int L, const float *g, const float *z, const float *coef, const float *zp,
                             const float *fin, const float *W, float *g_prev, float *sums, int n_blocks,
                             float *dw_part, float *db_part, void *stream);

/* per-row column sums: x viewed as [rows][T][V] -> out_partial [rows][V] = sum over T
 * (gradient of the graph-conv bias table; the caller sums rows of a channel). */
int p2r_colsum(int rows, int T, int V, const float *x, float *out_partial, void *stream);

/* sum over the leading axis of per-workgroup partials: in [P][M] f32 -> out [M], in a fixed order (deterministic): P < 32
 * rows one after the other in two interleaved accumulators; P >= 32 sixteen strided partial sums (rows p = r, r + 16, ...)
 * added in the order r = 0..15.
 * M % 4 == 0, in / out 16-byte aligned.  tr64 != 0: M = G * 4096 and every 64 x 64 block is written transposed
 * (the graph-conv weight-gradient kernel produces dW_k^T).  What `partials.sum(0)` did in the autograd wrappers. */
int p2r_sum_leading(int P, long long M, const float *in, float *out, int tr64, void *stream);

/* ---- fused vote aggregation ------------------------------------------------------------ */

/* replaces the chain ball_query -> group_points(features) -> [Conv2d 1x1 + ReLU] x2 ->
 * max over nsample of PointnetSAModuleVotes.forward (pointnet2_modules.py:220-259 with
 * mlp=[256,256,256], bn=False, use_xyz=False, pooling='max') in one launch.
 * xyz (b,n,3), new_xyz (b,m,3), features (b,256,n), w1 (256,256), b1 (256), w2 (256,256),
 * b2 (256) -> idx (b,m,16) i32 (identical to p2r_ball_query), out (b,256,m) f32.
 * nsample must be 16 and C0 = C1 = C2 = 256.
 * Training: G, H (b,256,m,16) f32 and amax (b,256,m) u8 non-NULL -> the grouped features, the
 * hidden activation and the arg-max sample of every (channel, ball) are saved for
 * p2r_sa_votes_backward; all three NULL for inference. */
int p2r_sa_votes_forward(int b, int n, int m, int nsample, float radius, int C0,
                         int C1, int C2, const float *xyz, const float *new_xyz,
                         const float *features, const float *w1, const float *b1,
                         const float *w2, const float *b2, int *idx, float *out,
                         float *G, float *H, unsigned char *amax, void *stream);

/* autograd of the above (max-pool -> ReLU -> conv -> ReLU -> conv): dout, out (b,256,m);
 * amax, H from the training forward; w2t / w1t (256,256) = transposed layer weights.
 * Writes dZ2, dZ1 (pre-activation gradients of the two layers) and dG (gradient of the
 * grouped features), all (b,256,m,16); dG goes to p2r_group_points_grad, (dZ2, H) and
 * (dZ1, G) to p2r_gemm_nt_256 for the weight gradients. */
int p2r_sa_votes_backward(int b, int m, int nsample, int C, const float *dout, const float *out,
                          const unsigned char *amax, const float *H, const float *w2t,
                          const float *w1t, float *dZ2, float *dZ1, float *dG, void *stream);

/* weight gradient of a 256 -> 256 pointwise convolution: A, B (nb,256,L) ->
 * partial [split][256][256], whose sum over the leading axis is sum_b A[b] . B[b]^T
 * (split-K MFMA product, deterministic). */
int p2r_gemm_nt_256(int nb, int L, int split, const float *A, const float *B, float *partial,
                    void *stream);


/* ---- point-wise (kernel size 1) convolution stacks of the vote / proposal heads ------------------
 *
 * replaces the `SingleConv` chains ('cbr' = Conv1d(k=1, no bias) -> BatchNorm1d -> ReLU, 'c' = Conv1d(k=1) with bias) of
 * CenterVoteModule.conv_input (models/p2rnet/modules/vote_center.py:28-48), ProposalNet.conv_center / conv_heading /
 * conv_size / conv_sem_obj (proposal_net.py:77-95,183-191) and the mixture heads' backbone + pi convolutions
 * (mdn.py:20-27,141-161) -- in the reference one cuDNN conv + one BatchNorm + one ReLU launch per layer forward and
 * five or more backward.  Here a layer is ONE launch each way, and independent layers of equal depth (the four
 * stems, the three mixture heads) share a launch as a JOB LIST.  Activations are (B, C, L) fp32 ("NCL", the Conv1d
 * layout) or (B, L, C) ("NLC", what `transpose(1, 2)` of it looks like in memory); L % 64 == 0.
 *
 * The BatchNorm + ReLU of layer i is never materialised: layer i+1 applies it to its input tile as it stages the
 * tile into LDS (forward: relu(z * scale + shift); backward: the gradient of the pre-BatchNorm output is formed the
 * same way as a*g + b*z + c from the stored masked gradient g, the stored conv output z and three per-channel
 * constants).  Batch statistics leave the forward kernel as (count, mean, M2) entries per 64-column tile, the two
 * BatchNorm-backward sums leave the data-gradient kernel per tile; one small multi-job launch finalises either.
 *
 * All structs are plain C; every pointer is a device pointer; arrays of jobs are HOST arrays (copied into the kernel
 * arguments), at most P2R_PW_MAX_JOBS per call. */
#define P2R_PW_MAX_JOBS 8

/* out[rows x cols] = op(W)[rows x k] . T(x)[k x cols]  (+ epilogue), one job of p2r_pw_gemm.
 *   x, x2 : the k-row input.  T = identity (tr == NULL), relu(x * tr[0][c] + tr[1][c]) (tr_mode 1: forward through a
 *           BatchNorm + ReLU), or tr[0][c] * x + tr[1][c] * x2 + tr[2][c] (tr_mode 2: the BatchNorm-backward form;
 *           x = masked gradient, x2 = saved conv output).  tr rows are `tr_ld` floats apart.
 *           x_ctot = channels of the tensor x (and x2) points INTO (the pointer is already offset to this job's
 *           first channel); x_nlc != 0: (B, L, x_ctot) memory, k % 4 == 0 then.
 *   w     : w_t == 0: [rows][k] row-major (a Conv1d weight, forward); w_t != 0: [k][rows] (the same weight read
 *           transposed: data gradient).
 *   bias  : [rows] or NULL.
 *   out   : NCL (out_nlc == 0) or NLC, out_ctot channels in the tensor, pointer pre-offset.
 *   epilogue 0: out = acc (+ bias).  stats != NULL: (count, mean, M2) of every output row over the 64-column tile
 *               -> stats [cols/64][rows][3].
 *   epilogue 1 (data gradient through the ReLU + BatchNorm in front): m = mz * mfin[2][r] + mfin[3][r] > 0,
 *               out = m ? acc : 0; stats [cols/64][rows][2] = (sum out, sum out * (mz - mfin[0][r]) * mfin[1][r]).
 *               mz: NCL with mz_ctot channels; mfin rows `mfin_ld` floats apart. */
typedef struct p2r_pw_job {
  const float *x, *x2, *tr, *w, *bias, *mz, *mfin;
  float *out, *stats;
  int k, rows, x_ctot, x_nlc, tr_mode, tr_ld, w_t, out_ctot, out_nlc, epilogue, mz_ctot, mfin_ld;
} p2r_pw_job;

/* njobs jobs over the same B x L columns; one workgroup per (job, 64-column tile). */
int p2r_pw_gemm(int njobs, const p2r_pw_job *jobs, int B, int L, void *stream);

/* weight (+ bias) gradient of one layer: dw_part [split][rows][k] partials of  sum_cols T(dz)[rows] * U(y)[k]^T,
 * db_part [split][rows] partials of the row sums of T(dz) (or NULL).  dz side: x / x2 / tr / tr_mode as above
 * (tr_mode 0 or 2); y side: y, ytr [2][k] (NULL or relu(y * ytr[0] + ytr[1])), rows `ytr_ld` apart. */
typedef struct p2r_pw_wjob {
  const float *x, *x2, *tr, *y, *ytr;
  float *dw_part, *db_part;
  int rows, k, x_ctot, x_nlc, tr_mode, tr_ld, y_ctot, y_nlc, ytr_ld, split;
} p2r_pw_wjob;
int p2r_pw_wgrad(int njobs, const p2r_pw_wjob *jobs, int B, int L, void *stream);

/* BatchNorm1d statistics of a job list.  part [P][C][3] (count, mean, M2) from p2r_pw_gemm, or NULL = evaluation
 * mode (running statistics).  fin: mean, invstd, scale = gamma * invstd, shift = beta - mean * scale at
 * fin[i * fin_ld + c].  Training (part != NULL, momentum >= 0): running_mean / running_var / num_batches_tracked
 * updated as nn.BatchNorm1d does (unbiased variance). */
typedef struct p2r_pw_bnjob {
  const float *part, *gamma, *beta;
  float *running_mean, *running_var, *fin;
  long long *num_batches_tracked;
  double eps, momentum;
  int P, C, fin_ld;
} p2r_pw_bnjob;
int p2r_pw_bn_finalize(int njobs, const p2r_pw_bnjob *jobs, void *stream);

/* BatchNorm1d backward constants.  part [P][C][2] = (sum g, sum g * xhat) from p2r_pw_gemm (epilogue 1);
 * dgamma = sum g * xhat, dbeta = sum g; coef (rows coef_ld apart) = a, b, c with  dz = a*g + b*z + c:
 * training  a = scale, b = -scale * invstd * m2, c = -scale * m1 + scale * invstd * m2 * mean  (m = sums / M);
 * evaluation-mode BatchNorm (train == 0)  a = scale, b = c = 0. */
typedef struct p2r_pw_bnbjob {
  const float *part, *fin;
  float *coef, *dgamma, *dbeta;
  double M;
  int P, C, fin_ld, coef_ld, train;
} p2r_pw_bnbjob;
int p2r_pw_bn_bwd_finalize(int njobs, const p2r_pw_bnbjob *jobs, void *stream);

/* out[j] = sum_p in[p * M + j]  for every job (partials of p2r_pw_wgrad and the like), pairwise-ordered, deterministic. */
typedef struct p2r_pw_rjob {
  const float *in;
  float *out;
  int P, M;
} p2r_pw_rjob;
#define P2R_PW_MAX_RJOBS 48
int p2r_pw_reduce(int njobs, const p2r_pw_rjob *jobs, void *stream);

/* Mixture-density read-out (mdn.py:34-83, MixtureDensityHead.generate_point_predictions with n_samples = 1 and
 * the mixture weights as gates), up to four heads per launch:
 *   pred[b, l, d] = sum_g sigmoid(logit[b, g, l]) * (mu[g, d] + exp(log_sigma[g, d]) * eps[(b * L + l), g, d]).
 * logit (B, G, L) f32 inside a tensor of logit_ctot channels (pointer pre-offset); mu / eps / pred / dpred / dmu are f32
 * (f64 == 0) or f64 (the heading head, whose `mu` is a float64 parameter); log_sigma, dlog_sigma f32; eps == NULL:
 * the mixture mean (get_mean, mdn.py:85-99).  pi (B, G, L) f32 optional output.  D <= 4.
 * Backward: dpred (B, L, D) -> dlogit (B, G, L) f32 (inside a tensor of dlogit_ctot channels), dmu [G][D],
 * dlog_sigma [G][D]; one workgroup per (head, component). */
typedef struct p2r_mix_head {
  const float *logit, *log_sigma;
  const void *mu, *eps, *dpred;
  void *pred, *dmu;
  float *pi, *dlogit, *dlog_sigma;
  int D, f64;
} p2r_mix_head;
#define P2R_MIX_MAX_HEADS 4
int p2r_mdn_mix_forward(int nheads, const p2r_mix_head *heads, int B, int G, int L, int logit_ctot, void *stream);
int p2r_mdn_mix_backward(int nheads, const p2r_mix_head *heads, int B, int G, int L, int logit_ctot, int dlogit_ctot,
                         void *stream);

/* ---- seams of the ST-GCN backbone (csrc/seed_ops.hip) ------------------------------------------------------------ */

/* frame gather in front of conv_joint (stgcn.py:142-149; conv_joint is pointwise in time, so the gather may come
 * first): x (b,c,t,j) f32, inds (b,s) int64 -> out (b,s,c*j), out[b,s,ci*j+ji] = x[b,ci,inds[b,s],ji]. */
int p2r_gather_frames(int b, int c, int t, int j, int s, const float *x, const long long *inds, float *out,
                      void *stream);
/* its gradient: dout (b,s,c*j) -> dx (b,c,t,j), every element written (zeros for frames no seed picked; rows of
 * seeds that share a frame are added in seed order). */
int p2r_gather_frames_grad(int b, int c, int t, int j, int s, const float *dout, const long long *inds, float *dx,
                           void *stream);
/* out[r] = scale * sum_v x[r*v_len + v], v_len <= 64 (mean over the 20-frame window of the position embedding,
 * stgcn.py:118-121; sum over the 53 joints = the gradient of its broadcast add, stgcn.py:129-130). */
int p2r_rowsum_short(long long rows, int v_len, float scale, const float *x, float *out, void *stream);

/* tail of the vote head (vote_center.py:50-58 + the feature normalisation of network.py:68-71) in one launch:
 * net (b,s,3+256) = conv_input's output per seed, seed_features (b,s,256), hip (b,s,3) -> vote_xyz (b,s,3) = hip + net[:3],
 * feat_ncl (b,256,s) = (seed_features + net[3:]) / |.|_2 channel-major (what the vote aggregation reads; the reference's
 * (b,s,256) tensor is its transposed view), inv_norm (b,s).  s % 64 == 0. */
int p2r_vote_finish(int b, int s, int C, const float *net, const float *seed_features, const float *hip,
                    float *vote_xyz, float *feat_ncl, float *inv_norm, void *stream);
/* gradient: d_xyz (b,s,3) / d_feat (b,256,s) (either may be NULL = zero) -> d_net (b,s,3+256), d_sf (b,s,256). */
int p2r_vote_finish_grad(int b, int s, int C, const float *d_xyz, const float *d_feat, const float *feat_ncl,
                         const float *inv_norm, float *d_net, float *d_sf, void *stream);

/* seed selection by arc length (stgcn.py:96-101): inds[b,s] = the FIRST t minimising |cum[b,t] - target[b,s]| (fp32, the
 * expression of `torch.argmin(torch.abs(cum.unsqueeze(-1) - target.unsqueeze(1)), dim=1)` without its (b,t,s) tensor).
 * cum (b,t), target (b,s) f32; inds (b,s) int64; t <= 40000. */
int p2r_nearest_prefix(int b, int t, int s, const float *cum, const float *target, long long *inds, void *stream);

/* ---- opt-in `split16` arithmetic of the ST-GCN blocks (csrc/split16.h) -----------------------------------------------
 * Same operators as the exact-fp32 entry points above (reference models/p2rnet/modules/stgcn_layers.py:50-67,399-439 and
 * their autograd), every fp32 product formed as three v_mfma_f32_16x16x32_f16 products of two-part fp16 operands with
 * fp32 accumulation.  Operand tensors are lifted into fp16's range by a power of two derived ON THE DEVICE from a range
 * word: `*_amax` arguments point at one uint32 = the float bits of max |x| over the tensor (NULL: scale 1).  Weights
 * arrive pre-split (fp16 planes in the kernel's lane order) with the inverse of their power-of-two scale as a device
 * float.  The default (exact) mode never calls these. */

/* range word of a tensor: *amax_bits = float bits of max |x[i]|, i < n (one read of x). */
int p2r_absmax_bits(long long n, const float *x, unsigned *amax_bits, void *stream);

/* p2r_bn_bwd_apply in the chain's form (ReLU mask BYTES, `relu` = 3) that also leaves the range word of dx. */
int p2r_bn_bwd_apply_amax(int N, int C, int L, const float *dy, const unsigned char *mask, const float *x,
                          const float *mean, const float *invstd, const float *kscale, const float *m1,
                          const float *m2, float *dx, float *dres, unsigned *amax_bits, void *stream);
/* p2r_bn_apply with ReLU (res / mask optional) that also leaves the range word of y. */
int p2r_bn_apply_amax(int N, int C, int L, const float *x, const float *scale, const float *shift, const float *res,
                      float *y, unsigned char *mask, unsigned *amax_bits, void *stream);
/* p2r_stgcn_tconv_weight_grad_dz that also leaves the range word of dz. */
int p2r_stgcn_tconv_weight_grad_dz_amax(int N, int T, int V, int taps, const float *x, const float *fin,
                                        const float *dout, const float *dh, const float *m12, float *dz, int n_blocks,
                                        float *dw_partial, float *dbias_partial, unsigned *amax_bits, void *stream);

/* The (3,1) temporal convolution of p2r_stgcn_tconv3_forward (taps = 3, V = 53, T % 16 == 0) in split16 arithmetic
 * (csrc/stgcn_tconvh.hip).  x, scale, shift, bias, out, bwd_z, bwd_fin as there.
 *   Wh    fp16 [3 parts][3 taps][2][4][64 lanes][8]: the parts (w1 = fp16(w), w2 = fp16(w - w1), 2^-11 w1) of
 *         w = 2^S_w W[tap][co][ci] in A-operand order,
 *         Wh[part][tap][ks][w][16 kg + r][i] = part of 2^S_w W[tap][16 w + r][32 ks + 8 kg + i]   (16-byte aligned)
 *   winv  device float: 2^-S_w
 *   x_amax range word of x (the data gradient's incoming gradient) or NULL (forward: the activation relu(x*scale+shift)
 *         is used at scale 1, clamped to fp16's largest finite value)
 *   stats_partial [*n_partials][64][3] = (count, mean, M2) per workgroup and channel (forward) or [*n_partials][64][2]
 *         (bwd_z / bwd_fin given: the two sums of the BatchNorm + ReLU backward); *n_partials = N * T / chunk with
 *         chunk = the longest of 64 / 32 / 16 frames dividing T; out == NULL queries it. */
int p2r_stgcn_tconvh_forward(int N, int T, int V, const float *x, const float *scale, const float *shift,
                             const void *Wh, const float *winv, const float *bias, float *out, float *stats_partial,
                             int *n_partials, const float *bwd_z, const float *bwd_fin, const unsigned *x_amax,
                             void *stream);

/* The fused graph convolution of p2r_stgcn_gcn3_forward in split16 arithmetic (csrc/stgcn_gcn3h_body.h): statically
 * scheduled for the P2RNet skeleton like the third generation, with plane PAIRS as the MFMA's K dimension (16 channels x
 * two planes).  p2r_stgcn_gcn3h_signature(form) = signature of the neighbour-table pattern the schedule of form 0
 * (column lists, forward) / 1 (row lists, data gradient) was generated for; p2r_stgcn_gcn3h_pairs(form, out[2 * 6])
 * returns the number of pairs and writes them (second plane -1 = none).  T % 16 == 0, tensors 16-byte aligned.
 *   Wh    fp16 [6 pairs][4 phases][3 parts][4 m][64 lanes][8]: the parts (w1, w2, 2^-11 w1) of 2^S_w [W_a | W_b] in
 *         A-operand order,
 *         Wh[pair][ph][part][m][16 kg + r][i] = part of 2^S_w W_{plane (i < 4 ? a : b)}[16 m + r][16 ph + kg + 4 (i & 3)]
 *         (forward: W_k [c][ci]; data gradient: W_k^T)
 *   winv  device float 2^-S_w;  coef [ltot][53] f32 as for p2r_stgcn_gcn3_forward (the kernel scales it by 2^S_x)
 *   x_amax / dz_amax: range word of the operand tensor (NULL: scale 1)
 *   stats_partial [*n_partials][64][3] (optional) as p2r_stgcn_gcn3_forward's forward statistics.
 * The data gradient adds `addend` (NULL: nothing) where the bytes of `addend_mask` are non-zero (NULL: everywhere). */
unsigned long long p2r_stgcn_gcn3h_signature(int form);
int p2r_stgcn_gcn3h_pairs(int form, int *pairs);
int p2r_stgcn_gcn3h_forward(int N, int T, int V, int K, int ltot, const float *x, const void *Wh, const float *winv,
                            const float *coef, const float *bias_cv, float *z, float *stats_partial, int *n_partials,
                            const unsigned *x_amax, void *stream);
int p2r_stgcn_gcn3h_data_gradient(int N, int T, int V, int K, int ltot, const float *dz, const void *Wh,
                                  const float *winv, const float *coef, const float *addend,
                                  const unsigned char *addend_mask, float *dx, const unsigned *dz_amax, void *stream);

/* Weight gradient of the graph convolution in split16 arithmetic (csrc/stgcn_gcn3dwh.hip): the operator and the outputs of
 * p2r_stgcn_gcn3_weight_grad -- dw_partial [n_blocks][K][64 ci][64 c] (dW_k transposed) and, optionally, dbias_partial
 * [n_blocks][64][53] (column sums of dz: the bias-table gradient), both summed over the leading axis by the caller -- with
 * K = 32 = (frame of a 4-frame tile) x (8 joints of a group) per MFMA.  coef [ltot][53]: the row-form coefficient table.
 * T % 4 == 0, x / dz 16-byte aligned; x_amax / dz_amax: range words of the two operand tensors (NULL: scale 1).
 * p2r_stgcn_gcn3h_weight_grad_signature() = signature of the row-form neighbour tables the schedule was generated for. */
unsigned long long p2r_stgcn_gcn3h_weight_grad_signature(void);
int p2r_stgcn_gcn3h_weight_grad(int N, int T, int V, int K, int ltot, const float *x, const float *dz, const float *coef,
                                int n_blocks, float *dw_partial, float *dbias_partial, const unsigned *x_amax,
                                const unsigned *dz_amax, void *stream);

/* Adjacency gradient of the graph convolution in split16 arithmetic (csrc/stgcn_gcn3h_grad.hip): arguments and result of
 * p2r_stgcn_gcn3_coef_grad; the product Y_k = W_k . x runs on two-part fp16 operands, its reduction against the gathered
 * rows of dz stays fp32 vector arithmetic (dz needs no range word).
 *   Wd    fp16 [K][4 ph][2 parts][2 ks][64 lanes][8]: the parts (w1, w2) of 2^S_w W_k (forward planes) in A-operand order,
 *         Wd[k][ph][part][ks][16 kg + r][i] = part of 2^S_w W_k[16 ph + r][32 ks + 16 (i >> 2) + 4 (i & 3) + kg]
 *   winv  device float 2^-S_w;  x_amax: range word of x (NULL: scale 1).  T % 16 == 0; x, dz, Wd 16-byte aligned. */
int p2r_stgcn_gcn3h_coef_grad(int N, int T, int V, int K, int ltot, const float *x, const float *dz, const void *Wd,
                              const float *winv, int n_blocks, float *dcoef_partial, const unsigned *x_amax,
                              void *stream);

#ifdef __cplusplus
}
#endif
#endif /* P2R_HIP_H */
