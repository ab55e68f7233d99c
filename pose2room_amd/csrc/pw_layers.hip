// pw_layers.hip -- point-wise convolution stacks of the vote / proposal heads (gfx950).
//
// Replaces the SingleConv chains of CenterVoteModule.conv_input (reference models/p2rnet/modules/vote_center.py:28-48),
// ProposalNet.conv_center / conv_heading / conv_size / conv_sem_obj (proposal_net.py:77-95,183-191) and the mixture
// heads (mdn.py:20-27,34-99,141-161): there every layer is a cuDNN convolution, a BatchNorm and a ReLU launch forward
// and five or more launches backward, on tensors of 2-16 MB -- ~330 launches of 4-10 us per train step.
//
// MI355X design (include/p2r_hip.h, "point-wise convolution stacks"):
//   * a layer = one launch each way; independent layers of equal depth ride in one launch as a job list;
//   * a workgroup owns 64 columns (positions) of one job: the k x 64 input tile is staged into LDS once, with the
//     BatchNorm + ReLU of the layer in front applied on the way in (forward) or the BatchNorm-backward affine form
//     a*g + b*z + c (backward), so normalised activations and pre-BatchNorm gradients never exist in HBM;
//   * the product runs on v_mfma_f32_16x16x4_f32 (exact fp32 accumulate): waves split the output rows, weights stream
//     from L2 as A operands (row-major for the forward, the same tensor read transposed for the data gradient),
//     software-pipelined two groups of 64 k ahead; B operands come from the LDS tile (row stride 68: the four k rows a
//     wave reads per step fall on two bank groups, two lanes per bank = the rate of a 64-lane ds_read_b32);
//   * batch statistics ((count, mean, M2) per 64-column tile) and the two BatchNorm-backward sums leave through the
//     epilogue; pw_bn_finalize / pw_bn_bwd_finalize merge them in fp64 (one wave per channel, all jobs in one launch);
//   * weight gradients are a split-K product over the columns with both operands transformed as they are staged
//     (pw_wgrad_kernel), partials summed by pw_reduce_kernel for all layers of a head at once (pairwise order);
//   * the mixture read-out (sigmoid, sampling, gate, sum over components) is one launch each way (mdn_mix_*).
// These tensors are L2-resident; the kernels are bound by launch latency and by the MFMA rate of the few CUs a
// 4096-16384 column problem can occupy, not by HBM.
#include "pw_mma.h"

namespace {

constexpr int PW_COLS = 64;       // columns per workgroup
constexpr int PW_KMAX = 560;      // largest (padded) reduction length: 560 * 68 * 4 = 152,320 bytes of LDS

struct PwJobs { p2r_pw_job j[P2R_PW_MAX_JOBS]; };

__global__ __launch_bounds__(256) void pw_gemm_kernel(PwJobs jobs, int L, int colblocks) {
  extern __shared__ float tile[];                    // [kpad][PW_RS]
  const p2r_pw_job &J = jobs.j[blockIdx.x / colblocks];
  const int cb = blockIdx.x % colblocks;
  const int col0 = cb * PW_COLS;
  const int b = col0 / L, l0 = col0 - b * L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int K = J.k, kpad = (K + 15) & ~15, rows = J.rows;

  // ---- stage T(x) [k][64 columns] ---------------------------------------------------------------------------------
  if (!J.x_nlc) {
    const int c4 = tid & 15;
    const size_t base = (size_t)b * J.x_ctot * L + l0 + 4 * c4;
    for (int row = tid >> 4; row < kpad; row += 16) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < K) {
        v = *reinterpret_cast<const float4 *>(J.x + base + (size_t)row * L);
        if (J.tr_mode == 1) {
          const float s = J.tr[row], t = J.tr[J.tr_ld + row];
          v.x = fmaxf(v.x * s + t, 0.f); v.y = fmaxf(v.y * s + t, 0.f);
          v.z = fmaxf(v.z * s + t, 0.f); v.w = fmaxf(v.w * s + t, 0.f);
        } else if (J.tr_mode == 2) {
          const float4 z = *reinterpret_cast<const float4 *>(J.x2 + base + (size_t)row * L);
          const float a = J.tr[row], bq = J.tr[J.tr_ld + row], c = J.tr[2 * J.tr_ld + row];
          v.x = a * v.x + bq * z.x + c; v.y = a * v.y + bq * z.y + c;
          v.z = a * v.z + bq * z.z + c; v.w = a * v.w + bq * z.w + c;
        }
      }
      *reinterpret_cast<float4 *>(tile + row * PW_RS + 4 * c4) = v;
    }
  } else if ((K & 3) != 0 || (J.x_ctot & 3) != 0) {   // (B, L, C) with rows that are not 16-byte multiples: scalar loads
    const int col = tid & 63;
    const size_t base = ((size_t)b * L + l0 + col) * J.x_ctot;
    for (int row = tid >> 6; row < kpad; row += 4) {
      float v = 0.f;
      if (row < K) {
        v = J.x[base + row];
        if (J.tr_mode == 1) v = fmaxf(v * J.tr[row] + J.tr[J.tr_ld + row], 0.f);
        else if (J.tr_mode == 2) v = J.tr[row] * v + J.tr[J.tr_ld + row] * J.x2[base + row] + J.tr[2 * J.tr_ld + row];
      }
      tile[row * PW_RS + col] = v;
    }
  } else {                                           // (B, L, C): lane = column, 16 bytes of channels per load
    const int col = tid & 63;
    const size_t base = ((size_t)b * L + l0 + col) * J.x_ctot;
    for (int c4 = tid >> 6; c4 < (kpad >> 2); c4 += 4) {
      const int row = 4 * c4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (row < K) {
        const float4 u = *reinterpret_cast<const float4 *>(J.x + base + row);
        v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
        if (J.tr_mode == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * J.tr[row + e] + J.tr[J.tr_ld + row + e], 0.f);
        } else if (J.tr_mode == 2) {
          const float4 z4 = *reinterpret_cast<const float4 *>(J.x2 + base + row);
          const float z[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = J.tr[row + e] * v[e] + J.tr[J.tr_ld + row + e] * z[e] + J.tr[2 * J.tr_ld + row + e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[(row + e) * PW_RS + col] = v[e];
    }
  }
  __syncthreads();

  // ---- product: wave w owns a contiguous run of 16-row tiles, four at a time -----------------------------------------
  const int mtiles = (rows + 15) >> 4;
  const int per = (mtiles + 3) >> 2;
  const int lo = wave * per, hi = min(lo + per, mtiles);
  const int nch = kpad >> 4;
  const bool nlc_vec = J.out_nlc && (rows & 3) == 0 && (J.out_ctot & 3) == 0 && (((size_t)J.out) & 15) == 0;
  for (int mt0 = lo; mt0 < hi; mt0 += 4) {
    const int nmt = min(4, hi - mt0);
    floatx4v acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = floatx4v{0.f, 0.f, 0.f, 0.f};
    int rowm[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) rowm[m] = min(16 * (mt0 + m) + r, rows - 1);
    if (J.w_t) pw_product<true>(J.w, tile, K, rows, nch, nmt, rowm, g, r, acc);
    else pw_product<false>(J.w, tile, K, rows, nch, nmt, rowm, g, r, acc);

    // ---- epilogue: lane (g, r) holds rows 16 (mt0 + m) + 4 g + q, columns 16 n + r -------------------------------------
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m >= nmt) continue;
      const int row0 = 16 * (mt0 + m) + 4 * g;
      float v[4][4];                                  // [q][n]
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = row0 + q;
        const bool ok = row < rows;
        if (J.epilogue == 0) {
          const float bb = (J.bias && ok) ? J.bias[row] : 0.f;
#pragma unroll
          for (int n = 0; n < 4; ++n) v[q][n] = acc[m][n][q] + bb;
          if (J.stats) {                              // every lane of the 16-lane row takes part
            float s = (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
            s = p2r_row16_sum(s);
            const float mean = s * (1.f / 64.f);
            float d2 = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) { const float d = v[q][n] - mean; d2 += d * d; }
            d2 = p2r_row16_sum(d2);
            if (r == 0 && ok) {
              float *e = J.stats + ((size_t)cb * rows + row) * 3;
              e[0] = 64.f; e[1] = mean; e[2] = d2;
            }
          }
        } else {
          float sc = 0.f, sh = 0.f, mean = 0.f, invstd = 0.f;
          if (ok) {
            mean = J.mfin[row]; invstd = J.mfin[J.mfin_ld + row];
            sc = J.mfin[2 * J.mfin_ld + row]; sh = J.mfin[3 * J.mfin_ld + row];
          }
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            float z = 0.f;
            if (ok) z = J.mz[((size_t)b * J.mz_ctot + row) * L + l0 + 16 * n + r];
            const float gv = (z * sc + sh > 0.f) ? acc[m][n][q] : 0.f;
            v[q][n] = gv;
            s1 += gv;
            s2 += gv * ((z - mean) * invstd);
          }
          if (J.stats) {
            s1 = p2r_row16_sum(s1);
            s2 = p2r_row16_sum(s2);
            if (r == 0 && ok) {
              float *e = J.stats + ((size_t)cb * rows + row) * 2;
              e[0] = s1; e[1] = s2;
            }
          }
        }
      }
      if (!J.out_nlc) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (row0 + q < rows) {
            float *o = J.out + ((size_t)b * J.out_ctot + row0 + q) * L + l0 + r;
#pragma unroll
            for (int n = 0; n < 4; ++n) o[16 * n] = v[q][n];
          }
      } else {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          float *o = J.out + ((size_t)b * L + l0 + 16 * n + r) * J.out_ctot + row0;
          if (nlc_vec) {
            if (row0 < rows) *reinterpret_cast<float4 *>(o) = make_float4(v[0][n], v[1][n], v[2][n], v[3][n]);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (row0 + q < rows) o[q] = v[q][n];
          }
        }
      }
    }
  }
}

// ---- weight gradient: split-K product over the columns, both operands transformed while staged ------------------------
struct PwWJobs { p2r_pw_wjob j[P2R_PW_MAX_JOBS]; int blk0[P2R_PW_MAX_JOBS + 1]; };
constexpr int PWW_RS = 68;

// 64 rows x 64 columns of an operand into registers: thread -> 4 x float4.  NCL: c4 = tid & 15, rows (tid >> 4) + 16 i;
// NLC: column = tid & 63, channel quads (tid >> 6) + 4 i.
__device__ __forceinline__ void pww_fetch(const float *__restrict__ p, int ctot, int nlc, int L, int b, int l0, int row_lo,
                                          int nrows, int tid, float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!nlc) {
      const int row = row_lo + (tid >> 4) + 16 * i;
      if (row < nrows) v[i] = *reinterpret_cast<const float4 *>(p + ((size_t)b * ctot + row) * L + l0 + 4 * (tid & 15));
    } else {
      const int row = row_lo + 4 * ((tid >> 6) + 4 * i);
      const float *q = p + ((size_t)b * L + l0 + (tid & 63)) * ctot;
      if (((nrows | ctot) & 3) == 0) {
        if (row < nrows) v[i] = *reinterpret_cast<const float4 *>(q + row);
      } else {
        if (row + 0 < nrows) v[i].x = q[row + 0];
        if (row + 1 < nrows) v[i].y = q[row + 1];
        if (row + 2 < nrows) v[i].z = q[row + 2];
        if (row + 3 < nrows) v[i].w = q[row + 3];
      }
    }
  }
}

__global__ __launch_bounds__(256) void pw_wgrad_kernel(PwWJobs jobs, int L, int chunks) {
  __shared__ float as[64 * PWW_RS], bs[64 * PWW_RS];
  int ji = 0;
#pragma unroll
  for (int i = 1; i < P2R_PW_MAX_JOBS; ++i)
    if ((int)blockIdx.x >= jobs.blk0[i]) ji = i;
  const p2r_pw_wjob &J = jobs.j[ji];
  const int local = blockIdx.x - jobs.blk0[ji];
  const int rows = J.rows, K = J.k;
  const int tiles_m = (rows + 63) >> 6, tiles_n = (K + 63) >> 6;
  const int sp = local / (tiles_m * tiles_n), t = local % (tiles_m * tiles_n);
  const int tm = t / tiles_n, tn = t % tiles_n;
  const int cps = (chunks + J.split - 1) / J.split;
  const int c_lo = sp * cps, c_hi = min(c_lo + cps, chunks);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int row_a = 64 * tm, row_b = 64 * tn;

  floatx4v acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = floatx4v{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool want_b = J.db_part != nullptr && tn == 0;

  float4 ra[4], ra2[4], rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ra2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c_lo < c_hi) {
    const int col0 = c_lo * 64, b = col0 / L, l0 = col0 - b * L;
    pww_fetch(J.x, J.x_ctot, J.x_nlc, L, b, l0, row_a, rows, tid, ra);
    if (J.tr_mode == 2) pww_fetch(J.x2, J.x_ctot, J.x_nlc, L, b, l0, row_a, rows, tid, ra2);
    pww_fetch(J.y, J.y_ctot, J.y_nlc, L, b, l0, row_b, K, tid, rb);
  }
  for (int c = c_lo; c < c_hi; ++c) {
    __syncthreads();
    // registers -> LDS, transforms applied here (NCL: one 16-byte store per register; NLC: four column-major scalars)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float va[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      const float vz[4] = {ra2[i].x, ra2[i].y, ra2[i].z, ra2[i].w};
      const float vb[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
      if (!J.x_nlc) {
        const int lr = (tid >> 4) + 16 * i, row = row_a + lr;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < rows) {
          if (J.tr_mode == 2) {
            const float a = J.tr[row], bq = J.tr[J.tr_ld + row], cc = J.tr[2 * J.tr_ld + row];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = a * va[e] + bq * vz[e] + cc;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = va[e];
          }
        }
        *reinterpret_cast<float4 *>(as + lr * PWW_RS + 4 * (tid & 15)) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int lr = 4 * ((tid >> 6) + 4 * i) + e, row = row_a + lr;
          float v = 0.f;
          if (row < rows) v = J.tr_mode == 2 ? J.tr[row] * va[e] + J.tr[J.tr_ld + row] * vz[e] + J.tr[2 * J.tr_ld + row] : va[e];
          as[lr * PWW_RS + (tid & 63)] = v;
        }
      }
      if (!J.y_nlc) {
        const int lr = (tid >> 4) + 16 * i, row = row_b + lr;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < K) {
          if (J.ytr) {
            const float sc = J.ytr[row], sh = J.ytr[J.ytr_ld + row];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(vb[e] * sc + sh, 0.f);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = vb[e];
          }
        }
        *reinterpret_cast<float4 *>(bs + lr * PWW_RS + 4 * (tid & 15)) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int lr = 4 * ((tid >> 6) + 4 * i) + e, row = row_b + lr;
          float v = 0.f;
          if (row < K) v = J.ytr ? fmaxf(vb[e] * J.ytr[row] + J.ytr[J.ytr_ld + row], 0.f) : vb[e];
          bs[lr * PWW_RS + (tid & 63)] = v;
        }
      }
    }
    __syncthreads();
    if (c + 1 < c_hi) {
      const int col0 = (c + 1) * 64, b = col0 / L, l0 = col0 - b * L;
      pww_fetch(J.x, J.x_ctot, J.x_nlc, L, b, l0, row_a, rows, tid, ra);
      if (J.tr_mode == 2) pww_fetch(J.x2, J.x_ctot, J.x_nlc, L, b, l0, row_a, rows, tid, ra2);
      pww_fetch(J.y, J.y_ctot, J.y_nlc, L, b, l0, row_b, K, tid, rb);
    }
    // wave w owns output columns 64 tn + 16 w + r for the four 16-row tiles of the 64 x 64 block; k = column index
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      const float bv = bs[(16 * wave + r) * PWW_RS + k + g];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[(16 * mt + r) * PWW_RS + k + g], bv, acc[mt], 0, 0, 0);
    }
    if (want_b) {
      const float *rowp = as + (tid >> 2) * PWW_RS + 16 * (tid & 3);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += rowp[i];
      bsum += s;
    }
  }
  float *outp = J.dw_part + (size_t)sp * rows * K;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = row_a + 16 * mt + 4 * g + q, col = row_b + 16 * wave + r;
      if (row < rows && col < K) outp[(size_t)row * K + col] = acc[mt][q];
    }
  if (want_b) {
    bsum += __shfl_xor(bsum, 1);
    bsum += __shfl_xor(bsum, 2);
    const int row = row_a + (tid >> 2);
    if ((tid & 3) == 0 && row < rows) J.db_part[(size_t)sp * rows + row] = bsum;
  }
}

// ---- BatchNorm statistics / backward constants: one wave per channel, all jobs of a level in one launch -------------
struct PwBnJobs { p2r_pw_bnjob j[P2R_PW_MAX_JOBS]; int ch0[P2R_PW_MAX_JOBS + 1]; };
struct PwBnbJobs { p2r_pw_bnbjob j[P2R_PW_MAX_JOBS]; int ch0[P2R_PW_MAX_JOBS + 1]; };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(64) void pw_bn_finalize_kernel(PwBnJobs jobs) {
  int ji = 0;
#pragma unroll
  for (int i = 1; i < P2R_PW_MAX_JOBS; ++i)
    if ((int)blockIdx.x >= jobs.ch0[i]) ji = i;
  const p2r_pw_bnjob &J = jobs.j[ji];
  const int c = blockIdx.x - jobs.ch0[ji];
  const int lane = threadIdx.x;
  double mean, var, count = 0.0;
  if (J.part) {
    double n = 0.0, s = 0.0;
    for (int p = lane; p < J.P; p += 64) {
      const float *e = J.part + ((size_t)p * J.C + c) * 3;
      n += (double)e[0]; s += (double)e[0] * (double)e[1];
    }
    n = wave_sum(n); s = wave_sum(s);
    mean = s / n;
    double m2 = 0.0;
    for (int p = lane; p < J.P; p += 64) {
      const float *e = J.part + ((size_t)p * J.C + c) * 3;
      const double d = (double)e[1] - mean;
      m2 += (double)e[2] + (double)e[0] * d * d;
    }
    m2 = wave_sum(m2);
    var = m2 / n;
    if (var < 0.0) var = 0.0;
    count = n;
  } else {
    mean = (double)J.running_mean[c];
    var = (double)J.running_var[c];
  }
  if (lane == 0) {
    const float mean_f = (float)mean, invstd_f = (float)(1.0 / sqrt(var + J.eps));
    const float scale = J.gamma[c] * invstd_f;
    J.fin[c] = mean_f;
    J.fin[J.fin_ld + c] = invstd_f;
    J.fin[2 * J.fin_ld + c] = scale;
    J.fin[3 * J.fin_ld + c] = J.beta[c] - mean_f * scale;
    if (J.part && J.momentum >= 0.0) {
      const float mom = (float)J.momentum;
      const double unbiased = var * (count / (count - 1.0 > 1.0 ? count - 1.0 : 1.0));
      J.running_mean[c] = J.running_mean[c] * (1.f - mom) + mom * mean_f;
      J.running_var[c] = J.running_var[c] * (1.f - mom) + mom * (float)unbiased;
      if (c == 0 && J.num_batches_tracked) *J.num_batches_tracked += 1;
    }
  }
}

__global__ __launch_bounds__(64) void pw_bn_bwd_finalize_kernel(PwBnbJobs jobs) {
  int ji = 0;
#pragma unroll
  for (int i = 1; i < P2R_PW_MAX_JOBS; ++i)
    if ((int)blockIdx.x >= jobs.ch0[i]) ji = i;
  const p2r_pw_bnbjob &J = jobs.j[ji];
  const int c = blockIdx.x - jobs.ch0[ji];
  const int lane = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int p = lane; p < J.P; p += 64) {
    const float *e = J.part + ((size_t)p * J.C + c) * 2;
    s += (double)e[0]; q += (double)e[1];
  }
  s = wave_sum(s); q = wave_sum(q);
  if (lane == 0) {
    if (J.dbeta) J.dbeta[c] = (float)s;
    if (J.dgamma) J.dgamma[c] = (float)q;
    const float mean = J.fin[c], invstd = J.fin[J.fin_ld + c], scale = J.fin[2 * J.fin_ld + c];
    if (J.train) {
      const float m1 = (float)(s / J.M), m2 = (float)(q / J.M);
      J.coef[c] = scale;
      J.coef[J.coef_ld + c] = -scale * invstd * m2;
      J.coef[2 * J.coef_ld + c] = -scale * m1 + scale * invstd * m2 * mean;
    } else {
      J.coef[c] = scale;
      J.coef[J.coef_ld + c] = 0.f;
      J.coef[2 * J.coef_ld + c] = 0.f;
    }
  }
}

// ---- sums of partials over the leading axis, many small jobs in one launch -------------------------------------------
struct PwRJobs { p2r_pw_rjob j[P2R_PW_MAX_RJOBS]; int blk0[P2R_PW_MAX_RJOBS + 1]; };

__global__ __launch_bounds__(256) void pw_reduce_kernel(PwRJobs jobs, int njobs) {
  int ji = 0;
  for (int i = 1; i < njobs; ++i)
    if ((int)blockIdx.x >= jobs.blk0[i]) ji = i;
  const p2r_pw_rjob &J = jobs.j[ji];
  const int i = (blockIdx.x - jobs.blk0[ji]) * 256 + threadIdx.x;
  if (i >= J.M) return;
  // four interleaved accumulators, combined pairwise: rounding grows with P / 4, order fixed (deterministic)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int p = 0;
  for (; p + 4 <= J.P; p += 4) {
    a0 += J.in[(size_t)(p + 0) * J.M + i];
    a1 += J.in[(size_t)(p + 1) * J.M + i];
    a2 += J.in[(size_t)(p + 2) * J.M + i];
    a3 += J.in[(size_t)(p + 3) * J.M + i];
  }
  for (; p < J.P; ++p) a0 += J.in[(size_t)p * J.M + i];
  J.out[i] = (a0 + a1) + (a2 + a3);
}

// ---- mixture read-out ------------------------------------------------------------------------------------------------
constexpr int MIX_DMAX = 4;
struct MixHeads { p2r_mix_head h[P2R_MIX_MAX_HEADS]; };

// forward: a workgroup owns 16 columns of one head; thread (c = tid & 15, s = tid >> 4) takes components s, s + 16, ...;
// the 16 slices meet in LDS.  logit reads are 64-byte runs, a column's noise (G * D values) stays in L1 for the block.
template <typename T>
__device__ __forceinline__ void mix_forward_body(const p2r_mix_head &H, int cols, int G, int L, int ctot, double *red) {
  const int tid = threadIdx.x, c = tid & 15, s = tid >> 4;
  const int col = blockIdx.x * 16 + c;
  const int D = H.D;
  const T *mu = (const T *)H.mu, *eps = (const T *)H.eps;
  T acc[MIX_DMAX];
#pragma unroll
  for (int d = 0; d < MIX_DMAX; ++d) acc[d] = (T)0;
  if (col < cols) {
    const int b = col / L, l = col - b * L;
    for (int gi = s; gi < G; gi += 16) {
      const float lg = H.logit[((size_t)b * ctot + gi) * L + l];
      const float p = 1.f / (1.f + expf(-lg));
      if (H.pi) H.pi[((size_t)b * G + gi) * L + l] = p;
#pragma unroll
      for (int d = 0; d < MIX_DMAX; ++d) {
        if (d < D) {
          T comp = mu[gi * D + d];
          if (eps) comp = eps[((size_t)col * G + gi) * D + d] * (T)expf(H.log_sigma[gi * D + d]) + comp;
          acc[d] += comp * (T)p;
        }
      }
    }
  }
#pragma unroll
  for (int d = 0; d < MIX_DMAX; ++d) red[(d * 16 + s) * 16 + c] = (double)acc[d];
  __syncthreads();
  if (tid < 16 * D) {                      // thread (d, c): slices added in order
    const int d = tid >> 4, cc = tid & 15;
    T tot = (T)0;
    for (int i = 0; i < 16; ++i) tot += (T)red[(d * 16 + i) * 16 + cc];
    const int ocol = blockIdx.x * 16 + cc;
    if (ocol < cols) ((T *)H.pred)[(size_t)ocol * D + d] = tot;
  }
}

__global__ __launch_bounds__(256) void mdn_mix_forward_kernel(MixHeads heads, int cols, int G, int L, int ctot) {
  __shared__ double red[MIX_DMAX * 16 * 16];
  const p2r_mix_head &H = heads.h[blockIdx.y];
  if (H.f64) mix_forward_body<double>(H, cols, G, L, ctot, red);
  else mix_forward_body<float>(H, cols, G, L, ctot, red);
}

template <typename T>
__device__ __forceinline__ void mix_backward_body(const p2r_mix_head &H, int cols, int G, int L, int ctot, int dctot,
                                                  double (*red)[4]) {
  const int gi = blockIdx.x, tid = threadIdx.x;
  const int D = H.D;
  const T *mu = (const T *)H.mu, *eps = (const T *)H.eps, *dpred = (const T *)H.dpred;
  T muv[MIX_DMAX];
  float sig[MIX_DMAX];
#pragma unroll
  for (int d = 0; d < MIX_DMAX; ++d) {
    muv[d] = d < D ? mu[gi * D + d] : (T)0;
    sig[d] = d < D ? expf(H.log_sigma[gi * D + d]) : 0.f;
  }
  T smu[MIX_DMAX], ssg[MIX_DMAX];
#pragma unroll
  for (int d = 0; d < MIX_DMAX; ++d) { smu[d] = (T)0; ssg[d] = (T)0; }
  for (int col = tid; col < cols; col += 256) {
    const int b = col / L, l = col - b * L;
    const float lg = H.logit[((size_t)b * ctot + gi) * L + l];
    const float p = 1.f / (1.f + expf(-lg));
    T t = (T)0;
#pragma unroll
    for (int d = 0; d < MIX_DMAX; ++d) {
      if (d < D) {
        const T dp = dpred[(size_t)col * D + d];
        const T e = eps ? eps[((size_t)col * G + gi) * D + d] : (T)0;
        const T comp = e * (T)sig[d] + muv[d];
        t += dp * comp;
        const T dsamp = dp * (T)p;
        smu[d] += dsamp;
        ssg[d] += dsamp * e;
      }
    }
    H.dlogit[((size_t)b * dctot + gi) * L + l] = (float)t * (p * (1.f - p));
  }
  // block sums (fp64 staging serves both element types)
#pragma unroll
  for (int d = 0; d < MIX_DMAX; ++d) {
    double a = (double)smu[d], c = (double)ssg[d];
    a = wave_sum(a); c = wave_sum(c);
    if ((tid & 63) == 0) { red[d][tid >> 6] = a; red[MIX_DMAX + d][tid >> 6] = c; }
  }
  __syncthreads();
  if (tid < D) {
    const double a = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
    const double c = (red[MIX_DMAX + tid][0] + red[MIX_DMAX + tid][1]) + (red[MIX_DMAX + tid][2] + red[MIX_DMAX + tid][3]);
    ((T *)H.dmu)[gi * D + tid] = (T)a;
    H.dlog_sigma[gi * D + tid] = (float)((T)c) * sig[tid];
  }
}

__global__ __launch_bounds__(256) void mdn_mix_backward_kernel(MixHeads heads, int cols, int G, int L, int ctot, int dctot) {
  __shared__ double red[2 * MIX_DMAX][4];
  const p2r_mix_head &H = heads.h[blockIdx.y];
  if (H.f64) mix_backward_body<double>(H, cols, G, L, ctot, dctot, red);
  else mix_backward_body<float>(H, cols, G, L, ctot, dctot, red);
}

}  // namespace

extern "C" int p2r_pw_gemm(int njobs, const p2r_pw_job *jobs, int B, int L, void *stream) {
  if (njobs < 1 || njobs > P2R_PW_MAX_JOBS || B < 0 || L <= 0 || L % PW_COLS != 0 || !jobs) return P2R_EINVAL;
  if (B == 0) return P2R_OK;
  PwJobs pj;
  int kmax = 0;
  for (int i = 0; i < njobs; ++i) {
    const p2r_pw_job &j = jobs[i];
    const int kpad = (j.k + 15) & ~15;
    if (j.k <= 0 || j.rows <= 0 || kpad > PW_KMAX || !j.x || !j.w || !j.out) return P2R_EINVAL;
    if (!j.w_t && (j.k % 16 != 0 || (((size_t)j.w) & 15) != 0)) return P2R_EINVAL;
    if ((((size_t)j.x) & 15) != 0 || (j.tr_mode == 2 && (!j.x2 || (((size_t)j.x2) & 15) != 0))) return P2R_EINVAL;
    if (j.tr_mode != 0 && !j.tr) return P2R_EINVAL;
    if (j.epilogue == 1 && (!j.mz || !j.mfin)) return P2R_EINVAL;
    if (j.epilogue != 0 && j.epilogue != 1) return P2R_EINVAL;
    pj.j[i] = j;
    if (kpad > kmax) kmax = kpad;
  }
  const int colblocks = (int)((long long)B * L / PW_COLS);
  const int lds = kmax * PW_RS * (int)sizeof(float);
  static unsigned char lds_ok[P2R_MAX_DEVICES];
  hipError_t e = p2r_allow_big_lds(pw_gemm_kernel, lds_ok, PW_KMAX * PW_RS * (int)sizeof(float));
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(pw_gemm_kernel, dim3((unsigned)(njobs * colblocks)), dim3(256), lds, p2r_stream(stream), pj, L,
                     colblocks);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_pw_wgrad(int njobs, const p2r_pw_wjob *jobs, int B, int L, void *stream) {
  if (njobs < 1 || njobs > P2R_PW_MAX_JOBS || B < 0 || L <= 0 || L % 64 != 0 || !jobs) return P2R_EINVAL;
  if (B == 0) return P2R_OK;
  PwWJobs pj;
  int total = 0;
  for (int i = 0; i < P2R_PW_MAX_JOBS + 1; ++i) pj.blk0[i] = 0x7fffffff;
  for (int i = 0; i < njobs; ++i) {
    const p2r_pw_wjob &j = jobs[i];
    if (j.rows <= 0 || j.k <= 0 || j.split < 1 || !j.x || !j.y || !j.dw_part) return P2R_EINVAL;
    if (j.tr_mode != 0 && j.tr_mode != 2) return P2R_EINVAL;
    if (j.tr_mode == 2 && (!j.x2 || !j.tr)) return P2R_EINVAL;
    if ((((size_t)j.x) & 15) != 0 || (((size_t)j.y) & 15) != 0 || (j.x2 && (((size_t)j.x2) & 15) != 0)) return P2R_EINVAL;
    pj.j[i] = j;
    pj.blk0[i] = total;
    total += ((j.rows + 63) / 64) * ((j.k + 63) / 64) * j.split;
  }
  pj.blk0[0] = 0;
  const int chunks = (int)((long long)B * L / 64);
  hipLaunchKernelGGL(pw_wgrad_kernel, dim3((unsigned)total), dim3(256), 0, p2r_stream(stream), pj, L, chunks);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_pw_bn_finalize(int njobs, const p2r_pw_bnjob *jobs, void *stream) {
  if (njobs < 1 || njobs > P2R_PW_MAX_JOBS || !jobs) return P2R_EINVAL;
  PwBnJobs pj;
  int total = 0;
  for (int i = 0; i < P2R_PW_MAX_JOBS + 1; ++i) pj.ch0[i] = 0x7fffffff;
  for (int i = 0; i < njobs; ++i) {
    const p2r_pw_bnjob &j = jobs[i];
    if (j.C <= 0 || !j.gamma || !j.beta || !j.fin || (j.part && j.P <= 0)) return P2R_EINVAL;
    if (!j.part && (!j.running_mean || !j.running_var)) return P2R_EINVAL;
    if (j.part && j.momentum >= 0.0 && (!j.running_mean || !j.running_var)) return P2R_EINVAL;
    pj.j[i] = j;
    pj.ch0[i] = total;
    total += j.C;
  }
  pj.ch0[0] = 0;
  hipLaunchKernelGGL(pw_bn_finalize_kernel, dim3((unsigned)total), dim3(64), 0, p2r_stream(stream), pj);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_pw_bn_bwd_finalize(int njobs, const p2r_pw_bnbjob *jobs, void *stream) {
  if (njobs < 1 || njobs > P2R_PW_MAX_JOBS || !jobs) return P2R_EINVAL;
  PwBnbJobs pj;
  int total = 0;
  for (int i = 0; i < P2R_PW_MAX_JOBS + 1; ++i) pj.ch0[i] = 0x7fffffff;
  for (int i = 0; i < njobs; ++i) {
    const p2r_pw_bnbjob &j = jobs[i];
    if (j.C <= 0 || j.P <= 0 || !j.part || !j.fin || !j.coef || j.M <= 0.0) return P2R_EINVAL;
    pj.j[i] = j;
    pj.ch0[i] = total;
    total += j.C;
  }
  pj.ch0[0] = 0;
  hipLaunchKernelGGL(pw_bn_bwd_finalize_kernel, dim3((unsigned)total), dim3(64), 0, p2r_stream(stream), pj);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_pw_reduce(int njobs, const p2r_pw_rjob *jobs, void *stream) {
  if (njobs < 1 || njobs > P2R_PW_MAX_RJOBS || !jobs) return P2R_EINVAL;
  PwRJobs pj;
  int total = 0;
  for (int i = 0; i < P2R_PW_MAX_RJOBS + 1; ++i) pj.blk0[i] = 0x7fffffff;
  for (int i = 0; i < njobs; ++i) {
    const p2r_pw_rjob &j = jobs[i];
    if (j.P <= 0 || j.M <= 0 || !j.in || !j.out) return P2R_EINVAL;
    pj.j[i] = j;
    pj.blk0[i] = total;
    total += (j.M + 255) / 256;
  }
  pj.blk0[0] = 0;
  hipLaunchKernelGGL(pw_reduce_kernel, dim3((unsigned)total), dim3(256), 0, p2r_stream(stream), pj, njobs);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_mdn_mix_forward(int nheads, const p2r_mix_head *heads, int B, int G, int L, int logit_ctot,
                                   void *stream) {
  if (nheads < 1 || nheads > P2R_MIX_MAX_HEADS || !heads || B < 0 || G <= 0 || L <= 0) return P2R_EINVAL;
  if (B == 0) return P2R_OK;
  MixHeads mh;
  for (int i = 0; i < nheads; ++i) {
    const p2r_mix_head &h = heads[i];
    if (h.D <= 0 || h.D > MIX_DMAX || !h.logit || !h.mu || !h.pred || (h.eps && !h.log_sigma)) return P2R_EINVAL;
    mh.h[i] = h;
  }
  const int cols = B * L;
  hipLaunchKernelGGL(mdn_mix_forward_kernel, dim3((unsigned)p2r_cdiv(cols, 16), (unsigned)nheads), dim3(256), 0,
                     p2r_stream(stream), mh, cols, G, L, logit_ctot);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}

extern "C" int p2r_mdn_mix_backward(int nheads, const p2r_mix_head *heads, int B, int G, int L, int logit_ctot,
                                    int dlogit_ctot, void *stream) {
  if (nheads < 1 || nheads > P2R_MIX_MAX_HEADS || !heads || B <= 0 || G <= 0 || L <= 0) return P2R_EINVAL;
  MixHeads mh;
  for (int i = 0; i < nheads; ++i) {
    const p2r_mix_head &h = heads[i];
    if (h.D <= 0 || h.D > MIX_DMAX || !h.logit || !h.mu || !h.log_sigma || !h.dpred || !h.dlogit || !h.dmu ||
        !h.dlog_sigma)
      return P2R_EINVAL;
    mh.h[i] = h;
  }
  hipLaunchKernelGGL(mdn_mix_backward_kernel, dim3((unsigned)G, (unsigned)nheads), dim3(256), 0, p2r_stream(stream), mh,
                     B * L, G, L, logit_ctot, dlogit_ctot);
  P2R_LAUNCH_CHECK();
  return P2R_OK;
}
