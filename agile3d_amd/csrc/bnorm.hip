// libagile3d_hip -- BatchNorm in TRAINING mode over the [N, C] rows of a sparse tensor (ME.MinkowskiBatchNorm =
// nn.BatchNorm1d over the concatenated rows of the batch, models/modules/common.py:22), forward and backward, with the
// residual add and the ReLU of BasicBlock.forward (resnet_block.py:48-64) fused in.  Second kernel family of the
// training path.  All of it is HBM-bound column statistics + element-wise work:
//   forward : mean_c, var_c (two passes, like torch: the variance is taken around the mean), y = (x-mean) rstd g + b
//             (+ res) (ReLU); running statistics updated with the unbiased variance
//   backward: g = dy (y > 0);  dbeta = sum g, dgamma = sum g xhat;  dx = gamma rstd (g - dbeta/N - xhat dgamma/N);
//             the residual branch receives g
// Column sums are two-stage and deterministic: row blocks -> partial[block][C] (fp32) -> one thread per column sums
// the partials in block order in fp64.
#include "common.h"

namespace a3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBnThreads = 192;     // divisible by C/4 for every channel count of the net (8..96 float4 columns)
constexpr int kBnMaxBlocks = 1024;

// mode 0: sum x            mode 1: sum (x - m)^2
// mode 2: two sums for the backward: g = dy * (y > 0 or 1), sum g and sum g * (x - m) * rstd   (out: [2][C])
struct ColArgs {
  const float *x, *y, *dy, *mean, *rstd;
  int ldx, ldy, lddy, n, C, relu, mode, rows_per_block;
  float* partial;   // [blocks][nsum][C]
};

__global__ void __launch_bounds__(kBnThreads) k_col_partial(const ColArgs a) {
  __shared__ f32x4 red[2][kBnThreads];
  const int c4n = a.C >> 2;
  const int col = threadIdx.x % c4n, rsub = threadIdx.x / c4n, rstep = kBnThreads / c4n;
  const int r0 = blockIdx.x * a.rows_per_block, r1 = min(a.n, r0 + a.rows_per_block);
  f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
  f32x4 m = s0, rs = s0;
  if (a.mode != 0) m = *(const f32x4*)(a.mean + 4 * col);
  if (a.mode == 2) rs = *(const f32x4*)(a.rstd + 4 * col);
  // four rows per round: their loads are independent and issued together (one row per round left a wave with a single 1 KB
  // access in flight -- 12 KB per CU, a fifth of what the HBM pipe needs); the sums are still added row by row, in order
  constexpr int U = 4;
  for (int rb = r0 + rsub; rb < r1; rb += U * rstep) {
    f32x4 xv[U], gv[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = rb + u * rstep;
      const bool ok = r < r1;
      const size_t rr = ok ? (size_t)r : (size_t)r0;
      xv[u] = *(const f32x4*)(a.x + rr * a.ldx + 4 * col);
      if (a.mode == 2) {
        gv[u] = *(const f32x4*)(a.dy + rr * a.lddy + 4 * col);
        if (a.relu) yv[u] = *(const f32x4*)(a.y + rr * a.ldy + 4 * col);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (rb + u * rstep >= r1) break;
      if (a.mode == 0) {
        s0 += xv[u];
      } else if (a.mode == 1) {
        const f32x4 d = xv[u] - m;
        s0 += d * d;
      } else {
        f32x4 g = gv[u];
        if (a.relu) {
#pragma unroll
          for (int t = 0; t < 4; ++t) g[t] = yv[u][t] > 0.f ? g[t] : 0.f;
        }
        s0 += g;
        s1 += g * ((xv[u] - m) * rs);
      }
    }
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  if (rsub == 0) {   // fold the row sub-lanes of this column in a fixed order
    for (int k = 1; k < rstep; ++k) {
      s0 += red[0][k * c4n + col];
      s1 += red[1][k * c4n + col];
    }
    const int nsum = a.mode == 2 ? 2 : 1;
    float* p = a.partial + (size_t)blockIdx.x * nsum * a.C + 4 * col;
    *(f32x4*)p = s0;
    if (nsum == 2) *(f32x4*)(p + a.C) = s1;
  }
}

// Forward statistics in ONE sweep kernel + one combine kernel (3 launches per BatchNorm layer with the apply pass;
// the first version took 7): every row block sums its rows, takes ITS mean, and sums the squared deviations from it in a
// second loop over the same rows (L2-resident by then); k_bn_combine merges the blocks' (count, mean, M2) in block order
// with Chan's update in fp64 -- the variance is still taken around a mean, never as E[x^2] - E[x]^2.
__global__ void __launch_bounds__(kBnThreads) k_bn_block_stats(const ColArgs a) {
  __shared__ f32x4 red[kBnThreads];
  __shared__ f32x4 bmean[kBnThreads];
  const int c4n = a.C >> 2;
  const int col = threadIdx.x % c4n, rsub = threadIdx.x / c4n, rstep = kBnThreads / c4n;
  const int r0 = blockIdx.x * a.rows_per_block, r1 = min(a.n, r0 + a.rows_per_block);
  f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + rsub; r < r1; r += rstep) s0 += *(const f32x4*)(a.x + (size_t)r * a.ldx + 4 * col);
  red[threadIdx.x] = s0;
  __syncthreads();
  if (rsub == 0) {
    for (int k = 1; k < rstep; ++k) s0 += red[k * c4n + col];
    bmean[col] = s0 * (1.f / (float)(r1 - r0));
    *(f32x4*)(a.partial + (size_t)blockIdx.x * 2 * a.C + 4 * col) = s0;
  }
  __syncthreads();
  const f32x4 m = bmean[col];
  f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + rsub; r < r1; r += rstep) {
    const f32x4 d = *(const f32x4*)(a.x + (size_t)r * a.ldx + 4 * col) - m;
    s1 += d * d;
  }
  red[threadIdx.x] = s1;
  __syncthreads();
  if (rsub == 0) {
    for (int k = 1; k < rstep; ++k) s1 += red[k * c4n + col];
    *(f32x4*)(a.partial + (size_t)blockIdx.x * 2 * a.C + a.C + 4 * col) = s1;
  }
}
// The second level of the column reductions.  One thread per column walking up to 1024 block partials is a chain of a
// thousand dependent L2 loads (38 us for a few KB of arithmetic; 130 of these launches per training iteration): a
// 256-thread workgroup takes 16 consecutive columns, thread (bl, cc) folds the blocks bl, bl + 16, ... of column cc (64-byte
// row segments per load), then one thread per column folds the 16 partial results in bl order -- fixed orders, fp64.
constexpr int kFinCols = 16, kFinLanes = 16;
__global__ void __launch_bounds__(256) k_bn_combine(const float* __restrict__ partial, int nblocks, int rows_per_block, int n,
                                                    int C, float eps, float* mean_out, float* rstd_out, float* running_mean,
                                                    float* running_var, float momentum) {
  __shared__ double s_cnt[kFinLanes][kFinCols], s_mean[kFinLanes][kFinCols], s_m2[kFinLanes][kFinCols];
  const int cc = threadIdx.x % kFinCols, bl = threadIdx.x / kFinCols;
  const int c = blockIdx.x * kFinCols + cc;
  double cnt = 0.0, mean = 0.0, m2 = 0.0;
  if (c < C) {
    for (int b = bl; b < nblocks; b += kFinLanes) {
      const double nb = (double)(min(n, (b + 1) * rows_per_block) - b * rows_per_block);
      const double bs = (double)partial[(size_t)b * 2 * C + c], bm2 = (double)partial[(size_t)b * 2 * C + C + c];
      // the block's M2 was taken around the fp32 block mean the kernel used, not around bs / nb: shift it (exact identity)
      const double bm_used = (double)((float)bs * (1.f / (float)nb)), bm = bs / nb;
      const double bm2c = bm2 - nb * (bm - bm_used) * (bm - bm_used);
      const double delta = bm - mean, tot = cnt + nb;
      mean += delta * nb / tot;
      m2 += bm2c + delta * delta * cnt * nb / tot;
      cnt = tot;
    }
  }
  s_cnt[bl][cc] = cnt, s_mean[bl][cc] = mean, s_m2[bl][cc] = m2;
  __syncthreads();
  if (bl != 0 || c >= C) return;
  for (int k = 1; k < kFinLanes; ++k) {          // Chan's merge of two partial (count, mean, M2) triples, in bl order
    const double nb = s_cnt[k][cc];
    if (nb == 0.0) continue;
    const double delta = s_mean[k][cc] - mean, tot = cnt + nb;
    mean += delta * nb / tot;
    m2 += s_m2[k][cc] + delta * delta * cnt * nb / tot;
    cnt = tot;
  }
  const float var = (float)(m2 / cnt);
  mean_out[c] = (float)mean;
  rstd_out[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    const float unbiased = cnt > 1.0 ? (float)(m2 / (cnt - 1.0)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// The same merge for MANY small blocks (the conv kernels' epilogues hand over one (sum, M2 around the block mean) pair per
// 64-row tile or 16-row group: 5 000 - 20 000 blocks at 320 k rows) without a dependent chain of divisions: pass 1 sums the
// block sums (fp64) -> the batch mean; pass 2 shifts every block's M2 to that mean with the exact identity
//   sum (v - mean)^2 = M2_b + 2 (m_b - mean) (S_b - n_b m_b) + n_b (m_b - mean)^2        (m_b = the fp32 mean the block used)
// 64 lanes per column, 16 columns per 1024-thread workgroup, every partial sum in a fixed order.
constexpr int kFin2Lanes = 256;
__global__ void __launch_bounds__(1024) k_bn_combine_tiles(const float* __restrict__ partial, int nblocks, int rows_per_block, int n,
                                                           int C, float eps, float* mean_out, float* rstd_out, float* running_mean,
                                                           float* running_var, float momentum) {
  // thread (bl, q): block lane bl of 256, column quad q of the workgroup's 4 (16 columns per workgroup): one 16-byte load
  // per block and thread, blocks bl, bl + 256, ... (20 rounds for the 5 000 tiles of a 320 k-row level)
  __shared__ double sh[kFin2Lanes][kFinCols + 1];
  __shared__ double s_mean[kFinCols];
  const int q = threadIdx.x & 3, bl = threadIdx.x >> 2;
  const int c0 = blockIdx.x * kFinCols + 4 * q;
  const bool live = c0 < C;                          // C is a multiple of 4 (of 32)
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (live)
    for (int b = bl; b < nblocks; b += kFin2Lanes) {
      const f32x4 v = *(const f32x4*)(partial + (size_t)b * 2 * C + c0);
#pragma unroll
      for (int t = 0; t < 4; ++t) s[t] += (double)v[t];
    }
#pragma unroll
  for (int t = 0; t < 4; ++t) sh[bl][4 * q + t] = s[t];
  __syncthreads();
  if (threadIdx.x < kFinCols) {
    double t = 0.0;
    for (int k = 0; k < kFin2Lanes; ++k) t += sh[k][threadIdx.x];
    s_mean[threadIdx.x] = t / (double)n;
  }
  __syncthreads();
  double m2s[4] = {0.0, 0.0, 0.0, 0.0};
  if (live)
    for (int b = bl; b < nblocks; b += kFin2Lanes) {
      const int nbi = min(n, (b + 1) * rows_per_block) - b * rows_per_block;
      const f32x4 bsf = *(const f32x4*)(partial + (size_t)b * 2 * C + c0);
      const f32x4 bq = *(const f32x4*)(partial + (size_t)b * 2 * C + C + c0);
      const double nb = (double)nbi;
      const float invf = 1.f / (float)nbi;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double m_used = (double)(bsf[t] * invf);
        const double dm = m_used - s_mean[4 * q + t];
        m2s[t] += (double)bq[t] + 2.0 * dm * ((double)bsf[t] - nb * m_used) + nb * dm * dm;
      }
    }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) sh[bl][4 * q + t] = m2s[t];
  __syncthreads();
  const int c = blockIdx.x * kFinCols + threadIdx.x;
  if (threadIdx.x >= kFinCols || c >= C) return;
  double m2 = 0.0;
  for (int k = 0; k < kFin2Lanes; ++k) m2 += sh[k][threadIdx.x];
  if (m2 < 0.0) m2 = 0.0;
  const double cnt = (double)n, mean = s_mean[threadIdx.x];
  const float var = (float)(m2 / cnt);
  mean_out[c] = (float)mean;
  rstd_out[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    const float unbiased = cnt > 1.0 ? (float)(m2 / (cnt - 1.0)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// column sums of the block partials, fp64: ncol = nsum * C columns of [nblocks][ncol] (launch: ceil(ncol / 16) x 256)
__device__ __forceinline__ double col_final_sum(const float* __restrict__ partial, int nblocks, int ncol, int c, int bl, int cc,
                                                double (*sh)[kFinCols]) {
  double s = 0.0;
  if (c < ncol)
    for (int b = bl; b < nblocks; b += kFinLanes) s += (double)partial[(size_t)b * ncol + c];
  sh[bl][cc] = s;
  __syncthreads();
  double t = 0.0;
  if (bl == 0)
    for (int k = 0; k < kFinLanes; ++k) t += sh[k][cc];
  return t;
}
__global__ void __launch_bounds__(256) k_col_final(const float* __restrict__ partial, int nblocks, int nsum, int C, double* out) {
  __shared__ double sh[kFinLanes][kFinCols];
  const int cc = threadIdx.x % kFinCols, bl = threadIdx.x / kFinCols;
  const int c = blockIdx.x * kFinCols + cc;
  const double t = col_final_sum(partial, nblocks, nsum * C, c, bl, cc, sh);
  if (bl == 0 && c < nsum * C) out[c] = t;
}

// the same sum written as fp32 (a3d_column_sums: one launch instead of k_col_final + a conversion kernel)
__global__ void __launch_bounds__(256) k_col_final_f32(const float* __restrict__ partial, int nblocks, int C, float* out) {
  __shared__ double sh[kFinLanes][kFinCols];
  const int cc = threadIdx.x % kFinCols, bl = threadIdx.x / kFinCols;
  const int c = blockIdx.x * kFinCols + cc;
  const double t = col_final_sum(partial, nblocks, C, c, bl, cc, sh);
  if (bl == 0 && c < C) out[c] = (float)t;
}

__global__ void k_scale_f64(const double* in, int C, double f, double* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) out[c] = in[c] * f;
}
__global__ void k_bn_mean(const double* sums, int C, double n, float* mean) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) mean[c] = (float)(sums[c] / n);
}
__global__ void k_bn_rstd(const double* sq, int C, double n, float eps, float* rstd, const float* mean,
                          float* running_mean, float* running_var, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float var = (float)(sq[c] / n);
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    const float unbiased = n > 1.0 ? (float)(sq[c] / (n - 1.0)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

struct ApplyArgs {
  const float *x, *mean, *rstd, *gamma, *beta, *res;
  int ldx, ldr, ldy, n, C, relu;
  int zero_row;   // != 0: row n of y is written as zeros (the row a missing neighbour gathers in the next conv)
  float* y;
};
__global__ void k_bn_apply(const ApplyArgs a) {
  const int c4n = a.C >> 2;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a.zero_row && e < (size_t)c4n) *(f32x4*)(a.y + (size_t)a.n * a.ldy + 4 * e) = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (e >= (size_t)a.n * c4n) return;
  const int r = (int)(e / c4n), col = (int)(e % c4n);
  const f32x4 xv = *(const f32x4*)(a.x + (size_t)r * a.ldx + 4 * col);
  const f32x4 m = *(const f32x4*)(a.mean + 4 * col), rs = *(const f32x4*)(a.rstd + 4 * col);
  const f32x4 ga = *(const f32x4*)(a.gamma + 4 * col), be = *(const f32x4*)(a.beta + 4 * col);
  f32x4 y = (xv - m) * rs * ga + be;
  if (a.res) y += *(const f32x4*)(a.res + (size_t)r * a.ldr + 4 * col);
  if (a.relu) {
#pragma unroll
    for (int t = 0; t < 4; ++t) y[t] = fmaxf(y[t], 0.f);
  }
  *(f32x4*)(a.y + (size_t)r * a.ldy + 4 * col) = y;
}

struct BwdArgs {
  const float *x, *y, *dy, *mean, *rstd, *gamma;
  const double* sums;         // [2][C]: sum g, sum g xhat over the rows the statistics were taken over
  const double* param_sums;   // [2][C]: the same over THIS call's rows (dgamma / dbeta)
  double n_stat;              // number of rows the statistics were taken over
  int zero_row;               // != 0: dx (and dres) have n + 1 rows, row n is written as zeros
  int ldx, ldy, lddy, lddx, lddres, n, C, relu;
  float *dx, *dres, *dgamma, *dbeta;
};
__global__ void k_bn_bwd_apply(const BwdArgs a) {
  const int c4n = a.C >> 2;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a.zero_row && e < (size_t)c4n) {
    *(f32x4*)(a.dx + (size_t)a.n * a.lddx + 4 * e) = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.dres) *(f32x4*)(a.dres + (size_t)a.n * a.lddres + 4 * e) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (e >= (size_t)a.n * c4n) return;
  const int r = (int)(e / c4n), col = (int)(e % c4n);
  const f32x4 xv = *(const f32x4*)(a.x + (size_t)r * a.ldx + 4 * col);
  f32x4 g = *(const f32x4*)(a.dy + (size_t)r * a.lddy + 4 * col);
  if (a.relu) {
    const f32x4 yv = *(const f32x4*)(a.y + (size_t)r * a.ldy + 4 * col);
#pragma unroll
    for (int t = 0; t < 4; ++t) g[t] = yv[t] > 0.f ? g[t] : 0.f;
  }
  if (a.dres) *(f32x4*)(a.dres + (size_t)r * a.lddres + 4 * col) = g;
  const f32x4 m = *(const f32x4*)(a.mean + 4 * col), rs = *(const f32x4*)(a.rstd + 4 * col);
  const f32x4 ga = *(const f32x4*)(a.gamma + 4 * col);
  const float inv_n = (float)(1.0 / a.n_stat);
  f32x4 dx;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float xh = (xv[t] - m[t]) * rs[t];
    const float db = (float)a.sums[4 * col + t], dg = (float)a.sums[a.C + 4 * col + t];
    dx[t] = ga[t] * rs[t] * (g[t] - db * inv_n - xh * dg * inv_n);
  }
  *(f32x4*)(a.dx + (size_t)r * a.lddx + 4 * col) = dx;
  if (r == 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a.dbeta[4 * col + t] = (float)a.param_sums[4 * col + t];
      a.dgamma[4 * col + t] = (float)a.param_sums[a.C + 4 * col + t];
    }
  }
}

// ---- LayerNorm over the channels of every row (nn.LayerNorm(128) of the decoder: attention_block.py:38,98,155,
// agile3d.py:137), forward and backward: one wave per row, lane = two channels at C = 128 (C <= 512: C / 64 per lane).
//   backward: gh = dy * gamma;  dx = rstd * (gh - mean(gh) - xhat * mean(gh * xhat));  dgamma = sum_rows dy * xhat,
//             dbeta = sum_rows dy  (row-block partials -> k_col_final, deterministic)
constexpr int kLnMaxPerLane = 8;
struct LnArgs {
  const float *x, *dy, *gamma, *beta;
  float *y, *dx, *partial;   // partial [blocks][2][C]
  int ldx, ldy, n, C, rows_per_block;
  float eps;
};
__global__ void __launch_bounds__(256) k_ln_forward(const LnArgs a) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n) return;
  const int per = a.C >> 6;
  float v[kLnMaxPerLane];
  float s = 0.f;
  for (int i = 0; i < per; ++i) {
    v[i] = a.x[(size_t)row * a.ldx + lane + 64 * i];
    s += v[i];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)a.C;
  float q = 0.f;
  for (int i = 0; i < per; ++i) q += (v[i] - mean) * (v[i] - mean);
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / (float)a.C + a.eps);
  for (int i = 0; i < per; ++i) {
    const int c = lane + 64 * i;
    a.y[(size_t)row * a.ldy + c] = (v[i] - mean) * rstd * a.gamma[c] + a.beta[c];
  }
}
__global__ void __launch_bounds__(256) k_ln_backward(const LnArgs a) {
  __shared__ float red[2][4][64 * kLnMaxPerLane];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = a.C >> 6;
  const int r0 = blockIdx.x * a.rows_per_block, r1 = min(a.n, r0 + a.rows_per_block);
  float dg[kLnMaxPerLane], db[kLnMaxPerLane], ga[kLnMaxPerLane];
  for (int i = 0; i < per; ++i) dg[i] = db[i] = 0.f, ga[i] = a.gamma[lane + 64 * i];
  for (int row = r0 + wave; row < r1; row += 4) {
    float v[kLnMaxPerLane], g[kLnMaxPerLane];
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
      v[i] = a.x[(size_t)row * a.ldx + lane + 64 * i];
      g[i] = a.dy[(size_t)row * a.ldy + lane + 64 * i];
      s += v[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)a.C;
    float q = 0.f;
    for (int i = 0; i < per; ++i) q += (v[i] - mean) * (v[i] - mean);
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)a.C + a.eps);
    float m1 = 0.f, m2 = 0.f;
    for (int i = 0; i < per; ++i) {
      v[i] = (v[i] - mean) * rstd;                  // xhat
      dg[i] += g[i] * v[i];
      db[i] += g[i];
      g[i] *= ga[i];                                // gh
      m1 += g[i];
      m2 += g[i] * v[i];
    }
    for (int o = 32; o > 0; o >>= 1) {
      m1 += __shfl_xor(m1, o, 64);
      m2 += __shfl_xor(m2, o, 64);
    }
    m1 /= (float)a.C, m2 /= (float)a.C;
    for (int i = 0; i < per; ++i) a.dx[(size_t)row * a.ldx + lane + 64 * i] = rstd * (g[i] - m1 - v[i] * m2);
  }
  for (int i = 0; i < per; ++i) {
    red[0][wave][lane + 64 * i] = dg[i];
    red[1][wave][lane + 64 * i] = db[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += 256) {
    // [2][C]: dbeta sums first, dgamma second (k_col_final sums column-wise over the blocks)
    a.partial[(size_t)blockIdx.x * 2 * a.C + c] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    a.partial[(size_t)blockIdx.x * 2 * a.C + a.C + c] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
  }
}
// C = 128 (every LayerNorm of the decoder): FOUR rows per wave -- 16 lanes per row, 8 channels (two 16-byte accesses) per
// lane, the row reductions over 16 lanes --, the next four rows' loads in flight behind the current ones' arithmetic.  The
// one-row-per-wave kernels above issue 4-byte accesses and four dependent shuffle trees per row: 950 us for the backward
// over 320 k rows (0.5 GB: 100 us at the HBM rate).
__device__ __forceinline__ float ln16_sum(float v) {
  v += __shfl_xor(v, 1, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 8, 16);
  return v;
}
__global__ void __launch_bounds__(256) k_ln_forward128(const LnArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane >> 4, j = lane & 15;
  const int row = blockIdx.x * 16 + wave * 4 + r;
  if (row >= a.n) return;
  const float* xp = a.x + (size_t)row * a.ldx + 8 * j;
  const f32x4 v0 = *(const f32x4*)xp, v1 = *(const f32x4*)(xp + 4);
  const float mean = ln16_sum(((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]))) * (1.f / 128.f);
  const f32x4 d0 = v0 - mean, d1 = v1 - mean;
  const float q = ln16_sum(((d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3])) +
                           ((d1[0] * d1[0] + d1[1] * d1[1]) + (d1[2] * d1[2] + d1[3] * d1[3])));
  const float rstd = 1.0f / sqrtf(q * (1.f / 128.f) + a.eps);
  const f32x4 g0 = *(const f32x4*)(a.gamma + 8 * j), g1 = *(const f32x4*)(a.gamma + 8 * j + 4);
  const f32x4 b0 = *(const f32x4*)(a.beta + 8 * j), b1 = *(const f32x4*)(a.beta + 8 * j + 4);
  float* yp = a.y + (size_t)row * a.ldy + 8 * j;
  *(f32x4*)yp = d0 * rstd * g0 + b0;
  *(f32x4*)(yp + 4) = d1 * rstd * g1 + b1;
}
__global__ void __launch_bounds__(256) k_ln_backward128(const LnArgs a) {
  __shared__ __attribute__((aligned(16))) float red[2][4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane >> 4, j = lane & 15;
  const int r0 = blockIdx.x * a.rows_per_block, r1 = min(a.n, r0 + a.rows_per_block);
  const f32x4 ga0 = *(const f32x4*)(a.gamma + 8 * j), ga1 = *(const f32x4*)(a.gamma + 8 * j + 4);
  f32x4 dg0 = (f32x4){0.f, 0.f, 0.f, 0.f}, dg1 = dg0, db0 = dg0, db1 = dg0;
  int row = r0 + wave * 4 + r;
  f32x4 nv0 = dg0, nv1 = dg0, ng0 = dg0, ng1 = dg0;
  auto fetch = [&](int rw) {
    if (rw < r1) {
      const float* xp = a.x + (size_t)rw * a.ldx + 8 * j;
      const float* gp = a.dy + (size_t)rw * a.ldy + 8 * j;
      nv0 = *(const f32x4*)xp, nv1 = *(const f32x4*)(xp + 4);
      ng0 = *(const f32x4*)gp, ng1 = *(const f32x4*)(gp + 4);
    }
  };
  fetch(row);
  // (a wave's four row slots run in lock step: the loop bound is the slot-0 row, slots past the end are masked)
  for (int base = r0 + wave * 4; base < r1; base += 16) {
    const bool ok = row < r1;
    f32x4 v0 = nv0, v1 = nv1, g0 = ng0, g1 = ng1;
    if (!ok) v0 = v1 = g0 = g1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    fetch(row + 16);
    const float mean = ln16_sum(((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]))) * (1.f / 128.f);
    f32x4 x0 = v0 - mean, x1 = v1 - mean;
    const float q = ln16_sum(((x0[0] * x0[0] + x0[1] * x0[1]) + (x0[2] * x0[2] + x0[3] * x0[3])) +
                             ((x1[0] * x1[0] + x1[1] * x1[1]) + (x1[2] * x1[2] + x1[3] * x1[3])));
    const float rstd = 1.0f / sqrtf(q * (1.f / 128.f) + a.eps);
    x0 = x0 * rstd, x1 = x1 * rstd;            // xhat
    dg0 += g0 * x0, dg1 += g1 * x1;
    db0 += g0, db1 += g1;
    g0 = g0 * ga0, g1 = g1 * ga1;              // gh
    const float m1 = ln16_sum(((g0[0] + g0[1]) + (g0[2] + g0[3])) + ((g1[0] + g1[1]) + (g1[2] + g1[3]))) * (1.f / 128.f);
    const float m2 = ln16_sum(((g0[0] * x0[0] + g0[1] * x0[1]) + (g0[2] * x0[2] + g0[3] * x0[3])) +
                              ((g1[0] * x1[0] + g1[1] * x1[1]) + (g1[2] * x1[2] + g1[3] * x1[3]))) * (1.f / 128.f);
    if (ok) {
      float* dp = a.dx + (size_t)row * a.ldx + 8 * j;
      *(f32x4*)dp = rstd * (g0 - m1 - x0 * m2);
      *(f32x4*)(dp + 4) = rstd * (g1 - m1 - x1 * m2);
    }
    row += 16;
  }
  // fold the four row slots of the wave (lanes j, j + 16, j + 32, j + 48), then the four waves through LDS: fixed orders
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    dg0[t] += __shfl_xor(dg0[t], 16, 64), dg0[t] += __shfl_xor(dg0[t], 32, 64);
    dg1[t] += __shfl_xor(dg1[t], 16, 64), dg1[t] += __shfl_xor(dg1[t], 32, 64);
    db0[t] += __shfl_xor(db0[t], 16, 64), db0[t] += __shfl_xor(db0[t], 32, 64);
    db1[t] += __shfl_xor(db1[t], 16, 64), db1[t] += __shfl_xor(db1[t], 32, 64);
  }
  if (r == 0) {
    *(f32x4*)&red[0][wave][8 * j] = dg0, *(f32x4*)&red[0][wave][8 * j + 4] = dg1;
    *(f32x4*)&red[1][wave][8 * j] = db0, *(f32x4*)&red[1][wave][8 * j + 4] = db1;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x;
    // [2][C]: dbeta sums first, dgamma second (k_col_final sums column-wise over the blocks)
    a.partial[(size_t)blockIdx.x * 256 + c] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    a.partial[(size_t)blockIdx.x * 256 + 128 + c] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
  }
}
__global__ void k_ln_params(const double* sums, int C, float* dgamma, float* dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) dbeta[c] = (float)sums[c], dgamma[c] = (float)sums[C + c];
}

static bool bn_shape_ok(int64_t n, int C, int ld0, int ld1, int ld2) {
  return n > 0 && n <= (int64_t)1 << 30 && C >= 32 && C % 32 == 0 && kBnThreads % (C / 4) == 0 && ld0 % 4 == 0 &&
         ld1 % 4 == 0 && ld2 % 4 == 0;
}
static int bn_blocks(int64_t n, int& rows_per_block);
// BatchNorm(train) from the conv epilogues' block partials (spconv.hip: a3d_conv_bn_train_forward): merge + apply
int bn_finish_from_partials(const float* partial, int nblocks, int rows_per_block, const float* x, int ldx, int64_t n, int C,
                            const float* gamma, const float* beta, float eps, const float* res, int ldr, int relu, float* y,
                            int ldy, int y_zero_row, float* save_mean, float* save_rstd, float* running_mean,
                            float* running_var, float momentum, hipStream_t st) {
  if (!partial || !x || !gamma || !beta || !y || !save_mean || !save_rstd || n <= 0 || n > (int64_t)1 << 30 || C < 32 || C % 32 ||
      (ldx & 3) || (ldy & 3) || (res && (ldr & 3)) || (running_mean == nullptr) != (running_var == nullptr) || nblocks < 1 ||
      (int64_t)(nblocks - 1) * rows_per_block >= n) {
    set_error("bn_finish_from_partials: bad arguments");
    return A3D_ERR_INVALID;
  }
  k_bn_combine_tiles<<<(unsigned)((C + kFinCols - 1) / kFinCols), 1024, 0, st>>>(partial, nblocks, rows_per_block, (int)n, C, eps, save_mean,
                                                                               save_rstd, running_mean, running_var, momentum);
  ApplyArgs a;
  a.x = x, a.mean = save_mean, a.rstd = save_rstd, a.gamma = gamma, a.beta = beta, a.res = res;
  a.ldx = ldx, a.ldr = ldr, a.ldy = ldy, a.n = (int)n, a.C = C, a.relu = relu, a.y = y, a.zero_row = y_zero_row;
  const size_t total = (size_t)n * (C / 4);
  k_bn_apply<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
// [blocks][2][C] fp32 partials (the input-gradient conv's epilogue) -> sums [2][C] fp64, blocks added in a fixed order
// column sums of MANY block partials (one per 64-row tile or 16-row group of a conv epilogue): 256 block lanes x 4 column
// quads per workgroup, one 16-byte load per block and thread (k_col_final's 16 lanes per column walked 300+ blocks each)
__global__ void __launch_bounds__(1024) k_col_final_tiles(const float* __restrict__ partial, int nblocks, int ncol, double* out) {
  __shared__ double sh[kFin2Lanes][kFinCols + 1];
  const int q = threadIdx.x & 3, bl = threadIdx.x >> 2;
  const int c0 = blockIdx.x * kFinCols + 4 * q;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (c0 < ncol)
    for (int b = bl; b < nblocks; b += kFin2Lanes) {
      const f32x4 v = *(const f32x4*)(partial + (size_t)b * ncol + c0);
#pragma unroll
      for (int t = 0; t < 4; ++t) s[t] += (double)v[t];
    }
#pragma unroll
  for (int t = 0; t < 4; ++t) sh[bl][4 * q + t] = s[t];
  __syncthreads();
  const int c = blockIdx.x * kFinCols + threadIdx.x;
  if (threadIdx.x >= kFinCols || c >= ncol) return;
  double t = 0.0;
  for (int k = 0; k < kFin2Lanes; ++k) t += sh[k][threadIdx.x];
  out[c] = t;
}
int bn_sums_from_partials(const float* partial, int nblocks, int C, double* sums, hipStream_t st) {
  if (nblocks >= 64)
    k_col_final_tiles<<<(unsigned)((2 * C + kFinCols - 1) / kFinCols), 1024, 0, st>>>(partial, nblocks, 2 * C, sums);
  else
    k_col_final<<<(unsigned)((2 * C + kFinCols - 1) / kFinCols), 256, 0, st>>>(partial, nblocks, 2, C, sums);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
static int bn_blocks(int64_t n, int& rows_per_block) {
  int blocks = (int)((n + 511) / 512);
  if (blocks > kBnMaxBlocks) blocks = kBnMaxBlocks;
  rows_per_block = (int)((n + blocks - 1) / blocks);
  return (int)((n + rows_per_block - 1) / rows_per_block);
}

}  // namespace a3d

using namespace a3d;

extern "C" size_t a3d_bn_workspace_bytes(int64_t n, int C) {
  if (n <= 0 || C <= 0) return 0;
  return align256((size_t)kBnMaxBlocks * 2 * C * sizeof(float)) + align256((size_t)2 * C * sizeof(double)) +
         align256((size_t)C * sizeof(float)) + 256;
}

extern "C" int a3d_bn_train_forward(const float* x_dev, int ldx, int64_t n, int C, const float* gamma_dev,
                                    const float* beta_dev, float eps, const float* res_dev, int ldr, int relu,
                                    float* y_dev, int ldy, float* save_mean_dev, float* save_rstd_dev,
                                    float* running_mean_dev, float* running_var_dev, float momentum, int y_zero_row,
                                    void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!x_dev || !gamma_dev || !beta_dev || !y_dev || !save_mean_dev || !save_rstd_dev || !workspace_dev ||
      !bn_shape_ok(n, C, ldx, ldy, res_dev ? ldr : 4) || (running_mean_dev == nullptr) != (running_var_dev == nullptr)) {
    set_error("a3d_bn_train_forward: bad arguments (C a multiple of 32 dividing 768, leading dimensions multiples of 4)");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < a3d_bn_workspace_bytes(n, C) || ((uintptr_t)workspace_dev & 15)) {
    set_error("a3d_bn_train_forward: workspace too small or misaligned");
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace_dev;
  double* sums = (double*)((char*)workspace_dev + align256((size_t)kBnMaxBlocks * 2 * C * sizeof(float)));
  ColArgs c;
  memset(&c, 0, sizeof(c));
  c.x = x_dev, c.ldx = ldx, c.n = (int)n, c.C = C, c.partial = partial;
  const int blocks = bn_blocks(n, c.rows_per_block);
  const unsigned cb = (unsigned)((2 * C + 255) / 256);
  (void)sums;
  (void)cb;
  k_bn_block_stats<<<blocks, kBnThreads, 0, st>>>(c);
  k_bn_combine<<<(unsigned)((C + kFinCols - 1) / kFinCols), 256, 0, st>>>(partial, blocks, c.rows_per_block, (int)n, C, eps, save_mean_dev,
                                                             save_rstd_dev, running_mean_dev, running_var_dev, momentum);
  ApplyArgs a;
  a.x = x_dev, a.mean = save_mean_dev, a.rstd = save_rstd_dev, a.gamma = gamma_dev, a.beta = beta_dev, a.res = res_dev;
  a.ldx = ldx, a.ldr = ldr, a.ldy = ldy, a.n = (int)n, a.C = C, a.relu = relu, a.y = y_dev, a.zero_row = y_zero_row;
  const size_t total = (size_t)n * (C / 4);
  k_bn_apply<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_bn_train_backward(const float* x_dev, int ldx, const float* y_dev, int ldy, const float* dy_dev,
                                     int lddy, int64_t n, int C, const float* gamma_dev, const float* save_mean_dev,
                                     const float* save_rstd_dev, int relu, float* dx_dev, int lddx, float* dres_dev,
                                     int lddres, float* dgamma_dev, float* dbeta_dev, int zero_row, void* workspace_dev,
                                     size_t workspace_bytes, void* stream) {
  if (!x_dev || !dy_dev || !gamma_dev || !save_mean_dev || !save_rstd_dev || !dx_dev || !dgamma_dev || !dbeta_dev ||
      !workspace_dev || (relu && !y_dev) || !bn_shape_ok(n, C, ldx, lddy, lddx) || (relu && ldy % 4) ||
      (dres_dev && lddres % 4)) {
    set_error("a3d_bn_train_backward: bad arguments");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < a3d_bn_workspace_bytes(n, C) || ((uintptr_t)workspace_dev & 15)) {
    set_error("a3d_bn_train_backward: workspace too small or misaligned");
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace_dev;
  double* sums = (double*)((char*)workspace_dev + align256((size_t)kBnMaxBlocks * 2 * C * sizeof(float)));
  ColArgs c;
  memset(&c, 0, sizeof(c));
  c.x = x_dev, c.y = y_dev, c.dy = dy_dev, c.mean = save_mean_dev, c.rstd = save_rstd_dev;
  c.ldx = ldx, c.ldy = ldy, c.lddy = lddy, c.n = (int)n, c.C = C, c.relu = relu, c.mode = 2, c.partial = partial;
  const int blocks = bn_blocks(n, c.rows_per_block);
  k_col_partial<<<blocks, kBnThreads, 0, st>>>(c);
  k_col_final<<<(unsigned)((2 * C + kFinCols - 1) / kFinCols), 256, 0, st>>>(partial, blocks, 2, C, sums);
  BwdArgs b;
  b.x = x_dev, b.y = y_dev, b.dy = dy_dev, b.mean = save_mean_dev, b.rstd = save_rstd_dev, b.gamma = gamma_dev;
  b.sums = sums, b.param_sums = sums, b.n_stat = (double)n, b.zero_row = zero_row;
  b.ldx = ldx, b.ldy = ldy, b.lddy = lddy, b.lddx = lddx, b.lddres = lddres, b.n = (int)n, b.C = C;
  b.relu = relu, b.dx = dx_dev, b.dres = dres_dev, b.dgamma = dgamma_dev, b.dbeta = dbeta_dev;
  const size_t total = (size_t)n * (C / 4);
  k_bn_bwd_apply<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(b);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

// column sums of a [n][C] matrix (bias gradient of lin_squeeze_head: sum over rows of dy)
// ---- the same BatchNorm in pieces, for statistics that span the data-parallel ranks (SyncBN): the caller exchanges
// the per-rank numbers (torch.distributed) between the calls.  Reference semantics: ME.MinkowskiBatchNorm normalises
// over ALL rows of the batch on one device (models/modules/common.py:20-22); with one scene per rank the strict
// equivalent needs [2C+1] numbers per layer across the ranks.
//   a3d_bn_local_stats     : stats[0..C) = mean of this rank's rows, stats[C..2C) = sum (x - that mean)^2   (fp64)
//   a3d_bn_apply           : y = relu?((x - mean) rstd gamma + beta (+ res)) with the GIVEN mean / rstd
//   a3d_bn_backward_sums   : sums[0..C) = sum g, sums[C..2C) = sum g xhat over this rank's rows (fp64), g = dy (y > 0)
//   a3d_bn_backward_apply  : dx from the GLOBAL sums and row count; dgamma / dbeta = the LOCAL sums (averaged over the
//                            ranks with every other parameter gradient afterwards, as DDP + SyncBatchNorm do)
extern "C" int a3d_bn_local_stats(const float* x_dev, int ldx, int64_t n, int C, double* stats_dev, void* workspace_dev,
                                  size_t workspace_bytes, void* stream) {
  if (!x_dev || !stats_dev || !workspace_dev || !bn_shape_ok(n, C, ldx, 4, 4)) {
    set_error("a3d_bn_local_stats: bad arguments");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < a3d_bn_workspace_bytes(n, C) || ((uintptr_t)workspace_dev & 15)) {
    set_error("a3d_bn_local_stats: workspace too small or misaligned");
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace_dev;
  double* sums = (double*)((char*)workspace_dev + align256((size_t)kBnMaxBlocks * 2 * C * sizeof(float)));
  float* meanf = (float*)((char*)sums + align256((size_t)2 * C * sizeof(double)));
  ColArgs c;
  memset(&c, 0, sizeof(c));
  c.x = x_dev, c.ldx = ldx, c.n = (int)n, c.C = C, c.partial = partial;
  const int blocks = bn_blocks(n, c.rows_per_block);
  const unsigned cb = (unsigned)((2 * C + 255) / 256);
  c.mode = 0;
  k_col_partial<<<blocks, kBnThreads, 0, st>>>(c);
  const unsigned fb = (unsigned)((C + kFinCols - 1) / kFinCols);
  k_col_final<<<fb, 256, 0, st>>>(partial, blocks, 1, C, sums);
  k_bn_mean<<<cb, 256, 0, st>>>(sums, C, (double)n, meanf);
  k_scale_f64<<<cb, 256, 0, st>>>(sums, C, 1.0 / (double)n, stats_dev);              // the mean in fp64
  c.mode = 1, c.mean = meanf;
  k_col_partial<<<blocks, kBnThreads, 0, st>>>(c);
  k_col_final<<<fb, 256, 0, st>>>(partial, blocks, 1, C, stats_dev + C);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_bn_apply(const float* x_dev, int ldx, int64_t n, int C, const float* gamma_dev, const float* beta_dev,
                            const float* mean_dev, const float* rstd_dev, const float* res_dev, int ldr, int relu,
                            float* y_dev, int ldy, int y_zero_row, void* stream) {
  if (!x_dev || !gamma_dev || !beta_dev || !mean_dev || !rstd_dev || !y_dev || !bn_shape_ok(n, C, ldx, ldy, res_dev ? ldr : 4)) {
    set_error("a3d_bn_apply: bad arguments");
    return A3D_ERR_INVALID;
  }
  ApplyArgs a;
  a.x = x_dev, a.mean = mean_dev, a.rstd = rstd_dev, a.gamma = gamma_dev, a.beta = beta_dev, a.res = res_dev;
  a.ldx = ldx, a.ldr = ldr, a.ldy = ldy, a.n = (int)n, a.C = C, a.relu = relu, a.y = y_dev, a.zero_row = y_zero_row;
  const size_t total = (size_t)n * (C / 4);
  k_bn_apply<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(a);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_bn_backward_sums(const float* x_dev, int ldx, const float* y_dev, int ldy, const float* dy_dev, int lddy,
                                    int64_t n, int C, const float* mean_dev, const float* rstd_dev, int relu,
                                    double* sums_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!x_dev || !dy_dev || !mean_dev || !rstd_dev || !sums_dev || !workspace_dev || (relu && !y_dev) ||
      !bn_shape_ok(n, C, ldx, lddy, 4) || (relu && ldy % 4)) {
    set_error("a3d_bn_backward_sums: bad arguments");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < a3d_bn_workspace_bytes(n, C) || ((uintptr_t)workspace_dev & 15)) {
    set_error("a3d_bn_backward_sums: workspace too small or misaligned");
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace_dev;
  ColArgs c;
  memset(&c, 0, sizeof(c));
  c.x = x_dev, c.y = y_dev, c.dy = dy_dev, c.mean = mean_dev, c.rstd = rstd_dev;
  c.ldx = ldx, c.ldy = ldy, c.lddy = lddy, c.n = (int)n, c.C = C, c.relu = relu, c.mode = 2, c.partial = partial;
  const int blocks = bn_blocks(n, c.rows_per_block);
  k_col_partial<<<blocks, kBnThreads, 0, st>>>(c);
  k_col_final<<<(unsigned)((2 * C + kFinCols - 1) / kFinCols), 256, 0, st>>>(partial, blocks, 2, C, sums_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_bn_backward_apply(const float* x_dev, int ldx, const float* y_dev, int ldy, const float* dy_dev, int lddy,
                                     int64_t n, int C, const float* gamma_dev, const float* mean_dev, const float* rstd_dev,
                                     int relu, const double* global_sums_dev, int64_t n_global, const double* local_sums_dev,
                                     float* dx_dev, int lddx, float* dres_dev, int lddres, float* dgamma_dev,
                                     float* dbeta_dev, int zero_row, void* stream) {
  if (!x_dev || !dy_dev || !gamma_dev || !mean_dev || !rstd_dev || !global_sums_dev || !local_sums_dev || !dx_dev ||
      !dgamma_dev || !dbeta_dev || n_global < n || (relu && !y_dev) || !bn_shape_ok(n, C, ldx, lddy, lddx) ||
      (relu && ldy % 4) || (dres_dev && lddres % 4)) {
    set_error("a3d_bn_backward_apply: bad arguments");
    return A3D_ERR_INVALID;
  }
  BwdArgs b;
  b.x = x_dev, b.y = y_dev, b.dy = dy_dev, b.mean = mean_dev, b.rstd = rstd_dev, b.gamma = gamma_dev;
  b.sums = global_sums_dev, b.param_sums = local_sums_dev, b.n_stat = (double)n_global, b.zero_row = zero_row;
  b.ldx = ldx, b.ldy = ldy, b.lddy = lddy, b.lddx = lddx, b.lddres = lddres, b.n = (int)n, b.C = C;
  b.relu = relu, b.dx = dx_dev, b.dres = dres_dev, b.dgamma = dgamma_dev, b.dbeta = dbeta_dev;
  const size_t total = (size_t)n * (C / 4);
  k_bn_bwd_apply<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(b);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_column_sums(const float* x_dev, int ldx, int64_t n, int C, float* out_dev, void* workspace_dev,
                               size_t workspace_bytes, void* stream) {
  if (!x_dev || !out_dev || !workspace_dev || !bn_shape_ok(n, C, ldx, 4, 4)) {
    set_error("a3d_column_sums: bad arguments");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < a3d_bn_workspace_bytes(n, C) || ((uintptr_t)workspace_dev & 15)) {
    set_error("a3d_column_sums: workspace too small or misaligned");
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace_dev;
  ColArgs c;
  memset(&c, 0, sizeof(c));
  c.x = x_dev, c.ldx = ldx, c.n = (int)n, c.C = C, c.partial = partial, c.mode = 0;
  const int blocks = bn_blocks(n, c.rows_per_block);
  k_col_partial<<<blocks, kBnThreads, 0, st>>>(c);
  k_col_final_f32<<<(unsigned)((C + kFinCols - 1) / kFinCols), 256, 0, st>>>(partial, blocks, C, out_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_layernorm_forward(const float* x_dev, int ldx, int64_t n, int C, const float* gamma_dev,
                                     const float* beta_dev, float eps, float* y_dev, int ldy, void* stream) {
  if (!x_dev || !gamma_dev || !beta_dev || !y_dev || n <= 0 || n > (int64_t)1 << 30 || C < 64 || C % 64 ||
      C > 64 * kLnMaxPerLane || ldx < C || ldy < C) {
    set_error("a3d_layernorm_forward: bad arguments (C a multiple of 64, <= 512)");
    return A3D_ERR_INVALID;
  }
  LnArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x_dev, a.gamma = gamma_dev, a.beta = beta_dev, a.y = y_dev, a.ldx = ldx, a.ldy = ldy, a.n = (int)n, a.C = C, a.eps = eps;
  if (C == 128 && !(ldx & 3) && !(ldy & 3)) k_ln_forward128<<<(unsigned)((n + 15) / 16), 256, 0, (hipStream_t)stream>>>(a);
  else k_ln_forward<<<(unsigned)((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(a);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_layernorm_backward(const float* x_dev, int ldx, const float* dy_dev, int lddy, int64_t n, int C,
                                      const float* gamma_dev, float eps, float* dx_dev, float* dgamma_dev,
                                      float* dbeta_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!x_dev || !dy_dev || !gamma_dev || !dx_dev || !dgamma_dev || !dbeta_dev || !workspace_dev || n <= 0 ||
      n > (int64_t)1 << 30 || C < 64 || C % 64 || C > 64 * kLnMaxPerLane || ldx < C || lddy < C) {
    set_error("a3d_layernorm_backward: bad arguments (C a multiple of 64, <= 512; dx has the layout of x)");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < a3d_bn_workspace_bytes(n, C) || ((uintptr_t)workspace_dev & 15)) {
    set_error("a3d_layernorm_backward: workspace too small or misaligned (a3d_bn_workspace_bytes)");
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace_dev;
  double* sums = (double*)((char*)workspace_dev + align256((size_t)kBnMaxBlocks * 2 * C * sizeof(float)));
  LnArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x_dev, a.dy = dy_dev, a.gamma = gamma_dev, a.dx = dx_dev, a.partial = partial;
  a.ldx = ldx, a.ldy = lddy, a.n = (int)n, a.C = C, a.eps = eps;
  const int blocks = bn_blocks(n, a.rows_per_block);
  if (C == 128 && !(ldx & 3) && !(lddy & 3)) k_ln_backward128<<<blocks, 256, 0, st>>>(a);
  else k_ln_backward<<<blocks, 256, 0, st>>>(a);
  k_col_final<<<(unsigned)((2 * C + kFinCols - 1) / kFinCols), 256, 0, st>>>(partial, blocks, 2, C, sums);
  k_ln_params<<<(unsigned)((C + 255) / 256), 256, 0, st>>>(sums, C, dgamma_dev, dbeta_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
