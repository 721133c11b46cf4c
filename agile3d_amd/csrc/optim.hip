// libagile3d_hip -- the optimiser step of the training path: AdamW and the global
// gradient-norm clip of the reference's loop (main.py:125-127: torch.optim.AdamW(lr, weight_decay);
// engine.py:145-150: clip_grad_norm_(max_norm) then optimizer.step()).  Element-wise, HBM-bound.
//   a3d_sum_squares : sum_i g_i^2 of one tensor (fp64 result, deterministic two-stage reduction); the host adds the
//                     tensors' sums, norm = sqrt(total), clip coefficient = min(1, max_norm / (norm + 1e-6))
//   a3d_adamw_step  : torch's single-tensor AdamW update with the clip coefficient folded into the gradient read:
//                     p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//                     p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "common.h"

namespace a3d {

constexpr int kSqBlocks = 512;

__global__ void __launch_bounds__(256) k_sumsq_partial(const float* __restrict__ g, size_t n, double* partial) {
  __shared__ double red[256];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double v = (double)g[i];
    s += v * v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int stride = 128; stride >= 1; stride >>= 1) {
    if (threadIdx.x < stride) red[threadIdx.x] += red[threadIdx.x + stride];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void k_sumsq_final(const double* partial, int nb, double* out, int accumulate) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = accumulate ? *out : 0.0;
    for (int b = 0; b < nb; ++b) s += partial[b];
    *out = s;
  }
}

__global__ void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                        size_t n, float grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                        float bias1, float bias2_sqrt) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * grad_scale;
  float pi = p[i] * (1.f - lr * weight_decay);
  const float mi = beta1 * m[i] + (1.f - beta1) * gi;          // exp_avg.lerp_(grad, 1 - beta1)
  const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
  const float denom = sqrtf(vi) / bias2_sqrt + eps;
  pi -= (lr / bias1) * (mi / denom);
  p[i] = pi;
  m[i] = mi;
  v[i] = vi;
}

// ---- all parameter tensors in one launch: a workgroup handles one 4096-element chunk, its tensor found by a binary
// search over the tensors' first-chunk numbers (a training step touches 268 tensors: 2 x 268 + 268 launches otherwise)
__device__ __forceinline__ int mt_find(const a3d_mt_tensor* __restrict__ tab, int nt, int chunk) {
  int lo = 0, hi = nt;   // tab[lo].chunk0 <= chunk < tab[hi].chunk0
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tab[mid].chunk0 <= chunk) lo = mid; else hi = mid;
  }
  return lo;
}
__global__ void __launch_bounds__(256) k_mt_sumsq(const a3d_mt_tensor* __restrict__ tab, int nt, double* partial) {
  __shared__ double red[256];
  const int ti = mt_find(tab, nt, blockIdx.x);
  const a3d_mt_tensor t = tab[ti];
  const int64_t base = (int64_t)(blockIdx.x - t.chunk0) * A3D_MT_CHUNK;
  double s = 0.0;
  for (int e = threadIdx.x; e < A3D_MT_CHUNK; e += 256) {
    const int64_t i = base + e;
    if (i < t.n) {
      const double v = (double)t.g[i];
      s += v * v;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int stride = 128; stride >= 1; stride >>= 1) {
    if (threadIdx.x < stride) red[threadIdx.x] += red[threadIdx.x + stride];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(1024) k_mt_sumsq_final(const double* __restrict__ partial, int64_t nchunks, double* out) {
  __shared__ double red[1024];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < nchunks; i += 1024) s += partial[i];   // fixed assignment: deterministic
  red[threadIdx.x] = s;
  __syncthreads();
  for (int stride = 512; stride >= 1; stride >>= 1) {
    if (threadIdx.x < stride) red[threadIdx.x] += red[threadIdx.x + stride];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}
__global__ void __launch_bounds__(256) k_mt_adamw(const a3d_mt_tensor* __restrict__ tab, int nt, float grad_scale, float lr,
                                                  float beta1, float beta2, float eps, float weight_decay) {
  const int ti = mt_find(tab, nt, blockIdx.x);
  const a3d_mt_tensor t = tab[ti];
  const int64_t base = (int64_t)(blockIdx.x - t.chunk0) * A3D_MT_CHUNK;
  for (int e = threadIdx.x; e < A3D_MT_CHUNK; e += 256) {
    const int64_t i = base + e;
    if (i >= t.n) break;
    const float gi = t.g[i] * grad_scale;   // the same arithmetic, in the same order, as k_adamw
    float pi = t.p[i] * (1.f - lr * weight_decay);
    const float mi = beta1 * t.m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * t.v[i] + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / t.bias2_sqrt + eps;
    pi -= (lr / t.bias1) * (mi / denom);
    t.p[i] = pi;
    t.m[i] = mi;
    t.v[i] = vi;
  }
}

}  // namespace a3d

using namespace a3d;

extern "C" size_t a3d_sum_squares_workspace_bytes(void) { return (size_t)(kSqBlocks + 1) * sizeof(double) + 256; }

extern "C" int a3d_sum_squares(const float* g_dev, int64_t n, double* out_host, void* workspace_dev,
                               size_t workspace_bytes, void* stream) {
  if (!g_dev || n <= 0 || !out_host || !workspace_dev || workspace_bytes < a3d_sum_squares_workspace_bytes() ||
      ((uintptr_t)workspace_dev & 7)) {
    set_error("a3d_sum_squares: bad arguments");
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)workspace_dev;
  int nb = (int)((n + 256 * 8 - 1) / (256 * 8));
  nb = nb < 1 ? 1 : nb > kSqBlocks ? kSqBlocks : nb;
  k_sumsq_partial<<<nb, 256, 0, st>>>(g_dev, (size_t)n, partial);
  k_sumsq_final<<<1, 64, 0, st>>>(partial, nb, partial + kSqBlocks, 0);
  A3D_LAUNCH_CHECK();
  A3D_HIP_CHECK(hipMemcpyAsync(out_host, partial + kSqBlocks, sizeof(double), hipMemcpyDeviceToHost, st));
  A3D_HIP_CHECK(hipStreamSynchronize(st));
  return A3D_OK;
}

// the same sum added to *acc_dev (a device double the caller zeroes first): no host synchronisation, so the norm over
// hundreds of parameter tensors costs one read-back instead of one per tensor
extern "C" int a3d_sum_squares_accumulate(const float* g_dev, int64_t n, double* acc_dev, void* workspace_dev,
                                          size_t workspace_bytes, void* stream) {
  if (!g_dev || n <= 0 || !acc_dev || !workspace_dev || workspace_bytes < a3d_sum_squares_workspace_bytes() ||
      ((uintptr_t)workspace_dev & 7)) {
    set_error("a3d_sum_squares_accumulate: bad arguments");
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)workspace_dev;
  int nb = (int)((n + 256 * 8 - 1) / (256 * 8));
  nb = nb < 1 ? 1 : nb > kSqBlocks ? kSqBlocks : nb;
  k_sumsq_partial<<<nb, 256, 0, st>>>(g_dev, (size_t)n, partial);
  k_sumsq_final<<<1, 64, 0, st>>>(partial, nb, acc_dev, 1);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_adamw_step(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev,
                              int64_t n, int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                              float grad_scale, void* stream) {
  if (!param_dev || !grad_dev || !exp_avg_dev || !exp_avg_sq_dev || n <= 0 || step < 1) {
    set_error("a3d_adamw_step: bad arguments (step counts from 1)");
    return A3D_ERR_INVALID;
  }
  // bias corrections in double on the host, like torch's scalar path
  const double b1 = 1.0 - pow((double)beta1, (double)step), b2 = 1.0 - pow((double)beta2, (double)step);
  k_adamw<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(param_dev, grad_dev, exp_avg_dev, exp_avg_sq_dev,
                                                                       (size_t)n, grad_scale, lr, beta1, beta2, eps,
                                                                       weight_decay, (float)b1, (float)sqrt(b2));
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" size_t a3d_mt_workspace_bytes(int64_t n_chunks) {
  if (n_chunks < 0) return 0;
  return (size_t)(n_chunks + 1) * sizeof(double) + 256;
}

extern "C" int a3d_sum_squares_multi(const a3d_mt_tensor* table_dev, int n_tensors, int64_t n_chunks, double* out_dev,
                                     void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!table_dev || n_tensors < 1 || n_chunks < 1 || n_chunks > 0x7fffffff || !out_dev || !workspace_dev ||
      workspace_bytes < a3d_mt_workspace_bytes(n_chunks) || ((uintptr_t)workspace_dev & 7)) {
    set_error("a3d_sum_squares_multi: bad arguments");
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)workspace_dev;
  k_mt_sumsq<<<(unsigned)n_chunks, 256, 0, st>>>(table_dev, n_tensors, partial);
  k_mt_sumsq_final<<<1, 1024, 0, st>>>(partial, n_chunks, out_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_adamw_step_multi(const a3d_mt_tensor* table_dev, int n_tensors, int64_t n_chunks, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  if (!table_dev || n_tensors < 1 || n_chunks < 1 || n_chunks > 0x7fffffff) {
    set_error("a3d_adamw_step_multi: bad arguments");
    return A3D_ERR_INVALID;
  }
  k_mt_adamw<<<(unsigned)n_chunks, 256, 0, (hipStream_t)stream>>>(table_dev, n_tensors, grad_scale, lr, beta1, beta2, eps,
                                                                  weight_decay);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
