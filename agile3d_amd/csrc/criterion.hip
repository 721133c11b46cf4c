// criterion.hip -- the mask losses of the training/validation loop and their gradient w.r.t. the logits (gfx950).
// Replaces: SetCriterion.loss_bce / loss_dice / multiclass_dice_loss / dice_loss (models/criterion.py:14-110) as
// engine.py:126-128,238-240 use them (first piece of SURVEY.md section 8 row f-2).
//
// For one sample with logits z [n][C], targets t [n], click weights w [n]   (p = softmax(z_i)):
//   loss_bce  = mean_i  w_i * (-log p_i[t_i])                                   (F.cross_entropy, reduction none)
//   loss_dice = mean_i  w_i * d_i,  num_i = 2 p_i[t_i] / C, den_i = (sum_c p_i[c] + 1) / C,
//               d_i = num_i > eps ? 1 - (num_i + eps) / (den_i + eps) : 0        (the reference flattens per POINT)
// grad = d(coef_bce * loss_bce + coef_dice * loss_dice) / dz   (sum_c p = 1, so den carries no gradient).
#include "common.h"

namespace a3d {

constexpr int kMaxClasses = 64;

__global__ void __launch_bounds__(256) k_mask_losses(const float* __restrict__ z, const int32_t* __restrict__ t,
                                                     const float* __restrict__ w, int64_t n, int C, float eps,
                                                     float coef_bce, float coef_dice, double* sums, float* grad,
                                                     int* err) {
  __shared__ double red[2][4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double bce = 0.0, dice = 0.0;
  if (i < n) {
    const float* zi = z + i * C;
    float p[kMaxClasses];
    float m = zi[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, zi[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
      p[c] = expf(zi[c] - m);
      s += p[c];
    }
    const float inv = 1.f / s;
    float psum = 0.f;
    for (int c = 0; c < C; ++c) {
      p[c] *= inv;
      psum += p[c];
    }
    const int ti = t[i];
    if (ti < 0 || ti >= C) {
      atomicOr(err, 1);
    } else {
      const float wi = w ? w[i] : 1.f;
      const float logp = (zi[ti] - m) - logf(s);
      bce = (double)(-logp * wi);
      const float num = 2.f * p[ti] / (float)C;
      const float den = (psum + 1.f) / (float)C;
      const bool on = num > eps;
      const float d = on ? 1.f - (num + eps) / (den + eps) : 0.f;
      dice = (double)(d * wi);
      if (grad) {
        const float gb = coef_bce * wi / (float)n;
        const float gd = on ? -coef_dice * wi / (float)n * (2.f / (float)C) / (den + eps) * p[ti] : 0.f;
        for (int c = 0; c < C; ++c) {
          const float ind = c == ti ? 1.f : 0.f;
          grad[i * C + c] = gb * (p[c] - ind) + gd * (ind - p[c]);
        }
      }
    }
  }
  // block reduction in double, one atomic per block and loss (summation order across blocks is not fixed:
  // the sums are accumulated in fp64 so the fp32 result is stable to the last bit in practice)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    bce += __shfl_xor(bce, o, 64);
    dice += __shfl_xor(dice, o, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = bce;
    red[1][wave] = dice;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&sums[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

__global__ void k_loss_finish(const double* sums, int64_t n, float* out, const int* err) {
  if (threadIdx.x < 2) out[threadIdx.x] = *err ? __int_as_float(0x7fc00000) : (float)(sums[threadIdx.x] / (double)n);
}

}  // namespace a3d

using namespace a3d;

extern "C" int a3d_mask_losses(const float* logits_dev, const int32_t* target_dev, const float* weights_dev, int64_t n,
                               int n_classes, float coef_bce, float coef_dice, float* losses_dev,
                               float* grad_logits_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!logits_dev || !target_dev || !losses_dev || n <= 0 || n_classes < 1 || n_classes > kMaxClasses) {
    set_error("a3d_mask_losses: bad arguments (n=%lld classes=%d, at most %d classes)", (long long)n, n_classes,
              kMaxClasses);
    return A3D_ERR_INVALID;
  }
  if (!workspace_dev || workspace_bytes < 64) {
    set_error("a3d_mask_losses: workspace needs 64 bytes");
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  A3D_HIP_CHECK(hipMemsetAsync(workspace_dev, 0, 64, st));
  double* sums = (double*)workspace_dev;
  int* err = (int*)((char*)workspace_dev + 32);
  k_mask_losses<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(logits_dev, target_dev, weights_dev, n, n_classes, 1e-6f,
                                                            coef_bce, coef_dice, sums, grad_logits_dev, err);
  k_loss_finish<<<1, 64, 0, st>>>(sums, n, losses_dev, err);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
