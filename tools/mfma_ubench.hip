// Calibration microbenchmark: fp32 MFMA issue rate on gfx950 (not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * i;
  __syncthreads();
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  f32x4 a = (f32x4){seed, 1.f, 2.f, 3.f}, b = (f32x4){1.f, seed, 0.5f, 0.25f};
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
      a = *(const f32x4*)(lds + ((it * 64 + lane) & 1023) * 4);
      b = *(const f32x4*)(lds + ((it * 64 + lane + 512) & 1023) * 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc[i], 0, 0, 0);
  }
  f32x4 s = (f32x4){0, 0, 0, 0};
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// the conv kernel's stage shape: NRD ds_read_b128 (1 A + NCT B per half) then 8*NCT MFMAs
template <int NCT>
__global__ void __launch_bounds__(256) kstage(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[12288];
  for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = seed * i;
  __syncthreads();
  f32x4 acc[NCT];
  for (int i = 0; i < NCT; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  for (int it = 0; it < iters; ++it) {
    const float* Ab = lds + ((it & 1) * 2048) + (16 * wave + j) * 32;
    const float* Wb = lds + 4096 + (it & 1) * 2 * NCT * 256 + lane * 4;
    f32x4 av[2], bv[2][NCT];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      av[s] = *(const f32x4*)(Ab + ((4 * s + g) ^ (j & 7)) * 4);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) bv[s][ct] = *(const f32x4*)(Wb + (s * NCT + ct) * 256);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], bv[s][ct][t], acc[ct], 0, 0, 0);
  }
  f32x4 sum = (f32x4){0, 0, 0, 0};
  for (int i = 0; i < NCT; ++i) sum += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = sum[0] + sum[1] + sum[2] + sum[3];
}
template <int NCT>
void run_stage(int blocks_per_cu) {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = getenv("UB_ITERS") ? atoi(getenv("UB_ITERS")) : 4000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kstage<NCT><<<grid, 256>>>(out, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kstage<NCT><<<grid, 256>>>(out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 8 * NCT * 2048.0;
  printf("conv-stage shape NCT=%d (%d ds_read_b128 + %d MFMA) blocks/CU=%d  %7.1f TFLOP/s\n", NCT, 2 + 2 * NCT, 8 * NCT,
         blocks_per_cu, flops / ms / 1e9);
  hipFree(out);
}

template <int NACC, bool LDS>
void run(const char* name, int blocks_per_cu) {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = getenv("UB_ITERS") ? atoi(getenv("UB_ITERS")) : 4000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<NACC, LDS><<<grid, 256>>>(out, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, LDS><<<grid, 256>>>(out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 /*waves*/ * iters * 4 * NACC * 2048.0;
  printf("%-28s blocks/CU=%d  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", name, blocks_per_cu, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * iters * 4 * NACC));
  hipFree(out);
}

int main() {
  for (int b = 1; b <= 3; ++b) {
    run_stage<2>(b);
    run_stage<6>(b);
    run_stage<8>(b);
  }
  for (int b = 1; b <= 3; ++b) {
    run<1, false>("1 acc (dependent chain)", b);
    run<2, false>("2 acc", b);
    run<4, false>("4 acc", b);
    run<12, false>("12 acc", b);
    run<12, true>("12 acc + 2 ds_read_b128/48", b);
    run<4, true>("4 acc + 2 ds_read_b128/16", b);
  }
  return 0;
}
