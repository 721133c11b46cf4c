// Calibration microbenchmark: fp32 MFMA issue rate on gfx950 (not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
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

template <int NACC, bool LDS>
void run(const char* name, int blocks_per_cu) {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 4000, grid = 256 * blocks_per_cu;
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
    run<1, false>("1 acc (dependent chain)", b);
    run<2, false>("2 acc", b);
    run<4, false>("4 acc", b);
    run<12, false>("12 acc", b);
    run<12, true>("12 acc + 2 ds_read_b128/48", b);
    run<4, true>("4 acc + 2 ds_read_b128/16", b);
  }
  return 0;
}
