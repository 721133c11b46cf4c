// Calibration microbenchmark (not part of the library): can the fp32 VALU run NEXT TO the fp32 matrix pipe on gfx950?
// Per iteration 24 independent v_mfma_f32_16x16x4_f32 (6 accumulators x 4, the conv stage's shape) and NV vector FMAs on
// registers of their own, interleaved in program order (NV / 24 after every MFMA).  PK: v_pk_fma_f32 (2 FMAs per lane and
// instruction) instead of v_fma_f32.  Reported: time per iteration against the MFMA-only loop, and the sum of both pipes'
// FLOP/s.  hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/coissue_ubench.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, bool PK, bool MFMA>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  f32x4 acc[6];
  for (int i = 0; i < 6; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  f32x4 a = (f32x4){seed, 1.f, 2.f, 3.f}, b = (f32x4){1.f, seed, 0.5f, 0.25f};
  constexpr int NR = 16;               // vector accumulators (independent chains)
  f32x2 v[NR];
  for (int i = 0; i < NR; ++i) v[i] = (f32x2){seed * i, seed + i};
  f32x2 x = (f32x2){seed, 0.5f}, w = (f32x2){0.999f, 1.001f};
  constexpr int PER = NV / 24;
  for (int it = 0; it < iters; ++it) {
    int vi = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (MFMA) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          f32x2& r = v[vi % NR];
          if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(x), "v"(w));
          else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[0]) : "v"(x[0]), "v"(w[0]));
          ++vi;
        }
      }
  }
  f32x4 s = (f32x4){0, 0, 0, 0};
  for (int i = 0; i < 6; ++i) s += acc[i];
  float t = s[0] + s[1] + s[2] + s[3];
  for (int i = 0; i < NR; ++i) t += v[i][0] + v[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int NV, bool PK, bool MFMA>
double run(int blocks_per_cu, const char* name, double base_ms) {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 20000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<NV, PK, MFMA><<<grid, 256>>>(out, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NV, PK, MFMA><<<grid, 256>>>(out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)grid * 4;
  const double mf = MFMA ? waves * iters * 24 * 2048.0 : 0.0;
  const double vf = waves * iters * (double)NV * 64 * 2 * (PK ? 2 : 1);
  printf("%-34s waves/SIMD=%d  %8.3f ms  x%.3f of MFMA-only   MFMA %6.1f + VALU %6.1f = %6.1f TFLOP/s\n", name, blocks_per_cu, ms,
         base_ms > 0 ? ms / base_ms : 1.0, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
  hipFree(out);
  return ms;
}

// ---- the same question for the memory instructions a matrix-bound loop carries
// KIND 1: ds_read_b128 (conflict-free, results unused), 2: global_load_dwordx4 (one hot 1 KB line set)
template <int NX, int KIND>
__global__ void __launch_bounds__(256) kx(float* out, const float* src, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * i;
  __syncthreads();
  f32x4 acc[6];
  for (int i = 0; i < 6; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  f32x4 a = (f32x4){seed, 1.f, 2.f, 3.f}, b = (f32x4){1.f, seed, 0.5f, 0.25f};
  const int lane = threadIdx.x & 63;
  unsigned sreg = (unsigned)iters;
  const unsigned laddr = (unsigned)(size_t)lds + lane * 16;
  const float* gp = src + lane * 4;
  f32x4 sink = (f32x4){0, 0, 0, 0};
  constexpr int PER = NX / 24;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          if (KIND == 0) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sreg));
          else if (KIND == 1) { f32x4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(laddr)); sink[0] = r[0]; }
          else { f32x4 r; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(gp)); sink[0] = r[0]; }
        }
      }
    if (KIND != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  f32x4 s = (f32x4){0, 0, 0, 0};
  for (int i = 0; i < 6; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + (float)sreg + sink[0];
}
template <int NX, int KIND>
void runx(int blocks_per_cu, const char* name, double base_ms) {
  float *out, *src;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipMalloc(&src, 4096);
  hipMemset(src, 0, 4096);
  const int iters = 20000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kx<NX, KIND><<<grid, 256>>>(out, src, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kx<NX, KIND><<<grid, 256>>>(out, src, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s waves/SIMD=%d  %8.3f ms  x%.3f of MFMA-only\n", name, blocks_per_cu, ms, ms / base_ms);
  hipFree(out);
  hipFree(src);
}

int main() {
  for (int b = 1; b <= 4; b += 3) {
    const double base = run<0, false, true>(b, "24 MFMA", 0);
    runx<24, 1>(b, "24 MFMA + 24 ds_read_b128", base);
    runx<48, 1>(b, "24 MFMA + 48 ds_read_b128", base);
    runx<24, 2>(b, "24 MFMA + 24 global_load_dwordx4", base);
    runx<48, 2>(b, "24 MFMA + 48 global_load_dwordx4", base);
  }
  for (int b = 1; b <= 4; ++b) {
    const double base = run<0, false, true>(b, "24 MFMA", 0);
    run<24, false, true>(b, "24 MFMA + 24 v_fma", base);
    run<48, false, true>(b, "24 MFMA + 48 v_fma", base);
    run<96, false, true>(b, "24 MFMA + 96 v_fma", base);
    run<192, false, true>(b, "24 MFMA + 192 v_fma", base);
    run<24, true, true>(b, "24 MFMA + 24 v_pk_fma", base);
    run<48, true, true>(b, "24 MFMA + 48 v_pk_fma", base);
    run<96, true, true>(b, "24 MFMA + 96 v_pk_fma", base);
    run<192, true, true>(b, "24 MFMA + 192 v_pk_fma", base);
    run<192, false, false>(b, "192 v_fma alone", base);
    run<192, true, false>(b, "192 v_pk_fma alone", base);
  }
  return 0;
}
