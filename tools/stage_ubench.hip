// Calibration: the conv kernel's stage loop without any global traffic -- what does [36 ds_read_b128 + 144 MFMA
// (+ barrier)] per wave and stage sustain with 1 / 2 workgroups per CU?  (not part of the library)
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/stage_ubench.hip -o /tmp/stage_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCT, int NS, int MODE, bool BAR, bool LDSRD>   // MODE 0: step-wise double buffer (RG=1); 1: fragment at a time, 2 groups
__global__ void __launch_bounds__(256, 2) kst(float* out, int stages, float seed, int lds_floats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wring = (float*)smem;
  for (int i = threadIdx.x; i < lds_floats; i += 256) wring[i] = seed * (i & 1023);
  __syncthreads();
  constexpr int WF = NS * NCT * 256;
  const int lane = threadIdx.x & 63;
  constexpr int RG = MODE == 1 ? 2 : 1;
  f32x4 acc[RG][NCT];
  for (int r = 0; r < RG; ++r)
    for (int i = 0; i < NCT; ++i) acc[r][i] = (f32x4){0, 0, 0, 0};
  f32x4 A[RG][NS];
  for (int r = 0; r < RG; ++r)
    for (int s = 0; s < NS; ++s) A[r][s] = (f32x4){seed + r, 1.f + s, 2.f, 3.f + lane};
  for (int st = 0; st < stages; ++st) {
    const f32x4* Ws = (const f32x4*)(wring + (st & 1) * WF) + lane;
    if (MODE == 0) {
      f32x4 b[2][NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) b[0][ct] = LDSRD ? Ws[ct * 64] : (f32x4){1.f, seed, 2.f, 3.f};
#pragma unroll
      for (int Sx = 0; Sx < NS; ++Sx) {
        if (Sx + 1 < NS) {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) b[(Sx + 1) & 1][ct] = LDSRD ? Ws[((Sx + 1) * NCT + ct) * 64] : b[Sx & 1][ct];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
            acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[Sx & 1][ct][tt], A[0][Sx][tt], acc[0][ct], 0, 0, 0);
      }
    } else {
      f32x4 bc = Ws[0];
#pragma unroll
      for (int Sx = 0; Sx < NS; ++Sx) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          f32x4 bn = bc;
          if (Sx * NCT + ct + 1 < NS * NCT) bn = LDSRD ? Ws[(Sx * NCT + ct + 1) * 64] : bc;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < RG; ++r)
              acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(bc[tt], A[r][Sx][tt], acc[r][ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          bc = bn;
        }
      }
    }
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  f32x4 sum = (f32x4){0, 0, 0, 0};
  for (int r = 0; r < RG; ++r)
    for (int i = 0; i < NCT; ++i) sum += acc[r][i];
  out[blockIdx.x * 256 + threadIdx.x] = sum[0] + sum[1] + sum[2] + sum[3];
}

template <int NCT, int NS, int MODE, bool BAR, bool LDSRD>
void run(const char* name, int wg_per_cu) {
  float* out;
  hipMalloc(&out, 512 * 256 * 4);
  const int stages = 600, grid = 256 * wg_per_cu;
  const int lds_floats = 2 * NS * NCT * 256;
  const size_t lds = wg_per_cu == 1 ? 100 * 1024 : (size_t)lds_floats * 4 + 64;   // 1 per CU: make a second one impossible
  hipFuncSetAttribute((const void*)kst<NCT, NS, MODE, BAR, LDSRD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kst<NCT, NS, MODE, BAR, LDSRD><<<grid, 256, lds>>>(out, 10, 1.f, lds_floats);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kst<NCT, NS, MODE, BAR, LDSRD><<<grid, 256, lds>>>(out, stages, 1.f, lds_floats);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int rg = MODE == 1 ? 2 : 1;
  const double flops = (double)grid * 4 * stages * NS * 4 * NCT * rg * 2048.0;
  printf("%-58s WG/CU=%d  %7.1f TFLOP/s\n", name, wg_per_cu, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int b = 1; b <= 2; ++b) {
    run<6, 6, 0, false, false>("96x96 stage, MFMA only (no LDS reads, no barrier)", b);
    run<6, 6, 0, false, true>("96x96 stage, LDS weight reads, no barrier", b);
    run<6, 6, 0, true, true>("96x96 stage, LDS weight reads + barrier per stage", b);
    run<6, 6, 1, true, true>("96x96 stage, 2 groups/wave fragment-at-a-time + barrier", b);
    run<6, 6, 1, false, true>("96x96 stage, 2 groups/wave fragment-at-a-time, no barrier", b);
    run<8, 4, 0, true, true>("128x64 stage, LDS weight reads + barrier", b);
    run<4, 4, 0, true, true>("64x64 stage, LDS weight reads + barrier", b);
  }
  return 0;
}
