// Experiment (not part of the library): fp32 products emulated with SIX bf16 MFMAs per 16x16x32 block
// (x = h + m + l, three bf16 planes each; h*h + h*m + m*h + h*l + m*m + l*h, fp32 accumulation) against the exact
// v_mfma_f32_16x16x4_f32 chain the conv kernel uses:  (1) accuracy of both against a float64 product,
// (2) what the conv kernel's stage loop (32 input channels x 96 columns per stage, weights from LDS, a barrier per
// stage, four workgroups per CU) would sustain in fp32-equivalent TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/bf16x6_ubench.hip -o /tmp/bf16x6_ubench && /tmp/bf16x6_ubench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// x -> three bf16 planes by truncation (every remainder is exact in fp32): x = h + m + l + O(2^-24 x)
__device__ __forceinline__ void split3(const float (&x)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    uint32_t hh[2], mm[2], ll[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float v = x[2 * p + e];
      const uint32_t hb = __float_as_uint(v) & 0xffff0000u;
      const float r1 = v - __uint_as_float(hb);
      const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
      const float r2 = r1 - __uint_as_float(mb);
      hh[e] = hb, mm[e] = mb, ll[e] = __float_as_uint(r2) & 0xffff0000u;
    }
    h[p] = (hh[0] >> 16) | hh[1];   // element 2p in the low half, 2p+1 in the high half
    m[p] = (mm[0] >> 16) | mm[1];
    l[p] = (ll[0] >> 16) | ll[1];
  }
}
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 mma6(const u32x4 (&w)[3], const u32x4 (&x)[3], f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w[2]), as_bf(x[0]), acc, 0, 0, 0);   // smallest terms first
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w[0]), as_bf(x[2]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w[1]), as_bf(x[1]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w[1]), as_bf(x[0]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w[0]), as_bf(x[1]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w[0]), as_bf(x[0]), acc, 0, 0, 0);
  return acc;
}

// ---- (1) accuracy: C[16][16] = W[16][K] X[K][16], K = 32 * kblocks, one wave
__global__ void k_acc(const float* __restrict__ W, const float* __restrict__ X, int K, float* Cf32, float* Cemu) {
  const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
  f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 4)   // exact fp32 MFMA: W is the A operand (lane: row j, k = k0 + g), X the B operand
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(W[j * K + k0 + g], X[(k0 + g) * 16 + j], a, 0, 0, 0);
  for (int k0 = 0; k0 < K; k0 += 32) {   // lane: row / column j, k = k0 + 8 g .. + 7
    float wv[8], xv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) wv[e] = W[j * K + k0 + 8 * g + e], xv[e] = X[(k0 + 8 * g + e) * 16 + j];
    u32x4 w[3], x[3];
    split3(wv, w[0], w[1], w[2]);
    split3(xv, x[0], x[1], x[2]);
    b = mma6(w, x, b);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {   // C/D: col = lane & 15, row = 4 (lane >> 4) + r
    Cf32[(4 * g + r) * 16 + j] = a[r];
    Cemu[(4 * g + r) * 16 + j] = b[r];
  }
}

// ---- (2) stage loop: per stage and wave 6 column tiles x (3 ds_read_b128 + 6 MFMA), the activation split once
template <bool EMU>
__global__ void __launch_bounds__(256, 4) k_stage(float* out, int stages, float seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kSlot = EMU ? 18 * 1024 : 12 * 1024;   // bytes of weights per stage: 3 bf16 planes vs fp32
  uint32_t* ring = (uint32_t*)smem;
  for (int i = threadIdx.x; i < 2 * kSlot / 4; i += 256) ring[i] = 0x3f803f80u + (i & 255);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[6];
  for (int i = 0; i < 6; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float xin[8];
  for (int e = 0; e < 8; ++e) xin[e] = seed + 0.37f * e + 0.01f * lane;
  for (int st = 0; st < stages; ++st) {
    const char* slot = (const char*)ring + (st & 1) * kSlot + lane * 16;
    if (EMU) {
      u32x4 x[3];
      split3(xin, x[0], x[1], x[2]);
#pragma unroll
      for (int ct = 0; ct < 6; ++ct) {
        u32x4 w[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) w[p] = *(const u32x4*)(slot + (ct * 3 + p) * 1024);
        acc[ct] = mma6(w, x, acc[ct]);
      }
    } else {   // the product kernel's PAIR build in miniature: 2 k-steps of 16 channels, fragments two at a time
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ct = 0; ct < 6; ct += 2) {
          const f32x4 c0 = *(const f32x4*)(slot + ((s * 6 + ct) * 1024)), c1 = *(const f32x4*)(slot + ((s * 6 + ct + 1) * 1024));
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[tt], xin[4 * s + tt], acc[ct], 0, 0, 0);
            acc[ct + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[tt], xin[4 * s + tt], acc[ct + 1], 0, 0, 0);
          }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) xin[e] += 1e-3f;   // next stage's rows differ
    __builtin_amdgcn_s_barrier();
  }
  f32x4 sum = {0, 0, 0, 0};
  for (int i = 0; i < 6; ++i) sum += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = sum[0] + sum[1] + sum[2] + sum[3];
}

template <bool EMU>
static double run_stage(const char* name) {
  float* out;
  hipMalloc(&out, 1024 * 256 * 4);
  const int stages = 2000, grid = 256 * 4;
  const size_t lds = EMU ? 2 * 18 * 1024 + 64 : 2 * 12 * 1024 + 64;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_stage<EMU><<<grid, 256, lds>>>(out, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_stage<EMU><<<grid, 256, lds>>>(out, stages, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * stages * 6 * (16.0 * 16 * 32 * 2);   // fp32-equivalent: one 16x16x32 block per tile
  printf("%-70s %7.1f TFLOP/s fp32-equivalent (%.3f ms)\n", name, flops / ms / 1e9, ms);
  hipFree(out);
  return flops / ms / 1e9;
}

int main() {
  // ---- accuracy
  const int K = 27 * 96;   // the reduction length of a 3^3 conv with 96 input channels
  float *W, *X, *C0, *C1;
  hipMallocManaged(&W, 16 * K * 4);
  hipMallocManaged(&X, K * 16 * 4);
  hipMallocManaged(&C0, 256 * 4);
  hipMallocManaged(&C1, 256 * 4);
  srand(1);
  for (int i = 0; i < 16 * K; ++i) W[i] = (float)((rand() / (double)RAND_MAX - 0.5) * 0.1), X[i] = (float)((rand() / (double)RAND_MAX - 0.5) * 4.0);
  k_acc<<<1, 64>>>(W, X, K, C0, C1);
  hipDeviceSynchronize();
  double e0 = 0, e1 = 0, scale = 0;
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) {
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)W[r * K + k] * (double)X[k * 16 + c];
      e0 = fmax(e0, fabs(C0[r * 16 + c] - ref));
      e1 = fmax(e1, fabs(C1[r * 16 + c] - ref));
      scale = fmax(scale, fabs(ref));
    }
  printf("K = %d products per output, |C| up to %.3f:  exact fp32 MFMA max |err| %.3e   six bf16 MFMAs max |err| %.3e\n", K, scale, e0, e1);
  // ---- stage loop
  run_stage<false>("fp32 MFMA 16x16x4, 32 ch x 96 col stage, LDS weights, barrier, 4 WG/CU");
  run_stage<true>("6 x bf16 MFMA 16x16x32 + split, same stage, 18 KB weight slot, 4 WG/CU");
  return 0;
}
