// gfx950 v_permlane16_swap / v_permlane32_swap as the cross-row (lane ^ 16, lane ^ 32) all-reduce used by the decoder
// kernels, checked against __shfl_xor.  hipcc --offload-arch=gfx950 -O3 tools/permlane_check.hip -o /tmp/plc && /tmp/plc
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float rows_max(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__global__ void k(const float* in, float* out) {
  const float x = in[threadIdx.x];
  float m = fmaxf(x, __shfl_xor(x, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float s = x + __shfl_xor(x, 16, 64);
  s += __shfl_xor(s, 32, 64);
  out[threadIdx.x] = m;
  out[64 + threadIdx.x] = rows_max(x);
  out[128 + threadIdx.x] = s;
  out[192 + threadIdx.x] = rows_sum(x);
}
int main() {
  float h[64], o[256], *di, *dout;
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 101) - 50.f + 0.25f * i;
  hipMalloc(&di, sizeof(h));
  hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) bad += (o[i] != o[64 + i]) + (o[128 + i] != o[192 + i]);
  printf("permlane swap all-reduce vs shfl_xor: %d mismatches\n", bad);
  return bad != 0;
}
