// quantize.hip -- on-device voxelisation (gfx950): ME.utils.sparse_quantize(coordinates, quantization_size,
// return_index=True, return_inverse=True) as the reference's dataset calls it
// (datasets/InterMultiObj3DSegDataset.py:67-75; semantics: SURVEY.md App. B.2).
//
//   q[i]        = int32(floor(xyz[i] / quantization_size))        computed in the INPUT's dtype (fp32 or fp64):
//                 numpy keeps float32 / python-float in float32, and voxel boundaries depend on it
//   voxels      = distinct q rows, ordered by the first point that falls into them
//   unique_map  = that first point of every voxel;  inverse_map[i] = voxel row of point i
//
// Integer work, bit-exact by construction: 63-bit packed keys -> stable radix sort of (key, point) pairs
// (radix.hip) -> run heads; the first element of a run is the smallest point index of the voxel; a scan of
// "is a voxel's first point" flags IN POINT ORDER numbers the voxels by first occurrence without a second sort.
#include "common.h"

namespace a3d {

constexpr int kQOff = 1 << 20;   // |q| < 2^20 per axis

template <typename T>
__global__ void k_quant_keys(const T* __restrict__ xyz, int64_t n, T qs, uint64_t* keys, int* vals, int* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const T f = floor(xyz[3 * i + a] / qs);
    int q = 0;
    if (!(f >= (T)(-kQOff) && f < (T)kQOff)) atomicOr(err, 1);   // also catches NaN / inf
    else q = (int)f;
    key = (key << 21) | (uint64_t)(q + kQOff);
  }
  keys[i] = key;
  vals[i] = (int)i;
}

// sorted order: head[p] = 1 where a new voxel starts; first[vals[p]] = 1 for those p (flags in POINT order)
__global__ void k_quant_heads(const uint64_t* __restrict__ keys, const int* __restrict__ vals, int64_t n,
                              int* head, int* first) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int h = p == 0 || keys[p] != keys[p - 1];
  head[p] = h;
  if (h) first[vals[p]] = 1;
}

// seg[p] (inclusive scan of head) - 1 = run index in sorted order; rank_first[i] = voxel number of the voxel whose
// first point is i (exclusive scan of first[] at i).  run -> voxel number through the run's first point.
__global__ void k_quant_emit(const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                             const int* __restrict__ head, const int* __restrict__ seg_incl,
                             const int* __restrict__ rank_first, int64_t n, int* run_voxel, int32_t* coords_out,
                             int64_t* unique_map) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || !head[p]) return;
  const int i = vals[p];
  const int v = rank_first[i];
  run_voxel[seg_incl[p] - 1] = v;
  unique_map[v] = i;
  const uint64_t key = keys[p];
  coords_out[3 * (size_t)v + 0] = (int)((key >> 42) & 0x1fffff) - kQOff;
  coords_out[3 * (size_t)v + 1] = (int)((key >> 21) & 0x1fffff) - kQOff;
  coords_out[3 * (size_t)v + 2] = (int)(key & 0x1fffff) - kQOff;
}

__global__ void k_quant_inverse(const int* __restrict__ vals, const int* __restrict__ seg_incl,
                                const int* __restrict__ run_voxel, int64_t n, int64_t* inverse_map) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  inverse_map[vals[p]] = run_voxel[seg_incl[p] - 1];
}

struct QuantWs {
  uint64_t *keys_in, *keys;
  int *vals_in, *vals, *head, *first, *seg, *rank, *run_voxel, *flags;
  void* temp;
  size_t temp_bytes, bytes;
};
static QuantWs carve_quant(void* base, int64_t n) {
  QuantWs w;
  size_t off = 0;
  auto take = [&](size_t b) {
    void* p = base ? (char*)base + off : nullptr;
    off += align256(b);
    return p;
  };
  w.flags = (int*)take(256);
  w.keys_in = (uint64_t*)take((size_t)n * 8);
  w.keys = (uint64_t*)take((size_t)n * 8);
  w.vals_in = (int*)take((size_t)n * 4);
  w.vals = (int*)take((size_t)n * 4);
  w.head = (int*)take((size_t)n * 4);
  w.first = (int*)take((size_t)n * 4);
  w.seg = (int*)take((size_t)n * 4);
  w.rank = (int*)take((size_t)n * 4);
  w.run_voxel = (int*)take((size_t)n * 4);
  const size_t t1 = radix_sort_temp_bytes((int)n), t2 = scan2_temp_bytes(n);
  w.temp_bytes = (t1 > t2 ? t1 : t2) + 1024;
  w.temp = take(w.temp_bytes);
  w.bytes = off;
  return w;
}

template <typename T>
static int run_quantize(const T* xyz, int64_t n, double qs, int32_t* coords_out, int64_t* unique_map,
                        int64_t* inverse_map, int64_t* n_voxels, QuantWs& w, hipStream_t st) {
  const unsigned nb = (unsigned)((n + 255) / 256);
  A3D_HIP_CHECK(hipMemsetAsync(w.flags, 0, 256, st));
  A3D_HIP_CHECK(hipMemsetAsync(w.first, 0, (size_t)n * 4, st));
  k_quant_keys<T><<<nb, 256, 0, st>>>(xyz, n, (T)qs, w.keys_in, w.vals_in, w.flags);
  A3D_LAUNCH_CHECK();
  {
    RadixPass ps[kRadixMaxPasses];
    const int np = radix_passes(0, 63, ps);   // digits the cloud's extent does not touch are skipped on the device
    const int rc = radix_sort_pairs(w.temp, w.temp_bytes, w.keys_in, w.keys, w.vals_in, w.vals, (int)n, ps, np, st, w.flags);
    if (rc) return rc;
  }
  k_quant_heads<<<nb, 256, 0, st>>>(w.keys, w.vals, n, w.head, w.first);
  A3D_LAUNCH_CHECK();
  {
    const int rc = scan2_incl_excl(w.temp, w.temp_bytes, w.head, w.first, n, w.seg, w.rank, st);
    if (rc) return rc;
  }
  k_quant_emit<<<nb, 256, 0, st>>>(w.keys, w.vals, w.head, w.seg, w.rank, n, w.run_voxel, coords_out, unique_map);
  k_quant_inverse<<<nb, 256, 0, st>>>(w.vals, w.seg, w.run_voxel, n, inverse_map);
  A3D_LAUNCH_CHECK();
  int host[2];   // [0] = error flag, [1] = number of voxels (last element of the inclusive scan)
  A3D_HIP_CHECK(hipMemcpyAsync(&host[0], w.flags, 4, hipMemcpyDeviceToHost, st));
  A3D_HIP_CHECK(hipMemcpyAsync(&host[1], w.seg + (n - 1), 4, hipMemcpyDeviceToHost, st));
  A3D_HIP_CHECK(hipStreamSynchronize(st));
  if (host[0] < 0) {   // only the sort's look-back writes a negative code here
    set_error("a3d_sparse_quantize: the device-side sort gave up waiting for a workgroup (starved queue?)");
    return A3D_ERR_HIP;
  }
  if (host[0]) {
    set_error("a3d_sparse_quantize: a coordinate is NaN/inf or its voxel index is outside +-2^20");
    return A3D_ERR_COORD_RANGE;
  }
  *n_voxels = host[1];
  return A3D_OK;
}

}  // namespace a3d

using namespace a3d;

extern "C" size_t a3d_quantize_workspace_bytes(int64_t n_points) {
  if (n_points <= 0 || n_points > (int64_t)1 << 28) return 0;
  return carve_quant(nullptr, n_points).bytes;
}

extern "C" int a3d_sparse_quantize(const void* xyz_dev, int is_f64, int64_t n_points, double quantization_size,
                                   int32_t* coords_out_dev, int64_t* unique_map_dev, int64_t* inverse_map_dev,
                                   int64_t* n_voxels, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!xyz_dev || !coords_out_dev || !unique_map_dev || !inverse_map_dev || !n_voxels || n_points <= 0 ||
      n_points > (int64_t)1 << 28 || !(quantization_size > 0.0)) {
    set_error("a3d_sparse_quantize: bad arguments");
    return A3D_ERR_INVALID;
  }
  QuantWs w = carve_quant(workspace_dev, n_points);
  if (!workspace_dev || workspace_bytes < w.bytes || ((uintptr_t)workspace_dev & 255)) {
    set_error("a3d_sparse_quantize: workspace too small or misaligned (%zu < %zu)", workspace_bytes, w.bytes);
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (is_f64)
    return run_quantize<double>((const double*)xyz_dev, n_points, quantization_size, coords_out_dev, unique_map_dev,
                                inverse_map_dev, n_voxels, w, st);
  return run_quantize<float>((const float*)xyz_dev, n_points, quantization_size, coords_out_dev, unique_map_dev,
                             inverse_map_dev, n_voxels, w, st);
}
