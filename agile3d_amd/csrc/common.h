// Shared host/device helpers for libagile3d_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/agile3d_hip.h"

namespace a3d {

// ---- error plumbing (thread-local text, negative codes across the C ABI) -----------------
void set_error(const char* fmt, ...);
const char* get_error();

#define A3D_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      a3d::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return A3D_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

#define A3D_LAUNCH_CHECK() A3D_HIP_CHECK(hipGetLastError())

// ---- optional event timing ----------------------------------------------------------------
bool prof_enabled();
int prof_begin(hipStream_t st, int id, int bn = 0, int K = 0, int cin = 0, int cout = 0, int n_out = 0,
               int table = 0, int level = 0, int ksplit = 0);
void prof_end(hipStream_t st, int idx);
struct ProfScope {
  hipStream_t st;
  int idx;
  ProfScope(hipStream_t s, int id, int bn = 0, int K = 0, int cin = 0, int cout = 0, int n_out = 0, int table = 0,
            int level = 0, int ksplit = 0)
      : st(s), idx(prof_enabled() ? prof_begin(s, id, bn, K, cin, cout, n_out, table, level, ksplit) : -1) {}
  ~ProfScope() {
    if (idx >= 0) prof_end(st, idx);
  }
};

// Barrier over all workgroups of a launch whose workgroups are ALL RESIDENT (a few hundred small ones): a monotonic counter
// (`target` = workgroups x number of this barrier, counted from 1), release fence before the arrival and acquire after the wait
// (what other XCDs' workgroups wrote before is visible behind it).  The spin is bounded like every wait of this library: a
// give-up reports A3D_ERR_HIP through `fail` instead of hanging the device.
#ifdef __HIPCC__
__device__ inline void grid_barrier_counter(unsigned* bar, unsigned target, int* fail) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 26)) {
        if (fail) atomicMin(fail, A3D_ERR_HIP);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
#endif

// A kernel that spins at grid_barrier_counter needs EVERY workgroup of its grid resident at once: true when the current device,
// as this process sees it (a partitioned device exposes a fraction of the CUs), holds `grid` workgroups of `func` (radix.hip)
bool barrier_grid_fits(const void* func, int block_threads, size_t dyn_lds_bytes, int grid);

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

// ---- radix sort of (uint64 key, int32 value) pairs (radix.hip) ------------------------------
constexpr int kRadixMaxPasses = 8;
struct RadixPass {
  int shift, bits;          // digit = (key >> shift) & ((1 << bits) - 1), bits <= 8 ...
  int shift2 = 0, bits2 = 0;   // ... | ((key >> shift2) & ((1 << bits2) - 1)) << bits: a second, more significant field
                               // (bits + bits2 <= 8): two narrow fields of the key sorted in ONE pass
};
// digits covering key bits [bit_begin, bit_end), least significant first; returns their number (<= kRadixMaxPasses)
static inline int radix_passes(int bit_begin, int bit_end, RadixPass* out, int n_before = 0) {
  int np = n_before;
  for (int b = bit_begin; b < bit_end && np < kRadixMaxPasses; b += 8) {
    out[np].shift = b;
    out[np].bits = bit_end - b < 8 ? bit_end - b : 8;
    out[np].shift2 = out[np].bits2 = 0;
    ++np;
  }
  return np;
}
// Raise a kernel's dynamic-LDS limit.  A failure (e.g. a static __shared__ word sneaking into a kernel that asks for all
// 160 KB) must not stay behind as HIP's sticky last error, where the NEXT launch check would blame an innocent kernel: it
// is cleared and reported here; the launch that needs the LDS then fails under its own name.
#define A3D_ALLOW_LDS(BYTES, ...)                                                                                        \
  do {                                                                                                                   \
    if (hipFuncSetAttribute((const void*)(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (BYTES)) != hipSuccess) { \
      (void)hipGetLastError();                                                                                           \
      fprintf(stderr, "agile3d_hip: cannot raise the dynamic LDS limit of %s to %d bytes\n", #__VA_ARGS__, (int)(BYTES)); \
    }                                                                                                                    \
  } while (0)

size_t radix_sort_temp_bytes(int n_max);
// stable, ascending in the listed digits; digits that are constant over the input are skipped on the device;
// keys_in / vals_in are only read, temp must be 256-byte aligned
// one_launch: a scene-sized input (<= 128 k keys) may run as ONE launch with grid barriers inside.  Its workgroups spin at
// those barriers, so the caller must make sure that no second barrier kernel of this process can be in flight next to it
// (a3d_scene_create: only its first phase, which ends in a host synchronisation, and only one call at a time)
int radix_sort_pairs(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const int* vals_in,
                     int* vals_out, int n, const RadixPass* passes, int npass, hipStream_t st, int* fail_dev = nullptr,
                     bool one_launch = false);

// a_incl = inclusive prefix sums of a, b_excl = exclusive prefix sums of b (radix.hip; in place allowed)
size_t scan2_temp_bytes(int64_t n);
int scan2_incl_excl(void* temp, size_t temp_bytes, const int* a, const int* b, int64_t n, int* a_incl, int* b_excl,
                    hipStream_t st);

// BatchNorm(train) of a conv output whose block partials the conv's epilogue wrote (bnorm.hip; called from spconv.hip)
int bn_finish_from_partials(const float* partial, int nblocks, int rows_per_block, const float* x, int ldx, int64_t n, int C,
                            const float* gamma, const float* beta, float eps, const float* res, int ldr, int relu, float* y,
                            int ldy, int y_zero_row, float* save_mean, float* save_rstd, float* running_mean,
                            float* running_var, float momentum, hipStream_t st);

int bn_sums_from_partials(const float* partial, int nblocks, int C, double* sums, hipStream_t st);

// ---- geometry constants -------------------------------------------------------------------
constexpr int kTileRows = 128;   // output rows per conv workgroup
constexpr int kGroupRows = 16;   // MFMA row granularity (v_mfma_f32_16x16x4_f32)
constexpr int kSuperTile = 1 << 18;  // rows re-ordered by neighbour pattern inside one super tile (= whole level up to 262 k rows)
constexpr int kCoordOff = 1 << 17;
constexpr uint64_t kEmptyKey = ~0ull;

// ---- Morton keys / voxel hash (shared by scene.hip and spconv.hip)
__host__ __device__ inline uint64_t spread3(uint64_t x) {
  x &= 0x1fffffULL;
  x = (x | x << 32) & 0x1f00000000ffffULL;
  x = (x | x << 16) & 0x1f0000ff0000ffULL;
  x = (x | x << 8) & 0x100f00f00f00f00fULL;
  x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
  x = (x | x << 2) & 0x1249249249249249ULL;
  return x;
}
__host__ __device__ inline uint32_t compact3(uint64_t x) {
  x &= 0x1249249249249249ULL;
  x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ULL;
  x = (x ^ (x >> 4)) & 0x100f00f00f00f00fULL;
  x = (x ^ (x >> 8)) & 0x1f0000ff0000ffULL;
  x = (x ^ (x >> 16)) & 0x1f00000000ffffULL;
  x = (x ^ (x >> 32)) & 0x1fffffULL;
  return (uint32_t)x;
}
// key of voxel (b, X, Y, Z) given in units of level L.  key_{L+1}(parent) == key_L(child) >> 3.
__host__ __device__ inline uint64_t make_key(int b, int X, int Y, int Z, int L) {
  const int off = kCoordOff >> L;
  return spread3((uint64_t)(X + off)) | (spread3((uint64_t)(Y + off)) << 1) |
         (spread3((uint64_t)(Z + off)) << 2) | ((uint64_t)b << (54 - 3 * L));
}
__host__ __device__ inline void decode_key(uint64_t key, int L, int& b, int& X, int& Y, int& Z) {
  const int bits = 54 - 3 * L;
  const int off = kCoordOff >> L;
  b = (int)(key >> bits);
  const uint64_t low = key & ((1ULL << bits) - 1);
  X = (int)compact3(low) - off;
  Y = (int)compact3(low >> 1) - off;
  Z = (int)compact3(low >> 2) - off;
}
__device__ inline uint32_t hash64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return (uint32_t)k;
}
__device__ inline int hash_lookup(const uint64_t* __restrict__ hk, const int* __restrict__ hv,
                                  uint32_t hmask, uint64_t key) {
  uint32_t h = hash64(key) & hmask;
  for (uint32_t probe = 0; probe <= hmask; ++probe) {
    const uint64_t k = hk[h];
    if (k == key) return hv[h];
    if (k == kEmptyKey) return -1;
    h = (h + 1) & hmask;
  }
  return -1;
}


// ---- the scene ----------------------------------------------------------------------------
struct Level {
  int n = 0, npad = 0;
  uint64_t* keys = nullptr;     // [n]   Morton keys, Morton order
  int* perm = nullptr;          // [n]   Morton row -> internal row
  int* inv = nullptr;           // [n]   internal row -> Morton row
  int* parentM = nullptr;       // [n]   Morton row -> Morton row at level+1 (levels 0..3)
  int32_t* xyzb = nullptr;      // [n][4] internal order
  uint64_t* hkeys = nullptr;    // hash table (key -> internal row)
  int* hvals = nullptr;
  uint32_t hmask = 0;
  int* nbr27 = nullptr;         // [27][npad]
  uint32_t* gmask27 = nullptr;  // [npad/16]
  // exclusive prefix sums over 64-row tiles of the number of offsets a tile has (popcount of the OR of its four group
  // masks): the conv kernel cuts the (tile, offset) work of a layer into equal shares with them (spconv.hip)
  int* pre27 = nullptr;         // [npad/64 + 1]
  int* pre_down = nullptr;      // [npad(level+1)/64 + 1]   (levels 0..3)
  int* pre_up = nullptr;        // [npad/64 + 1]            (levels 0..3)
  int* child8 = nullptr;        // [8][npad(level+1)]     (levels 0..3)
  uint32_t* gmask_down = nullptr;
  int* up8 = nullptr;           // [8][npad]              (levels 0..3)
  uint32_t* gmask_up = nullptr;
  int* up_rows = nullptr;       // [npad]
  // level 0 only, when the bounding box of the batch is small enough (<= kGridCellsPerVoxel cells per voxel):
  // dense voxel -> row grid, x fastest, kGridPad empty cells on every side so that a 5^3 neighbourhood of an existing
  // voxel never leaves it (no bounds checks).  nullptr -> the hash table is used instead.
  int* grid = nullptr;
  int gorg[3] = {0, 0, 0};      // coordinate of cell (0, 0, 0)
  int gdim[3] = {0, 0, 0};
};
constexpr int kGridPad = 2;
constexpr int64_t kGridCellsPerVoxel = 64;
__device__ __forceinline__ size_t grid_cell(const Level& lv, int b, int x, int y, int z) {
  return (((size_t)b * lv.gdim[2] + (z - lv.gorg[2])) * lv.gdim[1] + (y - lv.gorg[1])) * lv.gdim[0] + (x - lv.gorg[0]);
}

}  // namespace a3d

namespace a3d {
constexpr int kBBox = 8 + 1024;        // kBBoxSlots x {min x,y,z, max x,y,z, -, -} of the level-0 coordinates (slot = block % slots:
constexpr int kBBoxSlots = 64;         // thousands of atomics on six addresses would serialise; the host folds the slots)
constexpr int kSizesBar = kBBox + 8 * kBBoxSlots;   // one word behind the bounding-box slots: the grid-barrier counter of k_heads_all (uploaded as 0)
constexpr int kSizesInts = kSizesBar + 8;  // device-side size/error/batch-start/bounding-box block read back by a3d_scene_create
}

struct a3d_scene {
  int64_t n0 = 0;
  int n_batch = 0;
  int batch_start[1024];        // first row of every batch sample (rows of a sample are contiguous)
  a3d::Level lv[A3D_NUM_LEVELS];
  int* orig_row = nullptr;      // [n0] internal level-0 row -> caller row
  void* workspace = nullptr;
  size_t workspace_bytes = 0;
  // weight-gradient work lists (a3d_scene_build_wgrad_lists, csrc/wgrad.hip): per kernel-map kind (0 = 3^3, 1 = stride-2
  // down, 2 = transposed up) and level owning the map's group masks, wg_list[k * wg_stride + i] = the i-th 16-position
  // group that has offset k (ascending), wg_count[k] = how many there are (host copy)
  int* wg_list[3][A3D_NUM_LEVELS] = {};
  int wg_stride[3][A3D_NUM_LEVELS] = {};
  int wg_count[3][A3D_NUM_LEVELS][27] = {};
  bool wg_ready = false;
  // the counts come back asynchronously: a3d_scene_build_wgrad_lists leaves a pending read-back (pinned buffer + event from a
  // process-wide pool) that the first a3d_conv_wgrad on the scene waits for (wgrad.hip: wgrad_lists_finish)
  void* wg_pending = nullptr;
  uint64_t serial = 0;          // unique per scene of the process (a3d_scene_create): key of the launch-plan cache in wgrad.hip
};

namespace a3d {
void wgrad_scene_release(a3d_scene* s);   // wgrad.hip: gives a pending read-back of the work lists' counts back to its pool
}
