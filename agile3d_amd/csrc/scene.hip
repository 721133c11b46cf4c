// Coordinate manager for the sparse-voxel backbone, built on the GPU.
//
// Replaces what MinkowskiEngine's coordinate manager does implicitly behind
// ME.SparseTensor(...) (reference engine.py:47-51) and every ME layer: voxel hash,
// stride-2 coordinate sets (res16unet.py:229,234,239,245), kernel maps for the 3^3
// (resnet_block.py:24-43) and 2^3-stride-2 kernels (and their transposes).
//
// MI355X-first design (DESIGN.md "scene"):
//   * voxels are sorted by a 64-bit Morton key (batch | z y x interleaved); siblings of a
//     coarse voxel are then contiguous, so each coarser level is a run-length compaction of
//     the finer one (one scan per level, no hash insert races, deterministic row ids);
//   * the rows of every level are then re-ordered by their 27-bit neighbour-presence pattern (one stable
//     radix sort over all levels, ties keep Morton order), so that a 16-row MFMA group shares one pattern and
//     the implicit-GEMM kernel can skip absent kernel offsets per group (27 -> 11.7 offsets per group);
//   * kernel maps are output-major neighbour tables int32[K][npad] (k-major: the 128 rows of a
//     workgroup tile are contiguous for every offset) with "missing" = the all-zero row n.
#include "common.h"
#include <atomic>
#include <algorithm>
#include <limits.h>
#include <stdarg.h>
#include <stdlib.h>

namespace a3d {

static thread_local char g_err[512] = "";
static std::atomic<int> g_phase1_calls{0};   // a3d_scene_create calls between their first launch and their host synchronisation
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

// ------------------------------------------------------------------------------ profiler
}  // namespace a3d
#include <vector>
namespace a3d {
struct ProfRec {
  a3d_prof_entry e;
  hipEvent_t e0, e1;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;
static hipEvent_t get_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
bool prof_enabled() { return g_prof_on; }
int prof_begin(hipStream_t st, int id, int bn, int K, int cin, int cout, int n_out, int table, int level, int ksplit) {
  ProfRec r;
  r.e = a3d_prof_entry{id, bn, K, cin, cout, n_out, table, level, ksplit, 0.f};
  r.e0 = get_event();
  r.e1 = get_event();
  (void)hipEventRecord(r.e0, st);
  g_prof.push_back(r);
  return (int)g_prof.size() - 1;
}
void prof_end(hipStream_t st, int idx) { (void)hipEventRecord(g_prof[idx].e1, st); }

// ------------------------------------------------------------------------------ block scan
__device__ inline int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
// inclusive scan over a 1024-thread block; lds must hold 17 ints; returns scan, *total = block sum
__device__ inline int block_incl_scan(int v, int* lds, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int s = wave_incl_scan(v);
  if (lane == 63) lds[w] = s;
  __syncthreads();
  if (w == 0) {
    const int x = lane < nw ? lds[lane] : 0;
    const int xs = wave_incl_scan(x);
    if (lane < nw) lds[lane] = xs - x;
    if (lane == nw - 1) lds[16] = xs;
  }
  __syncthreads();
  const int r = s + lds[w];
  *total = lds[16];
  __syncthreads();
  return r;
}

// ------------------------------------------------------------------------------ kernels
// sizes_dev: [0..4] level sizes, [5] error code, [6] number of batch segments, [7] largest batch index,
// [8 + b] first row of batch sample b (-1 = absent)
__global__ void k_make_keys(const int32_t* __restrict__ coords4, int n, uint64_t* keys, int* vals,
                            int* sizes_dev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
  if (i < n) {
    const int b = coords4[4 * i + 0], x = coords4[4 * i + 1], y = coords4[4 * i + 2], z = coords4[4 * i + 3];
    const int lim = kCoordOff;
    vals[i] = i;
    if (b < 0 || b > 1022 || x < -lim || x >= lim || y < -lim || y >= lim || z < -lim || z >= lim) {
      atomicMin(&sizes_dev[5], A3D_ERR_COORD_RANGE);
      keys[i] = 0;
    } else {
      keys[i] = make_key(b, x, y, z, 0);
      lo[0] = hi[0] = x, lo[1] = hi[1] = y, lo[2] = hi[2] = z;
      // batch samples must be contiguous row ranges (ME.utils.batched_coordinates): record where each starts
      const int prev = i > 0 ? coords4[4 * (i - 1)] : -1;
      if (b != prev) {
        atomicAdd(&sizes_dev[6], 1);
        atomicMax(&sizes_dev[7], b);
        if (atomicCAS(&sizes_dev[8 + b], -1, i) != -1) atomicMin(&sizes_dev[5], A3D_ERR_INVALID);
      }
    }
  }
  // bounding box of the batch (decides whether level 0 gets a dense voxel grid): wave shuffle -> LDS -> one atomic
  // pair per workgroup and axis, spread over kBBoxSlots slots
  __shared__ int blo[3], bhi[3];
  if (threadIdx.x < 3) blo[threadIdx.x] = INT_MAX, bhi[threadIdx.x] = INT_MIN;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      lo[a] = min(lo[a], __shfl_xor(lo[a], d));
      hi[a] = max(hi[a], __shfl_xor(hi[a], d));
    }
    if ((threadIdx.x & 63) == 0 && lo[a] <= hi[a]) {
      atomicMin(&blo[a], lo[a]);
      atomicMax(&bhi[a], hi[a]);
    }
  }
  __syncthreads();
  if (threadIdx.x < 3 && blo[threadIdx.x] <= bhi[threadIdx.x]) {
    int* slot = sizes_dev + kBBox + 8 * (blockIdx.x % kBBoxSlots);
    atomicMin(&slot[threadIdx.x], blo[threadIdx.x]);
    atomicMax(&slot[3 + threadIdx.x], bhi[threadIdx.x]);
  }
}

// ---- coarser levels: key_{L}(voxel) = key_0 >> 3L, so a row of level L is a run of equal (key_0 >> 3L) in the
// sorted level-0 keys and ALL four coarser levels come out of one pass over them (3 launches instead of 12):
// head_L[i] = first element of its level-L run; rank_L(i) = #heads_L in [0, i] - 1 = the level-L row of voxel i.
struct CoarseOut {
  uint64_t* keys[A3D_NUM_LEVELS - 1];   // keys[L-1] = level-L keys
  int* parentM[A3D_NUM_LEVELS - 1];     // parentM[L] : level-L row -> level-(L+1) row
  int* firstM[A3D_NUM_LEVELS - 1];      // firstM[L] : level-(L+1) row -> its first level-L row (siblings are consecutive)
};
// returns true when voxel i repeats its predecessor's key (a duplicate coordinate)
__device__ __forceinline__ bool head_flags(const uint64_t* __restrict__ keys, int n, int i, int (&f)[A3D_NUM_LEVELS - 1]) {
#pragma unroll
  for (int L = 1; L < A3D_NUM_LEVELS; ++L) f[L - 1] = 0;
  if (i < n) {
    const uint64_t k = keys[i], kp = i > 0 ? keys[i - 1] : 0;
#pragma unroll
    for (int L = 1; L < A3D_NUM_LEVELS; ++L) f[L - 1] = (i == 0) || ((k >> (3 * L)) != (kp >> (3 * L)));
    return i > 0 && k == kp;
  }
  return false;
}
// pass 1: heads per block of 1024 voxels, blocksums[L-1][block]; also the duplicate-coordinate check (equal neighbours
// in the sorted keys)
__global__ void __launch_bounds__(1024) k_heads_count(const uint64_t* __restrict__ keys, int n, int nb, int* blocksums,
                                                      int* sizes_dev) {
  __shared__ int lds[17];
  int f[A3D_NUM_LEVELS - 1];
  if (head_flags(keys, n, blockIdx.x * 1024 + threadIdx.x, f)) atomicMin(&sizes_dev[5], A3D_ERR_DUPLICATE);
#pragma unroll
  for (int L = 0; L < A3D_NUM_LEVELS - 1; ++L) {
    int total;
    block_incl_scan(f[L], lds, &total);
    if (threadIdx.x == 0) blocksums[L * nb + blockIdx.x] = total;
  }
}
// pass 2: exclusive scan of the block sums of every level (one workgroup per level); level sizes -> sizes_dev[1..4]
__global__ void __launch_bounds__(1024) k_heads_scan(int* blocksums, int nb, int* sizes_dev) {
  __shared__ int lds[17];
  int* bs = blocksums + blockIdx.x * nb;
  int running = 0;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? bs[i] : 0;
    int total;
    const int s = block_incl_scan(v, lds, &total);
    if (i < nb) bs[i] = running + s - v;
    running += total;
  }
  if (threadIdx.x == 0) sizes_dev[1 + blockIdx.x] = running;
}
// pass 3: keys of the coarser levels and the row -> parent-row maps
__global__ void __launch_bounds__(1024) k_heads_write(const uint64_t* __restrict__ keys, int n, int nb,
                                                      const int* __restrict__ blockoffs, CoarseOut out) {
  __shared__ int lds[17];
  const int i = blockIdx.x * 1024 + threadIdx.x;
  int f[A3D_NUM_LEVELS - 1], rank[A3D_NUM_LEVELS];
  head_flags(keys, n, i, f);
  rank[0] = i;
#pragma unroll
  for (int L = 0; L < A3D_NUM_LEVELS - 1; ++L) {
    int total;
    const int s = block_incl_scan(f[L], lds, &total);
    rank[L + 1] = blockoffs[L * nb + blockIdx.x] + s - 1;
  }
  if (i >= n) return;
  const uint64_t k = keys[i];
#pragma unroll
  for (int L = 1; L < A3D_NUM_LEVELS; ++L) {
    if (f[L - 1]) {
      out.keys[L - 1][rank[L]] = k >> (3 * L);
      out.firstM[L - 1][rank[L]] = rank[L - 1];   // a voxel that heads a level-L run heads its level-(L-1) run too
    }
    // voxel i is a row of level L-1 iff it heads its level-(L-1) run (every voxel is a row of level 0)
    if (L == 1 || f[L - 2]) out.parentM[L - 1][rank[L - 1]] = rank[L];
  }
}

// the three passes above in ONE launch for a scene-sized input (every workgroup resident: nb <= kHeadsOneLaunch): counts ->
// grid barrier -> every workgroup adds up the block sums in front of it (and the last one writes the level sizes) -> keys and
// parent maps.  The flags and in-block scans of the first pass stay in registers.
constexpr int kHeadsOneLaunch = 128;
__global__ void __launch_bounds__(1024) k_heads_all(const uint64_t* __restrict__ keys, int n, int nb, int* blocksums,
                                                    int* sizes_dev, CoarseOut out) {
  __shared__ int lds[17];
  __shared__ int offs[A3D_NUM_LEVELS - 1];
  const int i = blockIdx.x * 1024 + threadIdx.x;
  int f[A3D_NUM_LEVELS - 1], sc[A3D_NUM_LEVELS - 1], tot[A3D_NUM_LEVELS - 1];
  if (head_flags(keys, n, i, f)) atomicMin(&sizes_dev[5], A3D_ERR_DUPLICATE);
#pragma unroll
  for (int L = 0; L < A3D_NUM_LEVELS - 1; ++L) {
    sc[L] = block_incl_scan(f[L], lds, &tot[L]);
    if (threadIdx.x == 0) __hip_atomic_store(blocksums + L * nb + blockIdx.x, tot[L], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  grid_barrier_counter((unsigned*)(sizes_dev + kSizesBar), gridDim.x, sizes_dev + 5);
#pragma unroll
  for (int L = 0; L < A3D_NUM_LEVELS - 1; ++L) {   // nb <= 1024: one value per thread
    const int v = (int)threadIdx.x < (int)blockIdx.x
                      ? __hip_atomic_load(blocksums + L * nb + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    int before;
    block_incl_scan(v, lds, &before);
    if (threadIdx.x == 0) {
      offs[L] = before;
      if (blockIdx.x == gridDim.x - 1) sizes_dev[1 + L] = before + tot[L];
    }
  }
  __syncthreads();
  if (i >= n) return;
  int rank[A3D_NUM_LEVELS];
  rank[0] = i;
#pragma unroll
  for (int L = 0; L < A3D_NUM_LEVELS - 1; ++L) rank[L + 1] = offs[L] + sc[L] - 1;
  const uint64_t k = keys[i];
#pragma unroll
  for (int L = 1; L < A3D_NUM_LEVELS; ++L) {
    if (f[L - 1]) {
      out.keys[L - 1][rank[L]] = k >> (3 * L);
      out.firstM[L - 1][rank[L]] = rank[L - 1];
    }
    if (L == 1 || f[L - 2]) out.parentM[L - 1][rank[L - 1]] = rank[L];
  }
}

// ---- per-level table kernels, batched over levels -------------------------------------------------------
// The phase-2 kernels are tiny (a level has 140 .. 80 k rows) and launch-bound, so each of them is launched ONCE for
// all levels: the block index selects the level (S.blk = block prefix per level), the rest is the per-level kernel.
struct LevelSet {
  Level lv[A3D_NUM_LEVELS];
  int* nbrM[A3D_NUM_LEVELS];   // [27][npad] neighbour rows in Morton order
  const int* firstM[A3D_NUM_LEVELS - 1];   // level-(L+1) Morton row -> first level-L Morton row
  int off[A3D_NUM_LEVELS + 1]; // start of every level in the concatenated row list of the sorts
  int blk[A3D_NUM_LEVELS + 1]; // first block of every level in THIS launch
  int nlev;                    // levels taking part in this launch
  int st_shift, level_shift;
  uint64_t* cat_keys;
  int *cat_vals, *cat_sorted;
  const int* vals_sorted0;     // Morton row of level 0 -> caller row (values of the first sort)
  int* orig_row;               // internal row of level 0 -> caller row
};
__device__ __forceinline__ int ls_level(const LevelSet& S, int& local_block) {
  int L = 0;
  while (L + 1 < S.nlev && (int)blockIdx.x >= S.blk[L + 1]) ++L;
  local_block = blockIdx.x - S.blk[L];
  return L;
}

// hash table of a level (the keys start out as kEmptyKey: the host clears the whole region with one memset)
__global__ void k_hash_insert(const LevelSet S) {
  int lb;
  const Level& lv = S.lv[ls_level(S, lb)];
  const int i = lb * blockDim.x + threadIdx.x;
  if (i >= lv.n) return;
  const uint64_t key = lv.keys[i];
  if (lv.grid) {   // level 0 with a dense grid (cleared to -1 by the host): no hash table
    int b, X, Y, Z;
    decode_key(key, 0, b, X, Y, Z);
    lv.grid[grid_cell(lv, b, X, Y, Z)] = i;
    return;
  }
  uint32_t h = hash64(key) & lv.hmask;
  for (uint32_t probe = 0; probe <= lv.hmask; ++probe) {
    const unsigned long long prev =
        atomicCAS((unsigned long long*)&lv.hkeys[h], (unsigned long long)kEmptyKey, (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) {
      lv.hvals[h] = i;
      return;
    }
    h = (h + 1) & lv.hmask;
  }
}

// 3^3 neighbours in Morton row ids: nbrM[k][npad] (-1 = missing), AND the sort key of the row re-ordering (round 5: one
// thread per ROW with its 27 lookups in flight together -- the key is decoded once instead of 27 times, the three x-adjacent
// cells of a (y, z) pair are one 12-byte run of the level-0 grid, Morton-adjacent threads share those runs, and the presence
// mask falls out of the lookups: no second kernel re-reading the 27 tables.  One thread per (row, offset) read 3.2 GB for
// the 16-scene batch, a 64-byte sector per lookup).
// Rows are re-ordered inside super tiles of 2^st_shift consecutive rows by a small key (the 27-bit neighbour
// presence mask / the child slot): ALL levels go through one stable device-wide radix sort of
//   key = level << (27 + super-tile bits) | super tile << 27 | small key,   value = position in the concatenated row list.
__global__ void k_nbr_morton(const LevelSet S) {
  int lb;
  const int L = ls_level(S, lb);
  const Level& lv = S.lv[L];
  // sort key = the 27 presence bits, the 12 edge offsets most significant, then the 8 corners, the 6 faces, the centre:
  // rows are grouped 16 at a time, a group multiplies for every offset ANY of its rows has, and the offsets a sort
  // does not reach (the low bits of the key) end up in nearly every group's union -- so the high bits should be the ones
  // present in about half the rows (edges: 35-60 %), not the ones nearly every row has (faces: 60-85 %, centre: all).
  // Offsets issued per row on the bench scenes, plain k order -> this order: L1 12.28 -> 12.16, L2 15.83 -> 15.36,
  // L3 19.05 -> 18.49 (real neighbours per row: 11.65 / 13.61 / 14.94).
  constexpr int kKeyBit[27] = {14, 26, 13, 25, 6, 24, 12, 23, 11, 22, 5, 21, 4, 0, 3, 20, 2, 19, 10, 18, 9, 17, 1, 16, 8, 15, 7};
  if (!lv.grid) {
    // hash levels (the coarse levels; level 0 of far-apart inputs): a hash probe is a chain of dependent loads, so the 27
    // lookups of a row go to 27 LANES (half a wave per row, the launch gives these levels 32 threads per row) and the
    // presence mask is a ballot
    const int k = threadIdx.x & 31;
    const int i = lb * (blockDim.x >> 5) + (threadIdx.x >> 5);
    int r = -1;
    if (i < lv.n && k < 27) {
      int b, X, Y, Z;
      decode_key(lv.keys[i], L, b, X, Y, Z);
      const int lim = kCoordOff >> L;
      const int x = X + k % 3 - 1, y = Y + (k / 3) % 3 - 1, z = Z + k / 9 - 1;
      if (k == 13) r = i;
      else if (x >= -lim && x < lim && y >= -lim && y < lim && z >= -lim && z < lim)
        r = hash_lookup(lv.hkeys, lv.hvals, lv.hmask, make_key(b, x, y, z, L));
      S.nbrM[L][(size_t)k * lv.npad + i] = r;
    }
    const unsigned long long bal = __ballot(r >= 0);
    if (k == 0 && i < lv.n) {
      const uint32_t have = (uint32_t)(bal >> (threadIdx.x & 32)) & 0x7ffffffu;
      uint32_t m = 0;
#pragma unroll
      for (int kk = 0; kk < 27; ++kk) m |= ((have >> kk) & 1u) << kKeyBit[kk];
      S.cat_keys[S.off[L] + i] = ((uint64_t)L << S.level_shift) | ((uint64_t)(i >> S.st_shift) << 27) | m;
      S.cat_vals[S.off[L] + i] = S.off[L] + i;
    }
    return;
  }
  const int i = lb * blockDim.x + threadIdx.x;
  if (i >= lv.n) return;
  int b, X, Y, Z;
  decode_key(lv.keys[i], L, b, X, Y, Z);
  int r[27];
  const ptrdiff_t c0 = (ptrdiff_t)grid_cell(lv, b, X, Y, Z);
#pragma unroll
  for (int k = 0; k < 27; ++k)     // x fastest; the grid's empty border covers the 3^3 neighbourhood
    r[k] = lv.grid[c0 + ((ptrdiff_t)(k / 9 - 1) * lv.gdim[1] + ((k / 3) % 3 - 1)) * lv.gdim[0] + (k % 3 - 1)];
  r[13] = i;
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    S.nbrM[L][(size_t)k * lv.npad + i] = r[k];
    m |= (r[k] >= 0 ? 1u : 0u) << kKeyBit[k];
  }
  S.cat_keys[S.off[L] + i] = ((uint64_t)L << S.level_shift) | ((uint64_t)(i >> S.st_shift) << 27) | m;
  S.cat_vals[S.off[L] + i] = S.off[L] + i;
}
// sorted segment of a level -> perm (old row -> new row) and inv (new row -> old row); `up` selects the second
// sort (virtual rows of the transposed convs: only the new -> old map, up_rows, is kept)
__global__ void k_perm_from_sorted(const LevelSet S, int up) {
  int lb;
  const int L = ls_level(S, lb);
  const Level& lv = S.lv[L];
  const int p = lb * blockDim.x + threadIdx.x;
  if (p >= lv.n) return;
  const int src = S.cat_sorted[S.off[L] + p] - S.off[L];
  if (up) {
    lv.up_rows[p] = src;
  } else {
    lv.perm[src] = p;
    lv.inv[p] = src;
    if (L == 0) S.orig_row[p] = S.vals_sorted0[src];
  }
}

// nbr27[k][f] in internal row ids (missing / padding -> n) + per-16-row-group presence masks: one thread per ROW (its old
// position is read once, the 27 table entries and their new row ids are independent gathers in flight together), the group
// mask is the OR over the group's 16 lanes -- no atomics, every word written exactly once
__global__ void k_remap_nbr(const LevelSet S) {
  int lb;
  const int L = ls_level(S, lb);
  const Level& lv = S.lv[L];
  const int f = lb * blockDim.x + threadIdx.x;
  if (f >= lv.npad) return;            // npad is a multiple of 128: whole waves leave together
  int v[27];
  uint32_t word = 0;
  if (f < lv.n) {
    const int mrow = lv.inv[f];
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = S.nbrM[L][(size_t)k * lv.npad + mrow];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const bool present = v[k] >= 0;
      v[k] = present ? lv.perm[v[k]] : lv.n;
      word |= (present ? 1u : 0u) << k;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = lv.n;
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) lv.nbr27[(size_t)k * lv.npad + f] = v[k];
  word |= __shfl_xor((int)word, 1, 16);
  word |= __shfl_xor((int)word, 2, 16);
  word |= __shfl_xor((int)word, 4, 16);
  word |= __shfl_xor((int)word, 8, 16);
  if ((threadIdx.x & 15) == 0) lv.gmask27[f >> 4] = word;
}

// pre[t] = number of (tile, offset) pairs of the tiles before tile t, for the three mask tables of a level
// (blockIdx.y: 0 = 3^3, 1 = stride-2 down, 2 = transposed up) over 64-row tiles; one
// workgroup per (level, table), serial over 1024-tile chunks (a 1 M-voxel level has 16 k tiles)
__global__ void __launch_bounds__(1024) k_tile_prefix(const LevelSet S) {
  __shared__ int lds[17];
  __shared__ int carry;
  const Level& lv = S.lv[blockIdx.x];
  const int gpt = 4;   // 16-row groups per 64-row tile
  const uint32_t* gmask;
  int* pre;
  int npad;
  if (blockIdx.y == 0) {
    gmask = lv.gmask27, pre = lv.pre27, npad = lv.npad;
  } else {
    if ((int)blockIdx.x >= A3D_NUM_LEVELS - 1) return;
    if (blockIdx.y == 1) gmask = lv.gmask_down, pre = lv.pre_down, npad = S.lv[blockIdx.x + 1].npad;
    else gmask = lv.gmask_up, pre = lv.pre_up, npad = lv.npad;
  }
  const int ntile = npad / (16 * gpt);
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int t0 = 0; t0 < ntile; t0 += 1024) {
    const int t = t0 + threadIdx.x;
    int c = 0;
    if (t < ntile) {
      uint32_t un = 0;
      for (int q = 0; q < gpt; ++q) un |= gmask[gpt * t + q];
      c = __popc(un);
    }
    int total;
    const int incl = block_incl_scan(c, lds, &total);
    const int base = carry;
    if (t < ntile) pre[t] = base + incl - c;
    __syncthreads();
    if (threadIdx.x == 0) carry = base + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) pre[ntile] = carry;
}

// coordinates in the new row order + hash values -> new rows (one launch over max(n, capacity) per level)
__global__ void k_xyzb_hashfix(const LevelSet S) {
  int lb;
  const int L = ls_level(S, lb);
  const Level& lv = S.lv[L];
  const uint32_t f = lb * blockDim.x + threadIdx.x;
  if ((int)f < lv.n) {
    int b, X, Y, Z;
    decode_key(lv.keys[lv.inv[f]], L, b, X, Y, Z);
    lv.xyzb[4 * f + 0] = X;
    lv.xyzb[4 * f + 1] = Y;
    lv.xyzb[4 * f + 2] = Z;
    lv.xyzb[4 * f + 3] = b;
    // (the level-0 grid keeps the MORTON rows k_level0_grid wrote: its users -- the stem and its weight gradient -- walk the
    // voxels in Morton order, where the cells and feature rows of neighbouring threads share cache lines, and map to
    // internal rows through perm where they need them)
    if (L < A3D_NUM_LEVELS - 1) {   // sort key of the second sort: the child slot of internal row f inside its super tile
      S.cat_keys[S.off[L] + f] = ((uint64_t)L << S.level_shift) | ((uint64_t)(f >> S.st_shift) << 27) | (lv.keys[lv.inv[f]] & 7);
      S.cat_vals[S.off[L] + f] = S.off[L] + f;
    }
  }
  if (!lv.grid && f <= lv.hmask && lv.hkeys[f] != kEmptyKey) lv.hvals[f] = lv.perm[lv.hvals[f]];
}

// child8[slot][coarse row] = fine row (fine level L, coarse L+1; missing = n_fine) and gmask_down[g] = child slots present
// in the 16 coarse rows of group g.  One thread per coarse row: its children are consecutive Morton rows of the fine
// level, in slot order (firstM from the level compaction) -- every entry is written exactly once, nothing is cleared first,
// and a group is 16 consecutive lanes (ballots, no atomics).
__global__ void k_child8(const LevelSet S) {
  int lb;
  const int L = ls_level(S, lb);
  const Level& f = S.lv[L];
  const Level& c = S.lv[L + 1];
  const int pf = lb * blockDim.x + threadIdx.x;
  if (pf >= c.npad) return;   // npad is a multiple of 128: whole waves leave together
  int m = 0, m1 = 0;
  if (pf < c.n) {
    const int mc = c.inv[pf];
    m = S.firstM[L][mc];
    m1 = mc + 1 < c.n ? S.firstM[L][mc + 1] : f.n;
  }
  const int lane = threadIdx.x & 63;
  uint32_t word = 0;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    int v = f.n;
    if (m < m1 && (int)(f.keys[m] & 7) == s) v = f.perm[m++];
    f.child8[(size_t)s * c.npad + pf] = v;
    const unsigned long long bal = __ballot(v != f.n);
    if ((bal >> (lane & 48)) & 0xffffULL) word |= 1u << s;
  }
  if ((lane & 15) == 0) f.gmask_down[pf >> 4] = word;
}

// transposed-conv tables over virtual rows v (fine rows sorted by child slot inside super tiles)
__global__ void k_up(const LevelSet S) {
  int lb;
  const int L = ls_level(S, lb);
  const Level& f = S.lv[L];
  const Level& c = S.lv[L + 1];
  const int v = lb * blockDim.x + threadIdx.x;
  if (v >= f.npad) return;
  int slot = -1, pf = c.n;
  if (v < f.n) {
    const int m = f.inv[f.up_rows[v]];
    slot = (int)(f.keys[m] & 7);
    pf = c.perm[f.parentM[m]];
  }
  {   // slots present in each 16-row group (a group is 16 consecutive lanes: no atomics)
    const int lane = threadIdx.x & 63;
    uint32_t word = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const unsigned long long bal = __ballot(slot == s);
      if ((bal >> (lane & 48)) & 0xffffULL) word |= 1u << s;
    }
    if ((lane & 15) == 0) f.gmask_up[v >> 4] = word;
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) f.up8[(size_t)s * f.npad + v] = (s == slot) ? pf : c.n;
}

// ------------------------------------------------------------------------------ host side
static int super_tile_shift() {   // log2 of the super-tile size (kSuperTile rows: the whole level -- smaller windows measured
  static int sh = -1;             // slower in round 2: 552 / 537 / 517 scenes/s at 65 536 / 16 384 / 4 096 rows against 558)
  if (sh < 0) {
    sh = 0;
    while ((1 << sh) < kSuperTile) ++sh;
    if (sh > 18) sh = 18;
  }
  return sh;
}

struct Bump {
  char* base;
  size_t off = 0;
  explicit Bump(void* b) : base((char*)b) {}
  template <typename T>
  T* take(size_t count) {
    off = align256(off);
    T* p = base ? (T*)(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

static uint32_t hash_capacity(int n) {
  uint32_t c = 1024;
  while (c < 2u * (uint32_t)n) c <<= 1;
  return c;
}

// the three sorts of a scene build are radix.hip's (rocPRIM's radix_sort_pairs is a merge sort of ~18 dependent launches
// below 1 M items: 0.4 of the 0.74 ms of a 4-scene build, launch-bound at one scene)
static size_t sort_temp_bytes(int n) { return radix_sort_temp_bytes(n); }

// phase-1 arrays (sized by the n0 upper bound) and phase-2 tables (sized by the real level sizes)
struct Phase1 {
  uint64_t *keys_in, *keys[A3D_NUM_LEVELS];
  int *vals_in, *vals_sorted, *parentM[A3D_NUM_LEVELS - 1], *firstM[A3D_NUM_LEVELS - 1], *blocksums, *sizes_dev;
  void* sort_temp;
  size_t sort_temp_bytes;
};
static void carve_phase1(Bump& b, int n0, Phase1& p) {
  p.keys_in = b.take<uint64_t>(n0);
  p.vals_in = b.take<int>(n0);
  p.vals_sorted = b.take<int>(n0);
  for (int L = 0; L < A3D_NUM_LEVELS; ++L) p.keys[L] = b.take<uint64_t>(n0);
  for (int L = 0; L < A3D_NUM_LEVELS - 1; ++L) p.parentM[L] = b.take<int>(n0);
  for (int L = 0; L < A3D_NUM_LEVELS - 1; ++L) p.firstM[L] = b.take<int>(n0);
  p.blocksums = b.take<int>((size_t)(A3D_NUM_LEVELS - 1) * (n0 / 1024 + 2));
  p.sizes_dev = b.take<int>(kSizesInts);
  p.sort_temp_bytes = sort_temp_bytes(A3D_NUM_LEVELS * n0 + 1024);   // also used for the concatenated per-level row sorts (sum of n_L <= 5 n0)
  p.sort_temp = b.take<char>(p.sort_temp_bytes);
}
struct Phase2Tmp {
  int* nbrM[A3D_NUM_LEVELS];     // [27][npad] neighbour rows in Morton order, kept until the rows are re-ordered
  uint64_t *cat_keys, *cat_keys_sorted;
  int *cat_vals, *cat_vals_sorted;
  size_t zero_begin, zero_end;   // workspace byte range of the zero-initialised tables
  size_t ones_begin, ones_end;   // ... and of the tables that start out as all-ones bytes (empty hash keys, empty grid cells)
};
static void carve_phase2(Bump& b, const int* sizes, a3d_scene* sc, Phase2Tmp& t, int64_t grid_cells) {
  for (int L = 0; L < A3D_NUM_LEVELS; ++L) {
    Level& lv = sc->lv[L];
    lv.n = sizes[L];
    lv.npad = (int)round_up(lv.n > 0 ? lv.n : 1, kTileRows);
  }
  // tables that start out as zeros sit in one contiguous region (one memset per scene)
  b.off = align256(b.off);
  t.zero_begin = b.off;
  for (int L = 0; L < A3D_NUM_LEVELS; ++L) {
    Level& lv = sc->lv[L];
    lv.gmask27 = b.take<uint32_t>(lv.npad / 16);
    if (L < A3D_NUM_LEVELS - 1) {
      const int npadC = sc->lv[L + 1].npad;
      lv.gmask_down = b.take<uint32_t>(npadC / 16);
      lv.gmask_up = b.take<uint32_t>(lv.npad / 16);
      lv.up_rows = b.take<int>(lv.npad);
    }
  }
  b.off = align256(b.off);
  t.zero_end = b.off;
  // hash keys of every level (kEmptyKey = all ones) and the level-0 grid (-1 = empty) in one region: one memset instead
  // of a clearing kernel + a memset; a level 0 that has the grid keeps no hash table (a token 1024 slots)
  t.ones_begin = b.off;
  for (int L = 0; L < A3D_NUM_LEVELS; ++L) {
    Level& lv = sc->lv[L];
    lv.hmask = (L == 0 && grid_cells > 0 ? 1024u : hash_capacity(lv.n)) - 1;
    lv.hkeys = b.take<uint64_t>((size_t)lv.hmask + 1);
  }
  sc->lv[0].grid = grid_cells > 0 ? b.take<int>((size_t)grid_cells) : nullptr;
  b.off = align256(b.off);
  t.ones_end = b.off;
  int tot = 0;
  for (int L = 0; L < A3D_NUM_LEVELS; ++L) {
    Level& lv = sc->lv[L];
    lv.perm = b.take<int>(lv.npad);
    lv.inv = b.take<int>(lv.npad);
    lv.xyzb = b.take<int32_t>((size_t)lv.npad * 4);
    lv.hvals = b.take<int>((size_t)lv.hmask + 1);
    lv.nbr27 = b.take<int>((size_t)27 * lv.npad);
    lv.pre27 = b.take<int>(lv.npad / 64 + 1);
    if (L < A3D_NUM_LEVELS - 1) {
      const int npadC = sc->lv[L + 1].npad;
      lv.pre_down = b.take<int>(npadC / 64 + 1);
      lv.pre_up = b.take<int>(lv.npad / 64 + 1);
      lv.child8 = b.take<int>((size_t)8 * npadC);
      lv.up8 = b.take<int>((size_t)8 * lv.npad);
    }
    t.nbrM[L] = b.take<int>((size_t)27 * lv.npad);
    tot += lv.npad;
  }
  sc->orig_row = b.take<int>(sc->lv[0].npad);
  t.cat_keys = b.take<uint64_t>(tot);
  t.cat_keys_sorted = b.take<uint64_t>(tot);
  t.cat_vals = b.take<int>(tot);
  t.cat_vals_sorted = b.take<int>(tot);
}

}  // namespace a3d

using namespace a3d;

extern "C" int a3d_version(void) { return A3D_ABI_VERSION; }

extern "C" int a3d_profile_enable(int on) {
  g_prof_on = on != 0;
  return A3D_OK;
}
extern "C" int a3d_profile_read(a3d_prof_entry* out, int max_entries) {
  int n = 0, bad = 0;
  hipError_t first = hipSuccess;
  for (auto& r : g_prof) {
    hipError_t e = hipEventSynchronize(r.e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.e0, r.e1);
    if (e != hipSuccess) {   // the entry is still returned (ms = -1), the call reports the failure
      ms = -1.f;
      if (!bad++) first = e;
    }
    r.e.ms = ms;
    if (out && n < max_entries) out[n++] = r.e;
    g_event_pool.push_back(r.e0);
    g_event_pool.push_back(r.e1);
  }
  g_prof.clear();
  if (bad) {
    set_error("a3d_profile_read: %d event pair(s) could not be read: %s", bad, hipGetErrorString(first));
    return A3D_ERR_HIP;
  }
  return n;
}
extern "C" const char* a3d_last_error(void) { return a3d::get_error(); }

extern "C" int a3d_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  A3D_HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
  A3D_HIP_CHECK(hipStreamSynchronize(st));
  return A3D_OK;
}

extern "C" size_t a3d_scene_workspace_bytes(int64_t n_voxels) {
  if (n_voxels <= 0 || n_voxels > (int64_t)1 << 28) return 0;
  const int n0 = (int)n_voxels;
  Bump b(nullptr);
  Phase1 p1;
  carve_phase1(b, n0, p1);
  a3d_scene tmp;
  int sizes[A3D_NUM_LEVELS];
  for (int L = 0; L < A3D_NUM_LEVELS; ++L) sizes[L] = n0;  // upper bound: n_L <= n_0
  Phase2Tmp t;
  carve_phase2(b, sizes, &tmp, t, kGridCellsPerVoxel * n0);
  return align256(b.off) + 4096;
}

extern "C" int a3d_scene_create(const int32_t* coords4_dev, int64_t n_voxels, void* workspace_dev,
                                size_t workspace_bytes, void* stream, a3d_scene** out) {
  if (!coords4_dev || !workspace_dev || !out || n_voxels <= 0 || n_voxels > (int64_t)1 << 28) {
    set_error("a3d_scene_create: bad arguments (n=%lld)", (long long)n_voxels);
    return A3D_ERR_INVALID;
  }
  if (((uintptr_t)workspace_dev & 255) != 0) {
    set_error("a3d_scene_create: workspace must be 256-byte aligned");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < a3d_scene_workspace_bytes(n_voxels)) {
    set_error("a3d_scene_create: workspace too small (%zu < %zu)", workspace_bytes,
              a3d_scene_workspace_bytes(n_voxels));
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int n0 = (int)n_voxels;
  Bump b(workspace_dev);
  Phase1 p;
  carve_phase1(b, n0, p);
  const int T = 256;
  auto nblk = [](int64_t n, int t) { return (unsigned)((n + t - 1) / t); };

  // The two barrier kernels of phase 1 (the one-launch sort, k_heads_all) spin at grid barriers: all their workgroups must get
  // onto the chip, which two of them waiting side by side could deny each other.  Phase 1 ends in a host synchronisation, so
  // a process-wide count of the calls between here and there is exact: only a call that finds itself ALONE uses them, a
  // concurrent one (another host thread, another stream) takes the launch chains.
  struct SoloPhase1 {
    bool alone;
    SoloPhase1() : alone(g_phase1_calls.fetch_add(1, std::memory_order_acq_rel) == 0) {}
    ~SoloPhase1() { g_phase1_calls.fetch_sub(1, std::memory_order_acq_rel); }
  } solo;
  // ---- phase 1: keys, sort, levels (all sized by the n0 bound; real sizes stay on the device)
  // The one-launch forms need every workgroup of their grids on the chip at once: they are used only where the device AS
  // THIS PROCESS SEES IT can hold the grid (barrier_grid_fits: a partitioned device has a fraction of the CUs), and when a
  // grid barrier gives up all the same (another process's kernels hold the CUs: ranks sharing one device) phase 1 runs
  // again through the launch chains, which wait only for workgroups that already run.
  int sizes[kSizesInts];
  const int nb = (n0 + 1023) / 1024;
  for (int attempt = 0; attempt < 2; ++attempt) {
  const bool barriers_ok = attempt == 0 && solo.alone;
  const bool heads_one = barriers_ok && nb <= kHeadsOneLaunch && barrier_grid_fits((const void*)k_heads_all, 1024, 0, nb);
  int prof1 = prof_enabled() ? prof_begin(st, A3D_PROF_SCENE_SORT, 0, 0, 0, 0, n0) : -1;
  for (int i = 0; i < kSizesInts; ++i) sizes[i] = i < 8 || i >= kSizesBar ? 0 : -1;
  for (int sl = 0; sl < kBBoxSlots; ++sl)
    for (int a = 0; a < 3; ++a) sizes[kBBox + 8 * sl + a] = INT_MAX, sizes[kBBox + 8 * sl + 3 + a] = INT_MIN;
  sizes[0] = n0;
  A3D_HIP_CHECK(hipMemcpyAsync(p.sizes_dev, sizes, sizeof(sizes), hipMemcpyHostToDevice, st));
  k_make_keys<<<nblk(n0, T), T, 0, st>>>(coords4_dev, n0, p.keys_in, p.vals_in, p.sizes_dev);
  A3D_LAUNCH_CHECK();
  {   // all 64 key bits: the digits the batch's extent does not touch are skipped on the device
    RadixPass ps[kRadixMaxPasses];
    const int np = radix_passes(0, 64, ps);
    int rc = radix_sort_pairs(p.sort_temp, p.sort_temp_bytes, p.keys_in, p.keys[0], p.vals_in, p.vals_sorted, n0, ps, np, st,
                              p.sizes_dev + 5,   // a look-back that gives up reports A3D_ERR_HIP through the error word read below
                              barriers_ok);
    if (rc) return rc;
  }
  {
    CoarseOut co;
    for (int L = 0; L < A3D_NUM_LEVELS - 1; ++L) {
      co.keys[L] = p.keys[L + 1];
      co.parentM[L] = p.parentM[L];
      co.firstM[L] = p.firstM[L];
    }
    if (heads_one) {
      k_heads_all<<<nb, 1024, 0, st>>>(p.keys[0], n0, nb, p.blocksums, p.sizes_dev, co);
    } else {
      k_heads_count<<<nb, 1024, 0, st>>>(p.keys[0], n0, nb, p.blocksums, p.sizes_dev);
      k_heads_scan<<<A3D_NUM_LEVELS - 1, 1024, 0, st>>>(p.blocksums, nb, p.sizes_dev);
      k_heads_write<<<nb, 1024, 0, st>>>(p.keys[0], n0, nb, p.blocksums, co);
    }
    A3D_LAUNCH_CHECK();
  }
  if (prof1 >= 0) prof_end(st, prof1);
  A3D_HIP_CHECK(hipMemcpyAsync(sizes, p.sizes_dev, sizeof(sizes), hipMemcpyDeviceToHost, st));
  A3D_HIP_CHECK(hipStreamSynchronize(st));
  if (sizes[5] == A3D_ERR_HIP && barriers_ok) continue;   // a grid barrier gave up: once more through the launch chains
  break;
  }
  if (sizes[5] != 0) {
    set_error(sizes[5] == A3D_ERR_DUPLICATE     ? "a3d_scene_create: duplicate voxel coordinates"
              : sizes[5] == A3D_ERR_COORD_RANGE ? "a3d_scene_create: coordinate out of range"
              : sizes[5] == A3D_ERR_HIP         ? "a3d_scene_create: the device-side sort gave up waiting for a workgroup (starved queue?)"
                                                : "a3d_scene_create: rows of a batch sample are not contiguous");
    return sizes[5];
  }
  const int n_batch = sizes[7] + 1;
  bool dense = sizes[6] == n_batch;
  for (int bi = 0; dense && bi < n_batch; ++bi)
    dense = sizes[8 + bi] >= 0 && (bi == 0 ? sizes[8] == 0 : sizes[8 + bi] > sizes[8 + bi - 1]);
  if (!dense) {
    set_error("a3d_scene_create: batch indices must be 0..B-1, each a contiguous row range in ascending order");
    return A3D_ERR_INVALID;
  }

  // ---- phase 2: per-level tables with exact sizes
  a3d_scene* sc = new a3d_scene();
  {
    static std::atomic<uint64_t> next_serial{1};
    sc->serial = next_serial.fetch_add(1);
  }
  sc->n0 = n0;
  sc->n_batch = n_batch;
  for (int bi = 0; bi < n_batch; ++bi) sc->batch_start[bi] = sizes[8 + bi];
  sc->workspace = workspace_dev;
  sc->workspace_bytes = workspace_bytes;
  // level 0 gets a dense voxel -> row grid when the padded bounding box of the batch has at most
  // kGridCellsPerVoxel cells per voxel (indoor scans: 15-30); sparser inputs keep the hash table
  int64_t grid_cells = 0;
  int gdim[3], gorg[3];
  {
    static int use_grid = -1;
    if (use_grid < 0) {
      const char* e = getenv("A3D_GRID");
      use_grid = e ? atoi(e) : 1;
    }
    grid_cells = n_batch;
    for (int a = 0; a < 3; ++a) {
      int lo = INT_MAX, hi = INT_MIN;
      for (int sl = 0; sl < kBBoxSlots; ++sl) {
        lo = std::min(lo, sizes[kBBox + 8 * sl + a]);
        hi = std::max(hi, sizes[kBBox + 8 * sl + 3 + a]);
      }
      gorg[a] = lo - kGridPad;
      gdim[a] = hi - lo + 1 + 2 * kGridPad;
      grid_cells = grid_cells <= kGridCellsPerVoxel * n0 ? grid_cells * gdim[a] : grid_cells;
    }
    if (!use_grid || grid_cells > kGridCellsPerVoxel * n0) grid_cells = 0;
  }
  Phase2Tmp t;
  carve_phase2(b, sizes, sc, t, grid_cells);
  for (int a = 0; a < 3; ++a) sc->lv[0].gorg[a] = gorg[a], sc->lv[0].gdim[a] = gdim[a];
  if (b.off > workspace_bytes) {
    delete sc;
    set_error("a3d_scene_create: internal workspace overflow");
    return A3D_ERR_WORKSPACE;
  }
  ProfScope prof2(st, A3D_PROF_SCENE_TABLES, 0, 0, 0, 0, n0);
  A3D_HIP_CHECK(hipMemsetAsync((char*)workspace_dev + t.zero_begin, 0, t.zero_end - t.zero_begin, st));
  A3D_HIP_CHECK(hipMemsetAsync((char*)workspace_dev + t.ones_begin, 0xff, t.ones_end - t.ones_begin, st));   // empty keys, grid cells = -1
  LevelSet S;
  memset(&S, 0, sizeof(S));
  S.st_shift = super_tile_shift();
  // key = level << (27 + super-tile bits) | super tile << 27 | small key: the sorts only look at the bits in use
  int tile_bits = 0;
  while (((sc->lv[0].npad - 1) >> S.st_shift) >> tile_bits) ++tile_bits;
  S.level_shift = 27 + tile_bits;
  S.cat_keys = t.cat_keys;
  S.cat_vals = t.cat_vals;
  S.cat_sorted = t.cat_vals_sorted;
  S.vals_sorted0 = p.vals_sorted;
  S.orig_row = sc->orig_row;
  S.off[0] = 0;
  for (int L = 0; L < A3D_NUM_LEVELS; ++L) {
    Level& lv = sc->lv[L];
    lv.keys = p.keys[L];
    lv.parentM = L < A3D_NUM_LEVELS - 1 ? p.parentM[L] : nullptr;
    S.lv[L] = lv;
    S.nbrM[L] = t.nbrM[L];
    if (L < A3D_NUM_LEVELS - 1) S.firstM[L] = p.firstM[L];
    S.off[L + 1] = S.off[L] + lv.n;
  }
  // one launch per kernel for all levels: blocks [blk[L], blk[L+1]) belong to level L
  auto blocks = [&](int nlev, auto count) -> unsigned {
    S.nlev = nlev;
    S.blk[0] = 0;
    for (int L = 0; L < nlev; ++L) S.blk[L + 1] = S.blk[L] + (int)nblk((int64_t)count(L), T);
    return (unsigned)S.blk[nlev];
  };
  // key = level | super tile | small key (27-bit neighbour mask or 3-bit child slot): only the bits in use are sorted
  auto sort_cat = [&](int nlev, int small_bits) -> int {
    RadixPass ps[kRadixMaxPasses];
    int np;
    if (small_bits == 27) {
      np = radix_passes(0, S.level_shift + 3, ps);
    } else if (small_bits + S.level_shift + 3 - 27 <= 8) {   // child slot + super tile + level fit one digit
      ps[0].shift = 0, ps[0].bits = small_bits, ps[0].shift2 = 27, ps[0].bits2 = S.level_shift + 3 - 27;
      np = 1;
    } else {
      np = radix_passes(0, small_bits, ps);
      np = radix_passes(27, S.level_shift + 3, ps, np);
    }
    return radix_sort_pairs(p.sort_temp, p.sort_temp_bytes, t.cat_keys, t.cat_keys_sorted, t.cat_vals, t.cat_vals_sorted,
                            S.off[nlev], ps, np, st);
  };
  const int NL = A3D_NUM_LEVELS;
  unsigned g;
  // ---- stage A: hash, neighbours in Morton order, sort keys
  g = blocks(NL, [&](int L) { return (int64_t)sc->lv[L].n; });
  k_hash_insert<<<g, T, 0, st>>>(S);
  // neighbour rows + presence masks + the sort keys: one thread per row on the level-0 grid, 32 threads per row on hash levels
  g = blocks(NL, [&](int L) { return sc->lv[L].grid ? (int64_t)sc->lv[L].n : (int64_t)sc->lv[L].n * 32; });
  k_nbr_morton<<<g, T, 0, st>>>(S);
  g = blocks(NL, [&](int L) { return (int64_t)sc->lv[L].n; });
  A3D_LAUNCH_CHECK();
  // ---- stage B: ONE stable radix sort re-orders the rows of all levels inside their super tiles
  {
    int rc = sort_cat(NL, 27);
    if (rc) { delete sc; return rc; }
  }
  // ---- stage C: tables in the new row order
  k_perm_from_sorted<<<g, T, 0, st>>>(S, 0);
  g = blocks(NL, [&](int L) { return (int64_t)sc->lv[L].npad; });
  k_remap_nbr<<<g, T, 0, st>>>(S);
  g = blocks(NL, [&](int L) { return std::max<int64_t>(sc->lv[L].n, (int64_t)sc->lv[L].hmask + 1); });
  k_xyzb_hashfix<<<g, T, 0, st>>>(S);
  A3D_LAUNCH_CHECK();
  // ---- stage D: stride-2 tables + child-slot sort keys of the fine rows (levels 0..3), second sort, up tables
  g = blocks(NL - 1, [&](int L) { return (int64_t)sc->lv[L + 1].npad; });
  k_child8<<<g, T, 0, st>>>(S);
  A3D_LAUNCH_CHECK();
  {
    int rc = sort_cat(NL - 1, 3);
    if (rc) { delete sc; return rc; }
  }
  g = blocks(NL - 1, [&](int L) { return (int64_t)sc->lv[L].n; });
  k_perm_from_sorted<<<g, T, 0, st>>>(S, 1);
  g = blocks(NL - 1, [&](int L) { return (int64_t)sc->lv[L].npad; });
  k_up<<<g, T, 0, st>>>(S);
  A3D_LAUNCH_CHECK();
  k_tile_prefix<<<dim3(NL, 3), 1024, 0, st>>>(S);   // all three mask tables are final here
  A3D_LAUNCH_CHECK();
  *out = sc;
  return A3D_OK;
}

extern "C" void a3d_scene_destroy(a3d_scene* s) {
  if (s) a3d::wgrad_scene_release(s);
  delete s;
}

extern "C" int a3d_scene_batch_ranges(const a3d_scene* s, int64_t* starts_out, int max_out) {
  if (!s) return A3D_ERR_INVALID;
  for (int i = 0; i < s->n_batch && i < max_out; ++i) starts_out[i] = s->batch_start[i];
  return s->n_batch;
}

extern "C" int a3d_scene_grid_dims(const a3d_scene* s, int dims_out[3]) {
  if (!s) return A3D_ERR_INVALID;
  if (!s->lv[0].grid) return 0;
  if (dims_out)
    for (int a = 0; a < 3; ++a) dims_out[a] = s->lv[0].gdim[a];
  return 1;
}

extern "C" int64_t a3d_scene_level_size(const a3d_scene* s, int level) {
  if (!s || level < 0 || level >= A3D_NUM_LEVELS) return -1;
  return s->lv[level].n;
}

extern "C" int a3d_scene_table(const a3d_scene* s, int level, int which, const void** ptr_dev, int64_t* count) {
  if (!s || !ptr_dev || !count || level < 0 || level >= A3D_NUM_LEVELS) {
    set_error("a3d_scene_table: bad arguments");
    return A3D_ERR_INVALID;
  }
  const Level& lv = s->lv[level];
  const bool has_coarse = level < A3D_NUM_LEVELS - 1;
  const int npadC = has_coarse ? s->lv[level + 1].npad : 0;
  switch (which) {
    case A3D_TAB_XYZB: *ptr_dev = lv.xyzb; *count = (int64_t)lv.n * 4; break;
    case A3D_TAB_NBR27: *ptr_dev = lv.nbr27; *count = (int64_t)27 * lv.npad; break;
    case A3D_TAB_GMASK27: *ptr_dev = lv.gmask27; *count = lv.npad / 16; break;
    case A3D_TAB_PRE27: *ptr_dev = lv.pre27; *count = lv.npad / 64 + 1; break;
    case A3D_TAB_PREDOWN: if (!has_coarse) goto bad; *ptr_dev = lv.pre_down; *count = npadC / 64 + 1; break;
    case A3D_TAB_PREUP: if (!has_coarse) goto bad; *ptr_dev = lv.pre_up; *count = lv.npad / 64 + 1; break;
    case A3D_TAB_CHILD8: if (!has_coarse) goto bad; *ptr_dev = lv.child8; *count = (int64_t)8 * npadC; break;
    case A3D_TAB_GMASKDOWN: if (!has_coarse) goto bad; *ptr_dev = lv.gmask_down; *count = npadC / 16; break;
    case A3D_TAB_UP8: if (!has_coarse) goto bad; *ptr_dev = lv.up8; *count = (int64_t)8 * lv.npad; break;
    case A3D_TAB_GMASKUP: if (!has_coarse) goto bad; *ptr_dev = lv.gmask_up; *count = lv.npad / 16; break;
    case A3D_TAB_UPROWS: if (!has_coarse) goto bad; *ptr_dev = lv.up_rows; *count = lv.npad; break;
    case A3D_TAB_ORIGROW: if (level != 0) goto bad; *ptr_dev = s->orig_row; *count = lv.n; break;
    default: goto bad;
  }
  return A3D_OK;
bad:
  set_error("a3d_scene_table: table %d does not exist at level %d", which, level);
  return A3D_ERR_INVALID;
}
