// Stable LSD radix sort of (uint64 key, int32 value) pairs for the scene build (scene.hip): one histogram sweep for
// all digits, one planning workgroup, then ONE kernel per digit (chained scan with decoupled look-back: a workgroup
// publishes its digit counts, adds up the counts of the workgroups before it and scatters).  Digits whose value is the
// same for every key are skipped ON THE DEVICE (the pass exits at once): the level-0 Morton keys are 64 bits wide but
// only the bits the scene's extent touches differ, and which those are is not known on the host before the sort.
//
// Why not rocPRIM: below ~1 M items its radix_sort_pairs is a merge sort of ~18 dependent launches, and the scene build
// runs three sorts per batch (rows of level 0 by Morton key; rows of all levels by neighbour pattern; fine rows by child
// slot) -- 0.4 of the 0.74 ms of a 4-scene build, launch-bound at one scene.  Here a sort is 2 + (number of digits)
// launches, typically 3-7.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace a3d {

namespace {

constexpr int kRsThreads = 256;
constexpr int kRsKeysPerThread = 16;
constexpr int kRsBlockKeys = kRsThreads * kRsKeysPerThread;   // 4096
constexpr int kLook = 16;   // look-back words in flight (an agent-scope load is a ~1 us round trip past the XCD's L2)
constexpr uint32_t kFlagAgg = 1u << 30, kFlagInc = 1u << 31, kValMask = (1u << 30) - 1u;

struct RsPlanDev {      // written by k_rs_plan, read by every pass
  int skip[kRadixMaxPasses];
  int src[kRadixMaxPasses];   // 0 = the caller's input, 1 = the caller's output, 2 = the temporary pair
  int dst[kRadixMaxPasses];
};

struct RsArgs {
  const uint64_t* keys_in;
  const int* vals_in;
  uint64_t *keys_out, *keys_tmp;
  int *vals_out, *vals_tmp;
  int n, npass, nblocks;
  int shift[kRadixMaxPasses], bits[kRadixMaxPasses], shift2[kRadixMaxPasses], bits2[kRadixMaxPasses];
  uint32_t* ghist;     // [npass][256]  counts, then exclusive starts
  RsPlanDev* plan;
  int* tickets;        // [npass]
  uint32_t* status;    // [npass][nblocks][256]  look-back words
  int* fail;           // optional: receives A3D_ERR_HIP (atomicMin) when a look-back gives up waiting
};

__device__ __forceinline__ int rs_digit(uint64_t k, int shift, int bits, int shift2, int bits2) {
  return (int)(((k >> shift) & ((1u << bits) - 1u)) | (((k >> shift2) & ((1u << bits2) - 1u)) << bits));
}

// every digit's histogram in one sweep
__global__ void __launch_bounds__(kRsThreads) k_rs_hist(const RsArgs a) {
  __shared__ uint32_t h[kRadixMaxPasses * 256];
  for (int i = threadIdx.x; i < a.npass * 256; i += kRsThreads) h[i] = 0;
  __syncthreads();
  for (int i = blockIdx.x * kRsThreads + threadIdx.x; i < a.n; i += gridDim.x * kRsThreads) {
    const uint64_t k = a.keys_in[i];
    for (int p = 0; p < a.npass; ++p) atomicAdd(&h[p * 256 + rs_digit(k, a.shift[p], a.bits[p], a.shift2[p], a.bits2[p])], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.npass * 256; i += kRsThreads)
    if (h[i]) atomicAdd(&a.ghist[i], h[i]);
}

// one workgroup: exclusive starts per digit, which passes are trivial, which buffer each pass reads / writes
__global__ void __launch_bounds__(256) k_rs_plan(const RsArgs a) {
  __shared__ int lds[8];
  __shared__ int trivial[kRadixMaxPasses];
  const int d = threadIdx.x, lane = d & 63, w = d >> 6;
  for (int p = 0; p < a.npass; ++p) {
    const uint32_t c = a.ghist[p * 256 + d];
    if (d == 0) trivial[p] = 0;
    __syncthreads();
    if (c == (uint32_t)a.n) trivial[p] = 1;   // one digit value holds every key
    // exclusive scan of the 256 counts
    int v = (int)c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o, 64);
      if (lane >= o) v += t;
    }
    if (lane == 63) lds[w] = v;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < w; ++q) base += lds[q];
    a.ghist[p * 256 + d] = (uint32_t)(base + v - (int)c);
    __syncthreads();
  }
  if (d == 0) {
    int m = 0;
    for (int p = 0; p < a.npass; ++p) m += trivial[p] ? 0 : 1;
    if (m == 0) trivial[0] = 0, m = 1;   // nothing to sort: the first pass copies input -> output
    int j = 0, cur = 0;
    for (int p = 0; p < a.npass; ++p) {
      a.plan->skip[p] = trivial[p];
      if (trivial[p]) continue;
      ++j;
      const int dst = ((m - j) & 1) ? 2 : 1;   // the last executed pass lands in the caller's output
      a.plan->src[p] = cur;
      a.plan->dst[p] = dst;
      cur = dst;
    }
  }
}

// one digit of one 4096-key block, stable: block b of pass p from buffer s to buffer t (0 = the caller's input, 1 = the
// caller's output, 2 = the temporary pair); gstart = the exclusive digit starts of the pass.  b < 0: the block number is a
// ticket drawn here (k_rs_pass: workgroups before this one in key order are then already running).
template <int KPT>   // keys per thread: 16 in the launch chain, 4 in the one-launch sort (four times the workgroups, a quarter of the chain each)
__device__ __forceinline__ void rs_block(const RsArgs& a, int p, int b_in, int s, int t, const uint32_t* gstart) {
  constexpr int kBlockKeys = kRsThreads * KPT;
  __shared__ uint64_t skeys[kBlockKeys];
  __shared__ int svals[kBlockKeys];
  __shared__ uint32_t whist[4][256];   // per-wave digit counts, then per-wave exclusive offsets
  __shared__ int lstart[256];          // first slot of a digit in the workgroup's sorted order
  __shared__ int gpos[256];            // global position of slot 0 of a digit, minus lstart
  __shared__ int misc[8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int shift = a.shift[p], bits = a.bits[p], shift2 = a.shift2[p], bits2 = a.bits2[p];
  const uint64_t* kin = s == 0 ? a.keys_in : (s == 1 ? a.keys_out : a.keys_tmp);
  const int* vin = s == 0 ? a.vals_in : (s == 1 ? a.vals_out : a.vals_tmp);
  uint64_t* kout = t == 1 ? a.keys_out : a.keys_tmp;
  int* vout = t == 1 ? a.vals_out : a.vals_tmp;

  __syncthreads();   // (a workgroup that walks several blocks: the previous block's LDS reads are over)
  if (tid == 0) misc[0] = b_in >= 0 ? b_in : atomicAdd(&a.tickets[p], 1);
  for (int i = tid; i < 4 * 256; i += kRsThreads) (&whist[0][0])[i] = 0;
  __syncthreads();
  const int b = misc[0];
  const int base = b * kBlockKeys;
  const int cnt = min(kBlockKeys, a.n - base);

  // ---- rank every key among the keys of its wave with the same digit (rows of 64 in index order)
  uint64_t key[KPT];
  int val[KPT], rank[KPT];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int li = w * (kBlockKeys / 4) + r * 64 + lane;
    const bool ok = li < cnt;
    key[r] = ok ? kin[base + li] : ~0ull;
    val[r] = ok ? vin[base + li] : 0;
  }
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int li = w * (kBlockKeys / 4) + r * 64 + lane;
    const bool ok = li < cnt;
    const int d = rs_digit(key[r], shift, bits, shift2, bits2);
    unsigned long long peers = __ballot(ok);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      if (bit < bits + bits2) {   // uniform
        const unsigned long long bal = __ballot((d >> bit) & 1);
        peers &= ((d >> bit) & 1) ? bal : ~bal;
      }
    }
    int before = 0;
    if (ok) {
      const int leader = __builtin_ctzll(peers);
      if (lane == leader) {
        before = (int)whist[w][d];
        whist[w][d] = (uint32_t)(before + __builtin_popcountll(peers));
      }
      before = __shfl(before, leader, 64);
    }
    rank[r] = before + __builtin_popcountll(peers & lt);
  }
  __syncthreads();

  // ---- digit d = thread d: counts of the workgroup, per-wave offsets, look-back
  {
    const int d = tid;
    const uint32_t c0 = whist[0][d], c1 = whist[1][d], c2 = whist[2][d], c3 = whist[3][d];
    const uint32_t tot = c0 + c1 + c2 + c3;
    whist[0][d] = 0;
    whist[1][d] = c0;
    whist[2][d] = c0 + c1;
    whist[3][d] = c0 + c1 + c2;
    uint32_t* st = a.status + ((size_t)p * a.nblocks + b) * 256 + d;
    __hip_atomic_store(st, tot | (b == 0 ? kFlagInc : kFlagAgg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // exclusive scan of tot over the digits -> lstart
    int v = (int)tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int x = __shfl_up(v, o, 64);
      if (lane >= o) v += x;
    }
    if (lane == 63) misc[1 + w] = v;
    __syncthreads();
    int wbase = 0;
    for (int q = 0; q < w; ++q) wbase += misc[1 + q];
    const int ls = wbase + v - (int)tot;
    lstart[d] = ls;
    // look back, kLook workgroups at a time (their loads are in flight together; only a word not yet published is
    // polled): a workgroup meets mostly AGGREGATE words -- all workgroups of a pass run at once
    uint32_t excl = 0;
    constexpr int kLookN = KPT <= 4 ? 2 * kLook : kLook;   // the one-launch sort has four times the blocks to look back over
    for (int pb = b - 1; pb >= 0; pb -= kLookN) {
      uint32_t x[kLookN];
#pragma unroll
      for (int u = 0; u < kLookN; ++u)
        x[u] = pb - u >= 0 ? __hip_atomic_load(a.status + ((size_t)p * a.nblocks + (pb - u)) * 256 + d, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)
                           : kFlagInc;
      bool done = false;
#pragma unroll
      for (int u = 0; u < kLookN; ++u) {
        if (done) break;
        unsigned spins = 0;
        while ((x[u] & (kFlagAgg | kFlagInc)) == 0u) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 26)) {        // seconds: never in a healthy run (every workgroup publishes before it looks back);
            if (a.fail) atomicMin(a.fail, A3D_ERR_HIP);   // do not hang the device, and say that the result is not a sort
            break;
          }
          x[u] = __hip_atomic_load(a.status + ((size_t)p * a.nblocks + (pb - u)) * 256 + d, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        excl += x[u] & kValMask;
        done = (x[u] & kFlagInc) != 0u;
      }
      if (done) break;
    }
    if (b > 0) __hip_atomic_store(st, (excl + tot) | kFlagInc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gpos[d] = (int)(gstart[d] + excl) - ls;
  }
  __syncthreads();

  // ---- into sorted order inside the workgroup (LDS), then out in runs
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int li = w * (kBlockKeys / 4) + r * 64 + lane;
    if (li < cnt) {
      const int d = rs_digit(key[r], shift, bits, shift2, bits2);
      const int slot = lstart[d] + (int)whist[w][d] + rank[r];
      skeys[slot] = key[r];
      svals[slot] = val[r];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int slot = r * kRsThreads + tid;
    if (slot < cnt) {
      const uint64_t k = skeys[slot];
      const int dst = gpos[rs_digit(k, shift, bits, shift2, bits2)] + slot;
      kout[dst] = k;
      vout[dst] = svals[slot];
    }
  }
}

__global__ void __launch_bounds__(kRsThreads) k_rs_pass(const RsArgs a, int p) {
  if (a.plan->skip[p]) return;
  rs_block<kRsKeysPerThread>(a, p, -1, a.plan->src[p], a.plan->dst[p], a.ghist + p * 256);
}

// ---- the whole sort in ONE launch (small inputs: <= kRsOneLaunchBlocks blocks of 1024 keys, i.e. a scene or a small batch, where the
// 3-11 launches of the chain above ARE the sort's time: 93 us for 80 k keys, five of its eight digit launches exiting at
// once).  One workgroup per block, all resident: histogram sweep -> grid barrier -> every workgroup derives the plan (which
// digits are trivial, the buffers' ping-pong, a digit's exclusive starts) from the global histogram by itself -> per
// non-trivial digit the block pass above and a grid barrier.  The barrier is a monotonic counter (release fence before the
// arrival, acquire after the wait: a pass reads what workgroups of other XCDs wrote); its spin is bounded like the
// look-back's.  Large inputs keep the launch chain (their workgroups would not all be resident next to other streams' work),
// and so does every sort whose caller cannot rule out a second barrier kernel of the process in flight (the `one_launch` argument).
constexpr int kRsSmallKpt = 4;               // 1024 keys per workgroup
constexpr int kRsOneLaunchBlocks = 128;      // <= 131 072 keys: one scene.  (A 4-scene batch -- 313 blocks -- measured 2 % SLOWER with
                                             // four steps in flight: 313 workgroups spinning at barriers next to the other streams' kernels)
__global__ void __launch_bounds__(kRsThreads) k_rs_sort(const RsArgs a, unsigned* bar) {
  __shared__ uint32_t h[kRadixMaxPasses * 256];   // this workgroup's histograms, then the current digit's exclusive starts
  __shared__ int plan_l[3 * kRadixMaxPasses + 4];
  __shared__ int scan_l[8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned G = gridDim.x;
  for (int i = tid; i < a.npass * 256; i += kRsThreads) h[i] = 0;
  __syncthreads();
  for (int i = blockIdx.x * kRsThreads + tid; i < a.n; i += gridDim.x * kRsThreads) {
    const uint64_t k = a.keys_in[i];
    for (int p = 0; p < a.npass; ++p) atomicAdd(&h[p * 256 + rs_digit(k, a.shift[p], a.bits[p], a.shift2[p], a.bits2[p])], 1u);
  }
  __syncthreads();
  for (int i = tid; i < a.npass * 256; i += kRsThreads)
    if (h[i]) atomicAdd(&a.ghist[i], h[i]);
  unsigned epoch = 1;
  grid_barrier_counter(bar, G * epoch++, a.fail);
  // ---- the plan, by every workgroup for itself (as k_rs_plan)
  for (int i = tid; i < a.npass * 256; i += kRsThreads)
    h[i] = __hip_atomic_load(a.ghist + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid < kRadixMaxPasses) plan_l[tid] = 0;
  __syncthreads();
  for (int i = tid; i < a.npass * 256; i += kRsThreads)
    if (h[i] == (uint32_t)a.n) plan_l[i >> 8] = 1;   // one digit value holds every key: trivial
  __syncthreads();
  if (tid == 0) {
    int m = 0;
    for (int p = 0; p < a.npass; ++p) m += plan_l[p] ? 0 : 1;
    if (m == 0) plan_l[0] = 0, m = 1;
    int j = 0, cur = 0;
    for (int p = 0; p < a.npass; ++p) {
      if (plan_l[p]) continue;
      ++j;
      const int dst = ((m - j) & 1) ? 2 : 1;
      plan_l[kRadixMaxPasses + p] = cur;
      plan_l[2 * kRadixMaxPasses + p] = dst;
      cur = dst;
    }
  }
  __syncthreads();
  for (int p = 0; p < a.npass; ++p) {
    if (plan_l[p]) continue;   // uniform over the grid
    {   // exclusive scan of the digit's 256 counts, in place
      const uint32_t c = h[p * 256 + tid];
      int v = (int)c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int x = __shfl_up(v, o, 64);
        if (lane >= o) v += x;
      }
      if (lane == 63) scan_l[w] = v;
      __syncthreads();
      int base = 0;
      for (int q = 0; q < w; ++q) base += scan_l[q];
      h[p * 256 + tid] = (uint32_t)(base + v - (int)c);
      __syncthreads();
    }
    for (int b = blockIdx.x; b < a.nblocks; b += gridDim.x)
      rs_block<kRsSmallKpt>(a, p, b, plan_l[kRadixMaxPasses + p], plan_l[2 * kRadixMaxPasses + p], h + p * 256);
    grid_barrier_counter(bar, G * epoch++, a.fail);
  }
}

inline int rs_blocks(int n) { return (n + kRsBlockKeys - 1) / kRsBlockKeys; }
inline int rs_blocks_small(int n) { return (n + kRsThreads * kRsSmallKpt - 1) / (kRsThreads * kRsSmallKpt); }
// status words enough for every n <= n_max (a temporary sized once serves smaller sorts too): the chain's blocks at n_max, or
// the one-launch sort's (four times as many at the same n, but never more than kRsOneLaunchBlocks)
inline size_t rs_status_blocks(int n_max) {
  const size_t big = (size_t)rs_blocks(n_max), small = (size_t)rs_blocks_small(n_max);
  const size_t cap = small < (size_t)kRsOneLaunchBlocks ? small : (size_t)kRsOneLaunchBlocks;
  return big > cap ? big : cap;
}

}  // namespace

size_t radix_sort_temp_bytes(int n_max) {
  const size_t nb = rs_status_blocks(n_max > 0 ? n_max : 1);
  size_t b = 0;
  b += align256((size_t)n_max * sizeof(uint64_t));                        // keys_tmp
  b += align256((size_t)n_max * sizeof(int));                             // vals_tmp
  b += align256((size_t)kRadixMaxPasses * 256 * sizeof(uint32_t));        // ghist      } zeroed per sort,
  b += align256((size_t)kRadixMaxPasses * sizeof(int) + 64);              // tickets, grid-barrier counter   } one memset
  b += align256((size_t)kRadixMaxPasses * nb * 256 * sizeof(uint32_t));   // status     }
  b += align256(sizeof(RsPlanDev));
  return b + 256;
}

int radix_sort_pairs(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const int* vals_in,
                     int* vals_out, int n, const RadixPass* passes, int npass, hipStream_t st, int* fail_dev, bool allow_one_launch) {
  if (n <= 0) return A3D_OK;
  if (npass < 1 || npass > kRadixMaxPasses || !temp || ((uintptr_t)temp & 255) || temp_bytes < radix_sort_temp_bytes(n)) {
    set_error("radix_sort_pairs: bad arguments (n=%d, passes=%d, temp=%zu)", n, npass, temp_bytes);
    return A3D_ERR_INVALID;
  }
  RsArgs a;
  memset(&a, 0, sizeof(a));
  a.keys_in = keys_in;
  a.vals_in = vals_in;
  a.keys_out = keys_out;
  a.vals_out = vals_out;
  a.n = n;
  a.npass = npass;
  a.nblocks = rs_blocks(n);
  a.fail = fail_dev;
  for (int p = 0; p < npass; ++p) {
    if (passes[p].bits < 1 || passes[p].bits2 < 0 || passes[p].bits + passes[p].bits2 > 8 || passes[p].shift < 0 ||
        passes[p].shift + passes[p].bits > 64 || passes[p].shift2 < 0 || passes[p].shift2 + passes[p].bits2 > 64) {
      set_error("radix_sort_pairs: digit %d (shift %d, %d bits)", p, passes[p].shift, passes[p].bits);
      return A3D_ERR_INVALID;
    }
    a.shift[p] = passes[p].shift;
    a.bits[p] = passes[p].bits;
    a.shift2[p] = passes[p].shift2;
    a.bits2[p] = passes[p].bits2;
  }
  char* c = (char*)temp;
  a.keys_tmp = (uint64_t*)c;
  c += align256((size_t)n * sizeof(uint64_t));
  a.vals_tmp = (int*)c;
  c += align256((size_t)n * sizeof(int));
  char* zero_begin = c;
  a.ghist = (uint32_t*)c;
  c += align256((size_t)kRadixMaxPasses * 256 * sizeof(uint32_t));
  a.tickets = (int*)c;
  unsigned* bar = (unsigned*)(a.tickets + kRadixMaxPasses);   // 64 bytes behind the tickets
  c += align256((size_t)kRadixMaxPasses * sizeof(int) + 64);
  static int one_launch = -1;
  if (one_launch < 0) {
    const char* e = getenv("A3D_SORT_ONE_LAUNCH");   // 0: the launch chain for every size (A/B, tests)
    one_launch = e ? atoi(e) : 1;
  }
  const bool small = allow_one_launch && one_launch && rs_blocks_small(n) <= kRsOneLaunchBlocks &&
                     barrier_grid_fits((const void*)k_rs_sort, kRsThreads, 0, (int)rs_blocks_small(n));
  if (small) a.nblocks = rs_blocks_small(n);
  a.status = (uint32_t*)c;
  c += align256((size_t)npass * a.nblocks * 256 * sizeof(uint32_t));
  char* zero_end = c;
  a.plan = (RsPlanDev*)c;
  A3D_HIP_CHECK(hipMemsetAsync(zero_begin, 0, (size_t)(zero_end - zero_begin), st));
  if (small) {
    k_rs_sort<<<a.nblocks, kRsThreads, 0, st>>>(a, bar);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
  }
  int hb = (n + kRsThreads * 8 - 1) / (kRsThreads * 8);
  if (hb > 512) hb = 512;
  k_rs_hist<<<hb, kRsThreads, 0, st>>>(a);
  k_rs_plan<<<1, 256, 0, st>>>(a);
  for (int p = 0; p < npass; ++p) k_rs_pass<<<a.nblocks, kRsThreads, 0, st>>>(a, p);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

// ---- prefix sums of two int arrays in one go (quantize.hip: run numbers of the sorted keys and voxel numbers in point
// order): a inclusive, b exclusive.  Three launches: 4096-element block sums, one workgroup scans them, blocks re-scan.
namespace {
constexpr int kScanBlock = 4096, kScanThreads = 256, kScanPer = kScanBlock / kScanThreads;   // 16 per thread
__device__ __forceinline__ int scan_wave_incl(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
__global__ void __launch_bounds__(kScanThreads) k_scan2_sums(const int* __restrict__ a, const int* __restrict__ b, int64_t n,
                                                             int* sums /*[2][nblocks]*/, int nblocks) {
  __shared__ int red[2][4];
  const int64_t base = (int64_t)blockIdx.x * kScanBlock;
  int sa = 0, sb = 0;
  for (int e = threadIdx.x; e < kScanBlock; e += kScanThreads)
    if (base + e < n) sa += a[base + e], sb += b[base + e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sa += __shfl_xor(sa, o, 64), sb += __shfl_xor(sb, o, 64);
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = sa, red[1][threadIdx.x >> 6] = sb;
  __syncthreads();
  if (threadIdx.x == 0) {
    sums[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    sums[nblocks + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}
__global__ void __launch_bounds__(1024) k_scan2_blocks(int* sums, int nblocks) {   // exclusive, in place; grid = 2
  __shared__ int wsum[16];
  __shared__ int carry;
  int* s = sums + (size_t)blockIdx.x * nblocks;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int i0 = 0; i0 < nblocks; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    const int v = i < nblocks ? s[i] : 0;
    const int inc = scan_wave_incl(v, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int wbase = 0;
    for (int q = 0; q < w; ++q) wbase += wsum[q];
    const int c = carry;
    if (i < nblocks) s[i] = c + wbase + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + wbase + inc;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(kScanThreads) k_scan2_apply(const int* __restrict__ a, const int* __restrict__ b, int64_t n,
                                                              const int* __restrict__ sums, int nblocks, int* a_incl,
                                                              int* b_excl) {
  __shared__ int wsum[2][4];
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + (int64_t)threadIdx.x * kScanPer;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int va[kScanPer], vb[kScanPer], ta = 0, tb = 0;
#pragma unroll
  for (int e = 0; e < kScanPer; ++e) {
    va[e] = base + e < n ? a[base + e] : 0;
    vb[e] = base + e < n ? b[base + e] : 0;
    ta += va[e];
    tb += vb[e];
  }
  const int ia = scan_wave_incl(ta, lane), ib = scan_wave_incl(tb, lane);
  if (lane == 63) wsum[0][w] = ia, wsum[1][w] = ib;
  __syncthreads();
  int oa = sums[blockIdx.x] + ia - ta, ob = sums[nblocks + blockIdx.x] + ib - tb;
  for (int q = 0; q < w; ++q) oa += wsum[0][q], ob += wsum[1][q];
#pragma unroll
  for (int e = 0; e < kScanPer; ++e) {
    oa += va[e];
    if (base + e < n) {
      a_incl[base + e] = oa;
      b_excl[base + e] = ob;
    }
    ob += vb[e];
  }
}
}  // namespace

size_t scan2_temp_bytes(int64_t n) { return align256((size_t)2 * ((n + kScanBlock - 1) / kScanBlock) * sizeof(int)) + 256; }

int scan2_incl_excl(void* temp, size_t temp_bytes, const int* a, const int* b, int64_t n, int* a_incl, int* b_excl,
                    hipStream_t st) {
  if (n <= 0) return A3D_OK;
  if (!temp || temp_bytes < scan2_temp_bytes(n) || n > (int64_t)1 << 31) {
    set_error("scan2: bad arguments");
    return A3D_ERR_INVALID;
  }
  const int nblocks = (int)((n + kScanBlock - 1) / kScanBlock);
  int* sums = (int*)temp;
  k_scan2_sums<<<nblocks, kScanThreads, 0, st>>>(a, b, n, sums, nblocks);
  k_scan2_blocks<<<2, 1024, 0, st>>>(sums, nblocks);
  k_scan2_apply<<<nblocks, kScanThreads, 0, st>>>(a, b, n, sums, nblocks, a_incl, b_excl);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

bool barrier_grid_fits(const void* func, int block_threads, size_t dyn_lds_bytes, int grid) {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, func, block_threads, dyn_lds_bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;   // cannot tell: the launch chains need no residency
  }
  return (int64_t)per_cu * cus >= grid;
}

}  // namespace a3d

// Test / utility entry: stable sort of (key, value) pairs by the key bits [bit_begin, bit_end), ascending.
extern "C" size_t a3d_sort_pairs_workspace_bytes(int64_t n) {
  if (n <= 0 || n > (int64_t)1 << 28) return 0;
  return a3d::radix_sort_temp_bytes((int)n) + 256;   // + the give-up word of a one-launch sort
}
extern "C" int a3d_sort_pairs_u64(const uint64_t* keys_in_dev, const int32_t* vals_in_dev, int64_t n, int bit_begin,
                                  int bit_end, uint64_t* keys_out_dev, int32_t* vals_out_dev, void* workspace_dev,
                                  size_t workspace_bytes, void* stream) {
  using namespace a3d;
  if (n < 0 || n > (int64_t)1 << 28 || bit_begin < 0 || bit_end > 64 || bit_end <= bit_begin || !keys_in_dev || !vals_in_dev ||
      !keys_out_dev || !vals_out_dev) {
    set_error("a3d_sort_pairs_u64: bad arguments");
    return A3D_ERR_INVALID;
  }
  RadixPass ps[kRadixMaxPasses];
  const int np = radix_passes(bit_begin, bit_end, ps);
  // A stand-alone sort: the one-launch form for scene-sized inputs, like the first sort of a scene build -- with its own
  // give-up word (the first int of the workspace's tail, cleared here): a grid barrier that gave up leaves unsorted data, so
  // the word is read back (this entry synchronises the stream for such inputs) and the sort runs again as a launch chain
  hipStream_t st = (hipStream_t)stream;
  const bool maybe_one = n > 0 && rs_blocks_small((int)n) <= kRsOneLaunchBlocks;
  if (!maybe_one)
    return radix_sort_pairs(workspace_dev, workspace_bytes, keys_in_dev, keys_out_dev, vals_in_dev, vals_out_dev, (int)n, ps, np, st,
                            nullptr, false);
  const size_t need = radix_sort_temp_bytes((int)n);
  if (!workspace_dev || workspace_bytes < need + 256) {
    set_error("a3d_sort_pairs_u64: workspace too small (%zu < %zu)", workspace_bytes, need + 256);
    return A3D_ERR_WORKSPACE;
  }
  int* fail_dev = (int*)((char*)workspace_dev + need);
  A3D_HIP_CHECK(hipMemsetAsync(fail_dev, 0, sizeof(int), st));
  int rc = radix_sort_pairs(workspace_dev, need, keys_in_dev, keys_out_dev, vals_in_dev, vals_out_dev, (int)n, ps, np, st, fail_dev, true);
  if (rc) return rc;
  int fail = 0;
  A3D_HIP_CHECK(hipMemcpyAsync(&fail, fail_dev, sizeof(int), hipMemcpyDeviceToHost, st));
  A3D_HIP_CHECK(hipStreamSynchronize(st));
  if (fail == 0) return A3D_OK;
  return radix_sort_pairs(workspace_dev, need, keys_in_dev, keys_out_dev, vals_in_dev, vals_out_dev, (int)n, ps, np, st, nullptr, false);
}
