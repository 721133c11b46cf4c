// clicks.hip -- the interactive loop around forward_mask: label argmax, IoU counting and the click
// simulator (gfx950).
//
// Replaces (reference file:line):
//   p.argmax(-1) + "update prediction with sparse gt"      eval_multi_obj.py:119-134
//   mean_iou_scene / mean_iou_single                        utils/seg.py:10-18,44-59
//   get_simulated_clicks / measure_error_size / get_next_click_coo_torch   utils/seg.py:93-239
//   loss_weights                                            utils/seg.py:62-70
//
// The simulator's cost is measure_error_size: for every wrongly labelled point the distance to the
// nearest point that is NOT in its error cluster (the reference builds the full [other x cluster]
// torch.cdist matrix per cluster).  Here it is an exact brute-force pass for all clusters at once:
// candidates are wave-uniform, so they stream through the scalar cache as SGPR operands and the inner
// loop is 9 VALU ops per (query, candidate) with no LDS or vector-memory traffic.
// Only a cluster's MAXIMUM of these distances (and its first arg-max) is ever used, so the pass is bounded
// (round 3, exact): (A) the same kernel against every 16th point gives every wrong point an UPPER bound of its
// distance (a minimum over a subset), (B) the point with the largest upper bound of each cluster gets its exact
// distance -- a LOWER bound of the cluster's maximum --, (C) only points whose upper bound reaches their cluster's
// lower bound can be the arg-max (ties included) and go through the full pass.  Every pair distance is the same
// expression in all three phases, so the comparisons are exact in floating point.
// Round 4: the bounding runs TWICE, coarse to fine -- first against every 256th point (1/16 of phase A's pairs), whose
// survivors alone meet every 16th point; with large wrong regions (training with early weights, 300 k-voxel scenes) the
// first stage removes all but the few per cent of points near a cluster's deepest spot and phase A shrinks by ~10x
// (300 k voxels, 60 % wrong: 1.07 -> 0.79 ms; from kCoarseFrom points on -- at 80 k its launches cost what it saves).
// A cell-list search for the NEAR field in front of all this (points binned into a uniform grid, rings of cells scanned
// by 32 lanes per wrong point, only unresolved points to the brute-force search) was built, verified bit-identical and
// measured slower in every regime but one (profiles/r04_experiments.txt): removed.
#include "common.h"
#include <stdlib.h>

namespace a3d {

constexpr int kClusterTable = 1 << 15;      // cluster id = 96*label + 11*pred, label/pred <= 255
constexpr unsigned kInfBits = 0x7f800000u;
constexpr int kQueriesPerThread = 2;
constexpr int kNearestBlock = 256;
constexpr int kNearestSplit = 128;           // candidate chunks (grid.y): enough waves when only a few thousand points are wrong
constexpr int kSample = 16;                  // phase A: every kSample-th point is a candidate
constexpr int kSampleCoarse = 256;           // ... after a first bounding stage against every kSampleCoarse-th point (a subset of them)
constexpr int kCoarseFrom = 150000;          // ... from this many points (80 k voxels: its five launches cost what it saves)
constexpr int kMinChunk = 32;                // candidates per workgroup row of k_nearest_other at least (small samples: fewer, fuller blocks)
constexpr long long kSmallPairs = 1ll << 30;  // (wrong points) x (points) below which phase A is skipped: the plain pass is ~0.15 ms
constexpr int kMaxChamp = 1024;              // clusters that get a lower bound (phase B); further ones are not pruned
constexpr int kChampSplit = 8;

// ---- argmax over the 1+K mask logits of every point (first maximum wins, like torch.argmax) ------
__global__ void k_argmax_labels(const float* __restrict__ logits, int64_t n, int C, int32_t* __restrict__ pred) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = logits + i * C;
  float best = r[0];
  int arg = 0;
  for (int c = 1; c < C; ++c) {
    const float v = r[c];
    if (v > best) { best = v; arg = c; }
  }
  pred[i] = arg;
}

struct ClickList {
  int n;
  int row[A3D_MAX_CLICKS];
  unsigned char obj[A3D_MAX_CLICKS];
};
// sequential on purpose: a row clicked for two objects keeps the LAST one, as the reference's dict loop does
__global__ void k_click_overwrite(ClickList cl, int64_t n, int32_t* __restrict__ pred) {
  for (int i = 0; i < cl.n; ++i)
    if (cl.row[i] >= 0 && cl.row[i] < n) pred[cl.row[i]] = cl.obj[i];
}

// ---- IoU counts: counts[0][id] = |pred==id & label==id|, [1][id] = |pred==id|, [2][id] = |label==id| ----
__global__ void k_iou_counts(const int32_t* __restrict__ pred, const int64_t* __restrict__ inverse_map,
                             const int32_t* __restrict__ labels, int64_t n_full, int64_t n_pred, int n_ids,
                             unsigned long long* __restrict__ counts, int* __restrict__ err) {
  __shared__ unsigned h[3 * 256];
  for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_full; i += stride) {
    const int64_t src = inverse_map ? inverse_map[i] : i;
    if (src < 0 || src >= n_pred) { atomicOr(err, 1); continue; }
    const int p = pred[src], l = labels[i];
    if (p >= 0 && p < n_ids) atomicAdd(&h[256 + p], 1u);
    if (l >= 0 && l < n_ids) {
      atomicAdd(&h[512 + l], 1u);
      if (p == l) atomicAdd(&h[l], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) {
    const int id = i & 255;
    if (id < n_ids && h[i]) atomicAdd(&counts[(size_t)(i >> 8) * n_ids + id], (unsigned long long)h[i]);
  }
}

// ---- click simulator --------------------------------------------------------------------------
// cand[i] = (x, y, z, cluster id bits) for EVERY point (cluster -1 = correctly labelled);
// err_rows = rows of wrongly labelled points in arbitrary order (the result does not depend on it).
__global__ void k_err_compact(const float* __restrict__ xyz, const int32_t* __restrict__ pred,
                              const int32_t* __restrict__ labels, int64_t n, float4* __restrict__ cand,
                              int32_t* __restrict__ err_rows, unsigned* __restrict__ d2bits,
                              int* __restrict__ n_err, int* __restrict__ err, float4* __restrict__ samp, int stride,
                              float4* __restrict__ samp_coarse, int stride_coarse, int* __restrict__ max_cid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool wrong = false;
  int my_cid = -1;
  if (i < n) {
    const int p = pred[i], l = labels[i];
    const bool bad = p < 0 || p > 255 || l < 0 || l > 255;   // reported through err (the call fails); never used as a table index
    if (bad) atomicOr(err, 2);
    wrong = p != l && !bad;
    const int cid = wrong ? 96 * l + 11 * p : -1;
    my_cid = cid;
    const float4 c = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], __int_as_float(cid));
    cand[i] = c;
    if (samp && i % stride == 0) samp[i / stride] = c;
    if (samp_coarse && i % stride_coarse == 0) samp_coarse[i / stride_coarse] = c;
  }
  // one pair of atomics per WORKGROUP (the waves' counts and maxima meet in LDS): a sample's thousands of waves would queue
  // on the two counter words otherwise (58 us at 300 k points, an order of magnitude above the kernel's memory time)
  __shared__ int wcnt[4], wmax[4], wbase[4];
  const unsigned long long m = __ballot(wrong);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) my_cid = max(my_cid, __shfl_xor(my_cid, o));
  if (lane == 0) {
    wcnt[wv] = __popcll(m);
    wmax[wv] = my_cid;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    int b = 0;
    if (tot) {
      b = atomicAdd(n_err, tot);
      // the largest cluster id of the sample: k_cluster_list scans the table up to it (a handful of objects -> ids of a
      // few hundred, not 32 k)
      atomicMax(max_cid, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
    }
    wbase[0] = b;
    wbase[1] = b + wcnt[0];
    wbase[2] = b + wcnt[0] + wcnt[1];
    wbase[3] = b + wcnt[0] + wcnt[1] + wcnt[2];
  }
  __syncthreads();
  if (wrong) {
    const int slot = wbase[wv] + __popcll(m & ((1ull << lane) - 1));
    err_rows[slot] = (int)i;
    d2bits[slot] = kInfBits;
  }
}

// pts: all points (the queries' coordinates, by row); cand / n: the candidates of this pass (all points, or the sample)
__global__ __launch_bounds__(kNearestBlock) void k_nearest_other(const float4* __restrict__ pts,
                                                                 const float4* __restrict__ cand, int64_t n,
                                                                 const int32_t* __restrict__ err_rows,
                                                                 const int* __restrict__ n_err_p,
                                                                 unsigned* __restrict__ d2bits, int chunk,
                                                                 long long skip_below) {
  const int n_err = *n_err_p;
  const int q0 = blockIdx.x * (kNearestBlock * kQueriesPerThread);
  if (q0 >= n_err) return;
  // phase A of the bounded pass on a sample with few wrong points: not worth its time -- the upper bounds stay +inf,
  // every point survives and phase C is the plain pass
  if (skip_below && (long long)n_err * skip_below < kSmallPairs) return;   // skip_below = number of points (phase A only)
  float qx[kQueriesPerThread], qy[kQueriesPerThread], qz[kQueriesPerThread], best[kQueriesPerThread];
  int qc[kQueriesPerThread];
#pragma unroll
  for (int u = 0; u < kQueriesPerThread; ++u) {
    const int e = q0 + u * kNearestBlock + threadIdx.x;
    float4 c = make_float4(0.f, 0.f, 0.f, __int_as_float(-2));
    if (e < n_err) c = pts[err_rows[e]];
    qx[u] = c.x; qy[u] = c.y; qz[u] = c.z; qc[u] = __float_as_int(c.w);
    best[u] = __uint_as_float(kInfBits);
  }
  const int j0 = blockIdx.y * chunk;
  const int j1 = (int)min((int64_t)j0 + chunk, n);
  auto visit = [&](const float4 c) {
    const int cc = __float_as_int(c.w);
#pragma unroll
    for (int u = 0; u < kQueriesPerThread; ++u) {
      const float dx = qx[u] - c.x, dy = qy[u] - c.y, dz = qz[u] - c.z;
      const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      best[u] = fminf(best[u], cc != qc[u] ? d2 : __uint_as_float(kInfBits));
    }
  };
  int j = j0;                              // j is wave-uniform: cand[j..j+3] arrive as one s_load_dwordx16
  for (; j + 4 <= j1; j += 4) {
    const float4 c0 = cand[j], c1 = cand[j + 1], c2 = cand[j + 2], c3 = cand[j + 3];
    visit(c0); visit(c1); visit(c2); visit(c3);
  }
  for (; j < j1; ++j) visit(cand[j]);
#pragma unroll
  for (int u = 0; u < kQueriesPerThread; ++u) {
    const int e = q0 + u * kNearestBlock + threadIdx.x;
    if (e < n_err) atomicMin(&d2bits[e], __float_as_uint(best[u]));   // d2 >= 0: bit order == value order
  }
}

// largest "distance to the nearest outside point" per cluster; ties -> lowest row (torch.where(...)[0][0])
__global__ void k_cluster_best(const float4* __restrict__ cand, const int32_t* __restrict__ err_rows,
                               const int* __restrict__ n_err_p, const unsigned* __restrict__ d2bits,
                               unsigned long long* __restrict__ table, long long skip_below) {
  if (skip_below && (long long)*n_err_p * skip_below < kSmallPairs) return;   // phase B of a sample with few wrong points
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  int cid = -1;
  unsigned long long key = 0;
  if (e < *n_err_p) {
    const int row = err_rows[e];
    cid = __float_as_int(cand[row].w);
    key = ((unsigned long long)d2bits[e] << 32) | (0xffffffffu - (unsigned)row);
  }
  // one atomic per (wave, cluster) instead of one per point: a sample has a handful of clusters, so tens of thousands of
  // atomics queued on a few addresses (154 us at 90 k wrong points); the lanes of a wave mostly share one or two clusters
  unsigned long long todo = __ballot(cid >= 0);
  while (todo) {
    const int lead = __builtin_amdgcn_readlane(cid, __builtin_ctzll(todo));
    const bool mine = cid == lead;
    unsigned long long k = mine ? key : 0ull;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned long long other = __shfl_xor(k, o);
      k = other > k ? other : k;
    }
    const unsigned long long members = __ballot(mine);
    if ((threadIdx.x & 63) == __builtin_ctzll(members)) atomicMax(&table[lead], k);
    todo &= ~members;
  }
}

// ---- the bounded pass (see the head of the file) ---------------------------------------------------------
// phase B, step 1: the clusters' champions (largest upper bound, from k_cluster_best on the upper bounds) as a list;
// lbtab[cluster] starts at +inf for a listed cluster, stays 0 (= nothing is pruned) for one that did not fit
__global__ void k_champ_list(const unsigned long long* __restrict__ table_ub, int2* __restrict__ champ,
                             int* __restrict__ n_champ, unsigned* __restrict__ lbtab) {
  const int cid = blockIdx.x * blockDim.x + threadIdx.x;
  if (cid >= kClusterTable) return;
  const unsigned long long e = table_ub[cid];
  if (!e) return;
  const int slot = atomicAdd(n_champ, 1);
  if (slot >= kMaxChamp) return;
  champ[slot] = make_int2(cid, (int)(0xffffffffu - (unsigned)(e & 0xffffffffu)));
  lbtab[cid] = kInfBits;
}
// phase B, step 2: exact distance of every champion = lower bound of its cluster's maximum; block = (champion, candidate range)
__global__ void __launch_bounds__(256) k_champ_exact(const float4* __restrict__ cand, int64_t n, const int2* __restrict__ champ,
                                                     const int* __restrict__ n_champ, unsigned* __restrict__ lbtab) {
  const int nc = min(*n_champ, kMaxChamp);
  for (int c = blockIdx.x; c < nc; c += gridDim.x) {
  const int2 ch = champ[c];
  const float4 q = cand[ch.y];
  const int qc = __float_as_int(q.w);
  const int64_t per = (n + kChampSplit - 1) / kChampSplit;
  const int64_t j0 = (int64_t)blockIdx.y * per, j1 = min(j0 + per, n);
  float best = __uint_as_float(kInfBits);
  for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) {
    const float4 cc = cand[j];
    const float dx = q.x - cc.x, dy = q.y - cc.y, dz = q.z - cc.z;
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));   // the expression of k_nearest_other: bit-identical pair values
    best = fminf(best, __float_as_int(cc.w) != qc ? d2 : __uint_as_float(kInfBits));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) best = fminf(best, __shfl_xor(best, o));
  if ((threadIdx.x & 63) == 0) atomicMin(&lbtab[ch.x], __float_as_uint(best));
  }
}
// phase C, step 1: the points that can still be their cluster's arg-max
__global__ void k_survivors(const float4* __restrict__ cand, const int32_t* __restrict__ err_rows,
                            const int* __restrict__ n_err_p, const unsigned* __restrict__ ub2bits,
                            const unsigned* __restrict__ lbtab, int32_t* __restrict__ surv_rows,
                            unsigned* __restrict__ d2s, int* __restrict__ n_surv) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  bool keep = false;
  int row = 0;
  if (e < *n_err_p) {
    row = err_rows[e];
    keep = ub2bits[e] >= lbtab[__float_as_int(cand[row].w)];   // d2 >= 0: bit order == value order
  }
  const unsigned long long m = __ballot(keep);
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0 && m) base = atomicAdd(n_surv, __popcll(m));
  base = __shfl(base, 0);
  if (keep) {
    const int slot = base + __popcll(m & ((1ull << lane) - 1));
    surv_rows[slot] = row;
    d2s[slot] = kInfBits;
  }
}

// ordered compaction of the table (ascending cluster id, like torch.unique)
__global__ void k_cluster_list(const unsigned long long* __restrict__ table, const int32_t* __restrict__ pred,
                               const int32_t* __restrict__ labels, a3d_click_cluster* __restrict__ out,
                               int max_out, int32_t* __restrict__ n_out, const int* __restrict__ err,
                               const int* __restrict__ max_cid) {
  const int PER = min(kClusterTable / 1024, *max_cid / 1024 + 1);   // ids per thread: up to the largest one present
  __shared__ int sums[1024];
  const int t = threadIdx.x;
  int cnt = 0;
  for (int k = 0; k < PER; ++k) cnt += table[t * PER + k] != 0;
  sums[t] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = t >= off ? sums[t - off] : 0;
    __syncthreads();
    sums[t] += v;
    __syncthreads();
  }
  int pos = sums[t] - cnt;
  if (*err) {
    if (t == 1023) *n_out = -1;   // a label or prediction outside 0..255
    return;
  }
  for (int k = 0; k < PER; ++k) {
    const unsigned long long e = table[t * PER + k];
    if (!e) continue;
    if (pos < max_out) {
      const int row = (int)(0xffffffffu - (unsigned)(e & 0xffffffffu));
      a3d_click_cluster c;
      c.cluster_id = t * PER + k;
      c.row = row;
      c.label = labels[row];
      c.pred = pred[row];
      c.error_size = sqrtf(__uint_as_float((unsigned)(e >> 32)));
      out[pos] = c;
    }
    ++pos;
  }
  if (t == 1023) *n_out = sums[1023];
}

// ---- loss weights: alpha + (beta-alpha) * (1 - min(d, tita)/tita), d = distance to the nearest click ----
struct ClickRows {
  int n;
  int row[A3D_MAX_CLICKS];
};
__global__ void k_click_weights(const float* __restrict__ xyz, int64_t n, ClickRows cl, float tita, float alpha,
                                float beta, float* __restrict__ w) {
  __shared__ float cx[A3D_MAX_CLICKS], cy[A3D_MAX_CLICKS], cz[A3D_MAX_CLICKS];
  for (int i = threadIdx.x; i < cl.n; i += blockDim.x) {
    const int r = cl.row[i];
    cx[i] = xyz[3 * (int64_t)r]; cy[i] = xyz[3 * (int64_t)r + 1]; cz[i] = xyz[3 * (int64_t)r + 2];
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  float best = __uint_as_float(kInfBits);
  for (int c = 0; c < cl.n; ++c) {
    const float dx = x - cx[c], dy = y - cy[c], dz = z - cz[c];
    best = fminf(best, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
  }
  const float d = fminf(sqrtf(best), tita);
  w[i] = alpha + (beta - alpha) * (1.f - d / tita);
}

struct ClickWs {
  float4* cand;
  int32_t* err_rows;
  unsigned* d2bits;
  unsigned long long* table;
  int* n_err;
  int* err;
  // the bounded pass: table_ub / lbtab / counters sit right behind `table` (one memset clears them all)
  unsigned long long* table_ub;
  unsigned* lbtab;
  int *n_surv, *n_champ, *max_cid;
  int2* champ;
  float4* samp;
  int32_t* surv_rows;
  unsigned* d2s;
  // the coarse bounding stage in front of the fine one (its survivors: surv_c_rows / d2s_c; the fine stage's: surv_rows / d2s)
  unsigned long long* table_ub_c;
  unsigned* lbtab_c;
  int *n_surv_c, *n_champ_c;
  int2* champ_c;
  float4* samp_coarse;
  int32_t* surv_c_rows;
  unsigned* d2s_c;
  size_t zero_bytes;   // bytes from `table` on that start out as zeros
  size_t bytes;
};
static ClickWs carve_click(void* base, int64_t n) {
  ClickWs w;
  size_t off = 0;
  auto take = [&](size_t b) {
    void* p = base ? (char*)base + off : nullptr;
    off += align256(b);
    return p;
  };
  w.table = (unsigned long long*)take((size_t)kClusterTable * 8);
  w.n_err = (int*)take(256);
  w.err = w.n_err ? w.n_err + 1 : nullptr;
  w.n_surv = w.n_err ? w.n_err + 2 : nullptr;
  w.n_champ = w.n_err ? w.n_err + 3 : nullptr;
  w.max_cid = w.n_err ? w.n_err + 4 : nullptr;
  w.n_surv_c = w.n_err ? w.n_err + 5 : nullptr;
  w.n_champ_c = w.n_err ? w.n_err + 6 : nullptr;
  w.table_ub = (unsigned long long*)take((size_t)kClusterTable * 8);
  w.lbtab = (unsigned*)take((size_t)kClusterTable * 4);
  w.table_ub_c = (unsigned long long*)take((size_t)kClusterTable * 8);
  w.lbtab_c = (unsigned*)take((size_t)kClusterTable * 4);
  w.zero_bytes = off;
  w.champ = (int2*)take((size_t)kMaxChamp * 8);
  w.champ_c = (int2*)take((size_t)kMaxChamp * 8);
  w.cand = (float4*)take((size_t)n * 16);
  w.samp = (float4*)take((size_t)(n / 4 + 8) * 16);   // stride >= 4
  w.err_rows = (int32_t*)take((size_t)n * 4);
  w.d2bits = (unsigned*)take((size_t)n * 4);
  w.surv_rows = (int32_t*)take((size_t)n * 4);
  w.d2s = (unsigned*)take((size_t)n * 4);
  w.samp_coarse = (float4*)take((size_t)(n / kSampleCoarse + 8) * 16);
  w.surv_c_rows = (int32_t*)take((size_t)n * 4);
  w.d2s_c = (unsigned*)take((size_t)n * 4);
  w.bytes = off;
  return w;
}

}  // namespace a3d

using namespace a3d;

extern "C" int a3d_argmax_labels(const float* logits_dev, int64_t n, int n_classes, const int32_t* click_row,
                                 const int32_t* click_obj, int n_clicks, int32_t* pred_dev, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n < 0 || n_classes < 1 || n_clicks < 0 || n_clicks > A3D_MAX_CLICKS || !pred_dev || (n && !logits_dev)) {
    set_error("a3d_argmax_labels: bad arguments (n=%lld classes=%d clicks=%d)", (long long)n, n_classes, n_clicks);
    return A3D_ERR_INVALID;
  }
  if (n == 0) return A3D_OK;
  k_argmax_labels<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(logits_dev, n, n_classes, pred_dev);
  A3D_LAUNCH_CHECK();
  if (n_clicks) {
    ClickList cl;
    cl.n = n_clicks;
    for (int i = 0; i < n_clicks; ++i) {
      if (click_obj[i] < 0 || click_obj[i] > 255) {
        set_error("a3d_argmax_labels: click object %d out of range", click_obj[i]);
        return A3D_ERR_INVALID;
      }
      cl.row[i] = click_row[i];
      cl.obj[i] = (unsigned char)click_obj[i];
    }
    k_click_overwrite<<<1, 1, 0, st>>>(cl, n, pred_dev);
    A3D_LAUNCH_CHECK();
  }
  return A3D_OK;
}

extern "C" int a3d_iou_counts(const int32_t* pred_dev, int64_t n_pred, const int64_t* inverse_map_dev,
                              const int32_t* labels_dev, int64_t n_full, int n_ids, int64_t* counts_dev,
                              void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_full < 0 || n_pred < 0 || n_ids < 1 || n_ids > 256 || !counts_dev) {
    set_error("a3d_iou_counts: bad arguments (n_full=%lld n_ids=%d)", (long long)n_full, n_ids);
    return A3D_ERR_INVALID;
  }
  // counts_dev: [3][n_ids] int64 followed by one int32 error flag slot (caller passes 3*n_ids+1 int64)
  A3D_HIP_CHECK(hipMemsetAsync(counts_dev, 0, ((size_t)3 * n_ids + 1) * 8, st));
  if (n_full == 0) return A3D_OK;
  const int64_t want = (n_full + 1023) / 1024;
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
  k_iou_counts<<<grid, 256, 0, st>>>(pred_dev, inverse_map_dev, labels_dev, n_full, n_pred, n_ids,
                                     (unsigned long long*)counts_dev, (int*)(counts_dev + (size_t)3 * n_ids));
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" size_t a3d_click_workspace_bytes(int64_t n) {
  if (n <= 0 || n > (int64_t)1 << 30) return 0;
  return carve_click(nullptr, n).bytes;
}

extern "C" int a3d_click_clusters(const float* xyz_dev, const int32_t* pred_dev, const int32_t* labels_dev,
                                  int64_t n, a3d_click_cluster* out_dev, int max_out, int32_t* n_out_dev,
                                  void* workspace_dev, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0 || !xyz_dev || !pred_dev || !labels_dev || !out_dev || !n_out_dev || max_out < 1) {
    set_error("a3d_click_clusters: bad arguments");
    return A3D_ERR_INVALID;
  }
  ClickWs w = carve_click(workspace_dev, n);
  if (!workspace_dev || workspace_bytes < w.bytes) {
    set_error("a3d_click_clusters: workspace %zu < %zu", workspace_bytes, w.bytes);
    return A3D_ERR_WORKSPACE;
  }
  ProfScope prof(st, A3D_PROF_CLICKS);
  static int prune = -1;   // A3D_CLICK_PRUNE=0: the plain pass over all points (A/B)
  if (prune < 0) {
    const char* e = getenv("A3D_CLICK_PRUNE");
    prune = e ? atoi(e) : 1;
  }
  const int stride = kSample;   // strides 8 / 16 / 32 / 64 measured in round 3: 16 is the best or second best everywhere
  const bool bounded = prune && n >= 64 * stride;
  const bool two_stage = bounded && prune != 3 && n >= kCoarseFrom;   // A3D_CLICK_PRUNE=3: the one-stage search of round 3 (A/B)
  A3D_HIP_CHECK(hipMemsetAsync(w.table, 0, w.zero_bytes, st));   // tables + counters
  const unsigned nb = (unsigned)((n + 255) / 256);
  k_err_compact<<<nb, 256, 0, st>>>(xyz_dev, pred_dev, labels_dev, n, w.cand, w.err_rows, w.d2bits, w.n_err, w.err,
                                    bounded ? w.samp : nullptr, stride, two_stage ? w.samp_coarse : nullptr, kSampleCoarse,
                                    w.max_cid);
  A3D_LAUNCH_CHECK();
  const int per_block = kNearestBlock * kQueriesPerThread;
  auto nearest = [&](const float4* cands, int64_t n_cands, const int32_t* rows, const int* n_rows, unsigned* out,
                     long long skip_below) {
    int chunk = (int)((n_cands + kNearestSplit - 1) / kNearestSplit);
    chunk = chunk < kMinChunk ? kMinChunk : chunk;
    chunk = (chunk + 3) & ~3;
    dim3 grid((unsigned)((n + per_block - 1) / per_block), (unsigned)((n_cands + chunk - 1) / chunk));
    k_nearest_other<<<grid, kNearestBlock, 0, st>>>(w.cand, cands, n_cands, rows, n_rows, out, chunk, skip_below);
  };
  // one bounding stage: upper bounds of `rows` against a sample (A), one exact champion per cluster = lower bound of the
  // cluster's maximum (B), the rows that can still be their cluster's arg-max -> rows_out (C, step 1).  `skip`: the stage does
  // nothing (every row survives) when rows x skip < kSmallPairs -- decided on the device, the host never learns the count
  auto bounding_stage = [&](const float4* sample, int64_t n_sample, const int32_t* rows, const int* n_rows, unsigned* ub,
                            unsigned long long* table_ub, unsigned* lbtab, int2* champ, int* n_champ, int32_t* rows_out,
                            unsigned* d2_out, int* n_out, long long skip) {
    nearest(sample, n_sample, rows, n_rows, ub, skip);
    k_cluster_best<<<nb, 256, 0, st>>>(w.cand, rows, n_rows, ub, table_ub, skip);
    k_champ_list<<<kClusterTable / 256, 256, 0, st>>>(table_ub, champ, n_champ, lbtab);
    k_champ_exact<<<dim3(128, kChampSplit), 256, 0, st>>>(w.cand, n, champ, n_champ, lbtab);
    k_survivors<<<nb, 256, 0, st>>>(w.cand, rows, n_rows, ub, lbtab, rows_out, d2_out, n_out);
  };
  if (!bounded) {
    nearest(w.cand, n, w.err_rows, w.n_err, w.d2bits, 0);
    A3D_LAUNCH_CHECK();
    k_cluster_best<<<nb, 256, 0, st>>>(w.cand, w.err_rows, w.n_err, w.d2bits, w.table, 0);
    A3D_LAUNCH_CHECK();
  } else {
    const int64_t n_samp = (n + stride - 1) / stride;
    const int32_t* rows = w.err_rows;
    const int* n_rows = w.n_err;
    unsigned* ub = w.d2bits;
    if (two_stage) {
      // worth its five launches only with many wrong points (the fine stage's phase A is rows x n / 16 pairs): 2x the
      // threshold of the fine stage
      bounding_stage(w.samp_coarse, (n + kSampleCoarse - 1) / kSampleCoarse, rows, n_rows, ub, w.table_ub_c, w.lbtab_c, w.champ_c,
                     w.n_champ_c, w.surv_c_rows, w.d2s_c, w.n_surv_c, (long long)(n / 2));
      A3D_LAUNCH_CHECK();
      rows = w.surv_c_rows;
      n_rows = w.n_surv_c;
      ub = w.d2s_c;
    }
    // behind a coarse stage the rows are few and the launches are issued anyway: the fine stage runs from a quarter of the
    // pair count (what it saves is the final pass over ALL points for most of its rows)
    bounding_stage(w.samp, n_samp, rows, n_rows, ub, w.table_ub, w.lbtab, w.champ, w.n_champ, w.surv_rows, w.d2s, w.n_surv,
                   two_stage ? 4ll * n : (long long)n);
    A3D_LAUNCH_CHECK();
    nearest(w.cand, n, w.surv_rows, w.n_surv, w.d2s, 0);
    k_cluster_best<<<nb, 256, 0, st>>>(w.cand, w.surv_rows, w.n_surv, w.d2s, w.table, 0);
    A3D_LAUNCH_CHECK();
  }
  k_cluster_list<<<1, 1024, 0, st>>>(w.table, pred_dev, labels_dev, out_dev, max_out, n_out_dev, w.err, w.max_cid);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_click_loss_weights(const float* xyz_dev, int64_t n, const int32_t* click_row, int n_clicks,
                                      float tita, float alpha, float beta, float* weights_dev, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0 || n_clicks < 1 || n_clicks > A3D_MAX_CLICKS || !xyz_dev || !weights_dev || !(tita > 0.f)) {
    set_error("a3d_click_loss_weights: bad arguments (n=%lld clicks=%d)", (long long)n, n_clicks);
    return A3D_ERR_INVALID;
  }
  ClickRows cl;
  cl.n = n_clicks;
  for (int i = 0; i < n_clicks; ++i) {
    if (click_row[i] < 0 || click_row[i] >= n) {
      set_error("a3d_click_loss_weights: click row %d out of range", click_row[i]);
      return A3D_ERR_INVALID;
    }
    cl.row[i] = click_row[i];
  }
  k_click_weights<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(xyz_dev, n, cl, tita, alpha, beta, weights_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
