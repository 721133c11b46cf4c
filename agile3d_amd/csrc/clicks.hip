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
// Round 6: every kernel serves ALL samples of a call (device table, a3d_click_clusters_batch: ~17 launches a round
// whatever the batch size; they were ~12 per sample on side streams), which makes a bounding stage's launches cheap: the
// first stage runs from kCoarseFrom = 20 k points and its stage-skip rule from 2^26 pairs -- and, where the caller hands
// over a spatial order of the sample's points (a3d_click_spatial_order, once per scene), takes its upper bounds from a
// row's neighbours in that order instead of every 256th point (k_nearest_window: with predictions that are wrong nearly
// everywhere the sampled bounds pruned nothing; 4 x 80 k training samples: final pass 1.38 -> 0.87 ms per round).
// A cell-list search for the NEAR field in front of all this (points binned into a uniform grid, rings of cells scanned
// by 32 lanes per wrong point, only unresolved points to the brute-force search) was built, verified bit-identical and
// measured slower in every regime but one (profiles/r04_experiments.txt): removed.
#include "common.h"
#include <vector>
#include <stdlib.h>

namespace a3d {

constexpr int kClusterTable = 1 << 15;      // cluster id = 96*label + 11*pred, label/pred <= 255
constexpr unsigned kInfBits = 0x7f800000u;
constexpr int kQueriesPerThread = 2;
constexpr int kNearestBlock = 256;
constexpr int kNearestSplit = 128;           // candidate chunks (grid.y): enough waves when only a few thousand points are wrong
constexpr int kSample = 16;                  // phase A: every kSample-th point is a candidate
constexpr int kSampleCoarse = 256;           // ... after a first bounding stage against every kSampleCoarse-th point (a subset of them)
constexpr int kCoarseFrom = 20000;           // ... from this many points (150 k until round 6: the stage's five launches are shared by the samples of a call now)
constexpr int kMinChunk = 32;                // candidates per workgroup row of k_nearest_other at least (small samples: fewer, fuller blocks)
constexpr long long kSmallPairs = 1ll << 26;  // (wrong points) x (points) below which a bounding stage is skipped: the plain pass over them is ~20 us of the chip
                                              // (2^30 until round 6, when a stage cost five launches PER SAMPLE; 16 random-init samples then met the plain pass with
                                              //  14 k wrong points each: 4 ms a round)
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

// ---- the same two kernels for all samples of a round (blockIdx.y = sample; the per-sample pointers travel by value) ----
struct ArgmaxBatch {
  const float* logits[A3D_MAX_ROUND_SAMPLES];
  int32_t* pred[A3D_MAX_ROUND_SAMPLES];
  long long n[A3D_MAX_ROUND_SAMPLES];
  int C[A3D_MAX_ROUND_SAMPLES];
  int click_off[A3D_MAX_ROUND_SAMPLES + 1];   // sample i's clicks: entries click_off[i] .. click_off[i + 1] - 1 of the uploaded lists
};
__global__ void k_argmax_labels_b(const ArgmaxBatch b) {
  const int s = blockIdx.y;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n[s]) return;
  const int C = b.C[s];
  const float* r = b.logits[s] + i * C;
  float best = r[0];
  int arg = 0;
  for (int c = 1; c < C; ++c) {
    const float v = r[c];
    if (v > best) { best = v; arg = c; }
  }
  b.pred[s][i] = arg;
}
// one thread per sample, sequential like k_click_overwrite: a row clicked for two objects keeps the LAST one
__global__ void k_click_overwrite_b(const ArgmaxBatch b, const int32_t* __restrict__ rows, const int32_t* __restrict__ objs) {
  const int s = blockIdx.x;
  if (threadIdx.x) return;
  int32_t* pred = b.pred[s];
  const long long n = b.n[s];
  for (int i = b.click_off[s]; i < b.click_off[s + 1]; ++i)
    if (rows[i] >= 0 && rows[i] < n) pred[rows[i]] = objs[i];
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
// Every kernel below serves ALL samples of a call (a3d_click_clusters_batch; round 6): blockIdx.z (or .y) = sample, its
// pointers and sizes come from a device table (ClickDev), the grid is sized for the largest sample.  The click rounds of
// training (4 samples) and of a lock-step evaluation (16) were ~12 launches PER SAMPLE on side streams -- 50-200 launches a
// round whose issue time alone was most of the round's cluster search on a fitted model (a few hundred wrong points).
struct ClickWs {
  float4* cand;
  int32_t* err_rows;
  unsigned* d2bits;
  unsigned long long* table;
  int* n_err;
  int* err;
  // the bounded pass: table_ub / lbtab / counters sit right behind `table` (one clear covers them all)
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
  unsigned* lbs;       // per cluster: the largest EXACT distance among a sample of its rows (k_lb_sampled): a lower bound of its maximum
  int *n_surv_c, *n_champ_c;
  int2* champ_c;
  float4* samp_coarse;
  int32_t* surv_c_rows;
  unsigned* d2s_c;
  float4* sorted;      // cand in the sample's spatial order (k_sorted_cand)
  void* dev_table;     // the call's ClickDev table (first sample's workspace)
  size_t zero_bytes;   // bytes from `table` on that start out as zeros
  size_t bytes;
};
struct ClickDev {      // one sample of a call as the kernels see it
  const float* xyz;
  const int32_t *pred, *labels;
  int64_t n;
  a3d_click_cluster* out;
  int32_t* n_out;
  int max_out;
  int bounded, two_stage;   // which passes this sample runs (by its size)
  const int32_t *order, *inv;   // a spatial order of the sample's points (a3d_click_spatial_order) and its inverse, or nullptr
  ClickWs w;
};
constexpr int kMaxClickBatch = 64;
constexpr int kWindow = 64;                  // rows on either side of a row in the spatial order that its first upper bound looks at
enum { ST_PLAIN = 0, ST_COARSE = 1, ST_FINE = 2, ST_FINAL = 3, ST_LBS = 4 };
constexpr int kLbStep = 64;                  // every kLbStep-th wrong row gets its exact distance up front (k_lb_sampled)
constexpr int kLbMinRows = 4096;             // ... in samples with at least this many wrong rows
// what a pass reads and writes: PLAIN / FINAL = exact distances against all points into the cluster table; COARSE / FINE = a
// bounding stage (upper bounds against a sample of the points, champions, survivors)
struct StageView {
  bool on;
  const float4* cands;
  int64_t n_cands;
  const int32_t* rows;
  const int* n_rows;
  unsigned* d2;
  unsigned long long* table;
  unsigned* lbtab;
  int2* champ;
  int* n_champ;
  int32_t* rows_out;
  unsigned* d2_out;
  int* n_out;
  long long skip;     // the stage does nothing (every row survives) when rows x skip < kSmallPairs: decided on the device
};
template <int stage>
__device__ __forceinline__ StageView stage_view(const ClickDev& s) {
  const ClickWs& w = s.w;
  StageView v;
  v.on = false;
  v.lbtab = nullptr; v.champ = nullptr; v.n_champ = nullptr; v.rows_out = nullptr; v.d2_out = nullptr; v.n_out = nullptr;
  v.skip = 0;
  if constexpr (stage == ST_PLAIN) {
    v.on = !s.bounded;
    v.cands = w.cand; v.n_cands = s.n; v.rows = w.err_rows; v.n_rows = w.n_err; v.d2 = w.d2bits; v.table = w.table;
  } else if constexpr (stage == ST_COARSE) {
    v.on = s.two_stage != 0;
    // (with a spatial order the stage's upper bounds come from k_nearest_window instead of the coarse sample; n_cands = the
    // pairs a row costs there, for the skip rule)
    v.cands = w.samp_coarse; v.n_cands = s.order ? 2 * kWindow : (s.n + kSampleCoarse - 1) / kSampleCoarse;
    v.rows = w.err_rows; v.n_rows = w.n_err; v.d2 = w.d2bits; v.table = w.table_ub_c; v.lbtab = w.lbtab_c;
    v.champ = w.champ_c; v.n_champ = w.n_champ_c; v.rows_out = w.surv_c_rows; v.d2_out = w.d2s_c; v.n_out = w.n_surv_c;
    v.skip = s.order ? 0 : (long long)(s.n / 2);
  } else if constexpr (stage == ST_LBS) {
    // exact distances of every kLbStep-th wrong row against ALL points, into the rows' own upper-bound slots
    v.on = s.two_stage != 0;
    v.cands = w.cand; v.n_cands = s.n; v.rows = w.err_rows; v.n_rows = w.n_err; v.d2 = w.d2bits; v.table = nullptr;
  } else if constexpr (stage == ST_FINE) {
    v.on = s.bounded != 0;
    v.cands = w.samp; v.n_cands = (s.n + kSample - 1) / kSample;
    v.rows = s.two_stage ? w.surv_c_rows : w.err_rows; v.n_rows = s.two_stage ? w.n_surv_c : w.n_err;
    v.d2 = s.two_stage ? w.d2s_c : w.d2bits; v.table = w.table_ub; v.lbtab = w.lbtab;
    v.champ = w.champ; v.n_champ = w.n_champ; v.rows_out = w.surv_rows; v.d2_out = w.d2s; v.n_out = w.n_surv;
    // behind a coarse stage the rows are few and the launches are issued anyway: the fine stage runs from a quarter of the
    // pair count (what it saves is the final pass over ALL points for most of its rows)
    v.skip = s.two_stage ? 4ll * s.n : (long long)s.n;
  } else {
    v.on = s.bounded != 0;
    v.cands = w.cand; v.n_cands = s.n; v.rows = w.surv_rows; v.n_rows = w.n_surv; v.d2 = w.d2s; v.table = w.table;
  }
  return v;
}
// tables + counters of every sample back to zero (one launch instead of a memset per sample)
__global__ void k_click_clear(const ClickDev* __restrict__ tab) {
  const ClickDev& s = tab[blockIdx.y];
  uint4* p = (uint4*)s.w.table;
  const size_t n16 = s.w.zero_bytes / 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4(0u, 0u, 0u, 0u);
}
// cand[i] = (x, y, z, cluster id bits) for EVERY point (cluster -1 = correctly labelled);
// err_rows = rows of wrongly labelled points in arbitrary order (the result does not depend on it).
__global__ void k_err_compact(const ClickDev* __restrict__ tab) {
  const ClickDev& s = tab[blockIdx.y];
  const float* __restrict__ xyz = s.xyz;
  const int64_t n = s.n;
  float4* __restrict__ cand = s.w.cand;
  float4* samp = s.bounded ? s.w.samp : nullptr;
  float4* samp_coarse = s.two_stage ? s.w.samp_coarse : nullptr;
  const int stride = kSample, stride_coarse = kSampleCoarse;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((int64_t)blockIdx.x * blockDim.x >= n) return;   // a block beyond this sample (the grid serves the largest one)
  bool wrong = false;
  int my_cid = -1;
  if (i < n) {
    const int p = s.pred[i], l = s.labels[i];
    const bool bad = p < 0 || p > 255 || l < 0 || l > 255;   // reported through err (the call fails); never used as a table index
    if (bad) atomicOr(s.w.err, 2);
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
      b = atomicAdd(s.w.n_err, tot);
      // the largest cluster id of the sample: k_cluster_list scans the table up to it (a handful of objects -> ids of a
      // few hundred, not 32 k)
      atomicMax(s.w.max_cid, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
    }
    wbase[0] = b;
    wbase[1] = b + wcnt[0];
    wbase[2] = b + wcnt[0] + wcnt[1];
    wbase[3] = b + wcnt[0] + wcnt[1] + wcnt[2];
  }
  __syncthreads();
  if (wrong) {
    const int slot = wbase[wv] + __popcll(m & ((1ull << lane) - 1));
    s.w.err_rows[slot] = (int)i;
    s.w.d2bits[slot] = kInfBits;
  }
}

// candidates of a pass in chunks (grid.y): at least kMinChunk per workgroup row, a multiple of four
__device__ __host__ inline int nearest_chunk(int64_t n_cands) {
  int chunk = (int)((n_cands + kNearestSplit - 1) / kNearestSplit);
  chunk = chunk < kMinChunk ? kMinChunk : chunk;
  return (chunk + 3) & ~3;
}
// the queries' coordinates come from cand (all points, by row); the candidates of the pass: all points, or a sample of them
template <int stage>
__global__ __launch_bounds__(kNearestBlock) void k_nearest_other(const ClickDev* __restrict__ tab) {
  const ClickDev& s = tab[blockIdx.z];
  const StageView v = stage_view<stage>(s);
  if (!v.on) return;
  if (stage == ST_COARSE && s.order) return;   // this sample's first stage is k_nearest_window
  const float4* __restrict__ pts = s.w.cand;
  const float4* __restrict__ cand = v.cands;
  const int64_t n = v.n_cands;
  constexpr int STEP = stage == ST_LBS ? kLbStep : 1;                 // rows of this pass: every STEP-th of the stage's list
  const int n_all = *v.n_rows;
  if (stage == ST_LBS && n_all < kLbMinRows) return;
  const int n_err = (n_all + STEP - 1) / STEP;
  // phase A of the bounded pass on a sample with few wrong points: not worth its time -- the upper bounds stay +inf,
  // every point survives and phase C is the plain pass
  if (v.skip && (long long)n_err * v.skip < kSmallPairs) return;
  const int chunk = nearest_chunk(n);
  const int64_t j0l = (int64_t)blockIdx.y * chunk;
  if (j0l >= n) return;
  const int q0 = blockIdx.x * (kNearestBlock * kQueriesPerThread);
  if (q0 >= n_err) return;   // (the host does not know how many rows are wrong: the grid covers all of them.  A few workgroups WALKING
                             //  the query blocks instead measured slower in every regime, profiles/r06_experiments.txt)
  float qx[kQueriesPerThread], qy[kQueriesPerThread], qz[kQueriesPerThread], best[kQueriesPerThread];
  int qc[kQueriesPerThread];
#pragma unroll
  for (int u = 0; u < kQueriesPerThread; ++u) {
    const int e = q0 + u * kNearestBlock + threadIdx.x;
    float4 c = make_float4(0.f, 0.f, 0.f, __int_as_float(-2));
    if (e < n_err) c = pts[v.rows[e * STEP]];
    qx[u] = c.x; qy[u] = c.y; qz[u] = c.z; qc[u] = __float_as_int(c.w);
    best[u] = __uint_as_float(kInfBits);
  }
  const int j0 = (int)j0l;
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
    if (e < n_err) atomicMin(&v.d2[e * STEP], __float_as_uint(best[u]));   // d2 >= 0: bit order == value order
  }
}

// ---- upper bounds from a spatial order (round 6) ----------------------------------------------------------------------
// With predictions that are wrong nearly everywhere (early training, random weights) every point has a point of another
// cluster a voxel or two away, but an upper bound from every 16th / 256th point cannot show it (the sample's spacing is
// four voxels and more): nearly every wrong point survived the bounding stages and met ALL points in the final pass -- 1.4 ms
// per training click round, the largest kernel of the iteration.  A point's neighbours in a Morton order of the sample's
// coordinates (computed once per scene, a3d_click_spatial_order) are mostly its neighbours in space: the nearest point of
// another cluster among the 2 x kWindow rows around it is a far tighter bound for ~128 pairs per row.  Any permutation
// gives VALID bounds (a minimum over a subset of the outside points, the same pair expression): a stale or poor order
// costs time, never exactness.
__global__ void k_sorted_cand(const ClickDev* __restrict__ tab) {
  const ClickDev& s = tab[blockIdx.y];
  if (!s.order || !s.two_stage) return;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < s.n) s.w.sorted[i] = s.w.cand[s.order[i]];
}
__global__ void __launch_bounds__(256) k_nearest_window(const ClickDev* __restrict__ tab) {
  const ClickDev& s = tab[blockIdx.y];
  if (!s.order || !s.two_stage) return;
  const StageView v = stage_view<ST_COARSE>(s);
  const int n_err = *v.n_rows;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if ((int)(blockIdx.x * blockDim.x) >= n_err) return;
  if (e >= n_err) return;
  const float4* __restrict__ sorted = s.w.sorted;
  const int row = v.rows[e];
  const float4 q = s.w.cand[row];
  const int qc = __float_as_int(q.w);
  const int pos = s.inv[row];
  const int j0 = max(0, pos - kWindow), j1 = min((int)s.n, pos + kWindow + 1);
  float best = __uint_as_float(kInfBits);
  for (int j = j0; j < j1; ++j) {
    const float4 c = sorted[j];
    const float dx = q.x - c.x, dy = q.y - c.y, dz = q.z - c.z;
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));   // the expression of k_nearest_other: bit-identical pair values
    best = fminf(best, __float_as_int(c.w) != qc ? d2 : __uint_as_float(kInfBits));
  }
  v.d2[e] = __float_as_uint(best);
}

// largest "distance to the nearest outside point" per cluster; ties -> lowest row (torch.where(...)[0][0])
template <int stage>
__global__ void k_cluster_best(const ClickDev* __restrict__ tab) {
  const ClickDev& s = tab[blockIdx.y];
  const StageView v = stage_view<stage>(s);
  if (!v.on) return;
  const int n_err = *v.n_rows;
  if ((int)(blockIdx.x * blockDim.x) >= n_err) return;
  if (v.skip && (long long)n_err * v.skip < kSmallPairs) return;   // phase B of a sample with few wrong points
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  int cid = -1;
  unsigned long long key = 0;
  if (e < n_err) {
    const int row = v.rows[e];
    cid = __float_as_int(s.w.cand[row].w);
    key = ((unsigned long long)v.d2[e] << 32) | (0xffffffffu - (unsigned)row);
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
    if ((threadIdx.x & 63) == __builtin_ctzll(members)) atomicMax(&v.table[lead], k);
    todo &= ~members;
  }
}

// ---- the bounded pass (see the head of the file) ---------------------------------------------------------
// phase B, step 1: the clusters' champions (largest upper bound, from k_cluster_best on the upper bounds) as a list;
// lbtab[cluster] starts at +inf for a listed cluster, stays 0 (= nothing is pruned) for one that did not fit
template <int stage>
__global__ void k_champ_list(const ClickDev* __restrict__ tab) {
  const ClickDev& s = tab[blockIdx.y];
  const StageView v = stage_view<stage>(s);
  if (!v.on) return;
  const int cid = blockIdx.x * blockDim.x + threadIdx.x;
  if (cid >= kClusterTable || cid > *s.w.max_cid) return;
  const unsigned long long e = v.table[cid];
  if (!e) return;
  const int slot = atomicAdd(v.n_champ, 1);
  if (slot >= kMaxChamp) return;
  v.champ[slot] = make_int2(cid, (int)(0xffffffffu - (unsigned)(e & 0xffffffffu)));
  v.lbtab[cid] = kInfBits;
}
// phase B, step 2: exact distance of every champion = lower bound of its cluster's maximum; block = (champion, candidate range)
template <int stage>
__global__ void __launch_bounds__(256) k_champ_exact(const ClickDev* __restrict__ tab) {
  const ClickDev& s = tab[blockIdx.z];
  const StageView v = stage_view<stage>(s);
  if (!v.on) return;
  const float4* __restrict__ cand = s.w.cand;
  const int64_t n = s.n;
  const int nc = min(*v.n_champ, kMaxChamp);
  for (int c = blockIdx.x; c < nc; c += gridDim.x) {
  const int2 ch = v.champ[c];
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
  if ((threadIdx.x & 63) == 0) atomicMin(&v.lbtab[ch.x], __float_as_uint(best));
  }
}
// Lower bounds from a SAMPLE of rows (round 6).  The champion of a cluster -- its row with the largest upper bound -- tends to
// be a row whose bound is loose, so its exact distance sits far below the cluster's maximum and most rows survive it (a
// 58 k-point cluster of an early-training prediction kept 36 k of them; offline on such a scene: champion 0.027 against a
// maximum of 0.051, 6 214 survivors -- 368 with the bound below).  Every kLbStep-th wrong row gets its EXACT distance up
// front (k_nearest_other<ST_LBS>: ~n_err / 64 rows against all points, into the rows' own bound slots, which it also
// tightens) and the largest of them per cluster is a lower bound of that cluster's maximum like the champion's.
__global__ void k_lb_sampled(const ClickDev* __restrict__ tab) {
  const ClickDev& s = tab[blockIdx.y];
  if (!s.two_stage) return;
  const int n_all = *s.w.n_err;
  if (n_all < kLbMinRows) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = i * kLbStep;
  if (e >= n_all) return;
  const int cid = __float_as_int(s.w.cand[s.w.err_rows[e]].w);
  atomicMax(&s.w.lbs[cid], s.w.d2bits[e]);                            // d2 >= 0: bit order == value order
}
// phase C, step 1: the points that can still be their cluster's arg-max
template <int stage>
__global__ void k_survivors(const ClickDev* __restrict__ tab) {
  const ClickDev& s = tab[blockIdx.y];
  const StageView v = stage_view<stage>(s);
  if (!v.on) return;
  const int n_err = *v.n_rows;
  if ((int)(blockIdx.x * blockDim.x) >= n_err) return;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  bool keep = false;
  int row = 0;
  if (e < n_err) {
    row = v.rows[e];
    const int cid = __float_as_int(s.w.cand[row].w);
    keep = v.d2[e] >= max(v.lbtab[cid], s.w.lbs[cid]);              // d2 >= 0: bit order == value order
  }
  const unsigned long long m = __ballot(keep);
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0 && m) base = atomicAdd(v.n_out, __popcll(m));
  base = __shfl(base, 0);
  if (keep) {
    const int slot = base + __popcll(m & ((1ull << lane) - 1));
    v.rows_out[slot] = row;
    v.d2_out[slot] = kInfBits;
  }
}

// ordered compaction of the table (ascending cluster id, like torch.unique)
__global__ void k_cluster_list(const ClickDev* __restrict__ tab) {
  const ClickDev& s = tab[blockIdx.y];
  const unsigned long long* __restrict__ table = s.w.table;
  a3d_click_cluster* __restrict__ out = s.out;
  const int max_out = s.max_out;
  const int PER = min(kClusterTable / 1024, *s.w.max_cid / 1024 + 1);   // ids per thread: up to the largest one present
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
  if (*s.w.err) {
    if (t == 1023) *s.n_out = -1;   // a label or prediction outside 0..255
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
      c.label = s.labels[row];
      c.pred = s.pred[row];
      c.error_size = sqrtf(__uint_as_float((unsigned)(e >> 32)));
      out[pos] = c;
    }
    ++pos;
  }
  if (t == 1023) *s.n_out = sums[1023];
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
  w.lbs = (unsigned*)take((size_t)kClusterTable * 4);
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
  w.sorted = (float4*)take((size_t)n * 16);
  w.dev_table = take((size_t)kMaxClickBatch * sizeof(ClickDev));
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

struct IouBatch {
  const int32_t* pred[A3D_MAX_ROUND_SAMPLES];
  const int64_t* inv[A3D_MAX_ROUND_SAMPLES];
  const int32_t* lab[A3D_MAX_ROUND_SAMPLES];
  long long n_full[A3D_MAX_ROUND_SAMPLES];
  long long n_pred[A3D_MAX_ROUND_SAMPLES];
};
// k_iou_counts for sample blockIdx.y (the same per-workgroup histogram, the same atomics)
__global__ void k_iou_counts_b(const IouBatch b, int n_ids, unsigned long long* __restrict__ counts_all) {
  __shared__ unsigned h[3 * 256];
  const int s = blockIdx.y;
  const int32_t* __restrict__ pred = b.pred[s];
  const int64_t* __restrict__ inverse_map = b.inv[s];
  const int32_t* __restrict__ labels = b.lab[s];
  const long long n_full = b.n_full[s], n_pred = b.n_pred[s];
  unsigned long long* counts = counts_all + (size_t)s * (3 * n_ids + 1);
  int* err = (int*)(counts + (size_t)3 * n_ids);
  for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_full; i += stride) {
    const long long src = inverse_map ? inverse_map[i] : i;
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

extern "C" int a3d_iou_counts_batch(const a3d_iou_sample* samples, int n_samples, int n_ids, int64_t* counts_all_dev, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!samples || n_samples < 1 || n_samples > A3D_MAX_ROUND_SAMPLES || n_ids < 1 || n_ids > 256 || !counts_all_dev) {
    set_error("a3d_iou_counts_batch: bad arguments (1..%d samples, 1..256 ids)", A3D_MAX_ROUND_SAMPLES);
    return A3D_ERR_INVALID;
  }
  IouBatch b;
  memset(&b, 0, sizeof(b));
  long long n_max = 0;
  for (int i = 0; i < n_samples; ++i) {
    const a3d_iou_sample& sp = samples[i];
    if (sp.n_full < 0 || sp.n_pred < 0 || (sp.n_full > 0 && (!sp.pred_dev || !sp.labels_dev))) {
      set_error("a3d_iou_counts_batch: sample %d: bad arguments (n_full=%lld)", i, (long long)sp.n_full);
      return A3D_ERR_INVALID;
    }
    b.pred[i] = sp.pred_dev, b.inv[i] = sp.inverse_map_dev, b.lab[i] = sp.labels_dev;
    b.n_full[i] = sp.n_full, b.n_pred[i] = sp.n_pred;
    n_max = sp.n_full > n_max ? sp.n_full : n_max;
  }
  A3D_HIP_CHECK(hipMemsetAsync(counts_all_dev, 0, (size_t)n_samples * ((size_t)3 * n_ids + 1) * 8, st));
  if (n_max == 0) return A3D_OK;
  const long long want = (n_max + 1023) / 1024;
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
  k_iou_counts_b<<<dim3(grid, n_samples), 256, 0, st>>>(b, n_ids, (unsigned long long*)counts_all_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" size_t a3d_argmax_labels_batch_workspace_bytes(int n_samples) {
  if (n_samples < 1 || n_samples > A3D_MAX_ROUND_SAMPLES) return 0;
  return (size_t)n_samples * A3D_MAX_CLICKS * 8 + 256;
}
extern "C" int a3d_argmax_labels_batch(const a3d_argmax_sample* samples, int n_samples, void* workspace_dev, size_t workspace_bytes,
                                       void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!samples || n_samples < 1 || n_samples > A3D_MAX_ROUND_SAMPLES) {
    set_error("a3d_argmax_labels_batch: 1..%d samples per call", A3D_MAX_ROUND_SAMPLES);
    return A3D_ERR_INVALID;
  }
  ArgmaxBatch b;
  memset(&b, 0, sizeof(b));
  long long n_max = 0;
  int total = 0;
  for (int i = 0; i < n_samples; ++i) {
    const a3d_argmax_sample& sp = samples[i];
    if (sp.n < 0 || sp.n_classes < 1 || sp.n_clicks < 0 || sp.n_clicks > A3D_MAX_CLICKS || (sp.n && (!sp.logits_dev || !sp.pred_dev)) ||
        (sp.n_clicks && (!sp.click_row || !sp.click_obj))) {
      set_error("a3d_argmax_labels_batch: sample %d: bad arguments (n=%lld classes=%d clicks=%d)", i, (long long)sp.n, sp.n_classes,
                sp.n_clicks);
      return A3D_ERR_INVALID;
    }
    b.logits[i] = sp.logits_dev, b.pred[i] = sp.pred_dev, b.n[i] = sp.n, b.C[i] = sp.n_classes;
    b.click_off[i] = total;
    total += sp.n ? sp.n_clicks : 0;
    n_max = sp.n > n_max ? sp.n : n_max;
  }
  b.click_off[n_samples] = total;
  if (n_max == 0) return A3D_OK;
  k_argmax_labels_b<<<dim3((unsigned)((n_max + 255) / 256), n_samples), 256, 0, st>>>(b);
  A3D_LAUNCH_CHECK();
  if (total) {
    if (!workspace_dev || ((uintptr_t)workspace_dev & 15) || workspace_bytes < (size_t)total * 8) {
      set_error("a3d_argmax_labels_batch: workspace too small or misaligned (a3d_argmax_labels_batch_workspace_bytes)");
      return A3D_ERR_WORKSPACE;
    }
    int32_t host[2 * A3D_MAX_ROUND_SAMPLES * A3D_MAX_CLICKS / 8];     // this call's lists, rows then objects (most rounds: a few dozen clicks)
    std::vector<int32_t> big;
    int32_t* hp = host;
    if ((size_t)2 * total > sizeof(host) / sizeof(host[0])) {
      big.resize((size_t)2 * total);
      hp = big.data();
    }
    int at = 0;
    for (int i = 0; i < n_samples; ++i) {
      const a3d_argmax_sample& sp = samples[i];
      if (!sp.n) continue;
      for (int c = 0; c < sp.n_clicks; ++c, ++at) {
        const int o = sp.click_obj[c];
        if (o < 0 || o > 255) {
          set_error("a3d_argmax_labels_batch: sample %d: object id %d outside 0..255", i, o);
          return A3D_ERR_INVALID;
        }
        hp[at] = sp.click_row[c];
        hp[total + at] = o;
      }
    }
    int32_t* rows_dev = (int32_t*)workspace_dev;
    // (a pageable source: the runtime stages the bytes before the call returns, `host` may go out of scope)
    A3D_HIP_CHECK(hipMemcpyAsync(rows_dev, hp, (size_t)2 * total * sizeof(int32_t), hipMemcpyHostToDevice, st));
    k_click_overwrite_b<<<n_samples, 64, 0, st>>>(b, rows_dev, rows_dev + total);
    A3D_LAUNCH_CHECK();
  }
  return A3D_OK;
}

extern "C" size_t a3d_click_workspace_bytes(int64_t n) {
  if (n <= 0 || n > (int64_t)1 << 30) return 0;
  return carve_click(nullptr, n).bytes;
}

extern "C" int a3d_click_clusters_batch(const a3d_click_sample* samples, int n_samples, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!samples || n_samples < 1 || n_samples > kMaxClickBatch) {
    set_error("a3d_click_clusters_batch: 1..%d samples per call", kMaxClickBatch);
    return A3D_ERR_INVALID;
  }
  static int prune = -1;   // A3D_CLICK_PRUNE=0: the plain pass over all points (A/B)
  if (prune < 0) {
    const char* e = getenv("A3D_CLICK_PRUNE");
    prune = e ? atoi(e) : 1;
  }
  static int coarse_from = -1;   // the coarse bounding stage in front of the fine one from this many points on
  if (coarse_from < 0) {
    const char* e = getenv("A3D_CLICK_COARSE_FROM");
    coarse_from = e ? atoi(e) : kCoarseFrom;
  }
  ClickDev host[kMaxClickBatch];
  int64_t n_max = 0;
  bool any_plain = false, any_bounded = false, any_two = false, any_order = false, any_sampled = false;
  static int use_order = -1;   // A3D_CLICK_ORDER=0: ignore the samples' spatial orders (A/B, tests)
  if (use_order < 0) {
    const char* e = getenv("A3D_CLICK_ORDER");
    use_order = e ? atoi(e) : 1;
  }
  for (int i = 0; i < n_samples; ++i) {
    const a3d_click_sample& sp = samples[i];
    if (sp.n <= 0 || !sp.xyz_dev || !sp.pred_dev || !sp.labels_dev || !sp.out_dev || !sp.n_out_dev || sp.max_out < 1) {
      set_error("a3d_click_clusters: bad arguments (sample %d)", i);
      return A3D_ERR_INVALID;
    }
    ClickDev& d = host[i];
    d.w = carve_click(sp.workspace_dev, sp.n);
    if (!sp.workspace_dev || sp.workspace_bytes < d.w.bytes || ((uintptr_t)sp.workspace_dev & 15)) {
      set_error("a3d_click_clusters: workspace %zu < %zu (sample %d)", sp.workspace_bytes, d.w.bytes, i);
      return A3D_ERR_WORKSPACE;
    }
    d.xyz = sp.xyz_dev;
    d.pred = sp.pred_dev;
    d.labels = sp.labels_dev;
    d.n = sp.n;
    d.out = sp.out_dev;
    d.n_out = sp.n_out_dev;
    d.max_out = sp.max_out;
    // strides 8 / 16 / 32 / 64 measured in round 3: 16 is the best or second best everywhere
    d.bounded = prune && sp.n >= 64 * kSample;
    d.two_stage = d.bounded && prune != 3 && sp.n >= coarse_from;   // A3D_CLICK_PRUNE=3: the one-stage search of round 3 (A/B)
    d.order = use_order ? sp.order_dev : nullptr;
    d.inv = use_order ? sp.inv_dev : nullptr;
    if ((d.order == nullptr) != (d.inv == nullptr)) {
      set_error("a3d_click_clusters: order_dev and inv_dev come together (sample %d)", i);
      return A3D_ERR_INVALID;
    }
    any_order = any_order || (d.order && d.two_stage);
    any_sampled = any_sampled || (!d.order && d.two_stage);
    any_plain = any_plain || !d.bounded;
    any_bounded = any_bounded || d.bounded;
    any_two = any_two || d.two_stage;
    n_max = sp.n > n_max ? sp.n : n_max;
  }
  ProfScope prof(st, A3D_PROF_CLICKS);
  const ClickDev* tab = (const ClickDev*)host[0].w.dev_table;
  // pageable source: the runtime stages it before returning
  A3D_HIP_CHECK(hipMemcpyAsync((void*)tab, host, sizeof(ClickDev) * n_samples, hipMemcpyHostToDevice, st));
  const unsigned ns = (unsigned)n_samples;
  const unsigned nb = (unsigned)((n_max + 255) / 256);
  k_click_clear<<<dim3(64, ns), 256, 0, st>>>(tab);
  k_err_compact<<<dim3(nb, ns), 256, 0, st>>>(tab);
  A3D_LAUNCH_CHECK();
  const int per_block = kNearestBlock * kQueriesPerThread;
  auto nearest_grid = [&](int stage) {
    unsigned gy = 1;   // candidate chunks: every sample cuts ITS candidates (nearest_chunk), the grid holds the most any needs
    for (int i = 0; i < n_samples; ++i) {
      const int64_t nc = stage == ST_COARSE ? (host[i].n + kSampleCoarse - 1) / kSampleCoarse
                         : stage == ST_FINE ? (host[i].n + kSample - 1) / kSample : host[i].n;
      const unsigned c = (unsigned)((nc + nearest_chunk(nc) - 1) / nearest_chunk(nc));
      gy = c > gy ? c : gy;
    }
    return dim3((unsigned)((n_max + per_block - 1) / per_block), gy, ns);
  };
  // (the pass is a template parameter of the kernels: what it reads and writes is fixed at compile time)
  if (any_plain) {
    k_nearest_other<ST_PLAIN><<<nearest_grid(ST_PLAIN), kNearestBlock, 0, st>>>(tab);
    k_cluster_best<ST_PLAIN><<<dim3(nb, ns), 256, 0, st>>>(tab);
    A3D_LAUNCH_CHECK();
  }
  if (any_bounded) {
    // one bounding stage: upper bounds of the stage's rows against a sample of the points (A), one exact champion per
    // cluster = lower bound of the cluster's maximum (B), the rows that can still be their cluster's arg-max (C, step 1).
    // The coarse stage is worth its five launches only with many wrong points (the fine stage's phase A is rows x n / 16
    // pairs): its threshold is 2x the fine stage's (stage_view)
    if (any_two) {
      if (any_order) {   // the first stage's upper bounds from the window around the row in the sample's spatial order
        k_sorted_cand<<<dim3(nb, ns), 256, 0, st>>>(tab);
        k_nearest_window<<<dim3(nb, ns), 256, 0, st>>>(tab);
      }
      if (any_sampled) k_nearest_other<ST_COARSE><<<nearest_grid(ST_COARSE), kNearestBlock, 0, st>>>(tab);
      {   // exact distances of a sample of the wrong rows -> lower bounds of the clusters' maxima (behind the bounds above:
          // k_nearest_window stores, this pass lowers)
        dim3 gl = nearest_grid(ST_LBS);
        gl.x = (gl.x + kLbStep - 1) / kLbStep;
        k_nearest_other<ST_LBS><<<gl, kNearestBlock, 0, st>>>(tab);
        k_lb_sampled<<<dim3((unsigned)((n_max / kLbStep + 255) / 256 + 1), ns), 256, 0, st>>>(tab);
      }
      k_cluster_best<ST_COARSE><<<dim3(nb, ns), 256, 0, st>>>(tab);
      k_champ_list<ST_COARSE><<<dim3(kClusterTable / 256, ns), 256, 0, st>>>(tab);
      k_champ_exact<ST_COARSE><<<dim3(128, kChampSplit, ns), 256, 0, st>>>(tab);
      k_survivors<ST_COARSE><<<dim3(nb, ns), 256, 0, st>>>(tab);
    }
    k_nearest_other<ST_FINE><<<nearest_grid(ST_FINE), kNearestBlock, 0, st>>>(tab);
    k_cluster_best<ST_FINE><<<dim3(nb, ns), 256, 0, st>>>(tab);
    k_champ_list<ST_FINE><<<dim3(kClusterTable / 256, ns), 256, 0, st>>>(tab);
    k_champ_exact<ST_FINE><<<dim3(128, kChampSplit, ns), 256, 0, st>>>(tab);
    k_survivors<ST_FINE><<<dim3(nb, ns), 256, 0, st>>>(tab);
    A3D_LAUNCH_CHECK();
    k_nearest_other<ST_FINAL><<<nearest_grid(ST_FINAL), kNearestBlock, 0, st>>>(tab);
    k_cluster_best<ST_FINAL><<<dim3(nb, ns), 256, 0, st>>>(tab);
    A3D_LAUNCH_CHECK();
  }
  k_cluster_list<<<dim3(1, ns), 1024, 0, st>>>(tab);
  A3D_LAUNCH_CHECK();
  static int dbg = -1;   // A3D_CLICK_DBG=1: the stages' row counts of every sample on stderr (synchronises: tuning only)
  if (dbg < 0) {
    const char* e = getenv("A3D_CLICK_DBG");
    dbg = e ? atoi(e) : 0;
  }
  if (dbg) {
    (void)hipStreamSynchronize(st);
    for (int i = 0; i < n_samples; ++i) {
      int c[8];
      (void)hipMemcpy(c, host[i].w.n_err, sizeof(c), hipMemcpyDeviceToHost);
      fprintf(stderr, "clicks: sample %d n %lld wrong %d -> first stage survivors %d (champions %d) -> fine stage survivors %d (champions %d)%s\n", i,
              (long long)host[i].n, c[0], c[5], c[6], c[2], c[3], host[i].order ? " [spatial order]" : "");
    }
  }
  return A3D_OK;
}

// ---- a spatial order of a sample's points: Morton keys of the coordinates on a 2^16 grid over the bounding box, sorted
namespace a3d {
__global__ void k_so_minmax(const float* __restrict__ xyz, int64_t n, unsigned* __restrict__ mm) {   // mm: [3] min, [3] max as ordered uints
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = xyz[3 * i + a];
      mn[a] = fminf(mn[a], v);
      mx[a] = fmaxf(mx[a], v);
    }
  auto enc = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };   // order-preserving
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&mm[a], enc(mn[a]));
      atomicMax(&mm[3 + a], enc(mx[a]));
    }
  }
}
__device__ __forceinline__ unsigned long long spread16(unsigned v) {   // bit i of v -> bit 3 i
  unsigned long long x = v & 0xffffu;
  x = (x | (x << 16)) & 0x0000ff0000ffull;
  x = (x | (x << 8)) & 0x00f00f00f00full;
  x = (x | (x << 4)) & 0x0c30c30c30c3ull;
  x = (x | (x << 2)) & 0x249249249249ull;
  return x;
}
__global__ void k_so_keys(const float* __restrict__ xyz, int64_t n, const unsigned* __restrict__ mm, uint64_t* __restrict__ keys,
                          int* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto dec = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
  unsigned q[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float lo = dec(mm[a]), hi = dec(mm[3 + a]);
    const float t = hi > lo ? (xyz[3 * i + a] - lo) / (hi - lo) : 0.f;
    q[a] = (unsigned)fminf(fmaxf(t * 65535.f, 0.f), 65535.f);
  }
  keys[i] = spread16(q[0]) | (spread16(q[1]) << 1) | (spread16(q[2]) << 2);
  vals[i] = (int)i;
}
__global__ void k_so_inverse(const int32_t* __restrict__ order, int64_t n, int32_t* __restrict__ inv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inv[order[i]] = (int32_t)i;
}
struct SoWs {
  unsigned* mm;
  uint64_t *keys_in, *keys_out;
  int* vals_in;
  void* sort_temp;
  size_t sort_bytes, bytes;
};
static SoWs carve_so(void* base, int64_t n) {
  SoWs w;
  size_t off = 0;
  auto take = [&](size_t b) {
    void* p = base ? (char*)base + off : nullptr;
    off += align256(b);
    return p;
  };
  w.mm = (unsigned*)take(256);
  w.keys_in = (uint64_t*)take((size_t)n * 8);
  w.keys_out = (uint64_t*)take((size_t)n * 8);
  w.vals_in = (int*)take((size_t)n * 4);
  w.sort_bytes = radix_sort_temp_bytes((int)n);
  w.sort_temp = take(w.sort_bytes);
  w.bytes = off;
  return w;
}
}  // namespace a3d

extern "C" size_t a3d_click_spatial_order_workspace_bytes(int64_t n) {
  if (n <= 0 || n > (int64_t)1 << 28) return 0;
  return carve_so(nullptr, n).bytes;
}
extern "C" int a3d_click_spatial_order(const float* xyz_dev, int64_t n, int32_t* order_dev, int32_t* inv_dev, void* workspace_dev,
                                       size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0 || n > (int64_t)1 << 28 || !xyz_dev || !order_dev || !inv_dev) {
    set_error("a3d_click_spatial_order: bad arguments");
    return A3D_ERR_INVALID;
  }
  SoWs w = carve_so(workspace_dev, n);
  if (!workspace_dev || workspace_bytes < w.bytes || ((uintptr_t)workspace_dev & 255)) {
    set_error("a3d_click_spatial_order: workspace %zu < %zu", workspace_bytes, w.bytes);
    return A3D_ERR_WORKSPACE;
  }
  const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  A3D_HIP_CHECK(hipMemcpyAsync(w.mm, init, sizeof(init), hipMemcpyHostToDevice, st));
  const unsigned nb = (unsigned)((n + 255) / 256);
  k_so_minmax<<<nb < 512 ? nb : 512, 256, 0, st>>>(xyz_dev, n, w.mm);
  k_so_keys<<<nb, 256, 0, st>>>(xyz_dev, n, w.mm, w.keys_in, w.vals_in);
  A3D_LAUNCH_CHECK();
  RadixPass ps[kRadixMaxPasses];
  const int np = radix_passes(0, 48, ps);
  int rc = radix_sort_pairs(w.sort_temp, w.sort_bytes, w.keys_in, w.keys_out, w.vals_in, order_dev, (int)n, ps, np, st, nullptr, false);
  if (rc) return rc;
  k_so_inverse<<<nb, 256, 0, st>>>(order_dev, n, inv_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_click_clusters(const float* xyz_dev, const int32_t* pred_dev, const int32_t* labels_dev,
                                  int64_t n, a3d_click_cluster* out_dev, int max_out, int32_t* n_out_dev,
                                  void* workspace_dev, size_t workspace_bytes, void* stream) {
  a3d_click_sample sp;
  sp.xyz_dev = xyz_dev;
  sp.pred_dev = pred_dev;
  sp.labels_dev = labels_dev;
  sp.n = n;
  sp.out_dev = out_dev;
  sp.max_out = max_out;
  sp.n_out_dev = n_out_dev;
  sp.workspace_dev = workspace_dev;
  sp.workspace_bytes = workspace_bytes;
  sp.order_dev = nullptr;
  sp.inv_dev = nullptr;
  return a3d_click_clusters_batch(&sp, 1, stream);
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
