// Click-query transformer decoder on gfx950.
//
// Replaces Agile3d.forward_mask / mask_module (reference models/agile3d.py:183-384), the
// post-norm CrossAttentionLayer / SelfAttentionLayer / FFNLayer built on nn.MultiheadAttention
// (models/modules/attention_block.py:28-38,86-98,151-155) and the Fourier / click-time position
// encodings (models/position_embedding.py:13-41,123-152,210-225; agile3d.py:141-161).
//
// Decomposition per decoder iteration (DESIGN.md "decoder"):
//   N-point GEMMs (K/V/Q/out projections)       -> the fp32-MFMA GEMM of spconv.hip (a3d_linear)
//   click-to-scene attention over N keys        -> k_c2s_attn: flash-decoding with 16x16x4 MFMA,
//                                                  never materialises [heads,Q,N]; the label-derived
//                                                  mask (agile3d.py:367-380) is evaluated from one
//                                                  byte per point + per-label counts
//   everything of size [Q,128]                  -> k_query_block: one workgroup + FFN helpers per 64-query block, activations resident
//                                                  in LDS (Q <= 64); for 64 < Q <= 256 the queries are
//                                                  processed in blocks of 64 (one workgroup per block,
//                                                  split in two launches around the self attention)
//   scene-to-click attention (Q keys per point) -> k_s2c_attn: MFMA, softmax in registers
//   LayerNorm + mask head + argmax + histogram  -> k_ln_mask
// The position encoding is added to the GEMM input on the fly (K = (src + pos) Wk^T, Q likewise):
// k_dense reads both operands once, nothing click-independent is cached.
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <vector>

namespace a3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// A pointer read out of a descriptor table in memory is "generic" to the compiler: accesses through it become FLAT
// instructions, which count in BOTH vmcnt and lgkmcnt and are only ever waited for with s_waitcnt 0 -- every LDS wait of
// the loop then also waits for the point rows in flight from HBM and no prefetch survives.  These state the address
// space (global_load / global_store: vmcnt only, in order).
#define A3D_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ T gld(const T* p) { return *(const T A3D_GLOBAL*)p; }
template <typename T>
__device__ __forceinline__ void gst(T* p, T v) { *(T A3D_GLOBAL*)p = v; }
__device__ __forceinline__ f32x4 gld4(const float* p) { return *(const f32x4 A3D_GLOBAL*)p; }
__device__ __forceinline__ void gst4(float* p, f32x4 v) { *(f32x4 A3D_GLOBAL*)p = v; }
// "did any thread see its wait give up": through one word of the kernel's own dynamic LDS (__syncthreads_or keeps a static
// __shared__ word of its own, which the 160 KB kernels of this file cannot afford)
__device__ __forceinline__ bool block_any(int pred, int* slot) {
  if (threadIdx.x == 0) *slot = 0;
  __syncthreads();
  if (pred) *slot = 1;
  __syncthreads();
  return *(volatile int*)slot != 0;
}
// agent-scope (coherent across workgroups) relaxed accesses of the in-kernel hand-offs
template <typename T>
__device__ __forceinline__ T gld_agent(const T* p) {
  return __hip_atomic_load((const T A3D_GLOBAL*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void gst_agent(T* p, T v) {
  __hip_atomic_store((T A3D_GLOBAL*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int D = 128;        // hidden_dim
constexpr int H = 8;          // heads
constexpr int DH = 16;        // head dim
constexpr float kLnEps = 1e-5f;
constexpr float kNegBig = -1e30f;
constexpr int kC2SChunk = 128;   // points per c2s workgroup
constexpr int kPartStride = 18;  // m, l, acc[16]

// All-reduce over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48): the transposed MFMA layouts put the
// softmax / LayerNorm reductions there.  gfx950's v_permlane16_swap / v_permlane32_swap do each step as one VALU
// op (swap of the odd rows of one operand with the even rows of the other: with both operands = x the two results
// hold the two partners in every lane) instead of a ds_bpermute round trip through the LDS pipeline; same
// association order as the __shfl_xor(16) / __shfl_xor(32) pair (tools/permlane_check.hip).
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
// exp for the softmax weights (argument <= 0 after the max is subtracted): v_exp_f32 on x log2(e), ~2 ulp, against
// the ~15-instruction expf; the decoder's 1e-3 logits bar leaves four orders of magnitude of room
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
// The two attentions with the N points on one side work in the LOG2 domain: the query-side operand of their score product
// (the projected queries of click-to-scene, the keys of scene-to-click) is produced pre-scaled by 1 / sqrt(d_h) * log2(e),
// so a softmax weight is exp2(score - max) -- one v_exp_f32 without the multiply in front of it (a VALU instruction costs
// matrix-pipe time on gfx950), in every consumer: k_kv_c2s, k_c2s_attn, k_c2s_combine, k_s2c_out, k_q_s2c, k_s2c_attn_wide.
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kPointScoreScale = 0.25f * kLog2e;
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// ------------------------------------------------------------------------------ posenc
__global__ void __launch_bounds__(256) k_minmax_partial(const float* __restrict__ xyz, int n, float* part) {
  __shared__ float s[6][256];
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = xyz[3 * (size_t)i + a];
      mn[a] = fminf(mn[a], v);
      mx[a] = fmaxf(mx[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    s[a][threadIdx.x] = mn[a];
    s[3 + a][threadIdx.x] = mx[a];
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        s[a][threadIdx.x] = fminf(s[a][threadIdx.x], s[a][threadIdx.x + o]);
        s[3 + a][threadIdx.x] = fmaxf(s[3 + a][threadIdx.x], s[3 + a][threadIdx.x + o]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = s[threadIdx.x][0];
}
__global__ void k_minmax_final(const float* __restrict__ part, int nb, float* minmax) {
  // 6 waves, one per statistic (min x,y,z, max x,y,z) over the nb block partials; blockIdx.x = sample of a batch
  part += (size_t)blockIdx.x * nb * 6;
  minmax += 6 * blockIdx.x;
  const int a = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float v = a < 3 ? 3.4e38f : -3.4e38f;
  for (int b = lane; b < nb; b += 64) v = a < 3 ? fminf(v, part[b * 6 + a]) : fmaxf(v, part[b * 6 + a]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float w = __shfl_xor(v, o, 64);
    v = a < 3 ? fminf(v, w) : fmaxf(v, w);
  }
  if (lane == 0) minmax[a] = v;
}
// get_fourier_embeddings with normalize=True: u = (x-min)/(max-min); p = (2 pi u) @ B; [sin p, cos p]
__device__ __forceinline__ void fourier_row(const float* __restrict__ xyz, size_t i, int jj, const float* __restrict__ gaussB,
                                            const float* __restrict__ minmax, float* out) {
  const float two_pi = 6.283185307179586f;
  float p = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float mn = minmax[a], mx = minmax[3 + a];
    float u = (xyz[3 * i + a] - mn) / (mx - mn);
    u *= two_pi;
    p += u * gaussB[a * 64 + jj];
  }
  float sn, cs;
  sincosf(p, &sn, &cs);   // one argument reduction for both (full precision: the encoding's bar is 1e-4)
  out[i * D + jj] = sn;
  out[i * D + 64 + jj] = cs;
}
__global__ void k_fourier(const float* __restrict__ xyz, int n, const float* __restrict__ gaussB,
                          const float* __restrict__ minmax, float* out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)n * 64) return;
  fourier_row(xyz, e >> 6, (int)(e & 63), gaussB, minmax, out);
}
// ---- the same for every sample of a batch in three launches (rows of sample b = [start[b], start[b+1]) of one matrix)
constexpr int kPosBlocks = 16;   // partial-reduction blocks per sample
struct PosBatch {
  int ns;
  int start[64 + 1];
};
__global__ void __launch_bounds__(256) k_minmax_partial_b(const float* __restrict__ xyz, const PosBatch pb, float* part) {
  __shared__ float s[6][256];
  const int b = blockIdx.y, r0 = pb.start[b], r1 = pb.start[b + 1];
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int i = r0 + blockIdx.x * 256 + threadIdx.x; i < r1; i += kPosBlocks * 256) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = xyz[3 * (size_t)i + a];
      mn[a] = fminf(mn[a], v);
      mx[a] = fmaxf(mx[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    s[a][threadIdx.x] = mn[a];
    s[3 + a][threadIdx.x] = mx[a];
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        s[a][threadIdx.x] = fminf(s[a][threadIdx.x], s[a][threadIdx.x + o]);
        s[3 + a][threadIdx.x] = fmaxf(s[3 + a][threadIdx.x], s[3 + a][threadIdx.x + o]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 6) part[((size_t)b * kPosBlocks + blockIdx.x) * 6 + threadIdx.x] = s[threadIdx.x][0];
}
__global__ void k_fourier_b(const float* __restrict__ xyz, const PosBatch pb, const float* __restrict__ gaussB,
                            const float* __restrict__ minmax, float* out) {
  // blockIdx.y = sample (uniform: its row range comes from scalar loads), blockIdx.x = 4-row block inside it
  const int b = blockIdx.y, r0 = pb.start[b], r1 = pb.start[b + 1];
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = (size_t)r0 + (e >> 6);
  if (i >= (size_t)r1) return;
  fourier_row(xyz, i, (int)(e & 63), gaussB, minmax + 6 * b, out);
}

struct QueryMeta {   // device-resident, uploaded once per forward_mask
  int n_fg, n_bg_click, n_bgl, nq, K;
  int row[A3D_MAX_QUERIES];     // click row per query (-1 for learned bg)
  int time[A3D_MAX_QUERIES];    // click time per query
  int obj[A3D_MAX_QUERIES];     // object id per query (0 = background, -1 = padding)
  int qrange[A3D_MAX_QUERIES + 2];
};

struct QueryBufs {   // all [QP][...] fp32 in global scratch
  float *queries, *qpos, *qproj, *ks, *vs, *E;
  float *attn, *tmp, *tgt, *qk, *vc, *hidden;   // hidden: FFN partial sums of the helper workgroups [kQlMaxHelpers][QP][128]
  unsigned* sync;                                // [A3D_MAX_DEC_LAYERS][16] hand-off flags of k_query_block (zeroed per pass)
};
constexpr int kQlMaxHelpers = 8;
constexpr int kMaxQBlocks = A3D_MAX_QUERIES / 32;   // query blocks of a sample: 64 rows in the unfused path, 32 in the wide tier

// ---- one batch sample as the query-side kernels see it (k_query_init, k_c2s_combine, k_query_block: blockIdx.y / z)
struct QuerySample {
  const QueryMeta* meta;
  QueryBufs B;
  const float *feats, *posenc;   // the sample's rows: click features / encodings are gathered from them
  int* counts;
  const float* part;             // click-to-scene flash partials of this sample and how many there are
  int n_part;
  int n_part0;                   // ... in the first layer when it runs on cached keys / values (one per 128-point chunk)
  int qp;                        // rows of the sample's query-side buffers (the partials' [h][qp][part] layout)
};

// ---- one batch sample as the fused wide kernels see it (a3d_decoder_forward_batch) ------------------------------
// The three persistent wide kernels of a decoder layer (k_kv_c2s, k_q_s2c, k_out_ln_mask) are launched ONCE for all
// samples of a batch: workgroups [wg_begin, wg_end) work on this sample (its queries' data in their LDS), the waves
// of those workgroups stride over its 16-point groups.  One launch per layer instead of one per sample: the weight
// staging prologue and the last, partly filled round of groups are paid once per batch.
struct DecSampleDev {
  int wg_begin, wg_end;
  int n, nq, K, n_fg;
  const float *feats, *posenc;     // layer-0 input [n,128], position encoding [n,128]
  float *bufB, *bufC, *bufD;       // attention output; layer outputs (even layers -> bufC, odd -> bufD)
  float* logits;                   // [layers][n][K+1]
  unsigned char* labels;           // [n] arg-max object of the previous layer's mask
  int* counts;                     // [layers][A3D_MAX_QUERIES+1] points per object
  float* part;                     // click-to-scene flash partials [H][QP][slots][kPartStride]: the slots of one (head, query)
                                   // side by side, which is what k_c2s_combine walks (coalesced 72-byte records)
  const int *qobj, *qrange;        // QueryMeta::obj / qrange (device)
  const float *qproj, *ks, *vs, *E;
  const float* q0;                 // the scene's cached layer-0 scene-to-click queries (src + pos) Wq^T + bq [n][128], or nullptr
  const float *k0, *v0;            // the scene's cached layer-0 click-to-scene keys / values [n][128], or nullptr
  int qp, nqt;                     // rows of this sample's query-side buffers; its 16-query tiles (wide tier: the samples of a launch differ)
};
constexpr int kMaxBatchSamples = 64;
__device__ __forceinline__ const DecSampleDev& sample_of_wg(const DecSampleDev* samples, int ns) {
  int si = 0;
  while (si + 1 < ns && (int)blockIdx.x >= samples[si + 1].wg_begin) ++si;
  return samples[si];
}
__device__ __forceinline__ const float* layer_input(const DecSampleDev& sm, int l) {
  return l == 0 ? sm.feats : (((l - 1) & 1) ? sm.bufD : sm.bufC);
}

#include "decoder_wide.h"

// ------------------------------------------------------------------------------ click-to-scene
// One wave = one head over a chunk of points.  S = K_h q_h^T (A = key rows, B = q^T), online
// softmax per query column (lane-local: column = lane & 15), O^T += V_h^T P.
// nqt <= QT: the 16-query tiles this sample has (the batched launch serves samples of different query counts with the
// widest one's kernel: the tiles beyond a sample's own are skipped, not computed); chunk / nchunk: this workgroup's chunk of
// the sample's points and the number of them (the record layout of k_c2s_combine)
template <int QT>
__device__ __forceinline__ void c2s_attn_body(const float* __restrict__ Kc, const float* __restrict__ V, int n, const float* qproj,
                                              const int* qobj, const unsigned char* labels, const int* counts, float* part,
                                              int qp_total, int qb0, int nqt, int chunk, int nchunk) {
  const int lane = threadIdx.x & 63;
  qproj += (size_t)qb0 * D;
  qobj += qb0;
  const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int pbeg = chunk * kC2SChunk;
  const int pend = min(n, pbeg + kC2SChunk);

  f32x4 qf[QT];
  int obj[QT];
  bool qmask[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qr = qt < nqt ? qt * 16 + j : 0;      // tiles past the sample's own are never multiplied: any row will do
    qf[qt] = *(const f32x4*)(qproj + (size_t)qr * D + h * DH + 4 * g);
    obj[qt] = qobj[qr];
    // a query is masked only if its object currently owns at least one point (agile3d.py:369,375)
    qmask[qt] = labels != nullptr && obj[qt] >= 0 && counts[obj[qt]] > 0;
  }
  float m[QT], l[QT];
  f32x4 acc[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m[qt] = kNegBig;
    l[qt] = 0.f;
    acc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // the 16 points of the NEXT iteration are loaded while the current ones are multiplied
  f32x4 kf_n = (f32x4){0.f, 0.f, 0.f, 0.f};
  float vf_n[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned lab_n = 0;
  auto fetch = [&](int p0) {
    const int prow = p0 + j;
    kf_n = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (prow < n) kf_n = *(const f32x4*)(Kc + (size_t)prow * D + h * DH + 4 * g);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pr = p0 + 4 * g + t;
      vf_n[t] = pr < n ? V[(size_t)pr * D + h * DH + j] : 0.f;
    }
    lab_n = labels ? *(const unsigned*)(labels + p0 + 4 * g) : 0u;
  };
  if (pbeg < pend) fetch(pbeg);
  for (int p0 = pbeg; p0 < pend; p0 += 16) {
    const f32x4 kf = kf_n;
    float vf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) vf[t] = vf_n[t];
    const unsigned lab4 = lab_n;
    if (p0 + 16 < pend) fetch(p0 + 16);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (qt >= nqt) break;                 // wave-uniform
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t], qf[qt][t], s, 0, 0, 0);
      float mx = kNegBig;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pr = p0 + 4 * g + t;
        const int lab = (int)((lab4 >> (8 * t)) & 0xffu);
        const bool blocked = pr >= n || (qmask[qt] && lab != obj[qt]);
        s[t] = blocked ? kNegBig : s[t];
        mx = fmaxf(mx, s[t]);
      }
      mx = rows_max(mx);
      const float mnew = fmaxf(m[qt], mx);
      const float sc = exp2_fast(m[qt] - mnew);
      m[qt] = mnew;
      f32x4 p;
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        p[t] = exp2_fast(s[t] - mnew);
        ps += p[t];
      }
      l[qt] = l[qt] * sc + ps;
      acc[qt] *= sc;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[t], p[t], acc[qt], 0, 0, 0);
    }
  }
  // record of (head h, query q, chunk): [h][q][chunk] -- the chunks of a (head, query) are contiguous for k_c2s_combine
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    if (qt >= nqt) break;
    float lt = l[qt];
    lt = rows_sum(lt);
    float* pq = part + (((size_t)h * qp_total + qb0 + qt * 16 + j) * nchunk + chunk) * kPartStride;
    if (g == 0) {
      pq[0] = m[qt];
      pq[1] = lt;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) pq[2 + 4 * g + t] = acc[qt][t];
  }
}

template <int QT>
__global__ void __launch_bounds__(512) k_c2s_attn(const float* __restrict__ Kc, const float* __restrict__ V,
                                                  int n, const float* qproj, const int* qobj,
                                                  const unsigned char* labels, const int* counts,
                                                  float* part, int qp_total) {
  // blockIdx.y: the launch's query block of 16 QT rows
  c2s_attn_body<QT>(Kc, V, n, qproj, qobj, labels, counts, part, qp_total, (int)blockIdx.y * (QT * 16), QT, (int)blockIdx.x,
                    (int)gridDim.x);
}
// the first layer's attention over the scenes' CACHED keys / values for every sample of a call in ONE launch (blockIdx.y =
// sample; its own tile count and chunk count come from the table): 4 (training click rounds) .. 16 (lock-step evaluation)
// launches of 625 workgroups each -- 1.2 rounds of the device's 512 slots, i.e. two -- became one launch of 2 500 .. 10 000
template <int QT>
__global__ void __launch_bounds__(512) k_c2s_attn_b(const DecSampleDev* __restrict__ samples) {
  const DecSampleDev& sm = samples[blockIdx.y];
  const int nchunk = (sm.n + kC2SChunk - 1) / kC2SChunk;
  if ((int)blockIdx.x >= nchunk) return;
  c2s_attn_body<QT>(sm.k0, sm.v0, sm.n, sm.qproj, sm.qobj, nullptr, nullptr, sm.part, sm.qp, 0, min(sm.nqt, QT), (int)blockIdx.x, nchunk);
}


// ---- K/V projections fused into the click-to-scene attention (<= 32 queries) ------------------------------
// K = (src + pos) Wk^T + bk and V = src Wv^T + bv are produced per 16-point group straight into the MFMA
// operand layouts the attention consumes and never reach HBM: the transposed product (weights as the A
// operand) leaves K[point j][16h+4g..+3] in lane (g, j) = the A fragment of S = K_h q_h^T; the plain product
// leaves V[point 4g+t][16h+j] = the A fragment of O^T += V_h^T P.  Both packed weight matrices (2 x 64 KB)
// sit in LDS for the life of a persistent 8-wave workgroup; a wave walks its own sequence of 16-point groups,
// head by head (one 16-column slice of each GEMM at a time), and keeps the flash state of all 8 heads in
// registers.  Saves writing and re-reading K and V (4 x 41 MB per decoder iteration at 80 k points).
template <int QT>
__global__ void __launch_bounds__(512) k_kv_c2s(const DecSampleDev* __restrict__ samples, int ns, int layer,
                                                const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                const float* __restrict__ bk, const float* __restrict__ bv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int QP = QT * 16;
  // up to 32 queries: two waves share a sequence of point groups, four heads each (8 heads of flash state do not fit the
  // register file).  48 / 64 queries double the state per head: four waves share a group, two heads each.  At 64 queries the
  // projected queries get [64][128] rows with the float4 column XOR-ed by the row (conflict-free b128 reads without the
  // 4-float pad) and the biases stay in registers: 128 KB of weights + 32 KB = exactly the 160 KB of LDS
  constexpr int HW = QT <= 2 ? 4 : 2, WPG = H / HW, SPW = 8 / WPG;
  constexpr bool SWZ = QT == 4;
  constexpr int LDQ = SWZ ? 128 : 132;
  const DecSampleDev& sm = sample_of_wg(samples, ns);
  const int lb = blockIdx.x - sm.wg_begin, nwg = sm.wg_end - sm.wg_begin;
  const int n = sm.n, ngroups = (n + 15) / 16, qp_total = QP;
  const float* __restrict__ X = layer_input(sm, layer);
  const float* __restrict__ Pe = sm.posenc;
  const float* qproj = sm.qproj;
  const int* qobj = sm.qobj;
  const unsigned char* labels = layer > 0 ? sm.labels : nullptr;                       // previous layer's mask labels
  const int* counts = layer > 0 ? sm.counts + (size_t)(layer - 1) * (A3D_MAX_QUERIES + 1) : nullptr;
  float* part = sm.part;
  f32x4* Wkl = (f32x4*)smem;              // [8 S][8 ct][64 lanes]
  f32x4* Wvl = Wkl + 8 * 8 * 64;
  float* qp_l = (float*)(Wvl + 8 * 8 * 64);   // [QP][132] projected queries
  float* bk_l = qp_l + QP * LDQ;              // [128] + [128] biases: the loop below touches global memory only for
  float* bv_l = bk_l + D;                     // the point rows (vmcnt is in order: any other load would wait for them)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int slot = lb * SPW + wave / WPG, nslots = nwg * SPW;
  // rows of the first group: in flight behind the weight staging
  f32x4 xs[8], pe[8];
  unsigned lab_nx = 0u;
  if (slot < ngroups) {
    const size_t row = (size_t)min(slot * 16 + j, n - 1);
    if (labels) lab_nx = gld((const unsigned*)(labels + slot * 16 + 4 * g));
#pragma unroll
    for (int S = 0; S < 8; ++S) xs[S] = gld4(X + row * D + 16 * S + 4 * g);
#pragma unroll
    for (int S = 0; S < 8; ++S) pe[S] = gld4(Pe + row * D + 16 * S + 4 * g);
  }
  {
    constexpr int TOT = 2 * 8 * 8 * 64;
    for (int base = threadIdx.x; base < TOT; base += 8 * 512) {
      f32x4 t8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + u * 512;
        if (e < TOT) t8[u] = e < TOT / 2 ? ((const f32x4*)Wk)[e] : ((const f32x4*)Wv)[e - TOT / 2];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u * 512 < TOT) Wkl[base + u * 512] = t8[u];
    }
  }
  for (int e = threadIdx.x; e < QP * 32; e += 512) {
    const int r = e >> 5, c = e & 31;
    *(f32x4*)(qp_l + r * LDQ + (SWZ ? (c ^ (r & 15)) : c) * 4) = gld4(qproj + (size_t)r * D + c * 4);
  }
  if (!SWZ && threadIdx.x < D) {
    bk_l[threadIdx.x] = bk[threadIdx.x];
    bv_l[threadIdx.x] = bv[threadIdx.x];
  }
  int obj[QT];
  bool qmask[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    obj[qt] = gld(qobj + qt * 16 + j);
    qmask[qt] = labels != nullptr && obj[qt] >= 0 && gld(counts + obj[qt]) > 0;
  }
  __syncthreads();
  // WPG waves share a sequence of point groups, HW heads each
  const int h0 = (wave % WPG) * HW;
  f32x4 bkr[HW];
  float bvr[HW];
#pragma unroll
  for (int hl = 0; hl < HW; ++hl) {
    bkr[hl] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bvr[hl] = 0.f;
    if constexpr (SWZ) {
      bkr[hl] = *(const f32x4*)(bk + 16 * (h0 + hl) + 4 * g);
      bvr[hl] = bv[16 * (h0 + hl) + j];
    }
  }
  float m[HW][QT], l[HW][QT];
  f32x4 acc[HW][QT];
#pragma unroll
  for (int hl = 0; hl < HW; ++hl)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      m[hl][qt] = kNegBig;
      l[hl][qt] = 0.f;
      acc[hl][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  for (int grp = slot; grp < ngroups; grp += nslots) {
    const int p0 = grp * 16;
    const unsigned lab4 = lab_nx;
    // the next group of this wave pair (the last iteration re-reads its own rows: unconditional loads keep the
    // compiler from sinking the prefetch into one conditional block behind the attention phase)
    const int next = grp + nslots < ngroups ? grp + nslots : grp;
    const size_t nrow = (size_t)min(next * 16 + j, n - 1);
    if (labels) lab_nx = gld((const unsigned*)(labels + next * 16 + 4 * g));
    // ---- projections of this wave's four heads, k-step by k-step: K^T slice kf[hl] = channels 16h+4g..+3 of point j,
    // V slice vv[hl] = channel 16h+j of points 4g..4g+3.  A k-step's row fragments are dead once its MFMAs are
    // issued, so the SAME registers take the next group's rows right away: the loads have the rest of this phase
    // and the whole attention phase to land (rolling prefetch, no extra registers).
    f32x4 kf[HW], vv[HW];
#pragma unroll
    for (int hl = 0; hl < HW; ++hl) {
      kf[hl] = SWZ ? bkr[hl] : *(const f32x4*)(bk_l + 16 * (h0 + hl) + 4 * g);
      const float bvj = SWZ ? bvr[hl] : bv_l[16 * (h0 + hl) + j];
      vv[hl] = (f32x4){bvj, bvj, bvj, bvj};
    }
    {
      // weight fragments one (K-step, head) ahead of the eight MFMAs that use them: left to itself the compiler issues a
      // step's ds_read_b128 right in front of their MFMAs and waits out the LDS latency three times per K-step
      f32x4 wk = Wkl[(0 * 8 + h0) * 64 + lane], wv = Wvl[(0 * 8 + h0) * 64 + lane];
#pragma unroll
      for (int S = 0; S < 8; ++S) {
        const f32x4 xp = xs[S] + pe[S];
#pragma unroll
        for (int hl = 0; hl < HW; ++hl) {
          const int i2 = S * HW + hl + 1, S2 = i2 / HW, hl2 = i2 % HW;
          f32x4 nk = wk, nv = wv;
          if (i2 < 8 * HW) {
            nk = Wkl[(S2 * 8 + h0 + hl2) * 64 + lane];
            nv = Wvl[(S2 * 8 + h0 + hl2) * 64 + lane];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            kf[hl] = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[t], xp[t], kf[hl], 0, 0, 0);
            vv[hl] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[S][t], wv[t], vv[hl], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          wk = nk;
          wv = nv;
        }
        xs[S] = gld4(X + nrow * D + 16 * S + 4 * g);
        pe[S] = gld4(Pe + nrow * D + 16 * S + 4 * g);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- attention of the 16 points against every query, flash-style running state per (head, query tile)
    auto qfrag = [&](int qt, int h) {   // projected queries [16 qt + j][16 h + 4 g ..+3]
      const int c = 4 * h + g;
      return *(const f32x4*)(qp_l + (qt * 16 + j) * LDQ + (SWZ ? (c ^ j) : c) * 4);
    };
    f32x4 qf_n = qfrag(0, h0);   // the next tile's fragment, one tile ahead
    // which (point, query) pairs are blocked does not depend on the head: the mask is taken ONCE per group and query
    // tile as the initial value of the score accumulators (0 / kNegBig; kNegBig + a dot product is kNegBig again) instead
    // of a compare + select per score and head -- VALU instructions cost matrix-pipe time on gfx950 (tools/coissue_ubench.hip)
    f32x4 mb[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pr = p0 + 4 * g + t;
        const int lab = (int)((lab4 >> (8 * t)) & 0xffu);
        const bool blocked = pr >= n || (qmask[qt] && lab != obj[qt]);
        mb[qt][t] = blocked ? kNegBig : 0.f;
      }
#pragma unroll
    for (int hl = 0; hl < HW; ++hl) {
      const int h = h0 + hl;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const f32x4 qf = qf_n;
        {
          const int i2 = hl * QT + qt + 1;
          if (i2 < HW * QT) qf_n = qfrag(i2 % QT, h0 + i2 / QT);
        }
        f32x4 sc4 = mb[qt];
#pragma unroll
        for (int t = 0; t < 4; ++t) sc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[hl][t], qf[t], sc4, 0, 0, 0);
        float mx = kNegBig;
#pragma unroll
        for (int t = 0; t < 4; ++t) mx = fmaxf(mx, sc4[t]);
        mx = rows_max(mx);
        const float mnew = fmaxf(m[hl][qt], mx);
        const float scl = exp2_fast(m[hl][qt] - mnew);
        m[hl][qt] = mnew;
        f32x4 pw;
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          pw[t] = exp2_fast(sc4[t] - mnew);
          ps += pw[t];
        }
        l[hl][qt] = l[hl][qt] * scl + ps;
        acc[hl][qt] *= scl;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[hl][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[hl][t], pw[t], acc[hl][qt], 0, 0, 0);
      }
    }
  }
  // The SPW slot groups of the workgroup hold flash states of the same heads and queries over different points: they are
  // merged here, through the LDS the weights no longer need, and the workgroup writes ONE partial (m, l, acc[16]) per head
  // and query -- a quarter / half of the records k_c2s_combine has to walk (1 024 -> 256 per (head, query) for one scene).
  // A lane's l is still its own partial sum (rows_sum is linear), m is uniform over the four lanes of a query column.
  {
    constexpr int REC = 6 * 64;   // floats per (wave, head, query tile): m, l, acc[4] x 64 lanes, lane-contiguous
    float* red = (float*)smem;    // [SPW - 1][WPG][HW * QT][6][64]
    const int sg = wave / WPG, wl = wave % WPG;
    __syncthreads();              // every wave is through with the weights
    if (sg > 0) {
#pragma unroll
      for (int hl = 0; hl < HW; ++hl)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          float* r = red + ((size_t)((sg - 1) * WPG + wl) * (HW * QT) + hl * QT + qt) * REC + lane;
          r[0] = m[hl][qt];
          r[64] = l[hl][qt];
#pragma unroll
          for (int t = 0; t < 4; ++t) r[128 + 64 * t] = acc[hl][qt][t];
        }
    }
    __syncthreads();
    if (sg > 0) return;
#pragma unroll
    for (int hl = 0; hl < HW; ++hl)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mm = m[hl][qt], ll = l[hl][qt];
        f32x4 aa = acc[hl][qt];
#pragma unroll
        for (int o = 1; o < SPW; ++o) {
          const float* r = red + ((size_t)((o - 1) * WPG + wl) * (HW * QT) + hl * QT + qt) * REC + lane;
          const float mo = r[0];
          const float mnew = fmaxf(mm, mo);
          const float sa = exp2_fast(mm - mnew), sb = exp2_fast(mo - mnew);
          mm = mnew;
          ll = ll * sa + r[64] * sb;
#pragma unroll
          for (int t = 0; t < 4; ++t) aa[t] = aa[t] * sa + r[128 + 64 * t] * sb;
        }
        const int h = h0 + hl;
        const float lt = rows_sum(ll);
        float* pq = part + (((size_t)h * qp_total + qt * 16 + j) * nwg + lb) * kPartStride;
        if (g == 0) {
          gst(pq, mm);
          gst(pq + 1, lt);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) gst(pq + 2 + 4 * g + t, aa[t]);
      }
  }
}
constexpr int kFusedC2SGrid = 256;   // one persistent 8-wave workgroup per CU (128 KB of weights in LDS)

// merge the per-chunk flash partials (m, l, acc[16]) of one (query, head): one wave per pair
__global__ void __launch_bounds__(64) k_c2s_combine(const QuerySample* __restrict__ qs, int cached0) {
  const QuerySample& smp = qs[blockIdx.y];
  const int QP = smp.qp;
  const int q = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
  if (q >= smp.meta->nq) return;
  const float* __restrict__ part = smp.part;
  const int nchunk = cached0 ? smp.n_part0 : smp.n_part;
  float* attn = smp.B.attn;
  float m = kNegBig, l = 0.f, o[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] = 0.f;
  for (int ch = lane; ch < nchunk; ch += 64) {
    const float* p = part + (((size_t)h * QP + q) * nchunk + ch) * kPartStride;
    const float pm = gld(p);
    const float mn = fmaxf(m, pm);
    const float a = exp2f(m - mn), b = exp2f(pm - mn);   // the partials' maxima are log2-domain scores
    l = l * a + gld(p + 1) * b;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = o[d] * a + gld(p + 2 + d) * b;
    m = mn;
  }
  float M = m;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
  const float sc = exp2f(m - M);
  l *= sc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    float v = o[d] * sc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == d) gst(attn + (size_t)q * D + h * DH + d, v / l);
  }
}

// ------------------------------------------------------------------------------ scene-to-click
// Unfused scene-to-click attention (more than 64 queries, or A3D_FUSED_C2S=0): 8 waves x 16 points per workgroup,
// S^T = ks_h Qs_h^T with the queries' keys/values staged in LDS in blocks of 64 (row stride 132 floats:
// conflict-free b128 fragment reads), online softmax across blocks (per head: running max, partial sum, O^T
// accumulator in registers).
template <int QT>
__global__ void __launch_bounds__(512) k_s2c_attn_wide(const float* __restrict__ Qs, int n, const float* ks,
                                                       const float* vs, int nq, int nblk, float* O) {
  constexpr int QP = QT * 16, LD = 132;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ks_l = (float*)smem;
  float* vs_l = ks_l + QP * LD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int p0 = (blockIdx.x * 8 + wave) * 16;
  const bool active = p0 < n;
  const int prow = min(p0 + j, n - 1);
  const float* qrow = Qs + (size_t)prow * D;
  float m[H], l[H];
  f32x4 acc[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    m[h] = kNegBig;
    l[h] = 0.f;
    acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int kb = 0; kb < nblk; ++kb) {
    __syncthreads();
    for (int e = threadIdx.x; e < QP * 32; e += 512) {
      const int r = e >> 5, c4 = (e & 31) * 4;
      *(f32x4*)(ks_l + r * LD + c4) = *(const f32x4*)(ks + (size_t)(kb * QP + r) * D + c4);
      *(f32x4*)(vs_l + r * LD + c4) = *(const f32x4*)(vs + (size_t)(kb * QP + r) * D + c4);
    }
    __syncthreads();
    if (!active) continue;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const f32x4 qf = *(const f32x4*)(qrow + h * DH + 4 * g);
      f32x4 s[QT];
      float mx = kNegBig;
#pragma unroll
      for (int kt = 0; kt < QT; ++kt) {
        const f32x4 kf = *(const f32x4*)(ks_l + (kt * 16 + j) * LD + h * DH + 4 * g);
        s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t], qf[t], s[kt], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (kb * QP + kt * 16 + 4 * g + t >= nq) s[kt][t] = kNegBig;
          mx = fmaxf(mx, s[kt][t]);
        }
      }
      mx = rows_max(mx);
      const float mnew = fmaxf(m[h], mx);
      const float sc = exp2_fast(m[h] - mnew);
      m[h] = mnew;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < QT; ++kt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s[kt][t] = exp2_fast(s[kt][t] - mnew);
          ps += s[kt][t];
        }
      l[h] = l[h] * sc + ps;
      acc[h] *= sc;
#pragma unroll
      for (int kt = 0; kt < QT; ++kt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float vf = vs_l[(kt * 16 + 4 * g + t) * LD + h * DH + j];
          acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, s[kt][t], acc[h], 0, 0, 0);
        }
    }
  }
  if (!active) return;
  float* orow = O + (size_t)prow * D;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float lt = l[h];
    lt = rows_sum(lt);
    if (p0 + j < n) *(f32x4*)(orow + h * DH + 4 * g) = acc[h] * (1.f / lt);
  }
}

// ------------------------------------------------------------------------------ LN + mask head
// src_new = LayerNorm(Y) (in place); logits[p, q] = src_new[p] . E[q]; per-object max -> [1+K];
// argmax (first max) -> label byte; per-label histogram.  4 waves x 16 points per workgroup.
template <int QT>
__global__ void __launch_bounds__(256) k_ln_mask(float* Y, int n, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, const float* E, int nq,
                                                 const int* qrange /*[K+2]*/, int n_fg, int K, float* logits,
                                                 unsigned char* labels, int* counts, int nblk) {
  constexpr int QP = QT * 16, LD = 132;
  const int LL = nblk * QP + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* E_l = (float*)smem;                    // [QP][132]
  float* L_l = E_l + QP * LD;                   // [4 waves][16][LL]
  float* O_l = L_l + 4 * 16 * LL;               // [4 waves][16][K+1]
  int* hist = (int*)(O_l + 4 * 16 * (K + 1));   // [K+1]
  for (int e = threadIdx.x; e <= K; e += 256) hist[e] = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int p0 = (blockIdx.x * 4 + wave) * 16;
  const bool wave_active = p0 < n;
  const int prow = min(p0 + j, n - 1);
  float* Lw = L_l + wave * 16 * LL;
  float* Ow = O_l + wave * 16 * (K + 1);
  f32x4 y[8];
  if (wave_active) {
    float* yrow = Y + (size_t)prow * D;
    float sum = 0.f;
#pragma unroll
    for (int S = 0; S < 8; ++S) {
      y[S] = *(const f32x4*)(yrow + 16 * S + 4 * g);
      sum += y[S][0] + y[S][1] + y[S][2] + y[S][3];
    }
    sum = rows_sum(sum);
    const float mean = sum * (1.f / D);
    float var = 0.f;
#pragma unroll
    for (int S = 0; S < 8; ++S)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float d = y[S][t] - mean;
        var += d * d;
      }
    var = rows_sum(var);
    const float rstd = rsqrtf(var * (1.f / D) + kLnEps);
#pragma unroll
    for (int S = 0; S < 8; ++S) {
      const f32x4 ga = *(const f32x4*)(gamma + 16 * S + 4 * g);
      const f32x4 be = *(const f32x4*)(beta + 16 * S + 4 * g);
#pragma unroll
      for (int t = 0; t < 4; ++t) y[S][t] = (y[S][t] - mean) * rstd * ga[t] + be[t];
      if (p0 + j < n) *(f32x4*)(yrow + 16 * S + 4 * g) = y[S];
    }
  }
  for (int qb = 0; qb < nblk; ++qb) {   // mask-embedding rows of 64 queries at a time through LDS
    if (qb) __syncthreads();
    for (int e = threadIdx.x; e < QP * 32; e += 256) {
      const int r = e >> 5, c4 = (e & 31) * 4;
      *(f32x4*)(E_l + r * LD + c4) = *(const f32x4*)(E + (size_t)(qb * QP + r) * D + c4);
    }
    __syncthreads();
    if (!wave_active) continue;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int S = 0; S < 8; ++S) {
        const f32x4 ef = *(const f32x4*)(E_l + (qt * 16 + j) * LD + 16 * S + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(y[S][t], ef[t], acc, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) Lw[(4 * g + t) * LL + qb * QP + qt * 16 + j] = acc[t];
    }
  }
  __syncthreads();
  if (wave_active) {
    // per-object max over that object's queries: object o>=1 -> fg queries [qrange[o], qrange[o+1]);
    // object 0 -> all background queries [n_fg, nq)   (agile3d.py:348-365)
    for (int o = g; o <= K; o += 4) {
      const int qb = o == 0 ? n_fg : qrange[o], qe = o == 0 ? nq : qrange[o + 1];
      float mxv = -3.4e38f;
      for (int q = qb; q < qe; ++q) mxv = fmaxf(mxv, Lw[j * LL + q]);
      Ow[j * (K + 1) + o] = mxv;
    }
  }
  __syncthreads();
  if (wave_active) {
    if (lane < 16 && p0 + lane < n) {
      float best = Ow[lane * (K + 1)];
      int bi = 0;
      for (int o = 1; o <= K; ++o) {
        const float v = Ow[lane * (K + 1) + o];
        if (v > best) {
          best = v;
          bi = o;
        }
      }
      labels[p0 + lane] = (unsigned char)bi;
      atomicAdd(&hist[bi], 1);
    }
    const int rows = min(16, n - p0);
    for (int e = lane; e < rows * (K + 1); e += 64) logits[(size_t)p0 * (K + 1) + e] = Ow[e];
  }
  __syncthreads();
  for (int e = threadIdx.x; e <= K; e += 256)
    if (hist[e]) atomicAdd(&counts[e], hist[e]);
}



// ---- scene-to-click query projection fused into the attention (<= 64 queries) ---------------------------------
// Q = (src + pos) Wq^T + bq per 16-point group, head by head, in the transposed accumulator layout = the B fragment
// of S^T = ks_h Q_h^T; softmax over the (few) queries in registers; O^T = vs_h^T P.  Q never reaches HBM.
// Persistent 8-wave workgroups: packed Wq (64 KB) + the queries' keys / values in LDS.
// QC: the layer's queries come from the scene's cache (a3d_decoder_sample::kv0_dev, third block: they depend on the scene only
// in the first layer, and the interactive loop runs ~100 passes on one scene): no projection, no read of src and pos
template <int QT, bool QC = false>
__global__ void __launch_bounds__(512) k_q_s2c(const DecSampleDev* __restrict__ samples, int ns, int layer,
                                               const float* __restrict__ Wq, const float* __restrict__ bq) {
  constexpr int QP = QT * 16, LD = 132, NW = 8;
  const DecSampleDev& sm = sample_of_wg(samples, ns);
  const int lb = blockIdx.x - sm.wg_begin, nwg = sm.wg_end - sm.wg_begin;
  const int n = sm.n, ngroups = (n + 15) / 16, nq = sm.nq;
  const float* __restrict__ X = layer_input(sm, layer);
  const float* __restrict__ Pe = sm.posenc;
  const float* ks = sm.ks;
  const float* vs = sm.vs;
  float* __restrict__ O = sm.bufB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* Wl = (f32x4*)smem;                       // [8 S][8 ct][64]
  constexpr int LT = QP + 4;                      // row stride of the transposed values (16-byte rows, bank spread)
  float* ks_l = (float*)(Wl + 8 * 8 * 64);        // [QP][132]
  float* vt_l = ks_l + QP * LD;                   // [128][LT]  V^T: the PV fragment (4 keys of one channel) is one b128
  float* bq_l = vt_l + D * LT;                    // [128]
  if (threadIdx.x < D) bq_l[threadIdx.x] = bq[threadIdx.x];
  {
    constexpr int TOT = 8 * 8 * 64;
    for (int base = threadIdx.x; base < TOT; base += 8 * 512) {
      f32x4 t8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u * 512 < TOT) t8[u] = ((const f32x4*)Wq)[base + u * 512];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u * 512 < TOT) Wl[base + u * 512] = t8[u];
    }
  }
  for (int e = threadIdx.x; e < QP * 32; e += 512) {
    const int r = e >> 5, c4 = (e & 31) * 4;
    *(f32x4*)(ks_l + r * LD + c4) = gld4(ks + (size_t)r * D + c4);
    const f32x4 v4 = gld4(vs + (size_t)r * D + c4);
#pragma unroll
    for (int t = 0; t < 4; ++t) vt_l[(c4 + t) * LT + r] = v4[t];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int stride = nwg * NW;
  int grp = lb * NW + wave;
  f32x4 nx[8], np[8];
  auto fetch = [&](int gq) {
    const size_t row = (size_t)min(gq * 16 + j, n - 1);
    if constexpr (QC) {   // head h's fragment of the cached queries: the position-encoding rows' access pattern
      const float* qr = sm.q0 + row * D + 4 * g;
#pragma unroll
      for (int S = 0; S < 8; ++S) np[S] = gld4(qr + 16 * S);
    } else {
      const float* xr = X + row * D + 4 * g;
      const float* pr = Pe + row * D + 4 * g;
#pragma unroll
      for (int S = 0; S < 8; ++S) nx[S] = gld4(xr + 16 * S);
#pragma unroll
      for (int S = 0; S < 8; ++S) np[S] = gld4(pr + 16 * S);
    }
  };
  if (grp < ngroups) fetch(grp);
  while (grp < ngroups) {
    const int p0 = grp * 16;
    const int prow = min(p0 + j, n - 1);
    f32x4 xp[8];
#pragma unroll
    for (int S = 0; S < 8; ++S) xp[S] = QC ? np[S] : nx[S] + np[S];
    const int next = grp + stride;
    if (next < ngroups) fetch(next);
    float* orow = O + (size_t)prow * D;
    auto head = [&](int h, f32x4 qf) {
      if constexpr (!QC) {
      // bias from LDS: the only vector-memory traffic inside the loop is the prefetch and the stores (vmcnt is
      // in order -- a global load here would wait for the whole prefetch and the previous head's store)
      qf = *(const f32x4*)(bq_l + 16 * h + 4 * g);   // Q[point j][16h+4g..+3]
#pragma unroll
      for (int S = 0; S < 8; ++S) {
        const f32x4 w = Wl[(S * 8 + h) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 4; ++t) qf = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], xp[S][t], qf, 0, 0, 0);
      }
      }
      f32x4 sc[QT];
      float mx = kNegBig;
#pragma unroll
      for (int kt = 0; kt < QT; ++kt) {
        const f32x4 kf = *(const f32x4*)(ks_l + (kt * 16 + j) * LD + h * DH + 4 * g);
        sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t], qf[t], sc[kt], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (kt * 16 + 4 * g + t >= nq) sc[kt][t] = kNegBig;
          mx = fmaxf(mx, sc[kt][t]);
        }
      }
      mx = rows_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < QT; ++kt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          sc[kt][t] = exp2_fast(sc[kt][t] - mx);
          sum += sc[kt][t];
        }
      sum = rows_sum(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < QT; ++kt) {
        const f32x4 vf = *(const f32x4*)(vt_l + (h * DH + j) * LT + kt * 16 + 4 * g);   // V^T: keys 4g..4g+3 of channel 16h+j
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[t], sc[kt][t] * inv, acc, 0, 0, 0);
      }
      if (p0 + j < n) gst4(orow + h * DH + 4 * g, acc);
    };
    if constexpr (QC) {   // unrolled: xp[h] is a register, not an indexed array
#pragma unroll
      for (int h = 0; h < H; ++h) head(h, xp[h]);
    } else {
#pragma unroll 2
      for (int h = 0; h < H; ++h) head(h, (f32x4){0.f, 0.f, 0.f, 0.f});
    }
    grp = next;
  }
}

// ---- scene-to-click output projection fused with the residual, LayerNorm and the mask head (<= 64 queries) -----
// Y = LayerNorm(O Wo^T + bo + src) and everything k_ln_mask does with it, per 16-point group, straight from the
// transposed MFMA accumulators (lane (g, j) holds channels 16ct+4g..+3 of point j = the layout the LayerNorm and
// the logits MFMA want): the pre-norm activation never reaches HBM.  Persistent 8-wave workgroups keep the packed
// Wo (64 KB) and the mask embeddings E in LDS; waves walk their own sequences of groups (wave-private LDS scratch,
// no workgroup barrier inside the loop) with the next group's fragments in flight behind the current MFMAs.
template <int QT>
__global__ void __launch_bounds__(512) k_out_ln_mask(const DecSampleDev* __restrict__ samples, int ns, int layer,
                                                     const float* __restrict__ Wo, const float* __restrict__ bo,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta) {
  constexpr int QP = QT * 16, LD = 132, LL = QP + 1, NW = 8;
  const DecSampleDev& sm = sample_of_wg(samples, ns);
  const int lb = blockIdx.x - sm.wg_begin, nwg = sm.wg_end - sm.wg_begin;
  const int n = sm.n, ngroups = (n + 15) / 16, nq = sm.nq, n_fg = sm.n_fg, K = sm.K;
  const float* __restrict__ O = sm.bufB;
  const float* __restrict__ Xres = layer_input(sm, layer);
  const float* E = sm.E;
  const int* qrange = sm.qrange;
  float* __restrict__ Y = (layer & 1) ? sm.bufD : sm.bufC;
  float* logits = sm.logits + (size_t)layer * n * (K + 1);
  unsigned char* labels = sm.labels;
  int* counts = sm.counts + (size_t)layer * (A3D_MAX_QUERIES + 1);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* Wl = (f32x4*)smem;                       // [8 S][8 ct][64]
  float* bo_l = (float*)(Wl + 8 * 8 * 64);        // [128] x 3: bias, LayerNorm weight and bias (no vector-memory
  float* ga_l = bo_l + D;                         // loads inside the loop except the prefetch: vmcnt is in order)
  float* be_l = ga_l + D;
  float* E_l = be_l + D;                          // [QP][132]
  float* L_l = E_l + QP * LD;                     // [NW][16][LL]
  float* O_l = L_l + NW * 16 * LL;                // [NW][16][K+1]
  int* hist = (int*)(O_l + NW * 16 * (K + 1));    // [K+1]
  int* qr_l = hist + K + 1;                       // [K+2] query range of every object
  for (int e = threadIdx.x; e <= K + 1; e += 512) qr_l[e] = gld(qrange + e);
  {
    constexpr int TOT = 8 * 8 * 64;
    for (int base = threadIdx.x; base < TOT; base += 8 * 512) {
      f32x4 t8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u * 512 < TOT) t8[u] = ((const f32x4*)Wo)[base + u * 512];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u * 512 < TOT) Wl[base + u * 512] = t8[u];
    }
  }
  for (int e = threadIdx.x; e < QP * 32; e += 512) {
    const int r = e >> 5, c4 = (e & 31) * 4;
    *(f32x4*)(E_l + r * LD + c4) = gld4(E + (size_t)r * D + c4);
  }
  for (int e = threadIdx.x; e <= K; e += 512) hist[e] = 0;
  if (threadIdx.x < D) {
    bo_l[threadIdx.x] = bo[threadIdx.x];
    ga_l[threadIdx.x] = gamma[threadIdx.x];
    be_l[threadIdx.x] = beta[threadIdx.x];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  float* Lw = L_l + wave * 16 * LL;
  float* Ow = O_l + wave * 16 * (K + 1);
  const int stride = nwg * NW;
  int grp = lb * NW + wave;
  f32x4 nx[8], nr[8];
  auto fetch = [&](int gq) {
    const size_t row = (size_t)min(gq * 16 + j, n - 1);
    const float* orow = O + row * D + 4 * g;
    const float* rrow = Xres + row * D + 4 * g;
#pragma unroll
    for (int S = 0; S < 8; ++S) nx[S] = gld4(orow + 16 * S);
#pragma unroll
    for (int S = 0; S < 8; ++S) nr[S] = gld4(rrow + 16 * S);   // residual rows: channels 16ct+4g..+3
  };
  if (grp < ngroups) fetch(grp);
  while (grp < ngroups) {
    const int p0 = grp * 16;
    const int prow = min(p0 + j, n - 1);
    f32x4 a[8];
    f32x4 y[8];   // y[ct] = channels 16ct+4g..+3 of point j
#pragma unroll
    for (int S = 0; S < 8; ++S) a[S] = nx[S];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) y[ct] = *(const f32x4*)(bo_l + 16 * ct + 4 * g) + nr[ct];
    const int next = grp + stride;
    if (next < ngroups) fetch(next);
#pragma unroll
    for (int S = 0; S < 8; ++S) {
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        const f32x4 w = Wl[(S * 8 + ct) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 4; ++t) y[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], a[S][t], y[ct], 0, 0, 0);
      }
      asm volatile("" ::: "memory");
    }
    // LayerNorm over the 128 channels of point j (4 g-lanes x 8 ct x 4)
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) sum += y[ct][0] + y[ct][1] + y[ct][2] + y[ct][3];
    sum = rows_sum(sum);
    const float mean = sum * (1.f / D);
    float var = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float d = y[ct][t] - mean;
        var += d * d;
      }
    var = rows_sum(var);
    const float rstd = rsqrtf(var * (1.f / D) + kLnEps);
    float* yrow = Y + (size_t)prow * D;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const f32x4 ga = *(const f32x4*)(ga_l + 16 * ct + 4 * g);
      const f32x4 be = *(const f32x4*)(be_l + 16 * ct + 4 * g);
#pragma unroll
      for (int t = 0; t < 4; ++t) y[ct][t] = (y[ct][t] - mean) * rstd * ga[t] + be[t];
      if (p0 + j < n) gst4(yrow + 16 * ct + 4 * g, y[ct]);
    }
    // logits of the 16 points against every query (C layout: row = point 4g+t, column = query j)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int S = 0; S < 8; ++S) {
        const f32x4 ef = *(const f32x4*)(E_l + (qt * 16 + j) * LD + 16 * S + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(y[S][t], ef[t], acc, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) Lw[(4 * g + t) * LL + qt * 16 + j] = acc[t];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private scratch: written above, read below by other lanes
    for (int o = g; o <= K; o += 4) {
      const int qb = o == 0 ? n_fg : qr_l[o], qe = o == 0 ? nq : qr_l[o + 1];
      float mxv = -3.4e38f;
      for (int q = qb; q < qe; ++q) mxv = fmaxf(mxv, Lw[j * LL + q]);
      Ow[j * (K + 1) + o] = mxv;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < 16 && p0 + lane < n) {
      float best = Ow[lane * (K + 1)];
      int bi = 0;
      for (int o = 1; o <= K; ++o) {
        const float v = Ow[lane * (K + 1) + o];
        if (v > best) {
          best = v;
          bi = o;
        }
      }
      gst(labels + p0 + lane, (unsigned char)bi);
      atomicAdd(&hist[bi], 1);
    }
    const int rows = min(16, n - p0);
    for (int e = lane; e < rows * (K + 1); e += 64) gst(logits + (size_t)p0 * (K + 1) + e, Ow[e]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // scratch is rewritten by the next group
    grp = next;
  }
  __syncthreads();
  for (int e = threadIdx.x; e <= K; e += 512)
    if (hist[e]) atomicAdd(&counts[e], hist[e]);
}

// ---- the whole scene-to-click half of a layer in ONE pass (<= ~24 queries) --------------------------------------
// k_q_s2c + k_out_ln_mask without the round trip of the attention output O through HBM: per 16-point group the lane that
// holds O[point j][16h+4g..+3] after head h's P V product is holding exactly the B fragment of the output projection's
// K-step S = h, so y += Wo[S = h] O_h is accumulated head by head (y starts as bias + residual row) and LayerNorm, the
// mask head, label argmax and histogram follow from registers.  Reads src + pos, writes Y and the logits: 3 N d floats
// per layer instead of 6 N d (SURVEY 8(d)'s fused-ideal figure).  Both packed weight matrices (128 KB) are LDS
// resident, so the queries' keys / transposed values get compact rows (nq rounded up to 4) and the mask embeddings sit
// in registers; more queries than fit fall back to the two-kernel path.  The logits MFMA is taken transposed
// (queries x points): a lane ends up with the logits of ITS point, the per-object max is a masked in-lane max + one
// row reduction, no LDS scratch for the [16][Q] logits tile.
// NW = waves per workgroup: 8 (two per SIMD, the next group's rows prefetched into 64 registers) or 12 (three per SIMD at
// <= 168 registers: no register prefetch -- the third wave covers a wave's wait for its rows)
// KG: the queries' key rows come from global memory (16 KB, cache resident; requested a whole Q projection before their
// MFMAs) instead of LDS -- the build for 25..32 queries, where keys + transposed values + both weight matrices do not fit
// QC: the queries of the layer from the scene's cache (as k_q_s2c<.., QC>): the Q projection -- 256 of a group's ~700 MFMAs -- and
// the read of the position encodings are skipped
template <int QT, int NW, bool KG = false, bool QC = false>
__global__ void __launch_bounds__(NW * 64) k_s2c_out(const DecSampleDev* __restrict__ samples, int ns, int layer,
                                                 const float* __restrict__ Wq, const float* __restrict__ bq,
                                                 const float* __restrict__ Wo, const float* __restrict__ bo,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, int nqr_max,
                                                 int Kmax) {
  constexpr int LD = 136, NT = NW * 64;   // 136: the key fragments' ds_read_b128 (row 16 kt + j, floats 16 h + 4 g ..) are bank-
                                          // conflict-free ((8 j + 4 g) mod 64 distinct inside a 16-lane service group; 132 was 2-way)
  constexpr bool PF = NW == 8;   // register prefetch of the next group
  const DecSampleDev& sm = sample_of_wg(samples, ns);
  const int lb = blockIdx.x - sm.wg_begin, nwg = sm.wg_end - sm.wg_begin;
  const int n = sm.n, ngroups = (n + 15) / 16, nq = sm.nq, n_fg = sm.n_fg, K = sm.K;
  const float* __restrict__ X = layer_input(sm, layer);
  const float* __restrict__ Pe = sm.posenc;
  float* __restrict__ Y = (layer & 1) ? sm.bufD : sm.bufC;
  float* logits = sm.logits + (size_t)layer * n * (K + 1);
  unsigned char* labels = sm.labels;
  int* counts = sm.counts + (size_t)layer * (A3D_MAX_QUERIES + 1);
  const int LT = nqr_max + 4;                     // row stride of the transposed values
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* Wql = (f32x4*)smem;                      // [8 S][8 ct][64]
  f32x4* Wol = Wql + 8 * 8 * 64;                  // [8 S][8 ct][64]
  float* ks_l = (float*)(Wol + 8 * 8 * 64);       // [nqr_max][LD] (not with KG)
  float* vt_l = ks_l + (KG ? 0 : nqr_max * LD);   // [128][LT]
  float* bq_l = vt_l + D * LT;                    // [128] x 4: q bias, out bias, LayerNorm weight and bias
  float* bo_l = bq_l + D;
  float* ga_l = bo_l + D;
  float* be_l = ga_l + D;
  float* O_l = be_l + D;                          // [NW][16][Kmax+1] logits staging
  int* hist = (int*)(O_l + NW * 16 * (Kmax + 1)); // [Kmax+1]
  int* qr_l = hist + Kmax + 1;                    // [Kmax+2]
  float* sb_l = (float*)(qr_l + Kmax + 2);        // [QT*16] score bias: 0 for a real query, kNegBig for a padded one
  for (int e = threadIdx.x; e < QT * 16; e += NT) sb_l[e] = e < nq ? 0.f : kNegBig;
  for (int e = threadIdx.x; e <= K + 1; e += NT) qr_l[e] = gld(sm.qrange + e);
  for (int e = threadIdx.x; e <= K; e += NT) hist[e] = 0;
  if (threadIdx.x < D) {
    bq_l[threadIdx.x] = bq[threadIdx.x];
    bo_l[threadIdx.x] = bo[threadIdx.x];
    ga_l[threadIdx.x] = gamma[threadIdx.x];
    be_l[threadIdx.x] = beta[threadIdx.x];
  }
  {
    constexpr int TOT = 8 * 8 * 64;
    for (int base = threadIdx.x; base < TOT; base += 4 * NT) {
      f32x4 t8[8];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (base + u * NT < TOT) {
          t8[u] = ((const f32x4*)Wq)[base + u * NT];
          t8[4 + u] = ((const f32x4*)Wo)[base + u * NT];
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (base + u * NT < TOT) {
          Wql[base + u * NT] = t8[u];
          Wol[base + u * NT] = t8[4 + u];
        }
    }
  }
  for (int e = threadIdx.x; e < nqr_max * 32; e += NT) {
    const int r = e >> 5, c4 = (e & 31) * 4;
    f32x4 k4 = (f32x4){0.f, 0.f, 0.f, 0.f}, v4 = k4;
    if (r < nq) {
      k4 = gld4(sm.ks + (size_t)r * D + c4);
      v4 = gld4(sm.vs + (size_t)r * D + c4);
    }
    if constexpr (!KG) *(f32x4*)(ks_l + r * LD + c4) = k4;
#pragma unroll
    for (int t = 0; t < 4; ++t) vt_l[(c4 + t) * LT + r] = v4[t];
  }
  for (int e = threadIdx.x; e < D * 4; e += NT) vt_l[(e >> 2) * LT + nqr_max + (e & 3)] = 0.f;   // the pad columns are read (x 0)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  __syncthreads();
  float* Ow = O_l + wave * 16 * (Kmax + 1);
  const unsigned lds0 = (unsigned)(size_t)smem;                                         // 32-bit LDS address of Wql
  const unsigned ks_a0 = (unsigned)(size_t)ks_l + (unsigned)(j * LD + 4 * g) * 4u;      // key row j, floats 4 g ..
  const unsigned vt_a0 = (unsigned)(size_t)vt_l + (unsigned)(j * LT + 4 * g) * 4u;      // value channel j, keys 4 g ..
  auto lds4 = [](unsigned addr) -> f32x4 { return *(const __attribute__((address_space(3))) f32x4*)addr; };
  const int stride = nwg * NW;
  int grp = lb * NW + wave;
  // On gfx950 a VALU instruction takes the fp32 matrix pipe's issue time (tools/coissue_ubench.hip: 24 MFMAs + 24 v_fma run
  // 16 % longer than the MFMAs alone, packed or not), so what this lane would otherwise recompute per group is fixed here:
  //   sb_l:  the score accumulators start at 0 for a real query and at kNegBig for a padded one (no compare / select per
  //          score and head; read from LDS: eight more registers spill the twelve-wave build);
  //   oidp:  the object of each of this lane's QT x 4 query slots, one byte each (255 = padded): the per-object maximum
  //          of the logits is one SDWA byte compare + select + max per slot.
  unsigned oidp[QT];
#pragma unroll
  for (int kt = 0; kt < QT; ++kt) {
    oidp[kt] = 0u;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int q = kt * 16 + 4 * g + t;
      unsigned o = 255u;
      if (q < nq) {
        o = 0u;                                   // background: [n_fg, nq)
        for (int oo = 1; oo <= K; ++oo)
          if (q >= qr_l[oo] && q < qr_l[oo + 1]) o = (unsigned)oo;
      }
      oidp[kt] |= o << (8 * t);
    }
  }
  f32x4 nx[8], np[8];
  auto fetch = [&](int gq) {
    const size_t row = (size_t)min(gq * 16 + j, n - 1);
    const float* xr = X + row * D + 4 * g;
    const float* pr = (QC ? sm.q0 : Pe) + row * D + 4 * g;   // QC: head h's fragment of the cached queries sits where pos's S = h does
#pragma unroll
    for (int S = 0; S < 8; ++S) nx[S] = gld4(xr + 16 * S);
#pragma unroll
    for (int S = 0; S < 8; ++S) np[S] = gld4(pr + 16 * S);
  };
  if (PF && grp < ngroups) fetch(grp);
  while (grp < ngroups) {
    if constexpr (!PF) fetch(grp);
    const int p0 = grp * 16;
    const int prow = min(p0 + j, n - 1);
    f32x4 xp[8], y[8];
#pragma unroll
    for (int S = 0; S < 8; ++S) {
      xp[S] = QC ? np[S] : nx[S] + np[S];
      y[S] = *(const f32x4*)(bo_l + 16 * S + 4 * g) + nx[S];   // bias + residual row: channels 16 ct + 4 g ..+3 of point j
    }
    const int next = grp + stride;
    // heads two at a time: the Q projection of a head is a chain of 32 MFMAs on ONE accumulator and the P V product a chain
    // of 4 QT (40 cycles of dependent latency per 32-cycle MFMA); two heads' chains interleaved keep the matrix pipe fed
#pragma unroll 1
    for (int h = 0; h < H; h += 2) {
      // the next group's rows are requested late in the head loop (two heads + the LayerNorm / logits phase cover the
      // latency): 64 registers that would otherwise be live next to xp and y for the whole group
      if (PF && h == 4 && next < ngroups) fetch(next);
      // 32-bit LDS addresses of this head pair, each pinned in ONE register: every read below is base + a 16-bit
      // immediate.  (The weight matrices sit 64 / 128 KB into the LDS, beyond the ds_read offset field, and h is a run-time
      // value: left alone the compiler spends an address add per read -- 38 of the pair's 133 vector instructions, and a
      // vector instruction costs matrix-pipe time.)
      unsigned wq_a = lds0 + (unsigned)(h * 64 + lane) * 16u;                          // + (S * 512 + u * 64) * 16
      unsigned wo_a = lds0 + 65536u + (unsigned)(h * 512 + lane) * 16u;               // + (u * 512 + ct * 64) * 16
      unsigned ks_a = KG ? 0u : ks_a0 + (unsigned)(h * DH) * 4u;                       // + (kt * 16 * LD + u * DH) * 4
      unsigned vt_a = vt_a0 + (unsigned)(h * DH * LT) * 4u;                            // + kt * 64 (u = 1: vt_a + DH * LT * 4)
      asm volatile("" : "+v"(wq_a), "+v"(wo_a), "+v"(vt_a));
      if constexpr (!KG) asm volatile("" : "+v"(ks_a));
      const unsigned vt_a1 = vt_a + (unsigned)(DH * LT) * 4u;
      f32x4 qf[2];
      if constexpr (QC) {   // h is a run-time value (the loop is not unrolled): three selects per component instead of an indexed array
        qf[0] = h == 0 ? xp[0] : h == 2 ? xp[2] : h == 4 ? xp[4] : xp[6];
        qf[1] = h == 0 ? xp[1] : h == 2 ? xp[3] : h == 4 ? xp[5] : xp[7];
      } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) qf[u] = *(const f32x4*)(bq_l + 16 * (h + u) + 4 * g);   // Q[point j][16h+4g..+3]
      }
      f32x4 kfg[2][QT];
      if constexpr (KG) {
#pragma unroll
        for (int kt = 0; kt < QT; ++kt)
#pragma unroll
          for (int u = 0; u < 2; ++u) kfg[u][kt] = gld4(sm.ks + (size_t)min(kt * 16 + j, nq - 1) * D + (h + u) * DH + 4 * g);
      }
      if constexpr (!QC) {
        // weight fragments one K-step ahead of the MFMAs that use them (left to itself the compiler issues the two
        // ds_read_b128 of a step right in front of its eight MFMAs and waits out the LDS latency every 256 cycles)
        f32x4 w0 = lds4(wq_a), w1 = lds4(wq_a + 64 * 16);
#pragma unroll
        for (int S = 0; S < 8; ++S) {
          f32x4 n0 = w0, n1 = w1;
          if (S + 1 < 8) {
            n0 = lds4(wq_a + (S + 1) * 512 * 16);
            n1 = lds4(wq_a + ((S + 1) * 512 + 64) * 16);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            qf[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[t], xp[S][t], qf[0], 0, 0, 0);
            qf[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[t], xp[S][t], qf[1], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          w0 = n0;
          w1 = n1;
        }
      }
      f32x4 sc[2][QT];
      float mx[2] = {kNegBig, kNegBig};
      {
        f32x4 kf[2][QT];
#pragma unroll
        for (int kt = 0; kt < QT; ++kt)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if constexpr (KG) kf[u][kt] = kfg[u][kt];
            else kf[u][kt] = lds4(ks_a + (kt * 16 * LD + u * DH) * 4);
            sc[u][kt] = *(const f32x4*)(sb_l + kt * 16 + 4 * g);
          }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int kt = 0; kt < QT; ++kt)
#pragma unroll
            for (int u = 0; u < 2; ++u) sc[u][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][kt][t], qf[u][t], sc[u][kt], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int kt = 0; kt < QT; ++kt)
#pragma unroll
          for (int t = 0; t < 4; ++t) mx[u] = fmaxf(mx[u], sc[u][kt][t]);   // padded queries sit at kNegBig (sb_l)
      float inv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        mx[u] = rows_max(mx[u]);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < QT; ++kt)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            sc[u][kt][t] = exp2_fast(sc[u][kt][t] - mx[u]);
            sum += sc[u][kt][t];
          }
        sum = rows_sum(sum);
        inv[u] = __builtin_amdgcn_rcpf(sum);
      }
      f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
      // the first output-projection fragments are requested before the P V products, every later pair one step ahead
      f32x4 wo0 = lds4(wo_a), wo1 = lds4(wo_a + 64 * 16);
      {
        f32x4 vf[2][QT];
#pragma unroll
        for (int kt = 0; kt < QT; ++kt)
#pragma unroll
          for (int u = 0; u < 2; ++u) vf[u][kt] = lds4((u ? vt_a1 : vt_a) + kt * 64);   // V^T: keys 4g..4g+3 of channel 16h+j
#pragma unroll
        for (int kt = 0; kt < QT; ++kt)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[u][kt][t], sc[u][kt][t], acc[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] *= inv[u];   // the softmax normaliser once per output instead of once per weight
      // acc[u] = O[point j][16(h+u)+4g..+3] = the B fragment of the output projection's K-step h + u; two column tiles at a
      // time (consecutive MFMAs never hit the same accumulator)
#pragma unroll
      for (int i = 0; i < 8; ++i) {          // i = 4 u + ct / 2
        const int u = i >> 2, ct = 2 * (i & 3);
        f32x4 n0 = wo0, n1 = wo1;
        if (i + 1 < 8) {
          const int u2 = (i + 1) >> 2, ct2 = 2 * ((i + 1) & 3);
          n0 = lds4(wo_a + (u2 * 512 + ct2 * 64) * 16);
          n1 = lds4(wo_a + (u2 * 512 + ct2 * 64 + 64) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          y[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wo0[t], acc[u][t], y[ct], 0, 0, 0);
          y[ct + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wo1[t], acc[u][t], y[ct + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        wo0 = n0;
        wo1 = n1;
      }
    }
    // LayerNorm over the 128 channels of point j (4 g-lanes x 8 ct x 4)
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) sum += y[ct][0] + y[ct][1] + y[ct][2] + y[ct][3];
    sum = rows_sum(sum);
    const float mean = sum * (1.f / D);
    float var = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float d = y[ct][t] - mean;
        var += d * d;
      }
    var = rows_sum(var);
    const float rstd = rsqrtf(var * (1.f / D) + kLnEps);
    float* yrow = Y + (size_t)prow * D;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const f32x4 ga = *(const f32x4*)(ga_l + 16 * ct + 4 * g);
      const f32x4 be = *(const f32x4*)(be_l + 16 * ct + 4 * g);
#pragma unroll
      for (int t = 0; t < 4; ++t) y[ct][t] = (y[ct][t] - mean) * rstd * ga[t] + be[t];
      if (p0 + j < n) gst4(yrow + 16 * ct + 4 * g, y[ct]);
    }
    // mask embeddings of the queries as A fragments of the transposed logits product, E[query 16 qt + j][16 S + 4 g ..+3]:
    // re-read per group (16 KB, cache resident; requested behind the next group's rows, which went out three heads ago) --
    // held in registers for the whole group they cost 64 registers next to xp, y and the prefetch (spills)
    f32x4 ef[QT][8];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const float* er = sm.E + (size_t)min(qt * 16 + j, nq - 1) * D + 4 * g;
#pragma unroll
      for (int S = 0; S < 8; ++S) ef[qt][S] = gld4(er + 16 * S);
    }
    // logits^T: lg[qt][t] = logit of query 16 qt + 4 g + t for point j (rows >= nq repeat the last query: never selected)
    // (even / odd K-steps on separate accumulators, query tiles interleaved: no back-to-back dependent MFMAs)
    f32x4 lg[QT], lg2[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) lg[qt] = lg2[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int S = 0; S < 8; S += 2)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          lg[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ef[qt][S][t], y[S][t], lg[qt], 0, 0, 0);
          lg2[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ef[qt][S + 1][t], y[S + 1][t], lg2[qt], 0, 0, 0);
        }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) lg[qt] += lg2[qt];
    // per-object max over the object's queries (k_out_ln_mask's order of comparisons: ascending query index; max is
    // order-independent), first-max argmax over the objects
    float best = 0.f;
    int bi = 0;
    for (int o = 0; o <= K; ++o) {
      float mxv = -3.4e38f;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (((oidp[qt] >> (8 * t)) & 255u) == (unsigned)o) mxv = fmaxf(mxv, lg[qt][t]);
      mxv = rows_max(mxv);
      if (g == (o & 3)) Ow[j * (K + 1) + o] = mxv;
      if (o == 0 || mxv > best) {
        best = mxv;
        bi = o;
      }
    }
    if (g == 0 && p0 + j < n) {
      gst(labels + p0 + j, (unsigned char)bi);
      atomicAdd(&hist[bi], 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private staging: written above, read below by other lanes
    const int rows = min(16, n - p0);
    for (int e = lane; e < rows * (K + 1); e += 64) gst(logits + (size_t)p0 * (K + 1) + e, Ow[e]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // rewritten by the next group
    grp = next;
  }
  __syncthreads();
  for (int e = threadIdx.x; e <= K; e += NT)
    if (hist[e]) atomicAdd(&counts[e], hist[e]);
}

// ------------------------------------------------------------------------------ query side
struct QueryLayerW {
  const float *c2s_in_wt, *c2s_in_b, *c2s_out_wt, *c2s_out_b, *c2s_norm_w, *c2s_norm_b;
  const float *c2c_in_wt, *c2c_in_b, *c2c_out_wt, *c2c_out_b, *c2c_norm_w, *c2c_norm_b;
  const float *ffn_w1t, *ffn_b1, *ffn_w2t, *ffn_b2, *ffn_norm_w, *ffn_norm_b;
  const float *s2c_in_wt, *s2c_in_b;
  const float *dn_w, *dn_b, *m_w0t, *m_b0, *m_w2t, *m_b2;
  const float *next_c2s_in_wt, *next_c2s_in_b;   // nullptr on the last layer
  const float *qpack, *next_qpack, *mpack;       // fragment-order copies (k_query_block): this layer's, the next layer's, the mask head's
  int dim_ff;
  int layer;
};


// Y[q][n] = ((X[q][:] (+ Xadd[q][:])) . W[n][:] + bias[n]) * scale, optional relu -- a skinny GEMM
// on the matrix cores.  W is the torch weight [N][ldw] (K contiguous): lane (g, j) loads
// W[n0 + j][16 S + 4 g .. +3] as one float4 = the B fragments of four MFMAs (same K permutation as
// spconv.hip).  X (<= 64 rows, global memory) is staged through LDS in 256-column slices (row stride 260 floats:
// conflict-free b128 A-fragment reads; vector loads only -- X was written earlier in this kernel).
// 8 waves; wave w owns output column tiles w, w+8, ...
constexpr int kLinKC = 256, kLinLD = kLinKC + 4;
template <int QT>
__device__ __noinline__ void lin(const float* X, int ldx, const float* Xadd, int Q, int K, const float* __restrict__ W,
                                 int ldw, const float* __restrict__ bias, int N, float* Y, int ldy, bool relu,
                                 float scale, float* lds /*[QT*16][260]*/) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int ntiles = N >> 4;
  for (int t0 = 0; t0 < ntiles; t0 += nw) {
    const int ntile = t0 + wave;
    const bool active = ntile < ntiles;
    f32x4 acc[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) acc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < K; kc += kLinKC) {
      const int kcur = min(kLinKC, K - kc);
      __syncthreads();
      for (int e = tid; e < QT * 16 * (kcur >> 2); e += nt) {
        const int q = e / (kcur >> 2), k4 = (e - q * (kcur >> 2)) * 4;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (q < Q) {
          v = gld4(X + (size_t)q * ldx + kc + k4);
          if (Xadd) v += gld4(Xadd + (size_t)q * D + kc + k4);
        }
        *(f32x4*)(lds + q * kLinLD + k4) = v;
      }
      __syncthreads();
      if (active) {
        const float* wrow = W + (size_t)(ntile * 16 + j) * ldw + kc + 4 * g;
#pragma unroll 4
        for (int S = 0; S < (kcur >> 4); ++S) {
          const f32x4 b = gld4(wrow + 16 * S);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            const f32x4 a = *(const f32x4*)(lds + (qt * 16 + j) * kLinLD + 16 * S + 4 * g);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc[qt], 0, 0, 0);
          }
        }
      }
    }
    if (active) {
      const int col = ntile * 16 + j;
      const float b = bias ? gld(bias + col) : 0.f;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int q = qt * 16 + 4 * g + t;
          if (q < Q) {
            float y = (acc[qt][t] + b) * scale;
            if (relu) y = fmaxf(y, 0.f);
            gst(Y + (size_t)q * ldy + col, y);
          }
        }
    }
  }
  __syncthreads();
}

// out[q] = LayerNorm(a[q] + b[q]) * w + bias ; one wave per row
__device__ __noinline__ void add_ln(const float* a, const float* b, int Q, const float* __restrict__ w,
                       const float* __restrict__ bias, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int q = wave; q < Q; q += nw) {
    float x0 = a[(size_t)q * D + lane], x1 = a[(size_t)q * D + 64 + lane];
    if (b) {
      x0 += b[(size_t)q * D + lane];
      x1 += b[(size_t)q * D + 64 + lane];
    }
    float s = x0 + x1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * (1.f / D);
    const float d0 = x0 - mean, d1 = x1 - mean;
    float v = d0 * d0 + d1 * d1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const float rstd = rsqrtf(v * (1.f / D) + kLnEps);
    out[(size_t)q * D + lane] = d0 * rstd * w[lane] + bias[lane];
    out[(size_t)q * D + 64 + lane] = d1 * rstd * w[64 + lane] + bias[64 + lane];
  }
  __syncthreads();
}

template <int QT>
__global__ void __launch_bounds__(512) k_query_init(const QuerySample* __restrict__ qs, const float* bg_feat,
                                                     const float* bg_pos, const float* time_table,
                                                     const float* c2s_in_wt, const float* c2s_in_b, int n_counts) {
  constexpr int QP = QT * 16;
  const QueryMeta* meta = qs[blockIdx.y].meta;
  const float* feats128 = qs[blockIdx.y].feats;
  const float* posenc = qs[blockIdx.y].posenc;
  QueryBufs B = qs[blockIdx.y].B;
  int* counts = qs[blockIdx.y].counts;
  __shared__ __attribute__((aligned(16))) float lds[QP * kLinLD];
  const int q0 = blockIdx.x * QP;                 // this workgroup's block of queries
  if (q0 >= qs[blockIdx.y].qp) return;            // the grid serves the longest query list of the call
  const int Q = max(0, min(QP, meta->nq - q0)), n_fg = meta->n_fg, n_bgl = meta->n_bgl;
  B.queries += (size_t)q0 * D; B.qpos += (size_t)q0 * D; B.qproj += (size_t)q0 * D;
  B.ks += (size_t)q0 * D; B.vs += (size_t)q0 * D; B.E += (size_t)q0 * D;
  if (blockIdx.x == 0) {
    for (int e = threadIdx.x; e < n_counts; e += blockDim.x) gst(counts + e, 0);
    // the hand-off flags of this sample's k_query_block launches (every layer, every query block)
    for (int e = threadIdx.x; e < A3D_MAX_DEC_LAYERS * kMaxQBlocks * 16; e += blockDim.x) B.sync[e] = 0u;
  }
  const int rows_here = min(QP, qs[blockIdx.y].qp - q0);   // the last block of a 16- or 48-row buffer is short
  for (int e = threadIdx.x; e < rows_here * D; e += blockDim.x) {
    const int c = e & 127;
    const int q = q0 + (e >> 7);
    float f = 0.f, p = 0.f;
    if (q < meta->nq) {
      const int r = meta->row[q];
      if (r >= 0) {   // clicked query: feature + Fourier(click xyz) + time encoding (agile3d.py:213-264)
        f = gld(feats128 + (size_t)r * D + c);
        p = gld(posenc + (size_t)r * D + c) + time_table[(size_t)meta->time[q] * D + c];
      } else {        // learned background query
        const int b = q - n_fg;
        f = bg_feat[(size_t)b * D + c];
        p = bg_pos[(size_t)b * D + c];
      }
    }
    gst(B.queries + e, f);
    gst(B.qpos + e, p);
    gst(B.qproj + e, 0.f);
    gst(B.ks + e, 0.f);
    gst(B.vs + e, 0.f);
    gst(B.E + e, 0.f);
  }
  (void)n_bgl;
  __syncthreads();
  // c2s query projection of the first layer, pre-scaled by 1/sqrt(head_dim)
  lin<QT>(B.queries, D, B.qpos, Q, D, c2s_in_wt, D, c2s_in_b, D, B.qproj, D, false, kPointScoreScale, lds);
}

// ---- query-side layer with all [Q,128] activations resident in LDS ---------------------------
// 8 waves; in every GEMM round wave w owns output column tile(s) w (+8, ...); activations never leave LDS between
// rounds (row stride 132 floats: conflict-free b128 A-fragment reads).
constexpr int kQLD = 132;

template <int QT>
__device__ __forceinline__ void qmm(const float* Xl, const f32x4 (&wf)[8], f32x4 (&acc)[QT]) {
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
#pragma unroll
  for (int S = 0; S < 8; ++S)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const f32x4 x = *(const f32x4*)(Xl + (qt * 16 + j) * kQLD + 16 * S + 4 * g);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[t], wf[S][t], acc[qt], 0, 0, 0);
    }
}
template <int QT>
__device__ __forceinline__ void qzero(f32x4 (&acc)[QT]) {
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) acc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
// ---- the single-block layer (nq <= 64), second build ---------------------------------------------------------
// The first build (k_query_layer, removed in round 4) spent its time WAITING.  Measured on it
// (20 queries, phase marks): 62 us of which the MFMAs are ~6 -- every phase ended in an on-demand global load (a bias, a
// LayerNorm vector) whose vmcnt wait also drained the weight prefetch behind it, a 6-step LDS-shuffle reduction per
// LayerNorm row, and a scalar click-to-click attention on 2.5 waves.  Here:
//   * the 20 bias / LayerNorm vectors go to LDS once, next to the activations, in the load phase;
//   * four weight-fragment sets rotate, so every GEMM's weights were requested at least one full phase earlier and the
//     only vector-memory traffic of a phase is the prefetch;
//   * LayerNorm takes 16 rows per wave in the MFMA fragment order (lane (g, j): row j, channels 16 S + 4 g ..+3): a row's
//     sums are in-lane adds + two permlane swaps, and the writer also emits what the next phase stages (tgt + qpos, the
//     hand-off copy for the helper workgroups);
//   * click-to-click attention runs on the matrix cores, one head per wave (S^T = K Q^T, softmax over the keys in the
//     C layout, O^T = V^T P^T -- the chain of k_s2c_out).
// Fragment-order copies of the query side's matrices (a3d_decoder_pack_query_weights).  A [N][K] torch matrix becomes
// [N/16 tiles][K/16 steps][64 lanes] float4 with lane (g, j) of (tile t, step S) = W[16t + j][16S + 4g ..+3]: what qload_w
// gathers from 16 rows (64 B each) is one contiguous 1 KB here -- 35 -> 140 GB/s into one CU (tools/qload_ubench.hip).
constexpr size_t kQpC2sOut = 0, kQpC2cIn = 16384, kQpC2cOut = 65536, kQpS2cKV = 81920, kQpC2sQ = 114688, kQpFfn1 = 131072;
__host__ __device__ constexpr size_t qpack_ffn2(int dim_ff) { return kQpFfn1 + (size_t)dim_ff * 128; }
__host__ __device__ constexpr size_t qpack_floats(int dim_ff) { return kQpFfn1 + (size_t)2 * dim_ff * 128; }
constexpr size_t kMpW0 = 0, kMpW2 = 16384, kMpFloats = 32768;
__global__ void __launch_bounds__(256) k_pack_rows(const float* __restrict__ W, int N, int K, float* __restrict__ out) {
  const int KS = K >> 4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // one float4 of the output
  if (i >= (size_t)(N >> 4) * KS * 64) return;
  const int lane = (int)(i & 63), g = lane >> 4, j = lane & 15;
  const size_t ts = i >> 6;
  const int S = (int)(ts % KS), t = (int)(ts / KS);
  ((f32x4*)out)[i] = *(const f32x4*)(W + (size_t)(16 * t + j) * K + 16 * S + 4 * g);
}
// wf[S] = fragment (tile, S0 + S) of a packed matrix with KS steps per tile
__device__ __forceinline__ void qload_p(const float* P, int KS, int tile, int S0, f32x4 (&wf)[8]) {
  const f32x4 A3D_GLOBAL* p = (const f32x4 A3D_GLOBAL*)P + ((size_t)tile * KS + S0) * 64 + (threadIdx.x & 63);
#pragma unroll
  for (int S = 0; S < 8; ++S) wf[S] = p[S * 64];
}
constexpr int kQVec = 2560;   // floats of the vector table
enum {
  V_C2S_OUT_B = 0, V_C2S_NW = 128, V_C2S_NB = 256, V_C2C_IN_B = 384, V_C2C_OUT_B = 768, V_C2C_NW = 896, V_C2C_NB = 1024,
  V_FFN_B2 = 1152, V_FFN_NW = 1280, V_FFN_NB = 1408, V_S2C_IN_B = 1536, V_DN_W = 1920, V_DN_B = 2048, V_M_B0 = 2176,
  V_M_B2 = 2304, V_NEXT_B = 2432
};
__device__ __forceinline__ const float* ql_vec_src(const QueryLayerW& W, int s) {   // 128-float segment s of the table
  switch (s) {
    case 0: return W.c2s_out_b;
    case 1: return W.c2s_norm_w;
    case 2: return W.c2s_norm_b;
    case 3: return W.c2c_in_b;
    case 4: return W.c2c_in_b + 128;
    case 5: return W.c2c_in_b + 256;
    case 6: return W.c2c_out_b;
    case 7: return W.c2c_norm_w;
    case 8: return W.c2c_norm_b;
    case 9: return W.ffn_b2;
    case 10: return W.ffn_norm_w;
    case 11: return W.ffn_norm_b;
    case 12: return W.s2c_in_b;
    case 13: return W.s2c_in_b + 128;
    case 14: return W.s2c_in_b + 256;
    case 15: return W.dn_w;
    case 16: return W.dn_b;
    case 17: return W.m_b0;
    case 18: return W.m_b2;
    default: return W.next_c2s_in_b;
  }
}
// y[q][col0 + j] = act((acc + b) * scale) for q < Q, the lane's bias value already in a register
template <int QT, bool G = false>
__device__ __forceinline__ void qstore_b(const f32x4 (&acc)[QT], float b, float scale, bool relu, float* dst, int ld,
                                         int col0, int Q) {
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int q = qt * 16 + 4 * g + t;
      if (q < Q) {
        float y = (acc[qt][t] + b) * scale;
        if (relu) y = fmaxf(y, 0.f);
        if constexpr (G) gst(dst + (size_t)q * ld + col0 + j, y);
        else dst[(size_t)q * ld + col0 + j] = y;
      }
    }
}
// dst[q] = LayerNorm(a[q] + b[q]) (b optional) for the rows q < Q of [QP][132] LDS tiles, waves 0 .. QT-1, 16 rows each.
// Optional extra outputs of the same rows: dst2 = y + add2 (LDS), gdst = y (global), gagent = y (global, agent scope: the
// hand-off to the helper workgroups).  dst / dst2 may alias a / b: a lane only rewrites the elements it read.
template <int QT>
__device__ __forceinline__ void qln16(const float* a, const float* b, int Q, const float* w_l, const float* b_l, float* dst,
                                      float* dst2, const float* add2, float* gdst, float* gagent) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  if (wave >= QT) return;
  const int q = 16 * wave + j;
  f32x4 x[8];
  float s = 0.f;
#pragma unroll
  for (int S = 0; S < 8; ++S) {
    x[S] = *(const f32x4*)(a + q * kQLD + 16 * S + 4 * g);
    if (b) x[S] += *(const f32x4*)(b + q * kQLD + 16 * S + 4 * g);
    s += (x[S][0] + x[S][1]) + (x[S][2] + x[S][3]);
  }
  s = rows_sum(s);
  const float mean = s * (1.f / D);
  float v = 0.f;
#pragma unroll
  for (int S = 0; S < 8; ++S)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      x[S][t] -= mean;
      v += x[S][t] * x[S][t];
    }
  v = rows_sum(v);
  const float rstd = rsqrtf(v * (1.f / D) + kLnEps);
  if (q >= Q) return;
#pragma unroll
  for (int S = 0; S < 8; ++S) {
    const int c = 16 * S + 4 * g;
    const f32x4 y = x[S] * rstd * *(const f32x4*)(w_l + c) + *(const f32x4*)(b_l + c);
    *(f32x4*)(dst + q * kQLD + c) = y;
    if (dst2) *(f32x4*)(dst2 + q * kQLD + c) = y + *(const f32x4*)(add2 + q * kQLD + c);
    if (gdst) gst4(gdst + (size_t)q * D + c, y);
    if (gagent) {
#pragma unroll
      for (int t = 0; t < 4; ++t) gst_agent(gagent + (size_t)q * D + c + t, y[t]);
    }
  }
}
// click-to-click attention of head h = wave on the matrix cores, all operands [QP][132] LDS tiles: q_l holds the
// pre-scaled queries on entry and the attention output on exit (a wave rewrites only its own head's 16 columns, row tile
// by row tile after reading them).  Keys >= Qall are masked; output rows >= Q are zero.
template <int QT>
__device__ __forceinline__ void qattn_mfma(float* q_l, const float* k_l, const float* v_l, int Q, int Qall) {
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  f32x4 kf[QT], vf[QT];
#pragma unroll
  for (int kt = 0; kt < QT; ++kt) {
    kf[kt] = *(const f32x4*)(k_l + (16 * kt + j) * kQLD + 16 * h + 4 * g);   // K[key 16kt+j][16h+4g..+3]
#pragma unroll
    for (int t = 0; t < 4; ++t) vf[kt][t] = v_l[(16 * kt + 4 * g + t) * kQLD + 16 * h + j];   // V^T[16h+j][key 16kt+4g+t]
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const f32x4 qf = *(const f32x4*)(q_l + (16 * qt + j) * kQLD + 16 * h + 4 * g);
    f32x4 sc[QT];
    float mx = kNegBig;
#pragma unroll
    for (int kt = 0; kt < QT; ++kt) {
      sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][t], qf[t], sc[kt], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (16 * kt + 4 * g + t >= Qall) sc[kt][t] = kNegBig;
        mx = fmaxf(mx, sc[kt][t]);
      }
    }
    mx = rows_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < QT; ++kt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sc[kt][t] = fast_exp(sc[kt][t] - mx);
        sum += sc[kt][t];
      }
    sum = rows_sum(sum);
    const float inv = 1.f / sum;
    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < QT; ++kt)
#pragma unroll
      for (int t = 0; t < 4; ++t) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][t], sc[kt][t] * inv, o, 0, 0, 0);
    // o[t] = O[query 16qt+j][16h+4g+t]
    if (16 * qt + j >= Q) o = (f32x4){0.f, 0.f, 0.f, 0.f};
    *(f32x4*)(q_l + (16 * qt + j) * kQLD + 16 * h + 4 * g) = o;
  }
}

// The same attention with the keys / values of ALL query blocks in global memory (PART 2 of the multi-block layer): flash
// over 16-key tiles, the next tile's fragments requested before the current tile's products.  qk: this block's rows of
// [q (pre-scaled) | k]; all_qk / all_vc: every block's rows.  Keys >= Qall are masked (their V lanes zeroed: the rows
// were never written).
template <int QT>
__device__ __forceinline__ void qattn_mfma_global(float* o_l, const float* qk, const float* all_qk, const float* all_vc,
                                                  int Q, int Qall) {
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  f32x4 qf[QT], o[QT];
  float m[QT], l[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qf[qt] = gld4(qk + (size_t)(16 * qt + j) * 2 * D + 16 * h + 4 * g);
    o[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m[qt] = kNegBig;
    l[qt] = 0.f;
  }
  const int nkt = (Qall + 15) >> 4;
  auto fetch = [&](int kt, f32x4& kf, f32x4& vf) {
    kf = gld4(all_qk + (size_t)(16 * kt + j) * 2 * D + D + 16 * h + 4 * g);          // K[key 16kt+j][16h+4g..+3]
#pragma unroll
    for (int t = 0; t < 4; ++t) vf[t] = gld(all_vc + (size_t)(16 * kt + 4 * g + t) * D + 16 * h + j);   // V^T[16h+j][key]
  };
  f32x4 kn, vn;
  fetch(0, kn, vn);
  for (int kt = 0; kt < nkt; ++kt) {
    f32x4 kf = kn, vf = vn;
    if (kt + 1 < nkt) fetch(kt + 1, kn, vn);
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (16 * kt + 4 * g + t >= Qall) vf[t] = 0.f;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t], qf[qt][t], sc, 0, 0, 0);
      float mx = kNegBig;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (16 * kt + 4 * g + t >= Qall) sc[t] = kNegBig;
        mx = fmaxf(mx, sc[t]);
      }
      mx = rows_max(mx);
      const float mnew = fmaxf(m[qt], mx);
      const float scl = fast_exp(m[qt] - mnew);
      m[qt] = mnew;
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sc[t] = fast_exp(sc[t] - mnew);
        ps += sc[t];
      }
      l[qt] = l[qt] * scl + ps;
      o[qt] *= scl;
#pragma unroll
      for (int t = 0; t < 4; ++t) o[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[t], sc[t], o[qt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float lt = rows_sum(l[qt]);
    f32x4 r = o[qt] * (1.f / lt);
    if (16 * qt + j >= Q) r = (f32x4){0.f, 0.f, 0.f, 0.f};
    *(f32x4*)(o_l + (16 * qt + j) * kQLD + 16 * h + 4 * g) = r;
  }
}

// More than 64 queries: blocks of 64 rows, one workgroup (+ helpers) each, in two launches around the click-to-click
// attention, which needs every block's keys / values: PART 1 = steps 1-2 up to the q / k / v projections (to B.qk, B.vc; tgt
// to B.tgt), PART 2 = the attention (flash over all blocks' keys, read from global memory) and everything after it.
// PART 0 = the whole layer (one block).  Grid (helpers + 1, blocks, samples).
template <int QT, int PART>
__global__ void __launch_bounds__(512) k_query_block(const QuerySample* __restrict__ qs, QueryLayerW W) {
  constexpr int QP = QT * 16;
  const QueryMeta* meta = qs[blockIdx.z].meta;
  QueryBufs B = qs[blockIdx.z].B;
  const int blk = (int)blockIdx.y, q0 = blk * QP;
  const float* all_qk = B.qk;
  const float* all_vc = B.vc;
  B.queries += (size_t)q0 * D; B.qpos += (size_t)q0 * D; B.qproj += (size_t)q0 * D; B.attn += (size_t)q0 * D;
  B.ks += (size_t)q0 * D; B.vs += (size_t)q0 * D; B.E += (size_t)q0 * D; B.tgt += (size_t)q0 * D;
  B.qk += (size_t)q0 * 2 * D; B.vc += (size_t)q0 * D;
  B.hidden += (size_t)blk * kQlMaxHelpers * QP * D;
  // workgroup 0 runs the layer; workgroups 1.. are FFN helpers (hidden chunks hx, hx + nh, ... of the 1024-wide FFN, whose
  // 1 MB of weights one workgroup alone would pull through one CU), helpers 1..3 then take one projection of the layer's new
  // queries each; hand-off through global memory with agent-scope loads / stores + flags
  const int nh = (int)gridDim.x, hx = (int)blockIdx.x;
  const int Qall = gld(&meta->nq);
  if (PART != 0 && q0 >= Qall) return;   // a block of another sample's longer query list: no rows, nothing anybody reads
  const int Q = max(0, min(QP, Qall - q0));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qpos = (float*)smem;            // [QP][132]  query position encodings (c2c values in between; mask MLP hidden)
  float* cur = qpos + QP * kQLD;         // queries -> tgt -> queries
  float* xa = cur + QP * kQLD;           // GEMM input staging
  float* xb = xa + QP * kQLD;            // GEMM output staging / FFN hidden chunk
  float* vec_l = xb + QP * kQLD;         // [kQVec] biases and LayerNorm vectors
  int* any_slot = (int*)(vec_l + kQVec);   // one word behind the table (block_any)
  const int tid = threadIdx.x, nt = 512;
  const int wave = tid >> 6, lane = tid & 63, j = lane & 15;
  const int nchunk = W.dim_ff >> 7;
  unsigned* flags = B.sync + (W.layer * kMaxQBlocks + blk) * 16;
  f32x4 wfa[8], wfb[8], wfc[8], wfd[8], acc[QT];
  constexpr bool kDeep = QT <= 2;   // the fourth fragment set in flight across the load phase and the attention (registers)

  if (PART != 1 && hx > 0) {   // ---- FFN helper
    if (hx >= nchunk) return;
    // everything this workgroup will multiply by is requested before it starts to wait
    qload_p(W.qpack + kQpFfn1, 8, hx * 8 + wave, 0, wfa);
    qload_p(W.qpack + qpack_ffn2(W.dim_ff), W.dim_ff >> 4, wave, hx * 8, wfb);
    float fb1 = gld(W.ffn_b1 + hx * 128 + 16 * wave + j);
    const bool second = nh >= 4 && nchunk >= 4 && hx <= 3 && (hx < 3 || W.next_c2s_in_wt);
    float sb = 0.f;
    if (second) {
      const float* wsrc = hx == 1 ? W.qpack + kQpS2cKV : hx == 2 ? W.qpack + kQpS2cKV + 16384 : W.next_qpack + kQpC2sQ;
      const float* bsrc = hx == 1 ? W.s2c_in_b + D : hx == 2 ? W.s2c_in_b + 2 * D : W.next_c2s_in_b;
      qload_p(wsrc, 8, wave, 0, wfc);
      sb = gld(bsrc + 16 * wave + j);
    }
    // a wait that gives up (seconds: never in a healthy run; the device must not hang) poisons what this workgroup hands
    // on: the layer's result is then NaN, not a plausible number computed from data that never arrived
    int late = 0;
    if (tid == 0) {
      unsigned spins = 0;
      while (gld_agent(flags) == 0u) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 26)) {
          late = 1;
          break;
        }
      }
    }
    const float bad = block_any(late, any_slot) ? __builtin_nanf("") : 0.f;
    for (int e = tid; e < QP * 128; e += nt) {
      const int q = e >> 7, c = e & 127;
      cur[q * kQLD + c] = q < Q ? gld_agent(B.tgt + (size_t)q * D + c) : 0.f;
    }
    __syncthreads();
    f32x4 facc[QT];
    qzero<QT>(facc);
    for (int c = hx; c < nchunk; c += nh) {
      qzero<QT>(acc);
      qmm<QT>(cur, wfa, acc);
      qstore_b<QT>(acc, fb1, 1.f, true, xb, kQLD, 16 * wave, QP);
      if (c + nh < nchunk) {
        qload_p(W.qpack + kQpFfn1, 8, (c + nh) * 8 + wave, 0, wfa);
        fb1 = gld(W.ffn_b1 + (c + nh) * 128 + 16 * wave + j);
      }
      __syncthreads();
      qmm<QT>(xb, wfb, facc);
      if (c + nh < nchunk) qload_p(W.qpack + qpack_ffn2(W.dim_ff), W.dim_ff >> 4, wave, (c + nh) * 8, wfb);
      __syncthreads();
    }
    {   // partial sums out, coherent stores, then the flag
      const int g = lane >> 4;
      float* part = B.hidden + (size_t)hx * QP * D;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t) gst_agent(part + (size_t)(qt * 16 + 4 * g + t) * D + 16 * wave + j, facc[qt][t] + bad);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (tid == 0) gst_agent(flags + hx, 1u);
    // ---- second job of helpers 1..3: one of the projections that depend only on the layer's new queries
    if (!second) return;
    late = 0;
    if (tid == 0) {
      unsigned spins = 0;
      while (gld_agent(flags + 8) == 0u) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 26)) {
          late = 1;
          break;
        }
      }
    }
    if (block_any(late, any_slot)) sb = __builtin_nanf("");
    for (int e = tid; e < QP * 128; e += nt) {
      const int q = e >> 7, c = e & 127;
      float v = 0.f;
      if (q < Q) {
        v = gld_agent(B.tgt + (size_t)q * D + c);
        if (hx != 2) v += gld(B.qpos + (size_t)q * D + c);   // keys / next query projection take queries + qpos
      }
      xa[q * kQLD + c] = v;
    }
    __syncthreads();
    qzero<QT>(acc);
    qmm<QT>(xa, wfc, acc);
    if (hx == 1) qstore_b<QT, true>(acc, sb, kPointScoreScale, false, B.ks, D, 16 * wave, Q);        // keys pre-scaled (log2 domain)
    else if (hx == 2) qstore_b<QT, true>(acc, sb, 1.f, false, B.vs, D, 16 * wave, Q);
    else qstore_b<QT, true>(acc, sb, kPointScoreScale, false, B.qproj, D, 16 * wave, Q);
    return;
  }

  // ---- the layer's workgroup.  Load phase: the small vectors and the activations first (vmcnt is in order: their waits
  // must not stand behind the weights), then the weights of the first FOUR GEMMs
  const bool deleg = PART != 1 && nh >= 4 && nchunk >= 4;   // helpers 1..3 take the s2c keys / values / next qproj
  auto load_d = [&]() {
    if (deleg) qload_p(W.mpack + kMpW0, 8, wave, 0, wfd);
    else qload_p(W.qpack + kQpS2cKV, 8, wave, 0, wfd);             // s2c k rows
  };
  float fb1 = 0.f;
  {
    float tv[5];
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {
      tv[s5] = 0.f;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const int sgm = 4 * s5 + gi;
        const float* src = ql_vec_src(W, sgm);
        if ((tid >> 7) == gi && src) tv[s5] = gld(src + (tid & 127));
      }
    }
    f32x4 vq[QT], vp[QT], va[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      const int e = tid + u * nt, q = e >> 5, c4 = (e & 31) * 4;
      // all QP rows exist in the scratch buffers: the loads do not wait for the query count (rows >= Q are zeroed below)
      if constexpr (PART != 2) {
        vq[u] = gld4(B.queries + (size_t)q * D + c4);
        vp[u] = gld4(B.qpos + (size_t)q * D + c4);
        va[u] = gld4(B.attn + (size_t)q * D + c4);
      } else {
        vq[u] = gld4(B.tgt + (size_t)q * D + c4);                    // tgt of PART 1
        vp[u] = gld4(B.qpos + (size_t)q * D + c4);
      }
    }
    if constexpr (PART != 2) {
      qload_p(W.qpack + kQpC2sOut, 8, wave, 0, wfa);
      if constexpr (PART == 0) {
        qload_p(W.qpack + kQpC2cIn, 8, wave, 0, wfb);                  // q rows of the c2c in_proj
        qload_p(W.qpack + kQpC2cIn, 8, 8 + wave, 0, wfc);              // k rows
        if constexpr (kDeep) qload_p(W.qpack + kQpC2cIn, 8, 16 + wave, 0, wfd);   // v rows
      }
    } else {
      qload_p(W.qpack + kQpC2cOut, 8, wave, 0, wfa);
      fb1 = gld(W.ffn_b1 + 16 * wave + j);
      qload_p(W.qpack + kQpFfn1, 8, wave, 0, wfb);
      qload_p(W.qpack + qpack_ffn2(W.dim_ff), W.dim_ff >> 4, wave, 0, wfc);
    }
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) vec_l[(4 * s5 + (tid >> 7)) * 128 + (tid & 127)] = tv[s5];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      const int e = tid + u * nt, q = e >> 5, c4 = (e & 31) * 4;
      const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
      *(f32x4*)(cur + q * kQLD + c4) = q < Q ? vq[u] : z;
      *(f32x4*)(qpos + q * kQLD + c4) = q < Q ? vp[u] : z;
      if constexpr (PART != 2) *(f32x4*)(xa + q * kQLD + c4) = q < Q ? va[u] : z;
    }
    if constexpr (PART == 0 && !kDeep) qload_p(W.qpack + kQpC2cIn, 8, 16 + wave, 0, wfd);
  }
  __syncthreads();
  const int cw = 16 * wave + j;                           // this lane's output column in every 128-wide GEMM
  if constexpr (PART != 2) {
  // ---- 1. click-to-scene output projection + residual + LayerNorm (attention_block.py:95-96)
  qzero<QT>(acc);
  qmm<QT>(xa, wfa, acc);
  qstore_b<QT>(acc, vec_l[V_C2S_OUT_B + cw], 1.f, false, xb, kQLD, 16 * wave, QP);
  if constexpr (PART == 0) {
    qload_p(W.qpack + kQpC2cOut, 8, wave, 0, wfa);
  } else {   // 64-row blocks: the registers of the load phase do not leave room for these any earlier
    qload_p(W.qpack + kQpC2cIn, 8, wave, 0, wfb);
    qload_p(W.qpack + kQpC2cIn, 8, 8 + wave, 0, wfc);
  }
  __syncthreads();
  qln16<QT>(cur, xb, Q, vec_l + V_C2S_NW, vec_l + V_C2S_NB, cur, xa, qpos, PART == 1 ? B.tgt : nullptr, nullptr);   // cur = tgt, xa = tgt + qpos
  __syncthreads();
  // ---- 2. click-to-click self attention (attention_block.py:32-36): q | k from tgt + qpos, v from tgt
  if constexpr (PART == 1) {   // to the buffers every block reads in PART 2
    qload_p(W.qpack + kQpC2cIn, 8, 16 + wave, 0, wfd);
    qzero<QT>(acc);
    qmm<QT>(xa, wfc, acc);
    qstore_b<QT, true>(acc, vec_l[V_C2C_IN_B + D + cw], 1.f, false, B.qk, 2 * D, D + 16 * wave, Q);   // k
    qzero<QT>(acc);
    qmm<QT>(cur, wfd, acc);
    qstore_b<QT, true>(acc, vec_l[V_C2C_IN_B + 2 * D + cw], 1.f, false, B.vc, D, 16 * wave, Q);       // v
    qzero<QT>(acc);
    qmm<QT>(xa, wfb, acc);
    qstore_b<QT, true>(acc, vec_l[V_C2C_IN_B + cw], 0.25f, false, B.qk, 2 * D, 16 * wave, Q);         // q (pre-scaled)
    return;
  } else {
    // single block: all three stay in LDS (k -> xb, v -> the qpos buffer, q -> xa once every wave is done reading xa)
    qzero<QT>(acc);
    qmm<QT>(xa, wfc, acc);
    qstore_b<QT>(acc, vec_l[V_C2C_IN_B + D + cw], 1.f, false, xb, kQLD, 16 * wave, Q);            // k (xb, qpos: read before the barrier above)
    qzero<QT>(acc);
    qmm<QT>(cur, wfd, acc);
    qstore_b<QT>(acc, vec_l[V_C2C_IN_B + 2 * D + cw], 1.f, false, qpos, kQLD, 16 * wave, Q);      // v
    qzero<QT>(acc);
    qmm<QT>(xa, wfb, acc);
    __syncthreads();                                                                              // xa fully consumed
    qstore_b<QT>(acc, vec_l[V_C2C_IN_B + cw], 0.25f, false, xa, kQLD, 16 * wave, Q);              // q (pre-scaled)
  }
  fb1 = gld(W.ffn_b1 + cw);
  qload_p(W.qpack + kQpFfn1, 8, wave, 0, wfb);                       // FFN chunk 0, hidden tile `wave`
  qload_p(W.qpack + qpack_ffn2(W.dim_ff), W.dim_ff >> 4, wave, 0, wfc);   // linear2 rows (output columns), chunk 0 columns
  if constexpr (kDeep) load_d();
  __syncthreads();
  qattn_mfma<QT>(xa, xb, qpos, Q, Q);
  if constexpr (!kDeep) load_d();
  __syncthreads();
  if (!deleg) {   // the qpos buffer held v: restore the position encodings for step 4
    for (int e = tid; e < QP * 32; e += nt) {
      const int q = e >> 5, c4 = (e & 31) * 4;
      f32x4 vp = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (q < Q) vp = gld4(B.qpos + (size_t)q * D + c4);
      *(f32x4*)(qpos + q * kQLD + c4) = vp;
    }
  }
  } else {
    // ---- 2 (continued): attention of this block's queries over every block's keys / values
    qattn_mfma_global<QT>(xa, B.qk, all_qk, all_vc, Q, Qall);
    load_d();
    __syncthreads();
  }
  qzero<QT>(acc);
  qmm<QT>(xa, wfa, acc);
  qstore_b<QT>(acc, vec_l[V_C2C_OUT_B + cw], 1.f, false, xb, kQLD, 16 * wave, QP);
  if (deleg) qload_p(W.mpack + kMpW2, 8, wave, 0, wfa);
  else qload_p(W.qpack + kQpS2cKV, 8, 8 + wave, 0, wfa);           // s2c v rows
  __syncthreads();
  qln16<QT>(cur, xb, Q, vec_l + V_C2C_NW, vec_l + V_C2C_NB, cur, nullptr, nullptr, nullptr, nh > 1 ? B.tgt : nullptr);
  if (nh > 1) {   // tgt is with the FFN helpers once the flag is up
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (tid == 0) gst_agent(flags, 1u);
  } else {
    __syncthreads();
  }
  // ---- 3. FFN (attention_block.py:151-155) in hidden chunks of 128: wave w computes hidden tile w of the chunk into
  //         xb, then accumulates output tile w over the chunk
  f32x4 facc[QT];
  qzero<QT>(facc);
  for (int c = 0; c < nchunk; c += nh) {
    qzero<QT>(acc);
    qmm<QT>(cur, wfb, acc);
    qstore_b<QT>(acc, fb1, 1.f, true, xb, kQLD, 16 * wave, QP);
    if (c + nh < nchunk) {
      qload_p(W.qpack + kQpFfn1, 8, (c + nh) * 8 + wave, 0, wfb);
      fb1 = gld(W.ffn_b1 + (c + nh) * 128 + cw);
    }
    __syncthreads();
    qmm<QT>(xb, wfc, facc);
    if (c + nh < nchunk) qload_p(W.qpack + qpack_ffn2(W.dim_ff), W.dim_ff >> 4, wave, (c + nh) * 8, wfc);
    __syncthreads();
  }
  if (nh > 1) {   // the helpers' partial sums, in helper order
    const int nhelp = min(nh, nchunk) - 1;
    int late = 0;
    if (tid < nhelp) {
      unsigned spins = 0;
      while (gld_agent(flags + 1 + tid) == 0u) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 26)) {   // a helper that never reported: the layer's result is NaN, not a partial sum
          late = 1;
          break;
        }
      }
    }
    if (block_any(late, any_slot)) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) facc[qt] += __builtin_nanf("");
    }
    const int g = lane >> 4;
    for (int h = 1; h <= nhelp; ++h) {
      const float* part = B.hidden + (size_t)h * QP * D;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t) facc[qt][t] += gld_agent(part + (size_t)(qt * 16 + 4 * g + t) * D + 16 * wave + j);
    }
  }
  qstore_b<QT>(facc, vec_l[V_FFN_B2 + cw], 1.f, false, xa, kQLD, 16 * wave, QP);
  __syncthreads();
  // cur = the layer's new queries (also to global; to the helpers when they take the projections; + qpos for this workgroup's own)
  qln16<QT>(cur, xa, Q, vec_l + V_FFN_NW, vec_l + V_FFN_NB, cur, deleg ? nullptr : xa, qpos, B.queries, deleg ? B.tgt : nullptr);
  if (deleg) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (tid == 0) gst_agent(flags + 8, 1u);
  } else {
    __syncthreads();
  }
  // ---- 4. everything that depends only on the new queries: s2c keys / values, mask MLP, next layer's c2s query projection
  qln16<QT>(cur, nullptr, Q, vec_l + V_DN_W, vec_l + V_DN_B, xb, nullptr, nullptr, nullptr, nullptr);   // decoder_norm(queries)
  if (!deleg) {
    qzero<QT>(acc);
    qmm<QT>(xa, wfd, acc);
    qstore_b<QT, true>(acc, vec_l[V_S2C_IN_B + D + cw], kPointScoreScale, false, B.ks, D, 16 * wave, Q);   // keys pre-scaled (log2 domain)
    qload_p(W.mpack + kMpW0, 8, wave, 0, wfd);
    qzero<QT>(acc);
    qmm<QT>(cur, wfa, acc);
    qstore_b<QT, true>(acc, vec_l[V_S2C_IN_B + 2 * D + cw], 1.f, false, B.vs, D, 16 * wave, Q);
    qload_p(W.mpack + kMpW2, 8, wave, 0, wfa);
    if (W.next_c2s_in_wt) {
      qload_p(W.next_qpack + kQpC2sQ, 8, wave, 0, wfb);
      qzero<QT>(acc);
      qmm<QT>(xa, wfb, acc);
      qstore_b<QT, true>(acc, vec_l[V_NEXT_B + cw], kPointScoreScale, false, B.qproj, D, 16 * wave, Q);
    }
  }
  __syncthreads();                                                   // xb = decoder_norm rows are complete; qpos is free
  qzero<QT>(acc);
  qmm<QT>(xb, wfd, acc);
  qstore_b<QT>(acc, vec_l[V_M_B0 + cw], 1.f, true, qpos, kQLD, 16 * wave, QP);   // mask MLP hidden
  __syncthreads();
  qzero<QT>(acc);
  qmm<QT>(qpos, wfa, acc);
  qstore_b<QT, true>(acc, vec_l[V_M_B2 + cw], 1.f, false, B.E, D, 16 * wave, Q);
}

}  // namespace a3d

using namespace a3d;

// ------------------------------------------------------------------------------ C ABI
extern "C" int a3d_posenc_fourier(const float* xyz_dev, int64_t n, const float* gauss_B_dev, float* minmax_dev,
                                  float* out_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!xyz_dev || !gauss_B_dev || !minmax_dev || !out_dev || n <= 0 || n > (int64_t)1 << 30) {
    set_error("a3d_posenc_fourier: bad arguments");
    return A3D_ERR_INVALID;
  }
  const int nb = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
  if (!workspace_dev || workspace_bytes < (size_t)nb * 6 * 4) {
    set_error("a3d_posenc_fourier: workspace needs >= %zu bytes", (size_t)256 * 6 * 4);
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace_dev;
  ProfScope ps(st, A3D_PROF_POSENC, 0, 0, 3, 128, (int)n);
  k_minmax_partial<<<nb, 256, 0, st>>>(xyz_dev, (int)n, part);
  k_minmax_final<<<1, 384, 0, st>>>(part, nb, minmax_dev);
  const size_t total = (size_t)n * 64;
  k_fourier<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(xyz_dev, (int)n, gauss_B_dev, minmax_dev, out_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" size_t a3d_posenc_batch_workspace_bytes(int n_samples) {
  return n_samples >= 1 && n_samples <= 64 ? (size_t)n_samples * kPosBlocks * 6 * 4 : 0;
}
extern "C" int a3d_posenc_fourier_batch(const float* xyz_dev, const int64_t* starts_host, int n_samples,
                                        const float* gauss_B_dev, float* minmax_dev, float* out_dev, void* workspace_dev,
                                        size_t workspace_bytes, void* stream) {
  if (!xyz_dev || !starts_host || !gauss_B_dev || !minmax_dev || !out_dev || n_samples < 1 || n_samples > 64) {
    set_error("a3d_posenc_fourier_batch: bad arguments (1..64 samples)");
    return A3D_ERR_INVALID;
  }
  PosBatch pb;
  pb.ns = n_samples;
  for (int b = 0; b <= n_samples; ++b) {
    if (starts_host[b] < 0 || starts_host[b] > ((int64_t)1 << 30) || (b && starts_host[b] <= starts_host[b - 1])) {
      set_error("a3d_posenc_fourier_batch: sample %d has no rows (starts must ascend)", b - 1);
      return A3D_ERR_INVALID;
    }
    pb.start[b] = (int)starts_host[b];
  }
  if (!workspace_dev || workspace_bytes < a3d_posenc_batch_workspace_bytes(n_samples)) {
    set_error("a3d_posenc_fourier_batch: workspace needs >= %zu bytes", a3d_posenc_batch_workspace_bytes(n_samples));
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace_dev;
  const int n_total = pb.start[n_samples];
  ProfScope ps(st, A3D_PROF_POSENC, 0, 0, 3, 128, n_total);
  k_minmax_partial_b<<<dim3(kPosBlocks, n_samples), 256, 0, st>>>(xyz_dev, pb, part);
  k_minmax_final<<<n_samples, 384, 0, st>>>(part, kPosBlocks, minmax_dev);
  int n_max = 0;
  for (int b = 0; b < n_samples; ++b) n_max = std::max(n_max, pb.start[b + 1] - pb.start[b]);
  k_fourier_b<<<dim3((unsigned)(((size_t)n_max * 64 + 255) / 256), n_samples), 256, 0, st>>>(xyz_dev, pb, gauss_B_dev, minmax_dev,
                                                                                            out_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

static int check_weights(const a3d_decoder_weights* w) {
  if (!w || w->n_layers < 1 || w->n_layers > A3D_MAX_DEC_LAYERS || w->n_bg_queries < 0 ||
      w->n_bg_queries > A3D_MAX_QUERIES || w->dim_ff % 128 != 0) {
    set_error("decoder: bad weight struct");
    return A3D_ERR_INVALID;
  }
  return A3D_OK;
}

// the query-side packs are required by every forward path: refused BEFORE anything is enqueued into the caller's workspace
// and logits buffers (the per-layer check inside the launch loop only fires after layer 0's kernels are in flight)
static int check_packs(const a3d_decoder_weights* w) {
  if (!w->mask_pack) {
    set_error("a3d_decoder_forward: the decoder weights carry no mask-head pack (a3d_decoder_pack_query_weights fills "
              "a3d_decoder_layer::query_pack and a3d_decoder_weights::mask_pack)");
    return A3D_ERR_INVALID;
  }
  for (int l = 0; l < w->n_layers; ++l)
    if (!w->layers[l].query_pack) {
      set_error("a3d_decoder_forward: decoder layer %d carries no query-side pack (a3d_decoder_pack_query_weights fills "
                "a3d_decoder_layer::query_pack and a3d_decoder_weights::mask_pack)", l);
      return A3D_ERR_INVALID;
    }
  return A3D_OK;
}

namespace {
struct DecLayout {
  size_t buf[4], labels, counts, part, meta, desc, q[12], sync, total;
  int qp, nchunk;
};
int round_qp(int nq) { return nq <= 16 ? 16 : nq <= 32 ? 32 : nq <= 48 ? 48 : (nq + 63) / 64 * 64; }
void dec_layout(int64_t n, int nq, DecLayout& L) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += align256(bytes);
    return o;
  };
  L.qp = round_qp(nq);
  L.nchunk = (int)((n + kC2SChunk - 1) / kC2SChunk);
  for (int i = 0; i < 4; ++i) L.buf[i] = take((size_t)n * D * 4);
  L.labels = take((size_t)n + 64);
  L.counts = take((size_t)A3D_MAX_DEC_LAYERS * (A3D_MAX_QUERIES + 1) * 4);
  L.part = take((size_t)(L.nchunk > kFusedC2SGrid * 8 ? L.nchunk : kFusedC2SGrid * 8) * H * L.qp * kPartStride * 4);
  // everything a call uploads -- the samples' QueryMeta records and the two sample tables, packed back to back for the number
  // of samples at hand -- lives in the FIRST sample's workspace and travels in ONE host-to-device copy (round 5: a copy per
  // sample + two table copies + a memset were five 5-us slots of the stream in front of a 0.67 ms decoder pass)
  L.desc = take((sizeof(QueryMeta) + 16 + sizeof(DecSampleDev) + sizeof(QuerySample)) * kMaxBatchSamples + 64);
  L.meta = L.desc;
  const size_t qb = (size_t)L.qp * D * 4;
  for (int i = 0; i < 9; ++i) L.q[i] = take(qb);        // queries qpos qproj ks vs E attn tmp tgt
  L.q[9] = take(2 * qb);                                // qk
  L.q[10] = take(qb);                                   // vc
  L.q[11] = take((size_t)L.qp * 4096 * 4);              // hidden: kQlMaxHelpers x [qp][128] FFN partial sums fit (dim_ff <= 4096)
  L.sync = take((size_t)kMaxBatchSamples * A3D_MAX_DEC_LAYERS * kMaxQBlocks * 16 * 4);   // k_query_block flags of a batched call (first sample's workspace)
  L.total = off;
}
}  // namespace

extern "C" size_t a3d_decoder_workspace_bytes(int64_t n, int n_queries) {
  if (n <= 0 || n_queries < 1 || n_queries > A3D_MAX_QUERIES) return 0;
  DecLayout L;
  dec_layout(n, n_queries, L);
  return L.total + 256;
}

extern "C" size_t a3d_decoder_query_pack_floats(int32_t dim_ff) { return dim_ff > 0 && dim_ff % 128 == 0 ? qpack_floats(dim_ff) : 0; }
extern "C" size_t a3d_decoder_mask_pack_floats(void) { return kMpFloats; }
extern "C" int a3d_decoder_pack_query_weights(const a3d_decoder_weights* w, int32_t layer, float* out_dev, void* stream) {
  int rc = check_weights(w);
  if (rc) return rc;
  if (!out_dev || layer < -1 || layer >= w->n_layers) {
    set_error("a3d_decoder_pack_query_weights: bad arguments");
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  auto pack = [&](const float* W, int N, int K, size_t off) {
    const size_t n4 = (size_t)(N >> 4) * (K >> 4) * 64;
    k_pack_rows<<<(unsigned)((n4 + 255) / 256), 256, 0, st>>>(W, N, K, out_dev + off);
  };
  if (layer < 0) {
    if (!w->mask_w0 || !w->mask_w2) {
      set_error("a3d_decoder_pack_query_weights: mask head weights missing");
      return A3D_ERR_INVALID;
    }
    pack(w->mask_w0, D, D, kMpW0);
    pack(w->mask_w2, D, D, kMpW2);
  } else {
    const a3d_decoder_layer& LW = w->layers[layer];
    if (!LW.c2s_out_w || !LW.c2c_in_w || !LW.c2c_out_w || !LW.s2c_in_w || !LW.c2s_in_w || !LW.ffn_w1 || !LW.ffn_w2) {
      set_error("a3d_decoder_pack_query_weights: layer %d weights missing", layer);
      return A3D_ERR_INVALID;
    }
    pack(LW.c2s_out_w, D, D, kQpC2sOut);
    pack(LW.c2c_in_w, 3 * D, D, kQpC2cIn);
    pack(LW.c2c_out_w, D, D, kQpC2cOut);
    pack(LW.s2c_in_w + (size_t)D * D, 2 * D, D, kQpS2cKV);
    pack(LW.c2s_in_w, D, D, kQpC2sQ);
    pack(LW.ffn_w1, w->dim_ff, D, kQpFfn1);
    pack(LW.ffn_w2, D, w->dim_ff, qpack_ffn2(w->dim_ff));
  }
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

static bool fused_c2s() {   // A3D_FUSED_C2S=0 keeps the separate K / V GEMMs + k_c2s_attn (A/B switch)
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("A3D_FUSED_C2S");
    v = e ? atoi(e) : 1;
  }
  return v != 0;
}

static bool fused_wide() {   // A3D_FUSED_WIDE=0 (or A3D_FUSED_C2S=0) keeps the unfused kernels above 64 queries (A/B switch, tests)
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("A3D_FUSED_WIDE");
    v = e ? atoi(e) : 1;
  }
  return v != 0 && fused_c2s();
}

// The wide tier's kernels take a sample's tile count from the table, so they serve fewer than 65 queries too (under the
// 5-tile build).  From 33 queries on they are the default: a pass at 80 k points 0.83 -> 0.80 ms at 35 queries, 0.97 -> 0.83
// at 60 (k_q_s2c + k_out_ln_mask and a 64-row query block against k_s2c_w + k_out_w and two 32-row blocks), and those samples
// then share the launch group of a call's larger ones.  Up to 32 queries k_s2c_out (one kernel for the whole scene-to-click
// half) stays ahead.  A3D_WIDE_FROM=<queries> moves the edge (65: the round's first protocol; A/B, tests).
static bool fused_s2o() {   // A3D_FUSED_S2O=0: k_s2c_w + k_out_w for up to five query tiles too (A/B, tests)
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("A3D_FUSED_S2O");
    v = e ? atoi(e) : 1;
  }
  return v != 0;
}
static int wide_from() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("A3D_WIDE_FROM");
    v = e ? atoi(e) : 33;
    v = v < 17 ? 17 : v;
  }
  return v;
}

// one batch sample on the host: validated query list, workspace layout and the views into its workspace
struct Prepared {
  QueryMeta hm;
  DecLayout L;
  char* ws;
  const float *feats, *posenc;
  int n;
  float* logits;
  float* kv0 = nullptr;   // the scene's cached first-layer keys / values / scene-to-click queries [3][n][128] (a3d_decoder_sample::kv0_dev) and their state
  int kv0_state = 0;
  int qtw = 0;            // tile count of the fused wide tier (decoder_wide.h) that serves this sample, 0 = none
  // views
  float *bufA, *bufB, *bufC, *bufD, *part;
  unsigned char* labels;
  int* counts;
  QueryMeta* meta;
  QueryBufs B;
  int wg_begin, wg_end;
  void bind() {
    bufA = (float*)(ws + L.buf[0]);
    bufB = (float*)(ws + L.buf[1]);
    bufC = (float*)(ws + L.buf[2]);
    bufD = (float*)(ws + L.buf[3]);
    labels = (unsigned char*)(ws + L.labels);
    counts = (int*)(ws + L.counts);
    part = (float*)(ws + L.part);
    meta = (QueryMeta*)(ws + L.meta);
    B.queries = (float*)(ws + L.q[0]);
    B.qpos = (float*)(ws + L.q[1]);
    B.qproj = (float*)(ws + L.q[2]);
    B.ks = (float*)(ws + L.q[3]);
    B.vs = (float*)(ws + L.q[4]);
    B.E = (float*)(ws + L.q[5]);
    B.attn = (float*)(ws + L.q[6]);
    B.tmp = (float*)(ws + L.q[7]);
    B.tgt = (float*)(ws + L.q[8]);
    B.qk = (float*)(ws + L.q[9]);
    B.vc = (float*)(ws + L.q[10]);
    B.hidden = (float*)(ws + L.q[11]);
  }
};

// The device tables of one call -- the samples' QueryMeta records, the wide kernels' sample table (workgroups in proportion to
// the samples' point groups) and the query-side table -- built on the host and uploaded in ONE copy into the first sample's
// workspace.  wg_per_group: a workgroup (not a wave) takes a 16-point group (decoder_wide.h); part_per_wg: the click-to-scene
// kernel writes one flash partial per workgroup.
struct DecTables {
  int grid;
  DecSampleDev* samples_dev;
  QuerySample* qs_dev;
};
static int upload_tables(Prepared* P, int ns, bool wg_per_group, bool part_per_wg, hipStream_t st, DecTables& T) {
  // the upload block in the first sample's workspace: [ns QueryMeta][ns DecSampleDev][ns QuerySample]
  char* const up_dev = P[0].ws + P[0].L.desc;
  const size_t up_meta = 0, up_samples = (sizeof(QueryMeta) * ns + 15) & ~(size_t)15;
  const size_t up_qs = (up_samples + sizeof(DecSampleDev) * ns + 15) & ~(size_t)15;
  const size_t up_bytes = up_qs + sizeof(QuerySample) * ns;
  for (int si = 0; si < ns; ++si) P[si].meta = (QueryMeta*)(up_dev + up_meta) + si;
  // ---- sample table: workgroups of the persistent kernels in proportion to the samples' point groups
  int grid = 0;
  T.samples_dev = (DecSampleDev*)(up_dev + up_samples);
  T.qs_dev = (QuerySample*)(up_dev + up_qs);
  {
    std::vector<char> up_host(up_bytes);
    for (int si = 0; si < ns; ++si) memcpy(up_host.data() + up_meta + sizeof(QueryMeta) * si, &P[si].hm, sizeof(QueryMeta));
    DecSampleDev* hd = (DecSampleDev*)(up_host.data() + up_samples);
    int64_t tot_groups = 0;
    for (int si = 0; si < ns; ++si) tot_groups += (P[si].n + 15) / 16;
    const int max_grid = 256;
    for (int si = 0; si < ns; ++si) {
      Prepared& p = P[si];
      const int ngroups = (p.n + 15) / 16;
      int share = (int)((int64_t)max_grid * ngroups / tot_groups);
      const int useful = wg_per_group ? ngroups : (ngroups + 7) / 8;   // eight waves take a group each, or the workgroup takes one
      share = share < 1 ? 1 : share;
      share = share > useful ? useful : share;
      p.wg_begin = grid;
      p.wg_end = grid += share;
      DecSampleDev& d = hd[si];
      d.wg_begin = p.wg_begin;
      d.wg_end = p.wg_end;
      d.n = p.n;
      d.nq = p.hm.nq;
      d.K = p.hm.K;
      d.n_fg = p.hm.n_fg;
      d.feats = p.feats;
      d.posenc = p.posenc;
      d.bufB = p.bufB;
      d.bufC = p.bufC;
      d.bufD = p.bufD;
      d.logits = p.logits;
      d.labels = p.labels;
      d.counts = p.counts;
      d.part = p.part;
      d.qobj = p.meta->obj;          // addresses inside the device copy of QueryMeta
      d.qrange = p.meta->qrange;
      d.qproj = p.B.qproj;
      d.ks = p.B.ks;
      d.vs = p.B.vs;
      d.E = p.B.E;
      d.q0 = p.kv0 && p.kv0_state != 0 ? p.kv0 + (size_t)2 * p.n * D : nullptr;
      d.k0 = p.kv0 && p.kv0_state != 0 ? p.kv0 : nullptr;
      d.v0 = p.kv0 && p.kv0_state != 0 ? p.kv0 + (size_t)p.n * D : nullptr;
      d.qp = p.L.qp;
      d.nqt = (p.hm.nq + 15) / 16;
    }
    QuerySample* hq = (QuerySample*)(up_host.data() + up_qs);
    unsigned* sync0 = (unsigned*)(P[0].ws + P[0].L.sync);   // zeroed by k_query_init (a sample's first query block)
    for (int si = 0; si < ns; ++si) {
      Prepared& p = P[si];
      p.B.sync = sync0 + (size_t)si * A3D_MAX_DEC_LAYERS * kMaxQBlocks * 16;
      hq[si].meta = p.meta;
      hq[si].B = p.B;
      hq[si].feats = p.feats;
      hq[si].posenc = p.posenc;
      hq[si].counts = p.counts;
      hq[si].part = p.part;
      hq[si].n_part = part_per_wg ? (p.wg_end - p.wg_begin) : p.L.nchunk;   // k_kv_c2s merges its slot groups: one partial per workgroup
      hq[si].n_part0 = p.L.nchunk;
      hq[si].qp = p.L.qp;
    }
    // pageable source: the runtime stages it before returning
    A3D_HIP_CHECK(hipMemcpyAsync(up_dev, up_host.data(), up_bytes, hipMemcpyHostToDevice, st));
  }
  T.grid = grid;
  return A3D_OK;
}

// the query side of decoder layer l for every sample of a call: merge of the click-to-scene partials, then one workgroup
// (or chain of 64-query blocks, QT = 4) per sample
template <int QT>
static int launch_query_side(const a3d_decoder_weights* w, int l, QuerySample* qs_dev, int qp, int nblk, int ns, int nq_max,
                             bool cached0, hipStream_t st) {
  constexpr int QP = QT * 16;
  const a3d_decoder_layer& LW = w->layers[l];
    QueryLayerW QW;
  QW.c2s_in_wt = LW.c2s_in_w; QW.c2s_in_b = LW.c2s_in_b; QW.c2s_out_wt = LW.c2s_out_w; QW.c2s_out_b = LW.c2s_out_b;
  QW.c2s_norm_w = LW.c2s_norm_w; QW.c2s_norm_b = LW.c2s_norm_b;
  QW.c2c_in_wt = LW.c2c_in_w; QW.c2c_in_b = LW.c2c_in_b; QW.c2c_out_wt = LW.c2c_out_w; QW.c2c_out_b = LW.c2c_out_b;
  QW.c2c_norm_w = LW.c2c_norm_w; QW.c2c_norm_b = LW.c2c_norm_b;
  QW.ffn_w1t = LW.ffn_w1; QW.ffn_b1 = LW.ffn_b1; QW.ffn_w2t = LW.ffn_w2; QW.ffn_b2 = LW.ffn_b2;
  QW.ffn_norm_w = LW.ffn_norm_w; QW.ffn_norm_b = LW.ffn_norm_b;
  QW.s2c_in_wt = LW.s2c_in_w; QW.s2c_in_b = LW.s2c_in_b;
  QW.dn_w = w->decoder_norm_w; QW.dn_b = w->decoder_norm_b;
  QW.m_w0t = w->mask_w0; QW.m_b0 = w->mask_b0; QW.m_w2t = w->mask_w2; QW.m_b2 = w->mask_b2;
  QW.next_c2s_in_wt = l + 1 < w->n_layers ? w->layers[l + 1].c2s_in_w : nullptr;
  QW.next_c2s_in_b = l + 1 < w->n_layers ? w->layers[l + 1].c2s_in_b : nullptr;
  QW.qpack = LW.query_pack;
  QW.next_qpack = l + 1 < w->n_layers ? w->layers[l + 1].query_pack : nullptr;
  QW.mpack = w->mask_pack;
  QW.dim_ff = w->dim_ff;
  QW.layer = l;
  {   // one workgroup (or chain of query blocks) per sample: blockIdx.y
    ProfScope ps(st, A3D_PROF_QUERY, 0, 0, 0, 0, nq_max);
    k_c2s_combine<<<dim3(nq_max * H, ns), 64, 0, st>>>(qs_dev, cached0 ? 1 : 0);
    const size_t ql_lds = (size_t)4 * QP * kQLD * 4 + 16;   // four [QP][132] tiles + the word of block_any
    // FFN helper workgroups next to a block's workgroup (A3D_QL_HELPERS = total workgroups per block, 1 = none: the
    // switch the tests use to compare the hand-off with the single-workgroup chain)
    static int nh_env = -1;
    if (nh_env < 0) {
      const char* e = getenv("A3D_QL_HELPERS");
      nh_env = e ? atoi(e) : kQlMaxHelpers;
      nh_env = nh_env < 1 ? 1 : nh_env > kQlMaxHelpers ? kQlMaxHelpers : nh_env;
    }
    if (!(QW.qpack && QW.mpack && (QW.next_qpack || !QW.next_c2s_in_wt))) {
      set_error("a3d_decoder_forward: the decoder weights carry no query-side packs (a3d_decoder_pack_query_weights "
                "fills a3d_decoder_layer::query_pack and a3d_decoder_weights::mask_pack)");
      return A3D_ERR_INVALID;
    }
    const size_t qb_lds = ql_lds + (size_t)kQVec * 4;
    if (nblk == 1) {
      k_query_block<QT, 0><<<dim3(nh_env, 1, ns), 512, qb_lds, st>>>(qs_dev, QW);
    } else if constexpr (QT == 4 || QT == 2) {   // blocks of 64 rows (unfused path) or of 32 (wide tier: more, shorter chains)
      k_query_block<QT, 1><<<dim3(1, nblk, ns), 512, qb_lds, st>>>(qs_dev, QW);
      k_query_block<QT, 2><<<dim3(nh_env, nblk, ns), 512, qb_lds, st>>>(qs_dev, QW);
    }
  }
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

// All samples of one call share QT (= padded query count / 16).  Per decoder layer: ONE launch of each fused wide
// kernel for the whole batch (DecSampleDev table), the small query-side kernels once per sample in between.
template <int QT>
static int run_decoder(const a3d_decoder_weights* w, Prepared* P, int ns, hipStream_t st) {
  constexpr int QP = QT * 16;
  const int nblk = P[0].L.qp / QP;          // query blocks (1 unless nq > 64)
  {
    static bool big = false;
    if (!big) {
      big = true;
      const int big_lds = 160 * 1024;
      A3D_ALLOW_LDS(big_lds, k_query_block<1, 0>);
      A3D_ALLOW_LDS(big_lds, k_query_block<2, 0>);
      A3D_ALLOW_LDS(big_lds, k_query_block<3, 0>);
      A3D_ALLOW_LDS(big_lds, k_query_block<4, 0>);
      A3D_ALLOW_LDS(big_lds, k_query_block<4, 1>);
      A3D_ALLOW_LDS(big_lds, k_query_block<4, 2>);
      A3D_ALLOW_LDS(big_lds, k_s2c_attn_wide<4>);
      A3D_ALLOW_LDS(big_lds, k_q_s2c<1>);
      A3D_ALLOW_LDS(big_lds, k_q_s2c<2>);
      A3D_ALLOW_LDS(big_lds, k_q_s2c<3>);
      A3D_ALLOW_LDS(big_lds, k_q_s2c<4>);
      A3D_ALLOW_LDS(big_lds, k_out_ln_mask<1>);
      A3D_ALLOW_LDS(big_lds, k_out_ln_mask<2>);
      A3D_ALLOW_LDS(big_lds, k_out_ln_mask<3>);
      A3D_ALLOW_LDS(big_lds, k_out_ln_mask<4>);
      A3D_ALLOW_LDS(big_lds, k_s2c_out<1, 8>);
      A3D_ALLOW_LDS(big_lds, k_s2c_out<2, 8>);
      A3D_ALLOW_LDS(big_lds, k_s2c_out<1, 12>);
      A3D_ALLOW_LDS(big_lds, k_s2c_out<2, 12>);
      A3D_ALLOW_LDS(big_lds, (k_s2c_out<2, 12, true>));
      A3D_ALLOW_LDS(big_lds, (k_s2c_out<2, 8, true>));
      A3D_ALLOW_LDS(big_lds, (k_s2c_out<1, 12, true>));
      A3D_ALLOW_LDS(big_lds, (k_s2c_out<1, 8, true>));
      A3D_ALLOW_LDS(big_lds, (k_q_s2c<1, true>));
      A3D_ALLOW_LDS(big_lds, (k_q_s2c<2, true>));
      A3D_ALLOW_LDS(big_lds, (k_q_s2c<3, true>));
      A3D_ALLOW_LDS(big_lds, (k_q_s2c<4, true>));
      A3D_ALLOW_LDS(big_lds, (k_s2c_out<1, 12, false, true>));
      A3D_ALLOW_LDS(big_lds, (k_s2c_out<2, 12, false, true>));
      A3D_ALLOW_LDS(big_lds, (k_s2c_out<1, 12, true, true>));
      A3D_ALLOW_LDS(big_lds, (k_s2c_out<2, 12, true, true>));
      A3D_ALLOW_LDS(big_lds, k_kv_c2s<1>);
      A3D_ALLOW_LDS(big_lds, k_kv_c2s<2>);
      A3D_ALLOW_LDS(big_lds, k_kv_c2s<3>);
      A3D_ALLOW_LDS(big_lds, k_kv_c2s<4>);
      A3D_ALLOW_LDS(big_lds, k_ln_mask<4>);
    }
  }
  const int n_counts = A3D_MAX_DEC_LAYERS * (A3D_MAX_QUERIES + 1);
  int64_t n_total = 0;
  int Kmax = 0, nq_max = 0;
  for (int si = 0; si < ns; ++si) {
    Prepared& p = P[si];
    n_total += p.n;
    Kmax = p.hm.K > Kmax ? p.hm.K : Kmax;
    nq_max = p.hm.nq > nq_max ? p.hm.nq : nq_max;
  }
  const size_t s2c_lds = (size_t)2 * QP * 132 * 4;             // keys + values of the queries (k_s2c_attn_wide)
  const size_t qs2c_lds = ((size_t)QP * 132 + (size_t)D * (QP + 4) + D) * 4;   // k_q_s2c: keys, transposed values, bias
  const size_t fused_lds = (size_t)64 * 1024 + ((size_t)3 * D + QP * 132 + 8 * 16 * (QP + 1) + 8 * 16 * (Kmax + 1)) * 4 +
                           (size_t)(2 * Kmax + 3) * 4;
  for (int si = 0; si < ns; ++si) {
    const size_t lnm_lds = ((size_t)QP * 132 + 4 * 16 * (P[si].L.qp + 1) + 4 * 16 * (P[si].hm.K + 1)) * 4 + (size_t)(P[si].hm.K + 1) * 4;
    if (lnm_lds > 160 * 1024) {
      set_error("a3d_decoder_forward: %d queries x %d objects need %zu bytes of LDS in the mask head", P[si].hm.nq,
                P[si].hm.K, lnm_lds);
      return A3D_ERR_UNSUPPORTED;
    }
  }
  // which of the wide phases run as one fused launch for the batch
  const bool fuse_c2s = fused_c2s() && nblk == 1;   // up to 64 queries (k_kv_c2s<1..4>)
  const bool fuse_s2c = fused_c2s() && nblk == 1;
  const bool fuse_out = fuse_s2c && fused_lds <= 160 * 1024;
  // scene-to-click + output projection + LayerNorm + mask head as ONE kernel when both packed weight matrices and the
  // queries' compact keys / values fit the LDS (about 24 queries at 5 objects); A3D_FUSED_S2C=0 keeps the two kernels
  const int nqr_max = (nq_max + 3) & ~3;
  const size_t s2c_rest = ((size_t)D * (nqr_max + 4) + 4 * D + (size_t)8 * 16 * (Kmax + 1) + 2 * Kmax + 3 + QT * 16 + 1) * 4;
  const size_t s2c_keys = (size_t)nqr_max * 136 * 4;
  const size_t staging12 = (size_t)4 * 16 * (Kmax + 1) * 4;   // twelve waves' logits staging
  // the keys move out of LDS (k_s2c_out<.., KG>) when everything else still fits: 25..32 queries at 5 objects
  const bool s2c_kg = (size_t)128 * 1024 + s2c_keys + s2c_rest + staging12 > 160 * 1024;
  const size_t s2c_out_lds = (size_t)128 * 1024 + (s2c_kg ? 0 : s2c_keys) + s2c_rest;
  const size_t s2c_out_lds12 = s2c_out_lds + staging12;
  static int fused_s2c_env = -1;
  if (fused_s2c_env < 0) {
    const char* e = getenv("A3D_FUSED_S2C");
    fused_s2c_env = e ? atoi(e) : 1;
  }
  const bool fuse_all = fuse_out && fused_s2c_env && QT <= 2 && s2c_out_lds <= 160 * 1024;
  DecTables T;
  {
    const int rc_t = upload_tables(P, ns, false, fuse_c2s, st, T);
    if (rc_t) return rc_t;
  }
  const int grid = T.grid;
  DecSampleDev* const samples_dev = T.samples_dev;
  QuerySample* const qs_dev = T.qs_dev;
  {
    ProfScope ps(st, A3D_PROF_QUERY, 0, 0, 0, 0, nq_max);
    k_query_init<QT><<<dim3(nblk, ns), 512, 0, st>>>(qs_dev, w->bg_query_feat, w->bg_query_pos, w->time_table,
                                                    w->layers[0].c2s_in_w, w->layers[0].c2s_in_b, n_counts);
    A3D_LAUNCH_CHECK();
  }
  for (int l = 0; l < w->n_layers; ++l) {
    const a3d_decoder_layer& LW = w->layers[l];
    int rc;
    // ---- click-to-scene
    // first layer on the scene's cached keys / values (a3d_decoder_sample::kv0_dev): K = (feats + pos) Wk^T + bk and
    // V = feats Wv^T + bv depend on the scene only, the interactive loop runs ~100 passes on it
    bool cached0 = l == 0;
    for (int si = 0; si < ns && cached0; ++si) cached0 = P[si].kv0 != nullptr && P[si].kv0_state != 0;
    const bool qc0 = cached0;   // the scene-to-click half of this layer reads its queries from the cache too
    if (cached0) {
      for (int si = 0; si < ns; ++si) {
        Prepared& p = P[si];
        float* K0 = p.kv0;
        float* V0 = p.kv0 + (size_t)p.n * D;
        if (p.kv0_state == 1) {
          rc = a3d_linear(p.feats, D, p.posenc, D, p.n, D, D, LW.c2s_wk_packed, nullptr, LW.c2s_in_b + D, nullptr, 0, 0, K0, D, nullptr, 0, st);
          if (rc) return rc;
          rc = a3d_linear(p.feats, D, nullptr, 0, p.n, D, D, LW.c2s_wv_packed, nullptr, LW.c2s_in_b + 2 * D, nullptr, 0, 0, V0, D, nullptr, 0, st);
          if (rc) return rc;
          // ... and the scene-to-click QUERIES of the first layer, (feats + pos) Wq^T + bq: click-independent too (agile3d.py:305-312)
          rc = a3d_linear(p.feats, D, p.posenc, D, p.n, D, D, LW.s2c_wq_packed, nullptr, LW.s2c_in_b, nullptr, 0, 0,
                          p.kv0 + (size_t)2 * p.n * D, D, nullptr, 0, st);
          if (rc) return rc;
        }
        if (nblk == 1) continue;              // one query block: all samples in ONE launch below
        ProfScope ps(st, A3D_PROF_C2S, 0, 0, 0, 0, p.n);
        k_c2s_attn<QT><<<dim3(p.L.nchunk, nblk), 512, 0, st>>>(K0, V0, p.n, p.B.qproj, p.meta->obj, nullptr, nullptr, p.part, p.L.qp);
        A3D_LAUNCH_CHECK();
      }
      if (nblk == 1) {
        int nchunk_max = 0;
        for (int si = 0; si < ns; ++si) nchunk_max = P[si].L.nchunk > nchunk_max ? P[si].L.nchunk : nchunk_max;
        ProfScope ps(st, A3D_PROF_C2S, 0, 0, 0, 0, (int)n_total);
        k_c2s_attn_b<QT><<<dim3(nchunk_max, ns), 512, 0, st>>>(samples_dev);
        A3D_LAUNCH_CHECK();
      }
    } else if (fuse_c2s) {
      // K / V projections fused in (K, V never reach HBM), all samples in one launch
      ProfScope ps(st, A3D_PROF_C2S, 0, 0, 0, 0, (int)n_total);
      const size_t c2s_lds = (size_t)128 * 1024 + (QT == 4 ? (size_t)QP * 128 * 4 : ((size_t)QP * 132 + 2 * D) * 4);
      k_kv_c2s<QT><<<grid, 512, c2s_lds, st>>>(samples_dev, ns, l, LW.c2s_wk_packed, LW.c2s_wv_packed, LW.c2s_in_b + D,
                                               LW.c2s_in_b + 2 * D);
      A3D_LAUNCH_CHECK();
    } else {
      for (int si = 0; si < ns; ++si) {
        Prepared& p = P[si];
        const float* src = l == 0 ? p.feats : (((l - 1) & 1) ? p.bufD : p.bufC);
        const int* prev_counts = l > 0 ? p.counts + (size_t)(l - 1) * (A3D_MAX_QUERIES + 1) : nullptr;
        // K = (src + pos) Wk^T + bk, V = src Wv^T + bv   (attention_block.py:88-94)
        rc = a3d_linear(src, D, p.posenc, D, p.n, D, D, LW.c2s_wk_packed, nullptr, LW.c2s_in_b + D, nullptr, 0, 0, p.bufA, D, nullptr, 0, st);
        if (rc) return rc;
        rc = a3d_linear(src, D, nullptr, 0, p.n, D, D, LW.c2s_wv_packed, nullptr, LW.c2s_in_b + 2 * D, nullptr, 0, 0, p.bufB, D, nullptr, 0, st);
        if (rc) return rc;
        ProfScope ps(st, A3D_PROF_C2S, 0, 0, 0, 0, p.n);
        k_c2s_attn<QT><<<dim3(p.L.nchunk, nblk), 512, 0, st>>>(p.bufA, p.bufB, p.n, p.B.qproj, p.meta->obj,
                                                              l > 0 ? p.labels : nullptr, prev_counts, p.part, p.L.qp);
        A3D_LAUNCH_CHECK();
      }
    }
    // ---- query side, per sample
    rc = launch_query_side<QT>(w, l, qs_dev, P[0].L.qp, nblk, ns, nq_max, cached0, st);
    if (rc) return rc;
    // ---- scene-to-click: Q = (src + pos) Wq^T + bq; attention; Y = O Wo^T + bo + src; LN
    if (fuse_all) {
      // the whole half in one pass: O never reaches HBM (k_s2c_out)
      ProfScope ps(st, A3D_PROF_S2C, 0, 0, 0, 0, (int)n_total);
      if constexpr (QT <= 2) {
        if (qc0 && s2c_kg && s2c_out_lds12 <= 160 * 1024)
          k_s2c_out<QT, 12, true, true><<<grid, 768, s2c_out_lds12, st>>>(samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b, LW.s2c_wo_packed,
                                                                         LW.s2c_out_b, LW.s2c_norm_w, LW.s2c_norm_b, nqr_max, Kmax);
        else if (qc0 && !s2c_kg && s2c_out_lds12 <= 160 * 1024)
          k_s2c_out<QT, 12, false, true><<<grid, 768, s2c_out_lds12, st>>>(samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b, LW.s2c_wo_packed,
                                                                          LW.s2c_out_b, LW.s2c_norm_w, LW.s2c_norm_b, nqr_max, Kmax);
        else if (s2c_kg && s2c_out_lds12 <= 160 * 1024)
          k_s2c_out<QT, 12, true><<<grid, 768, s2c_out_lds12, st>>>(samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b, LW.s2c_wo_packed,
                                                                   LW.s2c_out_b, LW.s2c_norm_w, LW.s2c_norm_b, nqr_max, Kmax);
        else if (s2c_kg)
          k_s2c_out<QT, 8, true><<<grid, 512, s2c_out_lds, st>>>(samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b, LW.s2c_wo_packed,
                                                                LW.s2c_out_b, LW.s2c_norm_w, LW.s2c_norm_b, nqr_max, Kmax);
        else if (s2c_out_lds12 <= 160 * 1024)   // twelve waves (three per SIMD) when their logits staging fits, else eight
          k_s2c_out<QT, 12><<<grid, 768, s2c_out_lds12, st>>>(samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b, LW.s2c_wo_packed,
                                                             LW.s2c_out_b, LW.s2c_norm_w, LW.s2c_norm_b, nqr_max, Kmax);
        else
          k_s2c_out<QT, 8><<<grid, 512, s2c_out_lds, st>>>(samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b, LW.s2c_wo_packed,
                                                          LW.s2c_out_b, LW.s2c_norm_w, LW.s2c_norm_b, nqr_max, Kmax);
      }
      A3D_LAUNCH_CHECK();
      continue;
    }
    if (fuse_s2c) {
      ProfScope ps(st, A3D_PROF_S2C, 0, 0, 0, 0, (int)n_total);
      if (qc0) k_q_s2c<QT, true><<<grid, 512, (size_t)64 * 1024 + qs2c_lds, st>>>(samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b);
      else k_q_s2c<QT><<<grid, 512, (size_t)64 * 1024 + qs2c_lds, st>>>(samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b);
      A3D_LAUNCH_CHECK();
    } else {
      for (int si = 0; si < ns; ++si) {
        Prepared& p = P[si];
        const float* src = l == 0 ? p.feats : (((l - 1) & 1) ? p.bufD : p.bufC);
        const float* Qs = p.bufA;
        if (qc0) {
          Qs = p.kv0 + (size_t)2 * p.n * D;     // the cached queries of the first layer
        } else {
          rc = a3d_linear(src, D, p.posenc, D, p.n, D, D, LW.s2c_wq_packed, nullptr, LW.s2c_in_b, nullptr, 0, 0, p.bufA, D, nullptr, 0, st);
          if (rc) return rc;
        }
        ProfScope ps(st, A3D_PROF_S2C, 0, 0, 0, 0, p.n);
        k_s2c_attn_wide<QT><<<(p.n + 127) / 128, 512, s2c_lds, st>>>(Qs, p.n, p.B.ks, p.B.vs, p.hm.nq, nblk, p.bufB);
        A3D_LAUNCH_CHECK();
      }
    }
    if (fuse_out) {
      // ---- output projection + residual + LayerNorm + mask head in one pass (the pre-norm activation stays on chip)
      ProfScope ps(st, A3D_PROF_LNMASK, 0, 0, 0, 0, (int)n_total);
      k_out_ln_mask<QT><<<grid, 512, fused_lds, st>>>(samples_dev, ns, l, LW.s2c_wo_packed, LW.s2c_out_b, LW.s2c_norm_w,
                                                      LW.s2c_norm_b);
      A3D_LAUNCH_CHECK();
    } else {
      for (int si = 0; si < ns; ++si) {
        Prepared& p = P[si];
        const float* src = l == 0 ? p.feats : (((l - 1) & 1) ? p.bufD : p.bufC);
        float* Y = (l & 1) ? p.bufD : p.bufC;
        const int K = p.hm.K;
        const size_t lnm_lds = ((size_t)QP * 132 + 4 * 16 * (p.L.qp + 1) + 4 * 16 * (K + 1)) * 4 + (size_t)(K + 1) * 4;
        rc = a3d_linear(p.bufB, D, nullptr, 0, p.n, D, D, LW.s2c_wo_packed, nullptr, LW.s2c_out_b, src, D, 0, Y, D, nullptr, 0, st);
        if (rc) return rc;
        ProfScope ps(st, A3D_PROF_LNMASK, 0, 0, 0, 0, p.n);
        k_ln_mask<QT><<<(p.n + 63) / 64, 256, lnm_lds, st>>>(Y, p.n, LW.s2c_norm_w, LW.s2c_norm_b, p.B.E, p.hm.nq, p.meta->qrange,
                                                          p.hm.n_fg, K, p.logits + (size_t)l * p.n * (K + 1), p.labels,
                                                          p.counts + (size_t)l * (A3D_MAX_QUERIES + 1), nblk);
        A3D_LAUNCH_CHECK();
      }
    }
  }
  return A3D_OK;
}

// ---- the fused wide tier (decoder_wide.h): 65 .. 224 queries, QT = padded queries / 16 of the three point-side kernels; the
// query side runs its 64-query blocks on buffers of L.qp rows as before
template <int QT>
static int run_decoder_wide(const a3d_decoder_weights* w, Prepared* P, int ns, hipStream_t st) {
  int qp = 0;   // rows of the longest query-side buffers of the group
  for (int si = 0; si < ns; ++si) qp = std::max(qp, P[si].L.qp);
  {
    static bool big = false;
    if (!big) {
      big = true;
      const int big_lds = 160 * 1024;
      A3D_ALLOW_LDS(big_lds, k_query_block<2, 1>);
      A3D_ALLOW_LDS(big_lds, k_query_block<2, 2>);
      A3D_ALLOW_LDS(big_lds, k_out_w<QT>);      // its per-object maxima grow with the objects: past 64 KB from ~43 objects on
      if constexpr (QT == 5) {
        A3D_ALLOW_LDS(big_lds, (k_s2o_w<5, false>));
        A3D_ALLOW_LDS(big_lds, (k_s2o_w<5, true>));
      }
    }
  }
  const int n_counts = A3D_MAX_DEC_LAYERS * (A3D_MAX_QUERIES + 1);
  int64_t n_total = 0;
  int Kmax = 0, nq_max = 0;
  for (int si = 0; si < ns; ++si) {
    n_total += P[si].n;
    Kmax = P[si].hm.K > Kmax ? P[si].hm.K : Kmax;
    nq_max = P[si].hm.nq > nq_max ? P[si].hm.nq : nq_max;
  }
  const size_t stage_lds = (size_t)2 * 2 * kWTile * 4;                       // two slots of (rows, position encodings)
  const size_t out_lds = ((size_t)6 * kWTile + 2 * 256 + 2 * 32 * (Kmax + 1)) * 4 + (size_t)(Kmax + 1) * 4;   // k_out_w: O slots, rows, statistics, maxima, histogram
  if (out_lds > 160 * 1024) {
    set_error("a3d_decoder_forward: %d objects need %zu bytes of LDS in the output half", Kmax, out_lds);
    return A3D_ERR_UNSUPPORTED;
  }
  DecTables T;
  int rc = upload_tables(P, ns, true, true, st, T);
  if (rc) return rc;
  {
    ProfScope ps(st, A3D_PROF_QUERY, 0, 0, 0, 0, nq_max);
    // every row of the query-side buffers is initialised (the point-side kernels read 16 QT rows), 32 per workgroup
    k_query_init<2><<<dim3((qp + 31) / 32, ns), 512, 0, st>>>(T.qs_dev, w->bg_query_feat, w->bg_query_pos, w->time_table,
                                                      w->layers[0].c2s_in_w, w->layers[0].c2s_in_b, n_counts);
    A3D_LAUNCH_CHECK();
  }
  // the query side in blocks of 32 rows: as many as the longest query list of the call needs
  const int nblk = (nq_max + 31) / 32;
  for (int l = 0; l < w->n_layers; ++l) {
    const a3d_decoder_layer& LW = w->layers[l];
    // ---- click-to-scene; the first layer on the scene's cached keys / values when every sample has them
    bool cached0 = l == 0;
    for (int si = 0; si < ns && cached0; ++si) cached0 = P[si].kv0 != nullptr && P[si].kv0_state != 0;
    const bool qc0 = cached0;
    if (cached0) {
      int nchunk_max = 0, nqt_max = 0;
      for (int si = 0; si < ns; ++si) {
        Prepared& p = P[si];
        float* K0 = p.kv0;
        float* V0 = p.kv0 + (size_t)p.n * D;
        if (p.kv0_state == 1) {
          rc = a3d_linear(p.feats, D, p.posenc, D, p.n, D, D, LW.c2s_wk_packed, nullptr, LW.c2s_in_b + D, nullptr, 0, 0, K0, D, nullptr, 0, st);
          if (rc) return rc;
          rc = a3d_linear(p.feats, D, nullptr, 0, p.n, D, D, LW.c2s_wv_packed, nullptr, LW.c2s_in_b + 2 * D, nullptr, 0, 0, V0, D, nullptr, 0, st);
          if (rc) return rc;
          rc = a3d_linear(p.feats, D, p.posenc, D, p.n, D, D, LW.s2c_wq_packed, nullptr, LW.s2c_in_b, nullptr, 0, 0,
                          p.kv0 + (size_t)2 * p.n * D, D, nullptr, 0, st);
          if (rc) return rc;
        }
        nchunk_max = p.L.nchunk > nchunk_max ? p.L.nchunk : nchunk_max;
        const int t = (p.hm.nq + 15) / 16;
        nqt_max = t > nqt_max ? t : nqt_max;
      }
      {
        ProfScope ps(st, A3D_PROF_C2S, 0, 0, 0, 0, (int)n_total);
        // the unfused flash kernel over the cached keys / values, ALL samples in one launch: the widest sample's kernel, every
        // sample by its OWN tile count (c2s_attn_body skips the tiles past it)
        const dim3 cg(nchunk_max, ns);
        switch (wide_qt_of_tiles(nqt_max)) {
          case 1: k_c2s_attn_b<1><<<cg, 512, 0, st>>>(T.samples_dev); break;
          case 2: k_c2s_attn_b<2><<<cg, 512, 0, st>>>(T.samples_dev); break;
          case 3: k_c2s_attn_b<3><<<cg, 512, 0, st>>>(T.samples_dev); break;
          case 4: k_c2s_attn_b<4><<<cg, 512, 0, st>>>(T.samples_dev); break;
          case 5: k_c2s_attn_b<5><<<cg, 512, 0, st>>>(T.samples_dev); break;
          case 6: k_c2s_attn_b<6><<<cg, 512, 0, st>>>(T.samples_dev); break;
          case 7: k_c2s_attn_b<7><<<cg, 512, 0, st>>>(T.samples_dev); break;
          case 8: k_c2s_attn_b<8><<<cg, 512, 0, st>>>(T.samples_dev); break;
          case 10: k_c2s_attn_b<10><<<cg, 512, 0, st>>>(T.samples_dev); break;
          case 12: k_c2s_attn_b<12><<<cg, 512, 0, st>>>(T.samples_dev); break;
          default: k_c2s_attn_b<14><<<cg, 512, 0, st>>>(T.samples_dev); break;
        }
        A3D_LAUNCH_CHECK();
      }
    } else {
      ProfScope ps(st, A3D_PROF_C2S, 0, 0, 0, 0, (int)n_total);
      k_c2s_w<QT><<<T.grid, 512, stage_lds, st>>>(T.samples_dev, ns, l, LW.c2s_wk_packed, LW.c2s_wv_packed, LW.c2s_in_b + D,
                                                  LW.c2s_in_b + 2 * D);
      A3D_LAUNCH_CHECK();
    }
    rc = launch_query_side<2>(w, l, T.qs_dev, qp, nblk, ns, nq_max, cached0, st);
    if (rc) return rc;
    // ---- scene-to-click: up to five query tiles as ONE kernel (k_s2o_w), else attention, then output projection +
    // residual + LayerNorm + mask head
    if constexpr (QT == 5) {
      if (fused_s2o()) {
        ProfScope ps(st, A3D_PROF_S2C, 0, 0, 0, 0, (int)n_total);
        const size_t s2o_lds = ((size_t)(4 + 5) * kWTile + 2 * 256 + 2 * 32 * (Kmax + 1) + ((Kmax + 1 + 3) & ~3)) * 4;
        if (qc0) k_s2o_w<5, true><<<T.grid, 512, s2o_lds, st>>>(T.samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b, LW.s2c_wo_packed,
                                                              LW.s2c_out_b, LW.s2c_norm_w, LW.s2c_norm_b, Kmax);
        else k_s2o_w<5, false><<<T.grid, 512, s2o_lds + (size_t)4 * kWTile * 4, st>>>(T.samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b,
                                                                                     LW.s2c_wo_packed, LW.s2c_out_b, LW.s2c_norm_w,
                                                                                     LW.s2c_norm_b, Kmax);
        A3D_LAUNCH_CHECK();
        continue;
      }
    }
    {
      ProfScope ps(st, A3D_PROF_S2C, 0, 0, 0, 0, (int)n_total);
      if (qc0) k_s2c_w<QT, true><<<T.grid, 512, 0, st>>>(T.samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b);
      else k_s2c_w<QT, false><<<T.grid, 512, stage_lds, st>>>(T.samples_dev, ns, l, LW.s2c_wq_packed, LW.s2c_in_b);
      A3D_LAUNCH_CHECK();
    }
    {
      ProfScope ps(st, A3D_PROF_LNMASK, 0, 0, 0, 0, (int)n_total);
      k_out_w<QT><<<T.grid, 512, out_lds, st>>>(T.samples_dev, ns, l, LW.s2c_wo_packed, LW.s2c_out_b, LW.s2c_norm_w, LW.s2c_norm_b, Kmax);
      A3D_LAUNCH_CHECK();
    }
  }
  return A3D_OK;
}

// validate one sample and lay out its workspace (a3d_decoder_forward / a3d_decoder_forward_batch)
static int prepare_sample(const a3d_decoder_weights* w, const a3d_decoder_sample& sp, Prepared& P) {
  const float* feats128_dev = sp.feats128_dev;
  const float* posenc_dev = sp.posenc_dev;
  const int64_t n = sp.n;
  const int32_t *click_row = sp.click_row, *click_obj = sp.click_obj, *click_time = sp.click_time;
  const int n_clicks = sp.n_clicks, n_objects = sp.n_objects;
  if (!feats128_dev || !posenc_dev || !sp.logits_dev || n <= 0 || n > (int64_t)1 << 30 || n_objects < 1 ||
      n_clicks < n_objects || (n_clicks && (!click_row || !click_obj || !click_time))) {
    set_error("a3d_decoder_forward: bad arguments (every object needs >= 1 click, agile3d.py:353)");
    return A3D_ERR_INVALID;
  }
  const int nq = n_clicks + w->n_bg_queries;
  if (nq > A3D_MAX_QUERIES || n_objects > 254 || n_clicks > 200) {
    set_error("a3d_decoder_forward: %d queries > %d", nq, A3D_MAX_QUERIES);
    return A3D_ERR_UNSUPPORTED;
  }
  QueryMeta& hm = P.hm;
  memset(&hm, 0, sizeof(hm));
  hm.K = n_objects;
  hm.n_bgl = w->n_bg_queries;
  // order: foreground clicks object-major, learned background queries, background clicks
  int q = 0;
  for (int o = 1; o <= n_objects; ++o) {
    hm.qrange[o] = q;
    for (int i = 0; i < n_clicks; ++i)
      if (click_obj[i] == o) {
        hm.row[q] = click_row[i];
        hm.time[q] = click_time[i];
        hm.obj[q] = o;
        ++q;
      }
    if (q == hm.qrange[o]) {
      set_error("a3d_decoder_forward: object %d has no click", o);
      return A3D_ERR_INVALID;
    }
  }
  hm.qrange[n_objects + 1] = q;
  hm.n_fg = q;
  for (int b = 0; b < w->n_bg_queries; ++b) {
    hm.row[q] = -1;
    hm.obj[q] = 0;
    ++q;
  }
  for (int i = 0; i < n_clicks; ++i)
    if (click_obj[i] == 0) {
      hm.row[q] = click_row[i];
      hm.time[q] = click_time[i];
      hm.obj[q] = 0;
      ++q;
      ++hm.n_bg_click;
    }
  if (q != nq) {
    set_error("a3d_decoder_forward: click_obj outside 0..%d", n_objects);
    return A3D_ERR_INVALID;
  }
  hm.nq = nq;
  for (int i = 0; i < nq; ++i)
    if (hm.row[i] >= n || hm.time[i] < 0 || hm.time[i] >= 200) {
      set_error("a3d_decoder_forward: click row/time out of range");
      return A3D_ERR_INVALID;
    }
  for (int i = nq; i < A3D_MAX_QUERIES; ++i) {
    hm.obj[i] = -1;
    hm.row[i] = -1;
  }
  dec_layout(n, nq, P.L);
  P.qtw = fused_wide() ? wide_qt(nq) : 0;
  if (fused_wide() && nq >= wide_from() && nq <= 64) P.qtw = 5;   // (the smallest build; the kernels take the sample's own tile count)
  if (!sp.workspace_dev || sp.workspace_bytes < P.L.total || ((uintptr_t)sp.workspace_dev & 255)) {
    set_error("a3d_decoder_forward: workspace too small or misaligned (%zu < %zu)", sp.workspace_bytes, P.L.total);
    return A3D_ERR_WORKSPACE;
  }
  P.ws = (char*)sp.workspace_dev;
  P.feats = feats128_dev;
  P.posenc = posenc_dev;
  P.n = (int)n;
  P.logits = sp.logits_dev;
  P.kv0 = sp.kv0_dev;
  P.kv0_state = sp.kv0_dev ? sp.kv0_state : 0;
  if (P.kv0_state < 0 || P.kv0_state > 2 || ((uintptr_t)sp.kv0_dev & 15) || (P.kv0_state != 0 && sp.kv0_blocks < 3)) {
    set_error("a3d_decoder_forward: bad kv0 cache (state %d, %d blocks of [n][128]; keys, values and first-layer queries = 3)",
              sp.kv0_state, sp.kv0_blocks);
    return A3D_ERR_INVALID;
  }
  P.bind();
  return A3D_OK;
}

static int dispatch_decoder(const a3d_decoder_weights* w, Prepared* P, int ns, hipStream_t st) {
  int qtw = 0;   // the wide build that holds the longest query list of the group (0: the group runs the <= 64 / unfused kernels)
  if (P[0].qtw > 0)
    for (int i = 0; i < ns; ++i) qtw = std::max(qtw, std::max(P[i].qtw, wide_qt(P[i].hm.nq)));
  switch (qtw) {
    case 5: return run_decoder_wide<5>(w, P, ns, st);
    case 6: return run_decoder_wide<6>(w, P, ns, st);
    case 7: return run_decoder_wide<7>(w, P, ns, st);
    case 8: return run_decoder_wide<8>(w, P, ns, st);
    case 10: return run_decoder_wide<10>(w, P, ns, st);
    case 12: return run_decoder_wide<12>(w, P, ns, st);
    case 14: return run_decoder_wide<14>(w, P, ns, st);
    default: break;
  }
  switch (P[0].L.qp) {
    case 16: return run_decoder<1>(w, P, ns, st);
    case 32: return run_decoder<2>(w, P, ns, st);
    case 48: return run_decoder<3>(w, P, ns, st);
    default: return run_decoder<4>(w, P, ns, st);
  }
}

extern "C" int a3d_decoder_forward_batch(const a3d_decoder_weights* w, const a3d_decoder_sample* samples, int n_samples,
                                         void* stream) {
  int rc = check_weights(w);
  if (rc) return rc;
  rc = check_packs(w);
  if (rc) return rc;
  if (!samples || n_samples < 1 || n_samples > 1024) {
    set_error("a3d_decoder_forward_batch: bad arguments");
    return A3D_ERR_INVALID;
  }
  if (w->dim_ff > 4096) {
    set_error("a3d_decoder_forward: dim_feedforward > 4096");
    return A3D_ERR_UNSUPPORTED;
  }
  std::vector<Prepared> P((size_t)n_samples);
  for (int i = 0; i < n_samples; ++i) {
    rc = prepare_sample(w, samples[i], P[(size_t)i]);
    if (rc) return rc;
  }
  hipStream_t st = (hipStream_t)stream;
  // Launch groups.  The samples that the fused wide tier serves (more than 64 queries, decoder_wide.h) go through its kernels
  // TOGETHER, whatever their query counts (the kernels take each sample's tile count from the table; the build is the one
  // that holds the longest list): one launch per kernel and layer instead of one per sample -- four 80 k-point samples of a
  // training click round pay the persistent kernels' start-up and tail once.  The others: consecutive samples with the same
  // padded query count share the <= 64-query kernels.  Several groups are independent chains of launches on their own
  // workspaces: group g goes on side stream g mod 4 (forked from / joined to the caller's stream with events), group 0 stays.
  // A3D_WIDE_MERGE=0: one group per padded query count as before; 2: when the samples of a call do not share ONE padded
  // query count, all of them go through the wide kernels.
  static int merge_mode = -1;
  if (merge_mode < 0) {
    const char* e = getenv("A3D_WIDE_MERGE");
    merge_mode = e ? atoi(e) : 1;
  }
  {
    bool mixed = false;
    for (int i = 1; i < n_samples; ++i) mixed = mixed || P[(size_t)i].L.qp != P[0].L.qp || P[(size_t)i].qtw != P[0].qtw;
    if (fused_wide() && merge_mode == 2 && mixed)
      for (int i = 0; i < n_samples; ++i)
        if (P[(size_t)i].qtw == 0 && P[(size_t)i].hm.nq <= 224) P[(size_t)i].qtw = 5;
    if (merge_mode != 0) {   // wide samples first (stable): they form the leading group(s)
      std::vector<Prepared> Q;
      Q.reserve((size_t)n_samples);
      for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < n_samples; ++i)
          if ((P[(size_t)i].qtw > 0) == (pass == 0)) Q.push_back(P[(size_t)i]);
      P.swap(Q);
    }
  }
  auto same_group = [&](const Prepared& a, const Prepared& b) {
    if (merge_mode != 0 && a.qtw > 0 && b.qtw > 0) return true;
    return a.L.qp == b.L.qp && a.qtw == b.qtw;
  };
  struct Side {
    hipStream_t s[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, done[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ok = false, tried = false;
  };
  static thread_local Side sides[16];   // per device: a stream belongs to the device that was current when it was made
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  Side& side = sides[dev_id & 15];
  int n_groups = 0;
  for (int i = 0; i < n_samples;) {
    int e = i + 1;
    while (e < n_samples && e - i < kMaxBatchSamples && same_group(P[(size_t)i], P[(size_t)e])) ++e;
    ++n_groups;
    i = e;
  }
  bool use_side = n_groups > 1;
  if (use_side && !side.tried) {
    side.tried = true;
    bool ok = hipEventCreateWithFlags(&side.fork, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; ok && k < 4; ++k)
      ok = hipStreamCreateWithFlags(&side.s[k], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&side.done[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    side.ok = ok;
  }
  use_side = use_side && side.ok;
  bool used[4] = {false, false, false, false};
  if (use_side) A3D_HIP_CHECK(hipEventRecord(side.fork, st));
  int g = 0, rc_all = A3D_OK;
  for (int i = 0; i < n_samples && rc_all == A3D_OK;) {
    int e = i + 1;
    while (e < n_samples && e - i < kMaxBatchSamples && same_group(P[(size_t)i], P[(size_t)e])) ++e;
    hipStream_t gs = st;
    if (use_side && g > 0) {
      const int k = (g - 1) & 3;
      gs = side.s[k];
      if (!used[k]) {
        used[k] = true;
        if (hipStreamWaitEvent(gs, side.fork, 0) != hipSuccess) rc_all = A3D_ERR_HIP;
      }
    }
    if (rc_all == A3D_OK) rc_all = dispatch_decoder(w, &P[(size_t)i], e - i, gs);
    ++g;
    i = e;
  }
  // join: the caller's stream continues after every side stream's work (also when a later group failed)
  for (int k = 0; k < 4; ++k)
    if (used[k]) {
      if (hipEventRecord(side.done[k], side.s[k]) != hipSuccess || hipStreamWaitEvent(st, side.done[k], 0) != hipSuccess) {
        (void)hipStreamSynchronize(side.s[k]);
        if (rc_all == A3D_OK) {
          set_error("a3d_decoder_forward_batch: joining a side stream failed");
          rc_all = A3D_ERR_HIP;
        }
      }
    }
  return rc_all;
}

extern "C" int a3d_decoder_forward(const a3d_decoder_weights* w, const float* feats128_dev, const float* xyz_dev,
                                   const float* posenc_dev, const float* minmax_dev, int64_t n, const int32_t* click_row, const int32_t* click_obj,
                                   const int32_t* click_time, int n_clicks, int n_objects, float* logits_dev,
                                   void* workspace_dev, size_t workspace_bytes, void* stream) {
  (void)xyz_dev;
  (void)minmax_dev;   // click encodings equal the scene encoding rows
  a3d_decoder_sample sp;
  sp.feats128_dev = feats128_dev;
  sp.posenc_dev = posenc_dev;
  sp.n = n;
  sp.click_row = click_row;
  sp.click_obj = click_obj;
  sp.click_time = click_time;
  sp.n_clicks = n_clicks;
  sp.n_objects = n_objects;
  sp.logits_dev = logits_dev;
  sp.workspace_dev = workspace_dev;
  sp.workspace_bytes = workspace_bytes;
  sp.kv0_dev = nullptr;
  sp.kv0_state = 0;
  sp.kv0_blocks = 0;
  return a3d_decoder_forward_batch(w, &sp, 1, stream);
}
