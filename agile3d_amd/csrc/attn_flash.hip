// libagile3d_hip -- flash-formulation attention WITH its backward for the training path of the decoder (SURVEY.md
// section 8 row f-2): nn.MultiheadAttention as attention_block.py:25-26,88-94 composes it, for the two attentions whose
// score matrix has the N points on one side (8 heads x Q queries x 80 k points = 50-270 MB per matrix when materialised,
// several per layer and direction -- what attn_train.hip does, and what made the training decoder scale with Q):
//
//   click-to-scene  (few queries over MANY keys, masked):  a3d_flash_c2s_forward / _backward
//   scene-to-click  (MANY queries over few keys):          a3d_flash_s2c_forward / _backward
//
// Nothing of size [heads, Q, N] is ever written.  The forward pass keeps the softmax statistics (row maximum and
// row sum) per (head, query); the backward pass recomputes the probabilities tile by tile from them:
//   P = exp(S - m) / l,  dP = dO V^T,  D = rowsum(dO * O),  dS = P (dP - D),  dQ = dS K,  dK = dS^T Q,  dV = P^T dO.
// All products are 16x16x4 fp32 MFMAs (exact fp32, the reference's arithmetic); one wave = one head over a chunk of
// points, 8 waves = the 8 heads of the same chunk; a workgroup walks chunks wg, wg + G, ...  Reductions over the N points
// (dQ of click-to-scene, dK / dV of scene-to-click) add up per workgroup in its own slab (sequential read-modify-write)
// and a two-level reduction adds the <= 256 slabs in a fixed order: deterministic.
// The scores need each tile in two register layouts (rows x columns and columns x rows, because an MFMA contracts over
// the lane-group index of BOTH operands): they are simply computed twice -- 8 of the 28 MFMAs per tile.
#include "common.h"

namespace a3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
namespace {
constexpr int FD = 128, FH = 8, FDH = 16;
constexpr int kFlChunk = 64;        // points per workgroup (four 16-point groups: their accumulators stay in registers)
constexpr float kFlNeg = -1e30f;

__device__ __forceinline__ float fl_rows_max(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float fl_rows_sum(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// f32x4 = row[r][c .. c+3] with rows clamped (callers mask what lies beyond)
__device__ __forceinline__ f32x4 ld4(const float* base, int64_t row, int64_t nrows, int col) {
  const int64_t r = row < nrows ? row : nrows - 1;
  return *(const f32x4*)(base + r * FD + col);
}
#define FL_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------------ click-to-scene
// forward: partial flash state of one (workgroup, head) for every query: part[wg][h][q][18] = m, l, acc[16].  A workgroup
// walks the 64-key chunks wg, wg + G, ... (G = gridDim.x workgroups), the running state of a query tile lives in registers
// across all of them.
__global__ void __launch_bounds__(512) k_fl_c2s_fwd(const float* __restrict__ qs, const float* __restrict__ K,
                                                    const float* __restrict__ V, const unsigned char* __restrict__ mask,
                                                    int Lq, int Lk, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int nchunk = (Lk + kFlChunk - 1) / kFlChunk;
  const int nqt = (Lq + 15) / 16;
  for (int qt = 0; qt < nqt; ++qt) {
    const int qrow = qt * 16 + j;                                   // this lane's query in the (keys x queries) layout
    const f32x4 qf = ld4(qs, qrow, Lq, h * FDH + 4 * g);
    float m = kFlNeg, l = 0.f;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};                        // O^T[d = 4g+t][query j]
    for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
      const int pbeg = ch * kFlChunk, pend = min(Lk, pbeg + kFlChunk);
      for (int p0 = pbeg; p0 < pend; p0 += 16) {
        const f32x4 kf = ld4(K, p0 + j, Lk, h * FDH + 4 * g);       // A: [key j][d 4g+t]
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) s = FL_MFMA(kf[t], qf[t], s);   // s[t] = S[key 4g+t][query j]
        float mx = kFlNeg;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int pr = p0 + 4 * g + t;
          const bool blocked = pr >= Lk || qrow >= Lq || (mask && mask[(size_t)qrow * Lk + pr]);
          s[t] = blocked ? kFlNeg : s[t];
          mx = fmaxf(mx, s[t]);
        }
        mx = fl_rows_max(mx);
        const float mnew = fmaxf(m, mx);
        const float sc = expf(m - mnew);
        m = mnew;
        f32x4 p;
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          p[t] = s[t] <= kFlNeg ? 0.f : expf(s[t] - mnew);
          ps += p[t];
        }
        l = l * sc + ps;
        acc *= sc;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int pr = min(p0 + 4 * g + t, Lk - 1);
          const float vf = V[(size_t)pr * FD + h * FDH + j];        // A: [d j][key 4g+t]
          acc = FL_MFMA(vf, p[t], acc);
        }
      }
    }
    l = fl_rows_sum(l);
    float* pq = part + (((size_t)blockIdx.x * FH + h) * Lq + min(qrow, Lq - 1)) * 18;
    if (qrow < Lq) {
      if (g == 0) {
        pq[0] = m;
        pq[1] = l;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) pq[2 + 4 * g + t] = acc[t];
    }
  }
}

// merge the per-chunk partials of one (query, head): O[q][16h ..], stats[0][h][q] = m, stats[1][h][q] = l; one wave per pair,
// chunks in a fixed order per lane, lanes by a butterfly: deterministic
__global__ void __launch_bounds__(64) k_fl_c2s_combine(const float* __restrict__ part, int nchunk, int Lq, float* __restrict__ O,
                                                       float* __restrict__ stats) {
  const int q = blockIdx.x / FH, h = blockIdx.x % FH, lane = threadIdx.x;
  float m = kFlNeg, l = 0.f, o[FDH];
#pragma unroll
  for (int d = 0; d < FDH; ++d) o[d] = 0.f;
  for (int ch = lane; ch < nchunk; ch += 64) {
    const float* p = part + (((size_t)ch * FH + h) * Lq + q) * 18;
    const float pm = p[0];
    const float mn = fmaxf(m, pm);
    const float a = expf(m - mn), b = expf(pm - mn);
    l = l * a + p[1] * b;
#pragma unroll
    for (int d = 0; d < FDH; ++d) o[d] = o[d] * a + p[2 + d] * b;
    m = mn;
  }
  float M = m;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
  const float sc = expf(m - M);
  l *= sc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
#pragma unroll
  for (int d = 0; d < FDH; ++d) {
    float v = o[d] * sc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == d) O[(size_t)q * FD + h * FDH + d] = v / l;
  }
  if (lane == 0) {
    stats[(size_t)h * Lq + q] = M;
    stats[(size_t)(FH + h) * Lq + q] = l;
  }
}

// Dsum[h][q] = sum_d dO[q][16h+d] O[q][16h+d]   (rows = the SHORT side of either attention)
__global__ void k_fl_rowdot(const float* __restrict__ dO, const float* __restrict__ O, int L, float* __restrict__ Ds) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= L * FH) return;
  const int q = e / FH, h = e % FH;
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < FDH; ++d) s += dO[(size_t)q * FD + h * FDH + d] * O[(size_t)q * FD + h * FDH + d];
  Ds[(size_t)h * L + q] = s;
}

// backward: a workgroup walks the 64-key chunks wg, wg + G, ...: dK, dV rows of a chunk are final when it is done; the dQ
// contributions of its chunks add up in the workgroup's own slab dqp[wg][h][q][16] (read-modify-write by the one wave that
// owns the (head, query tile): sequential, deterministic)
__global__ void __launch_bounds__(512) k_fl_c2s_bwd(const float* __restrict__ qs, const float* __restrict__ K,
                                                    const float* __restrict__ V, const unsigned char* __restrict__ mask,
                                                    int Lq, int Lk, const float* __restrict__ stats,
                                                    const float* __restrict__ Ds, const float* __restrict__ dO,
                                                    float* __restrict__ dqp, float* __restrict__ dK, float* __restrict__ dV) {
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int nchunk = (Lk + kFlChunk - 1) / kFlChunk;
  const int nqt = (Lq + 15) / 16;
  constexpr int NG = kFlChunk / 16;
  const float* Ms = stats + (size_t)h * Lq;
  const float* Ls = stats + (size_t)(FH + h) * Lq;
  const float* Dh = Ds + (size_t)h * Lq;
  bool first = true;
  for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x, first = false) {
    const int pbeg = ch * kFlChunk, pend = min(Lk, pbeg + kFlChunk);
    f32x4 adk[NG], adv[NG];                                         // dK^T / dV^T [d 4g+t][key j] of the chunk's groups
#pragma unroll
    for (int i = 0; i < NG; ++i) adk[i] = adv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int qt = 0; qt < nqt; ++qt) {
      const int qj = qt * 16 + j, qjc = min(qj, Lq - 1);            // this lane's query as a COLUMN (keys x queries) ...
      const f32x4 qf = ld4(qs, qj, Lq, h * FDH + 4 * g);            // q[query j][d 4g+t]
      const f32x4 dof = ld4(dO, qj, Lq, h * FDH + 4 * g);           // dO[query j][d 4g+t]
      const float mj = Ms[qjc], rlj = 1.f / Ls[qjc], Dj = Dh[qjc];
      float mq[4], rlq[4], Dq[4], qT[4], doT[4];                    // ... and its four queries as ROWS (queries x keys)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int qr = min(qt * 16 + 4 * g + t, Lq - 1);
        mq[t] = Ms[qr];
        rlq[t] = 1.f / Ls[qr];
        Dq[t] = Dh[qr];
        qT[t] = qs[(size_t)qr * FD + h * FDH + j];                  // A: [d j][query 4g+t]
        doT[t] = dO[(size_t)qr * FD + h * FDH + j];
      }
      f32x4 adq = (f32x4){0.f, 0.f, 0.f, 0.f};                      // dQ^T[d 4g+t][query j]
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        const int p0 = pbeg + gi * 16;
        if (p0 >= pend) break;
        const f32x4 kf = ld4(K, p0 + j, Lk, h * FDH + 4 * g);       // K[key j][d 4g+t]
        const f32x4 vr = ld4(V, p0 + j, Lk, h * FDH + 4 * g);       // V[key j][d 4g+t]
        // scores and dP in both layouts
        f32x4 s_kq = (f32x4){0.f, 0.f, 0.f, 0.f}, s_qk = s_kq, dp_kq = s_kq, dp_qk = s_kq;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s_kq = FL_MFMA(kf[t], qf[t], s_kq);                       // [key 4g+t][query j]
          s_qk = FL_MFMA(qf[t], kf[t], s_qk);                       // [query 4g+t][key j]
          dp_kq = FL_MFMA(vr[t], dof[t], dp_kq);
          dp_qk = FL_MFMA(dof[t], vr[t], dp_qk);
        }
        f32x4 p_qk, ds_qk, ds_kq;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          {   // keys x queries: key 4g+t, query j
            const int pr = p0 + 4 * g + t;
            const bool blocked = pr >= Lk || qj >= Lq || (mask && mask[(size_t)qjc * Lk + pr]);
            const float p = blocked ? 0.f : expf(s_kq[t] - mj) * rlj;
            ds_kq[t] = p * (dp_kq[t] - Dj);
          }
          {   // queries x keys: query 4g+t, key j
            const int qr = qt * 16 + 4 * g + t, pr = p0 + j;
            const bool blocked = pr >= Lk || qr >= Lq || (mask && mask[(size_t)min(qr, Lq - 1) * Lk + min(pr, Lk - 1)]);
            const float p = blocked ? 0.f : expf(s_qk[t] - mq[t]) * rlq[t];
            p_qk[t] = p;
            ds_qk[t] = p * (dp_qk[t] - Dq[t]);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          adv[gi] = FL_MFMA(doT[t], p_qk[t], adv[gi]);              // dV^T[d][key] += dO^T[d][q] P[q][key]
          adk[gi] = FL_MFMA(qT[t], ds_qk[t], adk[gi]);              // dK^T[d][key] += q^T[d][q] dS[q][key]
          const int pr = min(p0 + 4 * g + t, Lk - 1);
          const float kT = K[(size_t)pr * FD + h * FDH + j];        // A: [d j][key 4g+t]
          adq = FL_MFMA(kT, ds_kq[t], adq);                         // dQ^T[d][query] += K^T[d][key] dS[key][query]
        }
      }
      if (qj < Lq) {
        f32x4* slot = (f32x4*)(dqp + (((size_t)blockIdx.x * FH + h) * Lq + qj) * FDH + 4 * g);
        *slot = first ? adq : *slot + adq;
      }
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int pr = pbeg + gi * 16 + j;
      if (pr < pend) {
        *(f32x4*)(dK + (size_t)pr * FD + h * FDH + 4 * g) = adk[gi];
        *(f32x4*)(dV + (size_t)pr * FD + h * FDH + 4 * g) = adv[gi];
      }
    }
  }
}

// out[r][16h + d] = sum over the partial slabs (ascending) of part[slab][h][r][d], in two levels so that the sums of one
// output run in parallel: slice y adds slabs [y S, (y + 1) S) into tmp[y], then the slices are added in order
__global__ void k_fl_reduce_slices(const float* __restrict__ part, int nslab, int per_slice, int L, float* __restrict__ tmp) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = L * FD;
  if (e >= total) return;
  const int hh = e / (L * FDH), rem = e - hh * (L * FDH);          // element order = the slab's own order: [h][r][d]
  const int s0 = blockIdx.y * per_slice, s1 = min(nslab, s0 + per_slice);
  float s = 0.f;
  for (int sl = s0; sl < s1; ++sl) s += part[((size_t)sl * FH + hh) * L * FDH + rem];
  tmp[(size_t)blockIdx.y * total + e] = s;
}
__global__ void k_fl_reduce_final(const float* __restrict__ tmp, int nslice, int L, float* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = L * FD;
  if (e >= total) return;
  const int hh = e / (L * FDH), rem = e - hh * (L * FDH), r = rem / FDH, d = rem % FDH;
  float s = 0.f;
  for (int y = 0; y < nslice; ++y) s += tmp[(size_t)y * total + e];
  out[(size_t)r * FD + hh * FDH + d] = s;
}

// ------------------------------------------------------------------------------------------------ scene-to-click
// forward: per 16-point group and head, online softmax over the key tiles; stats[n][h][2] = m, l
__global__ void __launch_bounds__(512) k_fl_s2c_fwd(const float* __restrict__ qs, const float* __restrict__ K,
                                                    const float* __restrict__ V, int Lq, int Lk, float* __restrict__ O,
                                                    float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int nkt = (Lk + 15) / 16;
  const int pbeg = blockIdx.x * kFlChunk, pend = min(Lq, pbeg + kFlChunk);
  for (int p0 = pbeg; p0 < pend; p0 += 16) {
    const f32x4 qf = ld4(qs, p0 + j, Lq, h * FDH + 4 * g);          // B: [d 4g+t][point j]
    float m = kFlNeg, l = 0.f;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};                        // O^T[d 4g+t][point j]
    for (int kt = 0; kt < nkt; ++kt) {
      const f32x4 kf = ld4(K, kt * 16 + j, Lk, h * FDH + 4 * g);    // A: [key j][d 4g+t]
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) s = FL_MFMA(kf[t], qf[t], s);     // [key 4g+t][point j]
      float mx = kFlNeg;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (kt * 16 + 4 * g + t >= Lk) s[t] = kFlNeg;
        mx = fmaxf(mx, s[t]);
      }
      mx = fl_rows_max(mx);
      const float mnew = fmaxf(m, mx);
      const float sc = expf(m - mnew);
      m = mnew;
      f32x4 p;
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        p[t] = s[t] <= kFlNeg ? 0.f : expf(s[t] - mnew);
        ps += p[t];
      }
      l = l * sc + ps;
      acc *= sc;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int kr = min(kt * 16 + 4 * g + t, Lk - 1);
        acc = FL_MFMA(V[(size_t)kr * FD + h * FDH + j], p[t], acc); // A: [d j][key 4g+t]
      }
    }
    l = fl_rows_sum(l);
    const float rl = 1.f / l;
    if (p0 + j < Lq) {
      *(f32x4*)(O + (size_t)(p0 + j) * FD + h * FDH + 4 * g) = acc * rl;
      if (g == 0) {
        stats[((size_t)(p0 + j) * FH + h) * 2] = m;
        stats[((size_t)(p0 + j) * FH + h) * 2 + 1] = l;
      }
    }
  }
}

// backward: a workgroup walks the 64-point chunks wg, wg + G, ...: dQ rows of a chunk are final when it is done; the
// dK / dV contributions of its chunks add up in the workgroup's own slabs dkp / dvp[wg][h][key][16]
__global__ void __launch_bounds__(512) k_fl_s2c_bwd(const float* __restrict__ qs, const float* __restrict__ K,
                                                    const float* __restrict__ V, int Lq, int Lk, const float* __restrict__ O,
                                                    const float* __restrict__ stats, const float* __restrict__ dO,
                                                    float* __restrict__ dQ, float* __restrict__ dkp, float* __restrict__ dvp) {
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int nkt = (Lk + 15) / 16;
  const int nchunk = (Lq + kFlChunk - 1) / kFlChunk;
  constexpr int NG = kFlChunk / 16;
  bool first = true;
  for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x, first = false) {
    const int pbeg = ch * kFlChunk, pend = min(Lq, pbeg + kFlChunk);
    f32x4 adq[NG];                                                  // dQ^T[d 4g+t][point j] of the chunk's groups
#pragma unroll
    for (int i = 0; i < NG; ++i) adq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nkt; ++kt) {
      const f32x4 kf = ld4(K, kt * 16 + j, Lk, h * FDH + 4 * g);    // K[key j][d 4g+t]
      const f32x4 vr = ld4(V, kt * 16 + j, Lk, h * FDH + 4 * g);    // V[key j][d 4g+t]
      float kT[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) kT[t] = K[(size_t)min(kt * 16 + 4 * g + t, Lk - 1) * FD + h * FDH + j];   // A: [d j][key 4g+t]
      f32x4 adk = (f32x4){0.f, 0.f, 0.f, 0.f}, adv = adk;           // dK^T / dV^T [d 4g+t][key j]
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        const int p0 = pbeg + gi * 16;
        if (p0 >= pend) break;
        const int pj = min(p0 + j, Lq - 1);
        const f32x4 qf = ld4(qs, p0 + j, Lq, h * FDH + 4 * g);      // q[point j][d 4g+t]
        const f32x4 dof = ld4(dO, p0 + j, Lq, h * FDH + 4 * g);     // dO[point j][d 4g+t]
        const f32x4 of = ld4(O, p0 + j, Lq, h * FDH + 4 * g);
        const float mj = stats[((size_t)pj * FH + h) * 2], rlj = 1.f / stats[((size_t)pj * FH + h) * 2 + 1];
        float Dj = dof[0] * of[0] + dof[1] * of[1] + dof[2] * of[2] + dof[3] * of[3];
        Dj = fl_rows_sum(Dj);                                        // sum_d dO[point j][16h+d] O[point j][16h+d]
        float mp[4], rlp[4], Dp[4], qT[4], doT[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int src = 4 * g + t;                                 // lane holding point 4g+t as ITS point j (any row)
          mp[t] = __shfl(mj, src, 64);
          rlp[t] = __shfl(rlj, src, 64);
          Dp[t] = __shfl(Dj, src, 64);
          const int pr = min(p0 + 4 * g + t, Lq - 1);
          qT[t] = qs[(size_t)pr * FD + h * FDH + j];                 // A: [d j][point 4g+t]
          doT[t] = dO[(size_t)pr * FD + h * FDH + j];
        }
        f32x4 s_kp = (f32x4){0.f, 0.f, 0.f, 0.f}, s_pk = s_kp, dp_kp = s_kp, dp_pk = s_kp;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s_kp = FL_MFMA(kf[t], qf[t], s_kp);                        // [key 4g+t][point j]
          s_pk = FL_MFMA(qf[t], kf[t], s_pk);                        // [point 4g+t][key j]
          dp_kp = FL_MFMA(vr[t], dof[t], dp_kp);
          dp_pk = FL_MFMA(dof[t], vr[t], dp_pk);
        }
        f32x4 ds_kp, p_pk, ds_pk;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          {   // keys x points
            const bool off = kt * 16 + 4 * g + t >= Lk || p0 + j >= Lq;
            const float p = off ? 0.f : expf(s_kp[t] - mj) * rlj;
            ds_kp[t] = p * (dp_kp[t] - Dj);
          }
          {   // points x keys
            const bool off = kt * 16 + j >= Lk || p0 + 4 * g + t >= Lq;
            const float p = off ? 0.f : expf(s_pk[t] - mp[t]) * rlp[t];
            p_pk[t] = p;
            ds_pk[t] = p * (dp_pk[t] - Dp[t]);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          adq[gi] = FL_MFMA(kT[t], ds_kp[t], adq[gi]);               // dQ^T[d][point] += K^T[d][key] dS[key][point]
          adk = FL_MFMA(qT[t], ds_pk[t], adk);                       // dK^T[d][key] += q^T[d][point] dS[point][key]
          adv = FL_MFMA(doT[t], p_pk[t], adv);                       // dV^T[d][key] += dO^T[d][point] P[point][key]
        }
      }
      const int kj = kt * 16 + j;
      if (kj < Lk) {
        f32x4* sk = (f32x4*)(dkp + (((size_t)blockIdx.x * FH + h) * Lk + kj) * FDH + 4 * g);
        f32x4* sv = (f32x4*)(dvp + (((size_t)blockIdx.x * FH + h) * Lk + kj) * FDH + 4 * g);
        *sk = first ? adk : *sk + adk;
        *sv = first ? adv : *sv + adv;
      }
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int pr = pbeg + gi * 16 + j;
      if (pr < pend) *(f32x4*)(dQ + (size_t)pr * FD + h * FDH + 4 * g) = adq[gi];
    }
  }
}

// workgroups of the persistent kernels: one per chunk up to 256 (one 8-wave workgroup per CU at ~200 registers)
inline int fl_grid(int64_t n) {
  const int64_t c = (n + kFlChunk - 1) / kFlChunk;
  return (int)(c < 256 ? c : 256);
}
constexpr int kFlSlices = 16;
// out [L][128] = sum of the slabs part[nslab][8][L][16], two levels (tmp: [kFlSlices][L * 128])
inline void fl_reduce(const float* part, int nslab, int L, float* tmp, float* out, hipStream_t st) {
  const int per = (nslab + kFlSlices - 1) / kFlSlices, ny = (nslab + per - 1) / per;
  const unsigned bx = (unsigned)((L * FD + 255) / 256);
  k_fl_reduce_slices<<<dim3(bx, ny), 256, 0, st>>>(part, nslab, per, L, tmp);
  k_fl_reduce_final<<<bx, 256, 0, st>>>(tmp, ny, L, out);
}
}  // namespace
}  // namespace a3d

using namespace a3d;

// workspace: forward partials [G][8][Lq][18] (the backward's dQ slabs [G][8][Lq][16] fit inside) + Dsum [8][Lq] + the
// reduction's slices [16][Lq * 128]
extern "C" size_t a3d_flash_c2s_workspace_bytes(int64_t Lq, int64_t Lk) {
  if (Lq <= 0 || Lk <= 0) return 0;
  return align256((size_t)fl_grid(Lk) * FH * Lq * 18 * 4) + align256((size_t)FH * Lq * 4) +
         align256((size_t)kFlSlices * Lq * FD * 4) + 256;
}

static int fl_check(const char* what, const void* a, const void* b, const void* c, int64_t Lq, int64_t Lk, const void* ws,
                    size_t ws_bytes, size_t need) {
  if (!a || !b || !c || Lq <= 0 || Lk <= 0 || Lq > (1 << 30) || Lk > (1 << 30)) {
    set_error("%s: bad arguments", what);
    return A3D_ERR_INVALID;
  }
  if (need && (!ws || ws_bytes < need || ((uintptr_t)ws & 255))) {
    set_error("%s: workspace too small or misaligned (%zu < %zu)", what, ws_bytes, need);
    return A3D_ERR_WORKSPACE;
  }
  return A3D_OK;
}

extern "C" int a3d_flash_c2s_forward(const float* q_scaled_dev, const float* k_dev, const float* v_dev,
                                     const unsigned char* mask_dev, int64_t Lq, int64_t Lk, float* o_dev, float* stats_dev,
                                     void* workspace_dev, size_t workspace_bytes, void* stream) {
  int rc = fl_check("a3d_flash_c2s_forward", q_scaled_dev, k_dev, v_dev, Lq, Lk, workspace_dev, workspace_bytes,
                    a3d_flash_c2s_workspace_bytes(Lq, Lk));
  if (rc) return rc;
  if (!o_dev || !stats_dev) {
    set_error("a3d_flash_c2s_forward: null output");
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const int G = fl_grid(Lk);
  float* part = (float*)workspace_dev;
  k_fl_c2s_fwd<<<G, 512, 0, st>>>(q_scaled_dev, k_dev, v_dev, mask_dev, (int)Lq, (int)Lk, part);
  k_fl_c2s_combine<<<(unsigned)(Lq * FH), 64, 0, st>>>(part, G, (int)Lq, o_dev, stats_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_flash_c2s_backward(const float* q_scaled_dev, const float* k_dev, const float* v_dev,
                                      const unsigned char* mask_dev, int64_t Lq, int64_t Lk, const float* o_dev,
                                      const float* stats_dev, const float* d_o_dev, float* dq_scaled_dev, float* dk_dev,
                                      float* dv_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  int rc = fl_check("a3d_flash_c2s_backward", q_scaled_dev, k_dev, v_dev, Lq, Lk, workspace_dev, workspace_bytes,
                    a3d_flash_c2s_workspace_bytes(Lq, Lk));
  if (rc) return rc;
  if (!o_dev || !stats_dev || !d_o_dev || !dq_scaled_dev || !dk_dev || !dv_dev) {
    set_error("a3d_flash_c2s_backward: null argument");
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const int G = fl_grid(Lk);
  float* dqp = (float*)workspace_dev;
  float* Ds = (float*)((char*)workspace_dev + align256((size_t)G * FH * Lq * 18 * 4));
  float* tmp = (float*)((char*)Ds + align256((size_t)FH * Lq * 4));
  k_fl_rowdot<<<(unsigned)((Lq * FH + 255) / 256), 256, 0, st>>>(d_o_dev, o_dev, (int)Lq, Ds);
  k_fl_c2s_bwd<<<G, 512, 0, st>>>(q_scaled_dev, k_dev, v_dev, mask_dev, (int)Lq, (int)Lk, stats_dev, Ds, d_o_dev, dqp, dk_dev,
                                  dv_dev);
  fl_reduce(dqp, G, (int)Lq, tmp, dq_scaled_dev, st);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

// workspace of the backward: dK and dV slabs [G][8][Lk][16] each + the reduction's slices [16][Lk * 128]
extern "C" size_t a3d_flash_s2c_workspace_bytes(int64_t Lq, int64_t Lk) {
  if (Lq <= 0 || Lk <= 0) return 0;
  return 2 * align256((size_t)fl_grid(Lq) * FH * Lk * FDH * 4) + align256((size_t)kFlSlices * Lk * FD * 4) + 256;
}

extern "C" int a3d_flash_s2c_forward(const float* q_scaled_dev, const float* k_dev, const float* v_dev, int64_t Lq, int64_t Lk,
                                     float* o_dev, float* stats_dev, void* stream) {
  int rc = fl_check("a3d_flash_s2c_forward", q_scaled_dev, k_dev, v_dev, Lq, Lk, nullptr, 0, 0);
  if (rc) return rc;
  if (!o_dev || !stats_dev) {
    set_error("a3d_flash_s2c_forward: null output");
    return A3D_ERR_INVALID;
  }
  const int nchunk = (int)((Lq + kFlChunk - 1) / kFlChunk);
  k_fl_s2c_fwd<<<nchunk, 512, 0, (hipStream_t)stream>>>(q_scaled_dev, k_dev, v_dev, (int)Lq, (int)Lk, o_dev, stats_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" int a3d_flash_s2c_backward(const float* q_scaled_dev, const float* k_dev, const float* v_dev, int64_t Lq, int64_t Lk,
                                      const float* o_dev, const float* stats_dev, const float* d_o_dev, float* dq_scaled_dev,
                                      float* dk_dev, float* dv_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  int rc = fl_check("a3d_flash_s2c_backward", q_scaled_dev, k_dev, v_dev, Lq, Lk, workspace_dev, workspace_bytes,
                    a3d_flash_s2c_workspace_bytes(Lq, Lk));
  if (rc) return rc;
  if (!o_dev || !stats_dev || !d_o_dev || !dq_scaled_dev || !dk_dev || !dv_dev) {
    set_error("a3d_flash_s2c_backward: null argument");
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const int G = fl_grid(Lq);
  const size_t half = align256((size_t)G * FH * Lk * FDH * 4);
  float* dkp = (float*)workspace_dev;
  float* dvp = (float*)((char*)workspace_dev + half);
  float* tmp = (float*)((char*)workspace_dev + 2 * half);
  k_fl_s2c_bwd<<<G, 512, 0, st>>>(q_scaled_dev, k_dev, v_dev, (int)Lq, (int)Lk, o_dev, stats_dev, d_o_dev, dq_scaled_dev, dkp,
                                  dvp);
  fl_reduce(dkp, G, (int)Lk, tmp, dk_dev, st);
  fl_reduce(dvp, G, (int)Lk, tmp, dv_dev, st);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
