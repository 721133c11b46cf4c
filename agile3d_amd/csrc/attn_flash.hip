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
// dS is needed in two register layouts (rows x columns and columns x rows, because an MFMA contracts over the lane-group
// index of BOTH operands): the backward kernels compute it once and transpose it through a wave-private LDS tile.
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
// exp through v_exp_f32 (2^x): one multiply + one transcendental instead of expf's ~15 instructions; |x| <= ~100 here, the
// argument's rounding moves the result by |x| 2^-24 relative -- inside the 2e-5 the float64 comparison allows (tests)
__device__ __forceinline__ float fl_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
// mask bytes [row][col .. col + 3] as one (unaligned) load; `ok`: the four bytes lie inside the row
__device__ __forceinline__ unsigned ld_mask4(const unsigned char* __restrict__ mask, size_t row, int col, int ncols) {
  const unsigned char* p = mask + row * (size_t)ncols + col;
  unsigned u = 0x01010101u;                       // beyond the row: blocked
  if (col + 4 <= ncols) {
    __builtin_memcpy(&u, p, 4);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (col + t < ncols) u = (u & ~(0xffu << (8 * t))) | ((unsigned)p[t] << (8 * t));
  }
  return u;
}

// ------------------------------------------------------------------------------------------------ click-to-scene
// forward: partial flash state of one (workgroup, head) for every query: part[wg][h][q][18] = m, l, acc[16].  A workgroup
// walks the 64-key chunks wg, wg + G, ... (G = gridDim.x workgroups); the running state of EVERY query tile lives in
// registers across all of them (QT = tiles the build holds, nqt <= QT the call's), so the keys and values are read once --
// the first build walked all chunks once per query tile -- and the next 16-key group's fragments and mask words are in
// flight while the current group is multiplied.
template <int QT>
__global__ void __launch_bounds__(512) k_fl_c2s_fwd(const float* __restrict__ qs, const float* __restrict__ K,
                                                    const float* __restrict__ V, const unsigned char* __restrict__ mask,
                                                    int Lq, int Lk, float* __restrict__ part, int q0) {
  // q0: first query of this launch (a call with more queries than one build holds runs it once per block of 16 QT)
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int ngroups = (Lk + 15) / 16;
  const int nqt = min(QT, (Lq - q0 + 15) / 16);
  f32x4 qf[QT], acc[QT];
  float m[QT], l[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qf[qt] = ld4(qs, q0 + qt * 16 + j, Lq, h * FDH + 4 * g);       // this lane's query in the (keys x queries) layout
    m[qt] = kFlNeg;
    l[qt] = 0.f;
    acc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};                          // O^T[d = 4g+t][query j]
  }
  // the workgroup's groups: chunk ch = blockIdx.x, + G, ... of four 16-key groups each
  constexpr int NG = kFlChunk / 16;
  auto group_of = [&](int it) { return ((it / NG) * (int)gridDim.x + (int)blockIdx.x) * NG + (it % NG); };
  f32x4 kf_n = (f32x4){0.f, 0.f, 0.f, 0.f}, vf_n = kf_n;
  unsigned mk_n[QT];
  auto fetch = [&](int grp) {
    const int p0 = grp * 16;
    kf_n = ld4(K, p0 + j, Lk, h * FDH + 4 * g);                     // A: [key j][d 4g+t]
#pragma unroll
    for (int t = 0; t < 4; ++t) vf_n[t] = V[(size_t)min(p0 + 4 * g + t, Lk - 1) * FD + h * FDH + j];   // A: [d j][key 4g+t]
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      mk_n[qt] = 0u;
      if (mask && qt < nqt) mk_n[qt] = ld_mask4(mask, (size_t)min(q0 + qt * 16 + j, Lq - 1), p0 + 4 * g, Lk);
    }
  };
  int it = 0, grp = group_of(0);
  if (grp < ngroups) fetch(grp);
  while (grp < ngroups) {
    const int p0 = grp * 16;
    const f32x4 kf = kf_n, vf = vf_n;
    unsigned mk[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) mk[qt] = mk_n[qt];
    const int nxt = group_of(++it);
    if (nxt < ngroups) fetch(nxt);
    bool off[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) off[t] = p0 + 4 * g + t >= Lk;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (qt < nqt) {
        f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) s4 = FL_MFMA(kf[t], qf[qt][t], s4);   // s4[t] = S[key 4g+t][query j]
        float mx = kFlNeg;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool blocked = off[t] || ((mk[qt] >> (8 * t)) & 0xffu) != 0u;
          s4[t] = blocked ? kFlNeg : s4[t];
          mx = fmaxf(mx, s4[t]);
        }
        mx = fl_rows_max(mx);
        const float mnew = fmaxf(m[qt], mx);
        const float sc = fl_exp(m[qt] - mnew);
        m[qt] = mnew;
        f32x4 pw;
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          pw[t] = s4[t] <= kFlNeg ? 0.f : fl_exp(s4[t] - mnew);
          ps += pw[t];
        }
        l[qt] = l[qt] * sc + ps;
        acc[qt] *= sc;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[qt] = FL_MFMA(vf[t], pw[t], acc[qt]);
      }
    }
    grp = nxt;
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    if (qt < nqt) {
      const int qrow = q0 + qt * 16 + j;
      const float lt = fl_rows_sum(l[qt]);                           // (every lane takes part in the row sum)
      if (qrow < Lq) {
        float* pq = part + (((size_t)blockIdx.x * FH + h) * Lq + qrow) * 18;
        if (g == 0) {
          pq[0] = m[qt];
          pq[1] = lt;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) pq[2 + 4 * g + t] = acc[qt][t];
      }
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
// owns the (head, query tile): sequential, deterministic).
// Round 6: the chunk's key-side fragments (K, V rows, K^T columns of its four groups) are loaded ONCE per chunk and held
// across the query tiles (they were re-loaded per tile and group: fourteen dependent loads in front of every 28 MFMAs -- the
// kernel ran at a tenth of the matrix rate), the query side of the NEXT tile and the mask words of all four groups are
// requested before the current tile's products, mask bytes come four to a load, exp is one v_exp_f32.
struct FlQSide {   // one query tile as a lane sees it: query j's row fragments, and queries 4g+t: statistics + transposed columns
  f32x4 qf, dof;
  float mq[4], rlq[4], Dq[4], qT[4], doT[4];
};
// dS of a 16 x 16 tile from the (rows 4g+t, column j) layout the products over the ROWS need into the (columns 4g+t, row j)
// layout the product over the COLUMNS needs, through a wave-private LDS tile (a wave's LDS instructions execute in order:
// no barrier).  The first build computed the scores and dP a second time in the other layout instead: 8 of its 28 MFMAs
// per tile, a second set of exponentials, masks and statistics.
constexpr int kFlTLd = 20;          // row stride of the tile in floats: 16-byte aligned rows, conflict-free column writes
__device__ __forceinline__ f32x4 fl_transpose(float* tile, const f32x4& v, int g, int j) {
#pragma unroll
  for (int t = 0; t < 4; ++t) tile[(4 * g + t) * kFlTLd + j] = v[t];
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  return *(const f32x4*)(tile + j * kFlTLd + 4 * g);
}
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
  auto load_q = [&](int qt, FlQSide& s) {
    const int qj = qt * 16 + j;
    s.qf = ld4(qs, qj, Lq, h * FDH + 4 * g);                        // q[query j][d 4g+t]
    s.dof = ld4(dO, qj, Lq, h * FDH + 4 * g);                       // dO[query j][d 4g+t]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int qr = min(qt * 16 + 4 * g + t, Lq - 1);
      s.mq[t] = Ms[qr];
      s.rlq[t] = 1.f / Ls[qr];
      s.Dq[t] = Dh[qr];
      s.qT[t] = qs[(size_t)qr * FD + h * FDH + j];                  // A: [d j][query 4g+t]
      s.doT[t] = dO[(size_t)qr * FD + h * FDH + j];
    }
  };
  __shared__ __attribute__((aligned(16))) float tr_l[FH][NG][16 * kFlTLd];
  bool first = true;
  for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x, first = false) {
    const int pbeg = ch * kFlChunk;
    f32x4 kf[NG], vr[NG], kT[NG], adk[NG], adv[NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int p0 = pbeg + gi * 16;
      kf[gi] = ld4(K, p0 + j, Lk, h * FDH + 4 * g);                 // K[key j][d 4g+t]
      vr[gi] = ld4(V, p0 + j, Lk, h * FDH + 4 * g);                 // V[key j][d 4g+t]
#pragma unroll
      for (int t = 0; t < 4; ++t) kT[gi][t] = K[(size_t)min(p0 + 4 * g + t, Lk - 1) * FD + h * FDH + j];   // A: [d j][key 4g+t]
      adk[gi] = adv[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};              // dK^T / dV^T [d 4g+t][key j]
    }
    FlQSide qn;
    load_q(0, qn);
    for (int qt = 0; qt < nqt; ++qt) {
      const FlQSide q = qn;
      const int qj = qt * 16 + j;
      // mask bytes of the four groups: rows of queries 4g+t, key j
      unsigned mqk[NG];
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        mqk[gi] = 0u;
        if (mask) {
          const int p0 = pbeg + gi * 16;
          const int pr = min(p0 + j, Lk - 1);
#pragma unroll
          for (int t = 0; t < 4; ++t)
            mqk[gi] |= (unsigned)mask[(size_t)min(qt * 16 + 4 * g + t, Lq - 1) * Lk + pr] << (8 * t);
        }
      }
      if (qt + 1 < nqt) load_q(qt + 1, qn);
      f32x4 adq = (f32x4){0.f, 0.f, 0.f, 0.f};                      // dQ^T[d 4g+t][query j]
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        const int p0 = pbeg + gi * 16;
        // scores and dP as [query 4g+t][key j]
        f32x4 s_qk = (f32x4){0.f, 0.f, 0.f, 0.f}, dp_qk = s_qk;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s_qk = FL_MFMA(q.qf[t], kf[gi][t], s_qk);
          dp_qk = FL_MFMA(q.dof[t], vr[gi][t], dp_qk);
        }
        f32x4 p_qk, ds_qk;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool blocked = p0 + j >= Lk || qt * 16 + 4 * g + t >= Lq || ((mqk[gi] >> (8 * t)) & 0xffu) != 0u;
          const float p = blocked ? 0.f : fl_exp(s_qk[t] - q.mq[t]) * q.rlq[t];
          p_qk[t] = p;
          ds_qk[t] = p * (dp_qk[t] - q.Dq[t]);
        }
        const f32x4 ds_kq = fl_transpose(tr_l[h][gi], ds_qk, g, j);   // [key 4g+t][query j]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          adv[gi] = FL_MFMA(q.doT[t], p_qk[t], adv[gi]);            // dV^T[d][key] += dO^T[d][q] P[q][key]
          adk[gi] = FL_MFMA(q.qT[t], ds_qk[t], adk[gi]);            // dK^T[d][key] += q^T[d][q] dS[q][key]
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) adq = FL_MFMA(kT[gi][t], ds_kq[t], adq);   // dQ^T[d][query] += K^T[d][key] dS[key][query]
      }
      if (qj < Lq) {
        f32x4* slot = (f32x4*)(dqp + (((size_t)blockIdx.x * FH + h) * Lq + qj) * FDH + 4 * g);
        *slot = first ? adq : *slot + adq;
      }
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int pr = pbeg + gi * 16 + j;
      if (pr < Lk) {
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
// forward: per 16-point group and head, online softmax over the key tiles; stats[n][h][2] = m, l.  The workgroup's four
// 16-point groups walk the key tiles TOGETHER (their flash states and query fragments stay in registers), so a key
// tile's fragments are loaded once per chunk -- with the next tile's in flight -- instead of once per group.
__global__ void __launch_bounds__(512) k_fl_s2c_fwd(const float* __restrict__ qs, const float* __restrict__ K,
                                                    const float* __restrict__ V, int Lq, int Lk, float* __restrict__ O,
                                                    float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int nkt = (Lk + 15) / 16;
  constexpr int NG = kFlChunk / 16;
  const int pbeg = blockIdx.x * kFlChunk;
  f32x4 qf[NG], acc[NG];
  float m[NG], l[NG];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    qf[gi] = ld4(qs, pbeg + gi * 16 + j, Lq, h * FDH + 4 * g);      // B: [d 4g+t][point j]
    m[gi] = kFlNeg;
    l[gi] = 0.f;
    acc[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};                          // O^T[d 4g+t][point j]
  }
  f32x4 kf_n, vf_n;
  auto fetch = [&](int kt) {
    kf_n = ld4(K, kt * 16 + j, Lk, h * FDH + 4 * g);                // A: [key j][d 4g+t]
#pragma unroll
    for (int t = 0; t < 4; ++t) vf_n[t] = V[(size_t)min(kt * 16 + 4 * g + t, Lk - 1) * FD + h * FDH + j];   // A: [d j][key 4g+t]
  };
  fetch(0);
  for (int kt = 0; kt < nkt; ++kt) {
    const f32x4 kf = kf_n, vf = vf_n;
    if (kt + 1 < nkt) fetch(kt + 1);
    bool off[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) off[t] = kt * 16 + 4 * g + t >= Lk;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) s4 = FL_MFMA(kf[t], qf[gi][t], s4);   // [key 4g+t][point j]
      float mx = kFlNeg;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (off[t]) s4[t] = kFlNeg;
        mx = fmaxf(mx, s4[t]);
      }
      mx = fl_rows_max(mx);
      const float mnew = fmaxf(m[gi], mx);
      const float sc = fl_exp(m[gi] - mnew);
      m[gi] = mnew;
      f32x4 pw;
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        pw[t] = off[t] ? 0.f : fl_exp(s4[t] - mnew);
        ps += pw[t];
      }
      l[gi] = l[gi] * sc + ps;
      acc[gi] *= sc;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[gi] = FL_MFMA(vf[t], pw[t], acc[gi]);
    }
  }
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int pr = pbeg + gi * 16 + j;
    const float lt = fl_rows_sum(l[gi]);
    if (pr < Lq) {
      *(f32x4*)(O + (size_t)pr * FD + h * FDH + 4 * g) = acc[gi] * (1.f / lt);
      if (g == 0) {
        stats[((size_t)pr * FH + h) * 2] = m[gi];
        stats[((size_t)pr * FH + h) * 2 + 1] = lt;
      }
    }
  }
}

// backward: a workgroup walks the 64-point chunks wg, wg + G, ...: dQ rows of a chunk are final when it is done; the
// dK / dV contributions of its chunks add up in the workgroup's own slabs dkp / dvp[wg][h][key][16].
// Round 6: the point side of the chunk's four groups (rows and transposed columns of q and dO, statistics, the dO . O row
// sums) is loaded ONCE per chunk and held across the key tiles, the next key tile's fragments are requested before the
// current tile's products (both were re-loaded inside the innermost loop, thirteen loads per 28 MFMAs).
__global__ void __launch_bounds__(512) k_fl_s2c_bwd(const float* __restrict__ qs, const float* __restrict__ K,
                                                    const float* __restrict__ V, int Lq, int Lk, const float* __restrict__ O,
                                                    const float* __restrict__ stats, const float* __restrict__ dO,
                                                    float* __restrict__ dQ, float* __restrict__ dkp, float* __restrict__ dvp) {
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int nkt = (Lk + 15) / 16;
  const int nchunk = (Lq + kFlChunk - 1) / kFlChunk;
  constexpr int NG = kFlChunk / 16;
  f32x4 kf_n, vr_n, kT_n;
  auto fetch = [&](int kt) {
    kf_n = ld4(K, kt * 16 + j, Lk, h * FDH + 4 * g);                // K[key j][d 4g+t]
    vr_n = ld4(V, kt * 16 + j, Lk, h * FDH + 4 * g);                // V[key j][d 4g+t]
#pragma unroll
    for (int t = 0; t < 4; ++t) kT_n[t] = K[(size_t)min(kt * 16 + 4 * g + t, Lk - 1) * FD + h * FDH + j];   // A: [d j][key 4g+t]
  };
  __shared__ __attribute__((aligned(16))) float tr_l[FH][NG][16 * kFlTLd];
  bool first = true;
  for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x, first = false) {
    const int pbeg = ch * kFlChunk;
    fetch(0);
    // ---- the chunk's point side, once
    f32x4 qf[NG], dof[NG], qT[NG], doT[NG], mp[NG], rlp[NG], Dp[NG], adq[NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      float mj, rlj, Dj;
      const int p0 = pbeg + gi * 16;
      const int pj = min(p0 + j, Lq - 1);
      qf[gi] = ld4(qs, p0 + j, Lq, h * FDH + 4 * g);                // q[point j][d 4g+t]
      dof[gi] = ld4(dO, p0 + j, Lq, h * FDH + 4 * g);               // dO[point j][d 4g+t]
      const f32x4 of = ld4(O, p0 + j, Lq, h * FDH + 4 * g);
      mj = stats[((size_t)pj * FH + h) * 2];
      rlj = 1.f / stats[((size_t)pj * FH + h) * 2 + 1];
      float d = dof[gi][0] * of[0] + dof[gi][1] * of[1] + dof[gi][2] * of[2] + dof[gi][3] * of[3];
      Dj = fl_rows_sum(d);                                           // sum_d dO[point j][16h+d] O[point j][16h+d]
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int src = 4 * g + t;                                   // lane holding point 4g+t as ITS point j (any row)
        mp[gi][t] = __shfl(mj, src, 64);
        rlp[gi][t] = __shfl(rlj, src, 64);
        Dp[gi][t] = __shfl(Dj, src, 64);
        const int pr = min(p0 + 4 * g + t, Lq - 1);
        qT[gi][t] = qs[(size_t)pr * FD + h * FDH + j];               // A: [d j][point 4g+t]
        doT[gi][t] = dO[(size_t)pr * FD + h * FDH + j];
      }
      adq[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};                         // dQ^T[d 4g+t][point j]
    }
    for (int kt = 0; kt < nkt; ++kt) {
      const f32x4 kf = kf_n, vr = vr_n, kT = kT_n;
      if (kt + 1 < nkt) fetch(kt + 1);
      f32x4 adk = (f32x4){0.f, 0.f, 0.f, 0.f}, adv = adk;           // dK^T / dV^T [d 4g+t][key j]
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        const int p0 = pbeg + gi * 16;
        f32x4 s_pk = (f32x4){0.f, 0.f, 0.f, 0.f}, dp_pk = s_pk;       // [point 4g+t][key j]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s_pk = FL_MFMA(qf[gi][t], kf[t], s_pk);
          dp_pk = FL_MFMA(dof[gi][t], vr[t], dp_pk);
        }
        f32x4 p_pk, ds_pk;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool off = kt * 16 + j >= Lk || p0 + 4 * g + t >= Lq;
          const float p = off ? 0.f : fl_exp(s_pk[t] - mp[gi][t]) * rlp[gi][t];
          p_pk[t] = p;
          ds_pk[t] = p * (dp_pk[t] - Dp[gi][t]);
        }
        const f32x4 ds_kp = fl_transpose(tr_l[h][gi], ds_pk, g, j);   // [key 4g+t][point j]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          adk = FL_MFMA(qT[gi][t], ds_pk[t], adk);                   // dK^T[d][key] += q^T[d][point] dS[point][key]
          adv = FL_MFMA(doT[gi][t], p_pk[t], adv);                   // dV^T[d][key] += dO^T[d][point] P[point][key]
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) adq[gi] = FL_MFMA(kT[t], ds_kp[t], adq[gi]);   // dQ^T[d][point] += K^T[d][key] dS[key][point]
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
      if (pr < Lq) *(f32x4*)(dQ + (size_t)pr * FD + h * FDH + 4 * g) = adq[gi];
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
  // the build that holds the call's query tiles (or 14 of them per launch)
  for (int q0 = 0; q0 < (int)Lq; q0 += 16 * 14) {
    const int tiles = (int)((Lq - q0 + 15) / 16);
    if (tiles <= 2) k_fl_c2s_fwd<2><<<G, 512, 0, st>>>(q_scaled_dev, k_dev, v_dev, mask_dev, (int)Lq, (int)Lk, part, q0);
    else if (tiles <= 4) k_fl_c2s_fwd<4><<<G, 512, 0, st>>>(q_scaled_dev, k_dev, v_dev, mask_dev, (int)Lq, (int)Lk, part, q0);
    else if (tiles <= 6) k_fl_c2s_fwd<6><<<G, 512, 0, st>>>(q_scaled_dev, k_dev, v_dev, mask_dev, (int)Lq, (int)Lk, part, q0);
    else if (tiles <= 8) k_fl_c2s_fwd<8><<<G, 512, 0, st>>>(q_scaled_dev, k_dev, v_dev, mask_dev, (int)Lq, (int)Lk, part, q0);
    else if (tiles <= 10) k_fl_c2s_fwd<10><<<G, 512, 0, st>>>(q_scaled_dev, k_dev, v_dev, mask_dev, (int)Lq, (int)Lk, part, q0);
    else k_fl_c2s_fwd<14><<<G, 512, 0, st>>>(q_scaled_dev, k_dev, v_dev, mask_dev, (int)Lq, (int)Lk, part, q0);
  }
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
