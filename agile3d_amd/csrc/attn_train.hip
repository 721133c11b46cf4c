// libagile3d_hip -- attention and mask-head primitives WITH their backward, for the training path of the decoder
// (nn.MultiheadAttention as attention_block.py composes it, Agile3d.mask_module
// agile3d.py:342-384).  The inference path uses fused flash-style kernels (decoder.hip) that keep nothing; training
// needs the probabilities again, so here the score matrix [heads, Lq, Lk] is materialised (51 MB for 20 queries x 80 k
// points) and every step is one simple, deterministic kernel:
//   scores   S[h,i,j] = sum_d q[i,h,d] k[j,h,d] (mask -> -inf)        also used for dP = dO v^T and for src E^T
//   softmax  rows of S in place;   backward dS = P (dP - sum_j P dP) in place of dP
//   apply    O[i,h,:]  = sum_j P[h,i,j] V[j,h,:]                     (o = P v, dq = dS k, dsrc = dlogits E)
//   apply_T  O2[j,h,:] = sum_i P[h,i,j] X[i,h,:]                     (dv = P^T dO, dk = dS^T q, dE = dlogits^T src)
//   group max / its routing backward for the per-object max over an object's queries
// Written for parity first (thread per output element, fp32 sums in a fixed order); the fused kernels stay the fast path.
#include "common.h"

namespace a3d {

static unsigned blocks_of(size_t total, int t) { return (unsigned)((total + t - 1) / t); }

__global__ void k_tr_scores(const float* __restrict__ q, const float* __restrict__ k, int Lq, int Lk, int H, int dh,
                            float scale, const unsigned char* __restrict__ mask, float* __restrict__ S) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)H * Lq * Lk;
  if (e >= total) return;
  const int j = (int)(e % Lk);
  const int i = (int)((e / Lk) % Lq);
  const int h = (int)(e / ((size_t)Lk * Lq));
  const float* qr = q + (size_t)i * H * dh + h * dh;
  const float* kr = k + (size_t)j * H * dh + h * dh;
  float s = 0.f;
  for (int d = 0; d < dh; ++d) s += qr[d] * kr[d];
  s *= scale;
  if (mask && mask[(size_t)i * Lk + j]) s = -INFINITY;
  S[e] = s;
}

// the same scores with one thread per (head, row of the LONG side): its dh-vector stays in registers, the short side's
// vectors are wave-uniform (scalar loads); by_k = 1: thread = key j (writes coalesced over j), 0: thread = query i
template <int DHT>
__global__ void __launch_bounds__(256) k_tr_scores_long(const float* __restrict__ q, const float* __restrict__ k, int Lq,
                                                        int Lk, int H, float scale, const unsigned char* __restrict__ mask,
                                                        float* __restrict__ S, int by_k) {
  const int h = blockIdx.y, C = H * DHT;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int nlong = by_k ? Lk : Lq, nshort = by_k ? Lq : Lk;
  if (t >= nlong) return;
  const float* mine = (by_k ? k : q) + (size_t)t * C + h * DHT;
  const float* other = (by_k ? q : k) + h * DHT;
  float r[DHT];
#pragma unroll
  for (int d = 0; d < DHT; ++d) r[d] = mine[d];
  for (int o = 0; o < nshort; ++o) {
    const float* orow = other + (size_t)o * C;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DHT; ++d) s += r[d] * orow[d];
    s *= scale;
    const int i = by_k ? o : t, j = by_k ? t : o;
    if (mask && mask[(size_t)i * Lk + j]) s = -INFINITY;
    S[((size_t)h * Lq + i) * Lk + j] = s;
  }
}

// one workgroup per row (long rows) or one thread per row (short rows); mode 0: softmax in place,
// mode 1: a = P, b = dP -> b = P * (dP - sum P dP)
__global__ void __launch_bounds__(256) k_tr_rows_block(float* __restrict__ a, float* __restrict__ b, int L, int mode) {
  __shared__ float red[256];
  float* row = a + (size_t)blockIdx.x * L;
  float* rb = b ? b + (size_t)blockIdx.x * L : nullptr;
  auto reduce = [&](float v, bool is_max) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] = is_max ? fmaxf(red[threadIdx.x], red[threadIdx.x + s]) : red[threadIdx.x] + red[threadIdx.x + s];
      __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
  };
  if (mode == 0) {
    float m = -INFINITY;
    for (int j = threadIdx.x; j < L; j += 256) m = fmaxf(m, row[j]);
    m = reduce(m, true);
    float s = 0.f;
    for (int j = threadIdx.x; j < L; j += 256) {
      const float p = expf(row[j] - m);
      row[j] = p;
      s += p;
    }
    s = reduce(s, false);
    const float inv = 1.f / s;
    for (int j = threadIdx.x; j < L; j += 256) row[j] *= inv;
  } else {
    float s = 0.f;
    for (int j = threadIdx.x; j < L; j += 256) s += row[j] * rb[j];
    s = reduce(s, false);
    for (int j = threadIdx.x; j < L; j += 256) rb[j] = row[j] * (rb[j] - s);
  }
}
// short rows (the click-to-click attention: 8 heads x Q rows of Q <= 256 scores): one WAVE per row, lanes stride over the
// row, shuffle reductions in a fixed order -- one thread per row walked the row three times at one element per round trip
// (50 us for 1 600 rows of 200)
__global__ void __launch_bounds__(256) k_tr_rows_wave(float* __restrict__ a, float* __restrict__ b, size_t R, int L, int mode) {
  const int lane = threadIdx.x & 63;
  const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float* row = a + r * L;
  constexpr int kMax = 8;                      // L <= 512
  float v[kMax], w[kMax];
  const int per = (L + 63) >> 6;
  if (mode == 0) {
    float m = -INFINITY;
    for (int i = 0; i < per; ++i) {
      const int j = lane + 64 * i;
      v[i] = j < L ? row[j] : -INFINITY;
      m = fmaxf(m, v[i]);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
      v[i] = lane + 64 * i < L ? expf(v[i] - m) : 0.f;
      s += v[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float inv = 1.f / s;
    for (int i = 0; i < per; ++i)
      if (lane + 64 * i < L) row[lane + 64 * i] = v[i] * inv;
  } else {
    float* rb = b + r * L;
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
      const int j = lane + 64 * i;
      v[i] = j < L ? row[j] : 0.f;
      w[i] = j < L ? rb[j] : 0.f;
      s += v[i] * w[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    for (int i = 0; i < per; ++i)
      if (lane + 64 * i < L) rb[lane + 64 * i] = v[i] * (w[i] - s);
  }
}
// softmax over the MIDDLE dimension of [H][Lq][Lk] (one thread per (h, j), walking i with stride Lk: coalesced over j).
// Scene-to-click attention keeps its scores transposed -- [head][query][point], the point index fastest -- so that every
// kernel touching the 80 k-long dimension reads and writes consecutive addresses.  mode 0 / 1 as above.
__global__ void k_tr_cols(float* __restrict__ a, float* __restrict__ b, int H, int Lq, int Lk, int mode) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)H * Lk) return;
  const int j = (int)(e % Lk), h = (int)(e / Lk);
  float* col = a + (size_t)h * Lq * Lk + j;
  if (mode == 0) {
    float m = -INFINITY;
    for (int i = 0; i < Lq; ++i) m = fmaxf(m, col[(size_t)i * Lk]);
    float s = 0.f;
    for (int i = 0; i < Lq; ++i) {
      const float p = expf(col[(size_t)i * Lk] - m);
      col[(size_t)i * Lk] = p;
      s += p;
    }
    const float inv = 1.f / s;
    for (int i = 0; i < Lq; ++i) col[(size_t)i * Lk] *= inv;
  } else {
    float* cb = b + (size_t)h * Lq * Lk + j;
    float s = 0.f;
    for (int i = 0; i < Lq; ++i) s += col[(size_t)i * Lk] * cb[(size_t)i * Lk];
    for (int i = 0; i < Lq; ++i) cb[(size_t)i * Lk] = col[(size_t)i * Lk] * (cb[(size_t)i * Lk] - s);
  }
}

__global__ void k_tr_apply(const float* __restrict__ P, const float* __restrict__ V, int Lq, int Lk, int H, int dh,
                           float scale, float* __restrict__ O) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C = H * dh;
  if (e >= (size_t)Lq * C) return;
  const int c = (int)(e % C), i = (int)(e / C), h = c / dh;
  const float* p = P + ((size_t)h * Lq + i) * Lk;
  float s = 0.f;
  for (int j = 0; j < Lk; ++j) s += p[j] * V[(size_t)j * C + c];
  O[e] = s * scale;
}
__global__ void k_tr_apply_t(const float* __restrict__ P, const float* __restrict__ X, int Lq, int Lk, int H, int dh,
                             float scale, float* __restrict__ O) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C = H * dh;
  if (e >= (size_t)Lk * C) return;
  const int c = (int)(e % C), j = (int)(e / C), h = c / dh;
  const float* p = P + (size_t)h * Lq * Lk + j;
  float s = 0.f;
  for (int i = 0; i < Lq; ++i) s += p[(size_t)i * Lk] * X[(size_t)i * C + c];
  O[e] = s * scale;
}

// long reductions (20 queries x 80 k points: 2 560 outputs, each a sum over 80 k terms): the reduced dimension is cut
// into `splits` ranges, one grid.y slice each, partial sums to part[split][out] and a second kernel adds them in order
__global__ void k_tr_apply_split(const float* __restrict__ P, const float* __restrict__ V, int Lq, int Lk, int H, int dh,
                                 int transposed, int chunk, float* __restrict__ part) {
  const int C = H * dh;
  const int rows = transposed ? Lk : Lq, Lred = transposed ? Lq : Lk;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)rows * C) return;
  const int c = (int)(e % C), r = (int)(e / C), h = c / dh;
  const int b = blockIdx.y * chunk, end = min(Lred, b + chunk);
  float s = 0.f;
  if (!transposed) {
    const float* p = P + ((size_t)h * Lq + r) * Lk;
    for (int j = b; j < end; ++j) s += p[j] * V[(size_t)j * C + c];
  } else {
    const float* p = P + (size_t)h * Lq * Lk + r;
    for (int i = b; i < end; ++i) s += p[(size_t)i * Lk] * V[(size_t)i * C + c];
  }
  part[(size_t)blockIdx.y * rows * C + e] = s;
}
__global__ void k_tr_apply_reduce(const float* __restrict__ part, int splits, size_t total, float scale, float* __restrict__ O) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += part[(size_t)k * total + e];
  O[e] = s * scale;
}
// transposed apply with one thread per (long index j, head): P[h][i][j] is read coalesced over j, the short side's rows
// X[i][h*dh ..] are wave-uniform, the dh outputs of the head stay in registers
template <int DHT>
__global__ void __launch_bounds__(256) k_tr_apply_t_head(const float* __restrict__ P, const float* __restrict__ X, int Lq,
                                                         int Lk, int H, float scale, float* __restrict__ O) {
  const int h = blockIdx.y, C = H * DHT;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Lk) return;
  float acc[DHT];
#pragma unroll
  for (int d = 0; d < DHT; ++d) acc[d] = 0.f;
  const float* p = P + (size_t)h * Lq * Lk + j;
  for (int i = 0; i < Lq; ++i) {
    const float pv = p[(size_t)i * Lk];
    const float* xr = X + (size_t)i * C + h * DHT;
#pragma unroll
    for (int d = 0; d < DHT; ++d) acc[d] += pv * xr[d];
  }
  float* o = O + (size_t)j * C + h * DHT;
#pragma unroll
  for (int d = 0; d < DHT; ++d) o[d] = acc[d] * scale;
}
// the same partial sums on the matrix cores for 16-wide heads: one wave = (16 output rows, head, range of the long
// dimension); D[16 rows][16 channels] += P[h][rows][j..j+15] V[j..j+15][h*16 ..].  K is permuted inside a 16-step (lane
// (g, m) loads P[row m][j0 + 4g .. +3] as one 16-byte access and the matching value rows V[j0 + 4g + t] one MFMA at a
// time), so the 80 k value rows are read once per 16 output rows instead of once per output row.
typedef float f32x4t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) k_tr_apply_mfma(const float* __restrict__ P, const float* __restrict__ V, int Lq, int Lk,
                                                      int H, int chunk, float* __restrict__ part) {
  const int lane = threadIdx.x, g = lane >> 4, m = lane & 15;
  const int i0 = blockIdx.x * 16, h = blockIdx.y, C = H * 16;
  const int b = blockIdx.z * chunk, end = min(Lk, b + chunk);
  const int row = i0 + m;
  const float* prow = P + ((size_t)h * Lq + min(row, Lq - 1)) * Lk;
  const bool row_ok = row < Lq;
  f32x4t acc = (f32x4t){0.f, 0.f, 0.f, 0.f};
  for (int j0 = b; j0 < end; j0 += 16) {
    const int jj = j0 + 4 * g;
    f32x4t a = (f32x4t){0.f, 0.f, 0.f, 0.f};
    if (row_ok) {
      if (jj + 3 < end && ((size_t)(prow + jj) & 15) == 0) a = *(const f32x4t*)(prow + jj);
      else {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = jj + t < end ? prow[jj + t] : 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = jj + t;
      const float v = j < end ? V[(size_t)j * C + h * 16 + m] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], v, acc, 0, 0, 0);
    }
  }
  // lane (g, n = m) holds D[4g + r][n]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + 4 * g + r;
    if (i < Lq) part[((size_t)blockIdx.z * Lq + i) * C + h * 16 + m] = acc[r];
  }
}

static int apply_splits(int64_t rows, int64_t Lred, int C) {
  if (Lred < 4096 || rows * C > (1 << 20)) return 1;
  int64_t s = (Lred + 511) / 512;
  return (int)(s > 256 ? 256 : s);
}

// group g covers queries [qbeg[g], qend[g]); out[n][g] = max, arg[n][g] = first arg max (torch.max picks the first)
__global__ void k_tr_group_max(const float* __restrict__ lq, int N, int Q, const int* __restrict__ qbeg,
                               const int* __restrict__ qend, int G, float* __restrict__ out, int* __restrict__ arg) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)N * G) return;
  const int g = (int)(e % G), n = (int)(e / G);
  float best = -INFINITY;
  int bi = qbeg[g];
  for (int q = qbeg[g]; q < qend[g]; ++q) {
    const float v = lq[(size_t)n * Q + q];
    if (v > best) best = v, bi = q;
  }
  out[e] = best;
  arg[e] = bi;
}
__global__ void k_tr_group_max_bwd(const float* __restrict__ dout, const int* __restrict__ arg, int N, int Q, int G,
                                   float* __restrict__ dlq) {
  const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (size_t)N) return;
  for (int q = 0; q < Q; ++q) dlq[n * Q + q] = 0.f;
  for (int g = 0; g < G; ++g) dlq[n * Q + arg[n * G + g]] += dout[n * G + g];   // groups are disjoint
}

}  // namespace a3d

using namespace a3d;

extern "C" int a3d_attn_scores(const float* q_dev, const float* k_dev, int64_t Lq, int64_t Lk, int H, int dh, float scale,
                               const unsigned char* mask_dev, float* S_dev, void* stream) {
  if (!q_dev || !k_dev || !S_dev || Lq <= 0 || Lk <= 0 || H < 1 || dh < 1 || (int64_t)H * Lq * Lk > (int64_t)1 << 33) {
    set_error("a3d_attn_scores: bad arguments");
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t nlong = Lq > Lk ? Lq : Lk;
  const int by_k = Lk >= Lq;
  if ((dh == 16 || dh == 128) && nlong >= 1024) {
    const dim3 grid(blocks_of((size_t)nlong, 256), H);
    if (dh == 16) k_tr_scores_long<16><<<grid, 256, 0, st>>>(q_dev, k_dev, (int)Lq, (int)Lk, H, scale, mask_dev, S_dev, by_k);
    else k_tr_scores_long<128><<<grid, 256, 0, st>>>(q_dev, k_dev, (int)Lq, (int)Lk, H, scale, mask_dev, S_dev, by_k);
  } else {
    k_tr_scores<<<blocks_of((size_t)H * Lq * Lk, 256), 256, 0, st>>>(q_dev, k_dev, (int)Lq, (int)Lk, H, dh, scale, mask_dev,
                                                                    S_dev);
  }
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
extern "C" int a3d_softmax_rows(float* S_dev, int64_t rows, int64_t L, void* stream) {
  if (!S_dev || rows <= 0 || L <= 0 || L > (int64_t)1 << 30 || rows > (int64_t)1 << 31) {
    set_error("a3d_softmax_rows: bad arguments");
    return A3D_ERR_INVALID;
  }
  if (L >= 512) k_tr_rows_block<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(S_dev, nullptr, (int)L, 0);
  else k_tr_rows_wave<<<blocks_of((size_t)rows, 4), 256, 0, (hipStream_t)stream>>>(S_dev, nullptr, (size_t)rows, (int)L, 0);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
extern "C" int a3d_softmax_rows_backward(const float* P_dev, float* dP_dev, int64_t rows, int64_t L, void* stream) {
  if (!P_dev || !dP_dev || rows <= 0 || L <= 0 || L > (int64_t)1 << 30 || rows > (int64_t)1 << 31) {
    set_error("a3d_softmax_rows_backward: bad arguments");
    return A3D_ERR_INVALID;
  }
  if (L >= 512) k_tr_rows_block<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>((float*)P_dev, dP_dev, (int)L, 1);
  else k_tr_rows_wave<<<blocks_of((size_t)rows, 4), 256, 0, (hipStream_t)stream>>>((float*)P_dev, dP_dev, (size_t)rows, (int)L, 1);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
extern "C" int a3d_softmax_cols(float* S_dev, int H, int64_t Lq, int64_t Lk, void* stream) {
  if (!S_dev || H < 1 || Lq <= 0 || Lk <= 0) {
    set_error("a3d_softmax_cols: bad arguments");
    return A3D_ERR_INVALID;
  }
  k_tr_cols<<<blocks_of((size_t)H * Lk, 256), 256, 0, (hipStream_t)stream>>>(S_dev, nullptr, H, (int)Lq, (int)Lk, 0);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
extern "C" int a3d_softmax_cols_backward(const float* P_dev, float* dP_dev, int H, int64_t Lq, int64_t Lk, void* stream) {
  if (!P_dev || !dP_dev || H < 1 || Lq <= 0 || Lk <= 0) {
    set_error("a3d_softmax_cols_backward: bad arguments");
    return A3D_ERR_INVALID;
  }
  k_tr_cols<<<blocks_of((size_t)H * Lk, 256), 256, 0, (hipStream_t)stream>>>((float*)P_dev, dP_dev, H, (int)Lq, (int)Lk, 1);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
extern "C" size_t a3d_attn_apply_workspace_bytes(int64_t Lq, int64_t Lk, int H, int dh, int transposed) {
  const int64_t rows = transposed ? Lk : Lq, Lred = transposed ? Lq : Lk;
  const int sp = apply_splits(rows, Lred, H * dh);
  return sp > 1 ? (size_t)sp * rows * H * dh * sizeof(float) + 256 : 256;
}
extern "C" int a3d_attn_apply(const float* P_dev, const float* V_dev, int64_t Lq, int64_t Lk, int H, int dh, int transposed,
                              float scale, float* O_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!P_dev || !V_dev || !O_dev || Lq <= 0 || Lk <= 0 || H < 1 || dh < 1) {
    set_error("a3d_attn_apply: bad arguments");
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = transposed ? Lk : Lq, Lred = transposed ? Lq : Lk;
  const int sp = apply_splits(rows, Lred, H * dh);
  if (sp > 1) {
    if (!workspace_dev || workspace_bytes < a3d_attn_apply_workspace_bytes(Lq, Lk, H, dh, transposed)) {
      set_error("a3d_attn_apply: workspace too small (a3d_attn_apply_workspace_bytes)");
      return A3D_ERR_WORKSPACE;
    }
    const size_t total = (size_t)rows * H * dh;
    int chunk = (int)((Lred + sp - 1) / sp);
    chunk = (chunk + 15) / 16 * 16;        // 16-step aligned ranges (trailing ranges may be empty: they write zeros)
    if (!transposed && dh == 16)
      k_tr_apply_mfma<<<dim3((unsigned)((Lq + 15) / 16), H, sp), 64, 0, st>>>(P_dev, V_dev, (int)Lq, (int)Lk, H, chunk,
                                                                           (float*)workspace_dev);
    else
      k_tr_apply_split<<<dim3(blocks_of(total, 256), sp), 256, 0, st>>>(P_dev, V_dev, (int)Lq, (int)Lk, H, dh, transposed, chunk,
                                                                        (float*)workspace_dev);
    k_tr_apply_reduce<<<blocks_of(total, 256), 256, 0, st>>>((const float*)workspace_dev, sp, total, scale, O_dev);
  } else if (transposed && Lk >= 1024 && (dh == 16 || dh == 128)) {
    const dim3 grid(blocks_of((size_t)Lk, 256), H);
    if (dh == 16) k_tr_apply_t_head<16><<<grid, 256, 0, st>>>(P_dev, V_dev, (int)Lq, (int)Lk, H, scale, O_dev);
    else k_tr_apply_t_head<128><<<grid, 256, 0, st>>>(P_dev, V_dev, (int)Lq, (int)Lk, H, scale, O_dev);
  } else if (transposed) {
    k_tr_apply_t<<<blocks_of((size_t)Lk * H * dh, 256), 256, 0, st>>>(P_dev, V_dev, (int)Lq, (int)Lk, H, dh, scale, O_dev);
  } else {
    k_tr_apply<<<blocks_of((size_t)Lq * H * dh, 256), 256, 0, st>>>(P_dev, V_dev, (int)Lq, (int)Lk, H, dh, scale, O_dev);
  }
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
extern "C" int a3d_group_max(const float* lq_dev, int64_t N, int Q, const int32_t* qbeg_dev, const int32_t* qend_dev, int G,
                             float* out_dev, int32_t* arg_dev, void* stream) {
  if (!lq_dev || !qbeg_dev || !qend_dev || !out_dev || !arg_dev || N <= 0 || Q <= 0 || G <= 0) {
    set_error("a3d_group_max: bad arguments");
    return A3D_ERR_INVALID;
  }
  k_tr_group_max<<<blocks_of((size_t)N * G, 256), 256, 0, (hipStream_t)stream>>>(lq_dev, (int)N, Q, qbeg_dev, qend_dev, G,
                                                                               out_dev, arg_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
// ---- the attention mask of the NEXT decoder layer from this layer's mask logits (agile3d.py:362-383; not differentiated)
// label[n] = first arg-max over the 1 + K mask logits of point n; count[g] = points labelled g;
// mask[q][n] = label[n] != group(q) && count[group(q)] > 0   ("all points blocked -> nothing blocked", agile3d.py:369,375, is
// "no point carries the group's label").  Two kernels instead of the tape's eight torch launches per sample and layer.
__global__ void __launch_bounds__(256) k_tr_labels(const float* __restrict__ logits, int N, int G, unsigned char* __restrict__ labels,
                                                   int* __restrict__ counts) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) {
    const float* r = logits + (size_t)n * G;
    float best = r[0];
    int arg = 0;
    for (int g = 1; g < G; ++g) {
      const float v = r[g];
      if (v > best) best = v, arg = g;
    }
    labels[n] = (unsigned char)arg;
    atomicAdd(&h[arg], 1);
  }
  __syncthreads();
  if ((int)threadIdx.x < G && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_tr_next_mask(const unsigned char* __restrict__ labels, const int* __restrict__ counts, int N,
                                                      const int32_t* __restrict__ gq, int Q, unsigned char* __restrict__ mask) {
  __shared__ int grp[A3D_MAX_QUERIES];   // the group a query blocks other labels for, or -1 when its group owns no point
  for (int q = threadIdx.x; q < Q; q += 256) {
    const int g = gq[q];
    grp[q] = counts[g] > 0 ? g : -1;
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int lab = labels[n];
  for (int q = 0; q < Q; ++q) mask[(size_t)q * N + n] = (unsigned char)(grp[q] >= 0 && lab != grp[q]);
}
extern "C" size_t a3d_next_layer_mask_workspace_bytes(int64_t N, int G) {
  return N > 0 && G > 0 && G <= 256 ? (size_t)1024 + (size_t)((N + 255) / 256 * 256) : 0;
}
extern "C" int a3d_next_layer_mask(const float* logits_dev, int64_t N, int G, const int32_t* group_of_query_dev, int Q,
                                   unsigned char* mask_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!logits_dev || !group_of_query_dev || !mask_dev || !workspace_dev || N <= 0 || N >= (int64_t)1 << 31 || G <= 0 || G > 256 ||
      Q <= 0 || Q > A3D_MAX_QUERIES || workspace_bytes < a3d_next_layer_mask_workspace_bytes(N, G)) {
    set_error("a3d_next_layer_mask: bad arguments (N=%lld G=%d Q=%d)", (long long)N, G, Q);
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  int* counts = (int*)workspace_dev;
  unsigned char* labels = (unsigned char*)workspace_dev + 1024;
  A3D_HIP_CHECK(hipMemsetAsync(counts, 0, 1024, st));
  k_tr_labels<<<blocks_of((size_t)N, 256), 256, 0, st>>>(logits_dev, (int)N, G, labels, counts);
  k_tr_next_mask<<<blocks_of((size_t)N, 256), 256, 0, st>>>(labels, counts, (int)N, group_of_query_dev, Q, mask_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
extern "C" int a3d_group_max_backward(const float* dout_dev, const int32_t* arg_dev, int64_t N, int Q, int G, float* dlq_dev,
                                      void* stream) {
  if (!dout_dev || !arg_dev || !dlq_dev || N <= 0 || Q <= 0 || G <= 0) {
    set_error("a3d_group_max_backward: bad arguments");
    return A3D_ERR_INVALID;
  }
  k_tr_group_max_bwd<<<blocks_of((size_t)N, 256), 256, 0, (hipStream_t)stream>>>(dout_dev, arg_dev, (int)N, Q, G, dlq_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
