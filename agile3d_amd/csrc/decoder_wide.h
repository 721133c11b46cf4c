// The fused decoder tier for 65 .. 224 queries (included by decoder.hip inside namespace a3d, after the sample tables).
//
// Reference: Agile3d.forward_mask's three attention blocks per decoder layer (models/agile3d.py:271-325) and mask_module
// (agile3d.py:342-384) for any number of queries; the multi-object protocol reaches num_obj x 20 clicks + 10 background
// queries (eval_multi_obj.py:70,116-118), training U[0,19] click rounds on up to 10 objects (engine.py:83-93).
//
// The <= 64-query kernels keep packed weight matrices (64 KB each) in LDS and give every wave its own 16-point group; past 64
// queries the queries' keys / values / embeddings no longer fit next to the weights.  Here the roles turn around:
//   * a WORKGROUP owns a 16-point group, a WAVE owns one 16-column slice of the layer's matrices -- head h of the K / V / Q
//     projections, output tile w of the output projection -- so a wave's slice of every weight matrix is 8 float4 per lane
//     and stays in REGISTERS for the life of a persistent workgroup, next to the wave's slice of the query side (head h of
//     the projected queries, of the queries' keys and transposed values; the mask embeddings of query tile w);
//   * LDS holds only what the eight waves share: the group's point rows (two slots, written a whole iteration ahead by
//     all 512 threads, one barrier per group) and the exchanges of the output half (LayerNorm statistics, the normalised
//     rows, the logits tile);
//   * nothing of size [heads, Q, N] or K / V / Q [N, 128] reaches HBM; the attention output O makes one round trip
//     (k_s2c_w -> k_out_w).
// Granularity: 19.5 point groups per workgroup for one 80 k-point sample instead of 2.4 per wave.
// Algorithmic work per 16-point group and layer (QT = padded queries / 16): c2s 512 + 64 QT MFMAs, s2c 256 + 64 QT, output
// half 256 + 32 QT (v_mfma_f32_16x16x4_f32, 2 048 FLOPs each).

constexpr int kWLD = 136;                       // row stride (floats) of the staged 16 x 128 tiles: conflict-free b128 fragment reads
constexpr int kWTile = 16 * kWLD;               // floats per staged tile
constexpr int kWideGridMax = 256;               // persistent workgroups (one per CU: 8 waves x up to 256 VGPRs)

// smallest wide-tier tile count that holds nq queries (0: not served -- more than 224 queries run the unfused kernels)
__host__ __device__ constexpr int wide_qt(int nq) {
  return nq <= 64 ? 0 : nq <= 80 ? 5 : nq <= 96 ? 6 : nq <= 112 ? 7 : nq <= 128 ? 8 : nq <= 160 ? 10 : nq <= 192 ? 12 : nq <= 224 ? 14 : 0;
}

// tile count of the k_c2s_attn build that holds `tiles` 16-query tiles of a sample whose buffers have round_qp(nq) rows
__host__ __device__ constexpr int wide_qt_of_tiles(int tiles) {
  return tiles <= 8 ? tiles : tiles <= 10 ? 10 : tiles <= 12 ? 12 : 14;
}

// the 512 threads of a workgroup move one group's rows of X and of the position encodings (16 x 128 floats each):
// thread -> (row lr, float4 column lc)
struct WideLoader {
  f32x4 rx, rp;
  __device__ __forceinline__ void issue(const float* __restrict__ X, const float* __restrict__ Pe, int grp, int n) {
    const int lr = threadIdx.x >> 5, lc = (threadIdx.x & 31) * 4;
    const size_t row = (size_t)min(grp * 16 + lr, n - 1);
    rx = gld4(X + row * D + lc);
    rp = gld4(Pe + row * D + lc);
  }
  __device__ __forceinline__ void commit(float* slot) const {
    const int lr = threadIdx.x >> 5, lc = (threadIdx.x & 31) * 4;
    *(f32x4*)(slot + lr * kWLD + lc) = rx;
    *(f32x4*)(slot + kWTile + lr * kWLD + lc) = rp;
  }
};

// ---- click-to-scene: K / V projections + flash attention of every query over the workgroup's points, wave = head ---------
// S = K_h q_h^T per 16-point group (A = the projected key slice, straight from the projection's accumulators), online
// softmax per query column, O^T += V_h^T P; one partial (m, l, acc[16]) per (head, query) and workgroup for k_c2s_combine.
template <int QT>
__global__ void __launch_bounds__(512) k_c2s_w(const DecSampleDev* __restrict__ samples, int ns, int layer,
                                               const float* __restrict__ Wk, const float* __restrict__ Wv,
                                               const float* __restrict__ bk, const float* __restrict__ bv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tiles = (float*)smem;   // [2 slots][X, P][16][kWLD]
  const DecSampleDev& sm = sample_of_wg(samples, ns);
  const int lb = blockIdx.x - sm.wg_begin, nwg = sm.wg_end - sm.wg_begin;
  const int n = sm.n, ngroups = (n + 15) / 16, nqt = sm.nqt;   // nqt <= QT: the sample's own query tiles (uniform)
  const float* __restrict__ X = layer_input(sm, layer);
  const float* __restrict__ Pe = sm.posenc;
  const unsigned char* labels = layer > 0 ? sm.labels : nullptr;                       // previous layer's mask labels
  const int* counts = layer > 0 ? sm.counts + (size_t)(layer - 1) * (A3D_MAX_QUERIES + 1) : nullptr;
  const int lane = threadIdx.x & 63, h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  // ---- the wave's constants: head h's slices of Wk / Wv (fragment order), its biases, the projected queries, the mask ids
  f32x4 wk[8], wv[8];
#pragma unroll
  for (int S = 0; S < 8; ++S) {
    wk[S] = ((const f32x4 A3D_GLOBAL*)Wk)[(S * 8 + h) * 64 + lane];
    wv[S] = ((const f32x4 A3D_GLOBAL*)Wv)[(S * 8 + h) * 64 + lane];
  }
  const f32x4 bk4 = gld4(bk + 16 * h + 4 * g);
  const float bvj = gld(bv + 16 * h + j);
  // The attention mask (agile3d.py:367-380: a query whose object currently owns points sees only those points) is ONE more
  // MFMA in front of each score tile instead of a compare + select per score: -2^20 (lab_p - obj_q)^2 is the rank-3 product
  // (lab^2, -2 lab, 1) . (1, obj, obj^2) (-2^20), exact in fp32 for ids <= 254 -- 0 for the object's own points, <= -2^20
  // (a weight of exactly 0) elsewhere, all zeros for an unmasked query -- and the fourth K slot carries -m, the query's
  // running reference maximum, so the accumulator comes out as score - m: a weight is one v_exp_f32 (a vector instruction
  // costs matrix-pipe time on gfx950).  mq = the lane's entry of that B operand: B[k = g][query j].
  f32x4 qf[QT];
  float mq[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qf[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mq[qt] = 0.f;
    if (qt < nqt) {
      qf[qt] = gld4(sm.qproj + (size_t)(qt * 16 + j) * D + 16 * h + 4 * g);
      const int o = gld(sm.qobj + qt * 16 + j);
      // a query is masked only if its object currently owns at least one point (agile3d.py:369,375)
      const bool masked = labels != nullptr && o >= 0 && gld(counts + o) > 0;
      const float fo = (float)o;
      mq[qt] = !masked || g == 3 ? 0.f : -1048576.f * (g == 0 ? 1.f : g == 1 ? fo : fo * fo);
    }
  }
  // m is a REFERENCE maximum, not the exact one: it moves only when a score exceeds it by more than 2^kLazy (the first group
  // sets it), so the common group costs no rescale of l and acc; what k_c2s_combine needs is l and acc relative to the m it reads
  constexpr float kLazy = 8.f;
  float m[QT], l[QT];
  f32x4 acc[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m[qt] = 0.f;
    l[qt] = 0.f;
    acc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  WideLoader ld;
  unsigned lab_nx = 0u;   // label byte of point j of the next group
  int grp = lb;
  if (grp < ngroups) {
    ld.issue(X, Pe, grp, n);
    if (labels) lab_nx = gld(labels + min(grp * 16 + j, n - 1));
    ld.commit(tiles);
  }
  __syncthreads();
  for (int it = 0; grp < ngroups; ++it, grp += nwg) {
    const int p0 = grp * 16;
    const float labj = (float)lab_nx;
    const int next = grp + nwg;
    const bool has_next = next < ngroups;
    if (has_next) {   // the next group's rows: the whole iteration to land, then into the other slot
      ld.issue(X, Pe, next, n);
      if (labels) lab_nx = gld(labels + min(next * 16 + j, n - 1));
    }
    const float* tx = tiles + (it & 1) * 2 * kWTile + j * kWLD + 4 * g;
    // ---- projections: kf = K[point j][16h+4g..+3] (transposed product), vv = V[points 4g..4g+3][16h+j]
    f32x4 kf = bk4, vv = (f32x4){bvj, bvj, bvj, bvj};
    {
      f32x4 xs = *(const f32x4*)tx, pe = *(const f32x4*)(tx + kWTile);
#pragma unroll
      for (int S = 0; S < 8; ++S) {
        f32x4 nxs = xs, npe = pe;
        if (S + 1 < 8) {
          nxs = *(const f32x4*)(tx + 16 * (S + 1));
          npe = *(const f32x4*)(tx + kWTile + 16 * (S + 1));
        }
        const f32x4 xp = xs + pe;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          kf = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[S][t], xp[t], kf, 0, 0, 0);
          vv = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[t], wv[S][t], vv, 0, 0, 0);
        }
        xs = nxs;
        pe = npe;
      }
    }
    // ---- attention of the 16 points against every query tile
    const float am = g == 0 ? labj * labj : g == 1 ? -2.f * labj : 1.f;   // A[point j][k = g] of the mask product
    const bool tail = p0 + 16 > n;                                        // rows beyond the sample: blocked for every query
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (qt < nqt) {   // (no break: the loop must stay fully unrolled -- its arrays are registers)
      f32x4 sc4;
      auto scores = [&]() {
        sc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, mq[qt], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) sc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t], qf[qt][t], sc4, 0, 0, 0);
        if (tail) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (p0 + 4 * g + t >= n) sc4[t] = kNegBig;
        }
      };
      scores();
      const float mx4 = fmaxf(fmaxf(sc4[0], sc4[1]), fmaxf(sc4[2], sc4[3]));
      if (it == 0 || __builtin_amdgcn_ballot_w64(mx4 > kLazy) != 0) {   // wave-uniform: move the reference, redo the tile
        const float mx = rows_max(mx4);
        // the first group centres every column (a column whose points are all blocked stops at -2^16: a later real
        // score is then still resolved to 2^-7 before the tile is recomputed against the new reference)
        const float delta = it == 0 ? fmaxf(mx, -65536.f) : fmaxf(mx, 0.f);
        const float scl = it == 0 ? 0.f : exp2_fast(-delta);
        m[qt] += delta;
        l[qt] *= scl;
        acc[qt] *= scl;
        if (g == 3) mq[qt] = -m[qt];
        scores();
      }
      f32x4 pw;
#pragma unroll
      for (int t = 0; t < 4; ++t) pw[t] = exp2_fast(sc4[t]);
      l[qt] += (pw[0] + pw[1]) + (pw[2] + pw[3]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[t], pw[t], acc[qt], 0, 0, 0);
    }
    }
    if (has_next) ld.commit(tiles + ((it + 1) & 1) * 2 * kWTile);
    __syncthreads();
  }
  // one partial per (head, query) and workgroup: [h][q][workgroup] -- what k_c2s_combine walks
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    if (qt < nqt) {
      const float lt = rows_sum(l[qt]);
      float* pq = sm.part + (((size_t)h * sm.qp + qt * 16 + j) * nwg + lb) * kPartStride;
      if (g == 0) {
        gst(pq, m[qt]);
        gst(pq + 1, lt);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) gst(pq + 2 + 4 * g + t, acc[qt][t]);
    }
  }
}

// ---- scene-to-click attention: Q projection + softmax over the queries + P V, wave = head -----------------------------
// Head h's slice of Wq, of the queries' keys (A fragments of S^T = ks_h Q_h^T) and of their transposed values (A fragments
// of O^T = vs_h^T P) are register constants; the group's attention output goes to sm.bufB (k_out_w reads it back).
// QC: the first layer's queries come from the scene's cache (as k_q_s2c<.., QC>): no projection, no staging, no barrier.
template <int QT, bool QC>
__global__ void __launch_bounds__(512) k_s2c_w(const DecSampleDev* __restrict__ samples, int ns, int layer,
                                               const float* __restrict__ Wq, const float* __restrict__ bq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tiles = (float*)smem;   // [2 slots][X, P][16][kWLD]
  const DecSampleDev& sm = sample_of_wg(samples, ns);
  const int lb = blockIdx.x - sm.wg_begin, nwg = sm.wg_end - sm.wg_begin;
  const int n = sm.n, ngroups = (n + 15) / 16, nq = sm.nq;
  const float* __restrict__ X = layer_input(sm, layer);
  const float* __restrict__ Pe = sm.posenc;
  float* __restrict__ O = sm.bufB;
  const int lane = threadIdx.x & 63, h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  f32x4 wq[8];
  f32x4 bq4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (!QC) {
#pragma unroll
    for (int S = 0; S < 8; ++S) wq[S] = ((const f32x4 A3D_GLOBAL*)Wq)[(S * 8 + h) * 64 + lane];
    bq4 = gld4(bq + 16 * h + 4 * g);
  }
  const int nqt = sm.nqt;   // the sample's own key tiles (<= QT, uniform); tiles beyond it are never touched
  f32x4 kf[QT], vf[QT];
#pragma unroll
  for (int kt = 0; kt < QT; ++kt) {
    kf[kt] = vf[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (kt < nqt) {
      kf[kt] = gld4(sm.ks + (size_t)(kt * 16 + j) * D + 16 * h + 4 * g);   // ks[key 16kt+j][16h+4g..+3] (pre-scaled, log2 domain)
#pragma unroll
      for (int t = 0; t < 4; ++t) vf[kt][t] = gld(sm.vs + (size_t)(kt * 16 + 4 * g + t) * D + 16 * h + j);   // V^T: keys 4g..4g+3 of channel 16h+j
    }
  }
  // score bias of the padded queries of the sample's LAST tile (rows of its buffers beyond nq are zero, k_query_init)
  f32x4 sbL;
#pragma unroll
  for (int t = 0; t < 4; ++t) sbL[t] = (nqt - 1) * 16 + 4 * g + t < nq ? 0.f : kNegBig;
  WideLoader ld;
  f32x4 qn = (f32x4){0.f, 0.f, 0.f, 0.f};
  int grp = lb;
  if (grp < ngroups) {
    if constexpr (QC) {
      qn = gld4(sm.q0 + (size_t)min(grp * 16 + j, n - 1) * D + 16 * h + 4 * g);
    } else {
      ld.issue(X, Pe, grp, n);
      ld.commit(tiles);
    }
  }
  if constexpr (!QC) __syncthreads();
  for (int it = 0; grp < ngroups; ++it, grp += nwg) {
    const int p0 = grp * 16;
    const int next = grp + nwg;
    const bool has_next = next < ngroups;
    f32x4 qf;
    if constexpr (QC) {
      qf = qn;
      if (has_next) qn = gld4(sm.q0 + (size_t)min(next * 16 + j, n - 1) * D + 16 * h + 4 * g);
    } else {
      if (has_next) ld.issue(X, Pe, next, n);
      const float* tx = tiles + (it & 1) * 2 * kWTile + j * kWLD + 4 * g;
      // Q[point j][16h+4g..+3]: even / odd K-steps on two accumulators (a chain of 32 dependent MFMAs otherwise)
      f32x4 q0 = bq4, q1 = (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 xa = *(const f32x4*)tx + *(const f32x4*)(tx + kWTile);
      f32x4 xb = *(const f32x4*)(tx + 16) + *(const f32x4*)(tx + kWTile + 16);
#pragma unroll
      for (int S = 0; S < 8; S += 2) {
        f32x4 na = xa, nb = xb;
        if (S + 2 < 8) {
          na = *(const f32x4*)(tx + 16 * (S + 2)) + *(const f32x4*)(tx + kWTile + 16 * (S + 2));
          nb = *(const f32x4*)(tx + 16 * (S + 3)) + *(const f32x4*)(tx + kWTile + 16 * (S + 3));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[S][t], xa[t], q0, 0, 0, 0);
          q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[S + 1][t], xb[t], q1, 0, 0, 0);
        }
        xa = na;
        xb = nb;
      }
      qf = q0 + q1;
    }
    // S^T[key 16kt+4g+t][point j]; key tiles two at a time (one uniform test per pair; an odd last tile's partner holds
    // zero keys and values: its weights are finite and meet zeros)
    f32x4 sc[QT + 1];
    float mx = kNegBig;
#pragma unroll
    for (int kt = 0; kt < QT; kt += 2) {
      if (kt < nqt) {   // (no break: the loops stay fully unrolled -- their arrays are registers)
      sc[kt] = sc[kt + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 kb = kt + 1 < QT ? kf[kt + 1] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][t], qf[t], sc[kt], 0, 0, 0);
        sc[kt + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[t], qf[t], sc[kt + 1], 0, 0, 0);
      }
      if (kt + 2 >= nqt) {   // the pair that holds the sample's last tile: padded queries out of the softmax
        if (kt + 1 == nqt) {
          sc[kt] += sbL;
          sc[kt + 1] = (f32x4){kNegBig, kNegBig, kNegBig, kNegBig};
        } else {
          sc[kt + 1] += sbL;
        }
      }
      mx = fmaxf(mx, fmaxf(fmaxf(sc[kt][0], sc[kt][1]), fmaxf(sc[kt][2], sc[kt][3])));
      mx = fmaxf(mx, fmaxf(fmaxf(sc[kt + 1][0], sc[kt + 1][1]), fmaxf(sc[kt + 1][2], sc[kt + 1][3])));
      }
    }
    mx = rows_max(mx);
    float sum = 0.f;
    // O^T[channel 16h+4g+t][point j]: one accumulator per tile of the pair
    f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
    for (int kt = 0; kt < QT; kt += 2) {
      if (kt < nqt) {
      const f32x4 vb = kt + 1 < QT ? vf[kt + 1] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sc[kt][t] = exp2_fast(sc[kt][t] - mx);
        sc[kt + 1][t] = exp2_fast(sc[kt + 1][t] - mx);
      }
      sum += ((sc[kt][0] + sc[kt][1]) + (sc[kt][2] + sc[kt][3])) + ((sc[kt + 1][0] + sc[kt + 1][1]) + (sc[kt + 1][2] + sc[kt + 1][3]));
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][t], sc[kt][t], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[t], sc[kt + 1][t], a1, 0, 0, 0);
      }
      }
    }
    sum = rows_sum(sum);
    const float inv = __builtin_amdgcn_rcpf(sum);
    if (p0 + j < n) gst4(O + (size_t)(p0 + j) * D + 16 * h + 4 * g, (a0 + a1) * inv);
    if constexpr (!QC) {
      if (has_next) ld.commit(tiles + ((it + 1) & 1) * 2 * kWTile);   // the rows have had the whole iteration to land
      __syncthreads();
    }
  }
}

// ---- output projection + residual + LayerNorm + mask head, wave = output column tile ------------------------------------
// y[point j][16w+4g..+3] = bo + src + O Wo^T from the wave's register-resident slice of Wo; LayerNorm over the eight waves'
// tiles through per-wave (mean, M2) pairs merged exactly (Chan); the normalised rows are exchanged through LDS and wave w
// multiplies them with the mask embeddings of query tiles w (and w + 8), also register constants.  A lane then holds the
// logits of four points against ONE query, whose object is a lane constant: the per-object maximum (agile3d.py:348-365) is
// one LDS float-max atomic per value into [point][object] -- no logits tile, no scan.  Two 16-point groups per iteration
// (four independent MFMA chains, three barriers per 32 points); label argmax, histogram and the logits rows as k_out_ln_mask.
template <int QT>
__global__ void __launch_bounds__(512) k_out_w(const DecSampleDev* __restrict__ samples, int ns, int layer,
                                               const float* __restrict__ Wo, const float* __restrict__ bo,
                                               const float* __restrict__ gamma, const float* __restrict__ beta, int Kmax) {
  constexpr int NTW = QT > 8 ? 2 : 1;          // query tiles per wave
  constexpr int MG = 2;                        // point groups per iteration
  typedef __attribute__((address_space(3))) float lds_float;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* o_l = (float*)smem;                   // [2 slots][MG][16][kWLD] attention output rows
  float* y_l = o_l + 2 * MG * kWTile;          // [MG][16][kWLD] normalised rows
  float* st_l = y_l + MG * kWTile;             // [MG][16 points][8 waves][2] LayerNorm partials
  float* Ol = st_l + MG * 16 * 16;             // [2 slots][MG * 16][Kmax+1] per-object maxima
  int* hist = (int*)(Ol + 2 * MG * 16 * (Kmax + 1));   // [Kmax+1]
  const DecSampleDev& sm = sample_of_wg(samples, ns);
  const int lb = blockIdx.x - sm.wg_begin, nwg = sm.wg_end - sm.wg_begin;
  const int n = sm.n, npairs = (n + 16 * MG - 1) / (16 * MG), K = sm.K, K1 = K + 1;
  const float* __restrict__ O = sm.bufB;
  const float* __restrict__ Xres = layer_input(sm, layer);
  float* __restrict__ Y = (layer & 1) ? sm.bufD : sm.bufC;
  float* logits = sm.logits + (size_t)layer * n * K1;
  unsigned char* labels = sm.labels;
  int* counts = sm.counts + (size_t)layer * (A3D_MAX_QUERIES + 1);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, j = lane & 15;
  for (int e = tid; e <= K; e += 512) hist[e] = 0;
  f32x4 wo[8];
#pragma unroll
  for (int S = 0; S < 8; ++S) wo[S] = ((const f32x4 A3D_GLOBAL*)Wo)[(S * 8 + w) * 64 + lane];
  const f32x4 bo4 = gld4(bo + 16 * w + 4 * g), ga4 = gld4(gamma + 16 * w + 4 * g), be4 = gld4(beta + 16 * w + 4 * g);
  // mask embeddings E[query 16 qt + j][16 S + 4 g ..+3] of the wave's query tiles (B fragments of the logits product) and
  // the object of query 16 qt + j (0 = background, -1 = padding)
  f32x4 ef[NTW][8];
  int oq[NTW];
  const int nqt = sm.nqt;   // the sample's own query tiles (<= QT, uniform)
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int qt = w + 8 * i;
    oq[i] = qt < nqt ? gld(sm.qobj + qt * 16 + j) : -1;
#pragma unroll
    for (int S = 0; S < 8; ++S)
      ef[i][S] = qt < nqt ? gld4(sm.E + (size_t)(qt * 16 + j) * D + 16 * S + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int lr = tid >> 5, lc = (tid & 31) * 4;
  int pr = lb;                                 // pair of point groups
  f32x4 ro[MG], rres[MG];
#pragma unroll
  for (int mg = 0; mg < MG; ++mg) ro[mg] = rres[mg] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (pr < npairs) {
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
      ro[mg] = gld4(O + (size_t)min((pr * MG + mg) * 16 + lr, n - 1) * D + lc);
      rres[mg] = gld4(Xres + (size_t)min((pr * MG + mg) * 16 + j, n - 1) * D + 16 * w + 4 * g);
      *(f32x4*)(o_l + mg * kWTile + lr * kWLD + lc) = ro[mg];
    }
  }
  __syncthreads();
  for (int it = 0; pr < npairs; ++it, pr += nwg) {
    const int p0 = pr * MG * 16;
    const int next = pr + nwg;
    const bool has_next = next < npairs;
    f32x4 res[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) res[mg] = rres[mg];
    if (has_next) {
#pragma unroll
      for (int mg = 0; mg < MG; ++mg) {
        ro[mg] = gld4(O + (size_t)min((next * MG + mg) * 16 + lr, n - 1) * D + lc);
        rres[mg] = gld4(Xres + (size_t)min((next * MG + mg) * 16 + j, n - 1) * D + 16 * w + 4 * g);
      }
    }
    // this iteration's per-object maxima start at -inf (the slot's previous readers are two iterations back)
    float* Oc = Ol + (it & 1) * MG * 16 * K1;
    for (int e = tid; e < MG * 16 * K1; e += 512) Oc[e] = -3.4e38f;
    // ---- y tiles: bias + residual + O Wo^T (even / odd K-steps and the two groups: four chains)
    const float* to = o_l + (it & 1) * MG * kWTile + j * kWLD + 4 * g;
    f32x4 y[MG];
    {
      f32x4 ya[MG], yb[MG];
#pragma unroll
      for (int mg = 0; mg < MG; ++mg) {
        ya[mg] = bo4 + res[mg];
        yb[mg] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int S = 0; S < 8; S += 2) {
        f32x4 oa[MG], ob[MG];
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
          oa[mg] = *(const f32x4*)(to + mg * kWTile + 16 * S);
          ob[mg] = *(const f32x4*)(to + mg * kWTile + 16 * (S + 1));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int mg = 0; mg < MG; ++mg) {
            ya[mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wo[S][t], oa[mg][t], ya[mg], 0, 0, 0);
            yb[mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wo[S + 1][t], ob[mg][t], yb[mg], 0, 0, 0);
          }
      }
#pragma unroll
      for (int mg = 0; mg < MG; ++mg) y[mg] = ya[mg] + yb[mg];
    }
    // ---- LayerNorm: the wave's 16 channels of point j -> (mean, M2), merged over the eight waves
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
      const float mw = rows_sum((y[mg][0] + y[mg][1]) + (y[mg][2] + y[mg][3])) * (1.f / 16.f);
      float m2 = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) m2 += (y[mg][t] - mw) * (y[mg][t] - mw);
      m2 = rows_sum(m2);
      if (g == 0) *(float2*)(st_l + mg * 256 + (j * 8 + w) * 2) = make_float2(mw, m2);
    }
    __syncthreads();                                                       // (1) statistics
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
      f32x4 s4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] = *(const f32x4*)(st_l + mg * 256 + j * 16 + 4 * u);   // waves 2u, 2u+1: (mean, M2) x 2
      float mean = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) mean += s4[u][0] + s4[u][2];
      mean *= 0.125f;
      float var = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d0 = s4[u][0] - mean, d1 = s4[u][2] - mean;
        var += (s4[u][1] + s4[u][3]) + 16.f * (d0 * d0 + d1 * d1);
      }
      const float rstd = rsqrtf(var * (1.f / D) + kLnEps);
#pragma unroll
      for (int t = 0; t < 4; ++t) y[mg][t] = (y[mg][t] - mean) * rstd * ga4[t] + be4[t];
      const int row = p0 + mg * 16 + j;
      if (row < n) gst4(Y + (size_t)row * D + 16 * w + 4 * g, y[mg]);
      *(f32x4*)(y_l + mg * kWTile + j * kWLD + 16 * w + 4 * g) = y[mg];
    }
    __syncthreads();                                                       // (2) normalised rows
    // ---- logits of the 2 x 16 points against the wave's query tiles (C layout: row = point 4g+t, column = query j), then
    // the per-object maxima
    // (a wave whose second tile lies beyond the sample's queries runs the one-tile form: samples of one launch differ)
    auto logits_tiles = [&](auto nt_c) {
      constexpr int NT = decltype(nt_c)::value;
      const float* ty = y_l + j * kWLD + 4 * g;
      f32x4 la[MG][NT], lc2[MG][NT];
#pragma unroll
      for (int mg = 0; mg < MG; ++mg)
#pragma unroll
        for (int i = 0; i < NT; ++i) la[mg][i] = lc2[mg][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int S = 0; S < 8; S += 2) {
        f32x4 ya[MG], yb[MG];
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
          ya[mg] = *(const f32x4*)(ty + mg * kWTile + 16 * S);
          yb[mg] = *(const f32x4*)(ty + mg * kWTile + 16 * (S + 1));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int mg = 0; mg < MG; ++mg)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
              la[mg][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[mg][t], ef[i][S][t], la[mg][i], 0, 0, 0);
              lc2[mg][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[mg][t], ef[i][S + 1][t], lc2[mg][i], 0, 0, 0);
            }
      }
#pragma unroll
      for (int mg = 0; mg < MG; ++mg)
#pragma unroll
        for (int i = 0; i < NT; ++i)
          if (oq[i] >= 0) {
            const f32x4 lg = la[mg][i] + lc2[mg][i];
#pragma unroll
            for (int t = 0; t < 4; ++t)
              __builtin_amdgcn_ds_fmaxf((lds_float*)(Oc + (mg * 16 + 4 * g + t) * K1 + oq[i]), lg[t], 0, 0, false);
          }
    };
    if (NTW == 2 && w + 8 < nqt) logits_tiles(std::integral_constant<int, NTW>{});
    else if (w < nqt) logits_tiles(std::integral_constant<int, 1>{});
    if (has_next) {   // the next pair's attention rows: issued at the top of the iteration
#pragma unroll
      for (int mg = 0; mg < MG; ++mg) *(f32x4*)(o_l + (((it + 1) & 1) * MG + mg) * kWTile + lr * kWLD + lc) = ro[mg];
    }
    __syncthreads();                                                       // (3) per-object maxima
    if (tid < MG * 16 && p0 + tid < n) {
      float best = Oc[tid * K1];
      int bi = 0;
      for (int o = 1; o <= K; ++o) {
        const float v = Oc[tid * K1 + o];
        if (v > best) {
          best = v;
          bi = o;
        }
      }
      gst(labels + p0 + tid, (unsigned char)bi);
      atomicAdd(&hist[bi], 1);
    }
    const int rows = min(MG * 16, n - p0);
    for (int e = tid; e < rows * K1; e += 512) gst(logits + (size_t)p0 * K1 + e, Oc[e]);
  }
  __syncthreads();
  for (int e = tid; e <= K; e += 512)
    if (hist[e]) atomicAdd(&counts[e], hist[e]);
}

// ---- the whole scene-to-click half in ONE kernel for up to 5 query tiles (33 .. 80 queries) ------------------------------
// k_s2c_w + k_out_w without the round trip of the attention output through HBM, the second launch and its start-up: a wave is
// head h of the attention AND output tile h of the projection behind it, so its slices of Wq and Wo, of the queries' keys
// and transposed values are register constants (80 + 8 QT registers: the reason for the limit) and the mask embeddings of
// query tile h sit in LDS; the eight heads' attention outputs of a 16-point group meet in an LDS tile, everything behind it
// is k_out_w's (two groups per iteration, LayerNorm by merged per-wave statistics, per-object maxima by LDS float-max
// atomics).  The staged rows are src + posenc (summed once, at the commit); the residual rows come from memory
// (L2: the staging loads fetched them); QC (the first layer's queries from the scene cache) stages nothing.
template <int QT, bool QC>
__global__ void __launch_bounds__(512) k_s2o_w(const DecSampleDev* __restrict__ samples, int ns, int layer,
                                               const float* __restrict__ Wq, const float* __restrict__ bq,
                                               const float* __restrict__ Wo, const float* __restrict__ bo,
                                               const float* __restrict__ gamma, const float* __restrict__ beta, int Kmax) {
  constexpr int MG = 2;
  typedef __attribute__((address_space(3))) float lds_float;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* o_l = (float*)smem;                                  // [MG][16][kWLD] attention output rows of the pair
  float* y_l = o_l + MG * kWTile;                             // [MG][16][kWLD] normalised rows
  float* st_l = y_l + MG * kWTile;                            // [MG][16 points][8 waves][2] LayerNorm partials
  float* Ol = st_l + MG * 16 * 16;                            // [2 slots][MG * 16][Kmax+1] per-object maxima
  int* hist = (int*)(Ol + 2 * MG * 16 * (Kmax + 1));          // [Kmax+1]
  float* e_l = (float*)(hist + ((Kmax + 1 + 3) & ~3));        // [QT][16][kWLD] mask embeddings (tile h is wave h's alone)
  float* tiles = e_l + QT * kWTile;                           // [2 slots][MG][16][kWLD] src + posenc rows (not with QC)
  const DecSampleDev& sm = sample_of_wg(samples, ns);
  const int lb = blockIdx.x - sm.wg_begin, nwg = sm.wg_end - sm.wg_begin;
  const int n = sm.n, npairs = (n + 16 * MG - 1) / (16 * MG), nq = sm.nq, K = sm.K, K1 = K + 1;
  const float* __restrict__ X = layer_input(sm, layer);
  const float* __restrict__ Pe = sm.posenc;
  float* __restrict__ Y = (layer & 1) ? sm.bufD : sm.bufC;
  float* logits = sm.logits + (size_t)layer * n * K1;
  unsigned char* labels = sm.labels;
  int* counts = sm.counts + (size_t)layer * (A3D_MAX_QUERIES + 1);
  const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, j = lane & 15;
  for (int e = tid; e <= K; e += 512) hist[e] = 0;
  // ---- the wave's constants
  f32x4 wq[8], wo[8];
  f32x4 bq4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int S = 0; S < 8; ++S) {
    if constexpr (!QC) wq[S] = ((const f32x4 A3D_GLOBAL*)Wq)[(S * 8 + h) * 64 + lane];
    wo[S] = ((const f32x4 A3D_GLOBAL*)Wo)[(S * 8 + h) * 64 + lane];
  }
  if constexpr (!QC) bq4 = gld4(bq + 16 * h + 4 * g);
  const f32x4 bo4 = gld4(bo + 16 * h + 4 * g), ga4 = gld4(gamma + 16 * h + 4 * g), be4 = gld4(beta + 16 * h + 4 * g);
  const int nqt = sm.nqt;
  f32x4 kf[QT], vf[QT];
#pragma unroll
  for (int kt = 0; kt < QT; ++kt) {
    kf[kt] = vf[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (kt < nqt) {
      kf[kt] = gld4(sm.ks + (size_t)(kt * 16 + j) * D + 16 * h + 4 * g);
#pragma unroll
      for (int t = 0; t < 4; ++t) vf[kt][t] = gld(sm.vs + (size_t)(kt * 16 + 4 * g + t) * D + 16 * h + j);
    }
  }
  f32x4 sbL;
#pragma unroll
  for (int t = 0; t < 4; ++t) sbL[t] = (nqt - 1) * 16 + 4 * g + t < nq ? 0.f : kNegBig;
  const int oq = h < nqt ? gld(sm.qobj + h * 16 + j) : -1;   // the object of query 16 h + j (mask embeddings of query tile h)
  if (h < nqt) {
#pragma unroll
    for (int S = 0; S < 8; ++S) *(f32x4*)(e_l + (h * 16 + j) * kWLD + 16 * S + 4 * g) = gld4(sm.E + (size_t)(h * 16 + j) * D + 16 * S + 4 * g);
  }
  // ---- staging: thread -> (row lr, float4 column lc) of each of the pair's tiles
  const int lr = tid >> 5, lc = (tid & 31) * 4;
  f32x4 rx[MG], rp[MG];
  auto issue = [&](int pr) {
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
      const size_t row = (size_t)min((pr * MG + mg) * 16 + lr, n - 1);
      rx[mg] = gld4(X + row * D + lc);
      rp[mg] = gld4(Pe + row * D + lc);
    }
  };
  auto commit = [&](int slot) {
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) *(f32x4*)(tiles + (slot * MG + mg) * kWTile + lr * kWLD + lc) = rx[mg] + rp[mg];
  };
  int pr = lb;
  if constexpr (!QC) {
    if (pr < npairs) {
      issue(pr);
      commit(0);
    }
  }
  __syncthreads();
  for (int it = 0; pr < npairs; ++it, pr += nwg) {
    const int p0 = pr * MG * 16;
    const int next = pr + nwg;
    const bool has_next = next < npairs;
    float* Oc = Ol + (it & 1) * MG * 16 * K1;
    for (int e = tid; e < MG * 16 * K1; e += 512) Oc[e] = -3.4e38f;
    // ---- scene-to-click attention of head h, one group after the other; O_h into the pair's LDS tiles
    f32x4 res[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
      f32x4 qf;
      if constexpr (QC) {
        const size_t row = (size_t)min(p0 + mg * 16 + j, n - 1);
        qf = gld4(sm.q0 + row * D + 16 * h + 4 * g);
        res[mg] = gld4(X + row * D + 16 * h + 4 * g);
      } else {
        const float* tx = tiles + ((it & 1) * MG + mg) * kWTile + j * kWLD + 4 * g;
        f32x4 q0 = bq4, q1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int S = 0; S < 8; S += 2) {
          const f32x4 xa = *(const f32x4*)(tx + 16 * S), xb = *(const f32x4*)(tx + 16 * (S + 1));
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[S][t], xa[t], q0, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[S + 1][t], xb[t], q1, 0, 0, 0);
          }
        }
        qf = q0 + q1;
      }
      f32x4 sc[QT + 1];
      float mx = kNegBig;
#pragma unroll
      for (int kt = 0; kt < QT; kt += 2) {
        if (kt < nqt) {
          sc[kt] = sc[kt + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
          const f32x4 kb = kt + 1 < QT ? kf[kt + 1] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][t], qf[t], sc[kt], 0, 0, 0);
            sc[kt + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[t], qf[t], sc[kt + 1], 0, 0, 0);
          }
          if (kt + 2 >= nqt) {
            if (kt + 1 == nqt) {
              sc[kt] += sbL;
              sc[kt + 1] = (f32x4){kNegBig, kNegBig, kNegBig, kNegBig};
            } else {
              sc[kt + 1] += sbL;
            }
          }
          mx = fmaxf(mx, fmaxf(fmaxf(sc[kt][0], sc[kt][1]), fmaxf(sc[kt][2], sc[kt][3])));
          mx = fmaxf(mx, fmaxf(fmaxf(sc[kt + 1][0], sc[kt + 1][1]), fmaxf(sc[kt + 1][2], sc[kt + 1][3])));
        }
      }
      mx = rows_max(mx);
      float sum = 0.f;
      f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int kt = 0; kt < QT; kt += 2) {
        if (kt < nqt) {
          const f32x4 vb = kt + 1 < QT ? vf[kt + 1] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            sc[kt][t] = exp2_fast(sc[kt][t] - mx);
            sc[kt + 1][t] = exp2_fast(sc[kt + 1][t] - mx);
          }
          sum += ((sc[kt][0] + sc[kt][1]) + (sc[kt][2] + sc[kt][3])) + ((sc[kt + 1][0] + sc[kt + 1][1]) + (sc[kt + 1][2] + sc[kt + 1][3]));
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][t], sc[kt][t], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[t], sc[kt + 1][t], a1, 0, 0, 0);
          }
        }
      }
      sum = rows_sum(sum);
      *(f32x4*)(o_l + mg * kWTile + j * kWLD + 16 * h + 4 * g) = (a0 + a1) * __builtin_amdgcn_rcpf(sum);   // O[point j][16h+4g..+3]
    }
    __syncthreads();                                                       // (0) the eight heads' outputs of both groups
    if constexpr (!QC) {
      if (has_next) issue(next);      // in flight behind the output half (ahead of the attention they would cost it 16 registers)
    }
    // ---- output projection (tile h) + residual, LayerNorm, mask head: k_out_w's
    f32x4 y[MG];
    {
      f32x4 ya[MG], yb[MG];
#pragma unroll
      for (int mg = 0; mg < MG; ++mg) {
        if constexpr (!QC) res[mg] = gld4(X + (size_t)min(p0 + mg * 16 + j, n - 1) * D + 16 * h + 4 * g);   // the residual (an L2 hit)
        ya[mg] = bo4;
        yb[mg] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const float* to = o_l + j * kWLD + 4 * g;
#pragma unroll
      for (int S = 0; S < 8; S += 2) {
        f32x4 oa[MG], ob[MG];
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
          oa[mg] = *(const f32x4*)(to + mg * kWTile + 16 * S);
          ob[mg] = *(const f32x4*)(to + mg * kWTile + 16 * (S + 1));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int mg = 0; mg < MG; ++mg) {
            ya[mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wo[S][t], oa[mg][t], ya[mg], 0, 0, 0);
            yb[mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wo[S + 1][t], ob[mg][t], yb[mg], 0, 0, 0);
          }
      }
#pragma unroll
      for (int mg = 0; mg < MG; ++mg) y[mg] = (ya[mg] + yb[mg]) + res[mg];
    }
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
      const float mw = rows_sum((y[mg][0] + y[mg][1]) + (y[mg][2] + y[mg][3])) * (1.f / 16.f);
      float m2 = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) m2 += (y[mg][t] - mw) * (y[mg][t] - mw);
      m2 = rows_sum(m2);
      if (g == 0) *(float2*)(st_l + mg * 256 + (j * 8 + h) * 2) = make_float2(mw, m2);
    }
    __syncthreads();                                                       // (1) statistics
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
      f32x4 s4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] = *(const f32x4*)(st_l + mg * 256 + j * 16 + 4 * u);
      float mean = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) mean += s4[u][0] + s4[u][2];
      mean *= 0.125f;
      float var = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d0 = s4[u][0] - mean, d1 = s4[u][2] - mean;
        var += (s4[u][1] + s4[u][3]) + 16.f * (d0 * d0 + d1 * d1);
      }
      const float rstd = rsqrtf(var * (1.f / D) + kLnEps);
#pragma unroll
      for (int t = 0; t < 4; ++t) y[mg][t] = (y[mg][t] - mean) * rstd * ga4[t] + be4[t];
      const int row = p0 + mg * 16 + j;
      if (row < n) gst4(Y + (size_t)row * D + 16 * h + 4 * g, y[mg]);
      *(f32x4*)(y_l + mg * kWTile + j * kWLD + 16 * h + 4 * g) = y[mg];
    }
    __syncthreads();                                                       // (2) normalised rows
    if (h < nqt) {
      const float* ty = y_l + j * kWLD + 4 * g;
      const float* te = e_l + (h * 16 + j) * kWLD + 4 * g;
      f32x4 la[MG], lc2[MG];
#pragma unroll
      for (int mg = 0; mg < MG; ++mg) la[mg] = lc2[mg] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int S = 0; S < 8; S += 2) {
        f32x4 ya[MG], yb[MG];
        const f32x4 ea = *(const f32x4*)(te + 16 * S), eb = *(const f32x4*)(te + 16 * (S + 1));
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
          ya[mg] = *(const f32x4*)(ty + mg * kWTile + 16 * S);
          yb[mg] = *(const f32x4*)(ty + mg * kWTile + 16 * (S + 1));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int mg = 0; mg < MG; ++mg) {
            la[mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[mg][t], ea[t], la[mg], 0, 0, 0);
            lc2[mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[mg][t], eb[t], lc2[mg], 0, 0, 0);
          }
      }
      if (oq >= 0) {
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
          const f32x4 lg = la[mg] + lc2[mg];
#pragma unroll
          for (int t = 0; t < 4; ++t)
            __builtin_amdgcn_ds_fmaxf((lds_float*)(Oc + (mg * 16 + 4 * g + t) * K1 + oq), lg[t], 0, 0, false);
        }
      }
    }
    if constexpr (!QC) {
      if (has_next) commit((it + 1) & 1);
    }
    __syncthreads();                                                       // (3) per-object maxima (and the next pair's rows)
    if (tid < MG * 16 && p0 + tid < n) {
      float best = Oc[tid * K1];
      int bi = 0;
      for (int o = 1; o <= K; ++o) {
        const float v = Oc[tid * K1 + o];
        if (v > best) {
          best = v;
          bi = o;
        }
      }
      gst(labels + p0 + tid, (unsigned char)bi);
      atomicAdd(&hist[bi], 1);
    }
    const int rows = min(MG * 16, n - p0);
    for (int e = tid; e < rows * K1; e += 512) gst(logits + (size_t)p0 * K1 + e, Oc[e]);
  }
  __syncthreads();
  for (int e = tid; e <= K; e += 512)
    if (hist[e]) atomicAdd(&counts[e], hist[e]);
}
