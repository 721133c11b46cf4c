// libagile3d_hip -- weight gradient of the sparse convolutions (gfx950), first kernel of the training path
// (what MinkowskiEngine's autograd computes for `losses.backward()`, engine.py:137-150).
//
//   dW[k][ci][co] = sum over the kernel map's pairs (input row r_in, output row r_out) of offset k of
//                   x[r_in][ci] * dy[r_out][co]
//
// for the four kernel-map kinds (3^3 stride 1, 2^3 stride 2, 2^3 transposed, 1x1), read from the same scene tables
// the forward kernels use.  Exact fp32 on v_mfma_f32_16x16x4_f32 with the ROWS as the MFMA k dimension: one MFMA
// multiplies a [16 ci x 4 rows] slice of x^T with a [4 rows x 16 co] slice of dy.  Lane (g, j) loads CX consecutive
// input channels of row g (one or two wide loads) and CY consecutive output channels: value tx of the x load is the
// A operand of the MFMAs for input-channel set {CX*j' + tx}, value ty of the dy load the B operand for output-channel
// set {CY*j' + ty} -- two loads feed CX*CY MFMAs (36 at 96 x 96 channels).  16-position groups that lack offset k are
// skipped with the forward kernels' group masks (rows are sorted by neighbour mask).  A workgroup = (row chunk,
// offset k, channel block); its four waves' accumulators are folded through LDS in wave order and the chunk partials
// are summed by a second kernel in chunk order: the result does not depend on scheduling.
#include "common.h"
#include <stdlib.h>

namespace a3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(8)));

struct WgradArgs {
  const float* x;        // [n_in][ldx]
  const float* dy;       // [n_out][ldy]
  int ldx, ldy, n_in, n_out;
  const int* tab;        // [K][tab_stride] position -> input row (>= n_in: no pair); nullptr = identity (1x1)
  int tab_stride;
  const int* out_map;    // position -> output row (transposed conv: virtual rows), nullptr = identity
  const uint32_t* gmask; // per 16 positions: bit k set if any of them has offset k; nullptr = all present
  int n_pos;             // positions (output rows of the kernel map)
  int K, cin, cout;
  int chunk_groups;      // 16-position groups per chunk
  int n_chunks;
  float* part;           // [chunks][K][cin][cout]
  float* bias_part;      // BIAS builds (K = 1): [chunks][cout] column sums of dy, the bias gradient of an nn.Linear
};

template <int C>
__device__ __forceinline__ void load_c(const float* p, bool ok, float (&v)[C]) {
#pragma unroll
  for (int i = 0; i < C; ++i) v[i] = 0.f;
  if (!ok) return;
  if constexpr (C == 2) {
    const f32x2 a = *(const f32x2*)p;
    v[0] = a[0], v[1] = a[1];
  } else if constexpr (C == 4) {
    const f32x4 a = *(const f32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
  } else if constexpr (C == 6) {   // 24 bytes per lane at 8-byte alignment: one dwordx4 + one dwordx2
    const f32x4u a = *(const f32x4u*)p;
    const f32x2 b = *(const f32x2*)(p + 4);
    v[0] = a[0], v[1] = a[1], v[2] = a[2], v[3] = a[3], v[4] = b[0], v[5] = b[1];
  } else {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i], v[4 + i] = b[i];
  }
}

template <int CX, int CY, bool BIAS = false>
__global__ void __launch_bounds__(256) k_wgrad(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fold[];   // [16 CX][16 CY] block of dW, waves fold in order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int nbx = a.cin / (16 * CX);
  // workgroup id -> (chunk, offset): consecutive ids go to different XCDs (id % 8), so the K offsets of one row chunk
  // are given to ONE XCD, back to back -- they read the same dy rows and neighbouring x rows, which then come out of
  // that XCD's L2 instead of being fetched once per offset
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int k = q % a.K;
  const int chunk = (q / a.K) * 8 + xcd;
  if (chunk >= a.n_chunks) return;
  const int bx = blockIdx.z % nbx, by = blockIdx.z / nbx;       // channel blocks
  const int ci0 = bx * 16 * CX + CX * j, co0 = by * 16 * CY + CY * j;
  const int ngroups = (a.n_pos + 15) >> 4;
  const int g_begin = chunk * a.chunk_groups;
  const int g_end = min(ngroups, g_begin + a.chunk_groups);
  f32x4 acc[CX][CY];
#pragma unroll
  for (int tx = 0; tx < CX; ++tx)
#pragma unroll
    for (int ty = 0; ty < CY; ++ty) acc[tx][ty] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int* tab = a.tab ? a.tab + (size_t)k * a.tab_stride : nullptr;

  // this wave's groups: g_begin + wave, +4, ... that have offset k.  Software pipeline across groups: the next
  // group's 16 row pairs are requested while the current group is multiplied, and its first four rows during the
  // current group's last step, so neither the table nor the row latency is exposed between groups.
  auto next_group = [&](int from) {
    int n = from;
    while (n < g_end && a.gmask && !((a.gmask[n] >> k) & 1u)) n += 4;
    return n;
  };
  auto load_pairs = [&](int grp, int& xi, int& yi) {
    const int p16 = grp * 16 + j;                                // lanes 0..15 of every 16-lane row hold the 16 pairs
    xi = a.n_in, yi = a.n_out;
    if (grp < g_end && p16 < a.n_pos) {
      xi = tab ? tab[p16] : p16;
      yi = a.out_map ? a.out_map[p16] : p16;
    }
  };
  float xv[CX], yv[CY], x1[CX], y1[CY], x2[CX], y2[CY];
  float ys[CY];          // BIAS: this lane's column sums of the dy rows it multiplies (rows without a pair load zeros)
#pragma unroll
  for (int i = 0; i < CY; ++i) ys[i] = 0.f;
  auto load_rows = [&](int xi, int yi, int s, float (&xd)[CX], float (&yd)[CY]) {   // rows of positions 4s .. 4s+3
    const int xr = __shfl(xi, 4 * s + g, 16), yr = __shfl(yi, 4 * s + g, 16);
    const bool ok = xr < a.n_in && yr < a.n_out;
    load_c<CX>(a.x + (size_t)(ok ? xr : 0) * a.ldx + ci0, ok, xd);
    load_c<CY>(a.dy + (size_t)(ok ? yr : 0) * a.ldy + co0, ok, yd);
  };
  int grp = next_group(g_begin + wave);
  int xi, yi;
  load_pairs(grp, xi, yi);
  if (grp < g_end) {
    load_rows(xi, yi, 0, x1, y1);
    load_rows(xi, yi, 1, x2, y2);
  }
  while (grp < g_end) {
    const int ngrp = next_group(grp + 4);
    int xi_n, yi_n;
    load_pairs(ngrp, xi_n, yi_n);
    // (round 5: four rotating row buffers with the step loop unrolled -- no register copies -- measured 8 % SLOWER on
    // <6,6> (462 -> 498 us): the copies are not what this loop waits for; profiles/r05_experiments.txt)
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < CX; ++i) xv[i] = x1[i], x1[i] = x2[i];
#pragma unroll
      for (int i = 0; i < CY; ++i) yv[i] = y1[i], y1[i] = y2[i];
      // two steps ahead (2 waves per SIMD do not cover a row fetch with one step of MFMAs): rows of step s+2 of this
      // group, or of the next group's first steps; past the end it re-reads rows that are never multiplied
      load_rows(s < 2 ? xi : xi_n, s < 2 ? yi : yi_n, (s + 2) & 3, x2, y2);
      if constexpr (BIAS) {
#pragma unroll
        for (int i = 0; i < CY; ++i) ys[i] += yv[i];
      }
#pragma unroll
      for (int tx = 0; tx < CX; ++tx)
#pragma unroll
        for (int ty = 0; ty < CY; ++ty)
          acc[tx][ty] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[tx], yv[ty], acc[tx][ty], 0, 0, 0);
    }
    grp = ngrp, xi = xi_n, yi = yi_n;
  }
  // fold the four waves in wave order, in the accumulators' own layout (one conflict-free 16-byte LDS access per
  // tile and lane), and write the partial in that layout too: [chunk][k][block][tile][lane] x 4 floats -- the reduce
  // kernel does the (tile, lane, r) -> (ci, co) mapping once per weight instead of once per workgroup
  f32x4* fold4 = (f32x4*)fold;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int tx = 0; tx < CX; ++tx)
#pragma unroll
        for (int ty = 0; ty < CY; ++ty) {
          f32x4* f = fold4 + (tx * CY + ty) * 64 + lane;
          *f = w == 0 ? acc[tx][ty] : *f + acc[tx][ty];
        }
    }
    __syncthreads();
  }
  constexpr int TILE4 = CX * CY * 64;   // f32x4 per block
  f32x4* P = (f32x4*)a.part + (((size_t)chunk * a.K + k) * gridDim.z + blockIdx.z) * TILE4;
  for (int e = threadIdx.x; e < TILE4; e += 256) P[e] = fold4[e];
  if constexpr (BIAS) {
    // the chunk's column sums of dy (the input-channel block 0 writes them): the lane's rows, then the four row lanes g,
    // then the four waves in wave order -- a fixed order like the weights'
    if (bx != 0) return;
#pragma unroll
    for (int i = 0; i < CY; ++i) {
      ys[i] += __shfl_xor(ys[i], 16, 64);
      ys[i] += __shfl_xor(ys[i], 32, 64);
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < CY; ++i) fold[wave * 16 * CY + CY * j + i] = ys[i];
    }
    __syncthreads();
    if (threadIdx.x < 16 * CY) {
      const int t = threadIdx.x;
      a.bias_part[(size_t)chunk * a.cout + by * 16 * CY + t] = ((fold[t] + fold[16 * CY + t]) + fold[32 * CY + t]) + fold[48 * CY + t];
    }
  }
}

// dw[k][ci][co] = sum over chunks of the fragment-layout partials: element e of a block is
// (tile = tx * CY + ty, lane = 16 g + j, r) -> ci = block_x * 16 CX + CX (4g + r) + tx, co = block_y * 16 CY + CY j + ty.
// 64 elements x kRedLanes chunk lanes per workgroup: lane p adds chunks p, p + kRedLanes, ... (independent loads in flight;
// one thread per element walking up to 256 chunks was a chain of dependent round trips: 25-30 us per weight, 120 launches
// per training iteration), the lanes' sums are folded in lane order -- a fixed order either way.
constexpr int kRedLanes = 8;
__global__ void __launch_bounds__(64 * kRedLanes) k_wgrad_reduce(const float* __restrict__ part, int nchunk, int K, int cin, int cout,
                                                                 int cx, int cy, float* __restrict__ dw) {
  __shared__ float sh[kRedLanes][64];
  const size_t total = (size_t)K * cin * cout;
  const int el = threadIdx.x & 63, p = threadIdx.x >> 6;
  const size_t e = (size_t)blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < total)
    for (int c = p; c < nchunk; c += kRedLanes) s += part[(size_t)c * total + e];
  sh[p][el] = s;
  __syncthreads();
  if (p != 0 || e >= total) return;
#pragma unroll
  for (int q = 1; q < kRedLanes; ++q) s += sh[q][el];
  const int block_elems = 16 * cx * 16 * cy, nbx = cin / (16 * cx);
  const int per_k = cin * cout;
  const int k = (int)(e / per_k), rem = (int)(e % per_k);
  const int blk = rem / block_elems, in_blk = rem % block_elems;
  const int tile = in_blk / 256, lane = (in_blk % 256) / 4, r = in_blk & 3;
  const int tx = tile / cy, ty = tile % cy, g = lane >> 4, j = lane & 15;
  const int ci = (blk % nbx) * 16 * cx + cx * (4 * g + r) + tx;
  const int co = (blk / nbx) * 16 * cy + cy * j + ty;
  dw[((size_t)k * cin + ci) * cout + co] = s;
}

// the same sum for an nn.Linear's gradients with the destination's layout as arguments (a3d_linear_wgrad_into): dW written
// as [cin][cout] or transposed ([cout][cin]: nn.Linear.weight's own layout) with leading dimension `ld`, assigned or
// ADDED to what is there; elements past the weights are the bias gradient (column sums of dy over the chunks)
__global__ void __launch_bounds__(64 * kRedLanes) k_wgrad_reduce_into(const float* __restrict__ part, const float* __restrict__ bias_part,
                                                                      int nchunk, int cin, int cout, int cx, int cy, float* dw, int ld,
                                                                      int transposed, int accumulate, float* db, int db_accumulate) {
  __shared__ float sh[kRedLanes][64];
  const int total = cin * cout;
  const int el = threadIdx.x & 63, p = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int all = total + (db ? cout : 0);
  float s = 0.f;
  if (e < total) {
    for (int c = p; c < nchunk; c += kRedLanes) s += part[(size_t)c * total + e];
  } else if (e < all) {
    for (int c = p; c < nchunk; c += kRedLanes) s += bias_part[(size_t)c * cout + (e - total)];
  }
  sh[p][el] = s;
  __syncthreads();
  if (p != 0 || e >= all) return;
#pragma unroll
  for (int q = 1; q < kRedLanes; ++q) s += sh[q][el];
  if (e >= total) {
    float* d = db + (e - total);
    *d = db_accumulate ? *d + s : s;
    return;
  }
  const int block_elems = 16 * cx * 16 * cy, nbx = cin / (16 * cx);
  const int blk = e / block_elems, in_blk = e % block_elems;
  const int tile = in_blk / 256, lane = (in_blk % 256) / 4, r = in_blk & 3;
  const int tx = tile / cy, ty = tile % cy, g = lane >> 4, j = lane & 15;
  const int ci = (blk % nbx) * 16 * cx + cx * (4 * g + r) + tx;
  const int co = (blk / nbx) * 16 * cy + cy * j + ty;
  float* d = dw + (transposed ? (size_t)co * ld + ci : (size_t)ci * ld + co);
  *d = accumulate ? *d + s : s;
}

struct WgradPlan {
  int cx, cy, nblocks, chunks, chunk_groups;
  size_t part_bytes;
};

static bool wgrad_plan(int n_pos, int K, int cin, int cout, WgradPlan& p) {
  if (cin % 32 || cout % 32 || cin < 32 || cout < 32) return false;
  int best = 0;
  const int cand[4] = {8, 6, 4, 2};
  for (int ix = 0; ix < 4; ++ix)
    for (int iy = 0; iy < 4; ++iy) {
      const int cx = cand[ix], cy = cand[iy];
      if ((cin / 16) % cx || (cout / 16) % cy || cx * cy > 36) continue;
      if (cx * cy > best) best = cx * cy, p.cx = cx, p.cy = cy;
    }
  if (!best) return false;
  p.nblocks = (cin / (16 * p.cx)) * (cout / (16 * p.cy));
  const int ngroups = (n_pos + 15) / 16;
  // workgroups over the whole launch: (chunk, offset) items differ a lot in work (rows are sorted by neighbour mask,
  // so an offset's pairs cluster in some chunks) -- many small items balance the big levels (measured: 3072 at 320 k
  // rows, 1536 below); every chunk costs a fold and a pass of the reduce kernel, so 1x1 maps stay at <= 256 chunks
  const int tgt = n_pos > 200000 ? 3072 : 1536;
  int chunks = (tgt + K * p.nblocks - 1) / (K * p.nblocks);
  if (chunks > 256) chunks = 256;
  // (round 5: fewer chunks on the small levels -- at least 9 row groups per wave, so that a 256 -> 256 layer of level 4
  // writes 14 MB of partials instead of 56 -- measured SLOWER, <8,4> 66 -> 88 us: a wave's groups are a chain of two-deep
  // row fetches, more workgroups hide it better than less partial traffic pays)
  if (chunks > (ngroups + 3) / 4) chunks = (ngroups + 3) / 4;      // at least one group per wave
  if (chunks < 1) chunks = 1;
  p.chunk_groups = (ngroups + chunks - 1) / chunks;
  p.chunks = (ngroups + p.chunk_groups - 1) / p.chunk_groups;
  p.part_bytes = (size_t)p.chunks * K * cin * cout * sizeof(float);
  return true;
}

static int wgrad_tables(const a3d_scene* s, int kind, int level_in, WgradArgs& a) {
  if (!s || level_in < 0 || level_in >= A3D_NUM_LEVELS) {
    set_error("a3d_conv_wgrad: bad scene / level");
    return A3D_ERR_INVALID;
  }
  a.tab = nullptr, a.out_map = nullptr, a.gmask = nullptr;
  switch (kind) {
    case A3D_OP_CONV3:
      a.K = 27, a.n_in = a.n_out = a.n_pos = s->lv[level_in].n;
      a.tab = s->lv[level_in].nbr27, a.tab_stride = s->lv[level_in].npad, a.gmask = s->lv[level_in].gmask27;
      break;
    case A3D_OP_DOWN:
      if (level_in >= A3D_NUM_LEVELS - 1) goto bad;
      a.K = 8, a.n_in = s->lv[level_in].n, a.n_out = a.n_pos = s->lv[level_in + 1].n;
      a.tab = s->lv[level_in].child8, a.tab_stride = s->lv[level_in + 1].npad, a.gmask = s->lv[level_in].gmask_down;
      break;
    case A3D_OP_UP:
      if (level_in < 1) goto bad;
      a.K = 8, a.n_in = s->lv[level_in].n, a.n_out = a.n_pos = s->lv[level_in - 1].n;
      a.tab = s->lv[level_in - 1].up8, a.tab_stride = s->lv[level_in - 1].npad, a.gmask = s->lv[level_in - 1].gmask_up;
      a.out_map = s->lv[level_in - 1].up_rows;
      break;
    case A3D_OP_LINEAR:
      a.K = 1, a.n_in = a.n_out = a.n_pos = s->lv[level_in].n;
      break;
    default:
    bad:
      set_error("a3d_conv_wgrad: kind %d does not exist at level %d", kind, level_in);
      return A3D_ERR_INVALID;
  }
  return A3D_OK;
}

// ---- weight gradient of the input convolution (5^3 or 3^3, 3 -> 32 channels; res16unet.py:225): the neighbours are
// looked up on the fly like in the forward kernel (dense level-0 grid or hash), one workgroup = (offset k, row split);
// a thread owns whole rows and keeps the 3 x 32 outer-product sums in registers, the workgroup folds them in a
// fixed-order LDS tree, the splits are summed in order by k_stem_wgrad_reduce.
constexpr int kStemSplits = 8;
__global__ void __launch_bounds__(256) k_stem_wgrad(const Level lv, const float* __restrict__ feats3,
                                                    const int* __restrict__ orig_row, const float* __restrict__ dy,
                                                    int lddy, int ks, float* __restrict__ part) {
  __shared__ float red[256][33];
  const int k = blockIdx.x, split = blockIdx.y, h = ks / 2;
  const int dx = k % ks - h, dyo = (k / ks) % ks - h, dz = k / (ks * ks) - h;
  const int per = (lv.n + kStemSplits - 1) / kStemSplits;
  const int r0 = split * per, r1 = min(lv.n, r0 + per);
  float acc[3][32];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int o = 0; o < 32; ++o) acc[c][o] = 0.f;
  for (int i = r0 + threadIdx.x; i < r1; i += 256) {
    const int4 c = *(const int4*)(lv.xyzb + 4 * i);
    const int X = c.x + dx, Y = c.y + dyo, Z = c.z + dz;
    int nb = -1;
    if (lv.grid && h <= kGridPad) {
      nb = lv.grid[grid_cell(lv, c.w, X, Y, Z)];
      if (nb >= 0) nb = lv.perm[nb];   // the grid holds Morton rows
    } else {
      const int lim = kCoordOff;
      if (X >= -lim && X < lim && Y >= -lim && Y < lim && Z >= -lim && Z < lim)
        nb = hash_lookup(lv.hkeys, lv.hvals, lv.hmask, make_key(c.w, X, Y, Z, 0));
    }
    if (nb < 0) continue;
    const float* fr = feats3 + (size_t)orig_row[nb] * 3;
    const float f0 = fr[0], f1 = fr[1], f2 = fr[2];
    const f32x4* dr = (const f32x4*)(dy + (size_t)i * lddy);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 d = dr[q];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0][4 * q + t] += f0 * d[t];
        acc[1][4 * q + t] += f1 * d[t];
        acc[2][4 * q + t] += f2 * d[t];
      }
    }
  }
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int o = 0; o < 32; ++o) red[threadIdx.x][o] = acc[c][o];
    __syncthreads();
    for (int stride = 128; stride >= 1; stride >>= 1) {
      for (int e = threadIdx.x; e < stride * 32; e += 256) red[e >> 5][e & 31] += red[(e >> 5) + stride][e & 31];
      __syncthreads();
    }
    if (threadIdx.x < 32) part[((size_t)split * gridDim.x + k) * 96 + c * 32 + threadIdx.x] = red[0][threadIdx.x];
    __syncthreads();
  }
}
// 64 elements x 8 part lanes per workgroup (as k_wgrad_reduce): lane p adds parts p, p + 8, ..., the lanes fold in order
__global__ void __launch_bounds__(512) k_stem_wgrad_reduce(const float* __restrict__ part, int total, float* __restrict__ dw,
                                                           int nparts, int stride) {
  __shared__ float sh[8][64];
  const int el = threadIdx.x & 63, p = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < total)
    for (int sp = p; sp < nparts; sp += 8) s += part[(size_t)sp * stride + e];
  sh[p][el] = s;
  __syncthreads();
  if (p != 0 || e >= total) return;
#pragma unroll
  for (int q = 1; q < 8; ++q) s += sh[q][el];
  dw[e] = s;
}

// The same gradient on the matrix cores (round 5; dense level-0 grid only): dW as ONE [384 x 32] accumulator per wave --
// m = 3 k + ci (375 of 384 rows used), n = co --, the voxels as the MFMA k dimension, four per step: lane (g, j) supplies
// A[m = 16 mt + j][voxel g] = the neighbour's colour (0 when the cell is empty) for the 24 row tiles and B[voxel g][n] = dy
// for the two column tiles, 48 MFMAs per step.  Voxels are walked in MORTON order (a wave's 4-voxel steps are neighbours:
// their 125-cell lookups share cache lines) against colours gathered into Morton order; the grid holds Morton rows, so a
// lookup is grid cell -> colour, two dependent loads.  The thread-per-row kernel above read every dy row once per offset
// (125 x) and folded 96 sums per thread through LDS: 1.18 ms at 320 k voxels.
constexpr int kStemMfmaWgs = 512;
__global__ void __launch_bounds__(256) k_stem_wgrad_mfma(const Level lv, const f32x4* __restrict__ feats4m,
                                                         const float* __restrict__ dy, int lddy, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float fold[];   // [384][32]
  constexpr int KS = 5, K = 125, NMT = 24;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  // this lane's 24 (offset, colour channel) pairs: m = 16 mt + j -> k = m / 3, ci = m % 3; relative cell offsets packed
  int cell_dx[NMT], cell_ci[NMT];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) {
    const int m = 16 * mt + j, k = m / 3;
    cell_ci[mt] = m < 3 * K ? m - 3 * k : -1;
    const int ox = k % KS - 2, oy = (k / KS) % KS - 2, oz = k / (KS * KS) - 2;
    cell_dx[mt] = m < 3 * K ? (oz * lv.gdim[1] + oy) * lv.gdim[0] + ox : 0;
  }
  f32x4 acc[NMT][2];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) acc[mt][0] = acc[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nsteps = (lv.n + 3) >> 2;
  const int W = gridDim.x * 4;
  const float* f1 = (const float*)feats4m;
  for (int st = blockIdx.x * 4 + wave; st < nsteps; st += W) {
    const int mr = 4 * st + g;                       // this lane's voxel (Morton row)
    const bool ok = mr < lv.n;
    size_t cell0 = 0;
    int irow = 0;
    if (ok) {
      int b_, X_, Y_, Z_;
      decode_key(lv.keys[mr], 0, b_, X_, Y_, Z_);
      cell0 = grid_cell(lv, b_, X_, Y_, Z_);
      irow = lv.perm[mr];
    }
    const float b0 = ok ? dy[(size_t)irow * lddy + j] : 0.f, b1 = ok ? dy[(size_t)irow * lddy + 16 + j] : 0.f;
    int nb[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) nb[mt] = (ok && cell_ci[mt] >= 0) ? lv.grid[(ptrdiff_t)cell0 + cell_dx[mt]] : -1;
    float av[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) av[mt] = nb[mt] >= 0 ? f1[4 * (size_t)nb[mt] + cell_ci[mt]] : 0.f;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], b0, acc[mt][0], 0, 0, 0);
      acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], b1, acc[mt][1], 0, 0, 0);
    }
  }
  // fold the four waves in wave order: lane (g, j) holds D[m = 16 mt + 4 g + r][n = 16 nt + j]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* f = fold + (16 * mt + 4 * g + r) * 32 + 16 * nt + j;
            *f = w == 0 ? acc[mt][nt][r] : *f + acc[mt][nt][r];
          }
    }
    __syncthreads();
  }
  float* P = part + (size_t)blockIdx.x * (384 * 32);
  for (int e = threadIdx.x; e < 384 * 32; e += 256) P[e] = fold[e];
}

}  // namespace a3d

using namespace a3d;

// dW [cin][cout] = x^T dy for plain row-major matrices of n rows (the nn.Linear layers of the decoder that run over all
// points or over the queries: attention_block.py, agile3d.py:51-55): the 1x1 case of a3d_conv_wgrad without a scene
extern "C" size_t a3d_linear_wgrad_workspace_bytes(int64_t n, int cin, int cout) {
  WgradPlan p;
  if (n <= 0 || n > (int64_t)1 << 30 || !wgrad_plan((int)n, 1, cin, cout, p)) {
    set_error("a3d_linear_wgrad: channels must be multiples of 32 (got %d -> %d)", cin, cout);
    return 0;
  }
  return p.part_bytes + 256;
}
extern "C" int a3d_linear_wgrad(const float* x_dev, int ldx, const float* dy_dev, int ldy, int64_t n, int cin, int cout,
                                float* dw_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  WgradPlan p;
  if (!x_dev || !dy_dev || !dw_dev || !workspace_dev || n <= 0 || n > (int64_t)1 << 30 || ldx < cin || ldy < cout ||
      (ldx & 1) || (ldy & 1) || !wgrad_plan((int)n, 1, cin, cout, p)) {
    set_error("a3d_linear_wgrad: bad arguments (channels multiples of 32, even leading dimensions)");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < p.part_bytes || ((uintptr_t)workspace_dev & 15)) {
    set_error("a3d_linear_wgrad: workspace too small or misaligned");
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x_dev, a.dy = dy_dev, a.ldx = ldx, a.ldy = ldy, a.cin = cin, a.cout = cout;
  a.K = 1, a.n_in = a.n_out = a.n_pos = (int)n;
  a.chunk_groups = p.chunk_groups;
  a.n_chunks = p.chunks;
  a.part = (float*)workspace_dev;
  const dim3 grid((unsigned)((p.chunks + 7) / 8 * 8), 1, p.nblocks);
  const size_t lds = (size_t)16 * p.cx * 16 * p.cy * sizeof(float);
#define A3D_WG(CX_, CY_)                                                                                          \
  if (p.cx == CX_ && p.cy == CY_) {                                                                               \
    A3D_ALLOW_LDS(64 * 1024, k_wgrad<CX_, CY_>); \
    k_wgrad<CX_, CY_><<<grid, 256, lds, st>>>(a);                                                                  \
  } else
  A3D_WG(2, 2) A3D_WG(2, 4) A3D_WG(2, 6) A3D_WG(2, 8) A3D_WG(4, 2) A3D_WG(4, 4) A3D_WG(4, 6) A3D_WG(4, 8)
  A3D_WG(6, 2) A3D_WG(6, 4) A3D_WG(6, 6) A3D_WG(8, 2) A3D_WG(8, 4) {
    set_error("a3d_linear_wgrad: no kernel for %d x %d channels per lane", p.cx, p.cy);
    return A3D_ERR_UNSUPPORTED;
  }
#undef A3D_WG
  A3D_LAUNCH_CHECK();
  k_wgrad_reduce<<<(unsigned)(((size_t)cin * cout + 63) / 64), 64 * kRedLanes, 0, st>>>(a.part, p.chunks, 1, cin, cout, p.cx, p.cy,
                                                                                        dw_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

// the same gradient written where the caller keeps it: dW as [cin][cout] or [cout][cin] (transposed = 1, nn.Linear.weight's
// layout; a row slice of a packed in_proj matrix is such a block at an offset) with leading dimension ld_dw, assigned or
// added (accumulate); db_dev != nullptr: the bias gradient (column sums of dy) from the SAME pass over dy -- what took a
// transposing copy, a zero-filled full-size matrix + slice copy + add per in_proj slice and two column-sum launches per layer
extern "C" size_t a3d_linear_wgrad_into_workspace_bytes(int64_t n, int cin, int cout) {
  WgradPlan p;
  if (n <= 0 || n > (int64_t)1 << 30 || !wgrad_plan((int)n, 1, cin, cout, p)) {
    set_error("a3d_linear_wgrad_into: channels must be multiples of 32 (got %d -> %d)", cin, cout);
    return 0;
  }
  return align256(p.part_bytes) + align256((size_t)p.chunks * cout * sizeof(float)) + 256;
}
extern "C" int a3d_linear_wgrad_into(const float* x_dev, int ldx, const float* dy_dev, int ldy, int64_t n, int cin, int cout,
                                     float* dw_dev, int ld_dw, int transposed, int accumulate, float* db_dev, int db_accumulate,
                                     void* workspace_dev, size_t workspace_bytes, void* stream) {
  WgradPlan p;
  if (!x_dev || !dy_dev || !dw_dev || !workspace_dev || n <= 0 || n > (int64_t)1 << 30 || ldx < cin || ldy < cout ||
      (ldx & 1) || (ldy & 1) || ld_dw < (transposed ? cin : cout) || !wgrad_plan((int)n, 1, cin, cout, p)) {
    set_error("a3d_linear_wgrad_into: bad arguments (channels multiples of 32, even leading dimensions, ld_dw >= row length)");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < a3d_linear_wgrad_into_workspace_bytes(n, cin, cout) - 256 || ((uintptr_t)workspace_dev & 15)) {
    set_error("a3d_linear_wgrad_into: workspace too small or misaligned");
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x_dev, a.dy = dy_dev, a.ldx = ldx, a.ldy = ldy, a.cin = cin, a.cout = cout;
  a.K = 1, a.n_in = a.n_out = a.n_pos = (int)n;
  a.chunk_groups = p.chunk_groups;
  a.n_chunks = p.chunks;
  a.part = (float*)workspace_dev;
  a.bias_part = (float*)((char*)workspace_dev + align256(p.part_bytes));
  const dim3 grid((unsigned)((p.chunks + 7) / 8 * 8), 1, p.nblocks);
  const size_t lds = (size_t)16 * p.cx * 16 * p.cy * sizeof(float);
#define A3D_WG(CX_, CY_)                                                                                          \
  if (p.cx == CX_ && p.cy == CY_) {                                                                               \
    A3D_ALLOW_LDS(64 * 1024, (k_wgrad<CX_, CY_, true>));                                                           \
    A3D_ALLOW_LDS(64 * 1024, (k_wgrad<CX_, CY_, false>));                                                          \
    if (db_dev) k_wgrad<CX_, CY_, true><<<grid, 256, lds, st>>>(a);                                                \
    else k_wgrad<CX_, CY_, false><<<grid, 256, lds, st>>>(a);                                                      \
  } else
  A3D_WG(2, 2) A3D_WG(2, 4) A3D_WG(2, 6) A3D_WG(2, 8) A3D_WG(4, 2) A3D_WG(4, 4) A3D_WG(4, 6) A3D_WG(4, 8)
  A3D_WG(6, 2) A3D_WG(6, 4) A3D_WG(6, 6) A3D_WG(8, 2) A3D_WG(8, 4) {
    set_error("a3d_linear_wgrad_into: no kernel for %d x %d channels per lane", p.cx, p.cy);
    return A3D_ERR_UNSUPPORTED;
  }
#undef A3D_WG
  A3D_LAUNCH_CHECK();
  const int all = cin * cout + (db_dev ? cout : 0);
  k_wgrad_reduce_into<<<(unsigned)((all + 63) / 64), 64 * kRedLanes, 0, st>>>(a.part, a.bias_part, p.chunks, cin, cout, p.cx, p.cy, dw_dev,
                                                                            ld_dw, transposed, accumulate, db_dev, db_accumulate);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" size_t a3d_stem_wgrad_workspace_bytes(int kernel_volume) {
  if (kernel_volume != 125 && kernel_volume != 27) return 0;
  // the larger of the two paths: per-offset splits (hash / 3^3) or the matrix-core kernel's per-workgroup partials
  const size_t a = (size_t)kStemSplits * kernel_volume * 96 * sizeof(float);
  const size_t b = (size_t)kStemMfmaWgs * 384 * 32 * sizeof(float);
  return (a > b ? a : b) + 256;
}
extern "C" size_t a3d_stem_wgrad_scene_workspace_bytes(const a3d_scene* s, int kernel_volume) {
  const size_t base = a3d_stem_wgrad_workspace_bytes(kernel_volume);
  if (!s || !base) return 0;
  return align256((size_t)kStemMfmaWgs * 384 * 32 * sizeof(float)) + (size_t)s->lv[0].npad * 16 + 256 > base
             ? align256((size_t)kStemMfmaWgs * 384 * 32 * sizeof(float)) + (size_t)s->lv[0].npad * 16 + 256
             : base;
}
// Morton-ordered colours for the matrix-core path (spconv.hip's k_gather_feats: feats4[f] = colours of Morton row f)
__global__ void k_stem_gather_morton(const float* __restrict__ feats3, const int* __restrict__ orig_row, const int* __restrict__ perm,
                                     int n, f32x4* __restrict__ feats4) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const float* p = feats3 + (size_t)orig_row[perm[f]] * 3;
  feats4[f] = (f32x4){p[0], p[1], p[2], 0.f};
}

extern "C" int a3d_stem_wgrad(const a3d_scene* s, const float* feats3_dev, const float* dy_dev, int lddy,
                              int kernel_volume, float* dw_dev, void* workspace_dev, size_t workspace_bytes,
                              void* stream) {
  if (!s || !feats3_dev || !dy_dev || !dw_dev || !workspace_dev || lddy < 32 || (lddy & 3) ||
      (kernel_volume != 125 && kernel_volume != 27)) {
    set_error("a3d_stem_wgrad: bad arguments (5^3 or 3^3 kernel, dy with 32 columns, leading dimension a multiple of 4)");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < a3d_stem_wgrad_workspace_bytes(kernel_volume) || ((uintptr_t)workspace_dev & 15)) {
    set_error("a3d_stem_wgrad: workspace too small or misaligned");
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int ks = kernel_volume == 125 ? 5 : 3;
  const int total = kernel_volume * 96;
  const Level& lv0 = s->lv[0];
  const size_t part_bytes = align256((size_t)kStemMfmaWgs * 384 * 32 * sizeof(float));
  if (ks == 5 && lv0.grid && lv0.n < (1 << 25) && workspace_bytes >= part_bytes + (size_t)lv0.npad * 16) {
    // matrix-core path: colours in Morton order (behind the partials in the workspace: a3d_stem_wgrad_scene_workspace_bytes),
    // one [384 x 32] accumulator per wave
    f32x4* f4 = (f32x4*)((char*)workspace_dev + part_bytes);
    k_stem_gather_morton<<<(lv0.n + 255) / 256, 256, 0, st>>>(feats3_dev, s->orig_row, lv0.perm, lv0.n, f4);
    const int steps = (lv0.n + 3) / 4;
    int wgs = (steps + 3) / 4;
    if (wgs > kStemMfmaWgs) wgs = kStemMfmaWgs;
    A3D_ALLOW_LDS(64 * 1024, k_stem_wgrad_mfma);
    k_stem_wgrad_mfma<<<wgs, 256, 384 * 32 * sizeof(float), st>>>(lv0, f4, dy_dev, lddy, (float*)workspace_dev);
    k_stem_wgrad_reduce<<<(total + 63) / 64, 512, 0, st>>>((const float*)workspace_dev, total, dw_dev, wgs, 384 * 32);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
  }
  k_stem_wgrad<<<dim3(kernel_volume, kStemSplits), 256, 0, st>>>(s->lv[0], feats3_dev, s->orig_row, dy_dev, lddy, ks,
                                                               (float*)workspace_dev);
  k_stem_wgrad_reduce<<<(total + 63) / 64, 512, 0, st>>>((const float*)workspace_dev, total, dw_dev, kStemSplits, total);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" size_t a3d_conv_wgrad_workspace_bytes(const a3d_scene* s, int kind, int level_in, int cin, int cout) {
  WgradArgs a;
  if (wgrad_tables(s, kind, level_in, a) != A3D_OK) return 0;
  WgradPlan p;
  if (!wgrad_plan(a.n_pos, a.K, cin, cout, p)) {
    set_error("a3d_conv_wgrad: channels must be multiples of 32 (got %d -> %d)", cin, cout);
    return 0;
  }
  return p.part_bytes + 256;
}

extern "C" int a3d_conv_wgrad(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx,
                              const float* dy_dev, int ldy, int cin, int cout, float* dw_dev, void* workspace_dev,
                              size_t workspace_bytes, void* stream) {
  WgradArgs a;
  int rc = wgrad_tables(s, kind, level_in, a);
  if (rc) return rc;
  WgradPlan p;
  if (!x_dev || !dy_dev || !dw_dev || !workspace_dev || ldx < cin || ldy < cout || (ldx & 1) || (ldy & 1) ||
      !wgrad_plan(a.n_pos, a.K, cin, cout, p)) {
    set_error("a3d_conv_wgrad: bad arguments (channels multiples of 32, even leading dimensions)");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < p.part_bytes || ((uintptr_t)workspace_dev & 15)) {
    set_error("a3d_conv_wgrad: workspace too small or misaligned (%zu < %zu)", workspace_bytes, p.part_bytes);
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  a.x = x_dev, a.dy = dy_dev, a.ldx = ldx, a.ldy = ldy, a.cin = cin, a.cout = cout;
  a.chunk_groups = p.chunk_groups;
  a.n_chunks = p.chunks;
  a.part = (float*)workspace_dev;
  const dim3 grid((unsigned)((p.chunks + 7) / 8 * 8 * a.K), 1, p.nblocks);
  const size_t lds = (size_t)16 * p.cx * 16 * p.cy * sizeof(float);
#define A3D_WG(CX_, CY_)                                                                                          \
  if (p.cx == CX_ && p.cy == CY_) {                                                                               \
    A3D_ALLOW_LDS(64 * 1024, k_wgrad<CX_, CY_>); \
    k_wgrad<CX_, CY_><<<grid, 256, lds, st>>>(a);                                                                  \
  } else
  A3D_WG(2, 2) A3D_WG(2, 4) A3D_WG(2, 6) A3D_WG(2, 8) A3D_WG(4, 2) A3D_WG(4, 4) A3D_WG(4, 6) A3D_WG(4, 8)
  A3D_WG(6, 2) A3D_WG(6, 4) A3D_WG(6, 6) A3D_WG(8, 2) A3D_WG(8, 4) {
    set_error("a3d_conv_wgrad: no kernel for %d x %d channels per lane", p.cx, p.cy);
    return A3D_ERR_UNSUPPORTED;
  }
#undef A3D_WG
  A3D_LAUNCH_CHECK();
  const size_t total = (size_t)a.K * cin * cout;
  k_wgrad_reduce<<<(unsigned)((total + 63) / 64), 64 * kRedLanes, 0, st>>>(a.part, p.chunks, a.K, cin, cout, p.cx, p.cy, dw_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
