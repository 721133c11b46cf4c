// libagile3d_hip -- weight gradient of the sparse convolutions (gfx950), first kernel of the training path
// (what MinkowskiEngine's autograd computes for `losses.backward()`, engine.py:137-150).
//
//   dW[k][ci][co] = sum over the kernel map's pairs (input row r_in, output row r_out) of offset k of
//                   x[r_in][ci] * dy[r_out][co]
//
// for the four kernel-map kinds (3^3 stride 1, 2^3 stride 2, 2^3 transposed, 1x1), read from the same scene tables
// the forward kernels use.  Exact fp32 on v_mfma_f32_16x16x4_f32 with the ROWS as the MFMA k dimension: one MFMA
// multiplies a [16 ci x 4 rows] slice of x^T with a [4 rows x 16 co] slice of dy.  Lane (g, j) loads CX input channels of
// row g (one or two wide loads, chan()) and CY output channels: value tx of the x load is the A operand of the MFMAs for
// input-channel set {chan(CX, j', tx)}, value ty of the dy load the B operand for output-channel set {chan(CY, j', ty)} --
// two loads feed CX*CY MFMAs (36 at 96 x 96 channels).  Only the 16-position groups that HAVE offset k are walked: the
// scene keeps, per map and offset, the ascending list of those groups (a3d_scene_build_wgrad_lists), and a workgroup =
// (offset k, a segment of k's list, channel block); all segments are equally long and every XCD gets the same number of them
// (wgrad_plan).  Its four waves' accumulators are folded through LDS in wave order and the segments' partials are summed by
// a second kernel in segment order: the result does not depend on scheduling.
#include "common.h"
#include <stdlib.h>
#include <mutex>
#include <unordered_map>
#include <vector>

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
  int n_pos;             // positions (output rows of the kernel map)
  int K, cin, cout;
  // work items: offset k's groups (list[k * list_stride + i], i < cnt[k]; nullptr: groups 0 .. cnt[0] - 1 of a 1x1 map) are
  // cut into segments of seg_len; segment s of offset k writes partial slot base[k] + s.  Workgroup b runs on XCD b % 8
  // (hardware) and takes item b / 8 of that XCD's items: offset k's segments seg_lo[k][x] .. seg_lo[k][x + 1] - 1 -- the
  // x-th eighth of its list, so an XCD's offsets work on about the same stretch of rows -- offset-major; xbase[x][k] =
  // items of XCD x before offset k.  The host makes the tables (wgrad_plan): every XCD gets the same number of items +- K / 8.
  const int* list;
  int list_stride;
  int seg_len;
  int cnt[27];
  int base[28];
  int xbase[8][28];
  int seg_lo[27][9];
  float* part;           // [slots][cin][cout] in fragment layout
  float* bias_part;      // BIAS builds (K = 1): [slots][cout] column sums of dy, the bias gradient of an nn.Linear
};

// Channels of a 16 C-wide block held by a lane: the 16 lanes of a row load CONTIGUOUS pieces -- min(C, 4) floats at
// 4 min(C, 4) j bytes, and for C = 6 / 8 a second piece of C - 4 floats behind the first 16 min(C, 4) floats -- so a
// wave's load instruction touches 2 (1) cache lines per row.  (Round 5's C consecutive floats per lane made the 16-byte
// and the 8-byte load of a 96-channel row touch all three of its lines each: 48 line accesses per step and wave, and
// the loads' time adds to the MFMAs' in this loop.)  Value t of lane j is channel chan(C, j, t) of the block.
__host__ __device__ constexpr int chan(int C, int j, int t) {
  return t < (C < 4 ? C : 4) ? (C < 4 ? C : 4) * j + t : 16 * 4 + (C - 4) * j + (t - 4);
}

// the lane's pieces at byte offset `off` (first piece) of a raw buffer: rows past the end (offset >= num_records) read as
// zeros in hardware -- no branch, no select, so the loop below is straight-line code whose loads the compiler counts
// exactly.  off2 - off = (64 - 4 j + (C - 4) j) floats: the second piece.
template <int C>
__device__ __forceinline__ void load_c(__amdgpu_buffer_rsrc_t r, int off, int off2, float (&v)[C]) {
  // (whole-vector bit casts: __builtin_bit_cast(float, a[i]) on an element of the returned vector makes this compiler load
  // one dword and use it for every element)
  if constexpr (C == 2) {
    const f32x2 a = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    v[0] = a[0], v[1] = a[1];
  } else if constexpr (C == 4) {
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
  } else if constexpr (C == 6) {
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    const f32x2 b = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off2, 0, 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
    v[4] = b[0], v[5] = b[1];
  } else {
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off2, 0, 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i], v[4 + i] = b[i];
  }
}

constexpr unsigned kNoRow = 0xFFFFF000u;   // byte offset of "no pair": past every buffer this kernel accepts (a3d: wgrad_fits)

// the two pieces of load_c on their own (the step loop spreads a step's loads over its MFMAs)
template <int C>
__device__ __forceinline__ void load_piece1(__amdgpu_buffer_rsrc_t r, int off, float (&v)[C]) {
  if constexpr (C == 2) {
    const f32x2 a = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    v[0] = a[0], v[1] = a[1];
  } else {
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
  }
}
template <int C>
__device__ __forceinline__ void load_piece2(__amdgpu_buffer_rsrc_t r, int off2, float (&v)[C]) {
  if constexpr (C == 6) {
    const f32x2 b = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off2, 0, 0));
    v[4] = b[0], v[5] = b[1];
  } else if constexpr (C == 8) {
    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off2, 0, 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) v[4 + i] = b[i];
  }
}

// 16 / 24 / 32 bytes through a plain pointer (WIDE builds: operands of 4 GB and more, which a buffer descriptor cannot span)
template <int C>
__device__ __forceinline__ void load_p(const float* p, const float* p2, float (&v)[C]) {
  if constexpr (C == 2) {
    const f32x2 a = *(const f32x2*)p;
    v[0] = a[0], v[1] = a[1];
  } else if constexpr (C == 4) {
    const f32x4u a = *(const f32x4u*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
  } else if constexpr (C == 6) {
    const f32x4u a = *(const f32x4u*)p;
    const f32x2 b = *(const f32x2*)p2;
    v[0] = a[0], v[1] = a[1], v[2] = a[2], v[3] = a[3], v[4] = b[0], v[5] = b[1];
  } else {
    const f32x4u a = *(const f32x4u*)p, b = *(const f32x4u*)p2;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i], v[4 + i] = b[i];
  }
}

template <int CX, int CY, bool BIAS = false, bool WIDE = false>
__global__ void __launch_bounds__(256) k_wgrad(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fold[];   // [16 CX][16 CY] block of dW, waves fold in order
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int nbx = a.cin / (16 * CX);
  // workgroup id -> (XCD x, item i of x) -> (offset k, segment)
  const int xcd = blockIdx.x & 7, item = blockIdx.x >> 3;
  if (item >= a.xbase[xcd][27]) return;
  int k = 0;
#pragma unroll
  for (int kk = 1; kk < 27; ++kk) k += item >= a.xbase[xcd][kk] ? 1 : 0;
  const int seg = a.seg_lo[k][xcd] + (item - a.xbase[xcd][k]);
  const int cnt = a.cnt[k];
  const int slot = a.base[k] + seg;
  const int bx = blockIdx.z % nbx, by = blockIdx.z / nbx;       // channel blocks
  const int e_begin = seg * a.seg_len;
  const int e_end = min(cnt, e_begin + a.seg_len);
  const int* glist = a.list ? a.list + (size_t)k * a.list_stride : nullptr;
  f32x4 acc[CX][CY];
#pragma unroll
  for (int tx = 0; tx < CX; ++tx)
#pragma unroll
    for (int ty = 0; ty < CY; ++ty) acc[tx][ty] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int* tab = a.tab ? a.tab + (size_t)k * a.tab_stride : nullptr;
  // x and dy as raw buffers of n rows: a row index >= n turns into an offset past the end, which the hardware answers
  // with zeros (the product of a missing pair)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((unsigned)a.n_in * (unsigned)a.ldx * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((unsigned)a.n_out * (unsigned)a.ldy * 4u), 0x00020000);
  const unsigned cx4 = (unsigned)(bx * 16 * CX + chan(CX, j, 0)) * 4u, cy4 = (unsigned)(by * 16 * CY + chan(CY, j, 0)) * 4u;
  const unsigned cx4b = (unsigned)(bx * 16 * CX + chan(CX, j, CX > 4 ? 4 : 0)) * 4u;      // the lane's second piece (CX = 6 / 8)
  const unsigned cy4b = (unsigned)(by * 16 * CY + chan(CY, j, CY > 4 ? 4 : 0)) * 4u;
  const unsigned ldx4 = (unsigned)a.ldx * 4u, ldy4 = (unsigned)a.ldy * 4u;

  // this wave's groups: entries e_begin + wave, +4, ... of offset k's list.  The 16 pairs of a group sit in lanes 0..15 of every
  // 16-lane row as BYTE OFFSETS of their two rows; step s (positions 4s .. 4s+3) fetches them with a bpermute and loads
  // CX + CY floats per lane.  Four row buffers rotate with the step loop unrolled: the rows of step s+3 are requested
  // before step s is multiplied and nothing is copied, so three steps of MFMAs (3 x CX CY x 8 passes) cover a row fetch.
  // (Rounds 3-5 rotated two buffers through register copies: the copy of a buffer one step after its load made every
  // step wait for ALL outstanding loads -- s_waitcnt vmcnt(0) -- and the "two steps ahead" was one.)
  auto group_of = [&](int e) { return e < e_end ? (glist ? glist[e] : e) : -1; };   // entry e of the list (wave-uniform: a scalar load)
  auto request_pairs = [&](int grp, int& xi, int& yi) {          // table entries of group grp (-1: none; judged later)
    const int p16 = grp * 16 + j;
    const int pc = (grp >= 0 && p16 < a.n_pos) ? p16 : 0;
    xi = tab ? tab[pc] : pc;
    yi = a.out_map ? a.out_map[pc] : pc;
  };
  auto pair_offsets = [&](int grp, int xi, int yi, unsigned& xo, unsigned& yo) {
    const bool ok = grp >= 0 && grp * 16 + j < a.n_pos && (unsigned)xi < (unsigned)a.n_in && (unsigned)yi < (unsigned)a.n_out;
    if constexpr (WIDE) {   // row indices instead of byte offsets; "no pair" = n (the row count)
      xo = ok ? (unsigned)xi : (unsigned)a.n_in;
      yo = ok ? (unsigned)yi : (unsigned)a.n_out;
    } else {
      xo = ok ? (unsigned)xi * ldx4 : kNoRow;
      yo = ok ? (unsigned)yi * ldy4 : kNoRow;
    }
  };
  int bperm[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) bperm[s] = ((lane & 48) + 4 * s + g) * 4;
  float xb[4][CX], yb[4][CY];
  float ys[CY];          // BIAS: this lane's column sums of the dy rows it multiplies (rows without a pair load zeros)
#pragma unroll
  for (int i = 0; i < CY; ++i) ys[i] = 0.f;
  auto load_rows = [&](unsigned xo, unsigned yo, int s, float (&xd)[CX], float (&yd)[CY]) {   // rows of positions 4s .. 4s+3
    if constexpr (WIDE) {
      const unsigned xr = (unsigned)__builtin_amdgcn_ds_bpermute(bperm[s], (int)xo);
      const unsigned yr = (unsigned)__builtin_amdgcn_ds_bpermute(bperm[s], (int)yo);
      const bool ok = xr < (unsigned)a.n_in;                       // both rows exist or neither (pair_offsets)
      const float* xp = a.x + (size_t)(ok ? xr : 0u) * a.ldx;
      const float* yp = a.dy + (size_t)(ok ? yr : 0u) * a.ldy;
      load_p<CX>(xp + (cx4 >> 2), xp + (cx4b >> 2), xd);
      load_p<CY>(yp + (cy4 >> 2), yp + (cy4b >> 2), yd);
#pragma unroll
      for (int i = 0; i < CX; ++i) xd[i] = ok ? xd[i] : 0.f;
#pragma unroll
      for (int i = 0; i < CY; ++i) yd[i] = ok ? yd[i] : 0.f;
    } else {
      const unsigned xr = (unsigned)__builtin_amdgcn_ds_bpermute(bperm[s], (int)xo);
      const unsigned yr = (unsigned)__builtin_amdgcn_ds_bpermute(bperm[s], (int)yo);
      load_c<CX>(rx, (int)(xr + cx4), (int)(xr + cx4b), xd);
      load_c<CY>(ry, (int)(yr + cy4), (int)(yr + cy4b), yd);
    }
  };
  auto multiply = [&](const float (&xv)[CX], const float (&yv)[CY]) {
    if constexpr (BIAS) {
#pragma unroll
      for (int i = 0; i < CY; ++i) ys[i] += yv[i];
    }
#pragma unroll
    for (int tx = 0; tx < CX; ++tx)
#pragma unroll
      for (int ty = 0; ty < CY; ++ty)
        acc[tx][ty] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[tx], yv[ty], acc[tx][ty], 0, 0, 0);
  };
  // one step: the MFMAs of (xv, yv) with the loads of a later step (rows 4 s2 .. of the group behind xo2 / yo2, into xd / yd)
  // SPREAD over them (same-box A/B against the step's two to four load instructions issued back to back: -2 .. -6 %)
  auto step = [&](const float (&xv)[CX], const float (&yv)[CY], unsigned xo2, unsigned yo2, int s2, float (&xd)[CX], float (&yd)[CY]) {
    if constexpr (WIDE) {
      load_rows(xo2, yo2, s2, xd, yd);
      __builtin_amdgcn_sched_barrier(0);
      multiply(xv, yv);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      const unsigned xr = (unsigned)__builtin_amdgcn_ds_bpermute(bperm[s2], (int)xo2);
      const unsigned yr = (unsigned)__builtin_amdgcn_ds_bpermute(bperm[s2], (int)yo2);
      if constexpr (BIAS) {
#pragma unroll
        for (int i = 0; i < CY; ++i) ys[i] += yv[i];
      }
      constexpr int nP = 2 + (CX > 4 ? 1 : 0) + (CY > 4 ? 1 : 0);           // load instructions of a step: x1 [x2] y1 [y2]
      constexpr int iY1 = CX > 4 ? 2 : 1, iY2 = iY1 + 1;
      auto slot = [](int i) { return i * CX / nP < CX - 1 ? i * CX / nP : CX - 1; };
#pragma unroll
      for (int tx = 0; tx < CX; ++tx) {
#pragma unroll
        for (int ty = 0; ty < CY; ++ty)
          acc[tx][ty] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[tx], yv[ty], acc[tx][ty], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (tx == slot(0)) load_piece1<CX>(rx, (int)(xr + cx4), xd);
        if (CX > 4 && tx == slot(1)) load_piece2<CX>(rx, (int)(xr + cx4b), xd);
        if (tx == slot(iY1)) load_piece1<CY>(ry, (int)(yr + cy4), yd);
        if (CY > 4 && tx == slot(iY2)) load_piece2<CY>(ry, (int)(yr + cy4b), yd);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  int e = e_begin + wave;                       // this wave's entries: e_begin + wave, + 4, ...
  int grp = group_of(e), ngrp = group_of(e + 4);
  int xi, yi, xi_n, yi_n;
  unsigned xo, yo, xo_n, yo_n;
  request_pairs(grp, xi, yi);
  request_pairs(ngrp, xi_n, yi_n);
  pair_offsets(grp, xi, yi, xo, yo);
  pair_offsets(ngrp, xi_n, yi_n, xo_n, yo_n);
  if (grp >= 0) {
    load_rows(xo, yo, 0, xb[0], yb[0]);
    load_rows(xo, yo, 1, xb[1], yb[1]);
    load_rows(xo, yo, 2, xb[2], yb[2]);
  }
  while (grp >= 0) {
    // the group after the next: its table entries are requested now and looked at only at the end of this iteration
    e += 4;
    const int nngrp = group_of(e + 4);
    int xi_nn, yi_nn;
    request_pairs(nngrp, xi_nn, yi_nn);
    // (scheduling barriers: left alone, the compiler sinks the loads behind two steps of MFMAs to save registers and
    // drains the queue to vmcnt(1) before it issues the next ones)
    step(xb[0], yb[0], xo, yo, 3, xb[3], yb[3]);
    step(xb[1], yb[1], xo_n, yo_n, 0, xb[0], yb[0]);     // past the last group: offsets are kNoRow, the loads touch no memory
    step(xb[2], yb[2], xo_n, yo_n, 1, xb[1], yb[1]);
    step(xb[3], yb[3], xo_n, yo_n, 2, xb[2], yb[2]);
    grp = ngrp, xo = xo_n, yo = yo_n;
    ngrp = nngrp;
    pair_offsets(nngrp, xi_nn, yi_nn, xo_n, yo_n);
  }
  // fold the four waves in wave order, in the accumulators' own layout (one conflict-free 16-byte LDS access per
  // tile and lane), and write the partial in that layout too: [slot][block][tile][lane] x 4 floats -- the reduce
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
  f32x4* P = (f32x4*)a.part + ((size_t)slot * gridDim.z + blockIdx.z) * TILE4;
  for (int e = threadIdx.x; e < TILE4; e += 256) P[e] = fold4[e];
  if constexpr (BIAS) {
    // the segment's column sums of dy (the input-channel block 0 writes them): the lane's rows, then the four row lanes g,
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
      for (int i = 0; i < CY; ++i) fold[wave * 16 * CY + chan(CY, j, i)] = ys[i];
    }
    __syncthreads();
    if (threadIdx.x < 16 * CY) {
      const int t = threadIdx.x;
      a.bias_part[(size_t)slot * a.cout + by * 16 * CY + t] = ((fold[t] + fold[16 * CY + t]) + fold[32 * CY + t]) + fold[48 * CY + t];
    }
  }
}

// dw[k][ci][co] = sum over offset k's slots of the fragment-layout partials: a block's tile (tx, ty) holds, in lane
// (g, j) and value r, the element ci = chan(CX, 4g + r, tx), co = chan(CY, j, ty).  One workgroup per (k, block, tx, half of
// the ty): the tiles ty = 0 .. 3 of a lane are four CONSECUTIVE output channels (4j .. 4j + 3; ty = 4 .. of CY = 6 / 8: 64 +
// (CY - 4) j ..), so a lane sums its tiles over the slots -- each tile of a partial is 1 KB contiguous, float4 per lane --
// and writes one 16-byte (8-byte) piece per r: the 16 lanes of a row fill 256 (128) contiguous bytes of an output row.
// Wave w adds slots w, w + NW, ... in ascending order, the waves' sums are folded in wave order through LDS: a fixed
// order.  (Rounds 3-5: one thread per element, consecutive threads on consecutive r = four different output rows, and a
// strided 4-byte read per slot: 0.75 TB/s on the 256 -> 256 layers of level 4.)
struct SlotBase {
  int base[28];          // offset k's partial slots are base[k] .. base[k + 1] - 1
};
constexpr int NW = 4;      // waves per workgroup (small workgroups: the kernel runs next to the input-gradient convs, whose
                           // workgroups leave a CU only a few wave slots at a time -- 1024-thread workgroups waited 100 us for a place)
__global__ void __launch_bounds__(64 * NW) k_wgrad_reduce(const float* __restrict__ part, const SlotBase sb, int cin, int cout, int cx, int cy,
                                                         float* __restrict__ dw) {
  __shared__ f32x4 sh[NW - 1][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  const int nhalf = cy > 4 ? 2 : 1, nbx = cin / (16 * cx), nblocks = nbx * (cout / (16 * cy));
  int u = blockIdx.x;
  const int half = u % nhalf;
  u /= nhalf;
  const int tx = u % cx;
  u /= cx;
  const int blk = u % nblocks, k = u / nblocks;
  const int t0 = half ? 4 : 0, nt = half ? cy - 4 : (cy < 4 ? cy : 4);      // 2 or 4 tiles
  const f32x4* P = (const f32x4*)part;
  const size_t tiles_per_slot = (size_t)nblocks * cx * cy;
  const size_t first = ((size_t)blk * cx * cy + tx * cy + t0) * 64 + lane;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // four slots per iteration in flight (a wave's slots are a chain of fetch -> add otherwise); the sums stay in slot order
  const size_t slot_stride = tiles_per_slot * 64;
  const int c_end = sb.base[k + 1];
  for (int c = sb.base[k] + wave; c < c_end; c += 4 * NW) {
    f32x4 v[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cq = c + q * NW;
      const f32x4* src = P + (size_t)(cq < c_end ? cq : c) * slot_stride + first;
#pragma unroll
      for (int t = 0; t < 4; ++t) v[q][t] = src[(t < nt ? t : 0) * 64];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (c + q * NW < c_end) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] += v[q][t];
      }
  }
  if (wave) {
#pragma unroll
    for (int t = 0; t < 4; ++t) sh[wave - 1][t][lane] = acc[t];
  }
  __syncthreads();
  if (wave) return;
#pragma unroll
  for (int w = 0; w < NW - 1; ++w)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] += sh[w][t][lane];
  const int co = (blk / nbx) * 16 * cy + chan(cy, j, t0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ci = (blk % nbx) * 16 * cx + chan(cx, 4 * g + r, tx);
    float* d = dw + ((size_t)k * cin + ci) * cout + co;
    if (nt == 4) *(f32x4*)d = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    else *(f32x2*)d = (f32x2){acc[0][r], acc[1][r]};
  }
}

constexpr int kRedLanes = 8;      // k_wgrad_reduce_into: 64 elements x 8 slot lanes per workgroup
// the same sum for an nn.Linear's gradients with the destination's layout as arguments (a3d_linear_wgrad_into): dW written
// as [cin][cout] or transposed ([cout][cin]: nn.Linear.weight's own layout) with leading dimension `ld`, assigned or
// ADDED to what is there; elements past the weights are the bias gradient (column sums of dy over the slots)
__global__ void __launch_bounds__(64 * kRedLanes) k_wgrad_reduce_into(const float* __restrict__ part, const float* __restrict__ bias_part,
                                                                      int nslots, int cin, int cout, int cx, int cy, float* dw, int ld,
                                                                      int transposed, int accumulate, float* db, int db_accumulate) {
  __shared__ float sh[kRedLanes][64];
  const int total = cin * cout;
  const int el = threadIdx.x & 63, p = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int all = total + (db ? cout : 0);
  float s = 0.f;
  if (e < total) {
    for (int c = p; c < nslots; c += kRedLanes) s += part[(size_t)c * total + e];
  } else if (e < all) {
    for (int c = p; c < nslots; c += kRedLanes) s += bias_part[(size_t)c * cout + (e - total)];
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
  const int ci = (blk % nbx) * 16 * cx + chan(cx, 4 * g + r, tx);
  const int co = (blk / nbx) * 16 * cy + chan(cy, j, ty);
  float* d = dw + (transposed ? (size_t)co * ld + ci : (size_t)ci * ld + co);
  *d = accumulate ? *d + s : s;
}

struct WgradPlan {
  int cx, cy, nblocks, seg_len, slots, grid_items;     // grid_items: items of the fullest XCD
  int cnt[27], base[28];
  int xbase[8][28], seg_lo[27][9];
  size_t part_bytes;
};

// the segment tables of one segment length; returns the items of the fullest XCD
static int wgrad_cut(const int* cnt, int K, int seg_len, WgradPlan& p) {
  p.seg_len = seg_len;
  p.base[0] = 0;
  for (int k = 0; k < 27; ++k) {
    const int ns = k < K ? (cnt[k] + seg_len - 1) / seg_len : 0;
    p.base[k + 1] = p.base[k] + ns;
    // eighths of the list; the remainder ns % 8 goes to different XCDs for different offsets
    for (int x = 0; x <= 8; ++x) p.seg_lo[k][x] = (int)(((long long)x * ns + (5 * k) % 8) / 8);
  }
  int most = 0;
  for (int x = 0; x < 8; ++x) {
    p.xbase[x][0] = 0;
    for (int k = 0; k < 27; ++k) p.xbase[x][k + 1] = p.xbase[x][k] + (p.seg_lo[k][x + 1] - p.seg_lo[k][x]);
    if (p.xbase[x][27] > most) most = p.xbase[x][27];
  }
  p.slots = p.base[27] > 0 ? p.base[27] : 1;
  p.grid_items = most;
  return most;
}

// counts: groups per offset (the scene's lists), nullptr: every one of the ngroups groups (1x1 maps)
static bool wgrad_plan(int ngroups, int K, const int* counts, int cin, int cout, WgradPlan& p) {
  if (cin % 32 || cout % 32 || cin < 32 || cout < 32 || K > 27 || ngroups < 1) return false;
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
  // Work items of EQUAL size, the same number on every XCD.  Rounds 1-5 cut the ROWS into equal chunks per offset; rows are
  // sorted by neighbour mask, so an offset's pairs cluster in some chunks and the (chunk, offset) items differed 0 .. 45
  // groups, and a chunk's offsets went to one XCD whatever they held: the compute units stood idle a quarter of the
  // level-0 launch (SQ_BUSY_CU_CYCLES), more on the small levels (level 4: five of eight XCDs had work at all).
  // The segment length is chosen by a model of the launch: items go to the XCD's workgroup slots in rounds (equal items
  // finish together), a slot's waves share their SIMD's matrix pipe with the other resident workgroups, every item pays a
  // fixed prologue / fold / partial write and a pass of the reduce kernel.
  static const int tgt_env = getenv("A3D_WGRAD_SEG") ? atoi(getenv("A3D_WGRAD_SEG")) : 0;     // experiments: a fixed segment length
  long long total = 0;
  int cmax = 0;
  for (int k = 0; k < 27; ++k) {
    p.cnt[k] = k < K ? (counts ? counts[k] : ngroups) : 0;
    total += p.cnt[k];
    if (p.cnt[k] > cmax) cmax = p.cnt[k];
  }
  if (cmax < 1) cmax = 1;
  const int regs = p.cx * p.cy * 4 + 4 * (p.cx + p.cy) + 24;                  // accumulators + four row buffers + the rest
  const int occ = regs > 170 ? 2 : regs > 128 ? 3 : regs > 102 ? 4 : regs > 85 ? 5 : 6;   // workgroups per CU (512 registers per lane and SIMD)
  const int slots_xcd = 32 * occ;
  const double step_clk = 4.0 * p.cx * p.cy * 32.0;                            // one group through one wave's MFMAs
  const double fixed_clk = 9000.0 + 40.0 * p.cx * p.cy;                        // prologue (three dependent fetches) + fold + partial
  double best_cost = 0;
  int best_seg = 0;
  const int kMinSeg = 4, kMaxSlots = 2048;
  for (int seg = kMinSeg; ; seg = seg + (seg + 7) / 8) {
    if (seg > cmax) seg = cmax;
    if (tgt_env > 0) seg = tgt_env < cmax ? tgt_env : cmax;
    const int most = wgrad_cut(p.cnt, K, seg, p);
    if (p.slots <= kMaxSlots || seg == cmax || tgt_env > 0) {
      const long long wgs = (long long)most * p.nblocks;                       // workgroups of the fullest XCD
      const int resident = wgs < slots_xcd ? (int)((wgs + 31) / 32) : occ;     // workgroups sharing a CU
      const double rounds = (double)((wgs + slots_xcd - 1) / slots_xcd);
      // one wave per SIMD keeps ~0.65 of the matrix pipe busy (it waits for its rows in between), r of them min(0.92, 0.65 r)
      const double pipe = 0.65 * resident < 0.92 ? 0.65 * resident : 0.92;
      const double item = ((seg + 3) / 4) * step_clk * resident / pipe + fixed_clk;
      // a launch + the partials written and read again (measured ~1 TB/s each way next to the input-gradient convs), in clocks
      const double reduce = 8000.0 + (double)p.slots * cin * cout * 4.0 / 200.0;
      const double cost = rounds * item + reduce;
      if (!best_seg || cost < best_cost) best_cost = cost, best_seg = seg;
    }
    if (seg >= cmax || tgt_env > 0) break;
  }
  wgrad_cut(p.cnt, K, best_seg, p);
  p.part_bytes = (size_t)p.slots * cin * cout * sizeof(float);
  return true;
}

// The plan of a (scene map, channel pair) is made once: the search over segment lengths costs the host 13-23 us, an
// iteration asks ~110 times (each entry point twice: workspace size, then the launch).  serial = 0: a 1x1 map without a scene.
struct PlanKey {
  uint64_t serial;
  int kind_level, ngroups, K, cin, cout;
  bool operator==(const PlanKey& o) const {
    return serial == o.serial && kind_level == o.kind_level && ngroups == o.ngroups && K == o.K && cin == o.cin && cout == o.cout;
  }
};
struct PlanKeyHash {
  size_t operator()(const PlanKey& k) const {
    uint64_t h = k.serial * 0x9E3779B97F4A7C15ull;
    for (int v : {k.kind_level, k.ngroups, k.K, k.cin, k.cout}) h = (h ^ (uint64_t)(uint32_t)v) * 0x100000001B3ull;
    return (size_t)h;
  }
};
static bool wgrad_plan_cached(uint64_t serial, int kind_level, int ngroups, int K, const int* counts, int cin, int cout, WgradPlan& p) {
  static std::mutex mu;
  static std::unordered_map<PlanKey, WgradPlan, PlanKeyHash> cache;
  const PlanKey key{serial, kind_level, ngroups, K, cin, cout};
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) {
      p = it->second;
      return true;
    }
  }
  if (!wgrad_plan(ngroups, K, counts, cin, cout, p)) return false;
  std::lock_guard<std::mutex> lock(mu);
  if (cache.size() > 2048) cache.clear();      // scenes come and go (one per training iteration): the plans of dead ones are never asked for again
  cache.emplace(key, p);
  return true;
}

static void wgrad_fill(const WgradPlan& p, WgradArgs& a) {
  a.seg_len = p.seg_len;
  memcpy(a.cnt, p.cnt, sizeof(a.cnt));
  memcpy(a.base, p.base, sizeof(a.base));
  memcpy(a.xbase, p.xbase, sizeof(a.xbase));
  memcpy(a.seg_lo, p.seg_lo, sizeof(a.seg_lo));
}
static void wgrad_reduce_launch(const float* part, const WgradPlan& p, int K, int cin, int cout, float* dw, hipStream_t st) {
  SlotBase sb;
  memcpy(sb.base, p.base, sizeof(sb.base));
  const unsigned units = (unsigned)(K * p.nblocks * p.cx * (p.cy > 4 ? 2 : 1));
  k_wgrad_reduce<<<units, 64 * NW, 0, st>>>(part, sb, cin, cout, p.cx, p.cy, dw);
}

// a buffer descriptor spans < 4 GB and the "no pair" offset must lie past the end with room for a row's column offset
static bool wgrad_fits(const WgradArgs& a) {
  return (size_t)a.n_in * a.ldx * 4 <= (size_t)kNoRow && (size_t)a.n_out * a.ldy * 4 <= (size_t)kNoRow && a.cin <= 1024 && a.cout <= 1024;
}

template <int CX, int CY>
static void wgrad_launch_one(const WgradArgs& a, dim3 grid, bool bias, bool wide, hipStream_t st) {
  const size_t lds = (size_t)16 * CX * 16 * CY * sizeof(float);
#define A3D_WG1(B_, W_)                                       \
  {                                                           \
    A3D_ALLOW_LDS(64 * 1024, (k_wgrad<CX, CY, B_, W_>));      \
    k_wgrad<CX, CY, B_, W_><<<grid, 256, lds, st>>>(a);       \
  }
  if (bias) {
    if (wide) A3D_WG1(true, true) else A3D_WG1(true, false)
  } else {
    if (wide) A3D_WG1(false, true) else A3D_WG1(false, false)
  }
#undef A3D_WG1
}

// the kernel of the plan's channels-per-lane pair; false: there is none
static bool wgrad_launch(const WgradArgs& a, const WgradPlan& p, dim3 grid, bool bias, hipStream_t st) {
  static const bool force_wide = getenv("A3D_WGRAD_WIDE") && atoi(getenv("A3D_WGRAD_WIDE")) != 0;   // tests: the pointer build on small operands
  const bool wide = force_wide || !wgrad_fits(a);
#define A3D_WG(CX_, CY_)                                     \
  if (p.cx == CX_ && p.cy == CY_) {                          \
    wgrad_launch_one<CX_, CY_>(a, grid, bias, wide, st);     \
    return true;                                             \
  }
  A3D_WG(2, 2) A3D_WG(2, 4) A3D_WG(2, 6) A3D_WG(2, 8) A3D_WG(4, 2) A3D_WG(4, 4) A3D_WG(4, 6) A3D_WG(4, 8)
  A3D_WG(6, 2) A3D_WG(6, 4) A3D_WG(6, 6) A3D_WG(8, 2) A3D_WG(8, 4)
#undef A3D_WG
  return false;
}

// ---- the read-back of the work lists' counts: pinned buffer + event, pooled over the process (hipHostMalloc / hipEventCreate
// cost more than the read-back they serve)
constexpr int kListJobs = 5 * 27 + 4 * 8 + 4 * 8;
struct ListReadback {
  int* counts = nullptr;      // pinned [kListJobs]
  hipEvent_t done = nullptr;
};
static std::mutex g_rb_mu;
static std::vector<ListReadback*> g_rb_free;
static ListReadback* readback_get() {
  {
    std::lock_guard<std::mutex> lock(g_rb_mu);
    if (!g_rb_free.empty()) {
      ListReadback* r = g_rb_free.back();
      g_rb_free.pop_back();
      return r;
    }
  }
  ListReadback* r = new ListReadback();
  if (hipHostMalloc((void**)&r->counts, sizeof(int) * kListJobs, hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&r->done, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    if (r->counts) (void)hipHostFree(r->counts);
    delete r;
    return nullptr;
  }
  return r;
}
static void readback_put(ListReadback* r) {
  std::lock_guard<std::mutex> lock(g_rb_mu);
  g_rb_free.push_back(r);
}
// waits for a pending read-back (if any) and files the counts in the scene
static int wgrad_lists_finish(a3d_scene* s) {
  static std::mutex mu;                  // two host threads asking for the same scene's first weight gradient
  std::lock_guard<std::mutex> lock(mu);
  ListReadback* r = (ListReadback*)s->wg_pending;
  if (!r) return A3D_OK;
  A3D_HIP_CHECK(hipEventSynchronize(r->done));
  int j = 0;
  for (int kind = 0; kind < 3; ++kind)
    for (int l = 0; l < A3D_NUM_LEVELS; ++l) {
      if (kind > 0 && l >= A3D_NUM_LEVELS - 1) continue;
      const int K = kind == 0 ? 27 : 8;
      for (int k = 0; k < K; ++k) s->wg_count[kind][l][k] = r->counts[j++];
    }
  s->wg_pending = nullptr;
  s->wg_ready = true;
  readback_put(r);
  return A3D_OK;
}
void wgrad_scene_release(a3d_scene* s) {
  ListReadback* r = (ListReadback*)s->wg_pending;
  if (!r) return;
  (void)hipEventSynchronize(r->done);    // the copy into the pinned buffer must not land in the next owner's hands
  s->wg_pending = nullptr;
  readback_put(r);
}

static int wgrad_tables(const a3d_scene* s, int kind, int level_in, WgradArgs& a, const int** counts, int* ngroups) {
  if (!s || level_in < 0 || level_in >= A3D_NUM_LEVELS) {
    set_error("a3d_conv_wgrad: bad scene / level");
    return A3D_ERR_INVALID;
  }
  a.tab = nullptr, a.out_map = nullptr, a.list = nullptr, a.list_stride = 0;
  int lk = -1, ll = 0;       // list kind / owning level
  switch (kind) {
    case A3D_OP_CONV3:
      a.K = 27, a.n_in = a.n_out = a.n_pos = s->lv[level_in].n;
      a.tab = s->lv[level_in].nbr27, a.tab_stride = s->lv[level_in].npad;
      lk = 0, ll = level_in;
      break;
    case A3D_OP_DOWN:
      if (level_in >= A3D_NUM_LEVELS - 1) goto bad;
      a.K = 8, a.n_in = s->lv[level_in].n, a.n_out = a.n_pos = s->lv[level_in + 1].n;
      a.tab = s->lv[level_in].child8, a.tab_stride = s->lv[level_in + 1].npad;
      lk = 1, ll = level_in;
      break;
    case A3D_OP_UP:
      if (level_in < 1) goto bad;
      a.K = 8, a.n_in = s->lv[level_in].n, a.n_out = a.n_pos = s->lv[level_in - 1].n;
      a.tab = s->lv[level_in - 1].up8, a.tab_stride = s->lv[level_in - 1].npad;
      a.out_map = s->lv[level_in - 1].up_rows;
      lk = 2, ll = level_in - 1;
      break;
    case A3D_OP_LINEAR:
      a.K = 1, a.n_in = a.n_out = a.n_pos = s->lv[level_in].n;
      break;
    default:
    bad:
      set_error("a3d_conv_wgrad: kind %d does not exist at level %d", kind, level_in);
      return A3D_ERR_INVALID;
  }
  *ngroups = (a.n_pos + 15) / 16;
  *counts = nullptr;
  if (lk >= 0) {
    if (s->wg_pending) {                 // the lists were requested; their counts arrive now at the latest
      const int rc = wgrad_lists_finish(const_cast<a3d_scene*>(s));
      if (rc) return rc;
    }
    if (!s->wg_ready) {
      set_error("a3d_conv_wgrad: the scene has no weight-gradient work lists (call a3d_scene_build_wgrad_lists once per scene)");
      return A3D_ERR_INVALID;
    }
    a.list = s->wg_list[lk][ll], a.list_stride = s->wg_stride[lk][ll];
    *counts = s->wg_count[lk][ll];
  }
  return A3D_OK;
}

// ---- the work lists: for every (map kind, level, offset k) the 16-position groups that have offset k, ascending.  One
// workgroup per (kind, level, k) walks the level's group masks 1024 at a time (ballot + prefix inside the wave, the 16
// waves' totals through LDS) and appends; the counts go to the host once per scene (the launch plan needs them).
constexpr int kListMaps = 5 + 4 + 4;          // 3^3 maps of five levels, stride-2 and transposed maps of four
struct ListSet {
  const uint32_t* gmask[kListMaps];
  int* list[kListMaps];                        // [K][stride]
  int ngroups[kListMaps], stride[kListMaps];
  int first_job[kListMaps + 1];                // workgroup (job) b belongs to map m with first_job[m] <= b < first_job[m + 1]: offset k = b - first_job[m]
  int* counts;                                 // [jobs]
};
struct ListJob {
  const uint32_t* gmask;
  int ngroups, k;
  int* list;             // this offset's list
  int* count;            // one int
};
__global__ void __launch_bounds__(1024) k_wgrad_lists(const ListSet S) {
  __shared__ int wsum[16];
  __shared__ int carry;
  int m = 0;
  while (m + 1 < kListMaps && (int)blockIdx.x >= S.first_job[m + 1]) ++m;
  ListJob jb;
  jb.k = blockIdx.x - S.first_job[m];
  jb.gmask = S.gmask[m], jb.ngroups = S.ngroups[m];
  jb.list = S.list[m] + (size_t)jb.k * S.stride[m];
  jb.count = S.counts + blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int g0 = 0; g0 < jb.ngroups; g0 += 1024) {
    const int g = g0 + threadIdx.x;
    const bool has = g < jb.ngroups && ((jb.gmask[g] >> jb.k) & 1u);
    const unsigned long long bal = __ballot(has);
    const int below = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (has) jb.list[before + below] = g;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wsum[w];
      carry += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *jb.count = carry;
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
  if (n <= 0 || n > (int64_t)1 << 30 || !wgrad_plan_cached(0, 0, (int)((n + 15) / 16), 1, nullptr, cin, cout, p)) {
    set_error("a3d_linear_wgrad: channels must be multiples of 32 (got %d -> %d)", cin, cout);
    return 0;
  }
  return p.part_bytes + 256;
}
extern "C" int a3d_linear_wgrad(const float* x_dev, int ldx, const float* dy_dev, int ldy, int64_t n, int cin, int cout,
                                float* dw_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  WgradPlan p;
  if (!x_dev || !dy_dev || !dw_dev || !workspace_dev || n <= 0 || n > (int64_t)1 << 30 || ldx < cin || ldy < cout ||
      (ldx & 1) || (ldy & 1) || !wgrad_plan_cached(0, 0, (int)((n + 15) / 16), 1, nullptr, cin, cout, p)) {
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
  wgrad_fill(p, a);
  a.part = (float*)workspace_dev;
  const dim3 grid((unsigned)(p.grid_items * 8), 1, p.nblocks);
  if (!wgrad_launch(a, p, grid, false, st)) {
    set_error("a3d_linear_wgrad: no kernel for %d x %d channels per lane", p.cx, p.cy);
    return A3D_ERR_UNSUPPORTED;
  }
  A3D_LAUNCH_CHECK();
  wgrad_reduce_launch(a.part, p, 1, cin, cout, dw_dev, st);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

// the same gradient written where the caller keeps it: dW as [cin][cout] or [cout][cin] (transposed = 1, nn.Linear.weight's
// layout; a row slice of a packed in_proj matrix is such a block at an offset) with leading dimension ld_dw, assigned or
// added (accumulate); db_dev != nullptr: the bias gradient (column sums of dy) from the SAME pass over dy -- what took a
// transposing copy, a zero-filled full-size matrix + slice copy + add per in_proj slice and two column-sum launches per layer
extern "C" size_t a3d_linear_wgrad_into_workspace_bytes(int64_t n, int cin, int cout) {
  WgradPlan p;
  if (n <= 0 || n > (int64_t)1 << 30 || !wgrad_plan_cached(0, 0, (int)((n + 15) / 16), 1, nullptr, cin, cout, p)) {
    set_error("a3d_linear_wgrad_into: channels must be multiples of 32 (got %d -> %d)", cin, cout);
    return 0;
  }
  return align256(p.part_bytes) + align256((size_t)p.slots * cout * sizeof(float)) + 256;
}
extern "C" int a3d_linear_wgrad_into(const float* x_dev, int ldx, const float* dy_dev, int ldy, int64_t n, int cin, int cout,
                                     float* dw_dev, int ld_dw, int transposed, int accumulate, float* db_dev, int db_accumulate,
                                     void* workspace_dev, size_t workspace_bytes, void* stream) {
  WgradPlan p;
  if (!x_dev || !dy_dev || !dw_dev || !workspace_dev || n <= 0 || n > (int64_t)1 << 30 || ldx < cin || ldy < cout ||
      (ldx & 1) || (ldy & 1) || ld_dw < (transposed ? cin : cout) || !wgrad_plan_cached(0, 0, (int)((n + 15) / 16), 1, nullptr, cin, cout, p)) {
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
  wgrad_fill(p, a);
  a.part = (float*)workspace_dev;
  a.bias_part = (float*)((char*)workspace_dev + align256(p.part_bytes));
  const dim3 grid((unsigned)(p.grid_items * 8), 1, p.nblocks);
  if (!wgrad_launch(a, p, grid, db_dev != nullptr, st)) {
    set_error("a3d_linear_wgrad_into: no kernel for %d x %d channels per lane", p.cx, p.cy);
    return A3D_ERR_UNSUPPORTED;
  }
  A3D_LAUNCH_CHECK();
  const int all = cin * cout + (db_dev ? cout : 0);
  k_wgrad_reduce_into<<<(unsigned)((all + 63) / 64), 64 * kRedLanes, 0, st>>>(a.part, a.bias_part, p.slots, cin, cout, p.cx, p.cy, dw_dev,
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
  const int* counts;
  int ngroups;
  if (wgrad_tables(s, kind, level_in, a, &counts, &ngroups) != A3D_OK) return 0;
  WgradPlan p;
  if (!wgrad_plan_cached(s->serial, kind * 16 + level_in, ngroups, a.K, counts, cin, cout, p)) {
    set_error("a3d_conv_wgrad: channels must be multiples of 32 (got %d -> %d)", cin, cout);
    return 0;
  }
  return p.part_bytes + 256;
}

extern "C" int a3d_conv_wgrad(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx,
                              const float* dy_dev, int ldy, int cin, int cout, float* dw_dev, void* workspace_dev,
                              size_t workspace_bytes, void* stream) {
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  const int* counts;
  int ngroups;
  int rc = wgrad_tables(s, kind, level_in, a, &counts, &ngroups);
  if (rc) return rc;
  WgradPlan p;
  if (!x_dev || !dy_dev || !dw_dev || !workspace_dev || ldx < cin || ldy < cout || (ldx & 1) || (ldy & 1) ||
      !wgrad_plan_cached(s->serial, kind * 16 + level_in, ngroups, a.K, counts, cin, cout, p)) {
    set_error("a3d_conv_wgrad: bad arguments (channels multiples of 32, even leading dimensions)");
    return A3D_ERR_INVALID;
  }
  if (workspace_bytes < p.part_bytes || ((uintptr_t)workspace_dev & 15)) {
    set_error("a3d_conv_wgrad: workspace too small or misaligned (%zu < %zu)", workspace_bytes, p.part_bytes);
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  a.x = x_dev, a.dy = dy_dev, a.ldx = ldx, a.ldy = ldy, a.cin = cin, a.cout = cout;
  wgrad_fill(p, a);
  a.part = (float*)workspace_dev;
  const dim3 grid((unsigned)(p.grid_items * 8), 1, p.nblocks);
  if (!wgrad_launch(a, p, grid, false, st)) {
    set_error("a3d_conv_wgrad: no kernel for %d x %d channels per lane", p.cx, p.cy);
    return A3D_ERR_UNSUPPORTED;
  }
  A3D_LAUNCH_CHECK();
  wgrad_reduce_launch(a.part, p, a.K, cin, cout, dw_dev, st);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

// ---- the scene's weight-gradient work lists (once per scene, before the first a3d_conv_wgrad on it) ----------------------
// the maps in (kind, level) order; lists packed behind one another with a stride of the level's group count
static int wgrad_list_set(const a3d_scene* s, int* base_dev, ListSet* S, size_t* ints) {
  int m = 0, nj = 0;
  size_t off = 0;
  for (int kind = 0; kind < 3; ++kind)
    for (int l = 0; l < A3D_NUM_LEVELS; ++l) {
      if (kind > 0 && l >= A3D_NUM_LEVELS - 1) continue;
      const Level& lv = s->lv[l];
      const int K = kind == 0 ? 27 : 8;
      const int npos = kind == 1 ? s->lv[l + 1].n : lv.n;
      const int ng = (npos + 15) / 16;
      const int stride = (ng + 63) / 64 * 64;
      if (S) {
        S->gmask[m] = kind == 0 ? lv.gmask27 : kind == 1 ? lv.gmask_down : lv.gmask_up;
        S->list[m] = base_dev + off;
        S->ngroups[m] = ng, S->stride[m] = stride, S->first_job[m] = nj;
      }
      off += (size_t)K * stride;
      nj += K, ++m;
    }
  if (S) S->first_job[m] = nj;
  *ints = off;
  return nj;
}
extern "C" size_t a3d_scene_wgrad_lists_bytes(const a3d_scene* s) {
  if (!s) return 0;
  size_t ints;
  wgrad_list_set(s, nullptr, nullptr, &ints);
  return align256(ints * sizeof(int)) + align256(kListJobs * sizeof(int)) + 256;
}

extern "C" int a3d_scene_build_wgrad_lists(a3d_scene* s, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!s || !workspace_dev || ((uintptr_t)workspace_dev & 15) || workspace_bytes < a3d_scene_wgrad_lists_bytes(s) - 256) {
    set_error("a3d_scene_build_wgrad_lists: bad scene or workspace (a3d_scene_wgrad_lists_bytes)");
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  size_t ints;
  ListSet S;
  int* base = (int*)workspace_dev;
  const int nj = wgrad_list_set(s, base, &S, &ints);
  S.counts = (int*)((char*)workspace_dev + align256(ints * sizeof(int)));
  if (s->wg_pending) wgrad_scene_release(s);      // a second request: the first one's read-back is dropped
  s->wg_ready = false;
  k_wgrad_lists<<<nj, 1024, 0, st>>>(S);
  A3D_LAUNCH_CHECK();
  ListReadback* r = readback_get();
  if (!r) {
    set_error("a3d_scene_build_wgrad_lists: no pinned buffer / event for the read-back");
    return A3D_ERR_HIP;
  }
  // no host round trip here (round 6, second half: the synchronisation stood between the scene build and the first conv
  // of the training forward -- 0.4 ms of an idle device per iteration): the counts go to a pinned buffer behind an event
  // that the first a3d_conv_wgrad / a3d_conv_wgrad_workspace_bytes on this scene waits for
  if (hipMemcpyAsync(r->counts, S.counts, sizeof(int) * nj, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipEventRecord(r->done, st) != hipSuccess) {
    set_error("a3d_scene_build_wgrad_lists: %s", hipGetErrorString(hipGetLastError()));
    readback_put(r);
    return A3D_ERR_HIP;
  }
  int m = 0;
  for (int kind = 0; kind < 3; ++kind)
    for (int l = 0; l < A3D_NUM_LEVELS; ++l) {
      if (kind > 0 && l >= A3D_NUM_LEVELS - 1) continue;
      s->wg_list[kind][l] = S.list[m];
      s->wg_stride[kind][l] = S.stride[m];
      ++m;
    }
  s->wg_pending = r;
  return A3D_OK;
}
