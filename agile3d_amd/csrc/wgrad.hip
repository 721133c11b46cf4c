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

namespace a3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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
  } else if constexpr (C == 6) {
    const f32x2 a = *(const f32x2*)p, b = *(const f32x2*)(p + 2), c = *(const f32x2*)(p + 4);
    v[0] = a[0], v[1] = a[1], v[2] = b[0], v[3] = b[1], v[4] = c[0], v[5] = c[1];
  } else {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i], v[4 + i] = b[i];
  }
}

template <int CX, int CY>
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

  for (int grp = g_begin + wave; grp < g_end; grp += 4) {
    if (a.gmask && !((a.gmask[grp] >> k) & 1u)) continue;        // no position of this group has offset k
    // the group's 16 (input row, output row) pairs: lanes 0..15 of every 16-lane row hold them
    const int p16 = grp * 16 + j;
    int xi = a.n_in, yi = a.n_out;
    if (p16 < a.n_pos) {
      xi = tab ? tab[p16] : p16;
      yi = a.out_map ? a.out_map[p16] : p16;
    }
    float xv[CX], yv[CY], xn[CX], yn[CY];
    {
      const int xr = __shfl(xi, g, 16), yr = __shfl(yi, g, 16);
      const bool ok = xr < a.n_in && yr < a.n_out;
      load_c<CX>(a.x + (size_t)(ok ? xr : 0) * a.ldx + ci0, ok, xn);
      load_c<CY>(a.dy + (size_t)(ok ? yr : 0) * a.ldy + co0, ok, yn);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < CX; ++i) xv[i] = xn[i];
#pragma unroll
      for (int i = 0; i < CY; ++i) yv[i] = yn[i];
      if (s < 3) {   // next four positions in flight behind this step's MFMAs
        const int xr = __shfl(xi, 4 * (s + 1) + g, 16), yr = __shfl(yi, 4 * (s + 1) + g, 16);
        const bool ok = xr < a.n_in && yr < a.n_out;
        load_c<CX>(a.x + (size_t)(ok ? xr : 0) * a.ldx + ci0, ok, xn);
        load_c<CY>(a.dy + (size_t)(ok ? yr : 0) * a.ldy + co0, ok, yn);
      }
#pragma unroll
      for (int tx = 0; tx < CX; ++tx)
#pragma unroll
        for (int ty = 0; ty < CY; ++ty)
          acc[tx][ty] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[tx], yv[ty], acc[tx][ty], 0, 0, 0);
    }
  }
  // fold the four waves in wave order; lane (g, j) holds dW[ci = CX (4g + r) + tx][co = CY j + ty] in acc[tx][ty][r]
  constexpr int BW = 16 * CY;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int tx = 0; tx < CX; ++tx)
#pragma unroll
        for (int ty = 0; ty < CY; ++ty)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* f = fold + (CX * (4 * g + r) + tx) * BW + CY * j + ty;
            *f = w == 0 ? acc[tx][ty][r] : *f + acc[tx][ty][r];
          }
    }
    __syncthreads();
  }
  float* P = a.part + (((size_t)chunk * a.K + k) * a.cin + (size_t)bx * 16 * CX) * a.cout + (size_t)by * BW;
  for (int e = threadIdx.x; e < 16 * CX * BW; e += 256) P[(size_t)(e / BW) * a.cout + e % BW] = fold[e];
}

// dw[e] = sum over chunks of part[chunk][e], chunk order
__global__ void k_wgrad_reduce(const float* __restrict__ part, int nchunk, size_t total, float* __restrict__ dw) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += part[(size_t)c * total + e];
  dw[e] = s;
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
  int chunks = (1536 + K * p.nblocks - 1) / (K * p.nblocks);      // ~6 workgroups per CU over the whole launch
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

}  // namespace a3d

using namespace a3d;

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
    (void)hipFuncSetAttribute((const void*)k_wgrad<CX_, CY_>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
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
  k_wgrad_reduce<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a.part, p.chunks, total, dw_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}
