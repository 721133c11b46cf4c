// Sparse convolution as an output-stationary implicit GEMM on the fp32 matrix cores.
//
// Replaces ME.MinkowskiConvolution / MinkowskiConvolutionTranspose (+ the MinkowskiBatchNorm,
// MinkowskiReLU, residual add and me.cat that follow them) at the reference's call sites:
// models/modules/common.py:137-155,170-188; models/modules/resnet_block.py:48-64;
// models/res16unet.py:222-295; models/agile3d.py:43-45,179.
//
//   out[u, :] = act( (sum_k in[nbr[k][u], :] @ W[k]) * scale + shift + res[u, :] )
//
// k_conv_sk covers the 3^3, 2^3-stride-2, transposed 2^3 and small 1x1 layers: they only differ in the
// neighbour table.  k_dense is the gather-free GEMM for the N-point linears of the decoder and the large 1x1
// convolutions.  k_stem is the 5^3 input convolution (Cin = 3).
//
// gfx950 mapping:
//   * workgroup = 4 waves = 64 (or 128) output rows x BN output channels; a wave owns one (or two) 16-row MFMA
//     groups; v_mfma_f32_16x16x4_f32 (exact fp32) accumulates 16x16 tiles in registers;
//   * stage = (kernel offset k, up to 96 input channels): every lane loads its own A fragments of the NEXT stage
//     straight from global memory (gathered row, 16 bytes per 16-channel step) while the current stage is
//     multiplied; the packed weight slice goes global -> LDS by LDS-DMA (global_load_lds_dwordx4, inline asm)
//     into a two-slot ring;
//   * offsets k absent from a whole 16-row group are skipped (no gather, no MFMA): rows were sorted by
//     neighbour pattern when the scene was built (scene.hip);
//   * a layer's (tile, offset, channel chunk) stages are cut into equal shares, one per workgroup; a tile cut by a
//     share boundary is finished inside the kernel (hand-off of partial accumulators, fixed summation order);
//   * the K dimension inside a 16-channel step is permuted (channel = 16S + 4g + t for lane group g, MFMA t)
//     identically for both operands, so one 16-byte load / ds_read_b128 feeds four MFMAs;
//   * weights are the MFMA A operand: a lane ends up with 4 consecutive output channels of one row, so
//     BatchNorm(eval) scale/shift, residual, ReLU and the channel-slice concat are 16-byte epilogue accesses.
#include "common.h"
#include <stdlib.h>

namespace a3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* in;
  int ldi, n_in;
  const int* nbr;
  int nbr_stride;
  const uint32_t* gmask;
  const float* w;
  int K, cin, cout;
  float* out;
  int ldo, n_out;
  const int* out_map;
  const float* scale;
  const float* shift;
  const float* res;
  int ldr;
  int relu;
  int zero_row;
  int tag_table, tag_level;   // profiling only
  int n_tiles;
  // fused residual projection (3^3 only): out += in2 W2, W2 = the 1x1 weight packed right behind the 27 offsets'
  // (as "offset 27" with cin2 / 16 channel steps); in2 rows = the output rows; cin2 = 0: none
  const float* in2;
  int ldi2, cin2;
  // fused 1x1 head on the finished output rows (HEAD builds): head_out[head_map[row]][0 .. head_cout) = out_row @ head_w + bias
  const float* head_w;
  const float* head_bias;
  float* head_out;
  const int* head_map;
  int head_ld, head_cout;
};

// ------------------------------------------------------------------------------ k_conv_sk
// The convolution kernel:
//   * a STATIC equal-share partition of the layer's work ("stream-K"): the (cout block, 64-row tile, offset, channel
//     chunk) stages of a layer form one line; workgroup w of G owns the stretch [w Tot / G, (w+1) Tot / G) of it
//     (cost of a tile = its number of stages + `ov` units for its prologue/epilogue, prefix sums from the scene's
//     pre27 / pre_down / pre_up tables).  No tile queue, no last partly-filled round, no split-K launch geometry
//     and no separate reduction kernel: a tile cut by a share boundary is finished by the workgroup that holds its
//     LAST stages (the owner); the others hand it their partial accumulators through a slab in global memory
//     (write-through stores + one flag per workgroup, agent-scope acquire on the reader: cdna_hip_programming.md
//     Guideline 16).  Summation order is fixed (ascending stage ranges), so results are bit-identical run to run.
//     Deadlock freedom does not depend on dispatch order or residency: the logical index w is a ticket drawn at
//     start-up, a workgroup computes the part it must PUBLISH first and the part it must WAIT for last, and it
//     only ever waits for lower tickets -- whose holders have started and publish without waiting for anybody.
//   * a lean stage: gather rows are byte offsets (32-bit VGPR offset + scalar base: no 64-bit vector address
//     arithmetic), the A fragments alternate between two register sets (stage loop unrolled by two: no copies),
//     the weight DMA addresses are scalar base + constant per-lane offsets.  The VALU instructions of a stage that
//     are not MFMAs compete with the co-resident workgroup's MFMA stream for the SIMD's issue port.
struct SkArgs {
  ConvArgs c;
  const int* pre;        // [n_tiles + 1] (tile, offset) pairs before each tile; nullptr: K per tile
  int nchunk, ov;
  int n_cblk;            // cout / BN
  int G;                 // workgroups launched (shares; the kernel uses min(G, total cost))
  int* ticket;           // zeroed by the caller; nullptr: whole tiles only (no hand-offs), w = blockIdx.x
  unsigned* flags;       // [G] zeroed by the caller
  float* slab;           // [G][64 * BN] partial accumulators
  unsigned in_row_bytes; // ldi * 4
  int* fail;             // set when a bounded wait ran out (never in a healthy run)
  int prio;              // 1: static wave priority by occupancy layer (ticket / 256), see the kernel
  int nchunk2;           // FUSE: stages of the fused 1x1 projection per tile (cin2 / CH), run as offset 27 after the tile's own
  unsigned in2_row_bytes;
  // STATS (training-mode BatchNorm): per 64-row tile and output column the sum of the tile's rows and the sum of their
  // squared deviations from the TILE mean, [n_tiles][2][stats_ld] -- what k_bn_combine merges into the batch statistics
  float* stats;
  int stats_ld;
  // STATS == 2 (input-gradient conv whose output is the COMPLETE gradient dy of a BatchNorm(+ReLU) unit's output y): the
  // epilogue masks it (g = dy where y > 0), stores g, and writes per tile the two column sums the BatchNorm backward needs,
  // stats[tile][0][c] = sum g, stats[tile][1][c] = sum g * xhat with xhat = (raw - mean) rstd -- no pass over dy, y and raw
  // for the sums afterwards
  const float* bw_y;
  const float* bw_raw;
  const float* bw_mean;
  const float* bw_rstd;
  int bw_ldy, bw_ldraw, bw_relu;
};
struct BwArgs {   // the same for k_conv_wl (and the host-side plumbing)
  const float *y, *raw, *mean, *rstd;
  int ldy, ldraw, relu;
};

// sum over the 16 lanes that share lane >> 4 (the 16 rows of an MFMA group), result in every lane; fixed order
// (DPP row operations: four VALU adds with a lane permutation as operand modifier.  The first version used __shfl_xor =
// ds_bpermute_b32: 192 of them per tile epilogue, four dependent LDS round trips per value -- the statistics epilogue cost
// 9-13 % of the conv's time, profiles/r05_experiments.txt)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2, 3, 0, 1]: every lane holds its quad's sum
  v += dpp_mov<0x141>(v);   // row_half_mirror: + the other quad of the 8-lane half
  v += dpp_mov<0x140>(v);   // row_mirror: + the other half of the 16-lane row
  return v;
}

__device__ __forceinline__ void store_sc1(float* p, f32x4 v) {   // write-through (agent scope) 16-byte store
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// global -> LDS DMA with a scalar base and a 32-bit per-lane byte offset; M0 = LDS destination (saved and restored:
// scalar instructions only)
__device__ __forceinline__ void glds16_s(const float* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
      : "memory");
}

// s_waitcnt vmcnt(0) through the builtin: unlike an asm statement the compiler's wait-count pass models it, so it
// knows every load it tracks (the A fragments) has landed and inserts no conservative vmcnt of its own in front of
// the MFMAs -- which would also wait for the LDS-DMA pieces it cannot see.  simm16: vmcnt 0, expcnt 7, lgkmcnt 15.
__device__ __forceinline__ void wait_all_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// BN output columns, CH input channels per stage; PAIR: weight fragments two at a time (16 registers for the B operand
// instead of 8 BN / 16: the low-register build, 3-4 workgroups per CU)
// PAIR == 2 (opt-in, A3D_CONV_EMU=1, 96 columns x 32 channels only): fp32 products from SIX bf16 MFMAs -- both operands
// split into three bf16 planes (x = h + m + l by truncation, every remainder exact), h h + h m + m h + h l + m m + l h on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Error: tools/bf16x6_ubench.hip (5.1e-6 against the exact chain's 7.5e-6
// over 2592 products) and, bounded per shape class on adversarial inputs, tests/test_gpu_conv.py::
// test_emulated_fp32_products_error_bound (a-priori (2^-20 + 6 (K cin / 32) 2^-24) sum|x||w|; domain |x| >= 2^-100 or 0: the
// matrix cores drop subnormal bf16 planes); stage loop 1.84x faster; weights packed as three planes.
// FUSE: the block's residual projection (BasicBlock.downsample: 1x1 conv + BatchNorm on the block input,
// resnet_block.py:59-61) as one more "offset" of the block's second conv: every tile ends with cin2 / CH stages that
// gather the output rows themselves from the block input and multiply them with the 1x1 weight (both BatchNorm scales
// folded into the packed weights by the host, the two shifts added).  Same stage loop, same hand-offs; a separate
// instantiation, so the plain kernels are untouched.
// STATS: the owner's epilogue also writes the tile's BatchNorm partial statistics (training path; a separate instantiation,
// the inference kernels are untouched).
// HEAD: the owner's epilogue multiplies the finished rows (all BN = cout columns are in this workgroup's registers, in exactly
// the B-operand layout of the next MFMA: lane (g, j) holds channels 16 ct + 4 g + t of row j) with a packed 1x1 weight and
// writes the product to a second output -- lin_squeeze_head behind block8's last conv, without re-reading the rows.
template <int BN, int CH, int PAIR, bool FUSE = false, int STATS = 0, bool HEAD = false>
__global__ void __launch_bounds__(256, PAIR ? ((BN <= 96 && CH <= 32) ? 4 : 3) : ((BN <= 96 && CH <= 48) ? 3 : 2))
    k_conv_sk(const SkArgs a) {
  static_assert(!FUSE || PAIR != 2, "the fused projection runs on the exact-fp32 builds");
  constexpr int RG = 1;   // 16-row groups per wave (two per wave -- 128-row tiles -- was tried and did not pay)
  constexpr bool EMU = PAIR == 2;
  static_assert(!EMU || CH == 32, "one 16x16x32 block per stage and column tile");
  constexpr int NCT = BN / 16, NS = CH / 16, NW = 4, kTile = 64 * RG;
  constexpr int NPIECE = EMU ? 3 * NCT : NS * NCT;   // 1 KB pieces of a stage's weights: 3 bf16 planes per column tile / fp32
  constexpr int WV = (NPIECE + NW - 1) / NW;
  constexpr int WF = NPIECE * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wring = (float*)smem;                            // [2][WF]
  int* misc = (int*)(wring + 2 * WF);
  const unsigned ring_addr = (unsigned)(size_t)wring;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int K = a.c.K, nchunk = a.nchunk;
  const int nchunk2 = FUSE ? a.nchunk2 : 0;   // stages of the fused projection: the LAST stages of every tile
  const int ov = a.ov;
  const int T = a.c.n_tiles;
  const int cin16 = a.c.cin >> 4, cout16 = a.c.cout >> 4;

  int w = blockIdx.x;
  const long long pre_T = a.pre ? (long long)a.pre[T] : (long long)K * T;   // requested before the ticket's round trip
  if (a.ticket) {
    if (tid == 0) misc[0] = atomicAdd(a.ticket, 1);
    __syncthreads();
    w = __builtin_amdgcn_readfirstlane(misc[0]);
  }
  if (w == 0 && a.c.zero_row >= 0)
    for (int cidx = tid; cidx < a.c.cout; cidx += 256) a.c.out[(size_t)a.c.zero_row * a.c.ldo + cidx] = 0.f;
  // The four waves of a workgroup sit on the four SIMDs of a CU, each next to the waves of the 2-3 other resident
  // workgroups, and meet at a barrier every stage: a wave that loses the arbitration on ITS SIMD holds up its three
  // siblings (SQ counters: a third of a wave's life parked at stage ends).  A static priority by occupancy layer
  // (workgroups are dispatched CU after CU, so ticket / 256 numbers the workgroups of a CU) makes every SIMD prefer
  // the SAME workgroup: its waves reach the barrier together, the next layer's run behind them.
  if (a.prio) {
    switch ((w >> 8) & 3) {
      case 0: __builtin_amdgcn_s_setprio(3); break;
      case 1: __builtin_amdgcn_s_setprio(2); break;
      case 2: __builtin_amdgcn_s_setprio(1); break;
      default: __builtin_amdgcn_s_setprio(0); break;
    }
  }

  auto prefix = [&](int t) -> long long {   // cost units of one cout block before tile t
    const long long n = a.pre ? (long long)a.pre[t] : (long long)K * t;
    return n * nchunk + (long long)(ov + nchunk2) * t;
  };
  const long long tile_tot = pre_T * nchunk + (long long)(ov + nchunk2) * T;   // = prefix(T): one pass over all tiles
  const long long tot = (long long)a.n_cblk * tile_tot;
  const int G = (int)min((long long)a.G, tot);
  if (w >= G) return;
  auto share_begin = [&](int ww) -> long long { return tot * ww / G; };
  // 64-ary search, every wave on its own (no LDS): lane i probes the i-th of 64 evenly spaced tiles of the current
  // range, a ballot finds the last one at or below r; three rounds cover 2^18 tiles (a binary search is 13+ dependent
  // loads per workgroup before its first MFMA)
  // the two ends of a share are searched in lockstep (their probes are independent loads: one round trip per round)
  auto locate2 = [&](long long x0, long long x1, int& u0, int& u1) {
    const int cb0 = (int)(x0 / tile_tot), cb1 = (int)(x1 / tile_tot);
    const long long r0 = x0 - (long long)cb0 * tile_tot, r1 = x1 - (long long)cb1 * tile_tot;
    int lo0 = 0, hi0 = T, lo1 = 0, hi1 = T;   // prefix(lo) <= r < prefix(hi)
    while (hi0 - lo0 > 1 || hi1 - lo1 > 1) {
      const int step0 = (hi0 - lo0 + 63) >> 6, step1 = (hi1 - lo1 + 63) >> 6;
      const int tp0 = lo0 + lane * step0, tp1 = lo1 + lane * step1;   // lane 0 probes lo itself (always <= r)
      const long long p0 = prefix(min(tp0, T)), p1 = prefix(min(tp1, T));
      const unsigned long long m0 = __ballot(tp0 < hi0 && p0 <= r0), m1 = __ballot(tp1 < hi1 && p1 <= r1);
      const int n0 = lo0 + (63 - __builtin_clzll(m0)) * step0, n1 = lo1 + (63 - __builtin_clzll(m1)) * step1;
      hi0 = min(hi0, n0 + step0);
      lo0 = n0;
      hi1 = min(hi1, n1 + step1);
      lo1 = n1;
    }
    u0 = cb0 * T + __builtin_amdgcn_readfirstlane(lo0);
    u1 = cb1 * T + __builtin_amdgcn_readfirstlane(lo1);
  };

  long long lo, hi;
  int u_lo, u_hi;
  if (a.ticket) {
    lo = share_begin(w);
    hi = share_begin(w + 1);
    if (hi <= lo) return;
    locate2(lo, hi - 1, u_lo, u_hi);
  } else {   // whole tiles: no tile is shared (w < a.G <= number of tiles: the launch guarantees it)
    const long long NU = (long long)a.n_cblk * T;
    u_lo = (int)(NU * w / a.G);
    u_hi = (int)(NU * (w + 1) / a.G) - 1;
    if (u_hi < u_lo) return;
    lo = 0;
    hi = tot;
  }

  // this wave's weight pieces q = wave + NW*i of a stage: constant per-lane source byte offset, LDS byte offset
  unsigned wsrc[WV], wdst[WV];
#pragma unroll
  for (int i = 0; i < WV; ++i) {
    const int q = wave + NW * i;
    wsrc[i] = EMU ? (unsigned)(q * 1024 + lane * 16)
                  : (unsigned)(((q / NCT) * cout16 + (q % NCT)) * 1024 + lane * 16);
    wdst[i] = (unsigned)q * 1024u;
  }
  const unsigned lane_a_off = EMU ? 32u * g : 16u * g;   // channels 4g..4g+3 of a 16-channel step (EMU: 8g..8g+7 of the 32)
  const int wrow = 16 * RG * wave + j;   // this lane's row inside the tile for its group 0 (group r: + 16 r)

  const int n_u = u_hi - u_lo + 1;
  for (int it = 0; it < n_u; ++it) {
    // order: the tile at the END of the share first (its part is published), the tile at the START last (it may wait)
    const int u = n_u == 1 ? u_lo : (it == 0 ? u_hi : (it == n_u - 1 ? u_lo : u_lo + it));
    const int cb = u / T, t = u - cb * T;
    const int ct0 = cb * NCT;
    const int r0 = t * kTile;
    // gm: offsets this wave works on = those any of its RG groups has (with two groups per wave a group that lacks
    // one of them gathers the zero row for it: rows are sorted by neighbour pattern, adjacent groups rarely differ)
    uint32_t un = K >= 32 ? 0xffffffffu : (1u << K) - 1u, gm = un;
    if (a.c.gmask) {
      const uint32_t* gp = a.c.gmask + (r0 >> 4);
      uint32_t o = 0;
#pragma unroll
      for (int q = 0; q < 4 * RG; ++q) o |= gp[q];
      un &= o;
      gm = 0;
#pragma unroll
      for (int r = 0; r < RG; ++r) gm |= gp[RG * wave + r];
    }
    un = __builtin_amdgcn_readfirstlane(un);
    gm = __builtin_amdgcn_readfirstlane(gm);
    const int S_own = __builtin_popcount(un) * nchunk;   // stages of the tile's own offsets
    const int S = S_own + nchunk2;
    if constexpr (FUSE) {   // "offset 27": present in every tile and every group (K = 27: bit 27 is free)
      un |= 1u << 27;
      gm |= 1u << 27;
    }
    int s0 = 0, s1 = S;
    bool owner = true;
    long long Cu = 0;
    if (a.ticket) {
      Cu = (long long)cb * tile_tot + prefix(t);
      const long long Cu1 = (long long)cb * tile_tot + prefix(t + 1);
      if (lo > Cu) s0 = (int)min((long long)S, max(0LL, lo - Cu - ov));
      if (hi < Cu1) {
        s1 = (int)min((long long)S, max(0LL, hi - Cu - ov));
        owner = false;
      }
      if (!owner && s1 <= s0) continue;   // only overhead units of this tile fall into the share
    }

    f32x4 acc[RG][NCT];
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (s1 > s0) {
      auto next_k = [&](int k) -> int {   // next offset of the tile after k (32 = none)
        const uint32_t rest = k >= 31 ? 0u : un & ~((2u << k) - 1u);
        return rest ? __builtin_ctz(rest) : 32;
      };
      // first stage of the part: skip s0 / nchunk offsets
      int k = __builtin_ctz(un), c = s0 % nchunk;
      if (FUSE && s0 >= S_own) {
        k = 27;
        c = s0 - S_own;
      } else {
        for (int i = s0 / nchunk; i > 0; --i) k = next_k(k);
      }
      int rem = s1 - s0;
      const float* wbase = a.c.w + (size_t)ct0 * 256;
      const char* inb = (const char*)a.c.in;
      // gather rows of offset kk for this lane's groups, straight from the neighbour table (requested one offset
      // before the A loads that need them); a missing neighbour is the zero row
      auto fetch_rows = [&](int kk, int (&rows)[RG]) {
        const int kc = kk < K ? kk : 0;    // unconditional loads (no select behind them: the loop-carried register is
                                           // the load's destination, nothing waits for it before its use a stage later);
                                           // 32 = no further offset, 27 (FUSE) = the projection: no table row, entry unused
#pragma unroll
        for (int r = 0; r < RG; ++r)
          rows[r] = a.c.nbr ? a.c.nbr[(size_t)kc * a.c.nbr_stride + r0 + wrow + 16 * r] : min(r0 + wrow + 16 * r, a.c.n_in - 1);
      };
      // stage (k, c): this lane's A fragments (for its groups that have k), this wave's weight pieces by LDS-DMA
      auto load_stage = [&](f32x4 (&A)[RG][NS], int kk, int cc, int slot, const int (&rows)[RG]) {
        const bool proj = FUSE && kk == 27;   // wave-uniform: the projection's stages read the block input at the output rows
        const char* ar = (proj ? (const char*)a.c.in2 : inb) + (size_t)cc * (CH * 4);
        const unsigned row_bytes = proj ? a.in2_row_bytes : a.in_row_bytes;
        if ((gm >> kk) & 1u) {
#pragma unroll
          for (int r = 0; r < RG; ++r) {
            // (the select sits at the USE of the table entry requested a stage earlier, not behind its load)
            const int row = proj ? min(r0 + wrow + 16 * r, a.c.n_out) : rows[r];   // n_out = the zero row of in2
            const unsigned roff = (unsigned)row * row_bytes + lane_a_off;
#pragma unroll
            for (int Sx = 0; Sx < NS; ++Sx) A[r][Sx] = *(const f32x4*)(ar + roff + (EMU ? 16 : 64) * Sx);
          }
        }
        const float* wst = EMU ? a.c.w + (((size_t)kk * nchunk + cc) * cout16 + ct0) * (3 * 256)
                               : wbase + ((size_t)kk * cin16 + (size_t)cc * NS) * cout16 * 256;
        const unsigned dst = ring_addr + (unsigned)slot * (WF * 4u);
#pragma unroll
        for (int i = 0; i < WV; ++i)
          if (NPIECE % NW == 0 || wave + NW * i < NPIECE) glds16_s(wst, wsrc[i], dst + wdst[i]);
      };
      auto compute = [&](const f32x4 (&A)[RG][NS], int kk, int slot) {
        if (!((gm >> kk) & 1u)) return;
        const f32x4* Ws = (const f32x4*)(wring + slot * WF) + lane;
        if constexpr (EMU) {
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
          // this lane's 8 channels of its row -> three bf16 planes (truncation: v = h + r1, r1 = m + r2, exact in fp32)
          u32x4 x[3];
#pragma unroll
          for (int pq = 0; pq < 4; ++pq) {
            uint32_t hh[2], mm[2], ll[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const float v = A[0][pq >> 1][2 * (pq & 1) + e];
              const uint32_t hb = __float_as_uint(v) & 0xffff0000u;
              const float r1 = v - __uint_as_float(hb);
              const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
              const float r2 = r1 - __uint_as_float(mb);
              hh[e] = hb, mm[e] = mb, ll[e] = __float_as_uint(r2) & 0xffff0000u;
            }
            x[0][pq] = (hh[0] >> 16) | hh[1];
            x[1][pq] = (mm[0] >> 16) | mm[1];
            x[2][pq] = (ll[0] >> 16) | ll[1];
          }
          const bf16x8 xh = __builtin_bit_cast(bf16x8, x[0]), xm = __builtin_bit_cast(bf16x8, x[1]),
                       xl = __builtin_bit_cast(bf16x8, x[2]);
          const u32x4* Wq = (const u32x4*)Ws;
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, Wq[(3 * ct + 0) * 64]), wm = __builtin_bit_cast(bf16x8, Wq[(3 * ct + 1) * 64]),
                         wl = __builtin_bit_cast(bf16x8, Wq[(3 * ct + 2) * 64]);
            f32x4 c = acc[0][ct];   // smallest terms first
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
            acc[0][ct] = c;
          }
        } else if constexpr (PAIR == 0) {
          // the weight fragments of a whole 16-channel step in registers, the next step's read behind this step's MFMAs;
          // consecutive MFMAs go to different accumulators
          f32x4 b[2][NCT];
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) b[0][ct] = Ws[ct * 64];
#pragma unroll
          for (int Sx = 0; Sx < NS; ++Sx) {
            if (Sx + 1 < NS) {
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct) b[(Sx + 1) & 1][ct] = Ws[((Sx + 1) * NCT + ct) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct)
                acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[Sx & 1][ct][tt], A[0][Sx][tt], acc[0][ct], 0, 0, 0);
          }
        } else {
          // two weight fragments at a time: 8 MFMAs on two alternating accumulators behind which the next pair's
          // ds_read_b128s are in flight -- 16 registers of B operand; with 3-4 waves per SIMD the other waves fill the
          // issue slots a wave's own short dependence chains leave
          static_assert(NCT % 2 == 0, "pairs of 16-column tiles");
          constexpr int NP = NS * NCT / 2;
          f32x4 c0 = Ws[0], c1 = Ws[64];
#pragma unroll
          for (int pi = 0; pi < NP; ++pi) {
            const int Sx = pi / (NCT / 2), ct = 2 * (pi % (NCT / 2));
            f32x4 n0 = c0, n1 = c1;
            if (pi + 1 < NP) {
              n0 = Ws[(2 * pi + 2) * 64];
              n1 = Ws[(2 * pi + 3) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
              acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[tt], A[0][Sx][tt], acc[0][ct], 0, 0, 0);
              acc[0][ct + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[tt], A[0][Sx][tt], acc[0][ct + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            c0 = n0;
            c1 = n1;
          }
        }
      };
      f32x4 A0[RG][NS], A1[RG][NS];
      int rows_cur[RG], rows_nxt[RG];
      int k1 = next_k(k);
      fetch_rows(k, rows_cur);
      fetch_rows(k1, rows_nxt);
      load_stage(A0, k, c, 0, rows_cur);
      wait_all_vmem();
      __builtin_amdgcn_s_barrier();
      for (;;) {
        int k2, c2;
        // ---- even stage: multiply A0 / slot 0, fetch A1 / slot 1
        if (c + 1 < ((FUSE && k == 27) ? nchunk2 : nchunk)) { k2 = k; c2 = c + 1; } else { k2 = k1; c2 = 0; }
        if (rem > 1) {
          if (k2 != k) {
#pragma unroll
            for (int r = 0; r < RG; ++r) rows_cur[r] = rows_nxt[r];
            k1 = next_k(k2);
            fetch_rows(k1, rows_nxt);
          }
          load_stage(A1, k2, c2, 1, rows_cur);
        }
        compute(A0, k, 0);
        wait_all_vmem();
        __builtin_amdgcn_s_barrier();
        if (--rem == 0) break;
        k = k2; c = c2;
        // ---- odd stage: multiply A1 / slot 1, fetch A0 / slot 0
        if (c + 1 < ((FUSE && k == 27) ? nchunk2 : nchunk)) { k2 = k; c2 = c + 1; } else { k2 = k1; c2 = 0; }
        if (rem > 1) {
          if (k2 != k) {
#pragma unroll
            for (int r = 0; r < RG; ++r) rows_cur[r] = rows_nxt[r];
            k1 = next_k(k2);
            fetch_rows(k1, rows_nxt);
          }
          load_stage(A0, k2, c2, 0, rows_cur);
        }
        compute(A1, k, 1);
        wait_all_vmem();
        __builtin_amdgcn_s_barrier();
        if (--rem == 0) break;
        k = k2; c = c2;
      }
    }

    if (!owner) {
      // publish this part: write-through stores, every wave drains, one flag
      float* P = a.slab + (size_t)w * (kTile * BN) + (size_t)tid * 4;
#pragma unroll
      for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) store_sc1(P + (r * NCT + ct) * 1024, acc[r][ct]);
      wait_all_vmem();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.flags + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (s0 > 0) {
        // parts of the lower tickets, down to the workgroup that holds stage 0
        // (the largest w' with share_begin(w') <= Cu + ov: floor(tot w' / G) <= X  <=>  w' < (X + 1) G / tot)
        const int wf = (int)(((Cu + ov + 1) * G + tot - 1) / tot) - 1;
        // every flag is polled by its own thread; ONE acquire for the workgroup; then the parts are read with all
        // their loads in flight and added in a fixed order
        for (int wp = wf + tid; wp < w; wp += 256) {
          unsigned spins = 0;
          while (__hip_atomic_load(a.flags + wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) {   // seconds: never in a healthy run; do not hang the device
              if (a.fail) *a.fail = 1;
              break;
            }
          }
        }
        __syncthreads();
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        const float* P0 = a.slab + (size_t)tid * 4;
        // fixed order: own part, then the parts of the tickets below, descending; U parts' loads in flight at a time
        constexpr int U = NCT <= 4 ? 2 : 1;
        for (int wp = w - 1; wp >= wf; wp -= U) {
          f32x4 pa[U][NCT];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (wp - u >= wf) {
              const float* Pa = P0 + (size_t)(wp - u) * (kTile * BN);
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct) pa[u][ct] = *(const f32x4*)(Pa + ct * 1024);
            } else {
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct) pa[u][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[0][ct] += pa[u][ct];
        }
      }
      if constexpr (STATS == 1) {
        // BatchNorm statistics of the RAW output (before scale / shift / residual / ReLU, which the training path does
        // not pass): column sums over the tile's valid rows, then the squared deviations from the tile mean -- wave-level
        // shuffles over the 16 rows of a group, the four waves through LDS behind the weight ring, fixed orders
        float* sst = (float*)(misc + 16);   // [2][4][BN]
        const bool valid = r0 + wrow < a.c.n_out;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {   // one column tile at a time (sched_barrier: no interleaving, four live temporaries)
          f32x4 sv = valid ? acc[0][ct] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) sv[tt] = row16_sum(sv[tt]);
          if (j == 0) *(f32x4*)(sst + wave * BN + ct * 16 + 4 * g) = sv;
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        const int cnt = min(kTile, a.c.n_out - r0);
        const float inv = 1.f / (float)cnt;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const float* sp = sst + ct * 16 + 4 * g;
          const f32x4 tot = ((*(const f32x4*)sp + *(const f32x4*)(sp + BN)) + *(const f32x4*)(sp + 2 * BN)) + *(const f32x4*)(sp + 3 * BN);
          f32x4 d = valid ? acc[0][ct] - tot * inv : (f32x4){0.f, 0.f, 0.f, 0.f};
          d = d * d;
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) d[tt] = row16_sum(d[tt]);
          if (j == 0) *(f32x4*)(sst + 4 * BN + wave * BN + ct * 16 + 4 * g) = d;
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (tid < BN) {
          const float* s1p = sst + tid;
          const float* s2p = sst + 4 * BN + tid;
          float* P = a.stats + (size_t)t * 2 * a.stats_ld + cb * BN + tid;
          P[0] = ((s1p[0] + s1p[BN]) + s1p[2 * BN]) + s1p[3 * BN];
          P[a.stats_ld] = ((s2p[0] + s2p[BN]) + s2p[2 * BN]) + s2p[3 * BN];
        }
      }
      if constexpr (STATS == 2) {
        // dy = acc (+ the gradient already in the buffer) -> g = dy masked by (y > 0) -> stored; column sums of g and g xhat
        // over the tile: shuffles over the 16 rows of a group, the four waves through LDS, fixed orders
        float* sst = (float*)(misc + 16);   // [2][4][BN]
        const int myrow = r0 + wrow;
        const bool valid = myrow < a.c.n_out;
        const int orow = valid ? (a.c.out_map ? a.c.out_map[myrow] : myrow) : 0;
        float* po = a.c.out + (size_t)orow * a.c.ldo + ct0 * 16 + 4 * g;
        const float* pr = a.c.res ? a.c.res + (size_t)orow * a.c.ldr + ct0 * 16 + 4 * g : nullptr;
        const float* py = a.bw_y + (size_t)orow * a.bw_ldy + ct0 * 16 + 4 * g;
        const float* pw = a.bw_raw + (size_t)orow * a.bw_ldraw + ct0 * 16 + 4 * g;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          f32x4 v = acc[0][ct];
          if (pr) v += *(const f32x4*)(pr + ct * 16);
          if (a.bw_relu) {
            const f32x4 yv = *(const f32x4*)(py + ct * 16);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) v[tt] = yv[tt] > 0.f ? v[tt] : 0.f;
          }
          const f32x4 xh = (*(const f32x4*)(pw + ct * 16) - *(const f32x4*)(a.bw_mean + (ct0 + ct) * 16 + 4 * g)) *
                           *(const f32x4*)(a.bw_rstd + (ct0 + ct) * 16 + 4 * g);
          if (valid) *(f32x4*)(po + ct * 16) = v;
          f32x4 s1 = valid ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
          f32x4 s2 = s1 * xh;
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) s1[tt] = row16_sum(s1[tt]), s2[tt] = row16_sum(s2[tt]);
          if (j == 0) {
            *(f32x4*)(sst + wave * BN + ct * 16 + 4 * g) = s1;
            *(f32x4*)(sst + 4 * BN + wave * BN + ct * 16 + 4 * g) = s2;
          }
        }
        __syncthreads();
        if (tid < BN) {
          const float* s1p = sst + tid;
          const float* s2p = sst + 4 * BN + tid;
          float* P = a.stats + (size_t)t * 2 * a.stats_ld + cb * BN + tid;
          P[0] = ((s1p[0] + s1p[BN]) + s1p[2 * BN]) + s1p[3 * BN];
          P[a.stats_ld] = ((s2p[0] + s2p[BN]) + s2p[2 * BN]) + s2p[3 * BN];
        }
        __syncthreads();   // the scratch is rewritten by this workgroup's next owned tile
      } else {
#pragma unroll
      for (int r = 0; r < RG; ++r) {
        const int myrow = r0 + wrow + 16 * r;
        if (myrow < a.c.n_out) {
          const int orow = a.c.out_map ? a.c.out_map[myrow] : myrow;
          float* po = a.c.out + (size_t)orow * a.c.ldo + ct0 * 16 + 4 * g;
          const float* pr = a.c.res ? a.c.res + (size_t)orow * a.c.ldr + ct0 * 16 + 4 * g : nullptr;
          f32x4 rv[NCT];
          if (pr) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) rv[ct] = *(const f32x4*)(pr + ct * 16);
          }
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            f32x4 v = acc[r][ct];
            if (a.c.scale) v *= *(const f32x4*)(a.c.scale + (ct0 + ct) * 16 + 4 * g);
            if (a.c.shift) v += *(const f32x4*)(a.c.shift + (ct0 + ct) * 16 + 4 * g);
            if (pr) v += rv[ct];
            if (a.c.relu) {
#pragma unroll
              for (int tt = 0; tt < 4; ++tt) v[tt] = fmaxf(v[tt], 0.f);
            }
            *(f32x4*)(po + ct * 16) = v;
            if constexpr (HEAD) acc[r][ct] = v;   // the finished row stays in the accumulator registers for the head below
          }
        }
      }
      }
      if constexpr (HEAD) {
        // second GEMM on the finished tile: pcd[row][16 c2 + 4 g + r] = sum over the BN channels.  The weight fragments come
        // straight from global memory (48 KB, shared by every workgroup: L1 / L2 hits), four output column tiles at a time
        const f32x4* Wh = (const f32x4*)a.c.head_w + lane;
        const int hct = a.c.head_cout >> 4;
        const int myrow = r0 + wrow;
        const bool valid = myrow < a.c.n_out;
        float* ho = nullptr;
        if (valid) ho = a.c.head_out + (size_t)(a.c.head_map ? a.c.head_map[myrow] : myrow) * a.c.head_ld + 4 * g;
        // (c0, Sx) steps of four weight fragments x 16 MFMAs; the NEXT step's fragments are requested before the current
        // step's MFMAs (an L2 round trip behind 16 MFMAs: without it every step waited out its own loads -- the head cost
        // 460 us on the 16-scene batch against 336 for the separate k_dense launch)
        f32x4 wv[4], wn[4];
        auto fetch = [&](int c0, int Sx, f32x4 (&w)[4]) {
#pragma unroll
          for (int q = 0; q < 4; ++q) w[q] = c0 + q < hct ? Wh[(Sx * hct + c0 + q) * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};
        };
        fetch(0, 0, wv);
        for (int c0 = 0; c0 < hct; c0 += 4) {
          f32x4 h2[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) h2[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int Sx = 0; Sx < NCT; ++Sx) {
            if (Sx + 1 < NCT) fetch(c0, Sx + 1, wn);
            else if (c0 + 4 < hct) fetch(c0 + 4, 0, wn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
              for (int q = 0; q < 4; ++q) h2[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[q][tt], acc[0][Sx][tt], h2[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = wn[q];
          }
          if (valid) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (c0 + q < hct) {
                f32x4 o = h2[q];
                if (a.c.head_bias) o += *(const f32x4*)(a.c.head_bias + (c0 + q) * 16 + 4 * g);
                *(f32x4*)(ho + (c0 + q) * 16) = o;
              }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------ k_conv_deep
// The same convolution for the SMALL levels (a few hundred to a few thousand rows: levels 2-4 of one scene), where k_conv_sk
// is bound by its chain of dependent round trips, not by the matrix cores: ticket -> prefix table -> share search -> group
// masks -> neighbour rows -> first fragments, then one weight-slice round trip per stage behind a two-slot ring (a 256 -> 256
// layer on 140 rows: 25 us for 2 us of MFMA work and 7 MB of weights; profiles/r02_small_level_ablations.txt).  Here
//   * the partition is static per (64-row tile, column block) UNIT: its stages (present offsets x channel chunks, + the fused
//     projection's) are cut into P equal parts, workgroup = (unit, part) = blockIdx.x -- no ticket, no prefix table, no search;
//     the only thing a workgroup needs before it can issue loads is its tile's four group masks;
//   * the tile's whole neighbour table (K x 64 row ids) goes to LDS in the same round trip as the masks;
//   * BOTH operands of a stage travel by LDS-DMA (global_load_lds_dwordx4: the weight slice as in k_conv_sk, and every wave's
//     gathered 16 rows x CH channels: a lane's 16 bytes land at lane x 16 of the wave's piece, which the same lane reads back
//     with one ds_read_b128) into a FOUR-slot ring, three stages ahead of the MFMAs: no load of the loop has a register
//     destination, so the loop's only waits are counted s_waitcnt vmcnt(N) (N = the DMA instructions of the stages still in
//     flight: loads retire in order) followed by the stage barrier (MI355X_MICROARCH.md item 7: what orders a ds_read behind
//     an LDS-DMA);
//   * a unit cut into parts is finished WITHOUT anybody waiting: every part writes its accumulators to the slab
//     (write-through) and counts itself on the unit's counter; the part that finds P - 1 others there adds all parts in part
//     order (its own from registers: the same values) and runs the epilogue.  No spinning, so no assumption about dispatch
//     order or residency, and the sum does not depend on who arrives last: bit-identical run to run.
// One workgroup per CU (96-135 KB of LDS); the launch is used when the layer has at most a few stages per CU (plan_deep).
struct DeepArgs {
  ConvArgs c;
  int P;               // parts per unit
  int n_cblk;          // cout / BN
  int nchunk, nchunk2; // stages per offset (cin / CH); stages of the fused projection (cin2 / CH)
  int* cnt;            // [units] zeroed by the caller
  float* slab;         // [units * P][64 * BN]
  unsigned in_row_bytes, in2_row_bytes;
  unsigned long long* dbg;   // A3D_DEEP_DBG: [G][6] timestamps (100 MHz) of a workgroup's phases; nullptr otherwise
  // STATS builds (training, as k_conv_sk's): per 64-row tile and column the BatchNorm partials [n_tiles][2][stats_ld];
  // STATS == 2: the input-gradient conv's mask + sums (bw: y, raw, mean, rstd of the unit)
  float* stats;
  int stats_ld;
  BwArgs bw;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {   // vmcnt <= N (expcnt 7, lgkmcnt 15: not waited for)
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// (the ablation builds the round's measurements used -- no gathered pieces / no weight pieces / no MFMAs, four ring slots --
// were compiled out again: profiles/r05_experiments.txt has their numbers)
template <int BN, int CH, bool FUSE, int D = 3, int STATS = 0>
__global__ void __launch_bounds__(256, 1) k_conv_deep(const DeepArgs a) {
  static_assert(!STATS || !FUSE, "BatchNorm statistics are taken of a raw convolution");
  constexpr int NCT = BN / 16, NS = CH / 16, NW = 4;
  constexpr int NPW = NS * NCT;          // 1 KB pieces of a stage's weight slice
  static_assert(NPW % NW == 0, "every wave moves the same number of weight pieces (the vmcnt arithmetic)");
  constexpr int WV = NPW / NW;
  constexpr int WF = NPW * 256;          // floats of the weight slice
  constexpr int AF = NW * NS * 256;      // floats of the four waves' gathered fragments
  constexpr int SLOT = WF + AF;
  constexpr int PW = D - 1;              // D ring slots; PW stages in flight ahead of the one multiplied
  constexpr int NL = NS + WV;            // DMA instructions per wave and stage
  static_assert((PW - 1) * NL <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ring = (float*)smem;                         // [D][SLOT]
  int* nbrs = (int*)(ring + D * SLOT);                // [28][64]
  int* misc = nbrs + 28 * 64;
  const unsigned ring_addr = (unsigned)(size_t)ring;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int K = a.c.K, nchunk = a.nchunk;
  const int nchunk2 = FUSE ? a.nchunk2 : 0;
  const int T = a.c.n_tiles, P = a.P;
  const int cin16 = a.c.cin >> 4, cout16 = a.c.cout >> 4;
  const int u = (int)blockIdx.x / P, p = (int)blockIdx.x - u * P;
  const int cb = u / T, t = u - cb * T;
  const int ct0 = cb * NCT, r0 = t * 64;
  const int wrow = 16 * wave + j;
  auto stamp = [&](int i) {
    if (a.dbg && tid == 0) {
      a.dbg[(size_t)blockIdx.x * 6 + i] = __builtin_amdgcn_s_memrealtime();
      if (blockIdx.x == 0 && (i == 0 || i == 3)) a.dbg[(size_t)gridDim.x * 6 + (i ? 1 : 0)] = __builtin_amdgcn_s_memtime();
    }
  };
  stamp(0);

  // ---- round trip 1: the tile's group masks and its neighbour table
  uint32_t un = K >= 32 ? 0xffffffffu : (1u << K) - 1u, gm = un;
  if (a.c.gmask) {
    const uint32_t* gp = a.c.gmask + (r0 >> 4);
    const uint32_t m0 = gp[0], m1 = gp[1], m2 = gp[2], m3 = gp[3];
    un &= (m0 | m1) | (m2 | m3);
    gm = wave == 0 ? m0 : wave == 1 ? m1 : wave == 2 ? m2 : m3;
  }
  if (a.c.nbr) {
    for (int idx = tid; idx < K * 64; idx += 256) nbrs[idx] = a.c.nbr[(size_t)(idx >> 6) * a.c.nbr_stride + r0 + (idx & 63)];
  } else if (tid < 64) {
    nbrs[tid] = min(r0 + tid, a.c.n_in - 1);
  }
  if (blockIdx.x == 0 && a.c.zero_row >= 0)
    for (int cidx = tid; cidx < a.c.cout; cidx += 256) a.c.out[(size_t)a.c.zero_row * a.c.ldo + cidx] = 0.f;
  un = __builtin_amdgcn_readfirstlane(un);
  gm = __builtin_amdgcn_readfirstlane(gm);
  const int S_own = __builtin_popcount(un) * nchunk;
  const int S = S_own + nchunk2;
  if constexpr (FUSE) {
    un |= 1u << 27;
    gm |= 1u << 27;
  }
  const int s0 = S * p / P, s1 = S * (p + 1) / P;   // S <= 28 x 12 stages, P <= 24: 32-bit
  const int n = s1 - s0;
  __syncthreads();   // the neighbour table is in LDS (no DMA is pending yet: a plain barrier)
  stamp(1);

  f32x4 acc[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (n > 0) {
    auto next_k = [&](int k) -> int {
      const uint32_t rest = k >= 31 ? 0u : un & ~((2u << k) - 1u);
      return rest ? __builtin_ctz(rest) : 32;
    };
    auto adv = [&](int& k, int& c) {
      if (c + 1 < ((FUSE && k == 27) ? nchunk2 : nchunk)) {
        ++c;
      } else {
        k = next_k(k);
        c = 0;
      }
    };
    int ik, ic;   // the stage issued next
    if (FUSE && s0 >= S_own) {
      ik = 27;
      ic = s0 - S_own;
    } else {
      ik = __builtin_ctz(un);
      for (int i = s0 / nchunk; i > 0; --i) ik = next_k(ik);
      ic = s0 % nchunk;
    }
    int ck = ik, cc = ic;   // the stage multiplied next
    unsigned wsrc[WV], wdst[WV];
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int q = wave + NW * i;
      wsrc[i] = (unsigned)(((q / NCT) * cout16 + (q % NCT)) * 1024 + lane * 16);
      wdst[i] = (unsigned)q * 1024u;
    }
    const float* wbase = a.c.w + (size_t)ct0 * 256;
    const unsigned a_dst = (unsigned)(WF * 4 + wave * NS * 1024);
    // the DMA instructions of one stage: NS gathered pieces of this wave's 16 rows, WV pieces of the weight slice.  In the
    // loop they are issued BETWEEN the MFMA blocks of the stage being multiplied (a piece costs 60-180 cycles of issue:
    // in front of the MFMAs they delayed every stage by a third of its matrix time)
    struct Stage {
      const float* ab;
      const float* wst;
      unsigned roff, sl;
    };
    auto stage_of = [&](int k, int c, int slot) -> Stage {
      const bool proj = FUSE && k == 27;   // wave-uniform: the projection reads the block input at the output rows
      const int row = proj ? min(r0 + wrow, a.c.n_out) : nbrs[(k < K ? k : 0) * 64 + wrow];
      Stage st;
      st.roff = (unsigned)row * (proj ? a.in2_row_bytes : a.in_row_bytes) + 16u * g;
      st.ab = (proj ? a.c.in2 : a.c.in) + (size_t)c * CH;
      st.sl = ring_addr + (unsigned)slot * (SLOT * 4u);
      st.wst = wbase + ((size_t)k * cin16 + (size_t)c * NS) * cout16 * 256;
      return st;
    };
    auto issue_piece = [&](const Stage& st, int q) {   // q compile-time after unrolling: 0 .. NL - 1
      constexpr int NA = NS;
      if (q < NA) glds16_s(st.ab + 16 * q, st.roff, st.sl + a_dst + q * 1024u);
      else glds16_s(st.wst, wsrc[q - NA], st.sl + wdst[q - NA]);
    };
    auto issue = [&](int k, int c, int slot) {
      const Stage st = stage_of(k, c, slot);
#pragma unroll
      for (int q = 0; q < NL; ++q) issue_piece(st, q);
    };
    // multiply stage (k, slot); `nx` != nullptr: the pieces of that stage are issued behind the MFMA blocks
    auto compute = [&](int k, int slot, const Stage* nx) {
      constexpr int PER = (NL + NS - 1) / NS;   // pieces per 16-channel step
      if (!((gm >> k) & 1u)) {
        if (nx) {
#pragma unroll
          for (int q = 0; q < NL; ++q) issue_piece(*nx, q);
        }
        return;
      }
      const f32x4* Ws = (const f32x4*)(ring + slot * SLOT) + lane;
      const f32x4* As = (const f32x4*)(ring + slot * SLOT + WF) + wave * NS * 64 + lane;
      f32x4 b[2][NCT], av[2];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) b[0][ct] = Ws[ct * 64];
      av[0] = As[0];
#pragma unroll
      for (int Sx = 0; Sx < NS; ++Sx) {
        if (Sx + 1 < NS) {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) b[(Sx + 1) & 1][ct] = Ws[((Sx + 1) * NCT + ct) * 64];
          av[(Sx + 1) & 1] = As[(Sx + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[Sx & 1][ct][tt], av[Sx & 1][tt], acc[ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (nx) {
#pragma unroll
          for (int q = Sx * PER; q < (Sx + 1) * PER && q < NL; ++q) issue_piece(*nx, q);
        }
      }
    };
    for (int q = 0; q < PW && q < n; ++q) {
      issue(ik, ic, q);
      adv(ik, ic);
    }
    int slot_cur = 0, slot_nx = PW % D;   // ring slots of stage i and of stage i + PW
    for (int i = 0; i < n; ++i) {
      // stages i + 1 .. i + PW - 1 (as far as the part goes) are the loads allowed to stay in flight
      const int rem = n - 1 - i;
      if (PW >= 3 && rem >= 2) wait_vmcnt<2 * NL>();
      else if (PW >= 2 && rem >= 1) wait_vmcnt<NL>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // every wave's pieces of stage i have landed; every wave is done reading slot (i - 1) % D
      if (i == 0) stamp(2);
      if (i + PW < n) {
        const Stage nx = stage_of(ik, ic, slot_nx);
        adv(ik, ic);
        compute(ck, slot_cur, &nx);
      } else {
        compute(ck, slot_cur, nullptr);
      }
      adv(ck, cc);
      slot_cur = slot_cur + 1 == D ? 0 : slot_cur + 1;
      slot_nx = slot_nx + 1 == D ? 0 : slot_nx + 1;
    }
  }
  stamp(3);

  if (P > 1) {
    float* Pm = a.slab + (size_t)blockIdx.x * (64 * BN) + (size_t)tid * 4;
    if (n > 0) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) store_sc1(Pm + ct * 1024, acc[ct]);
    }
    wait_all_vmem();
    __syncthreads();
    if (tid == 0) misc[0] = __hip_atomic_fetch_add(a.cnt + u, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    stamp(4);
    if (misc[0] != P - 1) return;
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    // all parts in part order, this workgroup's own from its registers
    f32x4 tot[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) tot[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* P0 = a.slab + (size_t)u * P * (64 * BN) + (size_t)tid * 4;
    // 16 loads of 16 bytes in flight per thread: a 10-part tile of 32 columns is two round trips, not five (measured on the
    // 140-row level: 4.0 -> 1.5 us of the last arriver's life)
    constexpr int U = 16 / NCT;
    for (int q0 = 0; q0 < P; q0 += U) {
      f32x4 pa[U][NCT];
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        const int q = q0 + uu;
        const bool has = q < P && S * (q + 1) / P > S * q / P;
        if (has && q != p) {
          const float* Pa = P0 + (size_t)q * (64 * BN);
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) pa[uu][ct] = *(const f32x4*)(Pa + ct * 1024);
        } else {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) pa[uu][ct] = (has && q == p) ? acc[ct] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int uu = 0; uu < U; ++uu)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) tot[ct] += pa[uu][ct];
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = tot[ct];
  }

  const int myrow = r0 + wrow;
  if constexpr (STATS == 1) {
    // BatchNorm statistics of the raw tile, exactly as k_conv_sk<..., STATS = 1> takes them (the same fixed orders: the
    // partials of a tile do not depend on which kernel produced it beyond the rounding of the accumulators)
    float* sst = (float*)(misc + 16);   // [2][4][BN]
    const bool valid = myrow < a.c.n_out;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      f32x4 sv = valid ? acc[ct] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) sv[tt] = row16_sum(sv[tt]);
      if (j == 0) *(f32x4*)(sst + wave * BN + ct * 16 + 4 * g) = sv;
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    const float inv = 1.f / (float)min(64, a.c.n_out - r0);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float* sp = sst + ct * 16 + 4 * g;
      const f32x4 tsum = ((*(const f32x4*)sp + *(const f32x4*)(sp + BN)) + *(const f32x4*)(sp + 2 * BN)) + *(const f32x4*)(sp + 3 * BN);
      f32x4 dv = valid ? acc[ct] - tsum * inv : (f32x4){0.f, 0.f, 0.f, 0.f};
      dv = dv * dv;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) dv[tt] = row16_sum(dv[tt]);
      if (j == 0) *(f32x4*)(sst + 4 * BN + wave * BN + ct * 16 + 4 * g) = dv;
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (tid < BN) {
      const float* s1p = sst + tid;
      const float* s2p = sst + 4 * BN + tid;
      float* Pq = a.stats + (size_t)t * 2 * a.stats_ld + cb * BN + tid;
      Pq[0] = ((s1p[0] + s1p[BN]) + s1p[2 * BN]) + s1p[3 * BN];
      Pq[a.stats_ld] = ((s2p[0] + s2p[BN]) + s2p[2 * BN]) + s2p[3 * BN];
    }
  }
  if constexpr (STATS == 2) {
    // input-gradient conv: g = (dy (+ the gradient already in the buffer)) masked by y > 0, stored; tile sums of g and g xhat
    float* sst = (float*)(misc + 16);   // [2][4][BN]
    const bool valid = myrow < a.c.n_out;
    const int orow = valid ? (a.c.out_map ? a.c.out_map[myrow] : myrow) : 0;
    float* po = a.c.out + (size_t)orow * a.c.ldo + ct0 * 16 + 4 * g;
    const float* pr = a.c.res ? a.c.res + (size_t)orow * a.c.ldr + ct0 * 16 + 4 * g : nullptr;
    const float* py = a.bw.y + (size_t)orow * a.bw.ldy + ct0 * 16 + 4 * g;
    const float* pw = a.bw.raw + (size_t)orow * a.bw.ldraw + ct0 * 16 + 4 * g;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      f32x4 v = acc[ct];
      if (pr) v += *(const f32x4*)(pr + ct * 16);
      if (a.bw.relu) {
        const f32x4 yv = *(const f32x4*)(py + ct * 16);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) v[tt] = yv[tt] > 0.f ? v[tt] : 0.f;
      }
      const f32x4 xh = (*(const f32x4*)(pw + ct * 16) - *(const f32x4*)(a.bw.mean + (ct0 + ct) * 16 + 4 * g)) *
                       *(const f32x4*)(a.bw.rstd + (ct0 + ct) * 16 + 4 * g);
      if (valid) *(f32x4*)(po + ct * 16) = v;
      f32x4 s1 = valid ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 s2 = s1 * xh;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) s1[tt] = row16_sum(s1[tt]), s2[tt] = row16_sum(s2[tt]);
      if (j == 0) {
        *(f32x4*)(sst + wave * BN + ct * 16 + 4 * g) = s1;
        *(f32x4*)(sst + 4 * BN + wave * BN + ct * 16 + 4 * g) = s2;
      }
    }
    __syncthreads();
    if (tid < BN) {
      const float* s1p = sst + tid;
      const float* s2p = sst + 4 * BN + tid;
      float* Pq = a.stats + (size_t)t * 2 * a.stats_ld + cb * BN + tid;
      Pq[0] = ((s1p[0] + s1p[BN]) + s1p[2 * BN]) + s1p[3 * BN];
      Pq[a.stats_ld] = ((s2p[0] + s2p[BN]) + s2p[2 * BN]) + s2p[3 * BN];
    }
    stamp(5);
    return;
  }
  if (myrow < a.c.n_out) {
    const int orow = a.c.out_map ? a.c.out_map[myrow] : myrow;
    float* po = a.c.out + (size_t)orow * a.c.ldo + ct0 * 16 + 4 * g;
    const float* pr = a.c.res ? a.c.res + (size_t)orow * a.c.ldr + ct0 * 16 + 4 * g : nullptr;
    f32x4 rv[NCT];
    if (pr) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) rv[ct] = *(const f32x4*)(pr + ct * 16);
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      f32x4 v = acc[ct];
      if (a.c.scale) v *= *(const f32x4*)(a.c.scale + (ct0 + ct) * 16 + 4 * g);
      if (a.c.shift) v += *(const f32x4*)(a.c.shift + (ct0 + ct) * 16 + 4 * g);
      if (pr) v += rv[ct];
      if (a.c.relu) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) v[tt] = fmaxf(v[tt], 0.f);
      }
      *(f32x4*)(po + ct * 16) = v;
    }
  }
  stamp(5);
}

// ------------------------------------------------------------------------------ k_conv_wl
// Gathered convolution with the WHOLE packed weight set resident in LDS (K cin cout 4 bytes <= 144 KB: the 3^3 32 -> 32
// layers of level 1, 110 KB, and the 2^3 stride-2 layers 32 -> 32 / 64 -> 64).  With 32 input channels a stage of
// k_conv_sk is 16 MFMAs per wave between two workgroup barriers, each behind a global-memory round trip: latency-bound
// (35 TF/s on 72 k rows).  Here nothing is shared after the one-time weight load, so there is no barrier and no weight
// DMA in the loop: one persistent workgroup per CU, 16 waves, every wave walks its own sequence of 16-row groups
// (group i -> wave i mod W: rows are sorted by neighbour mask, so every wave gets the same mix of light and heavy
// groups); per (group, present offset): one neighbour-index load two pairs ahead, the gathered A fragments one pair
// ahead, NS NCT ds_read_b128 + 4 NS NCT MFMAs.  Same summation order per output element as k_conv_sk run without
// hand-offs (offsets ascending, channels ascending), so results are bit-identical to it.
// STATS (training): every finished 16-row group also writes its BatchNorm partials (column sums, squared deviations from
// the GROUP mean) to stats[group][2][stats_ld] -- a group belongs to one wave, so this needs no LDS and no barrier.
template <int NS, int NCT, int STATS = 0>
__global__ void __launch_bounds__(1024) k_conv_wl(const ConvArgs c, int ngroups, float* stats = nullptr, int stats_ld = 0,
                                                  const BwArgs bw = BwArgs()) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* Wl = (f32x4*)smem;   // [K][NS][NCT][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int K = c.K;
  if (blockIdx.x == 0 && c.zero_row >= 0)
    for (int cidx = tid; cidx < c.cout; cidx += blockDim.x) c.out[(size_t)c.zero_row * c.ldo + cidx] = 0.f;
  const int W = gridDim.x * nw;
  int grp = wave * gridDim.x + blockIdx.x;   // consecutive groups on different CUs
  const uint32_t full = K >= 32 ? 0xffffffffu : (1u << K) - 1u;
  auto mask_of = [&](int gq) -> uint32_t {
    return gq < ngroups ? (c.gmask ? c.gmask[gq] & full : full) : 0u;
  };
  // ---- pipeline state: pair p0 = (g0, k0) is multiplied, p1 has its A fragments in flight, p2 its row index
  auto row_index = [&](int gq, int k) -> int {
    return c.nbr[(size_t)k * c.nbr_stride + gq * 16 + j];
  };
  const char* inb = (const char*)c.in;
  const unsigned row_bytes = (unsigned)c.ldi * 4u;
  auto load_a = [&](f32x4 (&A)[NS], int idx) {
    const char* ar = inb + (size_t)(unsigned)idx * row_bytes + 16u * g;
#pragma unroll
    for (int S = 0; S < NS; ++S) A[S] = *(const f32x4*)(ar + 64 * S);
  };
  // iterator over (group, offset) pairs of this wave: (gq, rest) -> next pair; groups without offsets are skipped
  int it_g = grp;
  uint32_t it_m = mask_of(it_g);
  uint32_t it_mnext = mask_of(it_g + W);   // the mask of the group after: requested a whole group ahead
  auto advance = [&](int& og, int& ok) -> bool {   // false: no more pairs
    while (it_m == 0u) {
      if (it_g >= ngroups) return false;
      it_g += W;
      it_m = it_mnext;
      it_mnext = mask_of(it_g + W);
    }
    ok = __builtin_ctz(it_m);
    og = it_g;
    it_m &= it_m - 1u;
    return true;
  };
  // pipeline: pair p0 is multiplied, p1 and p2 have their A fragments in flight, p3 its row index (a pair's 4 NS NCT
  // MFMAs take 0.2-0.9 us of a SIMD shared by four waves; a gathered row takes 1-2 us under load: two pairs of cover)
  int g0 = 0, k0 = 0, g1 = 0, k1 = 0, g2 = 0, k2 = 0, g3 = 0, k3 = 0;
  bool v0 = advance(g0, k0), v1 = v0 && advance(g1, k1), v2 = v1 && advance(g2, k2), v3 = v2 && advance(g3, k3);
  int idx3 = 0;
  f32x4 A0[NS], A1[NS], A2[NS];
  {
    int i0 = 0, i1 = 0, i2 = 0;
    if (v0) i0 = row_index(g0, k0);
    if (v1) i1 = row_index(g1, k1);
    if (v2) i2 = row_index(g2, k2);
    if (v3) idx3 = row_index(g3, k3);
    if (v0) load_a(A0, i0);
    if (v1) load_a(A1, i1);
    if (v2) load_a(A2, i2);
  }
  {  // the packed weights: 8 independent 16-byte loads in flight per thread, then the LDS stores
    const int TOT = K * NS * NCT * 64, bd = blockDim.x;
    for (int base = tid; base < TOT; base += 8 * bd) {
      f32x4 t8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u * bd < TOT) t8[u] = ((const f32x4*)c.w)[base + u * bd];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u * bd < TOT) Wl[base + u * bd] = t8[u];
    }
  }
  __syncthreads();
  f32x4 acc[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  while (v0) {
    // fragments of the pair three ahead (its index was requested an iteration ago), the index of the pair four ahead
    f32x4 A3[NS];
    if (v3) load_a(A3, idx3);
    int g4 = 0, k4 = 0;
    const bool v4 = v3 && advance(g4, k4);
    if (v4) idx3 = row_index(g4, k4);
    const f32x4* Ws = Wl + (size_t)k0 * (NS * NCT * 64) + lane;
#pragma unroll
    for (int S = 0; S < NS; ++S) {
      f32x4 b[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) b[ct] = Ws[(S * NCT + ct) * 64];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ct][tt], A0[S][tt], acc[ct], 0, 0, 0);
    }
    if (!v1 || g1 != g0) {   // last offset of group g0: epilogue (as k_conv_sk's)
      const int myrow = g0 * 16 + j;
      if constexpr (STATS == 2) {   // input-gradient conv: g = (dy (+ res)) masked by y > 0, stored; sums of g and g xhat per group
        const bool valid = myrow < c.n_out;
        const int orow = valid ? (c.out_map ? c.out_map[myrow] : myrow) : 0;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          f32x4 v = acc[ct];
          if (c.res) v += *(const f32x4*)(c.res + (size_t)orow * c.ldr + ct * 16 + 4 * g);
          if (bw.relu) {
            const f32x4 yv = *(const f32x4*)(bw.y + (size_t)orow * bw.ldy + ct * 16 + 4 * g);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) v[tt] = yv[tt] > 0.f ? v[tt] : 0.f;
          }
          const f32x4 xh = (*(const f32x4*)(bw.raw + (size_t)orow * bw.ldraw + ct * 16 + 4 * g) - *(const f32x4*)(bw.mean + ct * 16 + 4 * g)) *
                           *(const f32x4*)(bw.rstd + ct * 16 + 4 * g);
          if (valid) *(f32x4*)(c.out + (size_t)orow * c.ldo + ct * 16 + 4 * g) = v;
          f32x4 s1 = valid ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
          f32x4 s2 = s1 * xh;
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) s1[tt] = row16_sum(s1[tt]), s2[tt] = row16_sum(s2[tt]);
          if (j == 0) {
            float* P = stats + (size_t)g0 * 2 * stats_ld + ct * 16 + 4 * g;
            *(f32x4*)P = s1;
            *(f32x4*)(P + stats_ld) = s2;
          }
          acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      } else {
      if constexpr (STATS == 1) {
        const bool valid = myrow < c.n_out;
        const float inv = 1.f / (float)min(16, c.n_out - g0 * 16);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          f32x4 sv = valid ? acc[ct] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) sv[tt] = row16_sum(sv[tt]);
          f32x4 d = valid ? acc[ct] - sv * inv : (f32x4){0.f, 0.f, 0.f, 0.f};
          d = d * d;
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) d[tt] = row16_sum(d[tt]);
          if (j == 0) {
            float* P = stats + (size_t)g0 * 2 * stats_ld + ct * 16 + 4 * g;
            *(f32x4*)P = sv;
            *(f32x4*)(P + stats_ld) = d;
          }
        }
      }
      if (myrow < c.n_out) {
        const int orow = c.out_map ? c.out_map[myrow] : myrow;
        float* po = c.out + (size_t)orow * c.ldo + 4 * g;
        const float* pr = c.res ? c.res + (size_t)orow * c.ldr + 4 * g : nullptr;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          f32x4 v = acc[ct];
          if (c.scale) v *= *(const f32x4*)(c.scale + ct * 16 + 4 * g);
          if (c.shift) v += *(const f32x4*)(c.shift + ct * 16 + 4 * g);
          if (pr) v += *(const f32x4*)(pr + ct * 16);
          if (c.relu) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) v[tt] = fmaxf(v[tt], 0.f);
          }
          *(f32x4*)(po + ct * 16) = v;
        }
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int S = 0; S < NS; ++S) {
      A0[S] = A1[S];
      A1[S] = A2[S];
      A2[S] = A3[S];
    }
    g0 = g1, k0 = k1, v0 = v1;
    g1 = g2, k1 = k2, v1 = v2;
    g2 = g3, k2 = k3, v2 = v3;
    g3 = g4, k3 = k4, v3 = v4;
  }
}

static bool conv_wl_supported(const ConvArgs& c) {
  if (c.K <= 1 || !c.nbr || !c.gmask) return false;
  if (!(c.cin == 32 && c.cout == 32)) return false;   // 64 -> 64 stride-2 (128 KB) measured slower than k_conv_sk (21 vs 18 us)
  return (size_t)c.K * c.cin * c.cout * 4 <= 144 * 1024;
}

static int launch_conv_wl(const ConvArgs& c, hipStream_t st, float* stats = nullptr, int stats_ld = 0, const BwArgs* bw = nullptr) {
  const int ngroups = (c.n_out + 15) / 16;
  const size_t lds = (size_t)c.K * c.cin * c.cout * 4;
  // one workgroup per CU; 16 waves when there are groups for them, never fewer than 4
  int grid = 256, nw = 16;
  while (nw > 4 && (long long)grid * nw > 2LL * ngroups) nw >>= 1;
  if ((long long)grid * nw > ngroups) grid = (ngroups + nw - 1) / nw;
  if (grid < 1) grid = 1;
  ProfScope ps(st, A3D_PROF_SPCONV, c.cout, c.K, c.cin, c.cout, c.n_out, c.tag_table, c.tag_level, 0);   // stage width 0: k_conv_wl
  if (stats) {
    if (c.cin != 32) {
      set_error("spconv: no statistics build of the 64-channel LDS-resident kernel");
      return A3D_ERR_UNSUPPORTED;
    }
    if (bw) k_conv_wl<2, 2, 2><<<grid, 64 * nw, lds, st>>>(c, ngroups, stats, stats_ld, *bw);
    else k_conv_wl<2, 2, 1><<<grid, 64 * nw, lds, st>>>(c, ngroups, stats, stats_ld);
  } else if (c.cin == 32) k_conv_wl<2, 2><<<grid, 64 * nw, lds, st>>>(c, ngroups);
  else k_conv_wl<4, 4><<<grid, 64 * nw, lds, st>>>(c, ngroups);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

// ------------------------------------------------------------------------------ dense GEMM
// Y[n][16*NCT] = act(((X (+ X2))[n][16*NS] @ W) * scale + shift + res): the N-point nn.Linear pieces of the
// decoder and the 1x1 convolutions (no gather, no masks).  HBM-bound (reads X (+X2), writes Y once), so the
// kernel is built around that: the packed weight matrix (<= 64 KB) sits in LDS for the life of a persistent
// workgroup, every lane pulls its MFMA fragments of X straight from global memory (row j, channels
// 16S+4g..+3: the same K permutation as k_spconv, so the fragment is one 16-byte load), the loads of the
// wave's NEXT 16-row group are in flight behind the MFMAs of the current one, and the product is computed
// TRANSPOSED (weights as the A operand), which leaves
// four consecutive output channels of one row in each lane: 16-byte stores, 16-byte scale/shift/res loads.
template <int NS, int NCT, bool HAS2>   // HAS2: a second operand X2 is added to X (its 4 NS registers exist only then)
__global__ void __launch_bounds__(768) k_dense(const float* __restrict__ X, int ldx, const float* __restrict__ X2,
                                                   int ldx2, int n, const float* __restrict__ Wp,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   const float* __restrict__ res, int ldr, int relu,
                                                   float* __restrict__ Y, int ldy, int ngroups, int zero_row,
                                                   const int* __restrict__ out_map) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* Wl = (f32x4*)smem;                         // [NS][NCT][64 lanes]
  if (zero_row >= 0 && blockIdx.x == 0 && threadIdx.x < 16 * NCT) Y[(size_t)zero_row * ldy + threadIdx.x] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  // every wave walks its own sequence of 16-row groups (no workgroup-level coupling after the weight load)
  const int nw = blockDim.x >> 6;
  const int stride = gridDim.x * nw;
  int grp = blockIdx.x * nw + wave;
  f32x4 nx[NS], nx2[HAS2 ? NS : 1];
  auto fetch = [&](int gq) {
    const int row = min(gq * 16 + j, n - 1);
    const float* xr = X + (size_t)row * ldx + 4 * g;
#pragma unroll
    for (int S = 0; S < NS; ++S) nx[S] = *(const f32x4*)(xr + 16 * S);
    if constexpr (HAS2) {
      const float* x2r = X2 + (size_t)row * ldx2 + 4 * g;
#pragma unroll
      for (int S = 0; S < NS; ++S) nx2[S] = *(const f32x4*)(x2r + 16 * S);
    }
  };
  if (grp < ngroups) fetch(grp);
  {  // stage the packed weights: 8 independent 16-byte loads in flight per thread, then the LDS stores
    constexpr int TOT = NS * NCT * 64;
    const int bd = blockDim.x;
    for (int base = threadIdx.x; base < TOT; base += 8 * bd) {
      f32x4 t8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u * bd < TOT) t8[u] = ((const f32x4*)Wp)[base + u * bd];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u * bd < TOT) Wl[base + u * bd] = t8[u];
    }
  }
  __syncthreads();
  while (grp < ngroups) {
    f32x4 a[NS];
#pragma unroll
    for (int S = 0; S < NS; ++S) {
      if constexpr (HAS2) a[S] = nx[S] + nx2[S];
      else a[S] = nx[S];
    }
    const int next = grp + stride;
    if (next < ngroups) fetch(next);                // in flight behind this group's 32 NS NCT/8 k-cycles of MFMA
    f32x4 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      // two column tiles at a time on alternating accumulators (four MFMAs in a row on ONE accumulator wait out the
      // matrix pipe's latency: a dependent issue every ~40 cycles instead of 32), the next pair's ds_read_b128s in flight
      // behind the current pair's eight MFMAs (k_conv_sk's low-register loop)
      static_assert(NCT % 2 == 0, "pairs of 16-column tiles");
      constexpr int NP = NS * NCT / 2;
      const f32x4* Ws = Wl + lane;
      f32x4 c0 = Ws[0], c1 = Ws[64];
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) {
        const int S = pi / (NCT / 2), ct = 2 * (pi % (NCT / 2));
        f32x4 n0 = c0, n1 = c1;
        if (pi + 1 < NP) {
          n0 = Ws[(2 * pi + 2) * 64];
          n1 = Ws[(2 * pi + 3) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[t], a[S][t], acc[ct], 0, 0, 0);
          acc[ct + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[t], a[S][t], acc[ct + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        c0 = n0;
        c1 = n1;
      }
    }
    // acc[ct][t] = Y[16 grp + j][16 ct + 4 g + t]
    if (grp * 16 + j < n) {
      const int row = out_map ? out_map[grp * 16 + j] : grp * 16 + j;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int col = 16 * ct + 4 * g;
        f32x4 v = acc[ct];
        if (scale) v *= *(const f32x4*)(scale + col);
        if (shift) v += *(const f32x4*)(shift + col);
        if (res) v += *(const f32x4*)(res + (size_t)row * ldr + col);
        if (relu) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
        }
        *(f32x4*)(Y + (size_t)row * ldy + col) = v;
      }
    }
    grp = next;
  }
}

static void allow_big_lds();
static bool dense_supported(int cin, int cout) {
  return (cin == 96 || cin == 128) && (cout == 96 || cout == 128);
}
static int launch_dense(const float* X, int ldx, const float* X2, int ldx2, int n, int cin, int cout, const float* Wp,
                        const float* scale, const float* shift, const float* res, int ldr, int relu, float* Y,
                        int ldy, int zero_row, int tag_level, const int* out_map, hipStream_t st) {
  if ((ldx | ldx2 | ldr | ldy) & 3) {
    set_error("a3d_linear: leading dimensions must be multiples of 4");
    return A3D_ERR_INVALID;
  }
  allow_big_lds();
  const int ngroups = (n + 15) / 16;
  // workgroup shape by row count.  80 k rows, 128 -> 128: 4x2 41 us, 8x2 48, 8x1 47, 12x1 46, 6x2 60; 1.28 M rows (16-scene
  // batch), 128 -> 96 / 96 -> 128: 4x2 387 / 366 us, 6x2 461 / 441, 8x2 379 / 364, 12x1 350 / 335 -- the kernel is latency-bound
  // on its row loads (TCP pending-stall 0.5-0.66 of its active cycles), twelve waves per CU keep more of them in flight; a
  // 128-register build with sixteen waves per CU measured the same 351 / 334 us and was dropped
  const int nw = n >= 200000 ? 12 : 4, wg_per_cu = n >= 200000 ? 1 : 2;
  const size_t lds = (size_t)(cin / 16) * (cout / 16) * 1024;
  ProfScope ps(st, A3D_PROF_DENSE, 0, 1, cin, cout, n, A3D_OP_LINEAR, tag_level, 1);
  const int max_grid = 256 * wg_per_cu;
  const int grid = (ngroups + nw - 1) / nw < max_grid ? (ngroups + nw - 1) / nw : max_grid;
#define A3D_DENSE(NS_, NCT_) \
  do { \
    if (X2) k_dense<NS_, NCT_, true><<<grid, 64 * nw, lds, st>>>(X, ldx, X2, ldx2, n, Wp, scale, shift, res, ldr, relu, Y, ldy, ngroups, \
                                                                 zero_row, out_map); \
    else k_dense<NS_, NCT_, false><<<grid, 64 * nw, lds, st>>>(X, ldx, X2, ldx2, n, Wp, scale, shift, res, ldr, relu, Y, ldy, ngroups, \
                                                               zero_row, out_map); \
  } while (0)
  if (cin == 128 && cout == 128) A3D_DENSE(8, 8);
  else if (cin == 128 && cout == 96) A3D_DENSE(8, 6);
  else if (cin == 96 && cout == 128) A3D_DENSE(6, 8);
  else A3D_DENSE(6, 6);
#undef A3D_DENSE
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

// ------------------------------------------------------------------------------ stem
// 5^3 (or 3^3) conv with Cin = 3 (res16unet.py:39-47,225-227; conv1_kernel_size main.py:37).
// FLOPs are negligible (0.6 GF at 80 k voxels); the cost is 125 hash probes per voxel, so the
// kernel map is never materialised: probe -> LDS table -> 3x32 FMAs per existing neighbour.
template <int NT, int KS>   // threads per 64-voxel workgroup: NT / 64 threads share a voxel, 32 * 64 / NT output channels each;
                            // KS: kernel size (compile-time: the index arithmetic of the 125 lookups is divisions by it)
__global__ void __launch_bounds__(NT) k_stem(const int32_t* __restrict__ xyzb, int n,
                                              const uint64_t* __restrict__ hk, const int* __restrict__ hv,
                                              uint32_t hmask, const f32x4* __restrict__ feats4,
                                              const float* __restrict__ w, int /*ks*/,
                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                              int relu, float* out, int ldo, int zero_row, const Level glv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ks = KS, K = KS * KS * KS;
  float* W = (float*)smem;              // [K][3][32]
  int* nb = (int*)(W + K * 96);         // [64][K]
  const int tid = threadIdx.x;
  for (int e = tid; e < K * 96; e += NT) W[e] = w[e];
  const int v0 = blockIdx.x * 64;
  const int h = ks / 2;
  constexpr int TPV = NT / 64, CPT = 32 / TPV;
  constexpr int SEG = (K + TPV - 1) / TPV;   // neighbour entries per thread
  const int v = tid / TPV, cg = tid % TPV;
  const bool fused = glv.grid && n < (1 << 25);
  // Grid path: the workgroup's 64 voxels are 64 consecutive MORTON rows (a compact patch of the surface: their 5^3
  // neighbourhoods overlap, so the grid cells and -- the grid holds Morton rows, feats4 is in Morton order -- the gathered
  // feature rows of neighbouring threads share cache lines; in the internal, mask-sorted order every lookup was its own
  // 64-byte sector from the fabric).  The result goes to the voxel's internal row (perm).
  const int mrow = v0 + v;
  const int row = fused ? (mrow < n ? glv.perm[mrow] : n) : mrow;
  int mine[SEG];
  if (fused) {
    // dense level-0 grid (scene.hip): one 4-byte load per neighbour; the TPV threads of a voxel look up SEG consecutive
    // offsets each (x fastest: runs of ks consecutive cells), all of a thread's lookups in flight, results kept in registers
    int4 c = (int4){0, 0, 0, 0};
    if (mrow < n) {
      int b_, X_, Y_, Z_;
      decode_key(glv.keys[mrow], 0, b_, X_, Y_, Z_);
      c = (int4){X_, Y_, Z_, b_};
    }
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
      const int k = cg * SEG + u;
      mine[u] = -1;
      if (k < K && row < n)
        mine[u] = glv.grid[grid_cell(glv, c.w, c.x + (k % ks) - h, c.y + ((k / ks) % ks) - h, c.z + (k / (ks * ks)) - h)];
    }
  } else if (glv.grid) {
    for (int e0 = tid; e0 < 64 * K; e0 += 8 * NT) {
      int res[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * NT;
        const int vv = e / K, k = e - vv * K;
        const int rr = v0 + vv;
        res[u] = -1;
        if (e < 64 * K && rr < n) {
          const int4 c = *(const int4*)(xyzb + 4 * rr);
          res[u] = glv.grid[grid_cell(glv, c.w, c.x + (k % ks) - h, c.y + ((k / ks) % ks) - h, c.z + (k / (ks * ks)) - h)];
          if (res[u] >= 0) res[u] = glv.perm[res[u]];   // the grid holds Morton rows; this path works on internal rows
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e0 + u * NT < 64 * K) nb[e0 + u * NT] = res[u];
    }
  } else
  // hash probes, four independent lookups in flight per thread (the probes are L2-latency bound)
  for (int e0 = tid; e0 < 64 * K; e0 += 4 * NT) {
    uint64_t key[4];
    uint32_t hh[4];
    int res[4];
    bool open[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * NT;
      const int v = e / K, k = e - v * K;
      const int row = v0 + v;
      res[u] = -1;
      open[u] = false;
      key[u] = 0;
      hh[u] = 0;
      if (e < 64 * K && row < n) {
        const int X = xyzb[4 * row + 0] + (k % ks) - h;
        const int Y = xyzb[4 * row + 1] + ((k / ks) % ks) - h;
        const int Z = xyzb[4 * row + 2] + (k / (ks * ks)) - h;
        const int b = xyzb[4 * row + 3];
        const int lim = kCoordOff;
        if (X >= -lim && X < lim && Y >= -lim && Y < lim && Z >= -lim && Z < lim) {
          key[u] = make_key(b, X, Y, Z, 0);
          hh[u] = hash64(key[u]) & hmask;
          open[u] = true;
        }
      }
    }
    for (uint32_t probe = 0; probe <= hmask; ++probe) {
      uint64_t got[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) got[u] = open[u] ? hk[hh[u]] : kEmptyKey;
      bool any = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!open[u]) continue;
        if (got[u] == key[u]) {
          res[u] = hv[hh[u]];
          open[u] = false;
        } else if (got[u] == kEmptyKey) {
          open[u] = false;
        } else {
          hh[u] = (hh[u] + 1) & hmask;
          any = true;
        }
      }
      if (!any) break;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (e0 + u * NT < 64 * K) nb[e0 + u * NT] = res[u];
  }
  __syncthreads();
  float acc[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) acc[c] = 0.f;
  if (n < (1 << 25)) {
    // a surface voxel has ~40 of its 125 neighbours: the TPV threads of a voxel compact its list (ascending k, so the
    // sums keep their order), k packed above the row; then EIGHT gathers in flight per thread over the present
    // neighbours only (was: five in flight over all 125, 25 dependent rounds per workgroup)
    constexpr int seg = SEG;
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
      const int k = cg * seg + u;
      if (!fused) mine[u] = k < K ? nb[v * K + k] : -1;
      cnt += mine[u] >= 0 ? 1 : 0;
    }
    int incl = cnt;   // inclusive prefix over the TPV consecutive lanes of this voxel
#pragma unroll
    for (int o = 1; o < TPV; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (cg >= o) incl += t;
    }
    const int total = __shfl(incl, (threadIdx.x & 63) | (TPV - 1), 64);
    if (!fused) __syncthreads();   // every entry is in registers: the list can be overwritten
    int pos = incl - cnt;
#pragma unroll
    for (int u = 0; u < SEG; ++u)
      if (mine[u] >= 0) nb[v * K + pos++] = (int)((uint32_t)mine[u] | ((uint32_t)(cg * seg + u) << 25));
    __syncthreads();
    for (int i0 = 0; i0 < total; i0 += 8) {
      uint32_t e[8];
      f32x4 f[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) e[u] = i0 + u < total ? (uint32_t)nb[v * K + i0 + u] : 0u;
#pragma unroll
      for (int u = 0; u < 8; ++u) f[u] = feats4[e[u] & 0x1ffffffu];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (i0 + u >= total) continue;
        const float* wk = W + (e[u] >> 25) * 96 + cg * CPT;
#pragma unroll
        for (int c = 0; c < CPT; ++c) acc[c] += f[u][0] * wk[c] + f[u][1] * wk[32 + c] + f[u][2] * wk[64 + c];
      }
    }
  } else
  // neighbour features: five gathers in flight per thread (missing neighbours read row 0 and are masked)
  for (int k0 = 0; k0 < K; k0 += 5) {
    int r[5];
    f32x4 f[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) r[u] = k0 + u < K ? nb[v * K + k0 + u] : -1;
#pragma unroll
    for (int u = 0; u < 5; ++u) f[u] = feats4[r[u] >= 0 ? r[u] : 0];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      if (r[u] < 0) continue;
      const float* wk = W + (k0 + u) * 96 + cg * CPT;
#pragma unroll
      for (int c = 0; c < CPT; ++c) acc[c] += f[u][0] * wk[c] + f[u][1] * wk[32 + c] + f[u][2] * wk[64 + c];
    }
  }
  if (row < n) {
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int col = cg * CPT + c;
      float y = acc[c] * (scale ? scale[col] : 1.f) + (shift ? shift[col] : 0.f);
      if (relu) y = fmaxf(y, 0.f);
      out[(size_t)row * ldo + col] = y;
    }
  }
  if (zero_row >= 0 && blockIdx.x == 0 && tid < 32) out[(size_t)zero_row * ldo + tid] = 0.f;
}

// feats4[f] = the colours of internal row f -- or, with `perm` (Morton row -> internal row), of MORTON row f
__global__ void k_gather_feats(const float* __restrict__ feats3, const int* __restrict__ orig_row, int n,
                               f32x4* feats4, const int* __restrict__ perm) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const float* p = feats3 + (size_t)orig_row[perm ? perm[f] : f] * 3;
  feats4[f] = (f32x4){p[0], p[1], p[2], 0.f};
}

// W[K][cin][cout] -> Wp[K][cin/16][cout/16][lane = 16 g + j][t] = W[k][16 S + 4 g + t][16 ct + j]
__global__ void k_pack_weight(const float* __restrict__ w, int K, int cin, int cout, float* out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)K * cin * cout;
  if (e >= total) return;
  const int t = (int)(e & 3), lane = (int)((e >> 2) & 63);
  size_t rest = e >> 8;
  const int cout16 = cout >> 4, cin16 = cin >> 4;
  const int ct = (int)(rest % cout16);
  rest /= cout16;
  const int S = (int)(rest % cin16);
  const int k = (int)(rest / cin16);
  const int g = lane >> 4, j = lane & 15;
  out[e] = w[((size_t)k * cin + 16 * S + 4 * g + t) * cout + 16 * ct + j];
}

// the same weights as three bf16 planes for the emulated-fp32 build of the 96-column kernel:
//   Wb[K][cin/32][cout/16][plane h, m, l][lane = 16 g + j][e] = plane(W[k][32 c + 8 g + e][16 ct + j]),  8 bf16 per lane
__global__ void k_pack_weight_emu(const float* __restrict__ w, int K, int cin, int cout, uint16_t* out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per weight
  const size_t total = (size_t)K * cin * cout;
  if (e >= total) return;
  const int el = (int)(e & 7), lane = (int)((e >> 3) & 63);
  size_t rest = e >> 9;
  const int cout16 = cout >> 4, cin32 = cin >> 5;
  const int ct = (int)(rest % cout16);
  rest /= cout16;
  const int c = (int)(rest % cin32);
  const int k = (int)(rest / cin32);
  const int g = lane >> 4, j = lane & 15;
  const float v = w[((size_t)k * cin + 32 * c + 8 * g + el) * cout + 16 * ct + j];
  const uint32_t hb = __float_as_uint(v) & 0xffff0000u;
  const float r1 = v - __uint_as_float(hb);
  const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mb);
  const uint32_t lb = __float_as_uint(r2) & 0xffff0000u;
  const size_t tile = (((size_t)k * cin32 + c) * cout16 + ct) * 3;   // 1 KB planes
  out[(tile + 0) * 512 + lane * 8 + el] = (uint16_t)(hb >> 16);
  out[(tile + 1) * 512 + lane * 8 + el] = (uint16_t)(mb >> 16);
  out[(tile + 2) * 512 + lane * 8 + el] = (uint16_t)(lb >> 16);
}

// which layers run the emulated-fp32 build (and have their weights packed for it): opt-in
static bool conv_emu(int K, int cin, int cout) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("A3D_CONV_EMU");
    on = e ? atoi(e) : 0;
  }
  // on == 1: the 96-column kernels (levels 0 / 1 of the decoder side, 40 % of a step's kernel time); on == 2: every gathered conv
  if (!on || K <= 1 || cin % 32 != 0) return false;
  if (cout == 96) return true;
  return on >= 2 && (cout == 32 || cout == 64 || cout % 128 == 0);
}

// ------------------------------------------------------------------------------ host: launch
constexpr int kMaxQueuesPerOp = 1024;  // ints of zeroed per-op state: k_conv_sk ticket [0], failure word [1], hand-off flags [2..2+G)
constexpr int kSkMaxG = 1020;

// ---- k_conv_sk: launch geometry
struct SkPlan {
  int bn, ch, pair, nchunk, ov, n_cblk, G, ntile;
  size_t lds, slab_floats;
};
static int sk_ch(int cin, int bn) {   // input channels per stage
  // 128-column workgroups: 32-channel stages as well.  Measured on the 16-scene batch against 64-channel stages
  // (profiles/r03_experiments.txt): every <128,*> layer 1-3 % faster, L4 128 -> 256 63 -> 53 us
  if (bn == 128 && cin % 32 == 0) return 32;
  // 96-column workgroups: 32-channel stages -- 157 registers, a 25 KB weight ring: THREE workgroups per CU, the third
  // covers the per-tile prologues / epilogues and the stage barriers of the other two (measured on the 4-scene batch:
  // L0 96 -> 96 700 -> 620 us = 108 TF/s, 128 -> 96 850 -> 790 us = 113 TF/s against 96- / 64-channel stages with two)
  if (bn == 96 && cin % 32 == 0) return 32;
  const int cand[3] = {96, 64, 32};
  for (int i = 0; i < 3; ++i)
    if (cin % cand[i] == 0 && 2 * cand[i] * bn * 4 + 64 <= 80 * 1024) return cand[i];
  return 0;
}
// which build: PAIR = 1 is the low-register one (weight fragments two at a time).  Measured on the 4-scene batch: the
// 96-column kernels gain from it (127 registers -> FOUR workgroups per CU: L0 96 -> 96 640 -> 607 us = 110 TF/s,
// 128 -> 96 816 -> 770 us = 116 TF/s, L1 174 -> 169 us); 64- and 128-column kernels are within 2 % either way
// -- on the 16-scene batch the 128-column kernels with 32-channel stages gain 4-6 % from it on the big levels (L2 128 -> 128
// 303 -> 280 us, L3 256 -> 256 299 -> 283 us) and lose 2 % on L4 (2.5 k rows): by row count
static int sk_pair(int bn, int ch, int n_rows) {
  if (bn == 128) return ch == 32 && n_rows >= 8192;
  return bn == 96;
}
// resident workgroups per CU of k_conv_sk<bn, ch, pair>: registers (hipcc's allocation, -Rpass-analysis=
// kernel-resource-usage; PAIR = 0: 83 / 104 / 122 / 120 / 142 / 160 / 157 / 176 / 196 / 191 / 214 VGPRs) and LDS
static int sk_wgs_per_cu(int bn, int ch, int pair, size_t lds) {
  int by_regs = 2;
  if (pair) {
    by_regs = (bn <= 96 && ch <= 32) ? 4 : 3;
  } else {
    if (bn == 32) by_regs = ch <= 32 ? 5 : 4;
    else if (bn == 64) by_regs = ch <= 32 ? 4 : 3;
    else if (bn == 96) by_regs = ch <= 48 ? 3 : 2;
  }
  const int by_lds = (int)(160 * 1024 / lds);
  const int w = by_regs < by_lds ? by_regs : by_lds;
  return w < 1 ? 1 : (w > 4 ? 4 : w);
}
// force_ch > 0: that stage width instead of sk_ch's choice (a fused projection whose input channels the preferred width does
// not divide: the 32 -> 64 block of level 2 runs its 64 -> 64 conv with 32-channel stages)
static SkPlan plan_sk(int n_rows, int K, int cin, int cout, bool handoff, int force_ch = 0) {
  SkPlan p;
  p.ntile = (n_rows + 63) / 64;
  if (p.ntile < 1) p.ntile = 1;
  const int k_eff = K == 27 ? 13 : K;   // a 3^3 map has 11-17 of its 27 offsets per tile (the kernel uses the exact counts)
  int gmax = 512;
  for (int pass = 0; pass < 2; ++pass) {
    p.bn = (cout % 128 == 0) ? 128 : cout;
    // a level too small to give every CU a share with 128-column workgroups is cut into 64-column ones: twice the
    // shares for the same number of workgroups per tile (the gathers of a tile are repeated, its stages are not)
    if (pass == 1 && cout % 64 == 0 && p.bn > 64) p.bn = 64;
    p.ch = force_ch > 0 && cin % force_ch == 0 ? force_ch : sk_ch(cin, p.bn);
    p.nchunk = p.ch ? cin / p.ch : 1;
    p.n_cblk = cout / p.bn;
    p.pair = sk_pair(p.bn, p.ch, n_rows);
    if (conv_emu(K, cin, cout)) {
      p.ch = 32;
      p.nchunk = cin / 32;
      p.pair = 2;
    }
    const int mfma_per_stage = p.ch / 4 * (p.bn / 16);
    p.ov = mfma_per_stage >= 128 ? 1 : mfma_per_stage >= 64 ? 2 : 3;   // per-tile overhead in stages (swept in round 2)
    p.lds = (size_t)2 * p.ch * p.bn * (p.pair == 2 ? 6 : 4) + 64;   // three bf16 planes: 6 bytes per weight
    gmax = 256 * sk_wgs_per_cu(p.bn, p.ch, p.pair, p.lds);
    if (gmax > kSkMaxG) gmax = kSkMaxG;
    const long long est = (long long)p.n_cblk * p.ntile * ((long long)p.nchunk * k_eff + p.ov);
    const int min_share = mfma_per_stage >= 128 ? 6 : 8;
    if (handoff) {
      long long G = est / min_share;
      // a level too small to hand every CU a share of that size: shorter shares (each workgroup's chain of stages is
      // what the layer waits for; the hand-off of a tile grows with the number of its parts, so not below small_share)
      const int small_share = 4;   // 3 and 5 measured the same, 2 and 8 slower (round 2)
      if (G < 256 && small_share < min_share) {
        G = est / small_share;
        if (G > 256) G = 256;
      }
      p.G = (int)(G < 1 ? 1 : (G > gmax ? gmax : G));
      p.slab_floats = (size_t)p.G * 64 * p.bn;
      if (p.G >= 256 || p.bn <= 64 || cout % 64) break;
    } else {
      const long long nu = (long long)p.n_cblk * p.ntile;
      p.G = (int)(nu < gmax ? nu : gmax);
      p.slab_floats = 0;
      break;
    }
  }
  return p;
}

// ---- k_conv_deep: which small layers it takes, and their geometry
// The model behind the choice (us; fitted to per-workgroup phase timestamps and ablation builds on the levels of one 80 k-voxel
// scene, profiles/r05_experiments.txt): a workgroup's life = 2.6 (masks + table, then the first slices' round trip) + its
// stages x (0.3 skeleton + 0.04 per LDS-DMA instruction of a wave + the stage's MFMA time: the DMA issue does not hide behind
// the MFMAs of a lone wave) + the hand-off of a unit cut into P parts (1.0 publish + 0.3-0.45 per part read by the last
// arriver: round trips to write-through data of other XCDs, 65 GB/s into one CU); one workgroup per CU: layers that need more than 256 are left to the stream-K kernel.  A 64-row tile issues the UNION of its rows' offsets: 22 of 27
// on levels of up to a few thousand rows, 14-17 on the large ones (rows sorted by neighbour pattern).
// The kernel is preferred while the layer's matrix work per CU stays below A3D_DEEP_MAX_US (18): beyond that the stream-K
// kernel's two workgroups per CU win (measured: level 3 384 -> 256 a tie at 23 us of work per CU, level 2 192 -> 128 lost).
// A3D_CONV_DEEP=0 switches the kernel off (A/B).
struct DeepPlan {
  bool use;
  int bn, ch, P, G, n_cblk, ntile, nchunk, D;
  size_t lds, slab_floats;
  float est_us;
};
static int g_deep_mode = -1;
static int deep_mode() {
  if (g_deep_mode < 0) {
    const char* e = getenv("A3D_CONV_DEEP");
    g_deep_mode = e ? atoi(e) : 1;
  }
  return g_deep_mode;
}
// `up`: a transposed 2^3 layer -- its fine rows are sorted by child slot, a tile has one or two of the eight offsets
static DeepPlan plan_deep(int n_rows, int K, int cin, int cout, int cin2, bool up = false) {
  DeepPlan best;
  memset(&best, 0, sizeof(best));
  if (!deep_mode() || K < 2 || K > 27 || cin % 32 || cout % 32 || conv_emu(K, cin, cout)) return best;
  static float max_us = -1.f;
  if (max_us < 0.f) {
    const char* e = getenv("A3D_DEEP_MAX_US");
    max_us = e ? (float)atof(e) : 23.f;
  }
  const int ntile = n_rows > 0 ? (n_rows + 63) / 64 : 1;
  const int k_eff = K == 27 ? (n_rows < 8000 ? 22 : n_rows < 32000 ? 17 : 14) : up ? 2 : K;
  static const int cand[5][2] = {{128, 32}, {64, 64}, {64, 32}, {32, 64}, {32, 32}};
  auto lds_of = [](int bn, int ch, int D) -> size_t { return (size_t)D * ((ch / 16) * (bn / 16) + 4 * (ch / 16)) * 1024 + 28 * 64 * 4 + 64; };
  float best_t = 1e30f;
  for (int ci = 0; ci < 5; ++ci) {
    const int bn = cand[ci][0], ch = cand[ci][1];
    if (cout % bn || cin % ch || (cin2 > 0 && cin2 % ch)) continue;
    const int units = cout / bn * ntile;
    if (units > kMaxQueuesPerOp - 2) continue;
    const int S = k_eff * (cin / ch) + (cin2 > 0 ? cin2 / ch : 0);
    const int ns = ch / 16, nct = bn / 16;
    const float mf = (float)ns * (float)nct * 0.0533f;
    const float tstage = (ns < 4 ? 0.45f : 0.3f) + 0.04f * (float)(ns + ns * nct / 4) + mf;   // two-step stages expose their LDS reads
    if ((float)units * (float)S * mf / 256.f > max_us) continue;   // matrix-bound: the stream-K kernel's ground
    for (int D = 3; D >= 2; --D) {
      // three ring slots, or two when that lets a second workgroup onto the CU (its waves fill the issue slots the first
      // one's DMA instructions and barriers leave: measured 10-15 % on the 128- / 256-channel layers of levels 2 / 3)
      const int per_cu = 160 * 1024 / lds_of(bn, ch, D) >= 2 ? 2 : 1;
      if (D == 2 && (per_cu < 2 || 160 * 1024 / lds_of(bn, ch, 3) >= 2)) continue;
      for (int P = 1; P <= 24; ++P) {
        if (P > 1 && S / P < 2) break;
        const int G = units * P;
        if (G > 256 * per_cu) break;   // every workgroup resident at once: a second round would start when the first ends
        const int stages = (S + P - 1) / P;
        const float ts = G > 256 ? fmaxf(tstage + 0.1f, 2.2f * mf + 0.3f) : tstage + (D == 2 ? 0.05f : 0.f);   // two on a CU share its matrix cores
        const float t = 2.6f + (float)stages * ts + (P > 1 ? 1.0f + (float)P * (0.2f + 0.12f * bn / 32.f) * (G > 256 ? 1.5f : 1.f) : 0.f);
        if (t < best_t) {
          best_t = t;
          best.bn = bn, best.ch = ch, best.P = P, best.G = G, best.n_cblk = cout / bn, best.ntile = ntile, best.nchunk = cin / ch;
          best.D = D;
        }
      }
    }
  }
  // mode >= 16 (experiments, tools/conv_bench.py --sweep): bits 0-3 = bn / 32, bits 4-7 = ch / 32, bits 8-15 = P (0: 256 / units),
  // bit 19 = two ring slots instead of three: that geometry wherever it divides the layer
  int force[3] = {0, 0, 0};
  if (deep_mode() >= 16) force[0] = (deep_mode() & 15) * 32, force[1] = ((deep_mode() >> 4) & 15) * 32, force[2] = (deep_mode() >> 8) & 255;
  if (force[0] > 0 && force[1] > 0 && cout % force[0] == 0 && cin % force[1] == 0 && (cin2 <= 0 || cin2 % force[1] == 0)) {
    const int bn = force[0], ch = force[1], units = cout / bn * ntile;
    const int S = k_eff * (cin / ch) + (cin2 > 0 ? cin2 / ch : 0);
    int P = force[2];
    if (P <= 0) {
      P = 256 / units;
      if (P > S / 2) P = S / 2;
    }
    if (P < 1) P = 1;
    if (units * P <= kSkMaxG && units <= kMaxQueuesPerOp - 2) {
      best.bn = bn, best.ch = ch, best.P = P, best.G = units * P, best.n_cblk = cout / bn, best.ntile = ntile, best.nchunk = cin / ch;
      best.D = ((deep_mode() >> 19) & 1) ? 2 : 3;
      best_t = 0.f;
    }
  }
  if (best_t > 1e29f) return best;
  best.use = true;
  best.est_us = best_t;
  best.lds = lds_of(best.bn, best.ch, best.D);
  best.slab_floats = best.P > 1 ? (size_t)best.G * 64 * best.bn : 0;
  return best;
}

static void allow_big_lds() {
  static bool done = false;
  if (done) return;
  done = true;
#define A3D_BIG3(BN_, CH_) \
  A3D_ALLOW_LDS(160 * 1024, k_conv_sk<BN_, CH_, 0>); \
  A3D_ALLOW_LDS(160 * 1024, k_conv_sk<BN_, CH_, 1>); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_sk<BN_, CH_, 0, false, 1>)); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_sk<BN_, CH_, 1, false, 1>)); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_sk<BN_, CH_, 0, false, 2>)); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_sk<BN_, CH_, 1, false, 2>));
  A3D_BIG3(32, 32) A3D_BIG3(32, 64) A3D_BIG3(32, 96) A3D_BIG3(64, 32) A3D_BIG3(64, 64) A3D_BIG3(64, 96)
  A3D_BIG3(96, 32) A3D_BIG3(96, 48) A3D_BIG3(96, 64) A3D_BIG3(96, 96) A3D_BIG3(128, 32) A3D_BIG3(128, 64)
#undef A3D_BIG3
#define A3D_BIGF(BN_, CH_) \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_sk<BN_, CH_, 0, true>)); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_sk<BN_, CH_, 1, true>));
  A3D_BIGF(64, 32) A3D_BIGF(64, 64) A3D_BIGF(96, 32) A3D_BIGF(128, 32)
#undef A3D_BIGF
  A3D_ALLOW_LDS(160 * 1024, (k_conv_sk<96, 32, 1, false, 0, true>));
  A3D_ALLOW_LDS(160 * 1024, k_conv_sk<96, 32, 2>);
  A3D_ALLOW_LDS(160 * 1024, k_conv_sk<128, 32, 2>);
  A3D_ALLOW_LDS(160 * 1024, k_conv_sk<64, 32, 2>);
  A3D_ALLOW_LDS(160 * 1024, k_conv_sk<32, 32, 2>);
#define A3D_BIGD(BN_, CH_) \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_deep<BN_, CH_, false, 3>)); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_deep<BN_, CH_, true, 3>)); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_deep<BN_, CH_, false, 2>)); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_deep<BN_, CH_, true, 2>)); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_deep<BN_, CH_, false, 3, 1>)); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_deep<BN_, CH_, false, 2, 1>)); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_deep<BN_, CH_, false, 3, 2>)); \
  A3D_ALLOW_LDS(160 * 1024, (k_conv_deep<BN_, CH_, false, 2, 2>));
  A3D_BIGD(128, 32) A3D_BIGD(64, 64) A3D_BIGD(64, 32) A3D_BIGD(32, 64) A3D_BIGD(32, 32)
#undef A3D_BIGD
  A3D_ALLOW_LDS(160 * 1024, k_conv_wl<2, 2>);
  A3D_ALLOW_LDS(160 * 1024, k_conv_wl<4, 4>);
  A3D_ALLOW_LDS(160 * 1024, (k_conv_wl<2, 2, 1>));
  A3D_ALLOW_LDS(160 * 1024, (k_conv_wl<2, 2, 2>));
  A3D_ALLOW_LDS(160 * 1024, (k_dense<8, 8, false>));
  A3D_ALLOW_LDS(160 * 1024, (k_dense<8, 6, false>));
  A3D_ALLOW_LDS(160 * 1024, (k_dense<6, 8, false>));
  A3D_ALLOW_LDS(160 * 1024, (k_dense<6, 6, false>));
  A3D_ALLOW_LDS(160 * 1024, (k_dense<8, 8, true>));
  A3D_ALLOW_LDS(160 * 1024, (k_dense<8, 6, true>));
  A3D_ALLOW_LDS(160 * 1024, (k_dense<6, 8, true>));
  A3D_ALLOW_LDS(160 * 1024, (k_dense<6, 6, true>));
}

// does a fused-projection build of k_conv_sk exist for what plan_sk picks at this row count?  (the stage width and the
// column block depend on the rows; a3d_program_run falls back to conv + separate 1x1 launch when not)
static bool sk_fused_ok(int n_rows, int cin, int cout, int cin2) {
  if (cin % 32 != 0 || cout % 16 != 0 || !(cout % 128 == 0 || cout == 32 || cout == 64 || cout == 96)) return false;
  if (cin2 <= 0) return false;
  SkPlan p = plan_sk(n_rows, 27, cin, cout, true);
  if (p.ch && cin2 % p.ch != 0 && cin2 % 32 == 0) p = plan_sk(n_rows, 27, cin, cout, true, 32);
  if (!p.ch || p.pair == 2 || cin2 % p.ch != 0) return false;
  return (p.bn == 64 && (p.ch == 32 || p.ch == 64)) || (p.bn == 96 && p.ch == 32) || (p.bn == 128 && p.ch == 32);
}

// stats != nullptr (training): the kernel's epilogue also writes BatchNorm partials, [blocks][2][stats_ld] with
// *stats_rows rows per block (64: k_conv_sk tiles, 16: the groups of k_conv_wl); the op must carry no scale / shift / residual
// is there a fused-head build for what plan_sk picks here?  (a3d_program_run runs the 1x1 layer on its own otherwise)
static bool sk_head_ok(int n_rows, int K, int cin, int cout, int head_cout, bool handoff) {
  if (cout != 96 || head_cout <= 0 || head_cout % 16 || head_cout > 256 || cin % 32) return false;
  if (conv_emu(K, cin, cout)) return false;
  const SkPlan p = plan_sk(n_rows, K, cin, cout, handoff);
  return p.bn == 96 && p.ch == 32 && p.pair == 1;
}

// bw != nullptr (with stats): the BatchNorm-BACKWARD sums instead (STATS == 2: the op is an input-gradient conv, a residual =
// the gradient already accumulated in the buffer is allowed)
static int launch_conv_sk(ConvArgs c, const int* pre64, float* slab_ws, size_t slab_ws_floats, int* state,
                          hipStream_t st, float* stats = nullptr, int stats_ld = 0, int* stats_rows = nullptr,
                          const BwArgs* bw = nullptr) {
  allow_big_lds();
  if (stats && (c.scale || c.shift || c.relu || c.cin2 > 0 || conv_emu(c.K, c.cin, c.cout) || !stats_rows || (c.res && !bw))) {
    set_error("spconv: BatchNorm statistics are taken of a raw convolution (no epilogue, no fused projection, exact fp32)");
    return A3D_ERR_UNSUPPORTED;
  }
  if (c.cin % 32 != 0 || c.cout % 16 != 0 || !(c.cout % 128 == 0 || c.cout == 32 || c.cout == 64 || c.cout == 96)) {
    set_error("spconv: unsupported channels cin=%d cout=%d", c.cin, c.cout);
    return A3D_ERR_UNSUPPORTED;
  }
  if (c.K > 27) {
    set_error("spconv: kernel volume %d > 27", c.K);
    return A3D_ERR_UNSUPPORTED;
  }
  if ((uint64_t)(c.n_in + 1) * (uint64_t)c.ldi * 4ull >= (1ull << 32)) {
    set_error("spconv: input of %d rows x %d floats exceeds the 4 GB gather window", c.n_in, c.ldi);
    return A3D_ERR_UNSUPPORTED;
  }
  if (c.cin2 == 0 && conv_wl_supported(c) && !conv_emu(c.K, c.cin, c.cout)) {
    if (stats) *stats_rows = 16;
    return launch_conv_wl(c, st, stats, stats_ld, bw);
  }
  if (state && c.head_cout == 0 && c.K > 1) {
    const DeepPlan d = plan_deep(c.n_out, c.K, c.cin, c.cout, c.cin2, c.tag_table == A3D_OP_UP);
    if (d.use && d.slab_floats <= slab_ws_floats && (c.cin2 == 0 || (c.K == 27 && c.in2 && !(c.ldi2 & 3) &&
        (uint64_t)(c.n_out + 1) * (uint64_t)c.ldi2 * 4ull < (1ull << 32)))) {
      DeepArgs a;
      memset(&a, 0, sizeof(a));
      c.n_tiles = d.ntile;
      a.c = c;
      a.P = d.P;
      a.n_cblk = d.n_cblk;
      a.nchunk = d.nchunk;
      a.nchunk2 = c.cin2 > 0 ? c.cin2 / d.ch : 0;
      a.cnt = state + 2;
      a.slab = slab_ws;
      a.in_row_bytes = (unsigned)c.ldi * 4u;
      a.in2_row_bytes = (unsigned)c.ldi2 * 4u;
      size_t lds = d.lds;
      if (stats) {   // training: the last arriver's epilogue writes the tile's BatchNorm partials (k_conv_sk's layout and orders)
        a.stats = stats;
        a.stats_ld = stats_ld;
        *stats_rows = 64;
        if (bw) a.bw = *bw;
        lds += (size_t)8 * d.bn * 4;
      }
      // the kernel-volume field as k_conv_sk's (K | cin2 << 8); the column field carries 1000 + BN: "deep" in the layer tables
      ProfScope ps(st, A3D_PROF_SPCONV, 1000 + d.bn, c.K | (c.cin2 << 8), c.cin, c.cout, c.n_out, c.tag_table, c.tag_level, d.ch);
      // A3D_DEEP_DBG=1 (tuning): per-workgroup phase timestamps of every launch, summarised on stderr (synchronises the stream)
      static int dbg_on = -1;
      static unsigned long long* dbg_buf = nullptr;
      if (dbg_on < 0) {
        const char* e = getenv("A3D_DEEP_DBG");
        dbg_on = e ? atoi(e) : 0;
        if (dbg_on && hipMalloc(&dbg_buf, (size_t)(kSkMaxG * 6 + 2) * 8) != hipSuccess) dbg_on = 0;
      }
      if (dbg_on) {
        a.dbg = dbg_buf;
        (void)hipMemsetAsync(dbg_buf, 0, (size_t)kSkMaxG * 6 * 8, st);
      }
      if (stats) {
        const bool two = d.D == 2;
#define A3D_LT(BN_, CH_) \
  if (d.bn == BN_ && d.ch == CH_) { \
    if (bw) { if (two) k_conv_deep<BN_, CH_, false, 2, 2><<<d.G, 256, lds, st>>>(a); else k_conv_deep<BN_, CH_, false, 3, 2><<<d.G, 256, lds, st>>>(a); } \
    else { if (two) k_conv_deep<BN_, CH_, false, 2, 1><<<d.G, 256, lds, st>>>(a); else k_conv_deep<BN_, CH_, false, 3, 1><<<d.G, 256, lds, st>>>(a); } \
  } else
        A3D_LT(128, 32) A3D_LT(64, 64) A3D_LT(64, 32) A3D_LT(32, 64) A3D_LT(32, 32)
        { set_error("spconv: no deep kernel for BN %d CH %d", d.bn, d.ch); return A3D_ERR_UNSUPPORTED; }
#undef A3D_LT
      } else {
        const bool two = d.D == 2;
#define A3D_LD(BN_, CH_) \
  if (d.bn == BN_ && d.ch == CH_) { \
    if (c.cin2 > 0) { if (two) k_conv_deep<BN_, CH_, true, 2><<<d.G, 256, d.lds, st>>>(a); else k_conv_deep<BN_, CH_, true, 3><<<d.G, 256, d.lds, st>>>(a); } \
    else { if (two) k_conv_deep<BN_, CH_, false, 2><<<d.G, 256, d.lds, st>>>(a); else k_conv_deep<BN_, CH_, false, 3><<<d.G, 256, d.lds, st>>>(a); } \
  } else
        A3D_LD(128, 32) A3D_LD(64, 64) A3D_LD(64, 32) A3D_LD(32, 64) A3D_LD(32, 32)
        { set_error("spconv: no deep kernel for BN %d CH %d", d.bn, d.ch); return A3D_ERR_UNSUPPORTED; }
#undef A3D_LD
      }
      A3D_LAUNCH_CHECK();
      if (dbg_on) {
        static unsigned long long host[kSkMaxG * 6 + 2];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(host, dbg_buf, (size_t)(d.G * 6 + 2) * 8, hipMemcpyDeviceToHost);
        unsigned long long t0min = ~0ull, tend = 0;
        for (int w = 0; w < d.G; ++w) {
          if (host[w * 6] < t0min) t0min = host[w * 6];
          for (int q = 0; q < 6; ++q) if (host[w * 6 + q] > tend) tend = host[w * 6 + q];
        }
        double sum[6] = {0, 0, 0, 0, 0, 0}, mx[6] = {0, 0, 0, 0, 0, 0};
        int cntq[6] = {0, 0, 0, 0, 0, 0};
        for (int w = 0; w < d.G; ++w)
          for (int q = 0; q < 6; ++q)
            if (host[w * 6 + q]) {
              const double v = (double)(host[w * 6 + q] - t0min) * 0.01;
              sum[q] += v, ++cntq[q];
              if (v > mx[q]) mx[q] = v;
            }
        fprintf(stderr, "deep<%d,%d,D%d>%s K=%d %d->%d rows=%d units=%d P=%d G=%d span %.2f us | avg/max since first start:", d.bn, d.ch, d.D,
                c.cin2 > 0 ? "+p" : "", c.K, c.cin, c.cout, c.n_out, d.G / d.P, d.P, d.G, (double)(tend - t0min) * 0.01);
        const char* nm[6] = {"start", "table", "stage0", "loop", "publish", "end"};
        for (int q = 0; q < 6; ++q)
          fprintf(stderr, " %s %.2f/%.2f", nm[q], cntq[q] ? sum[q] / cntq[q] : 0.0, mx[q]);
        if (host[3] > host[0])   // workgroup 0: shader-clock ticks per 10 ns tick between its start and the end of its loop
          fprintf(stderr, " | clock %.0f MHz", (double)(host[d.G * 6 + 1] - host[d.G * 6]) / (double)(host[3] - host[0]) * 100.0);
        fprintf(stderr, "\n");
      }
      return A3D_OK;
    }
  }
  // hand-offs (shares cut inside tiles) pay where a tile is long and tiles are few: the 3^3 maps, and the 2^3 maps of
  // the small levels.  1x1 layers and 2^3 maps with a tile per workgroup slot or more run whole tiles: no ticket, no
  // search, no flags -- their fixed latency is what matters
  bool handoff = state != nullptr && c.K > 1;
  if (handoff && c.K <= 8) {
    const SkPlan q = plan_sk(c.n_out, c.K, c.cin, c.cout, false);
    if ((long long)q.ntile * q.n_cblk >= 192) handoff = false;
  }
  SkPlan p = plan_sk(c.n_out, c.K, c.cin, c.cout, handoff);
  if (c.cin2 > 0 && p.ch && c.cin2 % p.ch != 0 && c.cin2 % 32 == 0) p = plan_sk(c.n_out, c.K, c.cin, c.cout, handoff, 32);
  if (!p.ch) {
    set_error("spconv: no stage size for cin=%d with %d-column workgroups", c.cin, p.bn);
    return A3D_ERR_UNSUPPORTED;
  }
  if (handoff && p.slab_floats > slab_ws_floats) {
    set_error("spconv: hand-off workspace too small");
    return A3D_ERR_WORKSPACE;
  }
  SkArgs a;
  memset(&a, 0, sizeof(a));
  c.n_tiles = p.ntile;
  a.c = c;
  const int* pre = pre64;
  a.pre = c.K > 1 ? pre : nullptr;
  a.nchunk = p.nchunk;
  a.ov = p.ov;
  a.n_cblk = p.n_cblk;
  a.G = p.G;
  a.ticket = handoff ? state : nullptr;
  a.fail = handoff ? state + 1 : nullptr;
  a.flags = handoff ? (unsigned*)(state + 2) : nullptr;
  a.slab = slab_ws;
  a.in_row_bytes = (unsigned)c.ldi * 4u;
  a.prio = handoff && p.G > 256;   // static wave priorities: measured +1 % on the 96-column layers (L0 593 -> 588 us)
  if (c.K > 1 && !pre) {
    set_error("spconv: a gathered convolution needs the scene's tile prefix table");
    return A3D_ERR_INVALID;
  }
  // kernel-volume field: K, plus the fused projection's input channels in the bits above (cin2 << 8); last field: stage width
  // ... and a fused head's output columns above those (head_cout << 20)
  ProfScope ps(st, A3D_PROF_SPCONV, p.bn, c.K | (c.cin2 << 8) | (c.head_cout << 20), c.cin, c.cout, c.n_out, c.tag_table, c.tag_level,
               p.ch);
  if (c.cin2 > 0) {
    // fused residual projection: the exact-fp32 builds of the shapes the U-Net's second block convs run on
    if (c.K != 27 || !c.in2 || p.pair == 2 || c.cin2 % p.ch != 0 || (c.ldi2 & 3) ||
        (uint64_t)(c.n_out + 1) * (uint64_t)c.ldi2 * 4ull >= (1ull << 32)) {
      set_error("spconv: fused projection unsupported here (K=%d cin2=%d stage %d emulated=%d)", c.K, c.cin2, p.ch, p.pair == 2);
      return A3D_ERR_UNSUPPORTED;
    }
    a.nchunk2 = c.cin2 / p.ch;
    a.in2_row_bytes = (unsigned)c.ldi2 * 4u;
#define A3D_LF(BN_, CH_) \
  if (p.bn == BN_ && p.ch == CH_) { if (p.pair) k_conv_sk<BN_, CH_, 1, true><<<p.G, 256, p.lds, st>>>(a); else k_conv_sk<BN_, CH_, 0, true><<<p.G, 256, p.lds, st>>>(a); } else
    A3D_LF(64, 32) A3D_LF(64, 64) A3D_LF(96, 32) A3D_LF(128, 32)
    { set_error("spconv: no fused kernel for BN %d CH %d", p.bn, p.ch); return A3D_ERR_UNSUPPORTED; }
#undef A3D_LF
    A3D_LAUNCH_CHECK();
    return A3D_OK;
  }
  if (p.pair == 2) {
    if (p.bn == 96) k_conv_sk<96, 32, 2><<<p.G, 256, p.lds, st>>>(a);
    else if (p.bn == 128) k_conv_sk<128, 32, 2><<<p.G, 256, p.lds, st>>>(a);
    else if (p.bn == 64) k_conv_sk<64, 32, 2><<<p.G, 256, p.lds, st>>>(a);
    else k_conv_sk<32, 32, 2><<<p.G, 256, p.lds, st>>>(a);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
  }
  if (c.head_cout > 0) {
    if (!(p.bn == 96 && p.ch == 32 && p.pair == 1) || c.cin2 > 0 || stats || !c.head_w || !c.head_out) {
      set_error("spconv: no fused-head build for BN %d CH %d pair %d", p.bn, p.ch, p.pair);
      return A3D_ERR_UNSUPPORTED;
    }
    k_conv_sk<96, 32, 1, false, 0, true><<<p.G, 256, p.lds, st>>>(a);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
  }
  if (stats) {
    a.stats = stats;
    a.stats_ld = stats_ld;
    *stats_rows = 64;
    const size_t lds_s = p.lds + (size_t)2 * 4 * p.bn * 4;
    if (bw) {
      a.bw_y = bw->y, a.bw_raw = bw->raw, a.bw_mean = bw->mean, a.bw_rstd = bw->rstd;
      a.bw_ldy = bw->ldy, a.bw_ldraw = bw->ldraw, a.bw_relu = bw->relu;
#define A3D_LB(BN_, CH_) \
  if (p.bn == BN_ && p.ch == CH_) { if (p.pair) k_conv_sk<BN_, CH_, 1, false, 2><<<p.G, 256, lds_s, st>>>(a); else k_conv_sk<BN_, CH_, 0, false, 2><<<p.G, 256, lds_s, st>>>(a); } else
      A3D_LB(32, 32) A3D_LB(32, 64) A3D_LB(32, 96) A3D_LB(64, 32) A3D_LB(64, 64) A3D_LB(64, 96)
      A3D_LB(96, 32) A3D_LB(96, 48) A3D_LB(96, 64) A3D_LB(96, 96) A3D_LB(128, 32) A3D_LB(128, 64)
      { set_error("spconv: no kernel for BN %d CH %d", p.bn, p.ch); return A3D_ERR_UNSUPPORTED; }
#undef A3D_LB
      A3D_LAUNCH_CHECK();
      return A3D_OK;
    }
#define A3D_LS(BN_, CH_) \
  if (p.bn == BN_ && p.ch == CH_) { if (p.pair) k_conv_sk<BN_, CH_, 1, false, 1><<<p.G, 256, lds_s, st>>>(a); else k_conv_sk<BN_, CH_, 0, false, 1><<<p.G, 256, lds_s, st>>>(a); } else
    A3D_LS(32, 32) A3D_LS(32, 64) A3D_LS(32, 96) A3D_LS(64, 32) A3D_LS(64, 64) A3D_LS(64, 96)
    A3D_LS(96, 32) A3D_LS(96, 48) A3D_LS(96, 64) A3D_LS(96, 96) A3D_LS(128, 32) A3D_LS(128, 64)
    { set_error("spconv: no kernel for BN %d CH %d", p.bn, p.ch); return A3D_ERR_UNSUPPORTED; }
#undef A3D_LS
    A3D_LAUNCH_CHECK();
    return A3D_OK;
  }
#define A3D_L3(BN_, CH_) \
  if (p.bn == BN_ && p.ch == CH_) { if (p.pair) k_conv_sk<BN_, CH_, 1><<<p.G, 256, p.lds, st>>>(a); else k_conv_sk<BN_, CH_, 0><<<p.G, 256, p.lds, st>>>(a); } else
  A3D_L3(32, 32) A3D_L3(32, 64) A3D_L3(32, 96) A3D_L3(64, 32) A3D_L3(64, 64) A3D_L3(64, 96)
  A3D_L3(96, 32) A3D_L3(96, 48) A3D_L3(96, 64) A3D_L3(96, 96) A3D_L3(128, 32) A3D_L3(128, 64)
  { set_error("spconv: no kernel for BN %d CH %d", p.bn, p.ch); return A3D_ERR_UNSUPPORTED; }
#undef A3D_L3
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

// ------------------------------------------------------------------------------ program
struct ProgLayout {
  size_t buf_off[64];
  size_t feats4_off, partial_off, partial_floats, queue_off, total;
};

static int layout_program(const a3d_scene* s, const a3d_buf_desc* bufs, int n_bufs, const a3d_op* ops,
                          int n_ops, ProgLayout& L) {
  if (!s || n_bufs < 0 || n_bufs > 64 || (n_bufs && !bufs)) {
    set_error("program: bad buffer list");
    return A3D_ERR_INVALID;
  }
  size_t off = 0;
  for (int i = 0; i < n_bufs; ++i) {
    if (bufs[i].level < 0 || bufs[i].level >= A3D_NUM_LEVELS || bufs[i].channels <= 0 || bufs[i].channels % 4) {
      set_error("program: bad buffer %d", i);
      return A3D_ERR_INVALID;
    }
    L.buf_off[i] = off;
    off += align256((size_t)(s->lv[bufs[i].level].n + 1) * bufs[i].channels * 4);
  }
  L.feats4_off = off;
  off += align256((size_t)s->lv[0].npad * 16);
  size_t pf = 0;
  for (int i = 0; i < n_ops; ++i) {
    const a3d_op& o = ops[i];
    if (o.kind == A3D_OP_STEM) continue;
    int lvl_out = o.level_in;
    if (o.kind == A3D_OP_DOWN) lvl_out = o.level_in + 1;
    if (o.kind == A3D_OP_UP) lvl_out = o.level_in - 1;
    if (lvl_out < 0 || lvl_out >= A3D_NUM_LEVELS) {
      set_error("program: op %d leaves the level range", i);
      return A3D_ERR_INVALID;
    }
    SkPlan q = plan_sk(s->lv[lvl_out].n, o.kernel_volume, o.cin, o.cout, true);
    if (q.slab_floats > pf) pf = q.slab_floats;
    if (o.proj_cin > 0) {   // the fused projection may run with 32-channel stages (launch_conv_sk): another share count
      q = plan_sk(s->lv[lvl_out].n, o.kernel_volume, o.cin, o.cout, true, 32);
      if (q.slab_floats > pf) pf = q.slab_floats;
    }
    const DeepPlan dp = plan_deep(s->lv[lvl_out].n, o.kernel_volume, o.cin, o.cout, o.proj_cin, o.kind == A3D_OP_UP);
    if (dp.use && dp.slab_floats > pf) pf = dp.slab_floats;
  }
  L.partial_off = off;
  L.partial_floats = pf;
  off += align256(pf * 4);
  L.queue_off = off;
  off += align256((size_t)(n_ops > 0 ? n_ops : 1) * kMaxQueuesPerOp * 4);
  L.total = off;
  return A3D_OK;
}

}  // namespace a3d

using namespace a3d;

extern "C" int a3d_pack_conv_weight(const float* w_dev, int kernel_volume, int cin, int cout,
                                    float* packed_dev, void* stream) {
  if (!w_dev || !packed_dev || kernel_volume < 1 || cin % 16 || cout % 16) {
    set_error("a3d_pack_conv_weight: bad arguments (K=%d cin=%d cout=%d)", kernel_volume, cin, cout);
    return A3D_ERR_INVALID;
  }
  const size_t total = (size_t)kernel_volume * cin * cout;
  if (conv_emu(kernel_volume, cin, cout))
    k_pack_weight_emu<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(w_dev, kernel_volume, cin, cout,
                                                                                      (uint16_t*)packed_dev);
  else
    k_pack_weight<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(w_dev, kernel_volume, cin, cout, packed_dev);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

// every job's pack in one launch: a workgroup takes one 4096-element chunk, its job found by binary search (as optim.hip)
__global__ void __launch_bounds__(256) k_pack_weight_multi(const a3d_pack_job* __restrict__ tab, int nj) {
  int lo = 0, hi = nj;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tab[mid].chunk0 <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const a3d_pack_job jb = tab[lo];
  const size_t total = (size_t)jb.K * jb.cin * jb.cout;
  const size_t base = (size_t)((int)blockIdx.x - jb.chunk0) * A3D_MT_CHUNK;
  const int cout16 = jb.cout >> 4, cin16 = jb.cin >> 4;
  for (int i = threadIdx.x; i < A3D_MT_CHUNK; i += 256) {
    const size_t e = base + i;
    if (e >= total) break;
    const int t = (int)(e & 3), lane = (int)((e >> 2) & 63);
    size_t rest = e >> 8;
    const int ct = (int)(rest % cout16);
    rest /= cout16;
    const int S = (int)(rest % cin16);
    const int k = (int)(rest / cin16);
    const int ci = 16 * S + 4 * (lane >> 4) + t, co = 16 * ct + (lane & 15);
    size_t si;
    if (jb.transposed) si = ((size_t)(jb.flip ? jb.K - 1 - k : k) * jb.src_cin + jb.c0 + co) * jb.src_cout + ci;
    else si = ((size_t)k * jb.src_cin + ci) * jb.src_cout + co;
    jb.dst[e] = jb.src[si];
  }
}

extern "C" int a3d_pack_conv_weights_multi(const a3d_pack_job* table_dev, int n_jobs, int64_t n_chunks, void* stream) {
  if (!table_dev || n_jobs < 1 || n_chunks < 1 || n_chunks > (int64_t)1 << 30) {
    set_error("a3d_pack_conv_weights_multi: bad arguments");
    return A3D_ERR_INVALID;
  }
  k_pack_weight_multi<<<(unsigned)n_chunks, 256, 0, (hipStream_t)stream>>>(table_dev, n_jobs);
  A3D_LAUNCH_CHECK();
  return A3D_OK;
}

extern "C" size_t a3d_conv_weight_packed_floats(int kernel_volume, int cin, int cout) {
  const size_t total = (size_t)kernel_volume * cin * cout;
  return conv_emu(kernel_volume, cin, cout) ? total + total / 2 : total;   // three bf16 planes = 1.5 floats per weight
}

extern "C" size_t a3d_program_workspace_bytes(const a3d_scene* s, const a3d_buf_desc* bufs, int n_bufs,
                                              const a3d_op* ops, int n_ops) {
  ProgLayout L;
  if (layout_program(s, bufs, n_bufs, ops, n_ops, L) != A3D_OK) return 0;
  return L.total + 256;
}

extern "C" size_t a3d_program_buffer_offset(const a3d_scene* s, const a3d_buf_desc* bufs, int n_bufs, int i) {
  ProgLayout L;
  if (i < 0 || i >= n_bufs || layout_program(s, bufs, n_bufs, nullptr, 0, L) != A3D_OK) return (size_t)-1;
  return L.buf_off[i];
}

extern "C" int a3d_program_run(const a3d_scene* s, const a3d_buf_desc* bufs, int n_bufs, const a3d_op* ops,
                               int n_ops, const float* feats3_dev, float* ext_out_dev, int ext_out_ld,
                               void* workspace_dev, size_t workspace_bytes, void* stream) {
  ProgLayout L;
  int rc = layout_program(s, bufs, n_bufs, ops, n_ops, L);
  if (rc != A3D_OK) return rc;
  if (!workspace_dev || workspace_bytes < L.total || ((uintptr_t)workspace_dev & 255)) {
    set_error("a3d_program_run: workspace too small or misaligned (%zu < %zu)", workspace_bytes, L.total);
    return A3D_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace_dev;
  f32x4* feats4 = (f32x4*)(ws + L.feats4_off);
  float* partial = (float*)(ws + L.partial_off);
  int* queues = (int*)(ws + L.queue_off);
  A3D_HIP_CHECK(hipMemsetAsync(queues, 0, (size_t)n_ops * kMaxQueuesPerOp * 4, st));
  bool feats_ready = false;
  auto buf_ptr = [&](int id) -> float* { return (float*)(ws + L.buf_off[id]); };

  for (int i = 0; i < n_ops; ++i) {
    const a3d_op& o = ops[i];
    const int Lin = o.level_in;
    if (Lin < 0 || Lin >= A3D_NUM_LEVELS) {
      set_error("op %d: bad level", i);
      return A3D_ERR_INVALID;
    }
    // ---- output placement
    float* out = nullptr;
    int ldo = 0;
    const int* out_map = nullptr;
    int zero_row = -1;
    int lvl_out = Lin;
    if (o.kind == A3D_OP_DOWN) lvl_out = Lin + 1;
    if (o.kind == A3D_OP_UP) lvl_out = Lin - 1;
    if (o.out_buf == A3D_BUF_EXT_OUT) {
      if (!ext_out_dev || lvl_out != 0 || o.kind == A3D_OP_UP) {
        set_error("op %d: external output needs a level-0, non-transposed op", i);
        return A3D_ERR_INVALID;
      }
      out = ext_out_dev + o.out_coff;
      ldo = ext_out_ld;
      out_map = s->orig_row;
    } else {
      if (o.out_buf < 0 || o.out_buf >= n_bufs || bufs[o.out_buf].level != lvl_out ||
          o.out_coff + o.cout > bufs[o.out_buf].channels) {
        set_error("op %d: bad output buffer", i);
        return A3D_ERR_INVALID;
      }
      out = buf_ptr(o.out_buf) + o.out_coff;
      ldo = bufs[o.out_buf].channels;
      zero_row = s->lv[lvl_out].n;
    }
    const float* res = nullptr;
    int ldr = 0;
    if (o.res_buf != A3D_BUF_NONE) {
      if (o.res_buf < 0 || o.res_buf >= n_bufs || bufs[o.res_buf].level != lvl_out ||
          o.res_coff + o.cout > bufs[o.res_buf].channels || o.out_buf == A3D_BUF_EXT_OUT) {
        set_error("op %d: bad residual buffer", i);
        return A3D_ERR_INVALID;
      }
      res = buf_ptr(o.res_buf) + o.res_coff;
      ldr = bufs[o.res_buf].channels;
    }

    if (o.kind == A3D_OP_STEM) {
      if (Lin != 0 || o.cin != 3 || o.cout != 32 || !feats3_dev ||
          !(o.kernel_volume == 125 || o.kernel_volume == 27)) {
        set_error("op %d: stem expects level 0, 3->32 channels, 5^3 or 3^3 kernel", i);
        return A3D_ERR_UNSUPPORTED;
      }
      const Level& lv = s->lv[0];
      const int ks = o.kernel_volume == 125 ? 5 : 3;
      const bool morton = lv.grid && ks / 2 <= kGridPad && lv.n < (1 << 25);   // = k_stem's `fused` path
      if (!feats_ready) {
        k_gather_feats<<<(lv.n + 255) / 256, 256, 0, st>>>(feats3_dev, s->orig_row, lv.n, feats4, morton ? lv.perm : nullptr);
        feats_ready = true;
      }
      const size_t lds = (size_t)o.kernel_volume * 96 * 4 + (size_t)64 * o.kernel_volume * 4;
      ProfScope ps(st, A3D_PROF_STEM, 0, o.kernel_volume, 3, 32, lv.n);
      // 8 waves per 64-voxel workgroup (measured: 4 waves 229 us, 8 waves 164 us, 16 waves 159 us)
      Level glv = lv;
      if (ks / 2 > kGridPad) glv.grid = nullptr;   // the grid's empty border covers 5^3 neighbourhoods
      if (ks == 5)
        k_stem<512, 5><<<(lv.n + 63) / 64, 512, lds, st>>>(lv.xyzb, lv.n, lv.hkeys, lv.hvals, lv.hmask, feats4, o.w_dev, ks,
                                                           o.scale_dev, o.shift_dev, o.relu, out, ldo, zero_row, glv);
      else
        k_stem<512, 3><<<(lv.n + 63) / 64, 512, lds, st>>>(lv.xyzb, lv.n, lv.hkeys, lv.hvals, lv.hmask, feats4, o.w_dev, ks,
                                                           o.scale_dev, o.shift_dev, o.relu, out, ldo, zero_row, glv);
      A3D_LAUNCH_CHECK();
      continue;
    }

    // ---- input placement
    if (o.in_buf < 0 || o.in_buf >= n_bufs || bufs[o.in_buf].level != Lin ||
        o.in_coff + o.cin > bufs[o.in_buf].channels) {
      set_error("op %d: bad input buffer", i);
      return A3D_ERR_INVALID;
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = buf_ptr(o.in_buf) + o.in_coff;
    a.ldi = bufs[o.in_buf].channels;
    a.n_in = s->lv[Lin].n;
    a.w = o.w_dev;
    a.K = o.kernel_volume;
    a.cin = o.cin;
    a.cout = o.cout;
    a.out = out;
    a.ldo = ldo;
    a.out_map = out_map;
    a.scale = o.scale_dev;
    a.shift = o.shift_dev;
    a.res = res;
    a.ldr = ldr;
    a.relu = o.relu;
    a.zero_row = zero_row;
    a.n_out = s->lv[lvl_out].n;
    a.tag_table = o.kind;
    a.tag_level = Lin;
    if (o.proj_cin > 0) {   // fused residual projection (see a3d_op)
      if (o.kind != A3D_OP_CONV3 || o.proj_buf < 0 || o.proj_buf >= n_bufs || bufs[o.proj_buf].level != Lin ||
          o.proj_coff + o.proj_cin > bufs[o.proj_buf].channels || o.scale_dev) {
        set_error("op %d: bad fused projection", i);
        return A3D_ERR_INVALID;
      }
      a.in2 = buf_ptr(o.proj_buf) + o.proj_coff;
      a.ldi2 = bufs[o.proj_buf].channels;
      a.cin2 = o.proj_cin;
    }
    const int* pre = nullptr;
    switch (o.kind) {
      case A3D_OP_CONV3:
        if (o.kernel_volume != 27) { set_error("op %d: CONV3 needs kernel volume 27", i); return A3D_ERR_INVALID; }
        a.nbr = s->lv[Lin].nbr27;
        a.nbr_stride = s->lv[Lin].npad;
        a.gmask = s->lv[Lin].gmask27;
        pre = s->lv[Lin].pre27;
        break;
      case A3D_OP_DOWN:
        if (o.kernel_volume != 8 || Lin >= A3D_NUM_LEVELS - 1) { set_error("op %d: bad DOWN", i); return A3D_ERR_INVALID; }
        a.nbr = s->lv[Lin].child8;
        a.nbr_stride = s->lv[Lin + 1].npad;
        a.gmask = s->lv[Lin].gmask_down;
        pre = s->lv[Lin].pre_down;
        break;
      case A3D_OP_UP:
        if (o.kernel_volume != 8 || Lin < 1) { set_error("op %d: bad UP", i); return A3D_ERR_INVALID; }
        a.nbr = s->lv[Lin - 1].up8;
        a.nbr_stride = s->lv[Lin - 1].npad;
        a.gmask = s->lv[Lin - 1].gmask_up;
        a.out_map = s->lv[Lin - 1].up_rows;
        pre = s->lv[Lin - 1].pre_up;
        break;
      case A3D_OP_LINEAR:
        if (o.kernel_volume != 1) { set_error("op %d: LINEAR needs kernel volume 1", i); return A3D_ERR_INVALID; }
        break;
      default:
        set_error("op %d: unknown kind %d", i, o.kind);
        return A3D_ERR_INVALID;
    }
    bool head_after = false;
    if (o.head_cout > 0) {
      if (!o.head_w_dev || !ext_out_dev || lvl_out != 0 || o.kind == A3D_OP_UP || o.out_buf == A3D_BUF_EXT_OUT ||
          o.head_cout % 16 || o.head_cout > ext_out_ld) {
        set_error("op %d: a fused head needs a level-0 op into a buffer, packed head weights and the external output", i);
        return A3D_ERR_INVALID;
      }
      bool handoff_h = o.kernel_volume > 1;
      if (handoff_h && o.kernel_volume <= 8) {
        const SkPlan q = plan_sk(a.n_out, a.K, a.cin, a.cout, false);
        if ((long long)q.ntile * q.n_cblk >= 192) handoff_h = false;
      }
      if (o.kind != A3D_OP_LINEAR && a.cin2 == 0 && !(conv_wl_supported(a)) && sk_head_ok(a.n_out, a.K, a.cin, a.cout, o.head_cout, handoff_h)) {
        a.head_w = o.head_w_dev;
        a.head_bias = o.head_bias_dev;
        a.head_out = ext_out_dev;
        a.head_ld = ext_out_ld;
        a.head_cout = o.head_cout;
        a.head_map = s->orig_row;
      } else {
        head_after = true;   // same arithmetic as its own launch below
      }
    }
    if (a.cin2 > 0 && !sk_fused_ok(a.n_out, a.cin, a.cout, a.cin2)) {
      // no fused build for this (row count, shape): the same arithmetic as two launches on the same packed weights --
      // the 27 offsets into `out` without the ReLU, then the 1x1 slice (packed right behind them) added in place
      ConvArgs c1 = a;
      c1.in2 = nullptr, c1.cin2 = 0, c1.ldi2 = 0, c1.relu = 0;
      rc = launch_conv_sk(c1, pre, partial, L.partial_floats, queues + (size_t)i * kMaxQueuesPerOp, st);
      if (rc != A3D_OK) return rc;
      if (conv_emu(27, a.cin, a.cout) || a.out_map) {
        set_error("op %d: fused projection has no fallback in the emulated-fp32 build", i);
        return A3D_ERR_UNSUPPORTED;
      }
      ConvArgs c2;
      memset(&c2, 0, sizeof(c2));
      c2.in = a.in2, c2.ldi = a.ldi2, c2.n_in = a.n_out, c2.K = 1, c2.cin = a.cin2, c2.cout = a.cout;
      c2.w = a.w + (size_t)27 * a.cin * a.cout;
      c2.out = a.out, c2.ldo = a.ldo, c2.n_out = a.n_out, c2.res = a.out, c2.ldr = a.ldo, c2.relu = a.relu;
      c2.zero_row = a.zero_row, c2.tag_table = A3D_OP_LINEAR, c2.tag_level = Lin;
      rc = launch_conv_sk(c2, nullptr, nullptr, 0, nullptr, st);
      if (rc != A3D_OK) return rc;
    } else
    if (o.kind == A3D_OP_LINEAR && dense_supported(a.cin, a.cout) && a.n_out >= 4096)
      rc = launch_dense(a.in, a.ldi, nullptr, 0, a.n_out, a.cin, a.cout, a.w, a.scale, a.shift, a.res, a.ldr, a.relu,
                        a.out, a.ldo, a.zero_row, Lin, a.out_map, st);
    else
      rc = launch_conv_sk(a, pre, partial, L.partial_floats, queues + (size_t)i * kMaxQueuesPerOp, st);
    if (rc != A3D_OK) return rc;
    if (head_after) {   // the head as its own 1x1 launch over the op's finished rows
      if (dense_supported(o.cout, o.head_cout) && a.n_out >= 4096)
        rc = launch_dense(out, ldo, nullptr, 0, a.n_out, o.cout, o.head_cout, o.head_w_dev, nullptr, o.head_bias_dev, nullptr, 0, 0,
                          ext_out_dev, ext_out_ld, -1, Lin, s->orig_row, st);
      else {
        ConvArgs h;
        memset(&h, 0, sizeof(h));
        h.in = out, h.ldi = ldo, h.n_in = a.n_out, h.w = o.head_w_dev, h.K = 1, h.cin = o.cout, h.cout = o.head_cout;
        h.out = ext_out_dev, h.ldo = ext_out_ld, h.n_out = a.n_out, h.out_map = s->orig_row, h.shift = o.head_bias_dev;
        h.zero_row = -1, h.tag_table = A3D_OP_LINEAR, h.tag_level = Lin;
        rc = launch_conv_sk(h, nullptr, nullptr, 0, nullptr, st);
      }
      if (rc != A3D_OK) return rc;
    }
  }
  return A3D_OK;
}

// One sparse convolution on the caller's own buffers (the training tapes; the inference path runs whole programs):
//   y[n_out (+1)][ldy] = conv(x[n_in + 1][ldx]; Wp)      kind / level_in as in a3d_op, no epilogue
// x must carry the zero row (row n_in all zeros: what a missing neighbour gathers); row n_out of y is written as the
// zero row of the next layer when `y_zero_row` is set.  Transposed convs scatter through the scene's row map as in
// the program.  Workspace: a3d_conv_apply_workspace_bytes (hand-off state, zeroed here, + partial-accumulator slab).
extern "C" size_t a3d_conv_apply_workspace_bytes(const a3d_scene* s, int kind, int level_in, int cin, int cout) {
  if (!s || level_in < 0 || level_in >= A3D_NUM_LEVELS) return 0;
  int lvl_out = level_in + (kind == A3D_OP_DOWN ? 1 : kind == A3D_OP_UP ? -1 : 0);
  if (lvl_out < 0 || lvl_out >= A3D_NUM_LEVELS) return 0;
  const int K = kind == A3D_OP_CONV3 ? 27 : kind == A3D_OP_LINEAR ? 1 : 8;
  SkPlan q = plan_sk(s->lv[lvl_out].n, K, cin, cout, true);
  size_t slab = q.slab_floats;
  const DeepPlan dp = plan_deep(s->lv[lvl_out].n, K, cin, cout, 0, kind == A3D_OP_UP);
  if (dp.use && dp.slab_floats > slab) slab = dp.slab_floats;
  return align256((size_t)kMaxQueuesPerOp * 4) + align256(slab * 4) + 256;
}

// shared body of a3d_conv_apply / a3d_conv_apply_acc / a3d_conv_bn_train_forward
static int conv_apply_impl(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx, int cin,
                           const float* w_packed_dev, int cout, float* y_dev, int ldy, int y_zero_row, const float* res_dev,
                           int ldr, int* state_dev, float* stats, int stats_ld, int* stats_rows, void* workspace_dev,
                           size_t workspace_bytes, hipStream_t st, const char* who, const BwArgs* bw = nullptr) {
  if (!s || !x_dev || !w_packed_dev || !y_dev || level_in < 0 || level_in >= A3D_NUM_LEVELS || (ldx & 3) || (ldy & 3) ||
      ldx < cin || ldy < cout || (res_dev && ((ldr & 3) || ldr < cout))) {
    set_error("%s: bad arguments", who);
    return A3D_ERR_INVALID;
  }
  const int Lin = level_in;
  int lvl_out = Lin + (kind == A3D_OP_DOWN ? 1 : kind == A3D_OP_UP ? -1 : 0);
  if (lvl_out < 0 || lvl_out >= A3D_NUM_LEVELS) {
    set_error("%s: the op leaves the level range", who);
    return A3D_ERR_INVALID;
  }
  const size_t need = a3d_conv_apply_workspace_bytes(s, kind, level_in, cin, cout);
  if (!workspace_dev || workspace_bytes < need || ((uintptr_t)workspace_dev & 255)) {
    set_error("%s: workspace too small or misaligned (%zu < %zu)", who, workspace_bytes, need);
    return A3D_ERR_WORKSPACE;
  }
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x_dev;
  a.ldi = ldx;
  a.n_in = s->lv[Lin].n;
  a.w = w_packed_dev;
  a.cin = cin;
  a.cout = cout;
  a.out = y_dev;
  a.ldo = ldy;
  a.res = res_dev;
  a.ldr = ldr;
  a.n_out = s->lv[lvl_out].n;
  a.zero_row = y_zero_row ? s->lv[lvl_out].n : -1;
  a.tag_table = kind;
  a.tag_level = Lin;
  const int* pre = nullptr;
  switch (kind) {
    case A3D_OP_CONV3:
      a.K = 27;
      a.nbr = s->lv[Lin].nbr27;
      a.nbr_stride = s->lv[Lin].npad;
      a.gmask = s->lv[Lin].gmask27;
      pre = s->lv[Lin].pre27;
      break;
    case A3D_OP_DOWN:
      a.K = 8;
      a.nbr = s->lv[Lin].child8;
      a.nbr_stride = s->lv[Lin + 1].npad;
      a.gmask = s->lv[Lin].gmask_down;
      pre = s->lv[Lin].pre_down;
      break;
    case A3D_OP_UP:
      a.K = 8;
      a.nbr = s->lv[Lin - 1].up8;
      a.nbr_stride = s->lv[Lin - 1].npad;
      a.gmask = s->lv[Lin - 1].gmask_up;
      a.out_map = s->lv[Lin - 1].up_rows;
      pre = s->lv[Lin - 1].pre_up;
      break;
    case A3D_OP_LINEAR:
      a.K = 1;
      break;
    default:
      set_error("%s: unknown kind %d", who, kind);
      return A3D_ERR_INVALID;
  }
  // the hand-off state (ticket, failure word, flags): the caller's zeroed block, or the head of the workspace zeroed here
  int* state = state_dev ? state_dev : (int*)workspace_dev;
  float* slab = (float*)((char*)workspace_dev + align256((size_t)kMaxQueuesPerOp * 4));
  const size_t slab_floats = (workspace_bytes - align256((size_t)kMaxQueuesPerOp * 4)) / 4;
  if (!state_dev) A3D_HIP_CHECK(hipMemsetAsync(state, 0, (size_t)kMaxQueuesPerOp * 4, st));
  return launch_conv_sk(a, pre, slab, slab_floats, state, st, stats, stats_ld, stats_rows, bw);
}

extern "C" int a3d_conv_apply(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx, int cin,
                              const float* w_packed_dev, int cout, float* y_dev, int ldy, int y_zero_row,
                              void* workspace_dev, size_t workspace_bytes, void* stream) {
  return conv_apply_impl(s, kind, level_in, x_dev, ldx, cin, w_packed_dev, cout, y_dev, ldy, y_zero_row, nullptr, 0, nullptr,
                         nullptr, 0, nullptr, workspace_dev, workspace_bytes, (hipStream_t)stream, "a3d_conv_apply");
}

extern "C" size_t a3d_conv_state_bytes(void) { return (size_t)kMaxQueuesPerOp * 4; }

extern "C" int a3d_conv_deep_mode(int mode) {
  const int before = deep_mode();
  if (mode >= 0) g_deep_mode = mode;
  return before;
}

extern "C" int a3d_conv_apply_acc(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx, int cin,
                                  const float* w_packed_dev, int cout, float* y_dev, int ldy, int y_zero_row,
                                  const float* res_dev, int ldr, void* state_dev, void* workspace_dev,
                                  size_t workspace_bytes, void* stream) {
  return conv_apply_impl(s, kind, level_in, x_dev, ldx, cin, w_packed_dev, cout, y_dev, ldy, y_zero_row, res_dev, ldr,
                         (int*)state_dev, nullptr, 0, nullptr, workspace_dev, workspace_bytes, (hipStream_t)stream,
                         "a3d_conv_apply_acc");
}

// Input-gradient conv whose output is the complete gradient of a BatchNorm(+ReLU) unit's output: see a3d_conv_dgrad_bn in
// the header.  sums_dev [2][cout] fp64 = (sum g, sum g xhat) over all rows: what a3d_bn_backward_apply takes.
extern "C" int a3d_conv_dgrad_bn(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx, int cin,
                                 const float* w_packed_dev, int cout, float* g_dev, int ldg, int acc, const float* y_dev,
                                 int ldy, const float* raw_dev, int ld_raw, const float* mean_dev, const float* rstd_dev,
                                 int relu, double* sums_dev, void* state_dev, void* workspace_dev, size_t workspace_bytes,
                                 void* stream) {
  const size_t need = a3d_conv_bn_train_workspace_bytes(s, kind, level_in, cin, cout);
  if (!s || !need || !workspace_dev || workspace_bytes < need || ((uintptr_t)workspace_dev & 255) || !g_dev || !raw_dev ||
      !mean_dev || !rstd_dev || !sums_dev || (relu && !y_dev) || (ldy & 3) || (ld_raw & 3)) {
    set_error("a3d_conv_dgrad_bn: bad arguments or workspace too small (%zu < %zu)", workspace_bytes, need);
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t conv_bytes = align256(a3d_conv_apply_workspace_bytes(s, kind, level_in, cin, cout));
  float* partial = (float*)((char*)workspace_dev + conv_bytes);
  BwArgs bw;
  bw.y = y_dev, bw.raw = raw_dev, bw.mean = mean_dev, bw.rstd = rstd_dev, bw.ldy = ldy, bw.ldraw = ld_raw, bw.relu = relu;
  int rows_per_block = 0;
  int rc = conv_apply_impl(s, kind, level_in, x_dev, ldx, cin, w_packed_dev, cout, g_dev, ldg, 1, acc ? g_dev : nullptr, ldg,
                           (int*)state_dev, partial, cout, &rows_per_block, workspace_dev, conv_bytes, st, "a3d_conv_dgrad_bn",
                           &bw);
  if (rc != A3D_OK) return rc;
  const int lvl_out = level_in + (kind == A3D_OP_DOWN ? 1 : kind == A3D_OP_UP ? -1 : 0);
  const int n = s->lv[lvl_out].n;
  return bn_sums_from_partials(partial, (n + rows_per_block - 1) / rows_per_block, cout, sums_dev, st);
}

extern "C" size_t a3d_conv_bn_train_workspace_bytes(const a3d_scene* s, int kind, int level_in, int cin, int cout) {
  const size_t conv = a3d_conv_apply_workspace_bytes(s, kind, level_in, cin, cout);
  if (!conv) return 0;
  const int lvl_out = level_in + (kind == A3D_OP_DOWN ? 1 : kind == A3D_OP_UP ? -1 : 0);
  const size_t groups = ((size_t)s->lv[lvl_out].n + 15) / 16;                  // the finest partial granularity (k_conv_wl)
  return align256(conv) + align256(groups * 2 * (size_t)cout * 4) + 256;
}

extern "C" int a3d_conv_bn_train_forward(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx, int cin,
                                         const float* w_packed_dev, int cout, float* raw_dev, int ld_raw,
                                         const float* gamma_dev, const float* beta_dev, float eps, const float* res_dev,
                                         int ldr, int relu, float* y_dev, int ldy, int y_zero_row, float* save_mean_dev,
                                         float* save_rstd_dev, float* running_mean_dev, float* running_var_dev,
                                         float momentum, void* state_dev, void* workspace_dev, size_t workspace_bytes,
                                         void* stream) {
  const size_t need = a3d_conv_bn_train_workspace_bytes(s, kind, level_in, cin, cout);
  if (!s || !need || !workspace_dev || workspace_bytes < need || ((uintptr_t)workspace_dev & 255) || !raw_dev || !y_dev) {
    set_error("a3d_conv_bn_train_forward: bad arguments or workspace too small (%zu < %zu)", workspace_bytes, need);
    return A3D_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t conv_bytes = align256(a3d_conv_apply_workspace_bytes(s, kind, level_in, cin, cout));
  float* partial = (float*)((char*)workspace_dev + conv_bytes);
  int rows_per_block = 0;
  int rc = conv_apply_impl(s, kind, level_in, x_dev, ldx, cin, w_packed_dev, cout, raw_dev, ld_raw, 0, nullptr, 0,
                           (int*)state_dev, partial, cout, &rows_per_block, workspace_dev, conv_bytes, st,
                           "a3d_conv_bn_train_forward");
  if (rc != A3D_OK) return rc;
  const int lvl_out = level_in + (kind == A3D_OP_DOWN ? 1 : kind == A3D_OP_UP ? -1 : 0);
  const int n = s->lv[lvl_out].n;
  const int nblocks = (n + rows_per_block - 1) / rows_per_block;
  return bn_finish_from_partials(partial, nblocks, rows_per_block, raw_dev, ld_raw, n, cout, gamma_dev, beta_dev, eps, res_dev,
                                 ldr, relu, y_dev, ldy, y_zero_row, save_mean_dev, save_rstd_dev, running_mean_dev,
                                 running_var_dev, momentum, st);
}

extern "C" int a3d_linear(const float* in_dev, int ldi, const float* in_add_dev, int ldi_add, int64_t n, int cin,
                          int cout, const float* w_packed_dev, const float* scale_dev, const float* shift_dev,
                          const float* res_dev, int ldr, int relu, float* out_dev, int ldo, void* workspace_dev,
                          size_t workspace_bytes, void* stream) {
  if (!in_dev || !out_dev || !w_packed_dev || n <= 0 || n > (int64_t)1 << 30) {
    set_error("a3d_linear: bad arguments");
    return A3D_ERR_INVALID;
  }
  if (dense_supported(cin, cout))
    return launch_dense(in_dev, ldi, in_add_dev, ldi_add, (int)n, cin, cout, w_packed_dev, scale_dev, shift_dev, res_dev,
                        ldr, relu, out_dev, ldo, -1, -1, nullptr, (hipStream_t)stream);
  if (in_add_dev) {
    set_error("a3d_linear: a second input is only supported for 96/128 -> 96/128 channels (got %d -> %d)", cin, cout);
    return A3D_ERR_UNSUPPORTED;
  }
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in_dev;
  a.ldi = ldi;
  a.n_in = (int)n;
  a.w = w_packed_dev;
  a.K = 1;
  a.cin = cin;
  a.cout = cout;
  a.out = out_dev;
  a.ldo = ldo;
  a.n_out = (int)n;
  a.scale = scale_dev;
  a.shift = shift_dev;
  a.res = res_dev;
  a.ldr = ldr;
  a.relu = relu;
  a.zero_row = -1;
  a.tag_table = A3D_OP_LINEAR;
  a.tag_level = -1;
  (void)workspace_dev;
  (void)workspace_bytes;   // a 1x1 layer runs whole tiles (no hand-off state)
  return launch_conv_sk(a, nullptr, nullptr, 0, nullptr, (hipStream_t)stream);
}
