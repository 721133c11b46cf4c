/*
 * agile3d_hip.h -- C ABI of libagile3d_hip.so (MI355X / gfx950).
 *
 * The reference (ywyue/AGILE3D) is pure Python and has no FFI of its own: its hot path
 * calls MinkowskiEngine (third-party C++/CUDA) and torch.nn.  This header is the boundary a
 * maintainer would bind instead (ctypes stub in INTEGRATION.md).  Every entry point cites the
 * reference interface it replaces (paths relative to the reference repository).
 *
 * Conventions
 *   - every pointer named *_dev is a DEVICE pointer owned by the caller; the library never
 *     allocates device memory: callers pass workspaces sized by the *_workspace_bytes queries.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - all feature matrices are row-major fp32; coordinates are int32 (batch, x, y, z).
 *   - return value: 0 = ok, negative = error (a3d_last_error() gives the text; thread-local).
 *   - no exceptions cross the ABI; a scene/program handle is not thread-safe, distinct
 *     handles are independent.
 */
#ifndef AGILE3D_HIP_H
#define AGILE3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A3D_OK                 0
#define A3D_ERR_INVALID       -1   /* bad argument */
#define A3D_ERR_HIP           -2   /* HIP runtime error */
#define A3D_ERR_COORD_RANGE   -3   /* |xyz| >= 2^17 or batch index > 1022 */
#define A3D_ERR_DUPLICATE     -4   /* duplicate voxel coordinates */
#define A3D_ERR_WORKSPACE     -5   /* workspace too small */
#define A3D_ERR_UNSUPPORTED   -6   /* shape outside what the kernels are built for */

#define A3D_NUM_LEVELS 5           /* tensor strides 1,2,4,8,16 (res16unet.py:222-295) */

/* Version of this interface: bumped whenever a struct grows or a buffer contract changes (2: a3d_op's fused-head fields, the
 * third block of a3d_decoder_sample::kv0_dev + kv0_blocks; 3: a3d_conv_wgrad needs a3d_scene_build_wgrad_lists).  A host binding compares it with the header it was written
 * against before the first call (agile3d_amd/lib.py does). */
#define A3D_ABI_VERSION 3
int         a3d_version(void);
const char* a3d_last_error(void);
/* plain hipMemcpy device->host (+ stream sync); lets non-torch hosts and tests read tables */
int         a3d_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optional kernel timing (bench.py's live roofline numbers).  While enabled, every launch of the
 * kernels listed below is bracketed by a hipEvent pair on the launch stream.
 * ------------------------------------------------------------------------------------------ */
enum {
  A3D_PROF_SPCONV = 0,      /* k_spconv2<bn,ch> (3^3 / 2^3 s2 / transposed / small 1x1)      */
  A3D_PROF_SPLITK = 1,      /* retired (the split-K reduction launches of round 1)          */
  A3D_PROF_STEM = 2,        /* 5^3 stem                                                  */
  A3D_PROF_C2S = 3,         /* click-to-scene attention                                  */
  A3D_PROF_QUERY = 4,       /* query-side chain (one workgroup)                          */
  A3D_PROF_S2C = 5,         /* scene-to-click attention                                  */
  A3D_PROF_LNMASK = 6,      /* LayerNorm + mask head                                     */
  A3D_PROF_POSENC = 7,      /* Fourier position encoding                                 */
  A3D_PROF_SCENE_SORT = 8,  /* keys + radix sort + level compaction                      */
  A3D_PROF_SCENE_TABLES = 9,/* hash, neighbour tables, row clustering                    */
  A3D_PROF_CLICKS = 10,     /* click simulator (error clusters + nearest outside point)  */
  A3D_PROF_DENSE = 11       /* k_dense: N-point nn.Linear / 1x1 conv (no gather)          */
};
typedef struct {
  int32_t id, bn, kernel_volume, cin, cout, n_out, table, level, ksplit;
  float   ms;
} a3d_prof_entry;
int a3d_profile_enable(int on);
int a3d_profile_read(a3d_prof_entry* out, int max_entries);   /* returns #entries, clears */

/* ------------------------------------------------------------------------------------------
 * Scene = the coordinate manager.
 * Replaces: ME.SparseTensor(coordinates=, features=, device=) (engine.py:47-51,
 * eval_multi_obj.py:94-98) and the coordinate manager every ME layer consults implicitly
 * (voxel hash, stride-2 coordinate sets, kernel maps for 3^3 / 5^3 / 2^3-stride-2 kernels).
 * Rows of level 0 keep the caller's order at the API surface; internally every level is
 * re-ordered (Morton super-tiles, rows clustered by neighbour pattern) -- see DESIGN.md.
 * ------------------------------------------------------------------------------------------ */
typedef struct a3d_scene a3d_scene;

/* Stable ascending sort of (uint64 key, int32 value) pairs by the key bits [bit_begin, bit_end): the LSD radix sort the
 * scene build uses for its three sorts (csrc/radix.hip; replaces the sorts MinkowskiEngine's coordinate manager runs
 * inside `ME.SparseTensor(...)` / `MinkowskiConvolution` kernel-map construction, reference models/agile3d.py:163-170).
 * Exposed for the tests (a stand-alone call: up to 128 k pairs as ONE launch with grid barriers, so not next to another
 * barrier kernel of the process).  keys_in / vals_in are only read. */
size_t  a3d_sort_pairs_workspace_bytes(int64_t n);
int     a3d_sort_pairs_u64(const uint64_t* keys_in_dev, const int32_t* vals_in_dev, int64_t n, int bit_begin, int bit_end,
                           uint64_t* keys_out_dev, int32_t* vals_out_dev, void* workspace_dev, size_t workspace_bytes,
                           void* stream);

/* a3d_scene_create synchronises `stream` once (the level sizes come back to the host).  It may be called from several host
 * threads / on several streams at once: a scene-sized input (<= 128 k voxels) runs its first sort and the level compaction as
 * ONE launch each with grid barriers inside, but only when no other call of the process is between its first launch and that
 * synchronisation (a process-wide counter); every other case uses launch chains whose workgroups wait only for workgroups
 * that already run.  A3D_SORT_ONE_LAUNCH=0 keeps the chains everywhere. */
size_t  a3d_scene_workspace_bytes(int64_t n_voxels);
int     a3d_scene_create(const int32_t* coords4_dev, int64_t n_voxels,
                         void* workspace_dev, size_t workspace_bytes,
                         void* stream, a3d_scene** out);
void    a3d_scene_destroy(a3d_scene* s);
int64_t a3d_scene_level_size(const a3d_scene* s, int level);
/* batch samples are contiguous row ranges in ascending batch order (ME.utils.batched_coordinates,
 * datasets/InterMultiObj3DSegDataset.py:129); returns their number B and writes the first row of
 * sample i to starts_out[i] (i < max_out); sample i ends where sample i+1 starts (the last at n). */
int     a3d_scene_batch_ranges(const a3d_scene* s, int64_t* starts_out, int max_out);
/* Level-0 voxel lookup structure chosen by a3d_scene_create: returns 1 and the (x, y, z) cell counts when the batch's
 * bounding box was small enough for a dense voxel -> row grid (at most 64 cells per voxel), 0 when the hash table is
 * used (results are identical; A3D_GRID=0 in the environment forces the hash table). */
int     a3d_scene_grid_dims(const a3d_scene* s, int dims_out[3]);

/* read-only views of the scene tables (device pointers valid while the workspace lives) */
enum {
  A3D_TAB_XYZB      = 0,  /* int32 [n][4]  (x,y,z,batch) in level units, internal row order       */
  A3D_TAB_NBR27     = 1,  /* int32 [27][npad]  3^3 neighbour rows, missing -> n (the zero row)    */
  A3D_TAB_GMASK27   = 2,  /* uint32 [npad/16]  offsets present in each 16-row group               */
  A3D_TAB_CHILD8    = 3,  /* int32 [8][npad(level+1)] rows of `level` under each coarse row       */
  A3D_TAB_GMASKDOWN = 4,  /* uint32 [npad(level+1)/16]                                            */
  A3D_TAB_UP8       = 5,  /* int32 [8][npad] parent row (level+1) of virtual row v at its slot    */
  A3D_TAB_GMASKUP   = 6,  /* uint32 [npad/16]                                                     */
  A3D_TAB_UPROWS    = 7,  /* int32 [npad] virtual row -> row of `level`                           */
  A3D_TAB_ORIGROW   = 8,  /* int32 [n0]   internal level-0 row -> caller's row                    */
  /* 9: retired (tile order of the former dynamic tile queue) */
  A3D_TAB_PRE27     = 10, /* int32 [npad/64+1] (tile, offset) pairs of the 3^3 map before each 64-row tile */
  A3D_TAB_PREDOWN   = 11, /* int32 [npad(level+1)/64+1] same for the stride-2 map                       */
  A3D_TAB_PREUP     = 12  /* int32 [npad/64+1] same for the transposed map                              */
};
int a3d_scene_table(const a3d_scene* s, int level, int which, const void** ptr_dev, int64_t* count);

/* ------------------------------------------------------------------------------------------
 * Backbone program.
 * Replaces: Res16UNetBase.forward (models/res16unet.py:222-295), BasicBlock.forward
 * (models/modules/resnet_block.py:48-64), ME.MinkowskiConvolution / ConvolutionTranspose /
 * BatchNorm / ReLU / cat, and lin_squeeze_head (models/agile3d.py:43-45,179).
 * The host describes the network as a list of ops over numbered activation buffers (the
 * topology stays where the reference keeps it: in the host language); the library runs it.
 * ------------------------------------------------------------------------------------------ */
enum {
  A3D_OP_STEM   = 0,  /* 5^3 (or 3^3) conv, Cin=3, input = caller features          (res16unet.py:225) */
  A3D_OP_CONV3  = 1,  /* 3^3 stride-1 conv on `level_in`                           (resnet_block.py:24-43) */
  A3D_OP_DOWN   = 2,  /* 2^3 stride-2 conv level_in -> level_in+1                  (res16unet.py:229,...) */
  A3D_OP_UP     = 3,  /* 2^3 stride-2 transposed conv level_in -> level_in-1       (res16unet.py:253,...) */
  A3D_OP_LINEAR = 4   /* 1x1 conv                                                  (resnet.py:108-123)   */
};
#define A3D_BUF_NONE       -1
#define A3D_BUF_EXT_OUT    -2   /* out: caller's [n0][cout] matrix, rows in the CALLER's order */

typedef struct {
  int32_t level;       /* rows = a3d_scene_level_size(level) + 1 (last row = zeros) */
  int32_t channels;    /* row stride in floats */
} a3d_buf_desc;

typedef struct {
  int32_t kind;
  int32_t level_in;
  int32_t cin, cout;
  int32_t in_buf,  in_coff;    /* input  = columns [in_coff, in_coff+cin)   of buffer in_buf  */
  int32_t out_buf, out_coff;   /* output = columns [out_coff, out_coff+cout) of buffer out_buf */
  int32_t res_buf, res_coff;   /* residual added before the ReLU, or A3D_BUF_NONE             */
  int32_t relu;
  int32_t kernel_volume;       /* 125 / 27 / 8 / 1 */
  const float* w_dev;          /* weights packed by a3d_pack_conv_weight (STEM: raw [K][3][32]) */
  const float* scale_dev;      /* [cout] folded BatchNorm scale, or NULL (=1)                   */
  const float* shift_dev;      /* [cout] folded BatchNorm shift / bias, or NULL (=0)            */
  /* CONV3 only, proj_cin > 0: the block's residual projection fused into this conv (BasicBlock.downsample: 1x1 conv +
   * BatchNorm on the BLOCK INPUT, resnet_block.py:59-61; models/resnet.py:108-123) -- out = act(conv3(in) + proj_in W1x1 +
   * shift).  w_dev then holds the 27 offsets' packed weights followed by the packed 1x1 weight [proj_cin][cout], BOTH with
   * their BatchNorm scale already multiplied in (scale_dev = NULL), shift_dev the sum of the two shifts; proj_in = columns
   * [proj_coff, proj_coff + proj_cin) of buffer proj_buf (same level).  proj_cin = 0: no projection. */
  int32_t proj_buf, proj_coff, proj_cin, reserved_;
  /* head_cout > 0 (a level-0 op whose output has <= 128 columns): a 1x1 layer applied to this op's OUTPUT rows as a second
   * GEMM in the same kernel's epilogue -- lin_squeeze_head behind block8's last conv (models/agile3d.py:43-45,179): the
   * workgroup holds complete output rows, so ext_out[caller row][0 .. head_cout) = out_row @ head_w + head_bias is
   * written without reading the rows back.  head_w_dev: packed [1][cout][head_cout] (a3d_pack_conv_weight), head_bias_dev
   * [head_cout] or NULL; the result goes to a3d_program_run's ext_out (rows in the CALLER's order).  Shapes without a fused
   * build run the 1x1 layer as its own launch (same arithmetic). */
  const float* head_w_dev;
  const float* head_bias_dev;
  int32_t head_cout, reserved2_;
} a3d_op;

/* W[K][cin][cout] (ME layout, models/modules/common.py:137-155) -> MFMA B-fragment order */
int a3d_pack_conv_weight(const float* w_dev, int kernel_volume, int cin, int cout,
                         float* packed_dev, void* stream);
/* floats the packed form of a [kernel_volume][cin][cout] weight occupies: kernel_volume * cin * cout, except for the layers
 * the opt-in emulated-fp32 build runs (A3D_CONV_EMU=1: 96-column gathered convolutions, three bf16 planes = 1.5 floats per
 * weight).  Size `packed_dev` of a3d_pack_conv_weight with it. */
size_t a3d_conv_weight_packed_floats(int kernel_volume, int cin, int cout);

size_t a3d_program_workspace_bytes(const a3d_scene* s, const a3d_buf_desc* bufs, int n_bufs,
                                   const a3d_op* ops, int n_ops);
/* byte offset of activation buffer i inside the program workspace (for aux feature maps) */
size_t a3d_program_buffer_offset(const a3d_scene* s, const a3d_buf_desc* bufs, int n_bufs, int i);
int    a3d_program_run(const a3d_scene* s, const a3d_buf_desc* bufs, int n_bufs,
                       const a3d_op* ops, int n_ops,
                       const float* feats3_dev,     /* [n0][3] caller order (STEM input)   */
                       float* ext_out_dev, int ext_out_ld,
                       void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the sparse convolutions (first part): what MinkowskiEngine's autograd
 * computes for `losses.backward()` (engine.py:137-150) through conv / conv_tr (models/modules/common.py:125-188).
 * dL/dx needs no entry point of its own: it is a3d_program_run on the transposed kernel map with transposed weights
 * (3^3: W'[k] = W[26-k]^T, same op; stride 2 <-> transposed with W_s^T; 1x1: W^T) -- agile3d_amd/backward.py.
 * a3d_conv_wgrad: dW[k][ci][co] = sum over the pairs (r_in, r_out) of offset k of x[r_in][ci] * dy[r_out][co];
 * kind / level_in as in the op struct: A3D_OP_CONV3 / DOWN / UP / LINEAR, x [n_in][ldx], dy [n_out][ldy] in the scene's
 * internal row order, channels multiples of 32, dw_dev [K][cin][cout] (ME layout).  Deterministic (fixed summation
 * order).
 * a3d_scene_build_wgrad_lists: once per scene before the first a3d_conv_wgrad on it (the 3^3 / stride-2 / transposed kinds
 * refuse without it): for every kernel map of the scene and every offset k, the 16-position groups that have offset k, so
 * that the kernel's work items are equal cuts of these lists instead of equal cuts of the rows.  The workspace
 * (a3d_scene_wgrad_lists_bytes) must live as long as the scene is used for weight gradients.  The call does not synchronise:
 * the list lengths travel to a pinned host buffer behind an event, and the first a3d_conv_wgrad /
 * a3d_conv_wgrad_workspace_bytes on the scene waits for them (the launch plans are made on the host).
 * ------------------------------------------------------------------------------------------ */
size_t a3d_scene_wgrad_lists_bytes(const a3d_scene* s);
int    a3d_scene_build_wgrad_lists(a3d_scene* s, void* workspace_dev, size_t workspace_bytes, void* stream);
size_t a3d_conv_wgrad_workspace_bytes(const a3d_scene* s, int kind, int level_in, int cin, int cout);
int    a3d_conv_wgrad(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx,
                      const float* dy_dev, int ldy, int cin, int cout, float* dw_dev,
                      void* workspace_dev, size_t workspace_bytes, void* stream);

/* One sparse convolution on the caller's own buffers -- the forward AND (on the transposed map, see above) the input
 * gradient of the training tapes, which keep every activation; the inference path runs whole programs instead.
 *   y[n_out (+1)][ldy] = conv(x[n_in + 1][ldx]; w_packed)      no BatchNorm / ReLU / residual
 * x carries the zero row (row n_in all zeros: what a missing neighbour gathers); with y_zero_row != 0 row n_out of y is
 * written as zeros (the zero row of the next layer).  Deterministic.  Workspace: a3d_conv_apply_workspace_bytes. */
size_t a3d_conv_apply_workspace_bytes(const a3d_scene* s, int kind, int level_in, int cin, int cout);
int    a3d_conv_apply(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx, int cin,
                      const float* w_packed_dev, int cout, float* y_dev, int ldy, int y_zero_row,
                      void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same convolution with what the training tapes need on top (round 5):
 *   a3d_conv_apply_acc: y = conv(x) + res  -- res_dev [n_out][ldr] may be y itself: a gradient accumulated IN the conv's
 *     epilogue (the input-gradient convs of a node with several consumers: no separate add pass);
 *     state_dev: a3d_conv_state_bytes() bytes ZEROED by the caller (the kernel's hand-off ticket and flags; one block per
 *     conv of an iteration, cleared with one memset for all of them) or NULL (cleared here, one memset per call).
 *   a3d_conv_bn_train_forward: conv -> BatchNorm on the statistics of THIS batch (+ res)(ReLU), the block of
 *     BasicBlock.forward (resnet_block.py:48-64) in training mode: raw_dev [n_out][ld_raw] = the conv's output (kept for
 *     the backward), y_dev as a3d_bn_train_forward's.  The batch statistics come out of the conv kernel's epilogue (per
 *     64-row tile: column sums and squared deviations from the tile mean, merged in fp64 in a fixed order), so the raw
 *     output is not read again for them.  Exact-fp32 builds only (not under A3D_CONV_EMU). */
size_t a3d_conv_state_bytes(void);
/* Small levels (a layer with a few stages of work per CU: levels 2-4 of one scene; res16unet.py:89-147,242-259) run on
 * k_conv_deep -- static (tile, column block, part) workgroups, both operands by LDS-DMA three stages ahead, a cut tile
 * finished by its last arriver -- instead of the stream-K kernel.  mode 1 (default; A3D_CONV_DEEP in the environment) uses it
 * where its cost model prefers it, 0 never; mode < 0 only queries; mode >= 16 forces a geometry (tuning only: bn / 32 |
 * ch / 32 << 4 | parts << 8 | two-slot ring << 19, tools/conv_bench.py --sweep).  Returns the mode in force before the call.  Workspaces
 * sized under one mode stay valid under the other (the launch falls back to the stream-K kernel when the slab is short). */
int    a3d_conv_deep_mode(int mode);
int    a3d_conv_apply_acc(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx, int cin,
                          const float* w_packed_dev, int cout, float* y_dev, int ldy, int y_zero_row,
                          const float* res_dev, int ldr, void* state_dev, void* workspace_dev, size_t workspace_bytes,
                          void* stream);
/*   a3d_conv_dgrad_bn: the input-gradient conv (kind / level_in of the BACKWARD op, i.e. the transposed map) whose result
 *     completes dL/dy of a BatchNorm(+ReLU) unit's output y [n_out][ldy]: g = (conv(x) (+ g when acc)) masked by y > 0 is
 *     written to g_dev (row n_out zeroed) and sums_dev [2][cout] (fp64) = sum g, sum g xhat with xhat = (raw - mean) rstd come
 *     out of the conv kernel's epilogue -- what a3d_bn_backward_apply needs (with relu = 0: g is masked already), without a
 *     pass over dy, y and raw for the sums.  Workspace: a3d_conv_bn_train_workspace_bytes of the same op. */
int    a3d_conv_dgrad_bn(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx, int cin,
                         const float* w_packed_dev, int cout, float* g_dev, int ldg, int acc, const float* y_dev, int ldy,
                         const float* raw_dev, int ld_raw, const float* mean_dev, const float* rstd_dev, int relu,
                         double* sums_dev, void* state_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
size_t a3d_conv_bn_train_workspace_bytes(const a3d_scene* s, int kind, int level_in, int cin, int cout);
int    a3d_conv_bn_train_forward(const a3d_scene* s, int kind, int level_in, const float* x_dev, int ldx, int cin,
                                 const float* w_packed_dev, int cout, float* raw_dev, int ld_raw,
                                 const float* gamma_dev, const float* beta_dev, float eps, const float* res_dev, int ldr,
                                 int relu, float* y_dev, int ldy, int y_zero_row, float* save_mean_dev,
                                 float* save_rstd_dev, float* running_mean_dev, float* running_var_dev, float momentum,
                                 void* state_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* weight gradient of the input convolution conv0p1s1 (5^3 or 3^3, 3 -> 32; res16unet.py:225): feats3_dev in the
 * caller's row order as for a3d_program_run, dy_dev [n0][lddy >= 32] in internal row order, dw_dev [K][3][32] */
size_t a3d_stem_wgrad_workspace_bytes(int kernel_volume);
/* the size that also lets a3d_stem_wgrad run its matrix-core path (round 5: dW as one [384 x 32] MFMA accumulator per wave,
 * voxels walked in Morton order against Morton-ordered colours kept behind the partials; dense level-0 grid, 5^3 only) --
 * with the smaller workspace above the per-offset kernel runs */
size_t a3d_stem_wgrad_scene_workspace_bytes(const a3d_scene* s, int kernel_volume);
int    a3d_stem_wgrad(const a3d_scene* s, const float* feats3_dev, const float* dy_dev, int lddy, int kernel_volume,
                      float* dw_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* BatchNorm in training mode over the [n][C] rows (ME.MinkowskiBatchNorm = nn.BatchNorm1d over the rows of the whole
 * batch, models/modules/common.py:22), with the residual add and ReLU of BasicBlock.forward (resnet_block.py:48-64):
 *   forward : y = relu?((x - mean) * rstd * gamma + beta (+ res)); mean / rstd of THIS batch are saved for the
 *             backward; running statistics (optional) are updated like torch (momentum, unbiased variance);
 *             y_zero_row != 0: y has n + 1 rows and row n is written as zeros (what a missing neighbour gathers)
 *   backward: g = dy masked by (y > 0) when relu; dres (optional) = g; dbeta = sum g; dgamma = sum g * xhat;
 *             dx = gamma * rstd * (g - dbeta / n - xhat * dgamma / n); zero_row != 0: dx (and dres) have n + 1 rows,
 *             row n written as zeros
 * C a multiple of 32 that divides 768 (32 .. 384), leading dimensions multiples of 4.  Deterministic. */
size_t a3d_bn_workspace_bytes(int64_t n, int C);
int    a3d_bn_train_forward(const float* x_dev, int ldx, int64_t n, int C, const float* gamma_dev,
                            const float* beta_dev, float eps, const float* res_dev, int ldr, int relu,
                            float* y_dev, int ldy, float* save_mean_dev, float* save_rstd_dev,
                            float* running_mean_dev, float* running_var_dev, float momentum, int y_zero_row,
                            void* workspace_dev, size_t workspace_bytes, void* stream);
int    a3d_bn_train_backward(const float* x_dev, int ldx, const float* y_dev, int ldy, const float* dy_dev, int lddy,
                             int64_t n, int C, const float* gamma_dev, const float* save_mean_dev,
                             const float* save_rstd_dev, int relu, float* dx_dev, int lddx, float* dres_dev,
                             int lddres, float* dgamma_dev, float* dbeta_dev, int zero_row,
                             void* workspace_dev, size_t workspace_bytes, void* stream);
/* The same BatchNorm in pieces, for statistics that span the data-parallel ranks (SyncBN; the reference normalises over
 * all rows of the batch on ONE device, models/modules/common.py:20-22 -- with one scene per rank the strict equivalent
 * exchanges [2C+1] numbers per layer).  The caller combines the per-rank numbers between the calls
 * (agile3d_amd/backward.py: all_gather + Chan's parallel variance for the forward, all_reduce for the backward).
 *   a3d_bn_local_stats    : stats[0..C) = mean of THIS call's rows, stats[C..2C) = sum (x - that mean)^2   (fp64)
 *   a3d_bn_apply          : y = relu?((x - mean) rstd gamma + beta (+ res)) with the GIVEN mean / rstd
 *   a3d_bn_backward_sums  : sums[0..C) = sum g, sums[C..2C) = sum g xhat over THIS call's rows (fp64), g = dy (y > 0)
 *   a3d_bn_backward_apply : dx from the GLOBAL sums / row count; dgamma, dbeta = the LOCAL sums (they are averaged over
 *                           the ranks with every other parameter gradient afterwards, as DDP + SyncBatchNorm do) */
int    a3d_bn_local_stats(const float* x_dev, int ldx, int64_t n, int C, double* stats_dev, void* workspace_dev,
                          size_t workspace_bytes, void* stream);
int    a3d_bn_apply(const float* x_dev, int ldx, int64_t n, int C, const float* gamma_dev, const float* beta_dev,
                    const float* mean_dev, const float* rstd_dev, const float* res_dev, int ldr, int relu,
                    float* y_dev, int ldy, int y_zero_row, void* stream);
int    a3d_bn_backward_sums(const float* x_dev, int ldx, const float* y_dev, int ldy, const float* dy_dev, int lddy,
                            int64_t n, int C, const float* mean_dev, const float* rstd_dev, int relu,
                            double* sums_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int    a3d_bn_backward_apply(const float* x_dev, int ldx, const float* y_dev, int ldy, const float* dy_dev, int lddy,
                             int64_t n, int C, const float* gamma_dev, const float* mean_dev, const float* rstd_dev,
                             int relu, const double* global_sums_dev, int64_t n_global, const double* local_sums_dev,
                             float* dx_dev, int lddx, float* dres_dev, int lddres, float* dgamma_dev,
                             float* dbeta_dev, int zero_row, void* stream);
/* out[c] = sum over rows of x[i][c] (bias gradient of lin_squeeze_head, agile3d.py:43-45); workspace as above */
int    a3d_column_sums(const float* x_dev, int ldx, int64_t n, int C, float* out_dev,
                       void* workspace_dev, size_t workspace_bytes, void* stream);

/* nn.LayerNorm over the channels of every row (the decoder's norms: attention_block.py:38,98,155; agile3d.py:137),
 * forward and backward (C a multiple of 64, <= 512; dx has the layout of x; workspace: a3d_bn_workspace_bytes(n, C)) */
int    a3d_layernorm_forward(const float* x_dev, int ldx, int64_t n, int C, const float* gamma_dev,
                             const float* beta_dev, float eps, float* y_dev, int ldy, void* stream);
int    a3d_layernorm_backward(const float* x_dev, int ldx, const float* dy_dev, int lddy, int64_t n, int C,
                              const float* gamma_dev, float eps, float* dx_dev, float* dgamma_dev, float* dbeta_dev,
                              void* workspace_dev, size_t workspace_bytes, void* stream);
/* dW [cin][cout] = x^T dy for row-major [n][ld] matrices: weight gradient of the decoder's nn.Linear layers (the input
 * gradient dy W^T is a3d_linear with the transposed weight) */
size_t a3d_linear_wgrad_workspace_bytes(int64_t n, int cin, int cout);
int    a3d_linear_wgrad(const float* x_dev, int ldx, const float* dy_dev, int ldy, int64_t n, int cin, int cout,
                        float* dw_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
/* The same gradient written where the caller accumulates it (what autograd's AccumulateGrad does for an nn.Linear of
 * attention_block.py / agile3d.py:51-55): dW as [cin][cout] (transposed = 0) or [cout][cin] (transposed = 1: nn.Linear.weight's
 * layout; a row slice of nn.MultiheadAttention's packed in_proj_weight is such a block at an offset) with leading dimension
 * ld_dw, assigned (accumulate = 0) or added; db_dev != NULL: the bias gradient, the column sums of dy, from the same pass over
 * dy, assigned or added (db_accumulate) */
size_t a3d_linear_wgrad_into_workspace_bytes(int64_t n, int cin, int cout);
int    a3d_linear_wgrad_into(const float* x_dev, int ldx, const float* dy_dev, int ldy, int64_t n, int cin, int cout,
                             float* dw_dev, int ld_dw, int transposed, int accumulate, float* db_dev, int db_accumulate,
                             void* workspace_dev, size_t workspace_bytes, void* stream);

/* Attention and mask-head primitives with their backward (training path of the decoder; the inference path uses the
 * fused kernels behind a3d_decoder_forward).  nn.MultiheadAttention (attention_block.py) = scores -> softmax -> apply
 * on [heads, Lq, Lk]; q / k / v are row-major [L][H*dh].  mask_dev: uint8 [Lq][Lk], non-zero = blocked (-inf).
 *   a3d_attn_scores        S[h][i][j] = scale * sum_d q[i][h*dh+d] k[j][h*dh+d]
 *   a3d_softmax_rows       in place over the last dimension of [rows][L]
 *   a3d_softmax_rows_backward   dP <- P * (dP - sum_j P dP)
 *   a3d_attn_apply         transposed = 0: O[i][c] = scale * sum_j P[h(c)][i][j] V[j][c];  1: O[j][c] = scale * sum_i P[h(c)][i][j] V[i][c]
 *   a3d_group_max(_backward)    per-object max over an object's queries (agile3d.py:353-360) and its gradient routing */
int a3d_attn_scores(const float* q_dev, const float* k_dev, int64_t Lq, int64_t Lk, int H, int dh, float scale,
                    const unsigned char* mask_dev, float* S_dev, void* stream);
int a3d_softmax_rows(float* S_dev, int64_t rows, int64_t L, void* stream);
int a3d_softmax_rows_backward(const float* P_dev, float* dP_dev, int64_t rows, int64_t L, void* stream);
/* the same over the MIDDLE dimension of [H][Lq][Lk] (scores kept transposed: scene-to-click attention stores
 * [head][query][point] so that the 80 k-long point index is the fastest one everywhere) */
int a3d_softmax_cols(float* S_dev, int H, int64_t Lq, int64_t Lk, void* stream);
int a3d_softmax_cols_backward(const float* P_dev, float* dP_dev, int H, int64_t Lq, int64_t Lk, void* stream);
size_t a3d_attn_apply_workspace_bytes(int64_t Lq, int64_t Lk, int H, int dh, int transposed);
int a3d_attn_apply(const float* P_dev, const float* V_dev, int64_t Lq, int64_t Lk, int H, int dh, int transposed,
                   float scale, float* O_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int a3d_group_max(const float* lq_dev, int64_t N, int Q, const int32_t* qbeg_dev, const int32_t* qend_dev, int G,
                  float* out_dev, int32_t* arg_dev, void* stream);
int a3d_group_max_backward(const float* dout_dev, const int32_t* arg_dev, int64_t N, int Q, int G, float* dlq_dev,
                           void* stream);
/* The attention mask of the NEXT decoder layer from this layer's mask logits (agile3d.py:362-383: `attn_mask` of the
 * click-to-scene attention; not differentiated): label[n] = first arg-max over the G = 1 + K logits of point n,
 * mask[q][n] = (label[n] != group_of_query[q]) && (some point carries group_of_query[q])  -- uint8 [Q][N], non-zero = blocked;
 * the second term is the reference's "a query that would be blocked everywhere is blocked nowhere" (agile3d.py:369,375). */
size_t a3d_next_layer_mask_workspace_bytes(int64_t N, int G);
int a3d_next_layer_mask(const float* logits_dev, int64_t N, int G, const int32_t* group_of_query_dev, int Q,
                        unsigned char* mask_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same attention in the flash formulation, for the two attentions of a decoder layer that have the N points on one
 * side (attention_block.py:86-98 as called at agile3d.py:283-290 and :305-312; 8 heads x 16 channels fixed): nothing of
 * size [heads, Lq, Lk] is written; the forward pass keeps the softmax statistics (row maximum, row sum), the backward
 * pass recomputes the probabilities tile by tile from them.  q_scaled = q / sqrt(16) (the caller scales: exact), so
 * dL/dq = dq_scaled / 4.  mask_dev: uint8 [Lq][Lk], non-zero = blocked, or NULL.
 *   click-to-scene (few queries, Lk = N keys):  o [Lq][128], stats [2][8][Lq] (max, sum)
 *   scene-to-click (Lq = N queries, few keys):  o [Lq][128], stats [Lq][8][2]
 * Reductions over the N points (dq of click-to-scene; dk, dv of scene-to-click) are per-chunk partial sums added in
 * chunk order by a second kernel: results are bit-identical run to run. */
size_t a3d_flash_c2s_workspace_bytes(int64_t Lq, int64_t Lk);
int a3d_flash_c2s_forward(const float* q_scaled_dev, const float* k_dev, const float* v_dev, const unsigned char* mask_dev,
                          int64_t Lq, int64_t Lk, float* o_dev, float* stats_dev, void* workspace_dev,
                          size_t workspace_bytes, void* stream);
int a3d_flash_c2s_backward(const float* q_scaled_dev, const float* k_dev, const float* v_dev, const unsigned char* mask_dev,
                           int64_t Lq, int64_t Lk, const float* o_dev, const float* stats_dev, const float* d_o_dev,
                           float* dq_scaled_dev, float* dk_dev, float* dv_dev, void* workspace_dev, size_t workspace_bytes,
                           void* stream);
size_t a3d_flash_s2c_workspace_bytes(int64_t Lq, int64_t Lk);
int a3d_flash_s2c_forward(const float* q_scaled_dev, const float* k_dev, const float* v_dev, int64_t Lq, int64_t Lk,
                          float* o_dev, float* stats_dev, void* stream);
int a3d_flash_s2c_backward(const float* q_scaled_dev, const float* k_dev, const float* v_dev, int64_t Lq, int64_t Lk,
                           const float* o_dev, const float* stats_dev, const float* d_o_dev, float* dq_scaled_dev,
                           float* dk_dev, float* dv_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Optimiser step of the reference's training loop: torch.optim.AdamW(lr, weight_decay) (main.py:125-127) after
 * clip_grad_norm_(parameters, max_norm) (engine.py:145-150).  a3d_sum_squares returns sum g^2 of one tensor to the
 * host (the caller adds the tensors, clip coefficient = min(1, max_norm / (sqrt(total) + 1e-6))); a3d_adamw_step is
 * torch's single-tensor AdamW update with the gradient read multiplied by grad_scale (the clip coefficient);
 * step counts from 1. */
size_t a3d_sum_squares_workspace_bytes(void);
int    a3d_sum_squares(const float* g_dev, int64_t n, double* out_host, void* workspace_dev, size_t workspace_bytes,
                       void* stream);
/* the same, added to *acc_dev (device double, zeroed by the caller) without a host synchronisation */
int    a3d_sum_squares_accumulate(const float* g_dev, int64_t n, double* acc_dev, void* workspace_dev,
                                  size_t workspace_bytes, void* stream);
int    a3d_adamw_step(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n,
                      int step, float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                      void* stream);

/* The same two operations over ALL parameter tensors in one launch each (a training step updates 268 tensors).  The
 * caller uploads a table of a3d_mt_tensor entries, one per tensor in any fixed order: chunk0 = number of
 * A3D_MT_CHUNK-element chunks of the tensors before it, bias1 = 1 - beta1^step, bias2_sqrt = sqrt(1 - beta2^step) of
 * THAT tensor's step count (torch.optim.AdamW's per-parameter state['step']).  a3d_sum_squares_multi writes
 * sum over all tensors of sum g^2 to *out_dev (fp64, deterministic); a3d_adamw_step_multi = a3d_adamw_step on every
 * entry (bit-identical results). */
#define A3D_MT_CHUNK 4096
typedef struct a3d_mt_tensor {
  float* p; const float* g; float* m; float* v;
  int64_t n;
  int32_t chunk0;
  float bias1, bias2_sqrt;
  int32_t pad_;
} a3d_mt_tensor;
size_t a3d_mt_workspace_bytes(int64_t n_chunks);
int    a3d_sum_squares_multi(const a3d_mt_tensor* table_dev, int n_tensors, int64_t n_chunks, double* out_dev,
                             void* workspace_dev, size_t workspace_bytes, void* stream);
int    a3d_adamw_step_multi(const a3d_mt_tensor* table_dev, int n_tensors, int64_t n_chunks, float lr, float beta1,
                            float beta2, float eps, float weight_decay, float grad_scale, void* stream);

/* Many conv weights packed by ONE launch (a training iteration repacks both orientations of every sparse-conv kernel
 * after the optimiser step: ~230 a3d_pack_conv_weight calls plus the transposes / flips / slices feeding them).  Job i
 * writes the packed [K][cin][cout] weight to dst, reading
 *   transposed = 0:  W[k][ci][co]      = src[k][ci][co]                         (src is [K][cin][cout])
 *   transposed = 1:  W[k][ci][co]      = src[flip ? K-1-k : k][c0 + co][ci]     (src is [K][src_cin][cin]: the weight of
 *                                        the input-gradient conv, output channels = the slice [c0, c0 + cout) of src's inputs)
 * chunk0 = number of A3D_MT_CHUNK-element chunks of the jobs before it (as a3d_mt_tensor).  Exact-fp32 packs only
 * (a3d_conv_weight_packed_floats(K, cin, cout) == K * cin * cout). */
typedef struct a3d_pack_job {
  const float* src;
  float* dst;
  int32_t K, cin, cout;
  int32_t src_cin, src_cout;
  int32_t transposed, flip, c0;
  int32_t chunk0;
  int32_t pad_;
} a3d_pack_job;
int a3d_pack_conv_weights_multi(const a3d_pack_job* table_dev, int n_jobs, int64_t n_chunks, void* stream);

/* Dense row-major GEMM: out[n][cout] = act(((in (+ in_add))[n][cin] @ W) * scale + shift + res).
 * Replaces the nn.Linear / in_proj pieces of nn.MultiheadAttention that run over all N points
 * (models/modules/attention_block.py:91-94; `in_add` is the position encoding the reference adds to
 * its queries / keys before projecting them, attention_block.py:25-26,88-90).  96/128 -> 96/128
 * channels run on a dedicated HBM-bound kernel (k_dense); other shapes fall back to the sparse-conv
 * kernel with kernel volume 1 (no in_add there; the workspace arguments are not used: whole tiles are assigned
 * statically, pass NULL / 0). */
int a3d_linear(const float* in_dev, int ldi, const float* in_add_dev, int ldi_add,
               int64_t n, int cin, int cout,
               const float* w_packed_dev, const float* scale_dev, const float* shift_dev,
               const float* res_dev, int ldr, int relu, float* out_dev, int ldo,
               void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Position encoding.
 * Replaces: Agile3d.get_pos_encs (models/agile3d.py:141-161) ->
 * PositionEmbeddingCoordsSine.get_fourier_embeddings (models/position_embedding.py:123-152).
 * minmax_dev receives [min_x,min_y,min_z,max_x,max_y,max_z] of the sample.
 * ------------------------------------------------------------------------------------------ */
int a3d_posenc_fourier(const float* xyz_dev, int64_t n, const float* gauss_B_dev /*[3][64]*/,
                       float* minmax_dev /*[6]*/, float* out_dev /*[n][128]*/,
                       void* workspace_dev, size_t workspace_bytes, void* stream);
/* The same for every sample of a batch in three launches: rows [starts_host[b], starts_host[b+1]) of xyz_dev / out_dev
 * belong to sample b (1..64 samples, each with its own min / max: agile3d.py:141-161 loops over the batch);
 * minmax_dev is [n_samples][6].  Bit-identical to a3d_posenc_fourier per sample. */
size_t a3d_posenc_batch_workspace_bytes(int n_samples);
int a3d_posenc_fourier_batch(const float* xyz_dev, const int64_t* starts_host, int n_samples,
                             const float* gauss_B_dev /*[3][64]*/, float* minmax_dev /*[n_samples][6]*/,
                             float* out_dev /*[N][128]*/, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Click-query decoder.
 * Replaces: Agile3d.forward_mask + mask_module (models/agile3d.py:183-384) and the post-norm
 * CrossAttentionLayer / SelfAttentionLayer / FFNLayer (models/modules/attention_block.py).
 * One call = one batch sample, all `n_layers` decoder iterations.
 * ------------------------------------------------------------------------------------------ */
#define A3D_MAX_QUERIES 256       /* clicks + learned background queries of one sample */
#define A3D_MAX_DEC_LAYERS 8

typedef struct {
  /* nn.MultiheadAttention / LayerNorm / Linear parameters in torch layout (row-major [out][in]):
   *   *_in_w = in_proj_weight [384][128] (rows 0..127 q, 128..255 k, 256..383 v),
   *   *_out_w = out_proj.weight [128][128], ffn_w1 = linear1.weight [dim_ff][128],
   *   ffn_w2 = linear2.weight [128][dim_ff]. */
  const float *c2s_in_w, *c2s_in_b, *c2s_out_w, *c2s_out_b, *c2s_norm_w, *c2s_norm_b;
  const float *c2c_in_w, *c2c_in_b, *c2c_out_w, *c2c_out_b, *c2c_norm_w, *c2c_norm_b;
  const float *ffn_w1, *ffn_b1, *ffn_w2, *ffn_b2, *ffn_norm_w, *ffn_norm_b;
  const float *s2c_in_w, *s2c_in_b, *s2c_out_w, *s2c_out_b, *s2c_norm_w, *s2c_norm_b;
  /* the [128][128] blocks that multiply all N points (c2s Wk, Wv; s2c Wq, Wo), as W^T packed by
   * a3d_pack_conv_weight(kernel_volume=1) */
  const float *c2s_wk_packed, *c2s_wv_packed, *s2c_wq_packed, *s2c_wo_packed;
  /* every matrix the QUERY side of the layer multiplies by (c2s out_proj, c2c in/out_proj, s2c k/v rows, c2s q rows,
   * FFN), in MFMA fragment order, made by a3d_decoder_pack_query_weights(w, layer, ...): a wave's weight load is 1 KB
   * contiguous instead of 64 B of each of 16 rows (35 -> 140 GB/s into one CU, tools/qload_ubench.hip).  REQUIRED since
   * round 4 (the kernel that read the torch-layout matrices is gone): with NULL here or in
   * a3d_decoder_weights::mask_pack the decoder entry points fail with A3D_ERR_INVALID. */
  const float* query_pack;
} a3d_decoder_layer;

typedef struct {
  int32_t n_layers;
  int32_t n_bg_queries;                 /* learned background queries (agile3d.py:47-48) */
  int32_t dim_ff;
  a3d_decoder_layer layers[A3D_MAX_DEC_LAYERS];
  const float *decoder_norm_w, *decoder_norm_b;
  const float *mask_w0, *mask_b0, *mask_w2, *mask_b2;     /* mask_embed_head.{0,2} */
  const float *bg_query_feat, *bg_query_pos;              /* [n_bg][128] */
  const float *gauss_B;                                   /* [3][64] */
  const float *time_table;                                /* [200][128] PositionalEncoding1D */
  const float *mask_pack;                                 /* mask_w0, mask_w2 in fragment order (layer = -1 below); required */
} a3d_decoder_weights;

/* Fragment-order copies of the query-side matrices of one decoder layer (layer >= 0: a3d_decoder_query_pack_floats(dim_ff)
 * floats) or of the mask head (layer = -1: a3d_decoder_mask_pack_floats()), read from the torch-layout pointers of `w`.
 * Re-run after the parameters change. */
size_t a3d_decoder_query_pack_floats(int32_t dim_ff);
size_t a3d_decoder_mask_pack_floats(void);
int a3d_decoder_pack_query_weights(const a3d_decoder_weights* w, int32_t layer, float* out_dev, void* stream);

size_t a3d_decoder_workspace_bytes(int64_t n, int n_queries);

/* One batch sample of forward_mask (agile3d.py:192: the reference loops `for b in range(batch_size)`).  Fields as in
 * a3d_decoder_forward; every sample brings its own workspace (a3d_decoder_workspace_bytes(n, n_clicks + n_bg)). */
typedef struct a3d_decoder_sample {
  const float* feats128_dev;            /* [n][128] rows of this sample */
  const float* posenc_dev;              /* [n][128] */
  int64_t n;
  const int32_t *click_row, *click_obj, *click_time;   /* HOST arrays, see below */
  int32_t n_clicks, n_objects;
  float* logits_dev;                    /* n_layers x [n][1 + n_objects] */
  void* workspace_dev;
  size_t workspace_bytes;
  /* Per-scene cache of the click-independent part of a pass (eval_multi_obj.py:112-160 runs ~100 passes per scene on
   * the same backbone output): the key / value projections of the FIRST layer's click-to-scene attention depend on
   * the scene only (agile3d.py:283-290 with src = pcd_features).  kv0_dev: [3][n][128] floats owned by the caller
   * (keys, values, and -- round 5 -- the first layer's scene-to-click QUERIES (feats + pos) Wq^T + bq, agile3d.py:305-312, which
   * depend on the scene only as well);
   * kv0_state 0 = not used, 1 = computed into kv0_dev by this call and used, 2 = valid from an earlier call with the same
   * features, position encodings and weights.  With the cache the first layer's attention reads K / V instead of
   * projecting them inside the fused kernel (39 instead of 75 us at 80 k points). */
  float* kv0_dev;
  int32_t kv0_state;
  int32_t kv0_blocks;                   /* [n][128] blocks kv0_dev holds: the library refuses a cache of fewer than 3 */
} a3d_decoder_sample;
/* All samples of a batch in one call: per decoder layer the three wide kernels are launched once for the whole batch
 * (samples with the same padded query count share the launches); results per sample equal a3d_decoder_forward's up to
 * the summation order of the click-to-scene partials. */
int    a3d_decoder_forward_batch(const a3d_decoder_weights* w, const a3d_decoder_sample* samples, int n_samples,
                                 void* stream);
/* click arrays are HOST arrays, object-major as the reference builds its queries
 * (agile3d.py:249-264): all clicks of object 1, ..., object K, then background clicks.
 * click_obj[i] in 0..K (0 = background), click_row = row of the sample, click_time < 200.
 * logits_dev: n_layers matrices [n][1+K]; the LAST is 'pred_masks', the others 'aux_outputs'. */
int    a3d_decoder_forward(const a3d_decoder_weights* w,
                           const float* feats128_dev, const float* xyz_dev,
                           const float* posenc_dev, const float* minmax_dev,
                           int64_t n,
                           const int32_t* click_row, const int32_t* click_obj,
                           const int32_t* click_time, int n_clicks, int n_objects,
                           float* logits_dev,
                           void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * The interactive loop around forward_mask.
 * Replaces: `p.argmax(-1)` + "update prediction with sparse gt" (eval_multi_obj.py:119-134),
 * mean_iou_scene (utils/seg.py:44-59), get_simulated_clicks + measure_error_size +
 * get_next_click_coo_torch (utils/seg.py:93-239) and loss_weights (utils/seg.py:62-70).
 * Labels and predictions are int32 object ids in 0..255 (0 = background).
 * ------------------------------------------------------------------------------------------ */
#define A3D_MAX_CLICKS 256

/* pred[i] = argmax_c logits[i][c] (first maximum), then pred[click_row[k]] = click_obj[k] in the
 * given order; click arrays are HOST arrays. */
int    a3d_argmax_labels(const float* logits_dev, int64_t n, int n_classes,
                         const int32_t* click_row, const int32_t* click_obj, int n_clicks,
                         int32_t* pred_dev, void* stream);

/* The same for every sample of a round in TWO launches (arg-max of all samples; their click lists): the loop over the samples
 * of eval_multi_obj.py:119-134 / engine.py:96-101 was two launches PER SAMPLE.  click_row / click_obj are HOST arrays;
 * workspace_dev (a3d_argmax_labels_batch_workspace_bytes, >= 16-byte aligned) takes the device copy of the click lists. */
typedef struct a3d_argmax_sample {
  const float*   logits_dev;   /* [n][n_classes] */
  int64_t        n;
  int32_t        n_classes;
  int32_t        n_clicks;     /* <= A3D_MAX_CLICKS */
  const int32_t* click_row;    /* host */
  const int32_t* click_obj;    /* host */
  int32_t*       pred_dev;     /* [n] */
} a3d_argmax_sample;
#define A3D_MAX_ROUND_SAMPLES 64
size_t a3d_argmax_labels_batch_workspace_bytes(int n_samples);
int    a3d_argmax_labels_batch(const a3d_argmax_sample* samples, int n_samples, void* workspace_dev, size_t workspace_bytes,
                               void* stream);

/* counts_dev: int64 [3][n_ids] + 1 trailing int64 (non-zero = an inverse_map entry was out of
 * range): [0][id] = |pred==id & label==id|, [1][id] = |pred==id|, [2][id] = |label==id| over the
 * n_full points i, with pred taken at inverse_map_dev[i] (NULL = identity).  IoU(id) =
 * [0]/([1]+[2]-[0]). */
int    a3d_iou_counts(const int32_t* pred_dev, int64_t n_pred, const int64_t* inverse_map_dev,
                      const int32_t* labels_dev, int64_t n_full, int n_ids,
                      int64_t* counts_dev, void* stream);
/* a3d_iou_counts of every sample of a round in one launch + one clear: counts_all_dev = [n_samples][3 n_ids + 1] int64,
 * sample i's block as a3d_iou_counts lays it out. */
typedef struct a3d_iou_sample {
  const int32_t* pred_dev;
  int64_t        n_pred;
  const int64_t* inverse_map_dev;   /* or NULL */
  const int32_t* labels_dev;
  int64_t        n_full;
} a3d_iou_sample;
int    a3d_iou_counts_batch(const a3d_iou_sample* samples, int n_samples, int n_ids, int64_t* counts_all_dev, void* stream);

/* One entry per error cluster (cluster id = 96*label + 11*pred over the wrongly labelled points,
 * utils/seg.py:186): `row` is the cluster point farthest from every point outside the cluster
 * (lowest row on ties), `error_size` that distance -- the next simulated click and the key the
 * reference sorts clusters by.  Entries come out in ascending cluster id (torch.unique order). */
typedef struct {
  int32_t cluster_id, row, label, pred;
  float   error_size;
} a3d_click_cluster;
size_t a3d_click_workspace_bytes(int64_t n);
/* *n_out_dev = number of clusters (may exceed max_out: only max_out are written), -1 if a label or
 * prediction was outside 0..255.  The search is exact: only a cluster's largest distance and its first
 * arg-max are ever used, so wrong points are first bounded from above against every 16th point, each
 * cluster's best candidate is measured exactly (a lower bound of the cluster's maximum), and only the
 * points whose upper bound reaches it go through the pass over all points (clicks.hip). */
int    a3d_click_clusters(const float* xyz_dev, const int32_t* pred_dev, const int32_t* labels_dev,
                          int64_t n, a3d_click_cluster* out_dev, int max_out, int32_t* n_out_dev,
                          void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same for up to 64 samples in ONE set of launches (the samples of a training click round, engine.py:103-116, or of a
 * lock-step evaluation round, eval_multi_obj.py:162-166): every kernel takes the samples from a device table, so a round
 * issues ~15 launches whatever the batch size instead of ~12 per sample.  Results per sample are those of
 * a3d_click_clusters; every sample brings its own workspace (a3d_click_workspace_bytes(n)). */
typedef struct a3d_click_sample {
  const float*   xyz_dev;          /* [n][3] */
  const int32_t *pred_dev, *labels_dev;   /* [n] object ids 0..255 */
  int64_t        n;
  a3d_click_cluster* out_dev;      /* max_out records */
  int32_t*       n_out_dev;
  int32_t        max_out;
  void*          workspace_dev;
  size_t         workspace_bytes;
  /* optional: a spatial order of the sample's points and its inverse (a3d_click_spatial_order; the coordinates of a scene do
   * not change over its rounds).  With it the first bounding stage looks at a row's neighbours in that order instead of a
   * sparse sample of all points: far tighter upper bounds where predictions are wrong nearly everywhere.  Any permutation of
   * 0..n-1 is VALID (the bounds stay bounds, the search stays exact); both NULL = the sampled stage. */
  const int32_t *order_dev, *inv_dev;
} a3d_click_sample;
int    a3d_click_clusters_batch(const a3d_click_sample* samples, int n_samples, void* stream);
/* order_dev[s] = row of the s-th point in Morton order of the coordinates (2^16 cells per axis over the bounding box),
 * inv_dev[row] = s.  No reference counterpart: an index the search above may use (utils/seg.py:157-171 has none). */
size_t a3d_click_spatial_order_workspace_bytes(int64_t n);
int    a3d_click_spatial_order(const float* xyz_dev, int64_t n, int32_t* order_dev, int32_t* inv_dev, void* workspace_dev,
                               size_t workspace_bytes, void* stream);

/* weights[i] = alpha + (beta-alpha) * (1 - min(d_i, tita)/tita), d_i = distance of point i to the
 * nearest clicked point; click_row is a HOST array. */
int    a3d_click_loss_weights(const float* xyz_dev, int64_t n, const int32_t* click_row, int n_clicks,
                              float tita, float alpha, float beta, float* weights_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Mask losses (first piece of SURVEY.md section 8 row f-2).
 * Replaces: SetCriterion.loss_bce / loss_dice (models/criterion.py:14-110) for ONE sample and ONE
 * prediction level: losses_dev[0] = mean_i w_i * CE(logits_i, target_i), losses_dev[1] = mean_i w_i *
 * dice_i (per-point soft IoU of softmax(logits_i) against the one-hot target, eps = 1e-6; NaN in both if
 * a target is outside 0..n_classes-1).  grad_logits_dev (optional, [n][n_classes]) receives
 * d(coef_bce * losses[0] + coef_dice * losses[1]) / d logits.  weights_dev NULL = all ones;
 * workspace: 64 bytes.
 * ------------------------------------------------------------------------------------------ */
int    a3d_mask_losses(const float* logits_dev, const int32_t* target_dev, const float* weights_dev,
                       int64_t n, int n_classes, float coef_bce, float coef_dice,
                       float* losses_dev, float* grad_logits_dev,
                       void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * On-device voxelisation.
 * Replaces: ME.utils.sparse_quantize(coordinates, quantization_size, return_index=True,
 * return_inverse=True) (datasets/InterMultiObj3DSegDataset.py:67-75): q = int32(floor(xyz / size)) in
 * the input's own dtype (is_f64: 0 = float32 [n][3], 1 = float64), voxels in first-occurrence order,
 * unique_map[v] = first point of voxel v, inverse_map[i] = voxel of point i.  Output arrays must hold
 * n_points entries; *n_voxels (HOST) receives the voxel count (one stream synchronisation).
 * ------------------------------------------------------------------------------------------ */
size_t a3d_quantize_workspace_bytes(int64_t n_points);
int    a3d_sparse_quantize(const void* xyz_dev, int is_f64, int64_t n_points, double quantization_size,
                           int32_t* coords_out_dev, int64_t* unique_map_dev, int64_t* inverse_map_dev,
                           int64_t* n_voxels, void* workspace_dev, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AGILE3D_HIP_H */
