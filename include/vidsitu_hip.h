/*
 * vidsitu_hip.h -- C-ABI of libvidsitu_hip.so (gfx950 / MI355X only).
 *
 * The reference (TheShadow29/VidSitu) has no FFI of its own: every FLOP of its
 * hot path is reached through torch.nn modules of un-vendored packages
 * (SURVEY.md section 2.2).  Each entry point below therefore cites the
 * reference call site whose arithmetic it replaces.
 *
 * Conventions (all entry points):
 *   - raw DEVICE pointers, POD descriptors, caller's hipStream_t (as void*);
 *   - never allocates, never synchronises, never throws; returns 0 or a
 *     negative vs_status; vs_last_error_string() explains the last failure of
 *     the calling thread;
 *   - activations are dense channels-last "NDHWC" bf16 (a torch tensor of
 *     logical shape [N,C,T,H,W] in torch.channels_last_3d memory format), with
 *     an explicit row pitch `ld` (elements between consecutive positions) so a
 *     channel concat is a pointer offset into a wider buffer;
 *   - conv weights are [Cout][kT][kH][kW][Cin] bf16 (the channels_last_3d image
 *     of the reference's [Cout,Cin,kT,kH,kW] parameter);
 *   - statistics, affine parameters, gradients of parameters: fp32.
 */
#ifndef VIDSITU_HIP_H
#define VIDSITU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vs_status {
  VS_OK = 0,
  VS_ERR_BAD_ARG = -1,
  VS_ERR_UNSUPPORTED = -2,
  VS_ERR_WORKSPACE = -3,
  VS_ERR_LAUNCH = -4
} vs_status;

/* conv epilogue / mode flags */
#define VS_CONV_AFFINE 1   /* y = acc*scale[c] + shift[c]   (eval-mode BN folded) */
#define VS_CONV_RESIDUAL 2 /* y += residual                 (ResBlock add)        */
#define VS_CONV_RELU 4     /* y = max(y, 0)                                        */
#define VS_CONV_STATS 8    /* per-block per-channel sum / sum-of-squares partials  */
#define VS_CONV_NAIVE 16   /* debug: one-thread-per-output direct kernel           */
/* bits 8..11: forced tile config id + 1 (0 = built-in heuristic); ids index
 * {128x128, 64x128, 128x64, 64x64, 256x32, 256x16, 256x128, 128x256} (BM x BN). */
#define VS_CONV_TILE(id) (((id) + 1) << 8)
#define VS_CONV_RING(ns) (((ns) & 7) << 16) /* staging: 0 heuristic, 1 register pipeline, 2..4 LDS-DMA ring stages */
#define VS_CONV_NOCLASS (1 << 19) /* dgrad of a strided conv: disable the stride-class tiling (debug / A-B) */
#define VS_CONV_SPLITK (1 << 15) /* allow the split-K plan (fp32 slabs + fused reduce/epilogue) */
#define VS_CONV_NOHALO (1 << 21) /* keep a unit-stride [kT,1,1] / [1,kH,kW] conv on the implicit-GEMM kernel (A/B, tests) */
#define VS_CONV_FORCEHALO (1 << 22) /* run it on the halo-image kernel whenever the shape is eligible (A/B, tests) */
#define VS_CONV_NOPW (1 << 23) /* keep a shallow-K pointwise conv on the implicit-GEMM kernel instead of the persistent
                                  weight-resident kernel (A/B, tests) */
#define VS_CONV_DIRECTBNB (1 << 25) /* dgrad on the small-channel kernel: let it emit the BN-backward sums (measured
                                       slower in the step than the separate reduce pass; A/B, tests) */
#define VS_CONV_BNB2 (1 << 26) /* dgrad that emits the sums of TWO BN units (vs_dgrad_epilogue.bn_y2 ...): set it in the desc
                                  of both vs_conv_dgrad_bnstats_rows and the launch (the tile kernel's variant only) */
#define VS_CONV_FORCEPW (1 << 24) /* run it on that kernel whenever the shape is eligible, also where the plan would
                                     not (one block per CU; A/B, tests) */
#define VS_CONV_NODEEP (1 << 27) /* keep a >= 192-column deep-reduction conv on the 128 x 128 tile kernel instead of the
                                    256 x 256 deep-pipeline kernel (conv_deep.hip; A/B, tests) */
#define VS_CONV_FORCEDEEP (1 << 28) /* run it on the deep-pipeline kernel whenever the shape is eligible (A/B, tests) */
#define VS_CONV_SPLITK_IL (1 << 29) /* in-launch split-K on the 128 x 128 tile (S blocks per tile, the last arriver sums the
                                       partial accumulators in split order and runs the fused epilogue) wherever the shape is
                                       eligible; without the flag the plan picks it for under-tiled deep reductions */
#define VS_CONV_NOSPLITK_IL (1 << 30) /* never (A/B, tests) */

/* Geometry of one Conv3d (bias-free, groups 1, dilation 1).
 * Replaces nn.Conv3d reached from vidsitu_code/mdl_sf_base.py:22-33 (s1..s5,
 * s*_fuse of slowfast.models.video_model_builder). */
typedef struct vs_conv_desc {
  int32_t N, Ti, Hi, Wi, Cin;
  int32_t To, Ho, Wo, Cout;
  int32_t kT, kH, kW;
  int32_t sT, sH, sW;
  int32_t pT, pH, pW;
  int32_t x_ld;   /* input row pitch  (elements, >= Cin)  */
  int32_t y_ld;   /* output row pitch (elements, >= Cout) */
  int32_t res_ld; /* residual row pitch                   */
  int32_t flags;
} vs_conv_desc;

const char* vs_last_error_string(void);
int vs_version(void);
/* Kernel launches issued by this library so far (process-wide, relaxed atomic; every launch site counts).
 * bench.py reports launches per step from it -- the reference's step is a chain of cuDNN / ATen launches
 * (utils/trn_utils.py:590-615), the length of that chain is what bounds the batch-8 step here. */
int64_t vs_launch_count(void);

/* NCDHW (f32 or bf16) -> NDHWC bf16 with channels zero-padded to Cpad.
 * Replaces the implicit layout of the A0 batch contract
 * (vidsitu_code/dat_loader.py:454-501 -> mdl_sf_base.py:169-180). */
int vs_pack_input(const void* x, int x_is_bf16, void* y, int N, int C, int T, int H, int W,
                  int Cpad, void* stream);
/* uint8 RGB frames [N][Tin][H][W][3] -> normalised bf16 activations [N][Tout][H][W][Cpad]:
 * ((x / 255 - mean[c]) / std[c], the reference's fp32 operation order,
 * utils/video_utils.py:147-164 after vidsitu_code/dat_loader.py:183-191); t_index[Tout] (device,
 * int32) picks frames (the slow pathway's linspace gather, video_utils.py:59-65), NULL = all.
 * mean3 / std3 are HOST pointers to 3 floats; reverse_channels = cfg.DATA.REVERSE_INPUT_CHANNEL. */
int vs_frames_u8_pack(const uint8_t* frames, const int* t_index, void* y, int N, int Tin, int Tout,
                      int H, int W, int Cpad, const float* mean3, const float* std3,
                      int reverse_channels, void* stream);

/* Stem convolution Conv3d(3 -> Cout, [kT,7,7], stride [1,2,2], pad [kT/2,3,3]) (the two
 * `pathway{p}_stem.conv` of s1).  x4: NDHWC bf16 with C padded to 4 (vs_pack_input, Cpad = 4);
 * wp: bf16 [ceil16(Cout)][kT][7][8][4] (kw 7 -> 8 and Cin 3 -> 4 zero-padded);
 * flags: VS_CONV_AFFINE | RELU | STATS; stats_partial: [vs_stem_stats_rows][2][Cout]. */
int vs_stem_conv_fwd(const void* x4, const void* wp, void* y, int N, int T, int H, int W, int Cout,
                     int kT, int y_ld, int flags, const float* scale, const float* shift,
                     float* stats_partial, void* stream);
int vs_stem_stats_rows(int N, int T, int H, int W);
/* Stem weight gradient, same patch scheme; dwp: fp32 [Cout][kT][7][8][4] (the packed weight's
 * layout; kw = 7 and c = 3 entries are padding).  workspace: vs_stem_wgrad_workspace_bytes. */
size_t vs_stem_wgrad_workspace_bytes(int N, int T, int H, int W, int Cout, int kT);
int vs_stem_conv_wgrad(const void* dy, const void* x4, float* dwp, int N, int T, int H, int W,
                       int Cout, int kT, int dy_ld, void* workspace, size_t ws_bytes, void* stream);

/* Forward conv as implicit GEMM on bf16 MFMA, fp32 accumulate.
 *   stats_partial: [vs_conv_stats_rows(desc)][2][Cout] fp32 when VS_CONV_STATS. */
int vs_conv_fwd(const void* x, const void* w, void* y, const vs_conv_desc* d, const float* scale,
                const float* shift, const void* residual, float* stats_partial,
                void* workspace, size_t ws_bytes, void* stream);
int vs_conv_stats_rows(const vs_conv_desc* d);
/* Split-K workspace (few-tile, deep-K layers); 0 when the plan for this shape has no split.
 * dgrad = 1 sizes the workspace of vs_conv_dgrad for the same descriptor.
 * The buffer's head holds one arrival counter per tile for the in-launch split plan: it has to be ZERO before the first
 * launch that uses the buffer and belongs to the library from then on (every launch leaves the counters at zero);
 * launches that may run concurrently (different streams) need different buffers. */
size_t vs_conv_workspace_bytes(const vs_conv_desc* d, int dgrad);
/* The launch plan the library picks for this descriptor (dgrad = 1: for vs_conv_dgrad):
 * out[5] = {tile rows, tile cols, LDS-DMA ring stages (0 = register-staged), split-K factor,
 * 1 if the register-resident small-channel kernel runs}; out[4] == 2: the halo-image kernel (conv_halo.hip),
 * out[0..1] its tile, out[2] its weight-ring depth, out[3] its unrolled tap count; out[4] == 3: the persistent
 * pointwise kernel (conv_pw.hip); out[4] == 4: the deep-pipeline kernel (conv_deep.hip: 256 x 256 tile, out[2] = 8
 * LDS sub-buffers); out[4] == 5: the tile kernel with out[3] blocks per tile and the in-launch split-K sum.
 * Profiling / attribution only. */
int vs_conv_plan(const vs_conv_desc* d, int dgrad, int* out);
/* Apply on load (training, the b -> c edge of a bottleneck: slowfast resnet_helper.BottleneckTransform.forward
 * `x = self.b_relu(self.b_bn(x)); x = self.c(x)`): the 1x1x1 convolution takes the PRODUCER unit's raw convolution
 * output x and that unit's batch-norm constants (vs_bn_finalize's scale / shift) and multiplies
 * relu(x * in_scale[c] + in_shift[c]) rounded to bf16 -- bit for bit the tensor vs_bn_apply would have stored, formed
 * on the operand fragments and never written.  vs_conv_fwd_aol: epilogue VS_CONV_STATS only; vs_conv_wgrad_aol: the
 * weight gradient of the same convolution (dy, x as in vs_conv_wgrad).  The *_ok queries say for which descriptors
 * the plan lands on a kernel with the transform (1x1x1, unit stride, Cin <= 512; forward: the persistent pointwise
 * kernel's 128-column variant or the 128 x 128 two-stage-ring tile; weight gradient: the 128-row ring tiles); a caller
 * materialises the activation with vs_bn_apply where they return 0. */
int vs_conv_aol_ok(const vs_conv_desc* d);
int vs_conv_fwd_aol(const void* x, const void* w, void* y, const vs_conv_desc* d, const float* in_scale,
                    const float* in_shift, float* stats_partial, void* stream);
int vs_conv_wgrad_aol_ok(const vs_conv_desc* d);
int vs_conv_wgrad_aol(const void* dy, const void* x, float* dw, const vs_conv_desc* d, const float* in_scale,
                      const float* in_shift, void* workspace, size_t ws_bytes, void* stream);
/* Evaluation: conv b (+ folded BN + ReLU) and conv c (1x1x1, + folded BN + residual + ReLU) of a fast-pathway
 * bottleneck in ONE launch -- slowfast resnet_helper.BottleneckTransform.forward (b, b_bn, b_relu, c, c_bn) and
 * ResBlock.forward's `x + f(x)` / relu for the blocks whose inner width is 8, 16 or 32 channels (SlowFast-R50 fast
 * pathway res2 / res3 / res4, SURVEY.md 7 "Layout"): the inner tensor never leaves the CU.  d describes conv b (flags:
 * VS_CONV_AFFINE | VS_CONV_RELU only; d->y_ld is ignored); w_c: bf16 [cout_c][Cout of b]; residual rows
 * [positions][res_ld] or NULL; y rows [positions][y_ld].  vs_conv_fwd_bc_fusable: 1 if the pair is taken
 * (conv b not pointwise; Cout_b in {8, 16}, taps * Cin_b <= 192, cout_c <= 64; or Cout_b = 32, 256 < taps * Cin_b <= 288,
 * cout_c <= 128), else the caller runs two vs_conv_fwd. */
int vs_conv_fwd_bc_fusable(const vs_conv_desc* d, int cout_c);
int vs_conv_fwd_bc(const void* x, const void* w_b, const vs_conv_desc* d, const float* scale_b,
                   const float* shift_b, const void* w_c, int cout_c, const float* scale_c,
                   const float* shift_c, const void* residual, int res_ld, void* y, int y_ld, int relu_c,
                   void* stream);

/* Data gradient: dx[N,Ti,Hi,Wi,Cin] = conv_transpose(dy, w) (+ residual when
 * desc.flags has VS_CONV_RESIDUAL: the gradient arriving over the other branch
 * of a fan-out, pitch desc.res_ld; may alias dx).  wt is the
 * [Cin][kT][kH][kW][Cout] bf16 image made by vs_weight_transpose.
 * Replaces autograd's cudnn_convolution_backward_input for the same layers. */
int vs_conv_dgrad(const void* dy, const void* wt, void* dx, const vs_conv_desc* d,
                  const void* residual, void* workspace, size_t ws_bytes, void* stream);

/* The same dgrad for a convolution whose INPUT is the output z of a BatchNorm + ReLU unit and whose dx is that
 * unit's complete dz: the epilogue also emits the unit's BN-backward partial sums, one row per M-tile --
 * stats_partial[rows][2][Cin] fp32, [0] = sum g, [1] = sum g * xhat, xhat = (bn_y - mean) * invstd, g = dz
 * (as rounded to bf16) where the unit's ReLU passed, else 0 -- so that vs_bn_bwd_finalize(stats_partial, rows,
 * ...) can follow directly and the separate pass of vs_bn_bwd_reduce over dz and bn_y is not needed (slowfast's
 * BatchNorm3d backward, autograd of `resnet_helper.BottleneckTransform.forward` / `ResBlock.forward`).
 * Two pairings: no residual input (the a -> b and b -> c links inside a bottleneck): relu_bits NULL, the mask
 * is recomputed as gamma * xhat + beta > 0; desc.flags has VS_CONV_RESIDUAL (conv a of the next block, dx =
 * dgrad + the gradient over the identity branch, the unit is the previous block's c unit): relu_bits = the bit
 * mask vs_bn_apply_mask wrote, [positions][Cin / 8] bytes; gamma / beta unused.
 * bn_y: the unit's saved conv output [positions][bn_y_ld] bf16.  vs_conv_dgrad_bnstats_rows(desc) = rows, or
 * 0 when this dgrad cannot emit the sums (small-channel direct kernel, split-K plan, NAIVE, a strided dgrad
 * with a residual, tiles other than 128x128 / 64x128 / 64x64): use vs_bn_bwd_reduce then. */
int vs_conv_dgrad_bnstats_rows(const vs_conv_desc* d);
int vs_conv_dgrad_bnstats(const void* dy, const void* wt, void* dx, const vs_conv_desc* d, const void* residual,
                          const void* bn_y, int bn_y_ld, const uint8_t* relu_bits, const float* mean,
                          const float* invstd, const float* gamma, const float* beta, float* stats_partial,
                          void* workspace, size_t ws_bytes, void* stream);

/* The general form of the two calls above: every optional epilogue operand of a data gradient in one POD
 * (zero-initialise it; unused members stay NULL / 0).
 *   residual        the gradient arriving over the other branch of a fan-out (desc.flags must carry
 *                   VS_CONV_RESIDUAL, pitch desc.res_ld).  It may alias dx: an accumulating dgrad, the
 *                   form the strided shortcut convolution of a ResBlock uses -- dx already holds conv a's
 *                   data gradient and only the positions the stride reaches are read and rewritten
 *                   (autograd's accumulation of the two branches of `ResBlock.forward`'s `x`).
 *   residual_bits   residual is an UNMASKED gradient dz of a BatchNorm + ReLU unit and this its ReLU bit
 *                   mask [positions][Cin / 8] (vs_bn_apply_mask): dz is added where the bit is set, which
 *                   spares the unit's backward-apply pass the write of a masked copy.
 *   bn_y ... stats_partial   as vs_conv_dgrad_bnstats. */
typedef struct vs_dgrad_epilogue {
  const void* residual;
  const uint8_t* residual_bits;
  const void* bn_y;
  int bn_y_ld;
  const uint8_t* relu_bits;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  float* stats_partial;
  /* A second BN unit fed by the same masked gradient (RESIDUAL form, unit-stride dgrads): the shortcut unit of the
   * ResBlock whose c unit is described above -- its saved conv output, mean, invstd; stats_partial2 gets
   * [rows][2][Cin] with the same sum(g) and this unit's sum(g * xhat).  All NULL / 0: off. */
  const void* bn_y2;
  int bn_y2_ld;
  const float* mean2;
  const float* invstd2;
  float* stats_partial2;
} vs_dgrad_epilogue;
int vs_conv_dgrad_ex(const void* dy, const void* wt, void* dx, const vs_conv_desc* d,
                     const vs_dgrad_epilogue* ep, void* workspace, size_t ws_bytes, void* stream);
int vs_weight_transpose(const void* w, void* wt, int Cout, int taps, int Cin, void* stream);
/* Every dgrad weight image of a model in one launch: src/dst are bf16 arenas with equal
 * element offsets; table[i] = {offset, Cout, taps, Cin, first flat index} (int64 x 5). */
int vs_weight_transpose_batched(const void* src, void* dst, const int64_t* table, int n,
                                int64_t total, void* stream);
/* fp32 nn.Linear weights [N][K] -> [K][N] for all linears of a model in one launch (table rows
 * {element offset (same in src and dst), N, 1, K, first flat index}). */
int vs_transpose_f32_batched(const float* src, float* dst, const int64_t* table, int n, int64_t total,
                             void* stream);
/* The same two transposes, tiled through LDS (reads contiguous along Cin, writes along Cout): table as
 * above; tile_first[i] = first tile of entry i, tiles(i) = taps * ceil(Cout/TS) * ceil(Cin/TS) with
 * TS = 64 for 2-byte (bf16 conv weights) and 32 for 4-byte (fp32 linear weights) elements. */
int vs_weight_transpose_tiled(const void* src, void* dst, const int64_t* table, const int64_t* tile_first,
                              int n, int64_t total_tiles, int elem_bytes, void* stream);

/* Weight gradient: dw[Cout][taps][Cin] fp32 = sum_p dy[p][co] * x[p@tap][ci].
 * workspace: vs_conv_wgrad_workspace_bytes(desc) bytes of fp32 split-K slabs.
 * desc.flags of a weight gradient: bits 8..11 forced tile, 16..18 VS_CONV_RING, 24..31 block slots / 8 (tuning knobs),
 * VS_WGRAD_NODEEP / VS_WGRAD_FORCEDEEP: keep it off / put it on the deep-pipeline kernel (128 x 256 output tile,
 * 32-position units, six-slot LDS ring; the plan takes it from 40 000 positions on). */
#define VS_WGRAD_NODEEP (1 << 12)
#define VS_WGRAD_FORCEDEEP (1 << 13)
size_t vs_conv_wgrad_workspace_bytes(const vs_conv_desc* d);
int vs_conv_wgrad(const void* dy, const void* x, float* dw, const vs_conv_desc* d,
                  void* workspace, size_t ws_bytes, void* stream);
/* Grouped weight gradients (round 5): the Conv3d bwd-filter of up to 20 convolutions -- a ResBlock's a, b, c and shortcut, or those of two or three consecutive blocks
 * (slowfast resnet_helper.ResBlock, reached from mdl_sf_base.py:22-33) -- as ONE launch of deep-pipeline blocks (128 x 256
 * output tiles) whose problems share the chip, + one grouped slab reduce where a problem is still split over positions.
 * Together the problems need far fewer position splits than each alone.  Every item: taps <= 31, Cin and Cout multiples
 * of 8, >= 512 positions, no forced-plan flags (vs_conv_wgrad_group_ok).  dw: fp32 [Cout][taps][Cin], overwritten.
 * workspace: vs_conv_wgrad_group_workspace_bytes bytes (slabs), private to the stream until the launch retires.
 * Bitwise reproducible (fixed summation order per problem for a given group). */
typedef struct vs_wgrad_item {
  const void* dy;
  const void* x;
  float* dw;
  vs_conv_desc d;
} vs_wgrad_item;
int vs_conv_wgrad_group_ok(const vs_wgrad_item* items, int n);
size_t vs_conv_wgrad_group_workspace_bytes(const vs_wgrad_item* items, int n);
int vs_conv_wgrad_group(const vs_wgrad_item* items, int n, void* workspace, size_t ws_bytes, void* stream);
/* The same weight gradient without its slab reduce: the position-split partials stay in `slabs`
 * ([*splits][Cout][taps*Cin] fp32, vs_conv_wgrad_workspace_bytes(desc) bytes, owned by the caller until the
 * reduce) and *splits says how many there are; *splits == 1 means dw was written directly and slabs is
 * untouched.  vs_wgrad_reduce_batched then sums the slabs of MANY layers in one launch, in exactly
 * vs_conv_wgrad's order (bitwise the same dw): table[i] = {slab address, dw address, elements, splits, first
 * block}, first block = running sum of vs_wgrad_reduce_blocks(elements).  Autograd accumulates one weight
 * gradient per layer behind each convolution's backward; here the 108 per-layer reduce launches of a
 * SlowFast-R50 step become one per backward segment. */
int vs_conv_wgrad_partial(const void* dy, const void* x, float* dw, const vs_conv_desc* d, void* slabs,
                          size_t slab_bytes, int* splits, void* stream);
int64_t vs_wgrad_reduce_blocks(int64_t elements);
/* The reduce of ONE layer's slabs (what vs_conv_wgrad launches behind its kernel): callers that split the two
 * launches can release the wgrad's operands -- and let a stream that waits for them go on -- before the reduce. */
int vs_wgrad_reduce(const float* slabs, float* dw, int64_t elements, int splits, void* stream);
int vs_wgrad_reduce_batched(const int64_t* table, int n_entries, int64_t total_blocks, void* stream);
/* Deferred slab reduce (round 3).  mode 1: a slab reduce issued behind vs_conv_wgrad / vs_conv_pair_end on this host
 * thread is not launched but waits (one slot) for the next vs_bn_bwd_finalize on the same stream, which issues ONE grid
 * holding both kernels' blocks -- in a unit's backward (autograd of slowfast's BottleneckTransform, mdl_sf_base.py:21-34)
 * they are neighbours on the stream and independent of each other; bitwise the two launches.  Flushed (launched alone) by
 * a second deferred reduce, by a weight gradient handed the same workspace or dw, by vs_wgrad_reduce_flush() and by
 * mode 0.  mode 2: suspended -- reduces are launched as usual and nothing is flushed (launches on a side lane, whose
 * completion event must cover the reduce).  Thread-local state. */
int vs_wgrad_reduce_defer(int mode);
int vs_wgrad_reduce_flush(void);

/* BatchNorm3d (mdl_sf_base.py:22-33 via slowfast BN modules; eps 1e-5, mom 0.1).
 * finalize: reduce conv-epilogue partials -> batch mean / biased var ->
 *   scale = gamma*rsqrt(var+eps), shift = beta - mean*scale; updates running
 *   stats with the unbiased variance (training) -- or, with nparts == 0, folds
 *   the running stats (eval). */
int vs_bn_finalize(const float* partials, int nparts, double count, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum,
                   float eps, float* scale, float* shift, float* mean, float* invstd, int C,
                   void* stream);
/* The same (training mode) for ANY number of partial rows in one launch: (C / 32) x G blocks sum row groups, the last
 * arriver of a channel group adds the group sums in group order and closes the channels.  workspace:
 * vs_bn_finalize_workspace_bytes() bytes, zero before its first use (left zero), one per stream that may run such
 * launches concurrently.  Bitwise stable from run to run; for <= 256 rows bitwise vs_bn_finalize.  VS_BN_FIN2=0 (or a
 * NULL workspace): vs_bn_partials_reduce + vs_bn_finalize as before. */
size_t vs_bn_finalize_workspace_bytes(void);
int vs_bn_finalize_ws(const float* partials, int nparts, double count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float* scale,
                      float* shift, float* mean, float* invstd, int C, void* workspace, size_t ws_bytes,
                      void* stream);
/* Level-1 reduction of the conv-epilogue partials when there are thousands of rows:
 * out[G][2][C] (then passed to vs_bn_finalize with nparts = G). */
int vs_bn_partials_reduce(const float* partials, int nparts, float* out, int C, int G,
                          void* stream);
/* out = relu?(y*scale[c] + shift[c] (+ residual)), bf16 rows of C channels. */
int vs_bn_apply(const void* y, const float* scale, const float* shift, const void* residual,
                void* out, int64_t rows, int C, int y_ld, int res_ld, int out_ld, int relu,
                void* stream);
/* vs_bn_apply with ReLU that also writes the ReLU mask as bits: relu_bits[rows][C/8] bytes, bit e of
 * byte (row, c/8) = out[row][c + e] > 0.  vs_bn_bwd_reduce / vs_bn_bwd_apply take it in place of z
 * with relu = 2 (16x fewer mask bytes than re-reading the output).  C/8 must be a power of two. */
int vs_bn_apply_mask(const void* y, const float* scale, const float* shift, const void* residual,
                     void* out, uint8_t* relu_bits, int64_t rows, int C, int y_ld, int res_ld,
                     int out_ld, void* stream);
/* out = relu(y * scale + shift + bf16(y2 * scale2 + shift2)), relu_bits (may be NULL) as in vs_bn_apply_mask: the last
 * unit of a ResBlock whose shortcut unit hands over its raw convolution output y2 and BN constants -- bitwise
 * vs_bn_apply(y2, scale2, shift2, relu = 0) followed by vs_bn_apply_mask(y, ..., residual = that), one pass less over a
 * block-output-sized tensor.  C / 8 a power of two. */
int vs_bn_apply2(const void* y, const float* scale, const float* shift, const void* y2, const float* scale2,
                 const float* shift2, void* out, uint8_t* relu_bits, int64_t rows, int C, int y_ld, int y2_ld,
                 int out_ld, void* stream);
/* Backward of z = relu?(bn(y) (+res)):  g = dz * [z>0];
 *   pass 1 (reduce): partial[blk][2][C] = (sum g, sum g*xhat), xhat=(y-mean)*invstd
 *   pass 2 (apply):  dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M); dres = g. */
/* z may be NULL when relu is set and the unit has no residual input: the mask is then
 * recomputed from y as gamma*xhat + beta > 0 (one read less per pass). */
int vs_bn_bwd_reduce(const void* dz, const void* z, const void* y, const float* mean,
                     const float* invstd, const float* gamma, const float* beta, float* partial,
                     int64_t rows, int C, int dz_ld, int z_ld, int y_ld, int relu, void* stream);
int vs_bn_bwd_reduce_rows(int64_t rows, int C);
int vs_bn_bwd_finalize(const float* partial, int nparts, float* dgamma, float* dbeta, int C,
                       void* stream);
/* The same with a workspace (vs_bn_finalize_workspace_bytes(), as for vs_bn_finalize_ws): the rows are summed by
 * (C / 32) x G blocks in one launch instead of C / 32 blocks walking all of them. */
int vs_bn_bwd_finalize_ws(const float* partial, int nparts, float* dgamma, float* dbeta, int C, void* workspace,
                          size_t ws_bytes, void* stream);
int vs_bn_bwd_apply(const void* dz, const void* z, const void* y, const float* mean,
                    const float* invstd, const float* gamma, const float* beta,
                    const float* dgamma, const float* dbeta, void* dy, void* dres, int64_t rows,
                    int C, int dz_ld, int z_ld, int y_ld, int dy_ld, int dres_ld, int relu,
                    void* stream);
/* The backward apply of TWO units fed by the same masked gradient (a ResBlock's last unit and its shortcut unit:
 * g = dz where the block's ReLU bit is set) in one pass: bitwise two vs_bn_bwd_apply(relu = 2) launches, dz and the bits
 * read once.  C / 8 a power of two. */
int vs_bn_bwd_apply2(const void* dz, const uint8_t* relu_bits, const void* y_a, const float* mean_a,
                     const float* invstd_a, const float* gamma_a, const float* dgamma_a, const float* dbeta_a,
                     void* dy_a, const void* y_b, const float* mean_b, const float* invstd_b,
                     const float* gamma_b, const float* dgamma_b, const float* dbeta_b, void* dy_b, int64_t rows,
                     int C, int dz_ld, int ya_ld, int dya_ld, int yb_ld, int dyb_ld, void* stream);

/* fp32 residual stream of a stage (eval mode, optional): out32 = relu?(residual + branch), out16 = bf16(out32),
 * rows of C channels.  `branch` = the bottleneck's last unit with its BatchNorm folded (bf16, no residual, no
 * ReLU); the residual is either the previous block's fp32 output (res32) or the shortcut unit's bf16 output
 * (res16) -- exactly one of them.  Upstream `ResBlock.forward`: x = relu(branch1(x) | x  +  branch2(x))
 * (slowfast resnet_helper via mdl_sf_base.py:21-34), which the reference evaluates in fp32; with bf16 activations
 * the identity chain otherwise rounds once per block (north_star: logits within 1e-3).  r_ld counts elements of
 * whichever residual is given. */
int vs_residual_add_f32(const void* branch, const float* res32, const void* res16, float* out32, void* out16,
                        int64_t rows, int C, int b_ld, int r_ld, int o32_ld, int o16_ld, int relu, void* stream);

/* MaxPool3d([1,3,3], s[1,2,2], p[0,1,1]) of the stems (slowfast stem_helper via
 * mdl_sf_base.py:22).  idx: uint8 argmax tap (first max in (kh,kw) scan order,
 * as torch) per output element, used by the backward. */
int vs_maxpool_hw3s2_fwd(const void* x, void* y, uint8_t* idx, int N, int T, int H, int W, int C,
                         int x_ld, int y_ld, void* stream);
int vs_maxpool_hw3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int T, int H, int W,
                         int C, int dy_ld, int dx_ld, void* stream);
/* The stem's BatchNorm + ReLU + max-pool [1,3,3] / [1,2,2] / pad [0,1,1] (upstream ResNetBasicStem; mdl_sf_base.py:22
 * runs it as s1) without the full-resolution normalised tensor: one forward pass from the conv output to the pooled
 * tensor (+ argmax bytes), and the two BN-backward passes with the pool's gradient gathered from (d_pooled, idx)
 * instead of read from a dense tensor.  Bitwise vs_bn_apply + vs_maxpool_hw3s2_fwd and vs_maxpool_hw3s2_bwd +
 * vs_bn_bwd_reduce / vs_bn_bwd_apply (relu = 1, mask recomputed from y).  Fewer than 2^24 positions, C/8 a power
 * of two; VS_ERR_UNSUPPORTED / VS_ERR_BAD_ARG otherwise. */
int vs_bn_apply_maxpool(const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, int N, int T,
                        int H, int W, int C, int y_ld, int out_ld, void* stream);
int vs_bn_bwd_reduce_pool(const void* d_pooled, const uint8_t* idx, const void* y, const float* mean,
                          const float* invstd, const float* gamma, const float* beta, float* partial, int N, int T,
                          int H, int W, int C, int dp_ld, int y_ld, void* stream);
int vs_bn_bwd_apply_pool(const void* d_pooled, const uint8_t* idx, const void* y, const float* mean,
                         const float* invstd, const float* gamma, const float* beta, const float* dgamma,
                         const float* dbeta, void* dy, int N, int T, int H, int W, int C, int dp_ld, int y_ld,
                         int dy_ld, void* stream);
/* MaxPool3d k=s=[kt,1,1] (pathway0_pool of c2d / i3d, mdl_sf_base.py:49-51). */
int vs_maxpool_t_fwd(const void* x, void* y, uint8_t* idx, int N, int T, int HW, int C, int kt,
                     void* stream);
int vs_maxpool_t_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int T, int HW, int C,
                     int kt, void* stream);

/* AdaptiveAvgPool3d((1,1,1)) + channel concat (mdl_sf_base.py:97-113):
 * out[n][c_off + c] = mean over `rows_per_clip` positions, fp32 out. */
int vs_avgpool_fwd(const void* x, float* out, int N, int64_t rows_per_clip, int C, int x_ld,
                   int out_ld, int c_off, void* stream);
int vs_avgpool_bwd(const float* dout, void* dx, int N, int64_t rows_per_clip, int C, int dx_ld,
                   int dout_ld, int c_off, void* stream);

/* fp32 Linear for M <= 64*k rows: y[M,N] = act(x[M,K] @ W[N,K]^T + b).
 * Replaces nn.Linear of proj_head / vid_feat_encoder (mdl_sf_base.py:161-167,
 * 767-769) and of utils/transformer_code.py:51-79 (wq/wk/wv/wo, linear1/2). */
int vs_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K,
                  int relu, void* stream);
/* dx[M,K] = dy[M,N] @ W[N,K], given wt = W^T [K,N] from vs_transpose_f32;
 * dw[N,K] = dy^T x;  db[N] = sum_m dy. */
int vs_transpose_f32(const float* w, float* wt, int R, int C, void* stream);
int vs_linear_bwd_data(const float* dy, const float* wt, float* dx, int M, int N, int K,
                       void* stream);
int vs_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int M, int N,
                         int K, void* stream);
/* Both of the above behind ONE launch for M <= 8 rows (N % 4 == 0, N <= 4096, 16-byte aligned dy / relu_y / wt):
 * dx = dy_eff . W from the transposed image wt [K][N], dW = dy_eff^T x, db = column sums of dy_eff, where
 * dy_eff = dy * (relu_y > 0) if relu_y (the layer's ReLU output) is given, else dy.  Bitwise the separate
 * launches (relu backward, vs_linear_bwd_data, vs_linear_bwd_weight): the 8-token encoder / head section is a
 * chain of ~6 us launches on the critical path of the step (LinearFn.backward of utils/transformer_code.py:61-75's
 * Linear layers).  VS_ERR_UNSUPPORTED outside that envelope. */
int vs_linear_bwd_fused(const float* dy, const float* relu_y, const float* x, const float* wt, float* dx,
                        float* dw, float* db, int M, int N, int K, void* stream);
/* The same with dx = dy_eff . W + dx_res (dx_res [M,K], nullable): the gradient that reaches the layer's input over a
 * residual connection (utils/transformer_code.py:26-31, ResidualBlock: x + dropout(layer(x))) joins in the product's
 * epilogue instead of in an add launch of autograd's -- the same single rounding. */
int vs_linear_bwd_fused_res(const float* dy, const float* relu_y, const float* x, const float* wt, const float* dx_res,
                            float* dx, float* dw, float* db, int M, int N, int K, void* stream);

/* One launch for a unit's data gradient and weight gradient (the backward of a conv + BN unit runs both on the same dy;
 * slowfast ResBlock / BottleneckTransform backward via mdl_sf_base.py:21-34).  Between vs_conv_pair_begin() and
 * vs_conv_pair_end() the conv entry points RECORD an eligible launch (dgrad on the 128 x 128 tile with the two-stage
 * ring, no split-K; weight gradient on the 128 x 128 two-stage ring) instead of issuing it; _end issues the two as
 * ONE grid -- first the dgrad's tiles, then the weight gradient's blocks -- followed by the weight gradient's slab
 * reduce, or issues what was recorded alone.  Bitwise the separate launches.  Why: a fork + join pair inside a
 * replayed hipGraph costs ~17 us (profiles/r02_graph_edge_cost.txt).  Thread-local state; always call _end.
 * vs_conv_pair_count(): launches so far that held both kernels. */
int vs_conv_pair_begin(void);
int vs_conv_pair_end(void);
int64_t vs_conv_pair_count(void);

/* softmax(Q_h K_h^T / scale) V_h for short sequences (L <= 16), per head.
 * q,k,v,o: [B, L, H*dh] fp32.  utils/transformer_code.py:33-48,60-68 --
 * scale is sqrt(d_model) there, passed explicitly. */
/* drop_mask (nullable): [B,H,L,L] fp32 holding 0 or 1/(1-p), the train-mode dropout of the
 * attention probabilities (transformer_code.py:48); probs keeps the pre-dropout softmax. */
/* ld_qkv: row pitch (floats) of q, k, v -- and of dq, dk, dv -- e.g. 3*H*dh when they are the three column
 * blocks of one fused projection buffer [B*L, 3*H*dh]; 0 = dense (H*dh).  o / dout are dense. */
int vs_attn_small_fwd(const float* q, const float* k, const float* v, float* o, float* probs,
                      const float* drop_mask, int B, int L, int H, int dh, int ld_qkv, float scale,
                      void* stream);
int vs_attn_small_bwd(const float* q, const float* k, const float* v, const float* probs,
                      const float* dout, float* dq, float* dk, float* dv, const float* drop_mask,
                      int B, int L, int H, int dh, int ld_qkv, float scale, void* stream);

/* y = LayerNorm(x + r*rmask) (utils/transformer_code.py:21-30: x + dropout(layer(x))), fp32.
 * rmask (nullable): [rows,D] holding 0 or 1/(1-p).  bwd: dx = d(x), dr = dx*rmask. */
int vs_add_layernorm_fwd(const float* x, const float* r, const float* rmask, const float* gamma,
                         const float* beta, float* y, float* mean, float* rstd, int rows, int D,
                         float eps, void* stream);
int vs_add_layernorm_bwd(const float* dy, const float* x, const float* r, const float* rmask,
                         const float* gamma, const float* mean, const float* rstd, float* dx,
                         float* dr, float* dgamma, float* dbeta, int rows, int D, void* stream);

/* Fused cross-entropy (mean) fwd + dlogits (mdl_sf_base.py:226-231) and the
 * softmax -> descending sort -> top-k verb indices of EvalB
 * (vidsitu_code/evl_vsitu.py:39-42). */
int vs_softmax_xent(const float* logits, const int64_t* labels, float* loss, float* dlogits,
                    int rows, int V, void* stream);
int vs_softmax_topk(const float* logits, float* probs_out, int64_t* idx_out, int rows, int V,
                    int k, void* stream);

/* Adam (main_dist.py:50: betas (0.9, 0.99), eps 1e-8, no weight decay) on a flat
 * fp32 parameter / gradient arena; grad_scale folds the DDP 1/world_size. */
int vs_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, int step, float grad_scale, void* stream);
/* Same, with the step count kept in device memory (incremented by the call): safe to capture
 * in a hipGraph and replay. */
int vs_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                     float beta1, float beta2, float eps, int* step_counter, float grad_scale,
                     void* stream);
/* Same step, also writing the bf16 kernel-layout copy of the updated parameters (p_bf16[n]). */
int vs_adam_step_dev_cast(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n,
                          float lr, float beta1, float beta2, float eps, int* step_counter,
                          float grad_scale, void* stream);
/* Same step with the gradients read from a bf16 buffer: the summed payload of a bf16 gradient
 * all-reduce (the reference's DDP reducer moves fp32 buckets, main_dist.py:68-79; half the xGMI
 * bytes here).  Parameters and both moments stay fp32. */
int vs_adam_step_dev_cast_g16(float* p, const void* g_bf16, float* m, float* v, void* p_bf16, int64_t n,
                              float lr, float beta1, float beta2, float eps, int* step_counter,
                              float grad_scale, void* stream);
/* The same update for ONE RANGE of the arena, without touching the step count: a step whose gradient ranges
 * become final at different times (the segments of vidsitu_amd/train_step.py) ticks once (vs_adam_tick) and then
 * updates each range as soon as it is final, beside the rest of the backward pass.  g: fp32, or bf16 when
 * g_is_bf16; p_bf16 may be NULL.  Replaces the same optimizer.step() of utils/trn_utils.py:590-615. */
int vs_adam_tick(int* step_counter, void* stream);
int vs_adam_step_dev_range(float* p, const void* g, int g_is_bf16, float* m, float* v, void* p_bf16, int64_t n,
                           float lr, float beta1, float beta2, float eps, const int* step_counter,
                           float grad_scale, void* stream);
/* fp32 -> bf16 cast of the parameter arena (weights used by the conv kernels). */
int vs_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);

/* ---- GPT-2 decoder + beam-search scoring (fp32) --------------------------------------------
 * Replaces huggingface GPT2LMHeadModel as called by HuggingFaceGPT2Decoder
 * (vidsitu_code/hf_gpt2_fseq.py:150,165-203) and the per-step scoring of SeqGenCustom._generate
 * (vidsitu_code/seq_gen.py:310-385).  Weights are kept K-contiguous ([out][in], nn.Linear
 * layout; the host mirror transposes HF Conv1D tensors on load). */
/* y[M,N] = act(x[M,K] . w[N,K]^T + b[N]) + res[M,N];  b, res may be NULL; act 0 none, 1 relu,
 * 2 gelu_new.  M <= 16: one wave streams one weight row; 17..64 rows (K % 128 == 0): skinny fp32-MFMA
 * tiles straight from global memory; else 128x128 fp32 MFMA tiles through LDS. */
int vs_gemm_nt_f32(const float* x, const float* w, const float* b, const float* res, float* y, int M,
                   int N, int K, int act, void* stream);
/* Frame resize of the loader, `img.resize((224, 224))` in VsituDS.read_img
 * (vidsitu_code/dat_loader.py:183-191) = Pillow's two-pass bicubic resampling of 8-bit RGB (Resample.c,
 * pillow=7.2.0), bit for bit.  vs_resize_ksize / vs_resize_coeffs fill HOST tables for one axis
 * (bounds[out][2] = first source index and count, kk[out][ksize] = 22-bit fixed-point weights, computed
 * in double precision as Pillow does); the caller uploads them.  vs_resize_bicubic_u8: src u8
 * [frames][H0][W0][3] -> dst u8 [frames][Ho][Wo][3]; y0, y1 = first / one-past-last source row the
 * vertical pass reads (bounds_v[0][0] and bounds_v[Ho-1][0] + bounds_v[Ho-1][1]); tmp holds
 * frames * (y1 - y0) * Wo * 3 bytes (vs_resize_tmp_bytes gives the H0-row upper bound); a pass whose
 * size does not change is skipped, as in Pillow. */
int vs_resize_ksize(int in_size, int out_size);
int vs_resize_coeffs(int in_size, int out_size, int32_t* bounds_host, int32_t* kk_host);
size_t vs_resize_tmp_bytes(int64_t frames, int H0, int Wo);
int vs_resize_bicubic_u8(const uint8_t* src, uint8_t* dst, uint8_t* tmp, int64_t frames, int H0, int W0,
                         int Ho, int Wo, const int32_t* bounds_h, const int32_t* kk_h, int ksize_h,
                         const int32_t* bounds_v, const int32_t* kk_v, int ksize_v, int y0, int y1,
                         void* stream);

/* vs_gemm_nt_f32 with a split-K workspace: GEMMs of more than 64 rows with few 64x64 output tiles (a
 * 600-token batch against a 1024-wide layer) are cut into K slices whose partial tiles are added in slice
 * order by a second launch (bitwise reproducible).  vs_gemm_nt_f32_workspace_bytes: bytes needed (0 = the
 * plain call does the same). */
size_t vs_gemm_nt_f32_workspace_bytes(int M, int N, int K);
int vs_gemm_nt_f32_ws(const float* x, const float* w, const float* b, const float* res, float* y, int M,
                      int N, int K, int act, void* workspace, size_t ws_bytes, void* stream);

/* Decode-step (1..64 rows) GEMM on fragment-major operands.  vs_pack_rows_f32 copies a row-major
 * matrix src[R][K] (K % 16 == 0) into 1-KB blocks of 16 rows x 16 floats in v_mfma_f32_16x16x4_f32
 * operand order (block (r/16, k/16) at index (r/16)*(K/16) + k/16; lane (k%16/4)*16 + r%16 holds floats
 * k%4 = 0..3; rows up to the next multiple of 16 are zero): dst holds ceil16(R) * K floats.  Weights are
 * packed once; the activations are written in this layout by their producers (vs_layernorm_fwd_packed,
 * vs_attn_decode(out_packed), vs_gemm_nt_f32_packed(y_packed)), so every fragment load of the GEMM is
 * one contiguous KB.  y = act(x . w^T + b) + res as vs_gemm_nt_f32; res, y row-major unless y_packed;
 * K % 128 == 0; a packed y needs N % 16 == 0. */
int vs_pack_rows_f32(const float* src, float* dst, int R, int K, void* stream);
int vs_gemm_nt_f32_packed(const float* x_packed, const float* w_packed, const float* b, const float* res,
                          float* y, int M, int N, int K, int act, int y_packed, void* stream);
/* LayerNorm(x) * gamma + beta written fragment-major (D % 16 == 0, D <= 2048). */
int vs_layernorm_fwd_packed(const float* x, const float* gamma, const float* beta, float* y_packed,
                            int rows, int D, float eps, void* stream);
/* out[r,l,:] = wte[tokens[r,l]] + wpe[pos0 + l]  (GPT2Model embeddings, default position ids). */
int vs_gpt2_embed(const int64_t* tokens, const float* wte, const float* wpe, float* out, int R, int L,
                  int D, int pos0, int V, void* stream);
/* Causal self-attention on the fused c_attn output qkv[R,L,3D]; key_mask[R,L] (1 = attend, NULL =
 * all); masked scores are -1e4 exactly as modeling_gpt2 does; out[R,L,D] with heads merged. */
int vs_attn_causal_fwd(const float* qkv, const uint8_t* key_mask, float* out, int R, int L, int H,
                       int dh, void* stream);
/* One incremental step: append k,v of qkv[rows,3D] at position t of the caches
 * [rows][H][Lmax][dh], attend the new query over 0..t; key_mask[rows,Lmax] or NULL.
 * ancestry[rows][Lmax] (or NULL): position j < t of row r is read from cache row ancestry[r][j] --
 * a beam reorder then permutes this table (vs_beam_step) instead of gathering the cache.
 * out_packed: out is written fragment-major (vs_pack_rows_f32) for vs_gemm_nt_f32_packed. */
int vs_attn_decode(const float* qkv, float* kcache, float* vcache, const uint8_t* key_mask,
                   const int32_t* ancestry, float* out, int rows, int H, int dh, int Lmax, int t,
                   int out_packed, void* stream);
/* dst[r] = src[index[r]] for the first len positions of every head (beam reorder of a cache). */
int vs_kv_gather(const float* src, float* dst, const int64_t* index, int rows_out, int H, int dh,
                 int Lmax, int len, void* stream);
/* Beam-search step scoring: lp = log_softmax(logits/T), NaN -> -inf, lp[pad] = -inf,
 * lp[unk] -= unk_penalty, flags&1: only eos, flags&2: eos banned, forced[r] >= 0 (and != pad):
 * only that token; + cum[r]; the k (<= 32) best (value, token) per row, descending, ties ->
 * lowest token id.  cum, forced may be NULL.  With a workspace (vs_beam_topk_workspace_bytes) rows
 * of more than 2048 tokens are cut into slices (slices x rows blocks); NULL = one block per row. */
size_t vs_beam_topk_workspace_bytes(int rows, int V, int k);
int vs_beam_topk(const float* logits, const float* cum, const int64_t* forced, float* out_val,
                 int64_t* out_idx, int rows, int V, int k, int pad, int eos, int unk,
                 float unk_penalty, float temperature, int flags, void* workspace, size_t ws_bytes,
                 void* stream);
/* Mean token cross entropy with ignore_index (Simple_TxDec.forward, mdl_sf_base.py:660-664):
 * nll_rows[rows] scratch, loss_out[2] = {mean nll over counted rows, count}; ld = row pitch. */
int vs_xent_ignore(const float* logits, const int64_t* labels, float* nll_rows, float* loss_out,
                   int rows, int V, int64_t ld, int ignore_index, void* stream);

/* ---- GPT-2 decoder backward (fine-tuning the LM; Simple_TxDec.forward, mdl_sf_base.py:653-667) */
int vs_gelu_new_fwd(const float* x, float* y, int64_t n, void* stream);
int vs_gelu_new_bwd(const float* dy, const float* x_pre, float* dx, int64_t n, void* stream);
int vs_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
/* out[N] = column sums of x[M,N] (bias gradients), fixed row order. */
int vs_colsum_f32(const float* x, float* out, int M, int N, void* stream);
/* Backward of vs_attn_causal_fwd: dqkv[R,L,3D] from dout[R,L,D]; scratch of
 * vs_attn_causal_bwd_scratch_bytes(R,L,H) bytes (p and dS rows); no atomics. */
size_t vs_attn_causal_bwd_scratch_bytes(int R, int L, int H);
int vs_attn_causal_bwd(const float* qkv, const uint8_t* key_mask, const float* dout, float* dqkv,
                       void* scratch, size_t scratch_bytes, int R, int L, int H, int dh, void* stream);
/* dwte[tokens] += dh, dwpe[pos0 + l] += dh (fp32 atomics). */
int vs_gpt2_embed_bwd(const int64_t* tokens, const float* dh, float* dwte, float* dwpe, int R, int L,
                      int D, int pos0, int V, void* stream);
/* dlogits of vs_xent_ignore: (softmax - onehot) * grad_scale / count; loss_out from vs_xent_ignore. */
int vs_xent_ignore_grad(const float* logits, const int64_t* labels, const float* loss_out,
                        float* dlogits, int rows, int V, int64_t ld, int ignore_index, float grad_scale,
                        void* stream);
/* The same with the upstream gradient of the scalar loss read from device memory (autograd hands it over as a
 * 0-dim device tensor): no host synchronisation in the backward pass, so the step can be captured in a hipGraph. */
int vs_xent_ignore_grad_dev(const float* logits, const int64_t* labels, const float* loss_out,
                            float* dlogits, int rows, int V, int64_t ld, int ignore_index,
                            const float* grad_scale_dev, void* stream);

/* Device-side beam-search bookkeeping of one step (SeqGenCustom._generate between two decoder
 * calls, seq_gen.py:368-520, and finalize_hypos :579-697), one block per sentence, no host sync:
 * merges the per-row lists of vs_beam_topk into the 2*beam candidates (ties -> lowest beam*V+token),
 * finalizes eos hypotheses into fin_* ([bsz][beam][max_len+1] tokens / positional scores, score,
 * length), marks finished sentences (remaining[0] counts the others), picks the next beam's live
 * candidates, writes the gathered token / score rows (ping-pong buffers, row pitch max_len+2 /
 * max_len+1) and the parent-row index `reorder` for vs_kv_gather.  Finished sentences stay in the
 * batch as idle rows.  anc_in / anc_out ([bsz*beam][anc_ld] ping-pong, or both NULL): the ancestry
 * table of vs_attn_decode, gathered like the token rows, entry step+1 of a new row = the row itself. */
int vs_beam_step(const float* row_val, const int64_t* row_idx, const int64_t* tok_in, int64_t* tok_out,
                 const float* sc_in, float* sc_out, uint8_t* ignore, uint8_t* finished, int* nfin,
                 int* remaining, int64_t* fin_tok, float* fin_score, float* fin_pos, int* fin_len,
                 int64_t* reorder, const int32_t* anc_in, int32_t* anc_out, int anc_ld, int bsz, int beam,
                 int k, int V, int step, int max_len, int eos, int normalize, float len_penalty,
                 void* stream);

/* Non-local block pieces (slowfast nonlocal_helper.Nonlocal of the i3d_r50_nl_8x8 feature model,
 * Kinetics_c2_I3D_NLN_8x8_R50.yaml:25-28): MaxPool3d([1,2,2], stride [1,2,2]) of a channels-last bf16
 * activation [NT][H][W][C] with a byte argmax per element (0..3 = (dh, dw), first maximum wins) and its
 * backward; row softmax of bf16 scores [rows][P] (P % 4 == 0, P <= 4096; fp32 math; in place allowed) and
 * ds = scale * p * (dp - sum(dp * p)); out[c] = sum over rows of a bf16 matrix (row pitch ld).  The
 * theta.phi^T and softmax.g products run on vs_conv_fwd / vs_conv_wgrad per clip. */
int vs_maxpool_hw2_fwd(const void* x, void* y, uint8_t* idx, int64_t NT, int H, int W, int C, void* stream);
int vs_maxpool_hw2_bwd(const void* dy, const uint8_t* idx, void* dx, int64_t NT, int H, int W, int C,
                       void* stream);
int vs_softmax_rows_bf16(const void* x, void* y, int64_t rows, int P, void* stream);
int vs_softmax_rows_bwd_bf16(const void* p, const void* dp, void* ds, int64_t rows, int P, float scale,
                             void* stream);
int vs_colsum_bf16(const void* x, float* out, int64_t rows, int C, int ld, void* stream);

/* fairseq TransformerDecoder (TxDecoderReal, vidsitu_code/mdl_sf_base.py:435-446) pieces beside the
 * shared GEMM / attention / layernorm kernels: out[t] = scale * emb[tokens[t]] + pos_table[pos_idx[t]]
 * (embed_scale * embed_tokens + sinusoidal positions; the caller's table has a zero row for padding);
 * demb += scale * dx per non-padding token (nn.Embedding(padding_idx) backward; demb zero-filled by the
 * caller); dx = y > 0 ? dy : 0 (relu FFN). */
int vs_embed_pos_fwd(const int64_t* tokens, const float* emb, const float* pos_table, const int64_t* pos_idx,
                     float* out, int64_t n_tok, int D, float scale, void* stream);
int vs_embed_scatter_bwd(const int64_t* tokens, const float* dx, float* demb, int64_t n_tok, int D, float scale,
                         int64_t pad, void* stream);
int vs_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIDSITU_HIP_H */
