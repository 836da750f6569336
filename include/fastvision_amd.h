/*
 * fastvision_amd.h -- C ABI of libfastvision_amd.so: the MI355X (gfx950) kernels behind the
 * YOLOv3 training hot path of ielym/fastvision.
 *
 * The reference is pure Python on PyTorch and has NO FFI of its own: the boundary a
 * maintainer would bind is the set of ATen ops its modules issue.  Each entry point below names the
 * reference call site it replaces (paths relative to the reference root).  INTEGRATION.md shows the
 * ctypes binding the reference side would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller for the
 *     duration of the call (borrowed from torch.Tensor.data_ptr()); outputs are preallocated;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); nothing here
 *     synchronises the host, allocates, or copies to the host;
 *   - every function returns 0 on success, a negative fva_status otherwise; fva_last_error()
 *     returns a static message for the calling thread;
 *   - dtype: FVA_F32 (exact fp32, f32-input MFMA) or FVA_BF16 (bf16 storage, fp32 accumulate).
 *
 * Activation layout ("halo NHWC"): a feature map of logical shape [B,C,H,W] is stored as
 * [B][H+2*pad][W+2*pad][C] with a zero border of `pad` pixels (pad is 1 for every map a 3x3 conv or
 * its dgrad reads, 0 otherwise).  Channels are contiguous.  Raw conv outputs (pre-BN) are dense
 * [B*OH*OW][Cout] (pad 0).
 */
#ifndef FASTVISION_AMD_H
#define FASTVISION_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { FVA_F32 = 0, FVA_BF16 = 1 } fva_dtype;

typedef enum {
    FVA_OK = 0,
    FVA_ERR_ARG = -1,      /* unsupported shape / null pointer / bad enum */
    FVA_ERR_LAUNCH = -2,   /* hipLaunchKernel or hipGetLastError failed */
    FVA_ERR_WORKSPACE = -3 /* workspace too small: call the matching *_workspace() */
} fva_status;

const char* fva_last_error(void);
int fva_version(void);

/* Live timing of the MFMA convolution entry points with HIP events on their launch stream, taken inside the library (no
 * per-call host work in the caller).  fva_profile_start(max_spans) creates the events -- call it OUTSIDE the region being
 * timed -- and arms the spans; fva_profile_stop() synchronises the device, disarms and returns the number of spans with
 * cls (low byte: 0 forward incl. head, 1 dgrad, 2 wgrad incl. its reduce, 3 a forward launch that also carries the apply pass of the
 * block before it (fva_conv1x1_fwd_apply[_acc]; FLOPs = the convolution's); bits 8..: the layer's kernel size), algorithmic
 * FLOPs and elapsed milliseconds of each.
 * fva_profile_classes(mask, stride) restricts the spans to the classes whose bit is set and, of those calls, to every
 * stride-th one (default: all classes, stride 1).  A span costs two event packets on the stream and the kernels on either
 * side of them no longer overlap their launch with the neighbour's tail (measured: ~7 us of GPU time per span, 3 % of the
 * train step when every forward convolution is bracketed), so a throughput measurement brackets a sample of the kernel
 * class it reports. */
int fva_profile_start(int32_t max_spans);
int fva_profile_classes(uint32_t mask, int32_t stride);

/* A low-priority side stream for work that nothing waits for until the end of the backward pass (the weight gradients).
 * fork: the side stream (returned) waits for everything enqueued on main_stream so far.  join: main_stream waits for
 * everything enqueued on the side stream so far.  Buffers the side stream reads or writes must stay alive until a join.
 * renew: drain and drop the side stream; the next fork creates a fresh one (how a stream maps onto the hardware queues is decided at its
 * creation, and a stream that landed badly no longer yields to the launch stream -- ops.autotune_wgrad_side_stream() asks for another). */
int fva_side_stream_fork(void* main_stream, void** side_stream);
int fva_side_stream_join(void* main_stream);
int fva_side_stream_renew(void);
int32_t fva_profile_stop(int32_t* cls, double* flop, float* ms, int32_t cap);

/* ------------------------------------------------------------------------------------------------
 * Convolution (no bias), implicit GEMM on MFMA.  Replaces nn.Conv2d as used by ConvBlock3x3 /
 * ConvBlock1x1 (classfication/models/darknet53.py:5-9,22-44; detection/neck/yolov3neck.py:5-9,22-44;
 * demos/yolov3_u/models/{darknet,yolov3}.py:6-40) and its autograd backward.
 * ksize in {1,3}, padding = ksize/2, stride in {1,2}, groups 1, dilation 1.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dtype;   /* fva_dtype of activations and packed weights */
    int32_t B, H, W; /* logical input size */
    int32_t Cin, Cout;
    int32_t ksize, stride;
    int32_t in_pad;  /* zero border of the input buffer  (>= ksize/2) */
    int32_t dy_pad;  /* zero border of the dY buffer used by dgrad/wgrad (>= 1 when ksize==3) */
} fva_conv_desc;

/* fp32 OIHW master weights [Cout][Cin][k][k] -> the two packed operand layouts (dtype of desc):
 *   w_fwd [k*k (+1 pad tap if Cin==32 && bf16)][Cout][Cin]   and   w_dgrad [k*k (+pad)][Cin][Cout].
 * Either output may be NULL.  Sizes in elements: fva_conv_packed_elems(). */
int fva_conv_pack_weights(const fva_conv_desc* d, const float* w_oihw, void* w_fwd, void* w_dgrad, void* stream);
int64_t fva_conv_packed_elems(const fva_conv_desc* d, int for_dgrad);
/* The same for every conv layer of a model in ONE launch (after each optimizer step): `table` is a DEVICE array of n
 * entries; taps_fwd / taps_dgrad = fva_conv_packed_elems(...) / (Cout*Cin); max_elems = largest taps*Cout*Cin. */
typedef struct {
    const float* w;       /* fp32 OIHW master weight */
    void* w_fwd;          /* [taps_fwd][Cout][Cin]   (may be NULL) */
    void* w_dgrad;        /* [taps_dgrad][Cin][Cout] (may be NULL) */
    int32_t Cout, Cin, ksize, taps_fwd, taps_dgrad, dtype;
    int32_t dgrad_paired; /* 1 when taps_dgrad == 12: the paired stride-2 layout [6][2*Cin][Cout] thin layers use
                           * (fva_conv_packed_elems() of a stride-2 descriptor tells; see conv_igemm.hip dgrad_paired) */
    int32_t tile_start;   /* fva_conv_pack_weights_tiled: 32 x 32 (Cout x Cin) tiles of the entries BEFORE this one, i.e. the running sum of
                           * ceil(Cout / 32) * ceil(Cin / 32); ignored by fva_conv_pack_weights_multi */
} fva_pack_entry;
int fva_conv_pack_weights_multi(const fva_pack_entry* table, int32_t n, int64_t max_elems, void* stream);
/* The same result from a launch of ONE block per 32 x 32 weight tile of any layer (total_tiles = the running sum after the last
 * entry): no idle blocks for small layers, 16-byte loads and 8-byte stores on aligned bf16 tiles. */
int fva_conv_pack_weights_tiled(const fva_pack_entry* table, int32_t n, int32_t total_tiles, void* stream);

/* Which MFMA kernel the calling thread's last convolution entry point launched: "igemm8" (256x256 8-phase), "igemm128", "igemm256x64",
 * "pconv", "pdgrad2" (patch kernels); "wgrad8", "wgrad128", "wgrad128thin", "wgrad64f32", "pwgrad".  For tests that must know what they compared. */
const char* fva_conv_last_kernel(void);

/* Diagnostic: while set (non-NULL), every block of an 8-phase convolution launch writes eight values to stamps[block * 8 ..]:
 * wall_clock64 (100 MHz) at block entry, first k-tile ready, k-loop done and exit, then the shader-clock cycle counter at the
 * same four points (cycles / wall time = the shader clock under load).  `rows` = capacity of the buffer in blocks (8 values
 * each): blocks beyond it do not stamp.  NULL switches the stamps off.  With FVA_STAMP_IGEMM=1 in the environment the 128x128 /
 * 256x64 kernels stamp too, eight wall-clock values per block: entry, first DMA issued, first k-tile landed, k loop done, tile staged
 * in LDS, exit, source rows ready, epilogue operands requested (tools/tile_timing.py pw). */
int fva_conv_debug_stamps(void* stamps, int32_t rows);
/* The thin 3x3 stride-1 bf16 layers (reduction channels <= 128, outputs <= 128, maps of 64 x 64 and more) run on the patch kernel
 * (conv_igemm.hip pconv_kernel: the input patch of an 8 x 32 output tile is staged once for the nine taps).  This switches it off /
 * on for the process (default on; env FVA_PCONV=0) and returns the previous setting: fva_conv_stat_blocks / fva_conv_dgrad_stat_rows
 * follow the setting, so do not change it between sizing a table and the launch that fills it. */
int fva_conv_patch_kernel(int on);

/* y[B*OH*OW][Cout] = conv(x) (dense, dtype).  If stats_partial != NULL also writes per-row-block
 * partial sums for BatchNorm: stats_partial[blk][0][c] = sum_y, [blk][1][c] = sum_y^2 over the rows of
 * that block (blk < fva_conv_stat_blocks()); they are reduced by fva_bn_finalize(). */
int fva_conv_fwd(const fva_conv_desc* d, const void* x, const void* w_fwd, void* y, float* stats_partial, void* stream);
/* Training forward of a 1x1 layer whose input z has not been produced yet: z = SiLU(y_prev * scale + shift) (+ residual) -- the
 * apply pass of the block before it (fva_bn_silu_apply: same arithmetic, same bits) -- is computed in this launch's operand path,
 * written ONCE to the halo buffer z (border d->in_pad <= 1 included, for every other reader of z: the residual identity, the weight
 * gradient, the next block) and fed to the MFMAs from LDS without being read back: the separate apply launch and one read of z go
 * (classfication/models/darknet53.py:46-63: a residual block's conv1 always follows the previous block's SiLU + add).  y / statistics
 * as fva_conv_fwd (same tiles, same rows, same bits).  bf16; Cin % 64 == 0, Cin <= 512; Cout <= 128 (one column block, so that every element
 * is transformed once); y_prev dense [B*H*W][Cin]; residual (optional) a halo buffer of z's shape with border res_pad. */
int fva_conv1x1_fwd_apply(const fva_conv_desc* d, const void* y_prev, const float* scale, const float* shift, const void* residual,
                          int32_t res_pad, void* z, const void* w_fwd, void* y, float* stats_partial, void* stream);
/* Inference form: eval-mode BatchNorm folded into a per-channel affine and SiLU applied in the convolution's epilogue,
 * z = SiLU(conv(x) * scale[c] + shift[c]) (+ residual), written straight into the halo buffer z [B][OH+2p][OW+2p][Cout]
 * (interior by the MFMA kernel, zero border by a small second launch).  residual (optional) has z's geometry.
 * Replaces conv + bn (running statistics) + SiLU (+ add) of ConvBlock / ResidualBlock in eval mode
 * (classfication/models/darknet53.py:28-31,58-62) with one pass over the output instead of three. */
int fva_conv_fwd_bnact(const fva_conv_desc* d, const void* x, const void* w_fwd, const float* scale, const float* shift,
                       const void* residual, void* z, int32_t z_pad, void* stream);
int32_t fva_conv_stat_blocks(const fva_conv_desc* d);

/* dx[B][H][W][Cin] (dense, dtype) = conv_transpose(dy, w) (+ addend).  dy is halo NHWC with border d->dy_pad.
 * addend (optional, dense like dx, may alias dx) is added in the epilogue: the residual-branch gradient sum. */
int fva_conv_dgrad(const fva_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* addend, void* stream);

/* The same data gradient with the FIRST pass of the BatchNorm backward of the layer that produced this convolution's input taken
 * in its epilogue.  dx (incl. the addend) is dz of that producer block z = SiLU(BN(y)) (classfication/models/darknet53.py:28-31,
 * 58-62: what autograd's native_batch_norm_backward + SiLU backward reduce over the batch): per row block and channel the
 * epilogue adds up dU = dz * SiLU'(y * scale + shift) and dU * (y - mean) * rstd from the values it stores, reading y (dense
 * [B*H*W][Cin], the producer's pre-BN output) once.  partial: [fva_bn_partial_rows(fva_conv_dgrad_stat_rows(d))][2][Cin] floats (the rows
 * beyond fva_conv_dgrad_stat_rows(d) are fva_bn_bwd_finalize's scratch), fixed order -- feed it to
 * fva_bn_bwd_finalize (partial_rows = the rows allocated) in place of fva_bn_silu_bwd_reduce's table.  Valid only when dx IS the whole dz (the
 * producer's output has no other consumer than this convolution and, through `addend`, the residual identity). */
typedef struct {
    const void* y;
    const float *scale, *shift, *mean, *rstd;
    float* partial;
    int64_t* acc;        /* when not NULL the sums are ADDED to this accumulator (the producer layer's backward one, see "BatchNorm statistics
                            WITHOUT a finalize launch") and `partial` is not used */
    int32_t acc_replicas;
} fva_bn_bwd_fuse;
int fva_conv_dgrad_bnstats(const fva_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* addend,
                           const fva_bn_bwd_fuse* fuse, void* stream);
int32_t fva_conv_dgrad_stat_rows(const fva_conv_desc* d);

/* dw (fp32, OIHW) (+)= sum_pixels dy x.  Deterministic: split-K partial tiles go to `workspace`
 * (fva_conv_wgrad_workspace() bytes) and are reduced in fixed order. */
int fva_conv_wgrad(const fva_conv_desc* d, const void* x, const void* dy, float* dw_oihw, int accumulate,
                   void* workspace, int64_t workspace_bytes, void* stream);
int64_t fva_conv_wgrad_workspace(const fva_conv_desc* d);
/* The split-K plan of the weight gradients, process-wide: 0 = for a launch that has the chip to itself (default), 1 = for launches
 * that run beside the rest of the backward pass on the side stream (half the block slots: fewer, longer blocks).  The plan fixes
 * the split factor and with it the fp32 summation order of dW: results are bit-identical between runs under the SAME plan, and
 * differ in the last bits between plans.  Returns the previous setting; any other argument only queries.  The workspace size
 * covers both. */
int fva_conv_wgrad_plan(int beside);

/* Stem: conv 3x3 s1 p1 on fp32 NCHW images with Cin <= 3, Cout == 32 (darknet53.py:73 `conv0`).
 * y dense [B*H*W][Cout] dtype + BN partial stats;  wgrad from dy dense [B*H*W][Cout] (pad 0).
 * bf16 with W a multiple of 16 runs on MFMA from a bf16 NHWC4 copy of the images that the call builds in `workspace`
 * (fva_stem_fwd_workspace() bytes; 0 for the other cases, which use the exact fp32 direct kernel). */
int fva_stem_fwd(int dtype, const float* images_nchw, const float* w_oihw, void* y, float* stats_partial,
                 void* workspace, int64_t workspace_bytes, int B, int Cin, int H, int W, int Cout, void* stream);
int64_t fva_stem_fwd_workspace(int dtype, int B, int H, int W);
int32_t fva_stem_stat_blocks(int dtype, int B, int H, int W);
int32_t fva_stem_fused_blocks(int B, int H, int W);   /* partial rows written by fva_stem_fused modes 0 and 2 */
int fva_stem_wgrad(int dtype, const float* images_nchw, const void* dy, float* dw_oihw, int accumulate,
                   void* workspace, int64_t workspace_bytes, int B, int Cin, int H, int W, int Cout, void* stream);
int64_t fva_stem_wgrad_workspace(int B, int Cin, int H, int W, int Cout);
/* bf16 alternative on the MFMA weight-gradient kernel, from the NHWC4 image copy that fva_stem_fwd left in its workspace and
 * a halo dY [B][H+2][W+2][32]: dw_raw [32][16][3] fp32 = [co][kw*4 + ci][kh] (the caller keeps kw < 3, ci < Cin). */
int fva_stem_wgrad_mfma(const void* images_nhwc4, const void* dy_halo, float* dw_raw, void* workspace, int64_t workspace_bytes,
                        int B, int H, int W, void* stream);
int64_t fva_stem_wgrad_mfma_workspace(void);
/* bf16 training form that never stores conv0's pre-BN output: the four BatchNorm / SiLU passes recompute it on MFMA from the
 * NHWC4 image copy (fva_stem_pack; fva_stem_fwd_workspace() bytes).  mode 0: partial sums of y, y^2 -> part; 1: z = SiLU(BN(y))
 * -> halo buffer out [B][H+2][W+2][32]; 2: partial sums of dU, dU*xhat from dz (dense [B*H*W][32]) -> part; 3: dY -> halo
 * buffer out.  part: fva_stem_fused_blocks() rows of [2][32] filled here, ALLOCATED with
 * fva_bn_partial_rows(fva_stem_fused_blocks()) rows, reduced by fva_bn_finalize / fva_bn_bwd_finalize. */
int fva_stem_pack(const float* images_nchw, void* images_nhwc4, int64_t bytes, int B, int Cin, int H, int W, void* stream);
int fva_stem_fused(int mode, const void* images_nhwc4, const float* w_oihw, const void* dz, const float* scale, const float* shift,
                   const float* save_mean, const float* save_rstd, const float* coef, void* out, float* part, int B, int Cin, int H,
                   int W, void* stream);

/* Head: biased 1x1 conv to N = A*(5+C) channels (detection/head/yolov3head.py:50,60;
 * demos/yolov3_u/models/yolov3.py:119-135).  Output fp32, dense [B*H*W][N] (pixel-major, N contiguous):
 * the library's permuted [B,A,H,W,5+C] (yolov3head.py:63) and the demo's NCHW are strided views of it. */
int fva_head_fwd(const fva_conv_desc* d, const void* x, const void* w_fwd, const float* bias, float* out, void* stream);
/* dhead fp32 [B*H*W][N] * (*grad_scale, device scalar, may be NULL = 1) -> dy (dtype) as a halo buffer
 * [B][H+2][W+2][Npad] (border 1, pad columns N..Npad-1 zero) ready for fva_conv_dgrad / fva_conv_wgrad with
 * Cout = Npad, and dbias[N] (+)= column sums (deterministic two-stage reduce whose first stage rides in the pass that writes dy;
 * workspace >= 4*N*4096 bytes). */
int fva_head_bwd_prepare(int dtype, const float* dhead, const float* grad_scale, void* dy, float* dbias,
                         int accumulate, void* workspace, int B, int H, int W, int N, int Npad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (training + eval) fused with SiLU and the residual add.  Replaces nn.BatchNorm2d +
 * nn.SiLU (+ `identity + conv2`) in ConvBlock3x3 / ConvBlock1x1 / ResidualBlock (darknet53.py:11-17,28-31,58-62).
 * ---------------------------------------------------------------------------------------------- */
/* Reduce conv partial stats -> batch mean / biased var; update running stats (momentum, unbiased var) and
 * num_batches_tracked += 1 (both optional: NULL skips); emit save_mean, save_rstd and the fused
 * scale = gamma*rstd, shift = beta - mean*scale.
 * stats_partial must hold fva_bn_partial_rows(nblocks) rows of [2][C] floats: the conv kernels fill the first
 * nblocks rows; tables of >= 1024 rows are first folded in parallel into the extra rows (scratch).  partial_rows = the rows
 * the caller allocated: a table that is too short is refused (FVA_ERR_WORKSPACE) instead of written past. */
int32_t fva_bn_partial_rows(int32_t nblocks);
int fva_bn_finalize(float* stats_partial, int32_t nblocks, int32_t partial_rows, int64_t count, int32_t C,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, float momentum, float eps, float* save_mean, float* save_rstd,
                    float* scale, float* shift, void* stream);
/* Eval mode: scale/shift from running stats. */
int fva_bn_eval_coeffs(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, void* stream);
/* z = silu(y*scale + shift) (+ residual); writes the whole halo buffer z[B][H+2p][W+2p][C] incl. zero
 * border.  residual (optional) has the layout of z with border res_pad. */
int fva_bn_silu_apply(int dtype, const void* y, const float* scale, const float* shift, const void* residual,
                      int res_pad, void* z, int z_pad, int B, int H, int W, int C, void* stream);
/* ---- BatchNorm statistics WITHOUT a finalize launch (round 4) ------------------------------------------------------------------
 * The convolution's tiles ADD their per-channel partial sums into a fixed-point accumulator (two int64 words per sum: integer
 * addition is associative, so the total is the same bits whatever order the tiles arrive in -- deterministic run to run) and the
 * pass that consumes the statistics finalises them in its own prologue: every block computes mean / rstd / scale / shift of all
 * channels once (the arithmetic of fva_bn_finalize), block 0 also writes them out for the backward pass and updates the running
 * statistics (nn.BatchNorm2d in training mode, classfication/models/darknet53.py:11-12).  Replaces the 197 latency-bound finalize /
 * pre-reduce launches of a YOLOv3 step, each a dependent step of the chain.
 *   An accumulator is int64 [replicas][FVA_BN_ACC_WORDS][C] -- per channel the high and low words of the two sums and a count of
 *   non-finite partials (a channel that received one finalises to NaN, as a floating-point sum would) -- and must be ZERO when its first
 *   producer runs.  `replicas` (a power of two <= FVA_BN_ACC_MAX_REPLICAS; the same number for every producer and consumer of an
 *   accumulator): atomics on one address are served one after the other (~10 ns each), so a layer whose convolution runs thousands of
 *   tiles spreads them over several copies (block b adds to copy b mod replicas; about one copy per 65536 output pixels keeps the queue
 *   per address below 5 us) and the consumer adds the copies up.  A layer keeps one accumulator per direction (forward: sums of y and
 *   y^2; backward: sums of dU and dU * xhat), both with the same `replicas`.  A consumer cannot zero the accumulator it reads (its other
 *   blocks may not have read it yet), so each consumer zeroes the OTHER direction's (`zero`, may be NULL): the forward consumer clears
 *   the backward sums of the previous step, the backward consumer clears the forward sums.  fva_bn_acc_finalize, the only reader of
 *   its launch, zeroes both.  An accumulator nobody zeroed (a forward pass without backward) is the caller's to clear.
 * Forward producers: fva_conv_fwd_acc (= fva_conv_fwd with the accumulator in place of the table), fva_conv1x1_fwd_apply_acc;
 * forward consumers: fva_bn_silu_apply_acc, fva_conv1x1_fwd_apply_acc (of the block BEFORE it), fva_bn_acc_finalize.
 * Backward producers: fva_conv_dgrad_bnstats with fva_bn_bwd_fuse::acc set, fva_bn_silu_bwd_reduce_acc; consumer: fva_bn_silu_bwd_apply_acc. */
#define FVA_BN_ACC_WORDS 5
#define FVA_BN_ACC_MAX_REPLICAS 32
typedef struct {
    int64_t* acc;
    int64_t* zero;                          /* may be NULL: the other direction's accumulator, returned to zero by this launch */
    int32_t replicas;
    const float *gamma, *beta;
    float *running_mean, *running_var;      /* may be NULL */
    int64_t* num_batches_tracked;           /* may be NULL */
    float momentum, eps;
    float *save_mean, *save_rstd, *scale, *shift;   /* outputs, [C] each */
} fva_bn_fwd_acc;
typedef struct {
    int64_t* acc;
    int64_t* zero;                          /* may be NULL */
    int32_t replicas;
    const float* gamma;
    float *dgamma, *dbeta;                  /* outputs (+= when accumulate) */
    int32_t accumulate;
} fva_bn_bwd_acc;
int fva_conv_fwd_acc(const fva_conv_desc* d, const void* x, const void* w_fwd, void* y, int64_t* acc, int32_t replicas, void* stream);
int fva_bn_silu_apply_acc(int dtype, const void* y, const fva_bn_fwd_acc* acc, const void* residual, int res_pad, void* z, int z_pad,
                          int B, int H, int W, int C, void* stream);
/* The accumulator finalised by a small launch of its own (writes the four outputs, updates the running statistics, zeroes the
 * accumulator): for a consumer that cannot do it in its prologue -- the thin tile of the fused 1x1 form, foreign code. */
int fva_bn_acc_finalize(const fva_bn_fwd_acc* acc, int64_t M, int C, void* stream);
/* fva_conv1x1_fwd_apply with this layer's statistics added to acc_out, and the block before described by `prev`: prev->acc != NULL: its
 * statistics are finalised in this launch's prologue (Cout = 128: the wide tile; thinner layers run fva_bn_acc_finalize first, inside this
 * call); prev->acc == NULL: prev->scale / prev->shift are final already. */
int fva_conv1x1_fwd_apply_acc(const fva_conv_desc* d, const void* y_prev, const fva_bn_fwd_acc* prev, const void* residual, int32_t res_pad,
                              void* z, const void* w_fwd, void* y, int64_t* acc_out, int32_t replicas_out, void* stream);

/* Backward, pass 1: partial sums over pixels of dU = dz*silu'(u) and dU*xhat (u = y*scale+shift). */
int fva_bn_silu_bwd_reduce(int dtype, const void* dz, const void* y, const float* scale, const float* shift,
                           const float* save_mean, const float* save_rstd, float* partial, int32_t nblocks,
                           int64_t M, int C, void* stream);
int32_t fva_bn_bwd_blocks(int dtype, int64_t M, int C);
/* The accumulator forms of the two backward passes (see "BatchNorm statistics WITHOUT a finalize launch"): pass 1 adds its sums to
 * acc; pass 2 takes dgamma, dbeta and its coefficients from the accumulator in its prologue (no fva_bn_bwd_finalize). */
int fva_bn_silu_bwd_reduce_acc(int dtype, const void* dz, const void* y, const float* scale, const float* shift,
                               const float* save_mean, const float* save_rstd, int64_t* acc, int32_t replicas, int64_t M, int C, void* stream);
int fva_bn_silu_bwd_apply_acc(int dtype, const void* dz, const void* y, const float* scale, const float* shift,
                              const float* save_mean, const float* save_rstd, const fva_bn_bwd_acc* acc, void* dy, int dy_pad,
                              int B, int H, int W, int C, void* stream);
/* Backward, finalize: dgamma, dbeta (+)= and the per-channel coefficients of pass 2.  partial: the table of
 * fva_bn_silu_bwd_reduce or of fva_conv_dgrad_bnstats, ALLOCATED with fva_bn_partial_rows(nblocks) rows: a long table (1024 rows
 * or more) is folded in parallel first, into doubles kept behind the nblocks rows the producer fills; partial_rows = the rows
 * allocated (checked: FVA_ERR_WORKSPACE). */
int fva_bn_bwd_finalize(float* partial, int32_t nblocks, int32_t partial_rows, int64_t M, int C, const float* gamma,
                        const float* save_rstd, float* dgamma, float* dbeta, int accumulate, float* coef,
                        void* stream);
/* Backward, pass 2: dy = gamma*rstd*(dU - dbeta/n - xhat*dgamma/n) written as halo buffer (border dy_pad). */
int fva_bn_silu_bwd_apply(int dtype, const void* dz, const void* y, const float* scale, const float* shift,
                          const float* save_mean, const float* save_rstd, const float* coef, void* dy, int dy_pad,
                          int B, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FPN glue and layout conversion.  Replaces nn.Upsample(scale_factor=2,'nearest') + torch.cat
 * (yolov3neck.py:71,105,110; demos/yolov3_u/models/yolov3.py:96,100).
 * ---------------------------------------------------------------------------------------------- */
/* out[B][2h+2][2w+2][Cup+Cskip] (pad 1).  up: [B][h+2pu][w+2pu][Cup]; skip: [B][2h+2ps][2w+2ps][Cskip].
 * up_first != 0 -> channel order [up | skip] (library), else [skip | up] (demo). */
int fva_upsample2_concat_fwd(int dtype, const void* up, int up_pad, const void* skip, int skip_pad, void* out,
                             int B, int h, int w, int Cup, int Cskip, int up_first, void* stream);
/* dcat dense [B][2h][2w][Cup+Cskip] -> dup dense [B][h][w][Cup] (2x2 sums) and dskip dense [B][2h][2w][Cskip]. */
int fva_upsample2_concat_bwd(int dtype, const void* dcat, void* dup, void* dskip, int B, int h, int w, int Cup,
                             int Cskip, int up_first, void* stream);
/* Arbitrary-stride fp32/bf16 [B,C,H,W] tensor -> halo NHWC buffer of `dtype` (foreign inputs). */
int fva_pack_nchw(int dtype, const void* src, int src_is_bf16, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                  void* dst, int dst_pad, int B, int C, int H, int W, void* stream);
/* Cast / re-layout dense or halo NHWC -> dense NHWC of another dtype (gradient hand-off at the API edge). */
int fva_cast_nhwc(const void* src, int src_dtype, int src_pad, void* dst, int dst_dtype, int dst_pad,
                  int B, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Target assignment + loss, library surface.  Replaces Yolov3Loss.build_target / forward
 * (loss/yolov3_loss.py:29-124) with BiCrossEntropyLoss (loss/classification_loss.py:42-65),
 * CIOULoss (loss/iou_loss.py:88-107) and cal_iou/CIOU/DIOU/xyxy_iou (detection/tools/IOU.py).
 * Head tensors are addressed through explicit element strides so any view works.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    float* data;          /* fp32 head tensor of this level */
    float* grad;          /* fp32 gradient buffer with the SAME strides (may be NULL: loss only) */
    int64_t sb, sa, sy, sx, sk; /* element strides of (batch, anchor, grid_y, grid_x, channel) */
    int32_t B, A, H, W, K;      /* K = 5 + num_classes */
    float anchor_w[8], anchor_h[8]; /* PIXEL-unit anchors of this level (A <= 8) */
    float stride;               /* backbone stride of this level */
} fva_head_level;

/* Matcher (build_target, yolov3_loss.py:75-124).  targets [T][6] fp32.  Outputs per level, capacity T*A rows
 * in the reference's row order (target-major, anchor-minor): count[1] i32, b/gx/gy/a/cls i64 [cap],
 * xywh [cap][4] f32, anc [cap][2] f32.  Integer outputs are bit-exact with the reference. */
typedef struct {
    int32_t* count;
    int64_t *b, *gx, *gy, *a, *cls;
    float *xywh, *anc;
} fva_match_out;
int fva_yolov3_match(const float* targets, int32_t T, const fva_head_level* level, const fva_match_out* out, void* stream);

/* Loss forward + analytic backward for all levels in one call (yolov3_loss.py:29-72).
 * loss_out[4] = {total, box, conf, cls} (total already * ratios * B); grads (if level.grad != NULL) are
 * d total / d head ACCUMULATED into level.grad, which the caller zero-fills first (the objectness channel
 * of every cell is stored, matched rows are added).  workspace: fva_yolov3_loss_workspace(). */
int fva_yolov3_loss(const float* targets, int32_t T, const fva_head_level* levels, int32_t nlevels,
                    float ratio_box, float ratio_conf, float ratio_cls, float* loss_out,
                    void* workspace, int64_t workspace_bytes, void* stream);
int64_t fva_yolov3_loss_workspace(int32_t T, const fva_head_level* levels, int32_t nlevels);

/* Data-parallel form of the same loss.  The reference wraps the model in nn.DataParallel (demos/yolov3_u/train.py:85) and
 * evaluates yolov3_loss.py:29-72 ONCE on the batch gathered from all N replicas: its per-match means divide by the match count
 * of the whole job and its result is multiplied by the job's batch size.  With one process per GPU each rank calls this entry on
 * its own images and targets with norm_counts[nlevels] = the per-level match counts summed over all ranks (device memory: run
 * fva_yolov3_match per level, all-reduce the three counts) and norm_batch = N * B: loss_out[0] is then this rank's SHARE of that
 * loss and level.grad its gradient, so that a SUM all-reduce of the parameter gradients (and of loss_out[0]) reproduces the
 * reference's step.  norm_counts == NULL: plain fva_yolov3_loss. */
int fva_yolov3_loss_dp(const float* targets, int32_t T, const fva_head_level* levels, int32_t nlevels,
                       float ratio_box, float ratio_conf, float ratio_cls, const int32_t* norm_counts, int32_t norm_batch,
                       float* loss_out, void* workspace, int64_t workspace_bytes, void* stream);

/* x[n] (fp32, 16-byte aligned) *= *scale (device scalar), in place; an empty launch when the scalar is exactly 1.  The chain rule of a
 * loss whose gradient buffers were filled in its forward pass: autograd hands `loss.backward()` an upstream gradient of one, and the
 * 3 x 91 MB multiply that `grad * gout` would be is what the backward pass starts with (loss/yolov3_loss.py:29-72 via autograd). */
int fva_scale_by_device_scalar(float* x, int64_t n, const float* scale, void* stream);

/* Stand-alone BiCrossEntropyLoss (loss/classification_loss.py:36-65) forward + dl/dy.  y [numel] fp32 logits (or probabilities:
 * already_sigmoid) seen as rows of C classes; target = one-hot of label[numel / C] (int64) or, label == NULL, the dense float
 * target [numel]; weights: NULL, one value, or numel values.  loss_out[1] = sum of the weighted element losses (/ numel when
 * mean); grad (optional, [numel]) = d(sum of weighted element losses)/dy, NOT divided by numel.  workspace: 1024 floats.
 * Deterministic (fixed-order partial sums). */
int fva_bce_loss(const float* y, const int64_t* label, const float* dense_target, const float* weights, int64_t weights_numel,
                 int64_t numel, int32_t C, int32_t already_sigmoid, int32_t mean, float* loss_out, float* grad, float* workspace,
                 void* stream);

/* Losses of the two-stage head (demos/faster_rcnn/models/rpn.py:8-64,303-312; fast.py:173-201), value + gradient in one launch.
 * fva_row_loss: logits [R][C] fp32 with int64 labels[R]; mode 0 = F.cross_entropy(reduction='mean'), mode 1 = the RPN's FocalLoss
 * (-(1 - p_t)^gamma * log p_t over softmax probabilities, alpha 1, mean); grad (optional) [R][C] = d loss / d logits;
 * workspace: R floats.  fva_smooth_l1: F.smooth_l1_loss(pred, target, reduction='mean') (beta 1) over n elements; grad (optional)
 * [n] = d loss / d pred; workspace: 1024 floats.  Deterministic (fixed-order sums). */
int fva_row_loss(const float* logits, const int64_t* labels, int32_t R, int32_t C, int32_t mode, float gamma, float* loss_out, float* grad,
                 float* workspace, void* stream);
int fva_smooth_l1(const float* pred, const float* target, int64_t n, float* loss_out, float* grad, float* workspace, void* stream);

/* Demo loss (demos/yolov3_u/utils/lossv3.py:18-119): best-anchor assignment, BCE/MSE/BCE terms, IoU>0.5
 * ignore mask, masked objectness BCE.  level.anchor_* are FEATURE-scale here and level.stride is unused.
 * loss_out[5] = {total, xy, wh, cls, conf} (unweighted parts, as the reference prints them). */
int fva_demo_loss(const float* targets, int32_t T, const fva_head_level* levels, int32_t nlevels, float* loss_out,
                  void* workspace, int64_t workspace_bytes, void* stream);
int64_t fva_demo_loss_workspace(int32_t T, const fva_head_level* levels, int32_t nlevels);

/* IoU family on device (detection/tools/IOU.py, BOX.py).  kind: 0 IoU 1 GIoU 2 DIoU 3 CIoU;
 * mode: 0 xyxy 1 xywh 2 wh; variant: 0 library 1 demo (iou.py centre sums / minus sign).
 * pairwise: out[N] and (optional) grad_a[N][4 or 2] = d out / d a in the caller's box parametrisation (ties
 * split as torch.maximum/minimum do, CIoU alpha constant); batch: out[N][M]. */
int fva_iou_pairwise(int kind, int mode, int variant, const float* a, const float* b, float* out, float* grad_a, int64_t N,
                     float eps, void* stream);
int fva_iou_batch(int kind, int mode, int variant, const float* a, const float* b, float* out, int64_t N, int64_t M, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Validation side (scope row f-2): eval decode, candidate selection, non-maximum suppression.
 * ---------------------------------------------------------------------------------------------- */
/* Un-letterbox mapping of the demo's postProcess (demos/yolov3_u/inference.py:90-106). */
typedef struct {
    float resize_ratio, pad_left, pad_top; /* x_ori = (x - pad_left) / resize_ratio */
    float ori_w, ori_h;                    /* clamp bounds of the original image */
    float min_wh;                          /* boxes with w <= min_wh or h <= min_wh are dropped (5 in the reference) */
} fva_letterbox;
/* Decode the head tensors of all levels into out [B][rows_per_image][K] fp32 (level after level).
 * variant 0 -- Yolov3.forward eval branch (detection/models/yolov3.py:35-53): rows of a level ordered (a, y, x);
 *   xy = (sigmoid(t) + cell) * stride, wh = exp(t) * anchor (PIXEL anchors), everything else sigmoid.
 * variant 1 -- postProcess (demos/yolov3_u/inference.py:58-88): rows ordered (y, x, a);
 *   xy = (2 sigmoid(t) - 0.5 + cell) * stride, wh = (2 sigmoid(t))^2 * anchor * stride (FEATURE anchors).
 *   With lb != NULL rows additionally go through :90-106: un-letterbox, clamp, xywh -> clamped xyxy in columns 0..3;
 *   rows the reference drops (w or h <= min_wh) keep their place with objectness -1. */
int fva_yolo_decode(const fva_head_level* levels, int32_t nlevels, int32_t variant, const fva_letterbox* lb,
                    float* out, int64_t rows_per_image, void* stream);

typedef struct {
    int32_t box_mode;    /* 0: rows carry xywh (converted with x -/+ w/2), 1: rows carry xyxy */
    int32_t score_mode;  /* 0: score = max_c(cls_c * obj) (NMS.py:13-16, nms.py:76-82); 1: score = obj (nms.py:24-36) */
    int32_t rethreshold; /* 1: also drop candidates with max_c(cls_c * obj) <= conf_thres (nms.py:84) */
    int32_t max_det;     /* detections kept per image */
    int32_t max_nms;     /* > 0: only the max_nms best candidates enter NMS (nms.py:40-42: 30000) */
    float conf_thres, iou_thres;
    float class_gap;     /* 0: class-agnostic (NMS.py); 4096: boxes shifted by category * gap (nms.py:45-46) */
} fva_nms_params;
/* Stage 1: rows with objectness > conf_thres become candidates, in row order (what boolean-mask indexing gives).
 * pred [B][R][K] fp32; counts [B] i32; cand: opaque buffer of fva_nms_candidates_workspace(B, R) bytes. */
int64_t fva_nms_candidates_workspace(int32_t B, int32_t R);
int fva_nms_candidates(const float* pred, int32_t B, int32_t R, int32_t K, const fva_nms_params* p, void* cand,
                       int64_t cand_bytes, int32_t* counts, void* stream);
/* Stage 2: torchvision.ops.nms on every image's candidates (greedy, highest score first, suppress when
 * inter / (area_a + area_b - inter) > iou_thres; score ties broken by candidate order).  nmax >= max(counts)
 * (the caller reads counts back: the reference's boolean indexing synchronises at the same point).
 * out [B][max_det][6] = x1, y1, x2, y2, score, category; out_rows [B][max_det] = source row in pred;
 * keep_counts [B]. */
int64_t fva_nms_select_workspace(int32_t B, int32_t nmax);
int fva_nms_select(const void* cand, const int32_t* counts, int32_t B, int32_t R, int32_t nmax, const fva_nms_params* p,
                   void* workspace, int64_t workspace_bytes, float* out, int32_t* out_rows, int32_t* keep_counts,
                   void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input side (scope row f-3): decoded uint8 RGB (HWC) images -> the [B][3][H][W] fp32 input batch in one launch.
 * A "paste job" resizes one source image with OpenCV's 8-bit INTER_LINEAR arithmetic (an exact 2x decimation is
 * area-averaged, as cv2.resize does) to dst_w x dst_h, optionally mirrors it, and places it at (left, top) of canvas
 * `b`; pixels no job covers take `fill`.  Every byte then goes through lut[channel][value] (the caller tabulates
 * x/255 or (x/255 - mean)/std exactly as the reference computes it).  Replaces cv2.resize + Padding + np.fliplr/flipud
 * + Normalization + transpose + stack (datasets/detection_dataloader.py:44-103, datasets/common/padding.py,
 * datasets/common/augmentation.py:298-376) and ResizeByMax/Padding/flips/Mosaic01 (demos/yolov3_u/data_gen.py).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t src_offset;      /* byte offset of this image's [src_h][src_w][3] pixels in `src` */
    int32_t src_h, src_w;
    int32_t dst_h, dst_w;    /* size after the resize */
    int32_t top, left;       /* where the resized image lands on its canvas */
    int32_t flip_h, flip_v;  /* mirror the resized image (cv2.flip 1 / 0) */
    int32_t src_pitch;       /* bytes between source rows; 0 = src_w * 3 (a sub-image of a wider canvas otherwise) */
    int32_t reserved;
    double scale_x, scale_y; /* 1.0 / ((double)dst / src), as cv2.resize derives them */
} fva_paste_job;
/* jobs: DEVICE array; job_start: DEVICE int32[B+1], canvas b owns jobs job_start[b] .. job_start[b+1]-1, pasted in
 * that order (later jobs overwrite earlier ones); lut: DEVICE float[3][256]; out: [B][3][H][W] fp32. */
int fva_paste_resize_normalize(const uint8_t* src, const fva_paste_job* jobs, const int32_t* job_start, int32_t B,
                               int32_t H, int32_t W, int32_t fill, const float* lut, float* out, void* stream);
/* The same placement without the value table: out [B][H][W][3] uint8 (an intermediate image that a later call resizes
 * again -- the demo resizes every image to input_size before Mosaic01 resizes it to input_size/2). */
int fva_paste_resize_u8(const uint8_t* src, const fva_paste_job* jobs, const int32_t* job_start, int32_t B, int32_t H,
                        int32_t W, int32_t fill, uint8_t* out, void* stream);

/* Colour / blur extras of the demo's training image path (demos/yolov3_u/data_gen.py:26-33,120-146), on uint8 canvases
 * [B][H][W][3] resident in device memory (the output of fva_paste_resize_u8).  One job per canvas image. */
typedef struct {
    int32_t h, w;        /* valid region of the canvas image, anchored at its top-left corner */
    int32_t clahe;       /* 1: HistEqualize -- CLAHE (clip limit 2.0, 8 x 8 tiles) on the luma of OpenCV's 8-bit YUV */
    int32_t hsv;         /* 1: HueSaturationValue -- the image's three 256-byte tables applied in OpenCV's 8-bit HSV (H < 180) */
    int32_t blur;        /* fva_colour_blur_shuffle_normalize: 0 none, 1 cv2.blur 3x3, 2 cv2.medianBlur 3, 3 cv2.GaussianBlur 3x3 sigma 0 */
    int32_t perm[3];     /* fva_colour_blur_shuffle_normalize: output channel c is input channel perm[c] (ChannelShuffle) */
} fva_colour_job;
/* In place on the valid region of every canvas: [CLAHE] then [HSV tables].  jobs: DEVICE array [B]; hsv_luts: DEVICE bytes
 * [B][3][256] (hue, saturation, value tables; read only for jobs with hsv = 1); max_h / max_w: largest valid region;
 * workspace: fva_colour_workspace(B) bytes (the 64 tile tables of every image). */
int64_t fva_colour_workspace(int32_t B);
int fva_colour_clahe_hsv(uint8_t* canvases, int32_t B, int32_t H, int32_t W, const fva_colour_job* jobs, int32_t max_h, int32_t max_w,
                         const uint8_t* hsv_luts, uint8_t* workspace, void* stream);
/* canvases [B][H][W][3] uint8 -> out [B][3][H][W] fp32: per image a 3x3 blur (or none) over the whole canvas, the channel
 * permutation, then lut[channel][value] (x / 255 as the reference computes it: ToTensorV2 + `/ 255.`). */
int fva_colour_blur_shuffle_normalize(const uint8_t* canvases, int32_t B, int32_t H, int32_t W, const fva_colour_job* jobs,
                                      const float* lut, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer: torch.optim.Adam semantics (demos/yolov3_u/train.py:68), multi-tensor.
 * ptrs: device array [4][n] of {param, grad, exp_avg, exp_avg_sq} fp32 pointers; sizes: device int64[n].
 * step is the 1-based step count; lr/betas/eps/weight_decay as torch (L2-in-gradient decay).
 * ---------------------------------------------------------------------------------------------- */
int fva_adam_step(const void* const* ptrs, const int64_t* sizes, int32_t n, int64_t max_size, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream);
/* The same update with the step count and the learning rate in device memory, for a training step captured in a HIP graph
 * (utils/fit.py:52-66 replayed as one launch): state_dev[3] doubles = {step count, lr / (1 - beta1^step), sqrt(1 - beta2^step)};
 * the call first advances state_dev[0] by one and recomputes the two bias-corrected factors from *lr_dev (a device float the
 * host may rewrite between replays: LR schedules), then updates the tensors.  Same arithmetic as fva_adam_step. */
int fva_adam_step_dev(const void* const* ptrs, const int64_t* sizes, int32_t n, int64_t max_size, const float* lr_dev, float beta1,
                      float beta2, float eps, float weight_decay, double* state_dev, float grad_scale, void* stream);

/* Gradient bucket fill for the data-parallel all-reduce (parallel.GradientReducer; the reference's nn.DataParallel gathers
 * gradients tensor by tensor, demos/yolov3_u/train.py:85): dst[offs[t] + i] = src_t[i], converted to dst_dtype (FVA_F32 or
 * FVA_BF16), for n fp32 source tensors in one launch.  table_dev: 3n int64 in device memory = source pointers | element counts |
 * element offsets into dst; a null pointer or zero count skips the tensor.  max_size sizes the grid. */
int fva_gather_cast(const int64_t* table_dev, int32_t n, int64_t max_size, void* dst, int dst_dtype, void* stream);

/* ---- VGG blocks of the two-stage head's backbone (demos/faster_rcnn/models/vgg.py: Conv2d + bias -> ReLU,
 * MaxPool2d(2, 2); no BatchNorm) ---------------------------------------------------------------------------------------------
 * fva_conv_fwd_bias_act: z = act(conv(x) + bias[n]) straight into the halo buffer z (border zeroed), act 1 = ReLU, 2 = none;
 *   its backward uses fva_conv_dgrad / fva_conv_wgrad on dY from
 * fva_bias_relu_bwd: dY = dZ * (Z > 0) (dZ dense NHWC, Z the forward's halo output) into the halo buffer dY, and partial
 *   [fva_bias_relu_bwd_rows()][C] column sums of dY that fva_colsum adds up in fixed order (= dbias; scratch may be NULL).
 * fva_maxpool2_fwd: 2x2 / stride 2 max of the halo tensor x [B][H+2p][W+2p][C] into the halo tensor out [B][H/2+2q][W/2+2q][C]
 *   (floor mode; border zeroed).  fva_maxpool2_bwd: dz dense [B][H/2][W/2][C] -> dx dense [B][H][W][C]; the gradient goes to the
 *   first maximum of each window in scan order. */
int fva_conv_fwd_bias_act(const fva_conv_desc* d, const void* x, const void* w_fwd, const float* bias, int32_t act, void* z, int32_t z_pad,
                          void* stream);
int32_t fva_bias_relu_bwd_rows(int B, int H, int dy_pad);
int fva_bias_relu_bwd(int dtype, const void* dz, const void* z, int z_pad, void* dy, int dy_pad, float* partial, int B, int H, int W, int C,
                      void* stream);
int32_t fva_colsum_scratch_rows(int32_t rows);   /* rows of [C] floats fva_colsum wants as scratch for its two-pass form (0: none) */
int fva_colsum(const float* partial, int32_t rows, int C, float* out, float* scratch, void* stream);
/* Fully connected layers of the Fast head (Linear -> ReLU, demos/faster_rcnn/models/vgg.py classifier): dY = dZ * (Z > 0)
 * (relu = 0: dY = dZ) over rows [R][C] (dY rows dy_stride elements apart: the interior column of a halo buffer [1][R+2][3][C], which
 * is what fva_conv_wgrad / fva_conv_dgrad read) and per block of 32 rows the column sums of dY: partial [fva_rows_relu_bwd_rows(R)][C],
 * added up by fva_colsum (= dbias).  The GEMMs themselves are 1x1 convolutions over R "pixels" (fva_conv_fwd_bias_act, fva_conv_dgrad,
 * fva_conv_wgrad). */
int32_t fva_rows_relu_bwd_rows(int32_t R);
int fva_rows_relu_bwd(int dtype, const void* dz, const void* z, void* dy, int64_t dy_stride, float* partial, int32_t R, int32_t C, int32_t relu,
                      void* stream);
int fva_maxpool2_fwd(int dtype, const void* x, int x_pad, void* out, int out_pad, int B, int H, int W, int C, void* stream);
int fva_maxpool2_bwd(int dtype, const void* dz, const void* x, int x_pad, void* dx, int B, int H, int W, int C, void* stream);

/* ---- RPN proposal rows (two-stage head) ------------------------------------------------------------------------
 * filter_proposals of demos/faster_rcnn/models/rpn.py:162-186 up to the per-image selection: cls [B][H][W][A][2] logits,
 * deltas [B][H][W][A][4], anchors_wh [A][2] (already divided by the backbone stride; anchor centres are the cell indices) ->
 * out [B][H*W*A][6] = clamped x1, y1, x2, y2, softmax foreground score, 1.0: the rows fva_nms_candidates / fva_nms_select
 * take (box_mode 1, score_mode 1, max_nms = rpn_pre_nms_top_n, max_det = rpn_post_nms_top_n). */
int fva_rpn_decode(const float* cls, const float* deltas, const float* anchors_wh, float* out, int32_t B, int32_t H, int32_t W,
                   int32_t A, void* stream);

/* RPN anchor / ground-truth matcher: the labelling inside RPN.computet_loss (demos/faster_rcnn/models/rpn.py:209-277).
 * anchors_xywh [Na][4] in feature cells (make_anchors_xywh, flattened (y, x, a)); targets [T][6] = image index, class, normalised
 * xywh (scaled by the map size here, rpn.py:257).  labels [B][Na] i32: >= 0 index of the matched box WITHIN its image's boxes,
 * -1 negative (best IoU < neg_thr), -2 ignored; every box then claims its best anchor, later boxes overriding earlier ones.
 * An image without boxes gets -2 everywhere.  workspace: T int32 (best anchor per box row). */
int fva_rpn_match(const float* anchors_xywh, int32_t Na, const float* targets, int32_t T, int32_t B, int32_t feature_h,
                  int32_t feature_w, float pos_thr, float neg_thr, int32_t* labels, int32_t* workspace, void* stream);

/* Fast head: proposal / ground-truth labelling of select_positive_negative_samples (demos/faster_rcnn/models/fast.py:100-127)
 * for ONE image: proposals_xywh [N][4]; targets [T][6] as above but with xywh already in feature cells (fast.py:217), only the
 * rows of `image` count.  labels [N] i32: >= 0 matched box (best IoU >= pos_thr), -1 negative (neg_floor <= best IoU <
 * neg_thr; the reference's floor is 0.1), -2 ignored. */
int fva_fast_match(const float* proposals_xywh, int32_t N, const float* targets, int32_t T, int32_t image, float pos_thr,
                   float neg_thr, float neg_floor, int32_t* labels, void* stream);

/* ---- RoIAlign (two-stage head) -----------------------------------------------------------------------------
 * torchvision.ops.roi_align as the reference calls it (demos/faster_rcnn/models/fast.py:227-231,258): rois [K][5] = (batch
 * index, x1, y1, x2, y2), out [K][C][PH][PW] fp32 (the order torch.flatten(.., 1) feeds the classifier), aligned = False,
 * sampling_ratio <= 0 = adaptive grid.  feat: halo NHWC [B][H+2*feat_pad][W+2*feat_pad][C] of dtype.  Backward: grad_out in
 * the same [K][C][PH][PW] layout is scattered (float atomics) into dfeat, dense fp32 NHWC [B][H][W][C], which the caller
 * zeroes first.  K = 0 is a no-op. */
int fva_roi_align_fwd(int dtype, const void* feat, int feat_pad, const float* rois, int K, float* out, int B, int H, int W, int C,
                      int PH, int PW, float spatial_scale, int sampling_ratio, void* stream);
int fva_roi_align_bwd(const float* grad_out, const float* rois, int K, float* dfeat, int B, int H, int W, int C, int PH, int PW,
                      float spatial_scale, int sampling_ratio, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FASTVISION_AMD_H */
