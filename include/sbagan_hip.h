/*
 * sbagan_hip.h -- C ABI of libsbagan_hip.so: the MI355X (gfx950) kernels behind
 * the SBA-GAN adversarial training hot path.
 *
 * Conventions
 *  - every entry point is extern "C", takes raw DEVICE pointers, sizes and a
 *    hipStream_t (passed as void*), returns int: 0 = ok, <0 = SBA_E_* below.
 *    Nothing is allocated, freed or synchronised inside; no global state; the
 *    library is thread-safe per stream and graph-capturable.
 *  - activations are NHWC ("channels last").  `dtype` selects the storage and
 *    MFMA input type of activations / packed weights: SBA_F32 (exact f32 MFMA,
 *    v_mfma_f32_32x32x2_f32) or SBA_BF16 (v_mfma_f32_32x32x16_bf16, f32
 *    accumulate).  Statistics, losses, master weights, gradients of weights and
 *    optimizer state are always f32.
 *  - "reference" citations are relative to zhengfei0908/SBA-GAN AttnGAN2/code.
 *    The reference has no FFI of its own: each function below
 *    replaces the torch.nn call sequence cited next to it.
 */
#ifndef SBAGAN_HIP_H
#define SBAGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBA_F32 0
#define SBA_BF16 1
/* bf16 activations / packed weights as SBA_BF16, but the RAW conv output y -- the pre-BatchNorm tensor, which is
 * only ever read by the BatchNorm kernels, never an MFMA operand -- is stored as IEEE binary16 (values saturate
 * at +-65504): three more mantissa bits at the same bytes, which removes one of the two bf16 roundings per
 * conv + BatchNorm + activation layer.  Accepted by sba_conv_igemm (no addend / bias / ReLU / mask epilogue) and by
 * the sba_bn_* entry points (where it describes y; out, dout, residual and dy stay bf16). */
#define SBA_BF16_YH 2

#define SBA_OK 0
#define SBA_E_ARG (-1)      /* shape / alignment / enum argument the kernels do not support */
#define SBA_E_LAUNCH (-2)   /* hipGetLastError() != hipSuccess after the launch */
#define SBA_E_UNSUPPORTED (-3)   /* sba_replay_create: the captured graph holds a node kind that cannot be re-issued */

#define SBA_ACT_NONE 0
#define SBA_ACT_GLU 1       /* model.py:15-23 */
#define SBA_ACT_LRELU 2     /* LeakyReLU(0.2), model.py:544 */
#define SBA_ACT_RELU 3      /* sba_bn_act_fwd only: BasicConv2d of the Inception trunk in TRAINING mode (frozen weights,
                             * batch statistics: pretrain_DAMSM.py:51 cnn_model.train(), model.py:170-199); no backward */

#define SBA_MAX_TAPS 32

/* Geometry of one implicit-GEMM convolution launch.  Output pixel (oy, ox) of
 * the OHs x OWs sub-grid reads input pixel (oy*sy + ty[t], ox*sx + tx[t]) for
 * tap t (zero outside the logical input), and is stored at output pixel
 * (oy*osy + ooy, ox*osx + oox) of the OH x OW output.  ups != 0: the logical
 * input is the nearest-neighbour x2 upsampling of the physical IH x IW tensor
 * (reference nn.Upsample(scale_factor=2), model.py:41), read as (iy>>1, ix>>1).
 * Packed weights for the launch are [Cout][ntaps][Cin]. */
typedef struct sba_conv_geom {
    int32_t N, IH, IW, Cin;
    int32_t OH, OW, Cout;
    int32_t OHs, OWs;
    int32_t sy, sx;
    int32_t osy, osx, ooy, oox;
    int32_t ups;
    int32_t ntaps;
    int8_t ty[SBA_MAX_TAPS];
    int8_t tx[SBA_MAX_TAPS];
    /* channel strides / offsets (elements) of the input and output tensors, for reading from /
     * writing into a channel slice of a wider NHWC tensor (Inception concats); 0 = dense
     * (x_cstride = Cin, y_cstride = Cout).  Honoured by sba_conv_igemm only. */
    int32_t x_cstride, x_coff, y_cstride, y_coff;
    int32_t relu;               /* epilogue: y = max(acc + bias, 0) when set (with `bias`) */
    /* tile configuration of sba_conv_igemm: 0 = the library's rule table; 1..SBA_IGEMM_TILES = a specific one
     * (measured per layer shape by tools/tune_igemm.py into sbagan/igemm_table.json; bf16 only, an id the
     * geometry cannot use falls back to the rules).  ksplit: 0 = the library decides, n >= 1 = split K n ways. */
    int32_t tile, ksplit;
    /* sba_conv_wgrad only: non-zero = the caller knows dw is all zeros (just cleared, nothing accumulated yet):
     * kernel paths whose workgroups own their outputs exclusively then STORE instead of read-modify-write (the
     * GEMM-like layers' weight gradients are bound by that traffic: 268 MB for D_NET256's widest layer).  Paths
     * that add with atomics ignore it.  0 = always accumulate. */
    int32_t first_write;
    /* sba_conv_igemm: layout of the packed weights `w`.  0 = row-major [Cout][ntaps][Cin].  1 = FRAGMENT-MAJOR
     * (sba_pack_frag_multi): accepted only where sba_conv_igemm_plan (asked with w_layout = 1) reports family 0 or 4 -- the
     * halo-tile 3x3 kernels, bf16, stride 1, the nine taps = the nine cells of a 3 x 3 window, Cin % 32 == 0,
     * Cout % 32 == 0, no statistics for family 4 -- which keep the weight fragments in registers instead of LDS. */
    int32_t w_layout;
} sba_conv_geom;
#define SBA_IGEMM_TILES 18

const char* sba_version(void);

/* ---- deterministic-reduction mode ----------------------------------------------------------------------------
 * The one piece of process-wide state in the library.  Off (default): partial sums of BatchNorm / InstanceNorm
 * statistics, split-K tiles, pixel-split weight gradients, the attention / DAMSM gradients etc. meet in f32 atomics
 * (LDS and global), so results depend on workgroup arrival order in the last bits (like cuDNN's default algorithms
 * under the reference, torch.backends.cudnn.deterministic = False).  On: every such kernel writes its partial sums
 * to a private slot of the caller's `scratch` ring (plain stores) and an ordered fold adds the slots in slot order,
 * in-workgroup accumulation is done wave by wave, split-K is disabled and the conv epilogue's BatchNorm statistics
 * are replaced by the ordered sba_bn_stats pass over the stored tensor: two runs of the same launches on the same
 * inputs are BIT-IDENTICAL, whatever the stream / graph / replayer issues them.  Slower (about 1.3x on the step);
 * meant for parity tests and debugging, not for the benchmark.
 *   scratch: device memory, 256-byte aligned, >= 1 MiB; size it for the partial sums of one whole step (the
 *   B = 20 step uses about 0.6 GiB; sba_det_high_water() reports the largest amount handed out between two resets).
 *   The ring never wraps: an allocation that does not fit in what is left since the last sba_det_reset() fails
 *   (SBA_E_ARG from the launching entry point, one line on stderr) -- slots handed out earlier may still be waiting for
 *   their fold.  sba_det_reset() rewinds the ring (call it at the start of a
 *   step, before capture: the addresses are baked into captured graphs).  Switch the mode only while the device is
 *   idle. */
int sba_set_deterministic(int on, void* scratch, int64_t scratch_bytes);
int sba_get_deterministic(void);
int sba_det_reset(void);
int64_t sba_det_high_water(void);
/* Scratch ring for two-stage reductions in the DEFAULT mode (optional; NULL / 0 removes it).  Launches whose workgroups
 * would all add into the same few hundred addresses at their end (sba_d_stem_bwd's and sba_img_head_bwd's weight
 * gradients) then store per-workgroup partial sums into the ring and a second small launch folds them into the
 * destination; without a ring they use f32 atomics (correct, ~30 us slower per launch at 256 px).  Device memory,
 * 256-byte aligned, >= 1 MiB; a launch takes at most a quarter of it (larger requests fall back to atomics); regions
 * are handed out round-robin at host-issue time, so size it for the launches of one whole step (64 MiB covers the B = 20
 * step, which uses ~33 MB).  REUSE-DISTANCE ASSUMPTION: a region is handed out again after `scratch_bytes` of later
 * requests; the caller guarantees that the fold of its previous user has completed by then -- true when every step
 * joins its streams before the next one starts (GANStep.step does; a captured step re-uses the SAME regions every replay,
 * ordered by that join).  ONE ring per process = one device per process (the launch model of this library: one process
 * per GPU); the host refuses a second device (sbagan.ops.reduce_scratch). */
int sba_set_reduce_scratch(void* scratch, int64_t scratch_bytes);
/* SBA_BN_STAT_SLOTS the library was compiled with (the host sizes its statistics buffers with it) */
int sba_bn_stat_slots(void);

/* ---- convolutions (replace nn.Conv2d fwd / autograd bwd; model.py:32-35,552,563-574) ---- */
/* y[pixel][co] = sum_t sum_ci x[gather(pixel,t)][ci] * w[co][t][ci]  (+ addend[pixel][co]).
 * stats != NULL: also accumulates per-channel sum(y), sum(y^2) of the f32 accumulators into
 * stats[0..Cout) and stats[Cout..2Cout) (caller zeroes it) -- the BatchNorm batch statistics. */
/* workspace (may be NULL): scratch for split-K of small-M / long-K layers, used when it holds
 * >= 4*M*Cout bytes.  It must be ZERO-FILLED when first handed in; every call leaves it zero-filled
 * again, so stream-ordered reuse of one buffer needs no further memsets.  With SBA_SPLITK_FUSED=1 in the environment
 * (an experiment, off by default) and 64 KiB more room than the partial sums need, its LAST 64 KiB are arrival
 * tickets (one int32 per output tile): the last K split to arrive finishes the tile inside the GEMM kernel instead of
 * a separate finishing launch. */
int sba_conv_igemm(int dtype, const void* x, const void* w, void* y, const void* addend,
                   float* stats, const sba_conv_geom* g, void* workspace, int64_t workspace_bytes,
                   void* stream);
/* same with a per-output-channel f32 bias (may be NULL) added before the optional ReLU of g->relu
 * (frozen conv + folded BatchNorm(eval) + ReLU of the image encoder, model.py:170-199), and an optional
 * relu_mask tensor laid out like y: after the addend, outputs are zeroed where relu_mask <= 0 -- used by
 * the data-gradient that completes d(loss)/d(t) to apply the backward of the ReLU that produced t. */
int sba_conv_igemm_bias(int dtype, const void* x, const void* w, void* y, const void* addend,
                        float* stats, const float* bias, const void* relu_mask, const sba_conv_geom* g,
                        void* workspace, int64_t workspace_bytes, void* stream);
/* Which kernel sba_conv_igemm launches for a geometry (nothing is launched; measurement / reporting aid):
 * plan[0] = family (0 halo-tile 3x3 conv3x3_halo_kernel, 1 igemm_dma2_kernel, 2 igemm_dma_kernel, 3 igemm_kernel, 4 the general
 * register-weight halo kernel conv3x3_halo3g_kernel -- only with g->w_layout = 1),
 * plan[1] = tile id (families 1 / 2: the ids of sba_conv_geom.tile) or configuration, plan[2] = K splits. */
int sba_conv_igemm_plan(int dtype, const sba_conv_geom* g, int64_t workspace_bytes, int* plan);
/* GROUPED launch: n <= SBA_GROUP_MAX independent convolutions -- no output of one is an input of another, their outputs
 * do not overlap -- as ONE grid (bf16 only, no split-K, no statistics; bias / ReLU (g->relu) / addend /
 * relu_mask per item as in sba_conv_igemm_bias).  For the branches of an Inception block at one depth level (model.py:226-262:
 * the reference runs them one after the other): each alone is 120..273 workgroups of a 64 x 64 tile on 256 CUs.
 * tile: 1 = 64x64, 3 = 96x64, 5 = 128x64, 7 = 128x128 (0 = 1; 7 falls back to 5 when some Cin % 64 != 0).  The item array
 * is HOST memory, read during the call. */
#define SBA_GROUP_MAX 8
typedef struct sba_conv_group_item {
    const void* x; const void* w; void* y; const void* addend; const float* bias; const void* relu_mask;
    const sba_conv_geom* g;
} sba_conv_group_item;
int sba_conv_igemm_group(int dtype, int n, const sba_conv_group_item* items, int tile, void* stream);
/* The same with split-K: every item's K range (ntaps * Cin) is cut into `ksplit` slices (grid.z), partial sums meet in
 * f32 atomics in the item's own [M][Cout] slice of `workspace` (zero-filled when handed in, left zero-filled; needs
 * sum_i 4 * M_i * Cout_i bytes, 16-byte aligned) and ONE finishing launch for the whole group applies the epilogues.
 * Falls back to ksplit = 1 when the workspace is too small, some Cin % 64 != 0, or in the deterministic mode.
 * Used for the four parity classes of the data gradient of the 4x4 stride-2 conv (model.py:552 downBlock backward):
 * they read the same dy, write disjoint pixels of dx, and were four launches of 80..640 workgroups each. */
int sba_conv_igemm_group_splitk(int dtype, int n, const sba_conv_group_item* items, int tile, int ksplit,
                                void* workspace, int64_t workspace_bytes, void* stream);
/* dw[co][t][ci] += sum_pixel dy[pixel][co] * x[gather(pixel,t)][ci]   (f32 accumulate/output).
 * ksplit > 1 splits the pixel range over that many workgroups (atomic accumulation). */
int sba_conv_wgrad(int dtype, const void* x, const void* dy, float* dw, const sba_conv_geom* g,
                   int ksplit, void* stream);
/* weight packing: master f32 [Cout][KH][KW][Cin] (= torch channels_last storage of an OIHW
 * parameter) -> packed `dtype` weights.  mode 0: same order (cast).  mode 1: data-gradient
 * weights of a stride-1 'same' conv: out[ci][KH-1-kh][KW-1-kw][co].  mode 2: data-gradient of
 * the 4x4 stride-2 pad-1 conv, four parity classes: out[(py*2+px)][ci][j*2+i][co] with
 * kh = (1-py) + 2j, kw = (1-px) + 2i.  mode 3 (3x3 only): data-gradient of (nearest x2 -> conv3x3)
 * collapsed into ONE 4x4 stride-2 pad-1 conv over dy (16/36 of the FLOPs of the conv at the upsampled
 * resolution + 2x2 sum pooling): out[ci][dd*4+ee][co] = sum_{kh in S(dd), kw in S(ee)} w[co][kh][kw][ci]
 * with S(0)={2}, S(1)={1,2}, S(2)={0,1}, S(3)={0}; 16*Cin*Cout elements. */
int sba_pack_weight(int dtype, const float* w, void* out, int Cout, int KH, int KW, int Cin,
                    int mode, void* stream);
/* All packed copies of ONE network's conv weights in one launch (after its optimizer step, cf.
 * trainer.py:275,296): per tensor the forward operand (mode-0 cast, skipped when fwd is NULL)
 * and the data-gradient operand (mode 1, 2 or 3 as above, skipped when tr is NULL) are written from a
 * single read of the f32 master.  `descs` is a DEVICE array; workgroup b serves the tensor d with
 * tile_begin[d] <= b < tile_begin[d+1] (tiles: tap-major, then 64-row Cout tiles, then 64-column
 * Cin tiles; co_tiles = ceil(Cout/64), ci_tiles = ceil(Cin/64)); total_tiles = sum over tensors
 * of slots*co_tiles*ci_tiles with slots = KH*KW (16 for mode 3).  Cin must be a multiple of 4. */
typedef struct sba_pack_desc {
    const float* w;
    void* fwd;
    void* tr;
    int32_t Cout, KH, KW, Cin;
    int32_t mode;
    int32_t tile_begin;
    int32_t co_tiles, ci_tiles;
} sba_pack_desc;
int sba_pack_weights_multi(int dtype, const sba_pack_desc* descs, int ndesc, int total_tiles, void* stream);
/* Row-major packed bf16 conv operands [R][taps][K] (R, K multiples of 64: the outputs of sba_pack_weight /
 * sba_pack_weights_multi; R a multiple of 32, K of 16) -> FRAGMENT-MAJOR copies [ceil(R/64)][taps][2][K/16][64][8] (the
 * destination holds ceil(R/64)*64 * taps * K elements): the 64 lanes' 16-byte MFMA B fragments of one (tap, 32-row tile,
 * 16-deep k-step) contiguous (1 KB), lane = ((k >> 3) & 1) * 32 + (r & 31).  `descs` is a DEVICE
 * array; one work unit = 16 bytes, tensor d owns units [unit_begin[d], unit_begin[d+1]); total_units = sum R*taps*K/8. */
typedef struct sba_frag_desc {
    const void* src;
    void* dst;
    int32_t R, taps, K, unit_begin;
} sba_frag_desc;
int sba_pack_frag_multi(const sba_frag_desc* descs, int ndesc, int total_units, void* stream);
/* sum each 2x2 block: dx[n][y][x][c] = sum dup[n][2y+a][2x+b][c] (bwd of nearest x2). */
int sba_pool2x2_sum(int dtype, const void* dup, void* dx, int N, int H, int W, int C, void* stream);

/* ---- BatchNorm(train) + activation (model.py:43-44,62-65,543-544,553-554) ---- */
/* BatchNorm batch statistics are accumulated in SBA_BN_STAT_SLOTS replicas ("slots") of the
 * (sum[C], sumsq[C]) pair: a conv epilogue adds into slot (workgroup index mod SLOTS), which divides the
 * same-address atomic contention of thousands of workgroups by SLOTS; readers add the slots up.  A
 * statistics buffer is therefore stats[groups][SBA_BN_STAT_SLOTS][2C] floats, zeroed by the caller. */
#ifndef SBA_BN_STAT_SLOTS
#define SBA_BN_STAT_SLOTS 8
#endif
/* All BatchNorm entry points take `groups` >= 1 independent BatchNorm batches laid back to back
 * (`rows` NHWC rows each; the discriminator's real | fake passes of losses.py:139-140 share one conv
 * launch); per-group arrays are stats[groups][SLOTS][2C], aux[groups][4C] (= scale, shift, mean, rstd),
 * red[groups][2C].  C a power of two <= 4096. */
/* stats[g][slot][0..C) += sum(y), stats[g][slot][C..2C) += sum(y^2) (caller zeroes): only needed when the conv
 * epilogue could not produce them (groups > 1). */
int sba_bn_stats(int dtype, const void* y, float* stats, int64_t rows, int groups, int C, void* stream);
/* finalize + normalise + activation in one launch: scale/shift/mean/rstd from stats (training) or
 * from the running statistics (training == 0) -> aux; running stats update per group in order
 * (momentum, unbiased var), num_batches_tracked += groups; out = act(y*scale+shift) (+ residual).
 * GLU halves the channel count.  out may have a larger channel stride (out_cstride) and offset
 * (out_coff) so that it can be written into a concat. */
int sba_bn_act_fwd(int dtype, const void* y, const float* stats, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float* aux,
                   const void* residual, void* out, int64_t rows, int groups, int C, int act,
                   int out_cstride, int out_coff, float eps, float momentum, int training, void* stream);
/* pass 1 of the backward: red[g][slot][0..C) += sum dz, red[g][slot][C..2C) += sum dz*xhat, where dz is the
 * gradient w.r.t. the BN output (activation backward applied to dout on the fly).  Like the forward statistics the
 * sums are spread over SBA_BN_STAT_SLOTS replicas: red is [groups][SBA_BN_STAT_SLOTS][2C] floats, zeroed by the
 * caller; pass 2 adds the replicas up. */
int sba_bn_act_bwd_reduce(int dtype, const void* y, const void* dout, const float* aux, float* red,
                          int64_t rows, int groups, int C, int act, int dout_cstride, int dout_coff,
                          void* stream);
/* pass 2: dy = gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)); also dgamma += sum_slots red[g][.][C+c],
 * dbeta += sum_slots red[g][.][c]. */
int sba_bn_act_bwd_apply(int dtype, const void* y, const void* dout, const float* aux, const float* red,
                         void* dy, float* dgamma, float* dbeta, int64_t rows, int groups, int C, int act,
                         int dout_cstride, int dout_coff, void* stream);
/* training-mode forward for small maps WITHOUT conv-epilogue statistics (grouped passes): batch statistics,
 * finalize (aux, running stats per group in order, num_batches_tracked += groups) and normalise + activation
 * in ONE launch; no residual. */
int sba_bn_act_fwd_fused(int dtype, const void* y, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, int64_t* num_batches_tracked, float* aux, void* out, int64_t rows,
                         int groups, int C, int act, int out_cstride, int out_coff, float eps, float momentum,
                         void* stream);
/* both passes in ONE launch for small maps (a workgroup owns a channel vector over all rows of a group):
 * same results as reduce + apply; the host picks it when rows per group is at most a few thousand. */
int sba_bn_act_bwd_fused(int dtype, const void* y, const void* dout, const float* aux, void* dy,
                         float* dgamma, float* dbeta, int64_t rows, int groups, int C, int act,
                         int dout_cstride, int dout_coff, void* stream);
/* Linear(no bias)+BatchNorm1d(train)+GLU on [B][F] f32, output permuted to NHWC [B][4*4][F/2/16]
 * (INIT_STAGE_G.fc + view, model.py:353-356,372-373). */
int sba_bn1d_glu_fwd(int dtype, const float* y, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, int64_t* nbt, float* mean,
                     float* rstd, void* out, int B, int F, float eps, float momentum, void* stream);
int sba_bn1d_glu_bwd(int dtype, const float* y, const void* dout, const float* gamma,
                     const float* beta, const float* mean, const float* rstd, float* dy,
                     float* dgamma, float* dbeta, int B, int F, void* stream);

/* ---- small dense layers, f32 (CA_NET, MAPPING_NET, INIT fc, AdaIN style; model.py:278,306-313,330,354) ---- */
int sba_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N,
                   void* stream);
/* dx = dy*W (may be NULL); dw += dy^T x; dbias += sum dy (may be NULL). */
int sba_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw,
                   float* dbias, int B, int K, int N, void* stream);
/* CA_NET tail (model.py:281-294): h[B][4c] -> GLU -> mu|logvar; c = eps*exp(.5*logvar)+mu. */
int sba_ca_fwd(const float* h, const float* eps, float* c, float* mu, float* logvar, int B, int C,
               void* stream);
int sba_ca_bwd(const float* h, const float* eps, const float* dc, const float* dmu,
               const float* dlogvar, float* dh, int B, int C, void* stream);

/* attention key projection conv_context (GlobalAttention.py:75,95-97): src[b][i][l] = sum_c W[i][c] words[b][c][l].
 * bwd: dW += ..., dwords = ... (may be NULL). */
int sba_ctx_proj_fwd(const float* words, const float* W, float* src, int B, int idf, int cdf, int L,
                     void* stream);
/* sba_ctx_proj_fwd with FP8 (OCP e4m3) operands on v_mfma_f32_32x32x16_fp8_fp8, block-scaled per 32-row tile,
 * f32 accumulate (BASELINE config 5).  cdf % 16 == 0.  Relative L2 error of src <= 8e-2 on N(0,1) data. */
int sba_ctx_proj_fwd_fp8(const float* words, const float* W, float* src, int B, int idf, int cdf, int L,
                         void* stream);
int sba_ctx_proj_bwd(const float* words, const float* W, const float* dsrc, float* dW, float* dwords,
                     int B, int idf, int cdf, int L, void* stream);

/* ---- AdaIN (model.py:324-339) ---- */
/* per (n,c) sum/sumsq of h over HW -> mean/rstd (biased var, eps). */
int sba_instnorm_stats(int dtype, const void* h, float* mean, float* rstd, int N, int HW, int C,
                       float eps, void* stream);
/* out[..., coff + c] = (style[n][c]+1)*xhat + style[n][C+c]. */
int sba_adain_fwd(int dtype, const void* h, const float* mean, const float* rstd, const float* style,
                  void* out, int N, int HW, int C, int out_cstride, int out_coff, void* stream);
/* red[n][c][0..2) += (sum dout*xhat, sum dout)  -> dstyle; */
int sba_adain_bwd_reduce(int dtype, const void* h, const void* dout, const float* mean,
                         const float* rstd, float* red, int N, int HW, int C, int dout_cstride,
                         int dout_coff, void* stream);
/* dh (+)= (style+1)*rstd*(dout - mean(dout) - xhat*mean(dout*xhat)); dstyle[n][c] = red sumxhat,
 * dstyle[n][C+c] = red sum.  accumulate != 0 adds into dh. */
int sba_adain_bwd_apply(int dtype, const void* h, const void* dout, const float* mean,
                        const float* rstd, const float* style, const float* red, void* dh,
                        float* dstyle, int N, int HW, int C, int dout_cstride, int dout_coff,
                        int accumulate, void* stream);

/* ---- word attention (GlobalAttention.py:82-121) ---- */
/* src[B][idf][L] f32 is conv_context applied to the words (use sba_linear_fwd).
 * mask[B][L] uint8 or NULL.  mask_mode 0 = the reference's row order
 * (row r = b*Q+q of the score matrix takes mask[r % B]), 1 = per-sample mask[b].
 * ctx written at channel offset out_coff of a tensor with out_cstride channels.
 * att (f32 [B][L][Q]) may be NULL. */
int sba_word_attn_fwd(int dtype, const void* h, const float* src, const uint8_t* mask, void* ctx,
                      float* att, int B, int Q, int idf, int L, int mask_mode, int out_cstride,
                      int out_coff, void* stream);
/* BASELINE config 5 ("fp8 MFMA for the attention / context GEMM"): the same forward (bf16 activations) with BOTH
 * contractions of GlobalAttention.py:103,117 on v_mfma_f32_32x32x16_fp8_fp8 -- OCP e4m3 operands, f32 accumulate, h scaled
 * per 32-query tile and the keys per image by powers of two (exact un-scaling), probabilities x 256.  Forward only: the
 * backward (sba_word_attn_bwd) keeps the bf16 / f32 operands (straight-through w.r.t. the quantisation).  Stated tolerance:
 * context rel L2 <= 8e-2, attention map abs <= 0.12 on N(0,1)-scaled inputs (tests/test_kernels_gpu.py). */
int sba_word_attn_fwd_fp8(const void* h, const float* src, const uint8_t* mask, void* ctx, float* att, int B, int Q,
                          int idf, int L, int mask_mode, int out_cstride, int out_coff, void* stream);
/* dh = d/dh, dsrc[B][idf][L] += d/dsrc (f32, caller zeroes).  accumulate != 0: dh += . */
int sba_word_attn_bwd(int dtype, const void* h, const float* src, const uint8_t* mask,
                      const void* dctx, void* dh, float* dsrc, int B, int Q, int idf, int L,
                      int mask_mode, int dctx_cstride, int dctx_coff, int accumulate, void* stream);

/* ---- image head: conv3x3(ngf->3)+tanh, NHWC in, NCHW f32 out (model.py:426-437) ---- */
int sba_img_head_fwd(int dtype, const void* h, const float* w, float* img, int N, int H, int W,
                     int C, void* stream);
/* dh (+)= ..., dw[3][3][3][C] += ...  (w and dw in channels_last [co][kh][kw][ci]). */
int sba_img_head_bwd(int dtype, const void* h, const float* w, const float* img, const float* dimg,
                     void* dh, float* dw, int N, int H, int W, int C, int accumulate, void* stream);

/* ---- discriminator stem: conv4x4 s2 p1 (3->ndf) + LeakyReLU(0.2), NCHW f32 in, NHWC out
 *      (model.py:563-564) ---- */
int sba_d_stem_fwd(int dtype, const float* img, const float* w, void* out, int N, int S, int C,
                   void* stream);
/* dw[C][4][4][3] += ; dimg (NCHW f32, may be NULL) = . `out` is the saved forward output. */
int sba_d_stem_bwd(int dtype, const float* img, const float* w, const void* out, const void* dout,
                   float* dimg, float* dw, int N, int S, int C, void* stream);

/* ---- logits head: conv4x4 s4 (8ndf->1, bias) + sigmoid on a 4x4 map (model.py:590-592,606-607) ---- */
int sba_logits_fwd(int dtype, const void* h, const float* w, const float* bias, float* prob, int B,
                   int K, void* stream);
int sba_logits_bwd(int dtype, const void* h, const float* w, const float* prob, const float* dprob,
                   void* dh, float* dw, float* dbias, int B, int K, int accumulate, void* stream);
/* cat(h[B][16][C], sent[B][E] tiled 4x4) -> out[B][16][C+E]  (model.py:597-600) */
int sba_cond_cat_fwd(int dtype, const void* h, const float* sent, void* out, int B, int C, int E,
                     void* stream);
int sba_cond_cat_bwd(int dtype, const void* dout, void* dh, float* dsent, int B, int C, int E,
                     int accumulate, void* stream);

/* ---- frozen image encoder pieces (CNN_ENCODER, model.py:162-267; forward + backward-data only) ---- */
/* bilinear resize, align_corners=True (model.py:210), NCHW f32 [NC][S][S] -> [NC][D][D];
 * backward != 0: `in` is d(out) [NC][D][D] and `out` receives d(in) [NC][S][S]. */
int sba_resize_bilinear(const float* in, float* out, int NC, int S, int D, int backward, void* stream);
/* Conv2d_1a_3x3: conv3x3 s2 p0 (3->C) + bias + ReLU, NCHW f32 image -> NHWC features. */
int sba_enc_stem_fwd(int dtype, const float* img, const float* w, const float* bias, void* out, int N, int S, int C,
                     void* stream);
int sba_enc_stem_bwd(int dtype, const float* w, const void* out, const void* dout, float* dimg, int N, int S, int C,
                     void* stream);
/* both of the above in one launch: the stem conv (C = 32) reading the S -> D bilinear resize of the image on the fly -- the
 * D x D tensor (model.py:210: 299 x 299) is never written; the interpolation is sba_resize_bilinear's expression. */
int sba_enc_stem_resize_fwd(int dtype, const float* img, const float* w, const float* bias, void* out, int N, int S, int D,
                            int C, void* stream);
/* F.max_pool2d(x, 3, 2) on NHWC channel slices (cs = channel stride, co = channel offset); the backward
 * routes each window's gradient to its first maximum in scan order (torch's tie rule). */
int sba_maxpool3x3s2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, int xcs, int xco, int ycs,
                         int yco, void* stream);
int sba_maxpool3x3s2_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int C, int xcs,
                         int xco, int dycs, int dyco, int dxcs, int dxco, int accumulate, void* stream);
/* the same pair with the argmax kept: fwd also writes argmax[N][OH][OW][C] (uint8: kh*3+kw of the first maximum, 8-byte
 * aligned), bwd reads it instead of re-deriving it from x (4 instead of 36 input loads per vector) */
int sba_maxpool3x3s2_fwd_arg(int dtype, const void* x, void* y, uint8_t* argmax, int N, int H, int W, int C, int xcs,
                             int xco, int ycs, int yco, void* stream);
int sba_maxpool3x3s2_bwd_arg(int dtype, const uint8_t* argmax, const void* dy, void* dx, int N, int H, int W, int C,
                             int dycs, int dyco, int dxcs, int dxco, int accumulate, const void* relu_mask, void* stream);
/* (relu_mask, optional: a tensor laid out like dx -- the pooled tensor itself when a ReLU produced it; dx is zeroed where
 * it is <= 0, after the accumulation: the ReLU's backward without a pass of its own) */
/* F.avg_pool2d(x, 3, 1, 1) (count_include_pad); self-adjoint, so it is also its own backward. */
int sba_avgpool3x3(int dtype, const void* x, void* y, int N, int H, int W, int C, int xcs, int xco, int ycs, int yco,
                   int accumulate, void* stream);
/* dpre[rows][C] = (out > 0) ? dout : 0 on channel slices. */
int sba_relu_bwd(int dtype, const void* out, const void* dout, void* dpre, int64_t rows, int C, int ocs, int oco,
                 int dcs, int dco, void* stream);
/* F.avg_pool2d over the whole HW map: x NHWC -> y [N][C] f32; backward != 0: x receives dy / HW. */
int sba_global_avgpool(int dtype, void* x, float* y, int N, int HW, int C, int backward, void* stream);
/* NHWC features <-> NCHW f32 (to_nhwc != 0: nchw -> nhwc). */
int sba_layout_nhwc_nchw(int dtype, void* nhwc, float* nchw, int N, int HW, int C, int to_nhwc, void* stream);

/* ---- losses ---- */
/* loss[0] = sum_s weight[s] * mean BCE(prob_s, target[s]) over nseg segments given by
 * offsets[s]..offsets[s+1] into the concatenated prob array (nn.BCELoss, losses.py:144-158,175-178).
 * dprob = d loss / d prob (scaled later by the upstream gradient). */
int sba_bce_multi(const float* prob, const int32_t* offsets, const float* target, const float* weight,
                  int nseg, float* loss, float* dprob, void* stream);
/* KL_loss (losses.py:210-214): loss, dmu, dlogvar. */
int sba_kl_loss(const float* mu, const float* logvar, float* loss, float* dmu, float* dlogvar,
                int n, void* stream);
/* DAMSM word loss (losses.py:62-132 with func_attention GlobalAttention.py:31-69).
 * feat[B][nef][R] f32 (R = 17*17), words[B][nef][L] f32, cap_lens int64[B].
 * sim[j][i] (image j, caption i) = log sum_t exp(g2 * cos(word_t, attended context)).
 * Saves attn[B*B][Lmax][R] and attn1 (first softmax) for the backward. */
int sba_damsm_words_fwd(const float* feat, const float* words, const int64_t* cap_lens, float* sim,
                        float* attn, float* attn1, float* wctx, int B, int nef, int R, int L,
                        float gamma1, float gamma2, void* stream);
int sba_damsm_words_bwd(const float* feat, const float* words, const int64_t* cap_lens,
                        const float* sim, const float* attn, const float* attn1, const float* wctx,
                        const float* dsim, float* dfeat, float* dwords, int B, int nef, int R, int L,
                        float gamma1, float gamma2, void* stream);
/* sentence similarity matrix (losses.py:41-47): s[j][i] = cos(cnn[j], rnn[i]) * g3. */
/* The same loss on the bf16 MATRIX CORES (csrc/damsm_mfma.hip; nef % 64 == 0, R <= 384, L <= 32 -- sba_damsm_prep_bytes
 * returns -1 otherwise and the callers keep the f32 kernels above).  Every product is formed from bf16 hi + lo parts of its
 * f32 factors (xh yh + xl yh + xh yl, f32 sums): results agree with the f32 kernels to ~1e-5 relative.
 *   sba_damsm_prep: features / words -> hi + lo operands in `prep` (caller-owned scratch of sba_damsm_prep_bytes bytes,
 *     16-byte aligned; read by the forward AND the backward of the same inputs).
 *   forward: one workgroup per (caption, image) pair; same outputs as sba_damsm_words_fwd.
 *   backward: `scratch` (caller-owned, sba_damsm_bwd_bytes bytes, 16-byte aligned, any contents): pass 1 (per pair)
 *     fills it with d(context), the attention and d(scores) as bf16 hi + lo MFMA fragments, pass 2 forms
 *     dfeat[j] (= or +=, `accumulate`) sum over (caption, word) of A x dcontext + dS x q as ONE contraction per image --
 *     every element of dfeat has one owner: no atomics, the same bits in every run.  dwords (may be NULL) is accumulated
 *     with f32 atomics (deterministic mode: ordered fold). */
int64_t sba_damsm_prep_bytes(int B, int nef, int R, int L);
int64_t sba_damsm_bwd_bytes(int B, int nef, int R, int L);
int sba_damsm_prep(const float* feat, const float* words, const int64_t* cap_lens, void* prep, int64_t prep_bytes,
                   int B, int nef, int R, int L, void* stream);
int sba_damsm_words_fwd_mfma(const void* prep, const float* words, const int64_t* cap_lens, float* sim, float* attn,
                             float* attn1, float* wctx, int B, int nef, int R, int L, float gamma1, float gamma2,
                             void* stream);
int sba_damsm_words_bwd_mfma(const void* prep, const float* words, const int64_t* cap_lens, const float* sim,
                             const float* attn, const float* attn1, const float* wctx, const float* dsim,
                             void* scratch, int64_t scratch_bytes, float* dfeat, int accumulate, float* dwords,
                             int B, int nef, int R, int L, float gamma1, float gamma2, void* stream);
/* The loss heads of the generator step with a FROZEN text side (losses.py:187-204: the upstream gradient of the four
 * cross-entropy terms is the constant LAMBDA), each as ONE launch instead of ce_pair + combine2 + scalar arithmetic:
 *   sba_ce_pair_direct: loss_out[0] = lam (loss0 + loss1) of the two cross entropies over the B x B scores (x scale,
 *     masked), dscore = d(that) / d(score);
 *   sba_damsm_sent_direct: the whole sentence loss -- cosine scores x gamma3, both cross entropies,
 *     loss_out[0] = lam (loss0 + loss1), dcnn = d(that) / d(cnn) (STORED; may be NULL); one workgroup, B <= 96,
 *     deterministic (partners walked in order). */
int sba_ce_pair_direct(const float* score, const uint8_t* mask, float scale, float lam, float* loss_out,
                       float* dscore, int B, void* stream);
int sba_damsm_sent_direct(const float* cnn, const float* rnn, const uint8_t* mask, float gamma3, float eps, float lam,
                          float* loss_out, float* dcnn, int B, int nef, void* stream);
int sba_damsm_sent_fwd(const float* cnn, const float* rnn, float* s, int B, int nef, float gamma3,
                       float eps, void* stream);
int sba_damsm_sent_bwd(const float* cnn, const float* rnn, const float* ds, float* dcnn, float* drnn,
                       int B, int nef, float gamma3, float eps, void* stream);
/* two cross entropies over a B x B score matrix and its transpose with labels = arange(B) and the
 * same-class mask (uint8[B][B], may be NULL) filled with -inf (losses.py:51-56,123-129):
 * loss[0] = CE(rows), loss[1] = CE(cols); dscore0/dscore1 = their gradients. `scale` multiplies
 * the scores first (gamma3 for the word loss). */
int sba_ce_pair(const float* score, const uint8_t* mask, float scale, float* loss, float* dscore0,
                float* dscore1, int B, void* stream);

/* out[k] = ga[0]*a[k] + gb[0]*b[k]: combines the two CE gradients with their upstream (device) scalars. */
int sba_combine2(float* out, const float* a, const float* ga, const float* b, const float* gb, int n,
                 void* stream);

/* ---- optimizer (trainer.py:136-143,275,297-299) ---- */
/* state = {int32 step; float step_size; float inv_sqrt_bc2}: step += 1 and the Adam bias
 * corrections for (lr, beta1, beta2), computed on device so the step is graph-replayable. */
int sba_adam_prepare(void* state, float lr, float beta1, float beta2, void* stream);
/* Adam update of n contiguous f32 parameters; avg != NULL: EMA avg = .999*avg + .001*p;
 * shadow != NULL: bf16 copy of the updated parameters (the packed forward weights). */
int sba_adam_step(float* p, const float* g, float* m, float* v, float* avg, void* shadow,
                  const void* state, int64_t n, float beta1, float beta2, float eps, float grad_scale,
                  void* stream);
/* RNN_ENCODER.forward of the frozen text encoder (model.py:127-159; trainer.py:248-252 calls it under
 * eval/no_grad every step): Embedding -> one-layer bidirectional LSTM over packed sequences.
 *   captions [B][T] int64 token ids, cap_lens [B] int64 (device memory: replaces the reference's
 *   cap_lens.tolist() host sync, model.py:139), emb_weight [ntoken][ninput],
 *   w_ih [2][4H][ninput], w_hh [2][4H][H], b_ih / b_hh [2][4H]  (= weight_ih_l0 | weight_ih_l0_reverse ...,
 *   PyTorch gate order i|f|g|o), h0 / c0 [2][B][H] or both NULL (= init_hidden zeros),
 *   gx_scratch [2][B*T][4H] f32 workspace.
 * Outputs: words [B][2H][Lout] (zeros past each caption's length, like pad_packed_sequence; Lout <= T is
 * the reference's max(cap_lens) when the host knows it, else T), sent [B][2H] (last valid hidden state,
 * forward | backward).  H in {64, 128}, ninput % 4 == 0. */
int sba_lstm_bidir_fwd(const int64_t* captions, const int64_t* cap_lens, const float* emb_weight,
                       const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                       const float* h0, const float* c0, float* gx_scratch, float* words, float* sent,
                       int B, int T, int Lout, int ntoken, int ninput, int H, void* stream);
/* ---- multi-stream launch replayer (host side of the step loop; no counterpart in the reference, whose
 * trainer.py:245-299 issues every launch from Python) ----
 * sba_replay_create walks a CAPTURED hipGraph_t (stream capture of one training step, or of one phase of it:
 * kernel / 1-D memcpy / memset / empty nodes and their edges), assigns every node a stream -- a chain keeps its
 * stream, a fork takes the next of at most max_streams -- and stores the launch list; sba_replay_launch
 * re-issues the nodes in topological order as plain asynchronous launches on those streams, with events only
 * where a dependency crosses streams; it begins after everything already queued on `stream` and `stream` waits
 * for its end.  flags: bit 0 = print a summary to stderr; bit 1 = issue the first chain of the graph on `stream`
 * ITSELF (for the per-phase graphs: a single-chain graph then uses no stream of its own -- a process has 4 hardware
 * queues by default and every extra stream shares one; max_streams = 4 is the measured optimum for the whole step).
 * Do not set bit 1 for callers on the NULL stream.  Unlike hipGraphLaunch on ROCm 7.2, independent branches then really run concurrently, at ~3 us
 * of host time per launch.  The graph must outlive the handle (kernel arguments live in its nodes); the handle
 * owns its streams and events.  Returns SBA_E_UNSUPPORTED for node kinds it cannot re-issue (callers fall
 * back to hipGraphLaunch).  info8 = {nodes, kernels, copies, memsets, streams, cross-stream waits, events, 0}.
 * This is the one entry point family that creates HIP objects (streams / events), at create time only. */
int sba_replay_create(void* hip_graph, int max_streams, int flags, void** out_handle);
int sba_replay_launch(void* handle, void* stream);
int sba_replay_info(void* handle, int* info8);
/* PRIORITIES for the longest dependency path.  Kernels of independent chains that run side by side share the CUs, and the
 * chain the step's duration hangs on (generator forward -> image encoder -> DAMSM -> their backward passes) is slowed by the
 * three discriminator updates beside it as much as they are by it.  sba_replay_prioritize
 *   1. issues every recorded node ONCE, alone, in order on `stream` with an event between neighbours -- this IS one
 *      execution of the recording (host-call nodes included: the callback runs), the caller counts it as a replay -- and keeps
 *      the durations;
 *   2. computes every node's earliest start and longest path to the end from them; nodes whose slack against the longest
 *      path of the whole recording is <= slack_frac of it are CRITICAL;
 *   3. assigns streams again, in two pools: n_high streams for the critical nodes, max_streams - n_high for the rest, created
 *      with hipStreamCreateWithPriority: mode 1 = (high, normal), 2 = (high, low), 3 = (normal, low); 0 = one pool, no
 *      priorities (only max_streams changes).
 * The dependencies enforced are the recorded ones in every mode: results do not depend on it.  verbose: 1 = a summary on
 * stderr, 2 = also the longest path node by node (start, duration alone, kernel, grid), 3 = also every node (duration alone, slack). */
int sba_replay_prioritize(void* handle, void* stream, int mode, int max_streams, int n_high, float slack_frac, int verbose);
/* HOST-CALL nodes.  sba_replay_marker launches a no-op kernel carrying `tag` (>= 0) on `stream`; captured into the graph it
 * becomes a node with the dependencies of its stream position.  A replay does not launch that node: it calls the callback
 * registered with sba_replay_set_callback -- void fn(int tag, void* stream, void* user) -- with the stream the node was
 * assigned to, after everything the node depends on has been issued.  The host issues there what cannot be recorded: an
 * RCCL collective (on its own stream, ordered behind `stream`), the wait for one (make `stream` wait for it), launches
 * that depend on host state.  This is how the DATA-PARALLEL step is one recording: the gradient exchanges sit between
 * its launches exactly where the eager step has them.  Without a callback the node is skipped.  info8[7] of
 * sba_replay_info = the number of such nodes. */
int sba_replay_marker(int tag, void* stream);
int sba_replay_set_callback(void* handle, void* fn, void* user);
int sba_replay_destroy(void* handle);
/* Training path of the same encoder (DAMSM pre-training, pretrain_DAMSM.py:49-130).
 *   sba_lstm_recur_train: the packed bidirectional recurrence on pre-computed input projections
 *     gx [2][B*T][4H] (= x W_ih^T + b_ih + b_hh, a plain GEMM the caller runs), also storing what BPTT needs:
 *     gates [2][B*T][4H] (i | f | g | o after their nonlinearities), cs / hs [2][B*T][H];
 *   sba_lstm_recur_bwd: back-propagation through time given d words [B][2H][Lout] and d sent [B][2H]:
 *     dG [2][B*T][4H] = gradient w.r.t. the gate pre-activations and hprev [2][B*T][H] = the hidden state every
 *     step started from (both must be ZERO-FILLED by the caller: rows past a caption's length are not written).
 *     The caller finishes with plain GEMMs: dW_ih = dG^T x, dW_hh = dG^T hprev, db = sum dG, dx = dG W_ih. */
int sba_lstm_recur_train(const float* gx, const int64_t* cap_lens, const float* w_hh, const float* h0,
                         const float* c0, float* words, float* sent, float* gates, float* cs, float* hs,
                         int B, int T, int Lout, int H, void* stream);
int sba_lstm_recur_bwd(const int64_t* cap_lens, const float* w_hh, const float* h0, const float* c0,
                       const float* gates, const float* cs, const float* hs, const float* dwords,
                       const float* dsent, float* dG, float* hprev, int B, int T, int Lout, int H, void* stream);
/* ---- BertEncoder forward (model_bert.py:161-189: frozen BERT-base trunk of the bert / mix variants) ----
 * The dense layers are 1x1 convolutions on sba_conv_igemm_bias (rows = B*L tokens); these are the pieces between
 * them.  Activations [B*L][C] of `dtype`; LayerNorm statistics and the softmax in f32.
 *   embed_ln:  LayerNorm(word_emb[token] + pos_emb[position] + type_emb[0]), eps as given (BERT: 1e-12)
 *   add_ln:    LayerNorm(x + residual)
 *   attention: per (caption, head) softmax(q k^T / 8) v over L <= 32 tokens with NO mask (the reference passes
 *              none, model_bert.py:181); qkv [B*L][3C] = q | k | v, heads of 64 channels
 *   gelu:      exact (erf) GELU in place;  tanh_transpose: y[b][c][l] = tanh(x[b*L + l][c]) (f32 out). */
int sba_bert_embed_ln(int dtype, const int64_t* tokens, const float* word_emb, const float* pos_emb,
                      const float* type_emb, const float* gamma, const float* beta, void* out, int B, int L,
                      int C, int ntoken, float eps, void* stream);
int sba_bert_add_ln(int dtype, const void* x, const void* residual, const float* gamma, const float* beta,
                    void* out, int rows, int C, float eps, void* stream);
int sba_bert_attention(int dtype, const void* qkv, void* ctx, int B, int L, int C, int heads, void* stream);
int sba_bert_gelu(int dtype, void* x, int64_t n, void* stream);
int sba_bert_tanh_transpose(int dtype, const void* x, float* y, int B, int L, int C, void* stream);
/* y = cast(x) between f32 and dtype, n elements. */
int sba_cast(int dtype_dst, void* dst, int dtype_src, const void* src, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SBAGAN_HIP_H */
