/*
 * epipolar_amd.h -- C ABI of the MI355X-native Epipolar Transformer hot path.
 *
 * Drop-in boundary: these entry points are what a
 * binding of the reference's `Epipolar` operator
 * (modeling/layers/epipolar.py:11-269, called from
 * modeling/backbones/resnet.py:377-388) calls instead of the per-sample Python
 * loop.  Plain pointers and sizes only; no torch types.  Every pointer is a
 * DEVICE pointer unless its comment says "host".  The library allocates
 * nothing, never synchronises the device and enqueues all work on `stream`
 * (a hipStream_t passed as void*; NULL = the default stream), so it is
 * re-entrant under nn.DataParallel-style threading (SURVEY.md 8b "Threading").
 *
 * Return value: 0 on success, non-zero on error; et_last_error() then returns
 * a thread-local description.  No error is ever turned into NaNs silently.
 *
 * Feature maps are channels-last: (N, H, W, C) float32, C contiguous, C % 4 == 0.
 * (The reference hands NCHW tensors; et_nchw_to_nhwc / et_nhwc_to_nchw convert,
 * and a torch tensor in torch.channels_last memory format already IS this
 * layout.)
 */
#ifndef EPIPOLAR_AMD_H_
#define EPIPOLAR_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ET_ABI_VERSION 13

/* Static description of one layer call: the cfg keys the reference reads in
 * Epipolar.__init__ (epipolar.py:12-54) and at call time (epipolar.py:303-311,
 * 409-414; vision/multiview.py:25-57,154-163). */
typedef struct EtLayerDesc {
    int32_t N;                 /* (reference view, source view) pairs in the batch          */
    int32_t C;                 /* cfg.KEYPOINT.NFEATS (channels), multiple of 4             */
    int32_t H, W;              /* cfg.KEYPOINT.HEATMAP_SIZE                                 */
    int32_t K;                 /* cfg.EPIPOLAR.SAMPLESIZE, 2 <= K <= 256                    */
    float xmin, ymin;          /* epipolar.py:46-47 (first pixel centre, image coords)      */
    float xmax, ymax;          /* epipolar.py:48-49 (last pixel centre)                     */
    float eps;                 /* epipolar.py:20  (0.001)                                   */
    float downsample;          /* cfg.BACKBONE.DOWNSAMPLE            (multiview.py:159-163) */
    float image_resize;        /* cfg.DATASETS.IMAGE_RESIZE          (epipolar.py:411)      */
    float predict_resize;      /* cfg.DATASETS.PREDICT_RESIZE        (epipolar.py:411)      */
    int32_t correct_normalize; /* cfg.EPIPOLAR.USE_CORRECT_NORMALIZE (multiview.py:29,45)   */
    int32_t align_corners;     /* F.grid_sample semantics (epipolar.py:199): 0 = torch>=1.3 default */
    float softmax_scale;       /* cfg.EPIPOLAR.SOFTMAXSCALE          (epipolar.py:306)      */
    int32_t softmax_enabled;   /* cfg.EPIPOLAR.SOFTMAX_ENABLED       (epipolar.py:303-311)  */
    int32_t src_grad_mask;     /* backward only, cfg.EPIPOLAR.OTHER_GRAD (epipolar.py:141-153):
                                  bit0 = gradient reaches feat_src through the similarity ('other1'),
                                  bit1 = through the sampled values ('other2'); reference default 3 */
    int32_t variant;           /* kernel variant bits, 0 = library default (tuning/ablation) */
} EtLayerDesc;

/* Floats per pair in the `cam` array consumed below:
 *   [0..11]  P1inv  4x3 row-major : pinverse(P_ref)            epipolar.py:336
 *   [12..23] P2     3x4 row-major : P_src                      epipolar.py:340
 *   [24..26] e2     epipole P2 @ centre(P_ref), divided by z   epipolar.py:344-348
 * This O(N) algebra stays on the host in float32 torch. */
#define ET_CAM_STRIDE 27

/* Variant bits (EtLayerDesc.variant).  0 selects the tuned default of the forward kernel
 * (MULTI4 for C == 256 and K <= 64, else BATCH4 | OCC5 | PIXEL_INTERLEAVE); the bits exist
 * for ablation and tuning runs. */
#define ET_VARIANT_SAFE_REDUCE 1  /* cross-lane sums via ds_bpermute only (no permlane swaps / DPP) */
#define ET_VARIANT_NO_TAP_CACHE 2 /* reload all four taps for every sample                          */
#define ET_VARIANT_PIXEL_INTERLEAVE 4 /* wave w of a block takes pixels w, w+4, .. instead of 4w..4w+3 */
#define ET_VARIANT_BATCH4 8       /* cross-lane reductions per 4 samples (fewer registers) instead of 8 */
#define ET_VARIANT_OCC5 16        /* compile for >= 5 waves per SIMD (<= 96 VGPRs)                       */
#define ET_VARIANT_OCC6 32        /* compile for >= 6 waves per SIMD (<= 80 VGPRs, may spill)            */
#define ET_VARIANT_PIPELINE 512    /* multi kernels: request step k+1's rows right after step k is interpolated */
#define ET_VARIANT_MULTI2 1024     /* C == 256: two pixels per wave in lockstep (32 lanes x 8 channels each) */
#define ET_VARIANT_MULTI4 2048     /* C == 256: four pixels per wave in lockstep (16 lanes x 16 channels)    */
#define ET_VARIANT_BWD_ATOMIC 4096 /* backward: float-atomic scatter even when a workspace is given      */
#define ET_VARIANT_BWD_UNSORTED 8192 /* backward gather: sum in arrival order (faster, not bit-reproducible) */
#define ET_VARIANT_NO_TILE 16384  /* host wrappers: do not route C == 256 calls to et_epipolar_forward_tiled */
#define ET_VARIANT_TILE_SPLIT 32768 /* et_epipolar_forward_tiled, testing: 64-row tiles, so that tiles overflow and split */
#define ET_VARIANT_TILE_CLASSIC 65536 /* et_epipolar_forward_tiled: the one-block-per-tile kernel (split-fp16 GEMMs, exact-fp32 redo of overflowing tiles) instead of the warp-specialised persistent one */
/* (bit 131072 was ET_VARIANT_WS_V2, a second-generation persistent kernel on pre-split source planes: measured slower in rounds 3-4 (profiles/r03_ws2_vs_v1_timing.txt), retired in round 5; the bit is reserved and now selects nothing) */
#define ET_VARIANT_WS_SETPRIO 262144 /* warp-specialised kernel (first generation), tuning: s_setprio 1 on the matrix waves */
#define ET_VARIANT_WS_BAND 1048576 /* et_epipolar_forward_tiled / _fused, testing: the persistent kernel's instance for maps above 64 x 64 (288-row arrays, slot table over the tile's band) also for smaller maps */
#define ET_VARIANT_TILE_EXACT 524288 /* et_epipolar_forward_tiled, one-block-per-tile kernel: both GEMMs in exact fp32 (v_mfma_f32_32x32x2_f32) instead of split-fp16 products */
#define ET_VARIANT_BWD_SPLIT_IN_PLACE 2097152 /* et_epipolar_backward_tiled: split EVERY tile beyond the merged kernel's columns in place (rounds 2-4) instead of deferring the hard ones (those whose group chain would outlast the launch) to a second launch of the one-array kernel: equal to 3.5 % faster on the ring rig, 1.5-2 x slower on geometries with many such tiles */
#define ET_VARIANT_BASELINE 256    /* batches of 8, compiler-chosen registers, pixels 4w..4w+3 per wave   */
/* Bits 64 and 128 are reserved: in development builds of the library (-DET_DEV_ABLATE) they switch the per-pixel
 * kernel's tap loads off for roofline ablations (wrong results by construction); a product build rejects them. */
#ifdef ET_DEV_ABLATE
#define ET_VARIANT_ABLATE_NO_LOADS 64
#define ET_VARIANT_ABLATE_ONE_ROW 128
#endif

int et_abi_version(void);
const char *et_last_error(void);

/* grid2sample_locs (epipolar.py:323-418).
 *   xs[W], ys[H] : pixel-centre grid in image coordinates       epipolar.py:35-38
 *   steps[K]     : torch.range(0, 1, 1/(K-1))                   epipolar.py:54
 *   cam[N*27]    : see ET_CAM_STRIDE
 *   sample_locs  : (K, N, H, W, 2) normalised coordinates, as the reference returns them */
int et_sample_locs(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                   const float *cam, float *sample_locs, void *stream);

/* Fused hot loop of Epipolar.forward in the headline mode (MERGE late,
 * ATTENTION avg, SIMILARITY dot): epipolar.py:178-247 + epipolar_similarity
 * (272-321) + de_normalize (multiview.py:39-57), never materialising
 * sample_locs or the K x C x H x W sampled tensor.
 *   feat_ref, feat_src : (N,H,W,C)
 *   out                : (N,H,W,C)  sum_k attn_k * sampled_k           (epipolar.py:243)
 *   attn      nullable : (N,K,H,W)  the reference's `depth` return     (epipolar.py:263)
 *   corr_pos  nullable : (N,H,W,2)  de-normalised arg-max sample       (epipolar.py:237-242)
 *   res_base  nullable : (N,H,W,C)  feat_ref + res_bias[c] (res_bias nullable = 0): the additive
 *                        term of the residual fusion, written while the reference row is in
 *                        registers.  With eval-mode BN folded into z, `ret + feat` (resnet.py:388,
 *                        epipolar.py:250-253) is then ONE GEMM:  x = res_base + (I + W') out . */
int et_epipolar_forward(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                        const float *cam, const float *feat_ref, const float *feat_src, float *out,
                        float *attn, float *corr_pos, const float *res_bias, float *res_base,
                        void *stream);

/* The same operator in its MFMA tile formulation (C == 256 head): reference pixels are ordered by
 * their epipolar line, 32 neighbouring lines form a tile, and the channel-long work of the tile runs
 * as two GEMMs on the matrix cores against the union of the source rows the tile touches (each source
 * row is fetched per tile, not per pixel).  For K <= 64 on maps up to 96 x 96 with the soft-max on
 * (BASELINE configs[0..3]) the call is one persistent block per compute unit whose waves are
 * specialised: matrix waves run the two GEMMs of consecutive tiles back to back as split-fp16 MFMAs
 * with fp32 accumulation (~22 significant bits per product; the fp32 operands are scaled by powers of
 * two, split into fp16 hi + lo at the point of use and every converted value is range-checked), vector
 * waves do the geometry / row-set / soft-max work of the neighbouring tiles meanwhile (two instances:
 * 256-row arrays up to 64 x 64 maps; 288-row arrays and a slot table over the tile's band above).
 * Other shapes (maps above 96 x 96, K > 64) run one block per tile with the same split-fp16 GEMMs.  A tile with a value beyond
 * fp16's range is redone in exact fp32, and so is every call with the soft-max off
 * (EPIPOLAR.SOFTMAX_ENABLED False: the "attention" sim / K is unbounded) and every call with
 * ET_VARIANT_TILE_EXACT.  Same arguments and results as et_epipolar_forward (rounding differs at the
 * 1e-6 level: the sums are re-associated), plus
 *   workspace : device scratch of at least et_epipolar_forward_workspace_bytes(desc) bytes, 256-byte
 *               aligned, ZERO-INITIALISED ONCE by the caller when it is allocated.  It starts with a
 *               64-word header -- word 0 the overflow-tile count, word 1 a STICKY int32 error word
 *               (et_epipolar_forward_workspace_error_offset(desc) = 4 bytes in, whatever the shape, so
 *               that a workspace reused across shapes keeps ONE error word: the kernels only ever OR
 *               bits into it; bit 0 / bit 1: a wave of a persistent kernel gave up waiting at an
 *               internal barrier -- the results of that call are invalid.  Such a barrier exists in
 *               front of the third GEMM of et_epipolar_forward_fused (bit 1; bit 0 belonged to a kernel
 *               retired in round 5); the default kernel's waves never wait on each other
 *               outside the hardware barrier.  The library never synchronises, so the caller reads the word when
 *               it synchronises anyway: ops.check_tile_errors in the Python binding) -- followed by
 *               the per-pair pixel order, the overflow-tile list, per-pair scales, the epipolar
 *               segments in tile order, one base line per tile and the segments by pixel, and
 *                 - one int32 of statistics per tile ( U | groups << 16 : size of the tile's
 *                   source-row set, number of groups it was split into) starting
 *                   et_epipolar_forward_workspace_stats_offset(desc) bytes in, in tile order
 *                   (pair-major), readable by the caller after the call.
 * et_epipolar_forward_workspace_bytes returns 0 when the tile path does not apply to `desc`
 * (then et_epipolar_forward_tiled fails and et_epipolar_forward is the path to call). */
size_t et_epipolar_forward_workspace_bytes(const EtLayerDesc *desc);
size_t et_epipolar_forward_workspace_stats_offset(const EtLayerDesc *desc);
size_t et_epipolar_forward_workspace_error_offset(const EtLayerDesc *desc);
int et_epipolar_forward_tiled(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                              const float *cam, const float *feat_ref, const float *feat_src, float *out,
                              float *attn, float *corr_pos, const float *res_bias, float *res_base,
                              void *workspace, size_t workspace_bytes, void *stream);

/* The operator's parameterised / pooled / prior branches as one kernel.
 * The reference applies its optional 1x1 convolutions to the maps BEFORE sampling (epipolar.py:138-153), so these
 * branches are the headline operator over three tensors instead of two:
 *   q        : (N,H,W,c_sim)  feat1 or theta(feat1)                          (epipolar.py:144-145)
 *   map_sim  : (N,H,W,c_sim)  feat2 or phi(feat2): sampled for the similarity (epipolar.py:138-143, :199)
 *   map_val  : (N,H,W,c_val)  feat2 or g(feat2):   sampled for the output     (epipolar.py:147-153, :210)
 *   prior    nullable : (N,K',H,W)  EPIPOLAR.PRIOR: the (camera, other camera) prior of every pair, added to the masked
 *                        similarity (epipolar.py:300-301) or, with ET_GENERAL_PRIOR_MUL, multiplied onto the soft-max
 *                        output (epipolar.py:308-309)
 *   flags    : ET_GENERAL_POOLING -- EPIPOLAR.POOLING (epipolar.py:200-202, 211-213): per-channel maximum of samples
 *              k and k + K/2 of both sampled maps; K' = K / 2 similarities per pixel (K even).  Otherwise K' = K.
 *   out      : (N,H,W,c_val)  sum_k' attn_k' * (pooled) sample_k' of map_val   (epipolar.py:243)
 *   attn     nullable : (N,K',H,W);  corr_pos nullable : (N,H,W,2), the location of sample arg-max_k' attn of the
 *              unpooled list (epipolar.py:237-242).
 * ATTENTION avg | max, SIMILARITY dot | cos | prior; FIND_CORR rgb is q = ref1, map_sim = ref2 (3 channels); soft-max on or off
 * as `desc` says; desc->C is ignored
 * (c_sim <= 512, c_val <= 4096, any positive value).  Nothing of size K x C x H x W is materialised. */
#define ET_GENERAL_POOLING 1
#define ET_GENERAL_PRIOR_MUL 2
#define ET_GENERAL_COSINE 4          /* SIMILARITY cos: F.cosine_similarity(q, sample) (epipolar.py:290-293), then mask / prior / soft-max as for dot */
#define ET_GENERAL_ATTENTION_MAX 8   /* ATTENTION max (epipolar.py:282-286, 222-235): attn = the raw cosine similarity, out = the arg-max sample of map_val; no prior */
#define ET_GENERAL_SIM_PRIOR 16      /* SIMILARITY prior (epipolar.py:288-289) and an externally supplied `depth` (:217-218): attn = the given (N,K',H,W) weights as they are (no similarity, mask or soft-max; q / map_sim are not read); excludes PRIOR_MUL / COSINE; with ATTENTION_MAX the output is the arg-max sample of those weights */
int et_epipolar_forward_general(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                                const float *cam, const float *q, const float *map_sim, const float *map_val,
                                const float *prior, int c_sim, int c_val, int flags, float *out, float *attn,
                                float *corr_pos, void *stream);

/* Backward of et_epipolar_forward_general w.r.t. its three tensors and the prior, for every branch (same `flags` as
 * the forward; ABI 12 -- ABI 11 covered the dot-product branches without a prior only):  grad_out (N,H,W,c_val)  ->
 * grad_q (N,H,W,c_sim), written;  grad_map_sim (N,H,W,c_sim) and grad_map_val (N,H,W,c_val), each nullable (OTHER_GRAD
 * without 'other1' / 'other2', epipolar.py:138-150) and ACCUMULATED with float atomics -- the caller zeroes them first;
 * sums over pixels arrive in any order, so the result is reproducible to rounding only;  grad_prior nullable:
 * (N,K',H,W), written -- the gradient of every pair's prior rows (the caller sums the pairs of one camera pair into the
 * (camera, other camera) table, epipolar.py:73-80).  The similarities and the soft-max are recomputed from the inputs;
 * the gradient of POOLING's per-channel maximum goes to the sample that won (the first on a tie, as torch.max); cosine
 * similarity is differentiated as torch does (through the unclamped norms); ATTENTION max passes grad_out to the first
 * arg-max sample of map_val only (grad_q = 0, nothing into map_sim); SIMILARITY prior: grad_prior_k' = grad_out . V_k'.
 * Gradients through the `attn` / `corr_pos` outputs are not provided (the reference configurations never use them). */
int et_epipolar_backward_general(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                                 const float *cam, const float *q, const float *map_sim, const float *map_val,
                                 const float *prior, const float *grad_out, int c_sim, int c_val, int flags, float *grad_q,
                                 float *grad_map_sim, float *grad_map_val, float *grad_prior, void *stream);

/* Backward of et_epipolar_forward w.r.t. both feature maps (sample locations
 * carry no gradient, epipolar.py:178-183).  Everything is recomputed from the
 * inputs; nothing saved by the forward is needed.
 *   grad_out  : (N,H,W,C)
 *   grad_ref  : (N,H,W,C) written
 *   grad_src  : (N,H,W,C) written
 *   workspace : device scratch of at least et_epipolar_backward_workspace_bytes(desc)
 *               bytes, or NULL.  With a workspace d(feat_src) is computed in gather
 *               form (per-(pixel,row) coefficients -> counting sort by source row ->
 *               one wave per source pixel sums alpha*g_p + beta*f_p in ascending p):
 *               no float atomics, bit-reproducible.  With NULL (or variant bit
 *               ET_VARIANT_BWD_ATOMIC) grad_src is zero-filled and accumulated with
 *               float atomics (bilinear-transpose scatter, run-to-run rounding noise). */
size_t et_epipolar_backward_workspace_bytes(const EtLayerDesc *desc);
int et_epipolar_backward(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                         const float *cam, const float *feat_ref, const float *feat_src,
                         const float *grad_out, float *grad_ref, float *grad_src, void *workspace,
                         size_t workspace_bytes, void *stream);

/* The backward in the same MFMA tile formulation (C == 256, K <= 256): per 32-pixel tile the similarity and
 * g.S_k come from two GEMMs against the tile's source rows, d(feat_ref) from a third, and d(feat_src) from
 * two pixel-contracting GEMMs whose U x C results are added into grad_src with float atomics (grad_src is
 * zero-filled first).  ~50x fewer atomics than the scatter form and ~3x faster than the gather form, but the
 * cross-tile summation order is not fixed: reproducible to rounding only.  Same arguments as
 * et_epipolar_backward; workspace of et_epipolar_backward_tiled_workspace_bytes(desc) bytes (0: the tile path
 * does not apply to `desc`, call et_epipolar_backward). */
size_t et_epipolar_backward_tiled_workspace_bytes(const EtLayerDesc *desc);
int et_epipolar_backward_tiled(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                               const float *cam, const float *feat_ref, const float *feat_src,
                               const float *grad_out, float *grad_ref, float *grad_src, void *workspace,
                               size_t workspace_bytes, void *stream);

/* The same with the attention the forward returned ((N,K,H,W), what autograd saves in the reference): the soft-max
 * is not recomputed, i.e. one of the five GEMMs, its resampling and the soft-max forward are skipped.  attn may be
 * NULL (= et_epipolar_backward_tiled). */
int et_epipolar_backward_tiled_attn(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                                    const float *cam, const float *feat_ref, const float *feat_src,
                                    const float *attn, const float *grad_out, float *grad_ref, float *grad_src,
                                    void *workspace, size_t workspace_bytes, void *stream);

/* Residual fusion epilogue: x = feat + out + (y * scale[c] + shift[c])
 *   (epipolar.py:250-253 with ZRESIDUAL, then resnet.py:388 `ret + feat`),
 * y = z(out) is the 1x1-conv (GEMM) result, scale/shift = batch-norm affine
 * folded with its (running or batch) statistics.  y/scale/shift may be NULL
 * for the un-parameterised layer (x = feat + out).  finalout nullable:
 * receives out + y*scale + shift (what Epipolar.forward returns).  All (N,H,W,C). */
int et_residual_epilogue(int64_t num_pixels, int32_t C, const float *feat, const float *out,
                         const float *y, const float *scale, const float *shift, float *finalout,
                         float *x, void *stream);

/* The same fusion in eval mode as ONE GEMM for the 256-channel head (replaces the reference's conv1x1 + BN +
 * two adds, epipolar.py:250-253, resnet.py:388):
 *     x[m, :] = [feat[m, :] +] bias + out[m, :] . Wf^T,   Wf = diag(s) W [+ I],  bias = s b + beta - mean s,
 * s = gamma / sqrt(var + eps)  (the caller folds; Wf is (256 out, 256 in) row-major fp32).  The products run as
 * split-fp16 MFMAs with fp32 accumulation (fp32-level error; every row of `out` is scaled by its own power of
 * two).  et_residual_gemm_pack lays Wf out for the kernel into `packed` (et_residual_gemm_packed_bytes() bytes,
 * 16-byte aligned; repack when the weights change).  feat nullable (then x = bias + out . Wf^T, what
 * Epipolar.forward returns).  out / feat / x: (num_pixels, 256) fp32, x may not alias out or feat. */
size_t et_residual_gemm_packed_bytes(void);
int et_residual_gemm_pack(const float *wf, void *packed, void *stream);
int et_residual_gemm(int64_t num_pixels, int32_t C, const float *out, const float *feat, const void *packed,
                     const float *bias, float *x, void *stream);

/* First pass of the layer's TRAINING-mode epilogue (ABI 12): the z branch with BATCH statistics -- bn(z(out)) with
 * training = True (epipolar.py:250-251, modeling/layers/BN.py:59-82) -- for the 256-channel head:
 *   y    : (num_pixels, 256) written: out . Wz^T + z_bias, the batch norm's input (what autograd keeps for the backward)
 *   mean : (256) written: per-channel mean of y over all num_pixels rows
 *   var  : (256) written: per-channel BIASED variance of y (what the batch norm normalises with; the running estimate
 *          takes var * n / (n - 1))
 * `packed_wz` = et_residual_gemm_pack of the raw z weight (256 out x 256 in).  Per 64-row block the kernel keeps a mean and
 * a centred sum of squares per channel (workspace: et_z_batch_stats_workspace_bytes(num_pixels) bytes, no initialisation
 * needed), merged pairwise in double precision -- no sum-of-squares cancellation, no float atomics, bit-reproducible.
 * The second pass is et_residual_gemm with the statistics folded into the weight:
 *   Wf = diag(gamma / sqrt(var + eps)) Wz [+ I],  bias = (z_bias - mean) gamma / sqrt(var + eps) + beta. */
size_t et_z_batch_stats_workspace_bytes(int64_t num_pixels);
int et_z_batch_stats(int64_t num_pixels, int32_t C, const float *out, const void *packed_wz, const float *z_bias, float *y,
                     float *mean, float *var, void *workspace, size_t workspace_bytes, void *stream);

/* Backward of that epilogue, x = bn(z(out)) [+ out] (+ feat) with batch statistics, w.r.t. `out` and the batch norm's affine
 * parameters (ABI 12), from g = d loss / d x and what et_z_batch_stats left (y, mean, invstd = 1 / sqrt(var + eps)):
 *   grad_gamma, grad_beta : (256) written (per-block sums merged in double, no atomics)
 *   grad_y   : (num_pixels, 256) written: the batch norm's input gradient
 *              gamma invstd (g - mean_rows(g) - yhat mean_rows(g yhat)),  yhat = (y - mean) invstd;
 *              d Wz = grad_y^T out and d bz = sum_rows grad_y are left to the caller (library GEMM / reduction)
 *   grad_out : (num_pixels, 256) written: grad_y . Wz, + g when `zresidual` (EPIPOLAR.ZRESIDUAL, epipolar.py:253)
 * `packed_wzt` = et_residual_gemm_pack of the TRANSPOSED z weight.  d feat = g needs no kernel.
 * workspace: et_z_backward_workspace_bytes(num_pixels) bytes, no initialisation needed. */
size_t et_z_backward_workspace_bytes(int64_t num_pixels);
int et_z_backward(int64_t num_pixels, int32_t C, const float *g, const float *y, const float *mean, const float *invstd,
                  const float *gamma, const void *packed_wzt, int32_t zresidual, float *grad_out, float *grad_y, float *grad_gamma,
                  float *grad_beta, void *workspace, size_t workspace_bytes, void *stream);

/* The weight gradient that closes et_z_backward (ABI 12): grad_w (256 out x 256 in) = grad_y^T . out and grad_b (256) =
 * sum_rows grad_y over the num_pixels rows -- a GEMM contracted over the rows on the matrix cores (each fp32 value as three bf16
 * terms, six cross terms per product, fp32 accumulation: no scale, no range limit; rounds 5-6: exact fp32 MFMAs),
 * one block per compute unit, per-block partial results in `workspace` (et_z_wgrad_workspace_bytes bytes, no initialisation
 * needed) summed in a fixed order: no float atomics, bit-reproducible. */
size_t et_z_wgrad_workspace_bytes(int64_t num_pixels);
int et_z_wgrad(int64_t num_pixels, int32_t C, const float *grad_y, const float *out, float *grad_w, float *grad_b, void *workspace,
               size_t workspace_bytes, void *stream);

/* The whole eval-mode layer as ONE data kernel (ABI 11): the sampling + attention of et_epipolar_forward_tiled with
 *     x = feat_ref + bias + out . Wf^T
 * -- `bn(z(out)) + out` (epipolar.py:250-253) and the backbone's `ret + feat` (resnet.py:388) with the eval-mode BN
 * folded into z, i.e. what et_residual_gemm computes from `out` in a second pass -- as a third GEMM of the persistent
 * kernel, on the tile's 32 `out` rows while they are still on chip: `out` is neither written nor re-read (1.07 GB less
 * traffic and one launch less per forward at Config 2).  Applies where the warp-specialised kernel does (C == 256, maps
 * up to 96 x 96, K <= 64, soft-max on, no ET_VARIANT_TILE_CLASSIC); otherwise the call fails and
 * et_epipolar_forward_tiled + et_residual_gemm is the path.
 *   packed_w    : Wf laid out by et_residual_gemm_pack;   bias : (256)
 *   x           : (N,H,W,256)                              attn / corr_pos : as et_epipolar_forward, nullable
 *   out_scratch : (N,H,W,256), required: the rows of tiles the persistent kernel hands to its overflow list (a row set
 *                 beyond its arrays, a source value beyond fp16's range) are produced there by the one-block-per-tile
 *                 kernel and their x rows by a small follow-up kernel; with want_out != 0 every row of `out` is written.
 *   workspace   : as et_epipolar_forward_tiled (same size, same sticky error word). */
int et_epipolar_forward_fused(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                              const float *cam, const float *feat_ref, const float *feat_src, const void *packed_w,
                              const float *bias, float *x, float *attn, float *corr_pos, float *out_scratch,
                              int32_t want_out, void *workspace, size_t workspace_bytes, void *stream);

/* Peak finder of the pose head: find_tensor_peak_batch (modeling/backbones/basic_batch.py:17-63), which
 * PoseResNet.forward calls once per sample in a Python loop (resnet.py:424-430).  One launch for all maps.
 *   heatmaps : (num_maps, H, W) float32, num_maps = N * joints (the NCHW output of final_layer as it lies in memory)
 *   radius   : cfg.KEYPOINT.SIGMA ;  downsample : cfg.BACKBONE.DOWNSAMPLE ;  threshold : 1e-6 in the reference
 *   legacy_floor_division : index / W as integer division (torch < 1.5, what the authors trained with) instead of
 *                           the true division the torch of this image performs
 *   locs     : (num_maps, 2) image coordinates (x, y) ;  scores : (num_maps) the maximum of each map */
int et_heatmap_peaks(int64_t num_maps, int32_t H, int32_t W, const float *heatmaps, float radius, float downsample,
                     float threshold, int32_t legacy_floor_division, float *locs, float *scores, void *stream);

/* Layout converters between the reference's NCHW and the kernels' NHWC. */
int et_nchw_to_nhwc(int32_t N, int32_t C, int32_t H, int32_t W, const float *src, float *dst, void *stream);
int et_nhwc_to_nchw(int32_t N, int32_t C, int32_t H, int32_t W, const float *src, float *dst, void *stream);

/* Test hook (HOST pointers, runs on the CPU, no GPU needed): evaluates the
 * per-sample set-up code that is shared with the device kernels for ONE pair
 * (cam points at its 27 floats), pixel (h, w): taps[k*4 + r] = linear index
 * y*W+x of the source pixel routed to tap register r (-1: outside the image),
 * weights[k*4 + r] = its bilinear weight, locs[k*2 .. k*2+1] = normalised
 * sample location.  It computes no features and is not a CPU fallback of the
 * path. */
int et_debug_host_sample_setup(const EtLayerDesc *desc, const float *xs, const float *ys,
                               const float *steps, const float *cam, int32_t h, int32_t w,
                               int32_t *taps, float *weights, float *locs);

/* Diagnostic (no reference counterpart; bench.py reports it beside the backward's time): `blocks` x 4 waves each issue
 * `iters` float-atomic wave-instructions onto pseudo-random pixel rows of dst (rows, 256) fp32 -- the access pattern of the
 * tiled backward's d(feat_src) accumulation (two 128-byte runs in two rows per instruction) with nothing around it.  dst is
 * modified (1.0 is added); rows < 2^22. */
int et_debug_atomic_probe(float *dst, int64_t rows, int32_t blocks, int32_t iters, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EPIPOLAR_AMD_H_ */
