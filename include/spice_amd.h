/*
 * spice_amd.h — C ABI of the MI355X (gfx950) pruning hot path.
 *
 * One shared library (libspice_amd.so) exports every entry point below with C
 * linkage.  Conventions, identical for all of them:
 *   - every pointer named *_base / student / teacher / labels / preds / target /
 *     conf / out3 / grad / workspace / d_items is a DEVICE pointer owned by the
 *     caller; the library never allocates, frees or retains device memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     every call only enqueues work on it and returns without synchronising;
 *   - the return value is 0 on success, SPICE_ERR_* for an argument problem
 *     detected on the host before anything is enqueued, or a hipError_t value
 *     when a launch fails; no C++ exception crosses the boundary;
 *   - "plan" functions are pure host code: they turn a list of tensor shapes into
 *     a table of fixed-size work items which the caller uploads once and re-uses
 *     for every call on tensors of that structure.
 *
 * The reference toolkit (EPIC 2.0, Python) has no FFI of its own; each entry
 * point names the Python call site whose arithmetic it replaces.
 */
#ifndef SPICE_AMD_H
#define SPICE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPICE_ABI_VERSION 5

#define SPICE_OK 0
#define SPICE_ERR_ARG 10001         /* null pointer, negative size, misaligned base */
#define SPICE_ERR_UNSUPPORTED 10002 /* shape outside what the kernels implement */
#define SPICE_ERR_WORKSPACE 10003   /* workspace smaller than *_workspace_bytes() */
#define SPICE_ERR_BLAS 10004        /* a hipBLASLt call failed */
#define SPICE_ERR_STATIC_LIMIT 10005 /* input beyond the fixed scratch of spice_confusion_update_static */

/* ABI version the library was built with (compare with SPICE_ABI_VERSION). */
int spice_abi_version(void);
/* Static, human-readable text for a return code of this library. */
const char *spice_error_string(int code);

/* ------------------------------------------------------------------------- *
 * (1) Per-output-channel importance reduction over many weight tensors.
 *
 * Replaces, per Conv2d and per calibration batch,
 *   grad.abs() * weight.abs() -> .sum(dim=(1,2,3)) -> scores[module] += ...
 *     (pruning/sensitivity_utils.py:61-66)
 * and torch-pruning 1.6.0 MagnitudeImportance(p)'s per-layer term
 *   weight.flatten(1).abs().pow(p).sum(1)
 *     (constructed at pruning/pruner.py:268-270, pruning/pruning_core.py:143-145,
 *      pruning/strategies.py:53; evaluated inside pruner.step(), pruning/pruner.py:495).
 * All tensors of a model are addressed as element offsets from three base
 * pointers so that one launch covers every layer.
 * ------------------------------------------------------------------------- */
typedef struct spice_rows_desc {
    int64_t w_off;   /* first element of the [rows,row_len] row-major weight, from w_base */
    int64_t g_off;   /* same tensor's gradient, from g_base (ignored unless mode TAYLOR)   */
    int64_t out_off; /* first element of its [rows] score vector, from score_base          */
    int64_t rows;    /* out_channels                                                       */
    int64_t row_len; /* in_channels/groups * kh * kw                                       */
} spice_rows_desc;

typedef struct spice_rows_item {
    int64_t w_off;
    int64_t g_off;
    int64_t out_off;
    int32_t row_len;
    int32_t n_rows;
    int32_t lanes; /* lanes cooperating on one row: power of two, 1..64 */
    int32_t flags; /* bit0: rows are 16-byte aligned and row_len % 4 == 0 */
} spice_rows_item;

#define SPICE_IMP_L1 0     /* sum |w|          */
#define SPICE_IMP_L2 1     /* sum w*w          */
#define SPICE_IMP_TAYLOR 2 /* sum |g| * |w|    */

/* Host only.  Writes at most `capacity` items to `items` (may be NULL to count)
 * and returns the number of items the layers need, or a negative SPICE_ERR_*. */
int64_t spice_importance_plan(const spice_rows_desc *layers, int32_t n_layers,
                              spice_rows_item *items, int64_t capacity);

/* score[row] = (accumulate ? score[row] : 0) + reduction(row) for every row of
 * every planned layer.  Deterministic: a fixed reduction tree per row. */
int spice_importance_run(const float *w_base, const float *g_base, float *score_base,
                         const spice_rows_item *d_items, int64_t n_items, int32_t mode,
                         int32_t accumulate, void *stream);

/* Per-INPUT-channel magnitude term of torch-pruning's MagnitudeImportance for a
 * consumer layer: weight.transpose(0,1).flatten(1).abs().pow(p).sum(1).
 * `inner` = kh*kw; the weight is [rows, cols*inner] row-major. */
typedef struct spice_cols_desc {
    int64_t w_off;
    int64_t out_off; /* [cols] result vector, from score_base */
    int64_t rows;    /* out_channels */
    int64_t cols;    /* in_channels / groups */
    int64_t inner;   /* kh*kw */
} spice_cols_desc;

typedef struct spice_cols_item {
    int64_t w_off;   /* first element of the slab's first row, at the chunk's first column */
    int64_t out_off; /* where this item's flat-column partial sums go, from partial_base    */
    int32_t row_stride;
    int32_t n_rows;  /* rows in this slab */
    int32_t n_flat;  /* flat columns (cols*inner granularity) in this chunk, <= 1024 */
    int32_t flags;   /* bit0: 16-byte aligned float4 path */
} spice_cols_item;

typedef struct spice_cols_fold {
    int64_t part_off; /* first partial row of this layer, from partial_base */
    int64_t out_off;  /* [cols] result, from score_base */
    int32_t n_slabs;
    int32_t n_flat;   /* cols*inner */
    int32_t inner;
    int32_t cols;
} spice_cols_fold;

/* Host only.  Returns the item count (stage 1); *n_folds receives the stage-2 record
 * count (= n_layers) and *partial_elems the float elements of scratch stage 1 writes. */
int64_t spice_importance_cols_plan(const spice_cols_desc *layers, int32_t n_layers,
                                   spice_cols_item *items, int64_t capacity,
                                   spice_cols_fold *folds, int64_t *partial_elems);

int spice_importance_cols_run(const float *w_base, float *score_base, float *partial_base,
                              const spice_cols_item *d_items, int64_t n_items,
                              const spice_cols_fold *d_folds, int64_t n_folds, int32_t mode,
                              void *stream);

/* ------------------------------------------------------------------------- *
 * (2) Fused KD-Lite loss: temperature-softmax KL + cross-entropy in one pass.
 *
 * Replaces KDLiteLoss.forward (pruning/losses.py:127-162) = SegmentationLoss
 * (:58-84) + PixelKDLoss (:22-55) + optional boundary mask (:109-119) and
 * teacher-confidence weights (:121-125); called from the fine-tune loop at
 * pruning/trainer.py:179.  Logits are fp32 NCHW contiguous, labels int64 NHW.
 * ------------------------------------------------------------------------- */
typedef struct spice_kdlite_params {
    float temperature;
    float alpha;
    float boundary_weight;     /* > 1 enables the 3x3 label-boundary term          */
    int32_t confidence_weight; /* non-zero enables the teacher max-prob weighting  */
    int64_t ignore_index;      /* 255 in the reference                             */
    const float *class_weight; /* device [C] or NULL (reference: 5 weights iff C==5) */
} spice_kdlite_params;

size_t spice_kdlite_workspace_bytes(int64_t n, int64_t c, int64_t h, int64_t w);

/* out3 (device, 3 floats) = {total, ce, kd}.  If `grad` is non-NULL it receives
 * coef_ce * d(ce)/d(student) + coef_kd * d(kd)/d(student), shape [N,C,H,W]; with
 * coef_ce = alpha and coef_kd = 1 - alpha that is d(total)/d(student).
 * If `upstream` (device, 3 floats = incoming gradients of total, ce, kd) is
 * non-NULL the coefficients become upstream[0]*alpha + upstream[1] and
 * upstream[0]*(1-alpha) + upstream[2] and coef_ce/coef_kd are ignored.
 * The workspace's first int64 is incremented by the number of labels that are neither
 * ignore_index nor in [0,C) (the reference raises for them); such pixels contribute
 * nothing to ce.  The caller zeroes the workspace once after allocating it (bytes [8,12)
 * hold an arrival counter that every call returns to zero) and must not share one workspace
 * between calls that may overlap in time (use one per stream).
 * Bytes [16,80) of the workspace receive the eight fp64 pixel sums of the call:
 * {sum valid*KL, n_valid, sum w*nll, sum w, sum boundary*KL, n_boundary,
 *  sum conf*KL, sum conf}. */
int spice_kdlite_forward(const float *student, const float *teacher, const int64_t *labels,
                         int64_t n, int64_t c, int64_t h, int64_t w,
                         const spice_kdlite_params *params, float *out3, float *grad,
                         float coef_ce, float coef_kd, const float *upstream,
                         void *workspace, size_t workspace_bytes, void *stream);

/* The label pass of spice_kdlite_forward's gradient form by itself: the loss normalisers (number of pixels that
 * are not ignored, sum of the class weights of the labelled ones) depend on the labels alone, so a training step can
 * take them on another stream while the student's and teacher's forwards run (pruning/trainer.py:172-183 computes the
 * loss behind both forwards; the mean reductions of pruning/losses.py:39-55,144-160 are what the pass serves).  It
 * writes them into `workspace`; spice_kdlite_forward_prepared is spice_kdlite_forward (same arguments, same results bit
 * for bit) without that launch, on the same workspace, ordered behind the pass by the caller.  Both return
 * SPICE_ERR_UNSUPPORTED for the boundary / confidence options (their pass reads the teacher's logits) and the prepared
 * form for grad == NULL (the forward-only form has no separate pass): call spice_kdlite_forward then. */
int spice_kdlite_label_pass(const int64_t *labels, int64_t n, int64_t c, int64_t h, int64_t w,
                            const spice_kdlite_params *params, void *workspace, size_t workspace_bytes, void *stream);
int spice_kdlite_forward_prepared(const float *student, const float *teacher, const int64_t *labels,
                                  int64_t n, int64_t c, int64_t h, int64_t w,
                                  const spice_kdlite_params *params, float *out3, float *grad,
                                  float coef_ce, float coef_kd, const float *upstream,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* Once per teacher of a sensitivity sweep: probs [N,C,H,W] fp32 = softmax(teacher / temperature) over C
 * and hneg [N,H,W] fp32 = sum_c probs ln probs, the teacher-only parts of PixelKDLoss
 * (pruning/losses.py:40-58), for spice_sweep_head_f16.  SPICE_ERR_UNSUPPORTED when H*W % 4 != 0 or a
 * pointer is off 16 bytes. */
int spice_kd_teacher_prepare(const float *teacher, int64_t n, int64_t c, int64_t h, int64_t w,
                             float temperature, float *probs, float *hneg, void *stream);

/* The tail of a sensitivity-sweep candidate in ONE launch: the segmentation head's bilinear
 * up-sampling (`nn.UpsamplingBilinear2d` at the end of the models of pruning/models.py:35-39), the
 * KD-Lite loss of the up-sampled logits against the teacher's (pruning/losses.py:127-162) and the
 * arg-max + confusion-matrix update (baseline/eval/inference.py:47-48, baseline/eval/metrics.py:8-26),
 * without the full-size student logits ever being written: `low` [N,C,in_h,in_w] fp16 is
 * interpolated on the fly with the arithmetic of spice_upsample_bilinear (same bits), so
 *   out3 = spice_kdlite_forward(up(low), teacher, labels) and conf += histogram(labels, argmax up(low)).
 * teacher [N,C,H,W] fp32, labels [N,H,W] int64, out3 3 floats, conf [C,C] int64 (+=).  The two
 * workspaces are those of spice_kdlite_forward and spice_confusion_update (same sizes, same
 * rules).  align_corners / scale_h / scale_w as spice_upsample_bilinear.
 * teacher_hneg: NULL, or the second output of spice_kd_teacher_prepare for this teacher and
 * temperature; `teacher` must then be its first output (the probabilities) instead of the logits.
 * A sweep evaluates every candidate against the same teacher, so the teacher's softmax is taken once
 * and the per-candidate pass does one exponential per logit less; same loss within fp32 rounding.
 * SPICE_ERR_UNSUPPORTED (nothing launched; use the three separate calls) for the boundary and
 * confidence options, W % 4 != 0, in_w % 8 != 0, pointers off 16 bytes, or source rows too long
 * for the LDS stage. */
int spice_sweep_head_f16(const void *low, const float *teacher, const float *teacher_hneg,
                         const int64_t *labels, int64_t n, int64_t c, int64_t in_h, int64_t in_w, int64_t h, int64_t w,
                         int32_t align_corners, float scale_h, float scale_w,
                         const spice_kdlite_params *params, float *out3, int64_t *conf,
                         void *kd_workspace, size_t kd_workspace_bytes, void *conf_workspace,
                         size_t conf_workspace_bytes, void *stream);

/* The same launch without teacher and loss, for evaluation loops: conf[label, argmax up(low)] += 1 with the
 * head's bilinear up-sampling done on the fly (baseline/eval/inference.py:45-48 and
 * pruning/eval/components/core_evaluator.py:141-143 for models that end in convolution ->
 * nn.UpsamplingBilinear2d): identical to spice_upsample_bilinear followed by
 * spice_argmax_confusion_update, without the full-size logits.  SPICE_ERR_UNSUPPORTED as above. */
int spice_upsample_argmax_confusion_f16(const void *low, const int64_t *labels, int64_t n, int64_t c,
                                        int64_t in_h, int64_t in_w, int64_t h, int64_t w,
                                        int32_t align_corners, float scale_h, float scale_w, int64_t *conf,
                                        void *conf_workspace, size_t conf_workspace_bytes, void *stream);

/* x[i] *= *scale for i < n, skipped entirely on the device when *scale == 1. */
int spice_scale_inplace(float *x, int64_t n, const float *scale, void *stream);

/* ------------------------------------------------------------------------- *
 * (3) mIoU confusion-matrix histogram.
 *
 * Replaces update_confusion_matrix (baseline/eval/metrics.py:8-19, duplicated at
 * baseline/train_test_quant.py:14-24) and the confusion part of
 * SegmentationMetrics.update (pruning/eval/metrics.py:47-70):
 *   conf[t, p] += 1 for every pixel with 0 <= t < C   (rows = target, cols = pred).
 * ------------------------------------------------------------------------- */
size_t spice_confusion_workspace_bytes(int64_t n_pixels, int32_t num_classes);

/* conf: device [C,C] int64, updated in place, by ONE kernel launch.  The workspace's
 * first int64 is incremented by the number of pixels with a valid target but a
 * prediction outside [0,C) (the reference fails on those); they are not counted in
 * conf.  The caller zeroes the first 4096 bytes of the workspace once after allocating
 * it (they hold that count and arrival counters which every call returns to zero) and
 * must not share one workspace between calls that may overlap in time.
 * At most 2^40 pixels per call. */
int spice_confusion_update(const int64_t *preds, const int64_t *target, int64_t n_pixels,
                           int32_t num_classes, int64_t *conf, void *workspace,
                           size_t workspace_bytes, void *stream);

/* The same update for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart
 * in the reference.  The table of per-workgroup totals that spice_confusion_update keeps
 * in the caller's workspace lives in a fixed 4 MiB + 4 KiB device array inside the
 * library (one copy per GPU, held until the process ends); the call allocates and frees
 * nothing.  Limits: num_classes <= 32 and n_pixels <= 2^40, SPICE_ERR_STATIC_LIMIT beyond
 * them.  `invalid_count` (device int64, may be NULL) is incremented by the number of
 * pixels with a valid target and a prediction outside [0,C).  The device the stream
 * belongs to must be the current device.  The first call on a device looks up that
 * device's copy of the array (hipGetSymbolAddress); make it before a graph capture
 * begins.  All calls on one GPU share the array: calls that may overlap in time on one
 * GPU (different streams, different threads) race and must be serialised by the caller.
 * Nothing needs clearing between calls: every call overwrites the rows it later reads
 * and returns the arrival counters to zero. */
int spice_confusion_update_static(const int64_t *preds, const int64_t *target, int64_t n_pixels,
                                  int32_t num_classes, int64_t *conf, int64_t *invalid_count,
                                  void *stream);

/* Same update with the arg-max fused in: preds = argmax over C of logits
 * [N,C,HW] (first maximum wins, as torch.argmax), never materialised.
 * Replaces baseline/eval/inference.py:47-48 and
 * pruning/eval/components/core_evaluator.py:141-143.
 * logits_dtype: 0 = fp32, 1 = fp16, 2 = bf16. */
int spice_argmax_confusion_update(const void *logits, int32_t logits_dtype,
                                  const int64_t *target, int64_t n, int64_t hw,
                                  int32_t num_classes, int64_t *conf, void *workspace,
                                  size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 * (4) Channel gather: physically slice pruned tensors.
 *
 * Replaces torch-pruning 1.6.0's prune_conv_out_channels / prune_conv_in_channels /
 * prune_batchnorm_out_channels (index_select + contiguous per tensor), executed
 * inside pruner.step() (pruning/pruner.py:495).  One launch covers every tensor:
 *   dst[o, i, k] = src[keep_out[o], keep_in[i], k]
 * with 1-D tensors (bias, BN weight/bias/mean/var) as the cols = inner = 1 case.
 * Elements are moved as 32-bit words, so the copy is bit-exact.
 * ------------------------------------------------------------------------- */
typedef struct spice_gather_desc {
    int64_t src_off;     /* from src_base, elements */
    int64_t dst_off;     /* from dst_base, elements */
    int64_t idx_out_off; /* offset into idx_base of the dst_rows kept row ids, or -1 = identity */
    int64_t idx_in_off;  /* offset into idx_base of the dst_cols kept col ids, or -1 = identity */
    int64_t dst_rows;
    int64_t dst_cols;
    int64_t src_cols;
    int64_t inner;
} spice_gather_desc;

typedef struct spice_gather_item {
    int64_t src_off;     /* tensor origin (row 0) */
    int64_t dst_off;     /* first destination row of this item */
    int64_t idx_out_off; /* already advanced to this item's first row, or -1 */
    int64_t idx_in_off;
    int32_t row0;        /* first destination row of this item */
    int32_t n_rows;
    int32_t dst_cols;
    int32_t src_cols;
    int32_t inner;
    int32_t lanes;       /* reserved, 0 */
    int32_t flags;       /* bit0: whole rows may move as 16-byte words; bit1: dst_off % 4 == 0;
                            bit2: in-channel list with rows staged in LDS */
    int32_t pad_;
} spice_gather_item;

int64_t spice_gather_plan(const spice_gather_desc *tensors, int32_t n_tensors,
                          spice_gather_item *items, int64_t capacity);

/* plan_flags: bitwise OR of the `flags` of the planned items (the caller has them on the host);
 * staged items (bit2) are only honoured when it says so, otherwise they take the element path. */
int spice_gather_run(const float *src_base, float *dst_base, const int64_t *idx_base,
                     const spice_gather_item *d_items, int64_t n_items, int32_t plan_flags,
                     void *stream);

/* A plan's work items and index lists (or any bytes) from page-locked host memory to the device as a LAUNCH on
 * `stream`: a kernel reads the host memory through its device mapping, so the call returns at once however much the
 * stream already holds (a copy command from page-locked memory keeps the caller until the stream has run that far).
 * The sweep makes the plans of the next candidates while the device works on the current ones (reference:
 * pruning/sensitivity.py builds every candidate's pruner inside its loop, :120-160).  dst: device memory; src_host:
 * page-locked, device-readable host memory that stays untouched until the launch has run; both 16-byte aligned,
 * nbytes a multiple of 16.  SPICE_ERR_ARG otherwise. */
int spice_gather_upload(void *dst, const void *src_host, int64_t nbytes, void *stream);

/* ------------------------------------------------------------------------- *
 * (5) Bilinear up-sampling of NCHW planes (the models' decoder and head).
 *
 * Replaces torch's upsample_bilinear2d inside the forward of the models the
 * reference builds at pruning/models.py:35-39 and baseline/eval/model_factory.py:26-30
 * (DeepLabV3+ decoder x4, segmentation heads x4, FPN blocks x2), i.e. inside every
 * forward of the sensitivity sweep and of the evaluation loops
 * (baseline/eval/inference.py:46, pruning/eval/components/core_evaluator.py:130-137).
 * src is [planes, in_h, in_w], dst [planes, out_h, out_w], both contiguous;
 * dtypes: 0 = fp32, 1 = fp16 (arithmetic is fp32 in every case, as torch does
 * under autocast).  scale_h / scale_w are the scale factors the caller gave
 * torch (used only without align_corners; 0 = derive from the sizes).
 * Up-sampling only (out >= in in both directions).
 * ------------------------------------------------------------------------- */
int spice_upsample_bilinear(const void *src, int32_t src_dtype, void *dst, int32_t dst_dtype,
                            int64_t planes, int32_t in_h, int32_t in_w, int32_t out_h,
                            int32_t out_w, int32_t align_corners, float scale_h, float scale_w,
                            void *stream);

/* The same into a channel slice of a larger tensor (the decoder's concatenation buffer, so that no separate
 * concatenation pass is needed): the destination's planes lie in groups of planes_per_group contiguous planes
 * (the channels of one image), the groups dst_group_stride elements apart.  planes must be a multiple of
 * planes_per_group. */
int spice_upsample_bilinear_into(const void *src, int32_t src_dtype, void *dst, int32_t dst_dtype,
                                 int64_t planes, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                                 int32_t align_corners, float scale_h, float scale_w, int64_t planes_per_group,
                                 int64_t dst_group_stride, void *stream);

/* ------------------------------------------------------------------------- *
 * (6) Inference batch-norm with the residual add and the ReLU behind it fused in.
 *
 * Replaces, inside the forward of the same models (every forward of the sweep and
 * of the evaluation loops), the launch triple batch_norm (eval mode) -> add -> relu
 * of a residual block, or the pair batch_norm -> relu:
 *   y = relu?( bn(x) [+ residual] ),  bn(x) = weight * (x - mean) * rsqrt(var + eps) + bias
 * x, residual (may be NULL) and y are [n, c, hw] contiguous, all of `dtype`
 * (0 = fp32, 1 = fp16); the four parameter vectors are fp32 [c].  Every
 * intermediate is rounded to `dtype` exactly where the separate kernels would
 * have stored it.  fp16 needs an even hw (SPICE_ERR_UNSUPPORTED otherwise).
 * y may alias x.
 * ------------------------------------------------------------------------- */
int spice_bn_act(const void *x, const void *residual, void *y, int32_t dtype, const float *mean,
                 const float *var, const float *weight, const float *bias, float eps, int64_t n,
                 int64_t c, int64_t hw, int32_t relu, void *stream);

/* The same for a residual block with a projection shortcut (the first bottleneck of every ResNet stage),
 * whose shortcut has a batch-norm of its own:
 *   y = relu?( bn(x) + bn_r(residual) )
 * replacing batch_norm(residual) -> batch_norm(x) -> add -> relu.  bn_r(residual) is rounded to `dtype` where
 * its own kernel would have stored it, so the values are those of the four separate launches.
 * The four r_* vectors are fp32 [c]; residual must not be NULL. */
int spice_bn_act_dual(const void *x, const void *residual, void *y, int32_t dtype, const float *mean,
                      const float *var, const float *weight, const float *bias, float eps, const float *r_mean,
                      const float *r_var, const float *r_weight, const float *r_bias, float r_eps, int64_t n,
                      int64_t c, int64_t hw, int32_t relu, void *stream);

/* The stem of a ResNet encoder behind its 7x7 convolution in one pass (pruning/models.py:35-39's ResNet encoders:
 * bn1 -> relu -> maxpool):
 *   full   = relu?( bn(x) )                          [n, c, h, w]      (NULL: not wanted, not written)
 *   pooled = max_pool2d(full, 3, stride 2, padding 1) [n, c, h/2, w/2]
 * fp16 NCHW; the batch-norm is spice_bn_act's arithmetic and the maximum keeps NaN as torch's kernel does, so both
 * outputs equal those of the two launches bit for bit.  h must be even and w a multiple of 16 (else
 * SPICE_ERR_UNSUPPORTED); neither output may be x. */
int spice_bn_act_maxpool_f16(const void *x, void *full, void *pooled, const float *mean, const float *var,
                             const float *weight, const float *bias, float eps, int64_t n, int64_t c, int32_t h,
                             int32_t w, int32_t relu, void *stream);

/* ------------------------------------------------------------------------- *
 * (7) Depthwise 3x3 convolution (stride 1, padding 1, no bias), fp16 NCHW.
 *
 * Replaces, inside the forward of the same models, torch's conv2d call for the
 * depthwise half of the DeepLabV3+ decoder's separable convolutions, which
 * MIOpen serves with its reference kernel.  x and y are [n, c, h, w] fp16
 * contiguous; weight is [c, 1, 3, 3], fp16 (weight_dtype 1) or fp32 (0: rounded
 * to fp16 first, as autocast's cast does).  The nine products (exact in fp32) are
 * added to an fp32 sum in tap order, rows first, with fused multiply-adds, and the
 * result is rounded to fp16 once: the arithmetic of torch's own depthwise kernel
 * (within one fp16 step of the exact sum MIOpen's reference kernel computes).
 * w must be a multiple of 8 and <= 2048.
 * ------------------------------------------------------------------------- */
int spice_depthwise_conv3x3(const void *x, const void *weight, int32_t weight_dtype, void *y,
                            int64_t n, int64_t c, int32_t h, int32_t w, void *stream);

/* The same with padding = dilation, dilation a multiple of 4 (the ASPP branches: 12 / 24 / 36).
 * torch serves these calls with its own depthwise kernel, which sums the nine products in fp32 in
 * tap order; this entry point does exactly that (bit-identical results). */
int spice_depthwise_conv3x3_dilated(const void *x, const void *weight, int32_t weight_dtype, void *y,
                                    int64_t n, int64_t c, int32_t h, int32_t w, int32_t dilation,
                                    void *stream);

/* Depthwise k x k convolution, k = 3 or 5, stride 1 or 2, dilation 1 or 2, padding (k / 2) * dilation, no bias: the
 * depthwise convolutions of the EfficientNet encoders (pruning/models.py:35-39 with "timm-efficientnet-b3"), which
 * MIOpen serves image by image with an unfold pass and a matrix product.  x [n, c, h, w] and y [n, c, oh, ow] fp16
 * contiguous, oh = (h - 1) / stride + 1, ow likewise; weight [c, 1, k, k] fp16 (1) or fp32 (0).  Same arithmetic as
 * spice_depthwise_conv3x3: the k * k products added to an fp32 sum in tap order, one rounding.  w must be even.
 * mean / var / bn_weight / bn_bias (fp32 [c], or all NULL) + eps + act (0 none, 1 ReLU, 2 SiLU): the inference
 * batch-norm and activation behind the convolution in the same launch, computed from the rounded result as
 * spice_bn_act would from the stored tensor.
 * SPICE_ERR_UNSUPPORTED for every other kernel size, stride, dilation or width. */
int spice_depthwise_conv_f16(const void *x, const void *weight, int32_t weight_dtype, void *y, int64_t n, int64_t c,
                             int32_t h, int32_t w, int32_t kernel, int32_t stride, int32_t dilation, const float *mean,
                             const float *var, const float *bn_weight, const float *bn_bias, float eps, int32_t act,
                             void *stream);

/* Squeeze-and-excite of the EfficientNet blocks, fp16 NCHW, inference: y = x * sigmoid(W2 silu(W1 mean(x) + b1) + b2),
 * mean over each plane; x, y [n, c, hw] fp16 contiguous, w1 [r, c], b1 [r], w2 [c, r], b2 [c] fp32 (rounded to fp16
 * where used, as autocast casts them).  Replaces torch's reduction, two 1x1 convolutions through MIOpen and four
 * element-wise launches with four short launches; every intermediate is rounded to fp16 where torch's modules would store
 * it.  workspace: spice_squeeze_excite_workspace_bytes(n, c) bytes, 256-byte aligned.  c <= 4096, r <= 1024. */
size_t spice_squeeze_excite_workspace_bytes(int64_t n, int64_t c);
int spice_squeeze_excite_f16(const void *x, void *y, int64_t n, int64_t c, int64_t hw, const float *w1, const float *b1,
                             int64_t r, const float *w2, const float *b2, void *workspace, size_t workspace_bytes,
                             void *stream);

/* The image-pooling branch of an ASPP (the DeepLabV3+ decoders of pruning/models.py:35-39): plane means of x
 * [n, cin, hw] fp16 -> 1x1 convolution with weight [cout, cin] fp16 -> inference batch-norm (fp32 [cout] vectors, or
 * all NULL) -> ReLU / SiLU (relu: 0 none, 1, 2) -> the result spread over the hw pixels of y [n, cout, hw] fp16, whose
 * images lie y_image_stride elements apart (0: dense; a channel slice of a concatenation otherwise).  Two launches
 * where torch's modules take a reduction, a convolution through MIOpen, a batch-norm, an activation and an
 * up-sampling; fp32 sums, every intermediate rounded to fp16 where those modules store it.
 * workspace: spice_pooled_branch_workspace_bytes(n, cin) bytes. */
size_t spice_pooled_branch_workspace_bytes(int64_t n, int64_t cin);
int spice_pooled_branch_f16(const void *x, const void *weight, void *y, int64_t n, int64_t cin, int64_t hw, int64_t cout,
                            int64_t y_image_stride, const float *mean, const float *var, const float *bn_weight,
                            const float *bn_bias, float eps, int32_t relu, void *workspace, size_t workspace_bytes,
                            void *stream);

/* ------------------------------------------------------------------------- *
 * (8) 1x1 stride-1 convolution of fp16 NCHW activations, without bias:
 *   y[n][Cout, HW] = weight[Cout, Cin] @ x[n][Cin, HW]   (fp32 accumulation).
 *
 * Replaces torch's conv2d call for the 1x1 convolutions of the same models
 * (two of the three convolutions of every ResNet bottleneck, the ASPP and decoder
 * projections) on the inference path.  A hipBLASLt matrix product whose
 * descriptors and algorithm are prepared once per (n, Cin, HW, Cout) and kept for
 * the life of the process; the first call on a device creates a hipBLASLt handle
 * (the one place where the library holds state of its own).  Only algorithms that
 * need no scratch memory are used.
 * ------------------------------------------------------------------------- */
int spice_conv1x1_f16(const void *x, const void *weight, void *y, int64_t n, int64_t cin, int64_t hw,
                      int64_t cout, void *stream);

/* The same product with the batch-norm that follows it in those models and the tail of a residual block
 * applied before the result leaves the compute unit (pruning/models.py:35-39's networks: conv1/bn1 and
 * conv3/bn3 of every bottleneck, the ASPP branches and projections):
 *   y = relu?( bn(weight @ x) [+ residual | + bn_r(residual)] )
 * ONE launch of a hand-written matrix-core kernel (v_mfma_f32_32x32x16_f16) instead of the product and a
 * batch-norm pass over it.  The product is rounded to fp16 where spice_conv1x1_f16 would have stored it and
 * the tail is spice_bn_act's / spice_bn_act_dual's arithmetic, so the values are those of the two calls up to
 * the order of the fp32 sums.  mean == NULL: the plain product (no tail; residual must be NULL).
 * residual is [n, Cout, HW] fp16 or NULL; the r_* vectors are all NULL or all set (then residual is required).
 * p_mean .. p_bias (all NULL or all set, fp32 [Cin]): x is the RAW result of the convolution before and its own
 * batch-norm (+ ReLU if p_relu) is still pending: the kernel applies it to x on the way in, rounded to fp16 as
 * spice_bn_act would have stored it, so that launch and its round trip through memory are gone (conv2 -> bn2 -> relu ->
 * conv3 -> bn3 of a bottleneck: one launch behind the 3x3 convolution).  Cin <= 512, else SPICE_ERR_UNSUPPORTED.
 * y_image_stride: elements between the images of y (0 = Cout * HW, dense); larger when y is a channel slice of a
 * concatenation buffer.  HW and y_image_stride must be multiples of 8, x, y, residual 16-byte aligned, weight
 * 4-byte aligned, and the weight matrix and one image of x below 2^30 - 2^20 bytes each (32-bit offsets inside
 * the kernel), else SPICE_ERR_UNSUPPORTED (callers then use the two calls above).  Holds no state and uses no scratch memory. */
int spice_conv1x1_bn_act_f16(const void *x, const void *weight, void *y, int64_t n, int64_t cin, int64_t hw,
                             int64_t cout, const float *mean, const float *var, const float *bn_weight,
                             const float *bn_bias, float eps, const void *residual, const float *r_mean,
                             const float *r_var, const float *r_weight, const float *r_bias, float r_eps,
                             int32_t relu, int64_t y_image_stride, const float *p_mean, const float *p_var,
                             const float *p_weight, const float *p_bias, float p_eps, int32_t p_relu, void *stream);

/* The plain product for a 1x1 convolution of stride 2 (the projection shortcuts of the ResNet stages that halve the
 * resolution): y[n][Cout, (h/2)*(w/2)] = weight[Cout, Cin] @ x[n][Cin, even rows, even columns], by the same
 * matrix-core kernel, which picks the even pixels while it stages x (no strided copy of x, no layout transposes).
 * h must be even and w a multiple of 16, else SPICE_ERR_UNSUPPORTED. */
int spice_conv1x1_stride2_f16(const void *x, const void *weight, void *y, int64_t n, int64_t cin, int32_t h,
                              int32_t w, int64_t cout, void *stream);

/* A 7x7 convolution of stride 2 and padding 3 without bias (a ResNet stem) of fp16 NCHW activations, by the same
 * matrix-core kernel: y[n][Cout, (h/2)*(w/2)] = weight[Cout, Cin*49] @ the 49 Cin shifted, column-halved views of x[n],
 * which the kernel picks from x while it stages it (two 16-byte loads, the even or the odd halfs, zeros for the
 * padding): no im2col buffer, no NHWC copies.  weight is [Cout][Cin][7][7] as torch keeps it.  h must be even, w a
 * multiple of 16 and Cin <= 1024, else SPICE_ERR_UNSUPPORTED.  Replaces nothing in the reference itself: it is the
 * convolution torch hands to MIOpen for `nn.Conv2d(3, 64, 7, 2, 3)` (smp's ResNet encoders). */
int spice_conv7x7_stride2_f16(const void *x, const void *weight, void *y, int64_t n, int64_t cin, int32_t h,
                              int32_t w, int64_t cout, void *stream);

/* A 3x3 convolution (padding = dilation, no bias) of fp16 NCHW activations with the inference batch-norm (+ ReLU)
 * behind it, in one launch of the same matrix-core kernel: the nine taps are nine shifted 1x1 products in one sum
 * (conv2 + bn2 + relu of the bottlenecks of pruning/models.py:35-39's ResNet encoders, which torch serves with
 * MIOpen's Winograd / implicit-GEMM kernels and a separate batch-norm launch).
 *   y[n][Cout, (h/stride)*(w/stride)] = relu?( bn( sum over taps of weight9[:, tap, :] @ x[n] shifted by the tap ) )
 * weight9 is the convolution's weight as [Cout][9][Cin] (tap-major: weight.permute(0, 2, 3, 1)), fp16.
 * stride 1 or 2, dilation 1..7; h and w multiples of the stride, w / stride a multiple of 8, w >= 8; else
 * SPICE_ERR_UNSUPPORTED (callers keep torch's convolution).  fp32 sums in (tap, channel) order; mean == NULL: no
 * batch-norm.  Layers of at most 32 output channels (stride 1, dilation 1, Cin a multiple of 16: the last decoder
 * blocks and the 3x3 head of the U-Net models) take a kernel of their own (tiles of 4 x 128 pixels of all channels,
 * the patch under a tile staged once per block of input channels).  Holds no state and uses no scratch memory. */
int spice_conv3x3_bn_act_f16(const void *x, const void *weight9, void *y, int64_t n, int64_t cin, int32_t h,
                             int32_t w, int64_t cout, int32_t stride, int32_t dilation, const float *mean,
                             const float *var, const float *bn_weight, const float *bn_bias, float eps,
                             int32_t relu, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPICE_AMD_H */
