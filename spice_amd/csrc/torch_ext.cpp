// torch.ops.spice.*: the thin PyTorch extension over the C ABI of libspice_amd.so.
//
// Every operator takes at::Tensor arguments, checks device / dtype / contiguity / shape with
// TORCH_CHECK, makes the tensors' device current, takes torch's current HIP stream of that device
// and calls the extern "C" launcher declared in include/spice_amd.h.  Nothing is computed here and
// nothing falls back to stock ATen kernels, with one stated exception (conv1x1_f16, see there).
// Host-only planning (spice_*_plan) is not wrapped: it touches no tensor and stays a plain C call.
//
// Python surfaces these operators stand under (reference file:line):
//   compute_taylor_scores          pruning/sensitivity_utils.py:15-73     -> importance_run
//   tp MagnitudeImportance         pruning/pruner.py:268-270              -> importance_run, importance_cols_run
//   pruner.step() slicing          pruning/pruner.py:495                  -> gather_run
//   KDLiteLoss.forward             pruning/losses.py:127-162              -> kdlite_forward, scale_inplace
//   update_confusion_matrix        baseline/eval/metrics.py:8-19          -> confusion_update
//   argmax + update                baseline/eval/inference.py:47-48       -> argmax_confusion_update
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <hip/hip_runtime_api.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <limits>
#include <tuple>
#include <vector>

#include "../../include/spice_amd.h"

namespace {

using at::Tensor;

void *current_stream(const Tensor &t) {
    return static_cast<void *>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

void check_rc(int rc, const char *what) {
    if (rc == SPICE_OK) return;
    TORCH_CHECK_VALUE(rc != SPICE_ERR_ARG, what, ": ", spice_error_string(rc));
    TORCH_CHECK(false, what, ": ", spice_error_string(rc), " (code ", rc, ")");
}

void check_dev(const Tensor &t, const char *name, at::ScalarType dtype) {
    TORCH_CHECK(t.is_cuda(), name, " is on ", t.device(),
                "; the MI355X path has no CPU implementation (move the tensor to a HIP device)");
    TORCH_CHECK_TYPE(t.scalar_type() == dtype, name, " must be ", dtype, ", got ", t.scalar_type());
    TORCH_CHECK_VALUE(t.is_contiguous(), name, " must be contiguous");
}

void check_same_device(const Tensor &a, const Tensor &b, const char *name) {
    TORCH_CHECK(a.device() == b.device(), name, " is on ", b.device(), ", expected ", a.device());
}

// Lowest data pointer of a list of tensors: work-item tables address every tensor of a model as an
// element offset from it.
const float *list_base(const std::vector<Tensor> &ts, const char *name, const Tensor &like) {
    TORCH_CHECK_VALUE(!ts.empty(), name, ": no tensors");
    uintptr_t base = std::numeric_limits<uintptr_t>::max();
    for (const Tensor &t : ts) {
        check_dev(t, name, at::kFloat);
        check_same_device(like, t, name);
        base = std::min(base, reinterpret_cast<uintptr_t>(t.data_ptr()));
    }
    return reinterpret_cast<const float *>(base);
}

void check_items(const Tensor &items, int64_t n_items, size_t item_bytes, const char *name, const Tensor &like) {
    check_dev(items, name, at::kByte);
    check_same_device(like, items, name);
    TORCH_CHECK_VALUE(n_items >= 0 && items.numel() >= (int64_t)(n_items * item_bytes), name, " holds ",
                      items.numel(), " bytes, fewer than ", n_items, " records of ", item_bytes, " bytes");
}

// ---- (1) importance -------------------------------------------------------------------------

void importance_run(std::vector<Tensor> weights, std::vector<Tensor> grads, Tensor scores, const Tensor &items,
                    int64_t n_items, int64_t mode, bool accumulate) {
    check_dev(scores, "scores", at::kFloat);
    const float *w_base = list_base(weights, "weights", scores);
    const float *g_base = nullptr;
    if (mode == SPICE_IMP_TAYLOR) {
        TORCH_CHECK_VALUE(grads.size() == weights.size(), "importance_run: Taylor mode needs one gradient per weight");
        g_base = list_base(grads, "grads", scores);
    }
    check_items(items, n_items, sizeof(spice_rows_item), "items", scores);
    c10::DeviceGuard guard(scores.device());
    check_rc(spice_importance_run(w_base, g_base, scores.data_ptr<float>(),
                                  reinterpret_cast<const spice_rows_item *>(items.data_ptr()), n_items, (int32_t)mode,
                                  accumulate ? 1 : 0, current_stream(scores)),
             "spice_importance_run");
}

void importance_cols_run(std::vector<Tensor> weights, Tensor scores, Tensor partial, const Tensor &items,
                         int64_t n_items, const Tensor &folds, int64_t n_folds, int64_t mode) {
    check_dev(scores, "scores", at::kFloat);
    check_dev(partial, "partial", at::kFloat);
    check_same_device(scores, partial, "partial");
    const float *w_base = list_base(weights, "weights", scores);
    check_items(items, n_items, sizeof(spice_cols_item), "items", scores);
    check_items(folds, n_folds, sizeof(spice_cols_fold), "folds", scores);
    c10::DeviceGuard guard(scores.device());
    check_rc(spice_importance_cols_run(w_base, scores.data_ptr<float>(), partial.data_ptr<float>(),
                                       reinterpret_cast<const spice_cols_item *>(items.data_ptr()), n_items,
                                       reinterpret_cast<const spice_cols_fold *>(folds.data_ptr()), n_folds,
                                       (int32_t)mode, current_stream(scores)),
             "spice_importance_cols_run");
}

// ---- (4) gather -----------------------------------------------------------------------------

void gather_run(std::vector<Tensor> srcs, Tensor dst, const Tensor &idx, const Tensor &items, int64_t n_items,
                int64_t plan_flags) {
    check_dev(dst, "dst", at::kFloat);
    check_dev(idx, "idx", at::kLong);
    check_same_device(dst, idx, "idx");
    const float *src_base = list_base(srcs, "srcs", dst);
    check_items(items, n_items, sizeof(spice_gather_item), "items", dst);
    c10::DeviceGuard guard(dst.device());
    check_rc(spice_gather_run(src_base, dst.data_ptr<float>(), idx.data_ptr<int64_t>(),
                              reinterpret_cast<const spice_gather_item *>(items.data_ptr()), n_items,
                              (int32_t)plan_flags, current_stream(dst)),
             "spice_gather_run");
}

void gather_upload(Tensor dst, const Tensor &src) {
    check_dev(dst, "dst", at::kByte);
    TORCH_CHECK_VALUE(src.device().is_cpu() && src.scalar_type() == at::kByte && src.is_contiguous() && src.is_pinned(),
                      "src must be a contiguous uint8 tensor in page-locked host memory");
    TORCH_CHECK_VALUE(dst.numel() == src.numel() && dst.numel() % 16 == 0, "dst and src must hold the same multiple of 16 bytes");
    c10::DeviceGuard guard(dst.device());
    check_rc(spice_gather_upload(dst.data_ptr(), src.data_ptr(), dst.numel(), current_stream(dst)), "spice_gather_upload");
}

// ---- (3) confusion --------------------------------------------------------------------------

void check_conf(const Tensor &conf, int64_t num_classes) {
    check_dev(conf, "conf", at::kLong);
    TORCH_CHECK_VALUE(num_classes >= 1 && num_classes <= std::numeric_limits<int32_t>::max(), "num_classes must be >= 1");
    TORCH_CHECK_VALUE(conf.dim() == 2 && conf.size(0) == num_classes && conf.size(1) == num_classes, "conf must be [",
                      num_classes, ",", num_classes, "], got ", conf.sizes());
}

void confusion_update(Tensor conf, const Tensor &preds, const Tensor &target, int64_t num_classes, Tensor workspace) {
    check_conf(conf, num_classes);
    check_dev(preds, "preds", at::kLong);
    check_dev(target, "target", at::kLong);
    check_dev(workspace, "workspace", at::kByte);
    check_same_device(conf, preds, "preds");
    check_same_device(conf, target, "target");
    check_same_device(conf, workspace, "workspace");
    TORCH_CHECK_VALUE(preds.numel() == target.numel(), "preds and target differ in element count");
    c10::DeviceGuard guard(conf.device());
    check_rc(spice_confusion_update(preds.data_ptr<int64_t>(), target.data_ptr<int64_t>(), preds.numel(),
                                    (int32_t)num_classes, conf.data_ptr<int64_t>(), workspace.data_ptr(),
                                    (size_t)workspace.numel(), current_stream(conf)),
             "spice_confusion_update");
}

void confusion_update_static(Tensor conf, const Tensor &preds, const Tensor &target, int64_t num_classes,
                             c10::optional<Tensor> invalid_count) {
    check_conf(conf, num_classes);
    check_dev(preds, "preds", at::kLong);
    check_dev(target, "target", at::kLong);
    check_same_device(conf, preds, "preds");
    check_same_device(conf, target, "target");
    TORCH_CHECK_VALUE(preds.numel() == target.numel(), "preds and target differ in element count");
    int64_t *bad = nullptr;
    if (invalid_count.has_value()) {
        check_dev(*invalid_count, "invalid_count", at::kLong);
        check_same_device(conf, *invalid_count, "invalid_count");
        TORCH_CHECK_VALUE(invalid_count->numel() >= 1, "invalid_count must hold one int64");
        bad = invalid_count->data_ptr<int64_t>();
    }
    c10::DeviceGuard guard(conf.device());
    void *stream = current_stream(conf);
    {
        // Calls on one GPU share that GPU's table and tickets, so they must not overlap.  A caller that moves from one
        // stream to another gets the previous stream drained first (a host wait, on the change of stream only): a
        // mistake costs time, not a wrong matrix.  Streams under graph capture cannot be waited for and are left to
        // the caller, as the header says.
        static std::mutex mu;
        static void *last_stream[64];
        static bool used[64];
        const int d = conf.device().index();
        if (d >= 0 && d < 64) {
            std::lock_guard<std::mutex> lock(mu);
            if (used[d] && last_stream[d] != stream) {
                hipStreamCaptureStatus was = hipStreamCaptureStatusNone, now = hipStreamCaptureStatusNone;
                const bool known = hipStreamIsCapturing(static_cast<hipStream_t>(last_stream[d]), &was) == hipSuccess &&
                                   hipStreamIsCapturing(static_cast<hipStream_t>(stream), &now) == hipSuccess;
                if (known && was == hipStreamCaptureStatusNone && now == hipStreamCaptureStatusNone)
                    (void)hipStreamSynchronize(static_cast<hipStream_t>(last_stream[d]));
                else
                    (void)hipGetLastError();   // (a stream destroyed since: nothing of it can still be running)
            }
            used[d] = true;
            last_stream[d] = stream;
        }
    }
    check_rc(spice_confusion_update_static(preds.data_ptr<int64_t>(), target.data_ptr<int64_t>(), preds.numel(),
                                           (int32_t)num_classes, conf.data_ptr<int64_t>(), bad, stream),
             "spice_confusion_update_static");
}

int32_t logits_code(const Tensor &t, const char *name) {
    switch (t.scalar_type()) {
        case at::kFloat: return 0;
        case at::kHalf: return 1;
        case at::kBFloat16: return 2;
        default: TORCH_CHECK_TYPE(false, name, " dtype ", t.scalar_type(), " not supported");
    }
    return -1;
}

void argmax_confusion_update(Tensor conf, const Tensor &logits, const Tensor &target, int64_t num_classes,
                             Tensor workspace) {
    check_conf(conf, num_classes);
    const int32_t code = logits_code(logits, "logits");
    check_dev(logits, "logits", logits.scalar_type());
    check_dev(target, "target", at::kLong);
    check_dev(workspace, "workspace", at::kByte);
    check_same_device(conf, logits, "logits");
    check_same_device(conf, target, "target");
    check_same_device(conf, workspace, "workspace");
    TORCH_CHECK_VALUE(logits.dim() >= 2 && logits.size(1) == num_classes, "logits must be [N, num_classes, ...]");
    const int64_t n = logits.size(0);
    const int64_t hw = (n * num_classes) ? logits.numel() / (n * num_classes) : 0;
    TORCH_CHECK_VALUE(target.numel() == n * hw, "target must have N*H*W elements");
    c10::DeviceGuard guard(conf.device());
    check_rc(spice_argmax_confusion_update(logits.data_ptr(), code, target.data_ptr<int64_t>(), n, hw,
                                           (int32_t)num_classes, conf.data_ptr<int64_t>(), workspace.data_ptr(),
                                           (size_t)workspace.numel(), current_stream(conf)),
             "spice_argmax_confusion_update");
}

// ---- (2) KD-Lite ----------------------------------------------------------------------------

std::tuple<Tensor, Tensor> kdlite_forward(const Tensor &student, const Tensor &teacher, const Tensor &labels,
                                          double temperature, double alpha, int64_t ignore_index,
                                          double boundary_weight, bool confidence_weight,
                                          const c10::optional<Tensor> &class_weight, bool want_grad, double coef_ce,
                                          double coef_kd, const c10::optional<Tensor> &upstream, Tensor workspace,
                                          bool normalisers_ready) {
    check_dev(student, "student_logits", at::kFloat);
    check_dev(teacher, "teacher_logits", at::kFloat);
    check_dev(labels, "labels", at::kLong);
    check_dev(workspace, "workspace", at::kByte);
    check_same_device(student, teacher, "teacher_logits");
    check_same_device(student, labels, "labels");
    check_same_device(student, workspace, "workspace");
    TORCH_CHECK_VALUE(student.dim() == 4 && teacher.sizes() == student.sizes(),
                      "student/teacher logits must both be [N,C,H,W]");
    const int64_t n = student.size(0), c = student.size(1), h = student.size(2), w = student.size(3);
    TORCH_CHECK_VALUE(labels.dim() == 3 && labels.size(0) == n && labels.size(1) == h && labels.size(2) == w,
                      "labels must be [", n, ",", h, ",", w, "], got ", labels.sizes());
    spice_kdlite_params p;
    p.temperature = (float)temperature;
    p.alpha = (float)alpha;
    p.boundary_weight = (float)boundary_weight;
    p.confidence_weight = confidence_weight ? 1 : 0;
    p.ignore_index = ignore_index;
    p.class_weight = nullptr;
    if (class_weight.has_value()) {
        check_dev(*class_weight, "class_weight", at::kFloat);
        check_same_device(student, *class_weight, "class_weight");
        TORCH_CHECK_VALUE(class_weight->numel() == c, "class_weight must have C elements");
        p.class_weight = class_weight->data_ptr<float>();
    }
    const float *up = nullptr;
    if (upstream.has_value()) {
        check_dev(*upstream, "upstream", at::kFloat);
        check_same_device(student, *upstream, "upstream");
        TORCH_CHECK_VALUE(upstream->numel() == 3, "upstream must have 3 elements");
        up = upstream->data_ptr<float>();
    }
    c10::DeviceGuard guard(student.device());
    Tensor out3 = at::empty({3}, student.options());
    Tensor grad = want_grad ? at::empty_like(student) : at::empty({0}, student.options());
    const auto entry = normalisers_ready ? spice_kdlite_forward_prepared : spice_kdlite_forward;
    check_rc(entry(student.data_ptr<float>(), teacher.data_ptr<float>(), labels.data_ptr<int64_t>(), n, c, h, w, &p,
                   out3.data_ptr<float>(), want_grad ? grad.data_ptr<float>() : nullptr, (float)coef_ce, (float)coef_kd, up,
                   workspace.data_ptr(), (size_t)workspace.numel(), current_stream(student)),
             normalisers_ready ? "spice_kdlite_forward_prepared" : "spice_kdlite_forward");
    return std::make_tuple(out3, grad);
}

// The loss normalisers of the gradient form from the labels alone, into `workspace` (for kdlite_forward(...,
// normalisers_ready = true) on the same workspace).  False when the options have no separate label pass.
bool kdlite_label_pass(const Tensor &labels, int64_t num_classes, int64_t ignore_index, double boundary_weight,
                       bool confidence_weight, const c10::optional<Tensor> &class_weight, Tensor workspace) {
    check_dev(labels, "labels", at::kLong);
    check_dev(workspace, "workspace", at::kByte);
    check_same_device(labels, workspace, "workspace");
    TORCH_CHECK_VALUE(labels.dim() == 3, "labels must be [N,H,W], got ", labels.sizes());
    spice_kdlite_params p;
    p.temperature = 1.0f;
    p.alpha = 0.5f;
    p.boundary_weight = (float)boundary_weight;
    p.confidence_weight = confidence_weight ? 1 : 0;
    p.ignore_index = ignore_index;
    p.class_weight = nullptr;
    if (class_weight.has_value()) {
        check_dev(*class_weight, "class_weight", at::kFloat);
        check_same_device(labels, *class_weight, "class_weight");
        TORCH_CHECK_VALUE(class_weight->numel() == num_classes, "class_weight must have C elements");
        p.class_weight = class_weight->data_ptr<float>();
    }
    c10::DeviceGuard guard(labels.device());
    const int rc = spice_kdlite_label_pass(labels.data_ptr<int64_t>(), labels.size(0), num_classes, labels.size(1),
                                           labels.size(2), &p, workspace.data_ptr(), (size_t)workspace.numel(),
                                           current_stream(labels));
    if (rc == SPICE_ERR_UNSUPPORTED) return false;
    check_rc(rc, "spice_kdlite_label_pass");
    return true;
}

// Up-sampling + loss + arg-max histogram of a sweep candidate in one launch.  Returns out3, or a tensor
// without elements when the library does not cover the case (nothing was launched; the caller makes the
// three separate calls).
// The teacher-only parts of the distillation term, once per teacher: (softmax(teacher / T) over C, per-pixel
// sum p ln p).  Tensors without elements when the library does not cover the layout.
std::tuple<Tensor, Tensor> kd_teacher_prepare(const Tensor &teacher, double temperature) {
    check_dev(teacher, "teacher_logits", at::kFloat);
    TORCH_CHECK_VALUE(teacher.dim() == 4, "teacher_logits must be [N,C,H,W]");
    c10::DeviceGuard guard(teacher.device());
    const int64_t n = teacher.size(0), c = teacher.size(1), h = teacher.size(2), w = teacher.size(3);
    Tensor probs = at::empty_like(teacher);
    Tensor hneg = at::empty({n, h, w}, teacher.options());
    const int rc = spice_kd_teacher_prepare(teacher.data_ptr<float>(), n, c, h, w, (float)temperature,
                                            probs.data_ptr<float>(), hneg.data_ptr<float>(), current_stream(teacher));
    if (rc == SPICE_ERR_UNSUPPORTED) return std::make_tuple(at::empty({0}, teacher.options()), at::empty({0}, teacher.options()));
    check_rc(rc, "spice_kd_teacher_prepare");
    return std::make_tuple(probs, hneg);
}

Tensor sweep_head(const Tensor &low, const Tensor &teacher, const c10::optional<Tensor> &teacher_hneg,
                  const Tensor &labels, Tensor conf, bool align_corners,
                  double scale_h, double scale_w, double temperature, double alpha, int64_t ignore_index,
                  const c10::optional<Tensor> &class_weight, Tensor kd_workspace, Tensor conf_workspace,
                  const c10::optional<Tensor> &out) {
    check_dev(low, "low_logits", at::kHalf);
    check_dev(teacher, "teacher_logits", at::kFloat);
    check_dev(labels, "labels", at::kLong);
    check_dev(kd_workspace, "kd_workspace", at::kByte);
    check_dev(conf_workspace, "conf_workspace", at::kByte);
    check_same_device(low, teacher, "teacher_logits");
    check_same_device(low, labels, "labels");
    check_same_device(low, conf, "conf");
    check_same_device(low, kd_workspace, "kd_workspace");
    check_same_device(low, conf_workspace, "conf_workspace");
    TORCH_CHECK_VALUE(low.dim() == 4 && teacher.dim() == 4 && teacher.size(0) == low.size(0) && teacher.size(1) == low.size(1),
                      "low_logits [N,C,h,w] and teacher_logits [N,C,H,W] must agree in N and C");
    const int64_t n = low.size(0), c = low.size(1), h = teacher.size(2), w = teacher.size(3);
    check_conf(conf, c);
    TORCH_CHECK_VALUE(labels.dim() == 3 && labels.size(0) == n && labels.size(1) == h && labels.size(2) == w,
                      "labels must be [", n, ",", h, ",", w, "], got ", labels.sizes());
    const float *hneg = nullptr;
    if (teacher_hneg.has_value()) {   // `teacher` then holds the prepared probabilities
        check_dev(*teacher_hneg, "teacher_hneg", at::kFloat);
        check_same_device(low, *teacher_hneg, "teacher_hneg");
        TORCH_CHECK_VALUE(teacher_hneg->numel() == n * h * w, "teacher_hneg must have N*H*W elements");
        hneg = teacher_hneg->data_ptr<float>();
    }
    spice_kdlite_params p;
    p.temperature = (float)temperature;
    p.alpha = (float)alpha;
    p.boundary_weight = 1.0f;
    p.confidence_weight = 0;
    p.ignore_index = ignore_index;
    p.class_weight = nullptr;
    if (class_weight.has_value()) {
        check_dev(*class_weight, "class_weight", at::kFloat);
        check_same_device(low, *class_weight, "class_weight");
        TORCH_CHECK_VALUE(class_weight->numel() == c, "class_weight must have C elements");
        p.class_weight = class_weight->data_ptr<float>();
    }
    c10::DeviceGuard guard(low.device());
    Tensor out3;
    if (out.has_value()) {   // the three floats written where the caller keeps them
        check_dev(*out, "out", at::kFloat);
        check_same_device(low, *out, "out");
        TORCH_CHECK_VALUE(out->numel() == 3, "out must hold three floats");
        out3 = *out;
    } else {
        out3 = at::empty({3}, teacher.options());
    }
    const int rc = spice_sweep_head_f16(low.data_ptr(), teacher.data_ptr<float>(), hneg, labels.data_ptr<int64_t>(), n, c,
                                        low.size(2), low.size(3), h, w, align_corners ? 1 : 0, (float)scale_h,
                                        (float)scale_w, &p, out3.data_ptr<float>(), conf.data_ptr<int64_t>(),
                                        kd_workspace.data_ptr(), (size_t)kd_workspace.numel(), conf_workspace.data_ptr(),
                                        (size_t)conf_workspace.numel(), current_stream(low));
    if (rc == SPICE_ERR_UNSUPPORTED) return at::empty({0}, teacher.options());
    check_rc(rc, "spice_sweep_head_f16");
    return out3;
}

// Arg-max histogram of bilinearly up-sampled fp16 logits in one launch (evaluation loops).  False when the library
// does not cover the case (nothing was launched).
bool upsample_argmax_confusion(Tensor conf, const Tensor &low, const Tensor &labels, bool align_corners, double scale_h,
                               double scale_w, Tensor workspace) {
    check_dev(low, "low_logits", at::kHalf);
    check_dev(labels, "labels", at::kLong);
    check_dev(workspace, "workspace", at::kByte);
    check_same_device(low, labels, "labels");
    check_same_device(low, conf, "conf");
    check_same_device(low, workspace, "workspace");
    TORCH_CHECK_VALUE(low.dim() == 4 && labels.dim() == 3 && labels.size(0) == low.size(0),
                      "low_logits must be [N,C,h,w] and labels [N,H,W]");
    check_conf(conf, low.size(1));
    c10::DeviceGuard guard(low.device());
    const int rc = spice_upsample_argmax_confusion_f16(low.data_ptr(), labels.data_ptr<int64_t>(), low.size(0), low.size(1),
                                                       low.size(2), low.size(3), labels.size(1), labels.size(2),
                                                       align_corners ? 1 : 0, (float)scale_h, (float)scale_w,
                                                       conf.data_ptr<int64_t>(), workspace.data_ptr(),
                                                       (size_t)workspace.numel(), current_stream(low));
    if (rc == SPICE_ERR_UNSUPPORTED) return false;
    check_rc(rc, "spice_upsample_argmax_confusion_f16");
    return true;
}

void scale_inplace(Tensor x, const Tensor &scale) {
    check_dev(x, "x", at::kFloat);
    check_dev(scale, "scale", at::kFloat);
    check_same_device(x, scale, "scale");
    TORCH_CHECK_VALUE(scale.numel() >= 1, "scale must hold one float");
    c10::DeviceGuard guard(x.device());
    check_rc(spice_scale_inplace(x.data_ptr<float>(), x.numel(), scale.data_ptr<float>(), current_stream(x)),
             "spice_scale_inplace");
}

// ---- kernels inside the candidate's forward ----------------------------------------------------

int32_t act_code(const Tensor &t, const char *name) {
    if (t.scalar_type() == at::kFloat) return 0;
    TORCH_CHECK_TYPE(t.scalar_type() == at::kHalf, name, ": dtype ", t.scalar_type(), " not supported");
    return 1;
}

// `out`, where a kernel of the forward accepts one: a channel slice of a larger NCHW tensor (a concatenation
// buffer): [n, c, h, w] whose images are whole contiguous blocks, any distance apart.
void check_slice_out(const Tensor &out, const Tensor &like, int64_t n, int64_t c, int64_t h, int64_t w,
                     at::ScalarType dtype, const char *name) {
    TORCH_CHECK(out.device() == like.device(), name, ": out is on ", out.device(), ", expected ", like.device());
    TORCH_CHECK_TYPE(out.scalar_type() == dtype, name, ": out must be ", dtype, ", got ", out.scalar_type());
    TORCH_CHECK_VALUE(out.dim() == 4 && out.size(0) == n && out.size(1) == c && out.size(2) == h && out.size(3) == w,
                      name, ": out must be [", n, ", ", c, ", ", h, ", ", w, "]");
    TORCH_CHECK_VALUE(out.numel() == 0 || (out.stride(3) == 1 && out.stride(2) == w && out.stride(1) == h * w &&
                                           (n == 1 || out.stride(0) >= c * h * w)),
                      name, ": the images of out must be contiguous blocks (a channel slice of an NCHW tensor)");
}

Tensor upsample_bilinear(const Tensor &x, int64_t out_h, int64_t out_w, bool align_corners, double scale_h,
                         double scale_w, int64_t out_dtype, const c10::optional<Tensor> &out_opt) {
    const int32_t in_code = act_code(x, "upsample_bilinear");
    check_dev(x, "x", x.scalar_type());
    TORCH_CHECK_VALUE(x.dim() == 4, "upsample_bilinear: x must be [N, C, H, W]");
    TORCH_CHECK_TYPE(out_dtype == 0 || out_dtype == 1, "upsample_bilinear: output dtype code ", out_dtype,
                     " not supported");
    const at::ScalarType dt = out_dtype == 0 ? at::kFloat : at::kHalf;
    c10::DeviceGuard guard(x.device());
    Tensor out;
    if (out_opt.has_value()) {
        out = *out_opt;
        check_slice_out(out, x, x.size(0), x.size(1), out_h, out_w, dt, "upsample_bilinear");
    } else {
        out = at::empty({x.size(0), x.size(1), out_h, out_w}, x.options().dtype(dt));
    }
    if (x.numel() == 0) return out;
    const int64_t group_stride = x.size(0) > 1 ? out.stride(0) : 0;   // the images of out: dense or a slice's
    check_rc(spice_upsample_bilinear_into(x.data_ptr(), in_code, out.data_ptr(), (int32_t)out_dtype, x.size(0) * x.size(1),
                                          (int32_t)x.size(2), (int32_t)x.size(3), (int32_t)out_h, (int32_t)out_w,
                                          align_corners ? 1 : 0, (float)scale_h, (float)scale_w, x.size(1), group_stride,
                                          current_stream(x)),
             "spice_upsample_bilinear_into");
    return out;
}

Tensor bn_act(const Tensor &x, const Tensor &mean, const Tensor &var, const Tensor &weight, const Tensor &bias,
              double eps, const c10::optional<Tensor> &residual, bool relu, int64_t act) {
    const int32_t tail = act == 2 ? 2 : (relu || act == 1 ? 1 : 0);   // 0 none, 1 ReLU, 2 SiLU
    const int32_t code = act_code(x, "bn_act");
    TORCH_CHECK(x.is_cuda(), "x is on ", x.device(),
                "; the MI355X path has no CPU implementation (move the tensor to a HIP device)");
    TORCH_CHECK_VALUE(x.dim() == 4 && x.is_contiguous(), "bn_act: x must be a contiguous [N, C, H, W] tensor");
    const int64_t c = x.size(1);
    for (const Tensor *t : {&mean, &var, &weight, &bias})
        TORCH_CHECK_TYPE(t->scalar_type() == at::kFloat && t->device() == x.device() && t->numel() == c &&
                             t->is_contiguous(),
                         "bn_act: the four parameter vectors must be contiguous fp32 [", c, "] on ", x.device());
    const void *res = nullptr;
    if (residual.has_value()) {
        TORCH_CHECK_VALUE(residual->sizes() == x.sizes(), "bn_act: residual must have x's shape");
        TORCH_CHECK_TYPE(residual->scalar_type() == x.scalar_type() && residual->device() == x.device() &&
                             residual->is_contiguous(),
                         "bn_act: residual must be contiguous, of x's dtype and on x's device");
        res = residual->data_ptr();
    }
    c10::DeviceGuard guard(x.device());
    Tensor out = at::empty_like(x);
    check_rc(spice_bn_act(x.data_ptr(), res, out.data_ptr(), code, mean.data_ptr<float>(), var.data_ptr<float>(),
                          weight.data_ptr<float>(), bias.data_ptr<float>(), (float)eps, x.size(0), c,
                          x.size(2) * x.size(3), tail, current_stream(x)),
             "spice_bn_act");
    return out;
}

Tensor bn_act_dual(const Tensor &x, const Tensor &mean, const Tensor &var, const Tensor &weight, const Tensor &bias,
                   double eps, const Tensor &residual, const Tensor &r_mean, const Tensor &r_var, const Tensor &r_weight,
                   const Tensor &r_bias, double r_eps, bool relu) {
    const int32_t code = act_code(x, "bn_act_dual");
    TORCH_CHECK(x.is_cuda(), "x is on ", x.device(),
                "; the MI355X path has no CPU implementation (move the tensor to a HIP device)");
    TORCH_CHECK_VALUE(x.dim() == 4 && x.is_contiguous(), "bn_act_dual: x must be a contiguous [N, C, H, W] tensor");
    const int64_t c = x.size(1);
    for (const Tensor *t : {&mean, &var, &weight, &bias, &r_mean, &r_var, &r_weight, &r_bias})
        TORCH_CHECK_TYPE(t->scalar_type() == at::kFloat && t->device() == x.device() && t->numel() == c &&
                             t->is_contiguous(),
                         "bn_act_dual: the eight parameter vectors must be contiguous fp32 [", c, "] on ", x.device());
    TORCH_CHECK_VALUE(residual.sizes() == x.sizes(), "bn_act_dual: residual must have x's shape");
    TORCH_CHECK_TYPE(residual.scalar_type() == x.scalar_type() && residual.device() == x.device() &&
                         residual.is_contiguous(),
                     "bn_act_dual: residual must be contiguous, of x's dtype and on x's device");
    c10::DeviceGuard guard(x.device());
    Tensor out = at::empty_like(x);
    check_rc(spice_bn_act_dual(x.data_ptr(), residual.data_ptr(), out.data_ptr(), code, mean.data_ptr<float>(),
                               var.data_ptr<float>(), weight.data_ptr<float>(), bias.data_ptr<float>(), (float)eps,
                               r_mean.data_ptr<float>(), r_var.data_ptr<float>(), r_weight.data_ptr<float>(),
                               r_bias.data_ptr<float>(), (float)r_eps, x.size(0), c, x.size(2) * x.size(3), relu ? 1 : 0,
                               current_stream(x)),
             "spice_bn_act_dual");
    return out;
}

// (full, pooled) = (relu?(bn(x)), max_pool2d(that, 3, 2, 1)); `full` is an empty tensor when want_full is false
std::tuple<Tensor, Tensor> bn_act_maxpool(const Tensor &x, const Tensor &mean, const Tensor &var, const Tensor &weight,
                                          const Tensor &bias, double eps, bool relu, bool want_full) {
    check_dev(x, "x", at::kHalf);
    TORCH_CHECK_VALUE(x.dim() == 4 && x.is_contiguous(), "bn_act_maxpool: x must be a contiguous [N, C, H, W] tensor");
    const int64_t n = x.size(0), c = x.size(1), h = x.size(2), w = x.size(3);
    for (const Tensor *t : {&mean, &var, &weight, &bias})
        TORCH_CHECK_TYPE(t->scalar_type() == at::kFloat && t->device() == x.device() && t->numel() == c &&
                             t->is_contiguous(),
                         "bn_act_maxpool: the four parameter vectors must be contiguous fp32 [", c, "] on ", x.device());
    TORCH_CHECK_VALUE(h <= std::numeric_limits<int32_t>::max() && w <= std::numeric_limits<int32_t>::max(),
                      "bn_act_maxpool: plane too large");
    c10::DeviceGuard guard(x.device());
    Tensor full = want_full ? at::empty_like(x) : at::empty({0}, x.options());
    Tensor pooled = at::empty({n, c, h / 2, w / 2}, x.options());
    check_rc(spice_bn_act_maxpool_f16(x.data_ptr(), want_full ? full.data_ptr() : nullptr, pooled.data_ptr(),
                                      mean.data_ptr<float>(), var.data_ptr<float>(), weight.data_ptr<float>(),
                                      bias.data_ptr<float>(), (float)eps, n, c, (int32_t)h, (int32_t)w, relu ? 1 : 0,
                                      current_stream(x)),
             "spice_bn_act_maxpool_f16");
    return std::make_tuple(full, pooled);
}

Tensor depthwise_conv(const Tensor &x, const Tensor &weight, int64_t stride, int64_t dilation,
                      const c10::optional<Tensor> &mean, const c10::optional<Tensor> &var,
                      const c10::optional<Tensor> &bn_weight, const c10::optional<Tensor> &bn_bias, double eps, int64_t act) {
    check_dev(x, "x", at::kHalf);
    TORCH_CHECK_TYPE(weight.scalar_type() == at::kHalf || weight.scalar_type() == at::kFloat,
                     "depthwise_conv: weight dtype ", weight.scalar_type(), " not supported");
    check_dev(weight, "weight", weight.scalar_type());
    check_same_device(x, weight, "weight");
    TORCH_CHECK_VALUE(x.dim() == 4 && weight.dim() == 4 && weight.size(0) == x.size(1) && weight.size(1) == 1 &&
                          weight.size(2) == weight.size(3),
                      "depthwise_conv: x must be [N,C,H,W] and weight [C,1,k,k]");
    TORCH_CHECK_VALUE(stride >= 1 && stride <= 2 && x.size(2) >= 1 && x.size(3) >= 1, "depthwise_conv: stride must be 1 or 2");
    c10::DeviceGuard guard(x.device());
    const int64_t oh = (x.size(2) - 1) / stride + 1, ow = (x.size(3) - 1) / stride + 1;
    Tensor out = at::empty({x.size(0), x.size(1), oh, ow}, x.options());
    const int32_t wcode = weight.scalar_type() == at::kFloat ? 0 : 1;
    const float *bn[4] = {nullptr, nullptr, nullptr, nullptr};
    if (mean.has_value()) {
        TORCH_CHECK_VALUE(var.has_value() && bn_weight.has_value() && bn_bias.has_value(),
                          "depthwise_conv: the batch-norm needs all four vectors");
        const Tensor *parts[4] = {&*mean, &*var, &*bn_weight, &*bn_bias};
        for (int k = 0; k < 4; ++k) {
            TORCH_CHECK_TYPE(parts[k]->scalar_type() == at::kFloat && parts[k]->device() == x.device() &&
                                 parts[k]->numel() == x.size(1) && parts[k]->is_contiguous(),
                             "depthwise_conv: the batch-norm vectors must be contiguous fp32 [C] on x's device");
            bn[k] = parts[k]->data_ptr<float>();
        }
    }
    check_rc(spice_depthwise_conv_f16(x.data_ptr(), weight.data_ptr(), wcode, out.data_ptr(), x.size(0), x.size(1),
                                      (int32_t)x.size(2), (int32_t)x.size(3), (int32_t)weight.size(2), (int32_t)stride,
                                      (int32_t)dilation, bn[0], bn[1], bn[2], bn[3], (float)eps, (int32_t)act,
                                      current_stream(x)),
             "spice_depthwise_conv_f16");
    return out;
}

Tensor squeeze_excite(const Tensor &x, const Tensor &w1, const Tensor &b1, const Tensor &w2, const Tensor &b2) {
    check_dev(x, "x", at::kHalf);
    TORCH_CHECK_VALUE(x.dim() == 4, "squeeze_excite: x must be [N,C,H,W]");
    const int64_t n = x.size(0), c = x.size(1), hw = x.size(2) * x.size(3), r = b1.numel();
    for (const Tensor *t : {&w1, &b1, &w2, &b2}) {
        check_dev(*t, "squeeze_excite parameter", at::kFloat);
        check_same_device(x, *t, "squeeze_excite parameter");
    }
    TORCH_CHECK_VALUE(r >= 1 && w1.numel() == r * c && w2.numel() == c * r && b2.numel() == c,
                      "squeeze_excite: w1 [R,C], b1 [R], w2 [C,R], b2 [C] for x with C channels");
    c10::DeviceGuard guard(x.device());
    Tensor out = at::empty_like(x);
    if (x.numel() == 0) return out;
    const size_t need = spice_squeeze_excite_workspace_bytes(n, c);
    Tensor ws = at::empty({(int64_t)need}, x.options().dtype(at::kByte));
    check_rc(spice_squeeze_excite_f16(x.data_ptr(), out.data_ptr(), n, c, hw, w1.data_ptr<float>(), b1.data_ptr<float>(), r,
                                      w2.data_ptr<float>(), b2.data_ptr<float>(), ws.data_ptr(), need, current_stream(x)),
             "spice_squeeze_excite_f16");
    return out;
}

// mean over the plane -> 1x1 convolution -> batch-norm -> activation -> spread over the plane, into `out` (a channel
// slice of a concatenation) when given
Tensor pooled_branch(const Tensor &x, const Tensor &weight2d, const Tensor &mean, const Tensor &var, const Tensor &weight,
                     const Tensor &bias, double eps, int64_t act, const c10::optional<Tensor> &out_opt) {
    check_dev(x, "x", at::kHalf);
    check_dev(weight2d, "weight", at::kHalf);
    check_same_device(x, weight2d, "weight");
    TORCH_CHECK_VALUE(x.dim() == 4 && weight2d.dim() == 2 && weight2d.size(1) == x.size(1),
                      "pooled_branch: x must be [N,Cin,H,W] and weight [Cout,Cin]");
    const int64_t n = x.size(0), cin = x.size(1), h = x.size(2), w = x.size(3), c = weight2d.size(0);
    for (const Tensor *p : {&mean, &var, &weight, &bias})
        TORCH_CHECK_TYPE(p->scalar_type() == at::kFloat && p->device() == x.device() && p->numel() == c && p->is_contiguous(),
                         "pooled_branch: the four parameter vectors must be contiguous fp32 [", c, "] on ", x.device());
    c10::DeviceGuard guard(x.device());
    if (out_opt.has_value()) check_slice_out(*out_opt, x, n, c, h, w, at::kHalf, "pooled_branch");
    Tensor out = out_opt.has_value() ? *out_opt : at::empty({n, c, h, w}, x.options());
    if (n == 0 || h * w == 0) return out;
    const int64_t image_stride = out_opt.has_value() && n > 1 ? out.stride(0) : 0;
    const size_t need = spice_pooled_branch_workspace_bytes(n, cin);
    Tensor ws = at::empty({(int64_t)need}, x.options().dtype(at::kByte));
    check_rc(spice_pooled_branch_f16(x.data_ptr(), weight2d.data_ptr(), out.data_ptr(), n, cin, h * w, c, image_stride,
                                     mean.data_ptr<float>(), var.data_ptr<float>(), weight.data_ptr<float>(),
                                     bias.data_ptr<float>(), (float)eps, (int32_t)act, ws.data_ptr(), need, current_stream(x)),
             "spice_pooled_branch_f16");
    return out;
}

Tensor depthwise_conv3x3(const Tensor &x, const Tensor &weight, int64_t dilation) {
    check_dev(x, "x", at::kHalf);
    TORCH_CHECK_TYPE(weight.scalar_type() == at::kHalf || weight.scalar_type() == at::kFloat,
                     "depthwise_conv3x3: weight dtype ", weight.scalar_type(), " not supported");
    check_dev(weight, "weight", weight.scalar_type());
    check_same_device(x, weight, "weight");
    TORCH_CHECK_VALUE(x.dim() == 4 && weight.dim() == 4 && weight.size(0) == x.size(1) && weight.size(1) == 1 &&
                          weight.size(2) == 3 && weight.size(3) == 3,
                      "depthwise_conv3x3: x must be [N,C,H,W] and weight [C,1,3,3]");
    c10::DeviceGuard guard(x.device());
    Tensor out = at::empty_like(x);
    const int32_t wcode = weight.scalar_type() == at::kFloat ? 0 : 1;
    const int rc = dilation == 1
                       ? spice_depthwise_conv3x3(x.data_ptr(), weight.data_ptr(), wcode, out.data_ptr(), x.size(0),
                                                 x.size(1), (int32_t)x.size(2), (int32_t)x.size(3), current_stream(x))
                       : spice_depthwise_conv3x3_dilated(x.data_ptr(), weight.data_ptr(), wcode, out.data_ptr(),
                                                         x.size(0), x.size(1), (int32_t)x.size(2), (int32_t)x.size(3),
                                                         (int32_t)dilation, current_stream(x));
    check_rc(rc, "spice_depthwise_conv3x3");
    return out;
}

// SPICE_CONV1X1_FUSED=0 keeps the 1x1 convolutions off the hand-written matrix-core kernel (A/B measurements).
static bool fused_conv1x1_enabled() {
    static const bool on = [] {
        const char *e = getenv("SPICE_CONV1X1_FUSED");
        return e == nullptr || atoi(e) != 0;
    }();
    return on;
}

// The plain product.  Where hipBLASLt offers no scratch-free algorithm for a problem size (SPICE_ERR_UNSUPPORTED),
// the same matrix product goes through at::bmm (rocBLAS): the one operator with a route outside the library.
Tensor conv1x1_f16(const Tensor &x, const Tensor &weight2d, int64_t stride) {
    TORCH_CHECK_VALUE(stride == 1 || stride == 2, "conv1x1_f16: stride must be 1 or 2");
    TORCH_CHECK_TYPE(x.scalar_type() == at::kHalf && weight2d.scalar_type() == at::kHalf,
                     "conv1x1_f16: x and weight must be fp16");
    TORCH_CHECK(x.is_cuda(), "x is on ", x.device(),
                "; the MI355X path has no CPU implementation (move the tensor to a HIP device)");
    TORCH_CHECK(weight2d.device() == x.device(), "conv1x1_f16: weight must be on x's device");
    TORCH_CHECK_VALUE(x.dim() == 4 && weight2d.dim() == 2 && weight2d.size(1) == x.size(1) && x.is_contiguous() &&
                          weight2d.is_contiguous(),
                      "conv1x1_f16: x must be contiguous [N,Cin,H,W] and weight contiguous [Cout,Cin]");
    const int64_t n = x.size(0), cin = x.size(1), h = x.size(2), w = x.size(3), cout = weight2d.size(0);
    c10::DeviceGuard guard(x.device());
    if (stride == 2) {   // the even pixels, picked by the matrix-core kernel while it stages x
        TORCH_CHECK_VALUE(h % 2 == 0 && w % 16 == 0, "conv1x1_f16: stride 2 needs an even height and a width that is a multiple of 16");
        Tensor half = at::empty({n, cout, h / 2, w / 2}, x.options());
        if (half.numel() != 0)
            check_rc(spice_conv1x1_stride2_f16(x.data_ptr(), weight2d.data_ptr(), half.data_ptr(), n, cin, (int32_t)h, (int32_t)w,
                                               cout, current_stream(x)),
                     "spice_conv1x1_stride2_f16");
        return half;
    }
    Tensor out = at::empty({n, cout, h, w}, x.options());
    // short sums over many pixels are bound by memory: the hand-written kernel (no tail) moves them faster than the
    // library's product (MI355X, 64 -> 256 channels on 131,072 pixels: 17.8 us against 22.5 us); from K = 512 on
    // hipBLASLt's larger tiles win
    if (cin <= 256 && fused_conv1x1_enabled()) {
        const int frc = spice_conv1x1_bn_act_f16(x.data_ptr(), weight2d.data_ptr(), out.data_ptr(), n, cin, h * w, cout,
                                                 nullptr, nullptr, nullptr, nullptr, 0.0f, nullptr, nullptr, nullptr, nullptr,
                                                 nullptr, 0.0f, 0, 0, nullptr, nullptr, nullptr, nullptr, 0.0f, 0,
                                                 current_stream(x));
        if (frc != SPICE_ERR_UNSUPPORTED) {
            check_rc(frc, "spice_conv1x1_bn_act_f16 (plain product)");
            return out;
        }
    }
    const int rc = spice_conv1x1_f16(x.data_ptr(), weight2d.data_ptr(), out.data_ptr(), n, cin, h * w, cout,
                                     current_stream(x));
    if (rc == SPICE_ERR_UNSUPPORTED)
        return at::bmm(weight2d.unsqueeze(0).expand({n, cout, cin}), x.view({n, cin, h * w})).view({n, cout, h, w});
    check_rc(rc, "spice_conv1x1_f16");
    return out;
}

// relu?( bn(weight2d @ x) [+ residual | + bn_r(residual)] ) behind ONE call from Python.  First choice: the
// matrix-core kernel that applies the tail to the product before it is stored (one launch).  Where that kernel
// does not take the shapes (SPICE_ERR_UNSUPPORTED) or SPICE_CONV1X1_FUSED=0 is set: conv1x1_f16 followed by the
// batch-norm tail (bn_act or bn_act_dual) written in place over the product (two launches).  `route` 1 / 2 asks
// for the one-launch / two-launch route explicitly (1 raises where the kernel does not take the shapes).
Tensor conv1x1_bn_act(const Tensor &x, const Tensor &weight2d, const Tensor &mean, const Tensor &var,
                      const Tensor &weight, const Tensor &bias, double eps, const c10::optional<Tensor> &residual,
                      const c10::optional<Tensor> &r_mean, const c10::optional<Tensor> &r_var,
                      const c10::optional<Tensor> &r_weight, const c10::optional<Tensor> &r_bias, double r_eps, bool relu,
                      int64_t route, const c10::optional<Tensor> &out_opt, const c10::optional<Tensor> &p_mean,
                      const c10::optional<Tensor> &p_var, const c10::optional<Tensor> &p_weight,
                      const c10::optional<Tensor> &p_bias, double p_eps, bool p_relu, int64_t act) {
    const int32_t tail = act == 2 ? 2 : (relu || act == 1 ? 1 : 0);   // 0 none, 1 ReLU, 2 SiLU
    TORCH_CHECK_VALUE(route >= 0 && route <= 2, "conv1x1_bn_act: route must be 0 (choose), 1 (one launch) or 2 (two launches)");
    TORCH_CHECK_TYPE(x.scalar_type() == at::kHalf && weight2d.scalar_type() == at::kHalf,
                     "conv1x1_bn_act: x and weight must be fp16");
    TORCH_CHECK(x.is_cuda(), "x is on ", x.device(),
                "; the MI355X path has no CPU implementation (move the tensor to a HIP device)");
    TORCH_CHECK(weight2d.device() == x.device(), "conv1x1_bn_act: weight must be on x's device");
    TORCH_CHECK_VALUE(x.dim() == 4 && weight2d.dim() == 2 && weight2d.size(1) == x.size(1) && x.is_contiguous() &&
                          weight2d.is_contiguous(),
                      "conv1x1_bn_act: x must be contiguous [N,Cin,H,W] and weight contiguous [Cout,Cin]");
    const int64_t n = x.size(0), cin = x.size(1), h = x.size(2), w = x.size(3), c = weight2d.size(0);
    for (const Tensor *p : {&mean, &var, &weight, &bias})
        TORCH_CHECK_TYPE(p->scalar_type() == at::kFloat && p->device() == x.device() && p->numel() == c && p->is_contiguous(),
                         "conv1x1_bn_act: the four parameter vectors must be contiguous fp32 [", c, "] on ", x.device());
    const void *res = nullptr;
    if (residual.has_value()) {
        TORCH_CHECK_VALUE(residual->dim() == 4 && residual->size(0) == n && residual->size(1) == c &&
                              residual->size(2) == h && residual->size(3) == w,
                          "conv1x1_bn_act: residual must have the product's shape");
        TORCH_CHECK_TYPE(residual->scalar_type() == at::kHalf && residual->device() == x.device() &&
                             residual->is_contiguous(),
                         "conv1x1_bn_act: residual must be contiguous fp16 on x's device");
        res = residual->data_ptr();
    }
    const float *rm = nullptr, *rv = nullptr, *rw = nullptr, *rb = nullptr;
    if (r_mean.has_value()) {
        TORCH_CHECK_VALUE(res != nullptr && r_var.has_value() && r_weight.has_value() && r_bias.has_value(),
                          "conv1x1_bn_act: a residual batch-norm needs the residual and all four of its vectors");
        for (const Tensor *p : {&*r_mean, &*r_var, &*r_weight, &*r_bias})
            TORCH_CHECK_TYPE(p->scalar_type() == at::kFloat && p->device() == x.device() && p->numel() == c &&
                                 p->is_contiguous(),
                             "conv1x1_bn_act: the residual's parameter vectors must be contiguous fp32 [", c, "]");
        rm = r_mean->data_ptr<float>();
        rv = r_var->data_ptr<float>();
        rw = r_weight->data_ptr<float>();
        rb = r_bias->data_ptr<float>();
    }
    // a batch-norm (+ ReLU) of x that is still pending: per input channel
    const float *pm = nullptr, *pv = nullptr, *pw = nullptr, *pb = nullptr;
    if (p_mean.has_value()) {
        TORCH_CHECK_VALUE(p_var.has_value() && p_weight.has_value() && p_bias.has_value(),
                          "conv1x1_bn_act: a pending batch-norm of x needs all four of its vectors");
        for (const Tensor *p : {&*p_mean, &*p_var, &*p_weight, &*p_bias})
            TORCH_CHECK_TYPE(p->scalar_type() == at::kFloat && p->device() == x.device() && p->numel() == cin &&
                                 p->is_contiguous(),
                             "conv1x1_bn_act: the pending batch-norm's vectors must be contiguous fp32 [", cin, "]");
        pm = p_mean->data_ptr<float>();
        pv = p_var->data_ptr<float>();
        pw = p_weight->data_ptr<float>();
        pb = p_bias->data_ptr<float>();
    }
    c10::DeviceGuard guard(x.device());
    if (out_opt.has_value()) check_slice_out(*out_opt, x, n, c, h, w, at::kHalf, "conv1x1_bn_act");
    if (route == 1 || (route == 0 && fused_conv1x1_enabled())) {
        Tensor out = out_opt.has_value() ? *out_opt : at::empty({n, c, h, w}, x.options());
        const int64_t image_stride = out_opt.has_value() && n > 1 ? out.stride(0) : 0;
        const int rc = n == 0 ? SPICE_OK
                              : spice_conv1x1_bn_act_f16(x.data_ptr(), weight2d.data_ptr(), out.data_ptr(), n, cin, h * w, c,
                                                         mean.data_ptr<float>(), var.data_ptr<float>(),
                                                         weight.data_ptr<float>(), bias.data_ptr<float>(), (float)eps, res, rm,
                                                         rv, rw, rb, (float)r_eps, tail, image_stride, pm, pv, pw, pb,
                                                         (float)p_eps, p_relu ? 1 : 0, current_stream(x));
        if (rc != SPICE_ERR_UNSUPPORTED || route == 1) {
            check_rc(rc, "spice_conv1x1_bn_act_f16");
            return out;
        }
    }
    Tensor xin = x;
    if (pm != nullptr) {   // the pending batch-norm as a launch of its own
        xin = at::empty_like(x);
        if (x.numel() != 0)
            check_rc(spice_bn_act(x.data_ptr(), nullptr, xin.data_ptr(), 1, pm, pv, pw, pb, (float)p_eps, n, cin, h * w,
                                  p_relu ? 1 : 0, current_stream(x)),
                     "spice_bn_act (pending batch-norm of x)");
    }
    Tensor t = conv1x1_f16(xin, weight2d, 1);
    TORCH_CHECK_VALUE(t.is_contiguous(), "conv1x1_bn_act: the product is not contiguous");
    int rc;
    if (rm != nullptr) {
        rc = spice_bn_act_dual(t.data_ptr(), res, t.data_ptr(), 1, mean.data_ptr<float>(), var.data_ptr<float>(),
                               weight.data_ptr<float>(), bias.data_ptr<float>(), (float)eps, rm, rv, rw, rb, (float)r_eps,
                               t.size(0), c, t.size(2) * t.size(3), tail, current_stream(t));
    } else {
        rc = spice_bn_act(t.data_ptr(), res, t.data_ptr(), 1, mean.data_ptr<float>(), var.data_ptr<float>(),
                          weight.data_ptr<float>(), bias.data_ptr<float>(), (float)eps, t.size(0), c, t.size(2) * t.size(3),
                          tail, current_stream(t));
    }
    check_rc(rc, "spice_bn_act (behind conv1x1)");
    if (out_opt.has_value()) {   // the two-launch route writes a dense tensor: one more pass into the slice
        out_opt->copy_(t);
        return *out_opt;
    }
    return t;
}

// relu?( bn( conv2d(x, weight, stride, padding = dilation, dilation) ) ) for a 3x3 convolution; weight9 is the weight as
// [Cout, 9 * Cin], tap-major.  One launch of the matrix-core kernel.
// A 7x7 convolution of stride 2 and padding 3 without bias (the ResNet stem): weight as [Cout, Cin * 49], the
// convolution's own weight tensor flattened.
Tensor conv7x7_s2(const Tensor &x, const Tensor &weight2d) {
    TORCH_CHECK_TYPE(x.scalar_type() == at::kHalf && weight2d.scalar_type() == at::kHalf,
                     "conv7x7_s2: x and weight must be fp16");
    TORCH_CHECK(x.is_cuda(), "x is on ", x.device(),
                "; the MI355X path has no CPU implementation (move the tensor to a HIP device)");
    TORCH_CHECK(weight2d.device() == x.device(), "conv7x7_s2: weight must be on x's device");
    TORCH_CHECK_VALUE(x.dim() == 4 && weight2d.dim() == 2 && weight2d.size(1) == 49 * x.size(1) && x.is_contiguous() &&
                          weight2d.is_contiguous(),
                      "conv7x7_s2: x must be contiguous [N,Cin,H,W] and weight contiguous [Cout, Cin*49]");
    const int64_t n = x.size(0), cin = x.size(1), h = x.size(2), w = x.size(3), cout = weight2d.size(0);
    TORCH_CHECK_VALUE(h % 2 == 0 && w % 16 == 0 && h > 0 && w > 0,
                      "conv7x7_s2: needs an even height and a width that is a multiple of 16");
    c10::DeviceGuard guard(x.device());
    Tensor out = at::empty({n, cout, h / 2, w / 2}, x.options());
    if (out.numel() != 0)
        check_rc(spice_conv7x7_stride2_f16(x.data_ptr(), weight2d.data_ptr(), out.data_ptr(), n, cin, (int32_t)h, (int32_t)w, cout,
                                           current_stream(x)),
                 "spice_conv7x7_stride2_f16");
    return out;
}

Tensor conv3x3_bn_act(const Tensor &x, const Tensor &weight9, const Tensor &mean, const Tensor &var, const Tensor &weight,
                      const Tensor &bias, double eps, bool relu, int64_t stride, int64_t dilation, int64_t act) {
    TORCH_CHECK_TYPE(x.scalar_type() == at::kHalf && weight9.scalar_type() == at::kHalf,
                     "conv3x3_bn_act: x and weight must be fp16");
    TORCH_CHECK(x.is_cuda(), "x is on ", x.device(),
                "; the MI355X path has no CPU implementation (move the tensor to a HIP device)");
    TORCH_CHECK(weight9.device() == x.device(), "conv3x3_bn_act: weight must be on x's device");
    TORCH_CHECK_VALUE(x.dim() == 4 && weight9.dim() == 2 && weight9.size(1) == 9 * x.size(1) && x.is_contiguous() &&
                          weight9.is_contiguous(),
                      "conv3x3_bn_act: x must be contiguous [N,Cin,H,W] and weight contiguous [Cout, 9*Cin]");
    TORCH_CHECK_VALUE((stride == 1 || stride == 2) && dilation >= 1, "conv3x3_bn_act: stride 1 or 2, dilation >= 1");
    const int64_t n = x.size(0), cin = x.size(1), h = x.size(2), w = x.size(3), c = weight9.size(0);
    for (const Tensor *p : {&mean, &var, &weight, &bias})
        TORCH_CHECK_TYPE(p->scalar_type() == at::kFloat && p->device() == x.device() && p->numel() == c && p->is_contiguous(),
                         "conv3x3_bn_act: the four parameter vectors must be contiguous fp32 [", c, "] on ", x.device());
    TORCH_CHECK_VALUE(h % stride == 0 && w % stride == 0, "conv3x3_bn_act: the plane must be whole stride cells");
    c10::DeviceGuard guard(x.device());
    Tensor out = at::empty({n, c, h / stride, w / stride}, x.options());
    if (out.numel() != 0)
        check_rc(spice_conv3x3_bn_act_f16(x.data_ptr(), weight9.data_ptr(), out.data_ptr(), n, cin, (int32_t)h, (int32_t)w, c,
                                          (int32_t)stride, (int32_t)dilation, mean.data_ptr<float>(), var.data_ptr<float>(),
                                          weight.data_ptr<float>(), bias.data_ptr<float>(), (float)eps,
                                          act == 2 ? 2 : (relu || act == 1 ? 1 : 0), current_stream(x)),
                 "spice_conv3x3_bn_act_f16");
    return out;
}

int64_t abi_version() { return spice_abi_version(); }

}  // namespace

TORCH_LIBRARY(spice, m) {
    m.def("abi_version() -> int", &abi_version);
    m.def("importance_run(Tensor[] weights, Tensor[] grads, Tensor(a!) scores, Tensor items, int n_items, int mode, "
          "bool accumulate) -> ()", &importance_run);
    m.def("importance_cols_run(Tensor[] weights, Tensor(a!) scores, Tensor(b!) partial, Tensor items, int n_items, "
          "Tensor folds, int n_folds, int mode) -> ()", &importance_cols_run);
    m.def("gather_run(Tensor[] srcs, Tensor(a!) dst, Tensor idx, Tensor items, int n_items, int plan_flags) -> ()",
          &gather_run);
    m.def("gather_upload(Tensor(a!) dst, Tensor src) -> ()", &gather_upload);
    m.def("confusion_update(Tensor(a!) conf, Tensor preds, Tensor target, int num_classes, Tensor(b!) workspace) -> ()",
          &confusion_update);
    m.def("confusion_update_static(Tensor(a!) conf, Tensor preds, Tensor target, int num_classes, "
          "Tensor(b!)? invalid_count) -> ()", &confusion_update_static);
    m.def("upsample_argmax_confusion(Tensor(a!) conf, Tensor low, Tensor labels, bool align_corners, float scale_h, "
          "float scale_w, Tensor(b!) workspace) -> bool", &upsample_argmax_confusion);
    m.def("kd_teacher_prepare(Tensor teacher, float temperature) -> (Tensor, Tensor)", &kd_teacher_prepare);
    m.def("sweep_head(Tensor low, Tensor teacher, Tensor? teacher_hneg, Tensor labels, Tensor(a!) conf, bool align_corners, float scale_h, "
          "float scale_w, float temperature, float alpha, int ignore_index, Tensor? class_weight, "
          "Tensor(b!) kd_workspace, Tensor(c!) conf_workspace, Tensor(d!)? out) -> Tensor", &sweep_head);
    m.def("argmax_confusion_update(Tensor(a!) conf, Tensor logits, Tensor target, int num_classes, "
          "Tensor(b!) workspace) -> ()", &argmax_confusion_update);
    m.def("kdlite_forward(Tensor student, Tensor teacher, Tensor labels, float temperature, float alpha, "
          "int ignore_index, float boundary_weight, bool confidence_weight, Tensor? class_weight, bool want_grad, "
          "float coef_ce, float coef_kd, Tensor? upstream, Tensor(a!) workspace, bool normalisers_ready) -> (Tensor, Tensor)", &kdlite_forward);
    m.def("kdlite_label_pass(Tensor labels, int num_classes, int ignore_index, float boundary_weight, bool confidence_weight, "
          "Tensor? class_weight, Tensor(a!) workspace) -> bool", &kdlite_label_pass);
    m.def("scale_inplace(Tensor(a!) x, Tensor scale) -> ()", &scale_inplace);
    m.def("upsample_bilinear(Tensor x, int out_h, int out_w, bool align_corners, float scale_h, float scale_w, "
          "int out_dtype, Tensor(a!)? out) -> Tensor(a!)", &upsample_bilinear);
    m.def("bn_act(Tensor x, Tensor mean, Tensor var, Tensor weight, Tensor bias, float eps, Tensor? residual, "
          "bool relu, int act=0) -> Tensor", &bn_act);
    m.def("bn_act_dual(Tensor x, Tensor mean, Tensor var, Tensor weight, Tensor bias, float eps, Tensor residual, "
          "Tensor r_mean, Tensor r_var, Tensor r_weight, Tensor r_bias, float r_eps, bool relu) -> Tensor", &bn_act_dual);
    m.def("bn_act_maxpool(Tensor x, Tensor mean, Tensor var, Tensor weight, Tensor bias, float eps, bool relu, "
          "bool want_full) -> (Tensor, Tensor)", &bn_act_maxpool);
    m.def("depthwise_conv3x3(Tensor x, Tensor weight, int dilation) -> Tensor", &depthwise_conv3x3);
    m.def("depthwise_conv(Tensor x, Tensor weight, int stride, int dilation, Tensor? mean=None, Tensor? var=None, "
          "Tensor? bn_weight=None, Tensor? bn_bias=None, float eps=0.0, int act=0) -> Tensor", &depthwise_conv);
    m.def("squeeze_excite(Tensor x, Tensor w1, Tensor b1, Tensor w2, Tensor b2) -> Tensor", &squeeze_excite);
    m.def("pooled_branch(Tensor x, Tensor weight2d, Tensor mean, Tensor var, Tensor weight, Tensor bias, float eps, int act, Tensor? out) -> Tensor", &pooled_branch);
    m.def("conv1x1_f16(Tensor x, Tensor weight2d, int stride) -> Tensor", &conv1x1_f16);
    m.def("conv7x7_s2(Tensor x, Tensor weight2d) -> Tensor", &conv7x7_s2);
    m.def("conv3x3_bn_act(Tensor x, Tensor weight9, Tensor mean, Tensor var, Tensor weight, Tensor bias, float eps, "
          "bool relu, int stride, int dilation, int act=0) -> Tensor", &conv3x3_bn_act);
    m.def("conv1x1_bn_act(Tensor x, Tensor weight2d, Tensor mean, Tensor var, Tensor weight, Tensor bias, float eps, "
          "Tensor? residual, Tensor? r_mean, Tensor? r_var, Tensor? r_weight, Tensor? r_bias, float r_eps, bool relu, "
          "int route, Tensor(a!)? out, Tensor? p_mean, Tensor? p_var, Tensor? p_weight, Tensor? p_bias, float p_eps, "
          "bool p_relu, int act=0) -> Tensor(a!)", &conv1x1_bn_act);
}
