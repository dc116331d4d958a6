"""Tensor-level entry points of the MI355X hot path.

Every launch goes through `torch.ops.spice.*`, the thin torch C++ extension
(spice_amd/csrc/torch_ext.cpp -> lib/libspice_torch.so) that checks the tensors, takes torch's
current HIP stream and calls the C ABI of libspice_amd.so.  Host-only planning (work-item
tables) calls the C ABI directly (spice_amd/_native.py).  Nothing here computes on the CPU and
nothing falls back to stock torch ops.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as nat

_ops = None


def spice_ops():
    """torch.ops.spice, with lib/libspice_torch.so loaded; raises NativeLibraryError if it is missing."""
    global _ops
    if _ops is None:
        nat.lib()
        _ops = nat.load_torch_ops()
    return _ops

# --------------------------------------------------------------------------------------
# work-item tables


_UPLOAD_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def upload(host: torch.Tensor, device: torch.device) -> torch.Tensor:
    """A small host tensor (launch plans, index lists) copied to the device while the current stream is busy: the copy
    runs on a stream of its own that has nothing else queued (a copy from pageable memory returns when it has run; on
    the current stream that would be behind everything launched so far), and the current stream is ordered behind it."""
    device = torch.device(device)
    if device.type != "cuda" or host.numel() == 0:
        return host.to(device)
    cur = torch.cuda.current_stream(device)
    up = _UPLOAD_STREAMS.get(device.index)
    if up is None:
        up = _UPLOAD_STREAMS[device.index] = torch.cuda.Stream(device=device)
    with torch.cuda.stream(up):
        out = host.to(device)
    cur.wait_stream(up)
    out.record_stream(cur)
    return out


class _StagingRing:
    """Page-locked host memory for small uploads queued on ONE stream: two halves, a half is written again only after
    the launches that read it have run (an event recorded when the writer moved on to the other half)."""

    HALF = 1 << 21

    def __init__(self):
        self.buf = torch.empty(2 * self.HALF, dtype=torch.uint8, pin_memory=True)
        self.half, self.pos = 0, 0
        self.left = [None, None]   # the event behind the last copy out of each half

    def take(self, nbytes: int, stream: "torch.cuda.Stream") -> torch.Tensor:
        nbytes = (nbytes + 255) // 256 * 256
        if self.pos + nbytes > self.HALF:
            ev = torch.cuda.Event()
            ev.record(stream)
            self.left[self.half] = ev
            self.half, self.pos = 1 - self.half, 0
            if self.left[self.half] is not None:
                self.left[self.half].synchronize()
                self.left[self.half] = None
        start = self.half * self.HALF + self.pos
        self.pos += nbytes
        return self.buf[start:start + nbytes]


_STAGING: Dict[tuple, _StagingRing] = {}


def _device_bytes(size: int, device: torch.device) -> torch.Tensor:
    return torch.empty(size, dtype=torch.uint8, device=device)


def _queue_copy(dst: torch.Tensor, src: torch.Tensor) -> None:
    # a launch, not a copy command: a copy command from page-locked memory keeps the caller until the stream has run
    # that far (up to 5 ms inside a candidate loop that is ahead of the device)
    spice_ops().gather_upload(dst, src)


def upload_staged(parts: Sequence[np.ndarray], device: torch.device) -> List[torch.Tensor]:
    """Small host arrays (index lists, work-item tables) to the device as ONE launch on the CURRENT stream, in order
    with the launches that read them: the bytes are written to page-locked staging memory and a kernel
    (spice_gather_upload) reads them from there through the memory's device mapping, so the host moves on at once (a
    copy command, from pageable or page-locked memory alike, keeps the caller until the stream has run that far).
    Returns one device tensor per part (views of one allocation, each starting on a 256-byte boundary, typed like its
    part).  More than one staging half (2 MiB) of parts goes as several launches; a single part larger than that as an
    ordinary copy."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("upload_staged needs a HIP device")
    parts = [a if a.flags.c_contiguous else np.ascontiguousarray(a) for a in parts]
    stream = torch.cuda.current_stream(device)
    key = (device.index, stream.cuda_stream)
    out: List[Optional[torch.Tensor]] = [None] * len(parts)
    group, size = [], 0

    def flush():
        nonlocal group, size
        if not group:
            return
        ring = _STAGING.get(key)
        if ring is None:
            ring = _STAGING[key] = _StagingRing()
        seg = ring.take(size, stream)
        host = seg.numpy()
        dev = _device_bytes(size, device)
        pos = 0
        for k in group:
            a = parts[k]
            n = int(a.nbytes)
            if n:
                host[pos:pos + n] = a.reshape(-1).view(np.uint8)
            piece = dev[pos:pos + n]
            out[k] = piece if a.dtype == np.uint8 else piece.view(_TORCH_OF_NUMPY[a.dtype.type])
            pos += (n + 255) // 256 * 256
        _queue_copy(dev, seg[:size])
        group, size = [], 0

    for k, a in enumerate(parts):
        n = (int(a.nbytes) + 255) // 256 * 256
        if n > _StagingRing.HALF:   # too large for the staging memory: a copy of its own
            out[k] = upload(torch.from_numpy(a), device)
            continue
        if size + n > _StagingRing.HALF:
            flush()
        group.append(k)
        size += n
    flush()
    return out


_TORCH_OF_NUMPY = {np.int64: torch.int64, np.int32: torch.int32, np.float32: torch.float32, np.uint8: torch.uint8}


def _upload_items(items, count: int, device: torch.device) -> torch.Tensor:
    """Copy a ctypes array of `count` fixed-size records into a device byte tensor."""
    nbytes = ctypes.sizeof(items._type_) * count
    host = torch.empty(max(nbytes, 1), dtype=torch.uint8)
    if nbytes:
        ctypes.memmove(host.data_ptr(), ctypes.addressof(items), nbytes)
    return host.to(device)


@dataclass
class RowsLayout:
    """One weight tensor viewed as [rows, row_len], addressed by element offsets."""

    w_off: int
    g_off: int
    out_off: int
    rows: int
    row_len: int


class RowsPlan:
    """Work-item table of the per-output-channel reduction for a fixed set of layers."""

    def __init__(self, layers: Sequence[RowsLayout], device: torch.device):
        L = nat.lib()
        self.device = torch.device(device)
        self.layers = list(layers)
        descs = (nat.RowsDesc * max(len(self.layers), 1))()
        for i, l in enumerate(self.layers):
            descs[i] = nat.RowsDesc(l.w_off, l.g_off, l.out_off, l.rows, l.row_len)
        n = L.spice_importance_plan(descs, len(self.layers), None, 0)
        if n < 0:
            nat.check(-n, "spice_importance_plan")
        items = (nat.RowsItem * max(n, 1))()
        got = L.spice_importance_plan(descs, len(self.layers), items, n)
        assert got == n
        self.n_items = int(n)
        self.items = _upload_items(items, self.n_items, self.device)
        self.weight_elems = sum(l.rows * l.row_len for l in self.layers)
        self.score_elems = sum(l.rows for l in self.layers)

    def run(self, weights: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], scores: torch.Tensor,
            mode: int, accumulate: bool) -> None:
        """One launch over every planned layer.  The plan's offsets count from the lowest address
        among `weights` (and among `grads` in Taylor mode; ignored otherwise)."""
        spice_ops().importance_run(list(weights), list(grads), scores, self.items, self.n_items, mode,
                                   bool(accumulate))

    def algorithmic_bytes(self, mode: int) -> int:
        """Bytes one launch must move: every weight (and gradient) element once plus the scores."""
        if mode == nat.IMP_TAYLOR:
            return 8 * self.weight_elems + 8 * self.score_elems
        return 4 * self.weight_elems + 4 * self.score_elems


@dataclass
class ColsLayout:
    w_off: int
    out_off: int
    rows: int
    cols: int
    inner: int


class ColsPlan:
    """Work-item tables of the per-input-channel reduction (two stages, fixed order)."""

    def __init__(self, layers: Sequence[ColsLayout], device: torch.device):
        L = nat.lib()
        self.device = torch.device(device)
        self.layers = list(layers)
        nl = len(self.layers)
        descs = (nat.ColsDesc * max(nl, 1))()
        for i, l in enumerate(self.layers):
            descs[i] = nat.ColsDesc(l.w_off, l.out_off, l.rows, l.cols, l.inner)
        part = ctypes.c_int64(0)
        n = L.spice_importance_cols_plan(descs, nl, None, 0, None, ctypes.byref(part))
        if n < 0:
            nat.check(-n, "spice_importance_cols_plan")
        items = (nat.ColsItem * max(n, 1))()
        folds = (nat.ColsFold * max(nl, 1))()
        got = L.spice_importance_cols_plan(descs, nl, items, n, folds, ctypes.byref(part))
        assert got == n
        self.n_items, self.n_folds = int(n), nl
        self.items = _upload_items(items, self.n_items, self.device)
        self.folds = _upload_items(folds, self.n_folds, self.device)
        self.partial = torch.empty(max(int(part.value), 4), dtype=torch.float32, device=self.device)

    def run(self, weights: Sequence[torch.Tensor], scores: torch.Tensor, mode: int) -> None:
        spice_ops().importance_cols_run(list(weights), scores, self.partial, self.items, self.n_items,
                                        self.folds, self.n_folds, mode)


def _relative_offsets(tensors: Sequence[torch.Tensor], what: str) -> Tuple[int, List[int]]:
    """Base pointer (lowest address) and per-tensor element offsets from it."""
    ptrs = [t.data_ptr() for t in tensors]
    base = min(ptrs)
    offs = []
    for p in ptrs:
        d = p - base
        if d % 4:
            raise ValueError(f"{what}: tensors are not 4-byte aligned relative to each other")
        offs.append(d // 4)
    return base, offs


_rows_plan_cache: Dict[tuple, Tuple[RowsPlan, List[int]]] = {}


def channel_importance(weights: Sequence[torch.Tensor], mode: int,
                       grads: Optional[Sequence[torch.Tensor]] = None,
                       out: Optional[torch.Tensor] = None, accumulate: bool = False
                       ) -> Tuple[torch.Tensor, List[int]]:
    """Per-output-channel reduction of every tensor in `weights` (viewed as [dim0, rest]).

    mode: IMP_L1 (sum |w|), IMP_L2 (sum w^2) or IMP_TAYLOR (sum |g||w|, needs `grads`).
    Returns (flat fp32 score vector, start offset of each tensor's scores in it).
    One kernel launch covers all tensors.
    """
    if not weights:
        raise ValueError("channel_importance: no tensors")
    device = weights[0].device
    for i, w in enumerate(weights):
        nat.require_device_tensor(w, f"weights[{i}]", torch.float32)
    if mode == nat.IMP_TAYLOR:
        if grads is None or len(grads) != len(weights):
            raise ValueError("channel_importance: Taylor mode needs one gradient per weight")
        for i, (w, g) in enumerate(zip(weights, grads)):
            nat.require_device_tensor(g, f"grads[{i}]", torch.float32)
            if g.shape != w.shape:
                raise ValueError(f"grads[{i}] shape {tuple(g.shape)} != weight {tuple(w.shape)}")
    key = (mode == nat.IMP_TAYLOR, tuple((w.data_ptr(), tuple(w.shape)) for w in weights),
           tuple(g.data_ptr() for g in grads) if mode == nat.IMP_TAYLOR else ())
    hit = _rows_plan_cache.get(key)
    if hit is None:
        w_base, w_offs = _relative_offsets(weights, "weights")
        if mode == nat.IMP_TAYLOR:
            g_base, g_offs = _relative_offsets(grads, "grads")
        else:
            g_base, g_offs = 0, [0] * len(weights)
        out_offs, pos = [], 0
        layers = []
        for w, wo, go in zip(weights, w_offs, g_offs):
            rows = int(w.shape[0]) if w.dim() > 0 else 1
            row_len = w.numel() // max(rows, 1)
            layers.append(RowsLayout(wo, go, pos, rows, row_len))
            out_offs.append(pos)
            pos += rows
        plan = RowsPlan(layers, device)
        if len(_rows_plan_cache) > 64:
            _rows_plan_cache.clear()
        _rows_plan_cache[key] = (plan, out_offs)
        hit = _rows_plan_cache[key]
    plan, out_offs = hit
    if out is None:
        if accumulate:
            raise ValueError("channel_importance: accumulate=True needs an existing `out`")
        out = torch.empty(plan.score_elems, dtype=torch.float32, device=device)
    else:
        nat.require_device_tensor(out, "out", torch.float32)
        if out.numel() != plan.score_elems:
            raise ValueError(f"out has {out.numel()} elements, expected {plan.score_elems}")
    plan.run(weights, grads if mode == nat.IMP_TAYLOR else (), out, mode, accumulate)
    return out, out_offs


# --------------------------------------------------------------------------------------
# confusion matrix

_workspaces: Dict[tuple, torch.Tensor] = {}


def _dev_index(device) -> int:
    device = torch.device(device)
    return torch.cuda.current_device() if device.index is None else device.index


def _workspace(kind: str, device: torch.device, nbytes: int) -> torch.Tensor:
    """Per (kind, device, stream) scratch buffer, zero-initialised once."""
    key = (kind, _dev_index(device), nat.current_stream_ptr(device))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def confusion_update(conf: torch.Tensor, preds: torch.Tensor, target: torch.Tensor,
                     num_classes: int) -> None:
    """conf[t, p] += 1 for every pixel with 0 <= t < num_classes (in place, int64).
    Device, dtype, contiguity and shape are checked by the extension."""
    if not conf.is_cuda:
        nat.require_device_tensor(conf, "conf", torch.int64)
    ws = _workspace("conf%d" % num_classes, conf.device,
                    nat.lib().spice_confusion_workspace_bytes(preds.numel(), num_classes))
    spice_ops().confusion_update(conf, preds, target, num_classes, ws)


_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def argmax_confusion_update(conf: torch.Tensor, logits: torch.Tensor, target: torch.Tensor,
                            num_classes: int) -> None:
    """confusion_update(conf, logits.argmax(1), target) without materialising the arg-max."""
    if not conf.is_cuda:
        nat.require_device_tensor(conf, "conf", torch.int64)
    ws = _workspace("conf%d" % num_classes, conf.device,
                    nat.lib().spice_confusion_workspace_bytes(target.numel(), num_classes))
    spice_ops().argmax_confusion_update(conf, logits, target, num_classes, ws)


def invalid_prediction_count(device: torch.device, num_classes: int) -> int:
    """Pixels seen so far (this device and stream) whose prediction was outside [0, C).

    Synchronises.  The reference fails on such input; here they are skipped and counted.
    """
    key = ("conf%d" % num_classes, _dev_index(device), nat.current_stream_ptr(torch.device(device)))
    ws = _workspaces.get(key)
    if ws is None:
        return 0
    return int(ws[:8].view(torch.int64).item())


# --------------------------------------------------------------------------------------
# KD-Lite loss


def kdlite_forward(student: torch.Tensor, teacher: torch.Tensor, labels: torch.Tensor, *,
                   temperature: float, alpha: float, ignore_index: int = 255,
                   boundary_weight: float = 1.0, confidence_weight: bool = False,
                   class_weight: Optional[torch.Tensor] = None, want_grad: bool = False,
                   coef: Optional[Tuple[float, float]] = None,
                   upstream: Optional[torch.Tensor] = None,
                   workspace: Optional[torch.Tensor] = None, normalisers_ready: bool = False
                   ) -> Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]:
    """Fused KD-Lite loss.  Returns (out3 = [total, ce, kd], grad or None, sums[8] fp64).

    grad = coef[0]*d(ce)/d(student) + coef[1]*d(kd)/d(student); coef defaults to
    (alpha, 1-alpha), i.e. d(total)/d(student).  `upstream` (device, 3 floats: incoming
    gradients of total/ce/kd) overrides coef on the device.
    `workspace` + `normalisers_ready`: kdlite_label_pass has put the loss normalisers of these labels into this
    workspace already (gradient form only); the call is ordered behind that pass by the caller.
    """
    if not student.is_cuda:
        nat.require_device_tensor(student, "student_logits", torch.float32)
    if student.dim() != 4:
        raise ValueError("student/teacher logits must both be [N,C,H,W]")
    n, c, h, w = student.shape
    dev = student.device
    ws = workspace if workspace is not None else _workspace("kdlite", dev, nat.lib().spice_kdlite_workspace_bytes(n, c, h, w))
    cce, ckd = coef if coef is not None else (float(alpha), 1.0 - float(alpha))
    out3, grad = spice_ops().kdlite_forward(student, teacher, labels, float(temperature), float(alpha),
                                            int(ignore_index), float(boundary_weight), bool(confidence_weight),
                                            class_weight, bool(want_grad), cce, ckd, upstream, ws,
                                            bool(normalisers_ready and want_grad))
    if not want_grad:
        grad = None
    sums = ws[16:80].view(torch.float64)
    return out3, grad, sums


def kdlite_workspace(device: torch.device) -> torch.Tensor:
    """A zeroed workspace of a caller's own for kdlite_label_pass / kdlite_forward(workspace=...)."""
    return torch.zeros(nat.lib().spice_kdlite_workspace_bytes(1, 1, 1, 1), dtype=torch.uint8, device=device)


def kdlite_label_pass(labels: torch.Tensor, num_classes: int, workspace: torch.Tensor, *, ignore_index: int = 255,
                      boundary_weight: float = 1.0, confidence_weight: bool = False,
                      class_weight: Optional[torch.Tensor] = None) -> bool:
    """The label pass of the loss's gradient form by itself, on the current stream, into `workspace`: what the loss
    divides by depends on the labels alone, so a training step can take it beside the networks' forwards.  False when
    the options have no separate label pass (boundary / confidence weighting): nothing was launched."""
    if not labels.is_cuda:
        nat.require_device_tensor(labels, "labels", torch.int64)
    return bool(spice_ops().kdlite_label_pass(labels, int(num_classes), int(ignore_index), float(boundary_weight),
                                              bool(confidence_weight), class_weight, workspace))


def upsample_argmax_confusion(conf: torch.Tensor, low: torch.Tensor, target: torch.Tensor, *,
                              align_corners: bool = True, scale_factor: Optional[float] = None) -> bool:
    """argmax_confusion_update(conf, upsample_bilinear(low -> target's size), target) in one launch over the small
    fp16 logits `low` [N,C,h,w]; the full-size logits are never written.  False when the library does not cover the
    case (nothing was launched: up-sample, then argmax_confusion_update)."""
    if not low.is_cuda:
        nat.require_device_tensor(low, "low_logits", torch.float16)
    c = low.shape[1]
    ws = _workspace("conf%d" % c, conf.device, nat.lib().spice_confusion_workspace_bytes(target.numel(), c))
    sf = float(scale_factor) if scale_factor else 0.0
    return bool(spice_ops().upsample_argmax_confusion(conf, low, target, bool(align_corners), sf, sf, ws))


def kd_teacher_prepare(teacher: torch.Tensor, temperature: float) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """Once per teacher of a sweep: (softmax(teacher / T) over the class axis [N,C,H,W] fp32, per-pixel
    sum p ln p [N,H,W] fp32) for sweep_head(teacher=..., teacher_hneg=...); None when the library does not cover the
    layout (H*W % 4 != 0)."""
    if not teacher.is_cuda:
        nat.require_device_tensor(teacher, "teacher_logits", torch.float32)
    probs, hneg = spice_ops().kd_teacher_prepare(teacher, float(temperature))
    return (probs, hneg) if probs.numel() or not teacher.numel() else None


def sweep_head(low: torch.Tensor, teacher: torch.Tensor, labels: torch.Tensor, conf: torch.Tensor, *,
               temperature: float, alpha: float, ignore_index: int = 255,
               class_weight: Optional[torch.Tensor] = None, align_corners: bool = True,
               scale_factor: Optional[float] = None,
               teacher_hneg: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """The tail of a sweep candidate in one launch: bilinear up-sampling of the head's fp16 logits
    `low` [N,C,h,w] to the teacher's size, KD-Lite loss against `teacher` [N,C,H,W] fp32 and
    conf[label, argmax] += 1, without the full-size student logits ever being written.  Same values as
    upsample_bilinear -> kdlite_forward + argmax_confusion_update (interpolated logits bit for bit).
    With `teacher_hneg`, `teacher` holds the probabilities of kd_teacher_prepare (same temperature) instead of
    logits: one exponential per logit less in the per-candidate pass, same loss within fp32 rounding.
    Returns out3 = [total, ce, kd] (`out` itself when given: three contiguous floats), or None when the library
    does not cover the case (nothing was launched: make the three calls)."""
    if not low.is_cuda:
        nat.require_device_tensor(low, "low_logits", torch.float16)
    if low.dim() != 4 or teacher.dim() != 4:
        raise ValueError("low and teacher logits must both be 4-d")
    n, c, h, w = teacher.shape
    dev = low.device
    kd_ws = _workspace("kdlite", dev, nat.lib().spice_kdlite_workspace_bytes(n, c, h, w))
    conf_ws = _workspace("conf%d" % c, dev, nat.lib().spice_confusion_workspace_bytes(labels.numel(), c))
    sf = float(scale_factor) if scale_factor else 0.0
    out3 = spice_ops().sweep_head(low, teacher, teacher_hneg, labels, conf, bool(align_corners), sf, sf, float(temperature),
                                  float(alpha), int(ignore_index), class_weight, kd_ws, conf_ws, out)
    return out3 if out3.numel() else None


def kdlite_bad_label_count(device: torch.device) -> int:
    """Labels seen so far (this device and stream) that were neither ignore_index nor in [0,C)."""
    device = torch.device(device)
    ws = _workspaces.get(("kdlite", _dev_index(device), nat.current_stream_ptr(device)))
    return 0 if ws is None else int(ws[:8].view(torch.int64).item())


def scale_inplace(x: torch.Tensor, scale: torch.Tensor) -> None:
    """x *= scale (a device scalar); a no-op on the device when scale == 1."""
    spice_ops().scale_inplace(x, scale)


# --------------------------------------------------------------------------------------
# channel gather


@dataclass
class GatherSpec:
    """dst[o,i,k] = src[keep_out[o], keep_in[i], k]; offsets in elements, -1 = keep all."""

    src_off: int
    dst_off: int
    idx_out_off: int
    idx_in_off: int
    dst_rows: int
    dst_cols: int
    src_cols: int
    inner: int


class GatherPlan:
    def __init__(self, specs: Sequence[GatherSpec], device: torch.device, defer_upload: bool = False):
        """defer_upload: the work items stay on the host (`items_host`, bytes) for the caller to upload together with
        other plans' in one copy and hand back as `items`."""
        L = nat.lib()
        self.device = torch.device(device)
        self.specs = list(specs)
        nt = len(self.specs)
        descs = (nat.GatherDesc * max(nt, 1))()
        for i, s in enumerate(self.specs):
            descs[i] = nat.GatherDesc(s.src_off, s.dst_off, s.idx_out_off, s.idx_in_off,
                                      s.dst_rows, s.dst_cols, s.src_cols, s.inner)
        n = L.spice_gather_plan(descs, nt, None, 0)
        if n < 0:
            nat.check(-n, "spice_gather_plan")
        items = (nat.GatherItem * max(n, 1))()
        got = L.spice_gather_plan(descs, nt, items, n)
        assert got == n
        self.n_items = int(n)
        self.plan_flags = 0
        for k in range(self.n_items):
            self.plan_flags |= items[k].flags
        if defer_upload:
            nbytes = ctypes.sizeof(nat.GatherItem) * self.n_items
            self.items_host = torch.empty(max(nbytes, 1), dtype=torch.uint8)
            if nbytes:
                ctypes.memmove(self.items_host.data_ptr(), ctypes.addressof(items), nbytes)
            self.items = None
        else:
            self.items = _upload_items(items, self.n_items, self.device)
        self.moved_elems = sum(s.dst_rows * s.dst_cols * s.inner for s in self.specs)
        self.index_elems = sum((s.dst_rows if s.idx_out_off >= 0 else 0) +
                               (s.dst_cols if s.idx_in_off >= 0 else 0) for s in self.specs)

    def run(self, srcs: Sequence[torch.Tensor], dst: torch.Tensor, idx: torch.Tensor) -> None:
        """One launch for every planned tensor.  The plan's source offsets count from the lowest
        address among `srcs`, destination offsets from `dst`, index offsets from `idx`."""
        spice_ops().gather_run(list(srcs), dst, idx, self.items, self.n_items, self.plan_flags)

    def algorithmic_bytes(self) -> int:
        return 8 * self.moved_elems + 8 * self.index_elems


def gather_channels(src: torch.Tensor, keep_out: Optional[torch.Tensor],
                    keep_in: Optional[torch.Tensor]) -> torch.Tensor:
    """Single-tensor convenience: index_select on dim 0 and/or dim 1 in one pass."""
    nat.require_device_tensor(src, "src", torch.float32)
    rows = int(src.shape[0])
    cols = int(src.shape[1]) if src.dim() > 1 else 1
    inner = src.numel() // max(rows * cols, 1) if src.numel() else 1
    idx_parts, o_off, i_off, pos = [], -1, -1, 0
    d_rows, d_cols = rows, cols
    if keep_out is not None:
        nat.require_device_tensor(keep_out, "keep_out", torch.int64)
        o_off, pos, d_rows = pos, pos + keep_out.numel(), keep_out.numel()
        idx_parts.append(keep_out)
    if keep_in is not None:
        nat.require_device_tensor(keep_in, "keep_in", torch.int64)
        i_off, pos, d_cols = pos, pos + keep_in.numel(), keep_in.numel()
        idx_parts.append(keep_in)
    idx = torch.cat(idx_parts) if idx_parts else torch.zeros(1, dtype=torch.int64, device=src.device)
    shape = list(src.shape)
    shape[0] = d_rows
    if src.dim() > 1:
        shape[1] = d_cols
    dst = torch.empty(shape, dtype=torch.float32, device=src.device)
    plan = GatherPlan([GatherSpec(0, 0, o_off, i_off, d_rows, d_cols, cols, max(inner, 1))], src.device)
    plan.run([src], dst, idx)
    return dst


def upsample_bilinear(x: torch.Tensor, size: Optional[Tuple[int, int]] = None,
                      scale_factor: Optional[float] = None, align_corners: bool = False,
                      out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F.interpolate(x, size / scale_factor, mode="bilinear", align_corners) for an NCHW device
    tensor (fp32 or fp16), interpolated in fp32 and written as `out_dtype` (default: x's dtype, or out's).
    `out`: write into this tensor instead of a new one; it may be a channel slice of a larger NCHW tensor
    (a concatenation buffer).  Forward only; up-sampling only."""
    if x.dtype not in (torch.float32, torch.float16):
        raise TypeError(f"upsample_bilinear: dtype {x.dtype} not supported")
    nat.require_device_tensor(x, "x", x.dtype)
    if x.dim() != 4:
        raise ValueError("upsample_bilinear: x must be [N, C, H, W]")
    if (size is None) == (scale_factor is None):
        raise ValueError("upsample_bilinear: give exactly one of size and scale_factor")
    n, c, h, w = (int(v) for v in x.shape)
    if size is not None:
        oh, ow = int(size[0]), int(size[1])
        sh = sw = 0.0
    else:
        sh = sw = float(scale_factor)
        oh, ow = int(h * sh), int(w * sw)   # floor, as torch
    out_dtype = out_dtype or (out.dtype if out is not None else x.dtype)
    if out_dtype not in (torch.float32, torch.float16):
        raise TypeError(f"upsample_bilinear: output dtype {out_dtype} not supported")
    return spice_ops().upsample_bilinear(x, oh, ow, bool(align_corners), sh, sw, _DTYPE_CODE[out_dtype], out)


def bn_act_supported(x: torch.Tensor) -> bool:
    """Whether bn_act reproduces torch's kernels for this tensor: an NCHW fp32 tensor, or an fp16 one
    with an even number of elements per plane."""
    if x.dim() != 4 or not x.is_contiguous():
        return False
    if x.dtype == torch.float32:
        return True
    return x.dtype == torch.float16 and (x.shape[2] * x.shape[3]) % 2 == 0


_bn_fn = None


def bn_act(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
           eps: float, residual: Optional[torch.Tensor] = None, relu: bool = False, act: int = 0) -> torch.Tensor:
    """act?(batch_norm(x, running statistics) [+ residual]) in one launch; a new tensor of x's dtype.  `relu`, or
    `act` = 1 (ReLU) / 2 (SiLU).
    Called ~60 times per forward of a ResNet-50 model: the argument checks are the extension's."""
    global _bn_fn
    if _bn_fn is None:
        _bn_fn = spice_ops().bn_act
    return _bn_fn(x, mean, var, weight, bias, eps, residual, relu, act)


def bn_act_maxpool_supported(x: torch.Tensor) -> bool:
    """Whether bn_act_maxpool takes this tensor: contiguous fp16 NCHW, even height, width a multiple of 16."""
    return (x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float16 and x.shape[2] % 2 == 0
            and x.shape[3] % 16 == 0 and x.shape[3] <= 4096)


def bn_act_maxpool(x: torch.Tensor, bn: "torch.nn.BatchNorm2d", relu: bool = True, want_full: bool = True):
    """(relu?(bn(x)), max_pool2d(that, kernel 3, stride 2, padding 1)) in one launch: the stem of a ResNet encoder
    behind its first convolution.  With want_full=False the first tensor is not computed (None is returned)."""
    full, pooled = spice_ops().bn_act_maxpool(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, relu,
                                              want_full)
    return (full if want_full else None), pooled


_bn_dual_fn = None


def bn_act_dual(x: torch.Tensor, bn: "torch.nn.BatchNorm2d", residual: torch.Tensor, res_bn: "torch.nn.BatchNorm2d",
                relu: bool = False) -> torch.Tensor:
    """relu?(bn(x) + res_bn(residual)) with both modules' running statistics, in one launch: the tail of a
    residual block whose shortcut is a projection with a batch-norm of its own."""
    global _bn_dual_fn
    if _bn_dual_fn is None:
        _bn_dual_fn = spice_ops().bn_act_dual
    return _bn_dual_fn(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, residual,
                       res_bn.running_mean, res_bn.running_var, res_bn.weight, res_bn.bias, res_bn.eps, relu)


def depthwise_conv3x3_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """fp16 NCHW input whose width is a multiple of 8 (at most 2048), [C,1,3,3] fp16 or fp32 weight."""
    return (x.dim() == 4 and x.dtype == torch.float16 and x.is_contiguous() and x.shape[3] % 8 == 0
            and x.shape[3] <= 2048 and weight.dtype in (torch.float16, torch.float32)
            and tuple(weight.shape) == (x.shape[1], 1, 3, 3) and weight.is_contiguous())


def depthwise_conv3x3(x: torch.Tensor, weight: torch.Tensor, dilation: int = 1) -> torch.Tensor:
    """conv2d(x, weight, stride=1, padding=dilation, dilation=dilation, groups=C) for fp16 x.
    fp32 sums of the nine products in tap order, one rounding (what torch's own depthwise kernel computes;
    dilation 1 or a multiple of 4)."""
    return spice_ops().depthwise_conv3x3(x, weight, int(dilation))


def depthwise_conv_supported(x: torch.Tensor, weight: torch.Tensor, stride: int, dilation: int) -> bool:
    """Whether depthwise_conv takes this call: contiguous fp16 NCHW of even width, a [C,1,k,k] weight with k = 3 or 5,
    stride 1 or 2, dilation 1 or 2."""
    return (x.dim() == 4 and x.dtype == torch.float16 and x.is_contiguous() and x.shape[3] % 2 == 0 and x.shape[2] >= 1
            and weight.dim() == 4 and weight.shape[0] == x.shape[1] and weight.shape[1] == 1
            and weight.shape[2] == weight.shape[3] and weight.shape[2] in (3, 5) and weight.is_contiguous()
            and weight.dtype in (torch.float16, torch.float32) and stride in (1, 2) and dilation in (1, 2)
            and x.shape[0] * x.shape[1] < (1 << 24))


def depthwise_conv(x: torch.Tensor, weight: torch.Tensor, stride: int = 1, dilation: int = 1,
                   bn: Optional["torch.nn.BatchNorm2d"] = None, act: int = 0) -> torch.Tensor:
    """conv2d(x, weight, stride=stride, padding=(k // 2) * dilation, dilation=dilation, groups=C) for fp16 x and a
    k x k weight, k = 3 or 5: fp32 sums of the k * k products in tap order, one rounding (what torch's own depthwise
    kernel computes).  `bn` (+ `act`: 1 ReLU, 2 SiLU): the inference batch-norm and activation behind it in the same
    launch, from the rounded result as bn_act would compute them from the stored tensor."""
    if bn is None:
        return spice_ops().depthwise_conv(x, weight, int(stride), int(dilation))
    return spice_ops().depthwise_conv(x, weight, int(stride), int(dilation), *_bn_parts(bn), int(act))


def pooled_branch(x: torch.Tensor, weight2d: torch.Tensor, bn: "torch.nn.BatchNorm2d", act: int = 1,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The image-pooling branch of an ASPP in two launches: plane means of x [N,Cin,H,W] fp16 -> 1x1 convolution
    with weight2d [Cout,Cin] fp16 -> bn (inference) -> activation (0 none, 1 ReLU, 2 SiLU) -> the value spread over
    the plane, into `out` (a channel slice of a concatenation) when given.  What AdaptiveAvgPool2d(1), conv, bn, relu
    and a bilinear up-sampling to x's size compute under fp16 autocast, to the order of the convolution's fp32 sum."""
    return spice_ops().pooled_branch(x, weight2d, bn.running_mean, bn.running_var, bn.weight, bn.bias, float(bn.eps),
                                     int(act), out)


def squeeze_excite(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """x * sigmoid(conv1x1(silu(conv1x1(mean(x), w1, b1)), w2, b2)) for fp16 NCHW x and fp32 parameters (w1 [R,C,1,1] or
    [R,C], w2 [C,R,1,1] or [C,R]): the squeeze-and-excite of an EfficientNet block in four short launches, every
    intermediate rounded to fp16 where torch's modules would store it under fp16 autocast."""
    return spice_ops().squeeze_excite(x, w1, b1, w2, b2)


_conv1x1_fn = None

# Measurement hook (bench.py): a list to which every convolution call below appends
# (kind, start event, end event, multiply-adds, bytes of x + y [+ residual]); None = off, which costs one comparison.
_conv_probe = None


def set_conv_probe(records):
    """records: a list to append to, or None to switch the hook off."""
    global _conv_probe
    _conv_probe = records


def _probed(kind, fn, args, x, cout, k, residual=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = fn(*args)
    e1.record()
    pixels = y.numel() // cout
    nbytes = 2 * (x.numel() + y.numel() + (residual.numel() if residual is not None else 0))
    _conv_probe.append((kind, e0, e1, pixels * cout * k, nbytes))
    return y


def conv1x1_f16(x: torch.Tensor, weight2d: torch.Tensor, stride: int = 1) -> torch.Tensor:
    """weight2d[Cout, Cin] @ x[n][Cin, H*W] for fp16 NCHW x: a 1x1 convolution without bias.  Stride 1: the
    matrix-core kernel for K <= 256 and planes of a multiple of 8 pixels, else hipBLASLt with descriptors and
    algorithm cached per problem size (torch.bmm where hipBLASLt offers no scratch-free algorithm).  Stride 2
    (even height, width a multiple of 16): the matrix-core kernel, which picks the even pixels while it stages x."""
    global _conv1x1_fn
    if _conv1x1_fn is None:
        _conv1x1_fn = spice_ops().conv1x1_f16
    if _conv_probe is not None:
        kind = "matrix_core_plain" if (stride == 2 or weight2d.shape[1] <= 256) else "hipblaslt_plain"
        return _probed(kind, _conv1x1_fn, (x, weight2d, stride), x, weight2d.shape[0], weight2d.shape[1])
    return _conv1x1_fn(x, weight2d, stride)


_conv1x1_bn_fn = None


def _bn_parts(bn):
    """(running_mean, running_var, weight, bias, eps) of a batch-norm module, read from its buffer and parameter
    tables directly: nn.Module's attribute fallback costs several dictionary probes per name, and a forward asks for
    these a few hundred times."""
    b, p = bn._buffers, bn._parameters
    return b["running_mean"], b["running_var"], p["weight"], p["bias"], bn.eps


def conv1x1_bn_act(x: torch.Tensor, weight2d: torch.Tensor, bn: "torch.nn.BatchNorm2d",
                   residual: Optional[torch.Tensor] = None, res_bn: Optional["torch.nn.BatchNorm2d"] = None,
                   relu: bool = False, route: int = 0, out: Optional[torch.Tensor] = None,
                   pre_bn: Optional["torch.nn.BatchNorm2d"] = None, pre_relu: bool = False, act: int = 0) -> torch.Tensor:
    """relu?(bn(weight2d @ x') [+ residual | + res_bn(residual)]) from one call of the extension, with
    x' = x, or relu?(pre_bn(x)) when x is the raw result of the convolution before and its own batch-norm is still
    pending.  route 0: the matrix-core kernel that applies the tail to the product before storing it (and pre_bn to x
    on the way in): ONE launch, where it takes the shapes (plane size a multiple of 8; at most 512 input channels
    with pre_bn); else the separate launches (pre_bn, conv1x1_f16, the batch-norm tail in place); route 1 / 2: that
    kernel / the separate launches, explicitly.  `out`: write into this tensor, which may be a channel slice of a
    larger NCHW tensor (a concatenation buffer), instead of a new one."""
    global _conv1x1_bn_fn
    if _conv1x1_bn_fn is None:
        _conv1x1_bn_fn = spice_ops().conv1x1_bn_act
    r = (None, None, None, None, 0.0) if res_bn is None else _bn_parts(res_bn)
    p = (None, None, None, None, 0.0) if pre_bn is None else _bn_parts(pre_bn)
    args = (x, weight2d, *_bn_parts(bn), residual, *r, relu, route, out, *p, pre_relu, act)   # act: 1 ReLU, 2 SiLU
    if _conv_probe is not None:
        return _probed("matrix_core_pair", _conv1x1_bn_fn, args, x, weight2d.shape[0], weight2d.shape[1], residual)
    return _conv1x1_bn_fn(*args)


_UNIT_NORMS: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def conv1x1_bias(x: torch.Tensor, weight2d: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """weight2d[Cout, Cin] @ x[n][Cin, H*W] + bias for fp16 NCHW x and an fp32 bias: a 1x1 convolution with bias (a
    segmentation head's) as ONE launch of the matrix-core kernel, the bias taking the place of a batch-norm's shift
    (mean 0, variance 1, weight 1, eps 0: the product, rounded to fp16 as a stored convolution result is, plus the bias
    in fp32, rounded to fp16).  The plane must be a multiple of 8 pixels."""
    global _conv1x1_bn_fn
    if _conv1x1_bn_fn is None:
        _conv1x1_bn_fn = spice_ops().conv1x1_bn_act
    key = (x.device, int(weight2d.shape[0]))
    unit = _UNIT_NORMS.get(key)
    if unit is None:
        unit = _UNIT_NORMS[key] = (torch.zeros(key[1], dtype=torch.float32, device=x.device),
                                   torch.ones(key[1], dtype=torch.float32, device=x.device))
    none5 = (None, None, None, None, 0.0)
    args = (x, weight2d, unit[0], unit[1], unit[1], bias, 0.0, None, *none5, False, 1, None, *none5, False)
    if _conv_probe is not None:
        return _probed("matrix_core_pair", _conv1x1_bn_fn, args, x, weight2d.shape[0], weight2d.shape[1], None)
    return _conv1x1_bn_fn(*args)


def conv3x3_bias(x: torch.Tensor, weight9: torch.Tensor, bias: torch.Tensor, stride: int = 1, dilation: int = 1) -> torch.Tensor:
    """conv2d(x, W, bias, stride, padding=dilation, dilation) for a 3x3 convolution with an fp32 bias (a segmentation
    head's), fp16 NCHW: one launch of the matrix-core kernel, the bias where a batch-norm's shift would be (the sums,
    rounded to fp16 as a stored convolution result is, plus the bias in fp32, rounded to fp16)."""
    global _conv3x3_fn
    if _conv3x3_fn is None:
        _conv3x3_fn = spice_ops().conv3x3_bn_act
    key = (x.device, int(weight9.shape[0]))
    unit = _UNIT_NORMS.get(key)
    if unit is None:
        unit = _UNIT_NORMS[key] = (torch.zeros(key[1], dtype=torch.float32, device=x.device),
                                   torch.ones(key[1], dtype=torch.float32, device=x.device))
    args = (x, weight9, unit[0], unit[1], unit[1], bias, 0.0, False, stride, dilation, 0)
    if _conv_probe is not None:
        return _probed("matrix_core_3x3", _conv3x3_fn, args, x, weight9.shape[0], weight9.shape[1])
    return _conv3x3_fn(*args)


def conv3x3_plain(x: torch.Tensor, weight9: torch.Tensor, stride: int = 1, dilation: int = 1) -> torch.Tensor:
    """conv2d(x, W, stride, padding=dilation, dilation) for a 3x3 convolution without bias and with nothing behind it
    the kernel could take along (a group-norm follows), fp16 NCHW: one launch of the matrix-core kernel with the
    identity where the batch-norm would be (mean 0, variance 1, weight 1, bias 0, eps 0: the rounded sums as they are)."""
    global _conv3x3_fn
    if _conv3x3_fn is None:
        _conv3x3_fn = spice_ops().conv3x3_bn_act
    key = (x.device, int(weight9.shape[0]))
    unit = _UNIT_NORMS.get(key)
    if unit is None:
        unit = _UNIT_NORMS[key] = (torch.zeros(key[1], dtype=torch.float32, device=x.device),
                                   torch.ones(key[1], dtype=torch.float32, device=x.device))
    args = (x, weight9, unit[0], unit[1], unit[1], unit[0], 0.0, False, stride, dilation, 0)
    if _conv_probe is not None:
        return _probed("matrix_core_3x3", _conv3x3_fn, args, x, weight9.shape[0], weight9.shape[1])
    return _conv3x3_fn(*args)


_conv7x7_fn = None


def conv7x7_s2(x: torch.Tensor, weight2d: torch.Tensor) -> torch.Tensor:
    """conv2d(x, W, stride=2, padding=3) for a 7x7 convolution without bias (the ResNet stem), fp16 NCHW, in one launch
    of the matrix-core kernel.  weight2d: W flattened to [Cout, Cin * 49], fp16.  Height even, width a multiple of 16."""
    global _conv7x7_fn
    if _conv7x7_fn is None:
        _conv7x7_fn = spice_ops().conv7x7_s2
    if _conv_probe is not None:
        return _probed("matrix_core_7x7", _conv7x7_fn, (x, weight2d), x, weight2d.shape[0], weight2d.shape[1])
    return _conv7x7_fn(x, weight2d)


def conv3x3_supported(x: torch.Tensor, stride: int, dilation: int) -> bool:
    """Whether conv3x3_bn_act takes this input: contiguous fp16 NCHW, the plane whole stride cells, rows of the result
    whole 16-byte pieces."""
    return (x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float16 and stride in (1, 2) and 1 <= dilation <= 7
            and x.shape[2] % stride == 0 and x.shape[3] % stride == 0 and (x.shape[3] // stride) % 8 == 0
            and x.shape[3] >= 8 and x.shape[1] <= 65536)


_conv3x3_fn = None


def conv3x3_bn_act(x: torch.Tensor, weight9: torch.Tensor, bn: "torch.nn.BatchNorm2d", relu: bool = False,
                   stride: int = 1, dilation: int = 1, act: int = 0) -> torch.Tensor:
    """relu?(bn(conv2d(x, W, stride, padding=dilation, dilation))) for a 3x3 convolution without bias, fp16 NCHW, in ONE
    launch of the matrix-core kernel (nine shifted 1x1 products in one fp32 sum).  weight9: W as [Cout, 9 * Cin],
    tap-major (W.permute(0, 2, 3, 1)), fp16."""
    global _conv3x3_fn
    if _conv3x3_fn is None:
        _conv3x3_fn = spice_ops().conv3x3_bn_act
    args = (x, weight9, *_bn_parts(bn), relu, stride, dilation, act)   # act: 1 ReLU, 2 SiLU
    if _conv_probe is not None:
        return _probed("matrix_core_3x3", _conv3x3_fn, args, x, weight9.shape[0], weight9.shape[1])
    return _conv3x3_fn(*args)
