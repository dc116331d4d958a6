"""ctypes binding of libspice_amd.so (the C ABI declared in include/spice_amd.h).

The library is the only compute path of this package: there is no CPU or stock-op
fallback.  Loading fails loudly when the shared object has not been built.
torch is imported first so that the HIP runtime bundled with PyTorch-ROCm is the one
the library binds to (both carry the SONAME libamdhip64.so.7).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libspice_amd.so")
TORCH_LIB_PATH = os.path.join(_HERE, "lib", "libspice_torch.so")

ABI_VERSION = 5
ERR_ARG, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_BLAS, ERR_STATIC_LIMIT = 10001, 10002, 10003, 10004, 10005
IMP_L1, IMP_L2, IMP_TAYLOR = 0, 1, 2


class RowsDesc(Structure):
    _fields_ = [("w_off", c_int64), ("g_off", c_int64), ("out_off", c_int64),
                ("rows", c_int64), ("row_len", c_int64)]


class RowsItem(Structure):
    _fields_ = [("w_off", c_int64), ("g_off", c_int64), ("out_off", c_int64),
                ("row_len", c_int32), ("n_rows", c_int32), ("lanes", c_int32), ("flags", c_int32)]


class ColsDesc(Structure):
    _fields_ = [("w_off", c_int64), ("out_off", c_int64), ("rows", c_int64),
                ("cols", c_int64), ("inner", c_int64)]


class ColsItem(Structure):
    _fields_ = [("w_off", c_int64), ("out_off", c_int64), ("row_stride", c_int32),
                ("n_rows", c_int32), ("n_flat", c_int32), ("flags", c_int32)]


class ColsFold(Structure):
    _fields_ = [("part_off", c_int64), ("out_off", c_int64), ("n_slabs", c_int32),
                ("n_flat", c_int32), ("inner", c_int32), ("cols", c_int32)]


class KdliteParams(Structure):
    _fields_ = [("temperature", c_float), ("alpha", c_float), ("boundary_weight", c_float),
                ("confidence_weight", c_int32), ("ignore_index", c_int64),
                ("class_weight", c_void_p)]


class GatherDesc(Structure):
    _fields_ = [("src_off", c_int64), ("dst_off", c_int64), ("idx_out_off", c_int64),
                ("idx_in_off", c_int64), ("dst_rows", c_int64), ("dst_cols", c_int64),
                ("src_cols", c_int64), ("inner", c_int64)]


class GatherItem(Structure):
    _fields_ = [("src_off", c_int64), ("dst_off", c_int64), ("idx_out_off", c_int64),
                ("idx_in_off", c_int64), ("row0", c_int32), ("n_rows", c_int32),
                ("dst_cols", c_int32), ("src_cols", c_int32), ("inner", c_int32),
                ("lanes", c_int32), ("flags", c_int32), ("pad_", c_int32)]


# name -> (restype, argtypes); every symbol include/spice_amd.h declares
SIGNATURES = {
    "spice_abi_version": (c_int32, []),
    "spice_error_string": (c_char_p, [c_int32]),
    "spice_importance_plan": (c_int64, [POINTER(RowsDesc), c_int32, POINTER(RowsItem), c_int64]),
    "spice_importance_run": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                       c_int32, c_void_p]),
    "spice_importance_cols_plan": (c_int64, [POINTER(ColsDesc), c_int32, POINTER(ColsItem), c_int64,
                                             POINTER(ColsFold), POINTER(c_int64)]),
    "spice_importance_cols_run": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                            c_void_p, c_int64, c_int32, c_void_p]),
    "spice_kdlite_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "spice_kdlite_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                       c_int64, POINTER(KdliteParams), c_void_p, c_void_p, c_float,
                                       c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "spice_kdlite_forward_prepared": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                                c_int64, POINTER(KdliteParams), c_void_p, c_void_p, c_float,
                                                c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "spice_kdlite_label_pass": (c_int32, [c_void_p, c_int64, c_int64, c_int64, c_int64, POINTER(KdliteParams), c_void_p,
                                          c_size_t, c_void_p]),
    "spice_kd_teacher_prepare": (c_int32, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_float, c_void_p, c_void_p,
                                           c_void_p]),
    "spice_sweep_head_f16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                                       c_int64, c_int32, c_float, c_float, POINTER(KdliteParams), c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "spice_upsample_argmax_confusion_f16": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                                                      c_int64, c_int32, c_float, c_float, c_void_p, c_void_p, c_size_t,
                                                      c_void_p]),
    "spice_scale_inplace": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p]),
    "spice_confusion_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "spice_confusion_update": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p,
                                         c_size_t, c_void_p]),
    "spice_confusion_update_static": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "spice_argmax_confusion_update": (c_int32, [c_void_p, c_int32, c_void_p, c_int64, c_int64,
                                                c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "spice_gather_plan": (c_int64, [POINTER(GatherDesc), c_int32, POINTER(GatherItem), c_int64]),
    "spice_gather_run": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "spice_gather_upload": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "spice_bn_act": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                               c_int64, c_int64, c_int64, c_int32, c_void_p]),
    "spice_bn_act_dual": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int64, c_int64, c_int32,
                                    c_void_p]),
    "spice_depthwise_conv3x3": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int64, c_int32, c_int32,
                                          c_void_p]),
    "spice_pooled_branch_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "spice_pooled_branch_f16": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32, c_void_p, c_size_t,
                                          c_void_p]),
    "spice_squeeze_excite_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "spice_squeeze_excite_f16": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                                           c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "spice_depthwise_conv_f16": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int64, c_int32, c_int32,
                                           c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                           c_int32, c_void_p]),
    "spice_depthwise_conv3x3_dilated": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int64, c_int32,
                                                  c_int32, c_int32, c_void_p]),
    "spice_bn_act_maxpool_f16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                           c_int64, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "spice_conv3x3_bn_act_f16": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int64, c_int32,
                                           c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32, c_void_p]),
    "spice_conv1x1_stride2_f16": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int64, c_void_p]),
    "spice_conv7x7_stride2_f16": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int64, c_void_p]),
    "spice_conv1x1_f16": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "spice_conv1x1_bn_act_f16": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32, c_int64,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32, c_void_p]),
    "spice_upsample_bilinear": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int64, c_int32, c_int32,
                                          c_int32, c_int32, c_int32, c_float, c_float, c_void_p]),
    "spice_upsample_bilinear_into": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int64, c_int32, c_int32,
                                          c_int32, c_int32, c_int32, c_float, c_float, c_int64, c_int64, c_void_p]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """The loaded library; raises NativeLibraryError if it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `make` at the repository root "
            "(or __graft_entry__.build()); this package has no fallback compute path")
    handle = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as exc:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if handle.spice_abi_version() != ABI_VERSION:
        raise NativeLibraryError(
            f"{LIB_PATH} has ABI {handle.spice_abi_version()}, expected {ABI_VERSION}; rebuild it")
    _lib = handle
    return _lib


_torch_ops = None


def load_torch_ops():
    """torch.ops.spice after loading libspice_torch.so (TORCH_LIBRARY registrations over the C ABI)."""
    global _torch_ops
    if _torch_ops is not None:
        return _torch_ops
    if not os.path.exists(TORCH_LIB_PATH):
        raise NativeLibraryError(
            f"{TORCH_LIB_PATH} is missing: build it with `make` at the repository root "
            "(or __graft_entry__.build()); this package has no fallback compute path")
    lib()  # the C ABI library first: the extension links against it
    torch.ops.load_library(TORCH_LIB_PATH)
    if int(torch.ops.spice.abi_version()) != ABI_VERSION:
        raise NativeLibraryError(f"{TORCH_LIB_PATH} was built against another ABI version; rebuild it")
    _torch_ops = torch.ops.spice
    return _torch_ops


def check(code: int, what: str) -> None:
    """Turn a non-zero return code of the C ABI into a Python exception."""
    if code == 0:
        return
    text = lib().spice_error_string(int(code)).decode()
    if code == ERR_ARG:
        raise ValueError(f"{what}: {text}")
    raise RuntimeError(f"{what}: {text} (code {code})")


def current_stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_device_tensor(t: torch.Tensor, name: str, dtype: torch.dtype) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device.type != "cuda":
        raise RuntimeError(
            f"{name} is on {t.device}; the MI355X path has no CPU implementation "
            "(move the tensor to a HIP device)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
