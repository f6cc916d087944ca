"""ctypes binding of libtripled_hip.so (C ABI: include/tripled_hip.h).

There is deliberately no fallback: if the shared library is missing or a call returns an
error code, an exception is raised.  Nothing in here touches ``oracle/``.
"""
import ctypes
import os

import torch

from . import dispatch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtripled_hip.so")

TD_MAX_SRC = 4
_c_float_p = ctypes.c_void_p   # device pointers travel as integers
_c_u8_p = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/tripled_hip.h one to one
_PTRARR = ctypes.POINTER(ctypes.c_void_p)
_I, _F, _P = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
_LLARR = ctypes.POINTER(ctypes.c_longlong)     # host arrays
_IARR = ctypes.POINTER(ctypes.c_int)
ABI_VERSION = 3      # include/tripled_hip.h: TD_ABI_VERSION

SIGNATURES = {
    "td_abi_version": (_I, []),
    "td_error_string": (ctypes.c_char_p, [_I]),
    "td_last_hip_error": (ctypes.c_char_p, []),
    "td_photo_num_blocks": (_I, [_I, _I, _I]),
    "td_photo_bwd_num_blocks": (_I, [_I, _I, _I]),
    "td_photo_identity": (_I, [_P, _PTRARR, _I, _I, _I, _I, _P, _P, _PTRARR, _P]),
    "td_photo_fwd": (_I, [_P, _PTRARR, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P]),
    "td_photo_bwd": (_I, [_P, _PTRARR, _P, _PTRARR, _I, _P, _P, _P, _P, _P, _I, _P, _F, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P]),
    "td_pack_rgbx": (_I, [_P, _I, _I, _I, _P, _P]),
    "td_upsample_adjoint": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "td_upsample_adjoint_planes": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "td_reduce_dP": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "td_sum_scaled": (_I, [_P, _I, _F, _P, _P]),
    "td_area_downsample": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "td_smooth_num_blocks": (_I, [_I, _I, _I]),
    "td_smooth_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "td_smooth_finish": (_I, [_P, _I, _I, _I, _F, _P, _P]),
    "td_smooth_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _F, _P, _P, _P, _I, _P]),
    "td_maxpool5_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "td_maxpool5_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "td_maxpool5_bwd_add": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "td_conv1x1_fwd_sum": (_I, [_P, _P, ctypes.c_longlong, _I, _I, _P, _P, _P, _P]),
    "td_maxpool3s2_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "td_maxpool3s2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "td_join_fwd": (_I, [_P, _P, _P, _I, ctypes.c_longlong, _I, _I, _I, _P, _P]),
    "td_join_bwd": (_I, [_P, _I, ctypes.c_longlong, _I, _I, _I, _P, _P, _P, _P]),
    "td_join_up2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "td_join_up2_bwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "td_bn_workspace_floats": (ctypes.c_longlong, [ctypes.c_longlong, _I, _I]),
    "td_bn_fwd": (_I, [_P, _P, _I, _P, _P, _P, _P, _F, _F, _I, ctypes.c_longlong, _I, _I, _P, _P, _P, _P, _P]),
    "td_conv1x1_stat_rows": (_I, [ctypes.c_longlong, _I, _I]),
    "td_conv1x1_fwd": (_I, [_P, _P, ctypes.c_longlong, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "td_conv3x3_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "td_conv3x3_wgrad_workspace_floats": (ctypes.c_longlong, [_I, _I, _I, _I, _I]),
    "td_conv3x3_wgrad": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "td_conv1x1_wgrad_workspace_floats": (ctypes.c_longlong, [ctypes.c_longlong, _I, _I]),
    "td_conv1x1_wgrad": (_I, [_P, _P, ctypes.c_longlong, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "td_conv1x1_wgrad_group": (_I, [_I, _PTRARR, _PTRARR, _LLARR, _IARR, _IARR, _IARR, _IARR, _IARR, _IARR, _PTRARR, _PTRARR, _P]),
    "td_bn_fwd_from_partials": (_I, [_P, _P, _I, _P, _P, _P, _P, _F, _F, _I, ctypes.c_longlong, _I, _I, _P, _I, _P, _P, _P, _P]),
    "td_bias_act_workspace_floats": (ctypes.c_longlong, [ctypes.c_longlong, _I]),
    "td_bias_act_fwd": (_I, [_P, _P, _I, _I, ctypes.c_longlong, _I, _I, _P, _P]),
    "td_bias_act_bwd": (_I, [_P, _P, _I, ctypes.c_longlong, _I, _I, _P, _P, _I, _P, _P]),
    "td_gather_flat": (_I, [_PTRARR, _LLARR, _LLARR, _I, _I, _P, _P]),
    "td_adam_flat": (_I, [_P, _P, _P, _P, _P, ctypes.c_longlong, ctypes.c_longlong, _P, _P, _F, ctypes.c_double, ctypes.c_double, _F, _P, _F, _P]),
    "td_bn_partial_rows": (_I, [ctypes.c_longlong, _I, _I]),
    "td_bn_fwd_partials": (_I, [_P, _I, ctypes.c_longlong, _I, _I, _P, _P]),
    "td_bn_bwd_partials": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, ctypes.c_longlong, _I, _I, _P, _P]),
    "td_bn_bwd_from_partials": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, ctypes.c_longlong, _I, _I, _P, _I, _P, _P, _P, _P, _P]),
    "td_conv1x1_fwd_bnrelu": (_I, [_P, _P, ctypes.c_longlong, _I, _I, _I, _P, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P]),
    "td_conv1x1_dgrad": (_I, [_P, _P, ctypes.c_longlong, _I, _I, _I, _P, _P, _P]),
    "td_conv1x1_dgrad_bnsums": (_I, [_P, _P, ctypes.c_longlong, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "td_conv1x1_dgrad_bnbwd": (_I, [_P, _P, _P, ctypes.c_longlong, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "td_bn_bwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, ctypes.c_longlong, _I, _I, _P, _P, _P, _P, _P, _P]),
    "td_edge_weights": (_I, [_P, _I, _I, _I, _F, ctypes.POINTER(ctypes.c_float), _P, _P]),
    "td_featreg_num_blocks": (_I, [_I, _I, _I, _I]),
    "td_featreg_fwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P]),
    "td_featreg_bwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P]),
    "td_recon_num_tasks": (_I, [_I, _I, _I]),
    "td_recon_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "td_recon_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "td_reflpad1_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "td_reflpad1_bwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "td_up2_reflpad1_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "td_up2_reflpad1_bwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "td_featwarp_num_blocks": (_I, [_I, _I, _I]),
    "td_featwarp_fwd": (_I, [_P, _PTRARR, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P]),
    "td_featwarp_bwd": (_I, [_P, _PTRARR, _I, _I, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _F, _F, _P, _PTRARR,
                             _P, _P, _P]),
    "td_featwarp_dsrc_finish": (_I, [_P, _P, _F, _I, ctypes.c_longlong, _I, _P, _P]),
    "td_reduce_partials": (_I, [_P, _I, _I, _I, _P, _P]),
    "td_bn_sync_fwd_sums": (_I, [_P, _I, ctypes.c_longlong, _I, _I, _P, _P, _P]),
    "td_bn_sync_fwd_apply": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _F, _F, _I, ctypes.c_longlong, _I, _I, _P, _P, _P, _P]),
    "td_bn_sync_bwd_sums": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, ctypes.c_longlong, _I, _I, _P, _P, _P]),
    "td_bn_sync_bwd_dx": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, ctypes.c_longlong, _I, _I, _P, _P, _P, _P, _P, _P]),
    "td_fp8_num_blocks": (_I, [ctypes.c_longlong]),
    "td_fp8_amax_partials": (_I, [_P, _I, ctypes.c_longlong, _P, _P]),
    "td_fp8_quantize": (_I, [_P, _I, ctypes.c_longlong, _P, _P, _P, _P]),
    "td_color_jitter": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "td_l1map_fwd": (_I, [_P, _I, _LLARR, _P, _I, _I, _I, _I, _F, _P, _P]),
    "td_l1map_bwd": (_I, [_P, _I, _LLARR, _P, _P, _I, _I, _I, _I, _F, _P, _P]),
    "td_rgb2lab": (_I, [_P, _I, _I, _I, _F, _F, _F, _P, _P]),
    "td_pose_fwd": (_I, [_P, _P, _IARR, _P, _I, _I, _P, _P, _P]),
    "td_pose_bwd": (_I, [_P, _P, _IARR, _P, _I, _I, _P, _P, _P, _P, _P]),
}

DTYPE_CODES = {torch.float32: 0, torch.bfloat16: 1}


class NativeLibraryError(RuntimeError):
    pass


_lib = None


def load(path=None):
    """Load (once) and return the ctypes handle; raises NativeLibraryError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("TD_HIP_LIB") or LIB_PATH      # TD_HIP_LIB: alternative build of the same ABI
    if not os.path.exists(path):
        raise NativeLibraryError(
            "libtripled_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C <package>/csrc`; there is no non-HIP fallback" % path)
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise NativeLibraryError("cannot load %s: %s" % (path, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError("%s does not export %s" % (path, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.td_abi_version() != ABI_VERSION:
        raise NativeLibraryError("ABI version mismatch: library %d, binding %d" % (lib.td_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code, what):
    dispatch.hip(what)
    if code != 0:
        lib = load()
        msg = lib.td_error_string(code).decode()
        hip = lib.td_last_hip_error().decode()
        raise NativeLibraryError("%s failed: %s (%d)%s" % (what, msg, code, (" [" + hip + "]") if hip else ""))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be a dense CUDA/HIP tensor."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NativeLibraryError("libtripled_hip needs device tensors (got a %s tensor)" % t.device)
    if not t.is_contiguous():
        raise NativeLibraryError("libtripled_hip needs contiguous tensors")
    return ctypes.c_void_p(t.data_ptr())


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def strides_array(t):
    """Element strides of a 4-D tensor as a host long-long array."""
    return (ctypes.c_longlong * 4)(*[int(v) for v in t.stride()])


def int_array(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])
