"""The C-ABI library loads on a CPU-only host and exports every symbol include/tripled_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "tripled_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(td_[a-z_A-Z0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(native.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libtripled_hip.so does not export %s" % n
    assert set(names) == set(native.SIGNATURES), "ctypes binding and header disagree"


def test_argument_validation_without_gpu():
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    lib = native.load()
    assert lib.td_abi_version() == 3
    # wave tasks of 62 columns x R rows, R picked so that the tasks fill the 2048 resident wave slots in one round
    assert lib.td_photo_num_blocks(12, 192, 640) == 12 * 14 * 11      # R = 14 (even): 1848 tasks
    assert lib.td_photo_bwd_num_blocks(12, 192, 640) == 12 * 11 * 11  # R = 18: 1452 strip tasks x 2 frames = 2904 waves (3 per SIMD: 3072 slots)
    assert lib.td_photo_num_blocks(2, 32, 64) == 2 * 4 * 2            # small shapes keep 8-row tasks
    assert lib.td_smooth_num_blocks(12, 96, 320) == 12 * 6 * 5
    # null pointers / bad sizes are rejected before any launch
    assert lib.td_sum_scaled(None, 4, 1.0, None, None) == -1
    assert lib.td_photo_identity(None, None, 2, 1, 8, 8, None, None, None, None) == -1
    assert lib.td_smooth_finish(None, 1, 8, 8, 1.0, None, None) == -1
    assert b"bad argument" in lib.td_error_string(-1)
    # round-2 entry points: the same contract (nothing is launched on a bad argument)
    import ctypes
    four = (ctypes.c_longlong * 4)(1, 1, 1, 1)
    two = (ctypes.c_int * 2)(1, 0)
    assert lib.td_l1map_fwd(None, 0, four, None, 1, 3, 8, 8, 1.0, None, None) == -1
    assert lib.td_l1map_bwd(None, 0, four, None, None, 1, 3, 8, 8, 1.0, None, None) == -1
    assert lib.td_rgb2lab(None, 1, 8, 8, 50.0, 50.0, 110.0, None, None) == -1
    assert lib.td_pose_fwd(None, None, two, None, 2, 4, None, None, None) == -1
    assert lib.td_pose_bwd(None, None, two, None, 2, 4, None, None, None, None, None) == -1
    assert lib.td_color_jitter(None, None, 1, 8, 8, None, None, None, None) == -1
    assert lib.td_fp8_num_blocks(0) == 0 and lib.td_fp8_num_blocks(8192) == 1 and lib.td_fp8_num_blocks(8193) == 2
    assert lib.td_fp8_amax_partials(None, 1, 64, None, None) == -1
    assert lib.td_fp8_quantize(None, 1, 64, None, None, None, None) == -1
    assert lib.td_join_up2_fwd(None, None, None, 1, 1, 4, 4, 8, 8, 1, None, None) == -1
    assert lib.td_bn_sync_fwd_sums(None, 1, 64, 1, 64, None, None, None) == -1
    assert lib.td_adam_flat(None, None, None, None, None, 64, 0, None, None, 1e-3, 0.9, 0.999, 1e-8, None, 0.0, None) == -1
    assert lib.td_conv3x3_wgrad(None, None, 1, 8, 32, 64, 64, 1, 1, None, None, None) == -1 and lib.td_conv3x3_wgrad_workspace_floats(1, 8, 16, 64, 64) == 0
    assert lib.td_bias_act_fwd(None, None, 0, 1, 64, 16, 1, None, None) == -1 and lib.td_bias_act_workspace_floats(64, 12) == 0
    assert lib.td_bias_act_bwd(None, None, 1, 64, 16, 1, None, None, 0, None, None) == -1
    assert lib.td_maxpool5_bwd_add(None, None, None, 1, 1, 8, 8, 8, None, None) == -1
    assert lib.td_conv1x1_fwd_sum(None, None, 64, 64, 64, None, None, None, None) == -1
    assert lib.td_gather_flat(None, None, None, 0, 1, None, None) == -1
    assert lib.td_conv1x1_wgrad_group(1, None, None, None, None, None, None, None, None, None, None, None, None) == -1
    # round-4 entry points (fused bottleneck)
    assert lib.td_bn_partial_rows(0, 1, 64) == 0 and lib.td_bn_partial_rows(92160, 1, 64) >= 1
    assert lib.td_bn_partial_rows(1440, 1, 512) <= 16            # wide, short layers: the GEMM prologue finishes them directly
    assert lib.td_bn_fwd_partials(None, 1, 64, 1, 64, None, None) == -1
    assert lib.td_bn_bwd_partials(None, None, None, 1, None, None, None, None, 1, 64, 1, 64, None, None) == -1
    assert lib.td_bn_bwd_from_partials(None, None, None, 1, None, None, None, None, 1, 64, 1, 64, None, 1, None, None, None, None, None) == -1
    assert lib.td_conv1x1_fwd_bnrelu(None, None, 64, 1, 64, 64, None, 1, None, None, None, None, 0.1, 1e-5, None, None, None, None, None, None) == -1
    assert lib.td_conv1x1_dgrad(None, None, 64, 1, 64, 64, None, None, None) == -1
    assert lib.td_conv1x1_dgrad_bnsums(None, None, 64, 1, 64, 64, None, None, None, None, None, None, None, None) == -1
    assert lib.td_conv1x1_dgrad_bnbwd(None, None, None, 64, 1, 64, 64, None, 1, None, None, None, None, None, None, None, None, None, None) == -1
    assert lib.td_bn_sync_bwd_dx(None, None, None, 1, None, None, None, None, None, None, None, 0, 64, 1, 64,
                                 None, None, None, None, None, None) == -1


def test_product_fails_loudly_without_hip_device():
    """No silent CPU fallback: CPU tensors are refused by the product loss backend."""
    import torch
    import tripled_amd  # noqa: F401
    from tripled_amd import native, ops
    with pytest.raises(native.NativeLibraryError):
        ops.photo_identity(torch.rand(1, 3, 8, 8), [torch.rand(1, 3, 8, 8)])
    with pytest.raises(native.NativeLibraryError):
        native.load("/nonexistent/libtripled_hip.so") if native._lib is None else (_ for _ in ()).throw(
            native.NativeLibraryError("already loaded"))
