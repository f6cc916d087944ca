"""GPU parity of the small fused maps (csrc/td_pointwise.hip, csrc/td_pose.hip) through the C ABI: against vectors
produced by the reference's own functions (tests/golden/color_lab.npz, ops_small.npz) and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import color, geometry  # noqa: E402
from tests.util import kitti_K, rel_err, smooth_image  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import tripled_amd  # noqa: F401
    from tripled_amd import ops as o
    return o


def _T(z, k):
    return torch.from_numpy(np.ascontiguousarray(z[k]))


def test_rgb2lab_golden(ops, golden_dir):
    z = np.load(os.path.join(golden_dir, "color_lab.npz"))
    lab = ops.rgb2lab(_T(z, "rgb").cuda()).cpu()
    # powf on the device vs the reference's torch.pow on the host: a few ulp of values of magnitude <= 1
    assert float((lab - _T(z, "lab")).abs().max()) < 2e-6


def test_rgb2lab_full_size_vs_oracle(ops):
    g = torch.Generator().manual_seed(2)
    rgb = torch.rand(8, 3, 192, 640, generator=g)
    rgb[:, :, :4] = rgb[:, :, :4] * 0.05             # around the 0.04045 / 0.008856 branch thresholds
    lab = ops.rgb2lab(rgb.cuda()).cpu()
    ref = color.rgb2lab(rgb)
    assert float((lab - ref).abs().max()) < 2e-6


def test_l1_map_golden(ops, golden_dir):
    z = np.load(os.path.join(golden_dir, "color_lab.npz"))
    pred = _T(z, "pred").cuda().requires_grad_(True)
    m = ops.robust_l1_map(pred, _T(z, "rgb").cuda(), 5e-3)
    assert float((m.cpu() - _T(z, "l1map")).abs().max()) < 1e-9
    m.mean().backward()
    assert rel_err(pred.grad, _T(z, "d_pred")) < 1e-5


@pytest.mark.parametrize("layout", ["nchw_f32", "nhwc_bf16", "nhwc_bf16_slice"])
@pytest.mark.parametrize("B,C,H,W", [(2, 3, 24, 40), (3, 2, 17, 33), (12, 3, 192, 640)])
def test_l1_map_layouts_vs_oracle(ops, layout, B, C, H, W):
    """The decoder outputs arrive as channels-last bf16 tensors (and as channel slices of 8-channel convolution
    outputs): read in place, gradient written in the same dtype and layout."""
    g = torch.Generator().manual_seed(3)
    target = torch.rand(B, C, H, W, generator=g)
    if layout == "nchw_f32":
        pred = torch.rand(B, C, H, W, generator=g).cuda()
    elif layout == "nhwc_bf16":
        pred = torch.rand(B, C, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    else:
        wide = torch.rand(B, 8, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        pred = torch.sigmoid(wide[:, :C])                # what Conv3x3 + sigmoid hand to the loss
    pred = pred.detach().requires_grad_(True)
    m = ops.robust_l1_map(pred, target.cuda(), 0.7)
    up = torch.rand(B, 1, H, W, generator=g)
    (m * up.cuda()).sum().backward()
    pr = pred.detach().float().cpu().requires_grad_(True)
    ref = color.robust_l1_map(pr, target, 0.7)
    (ref * up).sum().backward()
    assert float((m.cpu() - ref).abs().max()) < 1e-6
    assert pred.grad.dtype == pred.dtype and pred.grad.stride() == pred.stride()
    tol = 1e-5 if pred.dtype == torch.float32 else 8e-3      # bf16 gradient: 8 significand bits
    assert rel_err(pred.grad.float(), pr.grad) < tol


def test_pose_transforms_golden(ops, golden_dir):
    """transformation_from_parameters of the reference (ops_small.npz: T_fwd / T_inv for the same vectors)."""
    z = np.load(os.path.join(golden_dir, "ops_small.npz"))
    a, t = _T(z, "axisangle"), _T(z, "translation")          # [B,1,3]
    B = a.shape[0]
    K, _ = kitti_K(B, 192, 640)
    T, P = ops.pose_transforms(torch.cat([a, a], 0).cuda(), torch.cat([t, t], 0).cuda(), K.cuda(), [True, False])
    assert float((T[0].cpu() - _T(z, "T_inv")).abs().max()) < 1e-6
    assert float((T[1].cpu() - _T(z, "T_fwd")).abs().max()) < 1e-6
    for i, key in enumerate(("T_inv", "T_fwd")):
        ref_P = torch.matmul(K, _T(z, key))[:, :3, :]
        assert rel_err(P[i], ref_P) < 1e-6


@pytest.mark.parametrize("B", [1, 12, 70])
def test_pose_transforms_gradients_vs_oracle(ops, B):
    g = torch.Generator().manual_seed(B)
    a = (0.05 * torch.randn(2 * B, 1, 3, generator=g))
    t = (0.3 * torch.randn(2 * B, 1, 3, generator=g))
    a[0] = 0.0                                               # zero rotation: torch.norm's subgradient is 0 there
    K, _ = kitti_K(B, 192, 640)
    gT, gP = torch.randn(2, B, 4, 4, generator=g), torch.randn(2, B, 3, 4, generator=g)
    ad, td = a.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    T, P = ops.pose_transforms(ad, td, K.cuda(), [True, False])
    ((T * gT.cuda()).sum() + (P * gP.cuda()).sum()).backward()
    ar, tr = a.clone().requires_grad_(True), t.clone().requires_grad_(True)
    Ts = [geometry.transformation_from_parameters(ar[i * B:(i + 1) * B], tr[i * B:(i + 1) * B, 0], invert=(i == 0))
          for i in range(2)]
    loss = sum((Ts[i] * gT[i]).sum() + (torch.matmul(K, Ts[i])[:, :3, :] * gP[i]).sum() for i in range(2))
    loss.backward()
    for i in range(2):
        assert float((T[i].cpu() - Ts[i]).abs().max()) < 1e-6
    assert rel_err(ad.grad, ar.grad) < 1e-4
    assert rel_err(td.grad, tr.grad) < 1e-5
    assert bool(torch.isfinite(ad.grad).all())
