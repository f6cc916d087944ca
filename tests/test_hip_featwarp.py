"""GPU parity: fused feature-metric kernels vs the unfused torch composition (forward + all gradients)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import geometry, photometric  # noqa: E402
from tests.util import assert_argmin_parity, grad_close, kitti_K, random_poses, rel_err, smooth_image  # noqa: E402


def _reference(tgt_f, src_fs, disp, K, invK, Ts, h, w, forced=None):
    up = F.interpolate(disp, [h, w], mode="bilinear", align_corners=False)
    _, depth = geometry.disp_to_depth(up, 0.1, 100.0)
    pts = geometry.backproject(depth, invK)
    cands = []
    for s, T in zip(src_fs, Ts):
        grid = geometry.project(pts, K, T, h, w)
        warped = F.grid_sample(s, grid, mode="bilinear", padding_mode="border", align_corners=False)
        cands.append(photometric.perceptional_loss(tgt_f, warped))
    stack = torch.cat(cands, 1)
    vals, idx = torch.min(stack, dim=1)
    if forced is not None:       # gradients under the kernel's own selection: near-ties cannot change which frame gets gradient
        vals = torch.gather(stack, 1, forced.unsqueeze(1)).squeeze(1)
    return vals.mean(), idx, stack


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,h,w,hs,ws", [(2, 64, 12, 20, 12, 20), (1, 64, 9, 37, 9, 37), (2, 128, 6, 10, 3, 5),
                                           (12, 64, 96, 320, 96, 320), (4, 64, 160, 512, 160, 512)])   # C2 / C4 full size
def test_feature_metric(dtype, B, C, h, w, hs, ws):
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(2)
    base = torch.randn(B, C, h + 4, w + 4, generator=g)
    base = F.avg_pool2d(base, 3, 1, 1)                      # smooth features so the warp gradient is meaningful
    if w > 64:
        # full size: coordinates up to 512 carry 16x the rounding of the small cases (ulp(512) = 6e-5 px) and the steep
        # regime of the robust-L1 slope (|tgt - warped| < 1e-3) amplifies it by 1e3; larger feature magnitudes keep
        # the share of elements in that regime small, so that the comparison measures the kernel, not the rounding
        base = base * 8.0
    tgt = base[:, :, 2:2 + h, 2:2 + w].contiguous()
    srcs = [base[:, :, 2:2 + h, 1:1 + w].contiguous(), base[:, :, 3:3 + h, 2:2 + w].contiguous()]
    K, invK = kitti_K(B, h, w)
    Ts = random_poses(g, B, rot=0.005, trans=0.05)
    disp = (0.2 + 0.6 * smooth_image(g, B, 1, max(hs, 8), max(ws, 8))[:, :, :hs, :ws]).contiguous()
    cl = lambda t: t.to(dtype).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    tg, sg = cl(tgt), [cl(s) for s in srcs]
    dg = disp.cuda().requires_grad_(True)
    Tg = [T.cuda().requires_grad_(True) for T in Ts]
    P = torch.stack([torch.matmul(K.cuda(), T)[:, :3, :] for T in Tg], 0)
    loss, idx = ops.feature_warp_min_loss(tg, sg, dg, P, invK.cuda(), 0.1, 100.0)
    (loss * 2.0).backward()

    ref_in = lambda t: t.to(dtype).float().clone().requires_grad_(True)   # reference computes on .float()
    tr, sr = ref_in(tgt), [ref_in(s) for s in srcs]
    dr = disp.clone().requires_grad_(True)
    Tr = [T.clone().requires_grad_(True) for T in Ts]
    ref, ridx, stack = _reference(tr, sr, dr, K, invK, Tr, h, w, forced=idx.cpu().long())
    (ref * 2.0).backward()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert abs(float(loss) - float(ref)) < tol * max(1e-3, abs(float(ref)))
    # index work: exact except where the two frames' candidates tie to within the warp rounding (the robust-L1
    # slope turns ~3e-6 of coordinate rounding into ~1e-5 of the 64-channel mean; bf16 inputs are exact in both paths)
    assert_argmin_parity(idx, ridx, stack, tol=3e-5, max_frac=1e-2)
    # the robust-L1 slope d/dx sqrt(x^2 + 1e-6) changes by 1e3 per unit near 0: ~3e-6 warp rounding -> ~3e-3
    gt = 1e-2 if dtype == torch.float32 else 3e-2
    if w <= 64:
        assert rel_err(tg.grad.float(), tr.grad) < gt
        for a, r in zip(sg, sr):
            assert rel_err(a.grad.float(), r.grad) < gt
    else:
        # all but 0.2 % of the gradient elements within gt of the largest, every element within 0.5 (an element
        # whose robust-L1 argument is below eps = 1e-3 can move by a large part of its unit slope)
        grad_close(tg.grad.float(), tr.grad, gt, outlier_frac=2e-3, outlier_tol=0.5)
        for a, r in zip(sg, sr):
            grad_close(a.grad.float(), r.grad, gt, outlier_frac=2e-3, outlier_tol=0.5)
        grad_close(dg.grad, dr.grad, gt, outlier_frac=2e-3, outlier_tol=0.5)
    if w <= 64:
        assert rel_err(dg.grad, dr.grad) < max(gt, 5e-3)
    for a, r in zip(Tg, Tr):
        # (full size: the pose gradient sums the steep-regime elements of 10^5..10^6 pixels)
        assert rel_err(a.grad, r.grad) < (max(gt, 5e-3) if w <= 64 else 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_backward_is_bit_reproducible(dtype):
    """The source-feature gradient is a scatter with many contributions per element (here three near-identical frames warp onto
    each other and a coarse disparity makes neighbouring pixels land on the same source texels).  Since round 4 it accumulates
    in int32 fixed point (integer atomics commute), so two launches give the same bits -- with float atomics they did not --
    (the quantum is 2^-14 of the largest possible term, below the bf16 rounding of the result; the full-size cases of
    test_feature_metric pile thousands of border-clamped contributions onto corner texels without wrapping)."""
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(9)
    B, C, h, w = 4, 64, 48, 160
    base = F.avg_pool2d(torch.randn(B, C, h + 4, w + 4, generator=g), 3, 1, 1) * 4.0
    tgt = base[:, :, 2:2 + h, 2:2 + w].contiguous()
    srcs = [base[:, :, 2:2 + h, 1:1 + w].contiguous(), base[:, :, 3:3 + h, 2:2 + w].contiguous()]
    K, invK = kitti_K(B, h, w)
    Ts = random_poses(g, B, rot=0.01, trans=0.3)
    disp = (0.2 + 0.6 * smooth_image(g, B, 1, 8, 8)[:, :, :6, :20]).contiguous()
    cl = lambda t: t.to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    P = torch.stack([torch.matmul(K.cuda(), T.cuda())[:, :3, :] for T in Ts], 0)
    grads = []
    for _ in range(3):
        tg, sg = cl(tgt).requires_grad_(True), [cl(s).requires_grad_(True) for s in srcs]
        dg = disp.cuda().requires_grad_(True)
        loss, _ = ops.feature_warp_min_loss(tg, sg, dg, P, invK.cuda(), 0.1, 100.0)
        (loss * 3.0).backward()
        torch.cuda.synchronize()
        grads.append([tg.grad.clone(), dg.grad.clone()] + [s.grad.clone() for s in sg])
    for other in grads[1:]:
        for a, b in zip(grads[0], other):
            assert torch.equal(a, b)
    assert all(float(s.abs().max()) > 0 for s in grads[0][2:])
