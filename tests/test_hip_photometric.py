"""GPU parity: hand-written HIP photometric kernels (through the C ABI) vs the CPU oracle and
vs the golden vectors produced by the reference.  Tolerances are stated per assertion."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import geometry, photometric  # noqa: E402
from tests.util import assert_argmin_parity, kitti_K, make_triplet, random_poses, rel_err, smooth_image  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import tripled_amd  # noqa: F401
    from tripled_amd import ops as o
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return o


def _P(K, Ts):
    return torch.stack([torch.matmul(K, T)[:, :3, :] for T in Ts], 0).contiguous()


def _case(seed, B, H, W, hs, ws):
    g = torch.Generator().manual_seed(seed)
    fr = make_triplet(g, B, H, W)
    K, invK = kitti_K(B, H, W)
    Ts = random_poses(g, B)
    disp = (0.05 + 0.9 * smooth_image(g, B, 1, max(hs, 8), max(ws, 8))[:, :, :hs, :ws]).contiguous()
    noise = torch.randn(2, B, H, W, generator=g)
    return fr, K, invK, Ts, disp, noise


@pytest.mark.parametrize("B,H,W", [(2, 40, 70), (1, 16, 64), (3, 33, 129)])
def test_identity_term(ops, B, H, W):
    g = torch.Generator().manual_seed(1)
    fr = make_triplet(g, B, H, W)
    out = ops.photo_identity(fr[0].cuda(), [fr[-1].cuda(), fr[1].cuda()]).cpu()
    for i, f in enumerate((-1, 1)):
        ref = photometric.reprojection_loss(fr[f], fr[0])[:, 0]
        # SSIM variances cancel to ~1e-7 and are divided by C2 = 9e-4: 5e-5 absolute
        assert float((out[:, i] - ref).abs().max()) < 5e-5


@pytest.mark.parametrize("B,H,W,hs,ws", [(2, 48, 96, 24, 48), (2, 48, 96, 3, 6), (1, 37, 130, 19, 65),
                                         (2, 32, 64, 32, 64), (1, 20, 200, 5, 50)])
@pytest.mark.parametrize("automask", [True, False])
def test_forward_matches_oracle(ops, B, H, W, hs, ws, automask):
    fr, K, invK, Ts, disp, noise = _case(7, B, H, W, hs, ws)
    srcs = [fr[-1], fr[1]]
    tgt = fr[0]
    idl = ops.photo_identity(tgt.cuda(), [s.cuda() for s in srcs]) if automask else None
    loss, argmin, warped = ops.photometric_scale_loss(
        disp.cuda(), _P(K, Ts).cuda(), tgt.cuda(), [s.cuda() for s in srcs], invK.cuda(), idl,
        noise.cuda() if automask else None, 0.1, 100.0, 4, keep_warped=True)
    nz = [noise[0].unsqueeze(1), noise[1].unsqueeze(1)] if automask else None
    ref_loss, ref_idx, ref_warped = photometric.photometric_scale_loss(
        tgt, srcs, disp, K, invK, Ts, nz, 0.1, 100.0, automask=automask, n_scales=4)
    for i in range(2):
        # coordinate rounding (ulp(x) ~ 1.5e-5 at x = 200) times image slope
        assert float((warped[i].cpu() - ref_warped[i]).abs().max()) < 1e-4
    _, _, stack = photometric.min_reprojection(tgt, srcs, ref_warped, nz, automask)
    assert_argmin_parity(argmin, ref_idx, stack)       # exact except at fp near-ties (< 2e-5 apart)
    assert abs(float(loss) - float(ref_loss)) < 2e-6 + 1e-5 * abs(float(ref_loss))


@pytest.mark.parametrize("B,H,W,hs,ws", [(2, 48, 96, 24, 48), (2, 48, 96, 6, 12), (1, 37, 130, 19, 65),
                                         (1, 24, 70, 24, 70)])
@pytest.mark.parametrize("automask", [True, False])
def test_backward_matches_oracle_autograd(ops, B, H, W, hs, ws, automask):
    fr, K, invK, Ts, disp, noise = _case(9, B, H, W, hs, ws)
    srcs = [fr[-1], fr[1]]
    tgt = fr[0]
    idl = ops.photo_identity(tgt.cuda(), [s.cuda() for s in srcs]) if automask else None
    d = disp.cuda().requires_grad_(True)
    P = _P(K, Ts).cuda().requires_grad_(True)
    loss, argmin, _ = ops.photometric_scale_loss(d, P, tgt.cuda(), [s.cuda() for s in srcs], invK.cuda(), idl,
                                                 noise.cuda() if automask else None, 0.1, 100.0, 4)
    (loss * 3.0).backward()

    dr = disp.clone().requires_grad_(True)
    Pr = _P(K, Ts).requires_grad_(True)
    # same selection as the kernel so that near-ties cannot change which branch gets gradient
    forced = argmin.cpu().long()
    warped = []
    for i in range(2):
        up = geometry.upsample_bilinear(dr, H, W)
        _, depth = geometry.disp_to_depth(up, 0.1, 100.0)
        pts = geometry.backproject(depth, invK)
        cam = torch.matmul(Pr[i], pts)
        uv = cam[:, :2] / (cam[:, 2:3] + 1e-7)
        gx = (uv[:, 0].reshape(B, H, W) / (W - 1) - 0.5) * 2
        gy = (uv[:, 1].reshape(B, H, W) / (H - 1) - 0.5) * 2
        warped.append(geometry.grid_sample_border(srcs[i], torch.stack([gx, gy], -1)))
    nz = [noise[0].unsqueeze(1), noise[1].unsqueeze(1)] if automask else None
    vals, _, _ = photometric.min_reprojection(tgt, srcs, warped, nz, automask, forced_index=forced)
    (vals.mean() / 4 * 3.0).backward()
    # gradients: 2e-3 of the tensor's max magnitude (fp32 re-association in a different order)
    assert rel_err(d.grad, dr.grad) < 2e-3
    assert rel_err(P.grad, Pr.grad) < 2e-3


@pytest.mark.parametrize("scale", [0, 1, 2, 3])
def test_against_reference_golden(ops, golden_dir, scale):
    """Directly against vectors produced by the reference's own code (tools/gen_golden.py)."""
    z = np.load(os.path.join(golden_dir, "photo_scales.npz"))
    p = "s%d_" % scale
    T = lambda k: torch.from_numpy(np.ascontiguousarray(z[k]))
    tgt, srcs = T("color_0"), [T("color_-1"), T("color_1")]
    K, invK = T("K"), T("inv_K")
    Ts = [T("T_-1"), T("T_1")]
    Tc = [t.cuda().requires_grad_(True) for t in Ts]
    Kc = K.cuda()
    P = torch.stack([torch.matmul(Kc, t)[:, :3, :] for t in Tc], 0)
    d = T(p + "disp").cuda().requires_grad_(True)
    idl = ops.photo_identity(tgt.cuda(), [s.cuda() for s in srcs])
    noise = T(p + "noise")[:, :, 0].contiguous()
    loss, argmin, warped = ops.photometric_scale_loss(d, P, tgt.cuda(), [s.cuda() for s in srcs], invK.cuda(),
                                                      idl, noise.cuda(), 0.1, 100.0, 4, keep_warped=True)
    assert float((warped[0].cpu() - T(p + "warped_-1")).abs().max()) < 3e-5
    assert float((warped[1].cpu() - T(p + "warped_1")).abs().max()) < 3e-5
    # reference torch.min indices over the reference's own candidate stack (mono_fm_joint_inpaint/net.py:114-117)
    assert_argmin_parity(argmin, T(p + "min_index"), T(p + "cands"), max_frac=2e-3)
    assert abs(float(loss) - float(z[p + "loss"])) < 1e-6
    loss.backward()
    assert rel_err(d.grad, T(p + "d_disp")) < 5e-3
    assert rel_err(Tc[0].grad, T(p + "d_T_-1")) < 5e-3
    assert rel_err(Tc[1].grad, T(p + "d_T_1")) < 5e-3


def test_full_size_properties(ops):
    """BASELINE size (B=12, 192x640): determinism, loss parity with the oracle, and the
    identity-pose property (zero translation/rotation, constant disparity => every warped pixel
    is the bilinear sample at u*W/(W-1)-0.5)."""
    B, H, W = 12, 192, 640
    fr, K, invK, Ts, disp, noise = _case(3, B, H, W, 96, 320)
    srcs = [fr[-1].cuda(), fr[1].cuda()]
    tgt = fr[0].cuda()
    idl = ops.photo_identity(tgt, srcs)
    P = _P(K, Ts).cuda()
    run = lambda: ops.photometric_scale_loss(disp.cuda(), P, tgt, srcs, invK.cuda(), idl, noise.cuda(), 0.1, 100.0, 4)
    l1, a1, _ = run()
    l2, a2, _ = run()
    assert float(l1) == float(l2) and bool((a1 == a2).all())   # bit-reproducible
    nz = [noise[0].unsqueeze(1), noise[1].unsqueeze(1)]
    ref_loss, ref_idx, ref_warped = photometric.photometric_scale_loss(fr[0], [fr[-1], fr[1]], disp, K, invK, Ts, nz, 0.1, 100.0)
    assert abs(float(l1) - float(ref_loss)) < 1e-6
    _, _, stack = photometric.min_reprojection(fr[0], [fr[-1], fr[1]], ref_warped, nz, True)
    assert_argmin_parity(a1, ref_idx, stack)           # 1 474 560 pixels: exact except at fp near-ties
    eye = torch.eye(4).unsqueeze(0).repeat(B, 1, 1)
    Pid = _P(K, [eye, eye]).cuda()
    _, _, w = ops.photometric_scale_loss(torch.full((B, 1, 96, 320), 0.3).cuda(), Pid, tgt, srcs, invK.cuda(),
                                         None, None, 0.1, 100.0, 4, keep_warped=True)
    xs = torch.arange(W, dtype=torch.float32) * W / (W - 1) - 0.5
    ys = torch.arange(H, dtype=torch.float32) * H / (H - 1) - 0.5
    gx = (xs + 0.5) / W * 2 - 1
    gy = (ys + 0.5) / H * 2 - 1
    grid = torch.stack(torch.broadcast_tensors(gx.view(1, 1, W), gy.view(1, H, 1)), -1).repeat(B, 1, 1, 1)
    expect = geometry.grid_sample_border(fr[-1], grid)
    assert float((w[0].cpu() - expect).abs().max()) < 2e-4


def _oracle_backward(fr, K, invK, Ts, disp, noise, forced, B, H, W, scale_by=1.0):
    """Oracle autograd of one photometric scale under a given arg-min -> (d disp, d P)."""
    srcs = [fr[-1], fr[1]]
    dr = disp.clone().requires_grad_(True)
    Pr = _P(K, Ts).requires_grad_(True)
    warped = []
    for i in range(2):
        up = geometry.upsample_bilinear(dr, H, W)
        _, depth = geometry.disp_to_depth(up, 0.1, 100.0)
        pts = geometry.backproject(depth, invK)
        cam = torch.matmul(Pr[i], pts)
        uv = cam[:, :2] / (cam[:, 2:3] + 1e-7)
        gx = (uv[:, 0].reshape(B, H, W) / (W - 1) - 0.5) * 2
        gy = (uv[:, 1].reshape(B, H, W) / (H - 1) - 0.5) * 2
        warped.append(geometry.grid_sample_border(srcs[i], torch.stack([gx, gy], -1)))
    nz = [noise[0].unsqueeze(1), noise[1].unsqueeze(1)]
    vals, _, _ = photometric.min_reprojection(fr[0], srcs, warped, nz, True, forced_index=forced)
    loss = vals.mean() / 4 * scale_by
    loss.backward()
    return float(loss), dr.grad, Pr.grad


def test_full_size_backward_matches_oracle(ops):
    """BASELINE size (B=12, 192x640, scale 1): the shape at which the wave tasks are 13/14 rows tall with a
    ragged last chunk (td_common.h::pick_rows) -- gradients w.r.t. the disparity and the projection matrices
    against the oracle's autograd, under the kernel's own arg-min (as in test_backward_matches_oracle_autograd)."""
    B, H, W, hs, ws = 12, 192, 640, 48, 160
    fr, K, invK, Ts, disp, noise = _case(11, B, H, W, hs, ws)
    tgt = fr[0].cuda()
    srcs = [fr[-1].cuda(), fr[1].cuda()]
    idl = ops.photo_identity(tgt, srcs)
    d = disp.cuda().requires_grad_(True)
    P = _P(K, Ts).cuda().requires_grad_(True)
    loss, amin, _ = ops.photometric_scale_loss(d, P, tgt, srcs, invK.cuda(), idl, noise.cuda(), 0.1, 100.0, 4)
    loss.backward()
    ref_loss, d_ref, P_ref = _oracle_backward(fr, K, invK, Ts, disp, noise, amin.cpu().long(), B, H, W)
    assert abs(float(loss) - ref_loss) < 2e-6
    assert rel_err(d.grad, d_ref) < 2e-3
    assert rel_err(P.grad, P_ref) < 2e-3


@pytest.mark.parametrize("n_src", [1, 3, 4])
@pytest.mark.parametrize("automask", [True, False])
def test_other_source_counts(ops, n_src, automask):
    """frame_ids with one source ([0, 's']), three ([0, -1, 1, 's']) and the ABI's maximum of four: forward (warps, exact
    arg-min, loss) and backward (d disp, d P) of the template instantiations the two-source tests never reach."""
    B, H, W, hs, ws = 2, 40, 72, 20, 36
    g = torch.Generator().manual_seed(20 + n_src)
    base = smooth_image(g, B, 3, H + 8, W + 8)
    tgt = base[:, :, 4:4 + H, 4:4 + W].contiguous()
    srcs = [(base[:, :, 4 + (i % 2):4 + (i % 2) + H, 3 + i:3 + i + W] + 0.01 * torch.randn(B, 3, H, W, generator=g)).clamp(0, 1).contiguous()
            for i in range(n_src)]
    K, invK = kitti_K(B, H, W)
    Ts = []
    for i in range(n_src):
        axis = 0.01 * torch.randn(B, 1, 3, generator=g)
        t = 0.15 * torch.randn(B, 1, 3, generator=g)
        Ts.append(geometry.transformation_from_parameters(axis, t[:, 0], invert=(i % 2 == 0)))
    disp = (0.05 + 0.9 * smooth_image(g, B, 1, hs, ws)).contiguous()
    noise = torch.randn(n_src, B, H, W, generator=g)
    cs = [s.cuda() for s in srcs]
    idl = ops.photo_identity(tgt.cuda(), cs) if automask else None
    d = disp.cuda().requires_grad_(True)
    P = _P(K, Ts).cuda().requires_grad_(True)
    loss, argmin, warped = ops.photometric_scale_loss(d, P, tgt.cuda(), cs, invK.cuda(), idl,
                                                      noise.cuda() if automask else None, 0.1, 100.0, 4, keep_warped=True)
    nz = [noise[i].unsqueeze(1) for i in range(n_src)] if automask else None
    ref_loss, ref_idx, ref_warped = photometric.photometric_scale_loss(tgt, srcs, disp, K, invK, Ts, nz, 0.1, 100.0,
                                                                       automask=automask, n_scales=4)
    for i in range(n_src):
        assert float((warped[i].cpu() - ref_warped[i]).abs().max()) < 1e-4
    _, _, stack = photometric.min_reprojection(tgt, srcs, ref_warped, nz, automask)
    assert stack.shape[1] == (2 * n_src if automask else n_src)
    assert_argmin_parity(argmin, ref_idx, stack)
    assert abs(float(loss) - float(ref_loss)) < 2e-6 + 1e-5 * abs(float(ref_loss))
    (loss * 3.0).backward()
    dr = disp.clone().requires_grad_(True)
    Pr = _P(K, Ts).requires_grad_(True)
    rw = []
    for i in range(n_src):
        up = geometry.upsample_bilinear(dr, H, W)
        _, depth = geometry.disp_to_depth(up, 0.1, 100.0)
        cam = torch.matmul(Pr[i], geometry.backproject(depth, invK))
        uv = cam[:, :2] / (cam[:, 2:3] + 1e-7)
        gx = (uv[:, 0].reshape(B, H, W) / (W - 1) - 0.5) * 2
        gy = (uv[:, 1].reshape(B, H, W) / (H - 1) - 0.5) * 2
        rw.append(geometry.grid_sample_border(srcs[i], torch.stack([gx, gy], -1)))
    vals, _, _ = photometric.min_reprojection(tgt, srcs, rw, nz, automask, forced_index=argmin.cpu().long())
    (vals.mean() / 4 * 3.0).backward()
    assert rel_err(d.grad, dr.grad) < 2e-3
    assert rel_err(P.grad, Pr.grad) < 2e-3


@pytest.mark.parametrize("B,H,W,scale", [(4, 320, 1024, 0), (4, 320, 1024, 3), (8, 192, 640, 0)])
def test_config_shapes_forward_and_backward(ops, B, H, W, scale):
    """The photometric kernels at the shapes of BASELINE configs 4 (cfg_kitti_tripleD at 320x1024, 4 images per GPU: 17 column
    strips, row tiling picked by td_common.h::pick_rows for THAT shape) and 5 (8 images per GPU, 192x640), scales 0 and 3:
    loss to 2e-6, EXACT arg-min against the oracle (near-ties only), gradients w.r.t. the disparity and the projection
    matrices to 2e-3 of their maximum under the kernel's own arg-min (disparity: outlier-aware, see below)."""
    hs, ws = H >> (scale + 1), W >> (scale + 1)
    fr, K, invK, Ts, disp, noise = _case(31 + scale, B, H, W, hs, ws)
    tgt = fr[0].cuda()
    srcs = [fr[-1].cuda(), fr[1].cuda()]
    idl = ops.photo_identity(tgt, srcs)
    d = disp.cuda().requires_grad_(True)
    P = _P(K, Ts).cuda().requires_grad_(True)
    loss, amin, _ = ops.photometric_scale_loss(d, P, tgt, srcs, invK.cuda(), idl, noise.cuda(), 0.1, 100.0, 4)
    loss.backward()
    nz = [noise[0].unsqueeze(1), noise[1].unsqueeze(1)]
    ref_loss, ref_idx, ref_warped = photometric.photometric_scale_loss(fr[0], [fr[-1], fr[1]], disp, K, invK, Ts, nz, 0.1, 100.0)
    assert abs(float(loss) - float(ref_loss)) < 2e-6
    _, _, stack = photometric.min_reprojection(fr[0], [fr[-1], fr[1]], ref_warped, nz, True)
    assert_argmin_parity(amin, ref_idx, stack)
    _, d_ref, P_ref = _oracle_backward(fr, K, invK, Ts, disp, noise, amin.cpu().long(), B, H, W)
    # 1.3 - 1.5 million pixels: a few land where the robust-L1 argument is below its eps (d/dx sqrt(x^2 + 1e-6) ~ x / 1e-3 turns
    # 1e-5 of coordinate rounding into 1e-2 of gradient, tests/util.py::grad_close): all but 0.2 % of the disparity gradient
    # within 2e-3 of its maximum, every element within 5e-2; the pose gradient (a sum over all pixels) within 1e-2
    from tests.util import grad_close
    grad_close(d.grad, d_ref, 2e-3)
    # the same outlier pixels (plus those whose sample coordinate rounds across an integer: the bilinear kernel's slope jumps
    # there) enter this 1.3-million-term sum: 1e-2 of the largest element (measured 2.6e-3 / 5.7e-3 / < 2e-3)
    assert rel_err(P.grad, P_ref) < 1e-2


def test_identity_kernel_emits_the_rgbx_frames(ops):
    """td_photo_identity writes the RGBX copies of the frames it reads (the per-scale kernels' input) as a by-product: they must
    equal td_pack_rgbx's, pixel for pixel, for every frame, and the loss computed from them must equal the loss from packed frames."""
    B, H, W = 3, 37, 130
    g = torch.Generator().manual_seed(4)
    fr = make_triplet(g, B, H, W)
    tgt, srcs = fr[0].cuda(), [fr[-1].cuda(), fr[1].cuda()]
    a = ops.pack_frames(tgt, srcs, pack=True)
    b = ops.pack_frames(tgt, srcs, pack=False)
    assert not b.packed
    idl = ops.photo_identity(b)
    assert b.packed
    assert torch.equal(a.tgt, b.tgt) and all(torch.equal(x, y) for x, y in zip(a.srcs, b.srcs))
    want = torch.cat([tgt, torch.zeros(B, 1, H, W, device="cuda")], 1).permute(0, 2, 3, 1)
    assert torch.equal(b.tgt, want)
    assert torch.equal(idl, ops.photo_identity(a))
