"""Pins the CPU oracle (oracle/) against vectors produced by running the
reference's own modules (tools/gen_golden.py -> tests/golden/*.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import geometry, photometric, smooth, metrics, losses


class Opt(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, atol, rtol=0.0):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def test_ops_geometry(golden_dir):
    z = load(golden_dir, "ops_small.npz")
    B, _, H, W = z["depth"].shape
    pts = geometry.backproject(T(z["depth"]), T(z["inv_K"]))
    close(pts, z["cam_points"], atol=1e-5, rtol=1e-6)
    grid = geometry.project(T(z["cam_points"]), T(z["K"]), T(z["T_fwd"]), H, W)
    close(grid, z["grid"], atol=2e-6)
    warped = geometry.grid_sample_border(T(z["img"]), T(z["grid"]))
    close(warped, z["warped"], atol=2e-6)
    up = geometry.upsample_bilinear(T(z["disp_lr"]), H, W)
    close(up, z["disp_up"], atol=1e-6)
    _, depth = geometry.disp_to_depth(T(z["disp_up"]), 0.1, 100.0)
    close(depth, z["depth_from_disp"], atol=0, rtol=1e-6)


def test_ops_pose(golden_dir):
    z = load(golden_dir, "ops_small.npz")
    close(geometry.rot_from_axisangle(T(z["axisangle"])), z["rot"], atol=1e-7)
    close(geometry.transformation_from_parameters(T(z["axisangle"]), T(z["translation"])[:, 0], False),
          z["T_fwd"], atol=1e-7)
    close(geometry.transformation_from_parameters(T(z["axisangle"]), T(z["translation"])[:, 0], True),
          z["T_inv"], atol=1e-7)
    eye = geometry.rot_from_axisangle(torch.zeros(3, 1, 3))
    close(eye, np.tile(np.eye(4, dtype=np.float32), (3, 1, 1)), atol=0)


def test_color_lab_and_l1_map(golden_dir):
    from oracle import color
    z = load(golden_dir, "color_lab.npz")
    close(color.rgb2lab(T(z["rgb"])), z["lab"], atol=2e-6)
    pred = T(z["pred"]).requires_grad_(True)
    m = color.robust_l1_map(pred, T(z["rgb"]), 5e-3)
    close(m, z["l1map"], atol=1e-9)
    m.mean().backward()
    close(pred.grad, z["d_pred"], atol=1e-10)


def test_ops_ssim_and_smooth(golden_dir):
    z = load(golden_dir, "ops_small.npz")
    close(photometric.ssim_loss(T(z["ssim_x"]), T(z["ssim_y"])), z["ssim"], atol=1e-6)
    close(photometric.ssim_loss(T(z["ssim_x"]), T(z["ssim_x"])), z["ssim_same"], atol=1e-6)
    assert float(np.abs(z["ssim_same"]).max()) < 1e-4  # SSIM(x, x) == 0 up to rounding
    close(photometric.reprojection_loss(T(z["ssim_x"]), T(z["ssim_y"])), z["reproj"], atol=1e-6)
    close(smooth.smooth_loss(T(z["smooth_disp"]), T(z["img"])), z["smooth"], atol=1e-7)
    close(smooth.feature_regularization_loss(T(z["feat"]), T(z["img"]), 1e-3, 1e-3), z["freg"], atol=1e-8)
    # known answers
    x = T(z["ssim_x"])
    close(photometric.robust_l1(x, x), np.full(x.shape, 1e-3, np.float32), atol=1e-9)
    assert float(smooth.smooth_loss(torch.full((1, 1, 8, 8), 0.3), T(z["img"])[:1])) == 0.0


@pytest.mark.parametrize("scale", [0, 1, 2, 3])
def test_photometric_scale(golden_dir, scale):
    z = load(golden_dir, "photo_scales.npz")
    p = "s%d_" % scale
    target = T(z["color_0"])
    srcs = [T(z["color_-1"]), T(z["color_1"])]
    disp = T(z[p + "disp"]).requires_grad_(True)
    Ts = [T(z["T_-1"]).requires_grad_(True), T(z["T_1"]).requires_grad_(True)]
    noise = [T(z[p + "noise"][0]), T(z[p + "noise"][1])]
    loss, idx, warped = photometric.photometric_scale_loss(
        target, srcs, disp, T(z["K"]), T(z["inv_K"]), Ts, noise, 0.1, 100.0, automask=True, n_scales=4)
    close(warped[0], z[p + "warped_-1"], atol=3e-5)  # coordinate ulp (~8e-6 px at x=64) times image slope
    close(warped[1], z[p + "warped_1"], atol=3e-5)  # coordinate ulp (~8e-6 px at x=64) times image slope
    _, _, stack = photometric.min_reprojection(target, srcs, warped, noise, True)
    close(stack, z[p + "cands"], atol=5e-5)  # SSIM amplifies 1e-6 warp differences in flat windows (C2 = 9e-4)
    assert (idx.numpy() == z[p + "min_index"]).mean() > 0.999
    close(loss, z[p + "loss"], atol=1e-7)
    loss.backward()
    close(disp.grad, z[p + "d_disp"], atol=2e-7, rtol=1e-3)
    close(Ts[0].grad, z[p + "d_T_-1"], atol=2e-6, rtol=1e-3)
    close(Ts[1].grad, z[p + "d_T_1"], atol=2e-6, rtol=1e-3)


class _Stub(torch.nn.Module):
    def __init__(self, state):
        super().__init__()
        n = len({k.split(".")[1] for k in state})
        self.convs = torch.nn.ModuleList()
        for i in range(n):
            w = state["convs.%d.weight" % i]
            self.convs.append(torch.nn.Conv2d(w.shape[1], w.shape[0], 3, 2, 1))
        self.load_state_dict({k: T(v) for k, v in state.items()})

    def forward(self, x):
        out = []
        for c in self.convs:
            x = torch.tanh(c(x))
            out.append(x)
        return out


def _inputs_from(z, with_mask=True):
    inputs = {("color", 0, 0): T(z["color_0"]), ("color", -1, 0): T(z["color_-1"]), ("color", 1, 0): T(z["color_1"]),
              "K": T(z["K"]), "inv_K": T(z["inv_K"])}
    if with_mask:
        inputs[("mask", 0, 0)] = T(z["mask_0"])
    return inputs


def test_compute_losses_tripled(golden_dir):
    z = load(golden_dir, "losses_tripled.npz")
    B, _, H, W = z["color_0"].shape
    opt = Opt(frame_ids=[0, -1, 1], imgs_per_gpu=B, height=H, width=W, scales=[0, 1, 2, 3], min_depth=0.1,
              max_depth=100.0, automask=True, disp_norm=True, dis=1e-3, cvt=1e-3, perception_weight=1e-3,
              smoothness_weight=1e-3, auto_res_weight=5e-3)
    stub = _Stub({k[5:]: v for k, v in z.items() if k.startswith("stub_")})
    inputs = _inputs_from(z)
    leaves = {k[3:]: T(v).requires_grad_(True) for k, v in z.items() if k.startswith("in_")}
    outputs = {}
    for s in range(4):
        outputs[("disp", 0, s)] = leaves["disp_%d" % s]
        outputs[("res_img", 0, s)] = leaves["res_img_%d" % s]
    outputs[("auto_res_img", 0, 0)] = leaves["auto_res_img_0"]
    for f in (-1, 1):
        outputs[("cam_T_cam", 0, f)] = leaves["T_%d" % f]
    features = [leaves["feat_%d" % i] for i in range(5)]
    noise = losses.NoiseSource([T(n) for n in z["noise"]])
    loss_dict, extra = losses.compute_losses_disentangle(opt, inputs, outputs, features, stub, noise)
    ref_keys = [k[5:] for k in z if k.startswith("loss_")]
    assert [str(k) for k in loss_dict.keys()] == ref_keys  # same keys, same order
    for k, v in loss_dict.items():
        close(v, z["loss_" + str(k)], atol=2e-7, rtol=2e-5)
    total = losses.total_loss(loss_dict)
    close(total, z["total"], atol=1e-6)
    total.backward()
    for n, t in leaves.items():
        g = z["grad_" + n]
        if g.size == 0:
            assert t.grad is None
            continue
        close(t.grad, g, atol=3e-7, rtol=2e-3)
    for n, p in stub.named_parameters():
        if "stubgrad_" + n in z:
            close(p.grad, z["stubgrad_" + n], atol=1e-6, rtol=2e-3)
    for s in range(4):
        assert (extra[("min_index", s)].numpy() == z["min_index_%d" % s]).mean() > 0.999
    assert (extra["min_index"].numpy() == z["perc_min_index"]).mean() > 0.999


def test_compute_losses_fm(golden_dir):
    z = load(golden_dir, "losses_fm.npz")
    B, _, H, W = z["color_0"].shape
    opt = Opt(frame_ids=[0, -1, 1], imgs_per_gpu=B, height=H, width=W, scales=[0, 1, 2, 3], min_depth=0.1,
              max_depth=100.0, automask=True, disp_norm=True, perception_weight=1e-3, smoothness_weight=1e-3)
    stub = _Stub({k[5:]: v for k, v in z.items() if k.startswith("stub_")})
    inputs = _inputs_from(z, with_mask=False)
    leaves = {k[3:]: T(v).requires_grad_(True) for k, v in z.items() if k.startswith("in_")}
    outputs = {("disp", 0, s): leaves["disp_%d" % s] for s in range(4)}
    for f in (-1, 1):
        outputs[("cam_T_cam", 0, f)] = leaves["T_%d" % f]
    noise = losses.NoiseSource([T(n) for n in z["noise"]])
    loss_dict, _ = losses.compute_losses_fm(opt, inputs, outputs, stub, noise)
    assert [str(k) for k in loss_dict.keys()] == [k[5:] for k in z if k.startswith("loss_")]
    for k, v in loss_dict.items():
        close(v, z["loss_" + str(k)], atol=2e-7, rtol=2e-5)
    total = losses.total_loss(loss_dict)
    total.backward()
    for n, t in leaves.items():
        close(t.grad, z["grad_" + n], atol=3e-7, rtol=2e-3)
    for n, p in stub.named_parameters():
        if "stubgrad_" + n in z:
            close(p.grad, z["stubgrad_" + n], atol=1e-6, rtol=2e-3)


def test_metrics(golden_dir):
    z = load(golden_dir, "metrics.npz")
    np.testing.assert_allclose(metrics.compute_errors(z["gt"], z["pred"]), z["errors"], rtol=1e-6)
    np.testing.assert_allclose(metrics.compute_errors(z["gt"], z["gt"]), [0, 0, 0, 0, 1, 1, 1], atol=0)
    sd, dp = metrics.disp_to_depth(z["disp"])
    np.testing.assert_allclose(sd, z["scaled_disp"], rtol=1e-6)
    np.testing.assert_allclose(dp, z["depth"], rtol=1e-6)
    sd0, dp0 = metrics.disp_to_depth(np.array([0.0, 1.0]))
    np.testing.assert_allclose(sd0, [0.01, 10.0])
    np.testing.assert_allclose(dp0, [100.0, 0.1])
    # eval protocol: a prediction that is an exact multiple of the ground truth is fully
    # repaired by median scaling
    gt = np.random.RandomState(0).uniform(2, 60, size=(40, 120)).astype(np.float32)
    errs, ratio = metrics.eval_single((1.0 / (gt * 0.5)).astype(np.float32), gt)
    assert abs(ratio - 2.0) < 1e-4 and errs[0] < 1e-5 and errs[4] == 1.0
