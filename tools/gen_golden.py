#!/usr/bin/env python3
"""Generate golden vectors by importing and running the REFERENCE's own Python
modules on CPU (fp32, seeded).  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--ref /root/reference]

Outputs small .npz fixtures under tests/golden/.  Fixtures are data only (inputs
and the reference's outputs); no reference source is copied.  The reference is
never shipped to the GPU box -- the fixtures are.

Import recipe: `import mono.model` fails in the reference
(missing segmentation_base, torchvision absent), so empty stub packages `mono`
and `mono.model` are registered with __path__ pointing into the reference tree,
an inert `torchvision` stub is provided, and Tensor.cuda() is made a no-op
because the reference hard-codes .cuda() calls.
"""
import argparse
import importlib
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")


class AttrDict(dict):
    """attr + item access + .get + item assignment (what the models need of mmcv.Config)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def install_reference(ref):
    for name, sub in (("mono", "mono"), ("mono.model", "mono/model")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(ref, sub)]
        sys.modules[name] = m
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.transforms = types.ModuleType("torchvision.transforms")
        tv.transforms.Grayscale = object
        tv.transforms.RandomCrop = object
        tv.transforms.functional = types.ModuleType("torchvision.transforms.functional")
        tv.models = types.ModuleType("torchvision.models")
        tv.models.utils = types.ModuleType("torchvision.models.utils")
        tv.models.utils.load_state_dict_from_url = None
        for k, v in (("torchvision", tv), ("torchvision.transforms", tv.transforms),
                     ("torchvision.transforms.functional", tv.transforms.functional),
                     ("torchvision.models", tv.models), ("torchvision.models.utils", tv.models.utils)):
            sys.modules[k] = v
    torch.Tensor.cuda = lambda self, *a, **k: self


class NoiseTap:
    """Record every torch.randn draw made while active."""

    def __init__(self):
        self.draws = []
        self._orig = torch.randn

    def __enter__(self):
        def tapped(*a, **k):
            t = self._orig(*a, **k)
            self.draws.append(t.clone())
            return t
        torch.randn = tapped
        return self

    def __exit__(self, *exc):
        torch.randn = self._orig


def bare(cls):
    """Instance of a reference model class without building its networks (methods only)."""
    obj = object.__new__(cls)
    nn.Module.__init__(obj)
    return obj


def kitti_K(b, h, w):
    K = np.array([[0.58 * w, 0, 0.5 * w, 0],
                  [0, 1.92 * h, 0.5 * h, 0],
                  [0, 0, 1, 0],
                  [0, 0, 0, 1]], dtype=np.float32)
    inv_K = np.linalg.pinv(K)
    K = torch.from_numpy(K).unsqueeze(0).repeat(b, 1, 1)
    inv_K = torch.from_numpy(inv_K).unsqueeze(0).repeat(b, 1, 1)
    return K, inv_K


def smooth_image(g, b, c, h, w):
    """Low-pass random image in [0,1] so SSIM windows are not pure noise."""
    x = torch.rand(b, c, h // 4 + 2, w // 4 + 2, generator=g)
    x = F.interpolate(x, size=(h, w), mode="bicubic", align_corners=False)
    x = x + 0.05 * torch.rand(b, c, h, w, generator=g)
    return x.clamp(0, 1).contiguous()


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB, %d arrays)" % (path, os.path.getsize(path) / 1024, len(arrays)))


# ---------------------------------------------------------------------------


def gen_ops(ref_layers, ref_net_cls):
    """Per-op I/O of Backproject, Project, grid_sample, SSIM, upsample, smoothness, pose."""
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 24, 40
    K, inv_K = kitti_K(B, H, W)
    depth = 0.5 + 20 * torch.rand(B, 1, H, W, generator=g)
    bp = ref_layers.Backproject(B, H, W)
    pj = ref_layers.Project(B, H, W)
    pts = bp(depth, inv_K)

    net = bare(ref_net_cls)
    axis = 0.05 * torch.randn(B, 1, 3, generator=g)
    trans = 0.3 * torch.randn(B, 1, 3, generator=g)
    T_fwd = ref_net_cls.transformation_from_parameters(net, axis, trans, invert=False)
    T_inv = ref_net_cls.transformation_from_parameters(net, axis, trans, invert=True)
    rot = ref_net_cls.rot_from_axisangle(net, axis)
    grid = pj(pts, K, T_fwd)
    img = smooth_image(g, B, 3, H, W)
    warped = F.grid_sample(img, grid, padding_mode="border")

    x = smooth_image(g, B, 3, H, W)
    y = (x + 0.1 * torch.randn(B, 3, H, W, generator=g)).clamp(0, 1)
    ssim = ref_layers.SSIM()(x, y)
    ssim_same = ref_layers.SSIM()(x, x)
    net.ssim = ref_layers.SSIM()
    reproj = ref_net_cls.compute_reprojection_loss(net, x, y)

    disp_lr = torch.rand(B, 1, H // 4, W // 4, generator=g)
    up = F.interpolate(disp_lr, [H, W], mode="bilinear", align_corners=False)
    _, depth_from_disp = ref_net_cls.disp_to_depth(net, up, 0.1, 100.0)

    disp = torch.rand(B, 1, H // 2, W // 2, generator=g)
    smooth = ref_net_cls.get_smooth_loss(net, disp, img)
    feat = torch.randn(B, 6, H // 4, W // 4, generator=g)
    net.opt = AttrDict(dis=1e-3, cvt=1e-3)
    freg = ref_net_cls.get_feature_regularization_loss(net, feat, img)

    save("ops_small.npz",
         K=npy(K), inv_K=npy(inv_K), depth=npy(depth), cam_points=npy(pts),
         axisangle=npy(axis), translation=npy(trans), T_fwd=npy(T_fwd), T_inv=npy(T_inv), rot=npy(rot),
         grid=npy(grid), img=npy(img), warped=npy(warped),
         ssim_x=npy(x), ssim_y=npy(y), ssim=npy(ssim), ssim_same=npy(ssim_same), reproj=npy(reproj),
         disp_lr=npy(disp_lr), disp_up=npy(up), depth_from_disp=npy(depth_from_disp),
         smooth_disp=npy(disp), smooth=npy(smooth), feat=npy(feat), freg=npy(freg))


def make_batch(g, B, H, W, erase=4, erase_hw=(6, 6)):
    base = smooth_image(g, B, 3, H + 8, W + 8)
    inputs = {}
    for f, (dy, dx) in ((0, (4, 4)), (-1, (4, 2)), (1, (5, 6))):
        img = base[:, :, dy:dy + H, dx:dx + W].contiguous()
        img = (img + 0.01 * torch.randn(B, 3, H, W, generator=g)).clamp(0, 1)
        inputs[("color", f, 0)] = img
        inputs[("color_aug", f, 0)] = img.clone()
    mask = torch.ones(B, 3, H, W)
    for b in range(B):
        for _ in range(erase):
            y0 = int(torch.randint(0, H - erase_hw[0], (1,), generator=g))
            x0 = int(torch.randint(0, W - erase_hw[1], (1,), generator=g))
            mask[b, :, y0:y0 + erase_hw[0], x0:x0 + erase_hw[1]] = 0
    inputs[("mask", 0, 0)] = mask
    K, inv_K = kitti_K(B, H, W)
    inputs["K"] = K
    inputs["inv_K"] = inv_K
    return inputs


def random_poses(g, B, ref_net_cls, net):
    Ts = {}
    for f in (-1, 1):
        axis = 0.01 * torch.randn(B, 1, 3, generator=g)
        trans = 0.15 * torch.randn(B, 1, 3, generator=g)
        Ts[f] = ref_net_cls.transformation_from_parameters(net, axis, trans, invert=(f < 0))
    return Ts


def gen_photo(ref_layers, ref_net_cls):
    """generate_images_pred + automask/min block per scale, with gradients w.r.t.
    disp_s and cam_T_cam, through the reference's own methods."""
    g = torch.Generator().manual_seed(23)
    B, H, W = 2, 32, 64
    inputs = make_batch(g, B, H, W)
    net = bare(ref_net_cls)
    net.ssim = ref_layers.SSIM()
    net.backproject = ref_layers.Backproject(B, H, W)
    net.project = ref_layers.Project(B, H, W)
    net.opt = AttrDict(height=H, width=W, min_depth=0.1, max_depth=100.0, frame_ids=[0, -1, 1],
                       scales=[0, 1, 2, 3], automask=True, imgs_per_gpu=B)
    Ts = random_poses(g, B, ref_net_cls, net)
    arrays = {}
    for k in (("color", 0, 0), ("color", -1, 0), ("color", 1, 0)):
        arrays["color_%d" % k[1]] = npy(inputs[k])
    arrays["K"] = npy(inputs["K"])
    arrays["inv_K"] = npy(inputs["inv_K"])
    for f in (-1, 1):
        arrays["T_%d" % f] = npy(Ts[f])
    target = inputs[("color", 0, 0)]
    for s in range(4):
        hs, ws = H >> (s + 1), W >> (s + 1)
        disp = (0.05 + 0.9 * smooth_image(g, B, 1, max(hs, 4), max(ws, 4))[:, :, :hs, :ws]).contiguous()
        disp.requires_grad_(True)
        T = {f: Ts[f].clone().requires_grad_(True) for f in (-1, 1)}
        outputs = {("disp", 0, s): disp, ("cam_T_cam", 0, -1): T[-1], ("cam_T_cam", 0, 1): T[1]}
        outputs = ref_net_cls.generate_images_pred(net, inputs, outputs, s)
        with NoiseTap() as tap:
            losses = []
            for f in (-1, 1):
                ident = ref_net_cls.compute_reprojection_loss(net, inputs[("color", f, 0)], target)
                ident += torch.randn(ident.shape).cuda() * 1e-5
                losses.append(ident)
            for f in (-1, 1):
                losses.append(ref_net_cls.compute_reprojection_loss(net, outputs[("color", f, s)], target))
            stack = torch.cat(losses, 1)
            mn, idx = torch.min(stack, dim=1)
            loss = mn.mean() / 4
        loss.backward()
        p = "s%d_" % s
        arrays[p + "disp"] = npy(disp)
        arrays[p + "noise"] = np.stack([npy(d) for d in tap.draws])
        arrays[p + "warped_-1"] = npy(outputs[("color", -1, s)])
        arrays[p + "warped_1"] = npy(outputs[("color", 1, s)])
        arrays[p + "cands"] = npy(stack)
        arrays[p + "min_index"] = npy(idx).astype(np.uint8)
        arrays[p + "loss"] = npy(loss)
        arrays[p + "d_disp"] = npy(disp.grad)
        arrays[p + "d_T_-1"] = npy(T[-1].grad)
        arrays[p + "d_T_1"] = npy(T[1].grad)
        # the same without automask (stereo-style configs)
    save("photo_scales.npz", **arrays)


class StubEncoder(nn.Module):
    """Tiny stand-in for the ResNet feature extractor so that compute_losses can be
    exercised without shipping ResNet weights: 5 feature maps at strides 2..32."""

    def __init__(self, chans=(8, 8, 12, 16, 16)):
        super().__init__()
        self.convs = nn.ModuleList()
        cin = 3
        for c in chans:
            self.convs.append(nn.Conv2d(cin, c, 3, 2, 1))
            cin = c

    def forward(self, x):
        feats = []
        for conv in self.convs:
            x = torch.tanh(conv(x))
            feats.append(x)
        return feats


def gen_losses_tripled(ref_inpaint):
    """compute_losses of mono_fm_joint_inpaint_disentangle (cfg_kitti_tripleD's model)
    on synthetic decoder outputs, with the ResNet extractor replaced by StubEncoder."""
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(37)
    B, H, W = 2, 64, 96
    opt = AttrDict(name="mono_fm_joint_inpaint_disentangle", depth_num_layers=18, pose_num_layers=18,
                   extractor_num_layers=18, frame_ids=[0, -1, 1], imgs_per_gpu=B, height=H, width=W,
                   scales=[0, 1, 2, 3], min_depth=0.1, max_depth=100.0, depth_pretrained_path=None,
                   pose_pretrained_path=None, extractor_pretrained_path=None, automask=True, disp_norm=True,
                   dis=1e-3, cvt=1e-3, perception_weight=1e-3, smoothness_weight=1e-3, auto_res_weight=5e-3,
                   disentangle_layers=[False, False, False, False, True], skip_connection_multiplier=1,
                   depth_skip_type=None, color_skip_type=None, color_skip_layers=[False] * 4,
                   depth_use_shuffle=False, depth_disentangle_type="use_half", freeze_extractor=False)
    model = ref_inpaint.mono_fm_joint_inpaint_disentangle(opt)
    model.train()
    stub = StubEncoder()
    model.Encoder = stub
    inputs = make_batch(g, B, H, W)
    Ts = random_poses(g, B, type(model), model)
    leaves = {}

    def leaf(name, t):
        t = t.detach().clone().requires_grad_(True)
        leaves[name] = t
        return t

    outputs = {}
    for s in range(4):
        hs, ws = H >> (s + 1), W >> (s + 1)
        outputs[("disp", 0, s)] = leaf("disp_%d" % s, 0.05 + 0.9 * smooth_image(g, B, 1, hs + 4, ws + 4)[:, :, :hs, :ws])
        outputs[("res_img", 0, s)] = leaf("res_img_%d" % s, smooth_image(g, B, 3, (H >> s) + 4, (W >> s) + 4)[:, :, :H >> s, :W >> s])
    outputs[("auto_res_img", 0, 0)] = leaf("auto_res_img_0", smooth_image(g, B, 3, H, W))
    for f in (-1, 1):
        outputs[("cam_T_cam", 0, f)] = leaf("T_%d" % f, Ts[f])
    with torch.no_grad():
        feats0 = stub(inputs[("color", 0, 0)])
    features = [leaf("feat_%d" % i, t) for i, t in enumerate(feats0)]

    with NoiseTap() as tap:
        loss_dict = model.compute_losses(inputs, outputs, features)
    total = sum(v.mean() for v in loss_dict.values())
    total.backward()

    arrays = {}
    for k in (("color", 0, 0), ("color", -1, 0), ("color", 1, 0), ("mask", 0, 0)):
        arrays["%s_%d" % (k[0], k[1])] = npy(inputs[k])
    arrays["K"] = npy(inputs["K"])
    arrays["inv_K"] = npy(inputs["inv_K"])
    for n, t in leaves.items():
        arrays["in_" + n] = npy(t)
        arrays["grad_" + n] = npy(t.grad) if t.grad is not None else np.zeros(0, np.float32)
    for n, p in stub.state_dict().items():
        arrays["stub_" + n] = npy(p)
    for n, p in stub.named_parameters():
        if p.grad is not None:
            arrays["stubgrad_" + n] = npy(p.grad)
    arrays["noise"] = np.stack([npy(d) for d in tap.draws])
    for k, v in loss_dict.items():
        arrays["loss_" + str(k)] = npy(v)
    arrays["total"] = npy(total)
    for s in range(4):
        arrays["min_index_%d" % s] = npy(outputs[("min_index", s)]).astype(np.uint8)
    arrays["perc_min_index"] = npy(outputs["min_index"]).astype(np.uint8)
    save("losses_tripled.npz", **arrays)


def gen_losses_fm(ref_fm):
    """compute_losses of mono_fm (cfg_kitti_fm, BASELINE config #1) with StubEncoder as extractor."""
    torch.manual_seed(6)
    g = torch.Generator().manual_seed(41)
    B, H, W = 2, 64, 96
    opt = AttrDict(name="mono_fm", depth_num_layers=18, pose_num_layers=18, extractor_num_layers=18,
                   frame_ids=[0, -1, 1], imgs_per_gpu=B, height=H, width=W, scales=[0, 1, 2, 3],
                   min_depth=0.1, max_depth=100.0, depth_pretrained_path=None, pose_pretrained_path=None,
                   extractor_pretrained_path=None, automask=True, disp_norm=True, perception_weight=1e-3,
                   smoothness_weight=1e-3)
    model = ref_fm.mono_fm(opt)
    model.train()
    stub = StubEncoder()
    model.extractor = stub
    inputs = make_batch(g, B, H, W)
    Ts = random_poses(g, B, type(model), model)
    leaves = {}
    outputs = {}
    for s in range(4):
        hs, ws = H >> (s + 1), W >> (s + 1)
        t = (0.05 + 0.9 * smooth_image(g, B, 1, hs + 4, ws + 4)[:, :, :hs, :ws]).clone().requires_grad_(True)
        leaves["disp_%d" % s] = t
        outputs[("disp", 0, s)] = t
    for f in (-1, 1):
        t = Ts[f].clone().requires_grad_(True)
        leaves["T_%d" % f] = t
        outputs[("cam_T_cam", 0, f)] = t
    with NoiseTap() as tap:
        loss_dict = model.compute_losses(inputs, outputs)
    total = sum(v.mean() for v in loss_dict.values())
    total.backward()
    arrays = {}
    for k in (("color", 0, 0), ("color", -1, 0), ("color", 1, 0)):
        arrays["%s_%d" % (k[0], k[1])] = npy(inputs[k])
    arrays["K"] = npy(inputs["K"])
    arrays["inv_K"] = npy(inputs["inv_K"])
    for n, t in leaves.items():
        arrays["in_" + n] = npy(t)
        arrays["grad_" + n] = npy(t.grad)
    for n, p in stub.state_dict().items():
        arrays["stub_" + n] = npy(p)
    for n, p in stub.named_parameters():
        if p.grad is not None:
            arrays["stubgrad_" + n] = npy(p.grad)
    arrays["noise"] = np.stack([npy(d) for d in tap.draws])
    for k, v in loss_dict.items():
        arrays["loss_" + str(k)] = npy(v)
    arrays["total"] = npy(total)
    save("losses_fm.npz", **arrays)


def gen_metrics(ref):
    spec = importlib.util.spec_from_file_location("ref_pixel_error",
                                                  os.path.join(ref, "mono/core/evaluation/pixel_error.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.RandomState(3)
    gt = rng.uniform(1.0, 80.0, size=5000).astype(np.float32)
    pred = (gt * rng.uniform(0.7, 1.4, size=5000)).astype(np.float32)
    errs = np.array(mod.compute_errors(gt, pred), dtype=np.float64)
    same = np.array(mod.compute_errors(gt, gt), dtype=np.float64)
    disp = rng.uniform(0, 1, size=100).astype(np.float32)
    sd, dp = mod.disp_to_depth(disp)
    save("metrics.npz", gt=gt, pred=pred, errors=errs, errors_same=same, disp=disp, scaled_disp=sd, depth=dp)


def gen_color():
    """rgb2lab of the colourisation models (mono/model/mono_fm_joint_inpaint/color_conversions.py:106-114) and the
    robust-L1 channel-mean map of compute_auto_res_loss (net.py:520-527) on small images."""
    import argparse as _ap
    cc = importlib.import_module("mono.model.mono_fm_joint_inpaint.color_conversions")
    g = torch.Generator().manual_seed(21)
    rgb = smooth_image(g, 2, 3, 24, 40)
    rgb[0, :, :2, :3] = 0.0                      # exercise both branches of the gamma / cube-root thresholds
    rgb[1, :, -2:, -3:] = 1.0
    rgb[0, :, 5, 5] = 0.03
    lab = cc.rgb2lab(rgb, _ap.Namespace(l_cent=50.0, l_norm=50.0, ab_norm=110.0))
    pred = smooth_image(g, 2, 3, 24, 40).requires_grad_(True)
    eps = 1e-3
    l1map = torch.sqrt(torch.pow(pred - rgb, 2) + eps ** 2).mean(1, True) * 5e-3      # net.py:59-65, :520-527
    l1map.mean().backward()
    save("color_lab.npz", rgb=npy(rgb), lab=npy(lab), pred=npy(pred), l1map=npy(l1map), d_pred=npy(pred.grad))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    install_reference(args.ref)
    ref_layers = importlib.import_module("mono.model.mono_fm_joint.layers")
    ref_joint = importlib.import_module("mono.model.mono_fm_joint.net")
    ref_inpaint = importlib.import_module("mono.model.mono_fm_joint_inpaint.net")
    ref_fm = importlib.import_module("mono.model.mono_fm.net")
    gen_ops(ref_layers, ref_joint.mono_fm_joint)
    gen_photo(ref_layers, ref_joint.mono_fm_joint)
    gen_losses_tripled(ref_inpaint)
    gen_losses_fm(ref_fm)
    gen_metrics(args.ref)
    gen_color()


if __name__ == "__main__":
    main()
