"""Oracle: the loss orchestrators (compute_losses) of the BASELINE model families,
restated on top of the per-op oracle functions.

Test infrastructure -- see oracle/__init__.py.  Everything is unfused plain
PyTorch, so autograd provides the reference gradients.
"""
import torch
import torch.nn.functional as F

from . import geometry, photometric, smooth


class NoiseSource:
    """Supplies the automask tie-break noise.  The reference draws
    torch.randn(shape) on the CPU generator once per source frame and scale
    (mono/model/mono_fm_joint_inpaint/net.py:105).  ``draws`` replays recorded
    tensors in that order; without it fresh draws are made (and recorded)."""

    def __init__(self, draws=None):
        self.replay = list(draws) if draws is not None else None
        self.used = []

    def __call__(self, shape, device):
        if self.replay is not None:
            t = self.replay.pop(0).reshape(shape).to(device)
        else:
            t = torch.randn(shape).to(device)
        self.used.append(t)
        return t


def feature_warp(opt, inputs, outputs, encoder):
    """generate_features_pred, mono/model/mono_fm_joint/net.py:196-223
    (mono/model/mono_fm/net.py:172-199 is the same): disp_0 -> (H/2, W/2),
    K rows 0,1 halved, inv_K = pinverse(K) per sample, warp encoder(src)[0]."""
    h2, w2 = int(opt.height / 2), int(opt.width / 2)
    disp = F.interpolate(outputs[("disp", 0, 0)], [h2, w2], mode="bilinear", align_corners=False)
    _, depth = geometry.disp_to_depth(disp, opt.min_depth, opt.max_depth)
    warped = {}
    for frame_id in opt.frame_ids[1:]:
        T = inputs["stereo_T"] if frame_id == "s" else outputs[("cam_T_cam", 0, frame_id)]
        K = inputs["K"].clone()
        K[:, 0, :] = K[:, 0, :] / 2
        K[:, 1, :] = K[:, 1, :] / 2
        inv_K = torch.stack([torch.pinverse(K[i]) for i in range(K.shape[0])])
        pts = geometry.backproject(depth, inv_K)
        grid = geometry.project(pts, K, T, h2, w2)
        src_f = encoder(inputs[("color", frame_id, 0)])[0]
        warped[frame_id] = geometry.grid_sample_border(src_f, grid)
    return warped


def _photometric_and_smooth(opt, inputs, outputs, scale, noise, loss_dict, out_extra):
    """Per-scale body shared by all families: generate_images_pred + automask +
    min-reprojection + disp normalisation + smoothness
    (mono/model/mono_fm_joint_inpaint/net.py:93-131)."""
    target = inputs[("color", 0, 0)]
    n_scales = len(opt.scales)
    disp = outputs[("disp", 0, scale)]
    srcs = [inputs[("color", f, 0)] for f in opt.frame_ids[1:]]
    Ts = [inputs["stereo_T"] if f == "s" else outputs[("cam_T_cam", 0, f)] for f in opt.frame_ids[1:]]
    draws = None
    if opt.automask:
        b, _, h, w = target.shape
        draws = [noise((b, 1, h, w), target.device) for _ in srcs]
    loss, idx, warped = photometric.photometric_scale_loss(
        target, srcs, disp, inputs["K"], inputs["inv_K"], Ts, draws,
        opt.min_depth, opt.max_depth, automask=opt.automask, n_scales=n_scales)
    for f, wimg in zip(opt.frame_ids[1:], warped):
        out_extra[("color", f, scale)] = wimg
    out_extra[("min_index", scale)] = idx
    loss_dict[("min_reconstruct_loss", scale)] = loss
    if opt.disp_norm:
        disp = smooth.mean_normalize(disp)
    sm = smooth.smooth_loss(disp, target)
    loss_dict[("smooth_loss", scale)] = opt.smoothness_weight * sm / (2 ** scale) / n_scales


def compute_losses_inpaint(opt, inputs, outputs, features, encoder, noise=None):
    """mono_fm_joint_inpaint.compute_losses, mono/model/mono_fm_joint_inpaint/net.py:47-133.
    Returns (loss_dict, extra outputs)."""
    noise = noise or NoiseSource()
    loss_dict = {}
    extra = {}
    target = inputs[("color", 0, 0)]
    mask = inputs[("mask", 0, 0)]
    recon_w = opt.get("img_reconstruct_weight", 1)
    if features is not None:
        for i in range(5):
            reg = smooth.feature_regularization_loss(features[i], target, opt.dis, opt.cvt)
            loss_dict[("feature_regularization_loss", i)] = reg / (2 ** i) / 5
        warped_f = feature_warp(opt, inputs, outputs, encoder)
        cands = []
        for f in opt.frame_ids[1:]:
            extra[("feature", f, 0)] = warped_f[f]
            cands.append(photometric.perceptional_loss(features[0], warped_f[f]))
        vals, extra["min_index"] = torch.min(torch.cat(cands, 1), dim=1)
        loss_dict["min_perceptional_loss"] = opt.perception_weight * vals.mean()
    for scale in opt.scales:
        if features is not None and recon_w != 0:
            loss_dict[("img_reconstruct_loss", scale)] = photometric.masked_reconstruction_loss(
                outputs[("res_img", 0, scale)], target, mask, len(opt.scales), recon_w)
        _photometric_and_smooth(opt, inputs, outputs, scale, noise, loss_dict, extra)
    return loss_dict, extra


def compute_losses_disentangle(opt, inputs, outputs, features, encoder, noise=None):
    """mono_fm_joint_inpaint_disentangle.compute_losses,
    mono/model/mono_fm_joint_inpaint/net.py:529-532 (+ compute_auto_res_loss :520-527)."""
    loss_dict, extra = compute_losses_inpaint(opt, inputs, outputs, features, encoder, noise)
    if opt.auto_res_weight > 0.0:
        target = inputs[("color", 0, 0)]
        loss_dict["auto_res_loss"] = photometric.perceptional_loss(
            target, outputs[("auto_res_img", 0, 0)]) * opt.auto_res_weight
    return loss_dict, extra


def compute_losses_fm(opt, inputs, outputs, extractor, noise=None):
    """mono_fm.compute_losses, mono/model/mono_fm/net.py:69-133: per scale the
    photometric block, a perceptual min-loss over warped extractor features
    (weight perception_weight / n_scales) and the smoothness term."""
    noise = noise or NoiseSource()
    loss_dict = {}
    extra = {}
    n_scales = len(opt.scales)
    for scale in opt.scales:
        # the reference orders the keys min_reconstruct, min_perceptional, smooth
        tmp = {}
        _photometric_and_smooth(opt, inputs, outputs, scale, noise, tmp, extra)
        warped_f = feature_warp(opt, inputs, outputs, extractor)
        cands = []
        for f in opt.frame_ids[1:]:
            extra[("feature", f, 0)] = warped_f[f]
            tgt_f = extractor(inputs[("color", 0, 0)])[0]
            cands.append(photometric.perceptional_loss(tgt_f, warped_f[f]))
        vals, _ = torch.min(torch.cat(cands, 1), dim=1)
        loss_dict[("min_reconstruct_loss", scale)] = tmp[("min_reconstruct_loss", scale)]
        loss_dict[("min_perceptional_loss", scale)] = opt.perception_weight * vals.mean() / n_scales
        loss_dict[("smooth_loss", scale)] = tmp[("smooth_loss", scale)]
    return loss_dict, extra


def total_loss(loss_dict):
    """batch_processor's reduction, mono/apis/trainer.py:37-47: sum of .mean() of every entry."""
    return sum(v.mean() for v in loss_dict.values())
