"""Oracle: SSIM + robust-L1 photometric term and the min-reprojection block.

Test infrastructure -- see oracle/__init__.py.
"""
import torch
import torch.nn.functional as F

from . import geometry

SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2
L1_EPS = 1e-3


def _box3_reflect(x):
    """ReflectionPad2d(1) followed by AvgPool2d(3, 1)
    (mono/model/mono_fm_joint/layers.py:88-93,98-101)."""
    return F.avg_pool2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1)


def ssim_loss(x, y):
    """SSIM.forward, mono/model/mono_fm_joint/layers.py:97-107.
    Returns clamp((1 - SSIM)/2, 0, 1) per pixel and channel."""
    mu_x = _box3_reflect(x)
    mu_y = _box3_reflect(y)
    sig_x = _box3_reflect(x * x) - mu_x * mu_x
    sig_y = _box3_reflect(y * y) - mu_y * mu_y
    sig_xy = _box3_reflect(x * y) - mu_x * mu_y
    num = (2 * mu_x * mu_y + SSIM_C1) * (2 * sig_xy + SSIM_C2)
    den = (mu_x * mu_x + mu_y * mu_y + SSIM_C1) * (sig_x + sig_y + SSIM_C2)
    return torch.clamp((1 - num / den) / 2, 0, 1)


def robust_l1(pred, target):
    """robust_l1, mono/model/mono_fm_joint/net.py:59-61: sqrt((t-p)^2 + 1e-6)."""
    return torch.sqrt((target - pred) ** 2 + L1_EPS ** 2)


def reprojection_loss(pred, target):
    """compute_reprojection_loss, mono/model/mono_fm_joint/net.py:67-71.
    0.85 * mean_c SSIM(pred, target) + 0.15 * mean_c robust_l1 -> [B,1,H,W]."""
    l1 = robust_l1(pred, target).mean(1, True)
    ss = ssim_loss(pred, target).mean(1, True)
    return 0.85 * ss + 0.15 * l1


def perceptional_loss(tgt_f, src_f):
    """compute_perceptional_loss, mono/model/mono_fm_joint/net.py:63-65."""
    return robust_l1(tgt_f, src_f).mean(1, True)


def min_reprojection(target, sources, warped, noise=None, automask=True, forced_index=None):
    """The automask + minimum-reprojection block,
    mono/model/mono_fm_joint_inpaint/net.py:101-117 (identical in
    mono_fm/net.py:90-106 and mono_fm_joint/net.py:109-128).

    sources / warped: lists (one per non-reference frame, in frame_ids[1:] order).
    noise: list of [B,1,H,W] N(0,1) draws (the reference draws them with
    torch.randn on the CPU generator, one per source frame); scaled by 1e-5 here.
    Candidate order along dim 1: identity terms first, then warped terms.
    Returns (per-pixel min [B,H,W], argmin [B,H,W] int64, stacked candidates).
    """
    cands = []
    if automask:
        for i, src in enumerate(sources):
            ident = reprojection_loss(src, target)
            if noise is not None:
                ident = ident + noise[i] * 1e-5
            cands.append(ident)
    for wimg in warped:
        cands.append(reprojection_loss(wimg, target))
    stack = torch.cat(cands, 1)
    if forced_index is None:
        vals, idx = torch.min(stack, dim=1)
    else:
        idx = forced_index
        vals = torch.gather(stack, 1, idx.unsqueeze(1)).squeeze(1)
    return vals, idx, stack


def photometric_scale_loss(target, sources, disp_s, K, inv_K, Ts, noise, min_depth, max_depth,
                           automask=True, n_scales=4, forced_index=None):
    """generate_images_pred (mono/model/mono_fm_joint/net.py:181-194) + the
    min-reprojection block for one scale; returns (loss scalar = mean(min)/n_scales,
    argmin, list of warped images)."""
    warped = [geometry.warp_source(src, disp_s, K, inv_K, T, min_depth, max_depth,
                                   target.shape[2], target.shape[3])
              for src, T in zip(sources, Ts)]
    vals, idx, _ = min_reprojection(target, sources, warped, noise, automask, forced_index)
    return vals.mean() / n_scales, idx, warped


def masked_reconstruction_loss(res_img, target, mask, n_scales=4, weight=1.0):
    """Auto-encoder / in-painting reconstruction term,
    mono/model/mono_fm_joint_inpaint/net.py:80-91: resize target and mask
    bilinearly to res_img's size, photometric loss, average over erased pixels
    (mask == 0 marks an erased pixel)."""
    h, w = res_img.shape[2:]
    t = F.interpolate(target, [h, w], mode="bilinear", align_corners=False)
    m = F.interpolate(mask, [h, w], mode="bilinear", align_corners=False)
    loss = reprojection_loss(res_img, t)
    loss = torch.sum(loss * (1 - m)) / torch.sum(1 - m)
    return loss / n_scales * weight
