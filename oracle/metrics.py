"""Oracle: KITTI depth metrics and the per-image evaluation protocol.

Test infrastructure -- see oracle/__init__.py.
"""
import numpy as np


def compute_errors(gt, pred):
    """compute_errors, mono/core/evaluation/pixel_error.py:27-40.
    Returns (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3)."""
    ratio = np.maximum(gt / pred, pred / gt)
    a1 = (ratio < 1.25).mean()
    a2 = (ratio < 1.25 ** 2).mean()
    a3 = (ratio < 1.25 ** 3).mean()
    diff = gt - pred
    rmse = np.sqrt((diff ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(diff) / gt)
    sq_rel = np.mean(diff ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def disp_to_depth(disp, min_depth=0.1, max_depth=100):
    """mono/core/evaluation/pixel_error.py:43-48."""
    lo = 1 / max_depth
    hi = 1 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1 / scaled


def resize_bilinear(img, out_h, out_w):
    """cv2.resize(img, (w, h)) default INTER_LINEAR as used at scripts/eval_depth.py:78
    and mono/core/evaluation/eval_hooks.py:227: half-pixel centres, edge replicate,
    no anti-aliasing.  cv2 is absent here -> parity unpinned for this helper."""
    in_h, in_w = img.shape

    def axis(n_out, n_in):
        s = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(s).astype(np.int64)
        lam = s - i0
        i1 = np.clip(i0 + 1, 0, n_in - 1)
        i0 = np.clip(i0, 0, n_in - 1)
        return i0, i1, lam

    y0, y1, ly = axis(out_h, in_h)
    x0, x1, lx = axis(out_w, in_w)
    img = img.astype(np.float64)
    rows = img[y0] * (1 - ly)[:, None] + img[y1] * ly[:, None]
    return (rows[:, x0] * (1 - lx)[None] + rows[:, x1] * lx[None]).astype(np.float32)


def eval_single(pred_disp, gt_depth, min_depth=1e-3, max_depth=80.0, stereo_scale=False):
    """Per-image protocol of scripts/eval_depth.py:73-101 (same as
    mono/core/evaluation/eval_hooks.py:225-262): resize the predicted disparity to
    the ground-truth size, invert, mask 1e-3 < gt < 80 within the Garg crop,
    median-scale (or x36 for stereo), clamp, compute_errors.
    Returns (errors tuple, ratio)."""
    gt_h, gt_w = gt_depth.shape
    pred_depth = 1.0 / resize_bilinear(pred_disp, gt_h, gt_w)
    mask = np.logical_and(gt_depth > min_depth, gt_depth < max_depth)
    crop = np.array([0.40810811 * gt_h, 0.99189189 * gt_h,
                     0.03594771 * gt_w, 0.96405229 * gt_w]).astype(np.int32)
    crop_mask = np.zeros(mask.shape)
    crop_mask[crop[0]:crop[1], crop[2]:crop[3]] = 1
    mask = np.logical_and(mask, crop_mask)
    pd = pred_depth[mask]
    gd = gt_depth[mask]
    ratio = 36.0 if stereo_scale else np.median(gd) / np.median(pd)
    pd = pd * ratio
    pd = np.clip(pd, min_depth, max_depth)
    return compute_errors(gd, pd), ratio
