"""KITTI depth metrics (reference: mono/core/evaluation/pixel_error.py)."""
import numpy as np


class AverageMeter:
    """Running (weighted) average."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def compute_errors(gt, pred):
    """(abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3) between two 1-D depth arrays (reference :27-40)."""
    ratio = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = [(ratio < 1.25 ** k).mean() for k in (1, 2, 3)]
    diff = gt - pred
    rmse = np.sqrt((diff ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(diff) / gt)
    sq_rel = np.mean((diff ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def disp_to_depth(disp, min_depth=0.1, max_depth=100):
    """sigmoid disparity -> (scaled disparity, depth) (reference :43-48)."""
    lo, hi = 1 / max_depth, 1 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1 / scaled
