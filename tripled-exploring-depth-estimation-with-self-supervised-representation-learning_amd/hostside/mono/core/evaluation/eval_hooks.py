"""Per-epoch KITTI depth evaluation hooks (reference: mono/core/evaluation/eval_hooks.py:95-291,
397-436; same protocol as scripts/eval_depth.py:73-101).  cv2 is not a dependency here: the
disparity is resized with an explicit half-pixel-centre bilinear kernel (cv2.INTER_LINEAR's
definition; parity with cv2 itself is unpinned -- cv2 is not installed in the build image).
Per-rank results are merged with all_gather_object instead of pickle files + barriers."""
import time

import contextlib

import numpy as np
import torch
import torch.distributed as dist
from mmcv.runner import Hook
from torch.utils.data import Dataset

from .pixel_error import AverageMeter, compute_errors, disp_to_depth

MIN_DEPTH = 1e-3
MAX_DEPTH = 80
STEREO_SCALE_FACTOR = 36
METRICS = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")


def resize_bilinear(img, out_h, out_w):
    in_h, in_w = img.shape[:2]

    def axis(n_out, n_in):
        s = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(s).astype(np.int64)
        lam = s - i0
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), lam

    y0, y1, ly = axis(out_h, in_h)
    x0, x1, lx = axis(out_w, in_w)
    img = img.astype(np.float64)
    rows = img[y0] * (1 - ly)[:, None] + img[y1] * ly[:, None]
    return (rows[:, x0] * (1 - lx)[None] + rows[:, x1] * lx[None]).astype(np.float32)


def evaluate_disparity(pred_disp, gt_depth, stereo_scale=False):
    """One image: resize the scaled disparity to the ground-truth size, invert, keep
    1e-3 < gt < 80 inside the Garg crop, median-scale (x36 for stereo), clamp, compute_errors.
    Returns a dict with the seven metrics and 'scale' (the median ratio)."""
    gt_h, gt_w = gt_depth.shape[:2]
    pred_depth = 1 / resize_bilinear(pred_disp, gt_h, gt_w)
    mask = np.logical_and(gt_depth > MIN_DEPTH, gt_depth < MAX_DEPTH)
    crop = np.array([0.40810811 * gt_h, 0.99189189 * gt_h, 0.03594771 * gt_w, 0.96405229 * gt_w]).astype(np.int32)
    crop_mask = np.zeros(mask.shape)
    crop_mask[crop[0]:crop[1], crop[2]:crop[3]] = 1
    mask = np.logical_and(mask, crop_mask)
    pred = pred_depth[mask]
    gt = gt_depth[mask]
    ratio = np.median(gt) / np.median(pred)
    pred = pred * (STEREO_SCALE_FACTOR if stereo_scale else ratio)
    pred = np.clip(pred, MIN_DEPTH, MAX_DEPTH)
    out = dict(zip(METRICS, (float(v) for v in compute_errors(gt, pred))))
    out["scale"] = float(ratio)
    return out


def _predict_one(model, sample):
    """dataset[idx] -> scaled disparity [h, w] (numpy) and ground-truth depth."""
    dev = next(model.parameters()).device
    batch = {}
    for k, v in sample.items():
        t = torch.as_tensor(v)
        if isinstance(k, tuple) and k and k[0] == "color_u8":
            if dev.type == "cuda":                       # 'uint8' wire format: expanded by the HIP kernel below
                batch[k] = t.unsqueeze(0).to(dev)
            else:                                        # evaluation on the host: plain ToTensor (no jitter in validation)
                img = t.float().div(255.0).unsqueeze(0)
                batch[("color", k[1], 0)], batch[("color_aug", k[1], 0)] = img, img
            continue
        batch[k] = t.float().unsqueeze(0).to(dev)
    if dev.type == "cuda":
        from mono.datasets import expand_device_batch
        expand_device_batch(batch)
    batch.pop("aug", None)
    inner = model.module if hasattr(model, "module") else model
    flat = getattr(inner, "_flat_store", None)
    ctx = flat.full_precision() if flat is not None else contextlib.nullcontext()
    with torch.no_grad(), ctx:        # (flat store: validate on the fp32 master weights, as the reference does)
        result = model(batch)
    scaled, _ = disp_to_depth(result[("disp", 0, 0)].float())
    gt = torch.as_tensor(sample["gt_depth"]).float().cpu().numpy()
    return scaled.cpu()[0, 0].numpy(), gt


def _publish(runner, results):
    meters = {k: AverageMeter() for k in METRICS + ("scale",)}
    for r in results:
        for k in meters:
            meters[k].update(r[k])
    out = runner.log_buffer.output
    for k in METRICS:
        out[k] = meters[k].avg
    out["scale mean"] = meters["scale"].avg
    out["scale std"] = float(np.std([r["scale"] for r in results]))
    runner.log_buffer.ready = True


class NonDistEvalHook(Hook):
    def __init__(self, dataset, cfg):
        assert isinstance(dataset, Dataset)
        self.dataset = dataset
        self.interval = cfg.get("validate_interval", 1)
        self.out_path = cfg.get("work_dir", "./")
        self.cfg = cfg

    def after_train_epoch(self, runner):
        if not self.every_n_epochs(runner, self.interval):
            return
        runner.model.eval()
        stereo = bool(self.cfg.data["stereo_scale"])
        results = []
        for idx in range(len(self.dataset)):
            disp, gt = _predict_one(runner.model, self.dataset[idx])
            results.append(evaluate_disparity(disp, gt, stereo))
        _publish(runner, results)


class DistEvalHook(Hook):
    def __init__(self, dataset, interval=1, cfg=None):
        assert isinstance(dataset, Dataset)
        self.dataset, self.interval, self.cfg = dataset, interval, cfg

    def after_train_epoch(self, runner):
        if not self.every_n_epochs(runner, self.interval):
            return
        runner.model.eval()
        stereo = bool(self.cfg.data["stereo_scale"])
        mine = {}
        t0 = time.time()
        for idx in range(runner.rank, len(self.dataset), runner.world_size):   # reference :212
            disp, gt = _predict_one(runner.model, self.dataset[idx])
            mine[idx] = evaluate_disparity(disp, gt, stereo)
        fps = len(mine) / max(time.time() - t0, 1e-9)
        if runner.world_size > 1:
            gathered = [None] * runner.world_size
            dist.all_gather_object(gathered, mine)
            merged = {}
            for part in gathered:
                merged.update(part)
        else:
            merged = mine
        if runner.rank == 0:
            runner.logger.info("evaluation: %d images, %.1f img/s on rank 0", len(merged), fps)
            self.evaluate(runner, [merged[i] for i in sorted(merged)])

    def evaluate(self, runner, results):
        raise NotImplementedError


class DistEvalMonoHook(DistEvalHook):
    def evaluate(self, runner, results):
        if not isinstance(results, list):
            raise TypeError("results must be a list of per-image metric dicts, not {}".format(type(results)))
        _publish(runner, results)
