#!/usr/bin/env python3
"""Offline KITTI depth evaluation (reference: scripts/eval_depth.py:22-109): load a checkpoint, predict the
disparity of every validation frame, resize to the ground-truth size, median-scale (x36 for stereo), clamp to
[1e-3, 80] inside the Garg crop and print the seven standard metrics.

  python scripts/eval_depth.py --config config/cfg_kitti_tripleD.py --checkpoint work/epoch_20.pth \
         --gt_depths /data/kitti_raw/gt_depths.npz
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tripled_amd  # noqa: F401,E402
from mmcv import Config  # noqa: E402
from mono.core.evaluation import disp_to_depth, evaluate_disparity  # noqa: E402
from mono.core.evaluation.eval_hooks import METRICS  # noqa: E402
from mono.datasets.get_dataset import get_dataset  # noqa: E402
from mono.model import MONO  # noqa: E402


def evaluate(model, dataset, stereo_scale=False, device="cuda"):
    """Returns (mean metrics dict, scale ratios) over a validation dataset whose samples carry 'gt_depth'."""
    model.eval().to(device)
    results = []
    with torch.no_grad():
        for idx in range(len(dataset)):
            sample = dataset[idx]
            batch = {k: torch.as_tensor(v).float().unsqueeze(0).to(device) for k, v in sample.items() if k != "gt_depth"}
            scaled, _ = disp_to_depth(model(batch)[("disp", 0, 0)].float(), 0.1, 100)
            gt = np.asarray(sample["gt_depth"], dtype=np.float32)
            results.append(evaluate_disparity(scaled.cpu()[0, 0].numpy(), gt, stereo_scale))
    mean = {k: float(np.mean([r[k] for r in results])) for k in METRICS}
    return mean, np.array([r["scale"] for r in results])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--gt_depths", default=None)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    args = ap.parse_args()
    cfg = Config.fromfile(args.config)
    if args.gt_depths:
        cfg.data["gt_depth_path"] = args.gt_depths
    cfg.model["imgs_per_gpu"] = 1
    model = MONO.module_dict[cfg.model["name"]](cfg.model)
    ckpt = torch.load(args.checkpoint, map_location="cpu", weights_only=True)      # executes nothing from the file
    model.load_state_dict(ckpt["state_dict"], strict=True)
    mean, ratios = evaluate(model, get_dataset(cfg.data, training=False), bool(cfg.data["stereo_scale"]), args.device)
    med = np.median(ratios)
    print("Scaling ratios | med: {:0.3f} | std: {:0.3f}".format(med, np.std(ratios / med)))
    print("\n  " + ("{:>8} | " * 7).format(*METRICS))
    print(("&{: 8.3f}  " * 7).format(*[mean[k] for k in METRICS]) + "\\\\")


if __name__ == "__main__":
    main()
