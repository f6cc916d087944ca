"""BASELINE config #1: cfg_kitti_fm, ResNet18 depth+pose, 2 random 192x640 triplets, CPU forward
+ reprojection loss (plumbing, no GPU).  The loss hot path runs through the oracle backend here;
the product backend itself refuses CPU tensors (see tests/test_abi.py)."""
import os

import numpy as np
import torch

import tripled_amd  # noqa: F401
from mmcv import Config
from mono.apis import batch_processor
from mono.core import compute_errors, evaluate_disparity
from mono.datasets import synthetic_batch
from mono.model import MONO
from oracle.backend import OracleLossBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cfg_kitti_fm_cpu_step():
    cfg = Config.fromfile(os.path.join(ROOT, "config", "cfg_kitti_fm.py"))
    m = cfg.model
    assert (m.depth_num_layers, m.pose_num_layers, m.imgs_per_gpu, m.height, m.width) == (18, 18, 2, 192, 640)
    torch.manual_seed(0)
    model = MONO.module_dict[m.name](m)
    model.set_loss_backend(OracleLossBackend())
    batch = synthetic_batch(2, 192, 640, seed=5)
    out = batch_processor(model, batch, train_mode=True)
    assert set(out) == {"loss", "log_vars", "num_samples"} and out["num_samples"] == 2
    keys = list(out["log_vars"])
    assert keys[:3] == ["('min_reconstruct_loss', 0)", "('min_perceptional_loss', 0)", "('smooth_loss', 0)"]
    assert keys[-1] == "loss" and len(keys) == 13
    assert torch.isfinite(out["loss"])
    out["loss"].backward()
    g = model.DepthDecoder.disp1[0].conv.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    model.eval()
    with torch.no_grad():
        pred = model({("color_aug", 0, 0): batch[("color_aug", 0, 0)]})
    assert pred[("disp", 0, 0)].shape == (2, 1, 96, 320) and pred[("disp", 0, 3)].shape == (2, 1, 12, 40)


def test_eval_protocol_known_answers():
    gt = np.random.RandomState(0).uniform(2, 60, size=(60, 200)).astype(np.float32)
    assert compute_errors(gt.ravel(), gt.ravel()) == (0.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0)
    r = evaluate_disparity((1.0 / (gt * 0.5)).astype(np.float32), gt)
    assert abs(r["scale"] - 2.0) < 1e-4 and r["abs_rel"] < 1e-5 and r["a1"] == 1.0
