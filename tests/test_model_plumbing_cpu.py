"""BASELINE config #1: cfg_kitti_fm, ResNet18 depth+pose, 2 random 192x640 triplets, CPU forward
+ reprojection loss (plumbing, no GPU).  The loss hot path runs through the oracle backend here;
the product backend itself refuses CPU tensors (see tests/test_abi.py)."""
import os

import numpy as np
import pytest
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


REF_CONFIGS = "/root/reference/config"



@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="reference checkout not present")
@pytest.mark.parametrize("name", ["cfg_kitti_tripleD", "cfg_kitti_fm", "cfg_kitti_fm_joint_inpaint_disentangle_distill_full_colorize",
                                  "cfg_kitti_fm_joint", "cfg_kitti_fm_joint_inpaint", "cfg_kitti_fm_joint_inpaint_disentangle",
                                  "cfg_kitti_fm_refine"])        # (frame_ids [0, -1, 1, 's']: the stereo pair, no auto-mask)
def test_reference_config_files_train_one_iteration_on_the_host(name):
    """The reference's OWN config file (unchanged; only the ResNet depths and the image size are reduced for the CPU) through
    this build's batch_processor: every loss entry finite, every trainable parameter the config's switches leave in the graph
    receives a gradient.  (At 64x128 the level-4 feature map is 2x4 and the second-order regulariser averages an empty tensor:
    NaN in the reference too -- hence 96x160.)"""
    cfg = Config.fromfile(os.path.join(REF_CONFIGS, name + ".py"))
    m = cfg.model
    for k in list(m.keys()):
        if k.endswith("pretrained_path"):
            m[k] = None
    for k in ("depth_num_layers", "pose_num_layers", "extractor_num_layers", "colorize_num_layers"):
        if k in m:
            m[k] = 18
    B, H, W = 1, 96, 160
    m["imgs_per_gpu"], m["height"], m["width"] = B, H, W
    torch.manual_seed(0)
    model = MONO.module_dict[m["name"]](m)
    model.set_loss_backend(OracleLossBackend())
    model.train()
    batch = synthetic_batch(B, H, W, seed=1, frame_ids=tuple(m["frame_ids"]))
    if "s" in m["frame_ids"]:
        stereo_T = torch.eye(4).repeat(B, 1, 1)
        stereo_T[:, 0, 3] = -0.015                   # left camera, no flip (mono_dataset.py:194-199)
        batch["stereo_T"] = stereo_T
    out = batch_processor(model, batch, train_mode=True)
    assert out["num_samples"] == B and torch.isfinite(out["loss"])
    assert all(np.isfinite(v) for v in out["log_vars"].values()), out["log_vars"]
    out["loss"].backward()
    with_grad = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is not None]
    assert len(with_grad) > 150
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
