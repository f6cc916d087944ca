"""Config loader + Runner/hook semantics of the mmcv shim (LR schedule, checkpoint/resume, logs)."""
import os

import pytest
import torch
import torch.nn as nn

import tripled_amd  # noqa: F401
from mmcv import Config
from mmcv.runner import Runner, load_checkpoint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["cfg_kitti_tripleD", "cfg_kitti_fm",
                                  "cfg_kitti_fm_joint_inpaint_disentangle_distill_full_colorize"])
def test_shipped_configs_load(name):
    cfg = Config.fromfile(os.path.join(ROOT, "config", name + ".py"))
    for key in ("data", "model", "optimizer", "optimizer_config", "lr_config", "checkpoint_config", "log_config",
                "dist_params", "workflow", "total_epochs", "imgs_per_gpu", "workers_per_gpu", "validate",
                "resume_from", "load_from", "finetune", "syncbn", "log_level"):
        assert key in cfg, key
    assert cfg.model.imgs_per_gpu == cfg.imgs_per_gpu
    assert cfg.model.get("nonexistent", 7) == 7
    cfg.model["color_skip_layers"] = (False,) * 4      # the models assign into opt
    assert cfg.model.color_skip_layers == (False,) * 4


@pytest.mark.skipif(not os.path.isdir("/root/reference/config"), reason="reference checkout not present")
@pytest.mark.parametrize("name", ["cfg_kitti_tripleD", "cfg_kitti_fm",
                                  "cfg_kitti_fm_joint_inpaint_disentangle_distill_full_colorize",
                                  # the reference's other configs whose model class this build registers
                                  "cfg_kitti_fm_joint", "cfg_kitti_fm_joint_inpaint", "cfg_kitti_fm_joint_inpaint_disentangle",
                                  "cfg_kitti_fm_refine"])
def test_reference_configs_drop_in(name):
    """The reference's own config files load unchanged and build the model (tiny override of the
    ResNet depth only to keep the CPU test fast)."""
    from mono.model import MONO
    cfg = Config.fromfile("/root/reference/config/%s.py" % name)
    m = cfg.model
    for k in list(m.keys()):
        if k.endswith("pretrained_path"):
            m[k] = None
    for k in ("depth_num_layers", "pose_num_layers", "extractor_num_layers", "colorize_num_layers"):
        if k in m:
            m[k] = 18
    model = MONO.module_dict[m["name"]](m)
    assert sum(p.numel() for p in model.parameters()) > 1e6


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc = nn.Linear(4, 1)

    def forward(self, x):
        return self.fc(x)


def _bp(model, data, train_mode):
    loss = (model(data["x"]) - data["y"]).pow(2).mean()
    return dict(loss=loss, log_vars={"loss": loss.detach()}, num_samples=len(data["x"]))


def _loader(n=6):
    g = torch.Generator().manual_seed(0)
    return [{"x": torch.randn(4, 4, generator=g), "y": torch.randn(4, 1, generator=g)} for _ in range(n)]


def test_runner_lr_schedule_checkpoint_resume(tmp_path):
    model = Toy()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    runner = Runner(model, _bp, opt, str(tmp_path), "INFO")
    lrs = []

    from mmcv.runner import Hook

    class Spy(Hook):
        def after_train_iter(self, r):
            lrs.append(r.current_lr()[0])

    runner.register_training_hooks(
        dict(policy="step", warmup="linear", warmup_iters=10, warmup_ratio=1.0 / 3, step=[1, 2], gamma=0.5),
        dict(grad_clip=dict(max_norm=35, norm_type=2)), dict(interval=1),
        dict(interval=2, hooks=[dict(type="TextLoggerHook"), dict(type="TensorboardLoggerHook")]))
    runner.register_hook(Spy(), "LOWEST")
    runner.run([_loader()], [("train", 1)], 3)
    assert runner.epoch == 3 and runner.iter == 18
    # linear warm-up over 10 iterations from base/3, epoch-wise step decay afterwards
    assert abs(lrs[0] - 1e-4 * (1 - (1 - 0 / 10) * (1 - 1 / 3))) < 1e-12
    assert abs(lrs[5] - 1e-4 * (1 - (1 - 5 / 10) * (1 - 1 / 3))) < 1e-12
    assert abs(lrs[9] - 0.5e-4 * (1 - (1 - 9 / 10) * (1 - 1 / 3))) < 1e-12    # epoch 1 regular lr = base*0.5
    assert abs(lrs[10] - 0.5e-4) < 1e-12
    assert abs(lrs[12] - 0.25e-4) < 1e-12
    for e in (1, 2, 3):
        assert os.path.exists(tmp_path / ("epoch_%d.pth" % e))
    ckpt = torch.load(tmp_path / "epoch_3.pth", weights_only=False)
    assert set(ckpt) == {"meta", "state_dict", "optimizer"} and ckpt["meta"]["epoch"] == 3 and ckpt["meta"]["iter"] == 18
    model2 = Toy()
    load_checkpoint(model2, str(tmp_path / "epoch_3.pth"), map_location="cpu")
    assert torch.equal(model2.fc.weight, model.fc.weight)
    runner2 = Runner(Toy(), _bp, torch.optim.Adam(model2.parameters(), lr=1e-4), str(tmp_path), "INFO")
    runner2.resume(str(tmp_path / "epoch_2.pth"), map_location="cpu")
    assert runner2.epoch == 2 and runner2.iter == 12
    assert any(f.endswith(".log.json") for f in os.listdir(tmp_path))


def test_resume_keeps_the_execution_flags_of_the_live_optimizer(tmp_path):
    """A reference / older checkpoint's param_groups carry no capturable / fused / foreach keys; torch's load_state_dict then
    defaults them (capturable False, fused None), which on a GPU would refuse graph capture and put a host sync into every
    eager step.  Runner.resume keeps the flags of the optimiser it was given (the flat store did so already)."""
    model = Toy()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    runner = Runner(model, _bp, opt, str(tmp_path), "INFO")
    runner.register_training_hooks(dict(policy="step", step=[1]), dict(grad_clip=None), dict(interval=1), dict(interval=50, hooks=[]))
    runner.run([_loader()], [("train", 1)], 1)
    ckpt = torch.load(tmp_path / "epoch_1.pth", weights_only=False)
    for g in ckpt["optimizer"]["param_groups"]:            # what the reference's files look like
        for k in ("capturable", "fused", "foreach", "differentiable"):
            g.pop(k, None)
    torch.save(ckpt, tmp_path / "old_format.pth")
    model2 = Toy()
    opt2 = torch.optim.Adam(model2.parameters(), lr=1e-4, foreach=True)        # (capturable needs a device: GPU test)
    runner2 = Runner(model2, _bp, opt2, str(tmp_path), "INFO")
    runner2.resume(str(tmp_path / "old_format.pth"), map_location="cpu")
    g = opt2.param_groups[0]
    assert g["foreach"] is True
    assert runner2.iter == 6 and abs(g["lr"] - ckpt["optimizer"]["param_groups"][0]["lr"]) < 1e-12
    opt2.zero_grad()
    model2.fc.weight.sum().backward()
    opt2.step()                                           # state from the file + the live flags work together
