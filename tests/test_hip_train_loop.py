"""GPU integration: the reference's training entry (mono.apis.train_mono via the Runner shim) for one short
epoch on synthetic triplets -- optimiser hook, LR warm-up, text log, checkpoint, per-epoch depth evaluation."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_mono_one_epoch(tmp_path):
    import tripled_amd  # noqa: F401
    from mmcv import Config
    from mono.apis import train_mono
    from mono.datasets import get_dataset
    from mono.model import MONO
    H, W, B = 96, 160, 2
    cfg = Config(dict(
        data=dict(name="synthetic", split="exp", height=H, width=W, frame_ids=[0, -1, 1], in_path=None,
                  gt_depth_path=None, png=True, stereo_scale=False, erase_shape=[8, 8], erase_count=4,
                  synthetic_length=8, synthetic_val_length=3),
        model=dict(name="mono_fm_joint_inpaint_disentangle", depth_num_layers=18, pose_num_layers=18,
                   extractor_num_layers=18, frame_ids=[0, -1, 1], imgs_per_gpu=B, height=H, width=W,
                   scales=[0, 1, 2, 3], min_depth=0.1, max_depth=100.0, depth_pretrained_path=None,
                   pose_pretrained_path=None, extractor_pretrained_path=None, automask=True, disp_norm=True,
                   dis=1e-3, cvt=1e-3, perception_weight=1e-3, smoothness_weight=1e-3, auto_res_weight=5e-3,
                   disentangle_layers=[False, False, False, False, True], skip_connection_multiplier=1,
                   depth_skip_type=None, color_skip_type=None, color_skip_layers=[False] * 4,
                   depth_use_shuffle=False, depth_disentangle_type="use_half", freeze_extractor=False),
        resume_from=None, finetune=None, load_from=None, total_epochs=1, imgs_per_gpu=B, learning_rate=1e-4,
        workers_per_gpu=0, validate=True, validate_interval=1,
        optimizer=dict(type="Adam", lr=1e-4, weight_decay=0),
        optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)),
        lr_config=dict(policy="step", warmup="linear", warmup_iters=3, warmup_ratio=1.0 / 3, step=[10, 20], gamma=0.5),
        checkpoint_config=dict(interval=1), log_config=dict(interval=2, hooks=[dict(type="TextLoggerHook")]),
        dist_params=dict(backend="nccl"), log_level="INFO", workflow=[("train", 1)], syncbn=False,
        work_dir=str(tmp_path), gpus=[0],
        # the execution mode of the benchmark, reached through train_mono (mono/apis/trainer.py::configure_execution);
        # strict: any HIP tensor that would fall back to an ATen composition raises
        amp="bf16", channels_last=True, strict_dispatch=True))
    from tripled_amd import dispatch
    dispatch.reset()
    torch.manual_seed(0)
    model = MONO.module_dict[cfg.model["name"]](cfg.model)
    before = model.DepthDecoder.disp1[0].conv.weight.detach().clone()
    train_mono(model, get_dataset(cfg.data, training=True), get_dataset(cfg.data, training=False), cfg,
               distributed=False, validate=True)
    dispatch.set_strict(False)
    assert sum(dispatch.fallbacks.values()) == 0, dict(dispatch.fallbacks)
    assert dispatch.hip_calls["td_bn_fwd"] > 0 and dispatch.hip_calls["td_maxpool5_fwd"] > 0      # the fast path ran
    assert next(model.parameters()).is_cuda and model.DepthEncoder.encoder.layer1[0].conv1.weight.is_contiguous(
        memory_format=torch.channels_last)
    assert os.path.exists(tmp_path / "epoch_1.pth")
    ckpt = torch.load(tmp_path / "epoch_1.pth", weights_only=False)
    assert ckpt["meta"]["iter"] == 4 and "DepthDecoder.disp1.0.conv.weight" in ckpt["state_dict"]
    assert not torch.equal(ckpt["state_dict"]["DepthDecoder.disp1.0.conv.weight"], before)
    # the flat mixed-precision store is the default in this mode; the file keeps the reference's layout: fp32 weights under
    # the module's keys, per-parameter Adam state in model.parameters() order
    flat = model._flat_store
    assert model.DepthDecoder.disp1[0].conv.weight.dtype == torch.bfloat16 and flat.n_lp > 0
    assert all(v.dtype == torch.float32 for v in ckpt["state_dict"].values() if v.is_floating_point())
    n_train = len([p for p in model.parameters() if p.requires_grad])
    assert len(ckpt["optimizer"]["state"]) == n_train and ckpt["optimizer"]["param_groups"][0]["params"] == list(range(n_train))
    assert ckpt["optimizer"]["state"][0]["exp_avg"].shape == next(model.parameters()).shape
    logs = [f for f in os.listdir(tmp_path) if f.endswith(".log.json")]
    records = [json.loads(l) for l in open(tmp_path / logs[0])]
    train = [r for r in records if r["mode"] == "train" and "loss" in r]
    assert train and all(k in train[0] for k in ("loss", "('min_reconstruct_loss', 0)", "('smooth_loss', 3)", "lr"))
    assert all(r["loss"] == r["loss"] for r in train)            # finite
    assert any("abs_rel" in r for r in records)                  # the evaluation hook ran and published metrics
    # resume into a fresh model: one more epoch from the file
    cfg.resume_from = str(tmp_path / "epoch_1.pth")
    cfg.total_epochs = 2
    torch.manual_seed(1)
    model2 = MONO.module_dict[cfg.model["name"]](cfg.model)
    train_mono(model2, get_dataset(cfg.data, training=True), get_dataset(cfg.data, training=False), cfg,
               distributed=False, validate=False)
    ckpt2 = torch.load(tmp_path / "epoch_2.pth", weights_only=False)
    assert ckpt2["meta"]["iter"] == 8 and float(ckpt2["optimizer"]["state"][0]["step"]) == 8.0
    assert not torch.equal(ckpt2["state_dict"]["DepthDecoder.disp1.0.conv.weight"], ckpt["state_dict"]["DepthDecoder.disp1.0.conv.weight"])
