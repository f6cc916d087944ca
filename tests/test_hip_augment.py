"""GPU parity of the device-side input expansion (csrc/td_augment.hip through the C ABI) against the float oracle
(oracle/augment.py, pinned against PIL in tests/test_augment_cpu.py), and the 'uint8' wire format end to end."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import augment  # noqa: E402


def _aug_rows(g, n):
    rows = []
    for i in range(n):
        if i % 4 == 3:
            rows.append(torch.zeros(9))
            continue
        u = lambda lo, hi: lo + (hi - lo) * float(torch.rand(1, generator=g))
        rows.append(torch.tensor([1.0] + [float(v) for v in torch.randperm(4, generator=g)] +
                                 [u(0.8, 1.2), u(0.8, 1.2), u(0.8, 1.2), u(-0.1, 0.1)]))
    return torch.stack(rows, 0)


@pytest.mark.parametrize("N,H,W", [(5, 16, 24), (3, 33, 70), (36, 192, 640)])       # last: 3 frames x 12 samples at C2
def test_color_jitter_matches_oracle(N, H, W):
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(N)
    frames = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8)
    frames[0, :, :2] = 0                     # grey / black pixels: the hue stage's max == min branch
    frames[0, :, 2:4] = 200
    aug = _aug_rows(g, N)
    color, color_aug = ops.color_jitter_expand(frames.cuda(), aug.cuda())
    ref_c, ref_a = augment.expand_frames(frames, aug)
    assert torch.equal(color.cpu(), ref_c)
    d = (color_aug.cpu() - ref_a).abs()
    # float re-association (the per-image grey mean is a different summation order) and hue-sector boundaries
    assert float(d.mean()) < 1e-6 and float((d > 1e-4).float().mean()) < 1e-5, (float(d.max()), float(d.mean()))
    assert float(color_aug.min()) >= 0 and float(color_aug.max()) <= 1


def test_uint8_wire_training_step(tmp_path):
    """train_mono on the byte wire format: loader -> pinned uint8 -> device -> td_color_jitter -> model, strict dispatch."""
    import tripled_amd  # noqa: F401
    from mmcv import Config
    from mono.apis import train_mono
    from mono.datasets import get_dataset
    from mono.model import MONO
    from tripled_amd import dispatch
    H, W, B = 96, 160, 2
    cfg = Config(dict(
        data=dict(name="synthetic", split="exp", height=H, width=W, frame_ids=[0, -1, 1], in_path=None,
                  gt_depth_path=None, png=True, stereo_scale=False, erase_shape=[8, 8], erase_count=4,
                  synthetic_length=4, synthetic_val_length=2, wire="uint8"),
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
        checkpoint_config=dict(interval=1), log_config=dict(interval=1, hooks=[dict(type="TextLoggerHook")]),
        dist_params=dict(backend="nccl"), log_level="INFO", workflow=[("train", 1)], syncbn=False,
        work_dir=str(tmp_path), gpus=[0], amp="bf16", channels_last=True, strict_dispatch=True))
    train = get_dataset(cfg.data, training=True)
    assert train[0][("color_u8", 0)].dtype == torch.uint8 and ("color", 0, 0) not in train[0]
    dispatch.reset()
    torch.manual_seed(0)
    model = MONO.module_dict[cfg.model["name"]](cfg.model)
    train_mono(model, train, get_dataset(cfg.data, training=False), cfg, distributed=False, validate=True)
    dispatch.set_strict(False)
    assert dispatch.hip_calls["td_color_jitter"] >= 2 and sum(dispatch.fallbacks.values()) == 0
    assert (tmp_path / "epoch_1.pth").exists()
