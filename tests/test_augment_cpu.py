"""Input pipeline, host side: the float colour-jitter restatement (oracle/augment.py, what csrc/td_augment.hip
implements) against PIL's own 8-bit primitives (what the reference's torchvision ColorJitter applies to its PIL
frames, mono/datasets/mono_dataset.py:83-101,146-152), and the 'uint8' wire format of the loaders."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

import tripled_amd  # noqa: F401
from mono.datasets import kitti_dataset as kd
from oracle import augment
from tests.util import smooth_image


def _image(seed, h=64, w=96):
    g = torch.Generator().manual_seed(seed)
    img = (smooth_image(g, 1, 3, h, w)[0] * 255).round().clamp(0, 255).to(torch.uint8)
    return img, Image.fromarray(img.permute(1, 2, 0).numpy(), "RGB")


@pytest.mark.parametrize("op,max_tol", [(0, 1.5 / 255), (1, 1.5 / 255), (2, 1.5 / 255), (3, 0.08)])
def test_single_stage_matches_pil(op, max_tol):
    """brightness / contrast / saturation agree with PIL to one 8-bit step (PIL rounds the result to uint8); PIL's hue
    shift works on an 8-bit HSV image with an integer shift, so that stage agrees to ~1 % on average and to a few 8-bit
    steps at low-saturation pixels (the same gap torchvision has between its own PIL and tensor back-ends)."""
    u8, pil = _image(1 + op)
    j = kd.ColorJitter((1.13, 1.13), (0.87, 0.87), (1.17, 1.17), (0.06, 0.06))
    j.order = [op]
    ref = kd.to_tensor(j(pil))
    got = augment.color_jitter(u8.float().unsqueeze(0) / 255, j.order, *j.factors)[0]
    d = (ref - got).abs()
    assert float(d.max()) <= max_tol and float(d.mean()) < 0.008


def test_random_chains_match_pil():
    u8, pil = _image(7)
    torch.manual_seed(3)
    for _ in range(20):
        j = kd.ColorJitter(kd.MonoDataset.brightness, kd.MonoDataset.contrast, kd.MonoDataset.saturation, kd.MonoDataset.hue)
        ref = kd.to_tensor(j(pil))
        got = augment.color_jitter(u8.float().unsqueeze(0) / 255, j.order, *j.factors)[0]
        d = (ref - got).abs()
        assert float(d.mean()) < 0.012 and float(d.max()) < 0.1, (j.order, j.factors, float(d.mean()), float(d.max()))


def test_expand_frames_contract():
    u8, _ = _image(9, 16, 24)
    frames = torch.stack([u8, u8.flip(2)], 0)
    aug = torch.tensor([[0, 0, 1, 2, 3, 1, 1, 1, 0], [1, 2, 0, 3, 1, 1.1, 0.9, 1.2, -0.05]], dtype=torch.float32)
    color, color_aug = augment.expand_frames(frames, aug)
    assert torch.equal(color, frames.float() / 255) and torch.equal(color_aug[0], color[0])     # disabled row: identity
    assert not torch.equal(color_aug[1], color[1]) and float(color_aug.min()) >= 0 and float(color_aug.max()) <= 1


def test_kitti_loader_uint8_wire(tmp_path):
    """wire='uint8': ("color_u8", f) bytes + one jitter row per sample instead of two float copies per frame."""
    root = tmp_path / "kitti"
    seq = root / "2011_09_26/2011_09_26_drive_0001_sync/image_02/data"
    os.makedirs(seq)
    rng = np.random.RandomState(0)
    for i in range(3):
        Image.fromarray(rng.randint(0, 255, (40, 120, 3), dtype=np.uint8)).save(seq / ("%010d.png" % i))
    files = ["2011_09_26/2011_09_26_drive_0001_sync 1 l"]
    cfg = dict(wire="uint8", erase_shape=[4, 4], erase_count=3)
    ds = kd.KITTIInpaintDataset(str(root), files, 32, 96, [0, -1, 1], cfg=cfg, is_train=True, img_ext=".png")
    s = ds[0]
    assert all(s[("color_u8", f)].dtype == torch.uint8 and s[("color_u8", f)].shape == (3, 32, 96) for f in (0, -1, 1))
    assert s["aug"].shape == (9,) and ("color", 0, 0) not in s and s[("mask", 0, 0)].shape == (3, 32, 96)
    ref = kd.KITTIInpaintDataset(str(root), files, 32, 96, [0, -1, 1], cfg=dict(cfg, wire="float32"), is_train=False,
                                 img_ext=".png")[0]
    u8 = kd.KITTIInpaintDataset(str(root), files, 32, 96, [0, -1, 1], cfg=cfg, is_train=False, img_ext=".png")[0]
    assert torch.equal(u8[("color_u8", 0)].float() / 255, ref[("color", 0, 0)])       # same pixels, 1/8 of the bytes


def test_device_expand_refuses_host_tensors():
    from mono.datasets import expand_device_batch
    from tripled_amd import native
    batch = {("color_u8", 0): torch.zeros(1, 3, 4, 4, dtype=torch.uint8), "aug": torch.zeros(1, 9)}
    with pytest.raises(native.NativeLibraryError):
        expand_device_batch(batch)
