"""KITTI dataset loader on a tiny synthetic directory tree (PNG files written with PIL)."""
import os

import numpy as np
import torch
from PIL import Image

import tripled_amd  # noqa: F401
from mmcv import ConfigDict
from mono.datasets.kitti_dataset import KITTIInpaintDataset, KITTIRAWDataset


def _make_tree(root, n=4):
    d = os.path.join(root, "2011_09_26/2011_09_26_drive_0001_sync/image_02/data")
    os.makedirs(d)
    rng = np.random.RandomState(0)
    for i in range(n):
        Image.fromarray(rng.randint(0, 255, size=(75, 248, 3), dtype=np.uint8)).save(os.path.join(d, "%010d.png" % i))
    return ["2011_09_26/2011_09_26_drive_0001_sync %d l" % i for i in range(n)]


def test_kitti_triplet_contract(tmp_path):
    files = _make_tree(str(tmp_path))
    cfg = ConfigDict(erase_shape=[8, 8], erase_count=4)
    ds = KITTIInpaintDataset(str(tmp_path), files, 32, 96, [0, -1, 1], cfg=cfg, is_train=True, img_ext=".png")
    torch.manual_seed(0)
    s = ds[1]
    for f in (0, -1, 1):
        assert s[("color", f, 0)].shape == (3, 32, 96) and s[("color_aug", f, 0)].shape == (3, 32, 96)
        assert 0.0 <= float(s[("color", f, 0)].min()) and float(s[("color", f, 0)].max()) <= 1.0
    assert s[("mask", 0, 0)].dtype == torch.uint8 and 0 < int((s[("mask", 0, 0)] == 0).sum()) <= 3 * 4 * 64
    K = s["K"].numpy()
    np.testing.assert_allclose(K[0, 0], 0.58 * 96, rtol=1e-6)
    np.testing.assert_allclose(K[1, 2], 0.5 * 32, rtol=1e-6)
    np.testing.assert_allclose(s["inv_K"].numpy() @ K, np.eye(4), atol=1e-5)
    # sequence boundaries repeat the centre frame (frame -1 of index 0 does not exist)
    first = KITTIRAWDataset(str(tmp_path), files, 32, 96, [0, -1, 1], is_train=False, img_ext=".png")[0]
    assert torch.equal(first[("color", -1, 0)], first[("color", 0, 0)])
    assert ds.flag.shape == (4,)


def test_erase_masks_consume_the_generator_like_the_reference():
    """KITTIInpaintDataset.preprocess_masks (reference kitti_dataset.py:167-182) draws `erase_count` (row, col) pairs with
    torch.LongTensor(1).random_(0, size - erase - 1) from the global generator; the mirror draws with torch.randint.  Same seed ->
    the same rectangles (the draw sequence is restated here from the reference's lines; its module needs torchvision / cv2 and
    cannot be imported in this image), and erase_count == 1 is the reference's centred square."""
    import types
    shape, eh, ew, count = (3, 192, 640), 16, 16, 16
    for seed in (0, 5, 123):
        fake = types.SimpleNamespace(cfg=ConfigDict(erase_shape=[eh, ew], erase_count=count))
        inputs = {("color", 0, 0): torch.zeros(shape)}
        torch.manual_seed(seed)
        KITTIInpaintDataset.postprocess(fake, inputs)
        expected = torch.ones(shape, dtype=torch.uint8)
        torch.manual_seed(seed)
        for _ in range(count):
            row = torch.LongTensor(1).random_(0, shape[1] - eh - 1)[0]
            col = torch.LongTensor(1).random_(0, shape[2] - ew - 1)[0]
            expected[:, row:row + eh, col:col + ew] = 0
        assert torch.equal(inputs[("mask", 0, 0)], expected), seed
    fake = types.SimpleNamespace(cfg=ConfigDict(erase_shape=[64, 64], erase_count=1))
    inputs = {("color", 0, 0): torch.zeros(shape)}
    KITTIInpaintDataset.postprocess(fake, inputs)
    centre = torch.ones(shape, dtype=torch.uint8)
    centre[:, 64:128, 64:128] = 0                       # offset = (192 - 64) / 2 on BOTH axes, as the reference computes it
    assert torch.equal(inputs[("mask", 0, 0)], centre)
