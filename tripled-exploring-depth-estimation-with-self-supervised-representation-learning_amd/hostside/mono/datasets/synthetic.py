"""Synthetic KITTI-shaped frame triplets (no dataset is available offline).  This is the input
contract of the hot path: ("color", f, 0) / ("color_aug", f, 0) float
[3,H,W] in [0,1] for f in frame_ids, ("mask", 0, 0) with `erase_count` zeroed squares,
K = normalised KITTI intrinsics scaled by (W, H) and inv_K = pinv(K)
(reference: mono/datasets/kitti_dataset.py:126-129,167-182, mono/datasets/mono_dataset.py:174-180)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

KITTI_K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)


def intrinsics(height, width):
    K = KITTI_K.copy()
    K[0, :] *= width
    K[1, :] *= height
    return K, np.linalg.pinv(K)


def _smooth_canvas(g, h, w):
    """Low-pass noise + fine texture: temporally coherent frames are crops of one canvas."""
    coarse = torch.rand(1, 3, h // 8 + 2, w // 8 + 2, generator=g)
    img = F.interpolate(coarse, size=(h, w), mode="bicubic", align_corners=False)[0]
    img = img + 0.08 * torch.rand(3, h, w, generator=g)
    return img.clamp_(0, 1)


def make_sample(seed, height, width, frame_ids=(0, -1, 1), erase_shape=(16, 16), erase_count=16, max_shift=4,
                with_mask=True, wire="float32", augment=False, coherent=True):
    """coherent=False: every frame is independent U[0,1) noise -- the adversarial input of SURVEY.md section 8d
    (no frame explains another, so the arg-min and the bilinear taps see uncorrelated values)."""
    g = torch.Generator().manual_seed(int(seed))
    canvas = _smooth_canvas(g, height + 2 * max_shift, width + 2 * max_shift)
    sample = {}
    for f in frame_ids:
        if f == 0:
            dy = dx = max_shift
        else:
            dy = int(torch.randint(0, 2 * max_shift + 1, (1,), generator=g))
            dx = int(torch.randint(0, 2 * max_shift + 1, (1,), generator=g))
        if coherent:
            img = canvas[:, dy:dy + height, dx:dx + width].contiguous()
        else:
            img = torch.rand(3, height, width, generator=g)
        if wire == "uint8":      # the byte wire format of the KITTI loader (kitti_dataset.py), expanded on the device
            sample[("color_u8", f)] = (img * 255.0).round().clamp_(0, 255).to(torch.uint8)
            continue
        sample[("color", f, 0)] = img
        sample[("color_aug", f, 0)] = img.clone()
    if wire == "uint8":
        row = torch.zeros(9)
        if augment and float(torch.rand(1, generator=g)) > 0.5:
            u = lambda lo, hi: lo + (hi - lo) * float(torch.rand(1, generator=g))
            row = torch.tensor([1.0] + [float(v) for v in torch.randperm(4, generator=g)] +
                               [u(0.8, 1.2), u(0.8, 1.2), u(0.8, 1.2), u(-0.1, 0.1)])
        sample["aug"] = row
    if with_mask:
        mask = torch.ones(3, height, width)
        eh, ew = erase_shape
        for _ in range(erase_count):
            y0 = int(torch.randint(0, max(1, height - eh), (1,), generator=g))
            x0 = int(torch.randint(0, max(1, width - ew), (1,), generator=g))
            mask[:, y0:y0 + eh, x0:x0 + ew] = 0
        sample[("mask", 0, 0)] = mask
    K, inv_K = intrinsics(height, width)
    sample["K"] = torch.from_numpy(K)
    sample["inv_K"] = torch.from_numpy(inv_K)
    return sample


class SyntheticTripletDataset(Dataset):
    def __init__(self, length, height, width, frame_ids=(0, -1, 1), erase_shape=(16, 16), erase_count=16,
                 seed=1000, with_mask=True, with_gt=False, wire="float32", augment=False):
        self.wire, self.augment = wire, augment
        self.length, self.height, self.width = length, height, width
        self.frame_ids, self.erase_shape, self.erase_count = tuple(frame_ids), tuple(erase_shape), erase_count
        self.seed, self.with_mask, self.with_gt = seed, with_mask, with_gt
        self.flag = np.zeros(length, dtype=np.int64)

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        s = make_sample(self.seed + idx, self.height, self.width, self.frame_ids, self.erase_shape,
                        self.erase_count, with_mask=self.with_mask, wire=self.wire, augment=self.augment)
        if self.with_gt:
            g = torch.Generator().manual_seed(self.seed + idx + 7)
            s["gt_depth"] = 2.0 + 60.0 * torch.rand(self.height, self.width, generator=g)
        return s


def synthetic_batch(batch_size, height, width, seed=1000, device=None, **kw):
    """A collated batch (dict of [B,...] tensors), optionally moved to `device`."""
    samples = [make_sample(seed + i, height, width, **kw) for i in range(batch_size)]
    batch = {k: torch.stack([s[k] for s in samples], 0) for k in samples[0]}
    if device is not None:
        batch = {k: v.to(device) for k, v in batch.items()}
    return batch


class ResidentBatches:
    """``n_iters`` iterations over ONE batch that already sits in device memory, shaped like a loader (``len``,
    iteration, ``sampler``) and like a dataset that brings its own loader (``as_loader``; mono.apis.trainer).  It is how
    bench.py drives the Runner path (``train_mono``) with the inputs resident in HBM, as the metric prescribes."""
    device_resident = True
    sampler = None

    def __init__(self, batch, n_iters, timed_from=None):
        """``timed_from``: iterations ``timed_from .. n_iters-1`` are timed (device-synchronised, barrier across the ranks,
        on both sides); the generator resumes after an iteration's hooks have run, so the bracket covers whole Runner
        iterations.  ``elapsed`` (seconds) and ``timed_iters`` are set when the epoch ends."""
        self.batch, self.n_iters = batch, int(n_iters)
        self.timed_from, self.elapsed, self.timed_iters = timed_from, None, 0

    def as_loader(self, imgs_per_gpu):
        if int(self.batch["K"].shape[0]) != int(imgs_per_gpu):
            raise ValueError("resident batch holds %d samples, the config asks for %d per GPU" % (self.batch["K"].shape[0], imgs_per_gpu))
        return self

    def __len__(self):
        return self.n_iters

    def _fence(self):
        import time
        import torch.distributed as dist
        torch.cuda.synchronize()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()
            torch.cuda.synchronize()
        return time.perf_counter()

    def __iter__(self):
        t0 = None
        for i in range(self.n_iters):
            if self.timed_from is not None and i == self.timed_from:
                t0 = self._fence()
            yield dict(self.batch)
        if t0 is not None:
            self.elapsed = self._fence() - t0
            self.timed_iters = self.n_iters - self.timed_from
