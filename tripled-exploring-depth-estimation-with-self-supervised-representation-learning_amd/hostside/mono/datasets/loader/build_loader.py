"""DataLoader construction (reference: mono/datasets/loader/build_loader.py:18-54)."""
from functools import partial

import torch
from mmcv.parallel import collate
from mmcv.runner import get_dist_info
from torch.utils.data import DataLoader

from .sampler import DistributedGroupSampler, DistributedSampler, GroupSampler


def build_dataloader(dataset, imgs_per_gpu, workers_per_gpu, num_gpus=1, dist=True, **kwargs):
    shuffle = kwargs.pop("shuffle", True)
    if dist:
        rank, world_size = get_dist_info()
        if shuffle:
            sampler = DistributedGroupSampler(dataset, imgs_per_gpu, world_size, rank)
        else:
            sampler = DistributedSampler(dataset, world_size, rank, shuffle=False)
        batch_size, num_workers = imgs_per_gpu, workers_per_gpu
    else:
        sampler = GroupSampler(dataset, imgs_per_gpu) if shuffle else None
        batch_size, num_workers = num_gpus * imgs_per_gpu, num_gpus * workers_per_gpu
    # the reference never pins memory (pin_memory=False, :49), which makes every H2D copy in
    # change_input_variable synchronous; pinned staging lets the copies overlap with compute
    kwargs.setdefault("pin_memory", torch.cuda.is_available())
    if num_workers > 0:
        kwargs.setdefault("persistent_workers", True)
    return DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
                      collate_fn=partial(collate, samples_per_gpu=imgs_per_gpu), drop_last=True, **kwargs)
