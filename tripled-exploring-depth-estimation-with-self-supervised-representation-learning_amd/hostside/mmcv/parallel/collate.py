"""collate / scatter for the plain tensor/ndarray dict batches the depth datasets produce
(no DataContainer objects on this path)."""
import collections.abc as cabc

import numpy as np
import torch
from torch.utils.data.dataloader import default_collate


def collate(batch, samples_per_gpu=1):
    if not isinstance(batch, cabc.Sequence):
        raise TypeError("{} is not supported.".format(type(batch)))
    first = batch[0]
    if isinstance(first, cabc.Mapping):
        return {key: collate([d[key] for d in batch], samples_per_gpu) for key in first}
    if isinstance(first, (tuple, list)) and not isinstance(first, str):
        return [collate(samples, samples_per_gpu) for samples in zip(*batch)]
    if isinstance(first, np.ndarray):
        return default_collate([torch.from_numpy(np.ascontiguousarray(b)) for b in batch])
    return default_collate(batch)


def _to_device(obj, device):
    if isinstance(obj, torch.Tensor):
        return obj.to(device, non_blocking=True)
    if isinstance(obj, cabc.Mapping):
        return {k: _to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(v, device) for v in obj)
    return obj


def scatter(inputs, target_gpus, dim=0):
    """One process drives one GPU here, so scattering is a move to that device."""
    if len(target_gpus) != 1:
        raise NotImplementedError("single-process multi-GPU scatter is legacy nn.DataParallel behaviour; "
                                  "launch one process per GPU instead")
    dev = torch.device("cuda", target_gpus[0]) if target_gpus[0] >= 0 else torch.device("cpu")
    return (_to_device(inputs, dev),)


def scatter_kwargs(inputs, kwargs, target_gpus, dim=0):
    ins = scatter(inputs, target_gpus, dim) if inputs else ((),)
    kw = scatter(kwargs, target_gpus, dim) if kwargs else ({},)
    return ins, kw
