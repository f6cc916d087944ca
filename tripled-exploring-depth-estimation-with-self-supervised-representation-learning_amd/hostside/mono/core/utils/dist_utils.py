"""Gradient synchronisation hook (reference: mono/core/utils/dist_utils.py).

The reference does two all-reduces per step: DDP's bucketed one during backward and then this
hook's flat one over all 337 MB afterwards (dist_utils.py:27-28,57) -- averaging tensors that
are already identical on every rank.  Here the DP wrapper (mmcv.parallel.MMDistributedDataParallel,
this build's RCCL engine) reports whether the gradients are already synchronised and the second
pass is skipped; ``allreduce_grads`` itself keeps the reference's semantics for callers that use
it on a bare model."""
from collections import OrderedDict

import torch
import torch.distributed as dist
from mmcv.runner import OptimizerHook


def _allreduce_coalesced(tensors, world_size, bucket_size_mb=-1):
    if bucket_size_mb > 0:
        limit = bucket_size_mb * 1024 * 1024
        buckets, cur, size = [], [], 0
        for t in tensors:
            cur.append(t)
            size += t.numel() * t.element_size()
            if size >= limit:
                buckets.append(cur)
                cur, size = [], 0
        if cur:
            buckets.append(cur)
    else:
        by_type = OrderedDict()
        for t in tensors:
            by_type.setdefault(t.type(), []).append(t)
        buckets = list(by_type.values())
    for bucket in buckets:
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat)
        flat.div_(world_size)
        off = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def allreduce_grads(model, coalesce=True, bucket_size_mb=-1):
    grads = [p.grad.data for p in model.parameters() if p.requires_grad and p.grad is not None]
    world_size = dist.get_world_size()
    if coalesce:
        _allreduce_coalesced(grads, world_size, bucket_size_mb)
    else:
        for t in grads:
            dist.all_reduce(t.div_(world_size))


class DistOptimizerHook(OptimizerHook):
    def __init__(self, grad_clip=None, coalesce=True, bucket_size_mb=-1):
        self.grad_clip = grad_clip
        self.coalesce = coalesce
        self.bucket_size_mb = bucket_size_mb

    def after_train_iter(self, runner):
        model = runner.model
        engine = getattr(model, "reducer", None)
        if engine is None:
            runner.optimizer.zero_grad()      # the DP engine zeroes its flat buffer in forward()
        runner.outputs["loss"].backward()
        synced = getattr(model, "grads_synchronised", None)
        if not (synced is not None and synced()):
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                allreduce_grads(model, self.coalesce, self.bucket_size_mb)
        if self.grad_clip is not None:
            self.clip_grads(model.parameters())
        runner.optimizer.step()


class FlatOptimizerHook(OptimizerHook):
    """The iteration of DistOptimizerHook (zero_grad -> backward -> gradient average over the ranks -> clip -> step,
    reference: dist_utils.py:54-60) on the flat mixed-precision parameter store (tripled_amd.flat_amp): gradients are
    gathered into one flat fp32 buffer after backward, all-reduced there in a few large buckets, clipped with one norm
    and stepped by one single-tensor Adam; the bf16 working copy of the convolution weights is refreshed by one cast."""

    def __init__(self, flat, grad_clip=None, **_unused):
        self.flat = flat
        self.grad_clip = grad_clip
        if grad_clip is not None and grad_clip.get("norm_type", 2) != 2:
            raise ValueError("the flat store clips with the 2-norm")
        self.flat.max_norm = grad_clip["max_norm"] if grad_clip else None

    def after_train_iter(self, runner):
        self.flat.zero_grad()
        runner.outputs["loss"].backward()
        self.flat.collect()
        self.flat.allreduce()
        self.flat.step()


class IterationDoneHook(OptimizerHook):
    """The optimiser hook's slot when the Runner's batch_processor is ``tripled_amd.step.RunnerIteration``: zero_grad,
    backward, gradient exchange, clip and step (reference: dist_utils.py:54-60) have already executed as part of the
    replayed HIP graph when ``after_train_iter`` comes, so nothing is left to do here.  The LR schedule still acts in
    ``before_train_iter`` (on the optimiser's device-side ``lr`` tensor), i.e. before the iteration, as in the reference."""

    def __init__(self, iteration, grad_clip=None, **_unused):
        self.iteration = iteration
        self.grad_clip = grad_clip

    def after_train_iter(self, runner):
        pass
