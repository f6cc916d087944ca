"""Rank bring-up for the one-process-per-GPU data-parallel run, plus seeding and the root logger.

Public names and call signatures are the ones train.py uses (reference: mono/apis/env.py --
``init_dist(launcher, backend, **kwargs)``, ``set_random_seed(seed)``, ``get_root_logger(log_level)``).
The bring-up itself is written for this build's launch model: ``python -m torch.distributed.run`` (or any
launcher that exports RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT) starts one rank per MI355X;
the rank binds its GPU FIRST and hands the bound device to the process group, so RCCL builds its xGMI
communicator eagerly on the right device instead of guessing it at the first collective.
"""
import logging
import os
import random

import numpy as np
import torch
import torch.distributed as dist

_LAUNCHERS = ("pytorch",)


class RankEnv:
    """The launcher-provided coordinates of this process."""

    def __init__(self, environ=os.environ):
        try:
            self.rank = int(environ["RANK"])
            self.world_size = int(environ["WORLD_SIZE"])
        except KeyError as e:
            raise RuntimeError("distributed launch needs %s in the environment (start the job with "
                               "`python -m torch.distributed.run --nproc-per-node N train.py ... --launcher pytorch`)"
                               % e.args[0]) from e
        # LOCAL_RANK is what torch.distributed.run exports; a launcher without it runs one node
        self.local_rank = int(environ.get("LOCAL_RANK", self.rank))

    def device(self):
        """The GPU this rank owns (None on a CPU-only host, e.g. the gloo tests)."""
        n = torch.cuda.device_count()
        return torch.device("cuda", self.local_rank % n) if n > 0 and torch.cuda.is_available() else None


def init_dist(launcher, backend="nccl", **kwargs):
    """Join the job's process group.  ``backend='nccl'`` is RCCL on ROCm; on a host without a GPU the
    group falls back to gloo so that the same entry point serves the CPU tests."""
    if launcher not in _LAUNCHERS:
        raise ValueError("unsupported launcher %r: this build starts its ranks with torch.distributed.run "
                         "(launcher='pytorch')" % (launcher,))
    if dist.is_initialized():
        return
    # the host driver only supports dmabuf IPC; RCCL's intra-node transport needs this before the first HIP call
    # (tripled_amd/__init__ sets it on import as well; RankEnv.device() below may already initialise HIP)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env = RankEnv()
    dev = env.device()
    if dev is None:
        dist.init_process_group("gloo" if backend == "nccl" else backend, **kwargs)
        return
    torch.cuda.set_device(dev)
    if backend == "nccl":
        kwargs.setdefault("device_id", dev)
    dist.init_process_group(backend, **kwargs)


def set_random_seed(seed):
    """Seed python, numpy and torch (host and every visible GPU generator)."""
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_root_logger(log_level=logging.INFO):
    """Root logger: full output on rank 0, errors only elsewhere."""
    logger = logging.getLogger()
    if not logger.hasHandlers():
        logging.basicConfig(format="%(asctime)s - %(levelname)s - %(message)s", level=log_level)
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if rank != 0:
        logger.setLevel(logging.ERROR)
    return logger
