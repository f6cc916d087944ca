"""Process-group bring-up, seeding, logging (reference: mono/apis/env.py).  One process per GPU;
backend 'nccl' on ROCm is RCCL over xGMI."""
import logging
import os
import random
import subprocess

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from mmcv.runner import get_dist_info


def init_dist(launcher, backend="nccl", **kwargs):
    if mp.get_start_method(allow_none=True) is None:
        mp.set_start_method("spawn")
    if launcher == "pytorch":
        _init_dist_pytorch(backend, **kwargs)
    elif launcher == "mpi":
        raise NotImplementedError
    elif launcher == "slurm":
        _init_dist_slurm(backend, **kwargs)
    else:
        raise ValueError("Invalid launcher type: {}".format(launcher))


def _bind_device(index):
    n = torch.cuda.device_count()
    if n > 0:
        torch.cuda.set_device(index % n)


def _init_dist_pytorch(backend, **kwargs):
    """reference :30-35 -- RANK / WORLD_SIZE / MASTER_* come from the launcher; the device is
    LOCAL_RANK when the launcher provides it, otherwise rank % device_count as in the reference."""
    rank = int(os.environ["RANK"])
    _bind_device(int(os.environ.get("LOCAL_RANK", rank)))
    if backend == "nccl" and not torch.cuda.is_available():
        backend = "gloo"        # CPU-only host (tests): RCCL needs a GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC, required by RCCL on this driver
    dist.init_process_group(backend=backend, **kwargs)


def _init_dist_slurm(backend, port=29500, **kwargs):
    proc_id = int(os.environ["SLURM_PROCID"])
    ntasks = int(os.environ["SLURM_NTASKS"])
    node_list = os.environ["SLURM_NODELIST"]
    _bind_device(proc_id)
    addr = subprocess.getoutput("scontrol show hostname {} | head -n1".format(node_list))
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MASTER_ADDR"] = addr
    os.environ["WORLD_SIZE"] = str(ntasks)
    os.environ["RANK"] = str(proc_id)
    dist.init_process_group(backend=backend)


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_root_logger(log_level=logging.INFO):
    logger = logging.getLogger()
    if not logger.hasHandlers():
        logging.basicConfig(format="%(asctime)s - %(levelname)s - %(message)s", level=log_level)
    rank, _ = get_dist_info()
    if rank != 0:
        logger.setLevel("ERROR")
    return logger
