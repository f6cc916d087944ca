#!/usr/bin/env python3
"""Training CLI with the reference's flags (reference: train.py:36-124):

  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py \
      --config config/cfg_kitti_tripleD.py --work_dir work/ [--resume_from ckpt] [--launcher pytorch]

One process per GPU; gradients are averaged by the bucketed RCCL engine in mmcv.parallel."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import tripled_amd  # noqa: F401,E402  (puts mono / mmcv on sys.path)

import torch  # noqa: E402
import mmcv  # noqa: E402
from mmcv import Config  # noqa: E402
from mmcv.runner import load_checkpoint  # noqa: E402
from mono.apis import get_root_logger, init_dist, set_random_seed, train_mono  # noqa: E402
from mono.datasets.get_dataset import get_dataset  # noqa: E402
from mono.model.registry import MONO  # noqa: E402
import mono.model  # noqa: F401,E402  (registers the model classes)


def parse_args():
    p = argparse.ArgumentParser(description="Train a self-supervised depth model")
    p.add_argument("--config", default=os.path.join(ROOT, "config", "cfg_kitti_tripleD.py"), help="train config file path")
    p.add_argument("--work_dir", default=os.path.join(ROOT, "work_dirs", "tripled"), help="the dir to save logs and models")
    p.add_argument("--resume_from", help="the checkpoint file to resume from")
    p.add_argument("--gpus", default="0", type=str, help="gpu ids (only applicable to non-distributed training)")
    p.add_argument("--seed", type=int, default=1024, help="random seed")
    p.add_argument("--launcher", choices=["none", "pytorch", "slurm", "mpi"], default="pytorch", help="job launcher")
    p.add_argument("--local_rank", "--local-rank", type=int, default=0)
    return p.parse_args()


def main():
    args = parse_args()
    cfg = Config.fromfile(args.config)
    cfg.work_dir = args.work_dir
    if cfg.get("cudnn_benchmark", False):
        torch.backends.cudnn.benchmark = True     # MIOpen find-mode on ROCm
    if args.resume_from is not None:
        cfg.resume_from = args.resume_from
    cfg.gpus = [int(g) for g in args.gpus.split(",")]
    distributed = args.launcher != "none"
    if distributed:
        init_dist(args.launcher, **cfg.dist_params)
    logger = get_root_logger(cfg.log_level)
    logger.info("Distributed training: {}".format(distributed))
    if args.seed is not None:
        logger.info("Set random seed to {}".format(args.seed))
        set_random_seed(args.seed)
    model = MONO.module_dict[cfg.model["name"]](cfg.model)
    if cfg.resume_from is not None:
        load_checkpoint(model, cfg.resume_from, map_location="cpu")
    elif cfg.finetune is not None:
        ckpt = torch.load(cfg.finetune, map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["state_dict"], strict=False)
    train_dataset = get_dataset(cfg.data, training=True)
    val_dataset = get_dataset(cfg.data, training=False) if cfg.validate else None
    mmcv.mkdir_or_exist(os.path.abspath(cfg.work_dir))
    cfg.dump(os.path.join(cfg.work_dir, os.path.basename(args.config)))
    train_mono(model, train_dataset, val_dataset, cfg, distributed=distributed, validate=cfg.validate)


if __name__ == "__main__":
    main()
