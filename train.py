#!/usr/bin/env python3
"""Training entry point.  The command line is the reference's (reference: train.py:36-124), the
launch model is this build's: one process per MI355X, started by torch.distributed.run.

  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py \
      --config config/cfg_kitti_tripleD.py --work_dir work/ [--resume_from ckpt]

  python train.py --launcher none ...        # one process, one GPU

Gradients are exchanged by the flat parameter store / the bucketed RCCL engine (mmcv.parallel);
precision, memory format and dispatch strictness come from the config (mono.apis.trainer.configure_execution)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import tripled_amd  # noqa: F401,E402  (puts mono / mmcv on sys.path)

import torch  # noqa: E402
import mmcv  # noqa: E402
from mono import apis  # noqa: E402
from mono.datasets.get_dataset import get_dataset  # noqa: E402
from mono.model.registry import MONO  # noqa: E402
import mono.model  # noqa: F401,E402  (registers the model classes)

FLAGS = (
    # name(s), keyword arguments -- names and defaults as in the reference's CLI
    (("--config",), dict(default=os.path.join(ROOT, "config", "cfg_kitti_tripleD.py"), help="config file")),
    (("--work_dir",), dict(default=os.path.join(ROOT, "work_dirs", "tripled"), help="logs and checkpoints go here")),
    (("--resume_from",), dict(default=None, help="checkpoint to resume (weights, optimiser state, epoch/iter)")),
    (("--gpus",), dict(default="0", help="comma-separated device ids of a --launcher none run")),
    (("--seed",), dict(type=int, default=1024)),
    (("--launcher",), dict(choices=("none", "pytorch"), default="pytorch",
                           help="pytorch: ranks started by torch.distributed.run; none: a single process")),
    (("--local_rank", "--local-rank"), dict(type=int, default=0, help="accepted for old launchers; LOCAL_RANK is what is read")),
)


def read_job(argv=None):
    """Command line + config file -> the cfg object train_mono() consumes."""
    parser = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for names, kw in FLAGS:
        parser.add_argument(*names, **kw)
    args = parser.parse_args(argv)
    cfg = mmcv.Config.fromfile(args.config)
    cfg.work_dir = args.work_dir
    cfg.gpus = [int(tok) for tok in args.gpus.split(",") if tok != ""]
    if args.resume_from:
        cfg.resume_from = args.resume_from
    return args, cfg


def initial_weights(model, cfg):
    """resume_from wins over finetune: the runner restores weights, optimiser state and epoch/iter from it
    (Runner.resume -> mmcv.runner.load_checkpoint), once.  Both files are read with ``weights_only=True``:
    nothing in them is executed."""
    if cfg.resume_from is not None:
        return
    if cfg.get("finetune") is not None:
        blob = torch.load(cfg.finetune, map_location="cpu", weights_only=True)
        model.load_state_dict(blob.get("state_dict", blob), strict=False)


def main(argv=None):
    args, cfg = read_job(argv)
    torch.backends.cudnn.benchmark = bool(cfg.get("cudnn_benchmark", False))     # = MIOpen find mode on ROCm
    multi_rank = args.launcher != "none"
    if multi_rank:
        apis.init_dist(args.launcher, **cfg.dist_params)
    log = apis.get_root_logger(cfg.log_level)
    log.info("launcher=%s  seed=%s  config=%s", args.launcher, args.seed, args.config)
    if args.seed is not None:
        apis.set_random_seed(args.seed)

    model = MONO.module_dict[cfg.model["name"]](cfg.model)
    initial_weights(model, cfg)
    splits = {"train": get_dataset(cfg.data, training=True),
              "val": get_dataset(cfg.data, training=False) if cfg.validate else None}

    out_dir = os.path.abspath(cfg.work_dir)
    mmcv.mkdir_or_exist(out_dir)
    cfg.dump(os.path.join(out_dir, os.path.basename(args.config)))      # the run keeps a copy of its config
    apis.train_mono(model, splits["train"], splits["val"], cfg, distributed=multi_rank, validate=cfg.validate)


if __name__ == "__main__":
    main()
