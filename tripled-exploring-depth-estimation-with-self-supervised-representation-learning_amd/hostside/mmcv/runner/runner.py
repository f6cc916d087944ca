"""Epoch-based training loop with hooks (mmcv 0.4.4 ``Runner`` semantics)."""
import logging
import os
import os.path as osp
import time

import torch

from . import hooks as H
from .checkpoint import load_checkpoint, save_checkpoint
from .log_buffer import LogBuffer
from .utils import get_dist_info, get_host_info, get_time_str, obj_from_dict


class Runner:
    def __init__(self, model, batch_processor, optimizer=None, work_dir=None, log_level=logging.INFO, logger=None):
        assert callable(batch_processor)
        self.model = model
        self.optimizer = self.init_optimizer(optimizer) if optimizer is not None else None
        self.batch_processor = batch_processor
        if isinstance(work_dir, str):
            self.work_dir = osp.abspath(work_dir)
            os.makedirs(self.work_dir, exist_ok=True)
        elif work_dir is None:
            self.work_dir = None
        else:
            raise TypeError('"work_dir" must be a str or None')
        inner = model.module if hasattr(model, "module") else model
        self._model_name = inner.__class__.__name__
        self._rank, self._world_size = get_dist_info()
        self.timestamp = get_time_str()
        self.logger = self.init_logger(work_dir, log_level) if logger is None else logger
        self.log_buffer = LogBuffer()
        self.mode = None
        self._hooks = []
        self._epoch = self._iter = self._inner_iter = 0
        self._max_epochs = self._max_iters = 0
        self.data_loader = None
        self.outputs = None

    model_name = property(lambda self: self._model_name)
    rank = property(lambda self: self._rank)
    world_size = property(lambda self: self._world_size)
    hooks = property(lambda self: self._hooks)
    epoch = property(lambda self: self._epoch)
    iter = property(lambda self: self._iter)
    inner_iter = property(lambda self: self._inner_iter)
    max_epochs = property(lambda self: self._max_epochs)
    max_iters = property(lambda self: self._max_iters)

    def init_optimizer(self, optimizer):
        if isinstance(optimizer, dict):
            return obj_from_dict(optimizer, torch.optim, dict(params=self.model.parameters()))
        if not isinstance(optimizer, torch.optim.Optimizer):
            raise TypeError("optimizer must be either an Optimizer object or a dict")
        return optimizer

    def init_logger(self, log_dir=None, level=logging.INFO):
        if isinstance(level, str):
            level = getattr(logging, level.upper())
        logging.basicConfig(format="%(asctime)s - %(levelname)s - %(message)s", level=level)
        logger = logging.getLogger(__name__)
        if log_dir and self.rank == 0:
            fh = logging.FileHandler(osp.join(log_dir, "{}.log".format(self.timestamp)), "w")
            fh.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
            fh.setLevel(level)
            logger.addHandler(fh)
        return logger

    def current_lr(self):
        if self.optimizer is None:
            raise RuntimeError("lr is not applicable because optimizer does not exist.")
        return [H.lr_value(group["lr"]) for group in self.optimizer.param_groups]

    def register_hook(self, hook, priority="NORMAL"):
        assert isinstance(hook, H.Hook)
        if hasattr(hook, "priority"):
            raise ValueError('"priority" is a reserved attribute for hooks')
        hook.priority = H.get_priority(priority)
        pos = len(self._hooks)
        while pos > 0 and self._hooks[pos - 1].priority > hook.priority:
            pos -= 1
        self._hooks.insert(pos, hook)

    def call_hook(self, fn_name):
        for hook in self._hooks:
            getattr(hook, fn_name)(self)

    def load_checkpoint(self, filename, map_location="cpu", strict=False):
        self.logger.info("load checkpoint from %s", filename)
        return load_checkpoint(self.model, filename, map_location, strict, self.logger)

    def save_checkpoint(self, out_dir, filename_tmpl="epoch_{}.pth", save_optimizer=True, meta=None):
        meta = dict(epoch=self.epoch + 1, iter=self.iter) if meta is None else dict(meta, epoch=self.epoch + 1, iter=self.iter)
        filename = filename_tmpl.format(self.epoch + 1)
        filepath = osp.join(out_dir, filename)
        save_checkpoint(self.model, filepath, optimizer=self.optimizer if save_optimizer else None, meta=meta)
        link = osp.join(out_dir, "latest.pth")
        try:
            if osp.lexists(link):
                os.remove(link)
            os.symlink(filename, link)
        except OSError:
            pass

    def train(self, data_loader, **kwargs):
        self.model.train()
        self.mode = "train"
        self.data_loader = data_loader
        self._max_iters = self._max_epochs * len(data_loader)
        self.call_hook("before_train_epoch")
        for i, data_batch in enumerate(data_loader):
            self._inner_iter = i
            self.call_hook("before_train_iter")
            outputs = self.batch_processor(self.model, data_batch, train_mode=True, **kwargs)
            if not isinstance(outputs, dict):
                raise TypeError("batch_processor() must return a dict")
            if "log_vars" in outputs:
                self.log_buffer.update(outputs["log_vars"], outputs["num_samples"])
            self.outputs = outputs
            self.call_hook("after_train_iter")
            self._iter += 1
        self.call_hook("after_train_epoch")
        self._epoch += 1

    def val(self, data_loader, **kwargs):
        self.model.eval()
        self.mode = "val"
        self.data_loader = data_loader
        self.call_hook("before_val_epoch")
        for i, data_batch in enumerate(data_loader):
            self._inner_iter = i
            self.call_hook("before_val_iter")
            with torch.no_grad():
                outputs = self.batch_processor(self.model, data_batch, train_mode=False, **kwargs)
            if "log_vars" in outputs:
                self.log_buffer.update(outputs["log_vars"], outputs["num_samples"])
            self.outputs = outputs
            self.call_hook("after_val_iter")
        self.call_hook("after_val_epoch")

    def resume(self, checkpoint, resume_optimizer=True, map_location="default"):
        if map_location == "default":
            if torch.cuda.is_available():
                dev = torch.cuda.current_device()
                ckpt = self.load_checkpoint(checkpoint, map_location=lambda storage, loc: storage.cuda(dev))
            else:
                ckpt = self.load_checkpoint(checkpoint, map_location="cpu")
        else:
            ckpt = self.load_checkpoint(checkpoint, map_location=map_location)
        self._epoch = ckpt["meta"]["epoch"]
        self._iter = ckpt["meta"]["iter"]
        if "optimizer" in ckpt and resume_optimizer and self.optimizer is not None:
            from .checkpoint import _unwrap, flat_store_of
            flat = flat_store_of(self.model)
            if flat is not None:
                flat.load_optimizer_state_dict(_unwrap(self.model), ckpt["optimizer"])
            else:
                # a device-side lr tensor (graph replay reads it) keeps its identity; the file's value is written into it
                held = [g["lr"] if torch.is_tensor(g["lr"]) else None for g in self.optimizer.param_groups]
                # the execution flags belong to THIS process, not to the file: a reference / older checkpoint carries no
                # capturable / fused / foreach keys, and torch's __setstate__ would default them to False / None -- the graph
                # capture would then be refused and the eager step would take the single-tensor path with a host sync per step
                flags = [{k: g[k] for k in ("capturable", "fused", "foreach", "differentiable") if k in g}
                         for g in self.optimizer.param_groups]
                self.optimizer.load_state_dict(ckpt["optimizer"])
                for group, t, fl in zip(self.optimizer.param_groups, held, flags):
                    group.update(fl)
                    if t is not None:
                        t.fill_(float(group["lr"]))
                        t._host_value = float(group["lr"])
                        group["lr"] = t
                if any(fl.get("capturable") or fl.get("fused") for fl in flags):     # torch keeps the step count on the device then
                    dev_of = {id(p): p.device for g in self.optimizer.param_groups for p in g["params"]}
                    for p, st in self.optimizer.state.items():
                        if "step" in st and torch.is_tensor(st["step"]) and st["step"].device != dev_of.get(id(p), st["step"].device):
                            st["step"] = st["step"].to(dev_of[id(p)], torch.float32)
        self.logger.info("resumed epoch %d, iter %d", self.epoch, self.iter)

    def run(self, data_loaders, workflow, max_epochs, **kwargs):
        assert isinstance(data_loaders, list) and len(data_loaders) == len(workflow)
        self._max_epochs = max_epochs
        self.logger.info("Start running, host: %s, work_dir: %s", get_host_info(), self.work_dir)
        self.logger.info("workflow: %s, max: %d epochs", workflow, max_epochs)
        self.call_hook("before_run")
        while self.epoch < max_epochs:
            for i, (mode, epochs) in enumerate(workflow):
                if not isinstance(mode, str) or not hasattr(self, mode):
                    raise ValueError('runner has no method named "{}" to run an epoch'.format(mode))
                epoch_runner = getattr(self, mode)
                for _ in range(epochs):
                    if mode == "train" and self.epoch >= max_epochs:
                        return
                    epoch_runner(data_loaders[i], **kwargs)
        time.sleep(0.01)
        self.call_hook("after_run")

    def register_lr_hooks(self, lr_config):
        if isinstance(lr_config, H.LrUpdaterHook):
            self.register_hook(lr_config, "VERY_HIGH")
        elif isinstance(lr_config, dict):
            assert "policy" in lr_config
            cfg = dict(lr_config)
            policy = cfg.pop("policy")
            name = policy.title() + "LrUpdaterHook"
            if not hasattr(H, name):
                raise ValueError('"{}" does not exist'.format(name))
            self.register_hook(getattr(H, name)(**cfg), "VERY_HIGH")
        else:
            raise TypeError('"lr_config" must be either a LrUpdaterHook object or dict')

    def register_logger_hooks(self, log_config):
        interval = log_config["interval"]
        for info in log_config["hooks"]:
            hook = obj_from_dict(dict(info), H, default_args=dict(interval=interval))
            self.register_hook(hook, priority="VERY_LOW")

    def register_training_hooks(self, lr_config, optimizer_config=None, checkpoint_config=None, log_config=None):
        optimizer_config = {} if optimizer_config is None else optimizer_config
        checkpoint_config = {} if checkpoint_config is None else checkpoint_config
        self.register_lr_hooks(lr_config)
        opt_hook = optimizer_config if isinstance(optimizer_config, H.Hook) else H.OptimizerHook(**optimizer_config)
        self.register_hook(opt_hook, "ABOVE_NORMAL")
        ckpt_hook = checkpoint_config if isinstance(checkpoint_config, H.Hook) else H.CheckpointHook(**checkpoint_config)
        self.register_hook(ckpt_hook, "NORMAL")
        self.register_hook(H.IterTimerHook(), "LOW")
        if log_config is not None:
            self.register_logger_hooks(log_config)
