"""Hook set of mmcv 0.4.4's Runner that the training configs exercise."""
import datetime
import json
import os
import time

import torch
from torch.nn.utils import clip_grad

PRIORITIES = {"HIGHEST": 0, "VERY_HIGH": 10, "HIGH": 30, "ABOVE_NORMAL": 40, "NORMAL": 50,
              "BELOW_NORMAL": 60, "LOW": 70, "VERY_LOW": 90, "LOWEST": 100}


def get_priority(priority):
    if isinstance(priority, int):
        if not 0 <= priority <= 100:
            raise ValueError("priority must be between 0 and 100")
        return priority
    if isinstance(priority, str):
        return PRIORITIES[priority.upper()]
    raise TypeError("priority must be an integer or a priority name")


def lr_value(lr):
    """A learning rate as a plain number.  A device-side lr tensor (graph replay reads it) carries the exact value the host
    wrote last in ``_host_value``: no device sync, and no fp32 rounding of the schedule's base value."""
    host = getattr(lr, "_host_value", None)
    return host if host is not None else float(lr)


class Hook:
    def before_run(self, runner): pass
    def after_run(self, runner): pass
    def before_epoch(self, runner): pass
    def after_epoch(self, runner): pass
    def before_iter(self, runner): pass
    def after_iter(self, runner): pass
    def before_train_epoch(self, runner): self.before_epoch(runner)
    def before_val_epoch(self, runner): self.before_epoch(runner)
    def after_train_epoch(self, runner): self.after_epoch(runner)
    def after_val_epoch(self, runner): self.after_epoch(runner)
    def before_train_iter(self, runner): self.before_iter(runner)
    def before_val_iter(self, runner): self.before_iter(runner)
    def after_train_iter(self, runner): self.after_iter(runner)
    def after_val_iter(self, runner): self.after_iter(runner)

    def every_n_epochs(self, runner, n):
        return (runner.epoch + 1) % n == 0 if n > 0 else False

    def every_n_inner_iters(self, runner, n):
        return (runner.inner_iter + 1) % n == 0 if n > 0 else False

    def every_n_iters(self, runner, n):
        return (runner.iter + 1) % n == 0 if n > 0 else False

    def end_of_epoch(self, runner):
        return runner.inner_iter + 1 == len(runner.data_loader)


class OptimizerHook(Hook):
    def __init__(self, grad_clip=None):
        self.grad_clip = grad_clip

    def clip_grads(self, params):
        clip_grad.clip_grad_norm_(filter(lambda p: p.requires_grad and p.grad is not None, params), **self.grad_clip)

    def after_train_iter(self, runner):
        runner.optimizer.zero_grad()
        runner.outputs["loss"].backward()
        if self.grad_clip is not None:
            self.clip_grads(runner.model.parameters())
        runner.optimizer.step()


class LrUpdaterHook(Hook):
    def __init__(self, by_epoch=True, warmup=None, warmup_iters=0, warmup_ratio=0.1, **kwargs):
        if warmup is not None and warmup not in ("constant", "linear", "exp"):
            raise ValueError('"{}" is not a supported type for warming up'.format(warmup))
        if warmup is not None:
            assert warmup_iters > 0 and 0 < warmup_ratio <= 1.0
        self.by_epoch, self.warmup = by_epoch, warmup
        self.warmup_iters, self.warmup_ratio = warmup_iters, warmup_ratio
        self.base_lr, self.regular_lr = [], []

    def _set_lr(self, runner, lr_groups):
        for group, lr in zip(runner.optimizer.param_groups, lr_groups):
            cur = group["lr"]
            if torch.is_tensor(cur):
                # a device-side learning rate (the iteration is replayed from a HIP graph that reads this tensor):
                # written in place, and only when the schedule moves it; the host keeps the value it wrote last
                if getattr(cur, "_host_value", None) != lr:
                    cur.fill_(lr)
                    cur._host_value = lr
            else:
                group["lr"] = lr

    def get_lr(self, runner, base_lr):
        raise NotImplementedError

    def get_regular_lr(self, runner):
        return [self.get_lr(runner, lr) for lr in self.base_lr]

    def get_warmup_lr(self, cur_iters):
        if self.warmup == "constant":
            return [lr * self.warmup_ratio for lr in self.regular_lr]
        if self.warmup == "linear":
            k = (1 - cur_iters / self.warmup_iters) * (1 - self.warmup_ratio)
            return [lr * (1 - k) for lr in self.regular_lr]
        k = self.warmup_ratio ** (1 - cur_iters / self.warmup_iters)
        return [lr * k for lr in self.regular_lr]

    def before_run(self, runner):
        for group in runner.optimizer.param_groups:
            if "initial_lr" not in group:
                group["initial_lr"] = lr_value(group["lr"])      # (a plain number also when lr is a device tensor)
        self.base_lr = [group["initial_lr"] for group in runner.optimizer.param_groups]

    def before_train_epoch(self, runner):
        if not self.by_epoch:
            return
        self.regular_lr = self.get_regular_lr(runner)
        self._set_lr(runner, self.regular_lr)

    def before_train_iter(self, runner):
        cur_iter = runner.iter
        if not self.by_epoch:
            self.regular_lr = self.get_regular_lr(runner)
            if self.warmup is None or cur_iter >= self.warmup_iters:
                self._set_lr(runner, self.regular_lr)
            else:
                self._set_lr(runner, self.get_warmup_lr(cur_iter))
        elif self.warmup is not None:
            if cur_iter > self.warmup_iters:
                return
            if cur_iter == self.warmup_iters:
                self._set_lr(runner, self.regular_lr)
            else:
                self._set_lr(runner, self.get_warmup_lr(cur_iter))


class FixedLrUpdaterHook(LrUpdaterHook):
    def get_lr(self, runner, base_lr):
        return base_lr


class StepLrUpdaterHook(LrUpdaterHook):
    def __init__(self, step, gamma=0.1, **kwargs):
        assert isinstance(step, (list, int))
        if isinstance(step, list):
            assert all(s > 0 for s in step)
        else:
            assert step > 0
        self.step, self.gamma = step, gamma
        super().__init__(**kwargs)

    def get_lr(self, runner, base_lr):
        progress = runner.epoch if self.by_epoch else runner.iter
        if isinstance(self.step, int):
            return base_lr * (self.gamma ** (progress // self.step))
        exp = len(self.step)
        for i, s in enumerate(self.step):
            if progress < s:
                exp = i
                break
        return base_lr * self.gamma ** exp


class CheckpointHook(Hook):
    def __init__(self, interval=-1, save_optimizer=True, out_dir=None, **kwargs):
        self.interval, self.save_optimizer, self.out_dir, self.args = interval, save_optimizer, out_dir, kwargs

    def after_train_epoch(self, runner):
        if runner.rank != 0 or not self.every_n_epochs(runner, self.interval):
            return
        runner.save_checkpoint(self.out_dir or runner.work_dir, save_optimizer=self.save_optimizer, **self.args)


class IterTimerHook(Hook):
    def before_epoch(self, runner):
        self.t = time.time()

    def before_iter(self, runner):
        runner.log_buffer.update({"data_time": time.time() - self.t})

    def after_iter(self, runner):
        runner.log_buffer.update({"time": time.time() - self.t})
        self.t = time.time()


class DistSamplerSeedHook(Hook):
    def before_epoch(self, runner):
        sampler = getattr(runner.data_loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(runner.epoch)


class LoggerHook(Hook):
    def __init__(self, interval=10, ignore_last=True, reset_flag=False):
        self.interval, self.ignore_last, self.reset_flag = interval, ignore_last, reset_flag

    def log(self, runner):
        raise NotImplementedError

    def before_run(self, runner):
        for hook in runner.hooks[::-1]:
            if isinstance(hook, LoggerHook):
                hook.reset_flag = True
                break

    def before_epoch(self, runner):
        runner.log_buffer.clear()

    def after_train_iter(self, runner):
        if self.every_n_inner_iters(runner, self.interval):
            runner.log_buffer.average(self.interval)
        elif self.end_of_epoch(runner) and not self.ignore_last:
            runner.log_buffer.average(self.interval)
        if runner.log_buffer.ready:
            self.log(runner)
            if self.reset_flag:
                runner.log_buffer.clear_output()

    def after_train_epoch(self, runner):
        if runner.log_buffer.ready:
            self.log(runner)
            if self.reset_flag:
                runner.log_buffer.clear_output()

    def after_val_epoch(self, runner):
        runner.log_buffer.average()
        self.log(runner)
        if self.reset_flag:
            runner.log_buffer.clear_output()


class TextLoggerHook(LoggerHook):
    def __init__(self, interval=10, ignore_last=True, reset_flag=False):
        super().__init__(interval, ignore_last, reset_flag)
        self.time_sec_tot = 0

    def before_run(self, runner):
        super().before_run(runner)
        self.start_iter = runner.iter
        self.json_log_path = os.path.join(runner.work_dir, "{}.log.json".format(runner.timestamp))

    def _memory_mb(self, runner):
        if not torch.cuda.is_available():
            return 0
        return int(torch.cuda.max_memory_allocated() / (1024 * 1024))

    def log(self, runner):
        out = runner.log_buffer.output
        mode = runner.mode
        record = {"mode": mode, "epoch": runner.epoch + 1, "iter": runner.inner_iter + 1,
                  "lr": runner.current_lr()[0] if runner.optimizer is not None else None}
        if mode == "train":
            text = "Epoch [{}][{}/{}]\tlr: {:.5f}, ".format(record["epoch"], record["iter"], len(runner.data_loader),
                                                           record["lr"])
            if "time" in out:
                self.time_sec_tot += out["time"] * self.interval
                avg = self.time_sec_tot / max(1, runner.iter - self.start_iter + 1)
                eta = str(datetime.timedelta(seconds=int(avg * (runner.max_iters - runner.iter - 1))))
                text += "eta: {}, time: {:.3f}, data_time: {:.3f}, ".format(eta, out["time"], out.get("data_time", 0.0))
                record["memory"] = self._memory_mb(runner)
                text += "memory: {}, ".format(record["memory"])
        else:
            text = "Epoch({}) [{}][{}]\t".format(mode, record["epoch"] - 1, record["iter"])
        items = []
        for name, val in out.items():
            if name in ("time", "data_time"):
                record[name] = val
                continue
            record[name] = val
            items.append("{}: {:.4f}".format(name, val) if isinstance(val, float) else "{}: {}".format(name, val))
        runner.logger.info(text + ", ".join(items))
        if runner.rank == 0:
            with open(self.json_log_path, "a+") as f:
                f.write(json.dumps(record) + "\n")


class TensorboardLoggerHook(LoggerHook):
    """Writes scalars to work_dir/tf_logs when a SummaryWriter implementation is importable;
    otherwise stays silent (tensorboard / tensorboardX are optional dependencies)."""

    def __init__(self, log_dir=None, interval=10, ignore_last=True, reset_flag=True):
        super().__init__(interval, ignore_last, reset_flag)
        self.log_dir, self.writer = log_dir, None

    def before_run(self, runner):
        super().before_run(runner)
        if runner.rank != 0:
            return
        try:
            from torch.utils.tensorboard import SummaryWriter
        except Exception:
            try:
                from tensorboardX import SummaryWriter
            except Exception:
                runner.logger.warning("TensorboardLoggerHook: no SummaryWriter available, hook disabled")
                return
        self.writer = SummaryWriter(self.log_dir or os.path.join(runner.work_dir, "tf_logs"))

    def log(self, runner):
        if self.writer is None:
            return
        for name, val in runner.log_buffer.output.items():
            if name in ("time", "data_time") or isinstance(val, str):
                continue
            self.writer.add_scalar("{}/{}".format(name, runner.mode), val, runner.iter)

    def after_run(self, runner):
        if self.writer is not None:
            self.writer.close()
