"""Training API (reference: mono/apis/trainer.py): train_mono, batch_processor, build_optimizer."""
import re
from collections import OrderedDict

import torch
from mmcv.parallel import MMDataParallel, MMDistributedDataParallel
from mmcv.runner import DistSamplerSeedHook, Runner, obj_from_dict

from mono.core import DistEvalMonoHook, DistOptimizerHook, NonDistEvalHook
from mono.datasets import build_dataloader


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


# Execution mode of the networks, set by train_mono from the config (keys this build adds; absent keys take the
# MI355X defaults, so the reference's configs run on the fast path unchanged):
#   amp            "bf16" (default on a HIP device) | "fp32"/None -- autocast dtype of the convolutional networks;
#                  the loss hot path always runs in the fp32 HIP kernels
#   channels_last  True (default on a HIP device) -- NHWC weights/activations: the layout the hand-written
#                  BatchNorm / pool / pad / join kernels and MIOpen's MFMA implicit-GEMM kernels take
#   strict_dispatch  False -- raise instead of warn when a HIP tensor falls back to an ATen composition
#   fp8_conv1x1    False -- forward GEMM of the 1x1 convolutions on the fp8 MFMA path (mono.model.networks.set_fp8_conv1x1)
#   flat_params    True with amp="bf16" on a HIP device, Adam without paramwise options -- the flat mixed-precision
#                  parameter store (tripled_amd.flat_amp): one cast instead of ~500 per-weight casts per step, one
#                  single-tensor Adam, gradient all-reduce on one flat buffer; checkpoints keep the reference's layout
#   hip_graph      True on a HIP device with Adam without paramwise options -- the Runner replays the whole iteration
#                  (forward, backward, gradient exchange, clip, Adam) from ONE HIP graph after ``graph_warmup_iters`` (3)
#                  eager iterations (tripled_amd.step.RunnerIteration); the LR schedule, logging, checkpoints and the
#                  evaluation hooks stay outside the graph
_MODE = {"autocast": None}


def configure_execution(model, cfg, dev):
    """Apply cfg.amp / cfg.channels_last / cfg.strict_dispatch to ``model`` (before it is wrapped and before the
    optimizer is built) and to batch_processor's autocast.  Returns the model."""
    on_gpu = dev.type == "cuda"
    amp = cfg.get("amp", "bf16" if on_gpu else None)
    if amp in (None, False, "fp32", "none"):
        _MODE["autocast"] = None
    elif amp == "bf16":
        _MODE["autocast"] = torch.bfloat16
    else:
        raise ValueError("cfg.amp must be 'bf16' or 'fp32', got %r" % (amp,))
    model = model.to(dev)
    if cfg.get("channels_last", on_gpu):
        model = model.to(memory_format=torch.channels_last)
    if on_gpu:
        from tripled_amd import dispatch
        dispatch.set_strict(bool(cfg.get("strict_dispatch", False)))
        from mono.model.networks import set_fp8_conv1x1
        set_fp8_conv1x1(bool(cfg.get("fp8_conv1x1", False)))
    return model


def stage_inputs(data):
    """Every entry of the batch dict -> float32 on the training device (frames of the 'uint8' wire format stay bytes).
    The copies are issued non-blocking (asynchronous when the loader pins memory; no-ops behind DevicePrefetcher)."""
    dev = _device()
    for k, v in data.items():
        if isinstance(k, tuple) and k and k[0] == "color_u8":       # 'uint8' wire format: bytes until expanded on the device
            data[k] = torch.as_tensor(v).to(dev, non_blocking=True)
        elif "kp" not in k:
            data[k] = torch.as_tensor(v).to(dev, dtype=torch.float32, non_blocking=True)
    return data


def change_input_variable(data):
    """reference :19-29: every entry of the batch dict -> float32 on the training device, then the device-side
    expansion of the uint8 wire format."""
    if isinstance(data, dict):
        stage_inputs(data)
        from mono.datasets import expand_device_batch
        expand_device_batch(data)
    else:
        dev = _device()
        data[0] = [torch.as_tensor(img).to(dev, dtype=torch.float32, non_blocking=True) for img in data[0]]
    return data


def batch_processor(model, data, train_mode):
    """reference :32-60.  loss = sum of the means of every loss_dict entry.  The per-key values
    are handed to the log buffer as device scalars (read back only when a log line is due)
    instead of ~30 blocking .item() calls per iteration."""
    if train_mode:
        model.train()
    data = change_input_variable(data)
    dtype = _MODE["autocast"]
    on_gpu = next(iter(data.values())).is_cuda if isinstance(data, dict) else False
    with torch.autocast("cuda" if on_gpu else "cpu", dtype=dtype, enabled=dtype is not None and on_gpu):
        model_out, losses = model(data)
    log_vars = OrderedDict()
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, list):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError("{} is not a tensor or list of tensors".format(name))
    loss = sum(v for v in log_vars.values())
    log_vars["loss"] = loss
    detached = OrderedDict((str(k), v.detach()) for k, v in log_vars.items())
    n = len(data[("color", 0, 0)]) if isinstance(data, dict) else len(data[0])
    return dict(loss=loss, log_vars=detached, num_samples=n)


def train_mono(model, dataset_train, dataset_val, cfg, distributed=False, validate=False):
    if distributed:
        _dist_train(model, dataset_train, dataset_val, cfg, validate=validate)
    else:
        _non_dist_train(model, dataset_train, dataset_val, cfg, validate=validate)


def build_optimizer(model, optimizer_cfg):
    """reference :77-144 (mmdet-style paramwise_options: bias_lr_mult, bias_decay_mult,
    norm_decay_mult)."""
    if hasattr(model, "module"):
        model = model.module
    optimizer_cfg = dict(optimizer_cfg)
    paramwise = optimizer_cfg.pop("paramwise_options", None)
    if paramwise is None:
        extra = dict(params=model.parameters())
        first = next(model.parameters(), None)
        if optimizer_cfg.get("type") in ("Adam", "AdamW") and first is not None and first.is_cuda:
            extra["fused"] = True        # one multi-tensor kernel per step instead of ~10 tiny kernels per parameter
        return obj_from_dict(optimizer_cfg, torch.optim, extra)
    assert isinstance(paramwise, dict)
    base_lr = optimizer_cfg["lr"]
    base_wd = optimizer_cfg.get("weight_decay", None)
    if "bias_decay_mult" in paramwise or "norm_decay_mult" in paramwise:
        assert base_wd is not None
    bias_lr_mult = paramwise.get("bias_lr_mult", 1.0)
    bias_decay_mult = paramwise.get("bias_decay_mult", 1.0)
    norm_decay_mult = paramwise.get("norm_decay_mult", 1.0)
    groups = []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        group = {"params": [param]}
        if re.search(r"(bn|gn)(\d+)?.(weight|bias)", name):
            if base_wd is not None:
                group["weight_decay"] = base_wd * norm_decay_mult
        elif name.endswith(".bias"):
            group["lr"] = base_lr * bias_lr_mult
            if base_wd is not None:
                group["weight_decay"] = base_wd * bias_decay_mult
        groups.append(group)
    return getattr(torch.optim, optimizer_cfg.pop("type"))(groups, **optimizer_cfg)


def _use_flat_store(cfg, dev):
    ocfg = cfg.optimizer
    # weight_decay: a parameter no gradient reaches has a zero slot in the flat gradient buffer, which Adam would still
    # decay; the per-parameter optimiser skips it (grad None), so configs with decay keep the per-parameter path
    default = (dev.type == "cuda" and _MODE["autocast"] is torch.bfloat16 and ocfg.get("type") == "Adam"
               and ocfg.get("paramwise_options") is None and not ocfg.get("weight_decay", 0))
    return bool(cfg.get("flat_params", default))


def _build_flat_store(model, cfg):
    """model's parameters move into the flat store; returns (store, optimiser hook).  The store hangs on the module as
    ``_flat_store`` (not a sub-module): the checkpoint shim and the evaluation hooks look for it there."""
    from tripled_amd.flat_amp import FlatMixedPrecision
    from mono.core import FlatOptimizerHook
    inner = model.module if hasattr(model, "module") else model
    ocfg = dict(cfg.optimizer)
    ocfg.pop("type")
    flat = FlatMixedPrecision(inner, lowp=True, **ocfg)
    inner._flat_store = flat
    return flat, FlatOptimizerHook(flat, **cfg.optimizer_config)


def _use_hip_graph(cfg, dev, model=None):
    """Graph replay needs an iteration without host-side state: the reference's CPU-generator auto-mask noise
    (automask_noise="cpu": drawn on the host and uploaded) or a recorded noise source (set_noise_source) would be
    recorded ONCE and replayed for every iteration, so such a run takes the eager iteration (and says so)."""
    ocfg = cfg.optimizer
    default = dev.type == "cuda" and ocfg.get("type") == "Adam" and ocfg.get("paramwise_options") is None
    want = bool(cfg.get("hip_graph", default))
    inner = getattr(model, "module", model)
    host_rng = cfg.model.get("automask_noise", "device") == "cpu" or getattr(inner, "_noise_fn", None) is not None
    if want and host_rng:
        import logging
        logging.getLogger(__name__).warning("hip_graph off: the model draws its auto-mask noise on the host (automask_noise='cpu' or a "
                                            "noise source): a captured iteration would replay one draw for ever")
        return False
    return want


def _graphed_iteration(model, cfg, dev, flat, logger=None):
    """(batch_processor, optimizer, optimiser hook) of a run whose iteration is replayed from a HIP graph."""
    from tripled_amd.step import RunnerIteration, TrainStep
    from mono.core import IterationDoneHook
    from mono.datasets import expand_device_batch
    step = TrainStep(model, cfg, None, _MODE["autocast"], flat=flat if flat is not None else False,
                     prepare=expand_device_batch, device=dev, lr_tensor=True, keep_outputs=False)
    iteration = RunnerIteration(step, stage_inputs, batch_processor, warmup_iters=cfg.get("graph_warmup_iters", 3),
                                logger=logger, syncbn=bool(cfg.get("syncbn", False)) and _world_size() > 1)
    return iteration, step.optimizer, IterationDoneHook(iteration, **cfg.optimizer_config)


def _world_size():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _loaders(dataset_train, cfg, dist):
    """The training loader.  A dataset object may bring its own loader (``as_loader``): bench.py's HBM-resident batches
    go through the Runner that way."""
    if hasattr(dataset_train, "as_loader"):
        return [dataset_train.as_loader(cfg.imgs_per_gpu)]
    if dist:
        return [build_dataloader(dataset_train, cfg.imgs_per_gpu, cfg.workers_per_gpu, dist=True)]
    return [build_dataloader(dataset_train, cfg.imgs_per_gpu, cfg.workers_per_gpu, len(cfg.gpus), dist=False)]


def _maybe_prefetch(loaders, cfg):
    """Overlap the host->device copy of batch t+1 with step t (cfg.device_prefetch, default on with a GPU)."""
    if torch.cuda.is_available() and cfg.get("device_prefetch", True):
        from mono.datasets import DevicePrefetcher
        return [dl if getattr(dl, "device_resident", False) else DevicePrefetcher(dl) for dl in loaders]
    return loaders


def _finish_runner(runner, cfg, data_loaders):
    data_loaders = _maybe_prefetch(data_loaders, cfg)
    if cfg.resume_from:
        runner.resume(cfg.resume_from)
    elif cfg.load_from:
        runner.load_checkpoint(cfg.load_from)
    runner.run(data_loaders, cfg.workflow, cfg.total_epochs)


def _dist_train(model, dataset_train, dataset_val, cfg, validate=False):
    data_loaders = _loaders(dataset_train, cfg, dist=True)
    dev = _device()
    if cfg.get("syncbn", False):
        # this build's BatchNorm layers exchange their statistics themselves (hand-written passes + one small
        # all-reduce per layer and direction); any other normalisation layer takes torch's SyncBatchNorm
        from mono.model.networks import enable_sync_batchnorm
        enable_sync_batchnorm(model)
    model = configure_execution(model, cfg, dev)
    use_flat, use_graph = _use_flat_store(cfg, dev), _use_hip_graph(cfg, dev, model)
    # under graph replay Python autograd hooks do not run: the bucket engine then exchanges after backward (overlap=False:
    # captured with the step on RCCL, eager between two graphs otherwise)
    model = MMDistributedDataParallel(model, find_unused_parameters=cfg.get("find_unused_parameters", False),
                                      device_ids=[dev.index] if dev.type == "cuda" else None,
                                      broadcast_buffers=False, gradient_engine=not use_flat, overlap=not use_graph)
    flat, processor = None, batch_processor
    if use_flat:      # the store all-reduces its own flat gradient buffer; the wrapper only broadcasts the initial state
        flat, opt_hook = _build_flat_store(model, cfg)
        optimizer = flat.optimizer
    elif not use_graph:
        optimizer, opt_hook = build_optimizer(model, cfg.optimizer), DistOptimizerHook(**cfg.optimizer_config)
    if use_graph:
        processor, optimizer, opt_hook = _graphed_iteration(model, cfg, dev, flat)
    runner = Runner(model, processor, optimizer, cfg.work_dir, cfg.log_level)
    runner.register_training_hooks(cfg.lr_config, opt_hook, cfg.checkpoint_config, cfg.log_config)
    runner.register_hook(DistSamplerSeedHook())
    if validate:
        if "num_classes" in cfg:
            raise NotImplementedError("segmentation evaluation is outside the depth training path")
        runner.register_hook(DistEvalMonoHook(dataset_val, cfg.get("validate_interval", 1), cfg))
    _finish_runner(runner, cfg, data_loaders)


def _non_dist_train(model, dataset_train, dataset_val, cfg, validate=False):
    data_loaders = _loaders(dataset_train, cfg, dist=False)
    dev = _device()
    model = MMDataParallel(configure_execution(model, cfg, dev), device_ids=list(range(len(cfg.gpus))))
    use_graph = _use_hip_graph(cfg, dev, model)
    flat, processor = None, batch_processor
    if _use_flat_store(cfg, dev):
        flat, opt_hook = _build_flat_store(model, cfg)
        optimizer = flat.optimizer
    elif not use_graph:
        optimizer, opt_hook = build_optimizer(model, cfg.optimizer), cfg.optimizer_config
    if use_graph:
        processor, optimizer, opt_hook = _graphed_iteration(model, cfg, dev, flat)
    runner = Runner(model, processor, optimizer, cfg.work_dir, cfg.log_level)
    runner.register_training_hooks(cfg.lr_config, opt_hook, cfg.checkpoint_config, cfg.log_config)
    if validate:
        if "num_classes" in cfg:
            raise NotImplementedError("segmentation evaluation is outside the depth training path")
        runner.register_hook(NonDistEvalHook(dataset_val, cfg))
    _finish_runner(runner, cfg, data_loaders)
