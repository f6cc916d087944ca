"""Checkpoint format of mmcv 0.4.4: {'meta': {...}, 'state_dict': CPU tensors without the
'module.' prefix, 'optimizer': optimizer.state_dict()} saved as work_dir/epoch_N.pth."""
import os
import time
from collections import OrderedDict

import torch


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def flat_store_of(model):
    """The flat mixed-precision parameter store attached to a model by mono.apis.trainer (None without one): its
    parameters are bf16 working copies, the checkpoint holds the fp32 masters (tripled_amd.flat_amp)."""
    return getattr(_unwrap(model), "_flat_store", None)


def load_state_dict(module, state_dict, strict=False, logger=None):
    flat = flat_store_of(module)
    if flat is not None:
        missing, unexpected = flat.load_module_state_dict(module, state_dict, strict=strict)
    else:
        result = module.load_state_dict(state_dict, strict=strict)
        missing, unexpected = result.missing_keys, result.unexpected_keys
    problems = []
    if unexpected:
        problems.append("unexpected key in source state_dict: {}".format(", ".join(unexpected)))
    if missing:
        problems.append("missing keys in source state_dict: {}".format(", ".join(missing)))
    if problems:
        msg = "The model and loaded state dict do not match exactly\n" + "\n".join(problems)
        if logger is not None:
            logger.warning(msg)
        else:
            print(msg)


def load_checkpoint(model, filename, map_location=None, strict=False, logger=None):
    if not os.path.isfile(filename):
        raise IOError("{} is not a checkpoint file".format(filename))
    # weights_only: nothing in the file is executed.  A checkpoint in the mmcv 0.4.4 layout (meta of str/int, state_dict,
    # optimizer.state_dict() of tensors and scalars) loads under it; a pickle that carries code is refused.
    checkpoint = torch.load(filename, map_location=map_location, weights_only=True)
    if isinstance(checkpoint, OrderedDict):
        state_dict = checkpoint
    elif isinstance(checkpoint, dict) and "state_dict" in checkpoint:
        state_dict = checkpoint["state_dict"]
    else:
        raise RuntimeError("No state_dict found in checkpoint file {}".format(filename))
    if list(state_dict.keys()) and list(state_dict.keys())[0].startswith("module."):
        state_dict = {k[7:]: v for k, v in state_dict.items()}
    load_state_dict(_unwrap(model), state_dict, strict, logger)
    return checkpoint


def weights_to_cpu(state_dict):
    out = OrderedDict()
    for key, val in state_dict.items():
        out[key] = val.cpu()
    return out


def _plain_numbers(osd):
    """optimizer.state_dict() with 0-d tensor hyper-parameters (the device-side lr of a graph-replayed run) as floats:
    the file keeps the reference's layout whatever the execution mode of the run that wrote it."""
    from .hooks import lr_value
    groups = [{k: (lr_value(v) if torch.is_tensor(v) and v.dim() == 0 else v) for k, v in g.items()} for g in osd["param_groups"]]
    return {"state": osd["state"], "param_groups": groups}


def save_checkpoint(model, filename, optimizer=None, meta=None):
    meta = {} if meta is None else dict(meta)
    from .. import __version__
    meta.update(mmcv_version=__version__, time=time.asctime())
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    module, flat = _unwrap(model), flat_store_of(model)
    weights = flat.module_state_dict(module) if flat is not None else module.state_dict()
    checkpoint = {"meta": meta, "state_dict": weights_to_cpu(weights)}
    if optimizer is not None:
        # with the flat store the optimiser runs on ONE flat parameter; the file keeps the per-parameter layout
        checkpoint["optimizer"] = flat.optimizer_state_dict(module) if flat is not None else _plain_numbers(optimizer.state_dict())
    torch.save(checkpoint, filename)
