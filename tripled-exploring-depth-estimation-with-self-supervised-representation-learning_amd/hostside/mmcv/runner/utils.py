import sys
import time
from getpass import getuser
from socket import gethostname

import torch.distributed as dist


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def get_host_info():
    try:
        return "{}@{}".format(getuser(), gethostname())
    except Exception:
        return "unknown@host"


def get_time_str():
    return time.strftime("%Y%m%d_%H%M%S", time.localtime())


def obj_from_dict(info, parent=None, default_args=None):
    """Build an object from ``dict(type=..., **kwargs)``; ``type`` is a class or a name looked up
    on ``parent`` (e.g. torch.optim)."""
    assert isinstance(info, dict) and "type" in info
    assert isinstance(default_args, dict) or default_args is None
    args = dict(info)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_type = getattr(parent, obj_type) if parent is not None else sys.modules[obj_type]
    elif not isinstance(obj_type, type):
        raise TypeError("type must be a str or valid type, but got {}".format(type(obj_type)))
    if default_args is not None:
        for name, value in default_args.items():
            args.setdefault(name, value)
    return obj_type(**args)
