"""Minimal stand-in for the slice of mmcv==0.4.4 that the reference's training path imports
(requirements.txt:9; import sites: train.py:7-8,15, mono/apis/trainer.py:11-12, mono/apis/env.py:13,
mono/core/utils/dist_utils.py:9, mono/datasets/loader/build_loader.py).  mmcv is not installable
here (no network) and 0.4.4 predates the torch this stack runs on, so the behaviour is restated
from that release's documented semantics -- parity unpinned, the reference
holds no tests at this boundary.  What IS native here: ``mmcv.parallel.MMDistributedDataParallel``
is this build's own bucketed RCCL gradient all-reduce engine."""
import json
import os

from .config import Config, ConfigDict  # noqa: F401

__version__ = "0.4.4+tripled.shim"


def mkdir_or_exist(dir_name, mode=0o777):
    if dir_name == "":
        return
    os.makedirs(os.path.expanduser(dir_name), mode=mode, exist_ok=True)


def dump(obj, file=None, file_format=None, **kwargs):
    """json / yaml dump (the formats train.py's Config.dump patch may ask for)."""
    if file_format is None and isinstance(file, str):
        file_format = file.rsplit(".", 1)[-1]
    if file_format in ("yml", "yaml"):
        import yaml
        text = yaml.safe_dump(obj, **kwargs)
    elif file_format == "json":
        text = json.dumps(obj, **kwargs)
    else:
        raise TypeError("Unsupported format: {}".format(file_format))
    if file is None:
        return text
    with open(file, "w") as f:
        f.write(text)
