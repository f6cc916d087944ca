from .utils import get_dist_info, get_host_info, get_time_str, obj_from_dict  # noqa: F401
from .checkpoint import load_checkpoint, load_state_dict, save_checkpoint, weights_to_cpu  # noqa: F401
from .log_buffer import LogBuffer  # noqa: F401
from .hooks import (Hook, OptimizerHook, LrUpdaterHook, StepLrUpdaterHook, FixedLrUpdaterHook,  # noqa: F401
                    CheckpointHook, IterTimerHook, DistSamplerSeedHook, LoggerHook, TextLoggerHook,
                    TensorboardLoggerHook, get_priority)
from .runner import Runner  # noqa: F401
