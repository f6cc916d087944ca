from .pixel_error import AverageMeter, compute_errors, disp_to_depth  # noqa: F401
from .eval_hooks import (DistEvalHook, DistEvalMonoHook, NonDistEvalHook, evaluate_disparity,  # noqa: F401
                         resize_bilinear)
