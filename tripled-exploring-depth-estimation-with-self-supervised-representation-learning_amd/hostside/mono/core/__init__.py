from .utils import DistOptimizerHook, FlatOptimizerHook, IterationDoneHook, allreduce_grads  # noqa: F401
from .evaluation import (AverageMeter, DistEvalHook, DistEvalMonoHook, NonDistEvalHook, evaluate_disparity,  # noqa: F401
                         compute_errors, disp_to_depth)
