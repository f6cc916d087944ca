from .dist_utils import DistOptimizerHook, FlatOptimizerHook, IterationDoneHook, allreduce_grads  # noqa: F401
