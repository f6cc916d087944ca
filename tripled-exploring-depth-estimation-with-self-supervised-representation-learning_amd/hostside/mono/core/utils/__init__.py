from .dist_utils import DistOptimizerHook, FlatOptimizerHook, allreduce_grads  # noqa: F401
