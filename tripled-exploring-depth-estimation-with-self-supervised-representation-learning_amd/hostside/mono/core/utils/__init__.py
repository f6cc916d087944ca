from .dist_utils import DistOptimizerHook, allreduce_grads  # noqa: F401
