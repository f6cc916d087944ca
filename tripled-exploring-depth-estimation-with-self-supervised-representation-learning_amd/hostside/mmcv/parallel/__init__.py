from .collate import collate, scatter, scatter_kwargs  # noqa: F401
from .data_parallel import MMDataParallel  # noqa: F401
from .distributed import MMDistributedDataParallel, GradientBuckets  # noqa: F401
