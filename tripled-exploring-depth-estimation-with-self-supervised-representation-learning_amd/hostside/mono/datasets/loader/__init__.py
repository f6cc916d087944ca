from .build_loader import build_dataloader  # noqa: F401
from .sampler import DistributedGroupSampler, DistributedSampler, GroupSampler  # noqa: F401
