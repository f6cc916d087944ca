from .loader import DistributedGroupSampler, DistributedSampler, GroupSampler, build_dataloader  # noqa: F401
from .get_dataset import get_dataset  # noqa: F401
from .synthetic import ResidentBatches, SyntheticTripletDataset, synthetic_batch  # noqa: F401
from .prefetch import DevicePrefetcher  # noqa: F401
from .device_expand import expand_device_batch, has_uint8_frames  # noqa: F401
