"""Reference path mono/model/mono_fm_joint/pose_decoder.py -> mono.model.networks."""
from ..networks import PoseDecoder  # noqa: F401
