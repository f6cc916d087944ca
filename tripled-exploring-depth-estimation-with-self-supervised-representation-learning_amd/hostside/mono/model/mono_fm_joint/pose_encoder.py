"""Reference path mono/model/mono_fm_joint/pose_encoder.py -> mono.model.networks."""
from ..networks import PoseEncoder  # noqa: F401
