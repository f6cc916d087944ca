"""Reference path mono/model/mono_fm_joint/depth_encoder.py -> mono.model.networks."""
from ..networks import DepthEncoder  # noqa: F401
