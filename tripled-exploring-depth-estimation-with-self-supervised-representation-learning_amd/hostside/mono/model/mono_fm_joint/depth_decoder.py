"""Reference path mono/model/mono_fm_joint/depth_decoder.py -> mono.model.networks.
HRDepthDecoder / DIFFDepthDecoder (flags off in every BASELINE config) are out of scope."""
from ..networks import DepthDecoder  # noqa: F401
