"""Reference path mono/model/mono_fm_joint/encoder.py -> mono.model.networks."""
from ..networks import Encoder  # noqa: F401
