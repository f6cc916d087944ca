"""Reference path mono/model/mono_fm_joint/decoder.py -> mono.model.networks."""
from ..networks import ColorDecoder, Decoder  # noqa: F401
