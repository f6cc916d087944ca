"""Reference path mono/model/mono_fm_joint/resnet.py -> mono.model.networks."""
from ..networks import BasicBlock, Bottleneck, ResNet, resnet18, resnet34, resnet50, resnet101  # noqa: F401
