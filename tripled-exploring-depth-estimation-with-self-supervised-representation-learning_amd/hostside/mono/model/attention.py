"""Attention gates on the depth skip connections: ``depth_skip_type`` 'ca' / 'pa' / 'asca' of
``mono_fm_joint_inpaint_disentangle`` (reference: mono/model/mono_fm_joint_inpaint/net.py:410-436 builds them,
mono/model/mono_fm_joint/layers.py:232-243, 283-332, 340-385 define them).  No BASELINE config selects them; they are here so
that a reference config or checkpoint that does loads and trains (same parameter names, same registration order, same values:
tests/test_model_vs_reference.py).  They run on a handful of [B, C, h, w] encoder maps per step through ATen -- a global pooling,
two tiny 1x1 convolutions and one gating multiply per level; nothing here is on the measured path."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _squeeze_excite(channel, reduction):
    return [nn.Conv2d(channel, channel // reduction, 1), nn.ReLU(inplace=True), nn.Conv2d(channel // reduction, channel, 1)]


def _moments(x):
    """Per-channel (std, mean) over the map, population variance (layers.py:289-293)."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = (x - mean).square().mean(dim=(2, 3), keepdim=True)
    return var.sqrt(), mean


class SqueezeAndExcitationBlock(nn.Module):
    """1x1 -> ReLU -> 1x1 on a [B, C, 1, 1] descriptor, no gate (layers.py:232-243)."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.block = nn.Sequential(*_squeeze_excite(channel, reduction))

    def forward(self, x):
        return self.block(x)


class ChannelDescriptorLayer(nn.Module):
    def forward(self, x):
        return _moments(x)


class CALayer(nn.Module):
    """x * sigmoid(excite(descriptor)): the descriptor is the channel mean ('ca'), the map itself ('pa': a per-pixel gate), or with
    ``contrast_aware`` std - mean / std of the channel (layers.py:340-385)."""

    def __init__(self, channel, reduction=16, contrast_aware=False, pix_att=False):
        super().__init__()
        self.pix_att, self.contrast_aware = pix_att, contrast_aware
        self.conv_att = nn.Sequential(*_squeeze_excite(channel, reduction), nn.Sigmoid())

    @staticmethod
    def rescaled_contrast_layer(x):
        std, mean = _moments(x)
        return std - mean / std

    def forward(self, x):
        if self.contrast_aware:
            gate = self.conv_att(self.rescaled_contrast_layer(x))
        elif self.pix_att:
            gate = self.conv_att(x)
        else:
            gate = self.conv_att(F.adaptive_avg_pool2d(x, 1))
        return x * gate


class AdaptivelyScaledCALayer(nn.Module):
    """Gate from BOTH channel moments: each refined by its own squeeze-excite block, fused by a 2C -> C bottleneck and a third
    block (layers.py:297-332).  Sub-modules are registered in the reference's order (the optimiser state of a checkpoint is
    indexed by parameter order)."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.local_channel_descriptors = ChannelDescriptorLayer()
        self.saeb_mean = SqueezeAndExcitationBlock(channel, reduction)
        self.saeb_std = SqueezeAndExcitationBlock(channel, reduction)
        self.small_descriptor_bottleneck = nn.Sequential(nn.Conv2d(2 * channel, channel, 1), nn.ReLU(inplace=True))
        self.saeb_final = SqueezeAndExcitationBlock(channel, reduction)
        self.gating_function = nn.Sigmoid()

    def forward(self, x):
        std, mean = self.local_channel_descriptors(x)
        fused = torch.cat((self.saeb_std(std), self.saeb_mean(mean)), 1)
        return x * self.gating_function(self.saeb_final(self.small_descriptor_bottleneck(fused)))


def skip_attention(kind, channels):
    """The gate ``depth_skip_type`` names, or None for the types without one ('use_half', '1x1', None)."""
    if kind == "ca":
        return CALayer(channels)
    if kind == "pa":
        return CALayer(channels, pix_att=True)
    if kind == "asca":
        return AdaptivelyScaledCALayer(channels)
    return None
