"""Op surface of the reference's layers.py (mono/model/mono_fm_joint/layers.py): the geometry
and SSIM modules keep their names and signatures for code that calls them directly.  They are
the UNFUSED compatibility surface built from stock torch ops; the training step does not go
through them -- compute_losses uses the fused HIP kernels in tripled_amd.ops.
Network building blocks are re-exported from mono.model.networks."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..networks import (Conv1x1, Conv3x3, Conv5x5, ConvBlock, CRPBlock, IdentityPartial,  # noqa: F401
                        upsample, upshuffle)
from ..attention import (AdaptivelyScaledCALayer, CALayer, ChannelDescriptorLayer,  # noqa: F401  (layers.py:232-243, 283-385)
                         SqueezeAndExcitationBlock)


def disp_to_depth(disp, min_depth, max_depth):
    """layers.py:33-38."""
    lo, hi = 1 / max_depth, 1 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1 / scaled


class Backproject(nn.Module):
    """layers.py:41-61: depth [B,1,H,W], inv_K [B,4,4] -> homogeneous points [B,4,H*W].
    The pixel grid is a registered (non-persistent) buffer, so it lives on the module's device
    instead of being re-uploaded on every call."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width
        ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32),
                                torch.arange(width, dtype=torch.float32), indexing="ij")
        pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(height * width)], 0)
        self.register_buffer("pix_coords", pix.unsqueeze(0), persistent=False)

    def forward(self, depth, inv_K):
        pix = self.pix_coords.to(depth.device)
        rays = torch.matmul(inv_K[:, :3, :3], pix)
        pts = depth.reshape(self.batch_size, 1, -1) * rays
        ones = torch.ones(self.batch_size, 1, self.height * self.width, device=depth.device, dtype=depth.dtype)
        return torch.cat([pts, ones], 1)


class Project(nn.Module):
    """layers.py:64-82: points [B,4,HW], K, T [B,4,4] -> sampling grid [B,H,W,2] in [-1,1]
    (normalised with W-1 / H-1 exactly as the reference does)."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        cam = torch.matmul(torch.matmul(K, T)[:, :3, :], points)
        uv = cam[:, :2, :] / (cam[:, 2:3, :] + self.eps)
        uv = uv.view(self.batch_size, 2, self.height, self.width).permute(0, 2, 3, 1)
        u = uv[..., 0] / (self.width - 1)
        v = uv[..., 1] / (self.height - 1)
        return (torch.stack([u, v], dim=-1) - 0.5) * 2


class SSIM(nn.Module):
    """layers.py:85-107: clamp((1 - SSIM)/2, 0, 1) over 3x3 reflect-padded windows."""

    def __init__(self):
        super().__init__()
        self.refl = nn.ReflectionPad2d(1)
        self.C1 = 0.01 ** 2
        self.C2 = 0.03 ** 2

    def _pool(self, x):
        return F.avg_pool2d(self.refl(x), 3, 1)

    def forward(self, x, y):
        mu_x, mu_y = self._pool(x), self._pool(y)
        sigma_x = self._pool(x * x) - mu_x * mu_x
        sigma_y = self._pool(y * y) - mu_y * mu_y
        sigma_xy = self._pool(x * y) - mu_x * mu_y
        n = (2 * mu_x * mu_y + self.C1) * (2 * sigma_xy + self.C2)
        d = (mu_x * mu_x + mu_y * mu_y + self.C1) * (sigma_x + sigma_y + self.C2)
        return torch.clamp((1 - n / d) / 2, 0, 1)
