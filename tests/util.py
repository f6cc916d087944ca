"""Shared helpers for the parity tests (synthetic KITTI-shaped inputs)."""
import numpy as np
import torch
import torch.nn.functional as F


def smooth_image(g, b, c, h, w):
    x = torch.rand(b, c, h // 4 + 2, w // 4 + 2, generator=g)
    x = F.interpolate(x, size=(h, w), mode="bicubic", align_corners=False)
    x = x + 0.05 * torch.rand(b, c, h, w, generator=g)
    return x.clamp(0, 1).contiguous()


def kitti_K(b, h, w):
    K = np.array([[0.58 * w, 0, 0.5 * w, 0], [0, 1.92 * h, 0.5 * h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    inv_K = np.linalg.pinv(K)
    return (torch.from_numpy(K).unsqueeze(0).repeat(b, 1, 1), torch.from_numpy(inv_K).unsqueeze(0).repeat(b, 1, 1))


def make_triplet(g, B, H, W):
    """target + two sources cut from one smooth canvas with small shifts (temporally coherent)."""
    base = smooth_image(g, B, 3, H + 8, W + 8)
    frames = {}
    for f, (dy, dx) in ((0, (4, 4)), (-1, (4, 2)), (1, (5, 6))):
        img = base[:, :, dy:dy + H, dx:dx + W]
        frames[f] = (img + 0.01 * torch.randn(B, 3, H, W, generator=g)).clamp(0, 1).contiguous()
    return frames


def random_poses(g, B, rot=0.01, trans=0.15):
    from oracle import geometry
    Ts = []
    for f in (-1, 1):
        axis = rot * torch.randn(B, 1, 3, generator=g)
        t = trans * torch.randn(B, 1, 3, generator=g)
        Ts.append(geometry.transformation_from_parameters(axis, t[:, 0], invert=(f < 0)))
    return Ts


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
