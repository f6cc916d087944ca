"""Differentiable sRGB -> CIE XYZ (D65) -> CIE Lab (reference:
mono/model/mono_fm_joint_inpaint/color_conversions.py:6-27, 52-75, 106-114).  Only the forward
direction is needed by the colourisation head."""
import torch

_RGB2XYZ = ((0.412453, 0.357580, 0.180423),
            (0.212671, 0.715160, 0.072169),
            (0.019334, 0.119193, 0.950227))
_WHITE = (0.95047, 1.0, 1.08883)


def rgb2xyz(rgb):
    """rgb in [0,1], NCHW.  sRGB gamma expansion (threshold 0.04045) then the linear map."""
    lin = torch.where(rgb > 0.04045, ((rgb + 0.055) / 1.055) ** 2.4, rgb / 12.92)
    r, g, b = lin[:, 0], lin[:, 1], lin[:, 2]
    rows = [m[0] * r + m[1] * g + m[2] * b for m in _RGB2XYZ]
    return torch.stack(rows, dim=1)


def xyz2lab(xyz):
    # per-channel scalar division: no host->device constant upload (legal inside HIP graph capture)
    s = torch.stack([xyz[:, 0] / _WHITE[0], xyz[:, 1] / _WHITE[1], xyz[:, 2] / _WHITE[2]], dim=1)
    f = torch.where(s > 0.008856, s ** (1 / 3.0), 7.787 * s + 16.0 / 116.0)
    L = 116.0 * f[:, 1] - 16.0
    a = 500.0 * (f[:, 0] - f[:, 1])
    b = 200.0 * (f[:, 1] - f[:, 2])
    return torch.stack([L, a, b], dim=1)


def rgb2lab(rgb, opt):
    """Normalised Lab: ((L - l_cent) / l_norm, a / ab_norm, b / ab_norm)."""
    lab = xyz2lab(rgb2xyz(rgb))
    return torch.cat(((lab[:, 0:1] - opt.l_cent) / opt.l_norm, lab[:, 1:] / opt.ab_norm), dim=1)
