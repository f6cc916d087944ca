"""Oracle (test infrastructure, see oracle/__init__.py): colour-space conversion and the image robust-L1 map of the
auxiliary heads, restated in plain torch fp32.  Pinned by tests/golden/color_lab.npz (tools/gen_golden.py::gen_color,
outputs of the reference's own functions)."""
import torch

_M = ((0.412453, 0.357580, 0.180423), (0.212671, 0.715160, 0.072169), (0.019334, 0.119193, 0.950227))
_WHITE = (0.95047, 1.0, 1.08883)


def rgb2lab(rgb, l_cent=50.0, l_norm=50.0, ab_norm=110.0):
    """rgb2lab, mono/model/mono_fm_joint_inpaint/color_conversions.py:106-114 (rgb2xyz :6-27: sRGB gamma expansion
    with threshold 0.04045, linear map; xyz2lab :52-75: white-point scaling, cube root above 0.008856)."""
    lin = torch.where(rgb > 0.04045, ((rgb + 0.055) / 1.055) ** 2.4, rgb / 12.92)
    xyz = [m[0] * lin[:, 0] + m[1] * lin[:, 1] + m[2] * lin[:, 2] for m in _M]
    s = [xyz[i] / _WHITE[i] for i in range(3)]
    f = [torch.where(v > 0.008856, v ** (1 / 3.0), 7.787 * v + 16.0 / 116.0) for v in s]
    L = 116.0 * f[1] - 16.0
    a = 500.0 * (f[0] - f[1])
    b = 200.0 * (f[1] - f[2])
    return torch.stack([(L - l_cent) / l_norm, a / ab_norm, b / ab_norm], 1)


def robust_l1_map(pred, target, weight=1.0):
    """compute_perceptional_loss(target, pred) * weight as a [B,1,H,W] map: compute_auto_res_loss,
    mono/model/mono_fm_joint_inpaint/net.py:520-527 (robust_l1 with eps = 1e-3: mono_fm_joint/net.py:59-65)."""
    return torch.sqrt((target - pred) ** 2 + 1e-6).mean(1, True) * weight
