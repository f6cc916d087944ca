"""Oracle: edge-aware first+second order smoothness and feature regularisation.

Test infrastructure -- see oracle/__init__.py.
"""
import torch
import torch.nn.functional as F


def gradient(D):
    """gradient, mono/model/mono_fm_joint/net.py:304-307: forward differences,
    returns (d/dx, d/dy), each one element shorter along its axis."""
    dy = D[:, :, 1:] - D[:, :, :-1]
    dx = D[:, :, :, 1:] - D[:, :, :, :-1]
    return dx, dy


def area_resize(img, h, w):
    """F.interpolate(img, (h, w), mode='area') (mono_fm_joint/net.py:283) ==
    adaptive average pooling; for integer factors a plain box average."""
    return F.adaptive_avg_pool2d(img, (h, w))


def _edge_terms(field, img, a):
    f_dx, f_dy = gradient(field)
    i_dx, i_dy = gradient(img)
    f_dxx, f_dxy = gradient(f_dx)
    f_dyx, f_dyy = gradient(f_dy)
    i_dxx, i_dxy = gradient(i_dx)
    i_dyx, i_dyy = gradient(i_dy)

    def term(fd, idf):
        return torch.mean(fd.abs() * torch.exp(-a * idf.abs().mean(1, True)))

    first = term(f_dx, i_dx) + term(f_dy, i_dy)
    second = term(f_dxx, i_dxx) + term(f_dxy, i_dxy) + term(f_dyx, i_dyx) + term(f_dyy, i_dyy)
    return first, second


def smooth_loss(disp, img):
    """get_smooth_loss, mono/model/mono_fm_joint/net.py:279-302 (a1 = a2 = 0.5);
    img is area-resized to disp's size first.  Returns smooth1 + smooth2."""
    h, w = disp.shape[2:]
    first, second = _edge_terms(disp, area_resize(img, h, w), 0.5)
    return first + second


def feature_regularization_loss(feature, img, dis, cvt):
    """get_feature_regularization_loss, mono/model/mono_fm_joint/net.py:309-330:
    same stencil with exp(-|dI|) (a = 1), combined as -dis*smooth1 + cvt*smooth2."""
    h, w = feature.shape[2:]
    first, second = _edge_terms(feature, area_resize(img, h, w), 1.0)
    return -dis * first + cvt * second


def mean_normalize(disp):
    """disp / (mean_HW(disp) + 1e-7), mono/model/mono_fm_joint_inpaint/net.py:122-124."""
    mean = disp.mean(2, True).mean(3, True)
    return disp / (mean + 1e-7)
