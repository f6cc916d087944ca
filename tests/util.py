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


def assert_argmin_parity(argmin, ref_idx, stack, tol=2e-5, max_frac=5e-3):
    """Index work is held to exact agreement: ``argmin`` (kernel) must equal ``ref_idx`` (oracle/reference torch.min
    over ``stack`` [B,n,H,W]) at every pixel, except where the two candidates involved are a floating-point near-tie,
    |stack[argmin] - stack[ref_idx]| < tol (the kernel's candidates carry ~1e-5 of fp32 re-association error).
    Also bounds the share of such pixels.  Returns that share."""
    a = argmin.detach().cpu().long()
    r = ref_idx.detach().cpu().long()
    stack = stack.detach().cpu()
    assert a.shape == r.shape and int(a.max()) < stack.shape[1]
    diff = a != r
    frac = float(diff.float().mean())
    if diff.any():
        ca = stack.gather(1, a.unsqueeze(1)).squeeze(1)
        cr = stack.gather(1, r.unsqueeze(1)).squeeze(1)
        gap = (ca - cr).abs()[diff]
        assert float(gap.max()) < tol, "arg-min differs at %d pixels that are not near-ties (largest gap %.3e)" % (
            int((gap >= tol).sum()), float(gap.max()))
    assert frac <= max_frac, frac
    return frac


def grad_close(a, b, tol, outlier_frac=2e-3, outlier_tol=None):
    """Max-normalised gradient comparison that tolerates a small share of elements in a non-smooth regime of the loss
    (e.g. robust-L1 arguments below eps, where d/dx sqrt(x^2 + eps^2) ~ x / eps turns 1e-5 of input rounding into
    1e-2 of gradient): all but ``outlier_frac`` of the elements within ``tol`` of the largest reference magnitude,
    and every element within ``outlier_tol`` (default 25 * tol).  Returns (quantile error, max error)."""
    a = a.detach().double().cpu().flatten()
    b = b.detach().double().cpu().flatten()
    err = (a - b).abs() / (b.abs().max() + 1e-30)
    k = max(1, int(err.numel() * (1 - outlier_frac)))
    q = float(err.kthvalue(k).values)
    mx = float(err.max())
    assert q < tol, "quantile error %.3e >= %.3e" % (q, tol)
    assert mx < (outlier_tol if outlier_tol is not None else 25 * tol), "max error %.3e" % mx
    return q, mx
