"""Oracle: depth -> camera points -> source-frame pixel -> bilinear sample.

Closed-form, per-pixel restatement (no materialised meshgrid buffers, no bmm)
of the reference's geometry chain.  Test infrastructure -- see oracle/__init__.py.
"""
import torch


def disp_to_depth(disp, min_depth, max_depth):
    """mono/model/mono_fm_joint/net.py:157-162 (same body layers.py:33-38).

    scaled = 1/max + (1/min - 1/max) * disp ; depth = 1/scaled.
    """
    lo = 1.0 / max_depth
    hi = 1.0 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1.0 / scaled


def upsample_bilinear(x, out_h, out_w):
    """F.interpolate(x, [H, W], mode="bilinear", align_corners=False) as called at
    mono/model/mono_fm_joint/net.py:183, written out as explicit index math.

    Source coordinate s = (dst + 0.5) * in/out - 0.5, clamped below at 0; the
    upper neighbour is clamped to the last row/column.
    """
    b, c, in_h, in_w = x.shape
    dev, dt = x.device, x.dtype

    def axis(n_out, n_in):
        dst = torch.arange(n_out, device=dev, dtype=dt)
        src = (dst + 0.5) * (float(n_in) / float(n_out)) - 0.5
        src = torch.clamp(src, min=0.0)
        i0 = torch.floor(src).to(torch.long)
        i0 = torch.clamp(i0, max=n_in - 1)
        i1 = torch.clamp(i0 + 1, max=n_in - 1)
        lam = src - i0.to(dt)
        return i0, i1, lam

    y0, y1, ly = axis(out_h, in_h)
    x0, x1, lx = axis(out_w, in_w)
    rows0 = x[:, :, y0, :]
    rows1 = x[:, :, y1, :]
    ly = ly.view(1, 1, -1, 1)
    rows = rows0 * (1.0 - ly) + rows1 * ly
    lx = lx.view(1, 1, 1, -1)
    return rows[:, :, :, x0] * (1.0 - lx) + rows[:, :, :, x1] * lx


def pixel_grid(h, w, device=None, dtype=torch.float32):
    """Row-major pixel coordinates (x varies fastest), as np.meshgrid(range(W),
    range(H), indexing='xy') flattened in mono/model/mono_fm_joint/layers.py:49-55."""
    ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=dtype),
                            torch.arange(w, device=device, dtype=dtype), indexing="ij")
    return xs, ys


def backproject(depth, inv_K):
    """Backproject.forward, mono/model/mono_fm_joint/layers.py:57-61.

    depth [B,1,H,W], inv_K [B,4,4] -> homogeneous camera points [B,4,H*W]
    (only the 3x3 block of inv_K is used; a row of ones is appended).
    """
    b, _, h, w = depth.shape
    xs, ys = pixel_grid(h, w, depth.device, depth.dtype)
    xs = xs.reshape(1, -1)
    ys = ys.reshape(1, -1)
    m = inv_K[:, :3, :3]
    rays = torch.stack([
        m[:, 0, 0:1] * xs + m[:, 0, 1:2] * ys + m[:, 0, 2:3],
        m[:, 1, 0:1] * xs + m[:, 1, 1:2] * ys + m[:, 1, 2:3],
        m[:, 2, 0:1] * xs + m[:, 2, 1:2] * ys + m[:, 2, 2:3],
    ], dim=1)
    pts = depth.reshape(b, 1, -1) * rays
    ones = torch.ones(b, 1, h * w, device=depth.device, dtype=depth.dtype)
    return torch.cat([pts, ones], dim=1)


def project(points, K, T, h, w, eps=1e-7):
    """Project.forward, mono/model/mono_fm_joint/layers.py:73-82.

    P = (K @ T)[:, :3, :]; pixel = P @ points; u,v = xy / (z + eps); the grid is
    normalised with (W-1)/(H-1) and mapped to [-1, 1].  Returns [B,H,W,2].
    """
    b = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    z = cam[:, 2:3, :] + eps
    uv = cam[:, :2, :] / z
    u = uv[:, 0, :].reshape(b, h, w) / (w - 1)
    v = uv[:, 1, :].reshape(b, h, w) / (h - 1)
    return torch.stack([(u - 0.5) * 2.0, (v - 0.5) * 2.0], dim=-1)


def grid_sample_border(img, grid):
    """F.grid_sample(img, grid, padding_mode="border") with torch's defaults
    (bilinear, align_corners=False) as called at mono/model/mono_fm_joint/net.py:193
    and :222, written out as explicit gathers.

    xs = clamp(((gx + 1) * W - 1) / 2, 0, W-1); 4-tap bilinear with the upper
    neighbour clamped to the last row/column.
    """
    b, c, h, w = img.shape
    gx = grid[..., 0]
    gy = grid[..., 1]
    xs = torch.clamp(((gx + 1.0) * w - 1.0) / 2.0, 0.0, float(w - 1))
    ys = torch.clamp(((gy + 1.0) * h - 1.0) / 2.0, 0.0, float(h - 1))
    x0f = torch.floor(xs)
    y0f = torch.floor(ys)
    lx = (xs - x0f).unsqueeze(1)
    ly = (ys - y0f).unsqueeze(1)
    x0 = x0f.to(torch.long)
    y0 = y0f.to(torch.long)
    x1 = torch.clamp(x0 + 1, max=w - 1)
    y1 = torch.clamp(y0 + 1, max=h - 1)
    flat = img.reshape(b, c, h * w)

    def tap(yy, xx):
        idx = (yy * w + xx).reshape(b, 1, -1).expand(b, c, -1)
        return torch.gather(flat, 2, idx).reshape(b, c, *xs.shape[1:])

    top = tap(y0, x0) * (1.0 - lx) + tap(y0, x1) * lx
    bot = tap(y1, x0) * (1.0 - lx) + tap(y1, x1) * lx
    return top * (1.0 - ly) + bot * ly


def warp_source(src, disp_s, K, inv_K, T, min_depth, max_depth, out_h=None, out_w=None):
    """One (scale, frame) pass of generate_images_pred,
    mono/model/mono_fm_joint/net.py:181-194: upsample disp to the frame size,
    convert to depth, back-project, project with T, bilinear-sample ``src``.
    """
    h = out_h if out_h is not None else src.shape[2]
    w = out_w if out_w is not None else src.shape[3]
    disp = upsample_bilinear(disp_s, h, w)
    _, depth = disp_to_depth(disp, min_depth, max_depth)
    pts = backproject(depth, inv_K)
    grid = project(pts, K, T, h, w)
    return grid_sample_border(src, grid)


# ---------------------------------------------------------------------------
# pose parameters -> 4x4 transform
# ---------------------------------------------------------------------------

def rot_from_axisangle(vec):
    """rot_from_axisangle, mono/model/mono_fm_joint/net.py:248-277.
    vec [B,1,3] -> [B,4,4] Rodrigues rotation, axis = v / (|v| + 1e-7)."""
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca = torch.cos(angle).reshape(-1)
    sa = torch.sin(angle).reshape(-1)
    C = 1.0 - ca
    x = axis[:, 0, 0]
    y = axis[:, 0, 1]
    z = axis[:, 0, 2]
    zero = torch.zeros_like(x)
    one = torch.ones_like(x)
    rows = [
        x * (x * C) + ca, x * (y * C) - z * sa, z * (x * C) + y * sa, zero,
        x * (y * C) + z * sa, y * (y * C) + ca, y * (z * C) - x * sa, zero,
        z * (x * C) - y * sa, y * (z * C) + x * sa, z * (z * C) + ca, zero,
        zero, zero, zero, one,
    ]
    return torch.stack(rows, dim=1).reshape(-1, 4, 4)


def translation_matrix(t):
    """get_translation_matrix, mono/model/mono_fm_joint/net.py:238-246."""
    b = t.shape[0]
    M = torch.eye(4, device=t.device, dtype=t.dtype).unsqueeze(0).repeat(b, 1, 1)
    col = t.reshape(b, 3, 1)
    top = torch.cat([M[:, :3, :3], col], dim=2)
    return torch.cat([top, M[:, 3:, :]], dim=1)


def transformation_from_parameters(axisangle, translation, invert=False):
    """transformation_from_parameters, mono/model/mono_fm_joint/net.py:225-236.
    invert: M = R^T @ Trans(-t), else M = Trans(t) @ R."""
    R = rot_from_axisangle(axisangle)
    t = translation
    if invert:
        R = R.transpose(1, 2)
        t = -t
    Tm = translation_matrix(t)
    return torch.matmul(R, Tm) if invert else torch.matmul(Tm, R)
