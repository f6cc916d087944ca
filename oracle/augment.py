"""Oracle (test infrastructure): the colour jitter of the input pipeline in torch float32, i.e. torchvision's tensor
formulas for ColorJitter's four stages (the reference applies torchvision.transforms.ColorJitter to PIL images,
mono/datasets/mono_dataset.py:83-101,146-152).  torchvision is not installed here, so the formulas are restated from
its documented semantics; tests/test_augment_cpu.py pins this restatement against PIL's own ImageEnhance / HSV
primitives (which is what torchvision calls for PIL inputs) to within the 8-bit quantisation of each stage."""
import torch


def _gray(x):
    return (0.299 * x[:, 0] + 0.587 * x[:, 1] + 0.114 * x[:, 2]).unsqueeze(1)


def _blend(a, b, f):
    return (f * a + (1.0 - f) * b).clamp(0, 1)


def _rgb2hsv(img):
    r, g, b = img.unbind(1)
    maxc = img.max(1).values
    minc = img.min(1).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), 1)


def _hsv2rgb(img):
    h, s, v = img.unbind(1)
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = (v * (1.0 - s)).clamp(0, 1)
    q = (v * (1.0 - s * f)).clamp(0, 1)
    t = (v * (1.0 - s * (1.0 - f))).clamp(0, 1)
    mask = i.unsqueeze(1) == torch.arange(6).view(1, -1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), 1)
    a2 = torch.stack((t, v, v, q, p, p), 1)
    a3 = torch.stack((p, p, t, v, v, q), 1)
    return torch.stack([(mask * a).sum(1) for a in (a1, a2, a3)], 1)


def color_jitter(img, order, brightness, contrast, saturation, hue):
    """img [N,3,H,W] float in [0,1]; one parameter set for the whole batch."""
    for op in order:
        if op == 0:
            img = _blend(img, torch.zeros_like(img), brightness)
        elif op == 1:
            img = _blend(img, _gray(img).mean((1, 2, 3), keepdim=True), contrast)
        elif op == 2:
            img = _blend(img, _gray(img), saturation)
        else:
            hsv = _rgb2hsv(img)
            h = (hsv[:, 0] + hue) % 1.0
            img = _hsv2rgb(torch.stack((h, hsv[:, 1], hsv[:, 2]), 1))
    return img


def expand_frames(frames_u8, aug):
    """frames_u8 [N,3,H,W] uint8, aug [N,9] -> (color, color_aug) float32, the contract of td_color_jitter."""
    color = frames_u8.float() / 255.0
    out = color.clone()
    for n in range(frames_u8.shape[0]):
        a = aug[n].tolist()
        if a[0] != 0:
            out[n:n + 1] = color_jitter(color[n:n + 1], [int(v) for v in a[1:5]], a[5], a[6], a[7], a[8])
    return color, out
