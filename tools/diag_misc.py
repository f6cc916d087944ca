#!/usr/bin/env python3
"""Two diagnostics: (1) the smoothness backward at the C4 scale-0 shape, term by term; (2) which modules of the C2
model give bitwise different outputs when the same forward runs twice from the same state and RNG."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import tripled_amd  # noqa: F401,E402
from tripled_amd import ops  # noqa: E402


def smooth_case():
    from oracle import smooth
    from tests.util import smooth_image
    for (B, h, w) in [(12, 96, 320), (4, 160, 512), (1, 160, 512), (4, 160, 448)]:
        g = torch.Generator().manual_seed(4)
        disp = (0.05 + 0.9 * smooth_image(g, B, 1, h, w)).contiguous()
        img = smooth_image(g, B, 3, h, w).contiguous()
        weight = 1e-3 / 2 / 4
        out = {}
        for normalize in (False, True):
            d = disp.cuda().requires_grad_(True)
            loss = ops.smooth_loss(d, img.cuda(), normalize, weight)
            loss.backward()
            dr = disp.double().clone().requires_grad_(True)
            dn = dr / (dr.mean((2, 3), keepdim=True) + 1e-7) if normalize else dr
            ref = weight * smooth.smooth_loss(dn, img.double())
            ref.backward()
            d32 = disp.clone().requires_grad_(True)
            dn32 = smooth.mean_normalize(d32) if normalize else d32
            (weight * smooth.smooth_loss(dn32, img)).backward()
            e_k = float((d.grad.cpu().double() - dr.grad).abs().max() / dr.grad.abs().max())
            e_o = float((d32.grad.double() - dr.grad).abs().max() / dr.grad.abs().max())
            print("smooth B=%d %dx%d normalize=%s: kernel-vs-f64 %.3e, f32-oracle-vs-f64 %.3e, loss %.6e vs %.6e" % (
                B, h, w, normalize, e_k, e_o, float(loss), float(ref)))


def determinism():
    from mmcv import Config
    from mono.datasets.synthetic import synthetic_batch
    from mono.model import MONO
    cfg = Config.fromfile(os.path.join(ROOT, "config", "cfg_kitti_tripleD.py"))
    m = cfg.model
    torch.manual_seed(1024)
    dev = torch.device("cuda", 0)
    model = MONO.module_dict[m["name"]](m).to(dev).to(memory_format=torch.channels_last)
    model.train()
    batch = synthetic_batch(m["imgs_per_gpu"], m["height"], m["width"], seed=1000, device=dev, frame_ids=tuple(m["frame_ids"]))
    records = []

    def hook(name):
        def fn(mod, args, out):
            if torch.is_tensor(out):
                records[-1].append((name, type(mod).__name__, out.detach().clone()))
        return fn
    for name, mod in model.named_modules():
        if not list(mod.children()):
            mod.register_forward_hook(hook(name))
    state = {k: v.clone() for k, v in model.state_dict().items()}
    for run in range(2):
        model.load_state_dict(state)
        torch.manual_seed(7)
        records.append([])
        with torch.autocast("cuda", dtype=torch.bfloat16):
            outputs, losses = model(dict(batch))
        torch.cuda.synchronize()
        records[-1].append(("LOSSES", "dict", torch.stack([v.float().mean() for v in losses.values()])))
    a, b = records
    print("modules recorded:", len(a), len(b))
    shown = 0
    for (n1, t1, x1), (n2, t2, x2) in zip(a, b):
        same = torch.equal(x1, x2)
        if not same:
            diff = (x1.float() - x2.float()).abs()
            print("  DIFFERS %-60s %-14s max %.3e  frac %.4f shape %s" % (n1, t1, float(diff.max()), float((diff > 0).float().mean()),
                                                                           tuple(x1.shape)))
            shown += 1
            if shown >= 12:
                break
    if not shown:
        print("  forward is bitwise reproducible")


if __name__ == "__main__":
    smooth_case()
    determinism()
