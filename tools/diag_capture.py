#!/usr/bin/env python3
"""Where does a captured training step stop being finite?

  python tools/diag_capture.py MODE [--steps N] [--find on|off] [--arch 50|18] [--batch B]

MODE = eager | default | side: run the C2 step eagerly, or captured on PyTorch's own capture stream, or
captured on the warm-up stream, and after every iteration print the total loss, every loss_dict entry,
the gradient norm, and the first non-finite output / gradient / parameter.
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import tripled_amd  # noqa: F401,E402
from mmcv import Config  # noqa: E402
from mono.datasets.synthetic import synthetic_batch  # noqa: E402
from mono.model import MONO  # noqa: E402
from tripled_amd.step import TrainStep, warm_up  # noqa: E402


def report(tag, step, model):
    torch.cuda.synchronize()
    loss = float(step.loss)
    gn = float(step.grad_norm) if step.grad_norm is not None else float("nan")
    print("[%s] loss %.6f grad_norm %.4f" % (tag, loss, gn))
    if not math.isfinite(loss) or not math.isfinite(gn):
        for k, v in step.losses.items():
            print("    loss[%s] = %r" % (k, float(v)))
        for k, v in step.outputs.items():
            if torch.is_tensor(v) and v.is_floating_point() and not bool(torch.isfinite(v).all()):
                print("    output %s: %d non-finite of %d" % (k, int((~torch.isfinite(v)).sum()), v.numel()))
    bad_g = [(n, int((~torch.isfinite(p.grad)).sum())) for n, p in model.named_parameters()
             if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    bad_p = [(n, int((~torch.isfinite(p)).sum())) for n, p in model.named_parameters()
             if not bool(torch.isfinite(p).all())]
    bad_b = [n for n, b in model.named_buffers() if b.is_floating_point() and not bool(torch.isfinite(b).all())]
    if bad_g:
        print("    %d gradients non-finite; first: %s" % (len(bad_g), bad_g[:4]))
    if bad_p:
        print("    %d parameters non-finite; first: %s" % (len(bad_p), bad_p[:4]))
    if bad_b:
        print("    %d buffers non-finite; first: %s" % (len(bad_b), bad_b[:4]))
    sys.stdout.flush()
    return math.isfinite(loss) and not bad_p


STASH = {}


def install_recon_probe():
    """Keep the inputs and the result of every masked_reconstruction_sum call (static tensors under replay)."""
    from tripled_amd import ops
    real = ops.masked_reconstruction_sum

    def probe(pred, target, hole):
        out = real(pred, target, hole)
        STASH[tuple(pred.shape[2:])] = (pred.detach(), target.detach(), hole.detach(), out.detach())
        return out
    ops.masked_reconstruction_sum = probe


def check_recon(step):
    from tripled_amd import ops
    for size, (pred, target, hole, out) in STASH.items():
        with torch.no_grad():
            again = ops._ReconSum.apply(pred.float().contiguous(), target.float().contiguous(), hole.float().contiguous())
        torch.cuda.synchronize()
        p = pred.float()
        print("    recon %s: graph S %.4f, recomputed-from-static-inputs S %.4f, sum(hole) %.1f, pred range [%.3g, %.3g] finite %s, "
              "target range [%.3g, %.3g], hole range [%.3g, %.3g]" % (
                  size, float(out), float(again), float(hole.sum()), float(p.min()), float(p.max()),
                  bool(torch.isfinite(p).all()), float(target.min()), float(target.max()), float(hole.min()), float(hole.max())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["eager", "default", "side"])
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--find", default="on")
    ap.add_argument("--config", default=os.path.join(ROOT, "config", "cfg_kitti_tripleD.py"))
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = Config.fromfile(a.config)
    torch.backends.cudnn.benchmark = a.find == "on"
    m = cfg.model
    torch.manual_seed(1024)
    model = MONO.module_dict[m["name"]](m).to(dev).to(memory_format=torch.channels_last)
    model.train()
    batch = synthetic_batch(m["imgs_per_gpu"], m["height"], m["width"], seed=1000, device=dev,
                            frame_ids=tuple(m["frame_ids"]))
    install_recon_probe()
    step = TrainStep(model, cfg, batch, torch.bfloat16)
    side = torch.cuda.Stream()
    for i in range(a.warmup):
        warm_up(step, 1, side)
        report("%s warm-up %d" % (a.mode, i), step, model)
    if a.mode == "eager":
        for i in range(a.steps):
            warm_up(step, 1, side)
            if not report("eager step %d" % i, step, model):
                break
        return
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side if a.mode == "side" else None):
        step()
    for i in range(a.steps):
        graph.replay()
        ok = report("%s replay %d" % (a.mode, i), step, model)
        if i < 2 or not ok:
            check_recon(step)
        if not ok:
            break


if __name__ == "__main__":
    main()
