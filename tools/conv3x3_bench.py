#!/usr/bin/env python3
"""Per-shape timing + check of the hand-written MFMA 3x3 convolution (csrc/td_conv3x3.hip, statistics epilogue on) against
MIOpen's kernel for the same convolution (F.conv2d, bf16 channels_last, find mode on).   python tools/conv3x3_bench.py"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import tripled_amd  # noqa: F401,E402
from tripled_amd import native  # noqa: E402
from tripled_amd.ops import _raw  # noqa: E402

SHAPES = []
for B in (12, 36):
    g = 1 if B == 12 else 3
    SHAPES += [(B, 48, 160, 64, 64, 1, 1, g), (B, 48, 160, 128, 128, 2, 1, g), (B, 24, 80, 128, 128, 1, 1, g), (B, 24, 80, 256, 256, 2, 1, g),
               (B, 12, 40, 256, 256, 1, 1, g), (B, 12, 40, 512, 512, 2, 1, g), (B, 6, 20, 512, 512, 1, 1, g)]
SHAPES += [(12, 50, 162, 520, 256, 1, 0, 1), (12, 50, 162, 256, 256, 1, 0, 1), (12, 26, 82, 520, 256, 1, 0, 1), (12, 8, 22, 2048, 256, 1, 0, 1)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    lib = native.load()
    out = []
    print("%-36s %9s %9s %8s %8s %9s" % ("B,Hi,Wi,Cin,N,stride,pad,groups", "td us", "MIOpen us", "TF/s td", "TF/s MI", "rel err"))
    for (B, Hi, Wi, C, N, stride, pad, groups) in SHAPES:
        x = torch.randn(B, C, Hi, Wi, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(N, C, 3, 3, device="cuda") / (9 * C) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        Ho, Wo = (Hi + 2 * pad - 3) // stride + 1, (Wi + 2 * pad - 3) // stride + 1
        M = B * Ho * Wo
        y = torch.empty(B, N, Ho, Wo, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        S = lib.td_conv1x1_stat_rows(M, groups, N)
        part = torch.empty(groups * S * N * 2, device="cuda")
        st = native.stream()

        def td():
            native.check(lib.td_conv3x3_fwd(_raw(x), _raw(w), B, groups, Hi, Wi, C, N, stride, pad, _raw(y), native.ptr(part), st), "conv3x3")

        def mi():
            return F.conv2d(x, w, stride=stride, padding=pad)

        td()
        ref = mi()
        torch.cuda.synchronize()
        err = float((y.float() - ref.float()).abs().max()) / max(float(ref.float().abs().max()), 1e-9)
        t_td, t_mi = timeit(td, args.iters), timeit(mi, args.iters)
        flops = 2.0 * M * 9 * C * N
        rec = dict(shape=[B, Hi, Wi, C, N, stride, pad, groups], td_us=round(t_td, 2), miopen_us=round(t_mi, 2),
                   td_TFps=round(flops / t_td / 1e6, 1), miopen_TFps=round(flops / t_mi / 1e6, 1), rel_err_vs_miopen=err)
        out.append(rec)
        print("%-36s %9.2f %9.2f %8.1f %8.1f %9.2e" % (",".join(map(str, rec["shape"])), t_td, t_mi, rec["td_TFps"], rec["miopen_TFps"], err))
    print("sum over shapes: td %.1f us, MIOpen %.1f us" % (sum(r["td_us"] for r in out), sum(r["miopen_us"] for r in out)))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
