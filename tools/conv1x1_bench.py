#!/usr/bin/env python3
"""Per-shape timing of the hand-written MFMA 1x1-convolution GEMM (csrc/td_conv1x1.hip, statistics epilogue on) against
MIOpen's kernel for the same convolution (F.conv2d, bf16 channels_last, find mode on) and against the bytes the layer must
move: 2 (M K + N K + M N).  ResNet50 shapes of cfg_kitti_tripleD (B = 12 depth encoder / B = 36 auto-encoder passes).

  python tools/conv1x1_bench.py [--iters 30] [--json out.json]
"""
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
    SHAPES += [(B, 48, 160, 64, 64, 1, g), (B, 48, 160, 64, 256, 1, g), (B, 48, 160, 256, 64, 1, g),
               (B, 48, 160, 256, 128, 1, g), (B, 48, 160, 256, 512, 2, g), (B, 24, 80, 128, 512, 1, g), (B, 24, 80, 512, 128, 1, g),
               (B, 24, 80, 512, 256, 1, g), (B, 24, 80, 512, 1024, 2, g), (B, 12, 40, 256, 1024, 1, g), (B, 12, 40, 1024, 256, 1, g),
               (B, 12, 40, 1024, 512, 1, g), (B, 12, 40, 1024, 2048, 2, g), (B, 6, 20, 512, 2048, 1, g), (B, 6, 20, 2048, 512, 1, g)]


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
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    lib = native.load()
    out = []
    print("%-34s %9s %9s %9s %8s" % ("B,Hi,Wi,K,N,stride,groups", "td us", "MIOpen us", "GB/s td", "TF/s td"))
    for (B, Hi, Wi, K, N, stride, groups) in SHAPES:
        x = torch.randn(B, K, Hi, Wi, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(N, K, 1, 1, device="cuda") / K ** 0.5).to(torch.bfloat16)
        Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
        M = B * Ho * Wo
        y = torch.empty(B, N, Ho, Wo, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        S = lib.td_conv1x1_stat_rows(M, groups, N)
        part = torch.empty(groups * S * N * 2, device="cuda")
        st = native.stream()

        def td():
            native.check(lib.td_conv1x1_fwd(_raw(x), _raw(w), M, groups, K, N, Hi, Wi, stride, _raw(y), native.ptr(part), st), "conv")

        wcl = w.contiguous(memory_format=torch.channels_last)

        def mi():
            F.conv2d(x, wcl, stride=stride)

        t_td, t_mi = timeit(td, args.iters), timeit(mi, args.iters)
        nbytes = 2.0 * (M * K + N * K + M * N)
        flops = 2.0 * M * K * N
        rec = dict(shape=[B, Hi, Wi, K, N, stride, groups], td_us=round(t_td, 2), miopen_us=round(t_mi, 2),
                   td_GBps=round(nbytes / t_td / 1e3, 1), td_TFps=round(flops / t_td / 1e6, 1))
        out.append(rec)
        print("%-34s %9.2f %9.2f %9.1f %8.1f" % (",".join(map(str, rec["shape"])), t_td, t_mi, rec["td_GBps"], rec["td_TFps"]))
    tot_td, tot_mi = sum(r["td_us"] for r in out), sum(r["miopen_us"] for r in out)
    print("sum over shapes: td %.1f us, MIOpen %.1f us" % (tot_td, tot_mi))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(dict(rows=out, sum_td_us=tot_td, sum_miopen_us=tot_mi), f, indent=1)


if __name__ == "__main__":
    main()
