#!/usr/bin/env python3
"""GPU time of MIOpen's 3x3 convolution backward pieces at the ResNet shapes (20 calls captured in a HIP graph, so that host
launch overhead does not mask the kernels): data gradient only, weight gradient only (incl. its zero-fill / cast helper
kernels).   python tools/wgrad3x3_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

SHAPES = [(12, 48, 160, 64, 64, 1), (12, 24, 80, 128, 128, 1), (12, 12, 40, 256, 256, 1), (12, 6, 20, 512, 512, 1),
          (36, 48, 160, 64, 64, 1), (36, 24, 80, 128, 128, 1), (36, 12, 40, 256, 256, 1), (36, 6, 20, 512, 512, 1),
          (12, 48, 160, 128, 128, 2), (24, 48, 160, 64, 64, 1)]


def graph_time(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


def main():
    torch.backends.cudnn.benchmark = True
    tot = [0.0, 0.0, 0.0]
    print("%-26s %10s %10s %10s %8s" % ("B,H,W,Cin,N,stride", "fwd us", "dgrad us", "wgrad us", "GF"))
    for B, H, W, C, N, st in SHAPES:
        x = torch.randn(B, C, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(N, C, 3, 3, device="cuda") / (9 * C) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        Ho, Wo = (H + 2 - 3) // st + 1, (W + 2 - 3) // st + 1
        dy = torch.randn(B, N, Ho, Wo, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        f = lambda: torch.nn.functional.conv2d(x, w, stride=st, padding=1)
        d = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])
        g = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
        t = [graph_time(f), graph_time(d), graph_time(g)]
        for i in range(3):
            tot[i] += t[i]
        print("%-26s %10.1f %10.1f %10.1f %8.1f" % (",".join(map(str, (B, H, W, C, N, st))), *t, 2.0 * B * Ho * Wo * 9 * C * N / 1e9))
    print("%-26s %10.1f %10.1f %10.1f" % ("sum", *tot))


if __name__ == "__main__":
    main()
