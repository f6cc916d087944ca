"""td_conv3x3_wgrad against MIOpen's weight gradient (aten.convolution_backward, find mode on) per shape of the training step.
Both paths are captured into a HIP graph of REPS calls and replayed (an event-timed eager loop of ~20 us kernels measures the
launch path); MIOpen's time includes the zero-fill / cast kernels it runs around its split-K kernels.
usage: python tools/conv3x3_wgrad_bench.py [--json out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import tripled_amd  # noqa: F401
from tripled_amd import native
from tripled_amd.ops import _raw

SHAPES = [
    # (B, Ho, Wo, C, N, pad, what)
    (12, 48, 160, 64, 64, 1, "R50 layer1 conv2"),
    (12, 24, 80, 128, 128, 1, "R50 layer2 conv2"),
    (12, 12, 40, 256, 256, 1, "R50 layer3 conv2"),
    (12, 6, 20, 512, 512, 1, "R50 layer4 conv2"),
    (24, 48, 160, 64, 64, 1, "R18 layer1 (2 pairs)"),
    (24, 24, 80, 128, 128, 1, "R18 layer2"),
    (24, 12, 40, 256, 256, 1, "R18 layer3"),
    (24, 6, 20, 512, 512, 1, "R18 layer4"),
    (12, 6, 20, 512, 256, 0, "DepthDecoder iconv4"),
    (12, 6, 20, 256, 256, 0, "DepthDecoder merge4"),
    (12, 12, 40, 256, 256, 0, "DepthDecoder merge3 / Decoder iconv5"),
    (12, 24, 80, 256, 256, 0, "DepthDecoder merge2"),
    (12, 48, 160, 256, 256, 0, "DepthDecoder merge1"),
    (12, 6, 20, 2048, 256, 0, "Decoder upconv5"),
    (12, 12, 40, 256, 128, 0, "Decoder upconv4"),
    (12, 24, 80, 128, 128, 0, "Decoder iconv4"),
    (12, 24, 80, 128, 64, 0, "Decoder upconv3"),
    (12, 48, 160, 64, 64, 0, "Decoder iconv3"),
]
REPS = 20


def graph_time(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        t0.record()
        g.replay()
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) * 1e3 / REPS)
    return best


def main():
    lib = native.load()
    torch.backends.cudnn.benchmark = True
    rows = []
    for B, Ho, Wo, C, N, pad, what in SHAPES:
        Hi, Wi = Ho + 2 - 2 * pad, Wo + 2 - 2 * pad
        x = torch.randn(B, C, Hi, Wi, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, N, Ho, Wo, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(N, C, 3, 3, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ws = torch.empty(lib.td_conv3x3_wgrad_workspace_floats(B, Ho, Wo, C, N), device="cuda")
        dw = torch.empty_like(w)

        def mine():
            native.check(lib.td_conv3x3_wgrad(_raw(dy), _raw(x), B, Ho, Wo, C, N, pad, 1, _raw(dw), native.ptr(ws), native.stream()),
                         "td_conv3x3_wgrad")

        def miopen():
            return torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]

        ref = miopen().float()
        mine()
        torch.cuda.synchronize()
        rel = float((dw.float() - ref).abs().max() / ref.abs().max())
        t_td, t_mi = graph_time(mine), graph_time(miopen)
        flop = 2.0 * B * Ho * Wo * C * N * 9
        rows.append(dict(shape=[B, Ho, Wo, C, N, pad], what=what, td_us=round(t_td, 1), miopen_us=round(t_mi, 1),
                         td_TFps=round(flop / t_td / 1e6, 1), miopen_TFps=round(flop / t_mi / 1e6, 1), rel_diff=rel))
        print("%-40s %-28s td %7.1f us  miopen %7.1f us  (%.2fx)  rel diff %.1e" % (what, rows[-1]["shape"], t_td, t_mi, t_mi / t_td, rel), flush=True)
    print("sum td %.1f us, miopen %.1f us" % (sum(r["td_us"] for r in rows), sum(r["miopen_us"] for r in rows)))
    if "--json" in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
