#!/usr/bin/env python3
"""Backward of the 1x1 convolutions at the ResNet50 shapes: ATen/MIOpen convolution_backward (data + weight gradient) against
plain GEMMs on the same channels-last matrices (dX = dY W through torch.mm / hipBLASLt, dW = dY^T X).  Decides which path
tripled_amd.ops._Conv1x1BatchNormAct.backward takes.   python tools/wgrad_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

SHAPES = [(12, 48, 160, 64, 64), (12, 48, 160, 64, 256), (12, 48, 160, 256, 64), (12, 24, 80, 128, 512), (12, 24, 80, 512, 128),
          (12, 12, 40, 256, 1024), (12, 12, 40, 1024, 256), (12, 6, 20, 512, 2048), (12, 6, 20, 2048, 512),
          (36, 48, 160, 64, 256), (36, 48, 160, 256, 64), (36, 24, 80, 128, 512), (36, 12, 40, 1024, 256), (36, 6, 20, 512, 2048)]


def timeit(fn, iters=20):
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
    only_td = "--only-td" in sys.argv          # under rocprofv3: just the hand-written kernels, per-kernel GPU time from the trace
    torch.backends.cudnn.benchmark = True
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    from tripled_amd.ops import _raw
    lib = native.load()
    tot = [0.0, 0.0, 0.0, 0.0, 0.0]
    print("%-24s %10s %10s %10s %10s %10s" % ("B,H,W,K,N", "aten both", "aten dgrad", "mm dgrad", "mm wgrad", "td wgrad"))
    for B, H, W, K, N in SHAPES:
        x = torch.randn(B, K, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(N, K, 1, 1, device="cuda") / K ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, N, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        M = B * H * W
        x2 = x.permute(0, 2, 3, 1).reshape(M, K)
        dy2 = dy.permute(0, 2, 3, 1).reshape(M, N)
        w2 = w.reshape(N, K)

        def both():
            torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, True, False])

        def dgrad():
            torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])

        def mm_d():
            torch.mm(dy2, w2)

        def mm_w():
            torch.mm(dy2.t(), x2)

        dw = torch.empty_like(w)
        ws = torch.empty(lib.td_conv1x1_wgrad_workspace_floats(M, K, N), device="cuda")
        st = native.stream()

        def td_w():      # the hand-written kernel + its ordered slab sum (csrc/td_conv1x1.hip)
            native.check(lib.td_conv1x1_wgrad(_raw(dy), _raw(x), M, K, N, H, W, 1, native.DTYPE_CODES[dw.dtype], _raw(dw), native.ptr(ws), st), "wgrad")

        t = [0.0, 0.0, 0.0, 0.0, timeit(td_w)] if only_td else [timeit(both), timeit(dgrad), timeit(mm_d), timeit(mm_w), timeit(td_w)]
        for i in range(5):
            tot[i] += t[i]
        print("%-24s %10.1f %10.1f %10.1f %10.1f %10.1f" % (",".join(map(str, (B, H, W, K, N))), *t))
    print("%-24s %10.1f %10.1f %10.1f %10.1f %10.1f   (us, sum)" % ("sum", *tot))


if __name__ == "__main__":
    main()
