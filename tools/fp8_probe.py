#!/usr/bin/env python3
"""Is an fp8 (OCP e4m3fn) GEMM available through hipBLASLt on this box, and how does it compare with the bf16 1x1
convolution it would replace?  (ResNet50 bottleneck 1x1 shapes at C2 / C5.)"""
import os
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch
import torch.nn.functional as F

dev = "cuda"


torch.backends.cudnn.benchmark = True      # MIOpen find mode, as in the training configs


def bench(fn, iters=20):
    """GPU time per call: the calls are replayed from a HIP graph so that host launch overhead does not count."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (5 * iters) * 1e6


print("torch", torch.__version__, "float8_e4m3fn:", hasattr(torch, "float8_e4m3fn"))
for (n, h, w, k, c) in [(12, 48, 160, 256, 64), (12, 48, 160, 64, 256), (12, 24, 80, 512, 128), (12, 24, 80, 128, 512),
                        (12, 12, 40, 1024, 256), (12, 6, 20, 2048, 512), (12, 48, 160, 256, 256)]:
    m = n * h * w
    x = torch.randn(n, k, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(c, k, 1, 1, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t_conv = bench(lambda: F.conv2d(x, wt))
    go = torch.randn(n, c, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t_bwd = bench(lambda: torch.ops.aten.convolution_backward(go, x, wt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                              [True, True, False]))
    a2 = x.permute(0, 2, 3, 1).reshape(m, k)
    g2 = go.permute(0, 2, 3, 1).reshape(m, c)
    w2 = wt.reshape(c, k)
    t_mm = bench(lambda: a2 @ w2.t())
    t_mbwd = bench(lambda: (g2 @ w2, g2.t() @ a2))
    line = "M=%6d K=%4d N=%4d: conv2d bf16 fwd %6.1f bwd %6.1f us | matmul bf16 fwd %6.1f bwd %6.1f us" % (m, k, c, t_conv, t_bwd, t_mm, t_mbwd)
    try:
        a8 = a2.to(torch.float8_e4m3fn)
        b8 = wt.reshape(c, k).to(torch.float8_e4m3fn)          # [N,K] row-major == [K,N] column-major
        one = torch.ones((), device=dev)
        f = lambda: torch._scaled_mm(a8, b8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
        y = f()
        ref = (a8.float() @ b8.float().t())
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        line += " | fp8 _scaled_mm %6.1f us (err %.0e)" % (bench(f), err)
        t_q = bench(lambda: a2.to(torch.float8_e4m3fn))
        line += ", bf16->fp8 cast %5.1f us" % t_q
    except Exception as e:      # noqa: BLE001
        line += ", _scaled_mm fp8 unavailable: %s: %s" % (type(e).__name__, str(e)[:120])
    print(line)
