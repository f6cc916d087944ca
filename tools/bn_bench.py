"""Time the HIP BatchNorm(+add)(+ReLU) against ATen/MIOpen on the encoder shapes of config C2 (bf16, NHWC)."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import tripled_amd  # noqa
from tripled_amd import ops

shapes = [(12, 64, 96, 320), (12, 64, 48, 160), (12, 256, 48, 160), (12, 128, 24, 80), (12, 512, 24, 80),
          (12, 256, 12, 40), (12, 1024, 12, 40), (12, 512, 6, 20), (12, 2048, 6, 20)]


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    g.replay()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for shp in shapes:
    N, C, H, W = shp
    x = torch.randn(shp, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    res = torch.randn(shp, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    dy = torch.randn(shp, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.ones(C, device="cuda", requires_grad=True); b = torch.zeros(C, device="cuda", requires_grad=True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    for with_res in (False, True):
        def aten():
            y = F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
            if with_res:
                y = y + res
            y = F.relu(y)
            torch.autograd.grad(y, [x, w, b] + ([res] if with_res else []), dy)
        def hip():
            y = ops.batchnorm_act(x, w, b, rm, rv, 0.1, 1e-5, residual=res if with_res else None, relu=True)
            torch.autograd.grad(y, [x, w, b] + ([res] if with_res else []), dy)
        ta, th = timeit(aten), timeit(hip)
        mb = N * C * H * W * 2 / 1e6
        print("%-22s res=%d  tensor %.1f MB  aten %.1f us  hip %.1f us  (%.2fx)" % (shp, with_res, mb, ta, th, ta / th), flush=True)
