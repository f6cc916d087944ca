import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import tripled_amd
from tripled_amd import ops
g = torch.Generator().manual_seed(0)
N, C, H, W = 2, 16, 7, 9
x0 = torch.randint(0, 6, (N, C, H, W), generator=g).float() * 0.25
go = torch.randn(N, C, H, W, generator=g)
def run(fn, dev, cl):
    x = x0.detach().clone().to(dev)
    if cl: x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    y = fn(x)
    gg = go.to(dev)
    if cl: gg = gg.contiguous(memory_format=torch.channels_last)
    y.backward(gg)
    return y.detach().cpu(), x.grad.cpu()
yc, gc = run(lambda t: F.max_pool2d(t, 5, 1, 2), "cpu", False)
ya, ga = run(lambda t: F.max_pool2d(t, 5, 1, 2), "cuda", True)
yn, gn = run(lambda t: F.max_pool2d(t, 5, 1, 2), "cuda", False)
ym, gm = run(ops.maxpool5, "cuda", True)
print("aten nhwc vs cpu", float((ya - yc).abs().max()), float((ga - gc).abs().max()))
print("aten nchw vs cpu", float((yn - yc).abs().max()), float((gn - gc).abs().max()))
print("mine vs cpu", float((ym - yc).abs().max()), float((gm - gc).abs().max()))
