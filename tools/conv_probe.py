"""Time candidate conv shapes (bf16, channels_last, MIOpen find) forward and backward."""
import torch, time
torch.backends.cudnn.benchmark = True
dev = "cuda"
shapes = [  # (name, Cin, Cout, k, stride, H, W, needs_input_grad)
    ("dec.disp1 16->3 @192x640", 16, 3, 3, 1, 194, 642, True),
    ("dec.iconv1 16->16 @192x640", 16, 16, 3, 1, 194, 642, True),
    ("dec.upconv1 32->16 @96x320", 32, 16, 3, 1, 98, 322, True),
    ("dec.iconv2 32->32 @96x320", 32, 32, 3, 1, 98, 322, True),
    ("dec.disp2 32->3 @96x320", 32, 3, 3, 1, 98, 322, True),
    ("depth.disp1 256->1 @96x320", 256, 1, 3, 1, 98, 322, True),
    ("depth.merge1 256->256 @48x160", 256, 256, 3, 1, 50, 162, True),
    ("depth.iconv1 513->256 @48x160", 513, 256, 3, 1, 50, 162, True),
    ("stem 3->64 7x7s2", 3, 64, 7, 2, 192, 640, False),
    ("pose stem 6->64 7x7s2", 6, 64, 7, 2, 192, 640, False),
]
for name, ci, co, k, s, H, W, ig in shapes:
    conv = torch.nn.Conv2d(ci, co, k, s, padding=(3 if k == 7 else 0)).to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(12, ci, H, W, device=dev).to(memory_format=torch.channels_last).bfloat16().requires_grad_(ig)
    def run():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(x)
        return y
    y = run(); g = torch.randn_like(y)
    for _ in range(3):
        y = run(); y.backward(g)
    torch.cuda.synchronize()
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    n = 10
    tf = tb = 0.0
    for _ in range(n):
        a.record(); y = run(); b.record(); y.backward(g); c.record(); torch.cuda.synchronize()
        tf += a.elapsed_time(b); tb += b.elapsed_time(c)
    flops = 2 * 12 * co * ci * k * k * y.shape[2] * y.shape[3]
    print("%-34s fwd %7.1f us (%6.1f TF/s)  bwd %7.1f us" % (name, tf / n * 1e3, flops / (tf / n * 1e-3) / 1e12, tb / n * 1e3), flush=True)
