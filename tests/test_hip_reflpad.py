"""GPU parity: channels-last ReflectionPad2d(1) kernel and its adjoint vs torch."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H,W", [(2, 16, 6, 20), (1, 256, 3, 3), (2, 8, 2, 5), (1, 64, 12, 40),
                                     (12, 520, 48, 160), (12, 16, 192, 640)])            # C2: widest decoder pad, image-decoder head
def test_reflpad1(dtype, N, C, H, W):
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, C, H, W, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya = ops.reflpad1(xa)
    yb = F.pad(xb, (1, 1, 1, 1), mode="reflect")
    assert torch.equal(ya, yb)
    go = torch.randn(N, C, H + 2, W + 2, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    ya.backward(go)
    yb.backward(go)
    tol = 1e-6 if dtype == torch.float32 else 2e-2     # bf16: one rounding here, one per atomic add in ATen
    assert float((xa.grad.float() - xb.grad.float()).abs().max()) <= tol * max(1.0, float(xb.grad.float().abs().max()))


def test_narrow_convs_on_wide_images_skip_the_miopen_search():
    """ROCm 7.2: MIOpen's tuning search for the 16-channel 3x3 backward-data problem at 322x1026 (iconv1 of the image
    decoders at the 320x1024 configuration) faults (profiles/r02/miopen_search_fault_c4.log); Conv3x3 routes such layers
    through an immediate-mode convolution.  Checked here: the guard engages for W > 1024 and the result equals the plain
    convolution's (forward and gradients)."""
    import tripled_amd  # noqa: F401
    from mono.model import networks
    torch.manual_seed(0)
    conv = networks.Conv3x3(16, 16).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(1, 16, 6, 1030).cuda().contiguous(memory_format=torch.channels_last)
    calls = []
    real = networks._ConvImmediate.apply
    networks._ConvImmediate.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
    prev = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = True
    try:
        xa = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(xa)
        y.float().sum().backward()
        ga, gw = xa.grad.clone(), conv.conv.weight.grad.clone()
    finally:
        torch.backends.cudnn.benchmark = prev
        networks._ConvImmediate.apply = real
    assert calls, "the guard did not engage"
    conv.conv.weight.grad = None
    xb = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yb = conv(xb)            # benchmark off: the plain path
    yb.float().sum().backward()
    assert float((y.float() - yb.float()).abs().max()) <= 2e-2 * float(yb.float().abs().max())
    assert float((ga.float() - xb.grad.float()).abs().max()) <= 2e-2 * float(xb.grad.float().abs().max())
    assert float((gw - conv.conv.weight.grad).abs().max()) <= 2e-2 * float(conv.conv.weight.grad.abs().max())
