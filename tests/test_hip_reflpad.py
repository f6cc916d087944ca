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
