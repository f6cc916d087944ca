"""GPU parity: hand-written 5x5 max-pool (channels_last) vs torch's max_pool2d, forward and backward."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H,W", [(2, 16, 7, 9), (1, 256, 12, 40), (3, 8, 1, 5), (2, 64, 24, 80),
                                     (1, 128, 37, 53), (3, 64, 9, 200),                  # scatter backward: ragged tiles / strips
                                     (12, 256, 48, 160), (4, 256, 80, 256)])             # C2 / C4 stage-1 CRP block
def test_maxpool5_matches_aten(dtype, N, C, H, W):
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(0)
    # quantised values create many exact ties: the arg-max tie-break must match ATen's
    x = (torch.randint(0, 6, (N, C, H, W), generator=g).float() * 0.25).to(dtype)
    x = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    y = ops.maxpool5(x)
    yr = F.max_pool2d(xr, 5, 1, 2)
    assert torch.equal(y, yr)                      # bit-exact selection
    go = torch.randn(N, C, H, W, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    y.backward(go)
    yr.backward(go)
    tol = 1e-5 if dtype == torch.float32 else 3e-2   # f32: summation order; bf16: f32 accumulation here, one rounding
    assert float((x.grad.float() - xr.grad.float()).abs().max()) <= tol * max(1.0, float(xr.grad.float().abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H,W", [(2, 16, 7, 9), (1, 64, 12, 40), (3, 8, 1, 5), (2, 64, 96, 320), (1, 8, 2, 2)])
def test_maxpool3s2_matches_aten(dtype, N, C, H, W):
    """ResNet stem pool MaxPool2d(3, 2, 1) (reference resnet.py:101): odd and even sizes, ties."""
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(1)
    x = (torch.randint(0, 6, (N, C, H, W), generator=g).float() * 0.25).to(dtype)
    x = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    y = ops.maxpool3s2(x)
    yr = F.max_pool2d(xr, 3, 2, 1)
    assert y.shape == yr.shape and torch.equal(y, yr)
    go = torch.randn(yr.shape, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    y.backward(go)
    yr.backward(go)
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    assert float((x.grad.float() - xr.grad.float()).abs().max()) <= tol * max(1.0, float(xr.grad.float().abs().max()))


@pytest.mark.parametrize("N,C,H,W", [(2, 16, 7, 9), (2, 64, 24, 80)])      # gather backward / scatter backward
def test_maxpool5_nan_and_inf_follow_aten(N, C, H, W):
    """NaN wins a window (the LAST one in row-major order takes the gradient, like ATen's scan); scattered -inf / +inf values
    (no window is entirely -inf: ATen's index for that case depends on its memory-format kernel)."""
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, H, W, generator=g)
    m = torch.rand(N, C, H, W, generator=g)
    x[m < 0.02] = float("nan")
    x[(m > 0.02) & (m < 0.10)] = float("-inf")
    x[(m > 0.10) & (m < 0.12)] = float("inf")
    x = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    y, yr = ops.maxpool5(x), F.max_pool2d(xr, 5, 1, 2)
    assert torch.equal(torch.isnan(y), torch.isnan(yr)) and torch.equal(torch.nan_to_num(y, nan=7.0), torch.nan_to_num(yr, nan=7.0))
    go = torch.randn(N, C, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    y.backward(go)
    yr.backward(go)
    assert float((x.grad - xr.grad).abs().max()) <= 1e-5 * max(1.0, float(xr.grad.abs().max()))
