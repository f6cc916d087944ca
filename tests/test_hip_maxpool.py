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


def _ramp(N, C, H, W, kind):
    """inputs whose 5x5 maxima sit at a chosen corner of every window"""
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    rank = yy * W + xx
    base = {"top_left": -rank, "bottom_right": rank, "top_right": -yy * W + xx, "bottom_left": yy * W - xx,
            "ties": torch.zeros(H, W)}[kind]
    # per-channel offsets keep the selection (they are constant over a window) and make channels distinguishable
    return (base / (H * W)).reshape(1, 1, H, W) + 0.01 * torch.arange(C).reshape(1, C, 1, 1) + torch.zeros(N, 1, 1, 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H,W", [(2, 64, 48, 160), (1, 128, 37, 61), (1, 64, 24, 80)])     # all take the LDS scatter (>= 1536 px, C % 64 == 0)
@pytest.mark.parametrize("kind", ["top_left", "bottom_right", "top_right", "bottom_left", "ties"])
@pytest.mark.parametrize("mass", ["all", "first_rows", "overlap_columns", "single_row"])
def test_maxpool5_scatter_backward_adversarial(dtype, N, C, H, W, kind, mass):
    """The LDS scatter backward (csrc/td_maxpool.hip, column turns over a 6-row ring) under selections and gradient masses that
    stress its synchronisation: EVERY output of a tile selects the same corner of its window (all 25 outputs around an input
    element target it; 'top_left' makes the first turn of the first row add into entries other threads own -- the case the
    missing barrier of round 3 raced on), gradient mass only in the first rows of the row strips, only in the columns where
    neighbouring 28-column tiles overlap, or in one row.  Gradients are small integers, so every summation order is exact in
    fp32 and bf16: the result must EQUAL ATen's max_pool2d backward bit for bit.  (A constant input = all ties: ATen's first
    maximum in row-major order.)"""
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    x = _ramp(N, C, H, W, kind)
    if dtype == torch.bfloat16:
        # bf16 cannot hold a 7680-step ramp: quantise rows and columns separately so that the corner stays the unique maximum
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        sy = {"top_left": -1, "bottom_right": 1, "top_right": -1, "bottom_left": 1, "ties": 0}[kind]
        sx = {"top_left": -1, "bottom_right": 1, "top_right": 1, "bottom_left": -1, "ties": 0}[kind]
        x = ((sy * (yy % 8) * 8 + sx * (xx % 8)) * 1.0).reshape(1, 1, H, W) + torch.zeros(N, C, 1, 1)
        # (period-8 ramps: inside a 5-wide window the corner is the maximum unless the window crosses a period boundary;
        #  whatever ATen selects there is the reference -- the comparison is against ATen on the same tensor)
    g = torch.Generator().manual_seed(5)
    up = torch.randint(0, 3, (N, C, H, W), generator=g).float()
    if mass == "first_rows":
        keep = torch.zeros(H)
        keep[0::4] = 1                     # the strips' first rows for every strip height the launch may pick (4 | TH), plus more
        keep[1] = 1
        up = up * keep.reshape(1, 1, H, 1)
    elif mass == "overlap_columns":
        keep = torch.zeros(W)
        for t in range(0, W, 28):          # tiles own 28 columns and read 2 + 2 neighbours
            for d in (-2, -1, 0, 1, 26, 27):
                if 0 <= t + d < W:
                    keep[t + d] = 1
        up = up * keep.reshape(1, 1, 1, W)
    elif mass == "single_row":
        keep = torch.zeros(H)
        keep[H // 2] = 1
        up = up * keep.reshape(1, 1, H, 1)
    xd = x.to(dtype).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xr = x.to(dtype).cuda().requires_grad_(True)
    upd = up.to(dtype).cuda()
    y = ops.maxpool5(xd)
    y.backward(upd.contiguous(memory_format=torch.channels_last))
    yr = torch.nn.functional.max_pool2d(xr, 5, 1, 2)
    yr.backward(upd)
    torch.cuda.synchronize()
    assert torch.equal(y, yr)
    assert torch.equal(xd.grad, xr.grad), float((xd.grad.float() - xr.grad.float()).abs().max())
    assert float(xd.grad.float().sum()) == float(upd.float().sum())      # no gradient lost or duplicated


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H,W", [(2, 16, 7, 9), (2, 64, 24, 80), (1, 128, 37, 53)])      # gather / scatter backward
def test_maxpool5_backward_with_added_gradient(dtype, N, C, H, W):
    """td_maxpool5_bwd_add == td_maxpool5_bwd + add (the CRP block's direct gradient), integer-valued data: exact."""
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    from tripled_amd.ops import _raw
    lib = native.load()
    g = torch.Generator().manual_seed(1)
    cl = lambda t: t.to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    x = cl(torch.randint(0, 6, (N, C, H, W), generator=g).float())
    go = cl(torch.randint(0, 3, (N, C, H, W), generator=g).float())
    add = cl(torch.randint(-3, 4, (N, C, H, W), generator=g).float())
    out, idx = torch.empty_like(x), torch.empty((N, H, W, C), device="cuda", dtype=torch.uint8)
    code = native.DTYPE_CODES[dtype]
    native.check(lib.td_maxpool5_fwd(_raw(x), code, N, H, W, C, _raw(out), _raw(idx), native.stream()), "fwd")
    a, b = torch.empty_like(x), torch.empty_like(x)
    native.check(lib.td_maxpool5_bwd(_raw(go), _raw(idx), code, N, H, W, C, _raw(a), native.stream()), "bwd")
    native.check(lib.td_maxpool5_bwd_add(_raw(go), _raw(idx), _raw(add), code, N, H, W, C, _raw(b), native.stream()), "bwd_add")
    torch.cuda.synchronize()
    assert torch.equal(b.float(), a.float() + add.float())
