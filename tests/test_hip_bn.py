"""HIP BatchNorm(+residual)(+ReLU) vs ATen's F.batch_norm -> add -> relu in fp32 (reference blocks:
mono/model/mono_fm_joint/resnet.py:30-49, 66-86)."""
import pytest
import torch
import torch.nn.functional as F

import tripled_amd  # noqa: F401
from tripled_amd import ops

pytestmark = pytest.mark.gpu


def _reference(x, w, b, rm, rv, res, relu, dy):
    x = x.detach().float().requires_grad_(True)
    w = w.detach().clone().requires_grad_(True)
    b = b.detach().clone().requires_grad_(True)
    res = res.detach().float().requires_grad_(True) if res is not None else None
    y = F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    y.backward(dy.float())
    return y.detach(), x.grad, w.grad, b.grad, (res.grad if res is not None else None)


@pytest.mark.parametrize("shape", [(2, 64, 5, 7), (3, 128, 17, 33), (12, 64, 48, 160), (4, 2048, 6, 20), (1, 64, 1, 3)])
@pytest.mark.parametrize("relu,with_res", [(True, False), (True, True), (False, False), (False, True)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batchnorm_act_matches_aten(shape, relu, with_res, dtype):
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(hash((shape, relu, with_res)) % 1000)
    N, C, H, W = shape
    x = (torch.randn(shape, generator=g) * 1.7 + 0.3).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    res = torch.randn(shape, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last) if with_res else None
    dy = torch.randn(shape, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.rand(C, generator=g) + 0.5).to(dev)
    b = (torch.randn(C, generator=g) * 0.2).to(dev)
    rm0, rv0 = torch.randn(C, generator=g).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)

    rm_ref, rv_ref = rm0.clone(), rv0.clone()
    y_ref, dx_ref, dw_ref, db_ref, dres_ref = _reference(x, w, b, rm_ref, rv_ref, res, relu, dy)

    xh = x.clone().requires_grad_(True)
    wh, bh = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    resh = res.clone().requires_grad_(True) if with_res else None
    rm, rv = rm0.clone(), rv0.clone()
    y = ops.batchnorm_act(xh, wh, bh, rm, rv, 0.1, 1e-5, residual=resh, relu=relu)
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    # fp32: summation order only; bf16: one output rounding (2^-8 relative) on O(1..5) values
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=1.6e-2)
    assert torch.allclose(y.float(), y_ref, **tol)
    assert torch.allclose(rm, rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rv, rv_ref, rtol=1e-5, atol=1e-6)
    M = N * H * W
    gtol = dict(rtol=1e-4, atol=1e-4 * M ** 0.5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2 * M ** 0.5)
    assert torch.allclose(wh.grad, dw_ref, **gtol) and torch.allclose(bh.grad, db_ref, **gtol)
    dtol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    assert torch.allclose(xh.grad.float(), dx_ref, **dtol)
    if with_res:
        assert torch.allclose(resh.grad.float(), dres_ref, **dtol)


def test_resnet_block_uses_fused_bn():
    """The encoders' blocks give the same result through the fused path and through ATen (TD_NO_FUSED_BN)."""
    from mono.model import networks
    torch.manual_seed(0)
    blk = networks.Bottleneck(256, 64).cuda().to(memory_format=torch.channels_last).train()
    x = torch.randn(2, 256, 12, 20, device="cuda").contiguous(memory_format=torch.channels_last)
    ya = blk(x)
    networks.FUSED_BN_OFF = True
    try:
        yb = blk(x)
    finally:
        networks.FUSED_BN_OFF = False
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 8, 5, 7, 1), (3, 256, 256, 12, 20, 1), (1, 8, 24, 3, 4, 5)])
def test_join_channels_matches_cat(shape, dtype):
    """DepthDecoder stage input (depth_decoder.py:89-103): cat + zero padding to a multiple of 8 channels."""
    N, C0, C1, H, W, C2 = shape
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    mk = lambda c: cl(torch.randn(N, c, H, W, device="cuda").to(dtype)).requires_grad_(True)
    a, b, t = mk(C0), mk(C1), mk(C2)
    out = ops.join_channels(a, b, t)
    ref = torch.cat((a, b, t, a.new_zeros(N, 8 - C2, H, W)), 1)
    assert out.shape == ref.shape and torch.equal(out, ref)
    g = cl(torch.randn_like(ref))
    out.backward(g)
    assert torch.equal(a.grad, g[:, :C0]) and torch.equal(b.grad, g[:, C0:C0 + C1])
    assert torch.equal(t.grad, g[:, C0 + C1:C0 + C1 + C2])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu,with_res", [(True, False), (True, True), (False, False)])
@pytest.mark.parametrize("dims", [(3, 4, 128, 9, 13), (2, 2, 64, 96, 160)])     # small (merged finalize) / large tensors
def test_grouped_batchnorm_equals_separate_passes(relu, with_res, dtype, dims):
    """groups=G on a stacked batch == G separate calls (statistics per pass, running stats updated in order,
    weight gradients summed): the pose pairs / source-frame features of mono_fm_joint/net.py:172-178, :221."""
    dev = torch.device("cuda")
    G, N, C, H, W = dims
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    g = torch.Generator().manual_seed(3)
    x = cl((torch.randn(G * N, C, H, W, generator=g) * torch.linspace(0.5, 2.0, G).repeat_interleave(N)[:, None, None, None]).to(dev).to(dtype))
    res = cl(torch.randn(G * N, C, H, W, generator=g).to(dev).to(dtype)) if with_res else None
    dy = cl(torch.randn(G * N, C, H, W, generator=g).to(dev).to(dtype))
    w, b = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)

    def run(groups):
        xs, ws, bs = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        rs = res.clone().requires_grad_(True) if with_res else None
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        if groups > 1:
            y = ops.batchnorm_act(xs, ws, bs, rm, rv, 0.1, 1e-5, residual=rs, relu=relu, groups=groups)
        else:
            y = torch.cat([ops.batchnorm_act(cl(xs[i * N:(i + 1) * N]), ws, bs, rm, rv, 0.1, 1e-5,
                                             residual=cl(rs[i * N:(i + 1) * N]) if with_res else None, relu=relu)
                           for i in range(G)], 0)
        y.backward(dy)
        return y.detach(), xs.grad, ws.grad, bs.grad, (rs.grad if with_res else None), rm, rv

    a, bsep = run(G), run(1)
    assert torch.equal(a[0], bsep[0]) and torch.equal(a[1], bsep[1])          # same kernels, same row ranges
    assert torch.allclose(a[2], bsep[2], rtol=1e-5, atol=1e-4) and torch.allclose(a[3], bsep[3], rtol=1e-5, atol=1e-4)
    if with_res:
        assert torch.equal(a[4], bsep[4])
    assert torch.allclose(a[5], bsep[5], rtol=1e-6, atol=1e-7) and torch.allclose(a[6], bsep[6], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 5, 7), (1, 8, 1, 1), (3, 64, 12, 20), (1, 8, 2, 1)])
def test_up2_reflpad1_matches_aten(shape, dtype):
    """ReflectionPad2d(1)(upsample x2 nearest), forward and adjoint (decoder.py:40-57)."""
    x = torch.randn(shape, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = ops.up2_reflpad1(x)
    xr = x.detach().float().requires_grad_(True)
    ref = F.pad(F.interpolate(xr, scale_factor=2, mode="nearest"), (1, 1, 1, 1), mode="reflect")
    assert out.shape == ref.shape and torch.equal(out.float(), ref)
    g = torch.randn_like(ref).to(dtype).contiguous(memory_format=torch.channels_last)
    out.backward(g)
    ref.backward(g.float())
    tol = 1e-5 if dtype == torch.float32 else 4e-2        # up to 16 bf16 gradients summed in fp32, rounded once
    assert torch.allclose(x.grad.float(), xr.grad, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 8, 6, 10, 1), (3, 256, 256, 12, 20, 1), (12, 256, 256, 24, 80, 1), (1, 8, 24, 2, 4, 5)])
def test_join_channels_up2_matches_cat_of_upsampled(shape, dtype):
    """torch.cat((a, interpolate(b, scale_factor=2, mode="nearest"), tail, zeros), 1) without the up-sampled tensor
    (depth_decoder.py:89-103), forward and the three gradients (the low-resolution one sums four pixels)."""
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    N, C0, C1, H, W, C2 = shape
    g = torch.Generator().manual_seed(sum(shape))
    mk = lambda c, h, w: torch.randn(N, c, h, w, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    a, b, t = mk(C0, H, W), mk(C1, H // 2, W // 2), mk(C2, H, W)
    ins = [v.clone().requires_grad_(True) for v in (a, b, t)]
    ref_in = [v.clone().requires_grad_(True) for v in (a, b, t)]
    out = ops.join_channels_up2(*ins)
    up = torch.nn.functional.interpolate(ref_in[1], scale_factor=2, mode="nearest")
    ref = torch.cat((ref_in[0], up, ref_in[2], torch.zeros(N, 8 - C2, H, W, device="cuda", dtype=dtype)), 1)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out, ref)
    go = torch.randn(ref.shape, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    out.backward(go)
    ref.backward(go)
    assert torch.equal(ins[0].grad, ref_in[0].grad) and torch.equal(ins[2].grad, ref_in[2].grad)
    tol = 1e-6 if dtype == torch.float32 else 2e-2            # bf16: four-term sum in f32 here, pairwise bf16 in ATen
    assert float((ins[1].grad.float() - ref_in[1].grad.float()).abs().max()) <= tol * max(1.0, float(ref_in[1].grad.float().abs().max()))
