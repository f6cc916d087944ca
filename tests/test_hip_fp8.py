"""fp8 (OCP e4m3fn) 1x1-convolution path of BASELINE config 5: the hand-written quantisation kernels against torch's own
fp8 cast, the fp8 GEMM against its fp32 emulation on the same quantised operands, and the all-aux-heads model with the
path switched on.  Stated tolerance: e4m3 keeps 3 significand bits (relative step 2^-4 per operand); over a K-term dot
product the output error is ~2^-4 / sqrt(K) of the output scale -- asserted as a relative Frobenius error < 4 % against
the bf16 convolution, and < 1 % (bf16 output rounding) against the emulation."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(4, 64, 6, 10), (12, 256, 48, 160), (1, 16, 1, 8)])
def test_quantize_matches_torch_cast(dtype, shape):
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 3).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    x[0, :7, 0, 0] = 0
    q, inv = ops.quantize_fp8(x)
    amax = x.float().abs().max()
    assert abs(float(inv) - float(amax / 448.0)) <= 1e-7 * float(amax)
    ref = (x.float() * (448.0 / amax)).to(torch.float8_e4m3fn)
    assert q.dtype == torch.float8_e4m3fn and q.stride() == x.stride()
    # same rounding (nearest even) and saturation as torch's cast; the two conversions may differ by one fp8 step on a
    # handful of elements of a 23 M-element f32 tensor (product x * scale within an ulp of a rounding boundary)
    diff = q.view(torch.uint8) != ref.view(torch.uint8)
    if bool(diff.any()):
        a, b = q.float()[diff], ref.float()[diff]
        assert float(diff.float().mean()) < 1e-5, (int(diff.sum()), a[:4].tolist(), b[:4].tolist())
        assert bool(((a - b).abs() <= 0.126 * torch.maximum(a.abs(), b.abs()) + 2.0 ** -9).all()), (a[:4].tolist(), b[:4].tolist())
    back = q.float() * inv
    assert float((back - x.float()).abs().max()) <= float(amax) * 2.0 ** -4


@pytest.mark.parametrize("N,C,H,W,K,bias", [(2, 64, 6, 10, 128, False), (12, 256, 24, 80, 64, False), (3, 512, 3, 5, 256, True)])
def test_conv1x1_fp8_forward_backward(N, C, H, W, K, bias):
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(C + K)
    x = torch.randn(N, C, H, W, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, 1, 1, generator=g) / C ** 0.5).cuda()
    b = torch.randn(K, generator=g).cuda() if bias else None
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bg = b.clone().requires_grad_(True) if bias else None
    y = ops.conv1x1_fp8(xg, wg, bg)
    assert y.shape == (N, K, H, W) and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    # (a) against the fp32 evaluation of the SAME quantised operands: only the bf16 rounding of the output differs
    xq, sx = ops.quantize_fp8(x)
    wq, sw = ops.quantize_fp8(w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    emu = F.conv2d(xq.float(), wq.float()) * (sx * sw)
    if bias:
        emu = emu + b.to(torch.bfloat16).float().view(1, -1, 1, 1)
    assert float((y.float() - emu).norm() / emu.norm()) < 1e-2
    # (b) against the bf16 convolution it replaces: fp8 quantisation noise
    ref = F.conv2d(x, w.to(torch.bfloat16), b.to(torch.bfloat16) if bias else None)
    assert float((y.float() - ref.float()).norm() / ref.float().norm()) < 4e-2
    # (c) the backward is the bf16 convolution's backward on the saved full-precision operands
    up = torch.randn(N, K, H, W, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    y.backward(up)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        F.conv2d(xr, wr, br).backward(up)
    assert float((xg.grad.float() - xr.grad.float()).norm() / xr.grad.float().norm()) < 1e-2
    assert float((wg.grad - wr.grad).norm() / wr.grad.norm()) < 1e-2
    if bias:
        assert float((bg.grad - br.grad).norm() / br.grad.norm()) < 1e-2


def test_all_aux_heads_model_with_fp8_path():
    """cfg_kitti_fm_joint_inpaint_disentangle_distill_full_colorize's class under bf16 autocast + channels_last with the
    1x1 convolutions on the fp8 path: same loss entries as the bf16 path to 5e-4 + 5 %, disparities to a mean of 1e-2,
    finite gradients, no ATen fallback."""
    import tripled_amd  # noqa: F401
    from mono.datasets import synthetic_batch
    from mono.model import MONO, networks
    from tripled_amd import dispatch
    from tests.test_hip_model_step import _opt
    name = "mono_fm_joint_inpaint_disentangle_distill_sep_colorize"
    B, H, W = 2, 96, 160
    torch.manual_seed(11)
    base = MONO.module_dict[name](_opt(name, B, H, W)).cuda().to(memory_format=torch.channels_last)
    base.train()
    base.DepthDecoder.do.eval()
    other = copy.deepcopy(base)
    batch = {k: v.cuda() for k, v in synthetic_batch(B, H, W, seed=3).items()}
    noise = [torch.randn(B, H, W, generator=torch.Generator().manual_seed(50 + i)).cuda() for i in range(8)]
    results = []
    for model, fp8 in ((base, False), (other, True)):
        pool = list(noise)
        model.set_noise_source(lambda shape, device: pool.pop(0))
        prev = networks.set_fp8_conv1x1(fp8)
        dispatch.reset()
        try:
            with dispatch.strict(), torch.autocast("cuda", dtype=torch.bfloat16):
                out, losses = model(dict(batch))
            sum(v.float().mean() for v in losses.values()).backward()
        finally:
            networks.set_fp8_conv1x1(prev)
        results.append((out, losses, dict(dispatch.hip_calls)))
        assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    (o_b, l_b, calls_b), (o_f, l_f, calls_f) = results
    assert calls_f.get("td_fp8_quantize", 0) > 40 and calls_b.get("td_fp8_quantize", 0) == 0
    for k in l_b:
        x, y = float(l_f[k].float().mean()), float(l_b[k].float().mean())
        assert abs(x - y) < 5e-4 + 5e-2 * abs(y), (k, x, y)
    for s in range(4):
        d = (o_f[("disp", 0, s)].float() - o_b[("disp", 0, s)].float()).abs()
        assert float(d.mean()) < 1e-2, (s, float(d.mean()))
