"""Bias + activation fused behind the decoders' convolutions (csrc/td_biasact.hip, tripled_amd.ops.conv_bias_act) against torch:
F.conv2d(x, w, b) -> ELU / leaky ReLU, forward and backward (reference: ConvBlock, mono/model/mono_fm_joint/layers.py:143-155;
depth_decoder.py:89-103).  Tolerances: the activation pass is exact up to the bf16 rounding of its output (1 spacing); the bias
gradient is a deterministic fp32 sum of the bf16 adjoint (1e-5 relative to sum |gy| against a float64 sum)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {None: lambda z: z, "elu": F.elu, "leaky_relu": F.leaky_relu}


@pytest.mark.parametrize("C", [8, 16, 32, 64, 128, 256])
@pytest.mark.parametrize("act", [None, "elu", "leaky_relu"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_bias_act_kernels(C, act, dtype):
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    from tripled_amd.ops import ACT_CODES, _raw
    lib = native.load()
    g = torch.Generator().manual_seed(C)
    M = 4 * 37 * 53                                             # odd row count: ragged tails in every block
    y = (torch.randn(M, C, generator=g) * 1.5).to(dtype).cuda()
    bias = torch.randn(C, generator=g).cuda()
    a = torch.empty_like(y)
    native.check(lib.td_bias_act_fwd(_raw(y), native.ptr(bias), 0, native.DTYPE_CODES[dtype], M, C, ACT_CODES[act], _raw(a),
                                     native.stream()), "td_bias_act_fwd")
    ref = ACTS[act](y.float() + bias)
    tol = 2.0 ** -8 * ref.abs() + 1e-6 if dtype == torch.bfloat16 else 1e-6 * ref.abs() + 1e-6
    assert bool(((a.float() - ref).abs() <= tol).all())
    up = torch.randn(M, C, generator=g).to(dtype).cuda()
    gy = torch.full_like(y, float("nan"))
    outs = []
    for _ in range(2):
        db = torch.full((C,), float("nan"), device="cuda")
        ws = torch.empty(lib.td_bias_act_workspace_floats(M, C), device="cuda")
        native.check(lib.td_bias_act_bwd(_raw(up), _raw(a), native.DTYPE_CODES[dtype], M, C, ACT_CODES[act],
                                         _raw(gy) if act is not None else None, native.ptr(db), 0, native.ptr(ws), native.stream()),
                     "td_bias_act_bwd")
        torch.cuda.synchronize()
        outs.append(db)
    assert torch.equal(outs[0], outs[1])
    af = a.float()
    d = {None: torch.ones_like(af), "elu": torch.where(af > 0, torch.ones_like(af), af + 1),
         "leaky_relu": torch.where(af > 0, torch.ones_like(af), torch.full_like(af, 0.01))}[act]
    gref = up.float() * d
    got = gy.float() if act is not None else up.float()
    tol = 2.0 ** -8 * gref.abs() + 1e-6 if dtype == torch.bfloat16 else 1e-6 * gref.abs() + 1e-7
    assert bool(((got - gref).abs() <= tol).all())
    sref = gref.double().sum(0)                 # the kernel sums the adjoint BEFORE its bf16 rounding
    assert float((outs[0].double() - sref).abs().max()) <= 1e-5 * float(gref.double().abs().sum(0).max())


@pytest.mark.parametrize("B,H,W,Cin,Cout,act", [(4, 26, 82, 128, 64, "elu"), (2, 50, 162, 64, 32, "elu"), (2, 98, 162, 32, 16, "elu"),
                                               (4, 14, 42, 520, 256, "leaky_relu"), (4, 26, 82, 256, 256, "leaky_relu"),
                                               (2, 50, 162, 64, 64, None)])
def test_conv_bias_act_against_torch(B, H, W, Cin, Cout, act):
    """conv_bias_act(x, w, b, act) against (a) the unfused bf16 path it replaces, act(F.conv2d(x, w, b)) through MIOpen + ATen
    (same convolution kernel; the bias / activation roundings differ: output within 2 bf16 spacings, 2^-6 |y| + 1e-3, gradients
    within 1 % of their maxima up to 1e-3 of the elements) and (b) fp32 autograd of the composite on the same bf16 operands
    (output 2^-6 |y| + 1 % of max|y| -- up to 4680 bf16 products per output --, dx / dw 3 %, db 2 %)."""
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(4)
    mk = lambda t: t.to(torch.bfloat16).cuda()
    x = mk(torch.randn(B, Cin, H, W, generator=g)).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = mk(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = mk(0.5 * torch.randn(Cout, generator=g)).requires_grad_(True)
    up = mk(torch.randn(B, Cout, H - 2, W - 2, generator=g)).contiguous(memory_format=torch.channels_last)
    y = ops.conv_bias_act(x, w, b, act)
    (y.float() * up.float()).sum().backward()

    def run(dtype):
        xr, wr, br = (t.detach().to(dtype).requires_grad_(True) for t in (x, w, b))
        yr = ACTS[act](F.conv2d(xr, wr, br))
        (yr.float() * up.float()).sum().backward()
        return yr.detach().float(), xr.grad.float(), wr.grad.float(), br.grad.float()

    def close(a, r, what, rel, outliers=0.0):
        err = (a.float() - r).abs() / float(r.abs().max())
        assert float((err >= rel).float().mean()) <= outliers, (what, float(err.max()))
    torch.cuda.synchronize()
    yb, dxb, dwb, dbb = run(torch.bfloat16)
    d = (y.float() - yb).abs()
    assert bool((d <= 2.0 ** -6 * yb.abs() + 1e-3).all()), float(d.max())
    close(x.grad, dxb, "dx vs bf16 path", 1e-2, 1e-3)
    close(w.grad, dwb, "dw vs bf16 path", 1e-2, 1e-3)
    close(b.grad, dbb, "db vs bf16 path", 1e-2)
    yf, dxf, dwf, dbf = run(torch.float32)
    d = (y.float() - yf).abs()
    assert bool((d <= 2.0 ** -6 * yf.abs() + 1e-2 * float(yf.abs().max())).all()), float(d.max())
    close(x.grad, dxf, "dx", 3e-2, 5e-3)       # sums of up to 2304 bf16-rounded products
    close(w.grad, dwf, "dw", 3e-2, 5e-3)
    # (leaky ReLU: 1-2 % of the pre-activations change sign between the bf16 and the fp32 convolution, each flips its slope
    #  between 1 and 0.01, and the bias gradient is a cancelling sum over all pixels: 5-10 % of its maximum is the precision of
    #  the bf16 forward itself -- the unfused bf16 path above agrees to 1 %)
    close(b.grad, dbf, "db", 2e-2 if act != "leaky_relu" else 0.15)


def test_decoder_blocks_use_the_fused_path():
    import tripled_amd  # noqa: F401
    from mono.model import networks
    from tripled_amd import dispatch
    torch.manual_seed(0)
    blk = networks.ConvBlock(64, 32).cuda().to(memory_format=torch.channels_last).train()
    x = torch.randn(2, 64, 24, 40, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    dispatch.reset()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = blk(x)
        z = blk.forward_up(x)
    (y.float().sum() + z.float().sum()).backward()
    assert dispatch.hip_calls["td_bias_act_fwd"] == 2 and dispatch.hip_calls["td_bias_act_bwd"] == 2
    prev = networks.FUSED_BIAS_ACT_OFF
    networks.FUSED_BIAS_ACT_OFF = True
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y0 = blk(x)
    finally:
        networks.FUSED_BIAS_ACT_OFF = prev
    assert float((y.float() - y0.float()).abs().max()) <= 0.02 + 2.0 ** -7 * float(y0.float().abs().max())
