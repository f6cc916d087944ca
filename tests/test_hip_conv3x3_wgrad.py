"""td_conv3x3_wgrad (csrc/td_conv3x3_wgrad.hip): the weight gradient of the 3x3 stride-1 convolutions on the MFMA -- nine taps per
staged tile, bands in padded row coordinates, ordered fp32 slab sum -- against the fp64 weight gradient of the same bf16 operands
(torch autograd of F.conv2d in double), at the ResNet18/50 block shapes and the decoder shapes of cfg_kitti_tripleD.
Reference: mono/model/mono_fm_joint/resnet.py:30-49, 57-58 (conv3x3, padding 1); mono/model/mono_fm_joint/layers.py:171-184
(Conv3x3: ReflectionPad2d(1) + conv, i.e. padding 0 on the padded input).

Tolerance (stated): |dW - ref| <= 2e-4 max|ref| + 2^-8 |ref| for a bf16 result (one rounding), 2e-4 max|ref| for f32
(fp32 accumulation over up to 10^5 products in another order).  Bit-reproducible across launches."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [
    # (B, Ho, Wo, C, N, pad)
    (12, 48, 160, 64, 64, 1),       # layer1 conv2 (ResNet50) / BasicBlock convs (ResNet18)
    (12, 24, 80, 128, 128, 1),      # layer2
    (12, 12, 40, 256, 256, 1),      # layer3
    (12, 6, 20, 512, 512, 1),       # layer4: 20-pixel rows (four row crossings per stage)
    (24, 24, 80, 128, 128, 1),      # pose encoder, two stacked pairs
    (12, 6, 20, 512, 256, 0),       # DepthDecoder.iconv4 on the reflection-padded map (8 x 22 input)
    (12, 24, 80, 256, 256, 0),      # DepthDecoder.merge2
    (12, 12, 40, 256, 128, 0),      # Decoder.upconv4
    (12, 48, 160, 64, 64, 0),       # Decoder.iconv3
    (3, 7, 23, 64, 128, 1),         # odd sizes: 483 pixels, ragged last stage, rows of 23
    (2, 5, 21, 128, 64, 0),
]


def _ref(dy, x, pad):
    xr = x.double().requires_grad_(False)
    w = torch.zeros(dy.shape[1], x.shape[1], 3, 3, dtype=torch.float64, device=x.device, requires_grad=True)
    y = F.conv2d(xr, w, padding=pad)
    (y * dy.double()).sum().backward()
    return w.grad


@pytest.mark.parametrize("B,Ho,Wo,C,N,pad", SHAPES)
@pytest.mark.parametrize("dw_dtype", [torch.bfloat16, torch.float32])
def test_weight_gradient_3x3(B, Ho, Wo, C, N, pad, dw_dtype):
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    from tripled_amd.ops import _raw
    lib = native.load()
    g = torch.Generator().manual_seed(11)
    Hi, Wi = Ho + 2 - 2 * pad, Wo + 2 - 2 * pad
    # asymmetric operands: a per-channel ramp on x, a per-position pattern on dy (a transposed or shifted tap would show)
    x = (torch.randn(B, C, Hi, Wi, generator=g) + 0.05 * torch.arange(C).reshape(1, C, 1, 1) / C).to(torch.bfloat16)
    dy = (torch.randn(B, N, Ho, Wo, generator=g) * (1 + 0.5 * torch.sin(torch.arange(Wo) * 0.7)).reshape(1, 1, 1, Wo)).to(torch.bfloat16)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    ws = torch.empty(lib.td_conv3x3_wgrad_workspace_floats(B, Ho, Wo, C, N), device="cuda")
    assert ws.numel() > 0
    outs = []
    for _ in range(2):
        dw = torch.full((N, C, 3, 3), float("nan"), device="cuda", dtype=dw_dtype).contiguous(memory_format=torch.channels_last)
        native.check(lib.td_conv3x3_wgrad(_raw(dyd), _raw(xd), B, Ho, Wo, C, N, pad, native.DTYPE_CODES[dw_dtype], _raw(dw),
                                          native.ptr(ws), native.stream()), "td_conv3x3_wgrad")
        torch.cuda.synchronize()
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])                      # deterministic
    ref = _ref(dyd, xd, pad)
    got = outs[0].double()
    scale = float(ref.abs().max())
    tol = 2e-4 * scale + (2.0 ** -8 * ref.abs() if dw_dtype == torch.bfloat16 else 0.0)
    err = (got - ref).abs()
    assert bool(torch.isfinite(got).all()) and bool((err <= tol).all()), (float((err - tol).max()), scale)


def test_unsupported_shapes_are_refused():
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    lib = native.load()
    t = torch.zeros(8, device="cuda")
    assert lib.td_conv3x3_wgrad(native.ptr(t), native.ptr(t), 1, 8, 16, 64, 64, 1, 1, native.ptr(t), native.ptr(t), None) == -2   # Wo < 20
    assert lib.td_conv3x3_wgrad(native.ptr(t), native.ptr(t), 1, 8, 32, 520, 64, 0, 1, native.ptr(t), native.ptr(t), None) == -2  # C % 64
    assert lib.td_conv3x3_wgrad_workspace_floats(1, 8, 16, 64, 64) == 0
