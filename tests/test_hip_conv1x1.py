"""The hand-written MFMA 1x1-convolution GEMM with BatchNorm statistics in its epilogue (csrc/td_conv1x1.hip,
td_conv1x1_fwd + td_bn_fwd_from_partials) against plain fp32 torch: F.conv2d + F.batch_norm(training=True) [+ residual]
[-> relu] (reference: mono/model/mono_fm_joint/resnet.py:66-86, 119-127), at the ResNet50 shapes of cfg_kitti_tripleD
(M = 12 x {48x160, 24x80, 12x40, 6x20} output pixels, K / N in {64 .. 2048}), the stride-2 down-sample branches, stacked passes
(groups = 3) and a ragged pixel count.

Tolerances (stated): the GEMM output is the fp32 convolution of the SAME bf16 operands rounded once to bf16: |delta| <= 2^-8 |y| +
1e-3 * max|y| (fp32 accumulation order differs from ATen's); the partial statistics summed over the row tiles equal the
sums over the stored bf16 tensor to 1e-5 relative (same values, different order); the normalised output within 2 bf16 ulp of
its magnitude (0.02 + 2^-7 |y|); gradients within 2-3 % of each tensor's maximum against fp32 autograd of the composite
(whose convolution output is rounded to bf16 like the stored tensor, so that both ReLU masks act on the same values).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [
    # (batch, Hi, Wi, K, N, stride, groups)
    (12, 48, 160, 64, 64, 1, 1),        # layer1.0.conv1
    (12, 48, 160, 64, 256, 1, 1),       # layer1.x.conv3 / layer1.0.downsample (128x128 tiles)
    (12, 48, 160, 256, 64, 1, 1),       # layer1.1.conv1
    (12, 48, 160, 256, 512, 2, 1),      # layer2.0.downsample (stride 2)
    (12, 24, 80, 128, 512, 1, 1),       # layer2.x.conv3
    (12, 12, 40, 1024, 256, 1, 1),      # layer3.x.conv1
    (12, 12, 40, 256, 1024, 1, 1),      # layer3.x.conv3
    (12, 12, 40, 1024, 2048, 2, 1),     # layer4.0.downsample
    (12, 6, 20, 2048, 512, 1, 1),       # layer4.x.conv1 (K = 2048: 32 K-steps)
    (12, 6, 20, 512, 2048, 1, 1),       # layer4.x.conv3 (64x64 tiles, 1440 = 22.5 tiles of 64 rows: ragged)
    (36, 24, 80, 128, 512, 1, 3),       # the auto-encoder's three stacked frame passes
    (6, 7, 11, 64, 128, 1, 2),          # odd pixel counts: 231 rows per group
]


def _data(B, Hi, Wi, K, N, stride, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, K, Hi, Wi, generator=g) * 0.7 + 0.1).to(torch.bfloat16)
    w = (torch.randn(N, K, 1, 1, generator=g) / K ** 0.5).to(torch.bfloat16)
    gamma = 0.5 + torch.rand(N, generator=g)
    beta = 0.2 * torch.randn(N, generator=g)
    return x, w, gamma, beta


@pytest.mark.parametrize("B,Hi,Wi,K,N,stride,groups", SHAPES)
def test_gemm_and_epilogue_statistics(B, Hi, Wi, K, N, stride, groups):
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    lib = native.load()
    x, w, _, _ = _data(B, Hi, Wi, K, N, stride)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda()
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    M = B * Ho * Wo
    y = torch.empty(B, N, Ho, Wo, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    S = lib.td_conv1x1_stat_rows(M, groups, N)
    assert S >= 1
    part = torch.full((groups, S, N, 2), float("nan"), device="cuda")
    from tripled_amd.ops import _raw
    native.check(lib.td_conv1x1_fwd(_raw(xd), _raw(wd), M, groups, K, N, Hi, Wi, stride, _raw(y), native.ptr(part),
                                    native.stream()), "td_conv1x1_fwd")
    torch.cuda.synchronize()
    ref = F.conv2d(x.float(), w.float(), stride=stride)
    got = y.float().cpu()
    tol = 2.0 ** -8 * ref.abs() + 1e-3 * float(ref.abs().max())
    assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() - tol).max())
    # epilogue statistics == sums over the stored (bf16) tensor, per statistics group
    rows = y.permute(0, 2, 3, 1).reshape(groups, M // groups, N).double()
    sums = part.double().sum(1).cpu()
    assert bool(torch.isfinite(sums).all())
    ref_s, ref_q = rows.sum(1).cpu(), (rows * rows).sum(1).cpu()
    assert torch.allclose(sums[..., 0], ref_s, rtol=1e-5, atol=1e-3 * float(ref_s.abs().max()) + 1e-6)
    assert torch.allclose(sums[..., 1], ref_q, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,Hi,Wi,K,N,stride,groups", SHAPES)
@pytest.mark.parametrize("dw_dtype", [torch.bfloat16, torch.float32])
def test_weight_gradient_kernel(B, Hi, Wi, K, N, stride, groups, dw_dtype):
    """td_conv1x1_wgrad (MFMA over the pixel index through ds_read_b64_tr_b16, ordered slab sum) against the fp64 product of
    the same bf16 operands.  The inputs are ASYMMETRIC on purpose (x carries a per-channel ramp, dy a different one): a
    transposed or permuted operand map cannot pass.  fp32 slab: 2e-4 of the largest element (fp32 accumulation over up to
    276 480 pixels); bf16 output: plus its rounding, 2^-8 relative."""
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    from tripled_amd.ops import _raw
    lib = native.load()
    g = torch.Generator().manual_seed(3)
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    M = B * Ho * Wo
    x = (torch.randn(B, K, Hi, Wi, generator=g) + torch.linspace(-1, 1, K).view(1, K, 1, 1)).to(torch.bfloat16)
    dy = (torch.randn(B, N, Ho, Wo, generator=g) * torch.linspace(0.2, 2.0, N).view(1, N, 1, 1)).to(torch.bfloat16)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    dw = torch.full((N, K, 1, 1), float("nan"), device="cuda", dtype=dw_dtype)
    ws = torch.empty(lib.td_conv1x1_wgrad_workspace_floats(M, K, N), device="cuda")
    native.check(lib.td_conv1x1_wgrad(_raw(dyd), _raw(xd), M, K, N, Hi, Wi, stride, native.DTYPE_CODES[dw_dtype], _raw(dw), native.ptr(ws),
                                      native.stream()), "td_conv1x1_wgrad")
    torch.cuda.synchronize()
    xs = x[:, :, ::stride, ::stride].permute(0, 2, 3, 1).reshape(M, K).double()
    ref = dy.permute(0, 2, 3, 1).reshape(M, N).double().t() @ xs
    got = dw.double().cpu().reshape(N, K)
    assert bool(torch.isfinite(got).all())
    tol = 2e-4 * float(ref.abs().max()) + (2.0 ** -8 * ref.abs() if dw_dtype == torch.bfloat16 else 0.0)
    assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() - tol).max())


@pytest.mark.parametrize("B,Hi,Wi,K,N,stride,groups", [SHAPES[1], SHAPES[3], SHAPES[6], SHAPES[9], SHAPES[10], SHAPES[11]])
@pytest.mark.parametrize("relu,with_res", [(True, False), (True, True), (False, False)])
def test_conv_bn_act_forward_backward_vs_fp32(B, Hi, Wi, K, N, stride, groups, relu, with_res):
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    x, w, gamma, beta = _data(B, Hi, Wi, K, N, stride, seed=1)
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    g = torch.Generator().manual_seed(5)
    res = (torch.randn(B, N, Ho, Wo, generator=g)).to(torch.bfloat16) if with_res else None
    up = torch.randn(B, N, Ho, Wo, generator=g).to(torch.bfloat16)

    # fp32 reference of the composite on the same bf16 operands, per statistics group
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = res.float().requires_grad_(True) if with_res else None
    rm, rv = torch.zeros(N), torch.ones(N)
    outs = []
    for xc in xr.chunk(groups, 0):
        yc = F.conv2d(xc, wr, stride=stride)
        # the convolution output is STORED in bf16 (as under autocast with any convolution kernel) and normalised from there;
        # straight-through rounding keeps the ReLU masks of the two evaluations on the same values
        yc = yc + (yc.to(torch.bfloat16).float() - yc).detach()
        outs.append(F.batch_norm(yc, rm, rv, gr, br, True, 0.1, 1e-5))
    ref = torch.cat(outs, 0)
    if with_res:
        ref = ref + rr
    if relu:
        ref = F.relu(ref)
    (ref * up.float()).sum().backward()

    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    gd, bd = gamma.cuda().requires_grad_(True), beta.cuda().requires_grad_(True)
    rd = res.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) if with_res else None
    rmd, rvd = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    out = ops.conv1x1_bn_act(xd, wd, gd, bd, rmd, rvd, 0.1, 1e-5, residual=rd, relu=relu, groups=groups, stride=stride)
    (out.float() * up.cuda().float()).sum().backward()
    torch.cuda.synchronize()

    d = (out.float().cpu() - ref.detach()).abs()
    assert bool((d <= 0.02 + 2.0 ** -7 * ref.detach().abs()).all()), float(d.max())
    assert torch.allclose(rmd.cpu(), rm, atol=2e-3) and torch.allclose(rvd.cpu(), rv, rtol=2e-2, atol=2e-3)

    def close(a, b, what, rel=2e-2, outliers=0.0):
        """max |a - b| <= rel * max|b|, except a fraction `outliers` of the elements: a pre-activation that the two fp32
        accumulation orders round to different bf16 values across zero flips its ReLU mask (measured: a handful per million)."""
        a, b = a.float().cpu(), b.float()
        err = (a - b).abs() / max(float(b.abs().max()), 1e-12)
        bad = float((err >= rel).float().mean())
        assert bad <= outliers, (what, float(err.max()), bad)
    flips = 1e-4 if relu else 0.0
    close(xd.grad, xr.grad, "dx", 3e-2, flips * 64)       # one flipped channel touches the dx of its whole pixel
    close(wd.grad, wr.grad, "dw", 3e-2)
    close(gd.grad, gr.grad, "dgamma")
    close(bd.grad, br.grad, "dbeta")
    if with_res:
        close(rd.grad, rr.grad, "dresidual", 2e-2, flips)


def test_bottleneck_uses_the_mfma_path_and_matches_miopen():
    """networks.Bottleneck under bf16 autocast: same block through (a) the MFMA GEMM + epilogue statistics and (b) MIOpen +
    td_bn_fwd (TD_NO_MFMA_1X1 equivalent): outputs within 2 bf16 ulp, and the dispatch counters show which ran."""
    import tripled_amd  # noqa: F401
    from mono.model import networks
    from tripled_amd import dispatch
    torch.manual_seed(0)
    ds = torch.nn.Sequential(networks._conv(256, 512, 1, 2), networks.BatchNorm(512))
    blk = networks.Bottleneck(256, 128, stride=2, downsample=ds).cuda().to(memory_format=torch.channels_last).train()
    x = torch.randn(4, 256, 24, 40, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dispatch.reset()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = blk(x)
    # the fused block (round 4): conv1 + down-sample GEMMs with statistics epilogues, conv3 with bn2 + relu in its operand staging
    assert dispatch.hip_calls["td_conv1x1_fwd"] == 2 and dispatch.hip_calls["td_conv1x1_fwd_bnrelu"] == 1 \
        and dispatch.hip_calls["td_bn_fwd_from_partials"] == 3 and dispatch.hip_calls["td_bn_fwd_partials"] == 1
    prev = networks.FUSED_1X1_OFF
    networks.FUSED_1X1_OFF = True
    try:
        dispatch.reset()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            b = blk(x)
        assert dispatch.hip_calls["td_conv1x1_fwd"] == 0 and dispatch.hip_calls["td_bn_fwd"] == 4
    finally:
        networks.FUSED_1X1_OFF = prev
    d = (a.float() - b.float()).abs()
    assert float(d.max()) <= 0.05 and float(d.mean()) < 4e-3, (float(d.max()), float(d.mean()))


@pytest.mark.parametrize("B,Hi,Wi,C,N,stride,pad,groups", [(12, 48, 160, 64, 64, 1, 1, 1), (6, 24, 80, 128, 128, 2, 1, 3),
                                                           (4, 12, 40, 256, 256, 1, 1, 1), (2, 26, 42, 520, 256, 1, 0, 1),
                                                           (3, 7, 11, 64, 128, 1, 1, 1)])
def test_conv3x3_implicit_gemm(B, Hi, Wi, C, N, stride, pad, groups):
    """td_conv3x3_fwd (the 1x1 GEMM's tiles, stages, MFMA block and epilogue with K = 9 * Cin walked tap by tap; zero-filled
    padding and a zero-filled last partial channel chunk for Cin = 520) against fp32 F.conv2d of the same bf16 operands, and its
    epilogue statistics against sums over the stored tensor.  Built and measured this round; NOT dispatched by the model: MIOpen's
    tuned implicit-GEMM kernels are 1.2-1.7x faster on these compute-bound shapes (profiles/r03/conv3x3_bench_v1.json)."""
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    from tripled_amd.ops import _raw
    lib = native.load()
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(B, C, Hi, Wi, generator=g) * 0.7 + 0.1).to(torch.bfloat16)
    w = (torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(torch.bfloat16)
    Ho, Wo = (Hi + 2 * pad - 3) // stride + 1, (Wi + 2 * pad - 3) // stride + 1
    M = B * Ho * Wo
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    y = torch.empty(B, N, Ho, Wo, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    S = lib.td_conv1x1_stat_rows(M, groups, N)
    part = torch.full((groups, S, N, 2), float("nan"), device="cuda")
    native.check(lib.td_conv3x3_fwd(_raw(xd), _raw(wd), B, groups, Hi, Wi, C, N, stride, pad, _raw(y), native.ptr(part), native.stream()),
                 "td_conv3x3_fwd")
    torch.cuda.synchronize()
    ref = F.conv2d(x.float(), w.float(), stride=stride, padding=pad)
    got = y.float().cpu()
    tol = 2.0 ** -8 * ref.abs() + 1e-3 * float(ref.abs().max())
    assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() - tol).max())
    rows = y.permute(0, 2, 3, 1).reshape(groups, M // groups, N).double()
    sums = part.double().sum(1).cpu()
    ref_s, ref_q = rows.sum(1).cpu(), (rows * rows).sum(1).cpu()
    assert torch.allclose(sums[..., 0], ref_s, rtol=1e-5, atol=1e-3 * float(ref_s.abs().max()) + 1e-6)
    assert torch.allclose(sums[..., 1], ref_q, rtol=1e-5, atol=1e-6)


def test_grouped_weight_gradients_equal_the_single_launches(monkeypatch):
    """td_conv1x1_wgrad_group over ALL shapes of SHAPES at once (two dtypes: 24 problems, and again 2 x 24 = 48 > the 40 of one
    launch) against td_conv1x1_wgrad one by one: the grouped form keeps every problem's tiling, row ranges and summation order,
    so the results are EQUAL bit for bit; and ops.deferred_wgrads() routes an autograd backward through it."""
    import tripled_amd  # noqa: F401
    from tripled_amd import native, ops
    lib = native.load()
    probs = []
    for rep in range(2):
        for (B, Hi, Wi, K, N, stride, groups) in SHAPES:
            for dt in (torch.bfloat16, torch.float32):
                g = torch.Generator().manual_seed(len(probs))
                Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
                x = torch.randn(B, K, Hi, Wi, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
                dy = torch.randn(B, N, Ho, Wo, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
                w = torch.empty(N, K, 1, 1, device="cuda", dtype=dt)
                probs.append((dy, x, w, B * Ho * Wo, K, N, Hi, Wi, stride))
    single = [ops.conv1x1_wgrad(*p) for p in probs]
    with ops.wgrad_group() as grp:                                         # "node" scope: tensors back at once, written at the exit
        node = [ops.conv1x1_wgrad(*p) for p in probs]
        assert len(grp.items) == (48 if ops.WGRAD_GROUP == "node" else 0)
    torch.cuda.synchronize()
    for a, b in zip(single, node):
        assert torch.equal(a, b)
    assert all(bool(torch.isfinite(a.float()).all()) for a in single)
    for p in probs:
        p[2].requires_grad_(True)
    monkeypatch.setattr(ops, "WGRAD_GROUP", "step")
    with ops.deferred_wgrads():                                            # "step" scope: autograd gets no gradient ...
        with ops.wgrad_group() as inner:                                   # (a node scope inside leaves the step's queue in place)
            assert all(ops.conv1x1_wgrad(*p) is None for p in probs)
        assert not inner.items and all(p[2].grad is None for p in probs)
    torch.cuda.synchronize()
    for a, p in zip(single, probs):                                        # ... the exit of the block stores it
        assert torch.equal(a, p[2].grad)
    with ops.deferred_wgrads():                                            # a weight used twice, and one with a gradient already
        ops.conv1x1_wgrad(*probs[0]), ops.conv1x1_wgrad(*probs[0])
    torch.cuda.synchronize()
    assert torch.equal(probs[0][2].grad, (single[0] + single[0]) + single[0])
    monkeypatch.setattr(ops, "WGRAD_GROUP", "off")
    with ops.wgrad_group() as grp:
        off = ops.conv1x1_wgrad(*probs[1])
    assert not grp.items and torch.equal(off, single[1])
