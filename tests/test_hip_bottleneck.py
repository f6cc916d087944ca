"""The fused bottleneck (round 4): BatchNorm work folded into the 1x1 GEMMs' operand staging and epilogues, both 1x1 data
gradients on the hand-written MFMA kernel (csrc/td_conv1x1.hip "fused forms", tripled_amd.ops.bottleneck).
Reference: Bottleneck.forward, mono/model/mono_fm_joint/resnet.py:66-86 and its autograd.

Every fused entry point is checked twice: against the UNFUSED hand-written path it replaces (same bf16 rounding points, so the
two agree to the effect of the statistics' summation order: stated per test) and against plain fp32 torch
(F.conv2d -> F.batch_norm(training=True) [+ residual] -> relu, forward and backward) at the ResNet50 shapes of cfg_kitti_tripleD.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (batch, H, W, Cout (reduction of the data gradient), Cin (its output width), groups)
DGRAD_SHAPES = [
    (12, 48, 160, 64, 256, 1),      # layer1.x.conv1: dx is 256 wide (128x128 tiles)
    (12, 48, 160, 256, 64, 1),      # layer1.x.conv3: 64-wide output, K = 256
    (12, 24, 80, 512, 128, 1),      # layer2.x.conv3
    (12, 12, 40, 256, 1024, 1),     # layer3.x.conv1
    (12, 6, 20, 2048, 512, 1),      # layer4.x.conv3 (32 K steps, ragged 64-row tiles)
    (12, 6, 20, 512, 2048, 1),      # layer4.x.conv1
    (36, 24, 80, 128, 512, 3),      # three stacked frame passes
    (6, 7, 11, 64, 128, 2),         # 231 rows per group
]


def _cl(t):
    return t.cuda().contiguous(memory_format=torch.channels_last)


def _rows(t):          # [B, C, H, W] channels-last tensor -> [M, C] view of its memory
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])


def _lib():
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    from tripled_amd.ops import _raw
    return native.load(), native, _raw


@pytest.mark.parametrize("B,H,W,Cout,Cin,groups", DGRAD_SHAPES)
@pytest.mark.parametrize("with_res", [False, True])
def test_data_gradient_kernel(B, H, W, Cout, Cin, groups, with_res):
    """td_conv1x1_dgrad: dx = dy . w with the weight read as the forward stores it (transposed LDS reads), against the fp32
    product of the same bf16 operands: |delta| <= 2^-8 |dx| + 1e-3 max|dx| (one rounding); with the residual the result is
    bf16(bf16(dy . w) + res), i.e. a second rounding: 2^-7 |.| + 1e-3 max."""
    lib, native, _raw = _lib()
    g = torch.Generator().manual_seed(1)
    dy = (torch.randn(B, Cout, H, W, generator=g) * (1 + torch.arange(Cout).reshape(1, -1, 1, 1) % 5)).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cout ** 0.5).to(torch.bfloat16)
    res = torch.randn(B, Cin, H, W, generator=g).to(torch.bfloat16) if with_res else None
    dyd, wd = _cl(dy), w.cuda()
    resd = _cl(res) if with_res else None
    M = B * H * W
    dx = torch.full((B, Cin, H, W), float("nan"), device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    native.check(lib.td_conv1x1_dgrad(_raw(dyd), _raw(wd), M, groups, Cout, Cin, _raw(resd) if with_res else None, _raw(dx),
                                      native.stream()), "td_conv1x1_dgrad")
    torch.cuda.synchronize()
    ref = _rows(dyd).float() @ wd.reshape(Cout, Cin).float()
    scale = float(ref.abs().max())
    if with_res:
        ref = ref + _rows(resd).float()
    got = _rows(dx).float()
    tol = (2.0 ** -7 if with_res else 2.0 ** -8) * ref.abs() + 1e-3 * scale + (2.0 ** -8 * scale if with_res else 0.0)
    assert bool(torch.isfinite(got).all())
    assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() - tol).max())


def _bn_forward(lib, native, _raw, z, gamma, beta, groups, relu=True):
    """the unfused hand-written BatchNorm forward: (y, save_mean, save_invstd, running_mean, running_var)"""
    B, C, H, W = z.shape
    M = B * H * W
    y = torch.empty_like(z, memory_format=torch.channels_last)
    mean, invstd = torch.empty(groups * C, device="cuda"), torch.empty(groups * C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    ws = torch.empty(lib.td_bn_workspace_floats(M, groups, C), device="cuda")
    native.check(lib.td_bn_fwd(_raw(z), None, 1, native.ptr(gamma), native.ptr(beta), native.ptr(rm), native.ptr(rv), 0.1, 1e-5,
                               int(relu), M, groups, C, _raw(y), native.ptr(mean), native.ptr(invstd), native.ptr(ws),
                               native.stream()), "td_bn_fwd")
    return y, mean, invstd, rm, rv


def _ulp_mismatch(a, b):
    """fraction of elements that differ, and the largest difference in units of the element's bf16 spacing"""
    a, b = a.float(), b.float()
    d = (a - b).abs()
    spacing = torch.clamp(b.abs(), min=1e-30) * 2.0 ** -7
    return float((d > 0).float().mean()), float((d / torch.clamp(spacing, min=2.0 ** -20)).max())


@pytest.mark.parametrize("B,H,W,K,N,groups", [(12, 48, 160, 64, 256, 1), (12, 24, 80, 128, 512, 1), (12, 12, 40, 256, 1024, 1),
                                             (12, 6, 20, 512, 2048, 1), (36, 24, 80, 128, 512, 3), (6, 7, 11, 64, 128, 2)])
def test_forward_gemm_with_batchnorm_relu_prologue(B, H, W, K, N, groups):
    """td_conv1x1_fwd_bnrelu(z) == td_conv1x1_fwd(td_bn_fwd(z) with relu): the staged operand (a_side) equals the unfused apply
    pass except where the two statistics sums (different summation order, ~1e-7 relative) round an element to the neighbouring
    bf16 value (<= 1e-3 of the elements, never more than one spacing); statistics to 1e-5 relative; outputs within
    2^-7 |y| + 2e-3 max|y|; epilogue statistics == sums over the stored output."""
    lib, native, _raw = _lib()
    g = torch.Generator().manual_seed(3)
    z = _cl((torch.randn(B, K, H, W, generator=g) * (0.5 + torch.rand(1, K, 1, 1, generator=g)) + 0.3 * torch.randn(1, K, 1, 1, generator=g)
             ).to(torch.bfloat16))
    w = (torch.randn(N, K, 1, 1, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    gamma, beta = (0.5 + torch.rand(K, generator=g)).cuda(), (0.2 * torch.randn(K, generator=g)).cuda()
    M = B * H * W
    a_ref, mean_ref, invstd_ref, rm_ref, rv_ref = _bn_forward(lib, native, _raw, z, gamma, beta, groups)
    y_ref = torch.empty(B, N, H, W, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    native.check(lib.td_conv1x1_fwd(_raw(a_ref), _raw(w), M, groups, K, N, H, W, 1, _raw(y_ref), None, native.stream()), "td_conv1x1_fwd")

    R = lib.td_bn_partial_rows(M, groups, K)
    assert R >= 1
    part = torch.full((groups * R * K * 2,), float("nan"), device="cuda")
    native.check(lib.td_bn_fwd_partials(_raw(z), 1, M, groups, K, native.ptr(part), native.stream()), "td_bn_fwd_partials")
    y = torch.full_like(y_ref, float("nan"))
    a_side = torch.full_like(z, float("nan"))
    mean, invstd = torch.empty(groups * K, device="cuda"), torch.empty(groups * K, device="cuda")
    rm, rv = torch.zeros(K, device="cuda"), torch.ones(K, device="cuda")
    S = lib.td_conv1x1_stat_rows(M, groups, N)
    stat = torch.full((groups, S, N, 2), float("nan"), device="cuda")
    native.check(lib.td_conv1x1_fwd_bnrelu(_raw(z), _raw(w), M, groups, K, N, native.ptr(part), R, native.ptr(gamma), native.ptr(beta),
                                           native.ptr(rm), native.ptr(rv), 0.1, 1e-5, native.ptr(mean), native.ptr(invstd),
                                           _raw(a_side), _raw(y), native.ptr(stat), native.stream()), "td_conv1x1_fwd_bnrelu")
    torch.cuda.synchronize()
    assert torch.allclose(mean, mean_ref, rtol=1e-5, atol=1e-6) and torch.allclose(invstd, invstd_ref, rtol=1e-5)
    assert torch.allclose(rm, rm_ref, rtol=1e-5, atol=1e-7) and torch.allclose(rv, rv_ref, rtol=1e-5)
    frac, worst = _ulp_mismatch(a_side, a_ref)
    assert frac <= 1e-3 and worst <= 1.01, (frac, worst)
    d = (y.float() - y_ref.float()).abs()
    tol = 2.0 ** -7 * y_ref.float().abs() + 2e-3 * float(y_ref.float().abs().max())
    assert bool(torch.isfinite(y.float()).all()) and bool((d <= tol).all()), float((d - tol).max())
    rows = _rows(y).reshape(groups, M // groups, N).double()
    sums = stat.double().sum(1)
    assert torch.allclose(sums[..., 0], rows.sum(1), rtol=1e-5, atol=1e-3 * float(rows.sum(1).abs().max()) + 1e-6)
    assert torch.allclose(sums[..., 1], (rows * rows).sum(1), rtol=1e-5, atol=1e-6)
    # against fp32 torch: conv(relu(batch_norm(z)))
    zf = z.float()
    af = torch.cat([F.relu(F.batch_norm(c, None, None, gamma, beta, True, 0.1, 1e-5)) for c in zf.chunk(groups, 0)], 0)
    yf = F.conv2d(af, w.float())
    d = (y.float() - yf).abs()
    assert bool((d <= 0.03 + 2.0 ** -6 * yf.abs()).all()), float(d.max())


@pytest.mark.parametrize("B,H,W,Cout,Cin,groups", [s for s in DGRAD_SHAPES if s[3] <= 512])
def test_data_gradient_with_backward_sums_epilogue(B, H, W, Cout, Cin, groups):
    """td_conv1x1_dgrad_bnsums: dx bit-equal to td_conv1x1_dgrad's; out_partials summed over the row tiles == the statistics pass
    td_bn_bwd_partials makes over the stored dx (same masked values, same fmaf mask; different order: 1e-5 relative to sum |g|)."""
    lib, native, _raw = _lib()
    g = torch.Generator().manual_seed(5)
    dy = _cl(torch.randn(B, Cout, H, W, generator=g).to(torch.bfloat16))
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cout ** 0.5).to(torch.bfloat16).cuda()
    z = _cl((torch.randn(B, Cin, H, W, generator=g) + 0.2).to(torch.bfloat16))
    gamma, beta = (0.5 + torch.rand(Cin, generator=g)).cuda(), (0.3 * torch.randn(Cin, generator=g)).cuda()
    M = B * H * W
    _, mean, invstd, _, _ = _bn_forward(lib, native, _raw, z, gamma, beta, groups)
    dx0 = torch.empty_like(z, memory_format=torch.channels_last)
    native.check(lib.td_conv1x1_dgrad(_raw(dy), _raw(w), M, groups, Cout, Cin, None, _raw(dx0), native.stream()), "td_conv1x1_dgrad")
    dx = torch.full_like(z, float("nan"))
    S = lib.td_conv1x1_stat_rows(M, groups, Cin)
    out = torch.full((groups, S, Cin, 2), float("nan"), device="cuda")
    native.check(lib.td_conv1x1_dgrad_bnsums(_raw(dy), _raw(w), M, groups, Cout, Cin, _raw(z), native.ptr(gamma), native.ptr(beta),
                                             native.ptr(mean), native.ptr(invstd), _raw(dx), native.ptr(out), native.stream()),
                 "td_conv1x1_dgrad_bnsums")
    R = lib.td_bn_partial_rows(M, groups, Cin)
    ref = torch.full((groups, R, Cin, 2), float("nan"), device="cuda")
    native.check(lib.td_bn_bwd_partials(_raw(dx0), _raw(z), None, 1, native.ptr(gamma), native.ptr(beta), native.ptr(mean),
                                        native.ptr(invstd), 1, M, groups, Cin, native.ptr(ref), native.stream()), "td_bn_bwd_partials")
    torch.cuda.synchronize()
    assert torch.equal(dx, dx0)
    a, b = out.double().sum(1), ref.double().sum(1)
    scale = float(_rows(dx0).float().abs().sum(0).max())
    assert bool(torch.isfinite(a).all()) and float((a - b).abs().max()) <= 1e-5 * scale, (float((a - b).abs().max()), scale)
    # td_bn_bwd_from_partials on those sums == td_bn_bwd (its own statistics pass): dx, dgamma, dbeta
    dz_ref, dz = torch.empty_like(z), torch.full_like(z, float("nan"))
    dg_ref, db_ref, dg, db = (torch.empty(Cin, device="cuda") for _ in range(4))
    ws = torch.empty(lib.td_bn_workspace_floats(M, groups, Cin), device="cuda")
    native.check(lib.td_bn_bwd(_raw(dx0), _raw(z), None, 1, native.ptr(gamma), native.ptr(beta), native.ptr(mean), native.ptr(invstd), 1, M,
                               groups, Cin, _raw(dz_ref), None, native.ptr(dg_ref), native.ptr(db_ref), native.ptr(ws), native.stream()), "td_bn_bwd")
    native.check(lib.td_bn_bwd_from_partials(_raw(dx), _raw(z), None, 1, native.ptr(gamma), native.ptr(beta), native.ptr(mean),
                                             native.ptr(invstd), 1, M, groups, Cin, native.ptr(out), S, _raw(dz), None, native.ptr(dg),
                                             native.ptr(db), native.stream()), "td_bn_bwd_from_partials")
    torch.cuda.synchronize()
    assert torch.allclose(dg, dg_ref, rtol=1e-4, atol=1e-4 * float(dg_ref.abs().max()))
    assert torch.allclose(db, db_ref, rtol=1e-4, atol=1e-4 * float(db_ref.abs().max()))
    frac, worst = _ulp_mismatch(dz, dz_ref)
    assert frac <= 2e-3 and worst <= 1.01, (frac, worst)


@pytest.mark.parametrize("B,H,W,Cout,Cin,groups", [s for s in DGRAD_SHAPES if s[3] <= 512])
@pytest.mark.parametrize("with_res", [False, True])
def test_data_gradient_with_batchnorm_backward_prologue(B, H, W, Cout, Cin, groups, with_res):
    """td_conv1x1_dgrad_bnbwd(g, z) == td_conv1x1_dgrad(td_bn_bwd(g, z)) [+ residual]: the staged operand (dz_side) against the
    unfused dx pass (coefficients from a differently ordered sum: <= 2e-3 of the elements one bf16 spacing apart), dgamma / dbeta
    to 1e-4, dx within 2^-7 |dx| + 3e-3 max|dx|; and against fp32 autograd of relu(batch_norm(z)) . w."""
    lib, native, _raw = _lib()
    gen = torch.Generator().manual_seed(7)
    z = _cl((torch.randn(B, Cout, H, W, generator=gen) * (0.5 + torch.rand(1, Cout, 1, 1, generator=gen)) + 0.1).to(torch.bfloat16))
    g = _cl(torch.randn(B, Cout, H, W, generator=gen).to(torch.bfloat16))
    w = (torch.randn(Cout, Cin, 1, 1, generator=gen) / Cout ** 0.5).to(torch.bfloat16).cuda()
    res = _cl(torch.randn(B, Cin, H, W, generator=gen).to(torch.bfloat16)) if with_res else None
    gamma, beta = (0.5 + torch.rand(Cout, generator=gen)).cuda(), (0.3 * torch.randn(Cout, generator=gen)).cuda()
    M = B * H * W
    _, mean, invstd, _, _ = _bn_forward(lib, native, _raw, z, gamma, beta, groups)
    # unfused
    dz_ref = torch.empty_like(z)
    dg_ref, db_ref, dg, db = (torch.full((Cout,), float("nan"), device="cuda") for _ in range(4))
    ws = torch.empty(lib.td_bn_workspace_floats(M, groups, Cout), device="cuda")
    native.check(lib.td_bn_bwd(_raw(g), _raw(z), None, 1, native.ptr(gamma), native.ptr(beta), native.ptr(mean), native.ptr(invstd), 1, M,
                               groups, Cout, _raw(dz_ref), None, native.ptr(dg_ref), native.ptr(db_ref), native.ptr(ws), native.stream()), "td_bn_bwd")
    dx_ref = torch.empty(B, Cin, H, W, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    native.check(lib.td_conv1x1_dgrad(_raw(dz_ref), _raw(w), M, groups, Cout, Cin, _raw(res) if with_res else None, _raw(dx_ref),
                                      native.stream()), "td_conv1x1_dgrad")
    # fused
    R = lib.td_bn_partial_rows(M, groups, Cout)
    part = torch.full((groups * R * Cout * 2,), float("nan"), device="cuda")
    native.check(lib.td_bn_bwd_partials(_raw(g), _raw(z), None, 1, native.ptr(gamma), native.ptr(beta), native.ptr(mean), native.ptr(invstd),
                                        1, M, groups, Cout, native.ptr(part), native.stream()), "td_bn_bwd_partials")
    dz = torch.full_like(z, float("nan"))
    dx = torch.full_like(dx_ref, float("nan"))
    native.check(lib.td_conv1x1_dgrad_bnbwd(_raw(g), _raw(z), _raw(w), M, groups, Cout, Cin, native.ptr(part), R, native.ptr(gamma),
                                            native.ptr(beta), native.ptr(mean), native.ptr(invstd), native.ptr(dg), native.ptr(db),
                                            _raw(dz), _raw(res) if with_res else None, _raw(dx), native.stream()), "td_conv1x1_dgrad_bnbwd")
    torch.cuda.synchronize()
    assert torch.allclose(dg, dg_ref, rtol=1e-4, atol=1e-4 * float(dg_ref.abs().max()))
    assert torch.allclose(db, db_ref, rtol=1e-4, atol=1e-4 * float(db_ref.abs().max()))
    frac, worst = _ulp_mismatch(dz, dz_ref)
    assert frac <= 2e-3 and worst <= 1.01, (frac, worst)
    d = (dx.float() - dx_ref.float()).abs()
    tol = 2.0 ** -7 * dx_ref.float().abs() + 3e-3 * float(dx_ref.float().abs().max())
    assert bool(torch.isfinite(dx.float()).all()) and bool((d <= tol).all()), float((d - tol).max())
    # fp32 autograd of the same composite
    zf = z.float().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = torch.cat([F.relu(F.batch_norm(c, None, None, gm, bt, True, 0.1, 1e-5)) for c in zf.chunk(groups, 0)], 0)
    a.backward(g.float())
    dxf = _rows(zf.grad.contiguous(memory_format=torch.channels_last)) @ w.reshape(Cout, Cin).float()
    if with_res:
        dxf = dxf + _rows(res).float()
    err = (_rows(dx).float() - dxf).abs() / float(dxf.abs().max())
    assert float((err >= 3e-2).float().mean()) <= 1e-3, (float(err.max()), float((err >= 3e-2).float().mean()))
    assert torch.allclose(dg, gm.grad, rtol=2e-2, atol=2e-2 * float(gm.grad.abs().max()))
    assert torch.allclose(db, bt.grad, rtol=2e-2, atol=2e-2 * float(bt.grad.abs().max()))


BLOCKS = [
    # (batch, H, W, inplanes, planes, stride, downsample, groups)
    (12, 48, 160, 64, 64, 1, True, 1),        # layer1.0 (stride-1 down-sample branch)
    (12, 48, 160, 256, 64, 1, False, 1),      # layer1.1
    (12, 48, 160, 256, 128, 2, True, 1),      # layer2.0 (stride 2)
    (12, 12, 40, 1024, 256, 1, False, 1),     # layer3.x
    (12, 6, 20, 2048, 512, 1, False, 1),      # layer4.x
    (6, 24, 80, 512, 128, 1, False, 3),       # stacked passes
]


def _fp32_block(x, p, stride, groups):
    def bn(t, g, b):
        return torch.cat([F.batch_norm(c, None, None, g, b, True, 0.1, 1e-5) for c in t.chunk(groups, 0)], 0)
    a1 = F.relu(bn(F.conv2d(x, p["conv1.weight"]), p["bn1.weight"], p["bn1.bias"]))
    a2 = F.relu(bn(F.conv2d(a1, p["conv2.weight"], stride=stride, padding=1), p["bn2.weight"], p["bn2.bias"]))
    z3 = bn(F.conv2d(a2, p["conv3.weight"]), p["bn3.weight"], p["bn3.bias"])
    sc = x
    if "downsample.0.weight" in p:
        sc = bn(F.conv2d(x, p["downsample.0.weight"], stride=stride), p["downsample.1.weight"], p["downsample.1.bias"])
    return F.relu(z3 + sc)


@pytest.mark.parametrize("B,H,W,inplanes,planes,stride,down,groups", BLOCKS)
def test_fused_block_against_per_layer_path_and_fp32(B, H, W, inplanes, planes, stride, down, groups):
    """networks.Bottleneck under bf16 autocast: the fused node (tripled_amd.ops.bottleneck) against (a) the per-layer nodes of
    round 3 (same rounding points: outputs within 2 bf16 spacings, gradients within 2 % of each tensor's maximum up to a
    1e-3 fraction of ReLU-mask flips, running statistics 1e-5) and (b) the fp32 composite (outputs 0.06 + 2^-5 |y|, dx within 6 % of the
    maximum up to 1 % of the elements, parameter gradients 12 % relative L2)."""
    import tripled_amd  # noqa: F401
    from mono.model import networks
    from tripled_amd import dispatch
    torch.manual_seed(3)
    ds = torch.nn.Sequential(networks._conv(inplanes, planes * 4, 1, stride), networks.BatchNorm(planes * 4)) if down else None
    blk = networks.Bottleneck(inplanes, planes, stride=stride, downsample=ds).cuda().to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, networks.BatchNorm):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    x0 = (torch.randn(B, inplanes, H, W, device="cuda") * 0.8).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    up = torch.randn(B, planes * 4, (H - 1) // stride + 1, (W - 1) // stride + 1, device="cuda").to(torch.bfloat16) \
        .contiguous(memory_format=torch.channels_last)
    state0 = {k: v.clone() for k, v in blk.state_dict().items()}

    def run(fused):
        blk.load_state_dict(state0)
        blk.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        prev = networks.FUSED_BLOCK_OFF
        networks.FUSED_BLOCK_OFF = not fused
        try:
            dispatch.reset()
            with torch.autocast("cuda", dtype=torch.bfloat16), networks.bn_groups(groups):
                y = blk(x)
            (y.float() * up.float()).sum().backward()
            calls = dict(dispatch.hip_calls)
        finally:
            networks.FUSED_BLOCK_OFF = prev
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().float().clone() for k, p in blk.named_parameters()}
        bufs = {k: v.detach().clone() for k, v in blk.named_buffers() if "running" in k}
        return y.detach().float(), x.grad.detach().float(), grads, bufs, calls

    y_f, dx_f, g_f, b_f, calls_f = run(True)
    y_u, dx_u, g_u, b_u, calls_u = run(False)
    assert calls_f.get("td_conv1x1_fwd_bnrelu") == 1 and calls_f.get("td_conv1x1_dgrad_bnbwd") == 1 \
        and calls_f.get("td_conv1x1_dgrad_bnsums") == 1 and calls_f.get("td_bn_bwd_from_partials") == 1, calls_f
    assert "td_conv1x1_fwd_bnrelu" not in calls_u and calls_u.get("td_bn_bwd", 0) >= 3, calls_u
    # (launch counts are compared from a kernel trace, profiles/r04: the fused node's data gradients are C-ABI calls where the
    # per-layer nodes call MIOpen, so the call counters here do not measure launches)

    def close(a, b, what, rel, outliers):
        err = (a - b).abs() / max(float(b.abs().max()), 1e-12)
        bad = float((err >= rel).float().mean())
        assert bad <= outliers, (what, float(err.max()), bad)
    d = (y_f - y_u).abs()
    assert bool((d <= 2.0 ** -6 * y_u.abs() + 0.02).all()), float(d.max())
    close(dx_f, dx_u, "dx", 2e-2, 1e-3 * 64)
    for k in g_u:
        close(g_f[k], g_u[k], k, 2e-2, 1e-3)
    for k in b_u:
        assert torch.allclose(b_f[k], b_u[k], rtol=1e-4, atol=1e-5), k

    # fp32 composite on the same (bf16-rounded) weights
    p = {k: v.detach().float().clone().requires_grad_(True) for k, v in blk.named_parameters()}
    for k in p:
        if k.endswith("weight") and p[k].dim() == 4:
            p[k] = p[k].detach().to(torch.bfloat16).float().requires_grad_(True)
    xr = x0.float().clone().requires_grad_(True)
    yr = _fp32_block(xr, p, stride, groups)
    (yr * up.float()).sum().backward()
    d = (y_f - yr.detach()).abs()
    assert bool((d <= 0.06 + 2.0 ** -5 * yr.detach().abs()).all()), float(d.max())
    close(dx_f, xr.grad, "dx vs fp32", 6e-2, 1e-2)

    def l2(a, b, what, rel):       # parameter gradients (sums over 1e4..1e6 bf16-rounded products): relative L2 error
        e = float((a - b).norm()) / max(float(b.norm()), 1e-30)
        assert e <= rel, (what, e)
    for k in g_u:
        l2(g_f[k], p[k].grad, k + " vs fp32", 0.12)


@pytest.mark.parametrize("B,H,W,C", [(12, 6, 20, 256), (4, 24, 80, 256), (2, 48, 160, 256), (3, 7, 11, 64)])
def test_fused_crp_block_against_per_op_path_and_fp32(B, H, W, C):
    """networks.CRPBlock under bf16 autocast: the fused node (tripled_amd.ops.crp_block: pools + GEMMs with the running sum in
    the epilogue + pool backward with the direct gradient added) against the per-op path (MIOpen 1x1 convolutions + tensor adds:
    same rounding points up to the order of the two roundings of a sum; output within 2^-6 |y| + 0.02, gradients within 2 % of
    their maxima up to 1e-3 of the elements: max-pool selections flip where two bf16 values tie differently) and against the
    fp32 composite (reference: layers.py:200-215)."""
    import tripled_amd  # noqa: F401
    from mono.model import networks
    from tripled_amd import dispatch
    torch.manual_seed(5)
    blk = networks.CRPBlock(C, C, 4).cuda().to(memory_format=torch.channels_last).train()
    x0 = torch.randn(B, C, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    up = torch.randn(B, C, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def run(fused):
        blk.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        prev = networks.FUSED_CRP_OFF
        networks.FUSED_CRP_OFF = not fused
        try:
            dispatch.reset()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(x)
            (y.float() * up.float()).sum().backward()
            calls = dict(dispatch.hip_calls)
        finally:
            networks.FUSED_CRP_OFF = prev
        torch.cuda.synchronize()
        return y.detach().float(), x.grad.float(), {k: p.grad.float().clone() for k, p in blk.named_parameters()}, calls

    y_f, dx_f, g_f, calls_f = run(True)
    y_u, dx_u, g_u, calls_u = run(False)
    assert calls_f.get("td_conv1x1_fwd_sum") == 4 and calls_f.get("td_maxpool5_bwd_add") == 4 and calls_f.get("td_conv1x1_dgrad") == 4
    assert "td_conv1x1_fwd_sum" not in calls_u

    def close(a, b, what, rel, outliers):
        err = (a - b).abs() / max(float(b.abs().max()), 1e-12)
        bad = float((err >= rel).float().mean())
        assert bad <= outliers, (what, float(err.max()), bad)
    d = (y_f - y_u).abs()
    assert bool((d <= 2.0 ** -6 * y_u.abs() + 0.02).all()), float(d.max())
    close(dx_f, dx_u, "dx", 2e-2, 2e-3)
    for k in g_u:
        close(g_f[k], g_u[k], k, 2e-2, 2e-3)
    # fp32 composite on the bf16-rounded weights
    ws = [p.detach().to(torch.bfloat16).float().requires_grad_(True) for _, p in blk.named_parameters()]
    xr = x0.float().clone().requires_grad_(True)
    top, out = xr, xr
    for w in ws:
        top = F.conv2d(F.max_pool2d(top, 5, 1, 2), w)
        out = out + top
    (out * up.float()).sum().backward()
    d = (y_f - out.detach()).abs()
    assert bool((d <= 0.05 + 2.0 ** -5 * out.detach().abs()).all()), float(d.max())
    l2 = lambda a, b: float((a - b).norm()) / max(float(b.norm()), 1e-30)
    assert l2(dx_f, xr.grad) <= 0.08, l2(dx_f, xr.grad)
    for (k, _), w in zip(blk.named_parameters(), ws):
        assert l2(g_f[k], w.grad) <= 0.08, (k, l2(g_f[k], w.grad))
