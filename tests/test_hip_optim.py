"""td_adam_flat (csrc/td_optim.hip): clip scale + Adam + bf16 working copy in one pass, against torch.nn.utils.clip_grad_norm_ +
torch.optim.Adam (what the reference's optimiser hook runs, mono/core/utils/dist_utils.py:54-60) over several steps.
Tolerance: 2e-6 relative + 2e-7 absolute on parameters and moments per step (fp32, same formula; torch's fused kernel may contract or order
two roundings differently); the bf16 copy equals bf16(w) of the kernel's own w bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("clip", [None, 0.5, 1e6])
@pytest.mark.parametrize("lr_on_device", [False, True])
def test_adam_flat_matches_torch(clip, lr_on_device):
    import tripled_amd  # noqa: F401
    from tripled_amd import native
    lib = native.load()
    torch.manual_seed(0)
    n, n_lp = 1 << 20, 3 << 18
    w0 = torch.randn(n, device="cuda") * 0.1
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.Adam([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=True)
    w, m, v = w0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    lp = torch.empty(n_lp, device="cuda", dtype=torch.bfloat16)
    step = torch.zeros((), device="cuda")
    lr_t = torch.tensor(1e-3, device="cuda")
    for it in range(6):
        g = torch.randn(n, device="cuda") * (0.01 if it % 2 else 1.0)
        p.grad = g.clone()
        total = None
        if clip is not None:
            total = torch.nn.utils.clip_grad_norm_([p], clip)
        opt.step()
        step += 1
        tn = torch.linalg.vector_norm(g) if clip is not None else None
        native.check(lib.td_adam_flat(native.ptr(w), native.ptr(g), native.ptr(m), native.ptr(v), native.ptr(lp), n, n_lp, native.ptr(step),
                                      native.ptr(lr_t) if lr_on_device else None, 0.0 if lr_on_device else 1e-3, 0.9, 0.999, 1e-8,
                                      native.ptr(tn) if tn is not None else None, float(clip) if clip is not None else 0.0,
                                      native.stream()), "td_adam_flat")
        torch.cuda.synchronize()
        st = opt.state[p]
        for a, b, what in ((w, p.detach(), "w"), (m, st["exp_avg"], "m"), (v, st["exp_avg_sq"], "v")):
            # atol: an update is lr * O(1) = 1e-3, its last bits (1e-7 relative: the division / sqrt roundings) are the difference
            assert torch.allclose(a, b, rtol=2e-6 * (it + 1), atol=2e-7 * (it + 1)), (what, it, float((a - b).abs().max()))
        assert torch.equal(lp, w[:n_lp].to(torch.bfloat16))
        if total is not None:
            assert torch.allclose(tn, total, rtol=1e-5)


def test_flat_store_step_fused_equals_torch_path(monkeypatch):
    """FlatMixedPrecision.step(): the one-pass kernel against the torch path (scale, optimizer.step(), cast) from the same state."""
    import copy
    import tripled_amd  # noqa: F401
    from tripled_amd.flat_amp import FlatMixedPrecision
    torch.manual_seed(1)

    def build():
        torch.manual_seed(1)
        net = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3), torch.nn.BatchNorm2d(16), torch.nn.Conv2d(16, 8, 1)).cuda()
        return net, FlatMixedPrecision(net, lr=1e-3, max_norm=0.1, lowp=True)
    (net_a, fa), (net_b, fb) = build(), build()
    for it in range(4):
        g = torch.randn_like(fa.flat_g)
        fa.flat_g.copy_(g)
        fb.flat_g.copy_(g)
        monkeypatch.delenv("TD_NO_FUSED_ADAM", raising=False)
        ta = fa.step()
        monkeypatch.setenv("TD_NO_FUSED_ADAM", "1")
        tb = fb.step()
        torch.cuda.synchronize()
        assert torch.allclose(ta, tb)
        assert torch.allclose(fa.flat_w, fb.flat_w, rtol=1e-5, atol=2e-7 * (it + 1))      # updates are lr * O(1) = 1e-3
        assert torch.equal(fa.flat_lp, fa.flat_w[:fa.n_lp].to(torch.bfloat16))
        sa, sb = fa.optimizer.state[fa.master], fb.optimizer.state[fb.master]
        assert float(sa["step"]) == float(sb["step"]) == it + 1
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-9)
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-12)


def test_gradient_gather_equals_the_concatenation_path(monkeypatch):
    """FlatMixedPrecision.collect(): td_gather_flat (pointers in the kernel arguments, bf16 -> fp32 on the way) against the torch.cat
    + cast path, bit for bit -- including a parameter no gradient reached (its slot must read zero), odd sizes and more than 64
    tensors (two launches per group)."""
    import tripled_amd  # noqa: F401
    from tripled_amd.flat_amp import FlatMixedPrecision
    torch.manual_seed(2)
    layers = []
    for i in range(40):
        layers += [torch.nn.Conv2d(8 + (i % 3), 8 + ((i + 1) % 3), 3 if i % 2 else 1), torch.nn.BatchNorm2d(8 + ((i + 1) % 3))]
    net = torch.nn.Sequential(*layers).cuda().to(memory_format=torch.channels_last)
    flat = FlatMixedPrecision(net, lr=1e-3, max_norm=None, lowp=True)
    assert len(flat.lowp) > 64 and len(flat.full) > 64
    for i, p in enumerate(flat.params):
        p.grad = None if i % 17 == 5 else torch.randn_like(p)
    flat.flat_g.fill_(123.0)                       # stale contents must not survive in the parameter slots
    with torch.no_grad():                          # ... and the pads are zero in both paths
        for p, off in zip(flat.params, flat.offsets):
            pass
    monkeypatch.setenv("TD_NO_NATIVE_GATHER", "1")
    flat.flat_g.zero_()
    flat.collect()
    ref = flat.flat_g.clone()
    monkeypatch.delenv("TD_NO_NATIVE_GATHER")
    flat.flat_g.zero_()
    flat.collect()
    torch.cuda.synchronize()
    assert torch.equal(flat.flat_g, ref)
    assert float(ref.abs().sum()) > 0
