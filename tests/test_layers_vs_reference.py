"""The building blocks of the reference's mono/model/mono_fm_joint/layers.py against this build's classes of the same names
(mono.model.mono_fm_joint.layers: the unfused compatibility surface + the network blocks of mono.model.networks + the attention
gates).  The reference file needs only torch and numpy and is loaded stand-alone from /root/reference.  Same constructor
arguments, the reference's state_dict loaded strictly (same parameter names), same inputs -> same outputs and input gradients.
Skipped where the reference checkout is absent (the GPU box)."""
import importlib.util
import os
import sys

import pytest
import torch

import tripled_amd  # noqa: F401
from mono.model.mono_fm_joint import layers as mine

REF_FILE = "/root/reference/mono/model/mono_fm_joint/layers.py"
pytestmark = pytest.mark.skipif(not os.path.isfile(REF_FILE), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("_reference_layers", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _rand(*shape, seed=0, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return lo + (hi - lo) * torch.rand(*shape, generator=g)


MODULES = {
    # name: (constructor args, constructor kwargs, input shape)
    "Conv1x1": ((8, 16), {}, (2, 8, 6, 10)),
    "Conv1x1_bias": ((8, 16), dict(bias=True), (2, 8, 6, 10)),
    "Conv3x3": ((8, 12), {}, (2, 8, 6, 10)),
    "Conv3x3_zero_pad": ((8, 12), dict(use_refl=False), (2, 8, 6, 10)),
    "Conv5x5": ((8, 12), {}, (2, 8, 7, 9)),
    "ConvBlock": ((8, 12), {}, (2, 8, 6, 10)),
    "CRPBlock": ((16, 16, 4), {}, (2, 16, 12, 20)),
    "IdentityPartial_left": ((), dict(part_ratio=2, use_right=False), (2, 8, 4, 4)),
    "IdentityPartial_right": ((), dict(part_ratio=4, use_right=True), (2, 8, 4, 4)),
    "SqueezeAndExcitationBlock": ((32,), {}, (2, 32, 1, 1)),
    "CALayer": ((32,), {}, (2, 32, 6, 10)),
    "CALayer_pixel": ((32,), dict(pix_att=True), (2, 32, 6, 10)),
    "CALayer_contrast": ((32,), dict(contrast_aware=True), (2, 32, 6, 10)),
    "AdaptivelyScaledCALayer": ((32,), {}, (2, 32, 6, 10)),
}


@pytest.mark.parametrize("case", sorted(MODULES))
def test_module_equals_the_reference_module(ref, case):
    args, kwargs, shape = MODULES[case]
    cls = case.split("_")[0]
    torch.manual_seed(3)
    a = getattr(ref, cls)(*args, **kwargs)
    b = getattr(mine, cls)(*args, **kwargs)
    missing = b.load_state_dict(a.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert [n for n, _ in b.named_parameters()] == [n for n, _ in a.named_parameters()]
    xa = _rand(*shape, seed=5, lo=-1.0, hi=1.0).requires_grad_(True)
    xb = xa.detach().clone().requires_grad_(True)
    ya, yb = a(xa), b(xb)
    assert ya.shape == yb.shape
    assert float((ya - yb).abs().max()) <= 1e-6 * max(1.0, float(ya.abs().max())), case
    w = _rand(*ya.shape, seed=6, lo=-1.0, hi=1.0)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * max(1.0, float(xa.grad.abs().max())), case
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert float((p.grad - q.grad).abs().max()) <= 1e-5 * max(1.0, float(p.grad.abs().max())), (case, n)


def test_geometry_and_ssim_equal_the_reference(ref, monkeypatch):
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)      # the reference's Backproject says .cuda() (layers.py:49-57)
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
    B, H, W = 2, 12, 20
    disp = _rand(B, 1, H, W, seed=1, lo=0.05, hi=0.95)
    for f in ("disp_to_depth",):
        ra, rb = getattr(ref, f)(disp, 0.1, 100.0), getattr(mine, f)(disp, 0.1, 100.0)
        assert all(torch.equal(x, y) for x, y in zip(ra, rb))
    _, depth = ref.disp_to_depth(disp, 0.1, 100.0)
    K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]]).repeat(B, 1, 1)
    inv_K = torch.linalg.pinv(K)
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, :3, 3] = _rand(B, 3, seed=2, lo=-0.05, hi=0.05)
    pa, pb = ref.Backproject(B, H, W)(depth, inv_K), mine.Backproject(B, H, W)(depth, inv_K)
    assert float((pa - pb).abs().max()) <= 1e-5 * float(pa.abs().max())
    ga, gb = ref.Project(B, H, W)(pa, K, T), mine.Project(B, H, W)(pa, K, T)
    assert ga.shape == gb.shape == (B, H, W, 2) and float((ga - gb).abs().max()) <= 1e-6
    x, y = _rand(B, 3, H, W, seed=3), _rand(B, 3, H, W, seed=4)
    sa, sb = ref.SSIM()(x, y), mine.SSIM()(x, y)
    assert float((sa - sb).abs().max()) <= 1e-6
    up = _rand(B, 4, 5, 7, seed=5)
    assert torch.equal(ref.upsample(up), mine.upsample(up))


def test_upshuffle_equals_the_reference(ref):
    """upshuffle(in_planes, upscale_factor): 3x3 convolution to in_planes * r^2 channels, pixel shuffle, ELU, with the reference's
    sub-pixel initialisation (layers.py:114-134)."""
    torch.manual_seed(0)
    a = ref.upshuffle(8, 2)
    b = mine.upshuffle(8, 2)
    missing = b.load_state_dict(a.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    x = _rand(2, 8, 5, 7, seed=9, lo=-1.0, hi=1.0)
    ya, yb = a(x), b(x)
    assert ya.shape == yb.shape and float((ya - yb).abs().max()) <= 1e-6
    # the initialisation itself: every r^2 group of output channels starts from the same kernel (ICNR), in both
    torch.manual_seed(1)
    wa = ref.upshuffle(8, 2)[1].weight
    torch.manual_seed(1)
    wb = mine.upshuffle(8, 2)[1].weight
    assert wa.shape == wb.shape
    assert torch.equal(wa, wb)                          # same generator consumption, same kernels
    assert torch.equal(wb[0::4], wb[1::4]) and torch.equal(wb[0::4], wb[3::4])
