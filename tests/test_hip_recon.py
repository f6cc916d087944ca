"""GPU parity: streaming masked-reconstruction kernels vs the CPU oracle (forward + gradient)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import photometric  # noqa: E402
from tests.util import rel_err, smooth_image  # noqa: E402


@pytest.mark.parametrize("B,h,w", [(2, 24, 80), (1, 3, 3), (2, 17, 131), (1, 48, 60), (3, 9, 62), (1, 8, 124),
                                   (12, 192, 640), (12, 24, 80), (4, 320, 1024)])      # C2 scale 0 / 3, C4 scale 0
def test_masked_reconstruction(B, h, w):
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(5)
    x = smooth_image(g, B, 3, max(h, 8), max(w, 8))[:, :, :h, :w].contiguous()
    y = (x + 0.08 * torch.randn(B, 3, h, w, generator=g)).clamp(0, 1)
    mask = (torch.rand(B, 1, h, w, generator=g) > 0.3).float().repeat(1, 3, 1, 1)
    hole3 = 1 - mask
    hole3[0, :, 0, 0] = 1.0                      # make sure corners and borders carry weight
    hole3[0, :, h - 1, w - 1] = 1.0
    xg = x.cuda().requires_grad_(True)
    S = ops.masked_reconstruction_sum(xg, y.cuda(), hole3.sum(1).cuda())
    (S * 0.5).backward()
    xr = x.clone().requires_grad_(True)
    ref = torch.sum(photometric.reprojection_loss(xr, y) * hole3)
    (ref * 0.5).backward()
    assert abs(float(S) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert rel_err(xg.grad, xr.grad) < 2e-3       # SSIM-adjoint conditioning (C2 = 9e-4), fast rcp/sqrt
