"""GPU parity: fused feature-regularisation kernels vs the CPU oracle (forward and backward)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import smooth  # noqa: E402
from tests.util import rel_err, smooth_image  # noqa: E402


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,h,w,H,W", [(2, 16, 12, 20, 24, 40), (1, 64, 6, 20, 192, 640), (2, 8, 3, 3, 6, 6),
                                         (1, 24, 17, 9, 17, 9)])
def test_feature_regularization(dtype, B, C, h, w, H, W):
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(B, C, h, w, generator=g).to(dtype)
    img = smooth_image(g, B, 3, max(H, 8), max(W, 8))[:, :, :H, :W].contiguous()
    dis, cvt = 1e-3, 2e-3
    f = feat.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    im = ops.area_downsample(img.cuda(), h, w)
    loss = ops.feature_regularization(f, im, dis, cvt)
    (loss * 5.0).backward()
    fr = feat.float().clone().requires_grad_(True)       # the reference works on feature.float()
    ref = smooth.feature_regularization_loss(fr, img, dis, cvt)
    (ref * 5.0).backward()
    assert abs(float(loss) - float(ref)) < 1e-9 + 5e-5 * abs(float(ref))
    # bf16: the gradient is rounded to bf16 once at the end (8 significant bits)
    assert rel_err(f.grad.float(), fr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
