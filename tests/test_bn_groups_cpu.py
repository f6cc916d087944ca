"""bn_groups(G): a stacked batch through BatchNorm equals G separate passes (ATen fallback on CPU)."""
import torch

import tripled_amd  # noqa: F401
from mono.model import networks


def test_grouped_batchnorm_fallback_equals_separate_passes():
    torch.manual_seed(0)
    net = networks.build_resnet(18).train()
    ref = networks.build_resnet(18).train()
    ref.load_state_dict(net.state_dict())
    a, b = torch.randn(2, 3, 32, 64), torch.randn(2, 3, 32, 64) * 2 + 1
    with networks.bn_groups(2):
        stacked = net.pyramid(torch.cat([a, b], 0))
    networks.bump_batch_counters(net)
    fa, fb = ref.pyramid(a), ref.pyramid(b)
    networks.bump_batch_counters(ref)
    for s, x, y in zip(stacked, fa, fb):
        assert torch.allclose(s[:2], x, atol=1e-5) and torch.allclose(s[2:], y, atol=1e-5)
    for (k, v), (_, r) in zip(net.state_dict().items(), ref.state_dict().items()):
        assert torch.allclose(v.float(), r.float(), atol=1e-6), k       # running stats and counters included
