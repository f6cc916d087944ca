"""FlatMixedPrecision == autocast + clip_grad_norm_ + Adam on the same model (CPU, bf16 autocast)."""
import copy

import torch
import torch.nn as nn

import tripled_amd  # noqa: F401
from tripled_amd.flat_amp import FlatMixedPrecision


def _net():
    torch.manual_seed(3)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(),
                         nn.Conv2d(8, 4, 1, bias=False), nn.BatchNorm2d(4)).to(memory_format=torch.channels_last)


def test_flat_store_matches_autocast_adam():
    ref, net = _net(), None
    net = copy.deepcopy(ref)
    x = torch.randn(4, 3, 12, 16)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    flat = FlatMixedPrecision(net, lr=1e-2, max_norm=0.5)
    assert net[1].weight.dtype == torch.float32
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            la = ref(x).float().square().mean()
        la.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        opt.step()

        flat.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            lb = net(x).float().square().mean()
        lb.backward()
        flat.collect()
        flat.allreduce()
        flat.step()
        assert abs(float(la.detach()) - float(lb.detach())) <= 2e-2 * abs(float(la.detach()))
    # masters follow the fp32 reference weights (bf16 rounding of the working copy is the only difference)
    twin = {id(p): q for p, q in zip(net.parameters(), ref.parameters())}
    for p, off in zip(flat.params, flat.offsets):      # flat order: convolution parameters first, then the fp32 ones
        master = torch.as_strided(flat.flat_w, p.size(), p.stride(), off)
        assert torch.allclose(master, twin[id(p)].detach(), atol=3e-3), p.shape
        if p.dtype == torch.bfloat16:   # (TD_FLAT_LOWP=0 keeps every parameter fp32)
            assert torch.equal(p.detach(), master.to(torch.bfloat16))
        assert off % 8 == 0
