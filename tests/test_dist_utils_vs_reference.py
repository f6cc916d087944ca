"""Row a13 against the REAL reference: mono/core/utils/dist_utils.py (allreduce_grads, DistOptimizerHook.after_train_iter) is
loaded from /root/reference as a stand-alone module on two gloo ranks -- its one non-torch import, mmcv.runner.OptimizerHook (only the
base class that supplies clip_grads), resolves to this build's mmcv shim -- and run beside this build's mono.core.utils.dist_utils
on the same model, gradients and optimiser.  Gradients after the exchange and parameters after the hook's step must be EQUAL (two
ranks: the sum of two numbers and the division by two are exact in any bucket layout).  Skipped where the reference is absent."""
import importlib.util
import os
import socket
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FILE = "/root/reference/mono/core/utils/dist_utils.py"
pytestmark = pytest.mark.skipif(not os.path.isfile(REF_FILE), reason="reference checkout not present")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(3, 8, 3, padding=1)
        self.b = nn.BatchNorm2d(8)
        self.frozen = nn.Linear(4, 4)            # requires_grad False: neither side may touch it
        self.unused = nn.Linear(4, 4)            # no gradient reaches it
        self.c = nn.Linear(8, 1)
        for p in self.frozen.parameters():
            p.requires_grad = False

    def forward(self, x):
        return self.c(torch.relu(self.b(self.a(x))).mean((2, 3)))


def _grads(net, x, y):
    for p in net.parameters():
        p.grad = None
    (net(x) - y).pow(2).mean().backward()


def _worker(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.dont_write_bytecode = True
    import tripled_amd  # noqa: F401  (puts the mmcv shim on the path)
    from mono.core.utils import dist_utils as mine
    spec = importlib.util.spec_from_file_location("_reference_dist_utils", REF_FILE)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(20 + rank)                  # different data per rank
    x, y = torch.randn(4, 3, 6, 6, generator=g), torch.randn(4, 1, generator=g)
    torch.manual_seed(1)
    a, b = Net(), Net()
    b.load_state_dict(a.state_dict())

    for kwargs in (dict(coalesce=True, bucket_size_mb=-1), dict(coalesce=True, bucket_size_mb=1), dict(coalesce=False)):
        _grads(a, x, y)
        _grads(b, x, y)
        ref.allreduce_grads(a, **kwargs)
        mine.allreduce_grads(b, **kwargs)
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            assert (p.grad is None) == (q.grad is None), n
            if p.grad is not None:
                assert torch.equal(p.grad, q.grad), (n, kwargs)
        # ... and it IS the average over the ranks
        other = [torch.zeros_like(a.c.weight.grad) for _ in range(world)]
        dist.all_gather(other, a.c.weight.grad)
        assert torch.equal(other[0], other[1])

    # the hook's iteration: zero_grad -> backward -> exchange -> clip -> step (reference :54-60), three iterations
    opt_a, opt_b = (torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-2) for m in (a, b))
    hook_a = ref.DistOptimizerHook(grad_clip=dict(max_norm=0.05, norm_type=2))         # small enough to clip
    hook_b = mine.DistOptimizerHook(grad_clip=dict(max_norm=0.05, norm_type=2))
    for it in range(3):
        for net, opt, hook in ((a, opt_a, hook_a), (b, opt_b, hook_b)):
            runner = types.SimpleNamespace(model=net, optimizer=opt, outputs=dict(loss=(net(x * (1 + it)) - y).pow(2).mean()))
            hook.after_train_iter(runner)
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            assert torch.equal(p, q), (it, n)
    flat = torch.cat([p.detach().reshape(-1) for p in b.parameters()])
    parts = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(parts, flat)
    assert torch.equal(parts[0], parts[1])                          # replicas stay identical
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reference_hook_and_this_builds_hook_agree():
    mp.spawn(_worker, args=(2, _free_port()), nprocs=2, join=True)
