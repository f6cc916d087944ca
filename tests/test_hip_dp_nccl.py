"""GPU-side test of the collective path: a one-rank RCCL ("nccl") process group on the MI355X, the bucket engine
of mmcv.parallel.MMDistributedDataParallel in overlap mode (ReduceOp.AVG, asynchronous all-reduce of flat-buffer
slices, join in the autograd-engine callback) and DistOptimizerHook on top of it.
Reference semantics: mono/core/utils/dist_utils.py:34-60, mono/apis/trainer.py:147-162."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.nn as nn

pytestmark = pytest.mark.gpu


class Net(nn.Module):
    """Convolutions with channels-last weights, a BatchNorm, and one parameter no gradient ever reaches."""

    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(3, 16, 3, padding=1, bias=False)   # (a bias in front of BatchNorm has a zero gradient: Adam would amplify its rounding noise)
        self.bn = nn.BatchNorm2d(16)
        self.b = nn.Conv2d(16, 8, 3, padding=1)
        self.head = nn.Linear(8, 1)
        self.unused = nn.Linear(4, 4)

    def forward(self, x):
        y = self.b(torch.relu(self.bn(self.a(x))))
        return self.head(y.mean((2, 3)))


@pytest.fixture(scope="module")
def nccl_group():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def _models():
    torch.manual_seed(3)
    ref = Net().cuda().to(memory_format=torch.channels_last)
    net = Net().cuda().to(memory_format=torch.channels_last)
    net.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 16, 24, generator=g).cuda()
    y = torch.randn(4, 1, generator=g).cuda()
    return ref, net, x, y


def test_overlapped_buckets_match_plain_backward(nccl_group):
    import tripled_amd  # noqa: F401
    from mmcv.parallel import MMDistributedDataParallel
    ref, net, x, y = _models()
    ddp = MMDistributedDataParallel(net, device_ids=[0], find_unused_parameters=True, bucket_cap_mb=0.0005,
                                    engine_at_world_1=True)
    eng = ddp.reducer
    assert eng is not None and eng.overlap and eng.use_avg and len(eng.buckets) >= 2
    ddp.train()
    ref.train()
    for _ in range(2):                                   # second pass: re-attach after the first finalize
        (ddp(x) - y).pow(2).mean().backward()
        assert ddp.grads_synchronised()
        ref.zero_grad(set_to_none=True)
        (ref(x) - y).pow(2).mean().backward()
        torch.cuda.synchronize()
        for (n, p), q in zip(net.named_parameters(), ref.parameters()):
            if q.grad is None:
                assert p.grad is None, n                 # the optimiser must skip it, as under torch DDP
                continue
            assert p.grad.stride() == p.stride(), n      # gradient views keep the parameter's (channels-last) layout
            assert p.grad.data_ptr() >= eng.flat.data_ptr() and \
                p.grad.data_ptr() < eng.flat.data_ptr() + eng.flat.numel() * 4, n
            assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7), n


def test_dist_optimizer_hook_skips_second_allreduce(nccl_group, tmp_path, monkeypatch):
    import tripled_amd  # noqa: F401
    from mmcv.parallel import MMDistributedDataParallel
    from mmcv.runner import Runner
    from mono.core import DistOptimizerHook
    from mono.core.utils import dist_utils
    ref, net, x, y = _models()
    ddp = MMDistributedDataParallel(net, device_ids=[0], find_unused_parameters=True, engine_at_world_1=True)

    def forbidden(*a, **k):
        raise AssertionError("the reference's second, post-backward all-reduce ran although the gradients were "
                             "already synchronised by the bucket engine")
    monkeypatch.setattr(dist_utils, "allreduce_grads", forbidden)

    def bp(model, data, train_mode):
        l = (model(data[0]) - data[1]).pow(2).mean()
        return dict(loss=l, log_vars={"loss": l.detach()}, num_samples=len(data[0]))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, fused=True)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2, fused=True)
    runner = Runner(ddp, bp, opt, str(tmp_path), "ERROR")
    runner.register_training_hooks(dict(policy="fixed"), DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)),
                                   dict(interval=100), None)
    batches = [(x, y), (x * 0.5, y)]
    runner.run([batches], [("train", 1)], 1)
    ref.train()
    for bx, by in batches:                               # mono/core/utils/dist_utils.py:54-60 on the bare model
        ropt.zero_grad()
        (ref(bx) - by).pow(2).mean().backward()
        torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.grad is not None], 35, 2)
        ropt.step()
    torch.cuda.synchronize()
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), n
    assert torch.equal(net.unused.weight, ref.unused.weight)       # never stepped


def test_deferred_allreduce_and_flat_store(nccl_group):
    """The forms bench.py uses around HIP graphs at N > 1: all buckets after backward, and the flat store's own
    all-reduce (ReduceOp.AVG on RCCL)."""
    import tripled_amd  # noqa: F401
    from mmcv.parallel import MMDistributedDataParallel
    from tripled_amd.flat_amp import FlatMixedPrecision
    ref, net, x, y = _models()
    ddp = MMDistributedDataParallel(net, device_ids=[0], overlap=False, engine_at_world_1=True)
    ddp.train()
    ref.train()
    (ddp(x) - y).pow(2).mean().backward()
    assert not ddp.grads_synchronised()
    ddp.reducer.allreduce_all()
    assert ddp.grads_synchronised()
    (ref(x) - y).pow(2).mean().backward()
    torch.cuda.synchronize()
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        if q.grad is not None:
            assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7), n

    ref2, net2, x, y = _models()
    flat = FlatMixedPrecision(net2, lr=1e-2, max_norm=35.0, lowp=False)
    flat.zero_grad()
    (net2(x) - y).pow(2).mean().backward()
    flat.collect()
    before = flat.flat_g.clone()
    flat.allreduce(force=True)
    torch.cuda.synchronize()
    assert torch.allclose(flat.flat_g, before)           # average over one rank
    flat.step()
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(p).all()) for p in net2.parameters())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu,with_res,groups", [(True, True, 1), (True, False, 2), (False, False, 1)])
def test_sync_batchnorm_kernels_one_rank(nccl_group, dtype, relu, with_res, groups):
    """The staged (sums -> RCCL all-reduce -> apply / dx) form of the BatchNorm kernels in a one-rank group must equal
    the fused local form bit for bit in the forward and to rounding in the backward."""
    import tripled_amd  # noqa: F401
    from tripled_amd import ops
    g = torch.Generator().manual_seed(1)
    N, C, H, W = 4, 128, 9, 13
    mk = lambda: torch.randn(N, C, H, W, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    x, res, up = mk(), (mk() if with_res else None), mk()
    w = (torch.rand(C, generator=g) + 0.5).cuda()
    b = torch.randn(C, generator=g).cuda()
    outs = []
    for sync in (False, True):
        xi = x.clone().requires_grad_(True)
        ri = res.clone().requires_grad_(True) if res is not None else None
        wi, bi = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
        if sync:
            y = ops.sync_batchnorm_act(xi, wi, bi, rm, rv, 0.1, 1e-5, None, residual=ri, relu=relu, groups=groups)
        else:
            y = ops.batchnorm_act(xi, wi, bi, rm, rv, 0.1, 1e-5, residual=ri, relu=relu, groups=groups)
        (y.float() * up.float()).sum().backward()
        outs.append((y.detach().float(), xi.grad.float(), wi.grad, bi.grad, rm, rv, ri.grad.float() if ri is not None else None))
    a, s = outs
    assert torch.equal(a[0], s[0])
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((a[1] - s[1]).abs().max()) <= tol * float(a[1].abs().max())
    assert torch.allclose(a[2], s[2], rtol=1e-4, atol=1e-3) and torch.allclose(a[3], s[3], rtol=1e-4, atol=1e-3)
    assert torch.allclose(a[4], s[4], atol=1e-6) and torch.allclose(a[5], s[5], atol=1e-6)
    if with_res:
        assert torch.equal(a[6], s[6])
