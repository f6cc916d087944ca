"""Multi-process data-parallel path on CPU (gloo, world_size 2): sampler sharding and the
bucketed gradient all-reduce engine behind mmcv.parallel.MMDistributedDataParallel."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 16)
        self.unused = nn.Linear(4, 4)          # never receives a gradient (like ResNet.fc)
        self.c = nn.Linear(16, 1)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import tripled_amd  # noqa: F401
    from mmcv.parallel import MMDistributedDataParallel
    from mmcv.runner import Runner
    from mono.core import DistOptimizerHook
    from mono.datasets import DistributedGroupSampler
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mono.apis import init_dist
    init_dist("pytorch", backend="gloo")
    torch.manual_seed(100 + rank)               # different init per rank: the wrapper must broadcast rank 0's
    net = Net()
    ddp = MMDistributedDataParallel(net, bucket_cap_mb=0.0002, find_unused_parameters=True)  # tiny buckets -> several
    assert len(ddp.reducer.buckets) >= 3
    w0 = [p.detach().clone() for p in net.parameters()]
    gathered = [None] * world
    dist.all_gather_object(gathered, [w.tolist() for w in w0])
    assert gathered[0] == gathered[1]

    g = torch.Generator().manual_seed(7 + rank)
    x, y = torch.randn(5, 8, generator=g), torch.randn(5, 1, generator=g)
    ddp.train()
    loss = (ddp(x) - y).pow(2).mean()
    loss.backward()
    assert ddp.grads_synchronised()
    # reference result: average over ranks of the local gradients
    local = Net()
    local.load_state_dict(net.state_dict())
    (local(x) - y).pow(2).mean().backward()
    for (n, p), q in zip(net.named_parameters(), local.parameters()):
        if q.grad is None:
            assert p.grad is None, n      # unused parameter: left without a gradient, the optimiser skips it (as torch DDP)
            continue
        t = q.grad.clone()
        dist.all_reduce(t)
        t /= world
        assert torch.allclose(p.grad, t, atol=1e-6), n

    # two optimiser steps through the Runner hook keep the replicas identical
    class DS:
        flag = np.zeros(8, dtype=np.int64)

        def __len__(self):
            return 8
    sampler = DistributedGroupSampler(DS(), 2, world, rank)
    mine = list(iter(sampler))
    allidx = [None] * world
    dist.all_gather_object(allidx, mine)
    assert sorted(allidx[0] + allidx[1]) == list(range(8)) and len(mine) == 4

    def bp(model, data, train_mode):
        l = (model(data[0]) - data[1]).pow(2).mean()
        return dict(loss=l, log_vars={"loss": l.detach()}, num_samples=len(data[0]))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    runner = Runner(ddp, bp, opt, os.path.join(out_dir, "r%d" % rank), "ERROR")
    runner.register_training_hooks(dict(policy="fixed"), DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)),
                                   dict(interval=100), None)
    runner.run([[(x, y), (x * 0.5, y)]], [("train", 1)], 1)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    parts = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(parts, flat)
    assert torch.equal(parts[0], parts[1])
    assert not torch.equal(flat, torch.cat([w.reshape(-1) for w in w0]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_sync(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def _worker_deferred(rank, world, port):
    """The benchmark's N > 1 form: no autograd hooks, all buckets reduced after backward (between two HIP
    graphs on the GPU), and the flat mixed-precision store's own all-reduce."""
    sys.path.insert(0, ROOT)
    import tripled_amd  # noqa: F401
    from mmcv.parallel import MMDistributedDataParallel
    from tripled_amd.flat_amp import FlatMixedPrecision
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)
    net = Net()
    ddp = MMDistributedDataParallel(net, bucket_cap_mb=0.0002, find_unused_parameters=True, overlap=False)
    g = torch.Generator().manual_seed(11 + rank)
    x, y = torch.randn(5, 8, generator=g), torch.randn(5, 1, generator=g)
    ddp.train()
    (ddp(x) - y).pow(2).mean().backward()
    local = Net()
    local.load_state_dict(net.state_dict())
    (local(x) - y).pow(2).mean().backward()
    mine = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    for (n, p), q in zip(net.named_parameters(), local.parameters()):      # nothing exchanged yet
        if q.grad is not None:
            assert torch.allclose(mine[n], q.grad, atol=1e-6), n
    ddp.reducer.allreduce_all()
    for (n, p), q in zip(net.named_parameters(), local.parameters()):
        if q.grad is None:
            continue
        t = q.grad.clone()
        dist.all_reduce(t)
        t /= world
        assert torch.allclose(p.grad, t, atol=1e-6), n

    # FlatMixedPrecision: gather -> bucketed all-reduce of the flat buffer -> clip + Adam keeps replicas identical
    torch.manual_seed(9)
    conv = nn.Sequential(nn.Conv2d(3, 4, 3, padding=1), nn.BatchNorm2d(4), nn.ReLU(), nn.Conv2d(4, 2, 1))
    flat = FlatMixedPrecision(conv, lr=1e-2, max_norm=1.0, bucket_bytes=256)
    assert len(flat.buckets) > 1
    img = torch.randn(2, 3, 6, 6, generator=g)
    for _ in range(2):
        flat.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            conv(img).float().square().mean().backward()
        flat.collect()
        local_g = flat.flat_g.clone()
        flat.allreduce()
        want = local_g.clone()
        dist.all_reduce(want)
        assert torch.allclose(flat.flat_g, want / world, atol=1e-7)
        flat.step()
    parts = [torch.zeros_like(flat.flat_w) for _ in range(world)]
    dist.all_gather(parts, flat.flat_w)
    assert torch.equal(parts[0], parts[1])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_deferred_allreduce_and_flat_store():
    mp.spawn(_worker_deferred, args=(2, _free_port()), nprocs=2, join=True)


def _worker_syncbn(rank, world, port):
    """enable_sync_batchnorm on CPU ranks (torch-op staging of the same algorithm the HIP kernels implement): the
    two half-batches must normalise exactly like one BatchNorm over the whole batch, forward and backward, including
    stacked passes (bn_groups) and the running statistics."""
    sys.path.insert(0, ROOT)
    import tripled_amd  # noqa: F401
    from mono.model import networks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mono.apis import init_dist
    init_dist("pytorch", backend="gloo")
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(9)
    full = torch.randn(8, 64, 5, 7, generator=g)
    up = torch.randn(8, 64, 5, 7, generator=g)
    for groups in (1, 2):
        bn = networks.BatchNorm(64)
        ref = torch.nn.BatchNorm2d(64)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(64, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(64, generator=g))
            ref.load_state_dict(bn.state_dict())
        assert networks.enable_sync_batchnorm(torch.nn.Sequential(bn)) == 1
        # rank r owns samples [r::2] of every stacked pass
        per = 8 // groups
        mine = torch.cat([full[k * per:(k + 1) * per][rank::world] for k in range(groups)], 0).clone().requires_grad_(True)
        mine_up = torch.cat([up[k * per:(k + 1) * per][rank::world] for k in range(groups)], 0)
        with networks.bn_groups(groups):
            y = bn(mine)
        (y * mine_up).sum().backward()
        xr = full.clone().requires_grad_(True)
        yr = torch.cat([ref(xr[k * per:(k + 1) * per]) for k in range(groups)], 0)      # groups separate passes
        (yr * up).sum().backward()
        want_y = torch.cat([yr[k * per:(k + 1) * per][rank::world] for k in range(groups)], 0)
        want_dx = torch.cat([xr.grad[k * per:(k + 1) * per][rank::world] for k in range(groups)], 0)
        assert torch.allclose(y, want_y, atol=2e-5), float((y - want_y).abs().max())
        assert torch.allclose(mine.grad, want_dx, atol=2e-5), float((mine.grad - want_dx).abs().max())
        assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-5)
        assert torch.allclose(bn.running_var, ref.running_var, atol=1e-5)
        # parameter gradients are local sums: their sum over the ranks is the single-process gradient
        gw, gb = bn.weight.grad.clone(), bn.bias.grad.clone()
        dist.all_reduce(gw)
        dist.all_reduce(gb)
        assert torch.allclose(gw, ref.weight.grad, atol=2e-4) and torch.allclose(gb, ref.bias.grad, atol=2e-4)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sync_batchnorm():
    port = _free_port()
    mp.spawn(_worker_syncbn, args=(2, port), nprocs=2, join=True)


def _worker_agreement(rank, world, port):
    """The harness logic that keeps N > 1 runs honest (bench.py, tripled_amd.step.RunnerIteration): the ranks settle on ONE
    step form (all_reduce(MIN) of each rank's outcome), the candidate order follows the backend and the config's syncbn, and
    replicas whose parameters differ are detected on every rank."""
    sys.path.insert(0, ROOT)
    import tripled_amd  # noqa: F401
    from tripled_amd.step import RunnerIteration, ranks_agree, replicas_agree
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mono.apis import init_dist
    init_dist("pytorch", backend="gloo")
    cpu = torch.device("cpu")
    assert ranks_agree(True, cpu) is True
    assert ranks_agree(rank == 0, cpu) is False            # a form that one rank could not bring up is dropped by all
    assert ranks_agree(False, cpu) is False

    class _Step:                                             # only what the mode logic touches
        device = cpu
    it = RunnerIteration(_Step(), lambda d: d, lambda *a, **k: None, syncbn=False)
    assert it._candidate_modes() == ["two-graph", "eager"]   # gloo collectives cannot be captured: never "one-graph"
    assert RunnerIteration(_Step(), lambda d: d, lambda *a, **k: None, syncbn=True)._candidate_modes() == ["eager"]
    assert it._agree(rank == 1) is False and it._agree(True) is True

    torch.manual_seed(3)
    net = Net()
    assert replicas_agree(net)
    if rank == 1:
        with torch.no_grad():
            net.c.weight[0, 0] += 1e-3
    assert not replicas_agree(net)                           # detected on BOTH ranks
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_mode_agreement_and_replica_check():
    mp.spawn(_worker_agreement, args=(2, _free_port()), nprocs=2, join=True)
