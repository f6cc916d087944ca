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
            assert float(p.grad.abs().max()) == 0.0, n
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
