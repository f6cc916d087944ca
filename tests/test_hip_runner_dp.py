"""Two data-parallel ranks through the reference's distributed entry (mono.apis.train_mono(distributed=True): init_dist ->
MMDistributedDataParallel -> Runner.run, reference: mono/apis/trainer.py:147-189) with the graph-replayed iteration, on ONE GPU
over gloo (the rehearsal this pool allows; RCCL needs one device per rank).  gloo collectives cannot be captured, so the ranks
must settle on the two-graph form (forward+backward graph | eager all-reduce of the flat gradient buffer | clip+Adam graph),
every rank must take the same decision, and the replicas must hold identical weights after the run although every rank trained
on different frames."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, work):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import logging
    import tripled_amd  # noqa: F401
    import torch.distributed as dist
    from mmcv import Config
    from mono.apis import init_dist, train_mono
    from mono.datasets import ResidentBatches, synthetic_batch
    from mono.model import MONO
    from tripled_amd.step import replicas_agree
    init_dist("pytorch", backend="gloo")
    torch.cuda.set_device(0)
    cfg = Config.fromfile(os.path.join(ROOT, "config", "cfg_kitti_tripleD.py"))
    cfg.model.update(depth_num_layers=18, pose_num_layers=18, extractor_num_layers=18, imgs_per_gpu=2, height=96, width=160)
    cfg.imgs_per_gpu, cfg.total_epochs, cfg.validate, cfg.gpus = 2, 1, False, [0]
    cfg.work_dir = os.path.join(work, "rank%d" % rank)
    cfg.log_config = dict(interval=1, hooks=[dict(type="TextLoggerHook")])
    cfg.log_level = "WARNING"
    cfg.syncbn, cfg.strict_dispatch, cfg.cudnn_benchmark, cfg.hip_graph = False, True, False, True
    records = []

    class Grab(logging.Handler):
        def emit(self, r):
            records.append(r.getMessage())
    logging.getLogger("tripled_amd.step").addHandler(Grab())
    logging.getLogger("tripled_amd.step").setLevel(logging.INFO)
    torch.manual_seed(100 + rank)                # different initial weights: the wrapper broadcasts rank 0's
    model = MONO.module_dict[cfg.model["name"]](cfg.model)
    m = cfg.model
    batch = synthetic_batch(2, m["height"], m["width"], seed=1000 + rank, device=torch.device("cuda", 0),
                            frame_ids=tuple(m["frame_ids"]))
    train_mono(model, ResidentBatches(batch, 7), None, cfg, distributed=True, validate=False)
    torch.cuda.synchronize()
    assert any("training iteration: two-graph" in r for r in records), records
    assert replicas_agree(model)
    flat = model._flat_store
    assert bool(torch.isfinite(flat.flat_w).all())
    # the ranks saw different frames, so identical weights mean the gradients were averaged in every iteration
    chk = torch.tensor([float(flat.flat_w.double().sum())], dtype=torch.float64)
    got = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(got, chk)
    assert float(got[0]) == float(got[1])
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_two_graph_form(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
