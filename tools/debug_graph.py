"""Find which part of the training step breaks HIP graph capture."""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tripled_amd
from mmcv import Config
from mono.datasets.synthetic import synthetic_batch
from mono.model import MONO
import bench

stage = sys.argv[1]
dev = torch.device("cuda", 0)
cfg = Config.fromfile(os.path.join(ROOT, "config", "cfg_kitti_tripleD.py"))
for k in ("depth_num_layers", "extractor_num_layers"):
    cfg.model[k] = 18
cfg.model["imgs_per_gpu"] = 2
torch.manual_seed(0)
model = bench.build_model(cfg, dev, True)
batch = synthetic_batch(2, 192, 640, seed=1, device=dev)
step = bench.TrainStep(model, cfg, batch, torch.bfloat16)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("warm ok", flush=True)
g = torch.cuda.CUDAGraph()
if stage == "ops":
    from tripled_amd import ops
    tgt = batch[("color", 0, 0)]; srcs = [batch[("color", -1, 0)], batch[("color", 1, 0)]]
    with torch.cuda.graph(g):
        idl = ops.photo_identity(tgt, srcs)
    g.replay(); torch.cuda.synchronize(); print("ops capture ok", float(idl.sum()))
elif stage == "fwd":
    with torch.cuda.graph(g):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            out, losses = model(dict(batch))
            tot = sum(v.float().mean() for v in losses.values())
    g.replay(); torch.cuda.synchronize(); print("fwd capture ok", float(tot))
elif stage == "fwdbwd":
    step.optimizer.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out, losses = model(dict(batch))
        tot = sum(v.float().mean() for v in losses.values())
        tot.backward()
    g.replay(); torch.cuda.synchronize(); print("fwdbwd capture ok", float(tot))
elif stage == "full":
    step.optimizer.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        step()
    g.replay(); torch.cuda.synchronize(); print("full capture ok", float(step.loss))
