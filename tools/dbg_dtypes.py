import sys, traceback, collections, torch
sys.path.insert(0, '.')
import bench
import torch.nn.functional as F
from mmcv import Config
from tripled_amd import ops
cfg = Config.fromfile('config/cfg_kitti_tripleD.py')
dev = torch.device('cuda', 0)
m = cfg.model
model = bench.build_model(cfg, dev, channels_last=True)
batch = bench.synthetic_batch(m['imgs_per_gpu'], m['height'], m['width'], seed=1000, device=dev, frame_ids=tuple(m['frame_ids']))
step = bench.TrainStep(model, cfg, batch, torch.bfloat16)
log = collections.Counter()
def where():
    st = traceback.extract_stack()[:-2]
    return " < ".join("%s:%d" % (f.filename.split('/')[-1], f.lineno) for f in st[-16:-1][::-1] if "module.py" not in f.filename)
oi = F.interpolate
def interp(x, *a, **k):
    if x.dtype == torch.float32:
        log[("interp", tuple(x.shape), k.get('mode', a[2] if len(a) > 2 else None), where())] += 1
    return oi(x, *a, **k)
F.interpolate = interp
orp = ops.reflpad1
def rp(x):
    if x.dtype == torch.float32:
        log[("reflpad1", tuple(x.shape), torch.is_autocast_enabled(), torch.is_grad_enabled(), where())] += 1
    return orp(x)
ops.reflpad1 = rp
step.forward_backward()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print(v, k)
