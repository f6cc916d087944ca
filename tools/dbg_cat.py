import sys, traceback, collections, torch
sys.path.insert(0, '.')
import bench
from mmcv import Config
cfg = Config.fromfile('config/cfg_kitti_tripleD.py')
dev = torch.device('cuda', 0)
m = cfg.model
model = bench.build_model(cfg, dev, channels_last=True)
batch = bench.synthetic_batch(m['imgs_per_gpu'], m['height'], m['width'], seed=1000, device=dev, frame_ids=tuple(m['frame_ids']))
step = bench.TrainStep(model, cfg, batch, torch.bfloat16)
step.forward_backward()
def where():
    st = traceback.extract_stack()[:-2]
    return " < ".join("%s:%d" % (f.filename.split('/')[-1], f.lineno) for f in st[-12:][::-1] if "module.py" not in f.filename)
def wrap(name, fn):
    def g(ts, *a, **k):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(ts, *a, **k); e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3
        if t > 60:
            print("%s %.0f us" % (name, t), [(tuple(x.shape), x.dtype, x.is_contiguous()) for x in ts][:6], a, k, where(), flush=True)
        return out
    return g
torch.cat = wrap("cat", torch.cat)
torch.stack = wrap("stack", torch.stack)
step.forward_backward()
