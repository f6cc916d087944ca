import sys, time, torch
sys.path.insert(0, '.')
import bench
from mmcv import Config
cfg = Config.fromfile('config/cfg_kitti_tripleD.py')
dev = torch.device('cuda', 0)
torch.backends.cudnn.benchmark = False
m = cfg.model
model = bench.build_model(cfg, dev, channels_last=True)
batch = bench.synthetic_batch(m['imgs_per_gpu'], m['height'], m['width'], seed=1000, device=dev, frame_ids=tuple(m['frame_ids']))
step = bench.TrainStep(model, cfg, batch, torch.bfloat16, flat=True)
for _ in range(2):
    step()
step.forward_backward()
fl = step.flat
def t(fn, n=10):
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
print('collect %.3f ms' % t(fl.collect))
k = len(fl.lowp)
srcs = fl._flat_sources(fl.lowp, fl.offsets[:k], fl.n_lp, fl.flat_lp.dtype)
print('n lowp srcs', len(srcs), 'n full', len(fl.full))
print('cat lowp %.3f ms' % t(lambda: torch.cat(srcs, out=fl.flat_glp)))
print('cast %.3f ms' % t(lambda: fl.flat_g[:fl.n_lp].copy_(fl.flat_glp)))
srcf = fl._flat_sources(fl.full, fl.offsets[k:], fl.flat_g.numel(), torch.float32)
print('cat full %.3f ms (%d srcs)' % (t(lambda: torch.cat(srcf, out=fl.flat_g[fl.n_lp:])), len(srcf)))
print('step %.3f ms' % t(fl.step))
print('numel', fl.flat_g.numel(), 'n_lp', fl.n_lp)
