"""Which ATen ops issue the device memsets / memcpys of one training step (torch.profiler, eager)."""
import collections
import sys
import torch
sys.path.insert(0, '.')
import bench
from mmcv import Config
from torch.profiler import profile, ProfilerActivity
cfg = Config.fromfile('config/cfg_kitti_tripleD.py')
dev = torch.device('cuda', 0)
m = cfg.model
model = bench.build_model(cfg, dev, channels_last=True)
batch = bench.synthetic_batch(m['imgs_per_gpu'], m['height'], m['width'], seed=1000, device=dev, frame_ids=tuple(m['frame_ids']))
step = bench.TrainStep(model, cfg, batch, torch.bfloat16)
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
cnt = collections.Counter()
for e in cpu:
    for k in e.kernels:
        if 'emset' in k.name or 'fillBuffer' in k.name or 'emcpy' in k.name or 'copyBuffer' in k.name:
            # innermost op only: skip if a child op also owns this kernel
            if not any(k in c.kernels for c in e.cpu_children):
                st = [f for f in (e.stack or []) if 'site-packages' not in f and 'dist-packages' not in f][:3]
                cnt[(k.name[:16], e.name, str(e.input_shapes)[:60], ' < '.join(x.split('/')[-1] for x in st))] += 1
for k, v in cnt.most_common(25):
    print(v, k)
