import sys, torch
sys.path.insert(0,'.')
import tripled_amd
from mono.model import networks as N
import numpy as np
import torch.nn.functional as F
dec = N.Decoder(np.array([64,256,512,1024,2048])).cuda().to(memory_format=torch.channels_last)
feats=[torch.randn(2,c,h,w,device='cuda').bfloat16().contiguous(memory_format=torch.channels_last) for c,h,w in [(64,32,64),(256,16,32),(512,8,16),(1024,4,8),(2048,2,4)]]
def hook(name):
    def f(m, i, o):
        print(name, type(m).__name__, [t.dtype for t in i if torch.is_tensor(t)], '->', o.dtype if torch.is_tensor(o) else type(o))
    return f
for n,m in dec.named_modules():
    if n and n.count('.')<=2: m.register_forward_hook(hook(n))
with torch.autocast('cuda', dtype=torch.bfloat16):
    out = dec(feats, 0)
    x = torch.randn(2,8,4,4,device='cuda').bfloat16()
    print('elu', F.elu(x).dtype, 'elu_', F.elu(x.clone(), inplace=True).dtype, 'leaky', F.leaky_relu(x).dtype, 'interp', F.interpolate(x, scale_factor=2).dtype)
    from tripled_amd import ops
    xc = torch.randn(2,8,4,4,device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    print('reflpad1', ops.reflpad1(xc).dtype)
