"""HIP-event timing of the fused BatchNorm kernels on the model's dominant shapes, as GB/s of the bytes each pass
must move (3 passes forward: statistics read, apply read + write; 5 backward)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch
import torch.nn as nn
from stp3_amd import _lib
if os.environ.get('EXP_LIB'):          # experiment builds of the library
    _lib.LIB_PATH = os.environ['EXP_LIB']
from stp3_amd import ops

SHAPES = [('trunk b2 expand', 72, 144, 112, 240), ('trunk b6 expand', 72, 192, 56, 120), ('trunk b10 expand', 72, 336, 28, 60),
          ('trunk b17 expand', 72, 960, 14, 30), ('trunk b2 project', 72, 32, 56, 120), ('bev 64ch', 12, 64, 200, 200),
          ('bev 128ch', 12, 128, 200, 200), ('temporal 35ch sliced', 12, 35, 200, 200)]


def ev(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


print('batch of 72 camera images (B=4 x T=3 x 6), bf16, channels-last')
for name, n, c, h, w in SHAPES:
    cp = (c + 7) // 8 * 8
    xf = torch.randn(n, cp, h, w, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = xf[:, :c].detach().requires_grad_(True)
    bn = nn.BatchNorm2d(c).cuda()
    gy = torch.randn(n, c, h, w, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    fwd = lambda: ops.bn_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, 0.1, 1e-5, act=ops.ACT_SWISH, group=False)
    t_f = ev(fwd)
    y = fwd()
    t_fb = ev(lambda: torch.autograd.grad(fwd(), x, gy))
    nbytes = n * c * h * w * 2
    print(f'{name:24s} fwd {t_f*1e6:8.1f} us {3*nbytes/t_f/1e9:7.0f} GB/s | bwd {(t_fb-t_f)*1e6:8.1f} us {5*nbytes/(t_fb-t_f)/1e9:7.0f} GB/s')
