"""HIP-event timing of the small streaming kernels added around the BEV stages -- bilinear up-sampling, the causal
frame pairing, BatchNorm in zero-padded 40-lane rows -- next to the torch operator chains they replace, on the shapes
of bench.py's default workload.   python scripts/time_small_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch
import torch.nn as nn
import torch.nn.functional as F
from stp3_amd import ops
from stp3_amd.layers import fused

CL = torch.channels_last


def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3        # us


def rnd(*shape):
    return torch.randn(*shape, device='cuda').to(torch.bfloat16).contiguous(memory_format=CL)


print('bilinear x2 up-sampling, bf16 channels-last: kernel vs torch under autocast (float32 interpolation + casts)')
for n, c, h, w in ((12, 64, 100, 100), (12, 128, 50, 50), (12, 256, 25, 25), (72, 160, 14, 30)):
    x = rnd(n, c, h, w).requires_grad_()
    gy = rnd(n, c, 2 * h, 2 * w)
    up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)

    def torch_fwd():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            return up(x).to(torch.bfloat16)
    t_kf = ev(lambda: ops.upsample_bilinear(x, 2))
    t_kfb = ev(lambda: torch.autograd.grad(ops.upsample_bilinear(x, 2), x, gy))
    t_tf = ev(torch_fwd)
    t_tfb = ev(lambda: torch.autograd.grad(torch_fwd(), x, gy))
    byt = 5 * n * c * h * w * 2                      # read 1, write 4 pixel vectors (and the reverse)
    print(f'  {n}x{c}x{h}x{w}: fwd {t_kf:6.1f} us ({byt / t_kf / 1e3:5.0f} GB/s) vs torch {t_tf:6.1f} | '
          f'bwd {t_kfb - t_kf:6.1f} us ({byt / max(t_kfb - t_kf, 1e-3) / 1e3:5.0f} GB/s) vs torch {t_tfb - t_tf:6.1f}')

print('causal frame pairing (B*T = 12 frames of 200x200): kernel vs zero frame + two concatenations')
for c in (40, 32):
    x = rnd(12, c, 200, 200).requires_grad_()
    gy = rnd(12, 2 * c, 200, 200)

    def torch_pair():
        x5 = x.view(4, 3, c, 200, 200)
        prev = torch.cat([torch.zeros_like(x5[:, :1]), x5[:, :-1]], dim=1).view(12, c, 200, 200)
        return torch.cat([prev, x], dim=1)
    t_kf = ev(lambda: ops.causal_pair(x, 3))
    t_kfb = ev(lambda: torch.autograd.grad(ops.causal_pair(x, 3), x, gy))
    t_tf = ev(torch_pair)
    t_tfb = ev(lambda: torch.autograd.grad(torch_pair(), x, gy))
    print(f'  C={c}: fwd {t_kf:6.1f} us vs torch {t_tf:6.1f} | bwd {t_kfb - t_kf:6.1f} us vs torch {t_tfb - t_tf:6.1f}')

print('BatchNorm + ReLU of a 35-channel layer, 12x200x200: 40-lane zero-padded rows vs the 35-channel slice of them')
bn = nn.BatchNorm2d(35).cuda()
base = rnd(12, 40, 200, 200)
for tag, make_x, make_g in (('40-lane rows (cpad)', lambda: base.detach().requires_grad_(), lambda: rnd(12, 40, 200, 200)),
                            ('35-channel slice  ', lambda: base[:, :35].detach().requires_grad_(), lambda: rnd(12, 35, 200, 200))):
    x, gy = make_x(), make_g()
    t_f = ev(lambda: fused.bn_act(bn, x, fused.ACT_RELU))
    t_fb = ev(lambda: torch.autograd.grad(fused.bn_act(bn, x, fused.ACT_RELU), x, gy))
    print(f'  {tag}: fwd {t_f:6.1f} us | bwd {t_fb - t_f:6.1f} us')
