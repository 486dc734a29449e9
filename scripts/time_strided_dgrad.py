"""HIP-event timing of the data gradient of the BEV decoder's strided convolutions (per-phase sub-convolutions,
ops._strided_dgrad) next to the same gradient over the zero-stuffed dy (what it replaced).
    python scripts/time_strided_dgrad.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch
from stp3_amd import ops

CL = torch.channels_last


def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, n, cin, h, w, cout, k, pad in (('decoder stem 7x7/2 64->64 @200x200', 12, 64, 200, 200, 64, 7, 3),
                                         ('layer2 3x3/2 64->128 @100x100', 12, 64, 100, 100, 128, 3, 1),
                                         ('layer2 downsample 1x1/2 64->128', 12, 64, 100, 100, 128, 1, 0),
                                         ('layer3 3x3/2 128->256 @50x50', 12, 128, 50, 50, 256, 3, 1)):
    ho, wo = (h + 2 * pad - k) // 2 + 1, (w + 2 * pad - k) // 2 + 1
    dy = torch.randn(n, cout, ho, wo, device='cuda').to(torch.bfloat16).contiguous(memory_format=CL)
    wb = (torch.randn(cout, cin, k, k, device='cuda') * 0.05).to(torch.bfloat16).contiguous(memory_format=CL)
    wt = wb.flip(2, 3).transpose(0, 1).contiguous(memory_format=CL)
    cache = {}

    def phases():
        return ops._strided_dgrad(dy, wt, (n, cin, h, w), 2, (pad, pad), cache)

    def stuffed():
        g = torch.empty((n, cout, h + 2 * pad - (k - 1), w + 2 * pad - (k - 1)), dtype=dy.dtype, device='cuda',
                        memory_format=CL).zero_()
        g[:, :, ::2, ::2][:, :, :ho, :wo] = dy
        return ops._conv2d_launch(g, wt, None, 1, (k - 1 - pad, k - 1 - pad), (1, 1), torch.bfloat16)
    a, b = phases(), stuffed()
    err = float((a.float() - b.float()).abs().max() / b.float().abs().max())
    print(f'{name:38s} per phase {ev(phases):7.1f} us | zero-stuffed {ev(stuffed):7.1f} us | max rel diff {err:.1e}')
