"""HIP-event timing of the MFMA convolution kernels (forward = data gradient kernel, weight gradient) on the model's
dominant shapes (B=4, T=3), TFLOP/s against the 2.5 PFLOP/s dense bf16 MFMA peak and GB/s of compulsory traffic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch
from stp3_amd import _lib
if os.environ.get('EXP_LIB'):          # an experiment build of the library
    _lib.LIB_PATH = os.environ['EXP_LIB']
from stp3_amd import ops

SHAPES = [
    # name, N, Cin, H, W, Cout, K, stride, pad, dil
    ('temporal head 3x3 128->128 @200x200', 12, 128, 200, 200, 128, 3, 1, 1, 1),
    ('temporal ASPP 3x3 d12 64->128 @200x200', 12, 64, 200, 200, 128, 3, 1, 12, 12),
    ('temporal ASPP project 1x1 512->128', 12, 512, 200, 200, 128, 1, 1, 0, 1),
    ('decoder stem 7x7/2 64->64 @200x200', 12, 64, 200, 200, 64, 7, 2, 3, 1),
    ('decoder head 3x3 64->64 @200x200', 12, 64, 200, 200, 64, 3, 1, 1, 1),
    ('temporal block 1x1 72->40 @200x200', 12, 72, 200, 200, 40, 1, 1, 0, 1),
    ('temporal path 3x3 40->40 @200x200', 12, 40, 200, 200, 40, 3, 1, 1, 1),
    ('decoder layer1 3x3 64->64 @100x100', 12, 64, 100, 100, 64, 3, 1, 1, 1),
    ('decoder layer2 3x3 128->128 @50x50', 12, 128, 50, 50, 128, 3, 1, 1, 1),
    ('encoder upconcat 3x3 216->64 @28x60', 72, 216, 28, 60, 64, 3, 1, 1, 1),
    ('trunk stem 3x3/2 8->48 @224x480', 72, 8, 225, 481, 48, 3, 2, 0, 1),
    ('decoder heads merged 3x3 64->320 @200x200', 12, 64, 200, 200, 320, 3, 1, 1, 1),
    ('decoder heads merged dgrad 3x3 320->64 @200x200', 12, 320, 200, 200, 64, 3, 1, 1, 1),
    ('temporal ASPP merged? 3x3 64->192', 12, 64, 200, 200, 192, 3, 1, 1, 1),
    ('trunk expand 1x1 24->144 @112x240', 72, 24, 112, 240, 144, 1, 1, 0, 1),
    ('trunk project 1x1 144->32 @56x120', 72, 144, 56, 120, 32, 1, 1, 0, 1),
    ('trunk expand 1x1 32->192 @56x120', 72, 32, 56, 120, 192, 1, 1, 0, 1),
    ('trunk expand 1x1 56->336 @28x60', 72, 56, 28, 60, 336, 1, 1, 0, 1),
    ('trunk expand 1x1 160->960 @14x30', 72, 160, 14, 30, 960, 1, 1, 0, 1),
    ('trunk project 1x1 960->160 @14x30', 72, 960, 14, 30, 160, 1, 1, 0, 1),
]


def ev(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


only = sys.argv[1] if len(sys.argv) > 1 else ''
for name, n, cin, h, w, cout, k, st, pad, dil in SHAPES:
    if only and only not in name:
        continue
    x = torch.randn(n, cin, h, w, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wgt = (torch.randn(cout, cin, k, k, device='cuda') * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = ops._conv2d_launch(x, wgt, None, st, (pad, pad), (dil, dil), torch.bfloat16)
    flops = 2.0 * y.numel() * cin * k * k
    t_f = ev(lambda: ops._conv2d_launch(x, wgt, None, st, (pad, pad), (dil, dil), torch.bfloat16))
    sums = torch.empty(2, cout, device='cuda')
    t_s = ev(lambda: ops._conv2d_launch(x, wgt, None, st, (pad, pad), (dil, dil), torch.bfloat16, sums_ptr=sums.data_ptr()))
    dy = torch.randn_like(y)
    t_w = ev(lambda: ops._conv2d_wgrad(dy, x, (cout, cin, k, k), st, (pad, pad), (dil, dil)))
    byts = (x.numel() + y.numel() + wgt.numel()) * 2
    print(f'{name:42s} fwd {t_f*1e3:8.1f} us {flops/t_f/1e9:7.1f} TF/s ({flops/t_f/1e9/2500*100:4.1f}% of peak) {byts/t_f/1e6:7.1f} GB/s'
          f' | +BN stats {t_s*1e3:8.1f} us | wgrad {t_w*1e3:8.1f} us {flops/t_w/1e9:7.1f} TF/s')
