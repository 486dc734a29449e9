"""HIP-event timing of the short-contraction 1x1 layers (trunk expand convolutions at B = 4, T = 3: 72 images; BEV layers)
on the streaming kernels (pointwise_rows_kernel / pointwise_direct_kernel, stp3_conv.hip) in the five epilogue modes; GB/s
of the mode's own algorithmic traffic beside each time.  (The same table on the tiled kernel, from the round in which the
library still had a switch between the two: profiles/r04p_time_pointwise.txt.)

    python scripts/time_pointwise.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))

SHAPES = [  # N, Cin, H, W, Cout
    (72, 24, 112, 240, 144), (72, 32, 56, 120, 192), (72, 56, 28, 60, 336), (72, 112, 14, 30, 672),
    (72, 32, 112, 240, 24 * 6), (12, 64, 200, 200, 64), (12, 128, 200, 200, 512), (12, 64, 200, 200, 128),
]


def child():
    import torch
    from stp3_amd import _lib, ops
    from stp3_amd.ops import check
    lib = _lib.lib()
    stream = torch.cuda.current_stream().cuda_stream

    def ev(fn, iters=20, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3

    for n, cin, h, w, cout in SHAPES:
        x = torch.randn(n, cin, h, w, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wb = (torch.randn(cout, cin, 1, 1, device='cuda') * 0.2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dz = torch.randn(n, cout, h, w, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = torch.empty_like(dz)
        dims = _lib.ConvDims(n, h, w, cin, h, w, cout, 1, 1, 1, 0, 0, 1, 1, cin, cout, _lib.DTYPE_BF16, 0)
        need = ctypes.c_size_t()
        check(lib.stp3_conv2d_fwd_workspace(ctypes.byref(dims), ctypes.byref(need)), 'ws')
        ws = torch.empty(need.value, dtype=torch.uint8, device='cuda')
        stat = torch.zeros(2, cout, device='cuda')
        coef = torch.rand(4, cout, device='cuda') + 0.5
        gs = torch.randn(2, cout, device='cuda')
        M = n * h * w
        xin, out = M * cin * 2, M * cout * 2
        runs = {
            'plain': (lambda: check(lib.stp3_conv2d_fwd(ctypes.byref(dims), x.data_ptr(), wb.data_ptr(), None, y.data_ptr(), None, None, 0, stream), 'f'), xin + out),
            'plain+stats': (lambda: check(lib.stp3_conv2d_fwd(ctypes.byref(dims), x.data_ptr(), wb.data_ptr(), None, y.data_ptr(), stat.data_ptr(), ws.data_ptr(), need.value, stream), 'f'), xin + out),
            'stats': (lambda: check(lib.stp3_conv2d_fwd_stats(ctypes.byref(dims), x.data_ptr(), wb.data_ptr(), stat.data_ptr(), ws.data_ptr(), need.value, stream), 's'), xin),
            'bnact': (lambda: check(lib.stp3_conv2d_fwd_bnact(ctypes.byref(dims), x.data_ptr(), wb.data_ptr(), coef.data_ptr(), 2, y.data_ptr(), stream), 'b'), xin + out),
            'bwd_reduce': (lambda: check(lib.stp3_conv2d_bn_bwd_reduce(ctypes.byref(dims), x.data_ptr(), wb.data_ptr(), dz.data_ptr(), cout, coef.data_ptr(), 2, stat.data_ptr(), ws.data_ptr(), need.value, stream), 'r'), xin + out),
            'bwd_apply': (lambda: check(lib.stp3_conv2d_bn_bwd_apply(ctypes.byref(dims), x.data_ptr(), wb.data_ptr(), dz.data_ptr(), cout, coef.data_ptr(), 2, gs.data_ptr(), float(M), y.data_ptr(), stream), 'a'), xin + 2 * out),
        }
        line = f'{cin:4d}->{cout:4d} @{h}x{w}x{n} ({out / 2**20:6.1f} MiB out):'
        for name, (fn, byts) in runs.items():
            t = ev(fn)
            line += f'  {name} {t:6.1f} us {byts / t / 1e3:5.0f} GB/s'
        print(line, flush=True)


if __name__ == '__main__':
    child()
